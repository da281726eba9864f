"""Kernel-level numerics on the GPU through the C ABI's test hooks: each HIP kernel against a
plain PyTorch float32 reference of the same op (bf16 inputs are exact in float32, so the only
differences are accumulation order and the final rounding)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from rho_tts_amd import _native
    c = _native.Context(0)
    yield c
    c.close()


def gemm(ctx, a, w, taps=1, tap_stride=1, tap_offset=0, rows_out=0, rows_in=0, bias=None, act=0, mode=0, split_k=1, M=None,
         split_prec=False):
    cin = w.shape[1] // taps
    M = M if M is not None else a.numel() // cin
    out = torch.empty(M, w.shape[0], device="cuda")
    torch.cuda.synchronize()
    rc = ctx.lib.rt_debug_gemm(ctx.handle, a.data_ptr(), (2 if split_prec else 1) if a.dtype == torch.float32 else 0, M, cin, taps, tap_stride, tap_offset,
                               rows_out, rows_in, w.data_ptr(), w.shape[0], bias.data_ptr() if bias is not None else None, act,
                               out.data_ptr(), mode, split_k)
    ctx.check(rc, "rt_debug_gemm")
    return out


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale)


@pytest.mark.parametrize("M,N,K", [(32, 256, 512), (1, 96, 96), (8, 3072, 2048), (33, 64, 128), (64, 160, 1024)])
@pytest.mark.parametrize("split", [1, 2, 0])
def test_skinny_gemm_matches_fp32(ctx, M, N, K, split):
    a = rnd(M, K, seed=1).to(torch.bfloat16).cuda()
    w = rnd(N, K, scale=0.05, seed=2).to(torch.bfloat16).cuda()
    if split and (K // 16) % split:
        pytest.skip("split does not divide")
    out = gemm(ctx, a, w, mode=1, split_k=split)
    ref = a.float() @ w.float().T
    assert float((out - ref).abs().max()) < 2e-3 * max(1.0, float(ref.abs().max()))
    # integer-valued operands: products and sums are exact in f32 -> bit-exact, catches any fragment-layout error
    ai = torch.randint(-4, 5, (M, K), generator=torch.Generator().manual_seed(3)).to(torch.bfloat16).cuda()
    wi = torch.randint(-4, 5, (N, K), generator=torch.Generator().manual_seed(4)).to(torch.bfloat16).cuda()
    assert torch.equal(gemm(ctx, ai, wi, mode=1, split_k=split), ai.float() @ wi.float().T)


@pytest.mark.parametrize("M,N,K", [(460, 2048, 2048), (416, 4096, 2048), (460, 2048, 6144), (65, 96, 192), (129, 64, 256), (1000, 160, 320),
                                   (333, 12288, 2048)])
def test_prefill_gemm_matches_fp32(ctx, M, N, K):
    """k_gemm_mid (prompt prefill, 65..1024 rows, whole K per 64 x 64 tile, final sums): the prefix / suffix prefill shapes of the
    1.7B preset, row and column tails, the shortest K; integer-valued operands bit-exact."""
    a = rnd(M, K, seed=1).to(torch.bfloat16).cuda()
    w = rnd(N, K, scale=0.05, seed=2).to(torch.bfloat16).cuda()
    ai = torch.randint(-4, 5, (M, K), generator=torch.Generator().manual_seed(3)).to(torch.bfloat16).cuda()
    wi = torch.randint(-4, 5, (N, K), generator=torch.Generator().manual_seed(4)).to(torch.bfloat16).cuda()
    ref = a.float() @ w.float().T
    try:
        for code in (1902, 1903, 1901):              # 64 x 64 tiles, 128 x 64 tiles, the automatic choice
            ctx.lib.rt_debug_tune(code, 0)
            out = gemm(ctx, a, w, mode=2)
            assert float((out - ref).abs().max()) < 2e-3 * max(1.0, float(ref.abs().max())), code
            assert torch.equal(gemm(ctx, ai, wi, mode=2), ai.float() @ wi.float().T), code
    finally:
        ctx.lib.rt_debug_tune(1901, 0)


@pytest.mark.parametrize("N,K", [(2048, 2048), (4096, 2048), (12288, 2048), (2048, 6144), (3072, 2048)])
def test_prefill_gemm_rows_do_not_depend_on_the_batch(ctx, N, K):
    """A prompt row must get the same float32 sums whether it is prefilled among 460 rows (k_gemm_mid) or among a dozen (the
    skinny kernel + its slab sum): the mid kernel adds K in the skinny kernel's segments and association.  Random operands,
    bit for bit, on the 1.7B talker's four projection shapes and its head."""
    a = rnd(460, K, seed=5).to(torch.bfloat16).cuda()
    w = rnd(N, K, scale=0.05, seed=6).to(torch.bfloat16).cuda()
    big = {}
    try:
        for code in (1902, 1903):
            ctx.lib.rt_debug_tune(code, 0)
            big[code] = gemm(ctx, a, w, mode=2)
    finally:
        ctx.lib.rt_debug_tune(1901, 0)
    assert torch.equal(big[1902], big[1903])                           # 64- and 128-row tiles: same sums
    for r0, m in ((0, 13), (100, 33), (396, 64)):
        small = gemm(ctx, a[r0:r0 + m].contiguous(), w, mode=1, split_k=0)   # the skinny kernel with its production split
        assert torch.equal(small, big[1902][r0:r0 + m]), (r0, m)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 200, 96), (1000, 96, 672), (5, 40, 16), (257, 384, 2048)])
@pytest.mark.parametrize("f32", [False, True])
def test_tiled_gemm_matches_fp32(ctx, M, N, K, f32):
    a = rnd(M, K, seed=5).to(torch.bfloat16)
    w = rnd(N, K, scale=0.05, seed=6).to(torch.bfloat16).cuda()
    a_dev = a.float().cuda() if f32 else a.cuda()
    bias = rnd(N, seed=7).cuda()
    for split in (1, 3):
        out = gemm(ctx, a_dev, w, bias=bias, mode=0, split_k=split)
        ref = a.float().cuda() @ w.float().T + bias
        assert float((out - ref).abs().max()) < 2e-3 * max(1.0, float(ref.abs().max()))
    ai = torch.randint(-4, 5, (M, K), generator=torch.Generator().manual_seed(8)).to(torch.bfloat16)
    wi = torch.randint(-4, 5, (N, K), generator=torch.Generator().manual_seed(9)).to(torch.bfloat16).cuda()
    a_dev = ai.float().cuda() if f32 else ai.cuda()
    assert torch.equal(gemm(ctx, a_dev, wi, mode=0), ai.float().cuda() @ wi.float().T)


def test_split_precision_gemm_is_f32_faithful(ctx):
    """hi + lo bf16 planes of an f32 activation: error vs the float64 product drops ~100x below plain bf16 feeding."""
    a = rnd(300, 672, seed=41).cuda()                                     # full-precision f32 activations
    w = rnd(96, 672, scale=0.05, seed=42).to(torch.bfloat16).cuda()
    ref = (a.double() @ w.double().T)
    e_plain = float((gemm(ctx, a, w).double() - ref).abs().max())
    e_split = float((gemm(ctx, a, w, split_prec=True).double() - ref).abs().max())
    scale = float(ref.abs().max())
    assert e_split < 2e-5 * scale and e_split * 30 < e_plain, (e_split, e_plain, scale)
    # ... also with K split over workgroups (the codec pre-transformer's float32-faithful projections at few rows)
    e_split_k = float((gemm(ctx, a, w, split_prec=True, split_k=3).double() - ref).abs().max())
    assert e_split_k < 2e-5 * scale, (e_split_k, scale)
    # causal dilated conv in split mode
    x = rnd(2, 100, 32, seed=43)
    wc = rnd(48, 32, 7, scale=0.1, seed=44).to(torch.bfloat16)
    wm = wc.permute(0, 2, 1).reshape(48, 7 * 32).contiguous().cuda()
    out = gemm(ctx, x.cuda(), wm, taps=7, tap_stride=3, tap_offset=-18, rows_out=100, rows_in=100, M=200, split_prec=True)
    refc = torch.nn.functional.conv1d(torch.nn.functional.pad(x.double().transpose(1, 2), (18, 0)), wc.double(), dilation=3)
    assert float((out.view(2, 100, 48).cpu().double() - refc.transpose(1, 2)).abs().max()) < 2e-5 * float(refc.abs().max())


@pytest.mark.parametrize("act", [0, 1, 2])
def test_tiled_gemm_epilogue_activations(ctx, act):
    a = rnd(70, 64, seed=1).to(torch.bfloat16).cuda()
    w = rnd(48, 64, scale=0.2, seed=2).to(torch.bfloat16).cuda()
    bias = rnd(48, seed=3).cuda()
    ref = a.float() @ w.float().T + bias
    ref = [ref, torch.nn.functional.silu(ref), torch.nn.functional.gelu(ref)][act]
    assert float((gemm(ctx, a, w, bias=bias, act=act) - ref).abs().max()) < 3e-3


@pytest.mark.parametrize("C_in,C_out,dil,T,B", [(16, 16, 1, 50, 2), (96, 96, 3, 333, 3), (32, 64, 9, 40, 1), (192, 192, 9, 700, 2)])
def test_causal_dilated_conv_as_gemm(ctx, C_in, C_out, dil, T, B):
    """Implicit-GEMM causal conv1d (k = 7) on channels-last activations == F.conv1d with left padding."""
    x = rnd(B, T, C_in, seed=11).to(torch.bfloat16)
    w = rnd(C_out, C_in, 7, scale=0.1, seed=12).to(torch.bfloat16)
    wm = w.permute(0, 2, 1).reshape(C_out, 7 * C_in).contiguous().cuda()
    out = gemm(ctx, x.cuda(), wm, taps=7, tap_stride=dil, tap_offset=-6 * dil, rows_out=T, rows_in=T, M=B * T)
    ref = torch.nn.functional.conv1d(torch.nn.functional.pad(x.float().transpose(1, 2), (6 * dil, 0)), w.float(), dilation=dil)
    assert float((out.view(B, T, C_out).cpu() - ref.transpose(1, 2)).abs().max()) < 3e-3


@pytest.mark.parametrize("C_in,C_out,dil,T,B", [(96, 96, 3, 333, 3), (32, 64, 9, 40, 1), (192, 192, 9, 700, 2), (64, 128, 1, 129, 2),
                                               (96, 96, 9, 128, 4), (384, 384, 3, 77, 5)])
def test_causal_dilated_conv_on_operand_planes(ctx, C_in, C_out, dil, T, B):
    """The codec decoder's k = 7 convs read hi / lo bf16 operand planes and keep their input window in LDS (k_conv_win; tile
    tails, item starts inside a tile, windows that reach into the previous item): == F.conv1d on the f32 input to split
    precision.  rt_debug_tune(1200) runs the same planes through the per-tap kernel."""
    x = rnd(B, T, C_in, seed=11)
    w = rnd(C_out, C_in, 7, scale=0.1, seed=12).to(torch.bfloat16)
    wm = w.permute(0, 2, 1).reshape(C_out, 7 * C_in).contiguous().cuda()
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    planes = torch.cat([hi.reshape(-1), lo.reshape(-1)]).cuda()
    ref = torch.nn.functional.conv1d(torch.nn.functional.pad(x.transpose(1, 2), (6 * dil, 0)), w.float(), dilation=dil).transpose(1, 2)
    outs = []
    try:
        for codes in ((1201, 1800), (1200, 1800), (1201, 1802)):      # window kernel, per-tap kernel, window kernel with 256-row tiles (forced)
            for code in codes:
                ctx.lib.rt_debug_tune(code, 0)
            out = torch.empty(B * T, C_out, device="cuda")
            torch.cuda.synchronize()
            ctx.check(ctx.lib.rt_debug_gemm(ctx.handle, planes.data_ptr(), 3, B * T, C_in, 7, dil, -6 * dil, T, T, wm.data_ptr(), C_out, None, 0,
                                            out.data_ptr(), 0, 1), "rt_debug_gemm")
            outs.append(out.view(B, T, C_out).cpu())
    finally:
        ctx.lib.rt_debug_tune(1201, 0)
        ctx.lib.rt_debug_tune(1801, 0)
    for out in outs:
        assert float((out - ref).abs().max()) < 2e-4 * max(1.0, float(ref.abs().max()))
    assert float((outs[0] - outs[1]).abs().max()) < 1e-4 * max(1.0, float(ref.abs().max()))
    assert torch.equal(outs[0], outs[2])                                # same products, same order: the tile height changes nothing


@pytest.mark.parametrize("C_in,C_out,r,T,B", [(32, 16, 3, 20, 2), (64, 32, 8, 45, 3), (96, 48, 2, 7, 1)])
def test_transposed_conv_as_gemm(ctx, C_in, C_out, r, T, B):
    """ConvTranspose1d(k = 2r, stride r) trimmed r on both sides == 2-tap GEMM over (x[m], x[m+1])."""
    x = rnd(B, T, C_in, seed=13).to(torch.bfloat16)
    w = rnd(C_in, C_out, 2 * r, scale=0.1, seed=14).to(torch.bfloat16)
    tap0 = w[:, :, r:].permute(2, 1, 0)
    tap1 = w[:, :, :r].permute(2, 1, 0)
    wm = torch.cat([tap0, tap1], dim=2).reshape(r * C_out, 2 * C_in).contiguous().cuda()
    out = gemm(ctx, x.cuda(), wm, taps=2, tap_stride=1, tap_offset=0, rows_out=T - 1, rows_in=T, M=B * (T - 1))
    y = torch.nn.functional.conv_transpose1d(x.float().transpose(1, 2), w.float(), stride=r)
    ref = y[..., r: y.shape[-1] - r].transpose(1, 2)                      # [B, (T-1) r, C_out]
    assert float((out.view(B, (T - 1) * r, C_out).cpu() - ref).abs().max()) < 3e-3


@pytest.mark.parametrize("d,heads,kvh", [(128, 16, 8), (64, 4, 4), (32, 4, 1), (128, 4, 2)])
@pytest.mark.parametrize("window", [0, 9])
def test_attention_matches_fp32(ctx, d, heads, kvh, window):
    slots, max_pos, M = 3, 70, 11
    g = torch.Generator().manual_seed(21)
    k = (torch.randn(slots, kvh, max_pos, d, generator=g)).to(torch.bfloat16).cuda()
    v = (torch.randn(slots, kvh, max_pos, d, generator=g)).to(torch.bfloat16).cuda()
    q = torch.randn(M, heads, d, generator=g).cuda()
    slot = torch.randint(0, slots, (M,), generator=g).to(torch.int32).cuda()
    pos = torch.randint(0, max_pos, (M,), generator=g).to(torch.int32)
    pos[0], pos[1] = 0, max_pos - 1
    pos = pos.cuda()
    out = torch.empty(M, heads * d, dtype=torch.bfloat16, device="cuda")
    torch.cuda.synchronize()
    ctx.check(ctx.lib.rt_debug_attention(ctx.handle, q.data_ptr(), M, heads, kvh, d, slot.data_ptr(), pos.data_ptr(), window,
                                         k.data_ptr(), v.data_ptr(), slots, max_pos, out.data_ptr()), "rt_debug_attention")
    rep = heads // kvh
    for r in range(M):
        hi = int(pos[r])
        lo = max(0, hi - window + 1) if window else 0
        K = k[int(slot[r]), :, lo:hi + 1].float().repeat_interleave(rep, 0)
        V = v[int(slot[r]), :, lo:hi + 1].float().repeat_interleave(rep, 0)
        s = torch.einsum("hd,htd->ht", q[r], K) * d ** -0.5
        ref = torch.einsum("ht,htd->hd", torch.softmax(s, -1), V).reshape(-1)
        assert float((out[r].float() - ref).abs().max()) < 2e-2 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("V,sup_from", [(3072, 2048), (2048, 2048), (2048, 1500), (1000, 640), (4096, 2730), (700, 690), (6000, 5000)])
def test_sampler_matches_oracle_draw_for_draw(ctx, V, sup_from):
    """Every sampler instantiation (4 / 8 / 16 waves for V <= 1024 / 2048 / 4096, the LDS kernel above) against
    oracle/sampling.py, draw for draw.  Rows built to hit the selection's corner cases: ties at the top, all values equal,
    logits quantised to 1/4 (ties AT the top-k cut, to be resolved by lowest index), fewer finite values than k (the rest
    suppressed), one finite value, and a row whose top values all sit in the last wave."""
    from oracle.sampling import SamplingParams, draw, uniform
    from rho_tts_amd._native_model import RtSampling
    M = 24
    allow = min(V - 1, sup_from + 102)
    g = torch.Generator().manual_seed(31 + V)
    logits = (torch.randn(M, V, generator=g) * 2.0)
    logits[3, 100] = logits[3, 200] = float(logits[3].max()) + 1.0          # a tie at the top
    logits[4, :] = 0.5                                                       # all equal
    logits[5] = torch.round(logits[5] * 4) / 4                               # many ties, some across the top-k cut
    logits[6] = torch.round(logits[6])
    logits[7, : sup_from - 7] = float("-inf")                                # 7 finite values (+ the allowed token): fewer than k
    logits[8, :] = float("-inf"); logits[8, 17] = 0.25                       # one finite value
    logits[9, sup_from - 40: sup_from] += 9.0                                # the whole top-k in the last waves
    logits[10, :64] += 9.0; logits[10, :64] = torch.round(logits[10, :64])   # the whole top-k in ONE wave, with ties
    seen_h = torch.rand(M, V, generator=g) < 0.05
    cases = [SamplingParams(False), SamplingParams(True, 0.9, 50, 1.0, 1.0), SamplingParams(True, 1.3, 64, 0.8, 1.05),
             SamplingParams(True, 0.7, 1, 1.0, 1.0), SamplingParams(True, 1.0, 5, 0.3, 1.2)]
    for ci, sp in enumerate(cases):
        seen = seen_h.to(torch.uint8).cuda()
        out = torch.empty(M, dtype=torch.int32, device="cuda")
        rs = RtSampling(int(sp.do_sample), sp.temperature, sp.top_k, sp.top_p, sp.repetition_penalty)
        lg = logits.cuda()
        torch.cuda.synchronize()
        ctx.check(ctx.lib.rt_debug_sample(ctx.handle, lg.data_ptr(), M, V, C.byref(rs), 789 + (5 << 32), 7, 3, sup_from, allow,
                                          seen.data_ptr(), out.data_ptr()), "rt_debug_sample")
        sup = np.zeros(V, bool)
        sup[sup_from:] = True
        sup[allow] = False
        want = [draw(logits[r].numpy(), sp, uniform(789 + (5 << 32), r, 7, 3), sup, seen_h[r].numpy()) for r in range(M)]
        got = out.cpu().tolist()
        # a draw can legitimately differ only when u falls within float rounding of a CDF boundary: allow at most one
        assert sum(int(a != b) for a, b in zip(got, want)) <= (1 if sp.do_sample else 0), (ci, got, want)
        new_seen = seen.cpu().bool()
        for r in range(M):
            assert new_seen[r, got[r]]


def _prefill_attention(ctx, q, slot, pos, k, v, prefix_slot, Lp, mode):
    M, heads, d = q.shape
    out = torch.empty(M, heads * d, dtype=torch.bfloat16, device="cuda")
    torch.cuda.synchronize()
    ctx.check(ctx.lib.rt_debug_attention_prefill(ctx.handle, q.data_ptr(), M, heads, k.shape[1], d, slot.data_ptr(), pos.data_ptr(), k.data_ptr(), v.data_ptr(),
                                                 k.shape[0], k.shape[2], prefix_slot, Lp, mode, out.data_ptr()), "rt_debug_attention_prefill")
    return out


@pytest.mark.parametrize("Lp,M", [(460, 43), (64, 9), (97, 1), (460, 416)])
def test_prefill_attention_behind_a_shared_prefix_on_the_matrix_cores(ctx, Lp, M):
    """The prompt rows' attention (texts' suffixes behind the voice prefix: 1.7B shape, 16 / 8 heads x 128) in its matrix-core form:
    against float32, against the vector-unit kernel, an EXACT key census (q = 0 -> uniform softmax, V[p] = indicator of p mod d, so
    output dim j is (number of visible positions congruent to j) / (pos + 1): a dropped or doubled key changes a count), and
    batch invariance - every row alone gives the bits it gives among the others, whatever 8-row block it falls into."""
    d, heads, kvh, slots, max_pos = 128, 16, 8, 5, 560
    g = torch.Generator().manual_seed(33 + Lp + M)
    k = (torch.randn(slots, kvh, max_pos, d, generator=g)).to(torch.bfloat16).cuda()
    v = (torch.randn(slots, kvh, max_pos, d, generator=g)).to(torch.bfloat16).cuda()
    q = torch.randn(M, heads, d, generator=g).cuda()
    prefix_slot = slots - 1
    slot = torch.randint(0, slots - 1, (M,), generator=g).to(torch.int32)
    pos = (Lp + torch.randint(0, 60, (M,), generator=g)).to(torch.int32)
    pos[0] = Lp                                                       # the first own row
    if M > 2:
        pos[1], pos[2] = Lp + 59, Lp + 31
    slot_c, pos_c = slot.cuda(), pos.cuda()
    out = _prefill_attention(ctx, q, slot_c, pos_c, k, v, prefix_slot, Lp, 1)
    vec = _prefill_attention(ctx, q, slot_c, pos_c, k, v, prefix_slot, Lp, 0)
    rep = heads // kvh
    for r in range(min(M, 48)):
        hi, sl = int(pos[r]), int(slot[r])
        K = torch.cat([k[prefix_slot, :, :Lp], k[sl, :, Lp:hi + 1]], 1).float().repeat_interleave(rep, 0)
        V = torch.cat([v[prefix_slot, :, :Lp], v[sl, :, Lp:hi + 1]], 1).float().repeat_interleave(rep, 0)
        s = torch.einsum("hd,htd->ht", q[r], K) * d ** -0.5
        ref = torch.einsum("ht,htd->hd", torch.softmax(s, -1), V).reshape(-1)
        assert float((out[r].float() - ref).abs().max()) < 2e-2 * max(1.0, float(ref.abs().max())), r
    assert float((out.float() - vec.float()).abs().max()) < 3e-2
    # batch invariance: rows alone, and a re-ordered batch
    for r in (0, M // 2, M - 1):
        alone = _prefill_attention(ctx, q[r:r + 1].contiguous(), slot_c[r:r + 1].contiguous(), pos_c[r:r + 1].contiguous(), k, v, prefix_slot, Lp, 1)
        assert torch.equal(alone[0], out[r]), r
    if M > 3:
        perm = torch.randperm(M, generator=g)
        shuffled = _prefill_attention(ctx, q[perm.cuda()].contiguous(), slot_c[perm.cuda()].contiguous(), pos_c[perm.cuda()].contiguous(), k, v, prefix_slot, Lp, 1)
        assert torch.equal(shuffled, out[perm.cuda()])
    # key census
    ind = torch.zeros(max_pos, d)
    ind[torch.arange(max_pos), torch.arange(max_pos) % d] = 1.0
    v1 = ind.to(torch.bfloat16)[None, None].repeat(slots, kvh, 1, 1).cuda().contiguous()
    cen = _prefill_attention(ctx, torch.zeros_like(q), slot_c, pos_c, k, v1, prefix_slot, Lp, 1).float().cpu().view(M, heads, d)
    for r in range(min(M, 48)):
        hi = int(pos[r])
        want = torch.bincount(torch.arange(hi + 1) % d, minlength=d).float() / (hi + 1)
        got = cen[r]
        assert float((got - want.to(torch.bfloat16).float()[None]).abs().max()) <= 2 ** -9 * float(want.max()), r


@pytest.mark.parametrize("M", [460, 64, 131])
def test_prefix_self_prefill_attention_on_the_matrix_cores(ctx, M):
    """rt_model_set_voice's own attention (the voice prefix attending to itself: M consecutive rows of one slot, causal): the keys in
    front of each 8-row block on the matrix cores, the block's own <= 8 keys on the vector unit - against float32 and with the exact
    key census (row p sees exactly positions 0 .. p)."""
    d, heads, kvh, slots, max_pos = 128, 16, 8, 2, 512
    g = torch.Generator().manual_seed(90 + M)
    k = (torch.randn(slots, kvh, max_pos, d, generator=g)).to(torch.bfloat16).cuda()
    v = (torch.randn(slots, kvh, max_pos, d, generator=g)).to(torch.bfloat16).cuda()
    q = torch.randn(M, heads, d, generator=g).cuda()
    slot = torch.ones(M, dtype=torch.int32).cuda()
    pos = torch.arange(M, dtype=torch.int32).cuda()
    out = _prefill_attention(ctx, q, slot, pos, k, v, 1, M, 2)
    rep = heads // kvh
    for r in list(range(0, M, 37)) + [7, 8, 31, 32, M - 1]:
        K = k[1, :, :r + 1].float().repeat_interleave(rep, 0)
        V = v[1, :, :r + 1].float().repeat_interleave(rep, 0)
        s = torch.einsum("hd,htd->ht", q[r], K) * d ** -0.5
        ref = torch.einsum("ht,htd->hd", torch.softmax(s, -1), V).reshape(-1)
        assert float((out[r].float() - ref).abs().max()) < 2e-2 * max(1.0, float(ref.abs().max())), r
    ind = torch.zeros(max_pos, d)
    ind[torch.arange(max_pos), torch.arange(max_pos) % d] = 1.0
    v1 = ind.to(torch.bfloat16)[None, None].repeat(slots, kvh, 1, 1).cuda().contiguous()
    cen = _prefill_attention(ctx, torch.zeros_like(q), slot, pos, k, v1, 1, M, 2).float().cpu().view(M, heads, d)
    for r in range(M):
        want = torch.bincount(torch.arange(r + 1) % d, minlength=d).float() / (r + 1)
        assert float((cen[r] - want.to(torch.bfloat16).float()[None]).abs().max()) <= 2 ** -9 * float(want.max()), r
