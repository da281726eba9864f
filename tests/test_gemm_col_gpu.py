"""The dominant decode kernel (gemm_col.hip k_gemm_col) and the fused decode attention on their own, through the C ABI's
test hooks, against plain PyTorch float32 references.

k_gemm_col is checked at the 12 decode GEMM shapes of the 1.7B / 0.6B presets (tools/bench_gemm_col.py), for its three
epilogues (STORE with bias and RMSNorm row scale; RESID with layer scale, sums of squares and the next operand; SILU),
at M in {1, 8, 17, 32, 33, 64}, sub-tile splits {1, 2, 4}, both weight-load policies and a non-zero row offset.
Integer-valued operands make every product and partial sum exact in f32, so the GEMM part is compared BIT FOR BIT (any
fragment-layout, K-split or tail error shows up as a wrong integer); random operands are compared to a stated tolerance.
"""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

STORE, RESID, SILU = 0, 1, 2

# (N, K, epi, norm, name): every decode projection of the 1.7B preset (talker H 2048 / I 6144, predictor H 1024 / I 3072)
SHAPES = [
    (4096, 2048, STORE, True, "talker qkv"), (2048, 2048, RESID, False, "talker o"), (12288, 2048, SILU, True, "talker gate/up"),
    (2048, 6144, RESID, False, "talker down"), (3072, 2048, STORE, True, "talker head"), (4096, 1024, STORE, True, "pred qkv / 0.6b qkv"),
    (1024, 2048, RESID, False, "pred o"), (6144, 1024, SILU, True, "pred gate/up / 0.6b gate/up"), (1024, 3072, RESID, False, "pred down / 0.6b down"),
    (2048, 1024, STORE, True, "pred head"), (1024, 2048, STORE, True, "mtp projection (bias)"), (3072, 1024, STORE, True, "0.6b head"),
]


@pytest.fixture(scope="module")
def ctx():
    from rho_tts_amd import _native
    c = _native.Context(0)
    yield c
    c.close()


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def run_col(ctx, a, w, epi, split=0, row_off=0, nt=1, rowsq=None, eps=1e-6, bias=None, scale=None, x=None, next_w=None):
    """Returns dict(out / x, next, rowsq_out, act) as the epilogue produces them."""
    M, K = a.shape
    N = w.shape[0]
    dev = a.device
    res = {}
    n_part = (N + 15) // 16 * (split if split > 0 else 4)          # upper bound; the hook writes ceil(N/16)*split_used columns
    xo = nxt = rq = act = None
    if epi == STORE:
        xo = torch.full((M, N), float("nan"), device=dev)
    elif epi == RESID:
        xo = x.clone()
        nxt = torch.zeros(M, N, dtype=torch.bfloat16, device=dev) if next_w is not None else None
        rq = torch.zeros(M * n_part, device=dev)
    else:
        act = torch.zeros(M, N // 2, dtype=torch.bfloat16, device=dev)
    torch.cuda.synchronize()
    rc = ctx.lib.rt_debug_gemm_col(ctx.handle, ptr(a), M, K, ptr(w), N, epi, split, row_off, nt, ptr(rowsq),
                                   0 if rowsq is None else rowsq.shape[1], eps, ptr(bias), ptr(scale), ptr(xo), ptr(next_w), ptr(nxt), ptr(rq), ptr(act))
    ctx.check(rc, "rt_debug_gemm_col")
    torch.cuda.synchronize()
    res.update(out=xo, next=nxt, rowsq_out=rq, act=act)
    return res


def used_split(ctx, N, split):
    if split > 0:
        return split
    info = ctx.device_info()
    return 2 if ((N + 15) // 16) * 2 <= info["n_cu"] else 1


def ints(shape, lo, hi, seed):
    return torch.randint(lo, hi + 1, shape, generator=torch.Generator().manual_seed(seed)).to(torch.bfloat16)


@pytest.mark.parametrize("N,K,epi,norm,name", SHAPES, ids=[s[4] for s in SHAPES])
@pytest.mark.parametrize("M", [1, 8, 17, 32, 33, 64])
def test_gemm_col_integer_exact(ctx, N, K, epi, norm, name, M):
    """Integer operands in [-3, 3]: A W^T is exact in f32 whatever the summation order, so STORE / RESID outputs (and the
    integers under the SILU) must come out bit for bit; the row scale is a power of two so it stays exact too."""
    a = ints((M, K), -3, 3, 1 + M).cuda()
    w = ints((N, K), -3, 3, 2).cuda()
    ref = a.float() @ w.float().T
    for split in ([1, 2, 4] if M in (17, 32, 64) else [0]):
        for row_off, nt in ((0, 1), (32, 0)) if M <= 32 and split in (0, 2) else ((0, 1),):
            if epi == STORE:
                bias = ints((N,), -8, 8, 3).float().cuda()
                got = run_col(ctx, a, w, STORE, split, row_off, nt, None, 0.0, bias)["out"]
                assert torch.equal(got, ref + bias), (name, M, split, row_off)
                if norm:            # RMSNorm row scale from 4 partials: rsqrt(4K / K) = 0.5 up to the hardware rsqrt's last bit
                    rowsq = torch.full((M, 4), float(K), device="cuda")
                    got = run_col(ctx, a, w, STORE, split, row_off, nt, rowsq, 0.0, bias)["out"]
                    assert float((got - (0.5 * ref + bias)).abs().max()) <= 1e-6 * float(ref.abs().max()), (name, M, split, row_off)
            elif epi == RESID:
                x0 = ints((M, N), -16, 16, 4).float().cuda()
                scale = torch.full((N,), 0.25, device="cuda")
                nw = torch.full((N,), 2.0, device="cuda")
                r = run_col(ctx, a, w, RESID, split, row_off, nt, None, 0.0, None, scale, x0, nw)
                want = x0 + 0.25 * ref
                assert torch.equal(r["out"], want), (name, M, split, row_off)
                assert torch.equal(r["next"].float(), (2.0 * want).to(torch.bfloat16).float())
                n_part = (N + 15) // 16 * used_split(ctx, N, split)
                rq = r["rowsq_out"][: M * n_part].view(M, n_part).sum(1)
                want_sq = (want.double() ** 2).sum(1)
                assert float(((rq.double() - want_sq).abs() / want_sq.clamp(min=1)).max()) < 1e-5
            else:
                got = run_col(ctx, a, w, SILU, split, row_off, nt, None, 0.0)["act"].float()
                g, u = ref[:, : N // 2], ref[:, N // 2:]
                want = torch.nn.functional.silu(g) * u
                # the gate / up integers are exact; only expf and the bf16 rounding of the product remain
                assert float(((got - want).abs() / want.abs().clamp(min=1.0)).max()) <= 2.0 ** -7, (name, M, split, row_off)
                assert bool((got[(g == 0) | (u == 0)] == 0).all())


@pytest.mark.parametrize("M", [1, 8, 16])
def test_gemm_col_sixteen_row_instantiation_is_exact(ctx, M):
    """rt_debug_tune(2301): launches of <= 16 rows take the 128-VGPR MT = 1 instantiation (two workgroups per CU, what decode lanes
    would run side by side - measured: no gain, DESIGN.md section 6, kept as a knob).  Same integer-exact checks on every shape."""
    ctx.lib.rt_debug_tune(2301, 0)
    try:
        for N, K, epi, norm, name in SHAPES:
            test_gemm_col_integer_exact(ctx, N, K, epi, norm, name, M)
    finally:
        ctx.lib.rt_debug_tune(2300, 0)


@pytest.mark.parametrize("M", [8, 17, 32, 33, 64])
def test_gate_up_one_and_a_half_pairs_per_workgroup_changes_no_bit(ctx, M):
    """The 1.7B talker's gate/up GEMM has 384 gate/up tile pairs for 256 CUs; by default it runs as 256 workgroups of 1.5 pairs
    (k_gemm_col<SILU, MT, NPRE, X>: the half pair's 8 gate + 8 up columns share one more MFMA tile).  Every column keeps its wave
    split and summation order, so the activations equal the one-pair-per-workgroup launch (rt_debug_tune 2400) bit for bit."""
    N, K = 12288, 2048
    assert ctx.device_info()["n_cu"] == 256
    g = torch.Generator().manual_seed(5 + M)
    x_prev = torch.randn(M, K, generator=g)
    a = ((1.0 + 0.1 * torch.randn(K, generator=g)) * x_prev).to(torch.bfloat16).cuda()
    w = (torch.randn(N, K, generator=g) * 0.03).to(torch.bfloat16).cuda()
    sq = (x_prev ** 2).view(M, 8, K // 8).sum(2).cuda()
    got = run_col(ctx, a, w, SILU, 0, 0, 1, sq, 1e-6)["act"]
    ctx.lib.rt_debug_tune(2400, 0)
    try:
        pairs = run_col(ctx, a, w, SILU, 0, 0, 1, sq, 1e-6)["act"]
    finally:
        ctx.lib.rt_debug_tune(2401, 0)
    assert torch.equal(got, pairs) and float(got.float().abs().max()) > 0.1
    ref = (a.float() @ w.float().T) * torch.rsqrt((x_prev ** 2).mean(1) + 1e-6).cuda()[:, None]
    want = torch.nn.functional.silu(ref[:, : N // 2]) * ref[:, N // 2:]
    assert float((got.float() - want).abs().max()) < 2.0 ** -7 * float(want.abs().max()) + 2e-3 * float(ref.abs().max())


@pytest.mark.parametrize("N,K,epi,norm,name", SHAPES, ids=[s[4] for s in SHAPES])
def test_gemm_col_random_operands(ctx, N, K, epi, norm, name):
    """Random bf16 operands, real RMSNorm partials: tolerance 2e-3 of the output scale (f32 accumulation in another order)."""
    M = 32
    g = torch.Generator().manual_seed(11)
    x_prev = torch.randn(M, K, generator=g)                       # the (un-normalised) residual stream the operand came from
    ln = 1.0 + 0.1 * torch.randn(K, generator=g)
    a = (ln * x_prev).to(torch.bfloat16).cuda()                   # what the producer stores: bf16(norm_w .* x)
    w = (torch.randn(N, K, generator=g) * 0.03).to(torch.bfloat16).cuda()
    eps = 1e-6
    parts = 8
    sq = (x_prev ** 2).view(M, parts, K // parts).sum(2).cuda() if norm else None
    inv = torch.rsqrt((x_prev ** 2).mean(1) + eps).cuda()[:, None] if norm else 1.0
    ref = (a.float() @ w.float().T) * inv
    tol = 2e-3 * float(ref.abs().max())
    if epi == STORE:
        bias = (torch.randn(N, generator=g) * 0.1).cuda()
        got = run_col(ctx, a, w, STORE, 0, 0, 1, sq, eps, bias)["out"]
        assert float((got - (ref + bias)).abs().max()) < tol
    elif epi == RESID:
        x0 = torch.randn(M, N, generator=g).cuda()
        scale = (0.5 + torch.rand(N, generator=g)).cuda()
        nw = (1.0 + 0.1 * torch.randn(N, generator=g)).cuda()
        r = run_col(ctx, a, w, RESID, 0, 0, 1, None, eps, None, scale, x0, nw)
        want = x0 + scale * ref
        assert float((r["out"] - want).abs().max()) < tol
        assert float((r["next"].float() - (nw * r["out"]).to(torch.bfloat16).float()).abs().max()) == 0.0      # the operand is bf16 of THIS x
        n_part = (N + 15) // 16 * used_split(ctx, N, 0)
        rq = r["rowsq_out"][: M * n_part].view(M, n_part).sum(1)
        assert float(((rq - (r["out"] ** 2).sum(1)).abs() / (r["out"] ** 2).sum(1)).max()) < 1e-5
    else:
        got = run_col(ctx, a, w, SILU, 0, 0, 1, sq, eps)["act"].float()
        want = torch.nn.functional.silu(ref[:, : N // 2]) * ref[:, N // 2:]
        assert float((got - want).abs().max()) < 2.0 ** -7 * float(want.abs().max()) + tol


def rope_ref(x, pos, theta, d):
    inv = 1.0 / (theta ** (torch.arange(0, d, 2, dtype=torch.float32) / d))
    fr = pos.float()[:, None] * inv[None]
    c, s = fr.cos()[:, None], fr.sin()[:, None]
    x1, x2 = x[..., : d // 2], x[..., d // 2:]
    return torch.cat([x1 * c - x2 * s, x2 * c + x1 * s], -1)


@pytest.mark.parametrize("d,heads,kvh,max_pos,M,Lp,pos_lo,pos_hi", [
    (128, 16, 8, 1024, 32, 460, 470, 530),     # the talker's decode step at C3: 16 waves, 460-row shared prefix, ctx ~ 500
    (128, 16, 8, 1024, 32, 0, 0, 40),          # no shared prefix, short contexts incl. position 0
    (128, 16, 8, 17, 32, 0, 2, 16),            # the predictor's passes: 4 waves, <= 16 cached rows
    (128, 16, 8, 1024, 8, 460, 461, 480),      # 0.6B batch 8
    (64, 4, 4, 256, 5, 33, 33, 100),           # REP 1, head_dim 64, prefix end inside a 128-B line
    (32, 4, 1, 128, 3, 7, 7, 60),              # REP 4 (no early batch), head_dim 32
])
@pytest.mark.parametrize("mfma", [0, 1, 2])
def test_fused_attention_matches_fp32(ctx, d, heads, kvh, max_pos, M, Lp, pos_lo, pos_hi, mfma):
    """mfma = 1 / 2 route the head_dim-128 shared-prefix cases through the matrix-core kernel (attention_mfma.hip, rt_debug_tune
    1501: four rows per workgroup, 1502: one row per workgroup), the others are unaffected by the switch."""
    if mfma and not (d == 128 and Lp >= 64):
        pytest.skip("the matrix-core path exists for head_dim 128 with a shared prefix only")
    ctx.lib.rt_debug_tune(1500 + mfma, 0)
    try:
        _fused_attention_matches_fp32(ctx, d, heads, kvh, max_pos, M, Lp, pos_lo, pos_hi)
    finally:
        ctx.lib.rt_debug_tune(1500, 0)


def _fused_attention_matches_fp32(ctx, d, heads, kvh, max_pos, M, Lp, pos_lo, pos_hi):
    g = torch.Generator().manual_seed(101)
    slots = M + 1
    pslot = M if Lp > 0 else -1
    width = (heads + 2 * kvh) * d
    qkv = torch.randn(M, width, generator=g)
    qn, kn = 1.0 + 0.1 * torch.randn(d, generator=g), 1.0 + 0.1 * torch.randn(d, generator=g)
    kc = torch.randn(slots, kvh, max_pos, d, generator=g).to(torch.bfloat16)
    vc = torch.randn(slots, kvh, max_pos, d, generator=g).to(torch.bfloat16)
    pos = torch.randint(pos_lo, pos_hi + 1, (M,), generator=g).to(torch.int32)
    pos[0], pos[-1] = pos_lo, pos_hi
    slot = torch.arange(M, dtype=torch.int32)
    theta, eps = 1e6, 1e-6
    inv = 1.0 / (theta ** (torch.arange(0, d, 2, dtype=torch.float32) / d))
    fr = torch.arange(max_pos, dtype=torch.float32)[:, None] * inv[None]
    cos, sin = fr.cos().contiguous().cuda(), fr.sin().contiguous().cuda()
    kd, vd = kc.cuda(), vc.cuda()
    out = torch.zeros(M, heads * d, dtype=torch.bfloat16, device="cuda")
    qkv_d, qn_d, kn_d, slot_d, pos_d = qkv.cuda(), qn.cuda(), kn.cuda(), slot.cuda(), pos.cuda()
    torch.cuda.synchronize()
    rc = ctx.lib.rt_debug_attention_fused(ctx.handle, ptr(qkv_d), M, heads, kvh, d, ptr(qn_d), ptr(kn_d), eps, ptr(cos), ptr(sin), ptr(slot_d),
                                          ptr(pos_d), 0, ptr(kd), ptr(vd), slots, max_pos, pslot, Lp, ptr(out))
    ctx.check(rc, "rt_debug_attention_fused")
    torch.cuda.synchronize()

    def rms(x, w):
        return w * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps))

    q = rope_ref(rms(qkv[:, : heads * d].view(M, heads, d), qn), pos, theta, d)
    k = rope_ref(rms(qkv[:, heads * d: (heads + kvh) * d].view(M, kvh, d), kn), pos, theta, d)
    v = qkv[:, (heads + kvh) * d:].view(M, kvh, d)
    k_new, v_new = kd.cpu(), vd.cpu()
    rep = heads // kvh
    worst = 0.0
    for r in range(M):
        p = int(pos[r])
        # the appended cache row: bf16 of the normed / rotated key, the raw value
        assert float((k_new[r, :, p].float() - k[r].to(torch.bfloat16).float()).abs().max()) <= 2.0 ** -7 * float(k[r].abs().max())
        assert torch.equal(v_new[r, :, p], v[r].to(torch.bfloat16))
        K = torch.empty(kvh, p + 1, d)
        V = torch.empty(kvh, p + 1, d)
        n_pre = min(Lp, p + 1) if pslot >= 0 else 0
        K[:, :n_pre], V[:, :n_pre] = kc[pslot, :, :n_pre].float(), vc[pslot, :, :n_pre].float()
        K[:, n_pre:p], V[:, n_pre:p] = kc[r, :, n_pre:p].float(), vc[r, :, n_pre:p].float()
        K[:, p], V[:, p] = k_new[r, :, p].float(), v_new[r, :, p].float()
        s = torch.einsum("hd,htd->ht", q[r], K.repeat_interleave(rep, 0)) * d ** -0.5
        ref = torch.einsum("ht,htd->hd", torch.softmax(s, -1), V.repeat_interleave(rep, 0)).reshape(-1)
        worst = max(worst, float((out[r].float().cpu() - ref).abs().max()) / max(1.0, float(ref.abs().max())))
    assert worst < 1e-2, worst                                          # bf16 output rounding is 4e-3 relative
    # nothing but the appended rows changed in the caches
    mask = torch.ones(slots, max_pos, dtype=torch.bool)
    mask[slot.long(), pos.long()] = False
    assert torch.equal(k_new.transpose(1, 2)[mask], kc.transpose(1, 2)[mask]) and torch.equal(v_new.transpose(1, 2)[mask], vc.transpose(1, 2)[mask])


@pytest.mark.parametrize("d,heads,kvh,max_pos,M,Lp,pos_lo,pos_hi", [
    (128, 16, 8, 1024, 32, 460, 461, 540),     # C3 talker step: 16 waves, shared 460-row prefix
    (128, 16, 8, 1024, 32, 0, 0, 300),         # no prefix
    (128, 16, 8, 17, 32, 0, 1, 16),            # predictor passes: 4 waves
    (64, 4, 4, 512, 6, 130, 130, 400),         # head_dim 64 (two cache rows per 128-B line)
    (32, 4, 1, 256, 4, 9, 9, 200),             # REP 4
])
@pytest.mark.parametrize("mfma", [0, 1, 2])
def test_fused_attention_key_census(ctx, d, heads, kvh, max_pos, M, Lp, pos_lo, pos_hi, mfma):
    if mfma and not (d == 128 and Lp >= 64):
        pytest.skip("the matrix-core path exists for head_dim 128 with a shared prefix only")
    ctx.lib.rt_debug_tune(1500 + mfma, 0)
    try:
        _fused_attention_key_census(ctx, d, heads, kvh, max_pos, M, Lp, pos_lo, pos_hi)
    finally:
        ctx.lib.rt_debug_tune(1500, 0)


def _fused_attention_key_census(ctx, d, heads, kvh, max_pos, M, Lp, pos_lo, pos_hi):
    """EXACT check of WHICH cache rows the fused decode attention reads.  With q = 0 every score is 0 and the softmax is
    uniform, so the output is the plain mean of the V rows in the context.  V[p] is the indicator of p mod d, hence output
    dim j is (number of context positions congruent to j) / (pos + 1): a dropped, duplicated or misplaced row (own slot
    instead of the prefix slot, a row past `pos`, a stale copy of the appended row) changes an integer count and shows up
    bit for bit - which a tolerance on random data cannot see (one key in 500 moves the output by 0.2 %)."""
    g = torch.Generator().manual_seed(7)
    slots = M + 1
    pslot = M if Lp > 0 else -1
    width = (heads + 2 * kvh) * d
    POISON = 64.0
    pos = torch.randint(pos_lo, pos_hi + 1, (M,), generator=g).to(torch.int32)
    pos[0], pos[-1] = pos_lo, pos_hi
    onehot = torch.eye(d)[torch.arange(max_pos) % d]                        # [max_pos, d]
    vc = torch.full((slots, kvh, max_pos, d), POISON)
    for r in range(M):
        vc[r, :, Lp: int(pos[r])] = onehot[Lp: int(pos[r])]                 # own rows [Lp, pos); row `pos` is appended by the launch
    if pslot >= 0:
        vc[pslot, :, :Lp] = onehot[:Lp]                                     # the shared prefix lives in its slot only
    kc = torch.randn(slots, kvh, max_pos, d, generator=g)
    qkv = torch.zeros(M, width)
    qkv[:, heads * d: (heads + kvh) * d] = torch.randn(M, kvh * d, generator=g)
    qkv[:, (heads + kvh) * d:] = onehot[pos.long()].repeat(1, kvh)
    inv = 1.0 / (1e6 ** (torch.arange(0, d, 2, dtype=torch.float32) / d))
    fr = torch.arange(max_pos, dtype=torch.float32)[:, None] * inv[None]
    cos, sin = fr.cos().contiguous().cuda(), fr.sin().contiguous().cuda()
    kd, vd = kc.to(torch.bfloat16).cuda(), vc.to(torch.bfloat16).cuda()
    out = torch.zeros(M, heads * d, dtype=torch.bfloat16, device="cuda")
    ones = torch.ones(d).cuda()
    qkv_d, slot_d, pos_d = qkv.cuda(), torch.arange(M, dtype=torch.int32).cuda(), pos.cuda()
    torch.cuda.synchronize()
    ctx.check(ctx.lib.rt_debug_attention_fused(ctx.handle, ptr(qkv_d), M, heads, kvh, d, ptr(ones), ptr(ones), 1e-6, ptr(cos), ptr(sin), ptr(slot_d),
                                               ptr(pos_d), 0, ptr(kd), ptr(vd), slots, max_pos, pslot, Lp, ptr(out)), "rt_debug_attention_fused")
    torch.cuda.synchronize()
    got = out.float().cpu().view(M, heads, d)
    for r in range(M):
        n = int(pos[r]) + 1
        count = onehot[:n].sum(0)
        want = (count / float(n)).to(torch.bfloat16).float()
        assert torch.equal(got[r], want[None].expand(heads, d)), (r, n, (got[r] - want).abs().max())


@pytest.mark.parametrize("pos", [0, 3, 15])
def test_fused_attention_without_slot_and_position_arrays(ctx, pos):
    """The residual-code predictor's passes: every row at the same position, row r in slot r - the launch takes NULL for both
    arrays (no dependent scalar loads in front of its K / V requests).  Same exact census as above, and bit-equal to the launch
    that is handed the arrays."""
    d, heads, kvh, max_pos, M = 128, 16, 8, 17, 32
    g = torch.Generator().manual_seed(11)
    slots, width = M, (heads + 2 * kvh) * d
    onehot = torch.eye(d)[torch.arange(max_pos) % d]
    vc = torch.full((slots, kvh, max_pos, d), 64.0)
    vc[:, :, :pos] = onehot[:pos]
    kc = torch.randn(slots, kvh, max_pos, d, generator=g)
    qkv = torch.zeros(M, width)
    qkv[:, heads * d: (heads + kvh) * d] = torch.randn(M, kvh * d, generator=g)
    qkv[:, (heads + kvh) * d:] = onehot[pos].repeat(M, kvh)
    inv = 1.0 / (1e6 ** (torch.arange(0, d, 2, dtype=torch.float32) / d))
    fr = torch.arange(max_pos, dtype=torch.float32)[:, None] * inv[None]
    cos, sin = fr.cos().contiguous().cuda(), fr.sin().contiguous().cuda()
    ones = torch.ones(d).cuda()
    qkv_d = qkv.cuda()
    outs = []
    for arrays in (False, True):
        kd, vd = kc.to(torch.bfloat16).cuda(), vc.to(torch.bfloat16).cuda()
        out = torch.zeros(M, heads * d, dtype=torch.bfloat16, device="cuda")
        slot_d = torch.arange(M, dtype=torch.int32).cuda() if arrays else None
        pos_d = torch.zeros(M, dtype=torch.int32).cuda() if arrays else None
        torch.cuda.synchronize()
        ctx.check(ctx.lib.rt_debug_attention_fused(ctx.handle, ptr(qkv_d), M, heads, kvh, d, ptr(ones), ptr(ones), 1e-6, ptr(cos), ptr(sin), ptr(slot_d),
                                                   ptr(pos_d), pos, ptr(kd), ptr(vd), slots, max_pos, -1, 0, ptr(out)), "rt_debug_attention_fused")
        torch.cuda.synchronize()
        outs.append((out.float().cpu(), kd.float().cpu(), vd.float().cpu()))
    got = outs[0][0].view(M, heads, d)
    want = (onehot[: pos + 1].sum(0) / float(pos + 1)).to(torch.bfloat16).float()
    assert torch.equal(got, want[None, None].expand(M, heads, d))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)                                   # output and both caches (the appended row) equal the array form
