"""End-to-end GPU parity of the model path (through the C ABI) against the CPU oracle on the same
seeded synthetic weights.  PARITY UNPINNED against the reference itself (qwen-tts is absent, see
oracle/model.py); what is pinned here is HIP == oracle.

Tolerances (stated per test): the GPU rounds activations to bf16 at every GEMM input and keeps the
KV cache in bf16 (as the reference's bf16 model does); the oracle keeps activations in float32.
  - teacher-forced logits: max |diff| <= 4 % of the logits' standard deviation
  - greedy code agreement on free-running decode: >= 90 % of the codes of the first frames
  - code2wav waveform RMSE < 1e-3 (BASELINE.json's stated bar), same codes in both
"""
import numpy as np
import pytest
import torch

from oracle.model import OracleModel, Voice
from oracle.sampling import SamplingParams
from rho_tts_amd import config, weights

pytestmark = pytest.mark.gpu
torch.set_num_threads(8)


@pytest.fixture(scope="module")
def ctx():
    from rho_tts_amd import _native
    c = _native.Context(0)
    yield c
    c.close()


def build(ctx, cfg, max_batch=8):
    from rho_tts_amd._native_model import NativeModel
    state = weights.synthetic_state(cfg, 789)
    nm = NativeModel(ctx, cfg, max_batch=max_batch)
    nm.load_state({k: v.cuda() for k, v in state.items()})
    return nm, OracleModel(cfg, state)


@pytest.fixture(scope="module", params=["tiny", "small"])
def models(request, ctx):
    cfg = config.PRESETS[request.param]()
    nm, om = build(ctx, cfg)
    yield cfg, nm, om
    nm.close()


def make_voice(cfg, clone=True, n_ref=9, seed=5):
    g = torch.Generator().manual_seed(seed)
    if clone:
        return Voice("english", speaker_embed=(torch.randn(cfg.talker.hidden, generator=g) * 0.05),
                     ref_text_ids=[int(v) for v in torch.randint(0, cfg.text_vocab - 64, (5,), generator=g)],
                     ref_codes=torch.randint(0, cfg.codec.codebook_size, (n_ref, cfg.n_groups), generator=g))
    return Voice("chinese", speaker="ryan")


def set_voice(nm, v):
    return nm.set_voice(v.language, v.speaker, v.speaker_embed, v.ref_text_ids, v.ref_codes)


TEXTS = [[10, 11, 12], [20, 21, 22, 23, 24, 25, 26], [30], [40, 41, 42, 43]]
FRAMES = [6, 5, 7, 6]


@pytest.mark.parametrize("clone", [True, False])
def test_teacher_forced_logits_match_oracle(models, clone):
    from rho_tts_amd._native_model import RtSampling
    cfg, nm, om = models
    v = make_voice(cfg, clone)
    n_prefix = set_voice(nm, v)
    assert n_prefix == om.prefix_embeddings(v).shape[0] == nm.prefix_len()
    tr_o = {}
    free = om.generate(v, TEXTS, FRAMES, SamplingParams(), trace=tr_o)          # oracle greedy trajectory
    codes, tr = nm.generate(TEXTS, FRAMES, RtSampling(0, 1.0, 1, 1.0, 1.0), forced_codes=free, trace=True)
    for a, b in zip(codes, free):
        assert torch.equal(a, b)                                                   # forced codes come back unchanged
    t_o = torch.stack(tr_o["talker_logits"])                                       # [T, B, V]
    T = t_o.shape[0]
    t_g = tr["talker"][:T].cpu()
    valid = torch.zeros(T, len(TEXTS), dtype=torch.bool)
    for b, n in enumerate(FRAMES):
        valid[:n, b] = True
    V0 = cfg.codec.codebook_size
    err = (t_g - t_o)[valid][:, :V0].abs().max()
    assert float(err) <= 0.04 * float(t_o[valid][:, :V0].std()), float(err)
    p_o = torch.stack(tr_o["pred_logits"]).view(T, cfg.n_groups - 1, len(TEXTS), -1)
    p_g = tr["predictor"][:T].cpu()
    vp = valid[:, None, :].expand(-1, cfg.n_groups - 1, -1)
    err = (p_g - p_o)[vp].abs().max()
    assert float(err) <= 0.04 * float(p_o[vp].std()), float(err)


def test_greedy_free_running_agreement(models):
    from rho_tts_amd._native_model import RtSampling
    cfg, nm, om = models
    v = make_voice(cfg, True)
    set_voice(nm, v)
    want = om.generate(v, TEXTS, FRAMES, SamplingParams())
    got = nm.generate(TEXTS, FRAMES, RtSampling(0, 1.0, 1, 1.0, 1.0))
    assert [g.shape for g in got] == [w.shape for w in want]
    # near-ties in random-weight logits can flip an arg-max and the trajectories then diverge for good,
    # so compare the first two frames of every item, where divergence has not compounded
    agree = np.mean([float((g[:2] == w[:2]).float().mean()) for g, w in zip(got, want)])
    assert agree >= 0.9, agree


def test_sampling_is_reproducible_and_batch_invariant(models):
    from rho_tts_amd._native_model import RtSampling
    cfg, nm, om = models
    set_voice(nm, make_voice(cfg, True))
    sp = RtSampling(1, 0.9, 50, 1.0, 1.05)
    a = nm.generate(TEXTS, FRAMES, sp, seed=123)
    b = nm.generate(TEXTS, FRAMES, sp, seed=123)
    c = nm.generate(TEXTS, FRAMES, sp, seed=124)
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    assert not all(torch.equal(x, y) for x, y in zip(a, c))
    solo = nm.generate([TEXTS[2]], [FRAMES[2]], sp, seed=123, item_ids=[2])
    assert torch.equal(solo[0], a[2])                      # item 2 alone == item 2 inside the batch
    for x in a:
        assert int(x[:, 0].max()) < cfg.codec.codebook_size and int(x.min()) >= 0


def test_eos_and_max_frames(models):
    from rho_tts_amd._native_model import RtSampling
    cfg, nm, om = models
    set_voice(nm, make_voice(cfg, False))
    G = cfg.n_groups
    forced = [torch.tensor([[1] * G, [2] * G, [cfg.codec_eos_id] + [0] * (G - 1)]), torch.randint(0, 60, (5, G))]
    out = nm.generate([[1, 2], [3]], [5, 4], RtSampling(0, 1, 1, 1, 1), ignore_eos=False, forced_codes=forced)
    assert out[0].shape[0] == 2 and out[1].shape[0] == 4
    with pytest.raises(RuntimeError, match="length"):
        nm.generate([[1] * 10], [cfg.max_positions], RtSampling(0, 1, 1, 1, 1))
    with pytest.raises(ValueError):
        nm.generate([[1]], [3], RtSampling(1, 0.9, 500, 1.0, 1.0))


def test_voice_blob_roundtrip(models):
    from rho_tts_amd._native_model import RtSampling
    cfg, nm, om = models
    v = make_voice(cfg, True)
    n = set_voice(nm, v)
    ref = nm.generate(TEXTS[:2], FRAMES[:2], RtSampling(0, 1, 1, 1, 1))
    blob = nm.export_voice().clone()
    set_voice(nm, make_voice(cfg, False))                  # overwrite with another voice
    other = nm.generate(TEXTS[:2], FRAMES[:2], RtSampling(0, 1, 1, 1, 1))
    nm.import_voice(n, blob)                               # what a non-root rank does after the RCCL broadcast
    back = nm.generate(TEXTS[:2], FRAMES[:2], RtSampling(0, 1, 1, 1, 1))
    assert all(torch.equal(a, b) for a, b in zip(ref, back))
    assert not all(torch.equal(a, b) for a, b in zip(ref, other))


def test_code2wav_rmse(models):
    cfg, nm, om = models
    g = torch.Generator().manual_seed(9)
    Q = cfg.codec.num_quantizers
    lens = [7, 12, 9]
    codes = [torch.randint(0, cfg.codec.codebook_size, (n, Q), generator=g) for n in lens]
    wavs = nm.code2wav(codes)
    for c, w in zip(codes, wavs):
        ref = om.code2wav(c.T[None])[0]
        assert w.shape[0] == ref.shape[0] == om.wav_length(c.shape[0]) == nm.wav_length(c.shape[0])
        rmse = float(torch.sqrt(torch.mean((w.cpu() - ref) ** 2)))
        assert rmse < 1e-3, rmse
        assert float(ref.abs().max()) > 0.05               # the comparison is not on silence
    # right-padding inside a ragged batch does not leak into shorter items (everything is causal up to the
    # transposed convs' one-step look-ahead, which the length trim removes)
    solo = nm.code2wav([codes[0]])[0]
    assert torch.equal(solo, wavs[0])


def test_legacy_and_column_decode_paths_agree(models):
    """The 9-launch split-K decode path (used when 2B > 64) and the 5-launch column-owner path compute the same logits."""
    from rho_tts_amd._native_model import RtSampling
    cfg, nm, om = models
    v = make_voice(cfg, True)
    set_voice(nm, v)
    free = om.generate(v, TEXTS, FRAMES, SamplingParams())
    lib = nm.lib
    try:
        lib.rt_debug_tune(100, 0)                                   # legacy
        _, tr_a = nm.generate(TEXTS, FRAMES, RtSampling(0, 1, 1, 1, 1), forced_codes=free, trace=True)
    finally:
        lib.rt_debug_tune(101, 0)                                   # column path (default)
    _, tr_b = nm.generate(TEXTS, FRAMES, RtSampling(0, 1, 1, 1, 1), forced_codes=free, trace=True)
    V0 = cfg.codec.codebook_size
    for key in ("talker", "predictor"):
        a, b = tr_a[key].cpu(), tr_b[key].cpu()
        if key == "talker":
            a, b = a[..., :V0], b[..., :V0]
        assert float((a - b).abs().max()) <= 0.06 * float(a.std()) + 1e-6      # each is within 4 % of the oracle


def test_decode_lanes_and_tile_split_do_not_change_results(ctx):
    """Items are independent, so cutting the batch into concurrently decoding lanes (rt_debug_tune 40n) must leave every
    code unchanged; splitting a 16-column GEMM tile over 2 or 4 workgroups (50n) only regroups the RMSNorm partial sums."""
    from rho_tts_amd._native_model import RtSampling
    cfg = config.PRESETS["tiny"]()
    nm, _ = build(ctx, cfg, max_batch=16)
    try:
        set_voice(nm, make_voice(cfg, True))
        g = torch.Generator().manual_seed(3)
        texts = [[int(v) for v in torch.randint(0, cfg.text_vocab - 64, (int(n),), generator=g)] for n in torch.randint(1, 9, (16,), generator=g)]
        frames = [int(v) for v in torch.randint(3, 8, (16,), generator=g)]
        sp = RtSampling(1, 0.9, 50, 1.0, 1.05)
        lib = nm.lib
        base, tr0 = nm.generate(texts, frames, sp, seed=77, trace=True)
        try:
            lib.rt_debug_tune(402, 0)
            two, tr2 = nm.generate(texts, frames, sp, seed=77, trace=True)
        finally:
            lib.rt_debug_tune(401, 0)
        assert all(torch.equal(a, b) for a, b in zip(base, two))
        for key in ("talker", "predictor"):
            assert torch.equal(tr0[key], tr2[key])
        try:        # sampler with the next-input embedding fused in (801, default) == separate gather launch (800)
            lib.rt_debug_tune(800, 0)
            sep, tr_sep = nm.generate(texts, frames, sp, seed=77, trace=True)
        finally:
            lib.rt_debug_tune(801, 0)
        assert all(torch.equal(a, b) for a, b in zip(base, sep))
        for key in ("talker", "predictor"):
            assert torch.equal(tr0[key], tr_sep[key])
        try:        # frame counter advanced by the talker-input launch's last workgroup (2701) == by its own launch (2700, default)
            lib.rt_debug_tune(2701, 0)
            inc, tr_inc = nm.generate(texts, frames, sp, seed=77, trace=True)
        finally:
            lib.rt_debug_tune(2700, 0)
        assert all(torch.equal(a, b) for a, b in zip(base, inc))
        for key in ("talker", "predictor"):
            assert torch.equal(tr0[key], tr_inc[key])
        try:        # two-position first pass of the predictor: q/k norm + RoPE + append as their own launch (2800) == inside the fused attention (2801, default)
            lib.rt_debug_tune(2800, 0)
            two_l, tr_2l = nm.generate(texts, frames, sp, seed=77, trace=True)
        finally:
            lib.rt_debug_tune(2801, 0)
        assert all(torch.equal(a, b) for a, b in zip(base, two_l))
        for key in ("talker", "predictor"):
            assert torch.equal(tr0[key], tr_2l[key])
        outs = {}
        try:
            for code in (501, 502, 504):
                lib.rt_debug_tune(code, 0)
                _, outs[code] = nm.generate(texts, frames, RtSampling(0, 1, 1, 1, 1), forced_codes=base, trace=True)
        finally:
            lib.rt_debug_tune(500, 0)
        V0 = cfg.codec.codebook_size
        for code in (502, 504):
            for key in ("talker", "predictor"):
                a, b = outs[501][key].cpu(), outs[code][key].cpu()
                if key == "talker":
                    a, b = a[..., :V0], b[..., :V0]
                # regrouped float32 partial sums move a row scale by an ulp; where that flips the bfloat16 rounding of one operand
                # element the logits move by a bfloat16 ulp of that element's term (K is 64 here, so one term is a visible part
                # of a logit) - a handful of such elements is expected, a wrong sum would move all of them
                d = (a - b).abs()
                print(f"tile split {code} {key}: max {float(d.max()):.2e} mean {float(d.mean()):.2e} std {float(a.std()):.3f}")
                assert float(d.max()) <= 2e-2 * float(a.std()) + 1e-6
                assert float(d.mean()) <= 2e-4 * float(a.std()) + 1e-7
    finally:
        nm.close()


def test_64_row_decode_launches_match_32_row_launches(ctx):
    """Batches of 17..32 sequences put 2B > 32 rows into the predictor's first pass: one 64-row column-GEMM launch
    (rt_debug_tune 701, default) must give bit-identical logits to two 32-row launches (700), and both stay within the
    legacy path's tolerance."""
    from rho_tts_amd._native_model import RtSampling
    cfg = config.PRESETS["tiny"]()
    nm, _ = build(ctx, cfg, max_batch=24)
    try:
        set_voice(nm, make_voice(cfg, True))
        g = torch.Generator().manual_seed(5)
        texts = [[int(v) for v in torch.randint(0, cfg.text_vocab - 64, (int(n),), generator=g)] for n in torch.randint(1, 9, (21,), generator=g)]
        frames = [int(v) for v in torch.randint(3, 7, (21,), generator=g)]
        lib = nm.lib
        sp = RtSampling(1, 0.9, 50, 1.0, 1.05)
        codes64, tr64 = nm.generate(texts, frames, sp, seed=99, trace=True)
        try:
            lib.rt_debug_tune(700, 0)
            codes32, tr32 = nm.generate(texts, frames, sp, seed=99, trace=True)
            lib.rt_debug_tune(100, 0)
            _, tr_legacy = nm.generate(texts, frames, RtSampling(0, 1, 1, 1, 1), forced_codes=codes64, trace=True)
        finally:
            lib.rt_debug_tune(701, 0)
            lib.rt_debug_tune(101, 0)
        assert all(torch.equal(a, b) for a, b in zip(codes64, codes32))
        for key in ("talker", "predictor"):
            assert torch.equal(tr64[key], tr32[key])
        _, tr_forced = nm.generate(texts, frames, RtSampling(0, 1, 1, 1, 1), forced_codes=codes64, trace=True)
        V0 = cfg.codec.codebook_size
        for key in ("talker", "predictor"):
            a, b = tr_legacy[key].cpu(), tr_forced[key].cpu()
            if key == "talker":
                a, b = a[..., :V0], b[..., :V0]
            assert float((a - b).abs().max()) <= 0.06 * float(a.std()) + 1e-6
    finally:
        nm.close()


def test_equal_width_predictor_takes_the_column_path_too(ctx):
    """0.6B-style model: predictor hidden == talker hidden, no mtp projection; the predictor's first input row is the talker's
    normalised hidden state itself.  Teacher-forced logits of the column path match the oracle (4 % of sigma) and the legacy path."""
    import dataclasses
    from rho_tts_amd._native_model import RtSampling
    base = config.PRESETS["tiny"]()
    cfg = dataclasses.replace(base, name="tiny-equal", predictor=dataclasses.replace(base.predictor, hidden=base.talker.hidden))
    assert not cfg.has_mtp_proj
    nm, om = build(ctx, cfg)
    try:
        v = make_voice(cfg, True)
        set_voice(nm, v)
        tr_o = {}
        free = om.generate(v, TEXTS, FRAMES, SamplingParams(), trace=tr_o)
        _, tr = nm.generate(TEXTS, FRAMES, RtSampling(0, 1.0, 1, 1.0, 1.0), forced_codes=free, trace=True)
        try:
            nm.lib.rt_debug_tune(100, 0)
            _, tr_legacy = nm.generate(TEXTS, FRAMES, RtSampling(0, 1.0, 1, 1.0, 1.0), forced_codes=free, trace=True)
        finally:
            nm.lib.rt_debug_tune(101, 0)
        T = len(tr_o["talker_logits"])
        valid = torch.zeros(T, len(TEXTS), dtype=torch.bool)
        for b, n in enumerate(FRAMES):
            valid[:n, b] = True
        V0 = cfg.codec.codebook_size
        t_o = torch.stack(tr_o["talker_logits"])
        err = (tr["talker"][:T].cpu() - t_o)[valid][:, :V0].abs().max()
        assert float(err) <= 0.04 * float(t_o[valid][:, :V0].std()), float(err)
        p_o = torch.stack(tr_o["pred_logits"]).view(T, cfg.n_groups - 1, len(TEXTS), -1)
        vp = valid[:, None, :].expand(-1, cfg.n_groups - 1, -1)
        err = (tr["predictor"][:T].cpu() - p_o)[vp].abs().max()
        assert float(err) <= 0.04 * float(p_o[vp].std()), float(err)
        for key in ("talker", "predictor"):
            a, b = tr_legacy[key].cpu(), tr[key].cpu()
            if key == "talker":
                a, b = a[..., :V0], b[..., :V0]
            assert float((a - b).abs().max()) <= 0.06 * float(a.std()) + 1e-6
    finally:
        nm.close()


def test_engine_chunked_vocode_matches_oracle_chunked_decode(ctx):
    """Sequences longer than codec.chunk_frames go through Engine.vocode's chunking (chunk + left context, the reference
    architecture's chunked_decode): same waveform as the oracle's chunked_code2wav, RMSE < 1e-3; short and long items mixed."""
    from rho_tts_amd.engine import Engine
    cfg = config.PRESETS["tiny"]()
    eng = Engine(cfg=cfg, model_path="tiny", device_ordinal=0, max_batch=4, weight_seed=789, synthetic=True)
    try:
        om = OracleModel(cfg, weights.synthetic_state(cfg, 789))
        g = torch.Generator().manual_seed(13)
        Q, cf = cfg.codec.num_quantizers, cfg.codec.chunk_frames
        lens = [cf + 5, 3, 2 * cf + 1, cf]
        codes = [torch.randint(0, cfg.codec.codebook_size, (n, Q), generator=g) for n in lens]
        wavs = eng.vocode([c.cuda() for c in codes])
        for c, w in zip(codes, wavs):
            ref = om.chunked_code2wav(c.T[None])[0]
            assert w.shape[0] == ref.shape[0]
            rmse = float(torch.sqrt(torch.mean((w.cpu() - ref) ** 2)))
            assert rmse < 1e-3, (c.shape[0], rmse)
    finally:
        eng.close()


@pytest.mark.parametrize("preset", ["tiny", "small"])
def test_64_rows_on_the_column_path(ctx, preset):
    """Batches of 33..64 rows decode on the column-owner path too (the talker's GEMMs take 64 rows per launch, the predictor's
    two-position first pass runs as two 64-row launches): 50 ragged items in ONE static batch come out bit for bit as each does
    alone; and the legacy split-K path (rt_debug_tune(2032): rows above 32 fall back to it) still agrees to float tolerance."""
    from rho_tts_amd._native_model import RtSampling
    cfg = config.PRESETS[preset]()
    nm, _ = build(ctx, cfg, max_batch=64)
    try:
        set_voice(nm, make_voice(cfg, True))
        g = torch.Generator().manual_seed(17)
        texts = [[int(v) for v in torch.randint(0, cfg.text_vocab - 64, (int(n),), generator=g)] for n in torch.randint(1, 7, (50,), generator=g)]
        frames = [int(v) for v in torch.randint(3, 9, (50,), generator=g)]
        sp = RtSampling(1, 0.9, 50, 1.0, 1.05)
        alone = [nm.generate([t], [f], sp, seed=5, item_ids=[i])[0] for i, (t, f) in enumerate(zip(texts, frames))]
        got = nm.generate(texts, frames, sp, seed=5, item_ids=list(range(50)))
        assert all(torch.equal(a, b) for a, b in zip(got, alone))
        # the legacy path on the same batch: teacher-forced logits within float tolerance of the column path's
        greedy = RtSampling(0, 1, 1, 1, 1)
        f4 = [4] * 50
        forced = [a[:4] if a.shape[0] >= 4 else torch.cat([a, a[-1:].repeat(4 - a.shape[0], 1)]) for a in alone]
        _, tr_col = nm.generate(texts, f4, greedy, forced_codes=forced, trace=True)
        nm.lib.rt_debug_tune(2032, 0)
        try:
            _, tr_leg = nm.generate(texts, f4, greedy, forced_codes=forced, trace=True)
        finally:
            nm.lib.rt_debug_tune(2064, 0)
        V0 = cfg.codec.codebook_size
        for key in ("talker", "predictor"):
            a, b = tr_col[key].cpu(), tr_leg[key].cpu()
            if key == "talker":
                a, b = a[..., :V0], b[..., :V0]
            assert float((a - b).abs().max()) <= 0.06 * float(b.std()) + 1e-6
    finally:
        nm.lib.rt_debug_tune(2064, 0)
        nm.close()


def test_voice_switch_does_not_replay_stale_graphs(models):
    """The decode graphs bake the voice-prefix length into their attention nodes.  After warm generates with voice A, a voice of
    ANOTHER prefix length decoded with the same batch shape must re-capture: graph replay == eager launches (rt_debug_tune 200),
    bit for bit, and both match the oracle."""
    from rho_tts_amd._native_model import RtSampling
    cfg, nm, om = models
    greedy = RtSampling(0, 1.0, 1, 1.0, 1.0)
    va, vb = make_voice(cfg, True, n_ref=9, seed=5), make_voice(cfg, True, n_ref=4, seed=6)
    na = set_voice(nm, va)
    for _ in range(3):                                             # graphs captured and replayed for voice A
        nm.generate(TEXTS, FRAMES, greedy)
    nb = set_voice(nm, vb)
    assert na != nb
    got_graph = nm.generate(TEXTS, FRAMES, greedy)                 # same launch signature as the warm calls but for the prefix length
    try:
        nm.lib.rt_debug_tune(200, 0)
        got_eager = nm.generate(TEXTS, FRAMES, greedy)
    finally:
        nm.lib.rt_debug_tune(201, 0)
    assert all(torch.equal(a, b) for a, b in zip(got_graph, got_eager))
    tr_o = {}
    free = om.generate(vb, TEXTS, FRAMES, SamplingParams(), trace=tr_o)
    _, tr_graph = nm.generate(TEXTS, FRAMES, greedy, forced_codes=free, trace=True)
    t_o = torch.stack(tr_o["talker_logits"])
    T, V0 = t_o.shape[0], cfg.codec.codebook_size
    valid = torch.zeros(T, len(TEXTS), dtype=torch.bool)
    for b, n in enumerate(FRAMES):
        valid[:n, b] = True
    err = (tr_graph["talker"][:T].cpu() - t_o)[valid][:, :V0].abs().max()
    assert float(err) <= 0.04 * float(t_o[valid][:, :V0].std()), float(err)


@pytest.mark.parametrize("preset", ["tiny", "small"])
def test_queued_items_take_over_finished_rows(ctx, preset):
    """Continuous batching (rt_generate with n_items > max_batch): 13 ragged items on 4 decode rows.  Every item must come
    out exactly as it does alone on row 0 of a one-item call - same RNG stream (item id, ITS frame number), fresh repetition
    history, its own positions - whatever row it landed on and whenever it started; with fixed budgets, with live (sampled)
    end-of-sequence, and for every flag-polling period.  The oracle is not needed: 'alone' is already pinned to it above."""
    from rho_tts_amd._native_model import RtSampling
    cfg = config.PRESETS[preset]()
    nm, _ = build(ctx, cfg, max_batch=4)
    try:
        set_voice(nm, make_voice(cfg, True))
        g = torch.Generator().manual_seed(17)
        n = 13
        texts = [[int(v) for v in torch.randint(0, cfg.text_vocab - 64, (int(k),), generator=g)] for k in torch.randint(1, 9, (n,), generator=g)]
        frames = [int(v) for v in torch.randint(3, 29, (n,), generator=g)]
        ids = [100 + 7 * i for i in range(n)]
        fixed = RtSampling(1, 0.9, 50, 1.0, 1.05)                      # repetition penalty on: the history must be reset per item
        live = RtSampling(1, 1.5, 64, 1.0, 1.0)                        # hot sampling over the full top-k: end-of-sequence gets drawn
        alone_fixed = [nm.generate([t], [f], fixed, seed=5, item_ids=[i])[0] for t, f, i in zip(texts, frames, ids)]
        alone_live = [nm.generate([t], [f], live, seed=6, item_ids=[i], ignore_eos=False, min_frames=2)[0] for t, f, i in zip(texts, frames, ids)]
        if preset == "tiny":                                            # (the larger vocabulary of `small` rarely has EOS in its top-64)
            assert any(a.shape[0] < f for a, f in zip(alone_live, frames)), "no item ended early: the live case tests nothing"
        for every in (4, 1, 7):
            nm.lib.rt_debug_tune(1700 + every, 0)
            got = nm.generate(texts, frames, fixed, seed=5, item_ids=ids)
            st = nm.generate_stats()
            assert [c.shape[0] for c in got] == frames
            assert all(torch.equal(a, b) for a, b in zip(got, alone_fixed)), every
            assert st["rows"] == 4 and st["frames_kept"] == sum(frames) and st["hand_overs"] >= 2
            # list scheduling: no worse than sum / rows + longest (+ the polling slack per item)
            assert st["frames_run"] <= (sum(frames) + n * every) // 4 + max(frames) + every + 1
            got = nm.generate(texts, frames, live, seed=6, item_ids=ids, ignore_eos=False, min_frames=2)
            assert all(torch.equal(a, b) for a, b in zip(got, alone_live)), every
            assert nm.generate_stats()["frames_kept"] == sum(a.shape[0] for a in alone_live)
        # fewer rows than the engine has (rt_generate_args::max_rows): same items, same codes, on 2 rows
        got = nm.generate(texts, frames, fixed, seed=5, item_ids=ids, max_rows=2)
        assert all(torch.equal(a, b) for a, b in zip(got, alone_fixed)) and nm.generate_stats()["rows"] == 2
        # a static batch afterwards still works (the graphs of the queued shape are not replayed for it)
        again = nm.generate(texts[:4], frames[:4], fixed, seed=5, item_ids=ids[:4])
        assert all(torch.equal(a, b) for a, b in zip(again, alone_fixed[:4]))
        with pytest.raises(ValueError):                                 # teacher forcing is per frame counter: static batches only
            nm.generate(texts, frames, fixed, forced_codes=[torch.zeros(f, cfg.n_groups, dtype=torch.int64) for f in frames])
    finally:
        nm.lib.rt_debug_tune(1704, 0)
        nm.close()


def test_queued_items_randomised(ctx):
    """Continuous batching under random schedules (tiny preset): number of items 1..40, rows 1..8, budgets 1..12 frames (also a
    single frame, also fewer items than rows), every polling period, fixed lengths and live end-of-sequence.  Each item must be
    the prefix of what it produces alone with the largest budget (its RNG stream is (item id, its own frame number))."""
    from rho_tts_amd._native_model import RtSampling
    cfg = config.PRESETS["tiny"]()
    nm, _ = build(ctx, cfg, max_batch=8)
    try:
        set_voice(nm, make_voice(cfg, True))
        g = torch.Generator().manual_seed(99)
        pool_n = 40
        texts = [[int(v) for v in torch.randint(0, cfg.text_vocab - 64, (int(k),), generator=g)] for k in torch.randint(1, 9, (pool_n,), generator=g)]
        ids = [1000 + 3 * i for i in range(pool_n)]
        sp_fixed, sp_live = RtSampling(1, 0.9, 50, 1.0, 1.05), RtSampling(1, 1.5, 64, 1.0, 1.0)
        full_fixed = [nm.generate([t], [12], sp_fixed, seed=3, item_ids=[i])[0] for t, i in zip(texts, ids)]
        full_live = [nm.generate([t], [12], sp_live, seed=4, item_ids=[i], ignore_eos=False, min_frames=1)[0] for t, i in zip(texts, ids)]
        for trial in range(24):
            n = int(torch.randint(1, pool_n + 1, (1,), generator=g))
            pick = torch.randperm(pool_n, generator=g)[:n].tolist()
            frames = [int(v) for v in torch.randint(1, 13, (n,), generator=g)]
            rows = int(torch.randint(1, 9, (1,), generator=g))
            every = (1, 2, 4, 7)[trial % 4]
            nm.lib.rt_debug_tune(1700 + every, 0)
            got = nm.generate([texts[i] for i in pick], frames, sp_fixed, seed=3, item_ids=[ids[i] for i in pick], max_rows=rows)
            assert nm.generate_stats()["rows"] == min(rows, n)
            for i, f, c in zip(pick, frames, got):
                assert torch.equal(c, full_fixed[i][:f]), (trial, n, rows, every, i)
            got = nm.generate([texts[i] for i in pick], frames, sp_live, seed=4, item_ids=[ids[i] for i in pick], max_rows=rows,
                              ignore_eos=False, min_frames=1)
            for i, f, c in zip(pick, frames, got):
                assert torch.equal(c, full_live[i][:f]), (trial, n, rows, every, i, "live")
    finally:
        nm.lib.rt_debug_tune(1704, 0)
        nm.close()
