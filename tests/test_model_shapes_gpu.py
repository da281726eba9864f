"""Value checks at the shapes bench.py runs (BASELINE.json configs[1] and configs[2]), through the C ABI, against the CPU
oracle with the bf16 rounding points of a bf16 model (oracle/model.py ``act_bf16``; PARITY UNPINNED against qwen-tts itself,
which is absent - see the oracle's header):

  C3  Qwen3-TTS-1.7B shapes, batch 32, 30-s reference clone (460-row shared prefix), 10-word sentences
  C2  Qwen3-TTS-0.6B shapes, batch 8, same prompt
  codec decoder at the real dimensions (1024-wide pre-transformer, 1536-channel decoder, rates 8/5/4/3), 44 frames

How tight can teacher-forced logits be held?  Measured on MI355X (tools/diag_logits.py, 1.7B, batch 32): the oracle with
bf16 rounding at the SAME points as the HIP path is no closer to the GPU (rms 0.74 % of the logits' sigma, max 3.3 %) than
the float32 oracle is (0.72 % / 3.6 %), and the two oracles differ from EACH OTHER by the same amount (0.71 % / 3.1 %).
Rounding to bf16 is discontinuous: a 1e-6 difference in summation order flips the rounding of a few GEMM inputs, each
flip injects a full bf16 ulp, and after a handful of the 28 x 4 rounding points two implementations hold decorrelated
rounding realisations.  So the floor for ANY two bf16 implementations is the bf16-vs-f32 distance itself, and the test is
self-calibrating: the GPU must be as close to the bf16 oracle as the bf16 oracle is to the f32 oracle (x RMS_SLACK on the
rms, x MAX_SLACK on the max over ~10^5 logits), with absolute caps on top.  A systematic error (wrong scale, dropped
residual, wrong position) moves the rms by multiples of the floor and fails; what this CANNOT see is one dropped key in
a 460-row context (0.2 % of sigma) - that is pinned exactly, at kernel level, by
tests/test_gemm_col_gpu.py::test_fused_attention_key_census.
Waveform RMSE < 1e-3 is BASELINE.json's stated bar.
"""
import os

import pytest
import torch

from oracle.model import OracleModel, Voice
from oracle.sampling import SamplingParams
from rho_tts_amd import config, weights

pytestmark = pytest.mark.gpu
torch.set_num_threads(min(32, os.cpu_count() or 8))

RMS_SLACK, MAX_SLACK = 1.3, 1.6          # GPU-vs-oracle distance as a multiple of the bf16-vs-f32 oracle distance
RMS_CAP, MAX_CAP = 0.012, 0.08            # absolute caps, in units of the logits' standard deviation

WORDS = ("time year people way day man thing woman life child world school state family student group country problem hand part "
         "place case week company system program question work government number night point home water room mother area").split()


@pytest.fixture(scope="module")
def ctx():
    from rho_tts_amd import _native
    c = _native.Context(0)
    yield c
    c.close()


def clone_voice(cfg, tok, seconds=30.0):
    from rho_tts_amd.voice import synthetic_reference_clip
    from tests.fake_voice import conditioning_from_audio
    clip = synthetic_reference_clip(seconds, cfg.sample_rate, 789)
    ref_text = " ".join(WORDS[i % len(WORDS)] for i in range(75))
    return conditioning_from_audio(cfg, clip, tok.encode(ref_text), "english", max_frames=cfg.max_positions // 2)


def sentences(n, n_words, seed):
    g = torch.Generator().manual_seed(seed)
    return [" ".join(WORDS[int(j)] for j in torch.randint(0, len(WORDS), (n_words,), generator=g)).capitalize() + "." for _ in range(n)]


@pytest.mark.parametrize("preset,B,n_frames", [("1.7b", 32, 2), ("0.6b", 8, 3)])
def test_teacher_forced_logits_at_bench_shapes(ctx, preset, B, n_frames):
    from rho_tts_amd._native_model import NativeModel, RtSampling
    from rho_tts_amd.tokenizer import HashTokenizer
    cfg = config.PRESETS[preset]()
    tok = HashTokenizer(cfg.text_vocab)
    state = weights.synthetic_state(cfg, 789, device="cuda")
    nm = NativeModel(ctx, cfg, max_batch=B)
    try:
        nm.load_state(state)
        cpu_state = {k: v.cpu() for k, v in state.items()}
        om, om32 = OracleModel(cfg, cpu_state, act_bf16=True), OracleModel(cfg, cpu_state)
        del state
        torch.cuda.empty_cache()
        cond = clone_voice(cfg, tok)
        v = Voice(cond.language, None, cond.speaker_embed, cond.ref_text_ids, cond.ref_codes)
        n_prefix = nm.set_voice(v.language, None, v.speaker_embed, v.ref_text_ids, v.ref_codes)
        assert n_prefix == 460                                       # 3 role + 4 control + speaker + bos + 75 words + codec_bos + 375 frames
        texts = [tok.encode(t) for t in sentences(B, 10, 789)]
        frames = [n_frames] * B
        tr_o, tr_32 = {}, {}
        with torch.no_grad():
            free = om.generate(v, texts, frames, SamplingParams(), trace=tr_o, share_prefix=True)       # oracle's greedy trajectory
            om32.generate(v, texts, frames, SamplingParams(), trace=tr_32, share_prefix=True, forced_codes=free)
        codes, tr = nm.generate(texts, frames, RtSampling(0, 1.0, 1, 1.0, 1.0), forced_codes=free, trace=True)
        assert all(torch.equal(a, b) for a, b in zip(codes, free))
        V0, G1 = cfg.codec.codebook_size, cfg.n_groups - 1

        def dist(x, y, sig):
            e = (x - y).abs()
            return float(e.pow(2).mean().sqrt()) / sig, float(e.max()) / sig

        for name, o16, o32, gpu in (
                ("talker", torch.stack(tr_o["talker_logits"])[..., :V0], torch.stack(tr_32["talker_logits"])[..., :V0],
                 tr["talker"][:n_frames].cpu()[..., :V0]),
                ("predictor", torch.stack(tr_o["pred_logits"]).view(n_frames, G1, B, -1), torch.stack(tr_32["pred_logits"]).view(n_frames, G1, B, -1),
                 tr["predictor"][:n_frames].cpu())):
            sig = float(o16.std())
            floor_rms, floor_max = dist(o16, o32, sig)
            rms, mx = dist(gpu, o16, sig)
            print(f"\n{cfg.name} B={B} {name}: GPU vs bf16 oracle rms {rms:.5f} max {mx:.5f} sigma; bf16 vs f32 oracle rms {floor_rms:.5f} max {floor_max:.5f}")
            assert rms <= RMS_SLACK * floor_rms and rms <= RMS_CAP, (name, rms, floor_rms)
            assert mx <= MAX_SLACK * floor_max and mx <= MAX_CAP, (name, mx, floor_max)
            assert abs(float((gpu - o16).mean())) / sig < 2e-4          # no systematic offset
        # greedy free-running decode lands on the oracle's codes for the first frame (no compounding yet) almost everywhere
        got = nm.generate(texts, frames, RtSampling(0, 1.0, 1, 1.0, 1.0))
        agree = sum(float((a[0] == b[0]).float().mean()) for a, b in zip(got, free)) / B
        assert agree >= 0.9, agree
    finally:
        nm.close()


def test_code2wav_at_real_codec_dimensions(ctx):
    """The codec decoder of the 1.7B / 0.6B presets (they share it): 16 codebooks x 2048, 1024-wide 8-layer pre-transformer
    (window 72), 2x ConvNeXt upsampling, 1536-channel decoder with rates 8/5/4/3 - 44 frames as in bench.py, and a shorter item
    beside it in the same launch.  RMSE < 1e-3 against the float32 oracle."""
    from rho_tts_amd._native_model import NativeModel
    cfg = config.PRESETS["0.6b"]()
    state = weights.synthetic_state(cfg, 789, device="cuda")
    nm = NativeModel(ctx, cfg, max_batch=2, max_positions=256)
    try:
        nm.load_state(state)
        om = OracleModel(cfg, {k: v.cpu() for k, v in state.items() if k.startswith("codec.")})
        del state
        torch.cuda.empty_cache()
        g = torch.Generator().manual_seed(9)
        Q = cfg.codec.num_quantizers
        codes = [torch.randint(0, cfg.codec.codebook_size, (n, Q), generator=g) for n in (44, 29)]
        wavs = nm.code2wav(codes)
        for c, w in zip(codes, wavs):
            with torch.no_grad():
                ref = om.code2wav(c.T[None])[0]
            assert w.shape[0] == ref.shape[0] == nm.wav_length(c.shape[0])
            rmse = float(torch.sqrt(torch.mean((w.cpu() - ref) ** 2)))
            print(f"\\ncode2wav {c.shape[0]} frames: rmse {rmse:.2e}, ref rms {float(ref.pow(2).mean().sqrt()):.3f}")
            assert rmse < 1e-3, rmse
            assert float(ref.abs().max()) > 0.05
        # the waveform of an item does not depend on what it is vocoded with: alone == beside a longer item, bit for bit
        # (no split or tile choice in the decoder depends on the number of rows)
        for c, w in zip(codes, wavs):
            assert torch.equal(nm.code2wav([c])[0], w)
        # the 96-channel residual units run their k = 7 conv and 1x1 conv as ONE launch (k_conv_win<..., FUSE>); as two launches
        # through hi / lo planes in HBM (rt_debug_tune 2100) the same planes are multiplied in another MFMA order: float rounding apart
        nm.lib.rt_debug_tune(2100, 0)
        try:
            two = nm.code2wav(codes)
        finally:
            nm.lib.rt_debug_tune(2101, 0)
        for a, b in zip(two, wavs):
            d = float((a - b).abs().max())
            print(f"\nfused vs two-launch conv pairs: max diff {d:.2e}")
            assert d < 2e-5 and not torch.equal(a, torch.zeros_like(a))
    finally:
        nm.close()


def test_continuous_batching_at_the_bench_shape(ctx):
    """rt_generate with more items than rows on the 1.7B preset with the 460-row clone prefix: 44 ragged items on 32 rows.
    Every item - the 32 that start on the rows and the 12 that take over finished rows after a hand-over prefill of a few rows -
    must come out bit for bit as it does ALONE in a one-item call, and as the first 32 do in a static batch: the prompt
    prefill gives a row the same float32 sums whatever it is batched with (k_gemm_mid adds K in the skinny kernel's
    segments), decode rows never see each other, and the RNG stream is (item id, the item's own frame number)."""
    from rho_tts_amd._native_model import NativeModel, RtSampling
    from rho_tts_amd.tokenizer import HashTokenizer
    cfg = config.PRESETS["1.7b"]()
    tok = HashTokenizer(cfg.text_vocab)
    state = weights.synthetic_state(cfg, 789, device="cuda")
    nm = NativeModel(ctx, cfg, max_batch=32)
    try:
        nm.load_state(state)
        del state
        torch.cuda.empty_cache()
        cond = clone_voice(cfg, tok)
        nm.set_voice(cond.language, None, cond.speaker_embed, cond.ref_text_ids, cond.ref_codes)
        g = torch.Generator().manual_seed(41)
        n = 44
        texts = [tok.encode(t) for k in torch.randint(3, 14, (n,), generator=g) for t in sentences(1, int(k), int(k) + 1000)]
        frames = [int(v) for v in torch.randint(3, 13, (n,), generator=g)]
        ids = list(range(500, 500 + n))
        sp = RtSampling(1, 0.9, 50, 1.0, 1.05)
        static = nm.generate(texts[:32], frames[:32], sp, seed=9, item_ids=ids[:32])
        got = nm.generate(texts, frames, sp, seed=9, item_ids=ids)
        st = nm.generate_stats()
        assert [c.shape[0] for c in got] == frames and st["rows"] == 32 and st["hand_overs"] >= 1
        assert all(torch.equal(a, b) for a, b in zip(got[:32], static))
        for i in list(range(0, 32, 5)) + list(range(32, n)):
            alone = nm.generate([texts[i]], [frames[i]], sp, seed=9, item_ids=[ids[i]])[0]
            assert torch.equal(got[i], alone), i
        # a first wave of MORE than 1024 prompt rows (32 texts of 38-44 tokens: the normal case for real segments) - the prefill
        # goes down in chunks of <= 1024 rows on the same kernel, so every item still equals itself alone
        long_texts = [tok.encode(t) for k in torch.randint(38, 45, (32,), generator=g) for t in sentences(1, int(k), int(k) + 2000)]
        assert sum(len(t) + 2 for t in long_texts) > 1024
        many = nm.generate(long_texts, [3] * 32, sp, seed=11, item_ids=ids[:32])
        for i in (0, 13, 31):
            assert torch.equal(many[i], nm.generate([long_texts[i]], [3], sp, seed=11, item_ids=[ids[i]])[0]), i
    finally:
        nm.close()


def test_batch_invariance_and_64_rows_at_the_0p6b_shape(ctx):
    """the 0.6B down-projection has K = 3072 - 192 k-tiles, which the skinny kernel used to cut into 32 segments of 6
    (not a whole number of the prompt-prefill kernel's 64-deep steps, so the two kernels summed K differently and an item
    prefilled among many rows could differ from itself alone).  Items in a 32-row first wave of > 1024 prompt rows, in a
    13-row wave (skinny kernel) and alone must produce the same codes, bit for bit."""
    from rho_tts_amd._native_model import NativeModel, RtSampling
    from rho_tts_amd.tokenizer import HashTokenizer
    cfg = config.PRESETS["0.6b"]()
    tok = HashTokenizer(cfg.text_vocab)
    state = weights.synthetic_state(cfg, 789, device="cuda")
    nm = NativeModel(ctx, cfg, max_batch=64, max_positions=1024)
    try:
        nm.load_state(state)
        del state
        torch.cuda.empty_cache()
        nm.set_voice("english", "vivian", None, [], None)
        g = torch.Generator().manual_seed(77)
        texts = [tok.encode(t) for k in torch.randint(36, 46, (32,), generator=g) for t in sentences(1, int(k), int(k) + 3000)]
        assert sum(len(t) + 2 for t in texts) > 1024
        ids = list(range(900, 932))
        sp = RtSampling(1, 0.9, 50, 1.0, 1.05)
        full = nm.generate(texts, [4] * 32, sp, seed=5, item_ids=ids)
        few = nm.generate(texts[:2], [4] * 2, sp, seed=5, item_ids=ids[:2])           # 2 x ~42 rows: k_gemm_mid with few rows
        assert all(torch.equal(a, b) for a, b in zip(few, full[:2]))
        for i in (0, 9, 31):
            alone = nm.generate([texts[i]], [4], sp, seed=5, item_ids=[ids[i]])[0]       # <= 64 rows: the skinny kernel
            assert torch.equal(full[i], alone), i
        # the 64-row column path END TO END at real dimensions (H = 1024, predictor 1024: the talker's GEMMs
        # carry 64 rows per launch, the predictor's two-position first pass runs as two 64-row launches) - 50 ragged items in
        # one static batch, a sample of them equal to themselves alone
        short = [tok.encode(t) for k in torch.randint(2, 9, (50,), generator=g) for t in sentences(1, int(k), int(k) + 4000)]
        fr = [int(v) for v in torch.randint(2, 6, (50,), generator=g)]
        wide = nm.generate(short, fr, sp, seed=6, item_ids=list(range(50)))
        assert nm.generate_stats()["rows"] == 50 and [c.shape[0] for c in wide] == fr
        for i in (0, 7, 16, 31, 32, 33, 41, 49):
            assert torch.equal(wide[i], nm.generate([short[i]], [fr[i]], sp, seed=6, item_ids=[i])[0]), i
    finally:
        nm.close()


def test_whole_sequence_at_c3_teacher_forced_and_free_running(ctx):
    """the bench shape END TO END in time - 1.7B, 32 rows on the GPU, 460-row clone prefix, all 44 frames (positions
    473 .. 517: RoPE far behind the prefix, repetition history filling up).  The oracle runs the first 8 items (its cost, not the
    GPU's, sets the slice); the GPU decodes all 32 rows teacher-forced on the oracle's greedy trajectory.  Checked: the
    self-calibrated logit bound at frames 0, 22 and 43 and over all frames; per frame, the share of codes whose GPU argmax under
    teacher forcing is the oracle's code; and the FREE-running greedy decode against the oracle's (compounding: one differing
    code changes everything behind it - reported as a curve, asserted where a floor exists: the first two frames)."""
    from rho_tts_amd._native_model import NativeModel, RtSampling
    from rho_tts_amd.tokenizer import HashTokenizer
    cfg = config.PRESETS["1.7b"]()
    B, NO, T = 32, 8, 44
    tok = HashTokenizer(cfg.text_vocab)
    state = weights.synthetic_state(cfg, 789, device="cuda")
    nm = NativeModel(ctx, cfg, max_batch=B)
    try:
        nm.load_state(state)
        cpu_state = {k: v.cpu() for k, v in state.items()}
        om, om32 = OracleModel(cfg, cpu_state, act_bf16=True), OracleModel(cfg, cpu_state)
        del state
        torch.cuda.empty_cache()
        cond = clone_voice(cfg, tok)
        v = Voice(cond.language, None, cond.speaker_embed, cond.ref_text_ids, cond.ref_codes)
        assert nm.set_voice(v.language, None, v.speaker_embed, v.ref_text_ids, v.ref_codes) == 460
        texts = [tok.encode(t) for t in sentences(B, 10, 789)]
        tr_o, tr_32 = {}, {}
        with torch.no_grad():
            free = om.generate(v, texts[:NO], [T] * NO, SamplingParams(), trace=tr_o, share_prefix=True)
            om32.generate(v, texts[:NO], [T] * NO, SamplingParams(), trace=tr_32, share_prefix=True, forced_codes=free)
        forced = list(free) + [free[b % NO] for b in range(NO, B)]          # rows 8..31 ride along (rows never see each other)
        greedy = RtSampling(0, 1.0, 1, 1.0, 1.0)
        codes, tr = nm.generate(texts, [T] * B, greedy, forced_codes=forced, trace=True)
        assert all(torch.equal(a, b) for a, b in zip(codes, forced))
        V0, G1 = cfg.codec.codebook_size, cfg.n_groups - 1
        t16, t32 = torch.stack(tr_o["talker_logits"])[..., :V0], torch.stack(tr_32["talker_logits"])[..., :V0]          # [T, NO, V0]
        p16 = torch.stack(tr_o["pred_logits"]).view(T, G1, NO, -1)
        p32 = torch.stack(tr_32["pred_logits"]).view(T, G1, NO, -1)
        tg = tr["talker"][:T, :NO, :V0].cpu()
        pg = tr["predictor"][:T, :, :NO].cpu()

        def dist(x, y, sig):
            e = (x - y).abs()
            return float(e.pow(2).mean().sqrt()) / sig, float(e.max()) / sig

        for name, o16, o32, gpu in (("talker", t16, t32, tg), ("predictor", p16, p32, pg)):
            sig = float(o16.std())
            for label, sl in (("frame 0", slice(0, 1)), ("frame 22", slice(22, 23)), ("frame 43", slice(43, 44)), ("all 44 frames", slice(0, T))):
                floor_rms, floor_max = dist(o16[sl], o32[sl], sig)
                rms, mx = dist(gpu[sl], o16[sl], sig)
                print(f"\n1.7B B=32 {name} {label}: GPU vs bf16 oracle rms {rms:.5f} max {mx:.5f} sigma; bf16 vs f32 oracle rms {floor_rms:.5f} max {floor_max:.5f}")
                assert rms <= RMS_SLACK * floor_rms and rms <= RMS_CAP, (name, label, rms, floor_rms)
                assert mx <= MAX_SLACK * floor_max and mx <= MAX_CAP, (name, label, mx, floor_max)
            assert abs(float((gpu - o16).mean())) / sig < 2e-4
        # teacher-forced choice agreement per frame: GPU argmax == the oracle's code (group 0 over the codebook ids, groups 1..15)
        want = torch.stack(list(free))                                          # [NO, T, G]
        a0 = (tg.argmax(-1).T == want[:, :, 0]).float()                         # [NO, T]
        ag = (pg.argmax(-1).permute(2, 0, 1) == want[:, :, 1:]).float()         # [NO, T, G1]
        tf_curve = ((a0 + ag.sum(-1)) / cfg.n_groups).mean(0)                   # per frame
        print("\nteacher-forced argmax agreement per frame:", " ".join(f"{float(x):.3f}" for x in tf_curve))
        assert float(tf_curve.min()) >= 0.93 and float(tf_curve[[0, 22, 43]].min()) >= 0.93, tf_curve
        # free-running greedy decode of the same 32 rows against the oracle's own trajectory
        got = nm.generate(texts, [T] * B, greedy)
        eq = torch.stack([(got[b] == free[b]).float().mean(-1) for b in range(NO)])          # [NO, T] share of the 16 codes
        curve = eq.mean(0)
        print("free-running code agreement per frame:", " ".join(f"{float(x):.3f}" for x in curve))
        # Measured on MI355X: 1.000 0.938 0.703 0.320 0.258 0.203 0.188 0.117 0.04 ... 0.0 at frame 43.  Greedy decoding of SEEDED
        # weights is chaotic - the logits of a frame are nearly flat, 2-5 % of the argmax choices sit inside the bf16 noise (the
        # teacher-forced curve above), and one differing code changes every later frame - so two correct bf16 implementations part
        # within a few frames and the free-running curve has no floor to assert at frame 43 (it is 0 there for ANY pair, the bf16
        # and float32 oracles included).  What holds at every frame is the teacher-forced agreement; free-running, only the start.
        assert float(curve[0]) >= 0.9 and float(curve[1]) >= 0.75, curve
    finally:
        nm.close()


def test_c2_through_the_provider_end_to_end(ctx, tmp_path, monkeypatch):
    """BASELINE.json configs[1] driven the way a user drives it: ``MI355XQwenTTS(model_path=...0.6B-Base, batch_size=8).generate()``
    with a 30-s reference WAV on disk and validation off (max_iterations = 1), sampled decoding.  Two of the eight delivered
    waveforms are rebuilt on the CPU from the codes the GPU chose - oracle code2wav -> oracle post-processing - and must agree to
    RMSE < 1e-3 END TO END (provider, engine, decode, codec decoder, fused post-processing, hand-over); the oracle, teacher-forced
    on those codes, must make the same draws (same uniforms: oracle/sampling.py) at all but the few frames where the bf16 logit
    distance moves a boundary of the inverse CDF."""
    import wave

    import numpy as np

    from oracle import postprocess as OP
    from oracle.sampling import draw, uniform
    from rho_tts_amd.provider import MI355XQwenTTS
    from rho_tts_amd.voice import synthetic_reference_clip
    monkeypatch.setenv("RHO_TTS_AMD_SYNTHETIC", "1")
    cfg = config.PRESETS["0.6b"]()
    clip = synthetic_reference_clip(30.0, cfg.sample_rate, 789)
    ref = str(tmp_path / "ref.wav")
    with wave.open(ref, "wb") as wf:
        wf.setnchannels(1); wf.setsampwidth(2); wf.setframerate(cfg.sample_rate)
        wf.writeframes((np.clip(clip, -1, 1) * 32767).astype("<i2").tobytes())
    ref_text = " ".join(WORDS[i % len(WORDS)] for i in range(75))
    texts = sentences(8, 10, 4242)
    t = MI355XQwenTTS(device="cuda", reference_audio=ref, reference_text=ref_text, model_path="Qwen/Qwen3-TTS-12Hz-0.6B-Base",
                      batch_size=8, max_iterations=1)
    try:
        res = t.generate(texts)
        assert res is not None and len(res) == 8 and all(r is not None and r.audio.numel() > 24000 for r in res)
        eng = t._load_engine()
        assert eng.cfg.name == cfg.name and eng.model.prefix_len() == 460
        codes = eng.generate_codes(texts, seed=int(t.seed), item_ids=list(range(8)))       # the codes that call decoded (same seed, same streams)
        state = {k: v.cpu() for k, v in weights.synthetic_state(cfg, 789, device="cuda").items()}
        om = OracleModel(cfg, state, act_bf16=True)
        del state
        Q = cfg.codec.num_quantizers
        post = OP.PostParams(sample_rate=cfg.sample_rate)
        for b in (0, 5):
            with torch.no_grad():
                wav = om.code2wav(codes[b][:, :Q].T[None])[0]
            y, ratio, ok = OP.finish_item([wav], post)
            got = res[b].audio.reshape(-1).cpu()
            assert got.numel() == y.numel(), (got.numel(), y.numel())
            rmse = float(torch.sqrt(torch.mean((got - y.reshape(-1)) ** 2)))
            print(f"\nC2 item {b}: delivered waveform vs oracle rebuilt from the GPU's codes: RMSE {rmse:.2e} over {got.numel()} samples, decay {ratio:.3f}")
            assert rmse < 1e-3, rmse
            assert abs(res[b].decay_ratio - ratio) < 1e-4 * max(1.0, ratio)
        # the oracle, teacher-forced on the GPU's codes, draws with the same uniforms: how often does it pick the GPU's code?
        v = Voice(eng.voice.language, None, eng.voice.speaker_embed, eng.voice.ref_text_ids, eng.voice.ref_codes)
        ids = [eng.tokenizer.encode(x) for x in (texts[0], texts[5])]
        fc = [codes[0], codes[5]]
        tr = {}
        sp_t, sp_p = eng.params.talker(), eng.params.predictor()
        from oracle.sampling import SamplingParams as SP
        st = SP(bool(sp_t.do_sample), sp_t.temperature, sp_t.top_k, sp_t.top_p, sp_t.repetition_penalty)
        spd = SP(bool(sp_p.do_sample), sp_p.temperature, sp_p.top_k, sp_p.top_p, sp_p.repetition_penalty)
        with torch.no_grad():
            om.generate(v, ids, [c.shape[0] for c in fc], st, spd, seed=int(t.seed), item_ids=[0, 5], forced_codes=fc, trace=tr, share_prefix=True)
        T = fc[0].shape[0]
        tl = torch.stack(tr["talker_logits"])                                    # [T, 2, V]
        pl = torch.stack(tr["pred_logits"]).view(T, cfg.n_groups - 1, 2, -1)
        hit = tot = 0
        for k, item in enumerate((0, 5)):
            seen = np.zeros(cfg.codec_vocab, dtype=bool)
            for f in range(T):
                c0 = draw(tl[f, k].numpy(), st, uniform(int(t.seed), item, f, 0), om.talker_suppress(False), seen)
                hit += int(c0 == int(fc[k][f, 0])); tot += 1
                seen[int(fc[k][f, 0])] = True
                for g in range(cfg.n_groups - 1):
                    cg = draw(pl[f, g, k].numpy(), spd, uniform(int(t.seed), item, f, g + 1))
                    hit += int(cg == int(fc[k][f, g + 1])); tot += 1
        print(f"oracle draws equal to the GPU's sampled codes under teacher forcing: {hit}/{tot} = {hit / tot:.4f}")
        # How often can two correct implementations agree on a DRAW?  The candidates are ordered by logit, the seeded weights give
        # nearly flat logits, and a 0.7 %-sigma perturbation swaps neighbours in that order: the float32 oracle, teacher-forced on
        # the same codes with the same uniforms, is the yardstick (measured: GPU vs bf16 oracle 0.81).
        om32 = OracleModel(cfg, {k: v.cpu() for k, v in weights.synthetic_state(cfg, 789, device="cuda").items()})
        tr32 = {}
        with torch.no_grad():
            om32.generate(v, ids, [c.shape[0] for c in fc], st, spd, seed=int(t.seed), item_ids=[0, 5], forced_codes=fc, trace=tr32, share_prefix=True)
        tl32 = torch.stack(tr32["talker_logits"])
        pl32 = torch.stack(tr32["pred_logits"]).view(T, cfg.n_groups - 1, 2, -1)
        same = 0
        for k, item in enumerate((0, 5)):
            seen = np.zeros(cfg.codec_vocab, dtype=bool)
            for f in range(T):
                u0 = uniform(int(t.seed), item, f, 0)
                same += int(draw(tl[f, k].numpy(), st, u0, om.talker_suppress(False), seen) == draw(tl32[f, k].numpy(), st, u0, om.talker_suppress(False), seen))
                seen[int(fc[k][f, 0])] = True
                for g in range(cfg.n_groups - 1):
                    ug = uniform(int(t.seed), item, f, g + 1)
                    same += int(draw(pl[f, g, k].numpy(), spd, ug) == draw(pl32[f, g, k].numpy(), spd, ug))
        print(f"bf16 oracle draws equal to the float32 oracle's: {same}/{tot} = {same / tot:.4f}")
        assert hit / tot >= same / tot - 0.06 and hit / tot >= 0.7, (hit, same, tot)
    finally:
        t.close()
