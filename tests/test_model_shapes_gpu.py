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
