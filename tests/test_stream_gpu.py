"""Sub-segment streaming (reference: BaseTTS.stream, base_tts.py:1132-1190, which yields whole segments - the
piece-wise hand-over is this build's extension): the resumable generation (rt_generate_begin / _step / _peek / _end) gives the
codes of the one-call rt_generate however the frames are cut, the chunk-wise vocoder equals the oracle's chunked decode, and
rt_stream_chunk equals its oracle (oracle/postprocess.py stream_chunk, built from the leaves pinned by the reference fixtures)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import postprocess as OP
from oracle.model import OracleModel
from rho_tts_amd import _native, config

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from rho_tts_amd.engine import Engine
    cfg = config.small()
    e = Engine(cfg=cfg, model_path=cfg.name, max_batch=4, synthetic=True)
    e.set_builtin_voice("vivian", "english")
    yield e
    e.close()


def _clip(n, lead, tail, seed):
    g = np.random.default_rng(seed)
    t = np.arange(n) / 24000.0
    x = 0.2 * np.sin(2 * np.pi * 173.0 * t) * (0.6 + 0.4 * np.sin(2 * np.pi * 3.0 * t)) + 0.01 * g.standard_normal(n) + 0.004
    x[:lead] = 1e-5 * g.standard_normal(lead)
    if tail:
        x[-tail:] = 1e-5 * g.standard_normal(tail)
    return x.astype(np.float32)


@pytest.mark.parametrize("cuts", [[9000, 31000], [23040], [], [400, 700, 1000]])
def test_stream_chunk_matches_oracle(eng, cuts):
    """First / middle / last chunks of one segment: lengths exact, samples to 3e-6, the carried (dc, gain) to 1e-12 relative."""
    x = _clip(48000, 2600, 3100, 5)
    if cuts == [400, 700, 1000]:                      # chunks shorter than two fades, first chunk all silence
        x = _clip(1300, 500, 0, 6)
    p_gpu = _native.make_post_params(sample_rate=24000, stages=0)
    p_cpu = OP.PostParams(sample_rate=24000)
    edges = [0] + cuts + [x.shape[0]]
    st_g, st_c = [0.0, 1.0], [0.0, 1.0]
    started, silent_firsts = False, 0
    for k in range(len(edges) - 1):
        piece = x[edges[k]: edges[k + 1]]
        first, last = not started, k == len(edges) - 2            # the caller's protocol: `first` until a chunk has measured the gain
        got = eng.ctx.stream_chunk(p_gpu, torch.from_numpy(piece).cuda(), st_g, first, last).cpu()
        want = OP.stream_chunk(piece, p_cpu, st_c, first, last)
        assert got.shape == want.shape, (k, got.shape, want.shape)
        assert abs(st_g[1] - st_c[1]) <= 1e-9 * abs(st_c[1]) and abs(st_g[0] - st_c[0]) <= 1e-7
        if want.numel():
            assert float((got - want).abs().max()) < 3e-6, k
        if first and st_c[1] > 0.0:
            started = True
        elif first:
            silent_firsts += 1                                     # lead-in silence: gain 0 = "measure again on the next chunk"
            assert st_g[1] == 0.0 and got.numel() <= p_cpu.window
    assert st_c[1] > 0.0 and started
    assert silent_firsts == (1 if cuts == [400, 700, 1000] else 0)
    # the gain is that of the chunk's AUDIBLE span: lead-in silence in the first chunk does not raise it
    if cuts == [9000, 31000]:
        voiced_rms = float(np.sqrt(np.mean(x[2600:9000].astype(np.float64) ** 2)))
        whole_rms = float(np.sqrt(np.mean(x[:9000].astype(np.float64) ** 2)))
        g_voiced, g_whole = 10 ** (-23.0 / 20) / voiced_rms, 10 ** (-23.0 / 20) / whole_rms
        assert abs(st_g[1] / g_voiced - 1.0) < 0.03 and g_whole / g_voiced > 1.15


@pytest.mark.parametrize("first,step", [(1, 1), (3, 5), (12, 36), (7, 1000)])
def test_resumable_generation_equals_one_call(eng, first, step):
    """The codes do not depend on how the frames are cut into steps - fixed lengths and live end-of-sequence, greedy and sampled."""
    text = "people think about the history of water and power every single day"
    for live in (False, True):
        eng.ignore_eos = not live
        eng.params.temperature = 1.5 if live else 0.9
        eng.params.top_k = 64 if live else 50
        try:
            whole = eng.generate_codes([text], seed=11, item_ids=[7], max_frames=[40])[0]
            parts = [c for c, _ in eng.stream_codes(text, seed=11, item_id=7, first_chunk=first, chunk=step, max_frames=40)]
            lasts = [l for _, l in eng.stream_codes(text, seed=11, item_id=7, first_chunk=first, chunk=step, max_frames=40)]
        finally:
            eng.ignore_eos = None
            eng.params.temperature, eng.params.top_k = 0.9, 50
        got = torch.cat(parts)
        assert torch.equal(got, whole), (live, got.shape, whole.shape)
        assert lasts[-1] and not any(lasts[:-1])
        if not live:
            assert whole.shape[0] == 40 and parts[0].shape[0] == min(first, 40)


def test_run_in_flight_survives_other_calls_and_is_dropped_cleanly(eng):
    """Between two steps the vocoder (same pool, same stream) may run; a run that is never ended is dropped by the next begin /
    generate; a voice change while a run is in flight is refused."""
    text = "a short line of text"
    ids = eng.tokenizer.encode(text)
    whole = eng.generate_codes([text], seed=3, item_ids=[0], max_frames=[20])[0]
    eng.model.generate_begin([ids], [20], eng.params.talker(), eng.params.predictor(), seed=3, item_ids=[0])
    run, done = eng.model.generate_step(6)
    assert run == 6 and not done
    c1, fin = eng.model.generate_peek(0, 0, 20)
    assert c1.shape[0] == 6 and not fin and torch.equal(c1, whole[:6])
    w = eng.model.code2wav([c1])[0]                                   # another entry point using the pool in between
    assert w.numel() == eng.model.wav_length(6)
    # a launch-plan switch while a generation is in flight is REFUSED (the run keeps the plan it began with) ...
    assert eng.ctx.lib.rt_debug_tune(200, 0) == _native.RT_ERR_STATE
    with pytest.raises(RuntimeError):
        eng.set_builtin_voice("ryan", "english")                      # RT_ERR_STATE
    run, done = eng.model.generate_step(1000)
    assert run == 20 and done
    out = eng.model.generate_end(collect=True)
    assert torch.equal(out[0], whole)
    # abandoned run, then a fresh one-call generate: the same codes again
    eng.model.generate_begin([ids], [20], eng.params.talker(), eng.params.predictor(), seed=3, item_ids=[0])
    eng.model.generate_step(4)
    assert torch.equal(eng.generate_codes([text], seed=3, item_ids=[0], max_frames=[20])[0], whole)
    with pytest.raises(RuntimeError):
        eng.model.generate_step(1)                                    # nothing in flight any more
    eng.model.generate_end()                                          # harmless
    assert eng.ctx.lib.rt_debug_tune(201, 0) == 0                     # ... and accepted once nothing is in flight


def test_tuning_from_another_thread_never_lands_inside_a_call(eng):
    """rt_debug_tune is process-wide; a thread that flips a switch while another thread generates must not
    change that call's plan mid-way (the switch waits for calls in flight) - the codes stay those of an undisturbed run."""
    import threading
    text = "a thread generates while another one keeps flipping switches"
    whole = eng.generate_codes([text], seed=9, item_ids=[2], max_frames=[24])[0]
    stop = threading.Event()
    flips = [0]

    def flip():
        lib = eng.ctx.lib
        while not stop.is_set():
            for code in (200, 201, 700, 701, 800, 801):               # eager / graph frames, 32- / 64-row launches, fused sampler off / on
                lib.rt_debug_tune(code, 0)
                flips[0] += 1
    th = threading.Thread(target=flip, daemon=True)
    th.start()
    try:
        for _ in range(6):
            assert torch.equal(eng.generate_codes([text], seed=9, item_ids=[2], max_frames=[24])[0], whole)
    finally:
        stop.set()
        th.join(timeout=10)
        for code in (201, 701, 801):
            assert eng.ctx.lib.rt_debug_tune(code, 0) == 0
    assert flips[0] > 0


def test_streamed_waveform_equals_oracle_chunked_decode(eng):
    """stream_wav: every batch of new frames vocoded with left context - the codec decoder's chunked decode with the chunk
    boundaries where the frames arrived.  Against the float32 oracle decoding the same slices: RMSE < 1e-3."""
    from rho_tts_amd.weights import synthetic_state
    cfg = eng.cfg
    om = OracleModel(cfg, {k: v for k, v in synthetic_state(cfg, 789).items() if k.startswith("codec.")})
    text = "one two three four five six seven eight nine ten eleven twelve"
    chunks = [(w.cpu(), last) for w, last in eng.stream_wav(text, seed=5, item_id=1, first_chunk=5, chunk=9, max_frames=30)]
    codes = eng.generate_codes([text], seed=5, item_ids=[1], max_frames=[30])[0]
    assert chunks[-1][1] and len(chunks) == 4                          # 5 + 9 + 9 + 7 frames
    c = cfg.codec
    q, up = c.num_quantizers, c.total_upsample
    start, want, emitted = 0, [], 0
    for n in (5, 9, 9, 7):
        ctx = c.left_context_frames if start - c.left_context_frames > 0 else start
        with torch.no_grad():
            w = om.code2wav(codes[start - ctx: start + n, :q].T[None])[0]
        want.append(w[max(0, emitted - (start - ctx) * up):])     # by absolute sample position: no gap at the boundaries
        emitted += want[-1].numel()
        start += n
    for (g, _), w in zip(chunks, want):
        assert g.shape == w.shape
        assert float(torch.sqrt(torch.mean((g - w) ** 2))) < 1e-3
    assert float(torch.cat(want).abs().max()) > 0.01
    # the pieces played back to back ARE the segment: as many samples as one decode of all 30 frames gives, and the same waveform
    # up to what the bounded left context changes (the decoder's receptive field reaches further back than the context at a cut)
    with torch.no_grad():
        whole = om.code2wav(codes[:, :q].T[None])[0]                  # (one decode of all 30 frames: more than the engine's chunk)
    got = torch.cat([g for g, _ in chunks])
    assert got.numel() == whole.numel() == eng.model.wav_length(30)
    # (measured on the `small` test codec, 4 frames of left context under seeded weights: RMSE 0.04 at a peak of 0.78 - the cuts are
    #  audible in the numbers, not gaps in the timeline; the real decoder keeps 25 frames of context)
    assert float(torch.sqrt(torch.mean((got - whole) ** 2))) < 0.1 * float(whole.abs().max())
    # ... and exactly the same where no cut can reach: the decoder is causal, so the first piece (minus its last two frames, which the
    # one-decode form computes with right-hand neighbours in the transposed convs' overlap) is the segment's own beginning
    first_n = chunks[0][0].numel()
    assert float((got[:first_n - 2 * up] - whole[:first_n - 2 * up]).abs().max()) < 1e-3


def test_provider_streams_sub_segment_chunks():
    """stream() with stream_chunk_frames set: several results per segment, first after `stream_chunk_frames` frames, the pieces
    of a segment adding up to (almost) the segment; default (0) keeps the reference's one result per segment."""
    from rho_tts_amd.provider import MI355XQwenTTS
    t = MI355XQwenTTS(device="cuda", speaker="Vivian", model_path="x/CustomVoice-small", batch_size=4)
    try:
        text = "First sentence of eight words is right here. Second one is a little bit shorter."
        t.force_sentence_split = True
        per_segment = list(t.stream(text))
        assert len(per_segment) == 2
        t.stream_chunk_frames, t.stream_next_chunk_frames = 6, 10
        pieces = list(t.stream(text))
        assert len(pieces) > 4 and all(p.audio.numel() > 0 and p.sample_rate == t.sample_rate for p in pieces)
        tot = sum(p.duration_sec for p in pieces)
        ref = sum(p.duration_sec for p in per_segment)
        assert 0.8 * ref < tot < 1.05 * ref, (tot, ref)
        assert pieces[0].duration_sec < 6 * 0.09                          # the first piece is the first 6 frames (minus leading silence)
        # cancellation between two pieces ends the stream
        from rho_tts_amd.api import CancellationToken
        tok = CancellationToken()
        got = []
        for p in t.stream(text, cancellation_token=tok):
            got.append(p)
            tok.cancel()
        assert len(got) == 1
        assert len(list(t.stream("Still works after a cancelled stream."))) >= 1
    finally:
        t.close()
