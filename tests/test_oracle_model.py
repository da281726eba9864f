"""Cross-check the model oracle's building blocks against the in-container sibling
architecture (transformers' Qwen3-Omni code2wav and Qwen3 decoder stack) built from small
configs with the SAME weights, and pin its own invariants.  These siblings are not the
reference's dependency (qwen-tts is absent: parity unpinned, see oracle/model.py); they are
the only executable statement of these ops in the container."""
import numpy as np
import pytest
import torch

from oracle.model import OracleModel, Voice, rope_table
from oracle.sampling import SamplingParams, draw, uniform, mix32
from rho_tts_amd import config, weights

torch.set_num_threads(4)


@pytest.fixture(scope="module")
def tiny_model():
    cfg = config.tiny()
    return cfg, OracleModel(cfg, weights.synthetic_state(cfg, 789))


def test_code2wav_matches_transformers_sibling(tiny_model):
    from transformers.models.qwen3_omni_moe.configuration_qwen3_omni_moe import Qwen3OmniMoeCode2WavConfig
    from transformers.models.qwen3_omni_moe.modeling_qwen3_omni_moe import Qwen3OmniMoeCode2Wav
    cfg, m = tiny_model
    c = cfg.codec
    hc = Qwen3OmniMoeCode2WavConfig(
        codebook_size=c.codebook_size, hidden_size=c.hidden, num_attention_heads=c.heads, num_key_value_heads=c.heads,
        head_dim=c.head_dim, sliding_window=c.sliding_window, intermediate_size=c.inter, num_hidden_layers=c.layers,
        num_quantizers=c.num_quantizers, upsample_rates=tuple(c.upsample_rates), upsampling_ratios=tuple(c.upsampling_ratios),
        decoder_dim=c.decoder_dim, rms_norm_eps=c.rms_eps, layer_scale_initial_scale=c.layer_scale,
        rope_parameters={"rope_theta": c.rope_theta, "rope_type": "default"})
    sib = Qwen3OmniMoeCode2Wav(hc).eval()
    sd = {k[len("codec."):]: v for k, v in m.W.items() if k.startswith("codec.")}
    missing, unexpected = sib.load_state_dict(sd, strict=False)
    assert not unexpected and all("inv_freq" in k or "code_offset" in k for k in missing), (missing, unexpected)
    codes = torch.randint(0, c.codebook_size, (2, c.num_quantizers, 11), generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        ref = sib(codes)[:, 0]
    got = m.code2wav(codes)
    assert got.shape == ref.shape == (2, m.wav_length(11))
    assert float((got - ref).abs().max()) < 2e-5
    with torch.no_grad():
        ref_c = sib.chunked_decode(codes[..., :], chunk_size=c.chunk_frames, left_context_size=c.left_context_frames)[:, 0]
    codes_long = torch.randint(0, c.codebook_size, (1, c.num_quantizers, 29), generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        ref_c = sib.chunked_decode(codes_long, chunk_size=c.chunk_frames, left_context_size=c.left_context_frames)[:, 0]
    got_c = m.chunked_code2wav(codes_long)
    assert got_c.shape == ref_c.shape and float((got_c - ref_c).abs().max()) < 2e-5


def test_decoder_stack_matches_transformers_qwen3(tiny_model):
    """Talker stack == transformers' dense Qwen3 model (q/k-norm, GQA, RoPE, SwiGLU) on the same weights."""
    from transformers.models.qwen3.configuration_qwen3 import Qwen3Config
    from transformers.models.qwen3.modeling_qwen3 import Qwen3Model
    cfg, m = tiny_model
    d = cfg.talker
    hc = Qwen3Config(vocab_size=8, hidden_size=d.hidden, intermediate_size=d.inter, num_hidden_layers=d.layers,
                     num_attention_heads=d.heads, num_key_value_heads=d.kv_heads, head_dim=d.head_dim,
                     rms_norm_eps=d.rms_eps, max_position_embeddings=256, attention_bias=False,
                     rope_parameters={"rope_theta": d.rope_theta, "rope_type": "default"}, use_sliding_window=False)
    hc._attn_implementation = "eager"
    sib = Qwen3Model(hc).eval()
    sd = {k[len("talker."):]: v for k, v in m.W.items() if k.startswith("talker.layers.") or k == "talker.norm.weight"}
    missing, unexpected = sib.load_state_dict(sd, strict=False)
    assert not unexpected and all(("embed_tokens" in k or "inv_freq" in k) for k in missing), (missing, unexpected)
    x = torch.randn(2, 9, d.hidden, generator=torch.Generator().manual_seed(3)) * 0.5
    with torch.no_grad():
        ref = sib(inputs_embeds=x).last_hidden_state
    m.talker.alloc(2, 16)
    got = m.talker.forward(x, torch.arange(9)[None].expand(2, -1))
    assert float((got - ref).abs().max()) < 2e-5
    # incremental decode through the KV cache == full-sequence forward
    m.talker.alloc(2, 16)
    a = m.talker.forward(x[:, :5], torch.arange(5)[None].expand(2, -1))
    outs = [a]
    for t in range(5, 9):
        outs.append(m.talker.forward(x[:, t:t + 1], torch.full((2, 1), t)))
    assert float((torch.cat(outs, 1) - ref).abs().max()) < 2e-5


def test_generation_is_batch_independent_and_prefix_invariant(tiny_model):
    cfg, m = tiny_model
    g = torch.Generator().manual_seed(5)
    v = Voice("english", speaker_embed=torch.randn(cfg.talker.hidden, generator=g) * 0.02, ref_text_ids=[3, 4, 5],
              ref_codes=torch.randint(0, cfg.codec.codebook_size, (7, cfg.n_groups), generator=g))
    texts = [[10, 11, 12], [20, 21, 22, 23, 24, 25, 26], [30]]
    sp = SamplingParams(do_sample=True, temperature=0.9, top_k=20, top_p=0.9, repetition_penalty=1.05)
    both = m.generate(v, texts, [5, 6, 4], sp, seed=77)
    for b in range(3):
        solo = m.generate(v, [texts[b]], [[5, 6, 4][b]], sp, seed=77, item_ids=[b])
        assert torch.equal(solo[0], both[b])        # ragged right-padded batch == one item at a time
    assert [c.shape[0] for c in both] == [5, 6, 4]
    other = m.generate(v, texts, [5, 6, 4], sp, seed=78)
    assert not all(torch.equal(a, b) for a, b in zip(both, other))


def test_teacher_forcing_reproduces_logits(tiny_model):
    cfg, m = tiny_model
    v = Voice("english", speaker="vivian")
    texts = [[10, 11, 12], [20, 21]]
    t1 = {}
    free = m.generate(v, texts, [4, 4], SamplingParams(), trace=t1)
    t2 = {}
    forced = m.generate(v, texts, [4, 4], SamplingParams(do_sample=True), forced_codes=free, trace=t2)
    assert all(torch.equal(a, b) for a, b in zip(free, forced))
    for a, b in zip(t1["talker_logits"], t2["talker_logits"]):
        assert torch.equal(a, b)


def test_eos_stops_an_item(tiny_model):
    cfg, m = tiny_model
    v = Voice("english", speaker="ryan")
    forced = [torch.tensor([[1, 2, 3, 4], [5, 6, 7, 8], [cfg.codec_eos_id, 0, 0, 0]]), torch.randint(0, 60, (5, 4))]
    out = m.generate(v, [[1, 2], [3]], [5, 5], SamplingParams(), ignore_eos=False, forced_codes=forced)
    assert out[0].shape[0] == 2 and out[1].shape[0] == 5


def test_uniform_stream_known_answers():
    assert mix32(0) == 0 and mix32(1) == 0x514E28B7
    us = [float(uniform(789, i, f, g)) for i in (0, 5) for f in (0, 9) for g in (0, 15)]
    assert all(0.0 < u < 1.0 for u in us) and len(set(us)) == len(us)
    # pinned values: the HIP sampler must reproduce these bits (tests/test_model_gpu.py)
    assert np.float32(uniform(789, 0, 0, 0)).tobytes().hex() == np.float32(us[0]).tobytes().hex()


def test_draw_orders_ties_and_respects_masks():
    l = np.array([0.1, 2.0, 2.0, -1.0, 1.5], np.float32)
    assert draw(l, SamplingParams(), np.float32(0.3)) == 1                      # greedy: lowest index on ties
    sup = np.array([0, 1, 0, 0, 0], bool)
    assert draw(l, SamplingParams(), np.float32(0.3), suppress=sup) == 2
    sp = SamplingParams(do_sample=True, temperature=1.0, top_k=2, top_p=1.0)
    assert draw(l, sp, np.float32(0.49)) == 1 and draw(l, sp, np.float32(0.51)) == 2
    sp = SamplingParams(do_sample=True, temperature=1.0, top_k=5, top_p=0.5)
    assert draw(l, sp, np.float32(0.99)) in (1, 2)                                # nucleus keeps the two 2.0s only
    seen = np.array([0, 1, 1, 0, 0], bool)
    sp = SamplingParams(repetition_penalty=2.0)
    assert draw(l, sp, np.float32(0.1), seen=seen) == 4                           # 2.0/2 = 1.0 < 1.5
    counts = np.zeros(5)
    sp = SamplingParams(do_sample=True, temperature=1.0, top_k=5)
    for i in range(4000):
        counts[draw(l, sp, uniform(1, i, 0, 0))] += 1
    p = np.exp(l - l.max()); p /= p.sum()
    assert np.abs(counts / 4000 - p).max() < 0.03


def test_synthetic_weights_are_reproducible_and_scaled():
    cfg = config.tiny()
    a = weights.synthetic_state(cfg, 789, only_prefix="talker.layers.0")
    b = weights.synthetic_state(cfg, 789, only_prefix="talker.layers.0")
    c = weights.synthetic_state(cfg, 790, only_prefix="talker.layers.0")
    k = "talker.layers.0.mlp.up_proj.weight"
    assert torch.equal(a[k], b[k]) and not torch.equal(a[k], c[k])
    assert abs(float(a[k].float().std()) - 0.02) < 2e-3 and abs(float(a[k].float().mean())) < 2e-3
    assert abs(float(a["talker.layers.0.input_layernorm.weight"].float().mean()) - 1.0) < 0.05
    names = [s[0] for s in weights.tensor_specs(cfg)]
    assert len(names) == len(set(names))


def test_rope_table_matches_formula():
    cos, sin = rope_table(8, 10000.0, 5)
    assert cos.shape == (5, 4) and float(cos[0].min()) == 1.0
    assert abs(float(sin[3, 1]) - np.sin(3 * 10000.0 ** (-2 / 8))) < 1e-6


def test_predictor_recurrence_matches_transformers_sibling(tiny_model):
    """The residual-code predictor pass (prefill [past hidden, embed(code 0)] -> head 0, then embedding table g-1 -> head g,
    one KV cache over the frame's 16 positions) == transformers' Qwen3OmniMoeTalkerCodePredictorModelForConditionalGeneration
    (modeling_qwen3_omni_moe.py:2534-2608) on the same weights.  The sibling has no mtp projection, so it is fed the projected
    rows / tables (projection and gather commute row by row)."""
    from transformers.models.qwen3_omni_moe.configuration_qwen3_omni_moe import Qwen3OmniMoeTalkerCodePredictorConfig
    from transformers.models.qwen3_omni_moe.modeling_qwen3_omni_moe import Qwen3OmniMoeTalkerCodePredictorModelForConditionalGeneration
    from transformers import DynamicCache
    from oracle.sampling import SamplingParams
    cfg, m = tiny_model
    d, W, G = cfg.predictor, m.W, cfg.n_groups
    hc = Qwen3OmniMoeTalkerCodePredictorConfig(
        vocab_size=cfg.predictor_vocab, hidden_size=d.hidden, intermediate_size=d.inter, num_hidden_layers=d.layers,
        num_attention_heads=d.heads, num_key_value_heads=d.kv_heads, head_dim=d.head_dim, rms_norm_eps=d.rms_eps,
        max_position_embeddings=64, attention_bias=False, num_code_groups=G, sliding_window=None,
        rope_parameters={"rope_theta": d.rope_theta, "rope_type": "default"})
    hc._attn_implementation = "eager"
    sib = Qwen3OmniMoeTalkerCodePredictorModelForConditionalGeneration(hc).eval()

    def proj(x):
        return x @ W["predictor.mtp_proj.weight"].T + W["predictor.mtp_proj.bias"] if cfg.has_mtp_proj else x

    sd = {"model." + k[len("predictor."):]: v for k, v in W.items() if k.startswith("predictor.layers.") or k == "predictor.norm.weight"}
    for g in range(G - 1):
        sd[f"model.codec_embedding.{g}.weight"] = proj(W[f"predictor.codec_embedding.{g}.weight"])
        sd[f"lm_head.{g}.weight"] = W[f"predictor.lm_head.{g}.weight"]
    missing, unexpected = sib.load_state_dict(sd, strict=False)
    assert not unexpected and all("inv_freq" in k for k in missing), (missing, unexpected)

    B = 3
    gen = torch.Generator().manual_seed(11)
    past = torch.randn(B, cfg.talker.hidden, generator=gen) * 0.5
    c0 = torch.randint(0, cfg.codec.codebook_size, (B,), generator=gen)
    trace = {}
    codes = m.predictor_frame(None, c0, SamplingParams(), 0, list(range(B)), 0, trace=trace, past_hidden=past)      # greedy
    with torch.no_grad():
        x = torch.stack([proj(past), proj(m.codec_embed(c0))], dim=1)
        out = sib(inputs_embeds=x, past_key_values=DynamicCache(config=hc), use_cache=True)
        ref_logits = [out.logits[:, -1]]
        ref_codes = [ref_logits[0].argmax(-1)]
        for g in range(1, G - 1):
            out = sib(input_ids=ref_codes[-1][:, None], past_key_values=out.past_key_values, use_cache=True, generation_steps=g)
            ref_logits.append(out.logits[:, -1])
            ref_codes.append(ref_logits[-1].argmax(-1))
    for g in range(G - 1):
        assert float((trace["pred_logits"][g] - ref_logits[g]).abs().max()) < 5e-5, g
    assert torch.equal(codes[:, 1:], torch.stack(ref_codes, 1))


def test_shared_prefix_fast_path_equals_full_prompt_forward(tiny_model):
    """share_prefix=True (prefix through the talker once, K/V copied to every item) gives the logits of the plain
    per-item full-prompt forward: causality makes them the same numbers up to matmul summation order."""
    from oracle.model import Voice
    from oracle.sampling import SamplingParams
    cfg, m = tiny_model
    g = torch.Generator().manual_seed(4)
    v = Voice("english", speaker_embed=torch.randn(cfg.talker.hidden, generator=g) * 0.05, ref_text_ids=[5, 6, 7],
              ref_codes=torch.randint(0, cfg.codec.codebook_size, (11, cfg.n_groups), generator=g))
    texts, frames = [[10, 11, 12], [20, 21, 22, 23, 24, 25], [30]], [4, 3, 4]
    tr_a, tr_b = {}, {}
    a = m.generate(v, texts, frames, SamplingParams(), trace=tr_a)
    b = m.generate(v, texts, frames, SamplingParams(), forced_codes=a, trace=tr_b, share_prefix=True)
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    G1 = cfg.n_groups - 1
    for key, per_frame in (("talker_logits", 1), ("pred_logits", G1)):
        for i, (x, y) in enumerate(zip(tr_a[key], tr_b[key])):
            live = torch.tensor([i // per_frame < n for n in frames])      # a finished item's later rows are don't-cares
            assert float((x - y)[live].abs().max()) < 2e-5 * max(1.0, float(x.abs().max()))


def test_bf16_activation_mode_stays_close_to_f32_and_rounds_where_stated():
    """act_bf16=True follows the same graph with bf16 rounding at GEMM inputs / KV cache: logits move by a few % of sigma
    (the size of the bound the f32 oracle needed against the GPU), never more; the cached K/V are bf16 values."""
    from oracle.model import Voice, bf16_round, normed_matmul
    from oracle.sampling import SamplingParams
    cfg = config.tiny()
    state = weights.synthetic_state(cfg, 789)
    m32, m16 = OracleModel(cfg, state), OracleModel(cfg, state, act_bf16=True)
    v = Voice("chinese", speaker="ryan")
    texts, frames = [[10, 11, 12], [20, 21]], [3, 3]
    tr32, tr16 = {}, {}
    free = m32.generate(v, texts, frames, SamplingParams(), trace=tr32)
    m16.generate(v, texts, frames, SamplingParams(), forced_codes=free, trace=tr16)
    a, b = torch.stack(tr32["talker_logits"]), torch.stack(tr16["talker_logits"])
    V0 = cfg.codec.codebook_size
    err = float((a - b)[..., :V0].abs().max()) / float(a[..., :V0].std())
    assert 1e-5 < err < 0.06, err
    for kc in m16.talker.k_cache:
        assert torch.equal(kc, bf16_round(kc))
    # decode-style rounding: bf16(w * x), row scale on the product
    x, w = torch.randn(3, 64), 1.0 + 0.1 * torch.randn(64)
    W = torch.randn(8, 64)
    got = normed_matmul(x, w, 1e-6, [W], "decode")[0]
    want = (bf16_round(w * x) @ W.T) * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6)
    assert torch.equal(got, want)
