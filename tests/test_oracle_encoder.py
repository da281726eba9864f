"""The encoder oracle (oracle/encoder.py) against the sibling modules in the container's transformers (Mimi) on shared weights:
conv encoder, replicate-padded down-sampling conv, split residual vector quantiser.  CPU only."""
import numpy as np
import pytest
import torch

from oracle import encoder as E
from rho_tts_amd import config, weights

torch.set_num_threads(4)


@pytest.fixture(scope="module")
def tiny():
    cfg = config.tiny()
    st = {k: v.float() for k, v in weights.synthetic_state(cfg, 789, only_prefix="enc.").items()}
    return cfg, st


def mimi_config(cfg):
    from transformers.models.mimi.configuration_mimi import MimiConfig
    c = cfg.codec
    return MimiConfig(audio_channels=1, num_filters=c.enc_filters, kernel_size=c.enc_kernel, residual_kernel_size=c.enc_res_kernel,
                      last_kernel_size=c.enc_last_kernel, upsampling_ratios=list(reversed(c.enc_ratios)), num_residual_layers=1,
                      dilation_growth_rate=2, compress=2, hidden_size=c.enc_hidden, use_causal_conv=True, pad_mode="constant",
                      use_conv_shortcut=False, codebook_size=c.codebook_size, codebook_dim=c.vq_dim, vector_quantization_hidden_dimension=c.vq_dim,
                      num_quantizers=c.num_quantizers, num_semantic_quantizers=1, sampling_rate=cfg.sample_rate,
                      frame_rate=cfg.sample_rate / c.total_upsample, num_hidden_layers=1, num_attention_heads=2, num_key_value_heads=2,
                      head_dim=c.enc_hidden // 2, intermediate_size=64)


def test_conv_encoder_matches_mimi(tiny):
    from transformers.models.mimi.modeling_mimi import MimiEncoder
    cfg, W = tiny
    enc = MimiEncoder(mimi_config(cfg)).eval()
    convs = [n for n, m in enc.named_modules() if n.endswith(".conv")]
    assert len(convs) == 2 + 3 * len(cfg.codec.enc_ratios)
    sd = {}
    for i, n in enumerate(convs):                       # execution order == registration order
        sd[n + ".weight"], sd[n + ".bias"] = W[f"enc.conv.{i}.weight"], W[f"enc.conv.{i}.bias"]
    missing, unexpected = enc.load_state_dict(sd, strict=False)
    assert not unexpected and not [m for m in missing if "conv" in m], (missing, unexpected)
    g = torch.Generator().manual_seed(1)
    for n_frames in (1, 7, 40):
        pcm = torch.randn(n_frames * cfg.codec.total_upsample, generator=g) * 0.3
        with torch.no_grad():
            ref = enc(pcm[None, None])[0].T
        got = E.seanet_encode(W, cfg, pcm)
        assert got.shape == ref.shape == (2 * n_frames, cfg.codec.enc_hidden)
        assert float((got - ref).abs().max()) < 1e-5 * max(1.0, float(ref.abs().max()))


def test_downsample_and_quantiser_match_mimi(tiny):
    from transformers.models.mimi.modeling_mimi import MimiConv1d, MimiSplitResidualVectorQuantizer
    cfg, W = tiny
    mc = mimi_config(cfg)
    c = cfg.codec
    down = MimiConv1d(mc, c.enc_hidden, c.enc_hidden, 4, stride=2, bias=False, pad_mode="replicate").eval()
    down.load_state_dict({"conv.weight": W["enc.downsample.weight"]}, strict=False)
    g = torch.Generator().manual_seed(2)
    h = torch.randn(30, c.enc_hidden, generator=g)
    with torch.no_grad():
        ref = down(h.T[None])[0].T
    got = E.causal_conv(h.T[None], W["enc.downsample.weight"], None, stride=2, pad_mode="replicate")[0].T
    assert got.shape == (15, c.enc_hidden) and float((got - ref).abs().max()) < 1e-5
    q = MimiSplitResidualVectorQuantizer(mc).eval()
    sd = {"semantic_residual_vector_quantizer.input_proj.weight": W["enc.vq.semantic.input_proj.weight"][:, :, None],
          "acoustic_residual_vector_quantizer.input_proj.weight": W["enc.vq.acoustic.input_proj.weight"][:, :, None]}
    for k in range(c.num_quantizers):
        p = "semantic_residual_vector_quantizer.layers.0" if k == 0 else f"acoustic_residual_vector_quantizer.layers.{k - 1}"
        sd[p + ".codebook.embed_sum"] = W[f"enc.vq.codebook.{k}"]
        sd[p + ".codebook.cluster_usage"] = torch.ones(c.codebook_size)
    missing, unexpected = q.load_state_dict(sd, strict=False)
    assert not unexpected and not [m for m in missing if "embed_sum" in m or "input_proj" in m], (missing, unexpected)
    emb = torch.randn(50, c.enc_hidden, generator=g)
    with torch.no_grad():
        ref_codes = q.encode(emb.T[None])[:, 0].T            # [K, B, T] -> [T, K]
    got_codes = E.rvq_encode(W, cfg, emb)
    assert got_codes.shape == (50, c.num_quantizers)
    # cdist (sqrt of a differently ordered sum) and the fixed-order sum can only disagree on near-ties
    assert float((got_codes == ref_codes).float().mean()) >= 0.99
    assert int(got_codes.max()) < c.codebook_size and int(got_codes.min()) >= 0


def test_rvq_level_definition_and_tie_break():
    x = np.array([[0.0, 0.0], [1.0, 1.0]], dtype=np.float32)
    cb = np.array([[1.0, 0.0], [0.0, 1.0], [1.0, 1.0], [0.0, 1.0]], dtype=np.float32)
    assert E.rvq_level(x, cb).tolist() == [0, 2]             # row 0: entries 0, 1, 3 tie at distance 1 -> lowest index
    g = np.random.default_rng(3)
    x, cb = g.standard_normal((20, 16)).astype(np.float32), g.standard_normal((64, 16)).astype(np.float32)
    brute = ((x[:, None, :].astype(np.float64) - cb[None].astype(np.float64)) ** 2).sum(-1).argmin(1)
    assert (E.rvq_level(x, cb) == brute).mean() >= 0.95


def test_encode_end_to_end_shapes_and_determinism(tiny):
    cfg, W = tiny
    g = torch.Generator().manual_seed(4)
    pcm = (torch.randn(23 * cfg.codec.total_upsample, generator=g) * 0.2).numpy()
    codes, spk, mid = E.encode(W, cfg, pcm, return_intermediates=True)
    assert codes.shape == (23, cfg.codec.num_quantizers) and spk.shape == (cfg.talker.hidden,)
    assert mid["feats"].shape == (46, cfg.codec.enc_hidden) and mid["emb"].shape == (23, cfg.codec.enc_hidden)
    codes2, spk2 = E.encode(W, cfg, pcm)
    assert torch.equal(codes, codes2) and torch.equal(spk, spk2)
    # causal: the first frames do not depend on what follows
    codes3, _ = E.encode(W, cfg, pcm[: 10 * cfg.codec.total_upsample])
    assert torch.equal(codes3, codes[:10])
    assert len(set(codes[:, 0].tolist())) > 3                 # not degenerate
    with pytest.raises(ValueError):
        E.encode(W, cfg, pcm[:-1])
