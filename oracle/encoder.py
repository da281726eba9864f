"""CPU oracle of the conditioning front-end: reference audio -> codec codes + speaker embedding (TEST INFRASTRUCTURE ONLY).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this file.

**PARITY UNPINNED.**  The reference reaches this step through the third-party ``qwen_tts`` package
(``generate_voice_clone(ref_audio=path, ref_text=...)``, providers/qwen.py:253-258), which is absent; no reference test or
fixture pins it.  The topology restated here is the Mimi codec shipped in the container's transformers
(models/mimi/modeling_mimi.py: MimiConv1d :210-347 causal padding, MimiResnetBlock :408-447, MimiEncoder :450-492,
down-sampling conv with replicate padding :1208-1216, MimiEuclideanCodebook / split RVQ encode :964-1124, _encode_frame
:1229-1262) - whose 12.5 Hz / 2048 entries / 16 codebooks match the model's code layout - with the project's own pre-norm
transformer (oracle/model.py ``Stack``) in the middle.  tests/test_oracle_encoder.py checks the conv encoder, the
down-sampling conv and the quantiser against those sibling modules on shared weights.

The quantiser's distance is DEFINED with a fixed float32 evaluation order (sum over the code dimension, ascending, of
``(x - c)^2`` with separate multiply and add) so that the HIP kernel can reproduce the argmin bit for bit given the same
input vectors; ties go to the lowest index.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from oracle.model import Stack


def causal_conv(x: torch.Tensor, w: torch.Tensor, b, stride: int = 1, dilation: int = 1, pad_mode: str = "constant") -> torch.Tensor:
    """[B,C,T] -> [B,C',T'] with MimiConv1d's causal padding: (k-1)*dilation + 1 - stride on the left, and on the right
    whatever makes the last window complete (modeling_mimi.py:269-281, 327-347)."""
    k = (w.shape[-1] - 1) * dilation + 1
    total = k - stride
    L = x.shape[-1]
    n_frames = int(np.ceil((L - k + total) / stride + 1)) - 1
    extra = n_frames * stride + k - total - L
    mode = "replicate" if pad_mode == "replicate" else "constant"
    x = F.pad(x, (total, extra), mode=mode)
    return F.conv1d(x, w, b, stride=stride, dilation=dilation)


def encoder_channels(cfg):
    c = cfg.codec
    return [c.enc_filters * (2 ** i) for i in range(len(c.enc_ratios) + 1)]


def seanet_encode(W: Dict[str, torch.Tensor], cfg, pcm: torch.Tensor) -> torch.Tensor:
    """pcm [T] float32 -> conv features [T / enc_stride, enc_hidden] (MimiEncoder.forward)."""
    c = cfg.codec
    h = pcm.reshape(1, 1, -1).to(torch.float32)
    h = causal_conv(h, W["enc.conv.0.weight"], W["enc.conv.0.bias"])
    i = 1
    for r in c.enc_ratios:
        y = causal_conv(F.elu(h), W[f"enc.conv.{i}.weight"], W[f"enc.conv.{i}.bias"])
        y = causal_conv(F.elu(y), W[f"enc.conv.{i + 1}.weight"], W[f"enc.conv.{i + 1}.bias"])
        h = h + y
        h = causal_conv(F.elu(h), W[f"enc.conv.{i + 2}.weight"], W[f"enc.conv.{i + 2}.bias"], stride=r)
        i += 3
    h = causal_conv(F.elu(h), W[f"enc.conv.{i}.weight"], W[f"enc.conv.{i}.bias"])
    return h[0].T.contiguous()


def rvq_level(x: np.ndarray, cb: np.ndarray) -> np.ndarray:
    """Nearest codebook entry per row.  x [T,D], cb [K,D] float32 -> int64 [T].  d[t,j] = sum_k (x[t,k] - cb[j,k])^2 with the
    sum taken in ascending k, every operation rounded to float32 (no fused multiply-add); argmin takes the lowest index."""
    T, D = x.shape
    acc = np.zeros((T, cb.shape[0]), dtype=np.float32)
    for k in range(D):
        diff = (x[:, k:k + 1] - cb[None, :, k]).astype(np.float32)
        acc = (acc + (diff * diff).astype(np.float32)).astype(np.float32)
    return acc.argmin(axis=1)


def rvq_encode(W: Dict[str, torch.Tensor], cfg, emb: torch.Tensor) -> torch.Tensor:
    """emb [T, enc_hidden] -> codes [T, num_quantizers]: codebook 0 quantises the semantic projection, codebooks 1.. the
    acoustic projection residually (MimiSplitResidualVectorQuantizer.encode)."""
    c = cfg.codec
    sem = (emb @ W["enc.vq.semantic.input_proj.weight"].T).numpy().astype(np.float32)
    aco = (emb @ W["enc.vq.acoustic.input_proj.weight"].T).numpy().astype(np.float32)
    codes = np.zeros((emb.shape[0], c.num_quantizers), dtype=np.int64)
    codes[:, 0] = rvq_level(sem, W["enc.vq.codebook.0"].numpy().astype(np.float32))
    res = aco
    for q in range(1, c.num_quantizers):
        cb = W[f"enc.vq.codebook.{q}"].numpy().astype(np.float32)
        idx = rvq_level(res, cb)
        codes[:, q] = idx
        res = (res - cb[idx]).astype(np.float32)
    return torch.from_numpy(codes)


def speaker_embed(W: Dict[str, torch.Tensor], cfg, feats: torch.Tensor) -> torch.Tensor:
    """Statistics pooling of the conv features over time -> fc1 + ReLU -> fc2: [T', enc_hidden] -> [talker hidden]."""
    mean = feats.mean(0)
    std = torch.sqrt(((feats - mean) ** 2).mean(0) + 1e-5)
    h = F.relu(torch.cat([mean, std]) @ W["enc.spk.fc1.weight"].T + W["enc.spk.fc1.bias"])
    return h @ W["enc.spk.fc2.weight"].T + W["enc.spk.fc2.bias"]


def encode(state: Dict[str, torch.Tensor], cfg, pcm, return_intermediates: bool = False):
    """pcm: 1-D float32 at cfg.sample_rate, length a multiple of codec.total_upsample (callers trim).  Returns
    (codes [T, num_quantizers] int64, speaker embedding [talker hidden] float32)."""
    from rho_tts_amd.weights import encoder_transformer_dims
    W = {k: v.detach().to("cpu", torch.float32) for k, v in state.items() if k.startswith("enc.")}
    c = cfg.codec
    pcm = torch.as_tensor(np.asarray(pcm, dtype=np.float32))
    if pcm.numel() % c.total_upsample:
        raise ValueError("encoder input must be a whole number of frames")
    feats = seanet_encode(W, cfg, pcm)                                     # [Te, hidden] at 2x the frame rate
    tf = Stack(W, "enc.transformer", encoder_transformer_dims(cfg), qk_norm=False, layer_scale=True, window=c.enc_window)
    Te = feats.shape[0]
    tf.alloc(1, Te)
    h = tf.forward(feats[None], torch.arange(Te)[None])[0]                 # causal, sliding window
    emb = causal_conv(h.T[None], W["enc.downsample.weight"], None, stride=2, pad_mode="replicate")[0].T.contiguous()
    codes = rvq_encode(W, cfg, emb)
    spk = speaker_embed(W, cfg, feats)
    if return_intermediates:
        return codes, spk, {"feats": feats, "tf": h, "emb": emb}
    return codes, spk
