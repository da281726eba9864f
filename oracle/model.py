"""CPU oracle of the Qwen3-TTS-shaped generation path (TEST INFRASTRUCTURE ONLY).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this file, and only as the checker / the timed CPU baseline.

**PARITY UNPINNED.**  The reference's model arithmetic lives in the third-party
PyPI package ``qwen-tts`` (declared unpinned, pyproject.toml:40-43; imported
lazily at providers/qwen.py:100; called at providers/qwen.py:247-258).  It is
not under /root/reference, not installed, and no reference test pins its
numerics (the only test touching it mocks it, tests/test_sound_decay.py:113-117).
This file therefore restates the *published architecture family* — the
Qwen3-Omni talker / code-predictor / code2wav stack shipped in the container's
``transformers`` (models/qwen3_omni_moe/modeling_qwen3_omni_moe.py: decoder
layer :2340-2379, q/k-norm attention :2250-2321, predictor recurrence
:2534-2608 and :3137-3176, CausalConvNet :3180-3213, CausalTransConvNet
:3216-3228, ConvNeXtBlock :3231-3263, SnakeBeta :3542-3580, residual unit
:3592-3608, decoder block :3611-3633, Code2Wav :3636-3696) — with dense MLPs
and plain RoPE, parametrised by ``rho_tts_amd.config.ModelConfig``.  Its
building blocks are cross-checked against those sibling modules built from
small configs (tests/test_oracle_model.py); the end-to-end result is what the
HIP path is compared with.

Numerics: weights are bf16 values held in float32 (the reference loads the
model with dtype=torch.bfloat16, providers/qwen.py:163), activations and
accumulation float32.

``OracleModel(..., act_bf16=True)`` additionally rounds activations to bf16 at
the points where a bf16 model does (the reference runs the model in
``dtype=torch.bfloat16``: every Linear input and the KV cache are bf16 there):
GEMM inputs, attention output, SwiGLU product, cached K/V.  The rounding points
follow the HIP path's (DESIGN.md "Precision policy"): the prompt prefill rounds
``bf16(rmsnorm(x) * w)``, a decode step rounds ``bf16(w * x)`` and applies the
row scale to the f32 accumulator.  Measured (tools/diag_logits.py, 1.7B, batch
32): mirroring the rounding points does NOT bring the GPU closer than the f32
oracle is - bf16 rounding realisations of two implementations decorrelate after
a few layers - so tests/test_model_shapes_gpu.py uses this mode to MEASURE the
bf16 noise floor (bf16 oracle vs f32 oracle) and holds the GPU to it.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

from oracle.sampling import SamplingParams, draw, uniform


@dataclass
class Voice:
    """Conditioning of one voice (what rt_set_voice receives)."""
    language: str = "english"
    speaker: Optional[str] = None                  # built-in voice (CustomVoice models, qwen.py:247-251)
    speaker_embed: Optional[torch.Tensor] = None   # [H] clone embedding (Base models, qwen.py:253-258)
    ref_text_ids: List[int] = field(default_factory=list)
    ref_codes: Optional[torch.Tensor] = None       # [T_ref, G] int64


def rms_norm(x, w, eps):
    v = x.pow(2).mean(-1, keepdim=True)
    return w * (x * torch.rsqrt(v + eps))


def bf16_round(x):
    return x.to(torch.bfloat16).to(torch.float32)


def normed_matmul(x, w_norm, eps, mats, mode):
    """[rmsnorm(x; w_norm) @ W.T for W in mats] with the rounding points of ``mode``:
    None: float32 throughout; "prefill": the GEMM input is bf16(rmsnorm(x) * w); "decode": the GEMM input is
    bf16(w * x) and the RMSNorm row scale multiplies the f32 product (gemm_col.hip's post-scale)."""
    if mode == "decode":
        a = bf16_round(w_norm * x)
        inv = torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps)
        return [(a @ W.T) * inv for W in mats]
    h = rms_norm(x, w_norm, eps)
    if mode == "prefill":
        h = bf16_round(h)
    return [h @ W.T for W in mats]


def rope_table(head_dim: int, theta: float, n_pos: int):
    """cos/sin [n_pos, head_dim/2] in float32 (the same table is uploaded to the GPU)."""
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    fr = torch.arange(n_pos, dtype=torch.float32)[:, None] * inv[None, :]
    return fr.cos(), fr.sin()


def apply_rope(x, cos, sin):
    """x [..., T, d]; cos/sin [..., T, d/2]; rotate-half convention."""
    d = x.shape[-1] // 2
    x1, x2 = x[..., :d], x[..., d:]
    return torch.cat([x1 * cos - x2 * sin, x2 * cos + x1 * sin], dim=-1)


class Stack:
    """A pre-norm decoder stack with GQA, optional q/k RMSNorm, optional LayerScale and sliding window."""

    def __init__(self, W: Dict[str, torch.Tensor], prefix: str, dims, qk_norm=True, layer_scale=False, window=None):
        self.W, self.p, self.d = W, prefix, dims
        self.qk_norm, self.layer_scale, self.window = qk_norm, layer_scale, window
        self.k_cache: List[torch.Tensor] = []
        self.v_cache: List[torch.Tensor] = []

    def alloc(self, batch: int, max_pos: int):
        d = self.d
        self.k_cache = [torch.zeros(batch, d.kv_heads, max_pos, d.head_dim) for _ in range(d.layers)]
        self.v_cache = [torch.zeros(batch, d.kv_heads, max_pos, d.head_dim) for _ in range(d.layers)]
        self.cos, self.sin = rope_table(d.head_dim, d.rope_theta, max_pos)

    def forward(self, x: torch.Tensor, pos: torch.Tensor, final_norm=True, mode=None) -> torch.Tensor:
        """x [B, T, H]; pos [B, T] absolute positions (also the cache rows written).
        Row (b, t) attends cache rows j <= pos[b, t] (and > pos - window when windowed).
        mode: bf16 rounding points (see normed_matmul); None = float32 activations."""
        d, W, p = self.d, self.W, self.p
        rnd = bf16_round if mode else (lambda t: t)
        B, T, _ = x.shape
        cos, sin = self.cos[pos][:, None], self.sin[pos][:, None]            # [B,1,T,d/2]
        n_ctx = int(pos.max()) + 1
        j = torch.arange(n_ctx)[None, None, :]
        mask = j <= pos[:, :, None]
        if self.window is not None:
            mask = mask & (pos[:, :, None] - j < self.window)
        bi = torch.arange(B)[:, None].expand(B, T)
        for i in range(d.layers):
            lp = f"{p}.layers.{i}"
            q, k, v = normed_matmul(x, W[f"{lp}.input_layernorm.weight"], d.rms_eps,
                                    [W[f"{lp}.self_attn.q_proj.weight"], W[f"{lp}.self_attn.k_proj.weight"], W[f"{lp}.self_attn.v_proj.weight"]], mode)
            q, k, v = q.view(B, T, d.heads, d.head_dim), k.view(B, T, d.kv_heads, d.head_dim), v.view(B, T, d.kv_heads, d.head_dim)
            if self.qk_norm:
                q = rms_norm(q, W[f"{lp}.self_attn.q_norm.weight"], d.rms_eps)
                k = rms_norm(k, W[f"{lp}.self_attn.k_norm.weight"], d.rms_eps)
            q = apply_rope(q.transpose(1, 2), cos, sin)                       # [B,h,T,d]
            k = apply_rope(k.transpose(1, 2), cos, sin)
            self.k_cache[i][bi, :, pos] = rnd(k.transpose(1, 2))
            self.v_cache[i][bi, :, pos] = rnd(v)
            rep = d.heads // d.kv_heads
            K = self.k_cache[i][:, :, :n_ctx].repeat_interleave(rep, dim=1)
            V = self.v_cache[i][:, :, :n_ctx].repeat_interleave(rep, dim=1)
            s = (q @ K.transpose(-1, -2)) * (d.head_dim ** -0.5)
            s = s.masked_fill(~mask[:, None], float("-inf"))
            a = torch.softmax(s, dim=-1) @ V                                  # [B,h,T,d]
            o = rnd(a.transpose(1, 2).reshape(B, T, d.q_dim)) @ W[f"{lp}.self_attn.o_proj.weight"].T
            if self.layer_scale:
                o = o * W[f"{lp}.self_attn_layer_scale.scale"]
            x = x + o
            gt, up = normed_matmul(x, W[f"{lp}.post_attention_layernorm.weight"], d.rms_eps,
                                   [W[f"{lp}.mlp.gate_proj.weight"], W[f"{lp}.mlp.up_proj.weight"]], mode)
            m = rnd(F.silu(gt) * up) @ W[f"{lp}.mlp.down_proj.weight"].T
            if self.layer_scale:
                m = m * W[f"{lp}.mlp_layer_scale.scale"]
            x = x + m
        return rms_norm(x, W[f"{p}.norm.weight"], d.rms_eps) if final_norm else x

    def head(self, x: torch.Tensor, mats, mode=None):
        """[final_norm(x) @ W.T for W in mats] for the un-normalised stack output x (forward(..., final_norm=False))."""
        return normed_matmul(x, self.W[f"{self.p}.norm.weight"], self.d.rms_eps, mats, mode)


# ---------------------------------------------------------------- codec decoder ops
def causal_conv1d(x, w, b, dilation=1, groups=1):
    """[B,C,T] -> [B,C',T]; left zero padding (k-1)*dilation (CausalConvNet, stride 1)."""
    k = w.shape[-1]
    return F.conv1d(F.pad(x, ((k - 1) * dilation, 0)), w, b, dilation=dilation, groups=groups)


def causal_trans_conv1d(x, w, b, stride):
    """ConvTranspose1d then trim ceil(k - stride) samples on BOTH sides (CausalTransConvNet :3221-3227)."""
    k = w.shape[-1]
    y = F.conv_transpose1d(x, w, b, stride=stride)
    pad = int(math.ceil(k - stride))
    return y[..., pad: y.shape[-1] - pad]


def snake_beta(x, alpha, beta):
    a = torch.exp(alpha)[None, :, None]
    bt = torch.exp(beta)[None, :, None]
    return x + (1.0 / (bt + 1e-9)) * torch.sin(x * a).pow(2)


class OracleModel:
    def __init__(self, cfg, state: Dict[str, torch.Tensor], act_bf16: bool = False):
        self.cfg = cfg
        self.act_bf16 = act_bf16
        self.W = {k: v.detach().to("cpu", torch.float32) for k, v in state.items()}
        from rho_tts_amd.weights import codec_transformer_dims
        self.talker = Stack(self.W, "talker", cfg.talker)
        self.pred = Stack(self.W, "predictor", cfg.predictor)
        self.codec_tf = Stack(self.W, "codec.pre_transformer", codec_transformer_dims(cfg), qk_norm=False,
                              layer_scale=True, window=cfg.codec.sliding_window)

    # ------------------------------------------------------------ embeddings
    def text_embed(self, ids: Sequence[int]) -> torch.Tensor:
        W = self.W
        e = W["talker.text_embedding.weight"][torch.as_tensor(list(ids), dtype=torch.long)]
        h = F.silu(e @ W["talker.text_projection.fc1.weight"].T + W["talker.text_projection.fc1.bias"])
        if self.act_bf16:
            h = bf16_round(h)
        return h @ W["talker.text_projection.fc2.weight"].T + W["talker.text_projection.fc2.bias"]

    def codec_embed(self, ids) -> torch.Tensor:
        return self.W["talker.codec_embedding.weight"][torch.as_tensor(ids, dtype=torch.long)]

    def frame_embed(self, codes: torch.Tensor) -> torch.Tensor:
        """Sum over the G codebooks of one or more frames: codes [..., G] -> [..., H]."""
        e = self.codec_embed(codes[..., 0])
        for g in range(1, self.cfg.n_groups):
            e = e + self.W[f"predictor.codec_embedding.{g - 1}.weight"][codes[..., g].long()]
        return e

    def prefix_embeddings(self, voice: Voice) -> torch.Tensor:
        """Shared prompt prefix of a voice (see DESIGN.md "Prompt layout"): role tokens, the codec
        control block (think ids, language, speaker), then the reference transcript and the reference
        codec frames.  Depends on the voice only, never on the text to speak."""
        c = self.cfg
        pad_t = self.text_embed([c.tts_pad_id])[0]
        bos_t = self.text_embed([c.tts_bos_id])[0]
        rows = [self.text_embed(c.role_ids)]
        lang = c.language_ids.get(voice.language.lower())
        if lang is None:
            raise ValueError(f"unsupported language {voice.language!r}")
        ctrl = self.codec_embed([c.codec_nothink_id, c.codec_think_bos_id, lang, c.codec_think_eos_id])
        rows.append(ctrl + pad_t)
        if voice.speaker_embed is not None:
            spk = voice.speaker_embed.to(torch.float32)
        elif voice.speaker is not None:
            sid = c.speaker_ids.get(voice.speaker.lower())
            if sid is None:
                raise ValueError(f"unknown speaker {voice.speaker!r}")
            spk = self.codec_embed([sid])[0]
        else:
            raise ValueError("voice needs a speaker embedding (clone) or a built-in speaker")
        rows.append((spk + pad_t)[None])
        rows.append((self.codec_embed([c.codec_pad_id])[0] + bos_t)[None])
        if voice.ref_codes is not None and voice.ref_codes.numel() > 0:
            pad_c = self.codec_embed([c.codec_pad_id])[0]
            if voice.ref_text_ids:
                rows.append(self.text_embed(voice.ref_text_ids) + pad_c)
            rows.append((self.codec_embed([c.codec_bos_id])[0] + pad_t)[None])
            rows.append(self.frame_embed(voice.ref_codes.long()) + pad_t)
        return torch.cat(rows, dim=0)

    def suffix_embeddings(self, text_ids: Sequence[int]) -> torch.Tensor:
        c = self.cfg
        pad_c = self.codec_embed([c.codec_pad_id])[0]
        pad_t = self.text_embed([c.tts_pad_id])[0]
        rows = [self.text_embed(list(text_ids) + [c.tts_eos_id]) + pad_c,
                (self.codec_embed([c.codec_bos_id])[0] + pad_t)[None]]
        return torch.cat(rows, dim=0)

    # ------------------------------------------------------------ generation
    def talker_suppress(self, allow_eos: bool) -> np.ndarray:
        c = self.cfg
        s = np.zeros(c.codec_vocab, dtype=bool)
        s[c.codec.codebook_size:] = True
        if allow_eos:
            s[c.codec_eos_id] = False
        return s

    def predictor_frame(self, talker_x, c0, sp, seed, items, frame, forced=None, trace=None, past_hidden=None):
        """Residual codes 1..G-1 of one frame (predictor recurrence, sibling :2534-2608, :3137-3176).
        talker_x [B,H]: the talker's output row BEFORE its final norm (or None with past_hidden [B,H], the
        post-norm state, given directly); c0 [B].  Returns codes [B,G]."""
        c, W = self.cfg, self.W
        B = c0.shape[0]
        self.pred.alloc(B, c.n_groups + 1)
        mode = "decode" if self.act_bf16 else None

        def proj(x):
            if c.has_mtp_proj:
                return x @ W["predictor.mtp_proj.weight"].T + W["predictor.mtp_proj.bias"]
            return x

        if past_hidden is not None:
            past = proj(past_hidden)
        elif c.has_mtp_proj:        # mtp_proj(final_norm(x)): a GEMM behind the talker's final norm
            past = self.talker.head(talker_x, [W["predictor.mtp_proj.weight"]], mode)[0] + W["predictor.mtp_proj.bias"]
        else:
            past = rms_norm(talker_x, W["talker.norm.weight"], c.talker.rms_eps)
        codes = torch.zeros(B, c.n_groups, dtype=torch.long)
        codes[:, 0] = c0
        x = torch.stack([past, proj(self.codec_embed(c0))], dim=1)                   # [B,2,Hp]
        pos = torch.tensor([[0, 1]]).expand(B, 2)
        h = self.pred.forward(x, pos, final_norm=False, mode=mode)[:, -1]
        for g in range(c.n_groups - 1):
            logits = self.pred.head(h, [W[f"predictor.lm_head.{g}.weight"]], mode)[0]
            if trace is not None:
                trace.setdefault("pred_logits", []).append(logits.clone())
            nxt = torch.zeros(B, dtype=torch.long)
            for b in range(B):
                if forced is not None:
                    nxt[b] = int(forced[b, g + 1])
                else:
                    nxt[b] = draw(logits[b].numpy(), sp, uniform(seed, items[b], frame, g + 1))
            codes[:, g + 1] = nxt
            if g < c.n_groups - 2:
                e = proj(W[f"predictor.codec_embedding.{g}.weight"][nxt])[:, None]
                h = self.pred.forward(e, torch.full((B, 1), g + 2), final_norm=False, mode=mode)[:, -1]
        return codes

    def generate(self, voice: Voice, texts: Sequence[Sequence[int]], max_frames: Sequence[int],
                 sp_talker: SamplingParams = SamplingParams(), sp_pred: Optional[SamplingParams] = None,
                 seed: int = 789, item_ids: Optional[Sequence[int]] = None, ignore_eos: bool = True,
                 min_frames: int = 2, forced_codes: Optional[Sequence[torch.Tensor]] = None, trace: Optional[dict] = None,
                 share_prefix: bool = False, timing: Optional[dict] = None):
        """Autoregressive decode of a batch.  Returns a list of int64 code tensors [T_i, G].

        forced_codes: teacher forcing — the chosen codes of item b at frame t are taken from
        forced_codes[b][t] while logits are still recorded into ``trace``.
        share_prefix: run the voice prefix through the talker once and copy its K/V into every item's cache
        (attention is causal, so the prefix rows do not depend on what follows them: same numbers up to the
        summation order of a differently shaped matmul; tests/test_oracle_model.py).  This is what makes a 460-row
        clone prefix at batch 32 affordable on the CPU."""
        c = self.cfg
        sp_pred = sp_pred or sp_talker
        B = len(texts)
        items = list(item_ids) if item_ids is not None else list(range(B))
        m_pre = "prefill" if self.act_bf16 else None
        m_dec = "decode" if self.act_bf16 else None
        prefix = self.prefix_embeddings(voice)
        Lp = prefix.shape[0]
        suffixes = [self.suffix_embeddings(t) for t in texts]
        P = torch.tensor([Lp + sfx.shape[0] for sfx in suffixes])
        T_max = int(max(max_frames))
        n_pos = int(P.max()) + T_max + 1
        if n_pos > c.max_positions:
            raise RuntimeError(f"prompt length + frames = {n_pos} exceeds max_positions {c.max_positions}")
        if share_prefix:
            self.talker.alloc(1, n_pos)
            self.talker.forward(prefix[None], torch.arange(Lp)[None], final_norm=False, mode=m_pre)
            pk = [k[:, :, :Lp].clone() for k in self.talker.k_cache]
            pv = [v[:, :, :Lp].clone() for v in self.talker.v_cache]
            self.talker.alloc(B, n_pos)
            for i in range(c.talker.layers):
                self.talker.k_cache[i][:, :, :Lp] = pk[i]
                self.talker.v_cache[i][:, :, :Lp] = pv[i]
            S = int(P.max()) - Lp
            x = torch.zeros(B, S, c.talker.hidden)
            for b, sfx in enumerate(suffixes):
                x[b, : sfx.shape[0]] = sfx
            x_all = self.talker.forward(x, (Lp + torch.arange(S))[None].expand(B, -1), final_norm=False, mode=m_pre)
            xt = x_all[torch.arange(B), P - 1 - Lp]
        else:
            self.talker.alloc(B, n_pos)
            x = torch.zeros(B, int(P.max()), c.talker.hidden)
            for b, sfx in enumerate(suffixes):
                x[b, :Lp] = prefix
                x[b, Lp: Lp + sfx.shape[0]] = sfx
            pos = torch.arange(int(P.max()))[None].expand(B, -1)
            x_all = self.talker.forward(x, pos, final_norm=False, mode=m_pre)
            xt = x_all[torch.arange(B), P - 1]                                # [B,H] last prompt row, before the final norm
        pad_t = self.text_embed([c.tts_pad_id])[0]
        seen = np.zeros((B, c.codec_vocab), dtype=bool)
        out = [[] for _ in range(B)]
        done = [False] * B
        if timing is not None:                      # bench.py's CPU baseline: prompt prefill vs per-frame decode
            import time as _time
            timing["prefill_done"] = _time.perf_counter()
        for t in range(T_max):
            logits = self.talker.head(xt, [self.W["talker.codec_head.weight"]], m_dec)[0]
            if trace is not None:
                trace.setdefault("talker_logits", []).append(logits.clone())
            c0 = torch.zeros(B, dtype=torch.long)
            for b in range(B):
                if forced_codes is not None and t < forced_codes[b].shape[0]:
                    c0[b] = int(forced_codes[b][t, 0])
                else:
                    allow_eos = (not ignore_eos) and t >= min_frames
                    c0[b] = draw(logits[b].numpy(), sp_talker, uniform(seed, items[b], t, 0),
                                 self.talker_suppress(allow_eos), seen[b])
                seen[b, int(c0[b])] = True
            forced_t = None
            if forced_codes is not None:
                forced_t = torch.stack([fc[min(t, fc.shape[0] - 1)] for fc in forced_codes])
            c0_in = torch.where(c0 == c.codec_eos_id, torch.zeros_like(c0), c0)  # eos rows produce no frame; keep ids in range
            codes = self.predictor_frame(xt, c0_in, sp_pred, seed, items, t, forced_t, trace)
            for b in range(B):
                if done[b]:
                    continue
                if int(c0[b]) == c.codec_eos_id:
                    done[b] = True                 # eos row: no frame emitted
                    continue
                out[b].append(codes[b].clone())
                if len(out[b]) >= max_frames[b]:
                    done[b] = True
            if all(done):
                break
            e = self.frame_embed(codes) + pad_t
            xt = self.talker.forward(e[:, None], (P + t)[:, None], final_norm=False, mode=m_dec)[:, 0]
        return [torch.stack(o) if o else torch.zeros(0, c.n_groups, dtype=torch.long) for o in out]

    # ------------------------------------------------------------ vocoder
    def code2wav(self, codes: torch.Tensor) -> torch.Tensor:
        """codes [B, G_q, T] (first num_quantizers groups) -> wav [B, L], clamp(-1,1) (Code2Wav.forward :3672-3684)."""
        c, W = self.cfg.codec, self.W
        B, Q, T = codes.shape
        off = (torch.arange(Q) * c.codebook_size)[None, :, None]
        h = W["codec.code_embedding.weight"][codes.long() + off].mean(1)      # [B,T,hidden]
        self.codec_tf.alloc(B, T)
        h = self.codec_tf.forward(h, torch.arange(T)[None].expand(B, -1))
        h = h.transpose(1, 2)                                                  # [B,C,T]
        for i, r in enumerate(c.upsampling_ratios):
            u = f"codec.upsample.{i}"
            h = causal_trans_conv1d(h, W[f"{u}.0.conv.weight"], W[f"{u}.0.conv.bias"], r)
            y = causal_conv1d(h, W[f"{u}.1.dwconv.conv.weight"], W[f"{u}.1.dwconv.conv.bias"], groups=h.shape[1])
            y = F.layer_norm(y.transpose(1, 2), (h.shape[1],), W[f"{u}.1.norm.weight"], W[f"{u}.1.norm.bias"], 1e-6)
            y = F.gelu(y @ W[f"{u}.1.pwconv1.weight"].T + W[f"{u}.1.pwconv1.bias"])
            y = (y @ W[f"{u}.1.pwconv2.weight"].T + W[f"{u}.1.pwconv2.bias"]) * W[f"{u}.1.gamma"]
            h = h + y.transpose(1, 2)
        h = causal_conv1d(h, W["codec.decoder.0.conv.weight"], W["codec.decoder.0.conv.bias"])
        for i, r in enumerate(c.upsample_rates):
            bp = f"codec.decoder.{i + 1}.block"
            h = snake_beta(h, W[f"{bp}.0.alpha"], W[f"{bp}.0.beta"])
            h = causal_trans_conv1d(h, W[f"{bp}.1.conv.weight"], W[f"{bp}.1.conv.bias"], r)
            for j, dil in enumerate((1, 3, 9)):
                u = f"{bp}.{j + 2}"
                y = snake_beta(h, W[f"{u}.act1.alpha"], W[f"{u}.act1.beta"])
                y = causal_conv1d(y, W[f"{u}.conv1.conv.weight"], W[f"{u}.conv1.conv.bias"], dilation=dil)
                y = snake_beta(y, W[f"{u}.act2.alpha"], W[f"{u}.act2.beta"])
                y = causal_conv1d(y, W[f"{u}.conv2.conv.weight"], W[f"{u}.conv2.conv.bias"])
                h = h + y
        n = len(c.upsample_rates) + 1
        h = snake_beta(h, W[f"codec.decoder.{n}.alpha"], W[f"codec.decoder.{n}.beta"])
        h = causal_conv1d(h, W[f"codec.decoder.{n + 1}.conv.weight"], W[f"codec.decoder.{n + 1}.conv.bias"])
        return h[:, 0].clamp(-1, 1)

    def chunked_code2wav(self, codes: torch.Tensor) -> torch.Tensor:
        """chunked_decode (:3686-3696): chunks of chunk_frames with left_context_frames of left context."""
        c = self.cfg.codec
        T = codes.shape[-1]
        outs, start = [], 0
        while start < T:
            end = min(start + c.chunk_frames, T)
            ctx = c.left_context_frames if start - c.left_context_frames > 0 else start
            w = self.code2wav(codes[..., start - ctx:end])
            outs.append(w[..., ctx * c.total_upsample:])
            start = end
        return torch.cat(outs, dim=-1)

    def wav_length(self, n_frames: int) -> int:
        """Samples produced for n_frames by one un-chunked code2wav call (both-side trimmed transposed convs)."""
        c = self.cfg.codec
        L = n_frames
        for r in c.upsampling_ratios:
            L = L * r
        for r in c.upsample_rates:
            L = (L - 1) * r
        return max(L, 0)
