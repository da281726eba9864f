"""Tensor inventory, synthetic initialisation and checkpoint loading.

There are no Qwen3-TTS weights offline, so benchmarks and
parity tests run on seeded synthetic weights of the configured shapes.  The
generator is a counter hash evaluated with exact integer arithmetic followed by
exactly-rounded float ops, so the same bytes come out on CPU and on the GPU.
A real checkpoint (safetensors, same tensor names) is loaded by ``load_safetensors``.
"""
from __future__ import annotations

import math
import os
import zlib
from typing import Dict, Iterator, List, Tuple

import torch

from .config import ModelConfig, TransformerDims

Spec = Tuple[str, Tuple[int, ...], str, float]  # name, shape, kind, scale


def _layer_specs(prefix: str, d: TransformerDims, std: float, layer_scale: bool = False, qk_norm: bool = True) -> List[Spec]:
    out: List[Spec] = []
    for i in range(d.layers):
        p = f"{prefix}.layers.{i}"
        out += [(f"{p}.input_layernorm.weight", (d.hidden,), "norm", 0.1),
                (f"{p}.self_attn.q_proj.weight", (d.q_dim, d.hidden), "mat", std),
                (f"{p}.self_attn.k_proj.weight", (d.kv_dim, d.hidden), "mat", std),
                (f"{p}.self_attn.v_proj.weight", (d.kv_dim, d.hidden), "mat", std),
                (f"{p}.self_attn.o_proj.weight", (d.hidden, d.q_dim), "mat", std)]
        if qk_norm:
            out += [(f"{p}.self_attn.q_norm.weight", (d.head_dim,), "norm", 0.1),
                    (f"{p}.self_attn.k_norm.weight", (d.head_dim,), "norm", 0.1)]
        out += [(f"{p}.post_attention_layernorm.weight", (d.hidden,), "norm", 0.1),
                (f"{p}.mlp.gate_proj.weight", (d.inter, d.hidden), "mat", std),
                (f"{p}.mlp.up_proj.weight", (d.inter, d.hidden), "mat", std),
                (f"{p}.mlp.down_proj.weight", (d.hidden, d.inter), "mat", std)]
        if layer_scale:
            out += [(f"{p}.self_attn_layer_scale.scale", (d.hidden,), "const", 0.0),
                    (f"{p}.mlp_layer_scale.scale", (d.hidden,), "const", 0.0)]
    out.append((f"{prefix}.norm.weight", (d.hidden,), "norm", 0.1))
    return out


def codec_transformer_dims(cfg: ModelConfig) -> TransformerDims:
    c = cfg.codec
    return TransformerDims(hidden=c.hidden, layers=c.layers, heads=c.heads, kv_heads=c.heads, head_dim=c.head_dim,
                           inter=c.inter, rope_theta=c.rope_theta, rms_eps=c.rms_eps)


def decoder_channels(cfg: ModelConfig) -> List[int]:
    c = cfg.codec
    return [c.decoder_dim // (2 ** i) for i in range(len(c.upsample_rates) + 1)]


def tensor_specs(cfg: ModelConfig) -> List[Spec]:
    """Every parameter of the model: (name, shape, kind, scale).  kinds:
    mat = N(0, scale^2)-like matrix; fan = scale/sqrt(fan_in) matrix (convs);
    norm = 1 + scale*u; bias = scale*u; const = fixed value (see _const_value)."""
    t, p, c = cfg.talker, cfg.predictor, cfg.codec
    std = 0.02
    s: List[Spec] = [
        ("talker.text_embedding.weight", (cfg.text_vocab, cfg.text_hidden), "mat", std),
        ("talker.text_projection.fc1.weight", (cfg.text_hidden, cfg.text_hidden), "mat", std),
        ("talker.text_projection.fc1.bias", (cfg.text_hidden,), "bias", 0.01),
        ("talker.text_projection.fc2.weight", (t.hidden, cfg.text_hidden), "mat", std),
        ("talker.text_projection.fc2.bias", (t.hidden,), "bias", 0.01),
        ("talker.codec_embedding.weight", (cfg.codec_vocab, t.hidden), "mat", std),
    ]
    s += _layer_specs("talker", t, std)
    s.append(("talker.codec_head.weight", (cfg.codec_vocab, t.hidden), "mat", 0.08))
    if cfg.has_mtp_proj:
        s += [("predictor.mtp_proj.weight", (p.hidden, t.hidden), "mat", std),
              ("predictor.mtp_proj.bias", (p.hidden,), "bias", 0.01)]
    for g in range(cfg.n_groups - 1):
        s.append((f"predictor.codec_embedding.{g}.weight", (cfg.predictor_vocab, t.hidden), "mat", std))
    s += _layer_specs("predictor", p, std)
    for g in range(cfg.n_groups - 1):
        s.append((f"predictor.lm_head.{g}.weight", (cfg.predictor_vocab, p.hidden), "mat", 0.08))
    # ---- codec decoder (code2wav) ------------------------------------------------
    s.append(("codec.code_embedding.weight", (c.codebook_size * c.num_quantizers, c.hidden), "mat", 1.0))
    s += _layer_specs("codec.pre_transformer", codec_transformer_dims(cfg), std, layer_scale=True, qk_norm=False)
    for i, r in enumerate(c.upsampling_ratios):
        u = f"codec.upsample.{i}"
        s += [(f"{u}.0.conv.weight", (c.hidden, c.hidden, r), "fanT", 1.0), (f"{u}.0.conv.bias", (c.hidden,), "bias", 0.01),
              (f"{u}.1.dwconv.conv.weight", (c.hidden, 1, 7), "fan", 1.0), (f"{u}.1.dwconv.conv.bias", (c.hidden,), "bias", 0.01),
              (f"{u}.1.norm.weight", (c.hidden,), "norm", 0.1), (f"{u}.1.norm.bias", (c.hidden,), "bias", 0.01),
              (f"{u}.1.pwconv1.weight", (4 * c.hidden, c.hidden), "fan", 1.0), (f"{u}.1.pwconv1.bias", (4 * c.hidden,), "bias", 0.01),
              (f"{u}.1.pwconv2.weight", (c.hidden, 4 * c.hidden), "fan", 1.0), (f"{u}.1.pwconv2.bias", (c.hidden,), "bias", 0.01),
              (f"{u}.1.gamma", (c.hidden,), "const", 0.0)]
    ch = decoder_channels(cfg)
    s += [("codec.decoder.0.conv.weight", (ch[0], c.hidden, 7), "fan", 1.0), ("codec.decoder.0.conv.bias", (ch[0],), "bias", 0.01)]
    for i, r in enumerate(c.upsample_rates):
        b = f"codec.decoder.{i + 1}.block"
        cin, cout = ch[i], ch[i + 1]
        s += [(f"{b}.0.alpha", (cin,), "bias", 0.3), (f"{b}.0.beta", (cin,), "bias", 0.3),
              (f"{b}.1.conv.weight", (cin, cout, 2 * r), "fanT", 1.0), (f"{b}.1.conv.bias", (cout,), "bias", 0.01)]
        for j in range(3):
            u = f"{b}.{j + 2}"
            s += [(f"{u}.act1.alpha", (cout,), "bias", 0.3), (f"{u}.act1.beta", (cout,), "bias", 0.3),
                  (f"{u}.conv1.conv.weight", (cout, cout, 7), "fan", 0.6), (f"{u}.conv1.conv.bias", (cout,), "bias", 0.01),
                  (f"{u}.act2.alpha", (cout,), "bias", 0.3), (f"{u}.act2.beta", (cout,), "bias", 0.3),
                  (f"{u}.conv2.conv.weight", (cout, cout, 1), "fan", 0.6), (f"{u}.conv2.conv.bias", (cout,), "bias", 0.01)]
    n = len(c.upsample_rates) + 1
    s += [(f"codec.decoder.{n}.alpha", (ch[-1],), "bias", 0.3), (f"codec.decoder.{n}.beta", (ch[-1],), "bias", 0.3),
          (f"codec.decoder.{n + 1}.conv.weight", (1, ch[-1], 7), "fan", 0.1), (f"codec.decoder.{n + 1}.conv.bias", (1,), "bias", 0.0)]
    if c.enc_filters > 0:            # the conditioning front-end is optional: CustomVoice checkpoints ship none (enc_filters = 0)
        s += encoder_specs(cfg)
    return s


def encoder_transformer_dims(cfg: ModelConfig) -> TransformerDims:
    c = cfg.codec
    return TransformerDims(hidden=c.enc_hidden, layers=c.enc_layers, heads=c.enc_heads, kv_heads=c.enc_heads, head_dim=c.enc_head_dim,
                           inter=c.enc_inter, rope_theta=c.rope_theta, rms_eps=c.rms_eps)


def encoder_channels(cfg: ModelConfig) -> List[int]:
    c = cfg.codec
    return [c.enc_filters * (2 ** i) for i in range(len(c.enc_ratios) + 1)]


def encoder_specs(cfg: ModelConfig) -> List[Spec]:
    """Conditioning front-end: conv encoder -> transformer -> stride-2 conv -> split RVQ, and the speaker head.
    ``enc.conv.{i}`` numbers the convolutions in execution order: 0 = input conv, then per stage [residual k3, residual k1,
    strided], then the last conv."""
    c = cfg.codec
    ch = encoder_channels(cfg)
    s: List[Spec] = [("enc.conv.0.weight", (ch[0], 1, c.enc_kernel), "fan", 1.0), ("enc.conv.0.bias", (ch[0],), "bias", 0.01)]
    i = 1
    for st, r in enumerate(c.enc_ratios):
        d = ch[st]
        s += [(f"enc.conv.{i}.weight", (d // 2, d, c.enc_res_kernel), "fan", 1.4), (f"enc.conv.{i}.bias", (d // 2,), "bias", 0.01),
              (f"enc.conv.{i + 1}.weight", (d, d // 2, 1), "fan", 1.4), (f"enc.conv.{i + 1}.bias", (d,), "bias", 0.01),
              (f"enc.conv.{i + 2}.weight", (2 * d, d, 2 * r), "fan", 1.4), (f"enc.conv.{i + 2}.bias", (2 * d,), "bias", 0.01)]
        i += 3
    s += [(f"enc.conv.{i}.weight", (c.enc_hidden, ch[-1], c.enc_last_kernel), "fan", 1.4), (f"enc.conv.{i}.bias", (c.enc_hidden,), "bias", 0.01)]
    s += _layer_specs("enc.transformer", encoder_transformer_dims(cfg), 0.02, layer_scale=True, qk_norm=False)
    s += [("enc.downsample.weight", (c.enc_hidden, c.enc_hidden, 4), "fan", 1.0)]
    s += [("enc.vq.semantic.input_proj.weight", (c.vq_dim, c.enc_hidden), "fan", 1.0),
          ("enc.vq.acoustic.input_proj.weight", (c.vq_dim, c.enc_hidden), "fan", 1.0)]
    for q in range(c.num_quantizers):
        s.append((f"enc.vq.codebook.{q}", (c.codebook_size, c.vq_dim), "mat", 0.6 * (0.8 ** min(q, 8))))
    s += [("enc.spk.fc1.weight", (c.spk_hidden, 2 * c.enc_hidden), "fan", 1.0), ("enc.spk.fc1.bias", (c.spk_hidden,), "bias", 0.01),
          ("enc.spk.fc2.weight", (cfg.talker.hidden, c.spk_hidden), "fan", 0.4), ("enc.spk.fc2.bias", (cfg.talker.hidden,), "bias", 0.01)]
    return s


def _const_value(name: str, cfg: ModelConfig) -> float:
    if name.endswith("layer_scale.scale"):
        return 0.25          # synthetic: larger than the trained 0.01 init so the layers matter numerically
    if name.endswith(".gamma"):
        return 0.25
    return 0.0


def _mix32(h: torch.Tensor) -> torch.Tensor:
    """murmur3 fmix32 on int64 tensors holding values < 2^32 (exact on CPU and GPU)."""
    m = 0xFFFFFFFF
    h = h ^ (h >> 16)
    h = (h * 0x85EBCA6B) & m
    h = h ^ (h >> 13)
    h = (h * 0xC2B2AE35) & m
    h = h ^ (h >> 16)
    return h


def hash_uniform(n: int, seed: int, device="cpu", chunk: int = 1 << 24) -> Iterator[Tuple[int, torch.Tensor]]:
    """Yield (offset, float32 tensor in [-0.5, 0.5)) chunks of a length-n stream."""
    for off in range(0, n, chunk):
        m = min(chunk, n - off)
        idx = torch.arange(off, off + m, dtype=torch.int64, device=device)
        h = _mix32((idx ^ (seed & 0xFFFFFFFF)) & 0xFFFFFFFF)
        h = _mix32((h + ((seed >> 32) & 0xFFFFFFFF) + 0x9E3779B9) & 0xFFFFFFFF)
        u = ((h >> 8).to(torch.float32) + 0.5) * (1.0 / (1 << 24)) - 0.5
        yield off, u


def tensor_seed(name: str, base_seed: int) -> int:
    return (zlib.crc32(name.encode()) & 0xFFFFFFFF) | ((base_seed & 0xFFFFFFFF) << 32)


def synth_tensor(spec: Spec, cfg: ModelConfig, base_seed: int, device="cpu") -> torch.Tensor:
    """bf16 tensor for ``spec``; uniform with the requested standard deviation."""
    name, shape, kind, scale = spec
    n = 1
    for d in shape:
        n *= d
    if kind == "const":
        return torch.full(shape, _const_value(name, cfg), dtype=torch.float32, device=device).to(torch.bfloat16)
    if kind == "fan":        # Conv1d / Linear weight [out, in(/groups), k]: fan_in = prod(shape[1:])
        fan = 1
        for d in shape[1:]:
            fan *= d
        std = scale / math.sqrt(fan)
    elif kind == "fanT":     # ConvTranspose1d weight [in, out, k]: each output sums ~ in * k/stride taps; use in*2
        std = scale / math.sqrt(shape[0] * 2)
    else:
        std = scale
    amp = std * math.sqrt(12.0)
    out = torch.empty(n, dtype=torch.bfloat16, device=device)
    for off, u in hash_uniform(n, tensor_seed(name, base_seed), device):
        v = u * amp
        if kind == "norm":
            v = v + 1.0
        out[off:off + u.numel()] = v.to(torch.bfloat16)
    return out.reshape(shape)


def synthetic_state(cfg: ModelConfig, seed: int = 789, device="cpu", only_prefix: str = "") -> Dict[str, torch.Tensor]:
    out: Dict[str, torch.Tensor] = {}
    on_gpu = str(device).startswith("cuda")
    for sp in tensor_specs(cfg):
        if sp[0].startswith(only_prefix):
            out[sp[0]] = synth_tensor(sp, cfg, seed, device)
            if on_gpu and len(out) % 8 == 0:
                # ~10 launches per tensor: keep the queue shallow (rocprofv3 --pmc on this image faults once a few hundred
                # dispatches are queued without a host wait: tools/pmc_probe.py flood / syncflood, profiles/README.md)
                torch.cuda.synchronize()
    return out


# Checkpoint tensor names -> this module's names.  The first group is the layout of the sibling architecture in the
# container's transformers (Qwen3-Omni talker / code predictor / code2wav: modeling_qwen3_omni_moe.py); whether a real
# Qwen3-TTS checkpoint uses exactly these is UNVERIFIED offline (no checkpoint, no qwen-tts source) - a name that matches
# no rule is reported, never guessed.
_HF_RULES = [
    (r"^talker\.model\.codec_embedding\.", "talker.codec_embedding."),
    (r"^talker\.model\.text_embedding\.", "talker.text_embedding."),
    (r"^talker\.model\.(layers\.\d+\.|norm\.)", r"talker.\1"),
    (r"^talker\.text_projection\.linear_fc1\.", "talker.text_projection.fc1."),
    (r"^talker\.text_projection\.linear_fc2\.", "talker.text_projection.fc2."),
    (r"^talker\.code_predictor\.model\.codec_embedding\.", "predictor.codec_embedding."),
    (r"^talker\.code_predictor\.model\.(layers\.\d+\.|norm\.)", r"predictor.\1"),
    (r"^talker\.code_predictor\.lm_head\.", "predictor.lm_head."),
    (r"^talker\.code_predictor\.(small_to_mtp_projection|mtp_proj)\.", "predictor.mtp_proj."),
    (r"^(code2wav|speech_tokenizer\.decoder)\.", "codec."),
    # conditioning front-end: this package's own names under a speech-tokenizer prefix, and the layout of the Mimi codec in the
    # container's transformers (models/mimi/modeling_mimi.py: encoder.layers.N.conv, encoder_transformer.layers.N, downsample.conv,
    # quantizer.{semantic,acoustic}_residual_vector_quantizer) - UNVERIFIED for Qwen3-TTS like everything else here
    (r"^speech_tokenizer\.(encoder\.)?enc\.", "enc."),
    (r"^speech_tokenizer\.encoder_transformer\.(layers\.\d+\.|norm\.)", r"enc.transformer.\1"),
    (r"^speech_tokenizer\.downsample\.conv\.", "enc.downsample."),
    (r"^speech_tokenizer\.quantizer\.semantic_residual_vector_quantizer\.input_proj\.", "enc.vq.semantic.input_proj."),
    (r"^speech_tokenizer\.quantizer\.acoustic_residual_vector_quantizer\.input_proj\.", "enc.vq.acoustic.input_proj."),
    (r"^speech_tokenizer\.speaker_encoder\.", "enc.spk."),
]

# Slots the kernels consume as float32 (nearest-neighbour search of the quantiser, speaker head): a float32 checkpoint tensor
# stays float32 for them - rounding a codebook to bf16 first can move a code decision.
_F32_SLOTS = ("enc.vq.codebook.", "enc.spk.")


def remap_name(name: str) -> str:
    """A checkpoint's tensor name in this module's naming (identity when it already is)."""
    import re
    for pat, rep in _HF_RULES:
        new, n = re.subn(pat, rep, name)
        if n:
            return new
    return name


def checkpoint_files(model_dir: str) -> List[str]:
    """``*.safetensors`` in ``model_dir`` and one level of sub-folders (a speech-tokenizer folder next to the talker's shards)."""
    out = []
    for root, dirs, files in os.walk(model_dir):
        if root != model_dir and os.path.dirname(root) != model_dir.rstrip("/"):
            continue
        out += [os.path.join(root, f) for f in sorted(files) if f.endswith(".safetensors")]
    return sorted(out)


def load_safetensors(cfg: ModelConfig, model_dir: str, device="cpu") -> Dict[str, torch.Tensor]:
    """Load a checkpoint (this module's tensor names, or names `remap_name` knows) as bf16.  safetensors only:
    nothing is executed from the files."""
    from safetensors import safe_open

    files = checkpoint_files(model_dir)
    if not files:
        raise FileNotFoundError(f"no .safetensors files in {model_dir}")
    want = {sp[0]: sp[1] for sp in tensor_specs(cfg)}
    enc_names = {sp[0] for sp in encoder_specs(cfg)} if cfg.codec.enc_filters > 0 else set()
    state: Dict[str, torch.Tensor] = {}
    unknown: List[str] = []
    for path in files:
        sub = os.path.relpath(os.path.dirname(path), model_dir)
        with safe_open(path, framework="pt", device=str(device)) as sf:
            for k in sf.keys():
                name = remap_name(k if sub == "." else f"{sub}.{k}")
                if name not in want:
                    name = remap_name(k)
                if name in want:
                    tns = sf.get_tensor(k)
                    if tuple(tns.shape) != tuple(want[name]):
                        raise ValueError(f"{k}: checkpoint shape {tuple(tns.shape)} != configured {want[name]}")
                    keep_f32 = tns.dtype == torch.float32 and name.startswith(_F32_SLOTS)
                    state[name] = tns if keep_f32 else tns.to(torch.bfloat16)
                else:
                    unknown.append(k)
    if enc_names and not (enc_names & set(state)):
        # no encoder tensors at all: a checkpoint without the conditioning front-end (CustomVoice models never use it).  The
        # model is built without one - rt_voice_encode then answers RT_ERR_UNSUPPORTED, as include/rho_tts_amd.h documents.
        cfg.codec.enc_filters = 0
        for k in enc_names:
            want.pop(k, None)
    missing = sorted(set(want) - set(state))
    if missing:
        raise ValueError(f"checkpoint is missing {len(missing)} of {len(want)} tensors, e.g. {missing[:4]}"
                         + (f"; {len(unknown)} tensors in the files matched no known name, e.g. {sorted(unknown)[:4]}" if unknown else ""))
    return state


def save_checkpoint(cfg: ModelConfig, state: Dict[str, torch.Tensor], model_dir: str, shard_bytes: int = 1 << 30) -> List[str]:
    """Write ``state`` as ``model-0000k.safetensors`` shards plus this package's ``config.json`` (what ``resolve`` reads back)."""
    from safetensors.torch import save_file

    os.makedirs(model_dir, exist_ok=True)
    shards, cur, size = [], {}, 0
    for k in sorted(state):
        t = state[k].detach().to("cpu").contiguous()
        if cur and size + t.numel() * t.element_size() > shard_bytes:
            shards.append(cur)
            cur, size = {}, 0
        cur[k] = t
        size += t.numel() * t.element_size()
    if cur:
        shards.append(cur)
    paths = []
    for i, sh in enumerate(shards):
        path = os.path.join(model_dir, f"model-{i + 1:05d}-of-{len(shards):05d}.safetensors")
        save_file(sh, path)
        paths.append(path)
    with open(os.path.join(model_dir, "config.json"), "w") as f:
        f.write(cfg.to_json())
    return paths


def param_count(cfg: ModelConfig) -> int:
    tot = 0
    for _, shape, _, _ in tensor_specs(cfg):
        n = 1
        for d in shape:
            n *= d
        tot += n
    return tot
