"""ctypes binding of the model group of the C ABI (include/rho_tts_amd.h, "model") and the
host-side re-layout of a checkpoint into the GEMM-ready tensors the library expects."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from .config import ModelConfig, TransformerDims
from .weights import codec_transformer_dims, decoder_channels, encoder_transformer_dims

DT_BF16, DT_F32 = 0, 1


class StackDims(C.Structure):
    _fields_ = [("hidden", C.c_int32), ("layers", C.c_int32), ("heads", C.c_int32), ("kv_heads", C.c_int32),
                ("head_dim", C.c_int32), ("inter", C.c_int32), ("rope_theta", C.c_float), ("rms_eps", C.c_float)]


class RtEncoderConfig(C.Structure):
    _fields_ = [("filters", C.c_int32), ("n_ratios", C.c_int32), ("ratios", C.c_int32 * 8), ("kernel", C.c_int32), ("res_kernel", C.c_int32),
                ("last_kernel", C.c_int32), ("tf", StackDims), ("window", C.c_int32), ("vq_dim", C.c_int32), ("spk_hidden", C.c_int32),
                ("max_ref_frames", C.c_int32)]


class RtModelConfig(C.Structure):
    _fields_ = [("talker", StackDims), ("predictor", StackDims), ("codec_tf", StackDims),
                ("codec_vocab", C.c_int32), ("predictor_vocab", C.c_int32), ("text_vocab", C.c_int32),
                ("text_hidden", C.c_int32), ("n_groups", C.c_int32), ("codebook_size", C.c_int32),
                ("num_quantizers", C.c_int32), ("codec_sliding_window", C.c_int32), ("n_upsampling", C.c_int32),
                ("upsampling_ratios", C.c_int32 * 4), ("n_upsample_rates", C.c_int32), ("upsample_rates", C.c_int32 * 8),
                ("decoder_dim", C.c_int32), ("codec_eos_id", C.c_int32), ("max_batch", C.c_int32),
                ("max_positions", C.c_int32), ("max_codec_frames", C.c_int32), ("reserved", C.c_int32 * 4), ("enc", RtEncoderConfig)]


class RtSampling(C.Structure):
    _fields_ = [("do_sample", C.c_int32), ("temperature", C.c_float), ("top_k", C.c_int32), ("top_p", C.c_float),
                ("repetition_penalty", C.c_float)]


class RtGenerateArgs(C.Structure):
    _fields_ = [("n_items", C.c_int32), ("h_text_ids", C.POINTER(C.c_int32)), ("h_text_offsets", C.POINTER(C.c_int32)),
                ("h_max_frames", C.POINTER(C.c_int32)), ("h_item_ids", C.POINTER(C.c_int64)), ("seed", C.c_uint64),
                ("talker", RtSampling), ("predictor", RtSampling), ("ignore_eos", C.c_int32), ("min_frames", C.c_int32),
                ("tts_eos_id", C.c_int32), ("tts_pad_id", C.c_int32), ("codec_pad_id", C.c_int32), ("codec_bos_id", C.c_int32),
                ("h_forced_codes", C.POINTER(C.c_int32)), ("h_forced_offsets", C.POINTER(C.c_int32)),
                ("h_cancel_flag", C.POINTER(C.c_int32)), ("h_codes", C.POINTER(C.c_int32)), ("h_n_frames", C.POINTER(C.c_int32)),
                ("d_trace_talker", C.c_void_p), ("d_trace_predictor", C.c_void_p), ("max_rows", C.c_int32)]


def declare(lib: C.CDLL) -> None:
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    lib.rt_model_create.argtypes = [vp, C.POINTER(RtModelConfig), C.POINTER(vp)]
    lib.rt_model_destroy.argtypes = [vp]
    lib.rt_model_tensor_count.argtypes = [vp]
    lib.rt_model_tensor_info.argtypes = [vp, i32, C.c_char_p, C.c_size_t, C.POINTER(i64), C.POINTER(i32)]
    lib.rt_model_set_tensor.argtypes = [vp, C.c_char_p, vp, i32, i64, i64, i32]
    lib.rt_model_finalize.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    lib.rt_model_weight_bytes.restype = i64
    lib.rt_model_weight_bytes.argtypes = [vp]
    lib.rt_model_set_voice.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(i32), i32, C.POINTER(C.c_float)]
    lib.rt_voice_encode.argtypes = [vp, C.POINTER(C.c_float), i64, C.POINTER(i32), i32, C.POINTER(i32), C.POINTER(C.c_float)]
    lib.rt_model_set_voice_pcm.argtypes = [vp, C.POINTER(C.c_float), i64, i32, C.POINTER(i32), C.POINTER(i32), i32, i32, i32, C.POINTER(i32), C.POINTER(i32)]
    lib.rt_voice_prefix_len.argtypes = [vp]
    lib.rt_voice_blob_bytes.restype = i64
    lib.rt_voice_blob_bytes.argtypes = [vp]
    lib.rt_voice_export.argtypes = [vp, vp, i64]
    lib.rt_voice_import.argtypes = [vp, i32, vp, i64]
    lib.rt_generate.argtypes = [vp, C.POINTER(RtGenerateArgs)]
    lib.rt_generate_stats.argtypes = [vp] + [C.POINTER(C.c_int64)] * 4
    lib.rt_generate_begin.argtypes = [vp, C.POINTER(RtGenerateArgs)]
    lib.rt_generate_step.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(i32)]
    lib.rt_generate_peek.argtypes = [vp, i32, i32, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    lib.rt_generate_end.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    lib.rt_wav_length.restype = i64
    lib.rt_wav_length.argtypes = [vp, i32]
    lib.rt_code2wav.argtypes = [vp, i32, i32, C.POINTER(i32), C.POINTER(i32), vp, i64, C.POINTER(i64)]
    lib.rt_profile_enable.argtypes = [vp, i32]
    lib.rt_profile_read.argtypes = [vp, C.POINTER(i64), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.rt_profile_read_class.argtypes = [vp, i32, C.POINTER(i64), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.rt_debug_gemm.argtypes = [vp, vp, i32, i64, i32, i32, i32, i32, i32, i32, vp, i32, vp, i32, vp, i32, i32]
    lib.rt_debug_attention.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, i32, vp, vp, i32, i32, vp]
    f32 = C.c_float
    lib.rt_debug_gemm_col.argtypes = [vp, vp, i32, i32, vp, i32, i32, i32, i32, i32, vp, i32, f32, vp, vp, vp, vp, vp, vp, vp]
    lib.rt_debug_attention_fused.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, f32, vp, vp, vp, vp, i32, vp, vp, i32, i32, i32, i32, vp]
    lib.rt_debug_attention_prefill.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]
    lib.rt_debug_sample.argtypes = [vp, vp, i32, i32, C.POINTER(RtSampling), C.c_uint64, i32, i32, i32, i32, vp, vp]


def _stack(d: TransformerDims) -> StackDims:
    return StackDims(d.hidden, d.layers, d.heads, d.kv_heads, d.head_dim, d.inter, d.rope_theta, d.rms_eps)


def rt_config(cfg: ModelConfig, max_batch: int, max_positions: int, max_codec_frames: int) -> RtModelConfig:
    c = RtModelConfig()
    c.talker, c.predictor, c.codec_tf = _stack(cfg.talker), _stack(cfg.predictor), _stack(codec_transformer_dims(cfg))
    c.codec_vocab, c.predictor_vocab, c.text_vocab = cfg.codec_vocab, cfg.predictor_vocab, cfg.text_vocab
    c.text_hidden, c.n_groups = cfg.text_hidden, cfg.n_groups
    c.codebook_size, c.num_quantizers = cfg.codec.codebook_size, cfg.codec.num_quantizers
    c.codec_sliding_window = cfg.codec.sliding_window
    c.n_upsampling = len(cfg.codec.upsampling_ratios)
    for i, r in enumerate(cfg.codec.upsampling_ratios):
        c.upsampling_ratios[i] = r
    c.n_upsample_rates = len(cfg.codec.upsample_rates)
    for i, r in enumerate(cfg.codec.upsample_rates):
        c.upsample_rates[i] = r
    c.decoder_dim = cfg.codec.decoder_dim
    c.codec_eos_id = cfg.codec_eos_id
    c.max_batch, c.max_positions, c.max_codec_frames = max_batch, max_positions, max_codec_frames
    e, k = c.enc, cfg.codec
    e.filters, e.n_ratios = k.enc_filters, len(k.enc_ratios)
    for i, r in enumerate(k.enc_ratios):
        e.ratios[i] = r
    e.kernel, e.res_kernel, e.last_kernel = k.enc_kernel, k.enc_res_kernel, k.enc_last_kernel
    e.tf = _stack(encoder_transformer_dims(cfg))
    e.window, e.vq_dim, e.spk_hidden = k.enc_window, k.vq_dim, k.spk_hidden
    e.max_ref_frames = max(1, max_positions // 2)          # a reference clip may take half of the KV rows (Engine.set_voice_from_audio)
    return c


def rope_table(head_dim: int, theta: float, n_pos: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """cos/sin [n_pos, head_dim/2] float32, computed on the host (rotate-half RoPE, default frequencies)."""
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    fr = torch.arange(n_pos, dtype=torch.float32)[:, None] * inv[None, :]
    return fr.cos().contiguous(), fr.sin().contiguous()


def to_native(state: Dict[str, torch.Tensor], cfg: ModelConfig) -> Dict[str, torch.Tensor]:
    """Checkpoint names/layouts -> the library's GEMM-ready tensors (bf16 matrices [N, K]; f32 vectors).

    Conv1d weight [Co, Ci, k]            -> [Co, k*Ci]   (column = tap*Ci + ci, channels-last im2col order)
    ConvTranspose1d [Ci, Co, r] (k = r)  -> [r*Co, Ci]   (row = phase*Co + co)
    ConvTranspose1d [Ci, Co, 2r]         -> [r*Co, 2*Ci] (tap 0 = x[m] with kernel index phase + r,
                                                          tap 1 = x[m+1] with kernel index phase)
    SnakeBeta alpha/beta                 -> exp(alpha), 1/(exp(beta)+1e-9) in float32
    """
    out: Dict[str, torch.Tensor] = {}
    bf = torch.bfloat16

    def f32(t):
        return t.detach().to(torch.float32).contiguous()

    def stack(src: str, dst: str, d: TransformerDims, qk_norm: bool, layer_scale: bool):
        for i in range(d.layers):
            s, t = f"{src}.layers.{i}", f"{dst}.l{i}"
            out[f"{t}.wqkv"] = torch.cat([state[f"{s}.self_attn.q_proj.weight"], state[f"{s}.self_attn.k_proj.weight"],
                                          state[f"{s}.self_attn.v_proj.weight"]], 0).to(bf).contiguous()
            out[f"{t}.wo"] = state[f"{s}.self_attn.o_proj.weight"].to(bf).contiguous()
            out[f"{t}.wgu"] = torch.cat([state[f"{s}.mlp.gate_proj.weight"], state[f"{s}.mlp.up_proj.weight"]], 0).to(bf).contiguous()
            out[f"{t}.wd"] = state[f"{s}.mlp.down_proj.weight"].to(bf).contiguous()
            out[f"{t}.ln1"] = f32(state[f"{s}.input_layernorm.weight"])
            out[f"{t}.ln2"] = f32(state[f"{s}.post_attention_layernorm.weight"])
            if qk_norm:
                out[f"{t}.qn"] = f32(state[f"{s}.self_attn.q_norm.weight"])
                out[f"{t}.kn"] = f32(state[f"{s}.self_attn.k_norm.weight"])
            if layer_scale:
                out[f"{t}.ls1"] = f32(state[f"{s}.self_attn_layer_scale.scale"])
                out[f"{t}.ls2"] = f32(state[f"{s}.mlp_layer_scale.scale"])
        out[f"{dst}.norm"] = f32(state[f"{src}.norm.weight"])

    out["talker.text_embedding"] = state["talker.text_embedding.weight"].to(bf).contiguous()
    out["talker.tp_fc1"] = state["talker.text_projection.fc1.weight"].to(bf).contiguous()
    out["talker.tp_fc1_b"] = f32(state["talker.text_projection.fc1.bias"])
    out["talker.tp_fc2"] = state["talker.text_projection.fc2.weight"].to(bf).contiguous()
    out["talker.tp_fc2_b"] = f32(state["talker.text_projection.fc2.bias"])
    out["talker.codec_embedding"] = state["talker.codec_embedding.weight"].to(bf).contiguous()
    out["talker.codec_head"] = state["talker.codec_head.weight"].to(bf).contiguous()
    stack("talker", "talker", cfg.talker, True, False)
    if cfg.has_mtp_proj:
        out["pred.mtp"] = state["predictor.mtp_proj.weight"].to(bf).contiguous()
        out["pred.mtp_b"] = f32(state["predictor.mtp_proj.bias"])
    for g in range(cfg.n_groups - 1):
        out[f"pred.emb{g}"] = state[f"predictor.codec_embedding.{g}.weight"].to(bf).contiguous()
        out[f"pred.head{g}"] = state[f"predictor.lm_head.{g}.weight"].to(bf).contiguous()
    stack("predictor", "pred", cfg.predictor, True, False)
    out["codec.code_embedding"] = state["codec.code_embedding.weight"].to(bf).contiguous()
    stack("codec.pre_transformer", "ctf", codec_transformer_dims(cfg), False, True)

    def conv_mat(w):      # [Co, Ci, k] -> [Co, k*Ci]
        return w.permute(0, 2, 1).reshape(w.shape[0], -1).to(bf).contiguous()

    def snake(alpha, beta):
        a = torch.exp(f32(alpha).cpu())
        ib = 1.0 / (torch.exp(f32(beta).cpu()) + 1e-9)
        return a.to(alpha.device), ib.to(alpha.device)

    c = cfg.codec
    for i, r in enumerate(c.upsampling_ratios):
        s, t = f"codec.upsample.{i}", f"codec.up{i}"
        w = state[f"{s}.0.conv.weight"]                                     # [Ci, Co, r]
        out[f"{t}.tconv"] = w.permute(2, 1, 0).reshape(r * w.shape[1], w.shape[0]).to(bf).contiguous()
        out[f"{t}.tconv_b"] = f32(state[f"{s}.0.conv.bias"]).repeat(r)
        out[f"{t}.dw_w"] = f32(state[f"{s}.1.dwconv.conv.weight"][:, 0, :].t()).reshape(-1)   # [7, C]
        out[f"{t}.dw_b"] = f32(state[f"{s}.1.dwconv.conv.bias"])
        out[f"{t}.ln_w"] = f32(state[f"{s}.1.norm.weight"])
        out[f"{t}.ln_b"] = f32(state[f"{s}.1.norm.bias"])
        out[f"{t}.pw1"] = state[f"{s}.1.pwconv1.weight"].to(bf).contiguous()
        out[f"{t}.pw1_b"] = f32(state[f"{s}.1.pwconv1.bias"])
        out[f"{t}.pw2"] = state[f"{s}.1.pwconv2.weight"].to(bf).contiguous()
        out[f"{t}.pw2_b"] = f32(state[f"{s}.1.pwconv2.bias"])
        out[f"{t}.gamma"] = f32(state[f"{s}.1.gamma"])
    out["codec.dec0"] = conv_mat(state["codec.decoder.0.conv.weight"])
    out["codec.dec0_b"] = f32(state["codec.decoder.0.conv.bias"])
    for i, r in enumerate(c.upsample_rates):
        s, t = f"codec.decoder.{i + 1}.block", f"codec.b{i}"
        out[f"{t}.sa"], out[f"{t}.sib"] = snake(state[f"{s}.0.alpha"], state[f"{s}.0.beta"])
        w = state[f"{s}.1.conv.weight"]                                     # [Ci, Co, 2r]
        ci, co = w.shape[0], w.shape[1]
        tap0 = w[:, :, r:].permute(2, 1, 0)                                 # [phase, Co, Ci] for x[m]
        tap1 = w[:, :, :r].permute(2, 1, 0)                                 # [phase, Co, Ci] for x[m+1]
        out[f"{t}.tconv"] = torch.cat([tap0, tap1], dim=2).reshape(r * co, 2 * ci).to(bf).contiguous()
        out[f"{t}.tconv_b"] = f32(state[f"{s}.1.conv.bias"]).repeat(r)
        for j in range(3):
            u, v = f"{s}.{j + 2}", f"{t}.u{j}"
            out[f"{v}.a1"], out[f"{v}.ib1"] = snake(state[f"{u}.act1.alpha"], state[f"{u}.act1.beta"])
            out[f"{v}.c1"] = conv_mat(state[f"{u}.conv1.conv.weight"])
            out[f"{v}.c1_b"] = f32(state[f"{u}.conv1.conv.bias"])
            out[f"{v}.a2"], out[f"{v}.ib2"] = snake(state[f"{u}.act2.alpha"], state[f"{u}.act2.beta"])
            out[f"{v}.c2"] = conv_mat(state[f"{u}.conv2.conv.weight"])
            out[f"{v}.c2_b"] = f32(state[f"{u}.conv2.conv.bias"])
    n = len(c.upsample_rates) + 1
    out["codec.fin_a"], out["codec.fin_ib"] = snake(state[f"codec.decoder.{n}.alpha"], state[f"codec.decoder.{n}.beta"])
    out["codec.fin_w"] = conv_mat(state[f"codec.decoder.{n + 1}.conv.weight"])                  # [1, 7*C]
    out["codec.fin_wv"] = f32(out["codec.fin_w"]).reshape(-1)                                     # the same taps as a plain vector
    out["codec.fin_b"] = f32(state[f"codec.decoder.{n + 1}.conv.bias"])
    # ---- conditioning front-end (every conv, strided ones included, is [Co][k*Ci] with column = tap*Ci + ci: a k = 2r, stride r
    # conv over [T][Ci] is the 2-tap conv over the clip viewed as [T/r][r*Ci], whose column order is the same)
    if c.enc_filters <= 0:            # a checkpoint without the conditioning front-end (weights.load_safetensors)
        return out
    out["enc.conv0_w"] = f32(state["enc.conv.0.weight"][:, 0, :]).reshape(-1)
    out["enc.conv0_b"] = f32(state["enc.conv.0.bias"])
    for i in range(1, 2 + 3 * len(c.enc_ratios)):
        out[f"enc.c{i}"] = conv_mat(state[f"enc.conv.{i}.weight"])
        out[f"enc.c{i}_b"] = f32(state[f"enc.conv.{i}.bias"])
    stack("enc.transformer", "etf", encoder_transformer_dims(cfg), False, True)
    out["enc.down"] = conv_mat(state["enc.downsample.weight"])
    out["enc.vq_sem"] = state["enc.vq.semantic.input_proj.weight"].to(bf).contiguous()
    out["enc.vq_aco"] = state["enc.vq.acoustic.input_proj.weight"].to(bf).contiguous()
    for q in range(c.num_quantizers):
        out[f"enc.cbT{q}"] = f32(state[f"enc.vq.codebook.{q}"]).t().contiguous().reshape(-1)      # [D][K]: entries contiguous per dimension
    out["enc.spk_fc1"], out["enc.spk_fc1_b"] = f32(state["enc.spk.fc1.weight"]).reshape(-1), f32(state["enc.spk.fc1.bias"])
    out["enc.spk_fc2"], out["enc.spk_fc2_b"] = f32(state["enc.spk.fc2.weight"]).reshape(-1), f32(state["enc.spk.fc2.bias"])
    return out


class NativeModel:
    """One ``rt_model``: weights, KV caches and workspaces of a Qwen3-TTS-shaped model in HBM."""

    def __init__(self, ctx, cfg: ModelConfig, max_batch: int = 32, max_positions: Optional[int] = None,
                 max_codec_frames: Optional[int] = None):
        self.ctx, self.cfg, self.lib = ctx, cfg, ctx.lib
        self.max_batch = max_batch
        self.max_positions = max_positions or cfg.max_positions
        self.max_codec_frames = max_codec_frames or (cfg.codec.chunk_frames + cfg.codec.left_context_frames)
        self.rt_cfg = rt_config(cfg, max_batch, self.max_positions, self.max_codec_frames)
        h = C.c_void_p()
        ctx.check(self.lib.rt_model_create(ctx.handle, C.byref(self.rt_cfg), C.byref(h)), "rt_model_create")
        self.handle = h
        self._keep = []

    def close(self):
        if getattr(self, "handle", None):
            self.lib.rt_model_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    # ------------------------------------------------------------------ weights
    def tensor_inventory(self) -> List[Tuple[str, Tuple[int, int], int]]:
        n = self.lib.rt_model_tensor_count(self.handle)
        out = []
        for i in range(n):
            name = C.create_string_buffer(128)
            shp = (C.c_int64 * 2)()
            kind = C.c_int32()
            self.ctx.check(self.lib.rt_model_tensor_info(self.handle, i, name, 128, shp, C.byref(kind)), "rt_model_tensor_info")
            out.append((name.value.decode(), (shp[0], shp[1]), kind.value))
        return out

    def load_state(self, state: Dict[str, torch.Tensor]) -> None:
        """Upload a checkpoint-named state dict (bf16 tensors, CPU or GPU) and finalize."""
        native = to_native(state, self.cfg)
        for name, shape, kind in self.tensor_inventory():
            t = native.pop(name)
            rows, cols = (t.shape[0], t.shape[1]) if t.dim() == 2 else (t.numel(), 1)
            dt = DT_BF16 if t.dtype == torch.bfloat16 else DT_F32
            t = t.contiguous()
            if t.is_cuda:
                torch.cuda.current_stream(t.device).synchronize()
            rc = self.lib.rt_model_set_tensor(self.handle, name.encode(), C.c_void_p(t.data_ptr()), dt, rows, cols, 1 if t.is_cuda else 0)
            self.ctx.check(rc, f"rt_model_set_tensor({name})")
        if native:
            raise ValueError(f"tensors not consumed by the library: {sorted(native)[:4]}")
        cos, sin = [], []
        dims = [self.cfg.talker, self.cfg.predictor, codec_transformer_dims(self.cfg)]
        npos = [self.max_positions, self.cfg.n_groups + 1, self.max_codec_frames]
        for d, n in zip(dims, npos):
            c_, s_ = rope_table(d.head_dim, d.rope_theta, n)
            cos.append(c_)
            sin.append(s_)
        cp = (C.c_void_p * 3)(*[C.c_void_p(t.data_ptr()) for t in cos])
        sp = (C.c_void_p * 3)(*[C.c_void_p(t.data_ptr()) for t in sin])
        self.ctx.check(self.lib.rt_model_finalize(self.handle, cp, sp), "rt_model_finalize")

    def weight_bytes(self) -> int:
        return int(self.lib.rt_model_weight_bytes(self.handle))

    # ------------------------------------------------------------------ voice
    def prefix_recipe(self, language: str, speaker: Optional[str], speaker_embed, ref_text_ids: Sequence[int], ref_codes):
        """Row recipe of the shared prompt prefix (DESIGN.md 'Prompt layout')."""
        c = self.cfg
        G = c.n_groups
        lang = c.language_ids.get(language.lower())
        if lang is None:
            raise ValueError(f"unsupported language {language!r}")
        text, codec = [], []

        def row(tid, cids=()):
            text.append(tid)
            r = [-1] * G
            for g, v in enumerate(cids):
                r[g] = int(v)
            codec.append(r)

        for t in c.role_ids:
            row(t)
        for cid in (c.codec_nothink_id, c.codec_think_bos_id, lang, c.codec_think_eos_id):
            row(c.tts_pad_id, [cid])
        spk_row = -1
        if speaker_embed is not None:
            spk_row = len(text)
            row(c.tts_pad_id)
        elif speaker is not None:
            sid = c.speaker_ids.get(speaker.lower())
            if sid is None:
                raise ValueError(f"unknown speaker {speaker!r}")
            row(c.tts_pad_id, [sid])
        else:
            raise ValueError("voice needs a speaker embedding (clone) or a built-in speaker")
        row(c.tts_bos_id, [c.codec_pad_id])
        if ref_codes is not None and len(ref_codes) > 0:
            for t in ref_text_ids:
                row(int(t), [c.codec_pad_id])
            row(c.tts_pad_id, [c.codec_bos_id])
            for fr in ref_codes.tolist():
                row(c.tts_pad_id, fr)
        return text, codec, spk_row

    def set_voice(self, language="english", speaker=None, speaker_embed=None, ref_text_ids=(), ref_codes=None) -> int:
        text, codec, spk_row = self.prefix_recipe(language, speaker, speaker_embed, ref_text_ids, ref_codes)
        n = len(text)
        G = self.cfg.n_groups
        t_arr = (C.c_int32 * n)(*text)
        c_arr = (C.c_int32 * (n * G))(*[v for r in codec for v in r])
        emb = None
        if spk_row >= 0:
            e = speaker_embed.detach().to("cpu", torch.float32).contiguous()
            if e.numel() != self.cfg.talker.hidden:
                raise ValueError("speaker embedding has the wrong size")
            emb = (C.c_float * e.numel())(*e.tolist())
        self.ctx.check(self.lib.rt_model_set_voice(self.handle, n, t_arr, c_arr, spk_row, emb), "rt_model_set_voice")
        return n

    # ------------------------------------------------------------------ conditioning front-end
    def encode_voice(self, pcm, max_frames: Optional[int] = None):
        """Reference audio (1-D float32 at cfg.sample_rate, host) -> (codes [T, num_quantizers] int64, speaker embedding [hidden])."""
        import numpy as np
        x = np.ascontiguousarray(np.asarray(pcm, dtype=np.float32).reshape(-1))
        cap = int(max_frames or self.rt_cfg.enc.max_ref_frames)
        Q = self.cfg.codec.num_quantizers
        codes = (C.c_int32 * (cap * Q))()
        nf = C.c_int32()
        spk = (C.c_float * self.cfg.talker.hidden)()
        rc = self.lib.rt_voice_encode(self.handle, x.ctypes.data_as(C.POINTER(C.c_float)), x.size, codes, cap, C.byref(nf), spk)
        self.ctx.check(rc, "rt_voice_encode")
        out = torch.tensor(list(codes[: nf.value * Q]), dtype=torch.int64).reshape(nf.value, Q)
        return out, torch.tensor(list(spk), dtype=torch.float32)

    def set_voice_pcm(self, pcm, language="english", ref_text_ids=(), max_frames: Optional[int] = None):
        """rt_model_set_voice_pcm: encode the clip and install the voice prefix in one call.  Returns (prefix rows, codes)."""
        import numpy as np
        x = np.ascontiguousarray(np.asarray(pcm, dtype=np.float32).reshape(-1))
        G = self.cfg.n_groups
        one = torch.zeros(1, G, dtype=torch.int64)
        text, codec, spk_row = self.prefix_recipe(language, None, torch.zeros(self.cfg.talker.hidden), ref_text_ids, one)
        text, codec = text[:-1], codec[:-1]                    # the recipe's head: everything before the reference frames
        n = len(text)
        cap = int(max_frames or self.rt_cfg.enc.max_ref_frames)
        codes = (C.c_int32 * (cap * G))()
        nf = C.c_int32()
        rc = self.lib.rt_model_set_voice_pcm(self.handle, x.ctypes.data_as(C.POINTER(C.c_float)), x.size, n, (C.c_int32 * n)(*text),
                                             (C.c_int32 * (n * G))(*[v for r in codec for v in r]), spk_row, self.cfg.tts_pad_id, cap, codes, C.byref(nf))
        self.ctx.check(rc, "rt_model_set_voice_pcm")
        return n + nf.value, torch.tensor(list(codes[: nf.value * G]), dtype=torch.int64).reshape(nf.value, G)

    def prefix_len(self) -> int:
        return int(self.lib.rt_voice_prefix_len(self.handle))

    def export_voice(self) -> torch.Tensor:
        nbytes = int(self.lib.rt_voice_blob_bytes(self.handle))
        blob = torch.empty(nbytes // 2, dtype=torch.bfloat16, device=f"cuda:{self.ctx.device_ordinal}")
        self.ctx.check(self.lib.rt_voice_export(self.handle, C.c_void_p(blob.data_ptr()), nbytes), "rt_voice_export")
        return blob

    def import_voice(self, prefix_len: int, blob: torch.Tensor) -> None:
        torch.cuda.current_stream(blob.device).synchronize()
        self.ctx.check(self.lib.rt_voice_import(self.handle, prefix_len, C.c_void_p(blob.data_ptr()), blob.numel() * 2), "rt_voice_import")

    # ------------------------------------------------------------------ decode
    def _generate_args(self, texts, max_frames, talker, predictor, seed, item_ids, ignore_eos, min_frames, forced_codes, cancel_flag, max_rows):
        """rt_generate_args for a list of token-id lists; returns (args, objects that must stay alive while the call runs)."""
        c = self.cfg
        B = len(texts)
        talker = talker or RtSampling(0, 0.9, 50, 1.0, 1.0)
        predictor = predictor or talker
        flat = [int(t) for tx in texts for t in tx]
        offs = [0]
        for tx in texts:
            offs.append(offs[-1] + len(tx))
        a = RtGenerateArgs()
        a.n_items = B
        ids = (C.c_int32 * max(1, len(flat)))(*flat)
        a.h_text_ids = ids
        a.h_text_offsets = (C.c_int32 * (B + 1))(*offs)
        a.h_max_frames = (C.c_int32 * B)(*[int(v) for v in max_frames])
        a.h_item_ids = (C.c_int64 * B)(*[int(v) for v in (item_ids if item_ids is not None else range(B))])
        a.seed = seed
        a.talker, a.predictor = talker, predictor
        a.ignore_eos, a.min_frames = int(ignore_eos), min_frames
        a.max_rows = int(max_rows)
        a.tts_eos_id, a.tts_pad_id, a.codec_pad_id, a.codec_bos_id = c.tts_eos_id, c.tts_pad_id, c.codec_pad_id, c.codec_bos_id
        keep = [ids]
        if forced_codes is not None:
            fo = [0]
            ff = []
            for fc in forced_codes:
                fo.append(fo[-1] + fc.shape[0])
                ff += [int(v) for v in fc.reshape(-1).tolist()]
            fa = (C.c_int32 * max(1, len(ff)))(*ff)
            a.h_forced_codes = fa
            a.h_forced_offsets = (C.c_int32 * (B + 1))(*fo)
            keep.append(fa)
        if cancel_flag is not None:
            a.h_cancel_flag = C.pointer(cancel_flag)          # a ctypes.c_int32 another thread may set to 1
            keep.append(cancel_flag)
        return a, keep

    def generate(self, texts: Sequence[Sequence[int]], max_frames: Sequence[int], talker=None, predictor=None, seed: int = 789,
                 item_ids: Optional[Sequence[int]] = None, ignore_eos: bool = True, min_frames: int = 2,
                 forced_codes: Optional[Sequence[torch.Tensor]] = None, trace: bool = False, cancel_flag=None, max_rows: int = 0):
        c = self.cfg
        B, G = len(texts), c.n_groups
        a, keep = self._generate_args(texts, max_frames, talker, predictor, seed, item_ids, ignore_eos, min_frames, forced_codes, cancel_flag, max_rows)
        tot = sum(int(v) for v in max_frames)
        codes = (C.c_int32 * (tot * G))()
        nfr = (C.c_int32 * B)()
        a.h_codes, a.h_n_frames = codes, nfr
        tr = {}
        if trace:
            tmax = max(int(v) for v in max_frames)
            dev = f"cuda:{self.ctx.device_ordinal}"
            tr["talker"] = torch.zeros(tmax, B, c.codec_vocab, device=dev)
            tr["predictor"] = torch.zeros(tmax, G - 1, B, c.predictor_vocab, device=dev)
            torch.cuda.synchronize()
            a.d_trace_talker = tr["talker"].data_ptr()
            a.d_trace_predictor = tr["predictor"].data_ptr()
        self.ctx.check(self.lib.rt_generate(self.handle, C.byref(a)), "rt_generate")
        out, off = [], 0
        import numpy as np
        flat_codes = (torch.from_numpy(np.frombuffer(codes, dtype=np.int32).astype(np.int64)).reshape(-1, G) if tot
                      else torch.zeros(0, G, dtype=torch.int64))     # (a view of the ctypes buffer: no per-element Python objects)
        for b in range(B):
            out.append(flat_codes[off: off + nfr[b]].clone())
            off += int(max_frames[b])
        return (out, tr) if trace else out

    # ---- the same generation in pieces (rt_generate_begin / _step / _peek / _end): sub-segment streaming
    def generate_begin(self, texts, max_frames, talker=None, predictor=None, seed: int = 789, item_ids=None, ignore_eos: bool = True,
                       min_frames: int = 2, cancel_flag=None, max_rows: int = 0) -> None:
        a, keep = self._generate_args(texts, max_frames, talker, predictor, seed, item_ids, ignore_eos, min_frames, None, cancel_flag, max_rows)
        self._run_keep = (keep, [int(v) for v in max_frames])
        self.ctx.check(self.lib.rt_generate_begin(self.handle, C.byref(a)), "rt_generate_begin")

    def generate_step(self, n_frames: int):
        """Run up to ``n_frames`` more frames; returns (frames run so far, every item has ended)."""
        run, done = C.c_int32(), C.c_int32()
        self.ctx.check(self.lib.rt_generate_step(self.handle, int(n_frames), C.byref(run), C.byref(done)), "rt_generate_step")
        return int(run.value), bool(done.value)

    def generate_peek(self, item: int, first_frame: int, max_frames: int):
        """Codes of ``item`` decoded so far from ``first_frame`` on: (int64 tensor [n, n_groups], the item has ended)."""
        import numpy as np
        G = self.cfg.n_groups
        buf = (C.c_int32 * max(1, int(max_frames) * G))()
        n, fin = C.c_int32(), C.c_int32()
        self.ctx.check(self.lib.rt_generate_peek(self.handle, int(item), int(first_frame), int(max_frames), buf, C.byref(n), C.byref(fin)), "rt_generate_peek")
        codes = torch.from_numpy(np.frombuffer(buf, dtype=np.int32)[: n.value * G].astype(np.int64)).reshape(n.value, G)
        return codes, bool(fin.value)

    def generate_end(self, collect: bool = False):
        """Release the generation in flight.  ``collect``: return every item's codes as ``generate`` does (needs a finished run)."""
        keep, max_frames = getattr(self, "_run_keep", (None, []))
        self._run_keep = None
        if not collect:
            self.ctx.check(self.lib.rt_generate_end(self.handle, None, None), "rt_generate_end")
            return None
        import numpy as np
        G, B, tot = self.cfg.n_groups, len(max_frames), sum(max_frames)
        codes = (C.c_int32 * max(1, tot * G))()
        nfr = (C.c_int32 * max(1, B))()
        self.ctx.check(self.lib.rt_generate_end(self.handle, codes, nfr), "rt_generate_end")
        flat = torch.from_numpy(np.frombuffer(codes, dtype=np.int32).astype(np.int64)).reshape(-1, G)
        out, off = [], 0
        for b in range(B):
            out.append(flat[off: off + nfr[b]].clone())
            off += max_frames[b]
        return out

    def generate_stats(self) -> dict:
        """Figures of the last ``generate``: decode frames launched, rows, frames kept, row hand-overs to queued items."""
        v = [C.c_int64() for _ in range(4)]
        self.ctx.check(self.lib.rt_generate_stats(self.handle, *[C.byref(x) for x in v]), "rt_generate_stats")
        run, rows, kept, swaps = (int(x.value) for x in v)
        return {"frames_run": run, "rows": rows, "frames_kept": kept, "hand_overs": swaps,
                "row_occupancy": kept / max(1, run * rows)}

    # ------------------------------------------------------------------ vocoder
    def wav_length(self, n_frames: int) -> int:
        return int(self.lib.rt_wav_length(self.handle, n_frames))

    def code2wav(self, codes: Sequence[torch.Tensor]) -> List[torch.Tensor]:
        """codes: list of int tensors [T_i, >= num_quantizers]; returns float32 GPU tensors [wav_length(T_i)]."""
        B, Q = len(codes), self.cfg.codec.num_quantizers
        T = max(int(c.shape[0]) for c in codes)
        buf = torch.zeros(B, T, Q, dtype=torch.int32)
        for b, c in enumerate(codes):
            buf[b, : c.shape[0]] = c[:, :Q].to(torch.int32)
        L = self.wav_length(T)
        dev = f"cuda:{self.ctx.device_ordinal}"
        wav = torch.empty(B, L, dtype=torch.float32, device=dev)
        nfr = (C.c_int32 * B)(*[int(c.shape[0]) for c in codes])
        lens = (C.c_int64 * B)()
        torch.cuda.synchronize()
        self.ctx.check(self.lib.rt_code2wav(self.handle, B, T, C.cast(buf.data_ptr(), C.POINTER(C.c_int32)), nfr,
                                            C.c_void_p(wav.data_ptr()), L, lens), "rt_code2wav")
        return [wav[b, : lens[b]] for b in range(B)]

    def profile(self, on: bool) -> None:
        self.ctx.check(self.lib.rt_profile_enable(self.handle, int(on)), "rt_profile_enable")

    def profile_read(self):
        n, ms, by = C.c_int64(), C.c_double(), C.c_double()
        self.ctx.check(self.lib.rt_profile_read(self.handle, C.byref(n), C.byref(ms), C.byref(by)), "rt_profile_read")
        return n.value, ms.value, by.value

    def profile_read_class(self, cls: int):
        """(launches, ms, algorithmic bytes) of the recorded decode GEMMs of one weight-stream class: 0 talker layers / codec head /
        mtp, 1 the predictor's first pass + heads, 2 predictor passes 2.. (Infinity-Cache re-stream)."""
        n, ms, by = C.c_int64(), C.c_double(), C.c_double()
        self.ctx.check(self.lib.rt_profile_read_class(self.handle, int(cls), C.byref(n), C.byref(ms), C.byref(by)), "rt_profile_read_class")
        return n.value, ms.value, by.value
