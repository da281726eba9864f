"""On-GPU speech-to-text for the validation loop: host side of the ``rt_stt_*`` group of the C ABI.

The reference validates a generated segment by writing it to a temporary WAV (base_tts.py:821-830) and transcribing that file
with Whisper "tiny" - faster-whisper on the CPU, or transformers' Whisper (validation/stt/stt_validator.py:42-148).  Here the
waveform never leaves HBM: ``WhisperTranscriber`` is a ``transcriber`` hook of the provider (provider.BatchedPipeline) - a
callable ``(audio, sample_rate) -> text`` - whose model is the hand-written HIP encoder-decoder of ``csrc/stt.hip``.

Weights: a local Whisper checkpoint directory (``model.safetensors`` with transformers' tensor names + ``config.json`` +
``tokenizer.json``) when there is one; there is no network here, so ``openai/whisper-tiny`` cannot be fetched - seeded synthetic
weights of the same architecture serve the parity tests and benchmarks (they transcribe nothing meaningful).
"""
from __future__ import annotations

import ctypes as C
import json
import math
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _native

DT_BF16, DT_F32 = 0, 1


@dataclass
class SttConfig:
    """Whisper "tiny" (multilingual) by default: models/whisper/configuration_whisper.py of the container's transformers."""
    d_model: int = 384
    heads: int = 6
    ffn: int = 1536
    enc_layers: int = 4
    dec_layers: int = 4
    n_mels: int = 80
    n_ctx: int = 1500
    n_text_ctx: int = 448
    vocab: int = 51865
    n_fft: int = 400
    hop: int = 160
    sample_rate: int = 16000
    chunk_seconds: int = 30
    eos_id: int = 50257
    # <|startoftranscript|> <|en|> <|transcribe|> <|notimestamps|>: what stt_validator.py:137 asks of faster-whisper (language="en")
    prefix: Tuple[int, ...] = (50258, 50259, 50359, 50363)
    suppress_from: int = 50257           # special tokens (language / task / timestamp ids) are never text
    begin_suppress: Tuple[int, ...] = (220, 50257)
    suppress_tokens: Tuple[int, ...] = ()  # further ids never produced (a generation config's `suppress_tokens`)
    max_new_tokens: int = 224

    @staticmethod
    def from_hf(js: dict, generation: Optional[dict] = None, preprocessor: Optional[dict] = None) -> "SttConfig":
        """From a checkpoint directory's ``config.json`` (+ ``generation_config.json`` for the forced prefix and the suppressed
        ids, + ``preprocessor_config.json`` for the feature extractor), as transformers' ``save_pretrained`` writes them."""
        c = SttConfig()
        c.d_model = js.get("d_model", c.d_model)
        c.heads = js.get("encoder_attention_heads", c.heads)
        c.ffn = js.get("encoder_ffn_dim", c.ffn)
        c.enc_layers, c.dec_layers = js.get("encoder_layers", c.enc_layers), js.get("decoder_layers", c.dec_layers)
        c.n_mels, c.n_ctx = js.get("num_mel_bins", c.n_mels), js.get("max_source_positions", c.n_ctx)
        c.n_text_ctx, c.vocab = js.get("max_target_positions", c.n_text_ctx), js.get("vocab_size", c.vocab)
        pre = preprocessor or {}
        c.n_fft, c.hop = int(pre.get("n_fft", c.n_fft)), int(pre.get("hop_length", c.hop))
        c.sample_rate, c.n_mels = int(pre.get("sampling_rate", c.sample_rate)), int(pre.get("feature_size", c.n_mels))
        # the encoder's two convolutions halve the frames: one chunk is 2 n_ctx hops (30 s for n_ctx 1500)
        c.chunk_seconds = int(pre.get("chunk_length", max(1, round(2 * c.n_ctx * c.hop / c.sample_rate))))
        gen = dict(js)
        gen.update({k: v for k, v in (generation or {}).items() if v is not None})
        eos = gen.get("eos_token_id", c.eos_id)
        c.eos_id = int(eos[0] if isinstance(eos, (list, tuple)) else eos)
        start = int(gen.get("decoder_start_token_id", c.prefix[0]))
        # Forced prefix.  The reference transcribes with language="en" (stt_validator.py:137), so position 1 is <|en|> whatever the
        # checkpoint's default is.  Published generation configs carry `forced_decoder_ids: [[1, null], [2, 50359]]` - a None means
        # "decided at generation time", not "absent" - next to lang_to_id / task_to_id / no_timestamps_token_id: a None (or missing)
        # position is filled from those, and the ids they name win over stale forced entries.
        forced = {int(p_): (None if t is None else int(t)) for p_, t in (gen.get("forced_decoder_ids") or [])}
        lang, task = gen.get("lang_to_id") or {}, gen.get("task_to_id") or {}
        nts = gen.get("no_timestamps_token_id")
        if forced or (lang and task):
            classic = start == SttConfig.prefix[0]                  # the multilingual vocabulary the defaults describe
            en = lang.get("<|en|>", forced.get(1))
            if en is None and classic:
                en = SttConfig.prefix[1]
            tr = task.get("transcribe", forced.get(2))
            if tr is None and classic:
                tr = SttConfig.prefix[2]
            nt = nts if nts is not None else forced.get(3)
            if nt is None and classic:
                nt = SttConfig.prefix[3]
            c.prefix = (start,) + tuple(int(t) for t in (en, tr, nt) if t is not None)
        elif start != c.prefix[0]:
            c.prefix = (start,)
        if gen.get("begin_suppress_tokens"):
            c.begin_suppress = tuple(int(t) for t in gen["begin_suppress_tokens"] if 0 <= int(t) < c.vocab)[:4]
        # Whisper's vocabularies put every special id (language, task, timestamps) behind end-of-sequence
        c.suppress_from = c.eos_id if 0 < c.eos_id < c.vocab else 0
        # `suppress_tokens`: ids never produced (punctuation-only / non-speech tokens and a few specials below end-of-sequence)
        c.suppress_tokens = tuple(int(t) for t in (gen.get("suppress_tokens") or []) if 0 <= int(t) < c.vocab and int(t) != c.eos_id)
        if gen.get("max_new_tokens"):
            c.max_new_tokens = int(gen["max_new_tokens"])
        c.max_new_tokens = max(1, min(c.max_new_tokens, c.n_text_ctx - len(c.prefix)))
        return c

    @staticmethod
    def from_dir(model_dir: str) -> "SttConfig":
        def read(name):
            path = os.path.join(model_dir, name)
            if not os.path.exists(path):
                return None
            with open(path) as f:
                return json.load(f)
        js = read("config.json")
        if js is None:
            raise ValueError(f"no config.json in {model_dir!r}")
        return SttConfig.from_hf(js, read("generation_config.json"), read("preprocessor_config.json"))


def tiny_test_config() -> SttConfig:
    """A few-thousand-parameter model of the same topology for quick tests (2-s chunks)."""
    return SttConfig(d_model=64, heads=2, ffn=128, enc_layers=2, dec_layers=2, n_mels=16, n_ctx=100, n_text_ctx=32, vocab=300,
                     chunk_seconds=2, eos_id=290, prefix=(291, 292, 293), suppress_from=290, begin_suppress=(7, 290), max_new_tokens=12)


class RtSttConfig(C.Structure):
    _fields_ = [("d_model", C.c_int32), ("heads", C.c_int32), ("ffn", C.c_int32), ("enc_layers", C.c_int32), ("dec_layers", C.c_int32),
                ("n_mels", C.c_int32), ("n_ctx", C.c_int32), ("n_text_ctx", C.c_int32), ("vocab", C.c_int32),
                ("n_fft", C.c_int32), ("hop", C.c_int32), ("sample_rate", C.c_int32), ("chunk_seconds", C.c_int32),
                ("eos_id", C.c_int32), ("n_prefix", C.c_int32), ("prefix", C.c_int32 * 8), ("suppress_from", C.c_int32),
                ("n_begin_suppress", C.c_int32), ("begin_suppress", C.c_int32 * 4), ("max_new_tokens", C.c_int32), ("reserved", C.c_int32 * 4)]


def _rt_config(c: SttConfig) -> RtSttConfig:
    r = RtSttConfig()
    for k in ("d_model", "heads", "ffn", "enc_layers", "dec_layers", "n_mels", "n_ctx", "n_text_ctx", "vocab", "n_fft", "hop", "sample_rate",
              "chunk_seconds", "eos_id", "suppress_from", "max_new_tokens"):
        setattr(r, k, int(getattr(c, k)))
    r.n_prefix = len(c.prefix)
    for i, t in enumerate(c.prefix):
        r.prefix[i] = int(t)
    r.n_begin_suppress = len(c.begin_suppress)
    for i, t in enumerate(c.begin_suppress):
        r.begin_suppress[i] = int(t)
    return r


_DECLARED = False


def _declare(lib: C.CDLL) -> None:
    global _DECLARED
    if _DECLARED:
        return
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    lib.rt_stt_create.argtypes = [vp, C.POINTER(RtSttConfig), C.POINTER(vp)]
    lib.rt_stt_destroy.argtypes = [vp]
    lib.rt_stt_tensor_count.argtypes = [vp]
    lib.rt_stt_tensor_info.argtypes = [vp, i32, C.c_char_p, C.c_size_t, C.POINTER(i64), C.POINTER(i32)]
    lib.rt_stt_set_tensor.argtypes = [vp, C.c_char_p, vp, i32, i64, i64, i32]
    lib.rt_stt_finalize.argtypes = [vp]
    lib.rt_stt_set_suppress.argtypes = [vp, C.POINTER(i32), i32]
    lib.rt_stt_transcribe.argtypes = [vp, vp, i64, i32, C.POINTER(i32), i32, C.POINTER(i32), vp]
    lib.rt_stt_log_mel.argtypes = [vp, vp, i64, i32, vp]
    lib.rt_stt_encode.argtypes = [vp, vp, i64, i32, vp]
    _DECLARED = True


# ---------------------------------------------------------------------------------------------- front-end constants
def hertz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    mels = 3.0 * f / 200.0
    log = f >= 1000.0
    return np.where(log, 15.0 + np.log(np.maximum(f, 1e-30) / 1000.0) * (27.0 / np.log(6.4)), mels)


def mel_to_hertz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    f = 200.0 * m / 3.0
    log = m >= 15.0
    return np.where(log, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), f)


def mel_filters(n_fft: int, n_mels: int, sample_rate: int, f_max: float = 8000.0) -> np.ndarray:
    """[n_fft // 2 + 1][n_mels] float32 slaney-normalised triangular filters: what WhisperFeatureExtractor builds with
    transformers.audio_utils.mel_filter_bank(..., norm="slaney", mel_scale="slaney") (feature_extraction_whisper.py:95-103)."""
    n_bins = n_fft // 2 + 1
    mel_pts = np.linspace(hertz_to_mel_slaney(0.0), hertz_to_mel_slaney(f_max), n_mels + 2)
    hz = mel_to_hertz_slaney(mel_pts)
    fft_freqs = np.linspace(0, sample_rate // 2, n_bins)
    diff = np.diff(hz)
    slopes = hz[None, :] - fft_freqs[:, None]
    down = -slopes[:, :-2] / diff[:-1]
    up = slopes[:, 2:] / diff[1:]
    fb = np.maximum(0.0, np.minimum(down, up))
    fb *= (2.0 / (hz[2: n_mels + 2] - hz[:n_mels]))[None, :]
    return fb.astype(np.float32)


def hann_window(n_fft: int) -> np.ndarray:
    """torch.hann_window(n_fft) (periodic) in float32."""
    return torch.hann_window(n_fft, periodic=True, dtype=torch.float32).numpy()


# ---------------------------------------------------------------------------------------------- weights
def tensor_specs(c: SttConfig) -> List[Tuple[str, Tuple[int, ...], str, float]]:
    """(name, shape, kind, scale) with transformers' Whisper tensor names (models/whisper/modeling_whisper.py)."""
    D, F = c.d_model, c.ffn
    s: List[Tuple[str, Tuple[int, ...], str, float]] = [
        ("model.encoder.conv1.weight", (D, c.n_mels, 3), "fan", 1.0), ("model.encoder.conv1.bias", (D,), "bias", 0.01),
        ("model.encoder.conv2.weight", (D, D, 3), "fan", 1.0), ("model.encoder.conv2.bias", (D,), "bias", 0.01),
        ("model.encoder.embed_positions.weight", (c.n_ctx, D), "mat", 0.02)]

    def layer(p: str, cross: bool):
        out = []
        blocks = [("self_attn", "self_attn_layer_norm")] + ([("encoder_attn", "encoder_attn_layer_norm")] if cross else [])
        for att, ln in blocks:
            out += [(f"{p}.{att}.k_proj.weight", (D, D), "mat", 0.05), (f"{p}.{att}.v_proj.weight", (D, D), "mat", 0.05),
                    (f"{p}.{att}.v_proj.bias", (D,), "bias", 0.01), (f"{p}.{att}.q_proj.weight", (D, D), "mat", 0.05),
                    (f"{p}.{att}.q_proj.bias", (D,), "bias", 0.01), (f"{p}.{att}.out_proj.weight", (D, D), "mat", 0.05),
                    (f"{p}.{att}.out_proj.bias", (D,), "bias", 0.01), (f"{p}.{ln}.weight", (D,), "norm", 0.1), (f"{p}.{ln}.bias", (D,), "bias", 0.02)]
        out += [(f"{p}.fc1.weight", (F, D), "mat", 0.05), (f"{p}.fc1.bias", (F,), "bias", 0.01), (f"{p}.fc2.weight", (D, F), "mat", 0.03),
                (f"{p}.fc2.bias", (D,), "bias", 0.01), (f"{p}.final_layer_norm.weight", (D,), "norm", 0.1), (f"{p}.final_layer_norm.bias", (D,), "bias", 0.02)]
        return out
    for i in range(c.enc_layers):
        s += layer(f"model.encoder.layers.{i}", False)
    s += [("model.encoder.layer_norm.weight", (D,), "norm", 0.1), ("model.encoder.layer_norm.bias", (D,), "bias", 0.02),
          ("model.decoder.embed_tokens.weight", (c.vocab, D), "mat", 0.05), ("model.decoder.embed_positions.weight", (c.n_text_ctx, D), "mat", 0.02)]
    for i in range(c.dec_layers):
        s += layer(f"model.decoder.layers.{i}", True)
    s += [("model.decoder.layer_norm.weight", (D,), "norm", 0.1), ("model.decoder.layer_norm.bias", (D,), "bias", 0.02)]
    return s


def synthetic_state(c: SttConfig, seed: int = 789, device="cpu") -> Dict[str, torch.Tensor]:
    """Seeded bf16-valued weights (the counter-hash generator of weights.py: the same bytes on CPU and GPU)."""
    from .weights import synth_tensor
    return {sp[0]: synth_tensor(sp, None, seed, device) for sp in tensor_specs(c)}


def load_checkpoint(c: SttConfig, model_dir: str, device="cpu") -> Dict[str, torch.Tensor]:
    """``*.safetensors`` of a local Whisper checkpoint (transformers' names); nothing is executed from the files."""
    from safetensors import safe_open
    want = {sp[0]: sp[1] for sp in tensor_specs(c)}
    state: Dict[str, torch.Tensor] = {}
    files = [os.path.join(model_dir, f) for f in sorted(os.listdir(model_dir)) if f.endswith(".safetensors")]
    if not files:
        raise FileNotFoundError(f"no .safetensors files in {model_dir}")
    for path in files:
        with safe_open(path, framework="pt", device=str(device)) as sf:
            for k in sf.keys():
                name = k if k in want else ("model." + k if "model." + k in want else None)
                if name is None:
                    continue
                t = sf.get_tensor(k)
                if tuple(t.shape) != tuple(want[name]):
                    raise ValueError(f"{k}: checkpoint shape {tuple(t.shape)} != configured {want[name]}")
                state[name] = t
    missing = sorted(set(want) - set(state))
    if missing:
        raise ValueError(f"Whisper checkpoint is missing {len(missing)} of {len(want)} tensors, e.g. {missing[:4]}")
    return state


def to_native(state: Dict[str, torch.Tensor], c: SttConfig) -> Dict[str, torch.Tensor]:
    """transformers' names / layouts -> the library's tensors (bf16 matrices [N][K], float32 vectors):
    Conv1d k3 [Co][Ci][3] -> [Co][tap * Ci + ci];  the stride-2 conv over rows [x[2t], x[2t+1]] -> [Co][0 | W0 | W1 | W2];
    q / k / v projections concatenated ([3D][D]; k_proj has no bias: zeros), cross-attention k / v likewise ([2D][D])."""
    bf, D = torch.bfloat16, c.d_model
    out: Dict[str, torch.Tensor] = {}

    def f32(t):
        return t.detach().to(torch.float32).contiguous()

    def mat(t):
        return t.detach().to(bf).contiguous()
    w1 = state["model.encoder.conv1.weight"]
    out["enc.conv1"] = mat(w1.permute(0, 2, 1).reshape(D, -1))
    out["enc.conv1_b"] = f32(state["model.encoder.conv1.bias"])
    w2 = state["model.encoder.conv2.weight"].to(torch.float32)                  # [Co][Ci][3]
    z = torch.zeros(D, D, dtype=torch.float32, device=w2.device)
    out["enc.conv2"] = mat(torch.cat([z, w2[:, :, 0], w2[:, :, 1], w2[:, :, 2]], dim=1))
    out["enc.conv2_b"] = f32(state["model.encoder.conv2.bias"])
    out["enc.pos"] = f32(state["model.encoder.embed_positions.weight"]).reshape(-1)

    def layer(src: str, dst: str, cross: bool):
        a = f"{src}.self_attn"
        zb = torch.zeros(D, dtype=torch.float32, device=state[f"{a}.q_proj.bias"].device)
        out[f"{dst}.ln1_w"], out[f"{dst}.ln1_b"] = f32(state[f"{src}.self_attn_layer_norm.weight"]), f32(state[f"{src}.self_attn_layer_norm.bias"])
        out[f"{dst}.wqkv"] = mat(torch.cat([state[f"{a}.q_proj.weight"], state[f"{a}.k_proj.weight"], state[f"{a}.v_proj.weight"]], 0))
        out[f"{dst}.bqkv"] = torch.cat([f32(state[f"{a}.q_proj.bias"]), zb, f32(state[f"{a}.v_proj.bias"])])
        out[f"{dst}.wo"], out[f"{dst}.bo"] = mat(state[f"{a}.out_proj.weight"]), f32(state[f"{a}.out_proj.bias"])
        if cross:
            e = f"{src}.encoder_attn"
            out[f"{dst}.lnc_w"], out[f"{dst}.lnc_b"] = f32(state[f"{src}.encoder_attn_layer_norm.weight"]), f32(state[f"{src}.encoder_attn_layer_norm.bias"])
            out[f"{dst}.cwq"], out[f"{dst}.cbq"] = mat(state[f"{e}.q_proj.weight"]), f32(state[f"{e}.q_proj.bias"])
            out[f"{dst}.cwkv"] = mat(torch.cat([state[f"{e}.k_proj.weight"], state[f"{e}.v_proj.weight"]], 0))
            out[f"{dst}.cbkv"] = torch.cat([zb, f32(state[f"{e}.v_proj.bias"])])
            out[f"{dst}.cwo"], out[f"{dst}.cbo"] = mat(state[f"{e}.out_proj.weight"]), f32(state[f"{e}.out_proj.bias"])
        out[f"{dst}.ln2_w"], out[f"{dst}.ln2_b"] = f32(state[f"{src}.final_layer_norm.weight"]), f32(state[f"{src}.final_layer_norm.bias"])
        out[f"{dst}.fc1"], out[f"{dst}.fc1_b"] = mat(state[f"{src}.fc1.weight"]), f32(state[f"{src}.fc1.bias"])
        out[f"{dst}.fc2"], out[f"{dst}.fc2_b"] = mat(state[f"{src}.fc2.weight"]), f32(state[f"{src}.fc2.bias"])
    for i in range(c.enc_layers):
        layer(f"model.encoder.layers.{i}", f"enc.l{i}", False)
    out["enc.ln_w"], out["enc.ln_b"] = f32(state["model.encoder.layer_norm.weight"]), f32(state["model.encoder.layer_norm.bias"])
    out["dec.tok"] = mat(state["model.decoder.embed_tokens.weight"])
    out["dec.pos"] = f32(state["model.decoder.embed_positions.weight"]).reshape(-1)
    for i in range(c.dec_layers):
        layer(f"model.decoder.layers.{i}", f"dec.l{i}", True)
    out["dec.ln_w"], out["dec.ln_b"] = f32(state["model.decoder.layer_norm.weight"]), f32(state["model.decoder.layer_norm.bias"])
    out["fe.window"] = torch.from_numpy(hann_window(c.n_fft))
    out["fe.melT"] = torch.from_numpy(mel_filters(c.n_fft, c.n_mels, c.sample_rate)).reshape(-1)
    return out


class NativeSTT:
    """One ``rt_stt``: the speech-to-text model in HBM, on the context (GPU, stream) of the TTS engine it validates."""

    def __init__(self, ctx: "_native.Context", cfg: SttConfig, state: Dict[str, torch.Tensor]):
        self.ctx, self.cfg, self.lib = ctx, cfg, ctx.lib
        _declare(self.lib)
        self.rt_cfg = _rt_config(cfg)
        h = C.c_void_p()
        ctx.check(self.lib.rt_stt_create(ctx.handle, C.byref(self.rt_cfg), C.byref(h)), "rt_stt_create")
        self.handle = h
        native = to_native(state, cfg)
        n = self.lib.rt_stt_tensor_count(self.handle)
        for i in range(n):
            name = C.create_string_buffer(128)
            shp = (C.c_int64 * 2)()
            kind = C.c_int32()
            ctx.check(self.lib.rt_stt_tensor_info(self.handle, i, name, 128, shp, C.byref(kind)), "rt_stt_tensor_info")
            t = native.pop(name.value.decode()).contiguous()
            rows, cols = (t.shape[0], t.shape[1]) if t.dim() == 2 else (t.numel(), 1)
            if t.is_cuda:
                torch.cuda.current_stream(t.device).synchronize()
            ctx.check(self.lib.rt_stt_set_tensor(self.handle, name.value, C.c_void_p(t.data_ptr()), DT_BF16 if t.dtype == torch.bfloat16 else DT_F32,
                                                 rows, cols, 1 if t.is_cuda else 0), f"rt_stt_set_tensor({name.value.decode()})")
        if native:
            raise ValueError(f"tensors not consumed by the library: {sorted(native)[:4]}")
        if cfg.suppress_tokens:
            ids = (C.c_int32 * len(cfg.suppress_tokens))(*[int(t) for t in cfg.suppress_tokens])
            ctx.check(self.lib.rt_stt_set_suppress(self.handle, ids, len(cfg.suppress_tokens)), "rt_stt_set_suppress")
        ctx.check(self.lib.rt_stt_finalize(self.handle), "rt_stt_finalize")

    def close(self) -> None:
        if getattr(self, "handle", None):
            self.lib.rt_stt_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def _pcm(self, audio) -> torch.Tensor:
        x = audio if isinstance(audio, torch.Tensor) else torch.as_tensor(np.asarray(audio, dtype=np.float32))
        x = x.detach().reshape(-1).to(device=f"cuda:{self.ctx.device_ordinal}", dtype=torch.float32).contiguous()
        torch.cuda.current_stream(x.device).synchronize()
        return x

    def transcribe_ids(self, audio, sample_rate: int, max_tokens: Optional[int] = None, first_logits: bool = False):
        x = self._pcm(audio)
        # audio longer than one chunk is transcribed window by window (rt_stt_transcribe): room for every window's ids
        windows = max(1, -(-int(x.numel()) // (int(self.cfg.chunk_seconds) * int(sample_rate))))
        cap = int(max_tokens or windows * self.cfg.max_new_tokens)
        toks = (C.c_int32 * cap)()
        n = C.c_int32()
        lg = torch.empty(self.cfg.vocab, dtype=torch.float32, device=x.device) if first_logits else None
        self.ctx.check(self.lib.rt_stt_transcribe(self.handle, C.c_void_p(x.data_ptr() if x.numel() else 0), x.numel(), int(sample_rate), toks, cap,
                                                  C.byref(n), C.c_void_p(lg.data_ptr()) if lg is not None else None), "rt_stt_transcribe")
        ids = [int(toks[i]) for i in range(n.value)]
        return (ids, lg) if first_logits else ids

    def log_mel(self, audio, sample_rate: int) -> torch.Tensor:
        """[n_mels][frames] float32 (the layout of WhisperFeatureExtractor's ``input_features``)."""
        x = self._pcm(audio)
        out = torch.empty(2 * self.cfg.n_ctx, self.cfg.n_mels, dtype=torch.float32, device=x.device)
        self.ctx.check(self.lib.rt_stt_log_mel(self.handle, C.c_void_p(x.data_ptr() if x.numel() else 0), x.numel(), int(sample_rate),
                                               C.c_void_p(out.data_ptr())), "rt_stt_log_mel")
        return out.t().contiguous()

    def encode(self, audio, sample_rate: int) -> torch.Tensor:
        x = self._pcm(audio)
        out = torch.empty(self.cfg.n_ctx, self.cfg.d_model, dtype=torch.float32, device=x.device)
        self.ctx.check(self.lib.rt_stt_encode(self.handle, C.c_void_p(x.data_ptr() if x.numel() else 0), x.numel(), int(sample_rate),
                                              C.c_void_p(out.data_ptr())), "rt_stt_encode")
        return out


class WhisperTranscriber:
    """``transcriber`` hook of the provider: ``(audio tensor, sample_rate) -> text`` (None = transcription failed), the
    tensor-level stand-in for ``transcribe_audio(path)`` (stt_validator.py:116-148).  ``model_dir``: a local Whisper checkpoint
    (safetensors + config.json + tokenizer.json); without one the model runs on seeded synthetic weights - only when that is asked
    for explicitly - and the "text" is the generated ids written out, which is all random weights can mean."""

    def __init__(self, ctx: "_native.Context", model_dir: Optional[str] = None, synthetic: bool = False, cfg: Optional[SttConfig] = None, seed: int = 789):
        self.tokenizer = None
        dev = f"cuda:{ctx.device_ordinal}"
        if model_dir and os.path.isdir(model_dir) and any(f.endswith(".safetensors") for f in os.listdir(model_dir)):
            if cfg is None:
                cfg = SttConfig.from_dir(model_dir) if os.path.exists(os.path.join(model_dir, "config.json")) else SttConfig()
            state = load_checkpoint(cfg, model_dir, device=dev)
            tok_path = os.path.join(model_dir, "tokenizer.json")
            if os.path.exists(tok_path):
                from tokenizers import Tokenizer
                self.tokenizer = Tokenizer.from_file(tok_path)
        elif synthetic:
            cfg = cfg or SttConfig()
            state = synthetic_state(cfg, seed, device=dev)
        else:
            raise ValueError(f"no local Whisper checkpoint at {model_dir!r} (this build cannot download one); pass synthetic=True for seeded "
                             "random weights of the architecture (parity tests, benchmarks)")
        self.cfg = cfg
        self.model = NativeSTT(ctx, cfg, state)

    def ids(self, audio, sample_rate: int) -> List[int]:
        return self.model.transcribe_ids(audio, sample_rate)

    def __call__(self, audio, sample_rate: int) -> Optional[str]:
        ids = self.ids(audio, sample_rate)
        if self.tokenizer is not None:
            return self.tokenizer.decode(ids, skip_special_tokens=True).strip()
        return " ".join(f"<{i}>" for i in ids)

    def close(self) -> None:
        self.model.close()
