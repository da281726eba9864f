"""Shape description of a Qwen3-TTS-style model (talker + code predictor + codec decoder).

The reference never spells these numbers out: the model lives in the third-party
``qwen-tts`` package (reference call sites providers/qwen.py:100,160-165,247-258)
and no ``config.json`` is available offline.  Everything here is therefore a
parameter; the named presets carry the dimensions SURVEY.md section 8d assumes
(marked UNVERIFIED there) and ``from_hf_config`` reads a real config when one is
present in ``model_path``.
"""
from __future__ import annotations

import json
import os
from dataclasses import asdict, dataclass, field
from typing import Dict, Tuple


@dataclass
class TransformerDims:
    hidden: int
    layers: int
    heads: int
    kv_heads: int
    head_dim: int
    inter: int
    rope_theta: float = 1_000_000.0
    rms_eps: float = 1e-6

    @property
    def q_dim(self) -> int:
        return self.heads * self.head_dim

    @property
    def kv_dim(self) -> int:
        return self.kv_heads * self.head_dim

    def weight_params(self) -> int:
        """Matrix parameters of the decoder layers (what a decode step streams)."""
        per = self.hidden * (self.q_dim + 2 * self.kv_dim) + self.q_dim * self.hidden + 3 * self.hidden * self.inter
        return self.layers * per


@dataclass
class CodecDims:
    codebook_size: int = 2048
    num_quantizers: int = 16
    hidden: int = 1024
    layers: int = 8
    heads: int = 16
    head_dim: int = 64
    inter: int = 3072
    sliding_window: int = 72
    rope_theta: float = 10_000.0
    rms_eps: float = 1e-5
    layer_scale: float = 0.01
    upsampling_ratios: Tuple[int, ...] = (2, 2)
    upsample_rates: Tuple[int, ...] = (8, 5, 4, 3)
    decoder_dim: int = 1536
    chunk_frames: int = 300
    left_context_frames: int = 25
    # ---- encoder side (reference audio -> codes).  A SEANet-style causal conv encoder (filters x 2 per
    # stage, strides enc_ratios), a pre-norm transformer at twice the frame rate, a stride-2 down-sampling conv and a split
    # residual vector quantiser (1 semantic + num_quantizers - 1 acoustic codebooks) - the topology of the Mimi codec in the
    # container's transformers (models/mimi/modeling_mimi.py), which the 12.5-Hz / 2048-entry / 16-codebook numbers match.
    # UNVERIFIED against the real Qwen3-TTS tokenizer, like every other dimension here.
    enc_filters: int = 64
    enc_ratios: Tuple[int, ...] = (4, 5, 6, 8)
    enc_kernel: int = 7
    enc_res_kernel: int = 3
    enc_last_kernel: int = 3
    enc_hidden: int = 512
    enc_layers: int = 8
    enc_heads: int = 8
    enc_head_dim: int = 64
    enc_inter: int = 2048
    enc_window: int = 250
    vq_dim: int = 256
    spk_hidden: int = 512            # speaker head: statistics pooling of the conv features -> fc -> talker width

    @property
    def enc_stride(self) -> int:
        """Input samples per encoder frame (before the final stride-2 conv)."""
        t = 1
        for r in self.enc_ratios:
            t *= r
        return t

    @property
    def total_upsample(self) -> int:
        t = 1
        for r in tuple(self.upsampling_ratios) + tuple(self.upsample_rates):
            t *= r
        return t


@dataclass
class ModelConfig:
    name: str
    talker: TransformerDims
    predictor: TransformerDims
    codec: CodecDims = field(default_factory=CodecDims)
    codec_vocab: int = 3072          # talker output vocabulary: 2048 codes + control ids
    predictor_vocab: int = 2048
    text_vocab: int = 151936
    text_hidden: int = 2048
    n_groups: int = 16
    sample_rate: int = 24000
    max_positions: int = 4096        # KV rows per sequence the engine allocates (prompt + frames): sized by this package, NOT by a checkpoint
    # ``max_position_embeddings`` of a checkpoint's config.json, if it has one.  The reference uses it for ONE thing: refining
    # the segment character limit (providers/qwen.py:131-139) - and that is all it feeds here (provider._max_model_chars).  It
    # must never size the KV caches: 32768 positions x 33 slots of the 1.7B talker would be 124 GB.
    hf_max_position_embeddings: int = 0
    # text-side control ids
    tts_pad_id: int = 151671
    tts_bos_id: int = 151672
    tts_eos_id: int = 151673
    role_ids: Tuple[int, ...] = (151644, 77091, 198)   # <|im_start|> assistant \n
    # codec-side control ids (inside [codebook_size, codec_vocab))
    codec_pad_id: int = 2148
    codec_bos_id: int = 2149
    codec_eos_id: int = 2150
    codec_think_id: int = 2154
    codec_nothink_id: int = 2155
    codec_think_bos_id: int = 2156
    codec_think_eos_id: int = 2157
    language_ids: Dict[str, int] = field(default_factory=lambda: {
        "english": 2050, "chinese": 2055, "japanese": 2058, "korean": 2064})
    speaker_ids: Dict[str, int] = field(default_factory=lambda: {
        n.lower(): 2200 + i for i, n in enumerate(
            ["Chelsie", "Aidan", "Vivian", "Ryan", "Aria", "Ethan", "Luna", "Harper", "James"])})

    @property
    def frame_rate(self) -> float:
        return self.sample_rate / self.codec.total_upsample

    @property
    def has_mtp_proj(self) -> bool:
        return self.talker.hidden != self.predictor.hidden

    def to_json(self) -> str:
        """``config.json`` of a checkpoint written by this package (weights.save_checkpoint): every field, under one key."""
        return json.dumps({"rho_tts_amd": asdict(self)}, indent=1)

    @staticmethod
    def from_dict(d: dict) -> "ModelConfig":
        d = dict(d)
        talker, predictor, codec = TransformerDims(**d.pop("talker")), TransformerDims(**d.pop("predictor")), dict(d.pop("codec"))
        for k in ("upsampling_ratios", "upsample_rates", "enc_ratios"):
            if k in codec:
                codec[k] = tuple(codec[k])
        d["role_ids"] = tuple(d["role_ids"])
        return ModelConfig(talker=talker, predictor=predictor, codec=CodecDims(**codec), **d)


def qwen3_tts_1p7b() -> ModelConfig:
    return ModelConfig(
        name="Qwen3-TTS-1.7B",
        talker=TransformerDims(hidden=2048, layers=28, heads=16, kv_heads=8, head_dim=128, inter=6144),
        predictor=TransformerDims(hidden=1024, layers=5, heads=16, kv_heads=8, head_dim=128, inter=3072),
        text_hidden=2048)


def qwen3_tts_0p6b() -> ModelConfig:
    return ModelConfig(
        name="Qwen3-TTS-0.6B",
        talker=TransformerDims(hidden=1024, layers=28, heads=16, kv_heads=8, head_dim=128, inter=3072),
        predictor=TransformerDims(hidden=1024, layers=5, heads=16, kv_heads=8, head_dim=128, inter=3072),
        text_hidden=2048)


def tiny(name: str = "tiny") -> ModelConfig:
    """A few-thousand-parameter model of the same topology for parity tests
    (sizes chosen so every code path — GQA, mtp projection, multi-tile GEMMs,
    all decoder stages — is exercised while the oracle runs in seconds)."""
    return ModelConfig(
        name=name,
        talker=TransformerDims(hidden=128, layers=2, heads=4, kv_heads=2, head_dim=32, inter=256),
        predictor=TransformerDims(hidden=64, layers=2, heads=2, kv_heads=1, head_dim=32, inter=128),
        codec=CodecDims(codebook_size=64, num_quantizers=4, hidden=64, layers=2, heads=2, head_dim=32, inter=128,
                        sliding_window=8, upsampling_ratios=(2,), upsample_rates=(3, 2), decoder_dim=64,
                        chunk_frames=12, left_context_frames=3,
                        enc_filters=16, enc_ratios=(2, 3), enc_hidden=32, enc_layers=2, enc_heads=2, enc_head_dim=32, enc_inter=64,
                        enc_window=6, vq_dim=16, spk_hidden=32),
        codec_vocab=128, predictor_vocab=64, text_vocab=512, text_hidden=96, n_groups=4, max_positions=256,
        tts_pad_id=500, tts_bos_id=501, tts_eos_id=502, role_ids=(503, 504, 505),
        codec_pad_id=70, codec_bos_id=71, codec_eos_id=72, codec_think_id=73, codec_nothink_id=74,
        codec_think_bos_id=75, codec_think_eos_id=76,
        language_ids={"english": 80, "chinese": 81, "japanese": 82, "korean": 83},
        speaker_ids={"vivian": 90, "ryan": 91})


def small(name: str = "small") -> ModelConfig:
    """Mid-size topology (head_dim 128, MFMA-tile-aligned dims) used by GPU parity tests."""
    return ModelConfig(
        name=name,
        talker=TransformerDims(hidden=512, layers=3, heads=8, kv_heads=4, head_dim=128, inter=1024),
        predictor=TransformerDims(hidden=256, layers=2, heads=4, kv_heads=2, head_dim=128, inter=512),
        codec=CodecDims(codebook_size=256, num_quantizers=8, hidden=256, layers=2, heads=4, head_dim=64, inter=512,
                        sliding_window=16, upsampling_ratios=(2, 2), upsample_rates=(4, 3, 2), decoder_dim=384,
                        chunk_frames=24, left_context_frames=4,
                        enc_filters=16, enc_ratios=(2, 4, 6), enc_hidden=128, enc_layers=2, enc_heads=4, enc_head_dim=32, enc_inter=256,
                        enc_window=12, vq_dim=64, spk_hidden=64),
        codec_vocab=384, predictor_vocab=256, text_vocab=4096, text_hidden=320, n_groups=8, max_positions=1024,
        tts_pad_id=4000, tts_bos_id=4001, tts_eos_id=4002, role_ids=(4003, 4004, 4005),
        codec_pad_id=300, codec_bos_id=301, codec_eos_id=302, codec_think_id=303, codec_nothink_id=304,
        codec_think_bos_id=305, codec_think_eos_id=306,
        language_ids={"english": 310, "chinese": 311, "japanese": 312, "korean": 313},
        speaker_ids={"vivian": 320, "ryan": 321})


PRESETS = {"1.7b": qwen3_tts_1p7b, "0.6b": qwen3_tts_0p6b, "tiny": tiny, "small": small}


def resolve(model_path: str) -> ModelConfig:
    """Pick a config from a model id / path the way the reference's users name models
    (``Qwen/Qwen3-TTS-12Hz-1.7B-Base``, providers/qwen.py:57; UI catalogue ui/config.py:33-60)."""
    cfg_file = os.path.join(model_path, "config.json") if os.path.isdir(model_path) else None
    if cfg_file and os.path.exists(cfg_file):
        with open(cfg_file) as f:
            js = json.load(f)
        if "rho_tts_amd" in js:
            return ModelConfig.from_dict(js["rho_tts_amd"])
        return from_hf_config(js, name=os.path.basename(model_path.rstrip("/")))
    low = model_path.lower()
    for key in ("tiny", "small"):
        if low.endswith(key):
            return PRESETS[key]()
    if "0.6b" in low:
        return qwen3_tts_0p6b()
    return qwen3_tts_1p7b()


def _dims(d: dict, default: TransformerDims) -> TransformerDims:
    return TransformerDims(
        hidden=d.get("hidden_size", default.hidden), layers=d.get("num_hidden_layers", default.layers),
        heads=d.get("num_attention_heads", default.heads), kv_heads=d.get("num_key_value_heads", default.kv_heads),
        head_dim=d.get("head_dim", default.head_dim), inter=d.get("intermediate_size", default.inter),
        rope_theta=float(d.get("rope_theta", default.rope_theta)), rms_eps=float(d.get("rms_norm_eps", default.rms_eps)))


def from_hf_config(js: dict, name: str = "hf") -> ModelConfig:
    """Best-effort reader of a Hugging Face style ``config.json`` (talker_config /
    code_predictor_config sub-dicts as in the Qwen3-Omni sibling architecture,
    transformers/models/qwen3_omni_moe/configuration_qwen3_omni_moe.py)."""
    base = qwen3_tts_1p7b()
    t = js.get("talker_config", js)
    p = t.get("code_predictor_config", js.get("code_predictor_config", {}))
    cfg = ModelConfig(name=name, talker=_dims(t.get("text_config", t), base.talker), predictor=_dims(p, base.predictor))
    cfg.codec_vocab = t.get("vocab_size", cfg.codec_vocab)
    cfg.predictor_vocab = p.get("vocab_size", cfg.predictor_vocab)
    cfg.n_groups = t.get("num_code_groups", cfg.n_groups)
    cfg.text_vocab = t.get("text_vocab_size", cfg.text_vocab)
    cfg.text_hidden = t.get("text_hidden_size", cfg.text_hidden)
    for k in ("codec_pad_id", "codec_bos_id", "codec_eos_id", "codec_think_id", "codec_nothink_id",
              "codec_think_bos_id", "codec_think_eos_id"):
        if k in t:
            setattr(cfg, k, t[k])
    if "codec_eos_token_id" in t:
        cfg.codec_eos_id = t["codec_eos_token_id"]
    if "max_position_embeddings" in t.get("text_config", t):       # the reference refines its segment limit from it (qwen.py:131-139)
        cfg.hf_max_position_embeddings = int(t.get("text_config", t)["max_position_embeddings"])
        cfg.max_positions = min(cfg.max_positions, cfg.hf_max_position_embeddings)      # (a model trained on fewer positions than the default allocation)
    c2w = js.get("code2wav_config", js.get("speech_tokenizer_config", {}))
    if c2w:
        d = cfg.codec
        for src, dst in (("codebook_size", "codebook_size"), ("num_quantizers", "num_quantizers"), ("hidden_size", "hidden"),
                         ("num_hidden_layers", "layers"), ("num_attention_heads", "heads"), ("intermediate_size", "inter"),
                         ("sliding_window", "sliding_window"), ("decoder_dim", "decoder_dim")):
            if src in c2w:
                setattr(d, dst, c2w[src])
        if "upsample_rates" in c2w:
            d.upsample_rates = tuple(c2w["upsample_rates"])
        if "upsampling_ratios" in c2w:
            d.upsampling_ratios = tuple(c2w["upsampling_ratios"])
    return cfg
