"""The real-checkpoint path (reference: providers/qwen.py:96-197 load, :131-139 max_position_embeddings), exercised offline:
a tiny synthetic state is written as safetensors shards + config.json and read back through the same functions a real
checkpoint directory goes through.  CPU only; tests/test_provider_gpu.py drives the same directory through the provider."""
import json
import os

import pytest
import torch

from rho_tts_amd import config, weights


def test_save_and_load_round_trip(tmp_path):
    cfg = config.tiny()
    state = weights.synthetic_state(cfg, 789)
    d = str(tmp_path / "tiny-ckpt")
    paths = weights.save_checkpoint(cfg, state, d, shard_bytes=200_000)
    assert len(paths) > 1 and all(os.path.exists(p) for p in paths)               # sharded
    cfg2 = config.resolve(d)
    assert cfg2 == cfg                                                             # config.json carries every field
    back = weights.load_safetensors(cfg2, d)
    assert set(back) == set(state)
    for k in state:
        assert back[k].dtype == torch.bfloat16 and torch.equal(back[k], state[k]), k


def test_missing_and_misshapen_tensors_are_reported(tmp_path):
    from safetensors.torch import save_file
    cfg = config.tiny()
    state = weights.synthetic_state(cfg, 789)
    d = str(tmp_path / "broken")
    os.makedirs(d)
    part = {k: v for k, v in state.items() if not k.startswith("codec.decoder.1")}
    part["some.unknown.tensor"] = torch.zeros(3)
    save_file(part, os.path.join(d, "model.safetensors"))
    with pytest.raises(ValueError, match="missing .* tensors.*matched no known name"):
        weights.load_safetensors(cfg, d)
    bad = dict(state)
    bad["talker.norm.weight"] = torch.zeros(7, dtype=torch.bfloat16)
    save_file(bad, os.path.join(d, "model.safetensors"))
    with pytest.raises(ValueError, match="talker.norm.weight: checkpoint shape"):
        weights.load_safetensors(cfg, d)
    os.makedirs(str(tmp_path / "empty"))
    with pytest.raises(FileNotFoundError):
        weights.load_safetensors(cfg, str(tmp_path / "empty"))


def test_hf_style_names_and_subfolder(tmp_path):
    """Sibling-architecture naming (transformers' Qwen3-Omni talker / code predictor / code2wav) + the codec in a sub-folder."""
    from safetensors.torch import save_file
    cfg = config.tiny()
    state = weights.synthetic_state(cfg, 789)

    def hf(k):
        if k.startswith("talker.layers.") or k == "talker.norm.weight":
            return k.replace("talker.", "talker.model.", 1)
        if k.startswith("talker.codec_embedding.") or k.startswith("talker.text_embedding."):
            return k.replace("talker.", "talker.model.", 1)
        if k.startswith("talker.text_projection.fc"):
            return k.replace("fc", "linear_fc")
        if k.startswith("predictor.layers.") or k == "predictor.norm.weight" or k.startswith("predictor.codec_embedding."):
            return k.replace("predictor.", "talker.code_predictor.model.", 1)
        if k.startswith("predictor."):
            return k.replace("predictor.", "talker.code_predictor.", 1)
        return k
    d = str(tmp_path / "hf")
    os.makedirs(os.path.join(d, "speech_tokenizer"))
    save_file({hf(k): v for k, v in state.items() if not k.startswith("codec.")}, os.path.join(d, "model.safetensors"))
    save_file({"decoder." + k[len("codec."):]: v for k, v in state.items() if k.startswith("codec.")},
              os.path.join(d, "speech_tokenizer", "model.safetensors"))
    for k in state:
        assert weights.remap_name(weights.remap_name(hf(k))) in state or k.startswith("codec.")
    back = weights.load_safetensors(cfg, d)
    assert set(back) == set(state) and all(torch.equal(back[k], state[k]) for k in state)


def test_from_hf_config_reads_a_handwritten_dict():
    js = {"talker_config": {"vocab_size": 3100, "num_code_groups": 12, "text_vocab_size": 150000, "text_hidden_size": 1536,
                            "codec_eos_token_id": 2160, "codec_pad_id": 2161,
                            "text_config": {"hidden_size": 1536, "num_hidden_layers": 20, "num_attention_heads": 12, "num_key_value_heads": 4,
                                            "head_dim": 128, "intermediate_size": 4096, "rope_theta": 500000, "rms_norm_eps": 1e-5,
                                            "max_position_embeddings": 8192},
                            "code_predictor_config": {"hidden_size": 768, "num_hidden_layers": 4, "num_attention_heads": 8,
                                                      "num_key_value_heads": 8, "head_dim": 96, "intermediate_size": 2048, "vocab_size": 1024}},
          "code2wav_config": {"codebook_size": 1024, "num_quantizers": 12, "hidden_size": 512, "num_hidden_layers": 6, "upsample_rates": [8, 5, 4, 2]}}
    c = config.from_hf_config(js, name="hand")
    assert (c.talker.hidden, c.talker.layers, c.talker.heads, c.talker.kv_heads, c.talker.inter) == (1536, 20, 12, 4, 4096)
    assert c.talker.rope_theta == 500000.0 and c.talker.rms_eps == 1e-5
    assert (c.predictor.hidden, c.predictor.layers, c.predictor.head_dim, c.predictor_vocab) == (768, 4, 96, 1024)
    assert (c.codec_vocab, c.n_groups, c.text_vocab, c.text_hidden, c.codec_eos_id, c.codec_pad_id) == (3100, 12, 150000, 1536, 2160, 2161)
    # max_position_embeddings feeds the segment character limit only; the KV allocation keeps this package's own size
    assert c.hf_max_position_embeddings == 8192 and c.max_positions == config.qwen3_tts_1p7b().max_positions and c.has_mtp_proj
    def with_pos(n):
        t = dict(js["talker_config"])
        t["text_config"] = dict(t["text_config"], max_position_embeddings=n)
        return dict(js, talker_config=t)
    assert config.from_hf_config(with_pos(32768)).max_positions == config.qwen3_tts_1p7b().max_positions   # 32768 x 33 slots would be 124 GB of KV
    assert config.from_hf_config(with_pos(32768)).hf_max_position_embeddings == 32768
    assert config.from_hf_config(with_pos(1024)).max_positions == 1024
    assert (c.codec.codebook_size, c.codec.num_quantizers, c.codec.hidden, c.codec.layers, c.codec.upsample_rates) == (1024, 12, 512, 6, (8, 5, 4, 2))
    # a bare dict (no sub-configs) falls back to the preset dimensions
    assert config.from_hf_config({}, name="x").talker.hidden == config.qwen3_tts_1p7b().talker.hidden


def test_engine_refuses_a_hub_id_without_the_synthetic_opt_in(monkeypatch):
    """No checkpoint on disk and no opt-in: a ValueError (configuration error, never retried by the pipeline) BEFORE any GPU
    is touched - never seeded random weights posing as the named model."""
    from rho_tts_amd import engine
    monkeypatch.delenv(engine.SYNTHETIC_ENV, raising=False)
    with pytest.raises(ValueError, match="no local checkpoint"):
        engine.Engine("Qwen/Qwen3-TTS-12Hz-1.7B-Base")
    monkeypatch.setenv(engine.SYNTHETIC_ENV, "0")
    with pytest.raises(ValueError, match="no local checkpoint"):
        engine.Engine("Qwen/Qwen3-TTS-12Hz-0.6B-Base")


def test_tokenizer_json_is_used_when_the_checkpoint_has_one(tmp_path):
    from tokenizers import Tokenizer
    from tokenizers.models import WordLevel
    from tokenizers.pre_tokenizers import Whitespace
    from rho_tts_amd.tokenizer import FileTokenizer, HashTokenizer, load_tokenizer
    tk = Tokenizer(WordLevel({"[UNK]": 0, "hello": 5, "world": 9, ".": 11}, unk_token="[UNK]"))
    tk.pre_tokenizer = Whitespace()
    tk.save(str(tmp_path / "tokenizer.json"))
    t = load_tokenizer(str(tmp_path), 512)
    assert isinstance(t, FileTokenizer) and t.encode("hello world.") == [5, 9, 11] and t.encode("hello there") == [5, 0]
    assert isinstance(load_tokenizer("Qwen/none", 512), HashTokenizer)


def test_checkpoint_without_audio_encoder_loads(tmp_path):
    """CustomVoice checkpoints never use the conditioning front-end and ship no encoder: the load must not demand enc.* tensors
     - the model is then built without an encoder (enc_filters = 0, rt_voice_encode -> RT_ERR_UNSUPPORTED)."""
    from safetensors.torch import save_file
    from rho_tts_amd._native_model import rt_config, to_native
    cfg = config.tiny()
    state = weights.synthetic_state(cfg, 789)
    d = str(tmp_path / "no-enc")
    os.makedirs(d)
    save_file({k: v for k, v in state.items() if not k.startswith("enc.")}, os.path.join(d, "model.safetensors"))
    cfg2 = config.tiny()
    back = weights.load_safetensors(cfg2, d)
    assert cfg2.codec.enc_filters == 0 and not any(k.startswith("enc.") for k in back)
    assert set(back) == {sp[0] for sp in weights.tensor_specs(cfg2)}
    native = to_native(back, cfg2)
    assert not any(k.startswith("enc.") or k.startswith("etf.") for k in native)
    assert rt_config(cfg2, 2, 64, 16).enc.filters == 0
    # a checkpoint with SOME encoder tensors is incomplete, not encoder-less
    part = {k: v for k, v in state.items() if k != "enc.downsample.weight"}
    save_file(part, os.path.join(d, "model.safetensors"))
    with pytest.raises(ValueError, match="missing 1 of"):
        weights.load_safetensors(config.tiny(), d)


def test_float32_codebooks_stay_float32_and_encoder_names_remap(tmp_path):
    from safetensors.torch import save_file
    cfg = config.tiny()
    state = weights.synthetic_state(cfg, 789)
    g = torch.Generator().manual_seed(1)
    cb = torch.randn(state["enc.vq.codebook.0"].shape, generator=g) * 0.37          # not bf16-representable
    assert not torch.equal(cb, cb.to(torch.bfloat16).float())

    def hf(k):
        if k.startswith("enc.transformer."):
            return k.replace("enc.transformer.", "speech_tokenizer.encoder_transformer.", 1)
        if k == "enc.downsample.weight":
            return "speech_tokenizer.downsample.conv.weight"
        if k.startswith("enc.vq.semantic.input_proj."):
            return k.replace("enc.vq.semantic.", "speech_tokenizer.quantizer.semantic_residual_vector_quantizer.", 1)
        if k.startswith("enc.vq.acoustic.input_proj."):
            return k.replace("enc.vq.acoustic.", "speech_tokenizer.quantizer.acoustic_residual_vector_quantizer.", 1)
        if k.startswith("enc.spk."):
            return k.replace("enc.spk.", "speech_tokenizer.speaker_encoder.", 1)
        if k.startswith("enc."):
            return "speech_tokenizer." + k
        return k
    d = str(tmp_path / "f32")
    os.makedirs(d)
    out = {hf(k): v for k, v in state.items()}
    out[hf("enc.vq.codebook.0")] = cb
    out["talker.norm.weight"] = state["talker.norm.weight"].float()                # other float32 tensors still go to bf16
    save_file(out, os.path.join(d, "model.safetensors"))
    back = weights.load_safetensors(cfg, d)
    assert set(back) == set(state)
    assert back["enc.vq.codebook.0"].dtype == torch.float32 and torch.equal(back["enc.vq.codebook.0"], cb)
    assert back["talker.norm.weight"].dtype == torch.bfloat16
    from rho_tts_amd._native_model import to_native
    assert torch.equal(to_native(back, cfg)["enc.cbT0"].reshape(cb.shape[1], cb.shape[0]).t(), cb)


def test_resolve_on_an_hf_config_json_keeps_the_kv_allocation(tmp_path):
    """a real checkpoint's config.json says max_position_embeddings = 32768; that number may refine the segment
    character limit (qwen.py:131-139) but must not size the KV caches (33 slots x 32768 rows of the 1.7B talker = 124 GB)."""
    from rho_tts_amd._native_model import rt_config
    d = tmp_path / "Qwen3-TTS-12Hz-1.7B-Base"
    d.mkdir()
    (d / "config.json").write_text(json.dumps({"talker_config": {"text_config": {"max_position_embeddings": 32768, "hidden_size": 2048}}}))
    cfg = config.resolve(str(d))
    base = config.qwen3_tts_1p7b()
    assert cfg.hf_max_position_embeddings == 32768 and cfg.max_positions == base.max_positions == 4096
    rc = rt_config(cfg, 32, cfg.max_positions, 325)
    assert rc.max_positions == 4096 and rc.enc.max_ref_frames == 2048
    t = cfg.talker
    kv_bytes = 33 * cfg.max_positions * t.layers * 2 * t.kv_heads * t.head_dim * 2
    assert kv_bytes < 16e9
