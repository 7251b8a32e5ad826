"""Checkpoint compatibility on the host side (no GPU): the product modules expose the reference's state_dict keys
(LVM/model.py:178-192 + HF Phi3 names; diffusers AutoencoderKL names) and load local checkpoint directories the way
LVM.from_pretrained / LVMPipeline.from_pretrained do (LVM/model.py:195-211, LVM/pipeline.py:74-95) — safetensors only,
nothing is unpickled."""
import importlib
import json
import os

import torch
from safetensors.torch import save_file

from oracle import restate as R
from oracle import vae_ref as VR

M = importlib.import_module("video-gpt_amd.model")
V = importlib.import_module("video-gpt_amd.vae")


def _write_lvm_dir(tmp, cfg, params):
    os.makedirs(tmp, exist_ok=True)
    with open(os.path.join(tmp, "config.json"), "w") as f:
        json.dump(dict(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                       num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                       num_key_value_heads=cfg.num_key_value_heads, hidden_act=cfg.hidden_act, rms_norm_eps=cfg.rms_norm_eps,
                       rope_theta=cfg.rope_theta, pad_token_id=cfg.pad_token_id, model_type="phi3"), f)
    save_file({k: v.contiguous() for k, v in params.items()}, os.path.join(tmp, "model.safetensors"))


def test_lvm_from_pretrained_local_directory(tmp_path):
    cfg = R.Phi3Cfg(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2,
                    num_key_value_heads=1, vocab_size=32, pos_embed_max_size=192)
    params = R.make_params(cfg, seed=5)
    _write_lvm_dir(str(tmp_path / "ckpt"), cfg, params)
    model = M.LVM.from_pretrained(str(tmp_path / "ckpt"))
    sd = model.state_dict()
    assert set(sd) == set(params)
    for k in params:
        assert torch.equal(sd[k], params[k]), k
    assert model.llm.config.num_key_value_heads == 1 and model.pos_embed.shape == (1, 192 * 192, 64)
    # the position table the constructor builds is the reference's (LVM/model.py:185-186)
    fresh = M.LVM(model.llm.config)
    assert torch.equal(fresh.pos_embed, R.make_pos_embed(cfg))


def test_state_dict_keys_match_reference_names():
    cfg = M.Phi3Config(vocab_size=16, hidden_size=64, intermediate_size=64, num_hidden_layers=1, num_attention_heads=1)
    keys = set(M.LVMTraining(cfg, pos_embed_max_size=8).state_dict())
    for must in ("x_embedder.proj.weight", "input_x_embedder.proj.bias", "time_token.mlp.0.weight", "t_embedder.mlp.2.bias",
                 "pos_embed", "final_layer.linear.weight", "final_layer.adaLN_modulation.1.weight",
                 "llm.embed_tokens.weight", "llm.layers.0.self_attn.qkv_proj.weight", "llm.layers.0.self_attn.o_proj.weight",
                 "llm.layers.0.mlp.gate_up_proj.weight", "llm.layers.0.mlp.down_proj.weight",
                 "llm.layers.0.input_layernorm.weight", "llm.layers.0.post_attention_layernorm.weight", "llm.norm.weight"):
        assert must in keys, must
    m = M.LVM(cfg, pos_embed_max_size=8)
    m.init_input_final_layer()
    assert "input_final_layer.weight" in m.state_dict()


def test_deepcopy_and_pickle_leave_the_sampler_engines_behind():
    """LVMScheduler keeps the last clips' engines on the model (GBs of buffers, a hipGraph, a reference back to the model):
    copy.deepcopy(model) -- how the reference makes its EMA (LVM/train/train_x1_stage1_noiseinput.py:229) -- and pickling must
    not try to copy them."""
    import copy
    import pickle
    cfg = M.Phi3Config(vocab_size=16, hidden_size=64, intermediate_size=64, num_hidden_layers=1, num_attention_heads=1)
    m = M.LVM(cfg, pos_embed_max_size=8)

    class Unpicklable:
        def __reduce__(self):
            raise RuntimeError("an engine must not be copied")
    m.__dict__["_vgpt_engine_cache"] = {"key": Unpicklable()}
    ema = copy.deepcopy(m)
    assert "_vgpt_engine_cache" not in ema.__dict__ and "_vgpt_engine_cache" in m.__dict__
    assert all(torch.equal(a, b) for a, b in zip(ema.state_dict().values(), m.state_dict().values()))
    again = pickle.loads(pickle.dumps(m))
    assert "_vgpt_engine_cache" not in again.__dict__
    m.release_engines()
    assert "_vgpt_engine_cache" not in m.__dict__


def test_vae_from_pretrained_local_directory(tmp_path):
    cfg = VR.TINY_VAE8
    p = VR.make_vae_params(cfg, seed=7)
    d = tmp_path / "vae"
    os.makedirs(d)
    with open(d / "config.json", "w") as f:
        json.dump(dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=list(cfg.block_out_channels),
                       layers_per_block=cfg.layers_per_block, norm_num_groups=cfg.norm_num_groups, scaling_factor=0.13025,
                       _class_name="AutoencoderKL", sample_size=512), f)
    save_file(p, str(d / "diffusion_pytorch_model.safetensors"))
    vae = V.AutoencoderKL.from_pretrained(str(d))
    sd = vae.state_dict()
    assert set(sd) == set(p) and all(torch.equal(sd[k], p[k]) for k in p)
    assert vae.config.scaling_factor == 0.13025 and vae.config.shift_factor is None


def test_hub_names_are_refused_offline():
    import pytest
    with pytest.raises(FileNotFoundError):
        M.LVM.from_pretrained("GrayShine/Video-GPT")
    with pytest.raises(FileNotFoundError):
        V.AutoencoderKL.from_pretrained("stabilityai/sdxl-vae")


def test_checkpoint_layouts_of_the_reference_inference_script(tmp_path):
    """model.pt, one pytorch_model.bin, or a pytorch_model.bin/ directory of shards (LVM/inference/...inference.py:48-68);
    torch files written by THIS test (plain tensor dicts) and read back with weights_only=True."""
    cfg = R.Phi3Cfg(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2,
                    num_key_value_heads=2, vocab_size=32, pos_embed_max_size=192)
    params = {k: v.contiguous() for k, v in R.make_params(cfg, seed=9).items()}
    keys = sorted(params)
    layouts = {}
    for name in ("pt", "bin", "shards"):
        d = str(tmp_path / name)
        _write_lvm_dir(d, cfg, params)
        os.remove(os.path.join(d, "model.safetensors"))
        layouts[name] = d
    torch.save(params, os.path.join(layouts["pt"], "model.pt"))
    torch.save(params, os.path.join(layouts["bin"], "pytorch_model.bin"))
    os.makedirs(os.path.join(layouts["shards"], "pytorch_model.bin"))
    half = len(keys) // 2
    torch.save({k: params[k] for k in keys[:half]}, os.path.join(layouts["shards"], "pytorch_model.bin", "shard-00001.bin"))
    torch.save({k: params[k] for k in keys[half:]}, os.path.join(layouts["shards"], "pytorch_model.bin", "shard-00002.bin"))
    for name, d in layouts.items():
        sd = M.LVM.from_pretrained(d).state_dict()
        assert all(torch.equal(sd[k], params[k]) for k in keys), name
    import pytest
    with pytest.raises(FileNotFoundError):
        M.load_checkpoint_state_dict(str(tmp_path))
