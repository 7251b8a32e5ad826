"""Pin the CPU oracle (oracle/restate.py) before trusting it:
  * against vectors produced by the reference's own classes (tests/golden/ref_*.npz): LVMScheduler,
    TimestepEmbedder, FinalLayer, PatchEmbedMR, 2-D sincos table;
  * against the installed transformers Phi3 classes for the decoder layer (the reference calls the
    un-vendored transformers==4.47.1; this is a live third-party cross-check, not the reference);
  * against its own frozen outputs for the full tiny case (oracle_tiny_e2e.npz; parity unpinned by
    the reference, which holds no test or fixture for the assembled model).
The product's host-side pieces that do not need a GPU (sigma table, position table) are checked here too."""
import importlib
import os

import numpy as np
import pytest
import torch

from oracle import restate as R
from tests import smoke_case as SC

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def T(a):
    return torch.from_numpy(np.asarray(a))


# ---- LVMScheduler (LVM/scheduler.py:119-208) -------------------------------------------------

def _stub(a, c, use_cfg, scale):
    def func(z, t):
        pred = [a[j] * z[j] + c[j] * (1 + t[j]) for j in range(len(z))]
        return pred
    return func


@pytest.mark.parametrize("steps", [1, 3])
@pytest.mark.parametrize("pt", ["x1", "v"])
@pytest.mark.parametrize("cfg_on", [True, False])
@pytest.mark.parametrize("shift,begin", [(1, None), (3.0, 0.2)])
def test_scheduler_matches_reference(steps, pt, cfg_on, shift, begin):
    d = np.load(os.path.join(GOLD, "ref_scheduler.npz"))
    key = f"steps{steps}_{pt}_cfg{int(cfg_on)}_shift{shift}_begin{begin}"
    z0, a, c = [list(T(d[k])) for k in ("z0", "a", "c")]
    sigma = R.scheduler_sigma(steps, shift, begin)
    assert torch.equal(sigma, T(d["sigma_" + key]))
    S = importlib.import_module("video-gpt_amd.scheduler")
    assert torch.equal(S.LVMScheduler(steps, shift, begin).sigma, sigma)

    def func(z, t):
        pred = [a[j] * z[j] + c[j] * (1 + t[j]) for j in range(len(z))]
        if cfg_on and pt == "v":
            h = len(pred) // 2
            cond = [pred[h + j] + 1.6 * (pred[j] - pred[h + j]) for j in range(h)]
            pred = cond + cond
        return pred
    out = torch.stack(R.scheduler_call(sigma, z0, func, cfg_on, 1.6, pt))
    assert torch.allclose(out, T(d[key]), rtol=0, atol=1e-6)


# ---- leaf modules (LVM/model.py:22-154) --------------------------------------------------------

def test_leaf_modules_match_reference():
    d = np.load(os.path.join(GOLD, "ref_leaf_modules.npz"))
    t = T(d["t"])
    assert torch.allclose(R.timestep_embedding(t, 256), T(d["te_sin"]), atol=1e-7)
    p = {"tt." + k[3:]: T(d[k]) for k in d.files if k.startswith("te.")}
    assert torch.allclose(R.timestep_embedder(p, "tt", t, torch.float32), T(d["te_out"]), atol=1e-6)
    pf = {"final_layer." + k[3:]: T(d[k]) for k in d.files if k.startswith("fl.")}
    assert torch.allclose(R.final_layer(pf, T(d["fl_x"]), T(d["fl_c"])), T(d["fl_out"]), atol=1e-5)
    assert torch.allclose(R.patch_embed(T(d["pe_x"]), T(d["pe.proj.weight"]), T(d["pe.proj.bias"]), 2), T(d["pe_out"]), atol=1e-6)
    assert np.array_equal(R.sincos_2d(64, 12, 1.0, 64), d["sincos_64_12_b64"])
    assert np.array_equal(R.sincos_2d(32, 7, 2.0, 1), d["sincos_32_7_b1_i2"])
    M = importlib.import_module("video-gpt_amd.model")
    assert np.array_equal(M.get_2d_sincos_pos_embed(64, 12, interpolation_scale=1.0, base_size=64), d["sincos_64_12_b64"])
    assert np.array_equal(M.get_2d_sincos_pos_embed(32, 7, interpolation_scale=2.0, base_size=1), d["sincos_32_7_b1_i2"])


def test_unpatchify_roundtrip():
    x = torch.arange(2 * 6 * 16, dtype=torch.float32).reshape(2, 6, 16)
    img = R.unpatchify(x, 4, 6, 2, 4)
    assert img.shape == (2, 4, 4, 6)
    # token (i,j) element (p*2+q)*C + c lands at [c, 2i+p, 2j+q]
    assert img[1, 3, 2 * 1 + 1, 2 * 2 + 0] == x[1, 1 * 3 + 2, (1 * 2 + 0) * 4 + 3]


# ---- Phi3 decoder layer vs installed transformers (formulas of 4.47.1 == 5.x) --------------------

def test_decoder_layer_matches_installed_transformers():
    tf = pytest.importorskip("transformers")
    try:
        from transformers import Phi3Config
        from transformers.models.phi3.modeling_phi3 import Phi3DecoderLayer, Phi3RotaryEmbedding
    except Exception as e:  # pragma: no cover
        pytest.skip(f"transformers Phi3 classes unavailable: {e}")
    cfg = R.TINY
    p, batch, _, _ = SC.build_case(cfg)
    hc = Phi3Config(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                    num_hidden_layers=1, num_attention_heads=cfg.num_attention_heads,
                    num_key_value_heads=cfg.num_key_value_heads, rms_norm_eps=cfg.rms_norm_eps, pad_token_id=2,
                    attn_implementation="eager")
    layer = Phi3DecoderLayer(hc, 0).eval()
    sd = {k[len("llm.layers.0."):]: v for k, v in p.items() if k.startswith("llm.layers.0.")}
    layer.load_state_dict(sd, strict=True)
    rot = Phi3RotaryEmbedding(hc)
    g = torch.Generator("cpu").manual_seed(9)
    B, L = batch["input_ids"].shape
    x = torch.randn(B, L, cfg.hidden_size, generator=g) * 0.5
    pos = batch["position_ids"]
    amask = R.additive_mask(batch["attention_mask"], torch.float32)
    with torch.no_grad():
        out = layer(x, attention_mask=amask, position_ids=pos, position_embeddings=rot(x, pos))
        out = out[0] if isinstance(out, tuple) else out
        cos, sin = R.rope_cos_sin(pos, cfg.head_dim, cfg.rope_theta, torch.float32)
        h = x + R.attention(p, cfg, 0, R.rmsnorm(x, p["llm.layers.0.input_layernorm.weight"], cfg.rms_norm_eps), amask, cos, sin)
        ref = h + R.mlp(p, cfg, 0, R.rmsnorm(h, p["llm.layers.0.post_attention_layernorm.weight"], cfg.rms_norm_eps))
    assert float((out - ref).abs().max()) <= 1e-5


# ---- frozen oracle outputs for the assembled tiny model ------------------------------------------

def test_oracle_tiny_case_is_frozen():
    d = np.load(os.path.join(GOLD, "oracle_tiny_e2e.npz"))
    cfg = R.TINY
    p, batch, z, cond = SC.build_case(cfg)
    out = torch.cat(SC.oracle_sample(cfg, p, batch, z, cond, 3, "x1"))
    assert torch.allclose(out, T(d["sample3_x1"]), atol=2e-5)
    okw = {k: batch[k] for k in ("input_ids", "input_image_sizes", "attention_mask", "position_ids",
                                 "denoise_image_sizes", "time_emb_inx")}
    fwd = torch.cat(R.frame_block_forward_with_cfg(p, cfg, z, torch.full((len(z),), 0.3), True, 1.6, "v",
                                                   input_img_latents=cond, **okw))
    assert torch.allclose(fwd, T(d["fwd_v"]), atol=2e-5)


def test_oracle_mask_semantics_additive_min_equals_bool():
    """finfo.min additive mask (OmniGen/transformer.py:139-145) == excluding masked keys when every
    row has a visible key — the property the bit-packed kernel mask relies on."""
    m = R.collate_inference(2, 2, 4)["attention_mask"]
    assert m.any(-1).all()
    g = torch.Generator("cpu").manual_seed(1)
    s = torch.randn(2, 1, m.shape[1], m.shape[1], generator=g) * 5
    a = torch.softmax(s + R.additive_mask(m, torch.float32), -1)
    b = torch.softmax(s.masked_fill(~m[:, None], float("-inf")), -1)
    assert torch.equal(a, b)


def test_zero_init_heads_make_default_model_output_zero():
    """LVM/model.py:241-244: freshly constructed heads are zero -> fixtures must re-randomise them."""
    M = importlib.import_module("video-gpt_amd.model")
    m = M.LVMTraining(M.Phi3Config(vocab_size=16, hidden_size=64, intermediate_size=64, num_hidden_layers=1,
                                   num_attention_heads=1), pos_embed_max_size=8)
    assert float(m.final_layer.linear.weight.abs().sum()) == 0 and float(m.x_embedder.proj.weight.abs().sum()) == 0
    assert float(M.LVM(m.llm.config, pos_embed_max_size=8).x_embedder.proj.weight.abs().sum()) > 0
