"""The C-ABI library loads without a GPU and exports every symbol include/vgpt.h declares."""
import importlib
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "vgpt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vgpt_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    pkg = importlib.import_module("video-gpt_amd")
    if not os.path.exists(pkg._lib.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    return pkg._lib


def test_header_declares_the_expected_surface():
    syms = header_symbols()
    assert len(syms) >= 25
    for must in ("vgpt_gemm_bf16", "vgpt_attn_blockmask_fwd", "vgpt_rmsnorm_fwd", "vgpt_rope_qk_inplace",
                 "vgpt_gated_mlp_act_fwd", "vgpt_euler_cfg_update", "vgpt_mask_tile_summary", "vgpt_last_error"):
        assert must in syms


def test_library_exports_every_declared_symbol(lib):
    cdll = lib.load()
    missing = [s for s in header_symbols() if not hasattr(cdll, s)]
    assert missing == []


def test_python_binding_covers_header(lib):
    assert sorted(lib.SIGNATURES) == header_symbols()
    assert lib.missing_exports() == []


def test_abi_version_and_error_string(lib):
    cdll = lib.load()
    assert cdll.vgpt_abi_version() == lib.ABI_VERSION
    hdr = open(os.path.join(ROOT, "include", "vgpt.h")).read()
    assert int(re.search(r"#define VGPT_ABI_VERSION (\d+)", hdr).group(1)) == lib.ABI_VERSION
    # argument validation happens on the host before any launch: no GPU needed
    rc = cdll.vgpt_rmsnorm_fwd(None, None, None, 1, 64, 1e-5, None)
    assert rc == -1 and b"null pointer" in cdll.vgpt_last_error()
    rc = cdll.vgpt_attn_supported(96), cdll.vgpt_attn_supported(80)
    assert rc == (1, 0)


def test_stale_library_is_refused(lib, monkeypatch):
    """A library whose vgpt_abi_version() differs from the constant the binding was written for is never called."""
    lib.load()
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "ABI_VERSION", lib.ABI_VERSION + 1)
    with pytest.raises(lib.VgptError, match="ABI version"):
        lib.load()


def test_product_path_has_no_cpu_fallback(lib):
    import torch
    ops = importlib.import_module("video-gpt_amd.ops")
    with pytest.raises(lib.VgptError, match="GPU tensor"):
        ops.rmsnorm(torch.zeros(2, 64, dtype=torch.bfloat16), torch.ones(64, dtype=torch.bfloat16), 1e-5)


def test_product_never_imports_oracle():
    pkg_dir = os.path.join(ROOT, "video-gpt_amd")
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"
