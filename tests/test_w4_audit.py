"""CPU test (hipcc cross-compiles without a GPU): the code objects of the four-wave GEMM kernels keep the contract their asm
main loop relies on -- see scripts/w4_audit.py."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_compiler_never_touches_the_accumulator_registers_of_the_asm_loop():
    spec = importlib.util.spec_from_file_location("w4_audit", os.path.join(ROOT, "scripts", "w4_audit.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.audit(verbose=False) == []


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_compiler_never_touches_the_state_registers_of_the_attention_bodies():
    """The hand-scheduled attention tile bodies keep O, m, l and S in v[72:191] from one asm statement to the next."""
    spec = importlib.util.spec_from_file_location("w4_audit", os.path.join(ROOT, "scripts", "w4_audit.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.audit_attn_p2(verbose=False) == []


def test_generated_loop_is_what_the_generator_emits():
    """csrc/gemm_w4_loop.inc is committed generator output: regenerate and compare (a hand edit, or a generator change
    without regenerating, fails here)."""
    import subprocess
    import sys
    gen = os.path.join(ROOT, "video-gpt_amd", "csrc", "gen", "gemm_w4_gen.py")
    out = subprocess.run([sys.executable, gen], check=True, capture_output=True, text=True).stdout
    assert out == open(os.path.join(ROOT, "video-gpt_amd", "csrc", "gemm_w4_loop.inc")).read()
    gen = os.path.join(ROOT, "video-gpt_amd", "csrc", "gen", "attn_p2_gen.py")
    out = subprocess.run([sys.executable, gen], check=True, capture_output=True, text=True).stdout
    assert out == open(os.path.join(ROOT, "video-gpt_amd", "csrc", "attn_p2_loop.inc")).read()
