"""video-gpt_amd — MI355X (gfx950) native engine for the Video-GPT next-clip diffusion hot path.

The directory name contains a hyphen (it mirrors the reference repo name), so import it with
`importlib.import_module("video-gpt_amd")`; on import the package also registers itself as
`video_gpt_amd` in `sys.modules`, after which `import video_gpt_amd.ops` works as usual.

Layout
  csrc/            hand-written HIP kernels + the C ABI (include/vgpt.h) -> libvgpt_hip.so
  _lib.py, ops.py  ctypes binding and tensor-level wrappers (no CPU fallback)
  processor.py     LVMCollator mirror (ids / positions / block masks / index dicts)
  model.py         LVM / LVMTraining mirrors running on the HIP kernels
  scheduler.py     LVMScheduler mirror (Euler loop, hipGraph-captured fast path)
  pipeline.py      LVMPipeline mirror (next-clip autoregressive inference)
  transform.py     replace_attention operator seam
"""
import sys as _sys

_sys.modules.setdefault("video_gpt_amd", _sys.modules[__name__])

from . import _lib  # noqa: E402,F401

__all__ = ["_lib"]
