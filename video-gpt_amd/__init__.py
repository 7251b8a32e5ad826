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
import importlib as _importlib
import importlib.abc as _abc
import importlib.machinery as _machinery
import sys as _sys

_sys.modules.setdefault("video_gpt_amd", _sys.modules[__name__])


class _AliasFinder(_abc.MetaPathFinder, _abc.Loader):
    """`video_gpt_amd.x` IS `video-gpt_amd.x`: one module object under both names (a second copy of model.py would
    carry its own LVM class and break every isinstance check between code imported under the two spellings)."""

    def find_spec(self, fullname, path=None, target=None):
        if fullname.startswith("video_gpt_amd."):
            return _machinery.ModuleSpec(fullname, self, origin=__name__ + fullname[len("video_gpt_amd"):])
        return None

    def create_module(self, spec):
        return _importlib.import_module(spec.origin)

    def exec_module(self, module):
        return None


if not any(isinstance(f, _AliasFinder) or type(f).__name__ == "_AliasFinder" for f in _sys.meta_path):
    _sys.meta_path.insert(0, _AliasFinder())

from . import _lib  # noqa: E402,F401

__all__ = ["_lib"]
