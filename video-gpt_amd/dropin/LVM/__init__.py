"""Drop-in import shim: put `video-gpt_amd/dropin` on PYTHONPATH and the reference's import lines
(`from LVM import LVMPipeline`, `from LVM.acceleration.parallel_states import init_npu_env, hccl_info`, ...) resolve to
the MI355X implementation (same names as the reference's LVM/__init__.py:1-4)."""
import importlib as _il

_pkg = _il.import_module("video-gpt_amd")
from video_gpt_amd.model import LVM, LVMTraining, LVMTraining_CP  # noqa: E402,F401
from video_gpt_amd.processor import LVMCollator, LVMProcessor  # noqa: E402,F401
from video_gpt_amd.scheduler import LVMScheduler  # noqa: E402,F401
from video_gpt_amd.pipeline import LVMPipeline  # noqa: E402,F401
