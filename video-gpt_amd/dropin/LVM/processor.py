from video_gpt_amd.processor import LVMCollator, LVMProcessor  # noqa: F401
