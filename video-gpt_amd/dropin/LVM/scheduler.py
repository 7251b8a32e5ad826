from video_gpt_amd.scheduler import LVMScheduler  # noqa: F401
