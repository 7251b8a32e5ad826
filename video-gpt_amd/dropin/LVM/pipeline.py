from video_gpt_amd.pipeline import LVMPipeline  # noqa: F401
