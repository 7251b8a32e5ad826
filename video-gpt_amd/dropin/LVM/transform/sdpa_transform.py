from video_gpt_amd.transform import replace_attention, hip_sdpa  # noqa: F401
