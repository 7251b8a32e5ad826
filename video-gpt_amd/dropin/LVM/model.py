from video_gpt_amd.model import *  # noqa: F401,F403
from video_gpt_amd.model import LVM, LVMTraining, LVMTraining_CP, TimestepEmbedder, FinalLayer, PatchEmbedMR  # noqa: F401
