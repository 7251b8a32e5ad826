from video_gpt_amd.train_data import TrainDataCollator, TrainDataCollator_FrameBlock  # noqa: F401
