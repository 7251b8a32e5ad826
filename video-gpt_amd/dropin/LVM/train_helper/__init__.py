from .data import TrainDataCollator, TrainDataCollator_FrameBlock  # noqa: F401
from .loss import training_losses_x1_noise_input  # noqa: F401
