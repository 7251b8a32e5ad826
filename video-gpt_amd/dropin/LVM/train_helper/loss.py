from video_gpt_amd.loss import (training_losses_x1_noise_input, draw_training_noise, sample_x0, sample_timestep,  # noqa: F401
                                sample_exp_timestep, sample_frame_block_timestep, sample_timestep_max_noise, mean_flat,
                                is_all_equal)
