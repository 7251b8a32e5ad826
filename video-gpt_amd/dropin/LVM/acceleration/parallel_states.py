from video_gpt_amd.parallel_states import *  # noqa: F401,F403
from video_gpt_amd.parallel_states import (COMM_INFO, hccl_info, init_npu_env, initialize_sequence_parallel_state,  # noqa: F401
                                           initialize_sequence_parallel_group, destroy_sequence_parallel_group,
                                           get_sequence_parallel_state, set_sequence_parallel_state)
