from video_gpt_amd.vae import vae_encode, vae_encode_list  # noqa: F401
