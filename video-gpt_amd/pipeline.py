"""LVMPipeline: next-clip autoregressive video prediction (mirror of LVM/pipeline.py:46-135,347-595).

Host orchestration only — windowing, prompt strings, noise draw, VAE encode of the condition frames,
sampler call, VAE decode, uint8 conversion, chaining clips — with every tensor op on the HIP path:
VAE through `vae.AutoencoderKL`, the sampler through `scheduler.LVMScheduler` (hipGraph fast path).

Behaviours kept from the reference (and worth knowing):
  * every round re-encodes the previous round's DECODED frames as the new condition (pipeline.py:419-420);
  * the noise generator is re-seeded with the same `seed` every round (:473-476);
  * `samples = samples[:len(samples)//2]` is applied unconditionally (:549) — correct with image CFG
    (the second half is the unconditional branch), and it silently drops half of the frames when CFG
    is off; `halve_without_cfg=False` opts out of that.
Differences: `generator_device="cpu"` draws the noise with a CPU generator and copies it to the GPU
(needed for "identical noise seeds" against the CPU oracle; the reference's default is a device
generator); model CPU offload / KV-cache flags are accepted and ignored (288 GB HBM; the reference's
sampler never uses its cache, LVM/scheduler.py:174).
"""
from __future__ import annotations

from typing import List, Optional, Union

import torch

from . import ops, ops_train
from .model import LVM
from .processor import LVMProcessor
from .scheduler import LVMScheduler


class LVMPipeline:
    def __init__(self, vae, model: LVM, processor: LVMProcessor, device: Union[str, torch.device] = None):
        self.vae, self.model, self.processor = vae, model, processor
        if device is None:
            if not torch.cuda.is_available():
                raise ops.VgptError("LVMPipeline needs an MI355X: there is no CPU path")
            device = torch.device("cuda")
        self.device = torch.device(device)
        self.model.eval()
        self.vae.eval()
        self.model_cpu_offload = False
        self.last_latents = None
        self.mask_format = "layout"   # "bool": have the collator paint the reference's dense (B,L,L) mask instead
        self.attention_precision = "bf16"   # "fp8": MX-fp8 attention in the sampler steps (scheduler.LVMScheduler)

    @classmethod
    def from_pretrained(cls, model_name, vae_path: str = None, load_llm_ckpt=True):
        """Local directories only (no hub access): <model_name>/{config.json, model.safetensors|model.pt[, vae/]}."""
        import os
        from .vae import AutoencoderKL
        model = LVM.from_pretrained(model_name, load_llm_ckpt=load_llm_ckpt)
        processor = LVMProcessor.from_pretrained(model_name)
        if os.path.exists(os.path.join(model_name, "vae")):
            vae = AutoencoderKL.from_pretrained(os.path.join(model_name, "vae"))
        elif vae_path is not None:
            vae = AutoencoderKL.from_pretrained(vae_path)
        else:
            raise FileNotFoundError("no VAE directory given and stabilityai/sdxl-vae cannot be downloaded offline")
        return cls(vae, model, processor)

    def to(self, device: Union[str, torch.device]):
        self.device = torch.device(device)
        self.model.to(self.device)
        self.vae.to(self.device)

    def enable_model_cpu_offload(self):
        self.model_cpu_offload = False  # not needed on a 288 GB device; kept for interface parity

    def disable_model_cpu_offload(self):
        self.model_cpu_offload = False

    def move_to_device(self, data):
        if isinstance(data, list):
            return [x.to(self.device) for x in data]
        return data.to(self.device)

    def vae_encode(self, x, dtype, noise: Optional[torch.Tensor] = None, dist=None):
        """LVM/pipeline.py:110-117: sample the posterior, (z - shift) * scaling, cast to `dtype`.  `dist`: the
        posterior of this image when the encoder already ran on a batch (vae_posteriors)."""
        d = dist if dist is not None else self.vae.encode(x).latent_dist
        if noise is None:
            p = d.parameters
            noise = torch.randn(p.shape[0], p.shape[1] // 2, *p.shape[2:], device=p.device, dtype=torch.float32)
        return d.sample_scaled(noise, self.vae.config.shift_factor or 0.0, self.vae.config.scaling_factor).to(dtype)

    def vae_posteriors(self, imgs):
        """Encoder convolutions of all condition frames of a round in ONE batched pass (the reference encodes them one
        by one, LVM/pipeline.py:482-486; the convolutions are per-sample, so the moments are the same and the
        launches are 4-16x larger).  Sampling stays per image, in the reference's RNG order."""
        from .vae import DiagonalGaussianDistribution
        if len(imgs) > 1 and all(t.shape == imgs[0].shape for t in imgs):
            mom = self.vae.encode(torch.cat(imgs, dim=0)).latent_dist.parameters
            return [DiagonalGaussianDistribution(mom[i:i + 1]) for i in range(len(imgs))]
        return [self.vae.encode(t).latent_dist for t in imgs]

    def _to_images(self, u8: torch.Tensor, output_type: str):
        if output_type == "pt":
            return [u8[i] for i in range(u8.shape[0])]
        from PIL import Image
        arr = u8.cpu().numpy()
        return [Image.fromarray(arr[i]) for i in range(arr.shape[0])]

    def _rebuilt_processor(self, max_input_image_size: int):
        """A processor for another max_input_image_size keeps the sequence-parallel padding of the one it replaces (the
        reference passes hccl_info.world_size, LVM/pipeline.py:232) and its mask format."""
        old = self.processor
        new = LVMProcessor(old.text_tokenizer, max_image_size=max_input_image_size,
                           sequence_parallel_size=getattr(old, "sequence_parallel_size", 1))
        new.collator.mask_format = old.collator.mask_format
        return new

    @torch.no_grad()
    def __call__(self, input_images=None, height: int = 1024, width: int = 1024, gen_num: int = 1,
                 num_inference_steps: int = 50, use_img_guidance: bool = True, img_guidance_scale: float = 1.6,
                 max_input_image_size: int = 1024, offload_model: bool = False, use_kv_cache: bool = True,
                 offload_kv_cache: bool = True, use_input_image_size_as_output: bool = False,
                 dtype: torch.dtype = torch.bfloat16, seed: int = None, output_type: str = "pil",
                 time_shifting_factor: float = 1.0, prediction_type: str = "v", clean_image_noise_level: float = None,
                 generator_device: str = "cuda", vae_noise: Optional[List[torch.Tensor]] = None,
                 renoise_noise: Optional[List[torch.Tensor]] = None):
        """Single-target generation, one image per round, every generated image joining the condition images of the next
        round (LVM/pipeline.py:138-343): sequence [<img> image </img> ... <|diffusion|> | time token | target tokens]
        through `LVM.forward_with_cfg`; CFG row = the empty prompt.  Returns the decoded condition images of the first
        round followed by the generated images."""
        prompt_img_len = len(input_images) if input_images is not None else 0
        if not use_input_image_size_as_output:
            assert height % 16 == 0 and width % 16 == 0, "The height and width must be a multiple of 16."
        if dtype != torch.bfloat16:
            raise ops.VgptError("the HIP denoiser computes in bf16")
        ori_input_images = list(input_images) if input_images is not None else None
        output_images, output_image = [], None
        if img_guidance_scale == 1:
            use_img_guidance = False
        ori_use_img_guidance = use_img_guidance
        self.last_latents, self.last_samples = [], []
        n_renoised = 0
        for gen_idx in range(gen_num):
            if len(output_images) != 0:
                ori_input_images = [output_image] if ori_input_images is None else ori_input_images + [output_image]
            use_img_guidance = False if ori_input_images is None else ori_use_img_guidance
            prompt = "".join(f"<img><|image_{i + 1}|></img>" for i in range(len(ori_input_images or [])))
            if max_input_image_size != self.processor.max_image_size:
                self.processor = self._rebuilt_processor(max_input_image_size)
            self.processor.collator.hidden_size = self.model.hidden_size
            self.model.to(self.device, dtype)
            input_data = self.processor([prompt], [list(ori_input_images)] if ori_input_images is not None else None,
                                        height=height, width=width, use_img_cfg=use_img_guidance,
                                        use_input_image_size_as_output=use_input_image_size_as_output)
            num_cfg = 1 if use_img_guidance else 0
            if use_input_image_size_as_output:
                height, width = input_data["input_pixel_values"][0].shape[-2:]
            lh, lw = height // 8, width // 8
            gdev = self.device if generator_device == "cuda" else torch.device("cpu")
            generator = torch.Generator(device=gdev).manual_seed(seed) if seed is not None else None
            latents = torch.randn(1, 4, lh, lw, device=gdev, generator=generator).to(self.device)
            latents = torch.cat([latents] * (1 + num_cfg), 0).to(dtype)

            input_img_latents = []
            dists = self.vae_posteriors([img.to(self.device) for img in input_data["input_pixel_values"]])
            for idx, d in enumerate(dists):
                vn = vae_noise[len(self.last_latents)] if vae_noise is not None else None
                lat = self.vae_encode(None, dtype, None if vn is None else vn.to(self.device), dist=d)
                self.last_latents.append(lat)
                if idx >= prompt_img_len:   # a generated image fed back as a condition is re-noised (pipeline.py:256-257)
                    c = clean_image_noise_level
                    if renoise_noise is not None:   # replay of a recorded reference run (tests)
                        noise = renoise_noise[n_renoised].to(lat.device, torch.float32).reshape(lat.shape)
                        n_renoised += 1
                    else:
                        noise = torch.randn(lat.shape, device=lat.device, dtype=torch.float32)
                    cvec = torch.full((lat.shape[0],), float(c), device=lat.device, dtype=torch.float32)
                    lat = ops_train.lerp_frames(noise, lat.float().contiguous(), cvec,
                                                 torch.empty(lat.shape, device=lat.device, dtype=torch.bfloat16)).to(dtype)
                input_img_latents.append(lat)

            model_kwargs = dict(input_ids=self.move_to_device(input_data["input_ids"]), input_img_latents=input_img_latents,
                                input_image_sizes=input_data["input_image_sizes"],
                                attention_mask=self.move_to_device(input_data["attention_mask"]),
                                position_ids=self.move_to_device(input_data["position_ids"]),
                                img_cfg_scale=img_guidance_scale, use_img_cfg=use_img_guidance, use_kv_cache=use_kv_cache,
                                offload_model=False)
            scheduler = LVMScheduler(num_steps=num_inference_steps, time_shifting_factor=time_shifting_factor)
            scheduler.attention_precision = self.attention_precision
            samples = scheduler(latents, self.model.forward_with_cfg, model_kwargs, use_kv_cache=use_kv_cache,
                                offload_kv_cache=offload_kv_cache, prediction_type=prediction_type, vae=self.vae)
            samples = samples.chunk(1 + num_cfg, dim=0)[0]
            self.last_samples.append(samples)
            if gen_idx == 0 and input_img_latents:
                # the single-target path lets every condition image keep its own cropped resolution (process_image /
                # crop_arr; the reference decodes them one by one, LVM/pipeline.py:307-318): one batched decode only when
                # all latents share a shape
                if len({tuple(t.shape) for t in input_img_latents}) == 1:
                    output_images.extend(self._to_images(self.vae.decode_to_uint8(torch.cat(input_img_latents, dim=0)), output_type))
                else:
                    for lat in input_img_latents:
                        output_images.extend(self._to_images(self.vae.decode_to_uint8(lat), output_type))
            u8 = self.vae.decode_to_uint8(samples)
            output_image = self._to_images(u8[:1], output_type)[0]
            output_images.append(output_image)
        return output_images

    @torch.no_grad()
    def prompt_condition_frame_block_autoregressive_inference(
            self, input_images=None, height: int = 1024, width: int = 1024, gen_nums: list = [1],
            num_inference_steps: int = 50, use_img_guidance: bool = True, img_guidance_scale: float = 1.6,
            max_input_image_size: int = 1024, offload_model: bool = False, use_kv_cache: bool = True,
            offload_kv_cache: bool = True, use_input_image_size_as_output: bool = False,
            dtype: torch.dtype = torch.bfloat16, seed: int = None, output_type: str = "pil",
            time_shifting_factor: float = 1.0, prediction_type: str = "v", clean_image_noise_level: float = None,
            max_frame_window: int = 16, generator_device: str = "cuda", vae_noise: Optional[List[torch.Tensor]] = None,
            halve_without_cfg: bool = True, renoise_noise: Optional[List[torch.Tensor]] = None):
        """LVM/pipeline.py:347-595.  Not in the reference's signature: generator_device, vae_noise / renoise_noise (the
        noise of the VAE posterior samples and of the condition re-noising, in draw order, for tests that replay a recorded
        reference run) and halve_without_cfg."""
        if not use_input_image_size_as_output:
            assert height % 16 == 0 and width % 16 == 0, "The height and width must be a multiple of 16."
        if dtype != torch.bfloat16:
            raise ops.VgptError("the HIP denoiser computes in bf16")
        output_images = []
        if img_guidance_scale == 1:
            use_img_guidance = False
        if input_images is None:
            use_img_guidance = False
        self.last_latents, self.last_samples = [], []
        n_renoised = 0
        for k, gen_num in enumerate(gen_nums):
            if k > 0:
                input_images = output_images
            if len(input_images) + gen_num > max_frame_window:
                input_images = input_images[gen_num + len(input_images) - max_frame_window:]
            prompt_img_len = len(input_images) if input_images is not None else 0
            prompt = "".join(f"<img><|image_{i + 1}|></img>" if i < prompt_img_len else f"<|diffusion|><|image_{i + 1}|>"
                             for i in range(prompt_img_len + gen_num))
            frame_blocks = [prompt_img_len, gen_num]
            if use_img_guidance:
                prompt_ = "".join(f"<|diffusion|><|image_{i + 1}|>" for i in range(gen_num))
                prompts, images = [prompt, prompt_], [list(input_images), []]
            else:
                prompts, images = [prompt], [list(input_images)]
            if max_input_image_size != self.processor.max_image_size:
                self.processor = self._rebuilt_processor(max_input_image_size)
            # the sampler path never materialises the (B,L,L) mask: the collator hands over per-token attributes and
            # the device expands them into the packed rows the attention kernel reads (layout.TokenLayout)
            self.processor.collator.mask_format = self.mask_format
            self.model.to(self.device, dtype)
            input_data = self.processor.prompt_condition_frame_block_inference(
                prompts, images, height=height, width=width, use_img_cfg=use_img_guidance,
                use_input_image_size_as_output=use_input_image_size_as_output, frame_blocks=frame_blocks)
            num_cfg = 1 if use_img_guidance else 0
            if use_input_image_size_as_output:
                height, width = input_data["input_pixel_values"][0].shape[-2:]
            lh, lw = height // 8, width // 8

            gdev = self.device if generator_device == "cuda" else torch.device("cpu")
            generator = torch.Generator(device=gdev).manual_seed(seed) if seed is not None else None
            latents = [torch.randn(1, 4, lh, lw, device=gdev, generator=generator).to(self.device, dtype)
                       for _ in range(gen_num)]
            latents = latents * (1 + num_cfg)

            input_img_latents = []
            dists = self.vae_posteriors([input_data["input_pixel_values"][idx].to(self.device)
                                         for idx in range(prompt_img_len)])
            for idx in range(prompt_img_len):
                vn = vae_noise[len(self.last_latents)] if vae_noise is not None else None
                lat = self.vae_encode(None, dtype, None if vn is None else vn.to(self.device), dist=dists[idx])
                self.last_latents.append(lat)
                if k > 0:  # re-noise the re-encoded condition frames (pipeline.py:496-497); 16 KB blend, torch RNG
                    c = clean_image_noise_level
                    # (1 - c) * lat + c * noise through the HIP lerp kernel (fp32 in, bf16 out); the noise is torch's RNG
                    if renoise_noise is not None:
                        noise = renoise_noise[n_renoised].to(lat.device, torch.float32).reshape(lat.shape)
                        n_renoised += 1
                    else:
                        noise = torch.randn(lat.shape, device=lat.device, dtype=torch.float32)
                    cvec = torch.full((lat.shape[0],), float(c), device=lat.device, dtype=torch.float32)
                    lat = ops_train.lerp_frames(noise, lat.float().contiguous(), cvec,
                                                 torch.empty(lat.shape, device=lat.device, dtype=torch.bfloat16)).to(dtype)
                input_img_latents.append(lat)

            model_kwargs = dict(
                input_ids=self.move_to_device(input_data["input_ids"]), input_img_latents=input_img_latents,
                input_image_sizes=input_data["input_image_sizes"],
                attention_mask=self.move_to_device(input_data["attention_mask"]),
                position_ids=self.move_to_device(input_data["position_ids"]),
                denoise_image_sizes=input_data["denoise_image_sizes"], time_emb_inx=input_data["time_emb_inx"],
                img_cfg_scale=img_guidance_scale, use_img_cfg=use_img_guidance, use_kv_cache=use_kv_cache,
                offload_model=False, vae=self.vae)
            scheduler = LVMScheduler(num_steps=num_inference_steps, time_shifting_factor=time_shifting_factor)
            scheduler.attention_precision = self.attention_precision
            samples = scheduler(latents, self.model.frame_block_forward_with_cfg, model_kwargs,
                                use_kv_cache=use_kv_cache, offload_kv_cache=offload_kv_cache,
                                prediction_type=prediction_type, vae=self.vae)
            if use_img_guidance or halve_without_cfg:
                samples = samples[:len(samples) // 2]
            self.last_samples.append(samples)

            if k == 0 and input_img_latents:
                u8 = self.vae.decode_to_uint8(torch.cat(input_img_latents, dim=0))
                output_images.extend(self._to_images(u8, output_type))
            if samples:
                u8 = self.vae.decode_to_uint8(torch.cat(samples, dim=0))
                output_images.extend(self._to_images(u8, output_type))
        return output_images
