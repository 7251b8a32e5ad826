"""LVMScheduler: Euler rectified-flow sampler (mirror of LVM/scheduler.py:119-208).

`__call__(z, func, model_kwargs, ...)` keeps the reference contract:
`func(z, timesteps, past_key_values=None, prediction_type=..., **model_kwargs) -> (pred, cache)`.

Two execution paths, same arithmetic:
  * fast path — `func` is `LVM.frame_block_forward_with_cfg` of this package and the latents are a
    list of equal-shape frames: the whole step (model + x1->v + CFG + Euler) runs from one
    hipGraph captured once per clip (engine.StaticDenoiser);
  * generic path — any other `func`: the model call is the caller's, the x1->v / CFG / Euler update
    is one HIP kernel per step.
The sampler state is kept in fp32 and rounded to the model dtype only where it enters the model
(the reference rounds the state to bf16 after every step; an fp32 state is closer to its fp32 CPU
path).  KV caching is not used, exactly as the reference passes `past_key_values=None` every step
(LVM/scheduler.py:174).
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch

from . import ops
from .ops import BF16, VgptError


class LVMScheduler:
    def __init__(self, num_steps: int = 50, time_shifting_factor: int = 1, begin_time=None):
        self.num_steps = num_steps
        self.time_shift = time_shifting_factor
        t = torch.linspace(0 if begin_time is None else begin_time, 1, num_steps + 1)
        self.sigma = t / (t + time_shifting_factor - time_shifting_factor * t)
        self.use_graph = True
        self.pack_padding = True
        self.reuse_condition_prefix = True   # compute the step-invariant condition rows once per clip (engine.py)
        self.hoist_special_rows = True       # ... and the <|diffusion|> / time rows of all steps in one pass (needs a TokenLayout mask)
        self.attention_precision = "bf16"    # "fp8": MX-fp8 attention in the sampler steps of the fast path (cfg-5 option)
        self.fuse_norms = None               # None: RMSNorms folded into the GEMMs wherever the step's shapes allow (engine.py); False: never
        # keep the engine of a clip on the model and re-use it for the next clip of an identical sequence (the rounds of a
        # rollout once the window is full: LVM/pipeline.py:418-422 re-creates the same prompt every round): buffers, attention
        # plan and the captured graph survive, the per-clip pass is redone on the new condition latents
        self.cache_engines = os.environ.get("VGPT_ENGINE_CACHE", "1") != "0"
        self.last_engine = None
        self.last_engine_reused = False

    # ---- fast path ----
    def _fast_path_engine(self, z, func, model_kwargs, prediction_type):
        from .engine import StaticDenoiser
        from .model import LVM
        owner = getattr(func, "__self__", None)
        if not isinstance(owner, LVM) or getattr(func, "__name__", "") != "frame_block_forward_with_cfg":
            return None
        if not isinstance(z, (list, tuple)) or len({tuple(t.shape) for t in z}) != 1:
            return None
        lat = model_kwargs.get("input_img_latents")
        if lat is not None and len(lat) > 0 and len({tuple(t.shape) for t in lat}) != 1:
            return None
        need = ("input_ids", "input_image_sizes", "attention_mask", "position_ids", "denoise_image_sizes",
                "time_emb_inx", "use_img_cfg", "img_cfg_scale")
        if any(k not in model_kwargs for k in need) or model_kwargs.get("offload_model"):
            return None
        self.last_engine_reused = False
        from .train import wait_for_pending_update
        wait_for_pending_update(owner)      # a trainer's overlapped AdamW update of these parameters may still be in flight
        self._owner_params = list(owner.parameters())
        key = self._engine_key(z, model_kwargs, prediction_type) if self.cache_engines else None
        cache = owner.__dict__.setdefault("_vgpt_engine_cache", {}) if key is not None else None
        if cache is not None and key in cache:
            eng = cache.pop(key)
            cache[key] = eng                     # most recently used last
            self.last_engine_reused = True
            return eng.rebind(lat)
        eng = self._build_engine(StaticDenoiser, owner, z, lat, model_kwargs, prediction_type)
        if cache is not None:
            cache[key] = eng
            while len(cache) > 2:                # a clip's buffers are a few GB at cfg-2: the last two layouts are kept
                cache.pop(next(iter(cache)))
        return eng

    def _engine_key(self, z, model_kwargs, prediction_type):
        """Everything a StaticDenoiser is built from except the condition latents' VALUES; None when the mask is not a
        TokenLayout (a dense mask would have to be compared element by element)."""
        from .layout import TokenLayout
        mask = model_kwargs["attention_mask"]
        if not isinstance(mask, TokenLayout):
            return None
        lat = model_kwargs.get("input_img_latents")
        tb = lambda t: t.detach().cpu().numpy().tobytes()
        return (tb(model_kwargs["input_ids"]), tb(model_kwargs["position_ids"]), mask.attr().tobytes(),
                repr(model_kwargs["input_image_sizes"]), repr(model_kwargs["denoise_image_sizes"]), repr(model_kwargs["time_emb_inx"]),
                len(z), tuple(z[0].shape), None if not lat else (len(lat), tuple(lat[0].shape)),
                bool(model_kwargs["use_img_cfg"]), float(model_kwargs["img_cfg_scale"]), prediction_type, tb(self.sigma),
                self.pack_padding, self.reuse_condition_prefix, self.hoist_special_rows, self.attention_precision,
                self.fuse_norms, str(z[0].device),
                # the captured graph holds the parameters' device addresses: parameters moved or re-allocated since
                # (model.to(...), a new state dict assigned tensor by tensor) must not meet a cached graph
                tuple(p_.data_ptr() for p_ in self._owner_params))

    def _build_engine(self, StaticDenoiser, owner, z, lat, model_kwargs, prediction_type):
        return StaticDenoiser(owner, model_kwargs["input_ids"], model_kwargs["position_ids"],
                              model_kwargs["attention_mask"], lat, model_kwargs["input_image_sizes"],
                              model_kwargs["denoise_image_sizes"], model_kwargs["time_emb_inx"], len(z),
                              tuple(z[0].shape[-2:]), model_kwargs["use_img_cfg"], model_kwargs["img_cfg_scale"],
                              prediction_type, sigma=self.sigma, pack_padding=self.pack_padding,
                              reuse_condition_prefix=self.reuse_condition_prefix,
                              hoist_special_rows=self.hoist_special_rows,
                              attention_precision=self.attention_precision, fuse_norms=self.fuse_norms)

    def __call__(self, z, func, model_kwargs, use_kv_cache: bool = True, offload_kv_cache: bool = True,
                 prediction_type: str = "v", vae=None, noise_level=None):
        is_list = isinstance(z, (list, tuple))
        frames = list(z) if is_list else [z]
        if not frames[0].is_cuda:
            raise VgptError("LVMScheduler runs on the MI355X HIP path only (latents must be on the GPU)")
        out_dtype = frames[0].dtype
        if noise_level is not None:  # LVM/scheduler.py:162-163 (RNG stays torch's)
            from . import ops_train   # noise_level * f + (1 - noise_level) * randn through the HIP lerp kernel
            mixed = []
            for f in frames:
                t = torch.full((f.shape[0],), float(noise_level), device=f.device, dtype=torch.float32)
                o = torch.empty(f.shape, device=f.device, dtype=torch.bfloat16)
                mixed.append(ops_train.lerp_frames(f.float().contiguous(), torch.randn_like(f).float().contiguous(), t, o)
                             .to(out_dtype))
            frames = mixed

        engine = self._fast_path_engine(frames, func, model_kwargs, prediction_type) if is_list else None
        if engine is not None:
            self.last_engine = engine
            stream = torch.cuda.current_stream()
            side = None
            if self.use_graph and stream.cuda_stream == 0:  # the legacy default stream cannot be captured
                side = torch.cuda.Stream()
                side.wait_stream(stream)
            with torch.cuda.stream(side) if side is not None else _null():
                engine.set_latents(torch.cat(frames, dim=0))
                zf = engine.run(self.num_steps, use_graph=self.use_graph)
            if side is not None:
                stream.wait_stream(side)
            zf = zf.view(len(frames), *frames[0].shape[1:]).to(out_dtype)
            return [zf[i:i + 1] for i in range(len(frames))]

        # ---- generic path ----
        # the state is a (rows, elems) fp32 matrix: one row per list entry, or per batch item of a tensor z
        # (LVM/scheduler.py:183-204 treats both the same way: x1 -> v, CFG on the two halves, Euler update)
        dev = frames[0].device
        if is_list:
            shapes = [tuple(f.shape) for f in frames]
            if len({f.numel() for f in frames}) != 1:
                raise VgptError("generic sampler path needs latents of one size (mixed resolutions: call per size)")
            n, elems = len(frames), frames[0].numel()
            zf = torch.cat([f.reshape(1, -1) for f in frames], dim=0).to(torch.float32).contiguous()
        else:
            zt = frames[0]
            n, elems = zt.shape[0], zt.numel() // max(zt.shape[0], 1)
            zf = zt.reshape(n, elems).to(torch.float32).contiguous()
        zm = torch.empty(n, elems, dtype=BF16, device=dev)
        ops.cast_f32_to_bf16(zf, zm)
        sigma = self.sigma.to(dev, torch.float32).contiguous()
        step = torch.zeros(1, dtype=torch.int32, device=dev)
        ts = torch.empty(n, dtype=torch.float32, device=dev)
        use_cfg = bool(model_kwargs.get("use_img_cfg", False))
        scale = float(model_kwargs.get("img_cfg_scale", 1.0))
        # for 'v' predictions the CFG combination already happened inside func (LVM/model.py:508-512, 555-562)
        kernel_cfg = use_cfg and prediction_type == "x1"
        if kernel_cfg and n % 2:
            raise VgptError("image CFG needs the conditional and unconditional halves (an even number of latents)")
        for _ in range(self.num_steps):
            ops.sampler_set_timesteps(sigma, step, ts)
            z_in = [zm[i].view(shapes[i]) for i in range(n)] if is_list else zm.view(frames[0].shape)
            pred, _cache = func(z_in, ts, past_key_values=None, prediction_type=prediction_type, **model_kwargs)
            if is_list:
                pred = torch.cat([p.reshape(1, -1) for p in pred], dim=0)
            pred = pred.reshape(n, elems).to(BF16).contiguous()
            ops.euler_cfg_update(zf, zm, pred, sigma, step, ops.PRED_X1 if prediction_type == "x1" else ops.PRED_V,
                                 kernel_cfg, scale)
            ops.sampler_advance(step)
        out = zf.to(out_dtype)
        if is_list:
            return [out[i].view(shapes[i]) for i in range(n)]
        return out.view(frames[0].shape)


class _null:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False
