"""Stage-1 pre-training step on the HIP path: forward with saved activations, per-frame MSE on x1,
explicit backward, gradient all-reduce (RCCL) overlapped with the backward, global-norm clipping and AdamW.

Mirrors the reference's hot loop (LVM/train/train_x1_stage1_noiseinput.py:351-405) and loss
(LVM/train_helper/loss.py:128-243, `training_losses_x1_noise_input`):
    xt = t*x1 + (1-t)*x0 per frame, clean inputs noised with t_in in [input_noise, 1]; pred = model(xt, t, ...);
    loss_i = mean((x1_i - pred_i)^2); loss.mean().backward(); clip_grad_norm_(1.0); AdamW(lr, weight_decay).step()
The reference delegates backward to torch.autograd, the gradient reduction to DeepSpeed ZeRO-2 and the
optimizer to DeepSpeed's bf16 AdamW (fp32 master weights).  Here:
  * every backward pass is a HIP kernel (ops_train.py); the big dX / dW products reuse the MFMA NT GEMM
    through padded transposes, attention has its own dQ / dKdV kernels;
  * data parallelism replicates the model (288 GB HBM holds params + fp32 master + Adam moments, ~60 GB
    at Phi-3-mini size; no ZeRO sharding) and all-reduces one flat bf16 gradient bucket per decoder
    layer (226 MB at full size) as soon as that layer's backward has produced it, on RCCL's stream,
    while the next layer's backward runs; the small fp32 gradients go in one last bucket;
  * clipping uses the norm of the averaged gradient; the 1/world factor and the clip coefficient are
    folded into the AdamW kernel's gradient scale (no separate pass over the gradients).
"""
from __future__ import annotations

import os
import weakref

from typing import Dict, List, Optional

import torch
import torch.distributed as dist

from . import ops
from . import ops_train as T
from .engine import _rows, count_left_pads, pack_left_padded
from .ops import BF16, VgptError

F32 = torch.float32


_TRAINER_OF = weakref.WeakKeyDictionary()   # model -> weakref to the trainer whose optimizer updates it on a stream of its own


def wait_for_pending_update(model) -> None:
    """Readers of `model`'s parameters outside Stage1Trainer.step() -- the sampler (validation clips through LVMPipeline),
    state_dict() -- call this: with `overlap_optimizer` the last AdamW update may still be running on the trainer's own stream.
    Makes the CURRENT stream wait for it; nothing to do otherwise."""
    ref = _TRAINER_OF.get(model)
    tr = ref() if ref is not None else None
    if tr is not None:
        tr.finish_optimizer()


def _state_dict_barrier(module, prefix, keep_vars):
    wait_for_pending_update(module)


class Stage1Trainer:
    def __init__(self, model, lr: float = 1e-4, weight_decay: float = 0.0, betas=(0.9, 0.999), eps: float = 1e-8,
                 max_grad_norm: Optional[float] = 1.0, input_noise: float = 0.9, pack_padding: bool = True,
                 lr_scheduler: str = "constant", lr_warmup_steps: int = 0, gradient_checkpointing: Optional[bool] = None,
                 forward_only: bool = False, lr_scheduler_steps_per_optimizer_step: int = 1,
                 overlap_optimizer: bool = False):
        """lr_scheduler / lr_warmup_steps: diffusers' get_scheduler("constant" | "constant_with_warmup")
        (train_x1_stage1_noiseinput.py:279-283; the scripts use constant_with_warmup): the k-th optimizer step (k = 0, 1,
        ...) runs at lr * min(1, k * lr_scheduler_steps_per_optimizer_step / warmup).
        Stepping convention mirrored (default 1 = the reference's scripts: they pass --deepspeed_plugin
        (pretrain_stage1_nv.sh:49), so `accelerator.prepare` wraps the scheduler in accelerate's DeepSpeedSchedulerWrapper
        whose step() is a no-op and the DeepSpeed engine advances it ONCE per optimizer step whatever the world size;
        num_warmup_steps there is lr_warmup_steps * gradient_accumulation_steps, :279-283, and the scripts use
        accumulation 1).  Without the DeepSpeed plugin accelerate's AcceleratedScheduler advances the schedule
        `num_processes` times per optimizer step (split_batches=False): pass lr_scheduler_steps_per_optimizer_step =
        world size to mirror that launch instead.  No fixture pins the LR trajectory of the reference's loop (it needs
        deepspeed, absent here): parity of the warm-up length is unpinned.  gradient_checkpointing (default: model.llm.gradient_checkpointing, set by
        `model.llm.gradient_checkpointing_enable()`, train...py:170-171): keep only each decoder layer's input and
        recompute the layer inside the backward (OmniGen/transformer.py:182-192).  forward_only: no gradient / optimizer
        state (loss evaluation through `loss.training_losses_x1_noise_input`)."""
        model._check_ready()
        if hasattr(model, "release_engines"):
            model.release_engines()          # a sampler engine cached on the model holds GBs the trainer's buffers want
        self.model = model
        self.cfg = model.llm.config
        self.lr, self.wd, self.betas, self.eps = lr, weight_decay, betas, eps
        if lr_scheduler not in ("constant", "constant_with_warmup"):
            raise VgptError(f"lr_scheduler {lr_scheduler!r}: only 'constant' and 'constant_with_warmup' are built")
        self.lr_scheduler, self.lr_warmup_steps = lr_scheduler, int(lr_warmup_steps)
        if int(lr_scheduler_steps_per_optimizer_step) < 1:
            raise VgptError("lr_scheduler_steps_per_optimizer_step must be >= 1")
        self.lr_sched_stride = int(lr_scheduler_steps_per_optimizer_step)
        self.gradient_checkpointing = (bool(getattr(model.llm, "gradient_checkpointing", False))
                                       if gradient_checkpointing is None else bool(gradient_checkpointing))
        self.forward_only = forward_only
        self.max_grad_norm = max_grad_norm
        self.input_noise = input_noise
        self.pack_padding = pack_padding
        self.dev = model.llm.norm.weight.device
        # AdamW behind the clip coefficient on a stream of its own (see optimizer_step): events of the update of the small
        # bucket and of every layer bucket, waited for where the next forward first reads those parameters
        self.overlap_optimizer = bool(overlap_optimizer) and self.dev.type == "cuda" and not forward_only
        self._opt_stream = torch.cuda.Stream(device=self.dev) if self.overlap_optimizer else None
        if self.overlap_optimizer:
            # parameters are read outside step() too: the sampler asks wait_for_pending_update(model), state_dict() through
            # this hook
            _TRAINER_OF[model] = weakref.ref(self)
            model.register_state_dict_pre_hook(_state_dict_barrier)
        self._opt_events = None      # (small, [layer 0 .. nl-1]) of the update in flight
        self.step_count = 0
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.skip_allreduce = False   # measurement only (bench.py's exposed-communication leg): ranks stop agreeing when set
        # VGPT_DP_OVERLAP=0: all buckets are reduced behind the backward instead of layer by layer under it (for A/B runs on a
        # multi-GPU node: RCCL's kernels share the CUs with the backward's GEMMs while they overlap)
        self.overlap_allreduce = os.environ.get("VGPT_DP_OVERLAP", "1") != "0"
        self.params = {n: p for n, p in model.named_parameters()}
        self._ws = {}
        self.last = {}
        self.grads: Dict[str, torch.Tensor] = {}
        if forward_only:
            return
        L = self.cfg.num_hidden_layers
        # ---- gradient storage: one flat bf16 bucket per decoder layer + one fp32 bucket for the rest ----
        self.layer_names = [[f"llm.layers.{i}.self_attn.qkv_proj.weight", f"llm.layers.{i}.self_attn.o_proj.weight",
                             f"llm.layers.{i}.mlp.gate_up_proj.weight", f"llm.layers.{i}.mlp.down_proj.weight"]
                            for i in range(L)]
        self.layer_buckets = []
        for names in self.layer_names:
            n = sum(self.params[k].numel() for k in names)
            flat = torch.zeros(n, dtype=BF16, device=self.dev)
            o = 0
            for k in names:
                sz = self.params[k].numel()
                self.grads[k] = flat[o:o + sz].view(self.params[k].shape)
                o += sz
            self.layer_buckets.append(flat)
        big = {k for names in self.layer_names for k in names}
        small = [k for k in self.params if k not in big]
        n_small = sum(self.params[k].numel() for k in small)
        self.small_bucket = torch.zeros(n_small, dtype=F32, device=self.dev)
        o = 0
        for k in small:
            sz = self.params[k].numel()
            self.grads[k] = self.small_bucket[o:o + sz].view(self.params[k].shape)
            o += sz
        # ---- optimizer state (fp32 master + moments), flat per bucket so AdamW is one launch per bucket ----
        def flat_params(names):
            return torch.cat([self.params[k].detach().reshape(-1).to(F32) for k in names])
        self.master_layers = [flat_params(names) for names in self.layer_names]
        self.master_small = flat_params(small)
        self.small_names = small
        z = lambda t: torch.zeros_like(t)
        self.m_layers = [z(t) for t in self.master_layers]
        self.v_layers = [z(t) for t in self.master_layers]
        self.m_small, self.v_small = z(self.master_small), z(self.master_small)
        # model parameters become views of flat bf16 buffers so the optimizer writes them in one launch
        self.param_layers = []
        for names in self.layer_names:
            flat = torch.cat([self.params[k].detach().reshape(-1) for k in names]).contiguous()
            o = 0
            for k in names:
                sz = self.params[k].numel()
                self.params[k].data = flat[o:o + sz].view(self.params[k].shape)
                o += sz
            self.param_layers.append(flat)
        flat = torch.cat([self.params[k].detach().reshape(-1) for k in small]).contiguous()
        o = 0
        for k in small:
            sz = self.params[k].numel()
            self.params[k].data = flat[o:o + sz].view(self.params[k].shape)
            o += sz
        self.param_small = flat
        self.sumsq = torch.zeros(1, dtype=F32, device=self.dev)
        self.coef = torch.ones(1, dtype=F32, device=self.dev)

        self.grad_norm = torch.zeros(1, dtype=F32, device=self.dev)

    @classmethod
    def for_evaluation(cls, model):
        """A forward-only trainer cached on the model (no gradient buckets, no optimizer state)."""
        tr = getattr(model, "_vgpt_eval_trainer", None)
        if tr is None:
            tr = cls(model, forward_only=True)
            object.__setattr__(model, "_vgpt_eval_trainer", tr)
        return tr

    def current_lr(self) -> float:
        """Learning rate of the NEXT optimizer step (diffusers get_constant_schedule_with_warmup's lambda at
        current_step = optimizer steps taken so far)."""
        k = self.step_count * self.lr_sched_stride
        if self.lr_scheduler == "constant_with_warmup" and k < self.lr_warmup_steps:
            return self.lr * k / max(1.0, float(self.lr_warmup_steps))
        return self.lr

    # ------------------------------------------------------------------------------------------------
    def _buf(self, name, shape, dtype=BF16):
        t = self._ws.get(name)
        if t is None or t.shape != torch.Size(shape) or t.dtype != dtype:
            t = torch.empty(*shape, dtype=dtype, device=self.dev)
            self._ws[name] = t
        return t

    def _prepare(self, batch):
        cfg = self.cfg
        ids, pos, mask = batch["input_ids"], batch["position_ids"], batch["attention_mask"]
        B, L = ids.shape
        row_of = lambda b, s: b * L + s
        pads = count_left_pads(mask) if self.pack_padding else []
        if any(pads):
            ids, pos, mask, offs = pack_left_padded(ids, pos, mask, pads)
            row_of = lambda b, s: offs[b] + s - pads[b]
            B, L = ids.shape
        i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=self.dev)
        x_rows = _rows(batch["denoise_image_sizes"], row_of, True)
        t_rows = _rows(batch["time_emb_inx"], row_of, False)
        c_rows = _rows(batch["input_image_sizes"], row_of, True)
        ntok_x = batch_ntok(batch["denoise_image_sizes"])
        keep = torch.ones(B * L, dtype=torch.uint8)
        for r0 in x_rows + c_rows:
            keep[r0:r0 + ntok_x] = 0
        for r0 in t_rows:
            keep[r0] = 0
        c_idx = torch.tensor([r0 + k for r0 in c_rows for k in range(ntok_x)], dtype=torch.int64, device=self.dev) if c_rows else None
        return dict(ids=ids.contiguous(), B=B, L=L, pm=ops.as_packed_mask(mask, self.dev), rope=self.model.llm.rope_tables(pos),
                    x_rows=i32(x_rows), t_rows=i32(t_rows), c_rows=i32(c_rows) if c_rows else None, c_idx=c_idx,
                    keep=keep.to(self.dev), ntok=ntok_x)

    # ------------------------------------------------------------------------------------------------
    def step(self, batch, x1: torch.Tensor, x0: torch.Tensor, t: torch.Tensor, clean: Optional[torch.Tensor],
             x0_in: Optional[torch.Tensor], t_in: Optional[torch.Tensor], update: bool = True, backward: bool = True,
             input_output_return: bool = False):
        """One optimisation step.  x1/x0: (F, C, h, w) fp32 target latents / noise, t: (F,) fp32;
        clean/x0_in/t_in: the clean-frame latents and their noise (loss.py:166-192).  Returns the per-frame losses."""
        m, cfg = self.model, self.cfg
        prep = self._prepare(batch)
        pending = self._opt_events           # the previous step's update may still be running on its own stream
        if pending is not None:
            torch.cuda.current_stream().wait_event(pending[0])      # embeddings, heads, final norm: read from the start
        B, L, M, H, I = prep["B"], prep["L"], prep["B"] * prep["L"], cfg.hidden_size, cfg.intermediate_size
        nq, nk, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        nl = cfg.num_hidden_layers
        nf, C, h, w = x1.shape
        ntok = prep["ntok"]
        Tn = nf * ntok
        x1 = x1.to(self.dev, F32).contiguous(); x0 = x0.to(self.dev, F32).contiguous()
        t = t.to(self.dev, F32).contiguous()
        # ---------------- forward ----------------
        xt = T.lerp_frames(x1, x0, t, self._buf("xt", (nf, C, h, w)))
        cl = None
        if clean is not None and clean.shape[0] > 0:
            cl = T.lerp_frames(clean.to(self.dev, F32).contiguous(), x0_in.to(self.dev, F32).contiguous(),
                               t_in.to(self.dev, F32).contiguous(), self._buf("cl", tuple(clean.shape)))
        hbuf = self._buf("h", (nl + 1, M, H))          # layer inputs (h[l]) and the last output
        seq = hbuf[0]
        ops.embed_gather(prep["ids"], m.llm.embed_tokens.weight, out=seq.view(B, L, H))
        pos = m.pos_embed[0]
        if cl is not None:
            ops.patch_embed(cl, m.input_x_embedder.proj.weight, m.input_x_embedder.proj.bias, pos, prep["c_rows"], seq,
                            m.pos_embed_max_size)
        sin = ops.timestep_sinusoid(t, m.time_token.freqs(self.dev))
        tt, te, ada = m.time_token.mlp, m.t_embedder.mlp, m.final_layer.adaLN_modulation[1]
        tt_pre = ops.linear_small(sin, tt[0].weight, tt[0].bias)
        tt_act = T.act_fwd(tt_pre, ops.ACT_SILU)
        ops.linear_small(tt_act, tt[2].weight, tt[2].bias, out=seq, out_row=prep["t_rows"], ldo=H)
        ops.patch_embed(xt, m.x_embedder.proj.weight, m.x_embedder.proj.bias, pos, prep["x_rows"], seq,
                        m.pos_embed_max_size)
        # saved activations: every layer's (normal) or ONE layer's worth, recomputed per layer in the backward (checkpointing)
        ck = self.gradient_checkpointing
        ns = 1 if ck else nl
        n1 = self._buf("n1", (ns, M, H)); qkv = self._buf("qkv", (ns, M, (nq + 2 * nk) * hd))
        ctx = self._buf("ctx", (ns, M, nq * hd)); h2 = self._buf("h2", (ns, M, H)); n2 = self._buf("n2", (ns, M, H))
        gu = self._buf("gu", (ns, M, 2 * I)); act = self._buf("act", (ns, M, I))
        lse = self._buf("lse", (ns, B, nq, L), F32)
        sv = lambda li: 0 if ck else li

        def layer_forward(li, with_output=True):
            layer = m.llm.layers[li]
            at, mlp, k = layer.self_attn, layer.mlp, sv(li)
            if pending is not None:
                torch.cuda.current_stream().wait_event(pending[1][li])   # this layer's parameters are updated
            ops.rmsnorm(hbuf[li], layer.input_layernorm.weight, layer.input_layernorm.variance_epsilon, out=n1[k])
            ops.linear_qkv_rope(n1[k], at.qkv_proj.weight, prep["rope"][0], prep["rope"][1], nq, nk, hd, out=qkv[k])
            T.attention_qkv_train(qkv[k].view(B, L, -1), prep["pm"], nq, nk, hd, ctx[k].view(B, L, -1), lse[k])
            ops.linear(ctx[k], at.o_proj.weight, residual=hbuf[li], out=h2[k])
            ops.rmsnorm(h2[k], layer.post_attention_layernorm.weight, layer.post_attention_layernorm.variance_epsilon,
                        out=n2[k])
            # gate_up_proj + act(gate) * up in one kernel that also keeps the bf16 [gate | up] for the backward: bit for bit
            # ops.linear(n2, W) followed by T.silu_mul_fwd (tests/test_train_gpu.py), without the (M, 2I) round trip
            ops.gated_mlp_act(n2[k], mlp.gate_up_proj.weight, mlp.act, out=act[k], gate_up_out=gu[k])
            if with_output:
                ops.linear(act[k], mlp.down_proj.weight, residual=h2[k], out=hbuf[li + 1])

        for li in range(nl):
            layer_forward(li)
        nrm = ops.rmsnorm(hbuf[nl], m.llm.norm.weight, m.llm.norm.variance_epsilon, out=self._buf("nrm", (M, H)))
        te_pre = ops.linear_small(sin, te[0].weight, te[0].bias)
        te_act = T.act_fwd(te_pre, ops.ACT_SILU)
        temb = ops.linear_small(te_act, te[2].weight, te[2].bias)
        st = T.act_fwd(temb, ops.ACT_SILU)
        mod = ops.linear_small(st, ada.weight, ada.bias)
        v = self._buf("v", (Tn, H)); xhat = self._buf("xhat", (Tn, H), F32); rstd = self._buf("rstd", (Tn,), F32)
        T.ln_mod_fwd(nrm, prep["x_rows"], mod, v, xhat, rstd, ntok)
        fl = m.final_layer.linear
        y16 = ops.linear(v, fl.weight, bias=fl.bias)                       # (Tn, 16)
        p2 = m.patch_size
        pred = y16.view(nf, h // p2, w // p2, p2, p2, C).permute(0, 5, 1, 3, 2, 4).reshape(nf, C, h, w).contiguous()
        loss = torch.empty(nf, dtype=F32, device=self.dev)
        dpred = self._buf("dpred", (nf, C, h, w))
        head = None
        if input_output_return:
            # LVM/model.py:832-841 + loss.py:220-225: the input_final_layer head predicts the CLEAN condition latents from the
            # last hidden state of their rows; its per-frame MSE terms are appended to the loss vector before .mean()
            if cl is None:
                raise VgptError("Stage1Trainer.step: input_output_return needs condition frames")
            head = m.input_final_layer           # AttributeError without init_input_final_layer(), as in the reference
            nfc = clean.shape[0]
            vin = T.gather_rows(nrm, prep["c_rows"], ntok)                   # (nfc * ntok, H)
            yin = ops.linear(vin, head.weight, bias=head.bias)
            pred_in = yin.view(nfc, h // p2, w // p2, p2, p2, C).permute(0, 5, 1, 3, 2, 4).reshape(nfc, C, h, w).contiguous()
            loss_in = torch.empty(nfc, dtype=F32, device=self.dev)
            dpred_in = self._buf("dpred_in", (nfc, C, h, w))
            T.mse_frames(pred_in, clean.to(self.dev, F32).contiguous(), loss_in, dpred_in, n_mean=nf + nfc)
            T.mse_frames(pred, x1, loss, dpred, n_mean=nf + nfc)
            loss = torch.cat([loss, loss_in])
            self.last = dict(pred=pred, loss=loss, xt=xt, pred_in=pred_in)
        else:
            T.mse_frames(pred, x1, loss, dpred)
            self.last = dict(pred=pred, loss=loss, xt=xt)
        if self.forward_only or not backward:
            if update:
                raise VgptError("Stage1Trainer.step: an optimizer step needs the backward pass")
            return loss
        # ---------------- backward ----------------
        g = self.grads
        self.small_bucket.zero_()
        dy16 = T.unpatchify_bwd(dpred)                                      # (Tn, 16)
        T.matmul(dy16, v, out=g["final_layer.linear.weight"], ta=True)     # dWf = dy16^T v
        T.colsum(dy16, g["final_layer.linear.bias"])
        dv = T.matmul(dy16, fl.weight)                                      # (Tn, H)
        dnrm = self._buf("dnrm", (M, H)); dnrm.zero_()
        dmod = torch.zeros(nf, 2 * H, dtype=F32, device=self.dev)
        T.ln_mod_bwd(dv, xhat, rstd, mod, prep["x_rows"], dnrm, dmod, ntok)
        if head is not None:
            dyin = T.unpatchify_bwd(dpred_in)                                # (nfc * ntok, 16)
            T.matmul(dyin, vin, out=g["input_final_layer.weight"], ta=True)
            T.colsum(dyin, g["input_final_layer.bias"])
            dnrm.index_copy_(0, prep["c_idx"], T.matmul(dyin, head.weight))  # condition rows: no other gradient reaches them here
        dh = self._buf("dh", (M, H)); dh_b = self._buf("dh_b", (M, H))
        T.rmsnorm_bwd(hbuf[nl], m.llm.norm.weight, dnrm, dh, g["llm.norm.weight"], m.llm.norm.variance_epsilon)
        # adaLN + t_embedder
        T.matmul(dmod, st, out=g["final_layer.adaLN_modulation.1.weight"], ta=True)
        T.colsum(dmod, g["final_layer.adaLN_modulation.1.bias"])
        dtemb = T.act_bwd(temb, T.matmul(dmod, ada.weight), ops.ACT_SILU)
        self._mlp_bwd("t_embedder", te, dtemb, te_act, te_pre, sin)
        # decoder layers, last to first
        sw = sa = sb = None   # dX / dW read their operands transposed inside the GEMM (vgpt_gemm_bf16_tr)
        dact = self._buf("dact", (M, I)); dgu = self._buf("dgu", (M, 2 * I)); dn = self._buf("dn", (M, H))
        dctx = self._buf("dctx", (M, nq * hd)); dqkv = self._buf("dqkv", (M, (nq + 2 * nk) * hd))
        delta = self._buf("delta", (B, nq, L), F32)
        nsin = self._neg_sin(prep)
        handles = []
        for li in range(nl - 1, -1, -1):
            layer = m.llm.layers[li]
            at, mlp = layer.self_attn, layer.mlp
            names = self.layer_names[li]
            if ck:      # same kernels on the same inputs as the forward pass: the recomputed activations are bit-identical
                layer_forward(li, with_output=False)
            k = sv(li)
            T.linear_dw(dh, act[k], sa, sb, g[names[3]])                               # dW_down
            T.linear_dx(dh, mlp.down_proj.weight, sw, out=dact)
            T.silu_mul_bwd(gu[k], dact, dgu, mlp.act)
            T.linear_dw(dgu, n2[k], sa, sb, g[names[2]])                               # dW_gate_up
            T.linear_dx(dgu, mlp.gate_up_proj.weight, sw, out=dn)
            T.rmsnorm_bwd(h2[k], layer.post_attention_layernorm.weight, dn, dh_b,
                          g[f"llm.layers.{li}.post_attention_layernorm.weight"],
                          layer.post_attention_layernorm.variance_epsilon, dres=dh)       # dh2
            T.linear_dw(dh_b, ctx[k], sa, sb, g[names[1]])                             # dW_o
            T.linear_dx(dh_b, at.o_proj.weight, sw, out=dctx)
            T.attention_qkv_bwd(qkv[k].view(B, L, -1), ctx[k].view(B, L, -1), dctx.view(B, L, -1), lse[k], delta,
                                dqkv.view(B, L, -1), prep["pm"], nq, nk, hd)
            ops.rope_qk_inplace(dqkv, prep["rope"][0], nsin, nq, nk, hd)                 # inverse rotation
            T.linear_dw(dqkv, n1[k], sa, sb, g[names[0]])                              # dW_qkv
            T.linear_dx(dqkv, at.qkv_proj.weight, sw, out=dn)
            T.rmsnorm_bwd(hbuf[li], layer.input_layernorm.weight, dn, dh,
                          g[f"llm.layers.{li}.input_layernorm.weight"], layer.input_layernorm.variance_epsilon,
                          dres=dh_b)                                                     # dh (layer input)
            if self.world > 1 and not self.skip_allreduce and self.overlap_allreduce:
                handles.append(dist.all_reduce(self.layer_buckets[li], async_op=True))
        # heads fed by dseq = dh
        dseq = dh
        dtt = T.gather_rows(dseq, prep["t_rows"], 1)
        self._mlp_bwd("time_token", tt, dtt, tt_act, tt_pre, sin)
        self._patch_bwd("x_embedder", T.gather_rows(dseq, prep["x_rows"], ntok), xt)
        if cl is not None:
            self._patch_bwd("input_x_embedder", T.gather_rows(dseq, prep["c_rows"], ntok), cl)
        T.embed_bwd(prep["ids"].view(-1), prep["keep"], dseq, g["llm.embed_tokens.weight"])
        if self.world > 1 and not self.skip_allreduce:
            if not self.overlap_allreduce:
                handles += [dist.all_reduce(b, async_op=True) for b in reversed(self.layer_buckets)]
            handles.append(dist.all_reduce(self.small_bucket, async_op=True))
            for hd_ in handles:
                hd_.wait()
        if update:
            self.optimizer_step()
        return loss

    def _neg_sin(self, prep):
        key = ("nsin", prep["rope"][1].data_ptr())
        if self._ws.get("nsin_key") != key:
            self._ws["nsin"] = (-prep["rope"][1]).contiguous()   # sign flip of a table: data prep, once per layout
            self._ws["nsin_key"] = key
        return self._ws["nsin"]

    def _mlp_bwd(self, prefix, mlp, dout, act_saved, pre_saved, x_in):
        """2-layer MLP (Linear, SiLU, Linear) backward: LVM/model.py:32-36."""
        g = self.grads
        T.matmul(dout, act_saved, out=g[f"{prefix}.mlp.2.weight"], ta=True)
        T.colsum(dout, g[f"{prefix}.mlp.2.bias"])
        dpre = T.act_bwd(pre_saved, T.matmul(dout, mlp[2].weight), ops.ACT_SILU)
        T.matmul(dpre, x_in, out=g[f"{prefix}.mlp.0.weight"], ta=True)
        T.colsum(dpre, g[f"{prefix}.mlp.0.bias"])

    def _patch_bwd(self, prefix, dtok, latents):
        g = self.grads
        patches = T.patchify(latents)
        T.matmul(dtok, patches, out=g[f"{prefix}.proj.weight"].view(-1, 16), ta=True)
        T.colsum(dtok, g[f"{prefix}.proj.bias"])

    # ------------------------------------------------------------------------------------------------
    def optimizer_step(self):
        # an update still running on the optimizer's stream (overlap_optimizer) reads self.coef / self.sumsq and writes the
        # master weights and moments this call is about to touch: wait for it (free in the normal flow, where the forward of
        # the step that produced these gradients already waited for every layer's event)
        self.finish_optimizer()
        lr = self.current_lr()
        self.last_lr = lr
        self.step_count += 1
        self.sumsq.zero_()
        for b in self.layer_buckets:
            T.sumsq(b, self.sumsq)
        T.sumsq(self.small_bucket, self.sumsq)
        w = float(self.world)
        # norm of the AVERAGED gradient = norm(sum)/world; coefficient already carries the 1/world factor
        T.clip_coef(self.sumsq, self.coef, self.grad_norm, (self.max_grad_norm or 0.0) * w, 1.0 / w)
        b1, b2 = self.betas
        if not self.overlap_optimizer:
            for i in range(len(self.layer_buckets)):
                T.adamw_step(self.master_layers[i], self.param_layers[i], self.layer_buckets[i], self.m_layers[i],
                             self.v_layers[i], lr, b1, b2, self.eps, self.wd, self.step_count, self.coef)
            T.adamw_step(self.master_small, self.param_small, self.small_bucket, self.m_small, self.v_small, lr, b1, b2,
                         self.eps, self.wd, self.step_count, self.coef)
            return
        # The update is a pure HBM stream (28 bytes per parameter) and the next step's forward is matrix work: they run side by
        # side.  Everything below goes to the optimizer's stream behind the clip coefficient; the small bucket (embeddings,
        # heads: read first) and then the layers in forward order, each followed by an event the next forward waits for right
        # where it first reads that layer.  Readers outside step() call finish_optimizer() first.
        main = torch.cuda.current_stream()
        ready = torch.cuda.Event()
        ready.record(main)
        self._opt_stream.wait_event(ready)
        with torch.cuda.stream(self._opt_stream):
            T.adamw_step(self.master_small, self.param_small, self.small_bucket, self.m_small, self.v_small, lr, b1, b2,
                         self.eps, self.wd, self.step_count, self.coef)
            ev_small = torch.cuda.Event()
            ev_small.record(self._opt_stream)
            evs = []
            for i in range(len(self.layer_buckets)):
                T.adamw_step(self.master_layers[i], self.param_layers[i], self.layer_buckets[i], self.m_layers[i],
                             self.v_layers[i], lr, b1, b2, self.eps, self.wd, self.step_count, self.coef)
                e = torch.cuda.Event()
                e.record(self._opt_stream)
                evs.append(e)
        self._opt_events = (ev_small, evs)

    def finish_optimizer(self):
        """With overlap_optimizer: makes the current stream wait for the update in flight (call before reading parameters or
        optimizer state outside step(): checkpoints, evaluation, the end of a timed region)."""
        if self._opt_events is not None:
            torch.cuda.current_stream().wait_event(self._opt_events[1][-1])
            self._opt_events = None


    # ---- checkpoints (LVM/train/train_x1_stage1_noiseinput.py:304-334,437-451: accelerate's checkpoint-{step}
    #      directories with auto-resume from the highest step).  Written with safetensors, nothing is unpickled on
    #      load: model.safetensors holds the bf16 state_dict under the reference's keys (loadable by
    #      LVM.from_pretrained), optimizer.safetensors the fp32 master weights and Adam moments per bucket. ----
    def save_checkpoint(self, results_dir: str, global_step: Optional[int] = None) -> str:
        """Rank 0 writes (replicas are identical under data parallelism); every rank returns after the files exist."""
        self.finish_optimizer()
        import json
        import os
        from safetensors.torch import save_file
        step = self.step_count if global_step is None else int(global_step)
        path = os.path.join(results_dir, f"checkpoint-{step}")
        distributed = dist.is_available() and dist.is_initialized()
        if not distributed or dist.get_rank() == 0:
            os.makedirs(path, exist_ok=True)
            save_file({k: v.detach().cpu().contiguous() for k, v in self.model.state_dict().items()},
                      os.path.join(path, "model.safetensors"))
            opt = {"master_small": self.master_small, "m_small": self.m_small, "v_small": self.v_small}
            for i in range(len(self.master_layers)):
                opt[f"master.{i}"], opt[f"m.{i}"], opt[f"v.{i}"] = self.master_layers[i], self.m_layers[i], self.v_layers[i]
            save_file({k: v.detach().cpu().contiguous() for k, v in opt.items()}, os.path.join(path, "optimizer.safetensors"))
            with open(os.path.join(path, "trainer_state.json"), "w") as f:
                json.dump({"step_count": self.step_count, "global_step": step, "lr": self.lr, "weight_decay": self.wd,
                           "betas": list(self.betas), "eps": self.eps, "lr_scheduler": self.lr_scheduler,
                           "lr_warmup_steps": self.lr_warmup_steps, "small_names": self.small_names}, f)
        if distributed:
            dist.barrier()
        return path

    def load_checkpoint(self, path: str, restore_hyperparameters: bool = True) -> int:
        """Restores parameters, fp32 master weights, Adam moments, the step counter and (by default) the optimizer
        hyper-parameters and LR schedule; returns the global step.  Everything is validated before anything is copied.
        Only checkpoints written by this trainer resume (the reference's are accelerate / DeepSpeed `save_state`
        directories, whose optimizer shards are pickles: warm-start from those through LVM.from_pretrained's weight
        loaders instead)."""
        self.finish_optimizer()
        import json
        import os
        from safetensors.torch import load_file
        with open(os.path.join(path, "trainer_state.json")) as f:
            st = json.load(f)
        if st["small_names"] != self.small_names:
            raise VgptError("checkpoint was written for a different parameter layout")
        opt = load_file(os.path.join(path, "optimizer.safetensors"))
        pairs = [(self.master_small, "master_small"), (self.m_small, "m_small"), (self.v_small, "v_small")]
        for i in range(len(self.master_layers)):
            pairs += [(self.master_layers[i], f"master.{i}"), (self.m_layers[i], f"m.{i}"), (self.v_layers[i], f"v.{i}")]
        model_sd = load_file(os.path.join(path, "model.safetensors"))
        own = self.model.state_dict()
        problems = [f"optimizer tensor {k} missing" for _, k in pairs if k not in opt]
        problems += [f"optimizer tensor {k}: shape {tuple(opt[k].shape)} != {tuple(d.shape)}" for d, k in pairs
                     if k in opt and opt[k].shape != d.shape]
        problems += [f"model tensor {k} missing" for k in own if k not in model_sd]
        problems += [f"model tensor {k}: shape {tuple(model_sd[k].shape)} != {tuple(v.shape)}" for k, v in own.items()
                     if k in model_sd and model_sd[k].shape != v.shape]
        if problems:
            raise VgptError(f"{path}: checkpoint does not match this trainer: " + "; ".join(problems[:8]))
        for dst, key in pairs:
            dst.copy_(opt[key])
        with torch.no_grad():     # parameters are views of the flat bf16 buffers: copy in place, keep the views
            for k, p_ in own.items():
                p_.copy_(model_sd[k])
        self.step_count = int(st["step_count"])
        if restore_hyperparameters:
            self.lr, self.wd, self.eps = float(st["lr"]), float(st["weight_decay"]), float(st["eps"])
            self.betas = tuple(st["betas"])
            self.lr_scheduler = st.get("lr_scheduler", self.lr_scheduler)
            self.lr_warmup_steps = int(st.get("lr_warmup_steps", self.lr_warmup_steps))
        return int(st["global_step"])

    def auto_resume(self, results_dir: str) -> Optional[int]:
        """Load the checkpoint-{N} with the largest N under results_dir, if any (train...stage1.py:304-315)."""
        import glob
        import os
        found = [d for d in glob.glob(os.path.join(results_dir, "checkpoint-*")) if d.rsplit("-", 1)[-1].isdigit()]
        if not found:
            return None
        return self.load_checkpoint(max(found, key=lambda d: int(d.rsplit("-", 1)[-1])))


def batch_ntok(sizes) -> int:
    ns = {e - s for v in sizes.values() for s, e in v}
    if len(ns) != 1:
        raise VgptError("Stage1Trainer needs frames of one resolution")
    return ns.pop()
