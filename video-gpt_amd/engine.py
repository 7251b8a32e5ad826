"""StaticDenoiser: one next-clip denoise step as a fixed, allocation-free launch sequence, and the
Euler sampling loop over it replayed from a hipGraph.

This is the execution plan behind `LVMScheduler.__call__` when it drives
`LVM.frame_block_forward_with_cfg` (LVM/scheduler.py:161-208 calling LVM/model.py:519-566 once per
step with `past_key_values=None`).  All buffers are allocated once per clip; a step is
  set_timesteps -> sequence assembly (embedding gather, condition patch-embed, time tokens,
  noisy patch-embed) -> 32 x [rmsnorm, qkv GEMM, RoPE, block-masked attention, o_proj GEMM +
  residual, rmsnorm, gate_up GEMM + act*up, down GEMM + residual] -> final norm -> t_embedder /
  adaLN -> final layer + unpatchify -> Euler / x1->v / CFG update -> step counter += 1,
every launch reading the step index and sigma table from device memory, so the captured graph is
identical for every step.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import weakref

import torch

from . import ops
from .layout import TokenLayout
from .ops import BF16, VgptError


def _rows(sizes: Dict[int, list], row_of, span: bool):
    out = []
    for b in sizes.keys():
        for item in sizes[b]:
            out.append(row_of(b, item[0] if span else item))
    return out


def count_left_pads(attention_mask) -> List[int]:
    """Left-pad length of every row, read off the mask itself: pad rows are the leading all-ones rows
    (LVM/processor.py:726-727); a real first token only sees itself."""
    if isinstance(attention_mask, TokenLayout):
        return attention_mask.left_pads()
    B, L, _ = attention_mask.shape
    if L <= 1:
        return [0] * B
    mb = attention_mask.to(torch.bool)
    lead = [int(v) for v in torch.cumprod(mb.all(-1).to(torch.int64), dim=1).sum(1).tolist()]
    # leading all-ones rows are padding only in the collator's pattern, where the first real row does not see the pad
    # columns (LVM/processor.py:722-727).  A mask whose first rows simply see everything (a caller's full bidirectional
    # mask) has no padding: packing it would drop real tokens.
    for b, n in enumerate(lead):
        if n and (n >= L or bool(mb[b, n, :n].any())):
            lead[b] = 0
    return lead


def pack_left_padded(input_ids, position_ids, attention_mask, pads: List[int]):
    """Drop the left-pad tokens of every batch row and lay the real tokens of all rows out as ONE
    sequence with a block-diagonal mask (row b's real-token sub-mask on the diagonal, everything
    between different rows masked).  Pad rows never influence real rows (pad columns are masked,
    LVM/processor.py:722-727) and the model only reads real positions, so the outputs are unchanged;
    the attention kernel skips the off-diagonal tiles through its tile summary.
    Returns (ids (1,M), positions (1,M), mask (1,M,M) bool, offsets, pads)."""
    B, L = input_ids.shape
    lens = [L - p for p in pads]
    offsets = [0]
    for n in lens[:-1]:
        offsets.append(offsets[-1] + n)
    M = sum(lens)
    ids = torch.cat([input_ids[b, pads[b]:] for b in range(B)]).view(1, M)
    pos = torch.cat([position_ids[b, pads[b]:] for b in range(B)]).view(1, M)
    if isinstance(attention_mask, TokenLayout):   # tokens keep their sequence id: block-diagonal by construction
        return ids.contiguous(), pos.contiguous(), attention_mask.pack(pads)[0], offsets
    mask = torch.zeros(1, M, M, dtype=torch.bool, device=attention_mask.device)
    for b in range(B):
        o, n, p = offsets[b], lens[b], pads[b]
        mask[0, o:o + n, o:o + n] = attention_mask[b, p:, p:].to(torch.bool)
    return ids.contiguous(), pos.contiguous(), mask, offsets


_FOLDED = weakref.WeakKeyDictionary()   # model -> (key, qkv weights, gate_up weights) with the RMSNorm gains folded in


def folded_weights(model):
    """Per decoder layer: qkv_proj.weight * input_layernorm.weight and gate_up_proj.weight * post_attention_layernorm.weight
    (per input column, rounded to bf16 once: ops.fold_norm_gain) for the per-step forward with folded RMSNorms.  Derived
    copies (5 GB at Phi-3-mini size), built once per model and rebuilt when a parameter involved was written or replaced."""
    layers = model.llm.layers
    ps = [p_ for l in layers for p_ in (l.self_attn.qkv_proj.weight, l.input_layernorm.weight, l.mlp.gate_up_proj.weight,
                                        l.post_attention_layernorm.weight)]
    key = tuple((p_.data_ptr(), p_._version) for p_ in ps)
    hit = _FOLDED.get(model)
    if hit is not None and hit[0] == key:
        return hit[1], hit[2]
    wq = [ops.fold_norm_gain(l.self_attn.qkv_proj.weight, l.input_layernorm.weight) for l in layers]
    wgu = [ops.fold_norm_gain(l.mlp.gate_up_proj.weight, l.post_attention_layernorm.weight) for l in layers]
    _FOLDED[model] = (key, wq, wgu)
    return wq, wgu


class StaticDenoiser:
    def __init__(self, model, input_ids, position_ids, attention_mask, input_img_latents, input_image_sizes,
                 denoise_image_sizes, time_emb_inx, n_frames: int, latent_hw, use_img_cfg: bool, img_cfg_scale: float,
                 prediction_type: str = "v", sigma: Optional[torch.Tensor] = None, pack_padding: bool = True,
                 reuse_condition_prefix: bool = False, hoist_special_rows: bool = True,
                 attention_precision: str = "bf16", fuse_norms: Optional[bool] = None):
        model._check_ready()
        # fuse_norms: the two RMSNorms of a decoder layer folded into the GEMMs around them in the per-step forward (ops:
        # linear_resid_ssq -> *_prenorm; include/vgpt.h).  None = on wherever the step's shapes allow it, VGPT_FUSE_NORMS=0
        # switches it off (same-box A/B); the per-clip passes and the generic model path keep the separate kernel.
        if attention_precision not in ("bf16", "fp8"):
            raise VgptError(f"StaticDenoiser: attention_precision must be 'bf16' or 'fp8' (got {attention_precision!r})")
        # "fp8": the per-step attention of the sampler runs on MX-fp8 operands (csrc/attn_fp8.hip; the cfg-5 option of
        # SURVEY.md §8d).  The per-clip passes (prefill, time rows) stay bf16.
        self.attn_fp8 = attention_precision == "fp8"
        # query rows per work item of the per-step bf16 attention: 128 = the four-wave kernel, two workgroups per CU (product).
        # 256 = the eight-wave kernel (one K / V tile staged per 256 rows: half the LDS-DMA instructions per wave and half the
        # L2 -> LDS bytes per FLOP; head dim 96): bit-identical results, measured SLOWER in round 3 -- 159.5 / 160.4 us per
        # layer against 148.3 us at the cfg-2 live rows, same box (a barrier across eight waves per tile costs more than the
        # staging it saves) -- and kept behind VGPT_ATTN_ITEM_ROWS=256 for A/B runs.
        import os
        self.attn_item_rows = int(os.environ.get("VGPT_ATTN_ITEM_ROWS", "128"))
        self.model = model
        cfg = model.llm.config
        self.cfg = cfg
        dev = input_ids.device
        self.dev = dev
        B, L = input_ids.shape
        row_of = lambda b, s: b * L + s
        self.packed = False
        seq_bounds = None   # packed layout: row range of every original batch row, for the attention plan
        pads = count_left_pads(attention_mask) if pack_padding and not isinstance(attention_mask, ops.PackedMask) else []
        if any(pads):
            input_ids, position_ids, attention_mask, offs = pack_left_padded(input_ids, position_ids, attention_mask, pads)
            row_of = lambda b, s: offs[b] + s - pads[b]
            seq_bounds = [(offs[b], offs[b] + (L - pads[b])) for b in range(B)]
            B, L = input_ids.shape
            self.packed = True
        # ---- condition-prefix reuse: rows before the first <|diffusion|> token never see a
        #      noisy / time token (LVM/processor.py:682-731), so their activations are identical at every denoise
        #      step; the reference recomputes them 50x (LVM/scheduler.py:174).  They are computed ONCE (prefill),
        #      their per-layer K/V stay in a full-length qkv buffer, and a step only runs the remaining rows. ----
        self.S = 0          # static prefix length in the (padded) layout == first computed row, multiple of 128
        self.S0 = 0         # rows the prefill computes (== S without hoisting: the alignment rows ride along)
        self.hoist = None   # special-row hoisting (below): dict(nf, steps tensors ...) when active
        hoist_seg = None
        if reuse_condition_prefix and B == 1 and not isinstance(attention_mask, ops.PackedMask):
            t_first = min(row_of(b, t) for b in time_emb_inx.keys() for t in time_emb_inx[b]) - 1
            is_layout = isinstance(attention_mask, TokenLayout)
            m2 = None if is_layout else attention_mask[0].to(torch.bool)
            static = t_first >= 128 and (attention_mask.prefix_is_static(t_first) if is_layout
                                         else not bool(m2[:t_first, t_first:].any()))
            plan = None
            if static and hoist_special_rows and is_layout:
                plan = self._hoist_plan(attention_mask, t_first, L, row_of, denoise_image_sizes, time_emb_inx)
            if plan is not None:
                # ---- special-row hoisting: the `<|diffusion|>` row of a noisy frame sees only `<|diffusion|>` columns
                #      and the condition prefix, its time row only those and the time columns — never an image column
                #      (LVM/processor.py:682-731) — so the first is step-invariant and the second a function of the step
                #      index alone.  Both leave the per-step row set: the sequence is re-ordered to
                #      [prefix | diffusion rows | time rows | gap | image rows], the image rows (n_frames x N: whole
                #      GEMM / attention tiles) are all a step computes, and the time rows' q/k/v of EVERY step come
                #      from the per-clip pass (_clip_pass) and are dropped in by vgpt_sampler_copy_step_rows. ----
                perm, S, inv = plan["perm"], plan["S"], plan["inv"]
                pt = torch.tensor([p_ if p_ >= 0 else 0 for p_ in perm], dtype=torch.int64, device=input_ids.device)
                gapm = torch.tensor([p_ < 0 for p_ in perm], dtype=torch.bool, device=input_ids.device)
                input_ids = torch.where(gapm, input_ids[0, 0], input_ids[0, pt]).view(1, -1)
                position_ids = torch.where(gapm, torch.zeros_like(position_ids[0, pt]), position_ids[0, pt]).view(1, -1)
                attention_mask = attention_mask.permute(perm)
                prev = row_of
                row_of = lambda b, s, prev=prev, inv=inv: inv[prev(b, s)]
                L = len(perm)
                self.S, self.S0 = S, t_first
                self.hoist = plan
                hoist_seg = plan["segments"]
                seq_bounds = None
            elif static:
                S0 = t_first
                S = (S0 + 127) // 128 * 128
                npad = S - S0
                dev0 = input_ids.device
                pad_ids = torch.full((1, npad), int(input_ids[0, 0]), dtype=input_ids.dtype, device=dev0)
                input_ids = torch.cat([input_ids[:, :S0], pad_ids, input_ids[:, S0:]], dim=1)
                position_ids = torch.cat([position_ids[:, :S0], torch.zeros(1, npad, dtype=position_ids.dtype, device=dev0),
                                          position_ids[:, S0:]], dim=1)
                L2 = L + npad
                if is_layout:
                    attention_mask = attention_mask.insert_gap(S0, npad)
                else:
                    mask2 = torch.zeros(1, L2, L2, dtype=torch.bool, device=m2.device)
                    mask2[0, :S0, :S0] = m2[:S0, :S0]
                    mask2[0, S:, :S0] = m2[S0:, :S0]
                    mask2[0, S:, S:] = m2[S0:, S0:]
                    attention_mask = mask2
                prev = row_of
                row_of = lambda b, s, prev=prev, S0=S0, npad=npad: (lambda r: r if r < S0 else r + npad)(prev(b, s))
                L = L2
                self.S = self.S0 = S
                if seq_bounds is not None:
                    sh = lambda r_: r_ if r_ < S0 else r_ + npad
                    seq_bounds = [(sh(a_), sh(e_ - 1) + 1) for a_, e_ in seq_bounds]
        H, I = cfg.hidden_size, cfg.intermediate_size
        self.B, self.L, self.H = B, L, H
        # query-row segments of the attention plan: cut where two packed sequences meet (their key sets differ) so
        # that no work item of the kernel straddles them; `seg_live` covers the rows computed at every step
        self.seg_all = self.seg_live = None
        if seq_bounds is not None and B == 1:
            cuts = sorted({0, L, *[a_ for a_, _ in seq_bounds]})
            self.seg_all = tuple((0, a_, e_) for a_, e_ in zip(cuts[:-1], cuts[1:]) if e_ > a_)
            if self.S:
                cl = sorted({self.S, L, *[a_ for a_, _ in seq_bounds if a_ > self.S]})
                self.seg_live = tuple((0, a_, e_) for a_, e_ in zip(cl[:-1], cl[1:]) if e_ > a_)
        if hoist_seg is not None:
            self.seg_live = hoist_seg
        self.nf = n_frames
        self.h, self.w = latent_hw
        C = model.in_channels
        self.use_cfg = bool(use_img_cfg)
        self.cfg_scale = float(img_cfg_scale)
        if prediction_type not in ("v", "x1"):
            raise VgptError(f"unknown prediction_type {prediction_type!r}")
        self.pred_type = ops.PRED_X1 if prediction_type == "x1" else ops.PRED_V
        if self.use_cfg and n_frames % 2:
            raise VgptError("CFG needs an even number of latent frames")

        # static inputs
        self.input_ids = input_ids.contiguous()
        self.pm = ops.as_packed_mask(attention_mask, dev)
        self.layout = attention_mask if isinstance(attention_mask, TokenLayout) else None
        self.position_ids = position_ids
        self.rope = model.llm.rope_tables(position_ids)
        i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=dev)
        self.cond = None
        if input_img_latents is not None and len(input_img_latents) > 0:
            shapes = {tuple(t.shape[-2:]) for t in input_img_latents}
            if len(shapes) != 1:
                raise VgptError("StaticDenoiser needs condition frames of one resolution")
            self.cond = torch.cat([t.to(BF16) for t in input_img_latents], dim=0).contiguous()
            rows = _rows(input_image_sizes, row_of, True)
            if len(rows) != self.cond.shape[0]:
                raise AssertionError("input_image_sizes and input_img_latents disagree")
            self.cond_rows = i32(rows)
        x_rows = _rows(denoise_image_sizes, row_of, True)
        t_rows = _rows(time_emb_inx, row_of, False)
        if len(x_rows) != n_frames or len(t_rows) != n_frames:
            raise AssertionError("denoise_image_sizes / time_emb_inx disagree with the number of latents")
        self.x_rows, self.t_rows = i32(x_rows), i32(t_rows)
        S = self.S
        self.Ma = B * L - S   # rows computed per step
        if S:
            self.x_rows_a = i32([r - S for r in x_rows])
            self.t_rows_a = None if self.hoist else i32([r - S for r in t_rows])
            self.ids_a = self.input_ids[:, S:].contiguous()
            self.rope_a = (self.rope[0][S:].contiguous(), self.rope[1][S:].contiguous())

        # per-step state
        M = B * L
        e = lambda *s, dt=BF16: torch.empty(*s, dtype=dt, device=dev)
        self.z = e(n_frames, C * self.h * self.w, dt=torch.float32)
        self.z_model = e(n_frames, C, self.h, self.w)
        self.pred = e(n_frames, C, self.h, self.w)
        self.ts = e(n_frames, dt=torch.float32)
        self.step = torch.zeros(1, dtype=torch.int32, device=dev)
        self.sigma = None
        if sigma is not None:
            self.set_sigma(sigma)
        # workspaces
        nq, nk, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        if S:
            Ma = self.Ma
            self.hid, self.nrm, self.ctx, self.act = e(1, Ma, H), e(1, Ma, H), e(1, Ma, nq * hd), e(1, Ma, I)
            self.qkv_full = torch.zeros(cfg.num_hidden_layers, L, (nq + 2 * nk) * hd, dtype=BF16, device=dev)
            if self.attn_fp8:
                # one fp8 workspace per layer (51 MB at cfg-2): prefill() quantises the step-invariant rows once, a step
                # only the rows from the first one it writes on (the time rows of a hoisted layout sit below S)
                self.fp8_ws = [ops.attention_fp8_workspace(1, L, nq, nk, hd, dev) for _ in range(cfg.num_hidden_layers)]
                first = (self.S0 + self.hoist["nf"]) if self.hoist else S
                self.fp8_from = first // 64 * 64
        else:
            self.hid = e(B, L, H)
            self.nrm = e(B, L, H)
            self.qkv = e(B, L, (nq + 2 * nk) * hd)
            self.ctx = e(B, L, nq * hd)
            self.act = e(B, L, I)
        self.fuse = None
        if fuse_norms is None:
            fuse_norms = os.environ.get("VGPT_FUSE_NORMS", "1") != "0"
        if fuse_norms:
            Ms = self.Ma if S else B * L
            wa, wb = ops.norm_workspace_bytes(Ms, H, nq * hd), ops.norm_workspace_bytes(Ms, H, I)
            if wa > 0 and wb > 0:
                wq, wgu = folded_weights(model)
                # rstd_in / rstd_post: 1 / rms of the stream in front of a layer's input / post-attention norm
                self.fuse = {"rstd_in": torch.empty(Ms, dtype=torch.float32, device=dev),
                             "rstd_post": torch.empty(Ms, dtype=torch.float32, device=dev),
                             "ws": ops.norm_workspace(max(wa, wb), dev), "wq": wq, "wgu": wgu}
        self.temb_sin = e(n_frames, 256)
        self.tt_h = e(n_frames, H)
        self.te_h = e(n_frames, H)
        self.temb = e(n_frames, H)
        self.mod = e(n_frames, 2 * H)
        self.graph = None
        self.time_qkv = None   # hoisting: (steps, layers, n_frames, 3H) q/k/v rows of the time tokens of every step
        self.mod_all = None    # (steps, 1, n_frames, 2H) adaLN shift / scale of the final layer for every step (_mod_pass)
        self.steps_taken = 0
        if self.sigma is not None:
            self._mod_pass()
        if S:
            if self.hoist and self.sigma is not None:
                self._clip_pass()
            else:
                self.prefill()

    def prefill(self):
        """One forward over the static prefix rows [0, S) ONLY -- they never see a later row (that is what makes them
        step-invariant), so nothing else is needed to produce them -- leaving every layer's (post-RoPE) q/k/v in
        qkv_full[l][:S].  Rows >= S of qkv_full are written by every step (zero until the first one: the buffer is
        zero-initialised so that masked keys are finite)."""
        m, cfg, H = self.model, self.cfg, self.H
        # with hoisting the step-invariant `<|diffusion|>` rows (right behind the prefix) are computed here as well: they
        # see the prefix and each other, nothing else
        S = self.S0 + (self.hoist["nf"] if self.hoist else 0)
        nq, nk, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        e = lambda *s: torch.empty(*s, dtype=BF16, device=self.dev)
        hid, nrm, ctx, act = e(1, S, H), e(1, S, H), e(1, S, nq * hd), e(1, S, cfg.intermediate_size)
        ops.embed_gather(self.input_ids[:, :S].contiguous(), m.llm.embed_tokens.weight, out=hid)
        if self.cond is not None:
            ops.patch_embed(self.cond, m.input_x_embedder.proj.weight, m.input_x_embedder.proj.bias, m.pos_embed[0],
                            self.cond_rows, hid.view(-1, H), m.pos_embed_max_size)
        rope = (self.rope[0][:S].contiguous(), self.rope[1][:S].contiguous())
        seg = ((0, 0, S),)   # includes the pad rows S0..S: no visible key -> zeros, which keeps their K/V finite
        for li, layer in enumerate(m.llm.layers):
            at, mlp = layer.self_attn, layer.mlp
            full = self.qkv_full[li]
            ops.rmsnorm(hid, layer.input_layernorm.weight, layer.input_layernorm.variance_epsilon, out=nrm)
            ops.linear_qkv_rope(nrm, at.qkv_proj.weight, rope[0], rope[1], nq, nk, hd, out=full[:S])
            if self.attn_fp8:   # the sampler steps read the prefix's K / V from the fp8 workspace of this layer
                ops.attention_fp8_quantize(full.view(1, self.L, -1), self.fp8_ws[li], nq, nk, hd)
            ops.attention_qkv_range(full.view(1, self.L, -1), self.pm, nq, nk, hd, 0, ctx, segments=seg)
            ops.linear(ctx, at.o_proj.weight, residual=hid, out=hid)
            ops.rmsnorm(hid, layer.post_attention_layernorm.weight, layer.post_attention_layernorm.variance_epsilon, out=nrm)
            ops.gated_mlp_act(nrm, mlp.gate_up_proj.weight, mlp.act, out=act)
            ops.linear(act, mlp.down_proj.weight, residual=hid, out=hid)
        torch.cuda.current_stream().synchronize()

    def rebind(self, input_img_latents):
        """The NEXT clip on an engine built for an identical sequence (same ids, positions, mask, frame geometry, sigma table:
        the rounds of a rollout once the frame window is full): new condition latents in, everything that depends on them
        recomputed (the per-clip pass), every buffer, the attention plan and the captured graph kept.  Nothing else of the
        previous clip survives: the live rows of qkv_full and of the fp8 workspaces are rewritten by every step before they
        are read, the sampler state by set_latents()."""
        n_new = 0 if input_img_latents is None else len(input_img_latents)
        if (self.cond is None) != (n_new == 0):
            raise VgptError("StaticDenoiser.rebind: the clip has a different number of condition frames")
        if self.cond is not None:
            new = torch.cat([t.to(BF16) for t in input_img_latents], dim=0)
            if tuple(new.shape) != tuple(self.cond.shape):
                raise VgptError("StaticDenoiser.rebind: condition latents of another shape")
            self.cond.copy_(new)
        self.per_clip_setup()
        self.steps_taken = 0
        return self

    def per_clip_setup(self):
        """Everything a clip computes once instead of once per step: the condition prefix and the special rows of every step
        (one pass, _clip_pass; prefill() alone when the layout cannot be hoisted), the final layer's adaLN modulation of
        every step."""
        if self.S:
            if self.hoist:
                self._clip_pass()
            else:
                self.prefill()
        self._mod_pass()

    def set_sigma(self, sigma: torch.Tensor):
        self.sigma = sigma.to(self.dev, torch.float32).contiguous()
        self.num_steps = self.sigma.numel() - 1
        if getattr(self, "mod", None) is not None:
            self._mod_pass()
        if getattr(self, "hoist", None) and getattr(self, "qkv_full", None) is not None:
            self._clip_pass()

    def _mod_pass(self):
        """t_embedder MLP + adaLN modulation of the final layer for EVERY step in one pass per clip: they depend on
        sigma_i alone (every frame of a step carries the same t, LVM/scheduler.py:169), while the reference recomputes
        them inside every model call (LVM/model.py:480-486).  A step then copies its row (step index read on the device)."""
        m, dev, H, T = self.model, self.dev, self.H, self.num_steps
        e = lambda *s_: torch.empty(*s_, dtype=BF16, device=dev)
        sin, te_h, temb, mod = e(T, 256), e(T, H), e(T, H), e(T, 2 * H)
        ops.timestep_sinusoid(self.sigma[:T].contiguous(), m.t_embedder.freqs(dev), out=sin)
        te, ada = m.t_embedder.mlp, m.final_layer.adaLN_modulation[1]
        for c in range(0, T, 32):   # the small-M kernel takes at most 32 rows per call
            ops.linear_small(sin[c:c + 32], te[0].weight, te[0].bias, post_act=ops.ACT_SILU, out=te_h[c:c + 32])
            ops.linear_small(te_h[c:c + 32], te[2].weight, te[2].bias, out=temb[c:c + 32])
            ops.linear_small(temb[c:c + 32], ada.weight, ada.bias, pre_act=ops.ACT_SILU, out=mod[c:c + 32])
        shape = (T, 1, self.nf, 2 * H)
        if self.mod_all is None or tuple(self.mod_all.shape) != shape:
            self.mod_all = e(*shape)          # a captured graph reads this buffer: re-allocating invalidates it
            self.graph = None
        self.mod_all.copy_(mod.view(T, 1, 1, 2 * H).expand(*shape))

    # ---- special-row hoisting ------------------------------------------------------------------------------------
    @staticmethod
    def _hoist_plan(layout: TokenLayout, S0: int, L: int, row_of, denoise_image_sizes, time_emb_inx):
        """Re-ordering [prefix | diffusion rows | time rows | gap | image rows] of a packed next-clip sequence, or None
        when the rows behind the prefix are not exactly whole noisy frames (then only the prefix is reused)."""
        from .layout import NOISY
        x_old = _rows(denoise_image_sizes, row_of, True)
        t_old = _rows(time_emb_inx, row_of, False)
        ntoks = {it[1] - it[0] for b in denoise_image_sizes.keys() for it in denoise_image_sizes[b]}
        nf = len(x_old)
        if nf == 0 or len(t_old) != nf or len(ntoks) != 1:
            return None
        ntok = ntoks.pop()
        if any(t != x - 1 for t, x in zip(t_old, x_old)):
            return None
        d_old = [t - 1 for t in t_old]
        rows = sorted(d_old + t_old + [x + j for x in x_old for j in range(ntok)])
        if rows != list(range(S0, L)) or not bool((layout.kind[0, S0:] == NOISY).all()):
            return None
        # the special rows must really be the offset-0 / offset-1 tokens of their frames (what makes them image-blind)
        if not (bool((layout.oc[0, d_old] == 0).all()) and bool((layout.oc[0, t_old] == 1).all())
                and bool((layout.oc[0, [x + j for x in x_old for j in range(ntok)]] == 2).all())):
            return None
        S = (S0 + 2 * nf + 127) // 128 * 128
        perm = list(range(S0)) + d_old + t_old + [-1] * (S - S0 - 2 * nf) + [x + j for x in x_old for j in range(ntok)]
        inv = {o: i for i, o in enumerate(perm) if o >= 0}
        # image rows of one sequence form one segment of the attention plan
        seqs = [int(layout.seq[0, x]) for x in x_old]
        segs, f0 = [], 0
        for f in range(1, nf + 1):
            if f == nf or seqs[f] != seqs[f0]:
                segs.append((0, S + f0 * ntok, S + f * ntok))
                f0 = f
        return dict(perm=perm, inv=inv, S=S, nf=nf, ntok=ntok, segments=tuple(segs))

    def _clip_pass(self):
        """Everything of a hoisted clip that does not depend on the latents, in ONE forward per clip: the condition prefix
        and the `<|diffusion|>` rows (step-invariant; what prefill() computes) and the time rows of EVERY denoise step (a
        function of the step index alone).  The reference recomputes all of them inside every model call
        (LVM/model.py:435-454, LVM/scheduler.py:174).  Sequence of the pass:
            [prefix 0..S0) | nf `<|diffusion|>` rows | step 0's nf time rows | step 1's | ... | step T-1's]
        with the clip's own token attributes, the time rows of step s carrying sub-group s + 1 (layout.TokenLayout): they
        see the prefix, their clip's `<|diffusion|>` columns (sub-group 0) and the time columns of their own step only.
        One sequence means every layer's weights stream once per clip and the GEMMs run S0 + nf + T nf rows (1.9 k at
        cfg-2 with 53 steps) instead of two passes of about half that.  Leaves qkv_full[l][:S0 + nf] (post-RoPE q/k/v of
        the cached rows) and time_qkv[step, l]."""
        import numpy as np
        m, cfg, H, dev = self.model, self.cfg, self.H, self.dev
        nq, nk, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        W3 = (nq + 2 * nk) * hd
        S0, nf, T = self.S0, self.hoist["nf"], self.num_steps
        Sc = S0 + nf                                       # rows the sampler steps read from the cache
        Lp = Sc + T * nf
        idx = np.concatenate([np.arange(Sc), np.tile(np.arange(Sc, Sc + nf), T)])
        lp = self.layout.permute(idx)
        sub = lp.sub.copy()
        sub[0, Sc:] = 1 + np.repeat(np.arange(T), nf)
        pm = lp.with_subgroups(sub).packed_mask(dev)
        e = lambda *s_, dt=BF16: torch.empty(*s_, dtype=dt, device=dev)
        hid, nrm, ctx, act = e(1, Lp, H), e(1, Lp, H), e(1, Lp, nq * hd), e(1, Lp, cfg.intermediate_size)
        # rows [0, Sc): token embeddings + the condition frames' patch embeddings, as prefill()
        ops.embed_gather(self.input_ids[:, :Sc].contiguous(), m.llm.embed_tokens.weight, out=hid[:, :Sc])
        if self.cond is not None:
            ops.patch_embed(self.cond, m.input_x_embedder.proj.weight, m.input_x_embedder.proj.bias, m.pos_embed[0],
                            self.cond_rows, hid.view(-1, H), m.pos_embed_max_size)
        # time_token(sigma_s): one value per step (every frame of a step carries the same t, LVM/scheduler.py:169),
        # through the same small-M kernels as the per-step path (at most 32 rows per call), then broadcast to the rows
        ts = self.sigma[:T].contiguous()
        sin, tt_h, tt_o = e(T, 256), e(T, H), e(T, H)
        ops.timestep_sinusoid(ts, m.time_token.freqs(dev), out=sin)
        tt = m.time_token.mlp
        for c in range(0, T, 32):
            ops.linear_small(sin[c:c + 32], tt[0].weight, tt[0].bias, post_act=ops.ACT_SILU, out=tt_h[c:c + 32])
            ops.linear_small(tt_h[c:c + 32], tt[2].weight, tt[2].bias, out=tt_o[c:c + 32])
        hid[0, Sc:].view(T, nf, H)[:] = tt_o[:, None, :]
        # cos / sin are ROWS OF THE CLIP'S OWN TABLE (self.rope, built from the full position_ids): a su / longrope
        # checkpoint picks short or long factors from the largest position of the WHOLE sequence (HF 4.47.1
        # Phi3LongRoPEScaledRotaryEmbedding), which a table rebuilt from this pass's rows alone would get wrong whenever
        # only the image rows cross original_max_position_embeddings
        rope = tuple(torch.cat([t_[:Sc], t_[Sc:Sc + nf].repeat(T, 1)]).contiguous() for t_ in self.rope)
        buf = e(1, Lp, W3)
        shape = (T, cfg.num_hidden_layers, nf, W3)
        if self.time_qkv is None or tuple(self.time_qkv.shape) != shape:
            self.time_qkv = e(*shape)      # a captured graph reads this buffer: re-allocating invalidates it
            self.graph = None
        seg = ((0, 0, Lp),)
        for li, layer in enumerate(m.llm.layers):
            at, mlp = layer.self_attn, layer.mlp
            full = self.qkv_full[li]
            ops.rmsnorm(hid, layer.input_layernorm.weight, layer.input_layernorm.variance_epsilon, out=nrm)
            ops.linear_qkv_rope(nrm, at.qkv_proj.weight, rope[0], rope[1], nq, nk, hd, out=buf)
            full[:Sc].copy_(buf[0, :Sc])
            self.time_qkv[:, li].copy_(buf[0, Sc:].view(T, nf, W3))
            if self.attn_fp8:   # the sampler steps read the cached rows' K / V from the fp8 workspace of this layer
                ops.attention_fp8_quantize(full.view(1, self.L, -1), self.fp8_ws[li], nq, nk, hd)
            ops.attention_qkv_range(buf, pm, nq, nk, hd, 0, ctx, segments=seg)
            ops.linear(ctx, at.o_proj.weight, residual=hid, out=hid)
            ops.rmsnorm(hid, layer.post_attention_layernorm.weight, layer.post_attention_layernorm.variance_epsilon, out=nrm)
            ops.gated_mlp_act(nrm, mlp.gate_up_proj.weight, mlp.act, out=act)
            ops.linear(act, mlp.down_proj.weight, residual=hid, out=hid)
        self.time_dst = self.qkv_full[:, Sc:Sc + nf]    # (layers, nf, 3H) view the step copy writes
        torch.cuda.current_stream().synchronize()

    def set_latents(self, z: torch.Tensor):
        """z: (n_frames, C, h, w) any float dtype; becomes the fp32 sampler state."""
        self.z.copy_(z.reshape(self.nf, -1).to(torch.float32))
        ops.cast_f32_to_bf16(self.z, self.z_model)
        self.step.zero_()
        self.steps_taken = 0

    # ---- one denoise forward: z_model, ts -> pred ----
    def forward_step(self, from_tables: bool = False):
        """from_tables: the step is sigma[*step] of the table (sampler_step), so everything that depends on the step
        alone comes from the per-clip passes; otherwise `self.ts` may hold any timesteps."""
        m, cfg, H = self.model, self.cfg, self.H
        seq2d = self.hid.view(-1, H)
        pos = m.pos_embed[0]
        S = self.S
        x_rows, t_rows = (self.x_rows_a, self.t_rows_a) if S else (self.x_rows, self.t_rows)
        rope = self.rope_a if S else self.rope
        if not self.hoist:   # a hoisted step's live rows are image rows only: the patch embedding below writes every one
            ops.embed_gather(self.ids_a if S else self.input_ids, m.llm.embed_tokens.weight, out=self.hid)
        if self.cond is not None and not S:
            ops.patch_embed(self.cond, m.input_x_embedder.proj.weight, m.input_x_embedder.proj.bias, pos,
                            self.cond_rows, seq2d, m.pos_embed_max_size)
        from_tables = from_tables and self.mod_all is not None
        if not (from_tables and self.hoist):
            ops.timestep_sinusoid(self.ts, m.time_token.freqs(self.dev), out=self.temb_sin)
        if self.hoist:
            if self.time_qkv is None:
                raise VgptError("StaticDenoiser: set_sigma() must run before the first step")
            ops.sampler_copy_step_rows(self.time_qkv, self.time_dst, self.step)
        else:
            tt = m.time_token.mlp
            ops.linear_small(self.temb_sin, tt[0].weight, tt[0].bias, post_act=ops.ACT_SILU, out=self.tt_h)
            ops.linear_small(self.tt_h, tt[2].weight, tt[2].bias, out=seq2d, out_row=t_rows, ldo=H)
        ops.patch_embed(self.z_model, m.x_embedder.proj.weight, m.x_embedder.proj.bias, pos, x_rows, seq2d,
                        m.pos_embed_max_size)
        nq, nk, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        fz = self.fuse
        if fz is not None:
            # the statistics of the first norm: the embedded rows are no GEMM's output.  From here on every residual stream is
            # written by linear_resid_ssq, which leaves the next norm's partial sums of squares behind
            ops.rms_rstd(self.hid, m.llm.layers[0].input_layernorm.variance_epsilon, out=fz["rstd_in"])

        def qkv_proj(li_, layer, out):
            at = layer.self_attn
            if fz is None:
                return ops.linear_qkv_rope(self.nrm, at.qkv_proj.weight, rope[0], rope[1], nq, nk, hd, out=out)
            return ops.linear_qkv_rope_prenorm(self.hid, fz["wq"][li_], rope[0], rope[1], fz["rstd_in"], nq, nk, hd, out=out)
        for li_, layer in enumerate(m.llm.layers):
            at, mlp = layer.self_attn, layer.mlp
            if fz is None:
                ops.rmsnorm(self.hid, layer.input_layernorm.weight, layer.input_layernorm.variance_epsilon, out=self.nrm)
            if S:
                full = self.qkv_full[li_]
                live = full[S:]                                  # this step's q/k/v rows, behind the cached prefix
                qkv_proj(li_, layer, live)
                if self.attn_fp8:
                    # the prefix rows were quantised once by prefill(); a step re-quantises from the first row it writes
                    ops.attention_qkv_fp8(full.view(1, self.L, -1), self.pm, nq, nk, hd, out=self.ctx, q_start=S,
                                          segments=self.seg_live, workspace=self.fp8_ws[li_], quant_from=self.fp8_from)
                else:
                    ops.attention_qkv_range(full.view(1, self.L, -1), self.pm, nq, nk, hd, S, self.ctx, segments=self.seg_live,
                                            item_rows=self.attn_item_rows)
            else:
                qkv_proj(li_, layer, self.qkv)
                if self.attn_fp8:
                    ops.attention_qkv_fp8(self.qkv, self.pm, nq, nk, hd, out=self.ctx, segments=self.seg_all)
                elif self.seg_all is not None:
                    ops.attention_qkv_range(self.qkv, self.pm, nq, nk, hd, 0, self.ctx, segments=self.seg_all,
                                            item_rows=self.attn_item_rows)
                else:
                    ops.attention_qkv(self.qkv, self.pm, nq, nk, hd, out=self.ctx)
            if fz is not None:
                ops.linear_resid_rstd(self.ctx, at.o_proj.weight, self.hid, fz["rstd_post"], fz["ws"],
                                      layer.post_attention_layernorm.variance_epsilon, out=self.hid)
                ops.gated_mlp_act_prenorm(self.hid, fz["wgu"][li_], fz["rstd_post"], mlp.act, out=self.act)
                # the statistic the NEXT layer's input norm reads (its eps; the last layer's goes unused: the final norm is a
                # separate kernel)
                nxt = m.llm.layers[min(li_ + 1, len(m.llm.layers) - 1)].input_layernorm.variance_epsilon
                ops.linear_resid_rstd(self.act, mlp.down_proj.weight, self.hid, fz["rstd_in"], fz["ws"], nxt, out=self.hid)
                continue
            ops.linear(self.ctx, at.o_proj.weight, residual=self.hid, out=self.hid)
            ops.rmsnorm(self.hid, layer.post_attention_layernorm.weight,
                        layer.post_attention_layernorm.variance_epsilon, out=self.nrm)
            ops.gated_mlp_act(self.nrm, mlp.gate_up_proj.weight, mlp.act, out=self.act)
            ops.linear(self.act, mlp.down_proj.weight, residual=self.hid, out=self.hid)
        ops.rmsnorm(self.hid, m.llm.norm.weight, m.llm.norm.variance_epsilon, out=self.nrm)
        # t_embedder + adaLN modulation: every frame of a step carries the same t (LVM/scheduler.py:169), so one row is
        # computed and broadcast (the small-M kernel re-reads its input rows for every output column)
        if from_tables:
            ops.sampler_copy_step_rows(self.mod_all, self.mod.view(1, self.nf, 2 * H), self.step)
        else:
            te = m.t_embedder.mlp
            ops.linear_small(self.temb_sin[:1], te[0].weight, te[0].bias, post_act=ops.ACT_SILU, out=self.te_h[:1])
            ops.linear_small(self.te_h[:1], te[2].weight, te[2].bias, out=self.temb[:1])
            ada = m.final_layer.adaLN_modulation[1]
            ops.linear_small(self.temb[:1], ada.weight, ada.bias, pre_act=ops.ACT_SILU, out=self.mod[:1])
            if self.nf > 1:
                self.mod[1:].copy_(self.mod[:1].expand(self.nf - 1, -1))
        ops.final_layer(self.nrm.view(-1, H), x_rows, self.mod, m.final_layer.linear.weight,
                        m.final_layer.linear.bias, self.pred)

    def sampler_step(self):
        """LVM/scheduler.py:168-204 for one i: timesteps, model call, x1->v, CFG, Euler, i += 1."""
        ops.sampler_set_timesteps(self.sigma, self.step, self.ts)
        self.forward_step(from_tables=True)
        ops.euler_cfg_update(self.z, self.z_model, self.pred, self.sigma, self.step, self.pred_type, self.use_cfg,
                             self.cfg_scale)
        ops.sampler_advance(self.step)

    def capture(self):
        """Capture one sampler step into a hipGraph (must run on a non-default stream)."""
        # one eager step first: kernels set their attributes on first launch, which is not capturable
        saved = (self.z.clone(), self.z_model.clone(), self.step.clone())
        self.sampler_step()
        torch.cuda.current_stream().synchronize()
        self.z.copy_(saved[0]); self.z_model.copy_(saved[1]); self.step.copy_(saved[2])
        torch.cuda.current_stream().synchronize()
        self.graph = ops.HipGraph().capture(self.sampler_step)
        return self

    def run(self, num_steps: Optional[int] = None, use_graph: bool = True):
        n = self.num_steps if num_steps is None else num_steps
        if self.steps_taken + n > self.num_steps:
            raise VgptError(f"StaticDenoiser.run: {n} more steps after {self.steps_taken} exceed the {self.num_steps}-step "
                            "sigma table")
        self.steps_taken += n
        if use_graph and self.graph is None and n > 0:
            # the first step runs eagerly (kernels set their launch attributes on first use, which a capture cannot
            # record) and counts as a real step; the capture that follows records without executing
            self.sampler_step()
            torch.cuda.current_stream().synchronize()
            self.graph = ops.HipGraph().capture(self.sampler_step)
            n -= 1
        for _ in range(n):
            if use_graph:
                self.graph.replay()
            else:
                self.sampler_step()
        return self.z
