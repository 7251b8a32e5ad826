"""Tensor-level wrappers of the training entry points of the C ABI (include/vgpt.h, "stage-1
pre-training step").  Same rules as ops.py: GPU tensors only, no torch arithmetic."""
from __future__ import annotations

import ctypes
import math
from typing import Optional

import torch

from . import _lib
from .ops import BF16, VgptError, _chk, _ptr, _stream, call, linear

F32 = torch.float32


def _f32flag(t: torch.Tensor) -> int:
    if t.dtype == F32:
        return 1
    if t.dtype == BF16:
        return 0
    raise VgptError(f"expected bf16 or fp32, got {t.dtype}")


def attention_qkv_train(qkv, pm, n_heads, n_kv_heads, head_dim, out, lse, scale=None):
    """Forward on the fused qkv buffer that also writes lse (B, n_heads, L) fp32 (base 2)."""
    _chk(qkv, BF16, "attn.qkv"); _chk(out, BF16, "attn.out"); _chk(lse, F32, "attn.lse")
    B, L, width = qkv.shape
    es = 2
    kq = qkv.data_ptr() + n_heads * head_dim * es
    vq = kq + n_kv_heads * head_dim * es
    sb, ss = L * width, width
    scale = 1.0 / math.sqrt(head_dim) if scale is None else scale
    strides = (sb, head_dim, ss) * 3 + (L * n_heads * head_dim, head_dim, n_heads * head_dim)
    from .ops import _attn_plan_call
    _attn_plan_call(qkv.data_ptr(), kq, vq, out.data_ptr(), lse.data_ptr(), pm, pm.plan(), B, L, n_heads, n_kv_heads,
                    head_dim, strides, scale)
    return out


def attention_qkv_bwd(qkv, out, dout, lse, delta_ws, dqkv, pm, n_heads, n_kv_heads, head_dim, scale=None):
    """dqkv (B, L, (n_q+2n_kv)*hd) <- gradients of q, k, v (post-RoPE) given dout (B, L, n_q*hd)."""
    for n, t in (("qkv", qkv), ("out", out), ("dout", dout), ("dqkv", dqkv)):
        _chk(t, BF16, f"attn_bwd.{n}")
    _chk(lse, F32, "attn_bwd.lse"); _chk(delta_ws, F32, "attn_bwd.delta")
    B, L, width = qkv.shape
    hq, hk = n_heads * head_dim, n_kv_heads * head_dim
    sb, ss = L * width, width
    ob, os_ = L * hq, hq
    st = [sb, head_dim, ss] * 3 + [ob, head_dim, os_] * 2 + [sb, head_dim, ss] * 3
    arr = (ctypes.c_int64 * 24)(*st)
    scale = 1.0 / math.sqrt(head_dim) if scale is None else scale
    q, k, v = qkv.data_ptr(), qkv.data_ptr() + hq * 2, qkv.data_ptr() + (hq + hk) * 2
    dq, dk, dv = dqkv.data_ptr(), dqkv.data_ptr() + hq * 2, dqkv.data_ptr() + (hq + hk) * 2
    call("vgpt_attn_blockmask_bwd", q, k, v, out.data_ptr(), dout.data_ptr(), lse.data_ptr(), delta_ws.data_ptr(), dq, dk,
         dv, pm.bits.data_ptr(), pm.summary.data_ptr(), B, L, n_heads, n_kv_heads, head_dim, arr, float(scale), _stream())
    return dqkv


def transpose_pad(x2d: torch.Tensor, out: torch.Tensor, rows_padded: int):
    """(R, C) -> (C, Rp) with zero columns for r >= R."""
    _chk(x2d, BF16, "transpose.x"); _chk(out, BF16, "transpose.out")
    R, C = x2d.shape
    if out.numel() < C * rows_padded:
        raise VgptError("transpose_pad: output too small")
    call("vgpt_transpose_pad_bf16", x2d.data_ptr(), out.data_ptr(), R, C, rows_padded, x2d.stride(0), _stream())
    return out[: C * rows_padded].view(C, rows_padded)


def silu_mul_fwd(gate_up, act_out, act):
    M = gate_up.numel() // gate_up.shape[-1]
    call("vgpt_silu_mul_fwd", gate_up.data_ptr(), act_out.data_ptr(), M, gate_up.shape[-1] // 2, act, _stream())
    return act_out


def silu_mul_bwd(gate_up, dact, dgate_up, act):
    M = gate_up.numel() // gate_up.shape[-1]
    call("vgpt_silu_mul_bwd", gate_up.data_ptr(), dact.data_ptr(), dgate_up.data_ptr(), M, gate_up.shape[-1] // 2, act,
         _stream())
    return dgate_up


def act_fwd(pre, act):
    y = torch.empty_like(pre)
    call("vgpt_act_fwd", pre.data_ptr(), y.data_ptr(), pre.numel(), act, _stream())
    return y


def act_bwd(pre, dy, act):
    dx = torch.empty_like(pre)
    call("vgpt_act_bwd", pre.data_ptr(), dy.data_ptr(), dx.data_ptr(), pre.numel(), act, _stream())
    return dx


_rstd_ws = {}


def rmsnorm_bwd(x, w, dy, dx, dw, eps, dres=None):
    _chk(x, BF16, "rmsnorm_bwd.x"); _chk(dy, BF16, "rmsnorm_bwd.dy"); _chk(dx, BF16, "rmsnorm_bwd.dx")
    _chk(dw, F32, "rmsnorm_bwd.dw")
    H = x.shape[-1]
    rows = x.numel() // H
    ws = _rstd_ws.get(x.device)
    if ws is None or ws.numel() < rows:
        ws = _rstd_ws[x.device] = torch.empty(max(rows, 1), dtype=F32, device=x.device)
    call("vgpt_rmsnorm_bwd", x.data_ptr(), w.data_ptr(), dy.data_ptr(), _ptr(dres), dx.data_ptr(), dw.data_ptr(),
         ws.data_ptr(), rows, H, float(eps), _stream())
    return dx


_splitk_ws = {}


def matmul(a, b, out=None, ta=False, tb=False, alpha=1.0, accumulate=False, out_dtype=BF16):
    """out = alpha * op(a) @ op(b) (+ out) on the generic strided kernel (small heads only)."""
    M, K = (a.shape[1], a.shape[0]) if ta else a.shape
    K2, N = (b.shape[1], b.shape[0]) if tb else b.shape
    if K != K2:
        raise VgptError("matmul: inner dimensions disagree")
    if out is None:
        out = torch.empty(M, N, dtype=out_dtype, device=a.device)
    sa_m, sa_k = (a.stride(1), a.stride(0)) if ta else (a.stride(0), a.stride(1))
    sb_k, sb_n = (b.stride(1), b.stride(0)) if tb else (b.stride(0), b.stride(1))
    # split-K reduction slices: sized by the C ABI's own query (include/vgpt.h), grown on demand, shared per device
    need = int(_lib.load().vgpt_matmul_generic_workspace_bytes(M, N, K)) // 4
    ws = _splitk_ws.get(a.device)
    if ws is None or ws.numel() < need:
        ws = _splitk_ws[a.device] = torch.empty(max(need, 1 << 18), dtype=F32, device=a.device)
    call("vgpt_matmul_generic", a.data_ptr(), _f32flag(a), sa_m, sa_k, b.data_ptr(), _f32flag(b), sb_k, sb_n,
         out.data_ptr(), _f32flag(out), out.stride(0), out.stride(1), M, N, K, float(alpha), int(accumulate),
         ws.data_ptr(), ws.numel(), _stream())
    return out


def colsum(x2d, out, accumulate=False):
    _chk(out, F32, "colsum.out")
    R, C = x2d.shape
    call("vgpt_colsum", x2d.data_ptr(), _f32flag(x2d), out.data_ptr(), R, C, x2d.stride(0), int(accumulate), _stream())
    return out


def lerp_frames(x1, x0, t, out):
    n = x1.shape[0]
    call("vgpt_lerp_frames", x1.data_ptr(), x0.data_ptr(), t.data_ptr(), out.data_ptr(), n, x1.numel() // max(n, 1),
         _stream())
    return out


def mse_frames(pred, x1, loss, dpred=None, n_mean=None):
    """loss[f] = mean((x1[f] - pred[f])^2); dpred = d(mean over n_mean terms)/dpred (n_mean: the length of the whole loss
    vector when these frames are only part of it; default: these frames alone)."""
    n = x1.shape[0]
    call("vgpt_mse_frames_mean", pred.data_ptr(), x1.data_ptr(), loss.data_ptr(), _ptr(dpred), n, n if n_mean is None else n_mean,
         x1.numel() // max(n, 1), _stream())
    return loss


def ln_mod_fwd(hidden2d, src_row, mod, v_out, xhat, rstd, ntok, eps=1e-6):
    nf = src_row.numel()
    call("vgpt_ln_mod_fwd", hidden2d.data_ptr(), src_row.data_ptr(), mod.data_ptr(), v_out.data_ptr(), xhat.data_ptr(),
         rstd.data_ptr(), nf, ntok, hidden2d.shape[-1], float(eps), _stream())


def ln_mod_bwd(dv, xhat, rstd, mod, dst_row, dhidden2d, dmod, ntok):
    nf = dst_row.numel()
    call("vgpt_ln_mod_bwd", dv.data_ptr(), xhat.data_ptr(), rstd.data_ptr(), mod.data_ptr(), dst_row.data_ptr(),
         dhidden2d.data_ptr(), dmod.data_ptr(), nf, ntok, dhidden2d.shape[-1], _stream())


def embed_bwd(ids, keep, dseq2d, dtable):
    call("vgpt_embed_bwd", ids.data_ptr(), keep.data_ptr(), dseq2d.data_ptr(), dtable.data_ptr(), ids.numel(),
         dseq2d.shape[-1], dtable.shape[0], _stream())


def patchify(x):
    nf, C, h, w = x.shape
    out = torch.empty(nf * (h // 2) * (w // 2), 16, dtype=BF16, device=x.device)
    call("vgpt_patchify", x.data_ptr(), out.data_ptr(), nf, C, h, w, _stream())
    return out


def unpatchify_bwd(dpred):
    nf, C, h, w = dpred.shape
    out = torch.empty(nf * (h // 2) * (w // 2), 16, dtype=BF16, device=dpred.device)
    call("vgpt_unpatchify_bwd", dpred.data_ptr(), out.data_ptr(), nf, C, h, w, _stream())
    return out


def gather_rows(x2d, row0, per):
    out = torch.empty(row0.numel() * per, x2d.shape[-1], dtype=BF16, device=x2d.device)
    call("vgpt_gather_rows", x2d.data_ptr(), row0.data_ptr(), out.data_ptr(), row0.numel(), per, x2d.shape[-1], _stream())
    return out


_partials = {}


def sumsq(g, out):
    ws = _partials.get(g.device)
    if ws is None:
        ws = _partials[g.device] = torch.empty(1024, dtype=F32, device=g.device)
    call("vgpt_sumsq", g.data_ptr(), _f32flag(g), out.data_ptr(), g.numel(), ws.data_ptr(), _stream())


def clip_coef(sumsq_t, coef, norm_out, max_norm, extra_scale=1.0):
    call("vgpt_clip_coef", sumsq_t.data_ptr(), coef.data_ptr(), _ptr(norm_out), float(max_norm), float(extra_scale),
         _stream())


def adamw_step(master, param, grad, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale=None):
    call("vgpt_adamw_step", master.data_ptr(), param.data_ptr(), grad.data_ptr(), _f32flag(grad), m.data_ptr(),
         v.data_ptr(), master.numel(), float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), int(step),
         _ptr(grad_scale), _stream())


def linear_dx(dy2d, weight, scratch=None, dres=None, out=None):
    """dX (M, K) = dY (M, N) @ W (N, K) (+ dres): the weight is read as the transposed operand of the GEMM
    (include/vgpt.h, vgpt_gemm_bf16_tr); `scratch` is unused (kept for callers of the transposing version)."""
    _chk(dy2d, BF16, "linear_dx.dy", contiguous=False); _chk(weight, BF16, "linear_dx.weight", contiguous=False)
    if dy2d.stride(1) != 1 or weight.stride(1) != 1:
        raise VgptError("linear_dx: rows must be contiguous")
    M, N = dy2d.shape
    N2, K = weight.shape
    if N != N2:
        raise VgptError("linear_dx: shape mismatch")
    if out is None:
        out = torch.empty(M, K, dtype=BF16, device=dy2d.device)
    epi, extra, ldr = (1, dres.data_ptr(), K) if dres is not None else (0, None, 0)
    call("vgpt_gemm_bf16_tr", dy2d.data_ptr(), weight.data_ptr(), out.data_ptr(), extra, M, K, N, dy2d.stride(0),
         weight.stride(0), K, ldr, epi, 0, 1, _stream())
    return out


def linear_dw(dy2d, x2d, scratch_a=None, scratch_b=None, out=None):
    """dW (N, K) = dY (M, N)^T @ X (M, K): both operands are read transposed, the reduction runs over the M rows
    (a partial last 64-row tile is zero-filled by the hardware)."""
    _chk(dy2d, BF16, "linear_dw.dy", contiguous=False); _chk(x2d, BF16, "linear_dw.x", contiguous=False)
    if dy2d.stride(1) != 1 or x2d.stride(1) != 1:
        raise VgptError("linear_dw: rows must be contiguous")
    M, N = dy2d.shape
    M2, K = x2d.shape
    if M != M2:
        raise VgptError("linear_dw: shape mismatch")
    if out is None:
        out = torch.empty(N, K, dtype=BF16, device=dy2d.device)
    call("vgpt_gemm_bf16_tr", dy2d.data_ptr(), x2d.data_ptr(), out.data_ptr(), None, N, K, M, dy2d.stride(0), x2d.stride(0),
         K, 0, 0, 1, 1, _stream())
    return out
