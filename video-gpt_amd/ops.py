"""Tensor-level wrappers over the C ABI (include/vgpt.h).

PyTorch is used for device memory and streams only: every function takes CUDA(ROCm) tensors,
checks shapes/dtypes on the host, and launches the HIP kernel on the current torch stream.
Nothing here falls back to torch ops for the arithmetic.
"""
from __future__ import annotations

import ctypes
import math
from typing import Optional

import torch

from . import _lib
from ._lib import (ACT_GELU, ACT_GELU_TANH, ACT_NONE, ACT_SILU, EPI_BIAS, EPI_NONE, EPI_RESID,
                   PRED_V, PRED_X1, VgptError, call)

BF16 = torch.bfloat16

_ACTS = {"silu": ACT_SILU, "swish": ACT_SILU, "gelu": ACT_GELU, "gelu_new": ACT_GELU_TANH,
         "gelu_pytorch_tanh": ACT_GELU_TANH, "gelu_tanh": ACT_GELU_TANH}


def act_code(name: str) -> int:
    if name not in _ACTS:
        raise VgptError(f"unsupported hidden_act {name!r}")
    return _ACTS[name]


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _chk(t: torch.Tensor, dtype, name: str, contiguous=True):
    if not t.is_cuda:
        raise VgptError(f"{name}: expected a GPU tensor (libvgpt_hip has no CPU path)")
    if t.dtype != dtype:
        raise VgptError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if contiguous and not t.is_contiguous():
        raise VgptError(f"{name}: expected a contiguous tensor")


# ---- transformer block ---------------------------------------------------------------------

def rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float, out: Optional[torch.Tensor] = None):
    _chk(x, BF16, "rmsnorm.x"); _chk(weight, BF16, "rmsnorm.weight")
    H = x.shape[-1]
    if weight.numel() != H:
        raise VgptError("rmsnorm: weight size mismatch")
    if out is None:
        out = torch.empty_like(x)
    else:
        _chk(out, BF16, "rmsnorm.out")
    if x.numel() == 0:
        return out
    call("vgpt_rmsnorm_fwd", x.data_ptr(), weight.data_ptr(), out.data_ptr(), x.numel() // H, H,
         float(eps), _stream())
    return out


def rope_inv_freq(head_dim: int, theta: float, device, ext_factors=None) -> torch.Tensor:
    """Phi3RotaryEmbedding.inv_freq, computed on the host exactly as transformers does; `ext_factors` (head_dim/2 values)
    are the short / long factors of a "su" / "longrope" checkpoint: inv_freq = 1 / (ext_factors * theta^(2i/d))."""
    shape = torch.arange(0, head_dim, 2, dtype=torch.int64).float() / head_dim
    if ext_factors is None:
        inv = 1.0 / (theta ** shape)
    else:
        ext = torch.tensor(list(ext_factors), dtype=torch.float32)
        if ext.numel() != shape.numel():
            raise VgptError(f"rope_scaling factors must have head_dim/2 = {shape.numel()} entries, got {ext.numel()}")
        inv = 1.0 / (ext * theta ** shape)
    return inv.to(device)


def rope_table(position_ids: torch.Tensor, inv_freq: torch.Tensor, round_bf16: bool = True, scale: float = 1.0):
    _chk(position_ids, torch.int64, "rope_table.position_ids"); _chk(inv_freq, torch.float32, "rope_table.inv_freq")
    tokens, half = position_ids.numel(), inv_freq.numel()
    cos = torch.empty(tokens, half, dtype=torch.float32, device=position_ids.device)
    sin = torch.empty_like(cos)
    call("vgpt_rope_table", position_ids.data_ptr(), inv_freq.data_ptr(), cos.data_ptr(), sin.data_ptr(),
         tokens, half, int(round_bf16), float(scale), _stream())
    return cos, sin


def rope_qk_inplace(qkv: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, n_q_heads: int,
                    n_kv_heads: int, head_dim: int):
    _chk(qkv, BF16, "rope.qkv"); _chk(cos, torch.float32, "rope.cos"); _chk(sin, torch.float32, "rope.sin")
    width = (n_q_heads + 2 * n_kv_heads) * head_dim
    if qkv.shape[-1] != width:
        raise VgptError("rope: qkv last dim mismatch")
    tokens = qkv.numel() // width
    if cos.numel() != tokens * (head_dim // 2) or sin.numel() != cos.numel():
        raise VgptError("rope: cos/sin table size mismatch")
    call("vgpt_rope_qk_inplace", qkv.data_ptr(), cos.data_ptr(), sin.data_ptr(), tokens, n_q_heads,
         n_kv_heads, head_dim, _stream())
    return qkv


def linear(x: torch.Tensor, weight: torch.Tensor, residual: Optional[torch.Tensor] = None,
           bias: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None):
    """out = x @ weight.T (+ residual | + bias) on the MFMA GEMM; x (..., K), weight (N, K)."""
    _chk(x, BF16, "linear.x"); _chk(weight, BF16, "linear.weight")
    K = x.shape[-1]
    N = weight.shape[0]
    if weight.shape[1] != K:
        raise VgptError("linear: K mismatch")
    M = x.numel() // K
    if out is None:
        out = torch.empty(*x.shape[:-1], N, dtype=BF16, device=x.device)
    else:
        _chk(out, BF16, "linear.out")
    if residual is not None and bias is not None:
        raise VgptError("linear: residual and bias together are not supported")
    epi, extra, ldr = EPI_NONE, None, 0
    if residual is not None:
        _chk(residual, BF16, "linear.residual")
        if residual.numel() != M * N:
            raise VgptError("linear: residual shape mismatch")
        epi, extra, ldr = EPI_RESID, residual, N
    elif bias is not None:
        _chk(bias, BF16, "linear.bias")
        epi, extra = EPI_BIAS, bias
    call("vgpt_gemm_bf16", x.data_ptr(), weight.data_ptr(), out.data_ptr(), _ptr(extra), M, N, K, K, K, N,
         ldr, epi, _stream())
    return out


# ---- RMSNorm folded into the GEMMs around it (include/vgpt.h: vgpt_gemm_bf16_resid_rstd -> *_prenorm) ----

def norm_workspace_bytes(M: int, N: int, K: int) -> int:
    """Workspace bytes of linear_resid_rstd for this shape; 0 = the shape keeps the separate RMSNorm."""
    return int(_lib.load().vgpt_gemm_norm_workspace_bytes(M, N, K))


def norm_workspace(nbytes: int, device) -> torch.Tensor:
    """A zeroed, 256-byte aligned workspace (its arrival counters are left at zero by every launch)."""
    return torch.zeros((nbytes + 255) // 256 * 256, dtype=torch.uint8, device=device)


def linear_resid_rstd(x: torch.Tensor, weight: torch.Tensor, residual: torch.Tensor, rstd: torch.Tensor, workspace: torch.Tensor,
                      eps: float, out: torch.Tensor):
    """out = x @ weight.T + residual, and rstd (M,) fp32 = 1 / rms of the rows of `out` (on the rounded values) for the RMSNorm
    that reads `out` next.  `out` may be `residual` (in place)."""
    _chk(x, BF16, "linear_resid_rstd.x"); _chk(weight, BF16, "linear_resid_rstd.weight"); _chk(residual, BF16, "linear_resid_rstd.residual")
    _chk(out, BF16, "linear_resid_rstd.out"); _chk(rstd, torch.float32, "linear_resid_rstd.rstd")
    _chk(workspace, torch.uint8, "linear_resid_rstd.workspace")
    K, N = x.shape[-1], weight.shape[0]
    M = x.numel() // K
    if weight.shape[1] != K or residual.numel() != M * N or out.numel() != M * N or rstd.numel() != M:
        raise VgptError("linear_resid_rstd: shape mismatch")
    call("vgpt_gemm_bf16_resid_rstd", x.data_ptr(), weight.data_ptr(), out.data_ptr(), residual.data_ptr(), rstd.data_ptr(),
         workspace.data_ptr(), workspace.numel(), float(eps), M, N, K, K, K, N, N, _stream())
    return out


def rms_rstd(x: torch.Tensor, eps: float, out: Optional[torch.Tensor] = None):
    """1 / rms of the rows of x (..., H): (M,) fp32."""
    _chk(x, BF16, "rms_rstd.x")
    H = x.shape[-1]
    M = x.numel() // H
    if out is None:
        out = torch.empty(M, dtype=torch.float32, device=x.device)
    else:
        _chk(out, torch.float32, "rms_rstd.out")
        if out.numel() != M:
            raise VgptError("rms_rstd: out size mismatch")
    call("vgpt_rms_rstd", x.data_ptr(), out.data_ptr(), M, H, H, float(eps), _stream())
    return out


def fold_norm_gain(weight: torch.Tensor, gain: torch.Tensor) -> torch.Tensor:
    """weight (N, K) * gain (K) per input column, rounded to bf16 once: the Linear behind an RMSNorm with the gain folded in."""
    _chk(weight, BF16, "fold_norm_gain.weight"); _chk(gain, BF16, "fold_norm_gain.gain")
    N, K = weight.shape
    if gain.numel() != K:
        raise VgptError("fold_norm_gain: gain size mismatch")
    out = torch.empty_like(weight)
    call("vgpt_fold_norm_gain", weight.data_ptr(), gain.data_ptr(), out.data_ptr(), N, K, _stream())
    return out


def linear_qkv_rope_prenorm(x: torch.Tensor, weight_folded: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, rstd: torch.Tensor,
                            n_q_heads: int, n_kv_heads: int, head_dim: int, out: torch.Tensor):
    """linear_qkv_rope(rmsnorm(x), W) with the norm folded in: x is the raw stream, weight_folded = fold_norm_gain(W, gain),
    rstd (M,) the rows' 1 / rms."""
    _chk(x, BF16, "qkv_prenorm.x"); _chk(weight_folded, BF16, "qkv_prenorm.weight"); _chk(out, BF16, "qkv_prenorm.out")
    _chk(cos, torch.float32, "qkv_prenorm.cos"); _chk(sin, torch.float32, "qkv_prenorm.sin"); _chk(rstd, torch.float32, "qkv_prenorm.rstd")
    K, N = x.shape[-1], weight_folded.shape[0]
    M = x.numel() // K
    if weight_folded.shape[1] != K or N != (n_q_heads + 2 * n_kv_heads) * head_dim or out.numel() != M * N:
        raise VgptError("linear_qkv_rope_prenorm: shape mismatch")
    if cos.numel() != M * (head_dim // 2) or sin.numel() != cos.numel() or rstd.numel() != M:
        raise VgptError("linear_qkv_rope_prenorm: table / statistics size mismatch")
    call("vgpt_gemm_bf16_rope_prenorm", x.data_ptr(), weight_folded.data_ptr(), out.data_ptr(), cos.data_ptr(), sin.data_ptr(),
         rstd.data_ptr(), M, N, K, K, K, N, n_q_heads + n_kv_heads, head_dim, _stream())
    return out


def gated_mlp_act_prenorm(x: torch.Tensor, w_gate_up_folded: torch.Tensor, rstd: torch.Tensor, act: int, out: torch.Tensor):
    """gated_mlp_act(rmsnorm(x), W) with the norm folded in (see linear_qkv_rope_prenorm)."""
    _chk(x, BF16, "gated_prenorm.x"); _chk(w_gate_up_folded, BF16, "gated_prenorm.w"); _chk(out, BF16, "gated_prenorm.out")
    _chk(rstd, torch.float32, "gated_prenorm.rstd")
    K = x.shape[-1]
    I = w_gate_up_folded.shape[0] // 2
    M = x.numel() // K
    if w_gate_up_folded.shape[1] != K or out.numel() != M * I or rstd.numel() != M:
        raise VgptError("gated_mlp_act_prenorm: shape mismatch")
    call("vgpt_gated_mlp_act_fwd_prenorm", x.data_ptr(), w_gate_up_folded.data_ptr(), out.data_ptr(), rstd.data_ptr(), M, I, K,
         K, K, I, act, _stream())
    return out


def linear_qkv_rope(x: torch.Tensor, weight: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, n_q_heads: int,
                    n_kv_heads: int, head_dim: int, out: Optional[torch.Tensor] = None):
    """qkv_proj + apply_rotary_pos_emb as one kernel (LVM/transform/sdpa_transform.py:39,52-53): == linear() followed by
    rope_qk_inplace(); cos / sin (tokens, head_dim/2) fp32 from rope_table, row t = token t of x."""
    _chk(x, BF16, "linear_qkv_rope.x"); _chk(weight, BF16, "linear_qkv_rope.weight")
    _chk(cos, torch.float32, "linear_qkv_rope.cos"); _chk(sin, torch.float32, "linear_qkv_rope.sin")
    K = x.shape[-1]
    N = weight.shape[0]
    if weight.shape[1] != K or N != (n_q_heads + 2 * n_kv_heads) * head_dim:
        raise VgptError("linear_qkv_rope: weight shape mismatch")
    M = x.numel() // K
    if cos.numel() != M * (head_dim // 2) or sin.numel() != cos.numel():
        raise VgptError("linear_qkv_rope: cos/sin table size mismatch")
    if out is None:
        out = torch.empty(*x.shape[:-1], N, dtype=BF16, device=x.device)
    else:
        _chk(out, BF16, "linear_qkv_rope.out")
        if out.numel() != M * N:
            raise VgptError("linear_qkv_rope: out shape mismatch")
    call("vgpt_gemm_bf16_rope", x.data_ptr(), weight.data_ptr(), out.data_ptr(), cos.data_ptr(), sin.data_ptr(), M, N, K,
         K, K, N, n_q_heads + n_kv_heads, head_dim, _stream())
    return out


def gated_mlp_act(x: torch.Tensor, w_gate_up: torch.Tensor, act: int = ACT_SILU,
                  out: Optional[torch.Tensor] = None, gate_up_out: Optional[torch.Tensor] = None):
    """act(x Wg^T) * (x Wu^T) with W_gate_up = [Wg ; Wu] (2I, K).  gate_up_out (M, 2I): the training forward's form --
    also stores the bf16 [gate | up] the backward needs and computes the activation from those rounded values."""
    _chk(x, BF16, "gated_mlp.x"); _chk(w_gate_up, BF16, "gated_mlp.w")
    K = x.shape[-1]
    I = w_gate_up.shape[0] // 2
    if w_gate_up.shape[1] != K or w_gate_up.shape[0] != 2 * I:
        raise VgptError("gated_mlp: weight shape mismatch")
    M = x.numel() // K
    if out is None:
        out = torch.empty(*x.shape[:-1], I, dtype=BF16, device=x.device)
    else:
        _chk(out, BF16, "gated_mlp.out")
    if gate_up_out is not None:
        _chk(gate_up_out, BF16, "gated_mlp.gate_up_out")
        if gate_up_out.numel() != M * 2 * I:
            raise VgptError("gated_mlp: gate_up_out must hold (M, 2I) values")
        call("vgpt_gated_mlp_act_fwd_keep", x.data_ptr(), w_gate_up.data_ptr(), out.data_ptr(), gate_up_out.data_ptr(), M, I, K,
             K, K, I, 2 * I, act, _stream())
        return out
    call("vgpt_gated_mlp_act_fwd", x.data_ptr(), w_gate_up.data_ptr(), out.data_ptr(), M, I, K, K, K, I,
         act, _stream())
    return out


# ---- attention --------------------------------------------------------------------------------

class AttnPlan:
    """Work items of a planned attention launch: items (n,4) int32, and one workspace sized by
    vgpt_attn_plan_workspace_bytes holding the per-(item, key tile) summary and the longest-first order."""

    def __init__(self, items, summary, order, n_items, item_rows=128):
        self.items, self.summary, self.order, self.n_items, self.item_rows = items, summary, order, n_items, item_rows


ITEM_ROWS = 128   # rows per work item of the planned forward kernel (four waves x 32 rows); 256 = the eight-wave kernel


class PackedMask:
    """Bit-packed (B,L,L) visibility mask + tile summary consumed by the attention kernel."""

    def __init__(self, bits: torch.Tensor, summary: torch.Tensor, B: int, L: int):
        self.bits, self.summary, self.B, self.L = bits, summary, B, L
        self._order = {}
        self._plans = {}

    def plan(self, segments=None, item_rows: int = ITEM_ROWS) -> "AttnPlan":
        """Work plan of the planned forward kernel (include/vgpt.h, vgpt_attn_plan_build) for the query rows of
        `segments` = ((batch, row_begin, row_end), ...); default: every row, one segment per batch item.  Each
        segment is cut into items of `item_rows` rows (128, or 256 for the eight-wave kernel: head dim 96); cut segments
        where packed sequences meet so that no item straddles two key sets."""
        if item_rows not in (128, 256):
            raise VgptError("attention plan: item_rows must be 128 or 256")
        skey = tuple(tuple(int(v) for v in s_) for s_ in segments) if segments is not None else None
        key = (skey, item_rows)
        p = self._plans.get(key)
        if p is None:
            segs = skey if skey is not None else tuple((b, 0, self.L) for b in range(self.B))
            items = []
            for b, r0, r1 in segs:
                if not (0 <= b < self.B and 0 <= r0 <= r1 <= self.L):
                    raise VgptError(f"attention plan: bad segment {(b, r0, r1)}")
                items += [(b, r, min(item_rows, r1 - r), 0) for r in range(r0, r1, item_rows)]
            dev = self.bits.device
            n_ = len(items)
            it = torch.tensor(items, dtype=torch.int32).reshape(n_, 4).to(dev)
            nbytes = int(_lib.load().vgpt_attn_plan_workspace_bytes(self.L, max(n_, 1)))
            if nbytes < 0:
                raise VgptError("vgpt_attn_plan_workspace_bytes: bad shape")
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            nkt = (self.L + 63) // 64
            s_bytes = (max(n_, 1) * nkt * 2 + 255) // 256 * 256
            summ = ws[: max(n_, 1) * nkt * 2].view(torch.int16).view(max(n_, 1), nkt)
            order = ws[s_bytes: s_bytes + max(n_, 1) * 4].view(torch.int32)
            if n_:
                call("vgpt_attn_plan_build", self.bits.data_ptr(), self.B, self.L, it.data_ptr(), n_,
                     summ.data_ptr(), order.data_ptr(), _stream())
            p = self._plans[key] = AttnPlan(it, summ, order, n_, item_rows)
            p.workspace = ws
        return p

    def order(self, q_start: int = 0) -> torch.Tensor:
        """Longest-first launch order of the q blocks from row q_start on (include/vgpt.h, vgpt_attn_qblock_order)."""
        t = self._order.get(q_start)
        if t is None:
            n = (self.L + 127) // 128 - q_start // 128
            t = torch.empty(self.B, max(n, 0), dtype=torch.int32, device=self.bits.device)
            call("vgpt_attn_qblock_order", self.summary.data_ptr(), self.B, self.L, q_start, t.data_ptr(), _stream())
            self._order[q_start] = t
        return t

    def count_empty_rows(self) -> int:
        cnt = torch.empty(1, dtype=torch.int32, device=self.bits.device)
        call("vgpt_mask_count_empty_rows", self.bits.data_ptr(), cnt.data_ptr(), self.B, self.L, _stream())
        return int(cnt.item())


def _alloc_mask(B: int, L: int, device):
    W = (L + 31) // 32
    bits = torch.empty(B, L, W, dtype=torch.int32, device=device)
    summary = torch.empty(B, (L + 127) // 128, (L + 63) // 64, dtype=torch.uint8, device=device)
    return bits, summary


def pack_mask(mask: torch.Tensor) -> PackedMask:
    """mask: (B,L,L) bool/uint8 (True = visible), or the additive (B,1,L,L) bf16/fp32 mask."""
    if not mask.is_cuda:
        raise VgptError("pack_mask: expected a GPU tensor")
    if mask.dim() == 4:
        if mask.shape[1] != 1:
            raise VgptError("pack_mask: additive mask must be (B,1,L,L)")
        B, _, L, L2 = mask.shape
        if mask.dtype not in (torch.float32, BF16):
            raise VgptError("pack_mask: additive mask must be bf16 or fp32")
        mask = mask.contiguous()
        bits, summary = _alloc_mask(B, L, mask.device)
        call("vgpt_mask_pack_additive", mask.data_ptr(), int(mask.dtype == torch.float32), bits.data_ptr(),
             B, L, _stream())
    else:
        if mask.dim() != 3:
            raise VgptError("attention_mask parameter was unavailable or invalid")
        B, L, L2 = mask.shape
        if mask.dtype == torch.bool:
            m8 = mask.contiguous().view(torch.uint8)
        elif mask.dtype == torch.uint8:
            m8 = mask.contiguous()
        else:
            m8 = (mask != 0).contiguous().view(torch.uint8)
        bits, summary = _alloc_mask(B, L, mask.device)
        call("vgpt_mask_pack_bool", m8.data_ptr(), bits.data_ptr(), B, L, _stream())
    if L != L2:
        raise VgptError("pack_mask: mask must be square")
    call("vgpt_mask_tile_summary", bits.data_ptr(), summary.data_ptr(), B, L, _stream())
    return PackedMask(bits, summary, B, L)


def build_mask_from_layout(attr: torch.Tensor, B: int, L: int) -> PackedMask:
    """attr: (B, L, 2) int32 token attributes on the GPU (layout.TokenLayout.attr) -> packed rows + tile summary,
    generated on the device (include/vgpt.h, vgpt_mask_build_tokens); no (B,L,L) tensor exists at any point."""
    if not attr.is_cuda:
        raise VgptError("build_mask_from_layout: expected a GPU tensor")
    if attr.dtype != torch.int32 or tuple(attr.shape) != (B, L, 2) or not attr.is_contiguous():
        raise VgptError("build_mask_from_layout: attr must be a contiguous (B, L, 2) int32 tensor")
    bits, summary = _alloc_mask(B, L, attr.device)
    call("vgpt_mask_build_tokens", attr.data_ptr(), bits.data_ptr(), B, L, _stream())
    call("vgpt_mask_tile_summary", bits.data_ptr(), summary.data_ptr(), B, L, _stream())
    return PackedMask(bits, summary, B, L)


def as_packed_mask(mask, device=None) -> PackedMask:
    """PackedMask | layout.TokenLayout | (B,L,L) bool | additive (B,1,L,L) -> PackedMask."""
    if isinstance(mask, PackedMask):
        return mask
    if hasattr(mask, "packed_mask"):
        if device is None:
            raise VgptError("as_packed_mask: a TokenLayout needs the target device")
        return mask.packed_mask(device)
    return pack_mask(mask)


def attention_qkv(qkv: torch.Tensor, pm: PackedMask, n_heads: int, n_kv_heads: int, head_dim: int,
                  scale: Optional[float] = None, out: Optional[torch.Tensor] = None, variant: int = 0):
    """Attention on the fused (B, L, (n_q+2n_kv)*hd) projection buffer (RoPE already applied).  variant 0: planned
    launch; 2: aligned 128-row q blocks, longest first; 1: the slow V layout."""
    _chk(qkv, BF16, "attention.qkv")
    B, L, width = qkv.shape
    if width != (n_heads + 2 * n_kv_heads) * head_dim or B != pm.B or L != pm.L:
        raise VgptError("attention: qkv / mask shape mismatch")
    if out is None:
        out = torch.empty(B, L, n_heads * head_dim, dtype=BF16, device=qkv.device)
    if scale is None:
        scale = 1.0 / math.sqrt(head_dim)
    es = qkv.element_size()
    kq = qkv.data_ptr() + n_heads * head_dim * es
    vq = kq + n_kv_heads * head_dim * es
    sb, ss = L * width, width
    if variant == 0:   # planned launch
        _attn_plan_call(qkv.data_ptr(), kq, vq, out.data_ptr(), None, pm,
                        pm.plan(None), B, L,
                        n_heads, n_kv_heads, head_dim,
                        (sb, head_dim, ss) * 3 + (L * n_heads * head_dim, head_dim, n_heads * head_dim), scale)
        return out
    if variant == 2:   # 4-wave kernel on aligned 128-row q blocks, launched longest-first
        call("vgpt_attn_blockmask_fwd_qrange", qkv.data_ptr(), kq, vq, out.data_ptr(), 0, pm.bits.data_ptr(),
             pm.summary.data_ptr(), pm.order(0).data_ptr(), B, L, n_heads, n_kv_heads, head_dim,
             sb, head_dim, ss, sb, head_dim, ss, sb, head_dim, ss,
             L * n_heads * head_dim, head_dim, n_heads * head_dim, float(scale), _stream())
        return out
    call("vgpt_attn_blockmask_fwd", qkv.data_ptr(), kq, vq, out.data_ptr(), pm.bits.data_ptr(),
         pm.summary.data_ptr(), B, L, n_heads, n_kv_heads, head_dim,
         sb, head_dim, ss, sb, head_dim, ss, sb, head_dim, ss,
         L * n_heads * head_dim, head_dim, n_heads * head_dim, float(scale), variant, _stream())
    return out


def _attn_plan_call(q, k, v, o, lse, pm, plan, B, L, n_heads, n_kv_heads, head_dim, strides, scale):
    if plan.n_items:
        call("vgpt_attn_fwd_plan", q, k, v, o, lse, pm.bits.data_ptr(), plan.items.data_ptr(), plan.summary.data_ptr(),
             plan.order.data_ptr(), plan.n_items, B, L, n_heads, n_kv_heads, head_dim,
             *strides, float(scale), int(plan.item_rows), _stream())


def attention_qkv_range(qkv_full: torch.Tensor, pm: PackedMask, n_heads: int, n_kv_heads: int, head_dim: int,
                        q_start: int, out_active: torch.Tensor, scale: Optional[float] = None, segments=None,
                        item_rows: int = ITEM_ROWS):
    """Attention of query rows [q_start, L) against all L rows of the fused (1, L, 3H-like) buffer; `out_active`
    holds the L - q_start computed rows (condition-prefix reuse, see include/vgpt.h).  segments: optional
    ((0, row_begin, row_end), ...) covering [q_start, L), cut where packed sequences meet."""
    _chk(qkv_full, BF16, "attention.qkv"); _chk(out_active, BF16, "attention.out")
    B, L, width = qkv_full.shape
    if B != 1 or pm.B != 1 or pm.L != L:
        raise VgptError("attention_qkv_range: needs the packed single-row layout")
    if scale is None:
        scale = 1.0 / math.sqrt(head_dim)
    hq = n_heads * head_dim
    kq = qkv_full.data_ptr() + hq * 2
    vq = kq + n_kv_heads * head_dim * 2
    o_base = out_active.data_ptr() - q_start * hq * 2   # absolute-row addressing of the active output buffer
    if item_rows == 256 and head_dim != 96:
        item_rows = ITEM_ROWS        # the eight-wave kernel exists for head dim 96 only
    plan = pm.plan(segments if segments is not None else ((0, q_start, L),), item_rows)
    _attn_plan_call(qkv_full.data_ptr(), kq, vq, o_base, None, pm, plan, 1, L, n_heads, n_kv_heads, head_dim,
                    (L * width, head_dim, width) * 3 + (L * hq, head_dim, hq), scale)
    return out_active


def attention_fp8_workspace(B: int, L: int, n_heads: int, n_kv_heads: int, head_dim: int, device) -> torch.Tensor:
    """Caller-owned workspace of the MX-fp8 attention, sized by the library (vgpt_attn_fp8_workspace_bytes)."""
    nbytes = int(_lib.load().vgpt_attn_fp8_workspace_bytes(B, L, n_heads, n_kv_heads, head_dim))
    if nbytes < 0:
        raise VgptError(f"attention_fp8: unsupported shape (head_dim {head_dim}: the fp8 kernel is built for 96)")
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


def attention_fp8_quantize(qkv: torch.Tensor, ws: torch.Tensor, n_heads: int, n_kv_heads: int, head_dim: int,
                           scale: Optional[float] = None, row_begin: int = 0):
    """Q / K / V of the fused (B, L, .) buffer, rows >= row_begin (a multiple of 64), into the fp8 workspace."""
    _chk(qkv, BF16, "attention.qkv")
    B, L, width = qkv.shape
    hq = n_heads * head_dim
    kq = qkv.data_ptr() + hq * 2
    vq = kq + n_kv_heads * head_dim * 2
    sb, ss = L * width, width
    if scale is None:
        scale = 1.0 / math.sqrt(head_dim)
    call("vgpt_attn_fp8_quantize", qkv.data_ptr(), kq, vq, ws.data_ptr(), B, L, int(row_begin), n_heads, n_kv_heads, head_dim,
         sb, head_dim, ss, sb, head_dim, ss, sb, head_dim, ss, float(scale), _stream())


_FP8_WS = {}


def attention_qkv_fp8(qkv: torch.Tensor, pm: PackedMask, n_heads: int, n_kv_heads: int, head_dim: int,
                      scale: Optional[float] = None, out: Optional[torch.Tensor] = None, q_start: int = 0, segments=None,
                      workspace: Optional[torch.Tensor] = None, quant_from: int = 0):
    """MX-fp8 attention (include/vgpt.h, vgpt_attn_fp8_quantize + vgpt_attn_fwd_plan_fp8) on the fused
    (B, L, (n_q + 2 n_kv) * hd) projection buffer (RoPE applied): rows [q_start, L) of `segments` (default: all of
    them) against all L keys.  `out` holds rows [q_start, L).  workspace / quant_from: a caller-owned workspace whose rows
    below quant_from (a multiple of 64) already hold their quantised form (the engine's cached prefix); by default a
    shared scratch workspace is filled from row 0.  Inference only; tolerance of fp8 operands."""
    _chk(qkv, BF16, "attention.qkv")
    B, L, width = qkv.shape
    if width != (n_heads + 2 * n_kv_heads) * head_dim or B != pm.B or L != pm.L:
        raise VgptError("attention_fp8: qkv / mask shape mismatch")
    if q_start and B != 1:
        raise VgptError("attention_fp8: q_start needs the packed single-row layout")
    hq = n_heads * head_dim
    if out is None:
        out = torch.empty(B, L - q_start, hq, dtype=BF16, device=qkv.device)
    ws = workspace
    if ws is None:
        if quant_from:
            raise VgptError("attention_fp8: quant_from needs a caller-owned workspace")
        key = (qkv.device, B, L, n_heads, n_kv_heads, head_dim)
        ws = _FP8_WS.get(key)
        if ws is None:
            ws = _FP8_WS[key] = attention_fp8_workspace(B, L, n_heads, n_kv_heads, head_dim, qkv.device)
    attention_fp8_quantize(qkv, ws, n_heads, n_kv_heads, head_dim, scale, quant_from)
    plan = pm.plan(segments if segments is not None else (tuple((b, q_start, L) for b in range(B))))
    if plan.n_items:
        call("vgpt_attn_fwd_plan_fp8", ws.data_ptr(), out.data_ptr() - q_start * hq * 2, pm.bits.data_ptr(),
             plan.items.data_ptr(), plan.summary.data_ptr(), plan.order.data_ptr(), plan.n_items, B, L, n_heads, n_kv_heads,
             head_dim, (L - q_start) * hq if B > 1 else L * hq, head_dim, hq, _stream())
    return out


def sdpa(query: torch.Tensor, key: torch.Tensor, value: torch.Tensor, attn_mask=None, dropout_p: float = 0.0,
         is_causal: bool = False, scale: Optional[float] = None, variant: int = 0):
    """Drop-in for the `local_attn` slot (F.scaled_dot_product_attention) on (B, heads, S, d) tensors.

    attn_mask: PackedMask, a (B,S,S) bool mask, or the additive (B,1,S,S) mask the reference builds.
    """
    if dropout_p != 0.0:
        raise VgptError("sdpa: dropout is not supported")
    if is_causal or attn_mask is None:
        raise VgptError("sdpa: an explicit block mask is required (is_causal is not used by the reference path)")
    for n, t in (("query", query), ("key", key), ("value", value)):
        _chk(t, BF16, f"sdpa.{n}", contiguous=False)
        if t.stride(-1) != 1:
            raise VgptError(f"sdpa.{n}: head_dim axis must be contiguous")
    B, Hq, S, d = query.shape
    Hkv = key.shape[1]
    pm = attn_mask if isinstance(attn_mask, PackedMask) else pack_mask(attn_mask)
    out = torch.empty(B, Hq, S, d, dtype=BF16, device=query.device)
    if scale is None:
        scale = 1.0 / math.sqrt(d)
    if variant == 0:
        _attn_plan_call(query.data_ptr(), key.data_ptr(), value.data_ptr(), out.data_ptr(), None, pm,
                        pm.plan(None), B, S, Hq, Hkv,
                        d, (query.stride(0), query.stride(1), query.stride(2), key.stride(0), key.stride(1), key.stride(2),
                            value.stride(0), value.stride(1), value.stride(2), out.stride(0), out.stride(1), out.stride(2)),
                        scale)
        return out
    call("vgpt_attn_blockmask_fwd", query.data_ptr(), key.data_ptr(), value.data_ptr(), out.data_ptr(),
         pm.bits.data_ptr(), pm.summary.data_ptr(), B, S, Hq, Hkv, d,
         query.stride(0), query.stride(1), query.stride(2), key.stride(0), key.stride(1), key.stride(2),
         value.stride(0), value.stride(1), value.stride(2), out.stride(0), out.stride(1), out.stride(2),
         float(scale), 0 if variant == 2 else variant, _stream())
    return out


# ---- model glue -----------------------------------------------------------------------------

def embed_gather(ids: torch.Tensor, table: torch.Tensor, out: Optional[torch.Tensor] = None):
    _chk(ids, torch.int64, "embed.ids"); _chk(table, BF16, "embed.table")
    V, H = table.shape
    if out is None:
        out = torch.empty(*ids.shape, H, dtype=BF16, device=ids.device)
    call("vgpt_embed_gather", ids.data_ptr(), table.data_ptr(), out.data_ptr(), ids.numel(), H, V, _stream())
    return out


def patch_embed(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, pos_embed: torch.Tensor,
                dst_row: torch.Tensor, seq: torch.Tensor, pos_max: int):
    """x: (n_frames, C, h, w); writes H-wide token rows into seq (rows, H) at dst_row[f] + t."""
    _chk(x, BF16, "patch_embed.x"); _chk(weight, BF16, "patch_embed.weight"); _chk(bias, BF16, "patch_embed.bias")
    _chk(pos_embed, BF16, "patch_embed.pos_embed"); _chk(dst_row, torch.int32, "patch_embed.dst_row")
    _chk(seq, BF16, "patch_embed.seq")
    nf, C, h, w = x.shape
    H = weight.shape[0]
    if dst_row.numel() != nf:
        raise VgptError("patch_embed: dst_row size mismatch")
    call("vgpt_patch_embed_fwd", x.data_ptr(), weight.data_ptr(), bias.data_ptr(), pos_embed.data_ptr(),
         dst_row.data_ptr(), seq.data_ptr(), nf, C, h, w, H, pos_max, _stream())
    return seq


def timestep_freqs(dim: int, device, max_period: float = 10000.0) -> torch.Tensor:
    """freqs of TimestepEmbedder.timestep_embedding (LVM/model.py:50-53), host-computed."""
    half = dim // 2
    f = torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32) / half)
    return f.to(device)


def timestep_sinusoid(t: torch.Tensor, freqs: torch.Tensor, out: Optional[torch.Tensor] = None):
    _chk(t, torch.float32, "timestep.t"); _chk(freqs, torch.float32, "timestep.freqs")
    n, half = t.numel(), freqs.numel()
    if out is None:
        out = torch.empty(n, 2 * half, dtype=BF16, device=t.device)
    call("vgpt_timestep_sinusoid", t.data_ptr(), freqs.data_ptr(), out.data_ptr(), n, half, _stream())
    return out


def linear_small(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], pre_act: int = ACT_NONE,
                 post_act: int = ACT_NONE, out: Optional[torch.Tensor] = None,
                 out_row: Optional[torch.Tensor] = None, ldo: Optional[int] = None):
    _chk(x, BF16, "linear_small.x"); _chk(weight, BF16, "linear_small.weight")
    M, K = x.shape
    N = weight.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=BF16, device=x.device)
    if ldo is None:
        ldo = out.shape[-1]
    call("vgpt_linear_small", x.data_ptr(), weight.data_ptr(), _ptr(bias), out.data_ptr(), _ptr(out_row), M, N,
         K, x.stride(0), ldo, pre_act, post_act, _stream())
    return out


def final_layer(hidden: torch.Tensor, src_row: torch.Tensor, mod: torch.Tensor, weight: torch.Tensor,
                bias: torch.Tensor, out: torch.Tensor, eps: float = 1e-6):
    """hidden (rows,H); mod (n_frames,2H) [shift|scale]; out (n_frames, C, h, w)."""
    _chk(hidden, BF16, "final.hidden"); _chk(src_row, torch.int32, "final.src_row"); _chk(mod, BF16, "final.mod")
    _chk(weight, BF16, "final.weight"); _chk(bias, BF16, "final.bias"); _chk(out, BF16, "final.out")
    nf, C, h, w = out.shape
    H = hidden.shape[-1]
    call("vgpt_final_layer_fwd", hidden.data_ptr(), src_row.data_ptr(), mod.data_ptr(), weight.data_ptr(),
         bias.data_ptr(), out.data_ptr(), nf, C, h, w, H, float(eps), _stream())
    return out


# ---- sampler ----------------------------------------------------------------------------------

def sampler_set_timesteps(sigma: torch.Tensor, step: torch.Tensor, timesteps: torch.Tensor):
    call("vgpt_sampler_set_timesteps", sigma.data_ptr(), step.data_ptr(), sigma.numel() - 1, timesteps.data_ptr(),
         timesteps.numel(), _stream())


def euler_cfg_update(z: torch.Tensor, z_model: torch.Tensor, pred: torch.Tensor, sigma: torch.Tensor,
                     step: torch.Tensor, pred_type: int, use_cfg: bool, cfg_scale: float):
    _chk(z, torch.float32, "euler.z"); _chk(z_model, BF16, "euler.z_model"); _chk(pred, BF16, "euler.pred")
    nf = z.shape[0]
    call("vgpt_euler_cfg_update", z.data_ptr(), z_model.data_ptr(), pred.data_ptr(), sigma.data_ptr(),
         step.data_ptr(), sigma.numel() - 1, nf, z.numel() // nf, pred_type, int(use_cfg), float(cfg_scale), _stream())


def sampler_copy_step_rows(src: torch.Tensor, dst: torch.Tensor, step: torch.Tensor):
    """src: (n_steps, n_layers, rows, W) contiguous; dst: a (n_layers, rows, W) VIEW whose rows are contiguous (layer
    stride arbitrary).  dst[l] = src[*step][l]."""
    n_steps, n_layers, rows, W = src.shape
    if not src.is_contiguous() or dst.shape != src.shape[1:] or dst.stride(2) != 1 or dst.stride(1) != W \
            or dst.dtype != src.dtype:
        raise VgptError("sampler_copy_step_rows: bad operand layout")
    es = src.element_size()
    call("vgpt_sampler_copy_step_rows", src.data_ptr(), dst.data_ptr(), step.data_ptr(), n_steps, n_layers,
         rows * W * es, n_layers * rows * W * es, rows * W * es, dst.stride(0) * es, _stream())


def sampler_advance(step: torch.Tensor):
    call("vgpt_sampler_advance", step.data_ptr(), _stream())


def cast_f32_to_bf16(src: torch.Tensor, dst: torch.Tensor):
    call("vgpt_cast_f32_to_bf16", src.data_ptr(), dst.data_ptr(), src.numel(), _stream())


class HipGraph:
    """hipGraph capture/replay of a fixed launch sequence on the current torch stream."""

    def __init__(self):
        self._exec = None

    def capture(self, fn):
        stream = _stream()
        call("vgpt_graph_begin_capture", stream)
        try:
            fn()
        finally:
            out = ctypes.c_void_p()
            call("vgpt_graph_end_capture", stream, ctypes.byref(out))
        self._exec = out.value
        return self

    def replay(self):
        if self._exec is None:
            raise VgptError("HipGraph.replay before capture")
        call("vgpt_graph_launch", self._exec, _stream())

    def __del__(self):
        if getattr(self, "_exec", None):
            try:
                call("vgpt_graph_destroy", self._exec)
            except Exception:
                pass
            self._exec = None


# ---- VAE (fp32, NCHW) -------------------------------------------------------------------------

F32 = torch.float32


def groupnorm_stats(x: torch.Tensor, groups: int, eps: float) -> torch.Tensor:
    _chk(x, F32, "groupnorm_stats.x")
    N, C = x.shape[:2]
    HW = x.numel() // max(N * C, 1)
    stats = torch.empty(N, groups, 2, dtype=F32, device=x.device)
    call("vgpt_groupnorm_stats", x.data_ptr(), stats.data_ptr(), N, C, HW, groups, float(eps), _stream())
    return stats


def conv2d(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
           resid: Optional[torch.Tensor] = None, gn=None, ksize: int = 3, stride: int = 1, upsample: bool = False,
           cout: Optional[int] = None, w_transposed: bool = False, ldw: Optional[int] = None,
           w_batch_stride: int = 0, out: Optional[torch.Tensor] = None):
    """y = conv(f(x), weight) + bias + resid on the fp32-MFMA implicit GEMM (see include/vgpt.h).
    gn = (stats, gamma, beta, groups, silu) fuses GroupNorm(+SiLU) into the input load."""
    _chk(x, F32, "conv2d.x"); _chk(weight, F32, "conv2d.weight")
    N, Cin, Hin, Win = x.shape
    if cout is None:
        cout = weight.shape[0]
    if ldw is None:
        ldw = Cin * ksize * ksize
    Hv, Wv = (Hin * 2, Win * 2) if upsample else (Hin, Win)
    Ho, Wo = (Hv // 2, Wv // 2) if stride == 2 else (Hv, Wv)
    if out is None:
        out = torch.empty(N, cout, Ho, Wo, dtype=F32, device=x.device)
    else:
        _chk(out, F32, "conv2d.out")
    if resid is not None:
        _chk(resid, F32, "conv2d.resid")
        if resid.numel() != out.numel():
            raise VgptError("conv2d: residual shape mismatch")
    if bias is not None:
        _chk(bias, F32, "conv2d.bias")
    stats = gamma = beta = None
    groups = silu = 0
    if gn is not None:
        stats, gamma, beta, groups, silu = gn
        _chk(stats, F32, "conv2d.gn_stats"); _chk(gamma, F32, "conv2d.gn_gamma"); _chk(beta, F32, "conv2d.gn_beta")
    call("vgpt_conv2d_fwd", x.data_ptr(), weight.data_ptr(), _ptr(bias), _ptr(resid), _ptr(stats), _ptr(gamma),
         _ptr(beta), out.data_ptr(), N, Cin, Hin, Win, cout, ksize, stride, int(upsample), int(groups), int(silu),
         int(w_transposed), int(ldw), int(w_batch_stride), _stream())
    return out


def conv_pack_bx3(weight: torch.Tensor):
    """(Cout, Cin, 3, 3) fp32 -> LDS-ready split-bf16 weight images for conv2d_bx3 (include/vgpt.h)."""
    _chk(weight, F32, "conv_pack_bx3.weight")
    Cout, Cin, kh, kw = weight.shape
    if (kh, kw) != (3, 3):
        raise VgptError("conv_pack_bx3: 3x3 kernels only")
    from ._lib import load
    nbytes = int(load().vgpt_conv_bx3_packed_bytes(Cout, Cin))
    packed = torch.empty(nbytes, dtype=torch.uint8, device=weight.device)
    call("vgpt_conv_pack_weights_bx3", weight.data_ptr(), packed.data_ptr(), Cout, Cin, _stream())
    return packed, Cout, Cin


def conv2d_bx3(x: torch.Tensor, packed, bias: Optional[torch.Tensor] = None, resid: Optional[torch.Tensor] = None, gn=None,
               upsample: bool = False, out: Optional[torch.Tensor] = None):
    """3x3 stride-1 convolution with split-bf16 operands on the bf16 MFMA (fp32 in / out); packed = conv_pack_bx3(w)."""
    _chk(x, F32, "conv2d_bx3.x")
    img, cout, cin = packed
    N, Cin, Hin, Win = x.shape
    if cin != Cin:
        raise VgptError("conv2d_bx3: packed weights do not match the input channels")
    Ho, Wo = (Hin * 2, Win * 2) if upsample else (Hin, Win)
    if out is None:
        out = torch.empty(N, cout, Ho, Wo, dtype=F32, device=x.device)
    if resid is not None:
        _chk(resid, F32, "conv2d_bx3.resid")
        if resid.numel() != out.numel():
            raise VgptError("conv2d_bx3: residual shape mismatch")
    stats = gamma = beta = None
    groups = silu = 0
    if gn is not None:
        stats, gamma, beta, groups, silu = gn
    call("vgpt_conv2d_bx3_fwd", x.data_ptr(), img.data_ptr(), _ptr(bias), _ptr(resid), _ptr(stats), _ptr(gamma), _ptr(beta),
         out.data_ptr(), N, Cin, Hin, Win, cout, int(upsample), int(groups), int(silu), _stream())
    return out


def conv1x1_pack_bx3(weight: torch.Tensor):
    """(Cout, Cin[, 1, 1]) fp32 -> split-bf16 weight images for conv1x1_bx3 (include/vgpt.h)."""
    _chk(weight, F32, "conv1x1_pack_bx3.weight")
    Cout, Cin = weight.shape[:2]
    if weight.numel() != Cout * Cin:
        raise VgptError("conv1x1_pack_bx3: 1x1 kernels only")
    from ._lib import load
    packed = torch.empty(int(load().vgpt_conv1x1_bx3_packed_bytes(Cout, Cin)), dtype=torch.uint8, device=weight.device)
    call("vgpt_conv1x1_pack_weights_bx3", weight.data_ptr(), packed.data_ptr(), Cout, Cin, _stream())
    return packed, Cout, Cin


def conv1x1_bx3(x: torch.Tensor, packed, bias: Optional[torch.Tensor] = None, resid: Optional[torch.Tensor] = None, gn=None,
                out: Optional[torch.Tensor] = None):
    """1x1 convolution with split-bf16 operands on the bf16 MFMA (fp32 in / out, Cin % 32 == 0); packed = conv1x1_pack_bx3(w)."""
    _chk(x, F32, "conv1x1_bx3.x")
    img, cout, cin = packed
    N, Cin, Hin, Win = x.shape
    if cin != Cin:
        raise VgptError("conv1x1_bx3: packed weights do not match the input channels")
    if out is None:
        out = torch.empty(N, cout, Hin, Win, dtype=F32, device=x.device)
    if resid is not None:
        _chk(resid, F32, "conv1x1_bx3.resid")
        if resid.numel() != out.numel():
            raise VgptError("conv1x1_bx3: residual shape mismatch")
    stats = gamma = beta = None
    groups = silu = 0
    if gn is not None:
        stats, gamma, beta, groups, silu = gn
    call("vgpt_conv1x1_bx3_fwd", x.data_ptr(), img.data_ptr(), _ptr(bias), _ptr(resid), _ptr(stats), _ptr(gamma), _ptr(beta),
         out.data_ptr(), N, Cin, Hin * Win, cout, int(groups), int(silu), _stream())
    return out


def col_softmax(s: torch.Tensor, scale: float):
    _chk(s, F32, "col_softmax.s")
    N, keys, queries = s.shape
    call("vgpt_col_softmax", s.data_ptr(), N, keys, queries, float(scale), _stream())
    return s


def vae_sample(moments: torch.Tensor, noise: torch.Tensor, shift: float, scaling: float):
    _chk(moments, F32, "vae_sample.moments"); _chk(noise, F32, "vae_sample.noise")
    N = moments.shape[0]
    per = moments.numel() // max(2 * N, 1)
    if noise.numel() != N * per:
        raise VgptError("vae_sample: noise shape mismatch")
    z = torch.empty_like(noise)
    call("vgpt_vae_sample", moments.data_ptr(), noise.data_ptr(), z.data_ptr(), N, per, float(shift), float(scaling),
         _stream())
    return z


def vae_postprocess_u8(x: torch.Tensor):
    _chk(x, F32, "vae_postprocess_u8.x")
    N, C, H, W = x.shape
    out = torch.empty(N, H, W, C, dtype=torch.uint8, device=x.device)
    call("vgpt_vae_postprocess_u8", x.data_ptr(), out.data_ptr(), N, C, H, W, _stream())
    return out


def affine_to_f32(x: torch.Tensor, mul: float, add: float = 0.0):
    if x.dtype not in (F32, BF16):
        raise VgptError("affine_to_f32: expected fp32 or bf16 input")
    x = x.contiguous()
    if not x.is_cuda:
        raise VgptError("affine_to_f32: expected a GPU tensor")
    y = torch.empty(x.shape, dtype=F32, device=x.device)
    call("vgpt_affine_to_f32", x.data_ptr(), int(x.dtype == BF16), y.data_ptr(), x.numel(), float(mul), float(add),
         _stream())
    return y
