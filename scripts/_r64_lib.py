"""ctypes access to the attention experiment library (`make -C video-gpt_amd/csrc experiment-r64` ->
video-gpt_amd/libvgpt_x_r64.so): 256-row plans and the 64-rows-per-wave kernel.  Not part of the product."""
import ctypes, importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module("video-gpt_amd")
ops = importlib.import_module("video-gpt_amd.ops")
L_ = importlib.import_module("video-gpt_amd._lib")
_P, _I64 = ctypes.c_void_p, ctypes.c_int64
xlib = ctypes.CDLL(os.path.join(ROOT, "video-gpt_amd", "libvgpt_x_r64.so"))
xlib.vgpt_x_attn_fwd_r64.restype = ctypes.c_int
xlib.vgpt_x_attn_fwd_r64.argtypes = [_P] * 9 + [_I64, _P, _I64, _I64] + [ctypes.c_int] * 3 + [_I64] * 12 + [ctypes.c_float, _P]


class Plan256:
    """Row segments cut into 256-row items, summarised and ordered by the product's vgpt_attn_plan_build."""

    def __init__(self, pm, segments, n_heads):
        dev = pm.bits.device
        items = [(b, r, min(256, r1 - r), 0) for b, r0, r1 in segments for r in range(r0, r1, 256)]
        self.n = len(items)
        self.items = torch.tensor(items, dtype=torch.int32, device=dev)
        nkt = (pm.L + 63) // 64
        self.summary = torch.empty(self.n, nkt, dtype=torch.int16, device=dev)
        self.order = torch.empty(self.n, dtype=torch.int32, device=dev)
        L_.call("vgpt_attn_plan_build", pm.bits.data_ptr(), pm.B, pm.L, self.items.data_ptr(), self.n, self.summary.data_ptr(),
                self.order.data_ptr(), ops._stream())
        self.flags = torch.zeros(4 * self.n * n_heads, dtype=torch.int32, device=dev)


def attention_r64(qkv, pm, plan, n_heads, head_dim, out, q_start=0, lse=None):
    """qkv: fused (1, L, 3 * n_heads * head_dim) buffer; out: rows [q_start, L) (absolute-row addressing as ops.attention_qkv_range)."""
    _, L, w = qkv.shape
    hq = n_heads * head_dim
    kq = qkv.data_ptr() + hq * 2
    vq = kq + hq * 2
    rc = xlib.vgpt_x_attn_fwd_r64(qkv.data_ptr(), kq, vq, out.data_ptr() - q_start * hq * 2, lse, pm.bits.data_ptr(),
                                  plan.items.data_ptr(), plan.summary.data_ptr(), plan.order.data_ptr(), plan.n, plan.flags.data_ptr(),
                                  1, L, n_heads, n_heads, head_dim, L * w, head_dim, w, L * w, head_dim, w, L * w, head_dim, w,
                                  L * hq, head_dim, hq, 1.0 / head_dim ** 0.5, ops._stream())
    if rc:
        raise RuntimeError(f"vgpt_x_attn_fwd_r64: {rc}")
    return out
