"""Developer probe (GPU): torch's scaled_dot_product_attention (the ROCm build's flash kernel) on the two dense problems a
hoisted cfg-2 step's attention consists of -- the image rows of a sequence see every key of their sequence, so there is no
mask: conditional 2048 queries x 3096 keys, unconditional 2048 x 2064, 32 heads x 96 -- beside the planned block-mask kernel's
in-step 149 us.  A yardstick only; nothing in the product calls it."""
import torch, torch.nn.functional as F
dev = "cuda:0"; BF = torch.bfloat16
def timeit(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
tot = 0.0
for (Lq, Lk) in ((2048, 3096), (2048, 2064)):
    q = torch.randn(1, 32, Lq, 96, device=dev).to(BF); k = torch.randn(1, 32, Lk, 96, device=dev).to(BF); v = torch.randn_like(k)
    for backend in ("flash", "efficient", "math"):
        try:
            from torch.nn.attention import sdpa_kernel, SDPBackend
            b = {"flash": SDPBackend.FLASH_ATTENTION, "efficient": SDPBackend.EFFICIENT_ATTENTION, "math": SDPBackend.MATH}[backend]
            with sdpa_kernel(b):
                t = timeit(lambda: F.scaled_dot_product_attention(q, k, v))
            fl = 4.0 * 32 * Lq * Lk * 96
            print(f"Lq {Lq} Lk {Lk} {backend:9s} {t:7.1f} us  {fl / t / 1e6:6.0f} TFLOP/s")
        except Exception as ex:
            print(f"Lq {Lq} Lk {Lk} {backend}: {type(ex).__name__} {str(ex)[:100]}")
