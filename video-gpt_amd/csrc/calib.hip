// Box calibration launches for bench.py's `calibration` object: what THIS device sustains right now, measured in the
// benchmark process next to the timed region, so that a step time can be attributed to the box or to the code
// (MI355X_MICROARCH.md, DVFS give-back item 5: devices differ by up to 12 % on an MFMA-dense loop).
//   vgpt_calib_mfma : 256 workgroups x 8 waves, nothing but mfma_f32_16x16x32_bf16 on register operands drawn from a
//                     per-lane hash in [-1, 1): the rate the matrix pipes hold on random data at the clock the part grants
//   vgpt_calib_copy : 16-byte-per-lane streaming copy (read n + write n bytes): the HBM rate
// Neither is on the product path (the sampler never calls them).
#include "common.h"

namespace {

__device__ __forceinline__ float hash_unit(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return (float)(x >> 8) * (1.0f / 8388608.0f) - 1.0f;
}

__global__ __launch_bounds__(512, 1) void calib_mfma_kernel(float* out, int iters) {
    const uint32_t id = blockIdx.x * 512u + threadIdx.x;
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            a[i][j] = (bf16)hash_unit(id * 67u + i * 8 + j);
            b[i][j] = (bf16)hash_unit(id * 131u + i * 8 + j + 7919u);
        }
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 16; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
    float total = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) total += acc[i][0] + acc[i][3];
    if (total == 123.456f) out[0] = total;  // never true in practice: keeps the loop alive
}

__global__ __launch_bounds__(256) void calib_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int64_t n16) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}

}  // namespace

VGPT_EXPORT int vgpt_calib_mfma(float* out, int iters, void* stream) {
    VGPT_REQUIRE(out && iters > 0 && iters <= (1 << 24), VGPT_ERR_INVALID, "vgpt_calib_mfma: bad arguments");
    hipLaunchKernelGGL(calib_mfma_kernel, dim3(256), dim3(512), 0, (hipStream_t)stream, out, iters);
    VGPT_CHECK_LAUNCH("vgpt_calib_mfma");
    return VGPT_OK;
}

VGPT_EXPORT double vgpt_calib_mfma_flops(int iters) {
    // 256 workgroups x 8 waves x iters x 16 MFMAs of 2 * 16 * 16 * 32 FLOP
    return 256.0 * 8.0 * (double)iters * 16.0 * (2.0 * 16 * 16 * 32);
}

VGPT_EXPORT int vgpt_calib_copy(const void* src, void* dst, int64_t n_bytes, void* stream) {
    VGPT_REQUIRE(src && dst && n_bytes > 0 && n_bytes % 16 == 0, VGPT_ERR_INVALID,
                 "vgpt_calib_copy: needs non-null pointers and a positive multiple of 16 bytes");
    VGPT_REQUIRE((((uintptr_t)src | (uintptr_t)dst) & 15) == 0, VGPT_ERR_INVALID, "vgpt_calib_copy: 16-byte alignment");
    hipLaunchKernelGGL(calib_copy_kernel, dim3(256 * 16), dim3(256), 0, (hipStream_t)stream, (const uint4*)src, (uint4*)dst,
                       n_bytes / 16);
    VGPT_CHECK_LAUNCH("vgpt_calib_copy");
    return VGPT_OK;
}
