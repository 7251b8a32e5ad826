// Probe (GPU): sustained bf16 MFMA rate of the two shapes under a full-chip load (256 workgroups x 8 waves, two waves per
// SIMD, random operands in registers, nothing but MFMAs in the loop): what clock the part holds on each.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_shape_rate mfma_shape_rate.hip && ./mfma_shape_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(512, 1) void k(const float* seed, float* out, int iters) {
    const int lane = threadIdx.x;
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) {
            a[i][j] = (__bf16)seed[(lane * 37 + i * 8 + j) & 4095];
            b[i][j] = (__bf16)seed[(lane * 53 + i * 8 + j + 1000) & 4095];
        }
    float total = 0.f;
    if (SHAPE == 16) {
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
        for (int i = 0; i < 16; ++i) total += acc[i][0] + acc[i][3];
    } else {
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i)
            for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[(i >> 2) & 1], acc[i], 0, 0, 0);
        for (int i = 0; i < 8; ++i) total += acc[i][0] + acc[i][15];
    }
    if (total == 123.456f) out[0] = total;
}

int main() {
    float *seed, *out;
    hipMalloc(&seed, 4096 * 4); hipMalloc(&out, 4);
    float h[4096];
    unsigned s = 12345;
    for (int i = 0; i < 4096; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) & 0xffff) / 32768.0f - 1.0f; }
    hipMemcpy(seed, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep)
        for (int shape : {16, 32}) {
            const int iters = 20000;
            if (shape == 16) hipLaunchKernelGGL(k<16>, dim3(256), dim3(512), 0, 0, seed, out, 200);
            else hipLaunchKernelGGL(k<32>, dim3(256), dim3(512), 0, 0, seed, out, 200);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            if (shape == 16) hipLaunchKernelGGL(k<16>, dim3(256), dim3(512), 0, 0, seed, out, iters);
            else hipLaunchKernelGGL(k<32>, dim3(256), dim3(512), 0, 0, seed, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = 256.0 * 8 * iters * (shape == 16 ? 16 * (2.0 * 16 * 16 * 32) : 8 * (2.0 * 32 * 32 * 16));
            printf("mfma_f32_%s_bf16: %.2f ms, %.0f TFLOP/s\n", shape == 16 ? "16x16x32" : "32x32x16", ms, flops / ms / 1e9);
        }
    return 0;
}
