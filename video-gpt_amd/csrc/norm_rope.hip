// RMSNorm and rotary embedding kernels (HBM-bound; wave64 shuffle reductions, 16-byte accesses).
//
// Reference arithmetic:
//   Phi3RMSNorm.forward (transformers==4.47.1 modeling_phi3.py; used at OmniGen/transformer.py:196-214)
//   Phi3RotaryEmbedding.forward + apply_rotary_pos_emb (LVM/transform/sdpa_transform.py:52-53)
#include "common.h"

// One wave per row. NCH = number of 512-element slabs cached in registers (H <= 512*NCH).
template <int NCH>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const bf16* __restrict__ x,
                                                      const bf16* __restrict__ w,
                                                      bf16* __restrict__ y, int64_t rows, int H,
                                                      float eps) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t wave_stride = (int64_t)gridDim.x * 4;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += wave_stride) {
        const bf16* xr = x + row * H;
        bf16x8 v[NCH];
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int off = (c * 64 + lane) * 8;
            if (off < H) {
                v[c] = *reinterpret_cast<const bf16x8*>(xr + off);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float f = bf2f(v[c][j]);
                    ss += f * f;
                }
            }
        }
        ss = wave_sum(ss);
        const float rstd = rsqrtf(ss / (float)H + eps);
        bf16* yr = y + row * H;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int off = (c * 64 + lane) * 8;
            if (off < H) {
                bf16x8 wv = *reinterpret_cast<const bf16x8*>(w + off);
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    // HF: hidden.to(input_dtype) then weight * hidden  -> two roundings
                    bf16 n = f2bf(bf2f(v[c][j]) * rstd);
                    o[j] = f2bf(bf2f(wv[j]) * bf2f(n));
                }
                *reinterpret_cast<bf16x8*>(yr + off) = o;
            }
        }
    }
}

// Generic H (any multiple of 8): re-reads the row for the second pass (L2 hit).
__global__ __launch_bounds__(256) void rmsnorm_kernel_big(const bf16* __restrict__ x,
                                                          const bf16* __restrict__ w,
                                                          bf16* __restrict__ y, int64_t rows, int H,
                                                          float eps) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t wave_stride = (int64_t)gridDim.x * 4;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += wave_stride) {
        const bf16* xr = x + row * H;
        float ss = 0.f;
        for (int off = lane * 8; off < H; off += 512) {
            bf16x8 v = *reinterpret_cast<const bf16x8*>(xr + off);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float f = bf2f(v[j]);
                ss += f * f;
            }
        }
        ss = wave_sum(ss);
        const float rstd = rsqrtf(ss / (float)H + eps);
        bf16* yr = y + row * H;
        for (int off = lane * 8; off < H; off += 512) {
            bf16x8 v = *reinterpret_cast<const bf16x8*>(xr + off);
            bf16x8 wv = *reinterpret_cast<const bf16x8*>(w + off);
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                bf16 n = f2bf(bf2f(v[j]) * rstd);
                o[j] = f2bf(bf2f(wv[j]) * bf2f(n));
            }
            *reinterpret_cast<bf16x8*>(yr + off) = o;
        }
    }
}

VGPT_EXPORT int vgpt_rmsnorm_fwd(const void* x, const void* w, void* y, int64_t rows, int64_t H,
                                 float eps, void* stream) {
    VGPT_REQUIRE(x && w && y, VGPT_ERR_INVALID, "vgpt_rmsnorm_fwd: null pointer");
    VGPT_REQUIRE(rows >= 0 && H > 0, VGPT_ERR_INVALID, "vgpt_rmsnorm_fwd: bad shape");
    VGPT_REQUIRE(H % 8 == 0, VGPT_ERR_UNSUPPORTED, "vgpt_rmsnorm_fwd: H=%ld not a multiple of 8",
                 (long)H);
    if (rows == 0) return VGPT_OK;
    hipStream_t s = (hipStream_t)stream;
    int grid = (int)std::min<int64_t>(cdiv(rows, 4), 256 * 8);
    const bf16* xp = (const bf16*)x;
    const bf16* wp = (const bf16*)w;
    bf16* yp = (bf16*)y;
    const int nch = (int)cdiv(H, 512);
#define RMS_CASE(N)                                                                    \
    case N:                                                                            \
        hipLaunchKernelGGL(rmsnorm_kernel<N>, dim3(grid), dim3(256), 0, s, xp, wp, yp, \
                           rows, (int)H, eps);                                         \
        break;
    switch (nch) {
        RMS_CASE(1) RMS_CASE(2) RMS_CASE(3) RMS_CASE(4) RMS_CASE(5) RMS_CASE(6) RMS_CASE(7)
        RMS_CASE(8)
        default:
            hipLaunchKernelGGL(rmsnorm_kernel_big, dim3(grid), dim3(256), 0, s, xp, wp, yp, rows,
                               (int)H, eps);
    }
#undef RMS_CASE
    VGPT_CHECK_LAUNCH("vgpt_rmsnorm_fwd");
    return VGPT_OK;
}

// ---------------------------------------------------------------------------------------------

__global__ void rope_table_kernel(const int64_t* __restrict__ pos, const float* __restrict__ inv_freq,
                                  float* __restrict__ cos_o, float* __restrict__ sin_o,
                                  int64_t tokens, int half, int round_bf16, float scale) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= tokens * half) return;
    const int64_t t = idx / half;
    const int i = (int)(idx % half);
    const float ang = (float)pos[t] * inv_freq[i];
    float c = cosf(ang) * scale, s = sinf(ang) * scale;   // longrope attention factor (1 for plain RoPE)
    if (round_bf16) {
        c = bf2f(f2bf(c));
        s = bf2f(f2bf(s));
    }
    cos_o[idx] = c;
    sin_o[idx] = s;
}

VGPT_EXPORT int vgpt_rope_table(const int64_t* position_ids, const float* inv_freq, float* cos_out,
                                float* sin_out, int64_t tokens, int half, int round_bf16,
                                float scale, void* stream) {
    VGPT_REQUIRE(position_ids && inv_freq && cos_out && sin_out, VGPT_ERR_INVALID,
                 "vgpt_rope_table: null pointer");
    VGPT_REQUIRE(tokens >= 0 && half > 0, VGPT_ERR_INVALID, "vgpt_rope_table: bad shape");
    if (tokens == 0) return VGPT_OK;
    const int64_t n = tokens * half;
    hipLaunchKernelGGL(rope_table_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0,
                       (hipStream_t)stream, position_ids, inv_freq, cos_out, sin_out, tokens, half,
                       round_bf16, scale);
    VGPT_CHECK_LAUNCH("vgpt_rope_table");
    return VGPT_OK;
}

// One thread rotates 8 (d, d+half) pairs of one head of one token.
__global__ __launch_bounds__(256) void rope_qk_kernel(bf16* __restrict__ qkv,
                                                      const float* __restrict__ cos_t,
                                                      const float* __restrict__ sin_t,
                                                      int64_t tokens, int n_rot_heads, int head_dim,
                                                      int64_t row_stride) {
    const int half = head_dim >> 1;
    const int cpr = half >> 3;  // 8-wide chunks per half head
    const int64_t per_tok = (int64_t)n_rot_heads * cpr;
    const int64_t total = tokens * per_tok;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = idx / per_tok;
        const int r = (int)(idx % per_tok);
        const int head = r / cpr;
        const int c = r % cpr;
        bf16* p = qkv + t * row_stride + (int64_t)head * head_dim + c * 8;
        bf16x8 x1 = *reinterpret_cast<bf16x8*>(p);
        bf16x8 x2 = *reinterpret_cast<bf16x8*>(p + half);
        const float* cp = cos_t + t * half + c * 8;
        const float* sp = sin_t + t * half + c * 8;
        f32x4 c0 = *reinterpret_cast<const f32x4*>(cp), c1 = *reinterpret_cast<const f32x4*>(cp + 4);
        f32x4 s0 = *reinterpret_cast<const f32x4*>(sp), s1 = *reinterpret_cast<const f32x4*>(sp + 4);
        bf16x8 o1, o2;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float cs = j < 4 ? c0[j & 3] : c1[j & 3];
            const float sn = j < 4 ? s0[j & 3] : s1[j & 3];
            const float a = bf2f(x1[j]), b = bf2f(x2[j]);
            // q*cos + rotate_half(q)*sin, rotate_half(x) = [-x2 | x1]
            o1[j] = f2bf(a * cs - b * sn);
            o2[j] = f2bf(b * cs + a * sn);
        }
        *reinterpret_cast<bf16x8*>(p) = o1;
        *reinterpret_cast<bf16x8*>(p + half) = o2;
    }
}

VGPT_EXPORT int vgpt_rope_qk_inplace(void* qkv, const float* cos_t, const float* sin_t,
                                     int64_t tokens, int n_q_heads, int n_kv_heads, int head_dim,
                                     void* stream) {
    VGPT_REQUIRE(qkv && cos_t && sin_t, VGPT_ERR_INVALID, "vgpt_rope_qk_inplace: null pointer");
    VGPT_REQUIRE(tokens >= 0 && n_q_heads > 0 && n_kv_heads > 0 && head_dim > 0, VGPT_ERR_INVALID,
                 "vgpt_rope_qk_inplace: bad shape");
    VGPT_REQUIRE(head_dim % 16 == 0, VGPT_ERR_UNSUPPORTED,
                 "vgpt_rope_qk_inplace: head_dim=%d needs head_dim/2 %% 8 == 0", head_dim);
    if (tokens == 0) return VGPT_OK;
    const int64_t row_stride = (int64_t)(n_q_heads + 2 * n_kv_heads) * head_dim;
    const int n_rot = n_q_heads + n_kv_heads;  // q heads then k heads are contiguous in qkv
    const int64_t total = tokens * n_rot * (head_dim / 16);
    int grid = (int)std::min<int64_t>(cdiv(total, 256), (int64_t)1 << 30);  // one element group per thread streams faster than a capped grid-stride loop
    hipLaunchKernelGGL(rope_qk_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (bf16*)qkv,
                       cos_t, sin_t, tokens, n_rot, head_dim, row_stride);
    VGPT_CHECK_LAUNCH("vgpt_rope_qk_inplace");
    return VGPT_OK;
}

// ---- RMSNorm folded into the GEMMs around it (gemm_bf16.hip: vgpt_gemm_bf16_resid_ssq -> *_prenorm) ----
// 1 / rms of the rows of a bf16 matrix: the statistics of the FIRST norm of a step, whose input no GEMM of ours produced
// (one wave per row, fp32; what the *_prenorm entries read).
__global__ __launch_bounds__(256) void rms_rstd_kernel(const bf16* __restrict__ x, float* __restrict__ out, int64_t rows, int H,
                                                       int64_t ldx, float eps) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t wave_stride = (int64_t)gridDim.x * 4;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += wave_stride) {
        const bf16* xr = x + row * ldx;
        float ss = 0.f;
        for (int off = lane * 8; off < H; off += 512) {
            bf16x8 v = *reinterpret_cast<const bf16x8*>(xr + off);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float f = bf2f(v[j]);
                ss += f * f;
            }
        }
        ss = wave_sum(ss);
        if (lane == 0) out[row] = rsqrtf(ss / (float)H + eps);
    }
}

VGPT_EXPORT int vgpt_rms_rstd(const void* x, float* rstd_out, int64_t rows, int64_t H, int64_t ldx, float eps, void* stream) {
    VGPT_REQUIRE(rows >= 0 && H > 0 && H % 8 == 0 && ldx >= H && ldx % 8 == 0 && eps >= 0.f, VGPT_ERR_INVALID,
                 "vgpt_rms_rstd: H and ldx must be multiples of 8, ldx >= H");
    VGPT_REQUIRE(rows == 0 || (x && rstd_out), VGPT_ERR_INVALID, "vgpt_rms_rstd: null pointer");
    VGPT_REQUIRE(((uintptr_t)x & 15) == 0, VGPT_ERR_UNSUPPORTED, "vgpt_rms_rstd: rows must be 16-byte aligned");
    if (rows == 0) return VGPT_OK;
    const int blocks = (int)((rows + 3) / 4 < 4096 ? (rows + 3) / 4 : 4096);
    hipLaunchKernelGGL(rms_rstd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, rstd_out, rows, (int)H, ldx, eps);
    VGPT_CHECK_LAUNCH("vgpt_rms_rstd");
    return VGPT_OK;
}

// W'[n][k] = bf16(W[n][k] * g[k]): a norm's gain folded into the weight of the Linear behind it (once per checkpoint)
__global__ __launch_bounds__(256) void fold_gain_kernel(const bf16* __restrict__ w, const bf16* __restrict__ g, bf16* __restrict__ out,
                                                        int64_t n_vec, int K) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_vec) return;
    const int k = (int)((i * 8) % K);
    const bf16x8 wv = *reinterpret_cast<const bf16x8*>(w + i * 8);
    const bf16x8 gv = *reinterpret_cast<const bf16x8*>(g + k);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = f2bf(bf2f(wv[j]) * bf2f(gv[j]));
    *reinterpret_cast<bf16x8*>(out + i * 8) = o;
}

VGPT_EXPORT int vgpt_fold_norm_gain(const void* W, const void* gain, void* W_out, int64_t N, int64_t K, void* stream) {
    VGPT_REQUIRE(N >= 0 && K > 0 && K % 8 == 0, VGPT_ERR_INVALID, "vgpt_fold_norm_gain: K must be a multiple of 8");
    VGPT_REQUIRE(N == 0 || (W && gain && W_out), VGPT_ERR_INVALID, "vgpt_fold_norm_gain: null pointer");
    VGPT_REQUIRE((((uintptr_t)W | (uintptr_t)gain | (uintptr_t)W_out) & 15) == 0, VGPT_ERR_UNSUPPORTED,
                 "vgpt_fold_norm_gain: operands must be 16-byte aligned");
    if (N == 0) return VGPT_OK;
    const int64_t n_vec = N * K / 8;
    hipLaunchKernelGGL(fold_gain_kernel, dim3((unsigned)((n_vec + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)W,
                       (const bf16*)gain, (bf16*)W_out, n_vec, (int)K);
    VGPT_CHECK_LAUNCH("vgpt_fold_norm_gain");
    return VGPT_OK;
}
