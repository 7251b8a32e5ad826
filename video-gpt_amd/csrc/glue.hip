// Model-glue kernels around the transformer: token-embedding gather, patch embedding with the
// cropped 2-D sincos position table, timestep embedding, small-M Linear, final adaLN layer with
// unpatchify, and the Euler / CFG sampler update. All HBM- or latency-bound; written so the whole
// denoise step is a fixed sequence of launches that can be captured into one hipGraph.
//
// Reference: LVM/model.py:22-83 (modulate, TimestepEmbedder, FinalLayer), :138-154 (PatchEmbedMR),
// :255-327 (unpatchify, cropped_pos_embed, patch_multiple_resolutions), :399-501 (frame_block_forward),
// LVM/scheduler.py:161-208 (Euler loop, x1->v, CFG).
#include "common.h"

namespace {

// ---- embed_tokens gather -------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_gather_kernel(const int64_t* __restrict__ ids,
                                                           const bf16* __restrict__ table,
                                                           bf16* __restrict__ out, int64_t rows,
                                                           int H, int64_t vocab) {
    const int cpr = H >> 3;
    const int64_t total = rows * cpr;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / cpr;
        const int c = (int)(i % cpr);
        int64_t id = ids[row];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        *reinterpret_cast<bf16x8*>(out + row * H + c * 8) =
            *reinterpret_cast<const bf16x8*>(table + id * H + c * 8);
    }
}

// ---- patch embed: Conv2d(C -> H, k=2, s=2) == 16-wide dot per (token, channel) ---------------
// grid.x = token tiles of 16 tokens, grid.y = frames; each thread owns 4 output channels for
// every token of the tile (weights stay in registers across the 16 tokens).
__global__ __launch_bounds__(256) void patch_embed_kernel(
    const bf16* __restrict__ x, const bf16* __restrict__ Wp, const bf16* __restrict__ bias,
    const bf16* __restrict__ pos, const int32_t* __restrict__ dst_row, bf16* __restrict__ seq,
    int C, int h, int w, int H, int pos_max) {
    __shared__ float patch[16][16];
    const int f = blockIdx.y;
    const int h2 = h >> 1, w2 = w >> 1;
    const int ntok = h2 * w2;
    const int t0 = blockIdx.x * 16;
    // stage the 16 patches (16 values each): thread -> (token, element)
    {
        const int tl = threadIdx.x >> 4, e = threadIdx.x & 15;
        const int t = t0 + tl;
        float v = 0.f;
        if (t < ntok) {
            const int ci = e >> 2, p = (e >> 1) & 1, q = e & 1;  // weight layout (H, C, 2, 2)
            const int i = t / w2, j = t % w2;
            v = bf2f(x[(((int64_t)f * C + ci) * h + 2 * i + p) * w + 2 * j + q]);
        }
        patch[tl][e] = v;
    }
    __syncthreads();
    const int top = (pos_max - h2) / 2, left = (pos_max - w2) / 2;
    const int64_t row0 = dst_row[f];
    for (int c0 = threadIdx.x * 4; c0 < H; c0 += 256 * 4) {
        float wreg[4][16];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const bf16x8 lo = *reinterpret_cast<const bf16x8*>(Wp + (int64_t)(c0 + cc) * 16);
            const bf16x8 hi = *reinterpret_cast<const bf16x8*>(Wp + (int64_t)(c0 + cc) * 16 + 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                wreg[cc][e] = bf2f(lo[e]);
                wreg[cc][8 + e] = bf2f(hi[e]);
            }
        }
        const bf16x4 bv = *reinterpret_cast<const bf16x4*>(bias + c0);
        for (int tl = 0; tl < 16; ++tl) {
            const int t = t0 + tl;
            if (t >= ntok) break;
            const int i = t / w2, j = t % w2;
            const bf16x4 pv = *reinterpret_cast<const bf16x4*>(
                pos + ((int64_t)(top + i) * pos_max + left + j) * H + c0);
            bf16x4 o;
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                float acc = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc += wreg[cc][e] * patch[tl][e];
                o[cc] = f2bf(acc + bf2f(bv[cc]) + bf2f(pv[cc]));
            }
            *reinterpret_cast<bf16x4*>(seq + (row0 + t) * H + c0) = o;
        }
    }
}

// ---- timestep sinusoid ------------------------------------------------------------------------
__global__ void timestep_sinusoid_kernel(const float* __restrict__ t, const float* __restrict__ freqs,
                                         bf16* __restrict__ out, int n, int half) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * half) return;
    const int m = idx / half, i = idx % half;
    const float arg = t[m] * freqs[i];
    out[(int64_t)m * 2 * half + i] = f2bf(cosf(arg));
    out[(int64_t)m * 2 * half + half + i] = f2bf(sinf(arg));
}

// ---- small-M Linear: one wave per output column, W row streamed once ---------------------------
template <int MT>
__global__ __launch_bounds__(256) void linear_small_kernel(
    const bf16* __restrict__ x, const bf16* __restrict__ W, const bf16* __restrict__ bias,
    bf16* __restrict__ out, const int32_t* __restrict__ out_row, int M, int N, int K, int64_t ldx,
    int64_t ldo, int pre_act, int post_act) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    for (int n = blockIdx.x * 4 + wave; n < N; n += gridDim.x * 4) {
        float acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = 0.f;
        const bf16* wr = W + (int64_t)n * K;
        for (int k = lane * 8; k < K; k += 512) {
            const bf16x8 wv = *reinterpret_cast<const bf16x8*>(wr + k);
            float wf[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) wf[j] = bf2f(wv[j]);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                if (m < M) {
                    const bf16x8 xv = *reinterpret_cast<const bf16x8*>(x + (int64_t)m * ldx + k);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        float xf = bf2f(xv[j]);
                        if (pre_act != VGPT_ACT_NONE) xf = bf2f(f2bf(act_apply(xf, pre_act)));
                        acc[m] += wf[j] * xf;
                    }
                }
            }
        }
        const float bn = bias ? bf2f(bias[n]) : 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            if (m < M) {
                float v = wave_sum(acc[m]) + bn;
                v = act_apply(v, post_act);
                if (lane == 0) {
                    const int64_t r = out_row ? (int64_t)out_row[m] : (int64_t)m;
                    out[r * ldo + n] = f2bf(v);
                }
            }
        }
    }
}

// ---- final layer: LayerNorm(no affine) + modulate + Linear(H -> 16) + unpatchify ---------------
// one wave per token; NCH = 512-element slabs cached in registers
template <int NCH>
__global__ __launch_bounds__(256) void final_layer_kernel(
    const bf16* __restrict__ hidden, const int32_t* __restrict__ src_row, const bf16* __restrict__ mod,
    const bf16* __restrict__ Wf, const bf16* __restrict__ bfin, bf16* __restrict__ out, int C, int h,
    int w, int H, float eps) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int f = blockIdx.y;
    const int h2 = h >> 1, w2 = w >> 1;
    const int ntok = h2 * w2;
    const int t = blockIdx.x * 4 + wave;
    if (t >= ntok) return;
    const bf16* xr = hidden + ((int64_t)src_row[f] + t) * H;
    const bf16* shift = mod + (int64_t)f * 2 * H;
    const bf16* scale = shift + H;
    float v[NCH][8];
    float s1 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int off = (c * 64 + lane) * 8;
        if (off < H) {
            const bf16x8 xv = *reinterpret_cast<const bf16x8*>(xr + off);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[c][j] = bf2f(xv[j]);
                s1 += v[c][j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[c][j] = 0.f;
        }
    }
    const float mean = wave_sum(s1) / (float)H;
    float s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int off = (c * 64 + lane) * 8;
        if (off < H) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = v[c][j] - mean;
                s2 += d * d;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(s2) / (float)H + eps);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int off = (c * 64 + lane) * 8;
        if (off < H) {
            const bf16x8 sh = *reinterpret_cast<const bf16x8*>(shift + off);
            const bf16x8 sc = *reinterpret_cast<const bf16x8*>(scale + off);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                v[c][j] = (v[c][j] - mean) * rstd * (1.0f + bf2f(sc[j])) + bf2f(sh[j]);
        }
    }
    // 16 outputs: y[o] = sum_k v[k] Wf[o][k] + b[o]
    float y = 0.f;  // lane o (< 16) keeps output o
    for (int o = 0; o < 16; ++o) {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int off = (c * 64 + lane) * 8;
            if (off < H) {
                const bf16x8 wv = *reinterpret_cast<const bf16x8*>(Wf + (int64_t)o * H + off);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc += v[c][j] * bf2f(wv[j]);
            }
        }
        acc = wave_sum(acc);
        if (lane == o) y = acc + bf2f(bfin[o]);
    }
    if (lane < 16) {
        // unpatchify: out[f][c][2i+p][2j+q] = y[(p*2+q)*C + c]
        const int c = lane % C, pq = lane / C;
        const int p = pq >> 1, q = pq & 1;
        const int i = t / w2, j = t % w2;
        out[(((int64_t)f * C + c) * h + 2 * i + p) * w + 2 * j + q] = f2bf(y);
    }
}

// ---- sampler -----------------------------------------------------------------------------------
// `n_steps` = entries of the sigma table - 1: a replay past the end of the table (step >= n_steps) is a no-op instead of
// an out-of-bounds read of sigma[step + 1]
__global__ void set_timesteps_kernel(const float* sigma, const int32_t* step, float* ts, int n, int n_steps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int st = *step;
    if (i < n && st >= 0 && st <= n_steps) ts[i] = sigma[st];
}

__global__ __launch_bounds__(256) void euler_cfg_kernel(float* __restrict__ z, bf16* __restrict__ zm,
                                                        const bf16* __restrict__ pred,
                                                        const float* __restrict__ sigma,
                                                        const int32_t* __restrict__ step, int n_steps, int n_frames,
                                                        int64_t elems, int pred_type, int use_cfg,
                                                        float cfg_scale) {
    const int st = *step;
    if (st < 0 || st >= n_steps) return;   // past the sigma table: leave the state untouched
    const float sg = sigma[st], sg_next = sigma[st + 1];
    const float dt = sg_next - sg;
    const float inv = 1.0f / (1.0f - sg);
    const int half = use_cfg ? n_frames / 2 : n_frames;
    const int64_t total = (int64_t)half * elems;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        float zc = z[i];
        float vc = bf2f(pred[i]);
        if (pred_type == VGPT_PRED_X1) vc = (vc - zc) * inv;
        if (use_cfg) {
            const int64_t iu = i + total;
            float zu = z[iu];
            float vu = bf2f(pred[iu]);
            if (pred_type == VGPT_PRED_X1) vu = (vu - zu) * inv;
            vc = vu + cfg_scale * (vc - vu);
            zu += dt * vc;
            z[iu] = zu;
            zm[iu] = f2bf(zu);
        }
        zc += dt * vc;
        z[i] = zc;
        zm[i] = f2bf(zc);
    }
}

__global__ void advance_kernel(int32_t* step) { *step += 1; }

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ src,
                                                            bf16* __restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = f2bf(src[i]);
}

}  // namespace

VGPT_EXPORT int vgpt_embed_gather(const int64_t* ids, const void* table, void* out, int64_t rows,
                                  int64_t H, int64_t vocab, void* stream) {
    VGPT_REQUIRE(ids && table && out, VGPT_ERR_INVALID, "vgpt_embed_gather: null pointer");
    VGPT_REQUIRE(rows >= 0 && H > 0 && vocab > 0, VGPT_ERR_INVALID, "vgpt_embed_gather: bad shape");
    VGPT_REQUIRE(H % 8 == 0, VGPT_ERR_UNSUPPORTED, "vgpt_embed_gather: H must be a multiple of 8");
    if (rows == 0) return VGPT_OK;
    int grid = (int)std::min<int64_t>(cdiv(rows * (H / 8), 256), (int64_t)1 << 30);
    hipLaunchKernelGGL(embed_gather_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, ids,
                       (const bf16*)table, (bf16*)out, rows, (int)H, vocab);
    VGPT_CHECK_LAUNCH("vgpt_embed_gather");
    return VGPT_OK;
}

VGPT_EXPORT int vgpt_patch_embed_fwd(const void* x, const void* Wp, const void* bias,
                                     const void* pos_embed, const int32_t* dst_row, void* seq,
                                     int n_frames, int C, int h, int w, int64_t H, int pos_max,
                                     void* stream) {
    VGPT_REQUIRE(x && Wp && bias && pos_embed && dst_row && seq, VGPT_ERR_INVALID,
                 "vgpt_patch_embed_fwd: null pointer");
    VGPT_REQUIRE(n_frames >= 0 && C > 0 && h > 0 && w > 0 && H > 0, VGPT_ERR_INVALID,
                 "vgpt_patch_embed_fwd: bad shape");
    VGPT_REQUIRE(C * 4 == 16 && h % 2 == 0 && w % 2 == 0, VGPT_ERR_UNSUPPORTED,
                 "vgpt_patch_embed_fwd: needs in_channels=4, patch 2, even latent size");
    VGPT_REQUIRE(H % 4 == 0, VGPT_ERR_UNSUPPORTED, "vgpt_patch_embed_fwd: H must be a multiple of 4");
    VGPT_REQUIRE(h / 2 <= pos_max && w / 2 <= pos_max, VGPT_ERR_INVALID,
                 "vgpt_patch_embed_fwd: latent larger than pos_embed_max_size");
    if (n_frames == 0) return VGPT_OK;
    const int ntok = (h / 2) * (w / 2);
    hipLaunchKernelGGL(patch_embed_kernel, dim3((unsigned)cdiv(ntok, 16), n_frames), dim3(256), 0,
                       (hipStream_t)stream, (const bf16*)x, (const bf16*)Wp, (const bf16*)bias,
                       (const bf16*)pos_embed, dst_row, (bf16*)seq, C, h, w, (int)H, pos_max);
    VGPT_CHECK_LAUNCH("vgpt_patch_embed_fwd");
    return VGPT_OK;
}

VGPT_EXPORT int vgpt_timestep_sinusoid(const float* t, const float* freqs, void* out, int n,
                                       int half, void* stream) {
    VGPT_REQUIRE(t && freqs && out, VGPT_ERR_INVALID, "vgpt_timestep_sinusoid: null pointer");
    VGPT_REQUIRE(n >= 0 && half > 0, VGPT_ERR_INVALID, "vgpt_timestep_sinusoid: bad shape");
    if (n == 0) return VGPT_OK;
    hipLaunchKernelGGL(timestep_sinusoid_kernel, dim3((unsigned)cdiv((int64_t)n * half, 256)),
                       dim3(256), 0, (hipStream_t)stream, t, freqs, (bf16*)out, n, half);
    VGPT_CHECK_LAUNCH("vgpt_timestep_sinusoid");
    return VGPT_OK;
}

VGPT_EXPORT int vgpt_linear_small(const void* x, const void* W, const void* bias, void* out,
                                  const int32_t* out_row, int M, int64_t N, int64_t K, int64_t ldx,
                                  int64_t ldo, int pre_act, int post_act, void* stream) {
    VGPT_REQUIRE(x && W && out, VGPT_ERR_INVALID, "vgpt_linear_small: null pointer");
    VGPT_REQUIRE(M >= 0 && N > 0 && K > 0, VGPT_ERR_INVALID, "vgpt_linear_small: bad shape");
    VGPT_REQUIRE(M <= 32, VGPT_ERR_UNSUPPORTED, "vgpt_linear_small: M=%d > 32", M);
    VGPT_REQUIRE(K % 8 == 0 && ldx % 8 == 0, VGPT_ERR_UNSUPPORTED,
                 "vgpt_linear_small: K and ldx must be multiples of 8");
    VGPT_REQUIRE(N < (1ll << 31) && K < (1ll << 31), VGPT_ERR_UNSUPPORTED,
                 "vgpt_linear_small: dimension too large");
    if (M == 0) return VGPT_OK;
    int grid = (int)std::min<int64_t>(cdiv(N, 4), 256 * 8);
    hipStream_t s = (hipStream_t)stream;
#define LS_CASE(MT)                                                                               \
    hipLaunchKernelGGL(linear_small_kernel<MT>, dim3(grid), dim3(256), 0, s, (const bf16*)x,       \
                       (const bf16*)W, (const bf16*)bias, (bf16*)out, out_row, M, (int)N, (int)K, \
                       ldx, ldo, pre_act, post_act)
    if (M <= 8) LS_CASE(8);
    else if (M <= 16) LS_CASE(16);
    else LS_CASE(32);
#undef LS_CASE
    VGPT_CHECK_LAUNCH("vgpt_linear_small");
    return VGPT_OK;
}

VGPT_EXPORT int vgpt_final_layer_fwd(const void* hidden, const int32_t* src_row, const void* mod,
                                     const void* Wf, const void* bf, void* out, int n_frames, int C,
                                     int h, int w, int64_t H, float eps, void* stream) {
    VGPT_REQUIRE(hidden && src_row && mod && Wf && bf && out, VGPT_ERR_INVALID,
                 "vgpt_final_layer_fwd: null pointer");
    VGPT_REQUIRE(n_frames >= 0 && C > 0 && h > 0 && w > 0 && H > 0, VGPT_ERR_INVALID,
                 "vgpt_final_layer_fwd: bad shape");
    VGPT_REQUIRE(C * 4 == 16 && h % 2 == 0 && w % 2 == 0, VGPT_ERR_UNSUPPORTED,
                 "vgpt_final_layer_fwd: needs out_channels=4, patch 2, even latent size");
    VGPT_REQUIRE(H % 8 == 0 && H <= 8 * 512, VGPT_ERR_UNSUPPORTED,
                 "vgpt_final_layer_fwd: H must be a multiple of 8 and <= 4096");
    if (n_frames == 0) return VGPT_OK;
    const int ntok = (h / 2) * (w / 2);
    dim3 grid((unsigned)cdiv(ntok, 4), n_frames);
    hipStream_t s = (hipStream_t)stream;
#define FL_CASE(N)                                                                              \
    case N:                                                                                     \
        hipLaunchKernelGGL(final_layer_kernel<N>, grid, dim3(256), 0, s, (const bf16*)hidden,   \
                           src_row, (const bf16*)mod, (const bf16*)Wf, (const bf16*)bf,         \
                           (bf16*)out, C, h, w, (int)H, eps);                                   \
        break;
    switch ((int)cdiv(H, 512)) {
        FL_CASE(1) FL_CASE(2) FL_CASE(3) FL_CASE(4) FL_CASE(5) FL_CASE(6) FL_CASE(7) FL_CASE(8)
    }
#undef FL_CASE
    VGPT_CHECK_LAUNCH("vgpt_final_layer_fwd");
    return VGPT_OK;
}

VGPT_EXPORT int vgpt_sampler_set_timesteps(const float* sigma, const int32_t* step, int n_steps, float* timesteps,
                                           int n, void* stream) {
    VGPT_REQUIRE(sigma && step && timesteps, VGPT_ERR_INVALID,
                 "vgpt_sampler_set_timesteps: null pointer");
    VGPT_REQUIRE(n >= 0 && n_steps > 0, VGPT_ERR_INVALID, "vgpt_sampler_set_timesteps: bad shape");
    if (n == 0) return VGPT_OK;
    hipLaunchKernelGGL(set_timesteps_kernel, dim3((unsigned)cdiv(n, 64)), dim3(64), 0,
                       (hipStream_t)stream, sigma, step, timesteps, n, n_steps);
    VGPT_CHECK_LAUNCH("vgpt_sampler_set_timesteps");
    return VGPT_OK;
}

VGPT_EXPORT int vgpt_euler_cfg_update(float* z, void* z_model, const void* pred, const float* sigma,
                                      const int32_t* step, int n_steps, int n_frames, int64_t elems, int pred_type,
                                      int use_cfg, float cfg_scale, void* stream) {
    VGPT_REQUIRE(z && z_model && pred && sigma && step, VGPT_ERR_INVALID,
                 "vgpt_euler_cfg_update: null pointer");
    VGPT_REQUIRE(n_frames >= 0 && elems >= 0 && n_steps > 0, VGPT_ERR_INVALID, "vgpt_euler_cfg_update: bad shape");
    VGPT_REQUIRE(pred_type == VGPT_PRED_V || pred_type == VGPT_PRED_X1, VGPT_ERR_INVALID,
                 "vgpt_euler_cfg_update: unknown prediction type %d", pred_type);
    VGPT_REQUIRE(!use_cfg || n_frames % 2 == 0, VGPT_ERR_INVALID,
                 "vgpt_euler_cfg_update: CFG needs an even number of frames");
    if (n_frames == 0 || elems == 0) return VGPT_OK;
    const int64_t total = (int64_t)(use_cfg ? n_frames / 2 : n_frames) * elems;
    int grid = (int)std::min<int64_t>(cdiv(total, 256), 2048);
    hipLaunchKernelGGL(euler_cfg_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, z,
                       (bf16*)z_model, (const bf16*)pred, sigma, step, n_steps, n_frames, elems, pred_type,
                       use_cfg, cfg_scale);
    VGPT_CHECK_LAUNCH("vgpt_euler_cfg_update");
    return VGPT_OK;
}

// dst[l][0:slab] = src[*step][l][0:slab] for every layer l, in 16-byte units (the time-token rows of the current step)
__global__ __launch_bounds__(256) void copy_step_slab_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst,
                                                             const int32_t* __restrict__ step, int n_steps, int64_t slab,
                                                             int64_t src_step_stride, int64_t src_layer_stride,
                                                             int64_t dst_layer_stride, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int s = min(max(*step, 0), n_steps - 1);
    const int64_t l = i / slab, o = i % slab;
    dst[l * dst_layer_stride + o] = src[s * src_step_stride + l * src_layer_stride + o];
}

VGPT_EXPORT int vgpt_sampler_copy_step_rows(const void* src, void* dst, const int32_t* step, int n_steps, int n_layers,
                                            int64_t slab_bytes, int64_t src_step_stride_bytes,
                                            int64_t src_layer_stride_bytes, int64_t dst_layer_stride_bytes, void* stream) {
    VGPT_REQUIRE(src && dst && step, VGPT_ERR_INVALID, "vgpt_sampler_copy_step_rows: null pointer");
    VGPT_REQUIRE(n_steps > 0 && n_layers >= 0 && slab_bytes >= 0, VGPT_ERR_INVALID, "vgpt_sampler_copy_step_rows: bad shape");
    VGPT_REQUIRE(((slab_bytes | src_step_stride_bytes | src_layer_stride_bytes | dst_layer_stride_bytes) & 15) == 0 &&
                     (((uintptr_t)src | (uintptr_t)dst) & 15) == 0,
                 VGPT_ERR_UNSUPPORTED, "vgpt_sampler_copy_step_rows: sizes, strides and pointers must be multiples of 16 bytes");
    const int64_t slab = slab_bytes / 16, total = slab * n_layers;
    if (total == 0) return VGPT_OK;
    hipLaunchKernelGGL(copy_step_slab_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint4*)src, (uint4*)dst, step, n_steps, slab, src_step_stride_bytes / 16,
                       src_layer_stride_bytes / 16, dst_layer_stride_bytes / 16, total);
    VGPT_CHECK_LAUNCH("vgpt_sampler_copy_step_rows");
    return VGPT_OK;
}

VGPT_EXPORT int vgpt_sampler_advance(int32_t* step, void* stream) {
    VGPT_REQUIRE(step, VGPT_ERR_INVALID, "vgpt_sampler_advance: null pointer");
    hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step);
    VGPT_CHECK_LAUNCH("vgpt_sampler_advance");
    return VGPT_OK;
}

VGPT_EXPORT int vgpt_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream) {
    VGPT_REQUIRE(src && dst, VGPT_ERR_INVALID, "vgpt_cast_f32_to_bf16: null pointer");
    VGPT_REQUIRE(n >= 0, VGPT_ERR_INVALID, "vgpt_cast_f32_to_bf16: bad shape");
    if (n == 0) return VGPT_OK;
    int grid = (int)std::min<int64_t>(cdiv(n, 256), 2048);
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src,
                       (bf16*)dst, n);
    VGPT_CHECK_LAUNCH("vgpt_cast_f32_to_bf16");
    return VGPT_OK;
}
