// VAE encode/decode conv stack (diffusers==0.29.0 AutoencoderKL / sdxl-vae architecture; reference
// call sites LVM/pipeline.py:110-117,558-590, LVM/utils.py:99-137) in fp32, as the reference runs it.
//
// One implicit-GEMM convolution kernel covers every Conv2d / Linear of the VAE:
//   y[n][co][oy][ox] = bias[co] + resid + sum_{ci,dy,dx} w[co][ci][dy][dx] * f(x)[n][ci][iy][ix]
//   - 3x3 stride 1 pad 1, 3x3 stride 2 with the (0,1,0,1) zero pad of Downsample2D, 1x1;
//   - optional nearest x2 upsample folded into the loader (Upsample2D never materialised);
//   - optional GroupNorm(+SiLU) prologue applied while the input patch is staged into LDS
//     (stats from vgpt_groupnorm_stats), so the normalised tensor never goes to HBM;
//   - bias + residual epilogue (resnet skip, attention residual);
//   - "weights" may be an activation tensor (per-image batch stride, optionally stored [k][co]):
//     the mid-block attention's Q K^T and P V products run through the same kernel.
// Math: v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain, 157 TF/s peak) — 64 output channels x 128
// pixels per 4-wave block, K chunked by 8 input channels (3x3) or 64 (1x1), operands staged in LDS.
#include "common.h"

namespace {

struct ConvArgs {
    const float* x; const float* w; const float* bias; const float* resid;
    const float* gn_stats; const float* gn_gamma; const float* gn_beta;
    float* y;
    int N, Cin, Hin, Win, Cout, Hout, Wout;
    int upsample, gn_groups, gn_silu, w_transposed;
    int64_t ldw, w_batch_stride;
    int tiles_x, tiles_y, tiles_co;
};

constexpr int TCO = 64, TH = 4, TW = 32;  // block tile: 64 output channels x (4 x 32) pixels

template <int KS, int STRIDE>
struct ConvCfg {
    static constexpr int CK = (KS == 3) ? 8 : 64;        // input channels per K chunk
    static constexpr int KC = CK * KS * KS;              // K per chunk (72 / 64)
    static constexpr int KPAD = KC + 1;                  // weight row stride in LDS (bank spread)
    static constexpr int PH = (TH - 1) * STRIDE + KS, PW = (TW - 1) * STRIDE + KS;
    static constexpr int W_FLOATS = TCO * KPAD, P_FLOATS = CK * PH * PW;
    static constexpr int LDS_BYTES = (W_FLOATS + P_FLOATS) * 4;
};

template <int KS, int STRIDE>
__global__ __launch_bounds__(256) void conv_kernel(ConvArgs a) {
    using C = ConvCfg<KS, STRIDE>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sW = reinterpret_cast<float*>(smem);
    float* sP = sW + C::W_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    int bid = blockIdx.x;
    const int tx = bid % a.tiles_x; bid /= a.tiles_x;
    const int ty = bid % a.tiles_y; bid /= a.tiles_y;
    const int tco = bid % a.tiles_co;
    const int n = bid / a.tiles_co;
    const int co0 = tco * TCO, oy0 = ty * TH, ox0 = tx * TW;
    constexpr int PAD = (KS == 3 && STRIDE == 1) ? 1 : 0;
    const int Hv = a.Hin << a.upsample, Wv = a.Win << a.upsample;
    const int iy0 = oy0 * STRIDE - PAD, ix0 = ox0 * STRIDE - PAD;
    const float* xn = a.x + (int64_t)n * a.Cin * a.Hin * a.Win;
    const float* wn = a.w + (int64_t)n * a.w_batch_stride;
    const int cpg = a.gn_groups ? a.Cin / a.gn_groups : 1;

    // wave -> output row `wave` of the tile; 4 co sub-tiles x 2 pixel sub-tiles of 16x16
    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int kq = lane >> 4, l16 = lane & 15;
    // per k-step patch offset of this lane's k (k = s*4 + kq -> ci_local, dy, dx)
    int koff[C::KC / 4];
#pragma unroll
    for (int s = 0; s < C::KC / 4; ++s) {
        const int kk = s * 4 + kq;
        const int cl = kk / (KS * KS), tap = kk % (KS * KS);
        koff[s] = cl * (C::PH * C::PW) + (tap / KS) * C::PW + (tap % KS);
    }
    const int prow = wave * STRIDE * C::PW;  // patch row of this wave's output row

    for (int c0 = 0; c0 < a.Cin; c0 += C::CK) {
        __syncthreads();
        // ---- stage weights: sW[co][k], k = ci_local*KS*KS + tap ----
        for (int i = tid; i < TCO * C::KC; i += 256) {
            int co, k;
            if (a.w_transposed) { k = i / TCO; co = i % TCO; } else { co = i / C::KC; k = i % C::KC; }
            const int ci = c0 + k / (KS * KS);
            float v = 0.f;
            if (co0 + co < a.Cout && ci < a.Cin) {
                const int64_t kg = (int64_t)c0 * (KS * KS) + k;
                v = a.w_transposed ? wn[kg * a.ldw + co0 + co] : wn[(int64_t)(co0 + co) * a.ldw + kg];
            }
            sW[co * C::KPAD + k] = v;
        }
        // ---- stage input patch with optional upsample / GroupNorm(+SiLU) ----
        for (int i = tid; i < C::P_FLOATS; i += 256) {
            const int cl = i / (C::PH * C::PW), r = i % (C::PH * C::PW);
            const int py = r / C::PW, px = r % C::PW;
            const int ci = c0 + cl, iy = iy0 + py, ix = ix0 + px;
            float v = 0.f;
            if (ci < a.Cin && iy >= 0 && iy < Hv && ix >= 0 && ix < Wv) {
                v = xn[((int64_t)ci * a.Hin + (iy >> a.upsample)) * a.Win + (ix >> a.upsample)];
                if (a.gn_groups) {
                    const float* st = a.gn_stats + ((int64_t)n * a.gn_groups + ci / cpg) * 2;
                    v = (v - st[0]) * st[1] * a.gn_gamma[ci] + a.gn_beta[ci];
                    if (a.gn_silu) v = v / (1.0f + expf(-v));
                }
            }
            sP[i] = v;
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < C::KC / 4; ++s) {
            float wf[4], pf[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) wf[i] = sW[(i * 16 + l16) * C::KPAD + s * 4 + kq];
#pragma unroll
            for (int j = 0; j < 2; ++j) pf[j] = sP[koff[s] + prow + (j * 16 + l16) * STRIDE];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i], pf[j], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue: lane holds pixel l16 of sub-tile j, channels (kq*4 + r) of sub-tile i ----
    const int oy = oy0 + wave;
    if (oy < a.Hout) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ox = ox0 + j * 16 + l16;
            if (ox >= a.Wout) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + i * 16 + kq * 4 + r;
                    if (co >= a.Cout) continue;
                    const int64_t o = (((int64_t)n * a.Cout + co) * a.Hout + oy) * a.Wout + ox;
                    float v = acc[i][j][r];
                    if (a.bias) v += a.bias[co];
                    if (a.resid) v += a.resid[o];
                    a.y[o] = v;
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 3x3 stride-1 convolution on the bf16 matrix cores at fp32-class accuracy ("bf16x3").
// Every fp32 operand is split x = hi + lo (hi = bf16(x), lo = bf16(x - hi)); the product keeps the three leading
// terms hi*hi + hi*lo + lo*hi in fp32 accumulators (the dropped lo*lo term is 2^-16 relative), i.e. ~16 mantissa
// bits against the 10 of the TF32 convolutions torch/cuDNN runs by default on the reference's hardware
// (torch.backends.cudnn.allow_tf32).  Three v_mfma_f32_16x16x32_bf16 (16 cycles each) replace eight
// v_mfma_f32_16x16x4_f32 (32 cycles each) per 32 reduction elements.
//
// Tile: 64 output channels x (16 rows x 32 pixels), EIGHT waves of two rows each (acc 4 x 4 sub-tiles; a template
// variant with one row per wave, 8 x 32 pixels, serves images too small to fill the chip with the big tile); the
// reduction runs in chunks of 16 input channels.  K order inside a chunk: k = tap * 16 + channel with TEN tap slots (the
// tenth is zero weights), so one MFMA k-step (32) is two taps x 16 channels and a lane's 8 consecutive k are 8
// consecutive channels of ONE tap: with the patch stored [py][px][channel] and the (pre-split, pre-ordered) weights
// stored [co][tap slot][channel], both fragments are single ds_read_b128 per hi / lo plane.  Pixel rows are 48 B and
// weight rows 336 B apart (12 and 84 dwords: the 16-lane b128 groups read conflict-free).
// What the shape buys over the first version (4 x 32 pixels, four waves, 32-channel chunks: a 76-KiB weight image per
// chunk and workgroup = 88 B/clk/CU of LDS-DMA at the matrix pipe's rate, several times what a CU can take in, fetched
// and waited for inside every chunk; one wave per SIMD, so nothing covered the GroupNorm / SiLU / split pass either):
// the weight image is 44 KiB (+4 padding) per 4 x the MFMAs, it is double-buffered -- the next chunk's image is in
// flight for a whole chunk -- and two waves per SIMD share the matrix pipe.
// Same GroupNorm(+SiLU) / upsample prologue and bias + residual epilogue as conv_kernel.
constexpr int BX_CK = 16, BX_TAPS = 10, BX_PW = TW + 2;
constexpr int BX_PSTRIDE = 48;                       // bytes per pixel in one patch plane (32 used)
constexpr int BX_WROW = (BX_TAPS * BX_CK + 8) * 2;   // bytes per output channel in one weight plane (320 used)
constexpr int BX_W_BYTES = TCO * BX_WROW;
constexpr int BX_IMG = 48 * 1024;                    // one (co tile, channel chunk) weight image: hi plane, lo plane, padding
constexpr int BX_P_MAX = (16 + 2) * BX_PW * BX_PSTRIDE;   // one patch plane of the 16-row tile
constexpr int BX_GN_OFF = 2 * BX_IMG + 2 * BX_P_MAX;  // scale / shift of a chunk's 16 channels, two chunks
constexpr int BX_LDS_TOTAL = BX_GN_OFF + 2 * 2 * BX_CK * 4;
static_assert(2 * BX_W_BYTES <= BX_IMG && BX_IMG % 8192 == 0, "weight image must split into whole 1-KiB pieces per wave");
static_assert(BX_LDS_TOTAL <= 160 * 1024, "LDS budget");

struct ConvBxArgs {
    const float* x; const char* wimg; const float* bias; const float* resid;
    const float* gn_stats; const float* gn_gamma; const float* gn_beta;
    float* y;
    int N, Cin, nch, Hin, Win, Cout, Hout, Wout;
    int upsample, gn_groups, gn_silu;
    int tiles_x, tiles_y, tiles_co;
};

__device__ __forceinline__ void bx_glds16(const char* base, uint32_t voffset, uint32_t lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voffset), "s"(base), "s"(lds_dst)
        : "memory");
}

// Per chunk c of 16 input channels:  [barrier: the MFMAs of chunk c-1 are done, and every wave's share of weight image c,
// issued a chunk ago, has landed]  weight image c+1 goes out by LDS-DMA into the other buffer;  raw patch values
// (prefetched into registers under the previous chunk's MFMAs) -> GroupNorm/SiLU/split -> LDS;  [barrier]  issue the raw
// patch loads of chunk c+1;  5 k-steps x 24 RPW MFMAs.
// VGPT_BX_DEBUG (diagnostic builds, results are WRONG): 1 = no GroupNorm / SiLU / split arithmetic (values stored as
// they come), 2 = no raw patch loads, 4 = no weight LDS-DMA, 8 = no epilogue, 16 = no MFMAs
#ifndef VGPT_BX_DEBUG
#define VGPT_BX_DEBUG 0
#endif
template <int RPW>   // rows per wave: 2 (16 x 32 pixel tile) or 1 (8 x 32)
__global__ __launch_bounds__(512, 1) void conv_bx3_kernel(ConvBxArgs a) {
    constexpr int TH_ = 8 * RPW, PH_ = TH_ + 2, P_BYTES = PH_ * BX_PW * BX_PSTRIDE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sPh = smem + 2 * BX_IMG;
    char* sPl = sPh + P_BYTES;
    float* sGN = reinterpret_cast<float*>(smem + BX_GN_OFF);   // [chunk parity][scale 16 | shift 16]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = blockIdx.x;
    const int tx = bid % a.tiles_x; bid /= a.tiles_x;
    const int ty = bid % a.tiles_y; bid /= a.tiles_y;
    const int tco = bid % a.tiles_co;
    const int n = bid / a.tiles_co;
    const int co0 = tco * TCO, oy0 = ty * TH_, ox0 = tx * TW;
    const int Hv = a.Hin << a.upsample, Wv = a.Win << a.upsample;
    const int iy0 = oy0 - 1, ix0 = ox0 - 1;
    const float* xn = a.x + (int64_t)n * a.Cin * a.Hin * a.Win;
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
    const int cpg = a.gn_groups ? a.Cin / a.gn_groups : 1;

    f32x4 acc[4][2 * RPW];   // [co sub-tile][2 * row + pixel half]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2 * RPW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int kq = lane >> 4, l16 = lane & 15;

    // staging item = (pixel of the PH_ x 34 patch, group of 8 channels) over 512 threads.  Loads are unconditional from
    // clamped addresses (a scalar base per channel + one 32-bit lane offset per item: no per-element address arithmetic,
    // no branches): channels past Cin meet zero weights, pixels outside the image are zeroed after GroupNorm / SiLU.
    constexpr int NPIX = PH_ * BX_PW, ITEMS = NPIX * 2, PER_T = (ITEMS + 511) / 512;
    float praw[PER_T][8];
    uint32_t pofs[PER_T];   // element offset (pixel + 8 q channel planes) of this item inside one image of the batch
    bool pin[PER_T];        // the pixel lies inside the image
    const int64_t plane = (int64_t)a.Hin * a.Win;
#pragma unroll
    for (int it = 0; it < PER_T; ++it) {
        const int i = min(tid + 512 * it, ITEMS - 1);
        const int pix = i % NPIX, py = pix / BX_PW, px = pix % BX_PW, q = i / NPIX;
        const int iy = iy0 + py, ix = ix0 + px;
        pin[it] = iy >= 0 && iy < Hv && ix >= 0 && ix < Wv;
        const int cy = min(max(iy, 0), Hv - 1) >> a.upsample, cx = min(max(ix, 0), Wv - 1) >> a.upsample;
        pofs[it] = (uint32_t)(cy * a.Win + cx) + (uint32_t)(q * 8) * (uint32_t)plane;
    }
    const int c_last = a.Cin - 1;
    const bool whole_chunks = (a.Cin % BX_CK) == 0;   // every layer but the 3- / 4-channel input convolutions
    auto load_raw = [&](int c0) {
        if constexpr ((VGPT_BX_DEBUG & 2) != 0) return;
        if (whole_chunks) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float* base = xn + (int64_t)(c0 + j) * plane;   // channel c0 + j (+ 8 q through the lane offset)
#pragma unroll
                for (int it = 0; it < PER_T; ++it) praw[it][j] = base[pofs[it]];
            }
        } else {   // a partial chunk: per-lane clamped channel (the weights of channels past Cin are zero)
#pragma unroll
            for (int it = 0; it < PER_T; ++it) {
                const int q = min(tid + 512 * it, ITEMS - 1) / NPIX;
                const uint32_t pix_off = pofs[it] - (uint32_t)(q * 8) * (uint32_t)plane;
#pragma unroll
                for (int j = 0; j < 8; ++j) praw[it][j] = xn[(int64_t)min(c0 + q * 8 + j, c_last) * plane + pix_off];
            }
        }
    };
    const char* wimg = a.wimg + (int64_t)tco * a.nch * BX_IMG;
    auto stage_w = [&](int c) {   // weight image of chunk c: 48 KiB, 6 x 1-KiB pieces per wave, straight into its LDS layout
        const uint64_t sa = (uint64_t)(uintptr_t)(wimg + (int64_t)c * BX_IMG);   // wave-uniform: keep it in SGPRs
        const char* src = reinterpret_cast<const char*>(
            ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(sa >> 32)) << 32) |
            (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)sa));
#pragma unroll
        for (int p = 0; p < BX_IMG / 8192; ++p) {
            if constexpr ((VGPT_BX_DEBUG & 4) != 0) continue;
            const uint32_t off = (uint32_t)((wave * (BX_IMG / 8192) + p) * 1024);
            bx_glds16(src, off + lane * 16, lds_base + (uint32_t)((c & 1) * BX_IMG) + off);
        }
    };
    auto gn_table = [&](int c) {   // GroupNorm as one FMA per element: scale = rstd * gamma, shift = beta - mean * scale
        if (tid < BX_CK) {
            float sc = 1.f, sh = 0.f;   // no GroupNorm: identity
            if (a.gn_groups) {
                const int ci = min(c * BX_CK + tid, a.Cin - 1);
                const float* st = a.gn_stats + ((int64_t)n * a.gn_groups + ci / cpg) * 2;
                sc = st[1] * a.gn_gamma[ci];
                sh = a.gn_beta[ci] - st[0] * sc;
            }
            sGN[(c & 1) * 2 * BX_CK + tid] = sc;
            sGN[(c & 1) * 2 * BX_CK + BX_CK + tid] = sh;
        }
    };
    stage_w(0);
    gn_table(0);
    load_raw(0);
    // patch offset of a lane's tap in k-step s: tap = 2 s + (kq >> 1) (the zero tenth slot reads the ninth tap's pixels)
    int tap_off[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int t = min(2 * s + (kq >> 1), 8);
        tap_off[s] = ((t / 3) * BX_PW + t % 3) * BX_PSTRIDE + (kq & 1) * 16;
    }

    for (int c = 0; c < a.nch; ++c) {
        const int c0 = c * BX_CK;
        // retire the raw loads of this chunk (the compiler's wait for them is vmcnt(0): weight image c, issued before
        // them, has landed with it)
#pragma unroll
        for (int it = 0; it < PER_T; ++it)
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(praw[it][j]));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // every wave is done with chunk c-1's MFMAs; weight image c and the GroupNorm table c are complete
        if (c + 1 < a.nch) {
            stage_w(c + 1);
            gn_table(c + 1);
        }
        // ---- patch: GroupNorm(+SiLU), split into hi / lo, [pixel][channel] planes ----
        const float* gn = sGN + (c & 1) * 2 * BX_CK;
#pragma unroll
        for (int it = 0; it < PER_T; ++it) {
            const int i = tid + 512 * it;
            if (i >= ITEMS) continue;
            const int pix = i % NPIX, q = i / NPIX;
            bf16x8 vh, vl;
            f32x4 sc[2], sh[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                sc[u] = *reinterpret_cast<const f32x4*>(gn + q * 8 + 4 * u);
                sh[u] = *reinterpret_cast<const f32x4*>(gn + BX_CK + q * 8 + 4 * u);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float v = praw[it][j];
                if constexpr ((VGPT_BX_DEBUG & 1) == 0) {
                    v = __builtin_fmaf(v, sc[j >> 2][j & 3], sh[j >> 2][j & 3]);
                    if (a.gn_silu) v = v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
                    v = pin[it] ? v : 0.f;   // zero padding applies to the normalised activation
                }
                const bf16 hi = f2bf(v);
                vh[j] = hi;
                vl[j] = (VGPT_BX_DEBUG & 1) ? hi : f2bf(v - bf2f(hi));
            }
            const int o = pix * BX_PSTRIDE + q * 16;
            *reinterpret_cast<bf16x8*>(sPh + o) = vh;
            *reinterpret_cast<bf16x8*>(sPl + o) = vl;
        }
        __syncthreads();
        if (c + 1 < a.nch) load_raw(c0 + BX_CK);            // in flight under the MFMAs below
        const char* sWh = smem + (c & 1) * BX_IMG;
        const char* sWl = sWh + BX_W_BYTES;
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            bf16x8 wh[4], wl[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int o = (i * 16 + l16) * BX_WROW + (s * 32 + kq * 8) * 2;
                wh[i] = *reinterpret_cast<const bf16x8*>(sWh + o);
                wl[i] = *reinterpret_cast<const bf16x8*>(sWl + o);
            }
#pragma unroll
            for (int j = 0; j < 2 * RPW; ++j) {   // j = 2 * row + pixel half
                const int o = ((RPW * wave + (j >> 1)) * BX_PW + (j & 1) * 16 + l16) * BX_PSTRIDE + tap_off[s];
                const bf16x8 ph = *reinterpret_cast<const bf16x8*>(sPh + o);
                const bf16x8 pl = *reinterpret_cast<const bf16x8*>(sPl + o);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if constexpr ((VGPT_BX_DEBUG & 16) != 0) {
                        acc[i][j][0] += bf2f(wl[i][0]) * bf2f(ph[0]) + bf2f(wh[i][1]) * bf2f(pl[1]);   // keeps the reads alive
                        continue;
                    }
                    // pixels are the MFMA's rows: a lane ends up with FOUR CONSECUTIVE PIXELS (kq*4 + r) of channel l16
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph, wl[i], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pl, wh[i], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph, wh[i], acc[i][j], 0, 0, 0);
                }
            }
        }
    }

    if constexpr ((VGPT_BX_DEBUG & 8) != 0) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2 * RPW; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (t == 12345.678f) a.y[0] = t;
        return;
    }
    // ---- epilogue: lane holds channel l16 of sub-tile i, pixels kq*4 .. +3 of sub-tile j: 16-byte stores ----
    const bool vec = (a.Wout & 3) == 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int co = co0 + i * 16 + l16;
        if (co >= a.Cout) continue;
        const float bv = a.bias ? a.bias[co] : 0.f;
#pragma unroll
        for (int j = 0; j < 2 * RPW; ++j) {
            const int oy = oy0 + RPW * wave + (j >> 1), ox = ox0 + (j & 1) * 16 + kq * 4;
            if (oy >= a.Hout || ox >= a.Wout) continue;
            const int64_t o = (((int64_t)n * a.Cout + co) * a.Hout + oy) * a.Wout + ox;
            f32x4 v = acc[i][j];
            if (vec) {   // Wout % 4 == 0: the four pixels are inside the row together, 16-byte aligned
                if (a.resid) {
                    const f32x4 rv = *reinterpret_cast<const f32x4*>(a.resid + o);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += rv[r];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += bv;
                *reinterpret_cast<f32x4*>(a.y + o) = v;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (ox + r < a.Wout) a.y[o + r] = v[r] + bv + (a.resid ? a.resid[o + r] : 0.f);
            }
        }
    }
}

// w (Cout, Cin, 3, 3) fp32 -> LDS-ready images: for every (64-channel co tile, 16-channel chunk) the hi plane then the lo
// plane, each [co_local 64][BX_WROW bytes] with k = tap * 16 + channel (zeros for the tenth tap slot and for channels /
// output channels past the end; the caller zero-fills the buffer)
__global__ void conv_pack_bx3_kernel(const float* __restrict__ w, char* __restrict__ img, int Cout, int Cin, int nch,
                                     int tiles_co) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one (tile, chunk, co_local, tap, channel)
    const int64_t total = (int64_t)tiles_co * nch * TCO * 9 * BX_CK;
    if (i >= total) return;
    const int cl = (int)(i % BX_CK), t = (int)((i / BX_CK) % 9), col = (int)((i / (BX_CK * 9)) % TCO);
    const int c = (int)((i / ((int64_t)BX_CK * 9 * TCO)) % nch), tc = (int)(i / ((int64_t)BX_CK * 9 * TCO * nch));
    const int co = tc * TCO + col, ci = c * BX_CK + cl;
    const float v = (co < Cout && ci < Cin) ? w[((int64_t)co * Cin + ci) * 9 + t] : 0.f;
    const bf16 h = f2bf(v);
    char* base = img + ((int64_t)tc * nch + c) * BX_IMG + (int64_t)col * BX_WROW + (t * BX_CK + cl) * 2;
    *reinterpret_cast<bf16*>(base) = h;
    *reinterpret_cast<bf16*>(base + BX_W_BYTES) = f2bf(v - bf2f(h));
}

// ---------------------------------------------------------------------------------------------------------------
// 1x1 convolution (channel mixing: the resnet shortcuts, the mid-block attention's q / k / v / out projections) in the
// same split-bf16 arithmetic.  y[n][co][p] = sum_ci w[co][ci] x[n][ci][p] is a GEMM whose activation operand is
// channel-major in memory, so it goes through the same register prefetch -> (GroupNorm / SiLU) -> hi / lo split ->
// [pixel][channel] LDS planes as the 3x3 kernel's patch, without the halo.  Tile: 64 output channels x 512 consecutive
// pixels of one image, eight waves of 64 pixels (acc 4 x 4 sub-tiles); chunks of 32 input channels = one MFMA k-step;
// weight images of 16 KiB per (co tile, chunk) (rows of 32 channels + 16 B pad, hi plane then lo plane) double-buffered
// by LDS-DMA.  At 256^2 these layers are HBM-bound (the 256 -> 128 shortcut moves 0.8 GB per 8 frames for 34 GFLOP); the
// exact-fp32 MFMA kernel they ran on before reached a fifth of that.
constexpr int B1_CK = 32, B1_TP = 512;
constexpr int B1_ROW = B1_CK * 2 + 16;                 // bytes per pixel / per output channel in one plane (80: 20 dwords)
constexpr int B1_P_BYTES = B1_TP * B1_ROW;              // 40 KiB per plane
constexpr int B1_W_BYTES = TCO * B1_ROW;                // 5 KiB per plane
constexpr int B1_IMG = 16 * 1024;
constexpr int B1_GN_OFF = 2 * B1_IMG + 2 * B1_P_BYTES;
constexpr int B1_LDS_TOTAL = B1_GN_OFF + 2 * 2 * B1_CK * 4;
static_assert(2 * B1_W_BYTES <= B1_IMG && B1_LDS_TOTAL <= 160 * 1024, "LDS budget");

struct Conv1Args {
    const float* x; const char* wimg; const float* bias; const float* resid;
    const float* gn_stats; const float* gn_gamma; const float* gn_beta;
    float* y;
    int N, Cin, nch, HW, Cout;
    int gn_groups, gn_silu;
    int tiles_p, tiles_co;
};

__global__ __launch_bounds__(512, 1) void conv1x1_bx3_kernel(Conv1Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sPh = smem + 2 * B1_IMG;
    char* sPl = sPh + B1_P_BYTES;
    float* sGN = reinterpret_cast<float*>(smem + B1_GN_OFF);   // [chunk parity][scale 32 | shift 32]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = blockIdx.x;
    const int tp = bid % a.tiles_p; bid /= a.tiles_p;
    const int tco = bid % a.tiles_co;
    const int n = bid / a.tiles_co;
    const int co0 = tco * TCO, p0 = tp * B1_TP;
    const float* xn = a.x + (int64_t)n * a.Cin * a.HW;
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
    const int cpg = a.gn_groups ? a.Cin / a.gn_groups : 1;
    const int kq = lane >> 4, l16 = lane & 15;

    f32x4 acc[4][4];   // [co sub-tile][pixel sub-tile of this wave's 64 pixels]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // thread = one pixel of the tile: its 32 channel values of a chunk (coalesced across the threads of a wave)
    const uint32_t pofs = (uint32_t)min(p0 + tid, a.HW - 1);
    float praw[B1_CK];
    auto load_raw = [&](int c0) {   // Cin % 32 == 0 (checked on the host): no clamping
#pragma unroll
        for (int j = 0; j < B1_CK; ++j) praw[j] = (xn + (int64_t)(c0 + j) * a.HW)[pofs];
    };
    const char* wimg = a.wimg + (int64_t)tco * a.nch * B1_IMG;
    auto stage_w = [&](int c) {   // 16 KiB image: 2 x 1-KiB pieces per wave
        const uint64_t sa = (uint64_t)(uintptr_t)(wimg + (int64_t)c * B1_IMG);
        const char* src = reinterpret_cast<const char*>(
            ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(sa >> 32)) << 32) |
            (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)sa));
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const uint32_t off = (uint32_t)((wave * 2 + p) * 1024);
            bx_glds16(src, off + lane * 16, lds_base + (uint32_t)((c & 1) * B1_IMG) + off);
        }
    };
    auto gn_table = [&](int c) {
        if (tid < B1_CK) {
            float sc = 1.f, sh = 0.f;
            if (a.gn_groups) {
                const int ci = c * B1_CK + tid;
                const float* st = a.gn_stats + ((int64_t)n * a.gn_groups + ci / cpg) * 2;
                sc = st[1] * a.gn_gamma[ci];
                sh = a.gn_beta[ci] - st[0] * sc;
            }
            sGN[(c & 1) * 2 * B1_CK + tid] = sc;
            sGN[(c & 1) * 2 * B1_CK + B1_CK + tid] = sh;
        }
    };
    stage_w(0);
    gn_table(0);
    load_raw(0);

    for (int c = 0; c < a.nch; ++c) {
#pragma unroll
        for (int j = 0; j < B1_CK; ++j) asm volatile("" : "+v"(praw[j]));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // chunk c-1's MFMAs are done; weight image c and GroupNorm table c are complete
        if (c + 1 < a.nch) {
            stage_w(c + 1);
            gn_table(c + 1);
        }
        const float* gn = sGN + (c & 1) * 2 * B1_CK;
#pragma unroll
        for (int q = 0; q < 4; ++q) {   // 8 channels -> one 16-byte store per plane
            bf16x8 vh, vl;
            f32x4 sc[2], sh[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                sc[u] = *reinterpret_cast<const f32x4*>(gn + q * 8 + 4 * u);
                sh[u] = *reinterpret_cast<const f32x4*>(gn + B1_CK + q * 8 + 4 * u);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float v = __builtin_fmaf(praw[q * 8 + j], sc[j >> 2][j & 3], sh[j >> 2][j & 3]);
                if (a.gn_silu) v = v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
                const bf16 hi = f2bf(v);
                vh[j] = hi;
                vl[j] = f2bf(v - bf2f(hi));
            }
            *reinterpret_cast<bf16x8*>(sPh + tid * B1_ROW + q * 16) = vh;
            *reinterpret_cast<bf16x8*>(sPl + tid * B1_ROW + q * 16) = vl;
        }
        __syncthreads();
        if (c + 1 < a.nch) load_raw((c + 1) * B1_CK);
        const char* sWh = smem + (c & 1) * B1_IMG;
        const char* sWl = sWh + B1_W_BYTES;
        bf16x8 wh[4], wl[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int o = (i * 16 + l16) * B1_ROW + kq * 16;
            wh[i] = *reinterpret_cast<const bf16x8*>(sWh + o);
            wl[i] = *reinterpret_cast<const bf16x8*>(sWl + o);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int o = (wave * 64 + j * 16 + l16) * B1_ROW + kq * 16;
            const bf16x8 ph = *reinterpret_cast<const bf16x8*>(sPh + o);
            const bf16x8 pl = *reinterpret_cast<const bf16x8*>(sPl + o);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph, wl[i], acc[i][j], 0, 0, 0);   // pixels = rows
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pl, wh[i], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph, wh[i], acc[i][j], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: lane holds channel l16 of sub-tile i, pixels kq*4 .. +3 of sub-tile j: 16-byte stores ----
    const bool vec = (a.HW & 3) == 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int co = co0 + i * 16 + l16;
        if (co >= a.Cout) continue;
        const float bv = a.bias ? a.bias[co] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = p0 + wave * 64 + j * 16 + kq * 4;
            if (p >= a.HW) continue;
            const int64_t o = ((int64_t)n * a.Cout + co) * a.HW + p;
            f32x4 v = acc[i][j];
            if (vec) {
                if (a.resid) {
                    const f32x4 rv = *reinterpret_cast<const f32x4*>(a.resid + o);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += rv[r];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += bv;
                *reinterpret_cast<f32x4*>(a.y + o) = v;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (p + r < a.HW) a.y[o + r] = v[r] + bv + (a.resid ? a.resid[o + r] : 0.f);
            }
        }
    }
}

// w (Cout, Cin) fp32 -> per (64-channel co tile, 32-channel chunk) a 16-KiB image: hi plane then lo plane, each
// [co_local 64][B1_ROW bytes] (the caller zero-fills the buffer)
__global__ void conv_pack_b1_kernel(const float* __restrict__ w, char* __restrict__ img, int Cout, int Cin, int nch,
                                    int tiles_co) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one (tile, chunk, co_local, channel)
    const int64_t total = (int64_t)tiles_co * nch * TCO * B1_CK;
    if (i >= total) return;
    const int cl = (int)(i % B1_CK), col = (int)((i / B1_CK) % TCO);
    const int c = (int)((i / ((int64_t)B1_CK * TCO)) % nch), tc = (int)(i / ((int64_t)B1_CK * TCO * nch));
    const int co = tc * TCO + col, ci = c * B1_CK + cl;
    const float v = (co < Cout && ci < Cin) ? w[(int64_t)co * Cin + ci] : 0.f;
    const bf16 h = f2bf(v);
    char* base = img + ((int64_t)tc * nch + c) * B1_IMG + (int64_t)col * B1_ROW + cl * 2;
    *reinterpret_cast<bf16*>(base) = h;
    *reinterpret_cast<bf16*>(base + B1_W_BYTES) = f2bf(v - bf2f(h));
}

// ---- GroupNorm statistics: one 1024-thread block per (n, group), ONE pass over the group ----
// Sums of d = x - K and d^2 with the shift K = the group's first element (close to the mean, so Q/n - (S/n)^2 does
// not cancel), four independent 16-byte loads per thread and iteration; per-thread partial sums, then a wave / LDS
// tree.  (The two-pass version read every group twice with scalar loads: 2 TB/s, a quarter of the VAE's time.)
__global__ __launch_bounds__(1024) void gn_stats_kernel(const float* __restrict__ x, float* __restrict__ stats,
                                                        int64_t group_elems, float eps) {
    __shared__ float red[2][16];
    const float* p = x + (int64_t)blockIdx.x * group_elems;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float K = p[0];
    float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
    const bool vec = (group_elems & 3) == 0 && (((uintptr_t)p) & 15) == 0;
    if (vec) {
        const int64_t n4 = group_elems >> 2;
        const f32x4* p4 = reinterpret_cast<const f32x4*>(p);
        int64_t i = threadIdx.x;
        for (; i + 3 * 1024 < n4; i += 4 * 1024) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = p4[i + u * 1024];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = v[u][j] - K;
                    s[u] += d;
                    q[u] += d * d;
                }
        }
        for (; i < n4; i += 1024) {
            const f32x4 v = p4[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = v[j] - K;
                s[0] += d;
                q[0] += d * d;
            }
        }
    } else {
        for (int64_t i = threadIdx.x; i < group_elems; i += 1024) {
            const float d = p[i] - K;
            s[0] += d;
            q[0] += d * d;
        }
    }
    const float sw = wave_sum((s[0] + s[1]) + (s[2] + s[3])), qw = wave_sum((q[0] + q[1]) + (q[2] + q[3]));
    if (lane == 0) {
        red[0][wave] = sw;
        red[1][wave] = qw;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float S = 0.f, Q = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            S += red[0][w];
            Q += red[1][w];
        }
        const float md = S / (float)group_elems;
        const float var = fmaxf(Q / (float)group_elems - md * md, 0.f);
        stats[blockIdx.x * 2] = K + md;
        stats[blockIdx.x * 2 + 1] = rsqrtf(var + eps);
    }
}

// ---- column softmax of S^T (n, keys, queries): softmax over keys for every query column, in place ----
// block = 64 query columns x 4 key quarters (one wave each: a wave reads 64 consecutive floats of a key row); the four
// partial (max, sum) pairs of a column meet in LDS, then every wave normalises its own quarter.
__global__ __launch_bounds__(256) void col_softmax_kernel(float* __restrict__ s, int keys, int queries, float scale) {
    __shared__ float sm[4][64], sl[4][64];
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int q = blockIdx.x * 64 + c;
    const bool live = q < queries;
    float* p = s + (int64_t)blockIdx.y * keys * queries + min(q, queries - 1);
    const int per = (keys + 3) / 4, k0 = g * per, k1 = min(k0 + per, keys);
    float m = -INFINITY, l = 0.f;
    for (int k = k0; k < k1; ++k) {
        const float v = p[(int64_t)k * queries] * scale;
        const float mn = fmaxf(m, v);
        l = l * expf(m - mn) + expf(v - mn);
        m = mn;
    }
    sm[g][c] = m;
    sl[g][c] = l;
    __syncthreads();
    float M = fmaxf(fmaxf(sm[0][c], sm[1][c]), fmaxf(sm[2][c], sm[3][c])), Lsum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) Lsum += sm[i][c] == -INFINITY ? 0.f : sl[i][c] * expf(sm[i][c] - M);
    const float inv = 1.0f / Lsum;
    if (!live) return;
    for (int k = k0; k < k1; ++k) {
        float* e = p + (int64_t)k * queries;
        *e = expf(*e * scale - M) * inv;
    }
}

// ---- latent sampling: z = (mean + exp(0.5 clamp(logvar,-30,20)) * noise - shift) * scaling ----
__global__ void vae_sample_kernel(const float* __restrict__ moments, const float* __restrict__ noise,
                                  float* __restrict__ z, int64_t per_image, int64_t total, float shift,
                                  float scaling) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t n = i / per_image, r = i % per_image;
    const float mean = moments[n * 2 * per_image + r];
    float lv = moments[n * 2 * per_image + per_image + r];
    lv = fminf(fmaxf(lv, -30.0f), 20.0f);
    z[i] = (mean + expf(0.5f * lv) * noise[i] - shift) * scaling;
}

// ---- (x*0.5+0.5).clamp(0,1)*255 -> uint8, NCHW -> NHWC (LVM/pipeline.py:585-588) ----
__global__ void vae_post_u8_kernel(const float* __restrict__ x, uint8_t* __restrict__ out, int N, int C, int H,
                                   int W) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)N * C * H * W;
    if (i >= total) return;
    const int c = i % C;
    const int64_t r = i / C;
    const int xw = r % W;
    const int64_t r2 = r / W;
    const int yh = r2 % H;
    const int n = r2 / H;
    float v = x[(((int64_t)n * C + c) * H + yh) * W + xw] * 0.5f + 0.5f;
    v = fminf(fmaxf(v, 0.f), 1.f) * 255.0f;
    out[i] = (uint8_t)v;  // truncation, as torch's .to(uint8)
}

// ---- scale / unscale helpers: y = x * mul + add (latent / scaling_factor + shift) ----
__global__ void affine_kernel(const void* __restrict__ x, int x_is_bf16, float* __restrict__ y, int64_t n, float mul,
                              float add) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x_is_bf16 ? bf2f(reinterpret_cast<const bf16*>(x)[i]) : reinterpret_cast<const float*>(x)[i];
    y[i] = v * mul + add;
}

template <int KS, int STRIDE>
int launch_conv(const ConvArgs& a, hipStream_t s) {
    using C = ConvCfg<KS, STRIDE>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)conv_kernel<KS, STRIDE>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) {
            vgpt_set_error("vgpt_conv2d_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
            return VGPT_ERR_HIP;
        }
        attr_set = true;
    }
    const int64_t blocks = (int64_t)a.tiles_x * a.tiles_y * a.tiles_co * a.N;
    hipLaunchKernelGGL((conv_kernel<KS, STRIDE>), dim3((unsigned)blocks), dim3(256), C::LDS_BYTES, s, a);
    VGPT_CHECK_LAUNCH("vgpt_conv2d_fwd");
    return VGPT_OK;
}

}  // namespace

VGPT_EXPORT int vgpt_groupnorm_stats(const float* x, float* stats, int64_t N, int C, int HW, int groups, float eps,
                                     void* stream) {
    VGPT_REQUIRE(x && stats, VGPT_ERR_INVALID, "vgpt_groupnorm_stats: null pointer");
    VGPT_REQUIRE(N >= 0 && C > 0 && HW > 0 && groups > 0 && C % groups == 0, VGPT_ERR_INVALID,
                 "vgpt_groupnorm_stats: bad shape");
    if (N == 0) return VGPT_OK;
    hipLaunchKernelGGL(gn_stats_kernel, dim3((unsigned)(N * groups)), dim3(1024), 0, (hipStream_t)stream, x, stats,
                       (int64_t)(C / groups) * HW, eps);
    VGPT_CHECK_LAUNCH("vgpt_groupnorm_stats");
    return VGPT_OK;
}

VGPT_EXPORT int vgpt_conv2d_fwd(const float* x, const float* w, const float* bias, const float* resid,
                                const float* gn_stats, const float* gn_gamma, const float* gn_beta, float* y, int N,
                                int Cin, int Hin, int Win, int Cout, int ksize, int stride, int upsample,
                                int gn_groups, int gn_silu, int w_transposed, int64_t ldw, int64_t w_batch_stride,
                                void* stream) {
    VGPT_REQUIRE(x && w && y, VGPT_ERR_INVALID, "vgpt_conv2d_fwd: null pointer");
    VGPT_REQUIRE(N >= 0 && Cin > 0 && Hin > 0 && Win > 0 && Cout > 0, VGPT_ERR_INVALID, "vgpt_conv2d_fwd: bad shape");
    VGPT_REQUIRE((ksize == 3 && (stride == 1 || stride == 2)) || (ksize == 1 && stride == 1), VGPT_ERR_UNSUPPORTED,
                 "vgpt_conv2d_fwd: only 3x3 (stride 1, 2) and 1x1 convolutions");
    VGPT_REQUIRE(!(upsample && stride != 1), VGPT_ERR_UNSUPPORTED, "vgpt_conv2d_fwd: upsample needs stride 1");
    VGPT_REQUIRE(gn_groups == 0 || (gn_stats && gn_gamma && gn_beta && Cin % gn_groups == 0), VGPT_ERR_INVALID,
                 "vgpt_conv2d_fwd: GroupNorm prologue needs stats/gamma/beta and Cin %% groups == 0");
    if (N == 0) return VGPT_OK;
    ConvArgs a;
    a.x = x; a.w = w; a.bias = bias; a.resid = resid; a.gn_stats = gn_stats; a.gn_gamma = gn_gamma; a.gn_beta = gn_beta;
    a.y = y; a.N = N; a.Cin = Cin; a.Hin = Hin; a.Win = Win; a.Cout = Cout;
    const int Hv = Hin << (upsample ? 1 : 0), Wv = Win << (upsample ? 1 : 0);
    a.Hout = stride == 2 ? Hv / 2 : Hv;
    a.Wout = stride == 2 ? Wv / 2 : Wv;
    a.upsample = upsample ? 1 : 0; a.gn_groups = gn_groups; a.gn_silu = gn_silu; a.w_transposed = w_transposed;
    a.ldw = ldw; a.w_batch_stride = w_batch_stride;
    a.tiles_x = (int)cdiv(a.Wout, TW); a.tiles_y = (int)cdiv(a.Hout, TH); a.tiles_co = (int)cdiv(Cout, TCO);
    hipStream_t s = (hipStream_t)stream;
    if (ksize == 1) return launch_conv<1, 1>(a, s);
    if (stride == 1) return launch_conv<3, 1>(a, s);
    return launch_conv<3, 2>(a, s);
}

VGPT_EXPORT int64_t vgpt_conv_bx3_packed_bytes(int Cout, int Cin) {
    return (int64_t)cdiv(Cout, TCO) * cdiv(Cin, BX_CK) * BX_IMG;
}

VGPT_EXPORT int vgpt_conv_pack_weights_bx3(const float* w, void* packed, int Cout, int Cin, void* stream) {
    VGPT_REQUIRE(w && packed && Cout > 0 && Cin > 0, VGPT_ERR_INVALID, "vgpt_conv_pack_weights_bx3: bad argument");
    const int tiles_co = (int)cdiv(Cout, TCO), nch = (int)cdiv(Cin, BX_CK);
    hipError_t e = hipMemsetAsync(packed, 0, vgpt_conv_bx3_packed_bytes(Cout, Cin), (hipStream_t)stream);   // row padding
    if (e != hipSuccess) {
        vgpt_set_error("vgpt_conv_pack_weights_bx3: memset: %s", hipGetErrorString(e));
        return VGPT_ERR_HIP;
    }
    const int64_t n = (int64_t)tiles_co * nch * TCO * 9 * BX_CK;
    hipLaunchKernelGGL(conv_pack_bx3_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, w, (char*)packed,
                       Cout, Cin, nch, tiles_co);
    VGPT_CHECK_LAUNCH("vgpt_conv_pack_weights_bx3");
    return VGPT_OK;
}

VGPT_EXPORT int vgpt_conv2d_bx3_fwd(const float* x, const void* packed, const float* bias, const float* resid,
                                    const float* gn_stats, const float* gn_gamma, const float* gn_beta, float* y, int N,
                                    int Cin, int Hin, int Win, int Cout, int upsample, int gn_groups, int gn_silu,
                                    void* stream) {
    VGPT_REQUIRE(x && packed && y, VGPT_ERR_INVALID, "vgpt_conv2d_bx3_fwd: null pointer");
    VGPT_REQUIRE(N >= 0 && Cin > 0 && Hin > 0 && Win > 0 && Cout > 0, VGPT_ERR_INVALID, "vgpt_conv2d_bx3_fwd: bad shape");
    VGPT_REQUIRE(gn_groups == 0 || (gn_stats && gn_gamma && gn_beta && Cin % gn_groups == 0), VGPT_ERR_INVALID,
                 "vgpt_conv2d_bx3_fwd: GroupNorm prologue needs stats/gamma/beta and Cin %% groups == 0");
    VGPT_REQUIRE(((uintptr_t)packed & 15) == 0, VGPT_ERR_UNSUPPORTED, "vgpt_conv2d_bx3_fwd: packed weights must be 16-byte aligned");
    if (N == 0) return VGPT_OK;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)conv_bx3_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, BX_LDS_TOTAL);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)conv_bx3_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, BX_LDS_TOTAL);
        if (e != hipSuccess) {
            vgpt_set_error("vgpt_conv2d_bx3_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
            return VGPT_ERR_HIP;
        }
        attr_set = true;
    }
    ConvBxArgs a;
    a.x = x; a.wimg = (const char*)packed; a.bias = bias; a.resid = resid;
    a.gn_stats = gn_stats; a.gn_gamma = gn_gamma; a.gn_beta = gn_beta; a.y = y;
    a.N = N; a.Cin = Cin; a.nch = (int)cdiv(Cin, BX_CK); a.Hin = Hin; a.Win = Win; a.Cout = Cout;
    a.upsample = upsample ? 1 : 0;
    a.Hout = Hin << a.upsample; a.Wout = Win << a.upsample;
    a.gn_groups = gn_groups; a.gn_silu = gn_silu;
    a.tiles_x = (int)cdiv(a.Wout, TW); a.tiles_co = (int)cdiv(Cout, TCO);
    // 16-row tiles unless they leave CUs without a workgroup (one workgroup per CU: a launch wants >= 256 of them)
    const int64_t per_row_tile = (int64_t)a.tiles_x * a.tiles_co * N;
    const bool big = per_row_tile * cdiv(a.Hout, 16) >= 256;
    a.tiles_y = (int)cdiv(a.Hout, big ? 16 : 8);
    const int64_t blocks = per_row_tile * a.tiles_y;
    if (big) hipLaunchKernelGGL(conv_bx3_kernel<2>, dim3((unsigned)blocks), dim3(512), BX_LDS_TOTAL, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(conv_bx3_kernel<1>, dim3((unsigned)blocks), dim3(512), BX_LDS_TOTAL, (hipStream_t)stream, a);
    VGPT_CHECK_LAUNCH("vgpt_conv2d_bx3_fwd");
    return VGPT_OK;
}

VGPT_EXPORT int64_t vgpt_conv1x1_bx3_packed_bytes(int Cout, int Cin) {
    return (int64_t)cdiv(Cout, TCO) * cdiv(Cin, B1_CK) * B1_IMG;
}

VGPT_EXPORT int vgpt_conv1x1_pack_weights_bx3(const float* w, void* packed, int Cout, int Cin, void* stream) {
    VGPT_REQUIRE(w && packed && Cout > 0 && Cin > 0, VGPT_ERR_INVALID, "vgpt_conv1x1_pack_weights_bx3: bad argument");
    const int tiles_co = (int)cdiv(Cout, TCO), nch = (int)cdiv(Cin, B1_CK);
    hipError_t e = hipMemsetAsync(packed, 0, vgpt_conv1x1_bx3_packed_bytes(Cout, Cin), (hipStream_t)stream);
    if (e != hipSuccess) {
        vgpt_set_error("vgpt_conv1x1_pack_weights_bx3: memset: %s", hipGetErrorString(e));
        return VGPT_ERR_HIP;
    }
    const int64_t n = (int64_t)tiles_co * nch * TCO * B1_CK;
    hipLaunchKernelGGL(conv_pack_b1_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, w, (char*)packed, Cout,
                       Cin, nch, tiles_co);
    VGPT_CHECK_LAUNCH("vgpt_conv1x1_pack_weights_bx3");
    return VGPT_OK;
}

VGPT_EXPORT int vgpt_conv1x1_bx3_fwd(const float* x, const void* packed, const float* bias, const float* resid,
                                     const float* gn_stats, const float* gn_gamma, const float* gn_beta, float* y, int N,
                                     int Cin, int HW, int Cout, int gn_groups, int gn_silu, void* stream) {
    VGPT_REQUIRE(x && packed && y, VGPT_ERR_INVALID, "vgpt_conv1x1_bx3_fwd: null pointer");
    VGPT_REQUIRE(N >= 0 && Cin > 0 && HW > 0 && Cout > 0, VGPT_ERR_INVALID, "vgpt_conv1x1_bx3_fwd: bad shape");
    VGPT_REQUIRE(Cin % B1_CK == 0, VGPT_ERR_UNSUPPORTED, "vgpt_conv1x1_bx3_fwd: Cin must be a multiple of %d (got %d)", B1_CK, Cin);
    VGPT_REQUIRE(gn_groups == 0 || (gn_stats && gn_gamma && gn_beta && Cin % gn_groups == 0), VGPT_ERR_INVALID,
                 "vgpt_conv1x1_bx3_fwd: GroupNorm prologue needs stats/gamma/beta and Cin %% groups == 0");
    VGPT_REQUIRE(((uintptr_t)packed & 15) == 0, VGPT_ERR_UNSUPPORTED, "vgpt_conv1x1_bx3_fwd: packed weights must be 16-byte aligned");
    VGPT_REQUIRE((int64_t)Cin * HW < (1ll << 31), VGPT_ERR_UNSUPPORTED, "vgpt_conv1x1_bx3_fwd: image too large");
    if (N == 0) return VGPT_OK;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)conv1x1_bx3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, B1_LDS_TOTAL);
        if (e != hipSuccess) {
            vgpt_set_error("vgpt_conv1x1_bx3_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
            return VGPT_ERR_HIP;
        }
        attr_set = true;
    }
    Conv1Args a;
    a.x = x; a.wimg = (const char*)packed; a.bias = bias; a.resid = resid;
    a.gn_stats = gn_stats; a.gn_gamma = gn_gamma; a.gn_beta = gn_beta; a.y = y;
    a.N = N; a.Cin = Cin; a.nch = Cin / B1_CK; a.HW = HW; a.Cout = Cout;
    a.gn_groups = gn_groups; a.gn_silu = gn_silu;
    a.tiles_p = (int)cdiv(HW, B1_TP); a.tiles_co = (int)cdiv(Cout, TCO);
    const int64_t blocks = (int64_t)a.tiles_p * a.tiles_co * N;
    hipLaunchKernelGGL(conv1x1_bx3_kernel, dim3((unsigned)blocks), dim3(512), B1_LDS_TOTAL, (hipStream_t)stream, a);
    VGPT_CHECK_LAUNCH("vgpt_conv1x1_bx3_fwd");
    return VGPT_OK;
}

VGPT_EXPORT int vgpt_col_softmax(float* s, int N, int keys, int queries, float scale, void* stream) {
    VGPT_REQUIRE(s, VGPT_ERR_INVALID, "vgpt_col_softmax: null pointer");
    VGPT_REQUIRE(N >= 0 && keys > 0 && queries > 0, VGPT_ERR_INVALID, "vgpt_col_softmax: bad shape");
    if (N == 0) return VGPT_OK;
    hipLaunchKernelGGL(col_softmax_kernel, dim3((unsigned)cdiv(queries, 64), N), dim3(256), 0, (hipStream_t)stream, s,
                       keys, queries, scale);
    VGPT_CHECK_LAUNCH("vgpt_col_softmax");
    return VGPT_OK;
}

VGPT_EXPORT int vgpt_vae_sample(const float* moments, const float* noise, float* z, int N, int64_t per_image,
                                float shift, float scaling, void* stream) {
    VGPT_REQUIRE(moments && noise && z, VGPT_ERR_INVALID, "vgpt_vae_sample: null pointer");
    VGPT_REQUIRE(N >= 0 && per_image > 0, VGPT_ERR_INVALID, "vgpt_vae_sample: bad shape");
    if (N == 0) return VGPT_OK;
    const int64_t total = (int64_t)N * per_image;
    hipLaunchKernelGGL(vae_sample_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, moments,
                       noise, z, per_image, total, shift, scaling);
    VGPT_CHECK_LAUNCH("vgpt_vae_sample");
    return VGPT_OK;
}

VGPT_EXPORT int vgpt_vae_postprocess_u8(const float* x, uint8_t* out, int N, int C, int H, int W, void* stream) {
    VGPT_REQUIRE(x && out, VGPT_ERR_INVALID, "vgpt_vae_postprocess_u8: null pointer");
    VGPT_REQUIRE(N >= 0 && C > 0 && H > 0 && W > 0, VGPT_ERR_INVALID, "vgpt_vae_postprocess_u8: bad shape");
    if (N == 0) return VGPT_OK;
    const int64_t total = (int64_t)N * C * H * W;
    hipLaunchKernelGGL(vae_post_u8_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, out,
                       N, C, H, W);
    VGPT_CHECK_LAUNCH("vgpt_vae_postprocess_u8");
    return VGPT_OK;
}

VGPT_EXPORT int vgpt_affine_to_f32(const void* x, int x_is_bf16, float* y, int64_t n, float mul, float add,
                                   void* stream) {
    VGPT_REQUIRE(x && y, VGPT_ERR_INVALID, "vgpt_affine_to_f32: null pointer");
    VGPT_REQUIRE(n >= 0, VGPT_ERR_INVALID, "vgpt_affine_to_f32: bad shape");
    if (n == 0) return VGPT_OK;
    hipLaunchKernelGGL(affine_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, x_is_bf16, y,
                       n, mul, add);
    VGPT_CHECK_LAUNCH("vgpt_affine_to_f32");
    return VGPT_OK;
}
