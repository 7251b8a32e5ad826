// Attention-mask preprocessing: (B,L,L) masks -> bit-packed rows + per-tile summary.
//
// The reference materialises a dense additive (B,1,L,L) mask in the model dtype on every forward
// (OmniGen/transformer.py:139-145) from the bool masks of LVM/processor.py:575-731. Here the mask
// is packed once per clip to 1 bit per (query, key) and summarised per 32x64 tile so that the
// attention kernel skips masked tiles and reads bits only for mixed ones; any mask stays exact.
#include "common.h"

namespace {

// One wave packs 64 consecutive keys of one query row per iteration (ballot).
template <typename T>
__global__ __launch_bounds__(256) void mask_pack_kernel(const T* __restrict__ mask,
                                                        uint32_t* __restrict__ bits, int64_t rows,
                                                        int L, int W) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int chunks = (L + 63) / 64;
    const int64_t total = rows * chunks;
    for (int64_t it = (int64_t)blockIdx.x * 4 + wave; it < total; it += (int64_t)gridDim.x * 4) {
        const int64_t row = it / chunks;
        const int c = (int)(it % chunks);
        const int key = c * 64 + lane;
        bool vis = false;
        if (key < L) {
            if constexpr (sizeof(T) == 1) {
                vis = mask[row * L + key] != 0;
            } else {
                // additive mask: 0 where visible, finfo.min where masked
                vis = (float)mask[row * L + key] > -1.0f;
            }
        }
        const unsigned long long bal = __ballot(vis);
        if (lane == 0) {
            bits[row * W + 2 * c] = (uint32_t)bal;
            if (2 * c + 1 < W) bits[row * W + 2 * c + 1] = (uint32_t)(bal >> 32);
        }
    }
}

// Mask rows straight from per-token attributes (video-gpt_amd/layout.py; the rule of LVM/processor.py:575-731 in closed
// form).  attr[t] = { (thr of a CLEAN token | sub of a NOISY one) | seq << 24, kind | oc << 2 | grp << 4 }.  Block = 256 consecutive query rows x 4 words
// (128 keys): the key attributes are the same for every lane (scalar loads), each lane keeps its own row's attributes
// in registers and writes its 4 words.
enum { TK_PAD = 0, TK_CLEAN = 1, TK_NOISY = 2, TK_GAP = 3 };
__global__ __launch_bounds__(256) void mask_tokens_kernel(const uint2* __restrict__ attr, uint32_t* __restrict__ bits,
                                                          int L, int W, int wgroups) {
    const int b = blockIdx.y;
    const int q = (int)(blockIdx.x / (unsigned)wgroups) * 256 + threadIdx.x;
    const int w0 = (int)(blockIdx.x % (unsigned)wgroups) * 4;
    const uint2 aq = attr[(int64_t)b * L + min(q, L - 1)];
    const uint32_t kq = aq.y & 3u, ocq = (aq.y >> 2) & 3u, gq = aq.y >> 4, sq = aq.x >> 24, subq = aq.x & 0xffffffu;
    const uint2* ak = attr + (int64_t)b * L;
    uint32_t word[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k0 = (w0 + i) * 32;
        const int nk = min(32, L - k0);  // <= 0 for words past the row (not stored)
        uint32_t wd = 0u;
        for (int j = 0; j < nk; ++j) {
            const uint2 a = ak[k0 + j];  // wave-uniform address
            const uint32_t kk = a.y & 3u;
            const bool clean = kk == TK_CLEAN && (a.x >> 24) == sq && (uint32_t)q >= (a.x & 0xffffffu);
            const uint32_t subk = a.x & 0xffffffu;   // sub-group of a NOISY key: 0 = seen by every sub-group of its clip
            const bool noisy = kk == TK_NOISY && kq == TK_NOISY && (a.y >> 4) == gq && ocq >= ((a.y >> 2) & 3u) &&
                               (subk == 0u || subk == subq);
            wd |= (uint32_t)(clean || noisy) << j;
        }
        if (kq == TK_PAD) wd = nk >= 32 ? 0xffffffffu : (nk > 0 ? (1u << nk) - 1u : 0u);  // pad rows see the whole row
        word[i] = wd;
    }
    if (q >= L) return;
    uint32_t* out = bits + ((int64_t)b * L + q) * W + w0;
    if ((W & 3) == 0) {
        *reinterpret_cast<uint4*>(out) = uint4{word[0], word[1], word[2], word[3]};
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (w0 + i < W) out[i] = word[i];
    }
}

// block = 4 waves = the four 32-row sub-blocks of one (b, 128-row q block, 64-key tile)
__global__ __launch_bounds__(256) void mask_summary_kernel(const uint32_t* __restrict__ bits,
                                                           uint8_t* __restrict__ summary, int L,
                                                           int W, int nqb, int nkt) {
    __shared__ int codes[4];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int kt = blockIdx.x, qb = blockIdx.y, b = blockIdx.z;
    const int q = qb * 128 + wave * 32 + (lane & 31);
    const int w = 2 * kt + (lane >> 5);
    // expected all-visible pattern of this word (keys past L can never be visible)
    uint32_t full = 0u;
    {
        const int k0 = w * 32;
        if (k0 + 32 <= L) full = 0xffffffffu;
        else if (k0 < L) full = (1u << (L - k0)) - 1u;
    }
    bool is_full = true, is_zero = true;
    if (q < L) {
        const uint32_t word = (w < W) ? bits[((int64_t)b * L + q) * W + w] : 0u;
        is_zero = (word == 0u);
        is_full = (word == 0xffffffffu) && (full == 0xffffffffu);
    }
    const bool all_zero = __all(is_zero);
    const bool all_full = __all(is_full);
    if (lane == 0) codes[wave] = all_zero ? 0 : (all_full ? 1 : 2);
    __syncthreads();
    if (threadIdx.x == 0)
        summary[((int64_t)b * nqb + qb) * nkt + kt] =
            (uint8_t)(codes[0] | (codes[1] << 2) | (codes[2] << 4) | (codes[3] << 6));
}

__global__ void mask_empty_rows_kernel(const uint32_t* __restrict__ bits, int32_t* count,
                                       int64_t rows, int W) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    uint32_t any = 0;
    for (int w = 0; w < W; ++w) any |= bits[row * W + w];
    if (!any) atomicAdd(count, 1);
}

// Longest-first launch order of the 128-row q blocks of one batch item: cost = key tiles with any visible key
// (what a workgroup of the attention kernel iterates over).  Rank sort, stable; more than ORDER_MAX q blocks
// keep their natural order.
constexpr int ORDER_MAX = 2048;
__global__ void qblock_order_kernel(const uint8_t* __restrict__ summary, int nqb, int nkt, int qb0,
                                    int32_t* __restrict__ order) {
    __shared__ int cnt[ORDER_MAX];
    const int b = blockIdx.x, n = nqb - qb0;
    int32_t* out = order + (int64_t)b * n;
    if (n > ORDER_MAX) {
        for (int i = threadIdx.x; i < n; i += blockDim.x) out[i] = qb0 + i;
        return;
    }
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const uint8_t* row = summary + ((int64_t)b * nqb + qb0 + i) * nkt;
        int c = 0;
        for (int t = 0; t < nkt; ++t) c += row[t] != 0;
        cnt[i] = c;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int ci = cnt[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += (cnt[j] > ci) || (cnt[j] == ci && j < i);
        out[rank] = qb0 + i;
    }
}

}  // namespace

VGPT_EXPORT int vgpt_mask_pack_bool(const uint8_t* mask, uint32_t* bits, int64_t B, int64_t L,
                                    void* stream) {
    VGPT_REQUIRE(mask && bits, VGPT_ERR_INVALID, "vgpt_mask_pack_bool: null pointer");
    VGPT_REQUIRE(B >= 0 && L >= 0 && L < (1 << 24), VGPT_ERR_INVALID, "vgpt_mask_pack_bool: bad shape");
    if (B == 0 || L == 0) return VGPT_OK;
    const int W = (int)cdiv(L, 32);
    const int64_t total = B * L * cdiv(L, 64);
    int grid = (int)std::min<int64_t>(cdiv(total, 4), 256 * 16);
    hipLaunchKernelGGL(mask_pack_kernel<uint8_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       mask, bits, B * L, (int)L, W);
    VGPT_CHECK_LAUNCH("vgpt_mask_pack_bool");
    return VGPT_OK;
}

VGPT_EXPORT int vgpt_mask_pack_additive(const void* mask, int is_f32, uint32_t* bits, int64_t B,
                                        int64_t L, void* stream) {
    VGPT_REQUIRE(mask && bits, VGPT_ERR_INVALID, "vgpt_mask_pack_additive: null pointer");
    VGPT_REQUIRE(B >= 0 && L >= 0 && L < (1 << 24), VGPT_ERR_INVALID,
                 "vgpt_mask_pack_additive: bad shape");
    if (B == 0 || L == 0) return VGPT_OK;
    const int W = (int)cdiv(L, 32);
    const int64_t total = B * L * cdiv(L, 64);
    int grid = (int)std::min<int64_t>(cdiv(total, 4), 256 * 16);
    if (is_f32)
        hipLaunchKernelGGL(mask_pack_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                           (const float*)mask, bits, B * L, (int)L, W);
    else
        hipLaunchKernelGGL(mask_pack_kernel<bf16>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                           (const bf16*)mask, bits, B * L, (int)L, W);
    VGPT_CHECK_LAUNCH("vgpt_mask_pack_additive");
    return VGPT_OK;
}

VGPT_EXPORT int vgpt_mask_build_tokens(const int32_t* attr, uint32_t* bits, int64_t B, int64_t L, void* stream) {
    VGPT_REQUIRE(attr && bits, VGPT_ERR_INVALID, "vgpt_mask_build_tokens: null pointer");
    VGPT_REQUIRE(B >= 0 && B < 65536 && L >= 0 && L < (1 << 24), VGPT_ERR_INVALID, "vgpt_mask_build_tokens: bad shape");
    VGPT_REQUIRE(((uintptr_t)attr & 7) == 0, VGPT_ERR_UNSUPPORTED, "vgpt_mask_build_tokens: attr must be 8-byte aligned");
    if (B == 0 || L == 0) return VGPT_OK;
    const int W = (int)cdiv(L, 32), wgroups = (int)cdiv(W, 4);
    VGPT_REQUIRE(cdiv(L, 256) * wgroups < (1ll << 31), VGPT_ERR_UNSUPPORTED, "vgpt_mask_build_tokens: L too large");
    VGPT_REQUIRE(((uintptr_t)bits & 15) == 0, VGPT_ERR_UNSUPPORTED, "vgpt_mask_build_tokens: bits must be 16-byte aligned");
    hipLaunchKernelGGL(mask_tokens_kernel, dim3((unsigned)(cdiv(L, 256) * wgroups), (unsigned)B), dim3(256), 0,
                       (hipStream_t)stream, (const uint2*)attr, bits, (int)L, W, wgroups);
    VGPT_CHECK_LAUNCH("vgpt_mask_build_tokens");
    return VGPT_OK;
}

VGPT_EXPORT int vgpt_mask_tile_summary(const uint32_t* bits, uint8_t* summary, int64_t B, int64_t L,
                                       void* stream) {
    VGPT_REQUIRE(bits && summary, VGPT_ERR_INVALID, "vgpt_mask_tile_summary: null pointer");
    VGPT_REQUIRE(B >= 0 && L >= 0 && L < (1 << 24) && B < 65536, VGPT_ERR_INVALID,
                 "vgpt_mask_tile_summary: bad shape");
    if (B == 0 || L == 0) return VGPT_OK;
    const int nqb = (int)cdiv(L, 128), nkt = (int)cdiv(L, 64);
    VGPT_REQUIRE(nqb < 65536, VGPT_ERR_UNSUPPORTED, "vgpt_mask_tile_summary: L too large");
    hipLaunchKernelGGL(mask_summary_kernel, dim3(nkt, nqb, (unsigned)B), dim3(256), 0,
                       (hipStream_t)stream, bits, summary, (int)L, (int)cdiv(L, 32), nqb, nkt);
    VGPT_CHECK_LAUNCH("vgpt_mask_tile_summary");
    return VGPT_OK;
}

VGPT_EXPORT int vgpt_attn_qblock_order(const uint8_t* summary, int64_t B, int64_t L, int64_t q_start, int32_t* order,
                                       void* stream) {
    VGPT_REQUIRE(summary && order, VGPT_ERR_INVALID, "vgpt_attn_qblock_order: null pointer");
    VGPT_REQUIRE(B >= 0 && L >= 0 && L < (1 << 24) && q_start >= 0 && q_start % 128 == 0 && q_start <= L,
                 VGPT_ERR_INVALID, "vgpt_attn_qblock_order: bad shape");
    const int nqb = (int)cdiv(L, 128), nkt = (int)cdiv(L, 64), qb0 = (int)(q_start / 128);
    if (B == 0 || nqb == qb0) return VGPT_OK;
    hipLaunchKernelGGL(qblock_order_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, summary, nqb, nkt, qb0,
                       order);
    VGPT_CHECK_LAUNCH("vgpt_attn_qblock_order");
    return VGPT_OK;
}

VGPT_EXPORT int vgpt_mask_count_empty_rows(const uint32_t* bits, int32_t* count, int64_t B,
                                           int64_t L, void* stream) {
    VGPT_REQUIRE(bits && count, VGPT_ERR_INVALID, "vgpt_mask_count_empty_rows: null pointer");
    VGPT_REQUIRE(B >= 0 && L >= 0, VGPT_ERR_INVALID, "vgpt_mask_count_empty_rows: bad shape");
    hipError_t e = hipMemsetAsync(count, 0, sizeof(int32_t), (hipStream_t)stream);
    if (e != hipSuccess) {
        vgpt_set_error("vgpt_mask_count_empty_rows: memset: %s", hipGetErrorString(e));
        return VGPT_ERR_HIP;
    }
    if (B == 0 || L == 0) return VGPT_OK;
    const int64_t rows = B * L;
    hipLaunchKernelGGL(mask_empty_rows_kernel, dim3((unsigned)cdiv(rows, 256)), dim3(256), 0,
                       (hipStream_t)stream, bits, count, rows, (int)cdiv(L, 32));
    VGPT_CHECK_LAUNCH("vgpt_mask_count_empty_rows");
    return VGPT_OK;
}
