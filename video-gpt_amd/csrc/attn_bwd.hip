// Block-masked flash attention BACKWARD for gfx950 (head_dim 96, bf16 in/out, fp32 softmax math).
//
// The reference obtains these gradients from torch.autograd through F.scaled_dot_product_attention
// (LVM/transform/sdpa_transform.py:78-86,152 inside accelerator.backward, train_x1_stage1_noiseinput.py:380).
//
// With P = exp2(c*S - LSE) (c = scale*log2 e, LSE from the forward), delta_q = sum_d dO[q][d] O[q][d]:
//     dV = P^T dO            dP = dO V^T           dS = P o (dP - delta) * scale
//     dQ = dS K              dK = dS^T Q
// Two kernels, both recomputing S and dP (no atomics, deterministic):
//   attn_bwd_dq_kernel  — a wave owns 32 QUERY rows and walks the key tiles exactly like the forward:
//                         S^T = K Q^T, dP^T = V dO^T (keys on accumulator rows, the query on the lane, so
//                         LSE/delta are per-lane scalars), dQ^T += K^T dS^T with K^T from ds_read_b64_tr_b16.
//   attn_bwd_dkv_kernel — a wave owns 32 KEYS and walks the query tiles: S = Q K^T, dP = dO V^T (queries on
//                         accumulator rows, the key on the lane), dV^T += dO^T P, dK^T += Q^T dS with the
//                         transposed operands again from ds_read_b64_tr_b16 on the [row][d] LDS images.
// Every LDS image is a contiguous [row][192 B] tile filled by LDS-DMA with the chunk XOR-swizzle
// (chunk ^= (row>>2)&3) that makes ds_read_b128 row reads conflict-free and leaves the 64-byte windows of
// the transposed reads intact (the XOR is constant over the 4 rows of a transposed-read block).
#include "common.h"

namespace {

constexpr int D = 96, KS = D / 16, DT = D / 32, CHUNKS = D / 8, ROWB = D * 2;
constexpr int TILE_BYTES = 64 * ROWB;  // 12 KiB image of 64 rows
constexpr int PIECES = TILE_BYTES / 1024 / 4;

struct BwdArgs {
    const bf16 *q, *k, *v, *o, *dout;
    bf16 *dq, *dk, *dv;
    const float* lse;    // (B, n_heads, L) base-2
    const float* delta;  // (B, n_heads, L)
    const uint32_t* bits;
    const uint8_t* summary;
    int B, L, n_heads, n_kv_heads, kv_group, W, nqb, nkt;
    int64_t q_sb, q_sh, q_ss, k_sb, k_sh, k_ss, v_sb, v_sh, v_ss, o_sb, o_sh, o_ss, do_sb, do_sh, do_ss;
    int64_t dq_sb, dq_sh, dq_ss, dk_sb, dk_sh, dk_ss, dv_sb, dv_sh, dv_ss;
    float scale, scale_log2e;
};

// LDS-DMA from inline asm (lane i's 4 / 16 bytes land at lds_dst + 4 i / 16 i): hipcc does not see LDS being written,
// so it neither drains vmcnt(0) before the transposed LDS reads nor at the barrier; completion is tracked by the
// explicit s_waitcnt vmcnt(0) + raw s_barrier at the top of every tile (same scheme as attn_fwd.hip).
__device__ __forceinline__ void dma4(const void* src, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(lds_dst)
                 : "memory");
}
__device__ __forceinline__ void dma16(const bf16* src, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(lds_dst)
                 : "memory");
}
__device__ __forceinline__ void tile_sync() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
constexpr int LIST_MAX = 1023;  // active tiles per list chunk (4 KiB of LDS)
// VGPT_BWD_EXPERIMENT (diagnostic builds of the dK/dV kernels, results are WRONG): 1 = no element-wise math,
// 2 = transposed fragments read once, 3 = no statistics / mask DMA, 4 = no Q/dO DMA after the first tile
#ifndef VGPT_BWD_EXPERIMENT
#define VGPT_BWD_EXPERIMENT 0
#endif

// row read of a swizzled [row][192 B] image: lane (r = row, h) gets elements [16s + 8h, +8)
__device__ __forceinline__ bf16x8 row_frag(const char* img, int row, int s, int h) {
    const int kc = (2 * s + h) ^ ((row >> 2) & 3);
    return *reinterpret_cast<const bf16x8*>(img + row * ROWB + kc * 16);
}

// transposed read: A operand of a 32x32x16 MFMA whose rows are d (32-wide tile dt) and whose k index runs
// over 16 image rows starting at row16 (+4h, +8 for the upper half): element j <-> row row16 + 8(j>>2) + 4h + (j&3)
__device__ __forceinline__ bf16x8 tr_frag(const char* img, int row16, int dt, int lane) {
    const int h = lane >> 5, li = lane & 15;
    const int row = row16 + 4 * h + (li >> 2);
    const int dcol = dt * 32 + ((lane >> 4) & 1) * 16 + 4 * (li & 3);  // first of 4 consecutive d
    const int f0 = (row >> 2) & 3, f1 = ((row + 8) >> 2) & 3;
    const char* p0 = img + row * ROWB + (((dcol >> 3) ^ f0) * 16) + (dcol & 4) * 2;
    const char* p1 = img + (row + 8) * ROWB + (((dcol >> 3) ^ f1) * 16) + (dcol & 4) * 2;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p0);
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p1);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

__device__ __forceinline__ bf16x8 pack8(const f32x16& x, int half) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = f2bf(x[8 * half + j]);
    return r;
}

// delta[b][h][q] = sum_d dO[q][d] * O[q][d]
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16* __restrict__ o, const bf16* __restrict__ dout,
                                                         float* __restrict__ delta, int B, int L, int n_heads,
                                                         int64_t o_sb, int64_t o_sh, int64_t o_ss, int64_t do_sb,
                                                         int64_t do_sh, int64_t do_ss) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)B * n_heads * L) return;
    // head fastest: the 64 lanes of a wave read 64 consecutive 192-byte head slices (two whole token rows of the
    // (B, L, heads*d) layout) instead of 64 slices 6 KB apart
    const int hd = (int)(idx % n_heads);
    const int q = (int)((idx / n_heads) % L);
    const int b = (int)(idx / ((int64_t)L * n_heads));
    const bf16* op = o + b * o_sb + hd * o_sh + (int64_t)q * o_ss;
    const bf16* dp = dout + b * do_sb + hd * do_sh + (int64_t)q * do_ss;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(op + c * 8);
        const bf16x8 d = *reinterpret_cast<const bf16x8*>(dp + c * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += bf2f(a[j]) * bf2f(d[j]);
    }
    delta[((int64_t)b * n_heads + hd) * L + q] = s;
}

// ------------------------------------------------------------------------------------------------------
// dQ: block = 4 waves = 128 query rows of one (batch, head); K/V tiles of 64 keys double-buffered in LDS
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(BwdArgs a) {
    constexpr int STAGE = 2 * TILE_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int total = a.nqb * a.n_heads * a.B;
    int wid = blockIdx.x;
    {
        const int xcd = wid & 7, qn = total >> 3, rn = total & 7;
        wid = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (wid >> 3);
    }
    const int qb = wid % a.nqb, head = (wid / a.nqb) % a.n_heads, b = wid / (a.nqb * a.n_heads);
    const int kvh = head / a.kv_group;
    const uint8_t* sum_row = a.summary + ((int64_t)b * a.nqb + qb) * a.nkt;
    const bf16* kbase = a.k + (int64_t)b * a.k_sb + (int64_t)kvh * a.k_sh;
    const bf16* vbase = a.v + (int64_t)b * a.v_sb + (int64_t)kvh * a.v_sh;

    const int q_row = qb * 128 + wave * 32 + r;
    const int q_ld = min(q_row, a.L - 1);
    const bf16* qp = a.q + (int64_t)b * a.q_sb + (int64_t)head * a.q_sh + (int64_t)q_ld * a.q_ss;
    const bf16* dop = a.dout + (int64_t)b * a.do_sb + (int64_t)head * a.do_sh + (int64_t)q_ld * a.do_ss;
    bf16x8 Qf[KS], dOf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        Qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s + 8 * h);
        dOf[s] = *reinterpret_cast<const bf16x8*>(dop + 16 * s + 8 * h);
    }
    const int64_t stat = ((int64_t)b * a.n_heads + head) * a.L + q_ld;
    const float lse = a.lse[stat], dlt = a.delta[stat];

    f32x16 dQ[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) dQ[dt][i] = 0.f;

    int g_key[PIECES], g_chunk[PIECES];
#pragma unroll
    for (int j = 0; j < PIECES; ++j) {
        const int unit = (wave * PIECES + j) * 64 + lane;
        g_key[j] = unit / CHUNKS;
        g_chunk[j] = (unit % CHUNKS) ^ ((g_key[j] >> 2) & 3);
    }
    const uint32_t lds_base =
        __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
    constexpr int LIST_OFF = 2 * STAGE, MASK_OFF = LIST_OFF + 4096;
    uint32_t* alist = reinterpret_cast<uint32_t*>(smem + LIST_OFF);
    auto stage = [&](int buf, int kt) {
        const uint32_t sk = lds_base + buf * STAGE;
#pragma unroll
        for (int j = 0; j < PIECES; ++j) {
            const int key = min(kt * 64 + g_key[j], a.L - 1);
            const int off = (wave * PIECES + j) * 1024;
            dma16(kbase + (int64_t)key * a.k_ss + g_chunk[j] * 8, sk + off);
            dma16(vbase + (int64_t)key * a.v_ss + g_chunk[j] * 8, sk + TILE_BYTES + off);
        }
    };
    // mask words of a mixed tile: one dword per lane (row lane>>1, word lane&1) into this wave's 256-byte slot
    const uint32_t* mrow_src = a.bits + ((int64_t)b * a.L + min(qb * 128 + wave * 32 + (lane >> 1), a.L - 1)) * a.W;
    auto mask_dma = [&](int buf, uint32_t e) {
        if (((e >> (2 * wave)) & 3) == 2)
            dma4(mrow_src + min(2 * (int)(e >> 8) + (lane & 1), a.W - 1), lds_base + MASK_OFF + buf * 1024 + wave * 256);
    };
    // everything loaded with ordinary global loads is retired here, before any LDS-DMA is in flight
#pragma unroll
    for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(Qf[s]), "+v"(dOf[s]));
    float lse_r = lse, dlt_r = dlt;
    asm volatile("" : "+v"(lse_r), "+v"(dlt_r));

    for (int chunk0 = 0; chunk0 < a.nkt; chunk0 += LIST_MAX) {
        __syncthreads();
        if (wave == 0) {  // compact the active key tiles of this q block into LDS: entry = tile << 8 | summary byte
            const int lim = min(chunk0 + LIST_MAX, a.nkt);
            int n = 0;
            for (int base = chunk0; base < lim; base += 64) {
                const int t = base + lane;
                const uint32_t c = t < lim ? sum_row[t] : 0u;
                const uint64_t bal = __ballot(c != 0);
                if (c) alist[1 + n + __popcll(bal & ((1ull << lane) - 1))] = ((uint32_t)t << 8) | c;
                n += __popcll(bal);
            }
            if (lane == 0) alist[0] = (uint32_t)n;
        }
        __syncthreads();
        const int n_act = __builtin_amdgcn_readfirstlane((int)alist[0]);
        if (n_act == 0) continue;
        uint32_t e_cur = __builtin_amdgcn_readfirstlane(alist[1]);
        uint32_t e_nxt = __builtin_amdgcn_readfirstlane(n_act > 1 ? alist[2] : 0u);
        int buf = 0;
        mask_dma(0, e_cur);
        stage(0, (int)(e_cur >> 8));
        for (int it = 0; it < n_act; ++it) {
            tile_sync();
            const uint32_t e_n2 = it + 2 < n_act ? alist[3 + it] : 0u;
            if (it + 1 < n_act) {
                mask_dma(buf ^ 1, e_nxt);
                stage(buf ^ 1, (int)(e_nxt >> 8));
            }
            const int code = (e_cur >> (2 * wave)) & 3;
            if (code) {
                const char* sk = smem + buf * STAGE;
                const char* sv = sk + TILE_BYTES;
                uint32_t mw0 = 0xffffffffu, mw1 = 0xffffffffu;
                if (code == 2) {
                    const uint2 mw = *reinterpret_cast<const uint2*>(smem + MASK_OFF + buf * 1024 + wave * 256 + r * 8);
                    mw0 = mw.x;
                    mw1 = (2 * (int)(e_cur >> 8) + 1 < a.W) ? mw.y : 0u;
                }
                // the two 32-key halves of the tile are processed one after the other (register budget: 2 waves/SIMD)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    bf16x8 Kf[KS], Vf[KS];
#pragma unroll
                    for (int s = 0; s < KS; ++s) {
                        Kf[s] = row_frag(sk, kb * 32 + r, s, h);
                        Vf[s] = row_frag(sv, kb * 32 + r, s, h);
                    }
                    // K^T fragments for dQ^T += K^T dS^T (independent of the element-wise math below)
                    bf16x8 Kt[DT][2];
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                        for (int t = 0; t < 2; ++t) Kt[dt][t] = tr_frag(sk, kb * 32 + t * 16, dt, lane);
                    __builtin_amdgcn_sched_barrier(0);
                    f32x16 S, dP;
#pragma unroll
                    for (int i = 0; i < 16; ++i) { S[i] = 0.f; dP[i] = 0.f; }
#pragma unroll
                    for (int s = 0; s < KS; ++s) {
                        S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Kf[s], Qf[s], S, 0, 0, 0);
                        dP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Vf[s], dOf[s], dP, 0, 0, 0);
                    }
                    // dS^T / scale = P^T o (dP^T - delta), P^T = exp2(c S^T - LSE); masked keys -> 0
                    const uint32_t w = (kb ? mw1 : mw0) >> (4 * h);
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int bit = (i & 3) + 8 * (i >> 2);
                        float p = __builtin_amdgcn_exp2f(__builtin_fmaf(S[i], a.scale_log2e, -lse_r));
                        // masked keys -> 0 without VCC: the bit spread to 0 / ~0 (v_bfe_i32) and one AND; an all-visible
                        // tile carries all-ones words, so no branch and no select
                        // (the empty asm keeps instcombine from turning the AND back into a select)
                        uint32_t keep = (uint32_t)((int32_t)(w << (31 - bit)) >> 31);
                        asm("" : "+v"(keep));
                        p = __uint_as_float(__float_as_uint(p) & keep);
                        S[i] = p * (dP[i] - dlt_r);   // the softmax scale multiplies dQ once, in the epilogue
                    }
                    bf16x8 dSf[2];
#pragma unroll
                    for (int t = 0; t < 2; ++t) dSf[t] = pack8(S, t);
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                        for (int t = 0; t < 2; ++t)
                            dQ[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Kt[dt][t], dSf[t], dQ[dt], 0, 0, 0);
                }
            }
            e_cur = e_nxt;
            e_nxt = __builtin_amdgcn_readfirstlane(e_n2);
            buf ^= 1;
        }
    }
    if (q_row < a.L) {
        bf16* op = a.dq + (int64_t)b * a.dq_sb + (int64_t)head * a.dq_sh + (int64_t)q_row * a.dq_ss;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                bf16x4 o;
#pragma unroll
                for (int t = 0; t < 4; ++t) o[t] = f2bf(dQ[dt][4 * g4 + t] * a.scale);
                *reinterpret_cast<bf16x4*>(op + dt * 32 + 8 * g4 + 4 * h) = o;
            }
    }
}

// ------------------------------------------------------------------------------------------------------
// dK, dV: block = 4 waves = 128 keys of one (batch, kv head); Q/dO tiles of 64 query rows in LDS
// ------------------------------------------------------------------------------------------------------
// WANT_DK / WANT_DV: both gradients in one launch need ~380 VGPRs (one wave per SIMD); each alone fits the 256-register
// budget of two waves per SIMD at the price of recomputing S = Q K^T.  Both forms are built (see the launch below).
template <bool WANT_DK, bool WANT_DV>
__global__ __launch_bounds__(256, (WANT_DK && WANT_DV) ? 1 : 2) void attn_bwd_dkv_kernel(BwdArgs a) {
    // stage: Q image | dO image | lse[64] | delta[64] | mask words [64][4]
    constexpr int STAT_OFF = 2 * TILE_BYTES, MASK_OFF = STAT_OFF + 512, STAGE = MASK_OFF + 1024;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int nkb = (a.L + 127) / 128;  // 128-key blocks
    const int total = nkb * a.n_kv_heads * a.B;
    int wid = blockIdx.x;
    {
        const int xcd = wid & 7, qn = total >> 3, rn = total & 7;
        wid = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (wid >> 3);
    }
    const int kblk = wid % nkb, kvh = (wid / nkb) % a.n_kv_heads, b = wid / (nkb * a.n_kv_heads);
    const int key_row = kblk * 128 + wave * 32 + r;
    const int key_ld = min(key_row, a.L - 1);
    const bf16* kp = a.k + (int64_t)b * a.k_sb + (int64_t)kvh * a.k_sh + (int64_t)key_ld * a.k_ss;
    const bf16* vp = a.v + (int64_t)b * a.v_sb + (int64_t)kvh * a.v_sh + (int64_t)key_ld * a.v_ss;
    bf16x8 Kf[KS], Vf[KS];  // B operands: lane holds K[key r][16s + 8h .. +8)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        Kf[s] = *reinterpret_cast<const bf16x8*>(kp + 16 * s + 8 * h);
        if constexpr (WANT_DK) Vf[s] = *reinterpret_cast<const bf16x8*>(vp + 16 * s + 8 * h);
    }
    f32x16 dK[DT], dV[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) { dK[dt][i] = 0.f; dV[dt][i] = 0.f; }

    int g_row[PIECES], g_chunk[PIECES];
#pragma unroll
    for (int j = 0; j < PIECES; ++j) {
        const int unit = (wave * PIECES + j) * 64 + lane;
        g_row[j] = unit / CHUNKS;
        g_chunk[j] = (unit % CHUNKS) ^ ((g_row[j] >> 2) & 3);
    }
    const int nqt = (a.L + 63) / 64;
    const uint32_t lds_base =
        __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
    uint32_t* alist = reinterpret_cast<uint32_t*>(smem + 2 * STAGE);
    // the K/V fragment loads are retired here, before any LDS-DMA is in flight
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        asm volatile("" : "+v"(Kf[s]));
        if constexpr (WANT_DK) asm volatile("" : "+v"(Vf[s]));
    }

    for (int gh = 0; gh < a.kv_group; ++gh) {
        const int head = kvh * a.kv_group + gh;
        const bf16* qbase = a.q + (int64_t)b * a.q_sb + (int64_t)head * a.q_sh;
        const bf16* dobase = a.dout + (int64_t)b * a.do_sb + (int64_t)head * a.do_sh;
        const float* lse_b = a.lse + ((int64_t)b * a.n_heads + head) * a.L;
        const float* dlt_b = a.delta + ((int64_t)b * a.n_heads + head) * a.L;
        // 4 code bits (two 32-row halves) of query tile qt against key tile kt
        auto tile_bits = [&](int qt, int kt) -> uint32_t {
            if (kt >= a.nkt) return 0u;
            const uint8_t sbyte = a.summary[((int64_t)b * a.nqb + (qt >> 1)) * a.nkt + kt];
            return (sbyte >> (4 * (qt & 1))) & 15u;
        };
        auto stage = [&](int buf, int qt) {
            const uint32_t sq = lds_base + buf * STAGE;
#pragma unroll
            for (int j = 0; j < PIECES; ++j) {
                const int row = min(qt * 64 + g_row[j], a.L - 1);
                const int off = (wave * PIECES + j) * 1024;
                dma16(qbase + (int64_t)row * a.q_ss + g_chunk[j] * 8, sq + off);
                dma16(dobase + (int64_t)row * a.do_ss + g_chunk[j] * 8, sq + TILE_BYTES + off);
            }
            // row statistics and mask words also arrive by LDS-DMA (4-byte form)
            if (VGPT_BWD_EXPERIMENT == 3) return;
            if (wave == 0) {
                const int row = min(qt * 64 + lane, a.L - 1);
                dma4(lse_b + row, sq + STAT_OFF);
                dma4(dlt_b + row, sq + STAT_OFF + 256);
            } else if (wave == 1) {
                // mask words as [32-key group of the block][query row]: the words a wave needs for four
                // consecutive query rows are then one 16-byte LDS read
                const int row = min(qt * 64 + lane, a.L - 1);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    dma4(a.bits + ((int64_t)b * a.L + row) * a.W + min(kblk * 4 + i, a.W - 1), sq + MASK_OFF + i * 256);
            }
        };

        for (int chunk0 = 0; chunk0 < nqt; chunk0 += LIST_MAX) {
        __syncthreads();  // previous list / buffers are free (also separates the heads)
        if (wave == 0) {  // active query tiles of this key block: entry = qt << 8 | bits(ktile 0) | bits(ktile 1) << 4
            const int lim = min(chunk0 + LIST_MAX, nqt);
            int n = 0;
            for (int base = chunk0; base < lim; base += 64) {
                const int t = base + lane;
                const uint32_t c = t < lim ? (tile_bits(t, kblk * 2) | (tile_bits(t, kblk * 2 + 1) << 4)) : 0u;
                const uint64_t bal = __ballot(c != 0);
                if (c) alist[1 + n + __popcll(bal & ((1ull << lane) - 1))] = ((uint32_t)t << 8) | c;
                n += __popcll(bal);
            }
            if (lane == 0) alist[0] = (uint32_t)n;
        }
        __syncthreads();
        const int n_act = __builtin_amdgcn_readfirstlane((int)alist[0]);
        if (n_act == 0) continue;
        uint32_t e_cur = __builtin_amdgcn_readfirstlane(alist[1]);
        uint32_t e_nxt = __builtin_amdgcn_readfirstlane(n_act > 1 ? alist[2] : 0u);
        int buf = 0;
        stage(0, (int)(e_cur >> 8));
        for (int it = 0; it < n_act; ++it) {
            tile_sync();
            const uint32_t e_n2 = it + 2 < n_act ? alist[3 + it] : 0u;
            if (it + 1 < n_act && VGPT_BWD_EXPERIMENT != 4) stage(buf ^ 1, (int)(e_nxt >> 8));
            const int qt = (int)(e_cur >> 8);
            const int codes = (e_cur >> (4 * (wave >> 1))) & 15;
            const char* sq = smem + buf * STAGE;
            const char* sdo = sq + TILE_BYTES;
            const float* st = reinterpret_cast<const float*>(sq + STAT_OFF);
            const uint32_t* mw = reinterpret_cast<const uint32_t*>(sq + MASK_OFF);
#pragma unroll
            for (int qh = 0; qh < 2; ++qh) {  // two 32-row halves of the query tile
                const int code = (codes >> (2 * qh)) & 3;
                if (!code) continue;
                f32x16 S, dP;
#pragma unroll
                for (int i = 0; i < 16; ++i) { S[i] = 0.f; dP[i] = 0.f; }
                {   // S = Q K^T, then dP = dO V^T: one row-fragment set live at a time (register budget)
                    bf16x8 Qr[KS];
#pragma unroll
                    for (int s = 0; s < KS; ++s) Qr[s] = row_frag(sq, qh * 32 + r, s, h);
#pragma unroll
                    for (int s = 0; s < KS; ++s) S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Qr[s], Kf[s], S, 0, 0, 0);
                }
                if constexpr (WANT_DK) {
                    bf16x8 dOr[KS];
#pragma unroll
                    for (int s = 0; s < KS; ++s) dOr[s] = row_frag(sdo, qh * 32 + r, s, h);
#pragma unroll
                    for (int s = 0; s < KS; ++s) dP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dOr[s], Vf[s], dP, 0, 0, 0);
                }
                // transposed operand for dV^T += dO^T P (k index = query row); requested before the element-wise math
                bf16x8 dOt[DT][2];
                if constexpr (WANT_DV) {
                    if (VGPT_BWD_EXPERIMENT != 2 || it == 0) {
#pragma unroll
                        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                            for (int t = 0; t < 2; ++t) dOt[dt][t] = tr_frag(sdo, qh * 32 + t * 16, dt, lane);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                // accumulator register i holds query row qh*32 + 8 (i>>2) + 4h + (i&3): the row statistics of four
                // consecutive registers are one 16-byte LDS read
                f32x4 lse4[4], dlt4[4];
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    lse4[g4] = *reinterpret_cast<const f32x4*>(st + qh * 32 + 8 * g4 + 4 * h);
                    if constexpr (WANT_DK) dlt4[g4] = *reinterpret_cast<const f32x4*>(st + 64 + qh * 32 + 8 * g4 + 4 * h);
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (VGPT_BWD_EXPERIMENT == 1) continue;
                    dP[i] = WANT_DK ? dP[i] - dlt4[i >> 2][i & 3] : 0.f;                                      // dP - delta
                    S[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(S[i], a.scale_log2e, -lse4[i >> 2][i & 3]));  // P
                }
                if (code == 2) {  // mixed tile: bit r of the word of (query row, this wave's 32 keys)
                    // bit r spread to 0 / ~0 (v_bfe_i32) and one AND per probability, no VCC (as in the dQ kernel)
                    auto keep = [&](float p, uint32_t w) {
                        uint32_t m = (uint32_t)((int32_t)(w << (31 - r)) >> 31);
                        asm("" : "+v"(m));
                        return __uint_as_float(__float_as_uint(p) & m);
                    };
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const uint4 w4 = *reinterpret_cast<const uint4*>(mw + wave * 64 + qh * 32 + 8 * g4 + 4 * h);
                        S[4 * g4 + 0] = keep(S[4 * g4 + 0], w4.x);
                        S[4 * g4 + 1] = keep(S[4 * g4 + 1], w4.y);
                        S[4 * g4 + 2] = keep(S[4 * g4 + 2], w4.z);
                        S[4 * g4 + 3] = keep(S[4 * g4 + 3], w4.w);
                    }
                }
                if (qt * 64 + 64 > a.L || kblk * 128 + 128 > a.L) {  // rows / keys past L (last tiles only)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int qrow = qh * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                        if (key_row >= a.L || qt * 64 + qrow >= a.L) S[i] = 0.f;
                    }
                }
                // S now holds P; dP holds dP - delta.  dS / scale = P (dP - delta) goes to dP, P stays in S.
                if constexpr (WANT_DK) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) dP[i] = S[i] * dP[i];   // the softmax scale multiplies dK once, in the epilogue
                }
                // no MFMA of the products below starts while the packed fp32 instructions above are in flight
                // (scripts/attn_issue_probe.py: a v_pk_* beside an MFMA costs 53 cycles a group instead of 32; hipcc interleaved
                // them): whole backward 866 / 875 against 880 / 881 us at the cfg-3 mask, scripts/attn_bwd_probe.py
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (WANT_DV) {
                    bf16x8 Pf[2];
#pragma unroll
                    for (int t = 0; t < 2; ++t) Pf[t] = pack8(S, t);
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                        for (int t = 0; t < 2; ++t)
                            dV[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dOt[dt][t], Pf[t], dV[dt], 0, 0, 0);
                }
                if constexpr (WANT_DK) {  // dK^T += Q^T dS
                    bf16x8 dSf[2];
#pragma unroll
                    for (int t = 0; t < 2; ++t) dSf[t] = pack8(dP, t);
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) {
                        bf16x8 Qt[2];
#pragma unroll
                        for (int t = 0; t < 2; ++t) Qt[t] = tr_frag(sq, qh * 32 + t * 16, dt, lane);
#pragma unroll
                        for (int t = 0; t < 2; ++t)
                            dK[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Qt[t], dSf[t], dK[dt], 0, 0, 0);
                    }
                }
            }
            e_cur = e_nxt;
            e_nxt = __builtin_amdgcn_readfirstlane(e_n2);
            buf ^= 1;
        }
        }
    }
    if (key_row < a.L) {
        bf16* kp_o = a.dk + (int64_t)b * a.dk_sb + (int64_t)kvh * a.dk_sh + (int64_t)key_row * a.dk_ss;
        bf16* vp_o = a.dv + (int64_t)b * a.dv_sb + (int64_t)kvh * a.dv_sh + (int64_t)key_row * a.dv_ss;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                bf16x4 ok, ov;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    ok[t] = f2bf(dK[dt][4 * g4 + t] * a.scale);
                    ov[t] = f2bf(dV[dt][4 * g4 + t]);
                }
                if constexpr (WANT_DK) *reinterpret_cast<bf16x4*>(kp_o + dt * 32 + 8 * g4 + 4 * h) = ok;
                if constexpr (WANT_DV) *reinterpret_cast<bf16x4*>(vp_o + dt * 32 + 8 * g4 + 4 * h) = ov;
            }
    }
}

}  // namespace

VGPT_EXPORT int vgpt_attn_blockmask_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout,
                                        const float* lse, float* delta_ws, void* dq, void* dk, void* dv,
                                        const uint32_t* bits, const uint8_t* summary, int64_t B, int64_t L, int n_heads,
                                        int n_kv_heads, int head_dim, const int64_t* strides, float scale, void* stream) {
    VGPT_REQUIRE(q && k && v && o && dout && lse && delta_ws && dq && dk && dv && bits && summary && strides,
                 VGPT_ERR_INVALID, "vgpt_attn_blockmask_bwd: null pointer");
    VGPT_REQUIRE(head_dim == 96, VGPT_ERR_UNSUPPORTED, "vgpt_attn_blockmask_bwd: head_dim %d unsupported (96)", head_dim);
    VGPT_REQUIRE(B >= 0 && L >= 0 && n_heads > 0 && n_kv_heads > 0 && n_heads % n_kv_heads == 0 && scale > 0.f,
                 VGPT_ERR_INVALID, "vgpt_attn_blockmask_bwd: bad shape");
    for (int i = 0; i < 24; ++i)
        VGPT_REQUIRE(strides[i] % (i < 15 ? 8 : 4) == 0, VGPT_ERR_UNSUPPORTED,
                     "vgpt_attn_blockmask_bwd: strides must be multiples of 8 (inputs) / 4 (outputs) elements");
    if (B == 0 || L == 0) return VGPT_OK;
    BwdArgs a;
    a.q = (const bf16*)q; a.k = (const bf16*)k; a.v = (const bf16*)v; a.o = (const bf16*)o; a.dout = (const bf16*)dout;
    a.dq = (bf16*)dq; a.dk = (bf16*)dk; a.dv = (bf16*)dv; a.lse = lse; a.delta = delta_ws; a.bits = bits;
    a.summary = summary;
    a.B = (int)B; a.L = (int)L; a.n_heads = n_heads; a.n_kv_heads = n_kv_heads; a.kv_group = n_heads / n_kv_heads;
    a.W = (int)cdiv(L, 32); a.nqb = (int)cdiv(L, 128); a.nkt = (int)cdiv(L, 64);
    const int64_t* s = strides;
    a.q_sb = s[0]; a.q_sh = s[1]; a.q_ss = s[2]; a.k_sb = s[3]; a.k_sh = s[4]; a.k_ss = s[5];
    a.v_sb = s[6]; a.v_sh = s[7]; a.v_ss = s[8]; a.o_sb = s[9]; a.o_sh = s[10]; a.o_ss = s[11];
    a.do_sb = s[12]; a.do_sh = s[13]; a.do_ss = s[14]; a.dq_sb = s[15]; a.dq_sh = s[16]; a.dq_ss = s[17];
    a.dk_sb = s[18]; a.dk_sh = s[19]; a.dk_ss = s[20]; a.dv_sb = s[21]; a.dv_sh = s[22]; a.dv_ss = s[23];
    a.scale = scale; a.scale_log2e = scale * 1.4426950408889634f;
    hipStream_t st = (hipStream_t)stream;
    static bool attr_set = false;
    constexpr int lds_dq = 2 * 2 * TILE_BYTES + 4096 + 2048, lds_dkv = 2 * (2 * TILE_BYTES + 512 + 1024) + 4096;
    if (!attr_set) {
        hipError_t e1 = hipFuncSetAttribute((const void*)attn_bwd_dq_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_dq);
        hipError_t e2 = hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<true, false>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, lds_dkv);
        hipError_t e3 = hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<false, true>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, lds_dkv);
        hipError_t e4 = hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<true, true>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, lds_dkv);
        if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess) {
            vgpt_set_error("vgpt_attn_blockmask_bwd: hipFuncSetAttribute failed");
            return VGPT_ERR_HIP;
        }
        attr_set = true;
    }
    const int64_t nstat = B * n_heads * L;
    hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)cdiv(nstat, 256)), dim3(256), 0, st, a.o, a.dout, delta_ws, a.B,
                       a.L, n_heads, a.o_sb, a.o_sh, a.o_ss, a.do_sb, a.do_sh, a.do_ss);
    hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3(a.nqb * n_heads * a.B), dim3(256), lds_dq, st, a);
    const dim3 grid_kv((unsigned)(cdiv(L, 128) * n_kv_heads * B));
    // dK and dV in one launch (one wave per SIMD, ~380 VGPRs) measured 5 % faster than two launches at two waves per
    // SIMD that each recompute S (927 vs 971 us on the cfg-3 mask); VGPT_ATTN_BWD_SPLIT=1 selects the latter for A/B runs
    static int split = -1;
    if (split < 0) { const char* e = getenv("VGPT_ATTN_BWD_SPLIT"); split = e ? atoi(e) : 0; }
    if (!split) {
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<true, true>), grid_kv, dim3(256), lds_dkv, st, a);
    } else {
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<false, true>), grid_kv, dim3(256), lds_dkv, st, a);
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<true, false>), grid_kv, dim3(256), lds_dkv, st, a);
    }
    VGPT_CHECK_LAUNCH("vgpt_attn_blockmask_bwd");
    return VGPT_OK;
}
