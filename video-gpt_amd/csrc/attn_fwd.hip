// Block-masked flash attention forward for gfx950 (bf16 in/out, fp32 softmax and accumulation).
//
// Replaces module.local_attn = F.scaled_dot_product_attention with the additive block mask
// (LVM/transform/sdpa_transform.py:78-86,152; mask built at OmniGen/transformer.py:139-145 from
// the (B,L,L) bool masks of LVM/processor.py:575-731). The mask arrives bit-packed together with
// a per-tile summary (mask.hip), so arbitrary masks stay exact while all-masked tiles are skipped
// and all-visible tiles never read mask bits.
//
// Structure:
//   - block = 4 waves = 128 query rows of one (batch, head); each wave owns 32 query rows.
//   - K/V tiles of 64 keys are staged through registers into double-buffered LDS, the next
//     active tile's global loads are in flight while the current tile is computed.
//   - S^T = K Q^T with mfma_f32_32x32x16_bf16 (keys on rows): each lane then holds 32 scores of
//     ONE query row, so row max / row sum are 31 in-register ops + one cross-half shuffle, and
//     the exponentiated accumulator registers are, unchanged, the B operand of O^T = V^T P^T.
//   - V^T fragments come from ds_read_b64_tr_b16 (hardware transposed read) on a [key][d] image
//     whose row stride keeps the four 64-byte windows of a half-wave on disjoint banks.
//   - K rows are padded by 16 B so the ds_read_b128 16-lane groups are conflict-free.
#include "common.h"

namespace {

struct AttnArgs {
    const bf16* q;
    const bf16* k;
    const bf16* v;
    bf16* o;
    const uint32_t* bits;
    const uint8_t* summary;
    float* lse;  // optional (B, n_heads, L): base-2 log-sum-exp of the scaled scores, for the backward
    int B, L, n_heads, kv_group;  // kv_group = n_heads / n_kv_heads
    int W;                        // mask words per row
    int nqb, nkt;                 // 128-row q blocks, 64-key tiles
    int qb0;                      // first q block computed (rows before qb0*128 are keys only: cached prefix)
    int64_t q_sb, q_sh, q_ss, k_sb, k_sh, k_ss, v_sb, v_sh, v_ss, o_sb, o_sh, o_ss;
    float scale_log2e;
};

template <int D>
struct Cfg {
    // D == 96: K/V tiles are contiguous [key][192 B] images filled by LDS-DMA (global_load_lds); the K
    // image is XOR-swizzled (chunk ^= (key>>2)&3, applied on the DMA source address and on the read) so the
    // ds_read_b128 groups are conflict-free without padding.  Other head dims stage through registers
    // into a padded K image.
    static constexpr bool GLDS = (D == 96);
    static constexpr int KROW = GLDS ? D * 2 : D * 2 + 16;  // bytes
    // [key][d] image for transposed reads: (row stride in dwords) % 64 must be 16 or 48
    static constexpr int VROW_TR = (D == 96) ? 192 : (D == 128 ? 320 : 192);
    static constexpr int VROW_T = 64 * 2 + 8;  // [d][key] image of the slow variant
    static constexpr int KBYTES = 64 * KROW;
    static constexpr int CHUNKS = D / 8;               // 16-B chunks per key row
    static constexpr int LOADS = (64 * CHUNKS) / 256;  // chunks per thread per tile
    static_assert((64 * CHUNKS) % 256 == 0, "tile must divide over 256 threads");
};

template <int D, bool TR>
constexpr int vbytes() { return TR ? 64 * Cfg<D>::VROW_TR : D * Cfg<D>::VROW_T; }

template <int D, bool TR>
__global__ __launch_bounds__(256, (D <= 96 ? 2 : 1)) void attn_fwd_kernel(AttnArgs a) {
    using C = Cfg<D>;
    constexpr int KS = D / 16;  // k-steps of the QK^T product
    constexpr int DT = D / 32;  // 32-wide d tiles of the output
    constexpr int STAGE = C::KBYTES + vbytes<D, TR>();
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;

    // ---- work item: keep all q blocks of a (batch, head) on one XCD (shared K/V in its L2) ----
    const int nqa = a.nqb - a.qb0;  // q blocks actually computed
    const int total = nqa * a.n_heads * a.B;
    int wid = blockIdx.x;
    {
        const int xcd = wid & 7, qn = total >> 3, rn = total & 7;
        wid = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (wid >> 3);
    }
    const int qb = a.qb0 + wid % nqa;
    const int head = (wid / nqa) % a.n_heads;
    const int b = wid / (nqa * a.n_heads);
    const int kvh = head / a.kv_group;

    const uint8_t* sum_row = a.summary + ((int64_t)b * a.nqb + qb) * a.nkt;
    const bf16* kbase = a.k + (int64_t)b * a.k_sb + (int64_t)kvh * a.k_sh;
    const bf16* vbase = a.v + (int64_t)b * a.v_sb + (int64_t)kvh * a.v_sh;

    // ---- Q fragments (B operand of S^T = K Q^T): lane holds Q[q][16s + 8h .. +8) ----
    const int q_row = qb * 128 + wave * 32 + r;
    const int q_ld = min(q_row, a.L - 1);
    const bf16* qp = a.q + (int64_t)b * a.q_sb + (int64_t)head * a.q_sh + (int64_t)q_ld * a.q_ss;
    bf16x8 Qf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) Qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s + 8 * h);
    const uint32_t* bits_row = a.bits + ((int64_t)b * a.L + q_ld) * a.W;

    f32x16 O[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) O[dt][i] = 0.f;
    float m_i = -INFINITY, l_i = 0.f;

    // ---- staging helpers ----
    constexpr bool GLDS = TR && C::GLDS;
    constexpr int PIECES = (64 * D * 2) / 1024 / 4;  // 1-KiB LDS-DMA pieces per wave per operand
    // per-lane (key, source chunk) of each DMA piece this wave issues
    int g_key[GLDS ? PIECES : 1], g_kchunk[GLDS ? PIECES : 1], g_vchunk[GLDS ? PIECES : 1];
    if constexpr (GLDS) {
#pragma unroll
        for (int j = 0; j < PIECES; ++j) {
            const int unit = (wave * PIECES + j) * 64 + lane;  // 16-byte unit inside the tile image
            g_key[j] = unit / C::CHUNKS;
            g_vchunk[j] = unit % C::CHUNKS;
            g_kchunk[j] = g_vchunk[j] ^ ((g_key[j] >> 2) & 3);
        }
    }
    auto glds_tile = [&](int buf, int kt) {
        if constexpr (GLDS) {
            char* sk = smem + buf * STAGE;
#pragma unroll
            for (int j = 0; j < PIECES; ++j) {
                const int key = min(kt * 64 + g_key[j], a.L - 1);
                const int piece_off = (wave * PIECES + j) * 1024;
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(kbase + (int64_t)key * a.k_ss + g_kchunk[j] * 8),
                    (__attribute__((address_space(3))) void*)(sk + piece_off), 16, 0, 0);
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(vbase + (int64_t)key * a.v_ss + g_vchunk[j] * 8),
                    (__attribute__((address_space(3))) void*)(sk + C::KBYTES + piece_off), 16, 0, 0);
            }
        }
    };
    bf16x8 kreg[GLDS ? 1 : C::LOADS], vreg[GLDS ? 1 : C::LOADS];
    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < C::LOADS; ++i) {
            const int c = tid + 256 * i;
            const int key = min(kt * 64 + c / C::CHUNKS, a.L - 1);
            const int part = c % C::CHUNKS;
            kreg[i] = *reinterpret_cast<const bf16x8*>(kbase + (int64_t)key * a.k_ss + part * 8);
            vreg[i] = *reinterpret_cast<const bf16x8*>(vbase + (int64_t)key * a.v_ss + part * 8);
        }
    };
    auto lds_store = [&](int buf) {
        char* sk = smem + buf * STAGE;
        char* sv = sk + C::KBYTES;
#pragma unroll
        for (int i = 0; i < C::LOADS; ++i) {
            const int c = tid + 256 * i;
            const int key = c / C::CHUNKS, part = c % C::CHUNKS;
            *reinterpret_cast<bf16x8*>(sk + key * C::KROW + part * 16) = kreg[i];
            if (TR) {
                *reinterpret_cast<bf16x8*>(sv + key * C::VROW_TR + part * 16) = vreg[i];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    *reinterpret_cast<bf16*>(sv + (part * 8 + j) * C::VROW_T + key * 2) = vreg[i][j];
            }
        }
    };
    auto next_active = [&](int kt) {
        for (int k2 = kt + 1; k2 < a.nkt; ++k2)
            if (sum_row[k2]) return k2;
        return -1;
    };

    // mask words of a mixed tile are fetched one tile ahead, BEFORE that tile's LDS-DMA is issued, so the
    // ordinary loads never sit behind an in-flight DMA in the (in-order) vmcnt queue
    auto mask_words = [&](int t, uint32_t& w0, uint32_t& w1) {
        w0 = w1 = 0xffffffffu;
        if (((sum_row[t] >> (2 * wave)) & 3) == 2) {
            w0 = (2 * t < a.W) ? bits_row[2 * t] : 0u;
            w1 = (2 * t + 1 < a.W) ? bits_row[2 * t + 1] : 0u;
        }
    };
    int kt = next_active(-1);
    int buf = 0;
    uint32_t mw0 = 0xffffffffu, mw1 = 0xffffffffu;
    if (kt >= 0) {
        mask_words(kt, mw0, mw1);
        if constexpr (GLDS) glds_tile(0, kt); else gload(kt);
    }
    while (kt >= 0) {
        if constexpr (GLDS) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of tile kt has landed
        } else {
            lds_store(buf);
        }
        __syncthreads();
        const int nxt = next_active(kt);
        uint32_t nw0 = 0xffffffffu, nw1 = 0xffffffffu;
        if (nxt >= 0) {
            mask_words(nxt, nw0, nw1);
            if constexpr (GLDS) glds_tile(buf ^ 1, nxt); else gload(nxt);
        }
        const int code = (sum_row[kt] >> (2 * wave)) & 3;
        if (code) {
            const char* sk = smem + buf * STAGE;
            const char* sv = sk + C::KBYTES;
            // ---- S^T = K Q^T : two 32-key halves.  All K fragments are requested up front (one LDS round
            //      trip for the whole tile instead of one per MFMA), then the two MFMA chains run ----
            bf16x8 Kf[2][KS];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const char* krow = sk + (kb * 32 + r) * C::KROW;
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const int kc = GLDS ? ((2 * s + h) ^ ((r >> 2) & 3)) : (2 * s + h);
                    Kf[kb][s] = *reinterpret_cast<const bf16x8*>(krow + kc * 16);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            f32x16 S[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int i = 0; i < 16; ++i) S[kb][i] = 0.f;
#pragma unroll
                for (int s = 0; s < KS; ++s)
                    S[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Kf[kb][s], Qf[s], S[kb], 0, 0, 0);
            }
            // ---- V^T fragments do not depend on the softmax: request them now so their LDS latency hides
            //      under the softmax VALU work ----
            bf16x8 Vf[DT][4];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int key0 = (t >> 1) * 32 + (t & 1) * 16 + 4 * h;
                    if (TR) {
                        const int li = lane & 15;
                        const char* p0 = sv + (key0 + (li >> 2)) * C::VROW_TR +
                                         (dt * 32 + ((lane >> 4) & 1) * 16 + 4 * (li & 3)) * 2;
                        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                            (__attribute__((address_space(3))) bf16x4*)(p0));
                        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                            (__attribute__((address_space(3))) bf16x4*)(p0 + 8 * C::VROW_TR));
                        Vf[dt][t] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    } else {
                        const char* p0 = sv + (dt * 32 + r) * C::VROW_T + key0 * 2;
                        bf16x4 lo = *reinterpret_cast<const bf16x4*>(p0);
                        bf16x4 hi = *reinterpret_cast<const bf16x4*>(p0 + 16);
                        Vf[dt][t] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- mask (mixed tiles only), row max on raw scores; scale folded into the exp2 FMA ----
            if (code == 2) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    const uint32_t w = (kb ? mw1 : mw0) >> (4 * h);
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int bit = (i & 3) + 8 * (i >> 2);
                        S[kb][i] = ((w >> bit) & 1u) ? S[kb][i] : -INFINITY;
                    }
                }
            }
            float mx = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) mx = fmaxf(mx, S[kb][i]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * a.scale_log2e;  // scale > 0: max commutes with it
            const float m_new = fmaxf(m_i, mx);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m_i - m_use);
            float rs = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(S[kb][i], a.scale_log2e, -m_use));
                    S[kb][i] = p;
                    rs += p;
                }
            rs += __shfl_xor(rs, 32, 64);
            l_i = l_i * alpha + rs;
            // rescale the accumulator only when some row's running max moved (wave-uniform branch)
            if (__any(m_new != m_i)) {
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) O[dt][i] *= alpha;
            }
            m_i = m_new;
            // ---- P^T fragments: accumulator registers 8*half..8*half+7 of S[kb] ----
            bf16x8 Pf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < 8; ++j) Pf[t][j] = f2bf(S[t >> 1][8 * (t & 1) + j]);
            // ---- O^T += V^T P^T ----
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    O[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Vf[dt][t], Pf[t], O[dt], 0, 0, 0);
        }
        kt = nxt;
        buf ^= 1;
        mw0 = nw0;
        mw1 = nw1;
    }

    // ---- epilogue: lane holds O^T[d = 32dt + (i&3) + 8(i>>2) + 4h][q = r] ----
    if (a.lse && q_row < a.L && h == 0)
        a.lse[((int64_t)b * a.n_heads + head) * a.L + q_row] = l_i > 0.f ? m_i + __builtin_amdgcn_logf(l_i) : INFINITY;
    if (q_row < a.L) {
        const float inv = l_i > 0.f ? 1.0f / l_i : 0.f;
        bf16* op = a.o + (int64_t)b * a.o_sb + (int64_t)head * a.o_sh + (int64_t)q_row * a.o_ss;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                bf16x4 o;
#pragma unroll
                for (int t = 0; t < 4; ++t) o[t] = f2bf(O[dt][4 * g4 + t] * inv);
                *reinterpret_cast<bf16x4*>(op + dt * 32 + 8 * g4 + 4 * h) = o;
            }
    }
}

template <int D, bool TR>
int launch(const AttnArgs& a, hipStream_t s) {
    constexpr int lds = 2 * (Cfg<D>::KBYTES + vbytes<D, TR>());
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)attn_fwd_kernel<D, TR>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            vgpt_set_error("vgpt_attn_blockmask_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
            return VGPT_ERR_HIP;
        }
        attr_set = true;
    }
    const int total = (a.nqb - a.qb0) * a.n_heads * a.B;
    hipLaunchKernelGGL((attn_fwd_kernel<D, TR>), dim3(total), dim3(256), lds, s, a);
    VGPT_CHECK_LAUNCH("vgpt_attn_blockmask_fwd");
    return VGPT_OK;
}

}  // namespace

VGPT_EXPORT int vgpt_attn_supported(int head_dim) {
    return head_dim == 64 || head_dim == 96 || head_dim == 128;
}

static int attn_fwd_impl(const void* q, const void* k, const void* v, void* o, float* lse, int64_t q_start,
                         const uint32_t* bits, const uint8_t* summary, int64_t B,
                         int64_t L, int n_heads, int n_kv_heads, int head_dim,
                         int64_t q_sb, int64_t q_sh, int64_t q_ss, int64_t k_sb,
                         int64_t k_sh, int64_t k_ss, int64_t v_sb, int64_t v_sh,
                         int64_t v_ss, int64_t o_sb, int64_t o_sh, int64_t o_ss,
                         float scale, int variant, void* stream) {
    VGPT_REQUIRE(q && k && v && o && bits && summary, VGPT_ERR_INVALID,
                 "vgpt_attn_blockmask_fwd: null pointer");
    VGPT_REQUIRE(B >= 0 && L >= 0 && n_heads > 0 && n_kv_heads > 0, VGPT_ERR_INVALID,
                 "vgpt_attn_blockmask_fwd: bad shape");
    VGPT_REQUIRE(n_heads % n_kv_heads == 0, VGPT_ERR_INVALID,
                 "vgpt_attn_blockmask_fwd: n_kv_heads must divide n_heads");
    VGPT_REQUIRE(vgpt_attn_supported(head_dim), VGPT_ERR_UNSUPPORTED,
                 "vgpt_attn_blockmask_fwd: head_dim=%d unsupported (64, 96, 128)", head_dim);
    VGPT_REQUIRE(scale > 0.f, VGPT_ERR_INVALID, "vgpt_attn_blockmask_fwd: scale must be positive");
    VGPT_REQUIRE(variant == 0 || variant == 1, VGPT_ERR_INVALID,
                 "vgpt_attn_blockmask_fwd: unknown variant %d", variant);
    const int64_t strides[] = {q_sb, q_sh, q_ss, k_sb, k_sh, k_ss, v_sb, v_sh, v_ss};
    for (int64_t st : strides)
        VGPT_REQUIRE(st % 8 == 0, VGPT_ERR_UNSUPPORTED,
                     "vgpt_attn_blockmask_fwd: q/k/v strides must be multiples of 8 elements");
    VGPT_REQUIRE(o_sb % 4 == 0 && o_sh % 4 == 0 && o_ss % 4 == 0, VGPT_ERR_UNSUPPORTED,
                 "vgpt_attn_blockmask_fwd: o strides must be multiples of 4 elements");
    VGPT_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) == 0 && ((uintptr_t)o & 7) == 0,
                 VGPT_ERR_UNSUPPORTED, "vgpt_attn_blockmask_fwd: q/k/v must be 16-byte aligned");
    VGPT_REQUIRE(L < (1 << 24) && B * n_heads * cdiv(L, 128) < (1ll << 31), VGPT_ERR_UNSUPPORTED,
                 "vgpt_attn_blockmask_fwd: problem too large");
    VGPT_REQUIRE(q_start >= 0 && q_start % 128 == 0 && q_start <= L, VGPT_ERR_INVALID,
                 "vgpt_attn_blockmask_fwd: q_start must be a multiple of 128 in [0, L]");
    if (B == 0 || L == 0 || q_start >= L) return VGPT_OK;
    AttnArgs a;
    a.q = (const bf16*)q; a.k = (const bf16*)k; a.v = (const bf16*)v; a.o = (bf16*)o;
    a.bits = bits; a.summary = summary; a.lse = lse;
    a.B = (int)B; a.L = (int)L; a.n_heads = n_heads; a.kv_group = n_heads / n_kv_heads;
    a.W = (int)cdiv(L, 32); a.nqb = (int)cdiv(L, 128); a.nkt = (int)cdiv(L, 64); a.qb0 = (int)(q_start / 128);
    a.q_sb = q_sb; a.q_sh = q_sh; a.q_ss = q_ss; a.k_sb = k_sb; a.k_sh = k_sh; a.k_ss = k_ss;
    a.v_sb = v_sb; a.v_sh = v_sh; a.v_ss = v_ss; a.o_sb = o_sb; a.o_sh = o_sh; a.o_ss = o_ss;
    a.scale_log2e = scale * 1.4426950408889634f;
    hipStream_t s = (hipStream_t)stream;
#define ATTN_CASE(DD)                                   \
    case DD:                                            \
        return variant == 0 ? launch<DD, true>(a, s) : launch<DD, false>(a, s);
    switch (head_dim) {
        ATTN_CASE(64) ATTN_CASE(96) ATTN_CASE(128)
    }
#undef ATTN_CASE
    return VGPT_ERR_UNSUPPORTED;
}

VGPT_EXPORT int vgpt_attn_blockmask_fwd(const void* q, const void* k, const void* v, void* o,
                                        const uint32_t* bits, const uint8_t* summary, int64_t B,
                                        int64_t L, int n_heads, int n_kv_heads, int head_dim,
                                        int64_t q_sb, int64_t q_sh, int64_t q_ss, int64_t k_sb,
                                        int64_t k_sh, int64_t k_ss, int64_t v_sb, int64_t v_sh,
                                        int64_t v_ss, int64_t o_sb, int64_t o_sh, int64_t o_ss,
                                        float scale, int variant, void* stream) {
    return attn_fwd_impl(q, k, v, o, nullptr, 0, bits, summary, B, L, n_heads, n_kv_heads, head_dim, q_sb, q_sh, q_ss,
                         k_sb, k_sh, k_ss, v_sb, v_sh, v_ss, o_sb, o_sh, o_ss, scale, variant, stream);
}

VGPT_EXPORT int vgpt_attn_blockmask_fwd_lse(const void* q, const void* k, const void* v, void* o, float* lse,
                                            const uint32_t* bits, const uint8_t* summary, int64_t B,
                                            int64_t L, int n_heads, int n_kv_heads, int head_dim,
                                            int64_t q_sb, int64_t q_sh, int64_t q_ss, int64_t k_sb,
                                            int64_t k_sh, int64_t k_ss, int64_t v_sb, int64_t v_sh,
                                            int64_t v_ss, int64_t o_sb, int64_t o_sh, int64_t o_ss,
                                            float scale, void* stream) {
    if (!lse) {
        vgpt_set_error("vgpt_attn_blockmask_fwd_lse: null lse");
        return VGPT_ERR_INVALID;
    }
    return attn_fwd_impl(q, k, v, o, lse, 0, bits, summary, B, L, n_heads, n_kv_heads, head_dim, q_sb, q_sh, q_ss,
                         k_sb, k_sh, k_ss, v_sb, v_sh, v_ss, o_sb, o_sh, o_ss, scale, 0, stream);
}

VGPT_EXPORT int vgpt_attn_blockmask_fwd_qrange(const void* q, const void* k, const void* v, void* o, int64_t q_start,
                                               const uint32_t* bits, const uint8_t* summary, int64_t B,
                                               int64_t L, int n_heads, int n_kv_heads, int head_dim,
                                               int64_t q_sb, int64_t q_sh, int64_t q_ss, int64_t k_sb,
                                               int64_t k_sh, int64_t k_ss, int64_t v_sb, int64_t v_sh,
                                               int64_t v_ss, int64_t o_sb, int64_t o_sh, int64_t o_ss,
                                               float scale, void* stream) {
    return attn_fwd_impl(q, k, v, o, nullptr, q_start, bits, summary, B, L, n_heads, n_kv_heads, head_dim, q_sb, q_sh,
                         q_ss, k_sb, k_sh, k_ss, v_sb, v_sh, v_ss, o_sb, o_sh, o_ss, scale, 0, stream);
}
