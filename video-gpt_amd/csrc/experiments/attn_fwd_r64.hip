// Block-masked flash attention forward, 64 query rows per wave (gfx950, head_dim 96, bf16 in/out, fp32 softmax).
//
// Same operator as attn_fwd.hip (module.local_attn = SDPA with the additive block mask,
// LVM/transform/sdpa_transform.py:78-86,152; mask of LVM/processor.py:575-731 bit-packed by mask.hip) for the planned
// launches of the sampler and the trainer (vgpt_attn_fwd_plan, 256-row work items).
//
// Why another structure: with 32 query rows per wave every 32x32x16 MFMA needs a fresh 1-KiB K or V fragment from LDS,
// i.e. 4 SIMDs x 1 KiB per 32 cycles = the LDS's whole 128 B/clk -- the 4-wave / 32-row kernel is LDS-bound at
// 0.37 of the matrix peak (profiles/r01_pmc_mfma.json).  Here a wave owns TWO 32-row query blocks (A, B) and every
// fragment, read just in time, feeds two consecutive MFMAs; the workgroup (4 waves, one per SIMD, the whole
// 512-register file) covers 256 rows, so K/V tiles are fetched from L2 half as often as well.
//
// One wave per SIMD has no partner wave to fill the matrix pipe while it does softmax arithmetic, so the loop is software
// pipelined by hand.  Iteration t runs 48 MFMA slots and the vector ALU runs ONE continuous softmax stream beside them,
// a slice (19 per job: 2 row-maximum, 1 decision, 16 x two probabilities) per slot:
//     slots  0..11   Q.K^T(A, t+1)   | softmax(B, t) slices 3..18 in slots 0..15
//     slots 12..23   Q.K^T(B, t+1)   |   (B's scores are rewritten only from slot 12 on, half by half, behind their last reader)
//     slots 24..47   P.V(A, t), P.V(B, t) on each of the 12 V fragments
//                                    | softmax(A, t+1) slices 0..18 in slots 24..42, softmax(B, t+1) slices 0..2 in 43..45
// K fragments are read just in time once per query block (LDS has the bandwidth: the reads that hurt were the per-MFMA
// ones), V fragments once for both.  The probabilities of A are double-buffered (its softmax writes tile t+1's while P.V
// still reads tile t's); the loop is unrolled by two for that parity.  sched_barrier after every slot keeps the interleave.
// K/V tiles of 64 keys arrive by LDS-DMA into a 4-slot ring three tiles ahead; one barrier per tile.
//
// Softmax in the log2 domain: p = exp2(s * scale*log2(e) - m), one FMA + one v_exp_f32 per score, with a per-row
// reference m that is NOT the running maximum: it is the maximum of the row's first visible tile and then stays put, so
// nothing is ever rescaled in the loop; p may exceed 1 (by up to 2^LAZY_THR = 2^60: sums stay far inside fp32, and the
// relative precision of bf16 / fp32 does not depend on the magnitude).  The row sums come from the matrix pipe too
// (one extra MFMA per 16 keys against a fragment of ones: the vector ALU is the unit without slack here).  A row whose
// scores climb more than 2^60 above its first tile's maximum cannot be handled this way: the wave notes it, finishes the
// item regardless and leaves a flag; attn_fwd_r64_slow_kernel, launched behind this kernel, recomputes flagged items
// with the textbook online softmax (running maximum, rescale every tile; plain, unpipelined code).
//
// How the interleave is held: hipcc moves plain arithmetic freely (sched_barrier only binds its machine scheduler, and
// at one wave per SIMD it puts every MFMA result in accumulation registers, which the softmax could only read through
// one v_accvgpr_read per score).  So every MFMA is a one-instruction `asm volatile` -- Q.K^T with its accumulator in
// ARCHITECTURAL registers, P.V in accumulation registers -- and every slot ends in an empty `asm volatile` that takes
// the slice's live values as in/out operands: volatile statements keep their order, so each slice sits between its
// MFMA and the next.  LDS fragment reads are ordinary loads whose address passes through such a pin (the compiler still
// counts lgkmcnt for them).  The compiler knows nothing about the latency of an MFMA issued this way: the slot order
// keeps every reader of a Q.K^T result at least 12 MFMAs behind it, and the readers of O (the rare rescale, the
// epilogue) wait on explicit s_nop.
#include <type_traits>

#include "../common.h"

namespace {

constexpr int D = 96;
constexpr int KS = D / 16;            // k-steps of Q.K^T
constexpr int DT = D / 32;            // 32-wide d tiles of the output
constexpr int KROW = D * 2;           // bytes per key row of the K image (XOR-swizzled 16-byte chunks)
constexpr int VROW = D * 2;           // bytes per key row of the V image (transposed reads)
constexpr int TILE_BYTES = 2 * 64 * D * 2;   // K + V images of one 64-key tile
constexpr int NSTAGE = 4;
constexpr int ACT_MAX = 1023;         // key tiles per active-list chunk
constexpr int LIST_OFF = NSTAGE * TILE_BYTES;
constexpr int MASK_OFF = LIST_OFF + 4096;
constexpr int MASK_STAGE = 4 * 2 * 256;          // 4 waves x 2 query blocks x (32 rows x 2 words)
constexpr int LDS_TOTAL = MASK_OFF + NSTAGE * MASK_STAGE;
constexpr float LAZY_THR = 60.0f;   // log2 units a row's scores may exceed its reference before the wave flags the item

struct R64Args {
    const bf16* q;
    const bf16* k;
    const bf16* v;
    bf16* o;
    float* lse;                // optional (B, n_heads, L): base-2 log-sum-exp of the scaled scores
    const uint32_t* bits;
    const int32_t* items;      // n_items x 4: batch, row0, nrows (<= 256), 0
    const uint16_t* isum;      // n_items x nkt: 2 bits per 32-row slab (0 none visible, 1 all, 2 mixed)
    const int32_t* order;      // items longest first
    int n_items, L, n_heads, kv_group, W, nkt;
    int64_t q_sb, q_sh, q_ss, k_sb, k_sh, k_ss, v_sb, v_sh, v_ss, o_sb, o_sh, o_ss;
    float scale_log2e;
    int32_t* fallback;         // n_items * n_heads * 4 (one per wave): 1 where the pipelined kernel's result is void
};

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// LDS-DMA from inline asm (the compiler must not see LDS being written: it would drain vmcnt(0) before later LDS reads)
__device__ __forceinline__ void dma16(const char* base, uint32_t off, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(off), "s"(base), "s"(lds_dst)
                 : "memory");
}
__device__ __forceinline__ void dma4(const uint32_t* src, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(lds_dst)
                 : "memory");
}


__device__ __forceinline__ void qk_first(f32x16& d, const bf16x8& kf, const bf16x8& qf) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(kf), "a"(qf));
}
__device__ __forceinline__ void qk_acc(f32x16& d, const bf16x8& kf, const bf16x8& qf) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(kf), "a"(qf));
}
// `tie`: a value the slice that follows reads -- listed as an in/out operand so that the slice cannot be moved in front
// of this MFMA (the instruction does not touch it)
__device__ __forceinline__ void pv_acc(f32x16& o, const bf16x8& vf, const bf16x8& pf, f32x16& tie) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0" : "+a"(o), "+v"(tie) : "v"(vf), "v"(pf));
}
// v_max3_f32 / v_max_f32 as single instructions: on values that come out of inline asm hipcc wraps every fmaxf operand in
// a canonicalising v_max_f32 x, x (three instructions per maximum)
__device__ __forceinline__ float max3(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ float max2(float a, float b) {
    float d;
    asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// empty volatile statements that tie values to a point of the instruction stream
#define PIN1(a) asm volatile("" : "+v"(a))
#define PIN_S(S, x, y, p) asm volatile("" : "+v"((S)[0]), "+v"((S)[1]), "+v"(x), "+v"(y), "+v"(p))
#define MFMA_DRAIN() asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory")

// wave-uniform copy of a pointer in scalar registers (a value read through a by-reference argument arrives in VGPRs)
__device__ __forceinline__ const char* uniform_ptr(const void* p) {
    const uint64_t v = (uint64_t)(uintptr_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<const char*>((uintptr_t)(((uint64_t)hi << 32) | lo));
}

__global__ __launch_bounds__(256, 1) void attn_fwd_r64_kernel(R64Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;

    // ---- work item: XCD x owns n_heads/8 consecutive heads (their K/V stay in its L2), items longest first ----
    int rank, head;
    if ((a.n_heads & 7) == 0) {
        const int per = a.n_heads >> 3, j = blockIdx.x >> 3;
        rank = j / per;
        head = (blockIdx.x & 7) * per + j % per;
    } else {
        rank = blockIdx.x / a.n_heads;
        head = blockIdx.x % a.n_heads;
    }
    const int item = a.order[rank];
    const int b = a.items[4 * item], row0 = a.items[4 * item + 1];
    const int row_last = row0 + a.items[4 * item + 2] - 1;
    const uint16_t* sum16 = a.isum + (int64_t)item * a.nkt;
    const int kvh = head / a.kv_group;
    const bf16* kbase = a.k + (int64_t)b * a.k_sb + (int64_t)kvh * a.k_sh;
    const bf16* vbase = a.v + (int64_t)b * a.v_sb + (int64_t)kvh * a.v_sh;
    const bool wave_live = row0 + wave * 64 <= row_last;   // a wave wholly behind the item's rows only moves tiles

    // ---- Q fragments of both query blocks (the scale goes into the exponent's FMA: scaling bf16 Q would round it) ----
    bf16x8 Q[2][KS];
    int q_row[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        q_row[qb] = row0 + wave * 64 + qb * 32 + r;
        const bf16* qp = a.q + (int64_t)b * a.q_sb + (int64_t)head * a.q_sh + (int64_t)min(q_row[qb], row_last) * a.q_ss;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            Q[qb][s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s + 8 * h);
        }
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int s = 0; s < KS; ++s) asm volatile("" : "+a"(Q[qb][s]));   // retire the global loads; park Q in accumulation registers

    f32x16 O[2][DT], LS[2];              // output accumulators; LS: every register = the row's sum of probabilities
    f32x16 SA[2], SB[2];                 // scores [kb]
    bf16x8 PA[2][4], PB[4];              // probabilities: A double-buffered (by tile parity), B single      [t]
    bf16x8 Kr[5], Vr[3];                 // just-in-time fragment rings: K read 4 slots ahead, V 2 fragments (4 slots) ahead
    float m_ref[2] = {-INFINITY, -INFINITY};
    bf16x8 ones;                         // A operand of the row-sum MFMA: O_l^T[d][q] = sum_k 1 * P^T[k][q]
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (bf16)1.0f;
    asm volatile("" : "+v"(ones));
    bool bad = false;                    // some row of this wave outgrew its reference (wave-uniform)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) O[qb][dt][i] = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) LS[qb][i] = 0.f;
    }

    // ---- tile staging: 12 + 12 one-KiB pieces per tile, 3 + 3 per wave ----
    const uint32_t lds_base =
        __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
    uint32_t g_koff[3], g_voff[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int unit = (wave * 3 + j) * 64 + lane;   // 16-byte unit of the tile image
        const int key = unit / 12;
        g_koff[j] = (uint32_t)key * (uint32_t)a.k_ss * 2u + (((unit % 12) ^ ((key >> 2) & 3)) * 16);
        g_voff[j] = (uint32_t)key * (uint32_t)a.v_ss * 2u + (unit % 12) * 16;
    }
    uint32_t* alist = reinterpret_cast<uint32_t*>(smem + LIST_OFF);
    const uint32_t* mrow_src[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
        mrow_src[qb] = a.bits + ((int64_t)b * a.L + min(row0 + wave * 64 + qb * 32 + (lane >> 1), row_last)) * a.W;

    // entry = tile << 16 | 16 summary bits; this wave's two slabs are 2*wave and 2*wave+1
    auto code_of = [&](uint32_t e, int qb) __attribute__((always_inline)) {
        const int c = (e >> (2 * (2 * wave + qb))) & 3;
        return c == 1 ? 1 : 2;    // a slab with no visible key inside an active tile: masked like a mixed one (its words are 0)
    };
    // one piece pair (K and V, 1 KiB each) of the 3 a wave moves per tile; `j` = 0..2, mask words ride with j = 2
    // hk = 0..5: piece j = hk / 2 of the K (even hk) or V (odd hk) image
    auto stage_piece = [&](int slot, uint32_t e, int hk) __attribute__((always_inline)) {
        const int j = hk >> 1;
        const int kt = (int)(e >> 16);
        const char* kt_base = reinterpret_cast<const char*>(kbase + (int64_t)kt * 64 * a.k_ss);
        const char* vt_base = reinterpret_cast<const char*>(vbase + (int64_t)kt * 64 * a.v_ss);
        uint32_t ko = g_koff[j], vo = g_voff[j];
        if (kt * 64 + 64 > a.L) {   // rare: recomputed from the lane id instead of held in registers
            const int unit = (wave * 3 + j) * 64 + lane, key0 = unit / 12;
            const uint32_t key = (uint32_t)(min(kt * 64 + key0, a.L - 1) - kt * 64);
            ko = key * (uint32_t)a.k_ss * 2u + (((unit % 12) ^ ((key0 >> 2) & 3)) * 16);
            vo = key * (uint32_t)a.v_ss * 2u + (unit % 12) * 16;
        }
        const uint32_t dst = lds_base + (uint32_t)(slot * TILE_BYTES + (wave * 3 + j) * 1024);
        if ((hk & 1) == 0) dma16(kt_base, ko, dst);
        else dma16(vt_base, vo, dst + 64 * KROW);
        if (hk == 5 && wave_live) {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
                if (code_of(e, qb) == 2)
                    dma4(mrow_src[qb] + min(2 * kt + (lane & 1), a.W - 1),
                         lds_base + (uint32_t)(MASK_OFF + slot * MASK_STAGE + (wave * 2 + qb) * 256));
        }
    };
    auto stage_tile = [&](int slot, uint32_t e) __attribute__((always_inline)) {
        const int kt = (int)(e >> 16);
        const char* kt_base = reinterpret_cast<const char*>(kbase + (int64_t)kt * 64 * a.k_ss);
        const char* vt_base = reinterpret_cast<const char*>(vbase + (int64_t)kt * 64 * a.v_ss);
        const bool tail = kt * 64 + 64 > a.L;   // clamp keys past the end onto the last row (their mask bits are 0)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            uint32_t ko = g_koff[j], vo = g_voff[j];
            if (tail) {   // rare: recomputed from the lane id instead of held in registers
                const int unit = (wave * 3 + j) * 64 + lane, key0 = unit / 12;
                const uint32_t key = (uint32_t)(min(kt * 64 + key0, a.L - 1) - kt * 64);
                ko = key * (uint32_t)a.k_ss * 2u + (((unit % 12) ^ ((key0 >> 2) & 3)) * 16);
                vo = key * (uint32_t)a.v_ss * 2u + (unit % 12) * 16;
            }
            const uint32_t dst = lds_base + (uint32_t)(slot * TILE_BYTES + (wave * 3 + j) * 1024);
            dma16(kt_base, ko, dst);
            dma16(vt_base, vo, dst + 64 * KROW);
        }
        if (wave_live) {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
                if (code_of(e, qb) == 2)
                    dma4(mrow_src[qb] + min(2 * kt + (lane & 1), a.W - 1),
                         lds_base + (uint32_t)(MASK_OFF + slot * MASK_STAGE + (wave * 2 + qb) * 256));
        }
    };

    // ---- just-in-time fragment reads.  The address register goes through a pin (an empty volatile asm with the
    //      register as in/out operand): the load cannot be issued earlier than that point of the stream. ----
    const int kswz = (r >> 2) & 3;
    int kaddr[KS];   // per k-step: LDS byte address of this lane's 16-byte chunk (XOR swizzle) in the K image to read NEXT
#pragma unroll
    for (int s = 0; s < KS; ++s) kaddr[s] = (int)lds_base + r * KROW + (((2 * s + h) ^ kswz) * 16);
    int kslot = 0, vslot = 0;   // ring slots kaddr / vaddr point at (wave-uniform)
    auto k_goto = [&](int slot) __attribute__((always_inline)) {   // point kaddr at the K image of ring slot `slot`
        const int delta = (slot - kslot) * TILE_BYTES;
#pragma unroll
        for (int s = 0; s < KS; ++s) kaddr[s] += delta;
        kslot = slot;
    };
    typedef const volatile __attribute__((address_space(3))) bf16x8* lds_frag_ptr;
    auto read_k = [&](int f) __attribute__((always_inline)) -> bf16x8 {   // fragment f = kb * KS + s of the image kaddr points at
        const int kb = f / KS, s = f % KS;
        // volatile: keeps its place among the volatile asm statements (and the compiler still counts lgkmcnt for it)
        return *(lds_frag_ptr)(uintptr_t)(uint32_t)(kaddr[s] + kb * 32 * KROW);
    };
    int vaddr = (int)lds_base + 64 * KROW + ((lane & 15) >> 2) * VROW + (((lane >> 4) & 1) * 16 + 4 * (lane & 3)) * 2 + 4 * h * VROW;
    auto read_v = [&](int f) __attribute__((always_inline)) -> bf16x8 {   // fragment f = dt * 4 + t of the V image vaddr points into
        const int dt = f / 4, t = f % 4;
        PIN1(vaddr);
        const uint32_t p0 = (uint32_t)(vaddr + ((t >> 1) * 32 + (t & 1) * 16) * VROW + dt * 64);
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(uintptr_t)p0);
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(uintptr_t)(p0 + 8 * VROW));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };

    // ---- softmax of one job in 19 slices (S: that job's two score tiles, Pd: where its probabilities go); the row sum
    //      of the probabilities is taken by the matrix pipe (LS), which has the slack the vector ALU lacks ----
    float mxa = -INFINITY, mxb = -INFINITY, m_use = 0.f;
    auto sm_slice = [&](auto qbc, auto kc, f32x16 (&S)[2], bf16x8 (&Pd)[4], int code, uint32_t e, int slot) __attribute__((always_inline)) {
        constexpr int QB = decltype(qbc)::value;
        constexpr int K = decltype(kc)::value;
        if constexpr (K == 0) {
            if (code == 2) {   // mixed tile: keys the row does not see leave the softmax (wave-uniform branch)
                const uint2 mw = *reinterpret_cast<const uint2*>(smem + MASK_OFF + slot * MASK_STAGE + (wave * 2 + QB) * 256 + r * 8);
                const uint32_t mw0 = mw.x, mw1 = (2 * (int)(e >> 16) + 1 < a.W) ? mw.y : 0u;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    const uint32_t w = (kb ? mw1 : mw0) >> (4 * h);
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        S[kb][i] = ((w >> ((i & 3) + 8 * (i >> 2))) & 1u) ? S[kb][i] : -INFINITY;
                }
            }
            mxa = max3(S[0][0], S[0][1], S[0][2]);
            mxb = max3(S[1][0], S[1][1], S[1][2]);
#pragma unroll
            for (int i = 3; i < 9; i += 2) {
                mxa = max3(mxa, S[0][i], S[0][i + 1]);
                mxb = max3(mxb, S[1][i], S[1][i + 1]);
            }
            PIN_S(S, mxa, mxb, m_use);
        } else if constexpr (K == 1) {
#pragma unroll
            for (int i = 9; i < 15; i += 2) {
                mxa = max3(mxa, S[0][i], S[0][i + 1]);
                mxb = max3(mxb, S[1][i], S[1][i + 1]);
            }
            mxa = max3(mxa, S[0][15], mxb);
            mxb = max2(mxb, S[1][15]);
            PIN_S(S, mxa, mxb, m_use);
        } else if constexpr (K == 2) {
            const float mx = max2(mxa, mxb);
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
            const float tmax = max2(__uint_as_float(sw[0]), __uint_as_float(sw[1])) * a.scale_log2e;   // row maximum, log2 domain
            if (m_ref[QB] == -INFINITY) m_ref[QB] = tmax;                    // the row's first visible tile sets its reference
            bad = bad | (__any(tmax > m_ref[QB] + LAZY_THR) != 0);
            m_use = m_ref[QB] == -INFINITY ? 0.f : -m_ref[QB];  // (negated) rows that have seen no key yet: exp2(-inf + 0) = 0
            PIN_S(S, mxa, mxb, m_use);
        } else {   // K = 3..18: two scores -> probabilities, row sum, bf16 pair of the P fragment
            constexpr int pidx = K - 3, kb = pidx >> 3, i0 = 2 * (pidx & 7);
            const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(S[kb][i0], a.scale_log2e, m_use));
            const float e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(S[kb][i0 + 1], a.scale_log2e, m_use));
            Pd[kb * 2 + (i0 >> 3)][i0 & 7] = f2bf(e0);
            Pd[kb * 2 + (i0 >> 3)][(i0 & 7) + 1] = f2bf(e1);
            PIN_S(S, mxa, m_use, Pd[kb * 2 + (i0 >> 3)]);
        }
    };
    using QA = std::integral_constant<int, 0>;
    using QB_ = std::integral_constant<int, 1>;

    // Iteration over tile t (parity PAR = t & 1), see the header.  QK: tile t+1 exists; e_cur / e_nxt: list entries of
    // tiles t and t+1; slot_t / slot_n: their ring slots.  On entry kaddr points at K(t+1) whose fragments 0..3 are
    // already in flight (requested by the previous iteration or the prologue), vaddr at V(t).
#ifdef VGPT_R64_STAMPS
    unsigned long long* stamps = reinterpret_cast<unsigned long long*>(a.lse);
    int stamp_i = 0;
#define STAMP()                                                                                     \
    do {                                                                                            \
        if (blockIdx.x == 0 && wave == 0 && stamp_i < 1000) {                                       \
            const unsigned long long ts_ = __builtin_amdgcn_s_memtime();                            \
            if (lane == 0) stamps[stamp_i] = ts_;                                                   \
            ++stamp_i;                                                                              \
        }                                                                                           \
    } while (0)
#else
#define STAMP() do { } while (0)
#endif
    auto iteration = [&](auto parc, auto qkc, uint32_t e_cur, uint32_t e_nxt, int slot_t, int slot_n, bool more_k,
                         uint32_t e_dma) __attribute__((always_inline)) {   // e_dma: list entry of tile t+3 to fetch (0: none), into the slot tile t-1 left
        constexpr int PAR = decltype(parc)::value;
        constexpr bool QK = decltype(qkc)::value;
        STAMP();
        const int codeB = code_of(e_cur, 1), codeA1 = code_of(e_nxt, 0), codeB1 = code_of(e_nxt, 1);
        static_for<0, 24>([&](auto kc) __attribute__((always_inline)) {
            constexpr int K = decltype(kc)::value, f = K % 12;
            if constexpr (QK) {
                if constexpr (K + 4 < 24) Kr[(K + 4) % 5] = read_k((K + 4) % 12);
                if constexpr (K < 12) {
                    if constexpr (f % KS == 0) qk_first(SA[f / KS], Kr[K % 5], Q[0][f % KS]);
                    else qk_acc(SA[f / KS], Kr[K % 5], Q[0][f % KS]);
                } else {
                    if constexpr (f % KS == 0) qk_first(SB[f / KS], Kr[K % 5], Q[1][f % KS]);
                    else qk_acc(SB[f / KS], Kr[K % 5], Q[1][f % KS]);
                }
            }
            if constexpr (K == 20 || K == 22) Vr[(K - 20) >> 1] = read_v((K - 20) >> 1);   // V fragments 0, 1 of tile t
            if constexpr (K < 16) sm_slice(QB_{}, std::integral_constant<int, K + 3>{}, SB, PB, codeB, e_cur, slot_t);
            if constexpr (QK && (K == 16 || K == 19 || K == 22)) {   // the softmax stream pauses here: a third of the tile fetch,
                if (e_dma) stage_piece((slot_t + 3) & 3, e_dma, (K - 16) / 3);   // one 1-KiB piece at a time (the rest beside P.V)
            }
        });
        STAMP();
        if (more_k) k_goto((slot_n + 1) & 3);   // kaddr -> K(t+2)
        static_for<0, 32>([&](auto kc) __attribute__((always_inline)) {   // 24 P.V slots, then 8 row-sum slots (4 k-steps x 2 query blocks)
            constexpr int K = decltype(kc)::value, f = K >> 1, dt = f / 4, t = f % 4;
            f32x16& tie = K < 19 ? SA[K < 11 ? 0 : 1] : SB[0];
            if constexpr (K < 24) {
                if constexpr ((K & 1) == 0) {
                    if constexpr (f + 2 < 12) Vr[(f + 2) % 3] = read_v(f + 2);
                    pv_acc(O[0][dt], Vr[f % 3], PA[PAR][t], tie);
                } else {
                    pv_acc(O[1][dt], Vr[f % 3], PB[t], tie);
                }
            } else {
                constexpr int ks = (K - 24) >> 1;
                if constexpr ((K & 1) == 0) pv_acc(LS[0], ones, PA[PAR][ks], tie);
                else pv_acc(LS[1], ones, PB[ks], tie);
            }
            if constexpr (QK && (K == 22 || K == 25 || K == 28)) {
                if (e_dma) stage_piece((slot_t + 3) & 3, e_dma, 3 + (K - 22) / 3);
            }
            if constexpr (K >= 28) {   // K fragments 0..3 of tile t+2 for the next iteration
                if (more_k) Kr[K - 28] = read_k(K - 28);
            }
            if constexpr (QK) {
                if constexpr (K < 19) sm_slice(QA{}, kc, SA, PA[PAR ^ 1], codeA1, e_nxt, slot_n);
                else if constexpr (K < 22) sm_slice(QB_{}, std::integral_constant<int, K - 19>{}, SB, PB, codeB1, e_nxt, slot_n);
            }
        });
        STAMP();
        vaddr += (slot_n - vslot) * TILE_BYTES;   // -> V(t+1)
        vslot = slot_n;
        STAMP();
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;

    // the key tiles some row of the item sees, compacted into LDS (entry = tile << 16 | summary bits); returns their count
    auto build_list = [&](int chunk0) __attribute__((always_inline)) {
        __syncthreads();   // every wave is done with the previous list and the ring
        if (wave == 0) {
            const int lim = min(chunk0 + ACT_MAX, a.nkt);
            int n = 0;
            for (int base = chunk0; base < lim; base += 64) {
                const int t = base + lane;
                const uint32_t c = t < lim ? (uint32_t)sum16[t] : 0u;
                const uint64_t bal = __ballot(c != 0);
                if (c) alist[1 + n + __popcll(bal & ((1ull << lane) - 1))] = ((uint32_t)t << 16) | c;
                n += __popcll(bal);
            }
            if (lane == 0) alist[0] = (uint32_t)n;
        }
        __syncthreads();
        return __builtin_amdgcn_readfirstlane((int)alist[0]);
    };
    for (int chunk0 = 0; chunk0 < a.nkt; chunk0 += ACT_MAX) {
        const int n_act = build_list(chunk0);
        if (n_act == 0) continue;
        auto entry = [&](int i) __attribute__((always_inline)) { return __builtin_amdgcn_readfirstlane(i < n_act ? alist[1 + i] : 0u); };

        // ---- prologue: tiles 0..2 in flight; Q.K^T(0) for A and B; softmax(A,0) and the first three slices of softmax(B,0) ----
        for (int i = 0; i < min(n_act, 3); ++i) stage_tile(i, entry(i));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        uint32_t e_cur = entry(0), e_nxt = entry(1);
        if (wave_live) {
            k_goto(0);
            vaddr -= vslot * TILE_BYTES;
            vslot = 0;
            static_for<0, 12>([&](auto fc) __attribute__((always_inline)) {
                constexpr int f = decltype(fc)::value;
                const bf16x8 kf = read_k(f);
                if constexpr (f % KS == 0) {
                    qk_first(SA[f / KS], kf, Q[0][f % KS]);
                    qk_first(SB[f / KS], kf, Q[1][f % KS]);
                } else {
                    qk_acc(SA[f / KS], kf, Q[0][f % KS]);
                    qk_acc(SB[f / KS], kf, Q[1][f % KS]);
                }
            });
            MFMA_DRAIN();
            const int cA = code_of(e_cur, 0), cB = code_of(e_cur, 1);
            static_for<0, 19>([&](auto kc) __attribute__((always_inline)) { sm_slice(QA{}, kc, SA, PA[0], cA, e_cur, 0); });
            static_for<0, 3>([&](auto kc) __attribute__((always_inline)) { sm_slice(QB_{}, kc, SB, PB, cB, e_cur, 0); });
            if (n_act > 1) {   // kaddr -> K(1); its first four fragments
                k_goto(1);
                static_for<0, 4>([&](auto fc) __attribute__((always_inline)) { Kr[decltype(fc)::value] = read_k(decltype(fc)::value); });
            }
        }
        // Tile t+3 goes into the slot tile t-1 leaves: every wave finished reading it in the previous iteration.  Waves
        // with rows issue their share of the fetch from inside the iteration (beside MFMAs), idle waves here.
        auto top = [&](int t) __attribute__((always_inline)) {
            if (t > 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            if (t + 3 < n_act && (!wave_live || t + 1 >= n_act)) stage_tile((t + 3) & 3, entry(t + 3));
        };
        auto dma_of = [&](int t) __attribute__((always_inline)) { return (t + 3 < n_act) ? entry(t + 3) : 0u; };
        int t = 0;
        for (; t + 2 < n_act; t += 2) {   // two tiles per trip: the double-buffered probabilities of A keep fixed registers
            top(t);
            if (wave_live) iteration(P0{}, T_{}, e_cur, e_nxt, t & 3, (t + 1) & 3, true, dma_of(t));
            e_cur = e_nxt;
            e_nxt = entry(t + 2);
            top(t + 1);
            if (wave_live) iteration(P1{}, T_{}, e_cur, e_nxt, (t + 1) & 3, (t + 2) & 3, t + 3 < n_act, dma_of(t + 1));
            e_cur = e_nxt;
            e_nxt = entry(t + 3);
        }
        if (n_act - t == 2) {
            top(t);
            if (wave_live) iteration(P0{}, T_{}, e_cur, e_nxt, t & 3, (t + 1) & 3, false, 0u);
            e_cur = e_nxt;
            top(t + 1);
            if (wave_live) iteration(P1{}, F_{}, e_cur, 0u, (t + 1) & 3, (t + 2) & 3, false, 0u);
        } else {
            top(t);
            if (wave_live) iteration(P0{}, F_{}, e_cur, 0u, t & 3, (t + 1) & 3, false, 0u);
        }
    }
    // Every wave reports whether one of its rows outgrew its reference; the loop above ran to its end all the same (what
    // it stored is then garbage, possibly inf / nan: the slow kernel behind this one rewrites the whole item).
    if (lane == 0) a.fallback[blockIdx.x * 4 + wave] = bad ? 1 : 0;

    float l_tot[2];
    if (wave_live) MFMA_DRAIN();
    l_tot[0] = LS[0][0];
    l_tot[1] = LS[1][0];

    // ---- epilogue: lane holds O^T[d = 32 dt + (i&3) + 8 (i>>2) + 4 h][q = r] of both query blocks ----
    if (!wave_live) return;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        if (q_row[qb] > row_last) continue;
        const float l_t = l_tot[qb];
#ifndef VGPT_R64_STAMPS
        if (a.lse && h == 0)
            a.lse[((int64_t)b * a.n_heads + head) * a.L + q_row[qb]] =
                l_t > 0.f ? m_ref[qb] + __builtin_amdgcn_logf(l_t) : INFINITY;
#endif
        const float inv = l_t > 0.f ? 1.0f / l_t : 0.f;
        bf16* op = a.o + (int64_t)b * a.o_sb + (int64_t)head * a.o_sh + (int64_t)q_row[qb] * a.o_ss;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                bf16x4 o;
#pragma unroll
                for (int t = 0; t < 4; ++t) o[t] = f2bf(O[qb][dt][4 * g4 + t] * inv);
                *reinterpret_cast<bf16x4*>(op + dt * 32 + 8 * g4 + 4 * h) = o;
            }
    }
}

}  // namespace

// Experiment entry (NOT part of libvgpt_hip.so; `make experiment-r64` builds libvgpt_x_r64.so, scripts/attn_r64_probe.py
// drives it).  Same arguments as vgpt_attn_fwd_plan (include/vgpt.h) with items of up to 256 rows, plus
// flags (4 * n_items * n_heads int32): 1 where a wave met a row whose scores outgrew the fixed reference -- such an
// item's rows are void and a product version would recompute them with the online softmax of attn_fwd.hip.
extern "C" __attribute__((visibility("default"))) int vgpt_x_attn_fwd_r64(
    const void* q, const void* k, const void* v, void* o, float* lse, const uint32_t* bits, const int32_t* items,
    const uint16_t* item_summary, const int32_t* order, int64_t n_items, int32_t* flags, int64_t B, int64_t L, int n_heads,
    int n_kv_heads, int head_dim, int64_t q_sb, int64_t q_sh, int64_t q_ss, int64_t k_sb, int64_t k_sh, int64_t k_ss,
    int64_t v_sb, int64_t v_sh, int64_t v_ss, int64_t o_sb, int64_t o_sh, int64_t o_ss, float scale, void* stream) {
    if (!(q && k && v && o && bits && items && item_summary && order && flags) || head_dim != D || n_items <= 0 ||
        n_heads % n_kv_heads || ((q_ss | k_ss | v_ss | q_sh | k_sh | v_sh) & 7))
        return VGPT_ERR_INVALID;
    if (hipFuncSetAttribute((const void*)attn_fwd_r64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL) != hipSuccess)
        return VGPT_ERR_HIP;
    R64Args a;
    a.q = (const bf16*)q; a.k = (const bf16*)k; a.v = (const bf16*)v; a.o = (bf16*)o; a.lse = lse;
    a.bits = bits; a.items = items; a.isum = item_summary; a.order = order;
    a.n_items = (int)n_items; a.L = (int)L; a.n_heads = n_heads; a.kv_group = n_heads / n_kv_heads;
    a.W = (int)cdiv(L, 32); a.nkt = (int)cdiv(L, 64);
    a.q_sb = q_sb; a.q_sh = q_sh; a.q_ss = q_ss; a.k_sb = k_sb; a.k_sh = k_sh; a.k_ss = k_ss;
    a.v_sb = v_sb; a.v_sh = v_sh; a.v_ss = v_ss; a.o_sb = o_sb; a.o_sh = o_sh; a.o_ss = o_ss;
    a.scale_log2e = scale * 1.4426950408889634f;
    a.fallback = flags;
    hipLaunchKernelGGL(attn_fwd_r64_kernel, dim3((unsigned)(n_items * n_heads)), dim3(256), LDS_TOTAL, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? VGPT_OK : VGPT_ERR_HIP;
}
