// Block-masked flash attention forward for gfx950 (bf16 in/out, fp32 softmax and accumulation).
//
// Replaces module.local_attn = F.scaled_dot_product_attention with the additive block mask
// (LVM/transform/sdpa_transform.py:78-86,152; mask built at OmniGen/transformer.py:139-145 from
// the (B,L,L) bool masks of LVM/processor.py:575-731). The mask arrives bit-packed together with
// a per-tile summary (mask.hip), so arbitrary masks stay exact while all-masked tiles are skipped
// and all-visible tiles never read mask bits.
//
// Structure:
//   - workgroup = 4 waves = up to 128 query rows of one (batch, head); each wave owns 32 query rows; two
//     workgroups per CU.  The rows come either from an aligned 128-row q block or from a PLAN item (arbitrary row
//     range, vgpt_attn_fwd_plan), dispatched longest first with every XCD owning n_heads/8 heads.
//   - the key tiles that are visible to the item are compacted ONCE into an LDS list; per tile the K/V images of 64
//     keys (and the mask words of mixed tiles) arrive by LDS-DMA issued from inline asm into a double buffer, one
//     tile ahead; completion is s_waitcnt vmcnt(0) + a raw s_barrier at the top of the tile, so the loop contains no
//     compiler-visible memory traffic whose wait would drain the prefetch (head dims 64 / 128 stage through registers).
//   - S^T = K Q^T with mfma_f32_32x32x16_bf16 (keys on rows): each lane then holds 32 scores of ONE query row, so
//     row max / row sum are in-register ops + one v_permlane32_swap, and the exponentiated accumulator registers
//     are, unchanged, the B operand of O^T = V^T P^T.
//   - V^T fragments come from ds_read_b64_tr_b16 (hardware transposed read) on a [key][d] image whose row stride
//     keeps the four 64-byte windows of a half-wave on disjoint banks; the K image is XOR-swizzled on the DMA source
//     address (chunk ^= (key>>2)&3) so the ds_read_b128 16-lane groups are conflict-free without padding.
#include <type_traits>

#include "common.h"

// Softmax arithmetic of the tile loop.  0 (product): the plain online softmax.  1: three arithmetic reductions, built to
// parity and measured on the same box (`make attn-variant-VGPT_ATTN_LAZY`, round 3) -- an EXPERIMENT, not in the product:
//   (a) scale * log2(e) folded into Q once per work item (Q' = bf16(Q * c)), so a score needs no multiply;
//   (b) the running reference m_ref of a query row is LAZY: the score accumulators start at -m_ref (the C operand of the
//       first QK^T MFMA is a register block holding -m_ref) so that P = exp2(S) directly -- no subtract, no per-tile
//       alpha -- and the reference moves (and O, l are rescaled) only when a tile's largest score exceeds it by more than
//       LAZY_THRESH (2^8: P <= 256, far inside fp32 / bf16 range) or when the row sees its first key;
//   (c) the row sums l come from the matrix pipe: one more 32-row output tile of P V with an all-ones V block.
// Per tile and wave it issues 16 v_max3 + 32 v_exp + 16 v_cvt_pk and 28 MFMAs where the plain form issues 16 v_max3 +
// 16 v_pk_fma + 32 v_exp + 16 v_pk_add + 16 v_cvt_pk + the alpha bookkeeping and 24 MFMAs: a third fewer vector issue
// cycles.  Measured (cfg-2 live rows, in the step's own launch order, same box): 148.7 / 151.3 us against 155.9 us -- 3-5 %,
// 0.15 ms of a 31 ms step: the loop is not bound by what it issues.  And (a) costs accuracy exactly where logits are large:
// Q' is a second rounding of Q, an error proportional to the score; tests/test_ops_gpu.py::test_attention_late_spike
// (keys 30x the usual norm, scores ~ 400 in log2 units) keeps rel-L2 < 1e-2 but its max-abs bound reads 0.135 against
// 0.06.  Without (a) the multiply stays and nothing is saved.  Not worth the accuracy: the product keeps the plain form.
#ifndef VGPT_ATTN_LAZY
#define VGPT_ATTN_LAZY 0
#endif
#define LAZY_THRESH 8.0f
// Diagnostics build (make attn-variant-VGPT_ATTN_STAMPS, scripts/attn_stamps.py; results unchanged, timing perturbed by
// ~4 x 50 cycles per tile): wave 0 of every workgroup sums, over its tiles, the shader cycles (s_memtime) of four stretches
// of the loop -- top of the iteration -> behind the barrier | -> QK^T MFMAs issued | -> softmax done | -> P V issued -- into
// rows [cap, 2 cap) of the vgpt_attn_trace buffer, and its first / last s_memtime and tile count into rows [2 cap, 3 cap).
#ifndef VGPT_ATTN_STAMPS
#define VGPT_ATTN_STAMPS 0
#endif
// Round 3 also built a software-pipelined form of the tile loop (the QK^T MFMAs of tile t+1 issued inside the softmax of
// tile t through sched_group_barrier, K one tile ahead of V in the same two staging buffers; 252 registers, parity-green on
// the whole suite) and measured it SLOWER on the same box: 162.8 / 163.5 us against 151.3 us per layer at the cfg-2 live
// rows (31.75 vs 31.51 ms per step).  Overlapping a wave's own matrix and vector work does not help a loop whose two
// waves per SIMD already fill the issue port between them (active_inst_any 0.39 per wave, profiles/r02_pmc_mfma.json);
// kept out of the product as csrc/experiments/attn_fwd_pipelined_loop.inc.

#include "attn_p2_loop.inc"
#define VGPT_P2_DRAIN_0_M VGPT_P2_DRAIN_0
#define VGPT_P2_DRAIN_1_M VGPT_P2_DRAIN_1

namespace {

struct AttnArgs {
    const bf16* q;
    const bf16* k;
    const bf16* v;
    bf16* o;
    const uint32_t* bits;
    const uint8_t* summary;
    const int32_t* order;  // optional (B, nqb - qb0): q blocks in launch order (vgpt_attn_qblock_order)
    // planned launches (vgpt_attn_fwd_plan with 128-row items): work item = (items[order[rank]], head); then
    // `summary`/`order` above are unused
    const int32_t* items;      // n_items x 4: batch, row0, nrows (<= 128), 0
    const uint16_t* isum;      // n_items x nkt, 2 bits per 32-row slab
    const int32_t* iorder;     // n_items
    int n_items;
    float* lse;  // optional (B, n_heads, L): base-2 log-sum-exp of the scaled scores, for the backward
    int B, L, n_heads, kv_group;  // kv_group = n_heads / n_kv_heads
    int W;                        // mask words per row
    int nqb, nkt;                 // 128-row q blocks, 64-key tiles
    int qb0;                      // first q block computed (rows before qb0*128 are keys only: cached prefix)
    int64_t q_sb, q_sh, q_ss, k_sb, k_sh, k_ss, v_sb, v_sh, v_ss, o_sb, o_sh, o_ss;
    float scale_log2e;
    unsigned long long* trace;  // diagnostics (vgpt_attn_trace): 4 x u64 per workgroup, or null
    long long trace_cap;        // rows of one region of the trace buffer (the stamps build writes three regions)
};

// One 1-KiB LDS-DMA piece: lane i's 16 bytes at base + off land at lds_dst + 16 i.  Issued from inline asm so
// the compiler does not know LDS is being written: it would otherwise drain vmcnt(0) before every later LDS read
// of the OTHER buffer (and at __syncthreads), exposing the prefetch latency.  Completion is tracked by hand:
// s_waitcnt vmcnt(0) + a raw s_barrier at the top of the tile loop.
__device__ __forceinline__ void glds16(const char* base, uint32_t off, uint32_t lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(off), "s"(base), "s"(lds_dst)
        : "memory");
}

// 4-byte variant (lane i's dword lands at lds_dst + 4 i): the mask words of a mixed tile.
__device__ __forceinline__ void glds4(const uint32_t* src, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(lds_dst)
                 : "memory");
}

#if VGPT_ATTN_STAMPS
__device__ __forceinline__ unsigned long long stamp_now() {
    unsigned long long t;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
#define STAMP(var) const unsigned long long var = stamp_now()
#else
#define STAMP(var)
#endif

template <int D>
struct Cfg {
    // D == 96: K/V tiles are contiguous [key][192 B] images filled by LDS-DMA (global_load_lds); the K
    // image is XOR-swizzled (chunk ^= (key>>2)&3, applied on the DMA source address and on the read) so the
    // ds_read_b128 groups are conflict-free without padding.  Other head dims stage through registers
    // into a padded K image.
    // (round 4: head dim 96 staged through registers like the other head dims -- global_load + ds_write instead of LDS-DMA --
    // measured 158 us per layer in the step against 145 us, same box: the LDS-DMA stays)
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

// NW = waves per workgroup: 4 (128 query rows, two workgroups per CU) or 8 (256 rows, one workgroup per CU, planned launches
// with 256-row items, head dim 96 on the LDS-DMA path): the same K / V tile then serves twice the query rows, so every wave
// issues half the LDS-DMA pieces per tile (3 instead of 6) and the CU takes in half the bytes per FLOP.
// P2: runs of tiles every wave sees in full go through the hand-scheduled, software-pipelined tile bodies of
// gen/attn_p2_gen.py (QK^T of tile t+1 beside the exponentials of tile t, P.V of tile t beside the maxima of tile t+1, packed
// fp32 instructions only where no MFMA is in flight); mixed tiles and everything around the tiles stay in C++.  Same
// arithmetic per element: bit-identical to the C++ body.
template <int D, bool TR, int NW = 4, bool P2 = false>
__global__ __launch_bounds__(64 * NW, (NW == 8 ? 2 : (D <= 96 ? 2 : 1))) void attn_fwd_kernel(AttnArgs a) {
    static_assert(NW == 4 || (NW == 8 && D == 96 && TR), "8 waves: head dim 96, LDS-DMA path");
    static_assert(!P2 || (D == 96 && TR && NW == 4), "hand-scheduled bodies: head dim 96, four waves");
    constexpr int CODE_BITS = 2 * NW;                       // summary bits of one tile in an active-list entry
    constexpr uint32_t CODE_MASK = (1u << CODE_BITS) - 1u;
    using C = Cfg<D>;
    constexpr int KS = D / 16;  // k-steps of the QK^T product
    constexpr int DT = D / 32;  // 32-wide d tiles of the output
    constexpr int STAGE = C::KBYTES + vbytes<D, TR>();
    constexpr int ACT_MAX = 1023;  // tiles per active-list chunk (list = 4 KiB behind the staging buffers)
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const unsigned long long t_start = a.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
    int n_tiles_done = 0;
#if VGPT_ATTN_STAMPS
    unsigned long long ph[4] = {0ull, 0ull, 0ull, 0ull};
    const unsigned long long c_start = stamp_now();
#endif

    // ---- work item ----
    int wid = blockIdx.x, head, b, row0, row_last;
    const uint8_t* sum8 = nullptr;
    const uint16_t* sum16 = nullptr;
    if (a.items) {
        // XCD x (workgroups x, x+8, ...) owns n_heads/8 consecutive heads -- their K/V stay in its L2 -- and walks the
        // items longest first, so every head's long items start before any short one
        int rank;
        if ((a.n_heads & 7) == 0) {
            const int per = a.n_heads >> 3, j = wid >> 3;
            rank = j / per;
            head = (wid & 7) * per + j % per;
        } else {
            rank = wid / a.n_heads;
            head = wid % a.n_heads;
        }
        const int item = a.iorder[rank];
        b = a.items[4 * item];
        row0 = a.items[4 * item + 1];
        row_last = row0 + a.items[4 * item + 2] - 1;
        sum16 = a.isum + (int64_t)item * a.nkt;
        wid = item;
    } else {
        // aligned 128-row q blocks; keep all q blocks of a (batch, head) on one XCD
        const int nqa = a.nqb - a.qb0;  // q blocks actually computed
        const int nbh = a.n_heads * a.B;
        const int total = nqa * nbh;
        int qrank, bh;
        if ((nbh & 7) == 0) {
            const int per = nbh >> 3, j = wid >> 3;
            qrank = j / per;
            bh = (wid & 7) * per + j % per;
            wid = bh * nqa + qrank;
        } else {
            const int xcd = wid & 7, qn = total >> 3, rn = total & 7;
            wid = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (wid >> 3);
            qrank = wid % nqa;
            bh = wid / nqa;
        }
        head = bh % a.n_heads;
        b = bh / a.n_heads;
        const int qb = a.order ? a.order[b * nqa + qrank] : a.qb0 + qrank;
        row0 = qb * 128;
        row_last = min(row0 + 127, a.L - 1);
        sum8 = a.summary + ((int64_t)b * a.nqb + qb) * a.nkt;
    }
    const int kvh = head / a.kv_group;
    const bf16* kbase = a.k + (int64_t)b * a.k_sb + (int64_t)kvh * a.k_sh;
    const bf16* vbase = a.v + (int64_t)b * a.v_sb + (int64_t)kvh * a.v_sh;

    // ---- Q fragments (B operand of S^T = K Q^T): lane holds Q[q][16s + 8h .. +8) ----
    const int q_row = row0 + wave * 32 + r;
    const int q_ld = min(q_row, row_last);
    const bf16* qp = a.q + (int64_t)b * a.q_sb + (int64_t)head * a.q_sh + (int64_t)q_ld * a.q_ss;
    bf16x8 Qf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) Qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s + 8 * h);
#if VGPT_ATTN_LAZY
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) Qf[s][j] = f2bf(bf2f(Qf[s][j]) * a.scale_log2e);
#endif

    f32x16 O[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) O[dt][i] = 0.f;
#if VGPT_ATTN_LAZY
    f32x16 Ol, Cm;      // row sums (every register of a lane: l of its query row); -m_ref, the C operand of S^T's first MFMA
#pragma unroll
    for (int i = 0; i < 16; ++i) { Ol[i] = 0.f; Cm[i] = 0.f; }
    float m_ref = 0.f;
    bool need_ref = true;   // this row has not seen a key yet: the first finite tile maximum becomes its reference
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = f2bf(1.0f);
#else
    float m_i = -INFINITY, l_i = 0.f;
#endif

    // ---- staging helpers ----
    constexpr bool GLDS = TR && C::GLDS;
    constexpr int PIECES = (64 * D * 2) / 1024 / 4;  // 1-KiB LDS-DMA pieces per wave per operand (4 waves share one operand)
    // 8 waves: waves 0-3 stage the K image, waves 4-7 the V image (3 pieces each); 4 waves: every wave stages both
    const int swave = NW == 8 ? (wave & 3) : wave;
    const bool do_k = NW == 4 || wave < 4, do_v = NW == 4 || wave >= 4;
    // per-lane (key, source chunk) of each DMA piece this wave issues
    // per-lane byte offset (inside a tile) of each 16-byte unit this wave moves; the tile base is wave-uniform
    int g_key[GLDS ? PIECES : 1];
    uint32_t g_koff[GLDS ? PIECES : 1], g_voff[GLDS ? PIECES : 1], g_kc[GLDS ? PIECES : 1], g_vc[GLDS ? PIECES : 1];
    if constexpr (GLDS) {
#pragma unroll
        for (int j = 0; j < PIECES; ++j) {
            const int unit = (swave * PIECES + j) * 64 + lane;  // 16-byte unit inside the tile image
            g_key[j] = unit / C::CHUNKS;
            g_vc[j] = (unit % C::CHUNKS) * 16;
            g_kc[j] = ((unit % C::CHUNKS) ^ ((g_key[j] >> 2) & 3)) * 16;
            g_koff[j] = (uint32_t)g_key[j] * (uint32_t)a.k_ss * 2u + g_kc[j];
            g_voff[j] = (uint32_t)g_key[j] * (uint32_t)a.v_ss * 2u + g_vc[j];
        }
    }
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane(
        (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
    auto glds_tile = [&](int buf, int kt, bool want_k = true, bool want_v = true) {
        if constexpr (GLDS) {
            const char* kt_base = reinterpret_cast<const char*>(kbase + (int64_t)kt * 64 * a.k_ss);
            const char* vt_base = reinterpret_cast<const char*>(vbase + (int64_t)kt * 64 * a.v_ss);
            const bool tail = kt * 64 + 64 > a.L;  // wave-uniform: clamp keys past the end onto the last row
#pragma unroll
            for (int j = 0; j < PIECES; ++j) {
                uint32_t ko = g_koff[j], vo = g_voff[j];
                if (tail) {
                    const uint32_t key = (uint32_t)(min(kt * 64 + g_key[j], a.L - 1) - kt * 64);
                    ko = key * (uint32_t)a.k_ss * 2u + g_kc[j];
                    vo = key * (uint32_t)a.v_ss * 2u + g_vc[j];
                }
                const uint32_t dst = lds_base + (uint32_t)(buf * STAGE + (swave * PIECES + j) * 1024);
                if (do_k && want_k) glds16(kt_base, ko, dst);
                if (do_v && want_v) glds16(vt_base, vo, dst + C::KBYTES);
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
    // ---- active-tile list.  The per-tile summary bytes of this q block are compacted ONCE into LDS
    //      (entry = tile << 8 | summary byte), ACT_MAX tiles at a time, so the tile loop never issues a global
    //      load of its own: any such load would sit in the in-order vmcnt queue behind the next tile's LDS-DMA
    //      and its wait would expose the whole DMA latency. ----
    uint32_t* alist = reinterpret_cast<uint32_t*>(smem + 2 * STAGE);
    // mask words of a mixed tile travel by LDS-DMA too (one dword per lane: row lane>>1, word lane&1), one tile
    // ahead like K/V, into a per-wave 256-byte slot: the tile loop then holds no compiler-tracked global load.
    constexpr int MASK_OFF = 2 * STAGE + 4096;
    const uint32_t* mrow_src =
        a.bits + ((int64_t)b * a.L + min(row0 + wave * 32 + (lane >> 1), row_last)) * a.W;
    auto mask_dma = [&](int buf_, uint32_t e) {
        if (((e >> (2 * wave)) & 3) == 2) {
            const int t = (int)(e >> CODE_BITS);
            glds4(mrow_src + min(2 * t + (lane & 1), a.W - 1), lds_base + (uint32_t)(MASK_OFF + buf_ * (NW * 256) + wave * 256));
        }
    };
    auto stage = [&](int buf_, int kt_) {
        if constexpr (GLDS) glds_tile(buf_, kt_); else gload(kt_);
    };
    // Pin the Q fragments: the compiler must retire their global loads HERE.  Left pending, their first use sits
    // inside the tile loop and the vmcnt(0) emitted for it would also drain the (asm-issued) LDS-DMA in flight.
#pragma unroll
    for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(Qf[s]));
#ifdef VGPT_ATTN_PRIO
    // experiment (make attn-variant-VGPT_ATTN_PRIO VAL=n): a static s_setprio 1 for one of the two workgroups of a CU
    if constexpr (P2) { if ((blockIdx.x >> VGPT_ATTN_PRIO) & 1) __builtin_amdgcn_s_setprio(1); }
#endif
    if constexpr (P2) asm volatile(VGPT_P2_INIT ::: VGPT_P2_CLOBBERS);
    for (int chunk0 = 0; chunk0 < a.nkt; chunk0 += ACT_MAX) {
    __syncthreads();  // every wave is done with the previous list and the staging buffers
    if (wave == 0) {
        const int lim = min(chunk0 + ACT_MAX, a.nkt);
        auto code_of_tile = [&](int t) { return t < lim ? (sum16 ? (uint32_t)sum16[t] & CODE_MASK : (uint32_t)sum8[t]) : 0u; };
        int n = 0;
        for (int base = chunk0; base < lim; base += 64) {
            const int t = base + lane;
            const uint32_t c = code_of_tile(t);
            const uint64_t bal = __ballot(c != 0);
            const int idx = n + __popcll(bal & ((1ull << lane) - 1));
            if (c) alist[1 + idx] = ((uint32_t)t << CODE_BITS) | c;
            n += __popcll(bal);
        }
        if (lane == 0) alist[0] = (uint32_t)n;
    }
    __syncthreads();
    const int n_act = __builtin_amdgcn_readfirstlane((int)alist[0]);
    if (n_act == 0) continue;
    n_tiles_done += n_act;
    uint32_t e_cur = __builtin_amdgcn_readfirstlane(alist[1]);
    uint32_t e_nxt = __builtin_amdgcn_readfirstlane(n_act > 1 ? alist[2] : 0u);
    int buf = 0;
    mask_dma(0, e_cur);
    stage(0, (int)(e_cur >> CODE_BITS));
    int it = 0;
    auto classic = [&]() {
        STAMP(sA);
        if constexpr (GLDS) {
            // this wave's share of the current tile has landed; the raw barrier then (a) publishes every wave's
            // share and (b) guarantees every wave is done reading the other buffer before it is refilled
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        } else {
            lds_store(buf);
            __syncthreads();
        }
        STAMP(sB);
        const uint32_t e_n2 = it + 2 < n_act ? alist[3 + it] : 0u;  // consumed at the end of the iteration
        if (it + 1 < n_act) {
            mask_dma(buf ^ 1, e_nxt);
            stage(buf ^ 1, (int)(e_nxt >> CODE_BITS));
        }
        const int code = (e_cur >> (2 * wave)) & 3;
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
#if VGPT_ATTN_LAZY
                S[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Kf[kb][0], Qf[0], Cm, 0, 0, 0);   // S starts at -m_ref
#pragma unroll
                for (int s = 1; s < KS; ++s)
                    S[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Kf[kb][s], Qf[s], S[kb], 0, 0, 0);
#else
#pragma unroll
                for (int i = 0; i < 16; ++i) S[kb][i] = 0.f;
#pragma unroll
                for (int s = 0; s < KS; ++s)
                    S[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Kf[kb][s], Qf[s], S[kb], 0, 0, 0);
#endif
            }
            STAMP(sC);
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
                // per score two integer instructions without VCC: the bit spread to 0 / ~0 (v_bfe_i32), then
                // (s & m) | (-inf & ~m) (v_bitop3_b32) -- as a select it was and + cmp + nop + cndmask per score
                const uint2 mw = *reinterpret_cast<const uint2*>(smem + MASK_OFF + buf * (NW * 256) + wave * 256 + r * 8);
                const uint32_t mw0 = mw.x, mw1 = (2 * (int)(e_cur >> CODE_BITS) + 1 < a.W) ? mw.y : 0u;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    const uint32_t w = (kb ? mw1 : mw0) >> (4 * h);
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int bit = (i & 3) + 8 * (i >> 2);
                        if constexpr (TR) {
                            const uint32_t m = (uint32_t)((int32_t)(w << (31 - bit)) >> 31);
                            S[kb][i] = __uint_as_float((__float_as_uint(S[kb][i]) & m) | (0xff800000u & ~m));
                        } else {   // the register-staged variant is at its register limit: plain select
                            S[kb][i] = ((w >> bit) & 1u) ? S[kb][i] : -INFINITY;
                        }
                    }
                }
            }
            float mxp[2] = {-INFINITY, -INFINITY};  // two chains: the max3 latency, not its issue, would pace one
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) mxp[kb] = fmaxf(mxp[kb], S[kb][i]);
#if VGPT_ATTN_LAZY
            // scores are in log2 units relative to the row's reference already: P = exp2(S) as it stands, unless a row
            // outgrew its reference (or has none yet) -- then that row's reference moves to the tile maximum
            const float mx = half_max(fmaxf(mxp[0], mxp[1]));
            if (__any(need_ref || mx > LAZY_THRESH)) {
                const bool fin = mx > -INFINITY;
                const float delta = need_ref ? (fin ? mx : 0.f) : (mx > LAZY_THRESH ? mx : 0.f);
                const float alpha = need_ref ? 1.f : __builtin_amdgcn_exp2f(-delta);   // a row without a key holds O = l = 0
                need_ref = need_ref && !fin;
                m_ref += delta;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) S[kb][i] -= delta;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) O[dt][i] *= alpha;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    Ol[i] *= alpha;
                    Cm[i] = -m_ref;
                }
            }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) S[kb][i] = __builtin_amdgcn_exp2f(S[kb][i]);
#else
            const float mx = half_max(fmaxf(mxp[0], mxp[1])) * a.scale_log2e;  // scale > 0: max commutes with it
            const float m_new = fmaxf(m_i, mx);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m_i - m_use);
            // two values per VALU instruction where the ISA has a packed fp32 form (v_pk_fma_f32, v_pk_add_f32):
            // accumulator registers 2j, 2j+1 are an aligned pair; same per-element arithmetic and summation order
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const f32x2 sc2 = {a.scale_log2e, a.scale_log2e}, nm2 = {-m_use, -m_use};
            f32x2 rsp[2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    f32x2 t = {S[kb][i], S[kb][i + 1]};
                    t = __builtin_elementwise_fma(t, sc2, nm2);
                    t[0] = __builtin_amdgcn_exp2f(t[0]);
                    t[1] = __builtin_amdgcn_exp2f(t[1]);
                    S[kb][i] = t[0];
                    S[kb][i + 1] = t[1];
                    rsp[(i >> 1) & 1] += t;
                }
            const float rs = half_sum((rsp[0][0] + rsp[0][1]) + (rsp[1][0] + rsp[1][1]));
            l_i = l_i * alpha + rs;
            // rescale the accumulator only when some row's running max moved (wave-uniform branch)
            if (__any(m_new != m_i)) {
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) O[dt][i] *= alpha;
            }
            m_i = m_new;
#endif
            STAMP(sD);
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
#if VGPT_ATTN_LAZY
#pragma unroll
            for (int t = 0; t < 4; ++t) Ol = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, Pf[t], Ol, 0, 0, 0);   // l += sum_k P
#endif
#if VGPT_ATTN_STAMPS
            {
                STAMP(sE);
                ph[0] += sB - sA; ph[1] += sC - sB; ph[2] += sD - sC; ph[3] += sE - sD;
            }
#endif
        }
        e_cur = e_nxt;
        e_nxt = __builtin_amdgcn_readfirstlane(e_n2);
        buf ^= 1;
        ++it;
    };
    if constexpr (!P2) {
        while (it < n_act) classic();
    } else {
        // per-lane LDS addresses of the fragment reads; buffer, key half, k-step and d block are immediates of the bodies
        const uint32_t sw = (uint32_t)(r >> 2) & 3u;
        const uint32_t ka0 = lds_base + (uint32_t)r * C::KROW + (((uint32_t)h) ^ sw) * 16u;
        const uint32_t ka1 = lds_base + (uint32_t)r * C::KROW + ((2u + (uint32_t)h) ^ sw) * 16u;
        const int li = lane & 15;
        const uint32_t va = lds_base + (uint32_t)(4 * h + (li >> 2)) * C::VROW_TR + (uint32_t)((((lane >> 4) & 1) * 16 + 4 * (li & 3)) * 2);
        static_assert(STAGE == 24576 && C::KBYTES == 12288 && C::KROW == 192 && C::VROW_TR == 192, "layout constants of gen/attn_p2_gen.py");
        const float scale = a.scale_log2e;
        const unsigned long long scale2 = ((unsigned long long)__float_as_uint(scale) << 32) | __float_as_uint(scale);
        const uint32_t ninf = 0xff800000u;
        [[maybe_unused]] uint32_t m0_keep;
        // (no outputs: O, m, l, alpha and S stay in v[72:191] from body to body -- VGPT_P2_INIT / VGPT_P2_EXPORT around the item)
#define VGPT_P2_OPERANDS                                                                                                        \
        : [m0keep] "=&s"(m0_keep)                                                                                                 \
        : [q0] "v"(Qf[0]), [q1] "v"(Qf[1]), [q2] "v"(Qf[2]), [q3] "v"(Qf[3]), [q4] "v"(Qf[4]), [q5] "v"(Qf[5]), [ka0] "v"(ka0),   \
          [ka1] "v"(ka1), [va] "v"(va), [scale] "s"(scale), [scale2] "s"(scale2), [ninf] "s"(ninf), [mw0] "v"(mw0), [mw1] "v"(mw1),                     \
          [kbase] "s"(k_src), [vbase] "s"(v_src), [kdst] "s"(k_dst), [vdst] "s"(v_dst), [ko0] "v"(ko[0]), [ko1] "v"(ko[1]),       \
          [ko2] "v"(ko[2]), [vo0] "v"(vo[0]), [vo1] "v"(vo[1]), [vo2] "v"(vo[2])                                                  \
        : VGPT_P2_CLOBBERS
#define VGPT_P2_BODY(KIND, masked)                                                                     \
        do {                                                                                           \
            if (buf == 0) {                                                                            \
                if (masked) asm volatile(VGPT_P2_##KIND##_0_M VGPT_P2_OPERANDS);                       \
                else asm volatile(VGPT_P2_##KIND##_0 VGPT_P2_OPERANDS);                                \
            } else {                                                                                   \
                if (masked) asm volatile(VGPT_P2_##KIND##_1_M VGPT_P2_OPERANDS);                       \
                else asm volatile(VGPT_P2_##KIND##_1 VGPT_P2_OPERANDS);                                \
            }                                                                                          \
        } while (0)
        auto top_sync = [&]() {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };
        // mask words of this wave's rows in tile entry E (slot of staging buffer BUFM): a wave that sees nothing of the tile
        // masks every key (its scores become -inf, P = 0: state unchanged)
#define VGPT_P2_MASK_WORDS(BUFM, E)                                                                                              \
        uint32_t mw0 = 0u, mw1 = 0u;            /* key bits of the lane's query row in the tile, >> 4 h */                       \
        const int code_ = ((E) >> (2 * wave)) & 3;                                                                               \
        const bool masked = code_ != 1;                                                                                          \
        if (code_ == 2) {                                                                                                        \
            const uint2 mw = *reinterpret_cast<const uint2*>(smem + MASK_OFF + (BUFM) * (NW * 256) + wave * 256 + r * 8);        \
            mw0 = mw.x >> (4 * h);                                                                                               \
            mw1 = ((2 * (int)((E) >> CODE_BITS) + 1 < a.W) ? mw.y : 0u) >> (4 * h);                                              \
        }
        // ---- first tile: S^T and the maxima ----
        top_sync();
        uint32_t e_n2 = __builtin_amdgcn_readfirstlane(n_act > 2 ? alist[3] : 0u);
        if (n_act > 1) {
            mask_dma(1, e_nxt);
            stage(1, (int)(e_nxt >> CODE_BITS));
        }
        {
            const char *k_src = nullptr, *v_src = nullptr;      // (operands of the steady bodies' LDS-DMA: unused here)
            const uint32_t k_dst = 0u, v_dst = 0u, ko[3] = {0u, 0u, 0u}, vo[3] = {0u, 0u, 0u};
            VGPT_P2_MASK_WORDS(0, e_cur)
            VGPT_P2_BODY(PRO, masked);
        }
        // ---- pipelined bodies: exponentials and P.V of tile `it`, QK^T and maxima of tile it + 1; the body also issues the
        //      LDS-DMA of K(it+2) (over K(it)) and V(it+1) (over V(it-1)), spread between its MFMAs.  Where there is no tile
        //      it + 2 the K image of the current tile is fetched again onto itself, and the first body repeats the V image the
        //      full stage above already requested: same bytes to the same place, no branch in the body ----
        uint32_t e_n3v = n_act > 3 ? alist[4] : 0u;       // entry it + 3, read one body ahead of its use
        for (; it + 1 < n_act; ++it) {
            // (the operands of the body's LDS-DMA are formed BEFORE the wave parks at the barrier)
            const int kt_k = (int)((it + 2 < n_act ? e_n2 : e_cur) >> CODE_BITS), kt_v = (int)(e_nxt >> CODE_BITS);
            const char* k_src = reinterpret_cast<const char*>(kbase + (int64_t)kt_k * 64 * a.k_ss);
            const char* v_src = reinterpret_cast<const char*>(vbase + (int64_t)kt_v * 64 * a.v_ss);
            // per-lane source offsets of the six pieces; a tile that runs past the last key clamps its keys onto it -- rare, so
            // the (key, chunk) of a lane's units are recomputed there instead of being held in nine registers across the bodies
            uint32_t ko[PIECES], vo[PIECES];
#pragma unroll
            for (int j = 0; j < PIECES; ++j) {
                ko[j] = g_koff[j];
                vo[j] = g_voff[j];
            }
            if (kt_k * 64 + 64 > a.L || kt_v * 64 + 64 > a.L) {
                int lane_t = threadIdx.x & 63;
                asm volatile("" : "+v"(lane_t));      // (opaque: keeps hipcc from hoisting these out of the loop and spilling them)
#pragma unroll
                for (int j = 0; j < PIECES; ++j) {
                    const int unit = (wave * PIECES + j) * 64 + lane_t, key = unit / C::CHUNKS, ch = unit % C::CHUNKS;
                    if (kt_k * 64 + 64 > a.L)
                        ko[j] = (uint32_t)(min(kt_k * 64 + key, a.L - 1) - kt_k * 64) * (uint32_t)a.k_ss * 2u + (uint32_t)((ch ^ ((key >> 2) & 3)) * 16);
                    if (kt_v * 64 + 64 > a.L)
                        vo[j] = (uint32_t)(min(kt_v * 64 + key, a.L - 1) - kt_v * 64) * (uint32_t)a.v_ss * 2u + (uint32_t)(ch * 16);
                }
            }
            const uint32_t k_dst = lds_base + (uint32_t)(buf * STAGE + wave * PIECES * 1024);
            const uint32_t v_dst = lds_base + (uint32_t)((buf ^ 1) * STAGE + wave * PIECES * 1024) + C::KBYTES;
            STAMP(sA);
            top_sync();                         // K(it+1), V(it) have landed; every wave is done with K(it) and V(it-1)
            STAMP(sB);
            if (it + 2 < n_act) mask_dma(buf, e_n2);                  // mask words of tile it + 2 over those of tile it
            VGPT_P2_MASK_WORDS(buf ^ 1, e_nxt)
            STAMP(sC);
            VGPT_P2_BODY(STEADY, masked);
#if VGPT_ATTN_STAMPS
            {
                STAMP(sD);
                ph[0] += sB - sA; ph[1] += sC - sB; ph[2] += sD - sC;
            }
#endif
            e_cur = e_nxt;
            e_nxt = e_n2;
            e_n2 = __builtin_amdgcn_readfirstlane(e_n3v);
            e_n3v = it + 4 < n_act ? alist[5 + it] : 0u;
            buf ^= 1;
        }
        // ---- the last tile ----
        top_sync();
        {
            const char *k_src = nullptr, *v_src = nullptr;
            const uint32_t k_dst = 0u, v_dst = 0u, ko[3] = {0u, 0u, 0u}, vo[3] = {0u, 0u, 0u};
            const uint32_t mw0 = 0u, mw1 = 0u;
            VGPT_P2_BODY(DRAIN, false);
        }
        ++it;
#undef VGPT_P2_BODY
#undef VGPT_P2_MASK_WORDS
#undef VGPT_P2_OPERANDS
    }
    }

    if constexpr (P2) asm volatile(VGPT_P2_EXPORT : "=&v"(O[0][0]), "=&v"(O[0][1]), "=&v"(O[0][2]), "=&v"(O[0][3]), "=&v"(O[0][4]), "=&v"(O[0][5]), "=&v"(O[0][6]), "=&v"(O[0][7]), "=&v"(O[0][8]), "=&v"(O[0][9]), "=&v"(O[0][10]), "=&v"(O[0][11]), "=&v"(O[0][12]), "=&v"(O[0][13]), "=&v"(O[0][14]), "=&v"(O[0][15]), "=&v"(O[1][0]), "=&v"(O[1][1]), "=&v"(O[1][2]), "=&v"(O[1][3]), "=&v"(O[1][4]), "=&v"(O[1][5]), "=&v"(O[1][6]), "=&v"(O[1][7]), "=&v"(O[1][8]), "=&v"(O[1][9]), "=&v"(O[1][10]), "=&v"(O[1][11]), "=&v"(O[1][12]), "=&v"(O[1][13]), "=&v"(O[1][14]), "=&v"(O[1][15]), "=&v"(O[2][0]), "=&v"(O[2][1]), "=&v"(O[2][2]), "=&v"(O[2][3]), "=&v"(O[2][4]), "=&v"(O[2][5]), "=&v"(O[2][6]), "=&v"(O[2][7]), "=&v"(O[2][8]), "=&v"(O[2][9]), "=&v"(O[2][10]), "=&v"(O[2][11]), "=&v"(O[2][12]), "=&v"(O[2][13]), "=&v"(O[2][14]), "=&v"(O[2][15]), "=&v"(m_i), "=&v"(l_i) :: VGPT_P2_CLOBBERS);
    if (a.trace && tid == 0) {
        unsigned hw_id, xcc_id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
        unsigned long long* t = a.trace + 4ull * blockIdx.x;
        t[0] = t_start;
        t[1] = __builtin_amdgcn_s_memrealtime();
        t[2] = ((unsigned long long)xcc_id << 32) | hw_id;
        t[3] = ((unsigned long long)(unsigned)wid << 32) | (unsigned)n_tiles_done;
#if VGPT_ATTN_STAMPS
        const unsigned long long c_end = stamp_now();
        unsigned long long* p1 = a.trace + 4ull * (a.trace_cap + blockIdx.x);
        unsigned long long* p2 = a.trace + 4ull * (2ull * a.trace_cap + blockIdx.x);
        p1[0] = ph[0]; p1[1] = ph[1]; p1[2] = ph[2]; p1[3] = ph[3];
        p2[0] = c_start; p2[1] = c_end; p2[2] = (unsigned long long)n_tiles_done; p2[3] = 0ull;
#endif
    }
    // ---- epilogue: lane holds O^T[d = 32dt + (i&3) + 8(i>>2) + 4h][q = r] ----
#if VGPT_ATTN_LAZY
    const float l_i = Ol[0], m_i = m_ref;   // every register of Ol holds the lane's row sum (all 64 keys of a tile: no half-sum)
#endif
    int lane_e = threadIdx.x & 63;
    if constexpr (P2) asm volatile("" : "+v"(lane_e));   // the hand-scheduled kernel leaves hipcc 64 registers in the tile loop: the
    const int r_e = lane_e & 31, h_e = lane_e >> 5;      // epilogue's lane-derived values are formed here, not carried through it
    const int q_row_e = row0 + wave * 32 + r_e;
    const bool q_valid_e = q_row_e <= row_last;
    if (a.lse && q_valid_e && h_e == 0)
        a.lse[((int64_t)b * a.n_heads + head) * a.L + q_row_e] = l_i > 0.f ? m_i + __builtin_amdgcn_logf(l_i) : INFINITY;
    if (q_valid_e) {
        const float inv = l_i > 0.f ? 1.0f / l_i : 0.f;
        bf16* op = a.o + (int64_t)b * a.o_sb + (int64_t)head * a.o_sh + (int64_t)q_row_e * a.o_ss;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                bf16x4 o;
#pragma unroll
                for (int t = 0; t < 4; ++t) o[t] = f2bf(O[dt][4 * g4 + t] * inv);
                *reinterpret_cast<bf16x4*>(op + dt * 32 + 8 * g4 + 4 * h_e) = o;
            }
    }
}

template <int D, bool TR, int NW = 4, bool P2 = false>
int launch(const AttnArgs& a, hipStream_t s) {
    constexpr int lds = 2 * (Cfg<D>::KBYTES + vbytes<D, TR>()) + 4096 + 2 * NW * 256;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)attn_fwd_kernel<D, TR, NW, P2>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            vgpt_set_error("vgpt_attn_blockmask_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
            return VGPT_ERR_HIP;
        }
        attr_set = true;
    }
    const int total = a.items ? a.n_items * a.n_heads : (a.nqb - a.qb0) * a.n_heads * a.B;
    hipLaunchKernelGGL((attn_fwd_kernel<D, TR, NW, P2>), dim3(total), dim3(64 * NW), lds, s, a);
    VGPT_CHECK_LAUNCH("vgpt_attn_blockmask_fwd");
    return VGPT_OK;
}

}  // namespace

static unsigned long long* g_trace = nullptr;
static int64_t g_trace_cap = 0;

VGPT_EXPORT int vgpt_attn_trace(void* buf, int64_t capacity_workgroups) {
    g_trace = (unsigned long long*)buf;
    g_trace_cap = buf ? capacity_workgroups : 0;
    return VGPT_OK;
}

VGPT_EXPORT int vgpt_attn_supported(int head_dim) {
    return head_dim == 64 || head_dim == 96 || head_dim == 128;
}

// Tile body of the head-dim-96 forward: 1 (default) = the hand-scheduled bodies of gen/attn_p2_gen.py, 0 = the compiler-scheduled
// body (bit-identical results).  vgpt_attn_set_hand_scheduled (include/vgpt.h): a test and measurement switch; the environment
// variable VGPT_ATTN_P2=0 sets the initial value, read once.
static int g_attn_p2 = -1;
static bool attn_p2_enabled() {
    if (g_attn_p2 < 0) {
        const char* e = getenv("VGPT_ATTN_P2");
        g_attn_p2 = (e && e[0] == '0') ? 0 : 1;
    }
    return g_attn_p2 != 0;
}
VGPT_EXPORT int vgpt_attn_set_hand_scheduled(int on) {
    const int prev = attn_p2_enabled() ? 1 : 0;
    if (on == 0 || on == 1) g_attn_p2 = on;
    return prev;
}

struct ItemPlan {
    const int32_t* items;
    const uint16_t* isum;
    const int32_t* order;
    int64_t n_items;
    int item_rows;   // upper bound of the items' row counts: 128 (4-wave kernel) or 256 (8-wave kernel, head dim 96)
};

static int attn_fwd_impl(const void* q, const void* k, const void* v, void* o, float* lse, int64_t q_start,
                         const uint32_t* bits, const uint8_t* summary, const int32_t* order, const ItemPlan* plan,
                         int64_t B,
                         int64_t L, int n_heads, int n_kv_heads, int head_dim,
                         int64_t q_sb, int64_t q_sh, int64_t q_ss, int64_t k_sb,
                         int64_t k_sh, int64_t k_ss, int64_t v_sb, int64_t v_sh,
                         int64_t v_ss, int64_t o_sb, int64_t o_sh, int64_t o_ss,
                         float scale, int variant, void* stream) {
    VGPT_REQUIRE(q && k && v && o && bits && (summary || plan), VGPT_ERR_INVALID,
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
    VGPT_REQUIRE(k_ss > 0 && v_ss > 0 && k_ss < (1 << 24) && v_ss < (1 << 24), VGPT_ERR_UNSUPPORTED,
                 "vgpt_attn_blockmask_fwd: key/value row strides must be in (0, 2^24) elements");
    VGPT_REQUIRE(q_start >= 0 && q_start % 128 == 0 && q_start <= L, VGPT_ERR_INVALID,
                 "vgpt_attn_blockmask_fwd: q_start must be a multiple of 128 in [0, L]");
    if (B == 0 || L == 0 || q_start >= L) return VGPT_OK;
    if (plan && plan->n_items == 0) return VGPT_OK;
    AttnArgs a;
    a.q = (const bf16*)q; a.k = (const bf16*)k; a.v = (const bf16*)v; a.o = (bf16*)o;
    a.bits = bits; a.summary = summary; a.order = order; a.lse = lse;
    a.B = (int)B; a.L = (int)L; a.n_heads = n_heads; a.kv_group = n_heads / n_kv_heads;
    a.W = (int)cdiv(L, 32); a.nqb = (int)cdiv(L, 128); a.nkt = (int)cdiv(L, 64); a.qb0 = (int)(q_start / 128);
    a.q_sb = q_sb; a.q_sh = q_sh; a.q_ss = q_ss; a.k_sb = k_sb; a.k_sh = k_sh; a.k_ss = k_ss;
    a.v_sb = v_sb; a.v_sh = v_sh; a.v_ss = v_ss; a.o_sb = o_sb; a.o_sh = o_sh; a.o_ss = o_ss;
    a.scale_log2e = scale * 1.4426950408889634f;
    a.items = plan ? plan->items : nullptr;
    a.isum = plan ? plan->isum : nullptr;
    a.iorder = plan ? plan->order : nullptr;
    a.n_items = plan ? (int)plan->n_items : 0;
    const int64_t n_wg = plan ? plan->n_items * n_heads : (int64_t)(a.nqb - a.qb0) * n_heads * B;
    a.trace = n_wg <= g_trace_cap ? g_trace : nullptr;
    a.trace_cap = g_trace_cap;
    hipStream_t s = (hipStream_t)stream;
    int rc = VGPT_ERR_UNSUPPORTED;
#define ATTN_CASE(DD)                                                                     \
    case DD:                                                                              \
        rc = variant == 0 ? launch<DD, true>(a, s) : launch<DD, false>(a, s);             \
        break;
    if (plan && plan->item_rows == 256) {
        VGPT_REQUIRE(head_dim == 96 && variant == 0, VGPT_ERR_UNSUPPORTED,
                     "vgpt_attn_fwd_plan: 256-row items need head_dim 96 (got %d)", head_dim);
        return launch<96, true, 8>(a, s);
    }
    if (head_dim == 96 && variant == 0 && attn_p2_enabled()) return launch<96, true, 4, true>(a, s);
    switch (head_dim) {
        ATTN_CASE(64) ATTN_CASE(96) ATTN_CASE(128)
    }
#undef ATTN_CASE
    return rc;
}

VGPT_EXPORT int vgpt_attn_blockmask_fwd(const void* q, const void* k, const void* v, void* o,
                                        const uint32_t* bits, const uint8_t* summary, int64_t B,
                                        int64_t L, int n_heads, int n_kv_heads, int head_dim,
                                        int64_t q_sb, int64_t q_sh, int64_t q_ss, int64_t k_sb,
                                        int64_t k_sh, int64_t k_ss, int64_t v_sb, int64_t v_sh,
                                        int64_t v_ss, int64_t o_sb, int64_t o_sh, int64_t o_ss,
                                        float scale, int variant, void* stream) {
    return attn_fwd_impl(q, k, v, o, nullptr, 0, bits, summary, nullptr, nullptr, B, L, n_heads, n_kv_heads, head_dim, q_sb, q_sh, q_ss,
                         k_sb, k_sh, k_ss, v_sb, v_sh, v_ss, o_sb, o_sh, o_ss, scale, variant, stream);
}

VGPT_EXPORT int vgpt_attn_blockmask_fwd_lse(const void* q, const void* k, const void* v, void* o, float* lse,
                                            const uint32_t* bits, const uint8_t* summary, const int32_t* order, int64_t B,
                                            int64_t L, int n_heads, int n_kv_heads, int head_dim,
                                            int64_t q_sb, int64_t q_sh, int64_t q_ss, int64_t k_sb,
                                            int64_t k_sh, int64_t k_ss, int64_t v_sb, int64_t v_sh,
                                            int64_t v_ss, int64_t o_sb, int64_t o_sh, int64_t o_ss,
                                            float scale, void* stream) {
    if (!lse) {
        vgpt_set_error("vgpt_attn_blockmask_fwd_lse: null lse");
        return VGPT_ERR_INVALID;
    }
    return attn_fwd_impl(q, k, v, o, lse, 0, bits, summary, order, nullptr, B, L, n_heads, n_kv_heads, head_dim, q_sb, q_sh, q_ss,
                         k_sb, k_sh, k_ss, v_sb, v_sh, v_ss, o_sb, o_sh, o_ss, scale, 0, stream);
}

VGPT_EXPORT int vgpt_attn_blockmask_fwd_qrange(const void* q, const void* k, const void* v, void* o, int64_t q_start,
                                               const uint32_t* bits, const uint8_t* summary, const int32_t* order,
                                               int64_t B,
                                               int64_t L, int n_heads, int n_kv_heads, int head_dim,
                                               int64_t q_sb, int64_t q_sh, int64_t q_ss, int64_t k_sb,
                                               int64_t k_sh, int64_t k_ss, int64_t v_sb, int64_t v_sh,
                                               int64_t v_ss, int64_t o_sb, int64_t o_sh, int64_t o_ss,
                                               float scale, void* stream) {
    return attn_fwd_impl(q, k, v, o, nullptr, q_start, bits, summary, order, nullptr, B, L, n_heads, n_kv_heads, head_dim, q_sb, q_sh,
                         q_ss, k_sb, k_sh, k_ss, v_sb, v_sh, v_ss, o_sb, o_sh, o_ss, scale, 0, stream);
}

namespace {
// ---- plan: per (item, key tile) summary and the longest-first order ----
// block = 256 threads, one per row of an item (items hold at most 256 rows); 2 bits per 32-row slab
__global__ __launch_bounds__(256) void item_summary_kernel(const uint32_t* __restrict__ bits,
                                                           const int32_t* __restrict__ items,
                                                           uint16_t* __restrict__ isum, int L, int W, int nkt) {
    __shared__ int codes[8];
    const int kt = blockIdx.x, item = blockIdx.y;
    const int b = items[4 * item], row0 = items[4 * item + 1], nrows = items[4 * item + 2];
    const int i = threadIdx.x, lane = i & 63;
    bool none = true, all = true;
    if (i < nrows) {
        const uint32_t* rowp = bits + ((int64_t)b * L + row0 + i) * W;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int w = 2 * kt + half, k0 = w * 32;
            const uint32_t full = k0 + 32 <= L ? 0xffffffffu : 0u;  // a tile reaching past L is never "all visible"
            const uint32_t word = w < W ? rowp[w] : 0u;
            none = none && word == 0u;
            all = all && word == 0xffffffffu && full == 0xffffffffu;
        }
    }
    const uint64_t bn = __ballot(none), ba = __ballot(all);
    if ((lane & 31) == 0) {
        const uint32_t n32 = (uint32_t)(bn >> (lane & 32)), a32 = (uint32_t)(ba >> (lane & 32));
        codes[i >> 5] = n32 == 0xffffffffu ? 0 : (a32 == 0xffffffffu ? 1 : 2);
    }
    __syncthreads();
    if (i == 0) {
        int c = 0;
        for (int s = 0; s < 8; ++s) c |= codes[s] << (2 * s);
        isum[(int64_t)item * nkt + kt] = (uint16_t)c;
    }
}

constexpr int PLAN_SORT_MAX = 2048;
__global__ void item_order_kernel(const int32_t* __restrict__ items, const uint16_t* __restrict__ isum, int n, int nkt,
                                  int32_t* __restrict__ order) {
    __shared__ int cnt[PLAN_SORT_MAX];
    if (n > PLAN_SORT_MAX) {
        for (int i = threadIdx.x; i < n; i += blockDim.x) order[i] = i;
        return;
    }
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const uint16_t* row = isum + (int64_t)i * nkt;
        int c = 0;
        for (int t = 0; t < nkt; ++t) c += row[t] != 0;
        cnt[i] = c;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int ci = cnt[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += (cnt[j] > ci) || (cnt[j] == ci && j < i);
        order[rank] = i;
    }
}

}  // namespace

VGPT_EXPORT int64_t vgpt_attn_plan_workspace_bytes(int64_t L, int64_t n_items) {
    if (L <= 0 || n_items < 0) return -1;
    // item_summary (n_items x ceil(L/64) uint16), then the order (n_items int32), each rounded up to 256 bytes
    const int64_t s = (n_items * cdiv(L, 64) * 2 + 255) / 256 * 256, o = (n_items * 4 + 255) / 256 * 256;
    return s + o;
}

VGPT_EXPORT int vgpt_attn_plan_build(const uint32_t* bits, int64_t B, int64_t L, const int32_t* items, int64_t n_items,
                                     uint16_t* item_summary, int32_t* order, void* stream) {
    VGPT_REQUIRE(bits && items && item_summary && order, VGPT_ERR_INVALID, "vgpt_attn_plan_build: null pointer");
    VGPT_REQUIRE(B > 0 && L > 0 && L <= (1 << 22) && n_items > 0 && n_items < 65536, VGPT_ERR_INVALID,
                 "vgpt_attn_plan_build: bad shape");
    const int nkt = (int)cdiv(L, 64);
    hipLaunchKernelGGL(item_summary_kernel, dim3(nkt, (unsigned)n_items), dim3(256), 0, (hipStream_t)stream, bits, items,
                       item_summary, (int)L, (int)cdiv(L, 32), nkt);
    hipLaunchKernelGGL(item_order_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, items, item_summary, (int)n_items,
                       nkt, order);
    VGPT_CHECK_LAUNCH("vgpt_attn_plan_build");
    return VGPT_OK;
}

VGPT_EXPORT int vgpt_attn_fwd_plan(const void* q, const void* k, const void* v, void* o, float* lse, const uint32_t* bits,
                                   const int32_t* items, const uint16_t* item_summary, const int32_t* order,
                                   int64_t n_items, int64_t B, int64_t L, int n_heads, int n_kv_heads, int head_dim,
                                   int64_t q_sb, int64_t q_sh, int64_t q_ss, int64_t k_sb, int64_t k_sh, int64_t k_ss,
                                   int64_t v_sb, int64_t v_sh, int64_t v_ss, int64_t o_sb, int64_t o_sh, int64_t o_ss,
                                   float scale, int item_rows, void* stream) {
    VGPT_REQUIRE(items && item_summary && order, VGPT_ERR_INVALID, "vgpt_attn_fwd_plan: null pointer");
    VGPT_REQUIRE(item_rows == 128 || item_rows == 256, VGPT_ERR_INVALID, "vgpt_attn_fwd_plan: item_rows must be 128 or 256");
    const ItemPlan plan = {items, item_summary, order, n_items, item_rows};
    return attn_fwd_impl(q, k, v, o, lse, 0, bits, nullptr, nullptr, &plan, B, L, n_heads, n_kv_heads, head_dim, q_sb, q_sh, q_ss,
                         k_sb, k_sh, k_ss, v_sb, v_sh, v_ss, o_sb, o_sh, o_ss, scale, 0, stream);
}
