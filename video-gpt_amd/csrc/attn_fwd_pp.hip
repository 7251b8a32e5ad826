// Block-masked flash attention forward, head_dim 96, 8-wave "ping-pong" structure for gfx950.
//
// Same contract as attn_fwd.hip (module.local_attn = F.scaled_dot_product_attention with the block mask,
// LVM/transform/sdpa_transform.py:78-86,152) but the work is described by a PLAN (vgpt_attn_plan_build):
//   - work item = up to 256 consecutive query rows of one batch item (row0, nrows); items need not be aligned, so a
//     caller can cut them at the boundaries of its sequences (the packed CFG layout of the sampler puts two sequences
//     with different key sets back to back, LVM/pipeline.py:479-493) and no item straddles two key sets;
//   - per (item, 64-key tile) a 16-bit summary: 2 bits per 32-row wave slab (0 nothing visible, 1 everything, 2 mixed);
//   - a longest-first launch order of the items.
//
// Structure of one workgroup (8 waves = 256 rows, one workgroup per CU, two waves per SIMD):
//   - waves 0-3 (group A) and 4-7 (group B) run the same per-tile program
//         M(t): O += V_{t-1} P_{t-1} (12 MFMA, K_t fragments read from LDS underneath), S_t = K_t Q^T (12 MFMA)
//         V(t): softmax of S_t -> P_t (VALU), LDS reads of the V_t fragments, LDS-DMA of tile t+3
//     separated by workgroup barriers, with B one phase behind A: while one wave of a SIMD is in its MFMA phase the
//     other is in its VALU/LDS phase, so the matrix pipe and the vector ALU of every SIMD are busy together.
//   - K/V tiles live in a 4-slot LDS ring filled by LDS-DMA three tiles ahead (group A moves K, group B moves V, each
//     wave also its own mask words); completion is tracked with counted s_waitcnt vmcnt, never a drain.
//   - items of at most 32 rows ("thin": the 2064 = 8*256 + 16 rows of a cfg-2 sequence leave one) split the KEY tiles
//     over the 8 waves instead and merge the partial (m, l, O) through LDS.
#include "common.h"

namespace {

constexpr int D = 96;
constexpr int KS = D / 16;             // k-steps of S = K Q^T
constexpr int DT = D / 32;             // 32-wide d tiles of O^T
constexpr int OPB = 64 * D * 2;        // bytes of one operand tile (64 keys)
constexpr int RING = 4;
constexpr int SLOT = 2 * OPB;          // K image then V image
constexpr int LIST_OFF = RING * SLOT;  // active-tile list: 1 count + LIST_MAX entries
constexpr int LIST_MAX = 1023;
constexpr int MASK_OFF = LIST_OFF + 4096;          // RING x 8 waves x 256 B of mask words
constexpr int LDS_TOTAL = MASK_OFF + RING * 2048;  // 110 592 B
constexpr int CHUNKS = D / 8;                      // 16-byte chunks per key row
constexpr int COMB_STRIDE = 32 * D + 64;           // floats per wave in the thin-item merge buffer

struct PArgs {
    const bf16* q;
    const bf16* k;
    const bf16* v;
    bf16* o;
    float* lse;
    const uint32_t* bits;
    const int32_t* items;   // n_items x 4: batch, row0, nrows, 0
    const uint16_t* isum;   // n_items x nkt
    const int32_t* order;   // n_items
    int n_items, L, n_heads, kv_group, W, nkt;
    int64_t q_sb, q_sh, q_ss, k_sb, k_sh, k_ss, v_sb, v_sh, v_ss, o_sb, o_sh, o_ss;
    float scale_log2e;
    unsigned long long* trace;
};

__device__ __forceinline__ void glds16(const char* base, uint32_t off, uint32_t lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(off), "s"(base), "s"(lds_dst)
        : "memory");
}
__device__ __forceinline__ void glds4(const uint32_t* src, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(lds_dst)
                 : "memory");
}
// workgroup barrier that the compiler may not move memory operations across
__device__ __forceinline__ void bar() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Diagnostic build only (make stamps): workgroup 0 records shader-clock stamps of its first 32 tiles behind the
// per-workgroup trace records: [wave][tile][4] = M start, M end, V start, V end.
// VGPT_PP_EXPERIMENT (diagnostic builds, results are WRONG): 1 = no softmax arithmetic, 2 = no LDS-DMA after the
// prologue, 3 = no LDS fragment reads after the first tile, 4 = no barriers inside the tile loop
#ifndef VGPT_PP_EXPERIMENT
#define VGPT_PP_EXPERIMENT 0
#endif
#ifdef VGPT_PP_STAMPS
#define PP_STAMP(k)                                                                                   \
    if (a.trace && blockIdx.x == 0 && t < 32 && lane == 0)                                            \
    a.trace[4ull * gridDim.x + (wave * 32 + t) * 4 + (k)] = __builtin_amdgcn_s_memtime()
#else
#define PP_STAMP(k)
#endif

template <bool THIN>
__device__ __forceinline__ void item_body(const PArgs& a, char* smem, const int item, const int head, const int b,
                                          const int row0, const int nrows, const unsigned long long t_start) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;  // 0 = A (moves K), 1 = B (moves V, one phase behind)
    const int r = lane & 31, h = lane >> 5;
    int n_tiles_done = 0;
    constexpr bool thin = THIN;
    const int kvh = head / a.kv_group;
    const int slab = thin ? 0 : wave;  // 32-row slab this wave computes
    const int cshift = 2 * slab;       // its 2 bits in a summary entry

    const uint16_t* sum_row = a.isum + (int64_t)item * a.nkt;
    const bf16* kbase = a.k + (int64_t)b * a.k_sb + (int64_t)kvh * a.k_sh;
    const bf16* vbase = a.v + (int64_t)b * a.v_sb + (int64_t)kvh * a.v_sh;
    const int row_last = row0 + nrows - 1;

    // ---- Q fragments (B operand of S^T = K Q^T): lane holds Q[q][16s + 8h .. +8) ----
    const int q_row = row0 + slab * 32 + r;
    const bool q_valid = q_row <= row_last;
    const bf16* qp = a.q + (int64_t)b * a.q_sb + (int64_t)head * a.q_sh + (int64_t)min(q_row, row_last) * a.q_ss;
    bf16x8 Qf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) Qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s + 8 * h);

    const uint32_t lds_base =
        __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
    uint32_t* alist = reinterpret_cast<uint32_t*>(smem + LIST_OFF);

    // ---- LDS-DMA shares: wave w < 4 moves K pieces 3w..3w+2, wave w >= 4 the same pieces of V.  K image is
    //      XOR-swizzled (chunk ^= (key>>2)&3, on the source address) for conflict-free ds_read_b128 ----
    const bf16* mybase = grp ? vbase : kbase;
    const uint32_t my_ss = (uint32_t)(grp ? a.v_ss : a.k_ss);
    // per-lane source offset (inside a tile) of the three 16-byte units this lane moves; the rare tail tile
    // (keys past L clamped onto the last row) recomputes its offsets instead of keeping key/chunk in registers
    auto unit_off = [&](int j, int kt_tail) {
        const int unit = (((wave & 3) * 3 + j) * 64) + lane;
        int key = unit / CHUNKS;
        const int c = unit % CHUNKS;
        const uint32_t cb = (uint32_t)(grp ? c : (c ^ ((key >> 2) & 3))) * 16u;
        if (kt_tail >= 0) key = min(kt_tail * 64 + key, a.L - 1) - kt_tail * 64;
        return (uint32_t)key * my_ss * 2u + cb;
    };
    uint32_t g_off[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) g_off[j] = unit_off(j, -1);
    const uint32_t* mrow_src = a.bits + ((int64_t)b * a.L + min(row0 + slab * 32 + (lane >> 1), row_last)) * a.W;
    // A wave moves 3 operand pieces per tile, plus its mask words when the tile is mixed for its rows.  `issue_piece`
    // issues one of them (j = 0..2 operand pieces, j = 3 the mask words) so that the tile loop can spread them between
    // its MFMAs: the texture path takes 1 KiB per ~16 cycles per CU and a burst from 8 waves stalls the issuers.
    auto issue_piece = [&](int j, int kt, int slot) {
        if (j < 3) {
            const char* tb = reinterpret_cast<const char*>(mybase + (int64_t)kt * 64 * my_ss);
            uint32_t off = g_off[j];
            if (kt * 64 + 64 > a.L) off = unit_off(j, kt);
            glds16(tb, off, lds_base + (uint32_t)(slot * SLOT + grp * OPB + ((wave & 3) * 3 + j) * 1024));
        } else {
            glds4(mrow_src + min(2 * kt + (lane & 1), a.W - 1), lds_base + (uint32_t)(MASK_OFF + slot * 2048 + wave * 256));
        }
    };
    auto needs_mask = [&](uint32_t e) { return ((e >> cshift) & 3u) != 1u; };  // 0 (empty) is handled as mixed
    auto issue_tile = [&](uint32_t e, int slot) {
#pragma unroll
        for (int j = 0; j < 3; ++j) issue_piece(j, (int)(e >> 16), slot);
        if (needs_mask(e)) issue_piece(3, (int)(e >> 16), slot);
    };
    // wait until at most the LDS-DMA of `younger` whole tiles (given by their entries) is still in flight
    auto wait_tiles = [&](bool has1, uint32_t e1_, bool has2, uint32_t e2_) {
        const int n = (has1 ? 3 + (int)needs_mask(e1_) : 0) + (has2 ? 3 + (int)needs_mask(e2_) : 0);
        switch (n) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        }
    };

    f32x16 O[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) O[dt][i] = 0.f;
    float m_i = -INFINITY, l_i = 0.f;

    // pin the Q loads here: pending, their first use would sit in the tile loop and its vmcnt(0) would drain the DMA
#pragma unroll
    for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(Qf[s]));

    bf16x8 Kf[2][KS], Vf[DT][4], Pf[4];
    f32x16 S[2];
    auto read_k = [&](int slot) {
        const char* sk = smem + slot * SLOT;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const char* krow = sk + (kb * 32 + r) * (D * 2);
#pragma unroll
            for (int s = 0; s < KS; ++s)
                Kf[kb][s] = *reinterpret_cast<const bf16x8*>(krow + (((2 * s + h) ^ ((r >> 2) & 3)) * 16));
        }
    };
    auto read_v = [&](int slot) {
        const char* sv = smem + slot * SLOT + OPB;
        const int li = lane & 15;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int key0 = (t >> 1) * 32 + (t & 1) * 16 + 4 * h;
                const char* p0 = sv + (key0 + (li >> 2)) * (D * 2) + (dt * 32 + ((lane >> 4) & 1) * 16 + 4 * (li & 3)) * 2;
                bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(p0));
                bf16x4 hi =
                    __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(p0 + 8 * D * 2));
                Vf[dt][t] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
    };
    auto qk = [&]() {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int i = 0; i < 16; ++i) S[kb][i] = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) S[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Kf[kb][s], Qf[s], S[kb], 0, 0, 0);
        }
    };
    auto pv = [&]() {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int t = 0; t < 4; ++t) O[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Vf[dt][t], Pf[t], O[dt], 0, 0, 0);
    };
    // online softmax of S (scores of 64 keys for query row r; this lane holds keys with bit 2 of (key>>0)... == h)
    auto softmax = [&](int code, int slot, int kt) {
        if (code == 2) {
            const uint2 mw = *reinterpret_cast<const uint2*>(smem + MASK_OFF + slot * 2048 + wave * 256 + r * 8);
            const uint32_t mw0 = mw.x, mw1 = (2 * kt + 1 < a.W) ? mw.y : 0u;
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
        float mxp[2] = {-INFINITY, -INFINITY};  // two chains: the max3 latency, not its issue, would pace one
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) mxp[kb] = fmaxf(mxp[kb], S[kb][i]);
        float mx = half_max(fmaxf(mxp[0], mxp[1])) * a.scale_log2e;  // scale > 0: max commutes with it
        const float m_new = fmaxf(m_i, mx);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m_i - m_use);
        float rsp[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(S[kb][i], a.scale_log2e, -m_use));
                S[kb][i] = p;
                rsp[i & 3] += p;
            }
        float rs = half_sum((rsp[0] + rsp[1]) + (rsp[2] + rsp[3]));
        l_i = l_i * alpha + rs;
        if (__any(m_new != m_i)) {  // rescale only when some row's running max moved (wave-uniform)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int i = 0; i < 16; ++i) O[dt][i] *= alpha;
        }
        m_i = m_new;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) Pf[t][j] = f2bf(S[t >> 1][8 * (t & 1) + j]);
    };

    for (int chunk0 = 0; chunk0 < a.nkt; chunk0 += LIST_MAX) {
        __syncthreads();  // everyone is done with the previous list and the ring
        if (wave == 0) {
            const int lim = min(chunk0 + LIST_MAX, a.nkt);
            int n = 0;
            for (int base = chunk0; base < lim; base += 64) {
                const int t = base + lane;
                const uint32_t c = t < lim ? sum_row[t] : 0u;
                const uint64_t bal = __ballot(c != 0);
                if (c) alist[1 + n + __popcll(bal & ((1ull << lane) - 1))] = ((uint32_t)t << 16) | c;
                n += __popcll(bal);
            }
            if (lane == 0) alist[0] = (uint32_t)n;
        }
        __syncthreads();
        const int n_act = __builtin_amdgcn_readfirstlane((int)alist[0]);
        if (n_act == 0) continue;
        n_tiles_done += n_act;
        auto entry = [&](int i) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)alist[1 + i]); };
        auto code_of = [&](uint32_t e) { return (int)((e >> cshift) & 3u); };

        // ---- prologue: tiles 0..2 in flight, tile 0 (then 1) landed ----
        const uint32_t p0 = entry(0), p1 = n_act > 1 ? entry(1) : 0u, p2 = n_act > 2 ? entry(2) : 0u;
        issue_tile(p0, 0);
        if (n_act > 1) issue_tile(p1, 1);
        if (n_act > 2) issue_tile(p2, 2);
        if (grp && !thin) bar();  // group B runs one phase behind
        wait_tiles(n_act > 1, p1, n_act > 2, p2);  // tile 0 landed
        bar();

        if constexpr (!thin) {
            // entries of tiles t .. t+3 stay in scalar registers; the one of t+4 is fetched a phase ahead.
            // No per-wave skipping here: a tile that is empty for this wave's rows (code 0) is computed like a mixed
            // one -- its mask words are all zero, so it contributes exp2(-inf) = 0 -- which keeps the loop free of
            // data-dependent control flow (the waves would wait at the barriers anyway).
            uint32_t e_cur = p0, e1 = p1, e2 = p2, e3 = n_act > 3 ? entry(3) : 0u;
            wait_tiles(n_act > 2, p2, false, 0u);  // tile 1 landed as well
            for (int t = 0; t < n_act; ++t) {
                const int slot = t & (RING - 1);
                const bool more = t + 3 < n_act;
                const int kt3 = (int)(e3 >> 16), slot3 = (t + 3) & (RING - 1);
                if (VGPT_PP_EXPERIMENT != 4) bar();
                PP_STAMP(0);
                // ---- M(t): K_t fragments are requested first, their LDS latency hides under O += V_{t-1} P_{t-1};
                //      the LDS-DMA pieces of tile t+3 go out between the MFMA groups ----
                if (VGPT_PP_EXPERIMENT != 3 || t == 0) read_k(slot);
                if (t > 0) pv();
                if (more && VGPT_PP_EXPERIMENT != 2) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) issue_piece(j, kt3, slot3);
                }
                qk();
                if (more && needs_mask(e3) && VGPT_PP_EXPERIMENT != 2) issue_piece(3, kt3, slot3);
                if (VGPT_PP_EXPERIMENT != 2) wait_tiles(more, e3, false, 0u);  // tile t+2 landed
                PP_STAMP(1);
                if (VGPT_PP_EXPERIMENT != 4) bar();
                PP_STAMP(2);
                // ---- V(t): V_t fragments for the next M phase, softmax ----
                const uint32_t e4_raw = t + 4 < n_act ? alist[5 + t] : 0u;  // consumed at the end of the phase
                if (VGPT_PP_EXPERIMENT != 3 || t == 0) read_v(slot);
                if (VGPT_PP_EXPERIMENT != 1) {
                    softmax(code_of(e_cur) == 1 ? 1 : 2, slot, (int)(e_cur >> 16));
                } else {
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                        for (int j = 0; j < 8; ++j) Pf[tt][j] = f2bf(S[tt >> 1][8 * (tt & 1) + j]);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                PP_STAMP(3);
                e_cur = e1;
                e1 = e2;
                e2 = e3;
                e3 = (uint32_t)__builtin_amdgcn_readfirstlane((int)e4_raw);
            }
            bar();
            pv();
            if (!grp) bar();  // group A: the barrier B spent up front
        } else {
            // ---- thin item: the 8 waves take the active tiles round-robin for the same 32 rows; one barrier per
            //      tile publishes tile t+1 and retires the readers of tile t ----
            for (int t = 0; t < n_act; ++t) {
                const uint32_t e = entry(t);
                const int slot = t & (RING - 1);
                if ((t & 7) == wave && code_of(e)) {
                    read_k(slot);
                    read_v(slot);
                    qk();
                    softmax(code_of(e), slot, (int)(e >> 16));
                    pv();
                }
                const uint32_t t2 = t + 2 < n_act ? entry(t + 2) : 0u, t3 = t + 3 < n_act ? entry(t + 3) : 0u;
                if (t + 3 < n_act) issue_tile(t3, (t + 3) & (RING - 1));
                wait_tiles(t + 2 < n_act, t2, t + 3 < n_act, t3);  // tile t+1 landed
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                bar();
            }
        }
    }

    if constexpr (thin) {
        // ---- merge the 8 partial results: O = sum_w O_w 2^(m_w - m*), l likewise ----
        __syncthreads();
        float* comb = reinterpret_cast<float*>(smem);
        float* mine = comb + wave * COMB_STRIDE;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) mine[(dt * 32 + (i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = O[dt][i];
        if (h == 0) {
            mine[32 * D + r] = m_i;
            mine[32 * D + 32 + r] = l_i;
        }
        __syncthreads();
        float m_all = -INFINITY;
#pragma unroll
        for (int w = 0; w < 8; ++w) m_all = fmaxf(m_all, comb[w * COMB_STRIDE + 32 * D + r]);
        float l_all = 0.f, fac[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            const float mw = comb[w * COMB_STRIDE + 32 * D + r];
            fac[w] = (mw == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(mw - m_all);
            l_all += comb[w * COMB_STRIDE + 32 * D + 32 + r] * fac[w];
        }
        if (a.lse && q_valid && wave == 0 && h == 0)
            a.lse[((int64_t)b * a.n_heads + head) * a.L + q_row] =
                l_all > 0.f ? m_all + __builtin_amdgcn_logf(l_all) : INFINITY;
        const float inv = l_all > 0.f ? 1.0f / l_all : 0.f;
        // wave w finishes d columns [12 w, 12 w + 12): lane (r, h) takes 6 of them
        bf16* op = a.o + (int64_t)b * a.o_sb + (int64_t)head * a.o_sh + (int64_t)min(q_row, row_last) * a.o_ss;
        bf16x2 ov[3];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int d = wave * 12 + h * 6 + j;
            float acc = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) acc += comb[w * COMB_STRIDE + d * 32 + r] * fac[w];
            ov[j >> 1][j & 1] = f2bf(acc * inv);
        }
        if (q_valid) {
#pragma unroll
            for (int j = 0; j < 3; ++j) *reinterpret_cast<bf16x2*>(op + wave * 12 + h * 6 + 2 * j) = ov[j];
        }
    } else {
        // ---- epilogue: lane holds O^T[d = 32dt + (i&3) + 8(i>>2) + 4h][q = r] ----
        if (a.lse && q_valid && h == 0)
            a.lse[((int64_t)b * a.n_heads + head) * a.L + q_row] =
                l_i > 0.f ? m_i + __builtin_amdgcn_logf(l_i) : INFINITY;
        if (q_valid) {
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
    if (a.trace && tid == 0) {
        unsigned hw_id, xcc_id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
        unsigned long long* t = a.trace + 4ull * blockIdx.x;
        t[0] = t_start;
        t[1] = __builtin_amdgcn_s_memrealtime();
        t[2] = ((unsigned long long)xcc_id << 32) | hw_id;
        t[3] = ((unsigned long long)(unsigned)item << 32) | (unsigned)n_tiles_done;
    }
}

__global__ __launch_bounds__(512, 2) void attn_fwd_pp_kernel(PArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned long long t_start = a.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
    // ---- work item: XCD x (workgroups x, x+8, ...) owns n_heads/8 consecutive heads and walks the items
    //      longest first, so every head's long items start before any short one ----
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
    const int b = a.items[4 * item], row0 = a.items[4 * item + 1], nrows = a.items[4 * item + 2];
    if (nrows <= 32) item_body<true>(a, smem, item, head, b, row0, nrows, t_start);
    else item_body<false>(a, smem, item, head, b, row0, nrows, t_start);
}

// ---- plan: per (item, key tile) summary and the longest-first order ----
// block = 256 threads = the 256 rows of one item; 2 bits per 32-row slab
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
                                  int thin_rows, int32_t* __restrict__ order) {
    __shared__ int cnt[PLAN_SORT_MAX];
    if (n > PLAN_SORT_MAX) {
        for (int i = threadIdx.x; i < n; i += blockDim.x) order[i] = i;
        return;
    }
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const uint16_t* row = isum + (int64_t)i * nkt;
        int c = 0;
        for (int t = 0; t < nkt; ++t) c += row[t] != 0;
        const int nparts = max((items[4 * i + 3] >> 8) & 255, 1);   // key-split work items walk 1/nparts of the tiles
        cnt[i] = items[4 * i + 2] <= thin_rows ? (c + 7) / 8 : (c + nparts - 1) / nparts;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int ci = cnt[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += (cnt[j] > ci) || (cnt[j] == ci && j < i);
        order[rank] = i;
    }
}

unsigned long long* g_pp_trace = nullptr;
int64_t g_pp_trace_cap = 0;

}  // namespace

void vgpt_attn_pp_set_trace(void* buf, int64_t cap) {
    g_pp_trace = (unsigned long long*)buf;
    g_pp_trace_cap = buf ? cap : 0;
}

int vgpt_attn_fwd_items128(const void* q, const void* k, const void* v, void* o, float* lse, const uint32_t* bits,
                           const int32_t* items, const uint16_t* item_summary, const int32_t* order, int64_t n_items,
                           const int32_t* split_items, int64_t n_split, float* split_ws, int64_t B, int64_t L, int n_heads,
                           int n_kv_heads, int head_dim, const int64_t* st, float scale, void* stream);  // attn_fwd.hip

VGPT_EXPORT int vgpt_attn_plan_build(const uint32_t* bits, int64_t B, int64_t L, const int32_t* items, int64_t n_items,
                                     int item_rows, uint16_t* item_summary, int32_t* order, void* stream) {
    VGPT_REQUIRE(item_rows == 128 || item_rows == 256, VGPT_ERR_INVALID, "vgpt_attn_plan_build: item_rows must be 128 or 256");
    VGPT_REQUIRE(bits && items && item_summary && order, VGPT_ERR_INVALID, "vgpt_attn_plan_build: null pointer");
    VGPT_REQUIRE(B > 0 && L > 0 && L <= (1 << 22) && n_items > 0 && n_items < 65536, VGPT_ERR_INVALID,
                 "vgpt_attn_plan_build: bad shape");
    const int nkt = (int)cdiv(L, 64);
    hipLaunchKernelGGL(item_summary_kernel, dim3(nkt, (unsigned)n_items), dim3(256), 0, (hipStream_t)stream, bits, items,
                       item_summary, (int)L, (int)cdiv(L, 32), nkt);
    hipLaunchKernelGGL(item_order_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, items, item_summary, (int)n_items,
                       nkt, item_rows == 256 ? 32 : 0, order);
    VGPT_CHECK_LAUNCH("vgpt_attn_plan_build");
    return VGPT_OK;
}

VGPT_EXPORT int vgpt_attn_fwd_plan(const void* q, const void* k, const void* v, void* o, float* lse, const uint32_t* bits,
                                   const int32_t* items, const uint16_t* item_summary, const int32_t* order,
                                   int64_t n_items, int item_rows, const int32_t* split_items, int64_t n_split,
                                   float* split_ws, int64_t B, int64_t L, int n_heads, int n_kv_heads, int head_dim,
                                   int64_t q_sb, int64_t q_sh, int64_t q_ss, int64_t k_sb, int64_t k_sh, int64_t k_ss,
                                   int64_t v_sb, int64_t v_sh, int64_t v_ss, int64_t o_sb, int64_t o_sh, int64_t o_ss,
                                   float scale, void* stream) {
    VGPT_REQUIRE(q && k && v && o && bits && items && item_summary && order, VGPT_ERR_INVALID,
                 "vgpt_attn_fwd_plan: null pointer");
    if (item_rows == 128) {
        const int64_t st[12] = {q_sb, q_sh, q_ss, k_sb, k_sh, k_ss, v_sb, v_sh, v_ss, o_sb, o_sh, o_ss};
        return vgpt_attn_fwd_items128(q, k, v, o, lse, bits, items, item_summary, order, n_items, split_items, n_split,
                                      split_ws, B, L, n_heads, n_kv_heads, head_dim, st, scale, stream);
    }
    VGPT_REQUIRE(n_split == 0, VGPT_ERR_UNSUPPORTED, "vgpt_attn_fwd_plan: key-split items need item_rows 128");
    VGPT_REQUIRE(item_rows == 256, VGPT_ERR_INVALID, "vgpt_attn_fwd_plan: item_rows must be 128 or 256");
    VGPT_REQUIRE(head_dim == D, VGPT_ERR_UNSUPPORTED,
                 "vgpt_attn_fwd_plan: the 256-row kernel needs head_dim 96 (got %d)", head_dim);
    VGPT_REQUIRE(B > 0 && L > 0 && L <= (1 << 22) && n_items >= 0 && n_items < 65536 && n_heads > 0 && n_kv_heads > 0 &&
                     n_heads % n_kv_heads == 0,
                 VGPT_ERR_INVALID, "vgpt_attn_fwd_plan: bad shape");
    VGPT_REQUIRE(scale > 0.f, VGPT_ERR_INVALID, "vgpt_attn_fwd_plan: scale must be positive");
    const int64_t strides[] = {q_sb, q_sh, q_ss, k_sb, k_sh, k_ss, v_sb, v_sh, v_ss};
    for (int64_t st : strides)
        VGPT_REQUIRE(st % 8 == 0, VGPT_ERR_UNSUPPORTED, "vgpt_attn_fwd_plan: q/k/v strides must be multiples of 8 elements");
    VGPT_REQUIRE(o_sb % 4 == 0 && o_sh % 4 == 0 && o_ss % 4 == 0, VGPT_ERR_UNSUPPORTED,
                 "vgpt_attn_fwd_plan: o strides must be multiples of 4 elements");
    VGPT_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) == 0 && ((uintptr_t)o & 7) == 0, VGPT_ERR_UNSUPPORTED,
                 "vgpt_attn_fwd_plan: q/k/v must be 16-byte aligned");
    VGPT_REQUIRE(k_ss > 0 && v_ss > 0 && k_ss < (1 << 24) && v_ss < (1 << 24), VGPT_ERR_UNSUPPORTED,
                 "vgpt_attn_fwd_plan: key/value row strides must be in (0, 2^24) elements");
    if (n_items == 0) return VGPT_OK;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)attn_fwd_pp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           LDS_TOTAL);
        if (e != hipSuccess) {
            vgpt_set_error("vgpt_attn_fwd_plan: hipFuncSetAttribute: %s", hipGetErrorString(e));
            return VGPT_ERR_HIP;
        }
        attr_set = true;
    }
    PArgs a;
    a.q = (const bf16*)q; a.k = (const bf16*)k; a.v = (const bf16*)v; a.o = (bf16*)o; a.lse = lse;
    a.bits = bits; a.items = items; a.isum = item_summary; a.order = order;
    a.n_items = (int)n_items; a.L = (int)L; a.n_heads = n_heads; a.kv_group = n_heads / n_kv_heads;
    a.W = (int)cdiv(L, 32); a.nkt = (int)cdiv(L, 64);
    a.q_sb = q_sb; a.q_sh = q_sh; a.q_ss = q_ss; a.k_sb = k_sb; a.k_sh = k_sh; a.k_ss = k_ss;
    a.v_sb = v_sb; a.v_sh = v_sh; a.v_ss = v_ss; a.o_sb = o_sb; a.o_sh = o_sh; a.o_ss = o_ss;
    a.scale_log2e = scale * 1.4426950408889634f;
    const int64_t total = n_items * n_heads;
    a.trace = total <= g_pp_trace_cap ? g_pp_trace : nullptr;
    hipLaunchKernelGGL(attn_fwd_pp_kernel, dim3((unsigned)total), dim3(512), LDS_TOTAL, (hipStream_t)stream, a);
    VGPT_CHECK_LAUNCH("vgpt_attn_fwd_plan");
    return VGPT_OK;
}
