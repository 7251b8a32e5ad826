// bf16 NT GEMM on MFMA for the transformer's Linear layers (qkv_proj, o_proj, gate_up_proj,
// down_proj): C[M,N] = A[M,K] * W[N,K]^T, fp32 accumulate.
//
// Reference arithmetic: nn.Linear calls at LVM/transform/sdpa_transform.py:39,89 and
// Phi3MLP.forward (transformers==4.47.1): down_proj(up * act(gate)), [gate|up] = gate_up_proj(x).
//
// Structure (gfx950):
//   - 256(m) x 256(n) x 64(k) tile, 512 threads = 8 waves in a 2(m) x 4(n) grid, wave tile 128 x 64 = 8 x 4 MFMA
//     16x16x32 sub-tiles (128 accumulator registers), one workgroup per CU; a 128 x 128 / 4-wave configuration with a
//     plain double-buffered loop serves small grids and the row remainder of a launch whose last round of 256-tiles
//     would be badly filled (launch<>()).
//   - A and W tiles go L2 -> LDS with global_load_lds_dwordx4 issued from inline asm (no VGPR round trip, no
//     compiler-inserted drain), double-buffered; the 256-tile loop is software-pipelined in 4 phases of 16 MFMAs with
//     the fragments of phase p+1 read under phase p's MFMAs, one barrier per k-tile, next tile's DMA in two halves.
//   - LDS rows are 128 B (64 bf16); the 16-byte chunk index is XOR-swizzled with (row & 7) on the SOURCE address
//     (the LDS-DMA destination is lane-linear), and the same XOR is applied on the ds_read_b128 side: conflict-free
//     for the 16-lane b128 groups.
//   - operands whose reduction index is their ROW index (the backward's dX = dY W, dW = dY^T X) are staged in their
//     natural layout and read with ds_read_b64_tr_b16 (template flags ATR / WTR below).
//   - operands are swapped (W is the MFMA "A" operand) so each lane ends up with 4 consecutive n of one m: 8-byte
//     bf16 stores; fused residual / bias / act(gate)*up epilogues.
//   - 1-D grid with an XCD-aware, grouped tile order so blocks that share an XCD's L2 work on neighbouring tiles.
#include <type_traits>

#include "common.h"

#ifndef VGPT_GEMM_SETPRIO
#define VGPT_GEMM_SETPRIO 0
#endif
// Diagnostics build (make gemm-debug-N, results are garbage): 1 = skip the LDS-DMA staging, 2 = skip the LDS fragment
// reads, 3 = both, 4 = skip the epilogue, 16 = no wait for the LDS-DMA (what the per-tile drain costs).  A COMPILE-time switch: as a run-time flag the skipped reads became conditional, and at the join
// hipcc's s_waitcnt insertion assumes the shorter path — every MFMA phase then waited for the fragment reads issued
// right in front of it (lgkmcnt(3..0) instead of (7..4)), exposing the LDS latency twice per k-tile.
#ifndef VGPT_GEMM_DEBUG_BUILD
#define VGPT_GEMM_DEBUG_BUILD 0
#endif

// EXPERIMENT switch (make gemm-variant-VGPT_GEMM_STORE_WT): the epilogue's 8-byte output stores as write-through
// (`sc0 sc1`: the line is not kept dirty in the XCD's L2), to see what the write-back of a GEMM's dirty output lines costs
// at the kernel boundary behind it (MI355X_MICROARCH.md, price list row `boundary`: + B / 6 TB/s for B dirty bytes).
// Measured in round 3: 35.3 ms per sampler step against 32.0 (o_proj 97 vs 72 us, qkv 242 vs 205): the consumer kernel
// finds its input in neither L2 nor -- apparently -- as readily in the Infinity Cache; the plain stores stay.  VAL=2 (`nt`,
// non-temporal) is worse still: gate_up 366 vs 326 us, qkv + RoPE 303 vs 201 us, 35.2 vs 30.8 ms per step -- the epilogue's
// stores are 8 bytes per lane and it is the write-back L2 that merges them into whole lines.
// Also measured and removed (round 3): software prefetch into the XCD's L2 -- wave 0 / 1 of every workgroup touching one dword
// per 128-byte line of a 1/4 (A) / 1/8 (W) slice of the k-pieces three k-tiles ahead, so that the first of the 4 / 8
// workgroups of an XCD that want a piece no longer misses L2 (12 of 64 piece fetches per k-tile do: the 16-19 % beyond-L2
// fills of the counters).  Same box: 34.0 ms per sampler step against 33.35 / 33.43, 8192^3 1254 vs 1306 TFLOP/s: the extra
// vector-memory instructions cost the loop more than the Infinity-Cache-served fills do.
#ifndef VGPT_GEMM_STORE_WT
#define VGPT_GEMM_STORE_WT 0
#endif

namespace {

__device__ __forceinline__ void store_out4(bf16* p, bf16x4 v) {
#if VGPT_GEMM_STORE_WT == 1
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    const u32x2_t d = __builtin_bit_cast(u32x2_t, v);
    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(d) : "memory");
#elif VGPT_GEMM_STORE_WT == 2   // non-temporal: the output streams through the L2 instead of displacing the operand panels
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    const u32x2_t d = __builtin_bit_cast(u32x2_t, v);
    asm volatile("global_store_dwordx2 %0, %1, off nt" ::"v"(p), "v"(d) : "memory");
#else
    *reinterpret_cast<bf16x4*>(p) = v;
#endif
}

constexpr int BK = 64;
constexpr int kDebug = VGPT_GEMM_DEBUG_BUILD;

// Tile configuration: BM x BN block tile, WM x WN waves, every wave owns (BM/WM) x (BN/WN).
//   Cfg128: 128x128, 2x2 waves of 64x64   (64 KiB LDS, 2 blocks/CU)  — small problems
//   Cfg256: 256x256, 2x4 waves of 128x64  (128 KiB LDS, 1 block/CU)  — halves the L2->LDS bytes per FLOP
template <int BM_, int BN_, int WM_, int WN_>
struct TileCfg {
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_;
    static constexpr int NWAVES = WM * WN, THREADS = NWAVES * 64;
    static constexpr int MI = BM / WM / 16, NI = BN / WN / 16;  // 16x16 sub-tiles per wave
    static constexpr int A_BYTES = BM * BK * 2, W_BYTES = BN * BK * 2;
    static constexpr int LDS_BYTES = 2 * (A_BYTES + W_BYTES);
    // 8-row slabs (1-KiB LDS-DMA pieces) per wave.  Where the W slabs do not divide over the waves (256 x 288) they are
    // dealt round-robin and the last round is issued by the first waves only.
    static constexpr int W_TOTAL = BN / 8;
    static constexpr bool W_EVEN = W_TOTAL % NWAVES == 0;
    static constexpr int A_SLABS = BM / 8 / NWAVES, W_SLABS = (W_TOTAL + NWAVES - 1) / NWAVES;
    static_assert(BM % (8 * NWAVES) == 0 && BN % 8 == 0, "slabs must divide over waves");
    static __device__ __host__ constexpr int w_slab(int wave, int i) { return W_EVEN ? wave * W_SLABS + i : i * NWAVES + wave; }
};
using Cfg128 = TileCfg<128, 128, 2, 2>;
using Cfg256 = TileCfg<256, 256, 2, 4>;
// 256 x 192: same loop with 3 instead of 4 n sub-tiles per wave (wave tile 128 x 48).  For problems whose 256-tile
// grid fills the last round badly: M = 4096 rows x N = 3072 is 192 tiles of 256 x 256 (a 75 % round) but exactly 256
// tiles of 256 x 192; x N = 9216 it is 2.25 rounds against 3 full rounds of 0.75-size tiles.
using Cfg192 = TileCfg<256, 192, 2, 4>;
// 256 x 288, 4 x 2 waves of 64 x 144 (MI = 4, NI = 9), six-phase loop (PIPE == 5): N = 9216 (qkv_proj of the Phi-3-mini-class
// denoiser) is 32 such tiles, so M = 4096 rows make exactly two rounds of 256 workgroups where 256-wide tiles make 2.25 and
// 192-wide ones three (and a 192-wide tile's k-step takes as long as a 256-wide one's: the loop is not bound by its MFMAs)
using Cfg288 = TileCfg<256, 288, 4, 2>;
// 256 x 256 with FOUR waves of 128 x 128 (MI = NI = 8: 256 accumulator registers, one wave per SIMD, 512-register budget)
// and the register-staged, fragment-streaming loop PIPE == 6: every fragment feeds 8 MFMAs (a third fewer LDS bytes per
// FLOP than the 128 x 64 wave tile) and no instruction of the loop is an LDS-DMA.  EXPERIMENT: VGPT_GEMM_TILE=512.
using Cfg256w4 = TileCfg<256, 256, 2, 2>;

enum { MODE_PLAIN = 0, MODE_GATED = 1, MODE_ROPE = 2 };

struct GemmArgs {
    const bf16* A;
    const bf16* W;
    bf16* C;
    const bf16* extra;
    int M, N, K;
    int64_t lda, ldw, ldc, ldr;
    int epi;   // VGPT_EPI_*
    int act;   // gated mode
    int I;     // gated mode: intermediate size
    int tiles_m, tiles_n;
    // MODE_ROPE (qkv_proj + apply_rotary_pos_emb): columns [0, rope_cols) are heads of head_dim columns rotated with the
    // per-token tables cos / sin (M, head_dim / 2) fp32; the remaining columns (V) are plain
    const float* rope_cos;
    const float* rope_sin;
    int rope_cols, head_dim;
    // MODE_GATED, training forward (vgpt_gated_mlp_act_fwd_keep): also store [gate | up] (M, 2I) rounded to bf16 -- what
    // gate_up_proj returns and the backward reads -- and compute act(gate) * up FROM the rounded values, bit for bit what
    // the plain GEMM followed by vgpt_silu_mul_fwd produces; null: inference (activation from the fp32 accumulators)
    bf16* gu_out = nullptr;
    int64_t ld_gu = 0;
    uint32_t* dbg = nullptr;   // diagnostics builds of the four-wave kernel only
    // RMSNorm folded into the GEMMs around it (vgpt_gemm_bf16_resid_rstd -> vgpt_gemm_bf16_rope_prenorm /
    // vgpt_gated_mlp_act_fwd_prenorm):
    //   producer (MODE_PLAIN + residual, four-wave kernel): every workgroup stores, per output row, the sum over ITS columns of the
    //   squares of the bf16-rounded outputs (ssq_out[p * M + m], p = tile column * 2 + wave column); the last workgroup to arrive
    //   at a 256-row block's counter adds the block's partials up in index order (deterministic: no float atomics) and writes
    //   rstd_out[m] = rsqrt(sum * nrm_inv_h + nrm_eps);
    //   consumer (MODE_ROPE / MODE_GATED): the accumulators of row m are multiplied by nrm_rstd[m] before anything else -- the
    //   norm's gain is folded into W by the caller (vgpt_fold_norm_gain)
    float* ssq_out = nullptr;          // producer: partial sums (n_partials, M), inside the caller's workspace
    int* ssq_cnt = nullptr;            // producer: one arrival counter per 256-row block (zero between launches)
    float* rstd_out = nullptr;         // producer: 1 / rms of every output row, written by the last workgroup of its row block
    const float* nrm_rstd = nullptr;   // consumer: 1 / rms of every row of A
    int64_t nrm_ld = 0;                // producer: rows of the whole matrix = stride between two partials
    float nrm_eps = 0.f, nrm_inv_h = 0.f;
};


__device__ __forceinline__ void glds16(const bf16* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0,
                                     0);
}

// LDS-DMA issued from inline asm (kernels with transposed operands): hipcc cannot tell that a ds_read_b64_tr_b16
// does not alias an LDS-DMA in flight and would drain vmcnt(0) before every such read; hidden from it, completion is
// tracked by the kernel's own s_waitcnt vmcnt(0) in front of its barrier.
typedef int v4i32 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void glds16_asm(const char* base, uint32_t voffset, uint32_t lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voffset), "s"(base), "s"(lds_dst)
        : "memory");
}
// 16 bytes per lane through a buffer descriptor covering [base, base + num_bytes): lanes whose offset falls outside
// read zeros (the zero fill of a partial reduction tile)
__device__ __forceinline__ v4i32 make_rsrc(const void* base, int num_bytes) {
    const uint64_t b = (uint64_t)(uintptr_t)base;
    v4i32 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)b);
    r[1] = __builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32) & 0xffff);
    r[2] = __builtin_amdgcn_readfirstlane(num_bytes);
    r[3] = 0x00020000;
    return r;
}
__device__ __forceinline__ void buf_glds16_asm(v4i32 rsrc, uint32_t voffset, int soffset, uint32_t lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voffset), "s"(rsrc), "s"(soffset), "s"(lds_dst)
        : "memory");
}

// weight row feeding n-slot s of a tile (gated mode interleaves gate/up every 16 slots)
template <int MODE>
__device__ __forceinline__ int w_row_of_slot(int n0, int s, int I) {
    if (MODE == MODE_GATED) {
        // n0 = first output column of this tile (tile covers 64 output columns)
        return ((s >> 4) & 1 ? I : 0) + n0 + (s >> 5) * 16 + (s & 15);
    }
    return n0 + s;
}

// MODE_ROPE: output column behind n-slot `gs` (global slot index).  Inside the rotated region every 16-slot MFMA
// sub-tile holds 8 (d, d + head_dim/2) pairs of one head -- slots 0..7 the lower-half columns, 8..15 their partners --
// so the partner of the 4 columns a lane holds lives in lane ^ 32 (one v_permlane32_swap in the epilogue).
__device__ __forceinline__ int rope_col_of_slot(int gs, int rope_cols, int head_dim) {
    if (gs >= rope_cols) return gs;
    const int u = gs >> 4, w = gs & 15, per = head_dim >> 4;
    return (u / per) * head_dim + (u % per) * 8 + (w & 7) + ((w & 8) ? (head_dim >> 1) : 0);
}

constexpr bool getenv_prio = VGPT_GEMM_SETPRIO;

// ATR / WTR: the operand is stored with the reduction index as its ROW index (A as [K][M], W as [K][N]) -- the dX and
// dW products of the backward (dX = dY W, dW = dY^T X) without materialising a transpose.  Such a tile is staged in
// its natural [64 reduction rows][256 columns] layout and its MFMA fragments come from ds_read_b64_tr_b16; the
// 32-byte units of a row are XOR-swizzled with ((row>>3)&1)<<2 | (row&3) (on the DMA source address and on the
// read) so that the 8 rows x 32 B a half-wave reads transposed hit 64 different banks.
template <int MODE, typename C, int PIPE, bool ATR = false, bool WTR = false>
__global__ __launch_bounds__(C::THREADS, 2) void gemm_bf16_kernel(GemmArgs g) {
    static_assert(!(ATR || WTR) || MODE == MODE_PLAIN, "transposed operands: plain kernel only");
    constexpr bool ROPE = MODE == MODE_ROPE;
    constexpr int BM = C::BM, BN = C::BN, MI = C::MI, NI = C::NI;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- tile order: XCD-aware remap (bijective), then grouped along m.  A workgroup walks the virtual tile ids
    //      blockIdx.x, blockIdx.x + gridDim.x, ... (persistent launch: gridDim.x = one round of the chip; a plain launch
    //      has gridDim.x = number of tiles and the walk ends after one).  gridDim.x % 8 == 0 or a single round, so a
    //      workgroup's tiles keep `vt & 7`, the XCD the remap assumes. ----
    const int nwg = g.tiles_m * g.tiles_n;
    // ---- staging addresses: wave w stages its share of 8-row slabs of both tiles ----
    const int srow = lane >> 3;            // row inside the 8-row slab
    const int schunk = (lane & 7) ^ srow;  // source 16-B chunk (XOR swizzle)
    // per-lane BYTE offsets of the pieces this wave stages, relative to the (wave-uniform) tile origin: one
    // 32-bit register per piece instead of a 64-bit pointer, and the k-tile advance stays in scalar registers
    uint32_t a_off[C::A_SLABS], w_off[C::W_SLABS];
    const int n_rows_w = (MODE == MODE_GATED) ? 2 * g.I : g.N;
    // transposed operand (tile = 64 reduction rows x BM or BN columns): a 1-KiB piece covers 2 (256 columns) or 4
    // (128 columns) consecutive rows; lane -> (row, 16-B chunk)
    auto tr_off = [&](int64_t ld, int piece, int col0, int width, int cols) {
        const int cpr = cols / 8;  // 16-B chunks per row
        const int row = piece * (64 / cpr) + lane / cpr, t_c = lane % cpr;
        const int key = (((row >> 3) & 1) << 2) | (row & 3);
        const int lchunk = ((((t_c >> 1) ^ key) << 1) | (t_c & 1));
        const int col = min(col0 + lchunk * 8, width - 8) - col0;
        return (uint32_t)(row * (int)ld + col) * 2u;
    };
    // A transposed operand is fetched with buffer_load ... lds through a descriptor that ends after reduction row
    // K-1: the rows of a partial last k-tile are out of range and the hardware returns zeros for them.
    v4i32 a_rs = {0, 0, 0, 0}, w_rs = {0, 0, 0, 0};
    int m0 = 0, n0 = 0;
    const char* a_org = nullptr;
    const char* w_org = nullptr;
    auto set_tile = [&](int vt, bool remap = true) {
        int bid = vt;
        if (remap) {
            const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
            bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        }
        constexpr int GROUP = 8;
        const int in_group = GROUP * g.tiles_n;
        const int group_id = bid / in_group;
        const int first_m = group_id * GROUP;
        const int gsz = min(g.tiles_m - first_m, GROUP);
        const int tm = first_m + (bid % in_group) % gsz;
        const int tn = (bid % in_group) / gsz;
        m0 = tm * BM;
        n0 = tn * (MODE == MODE_GATED ? BN / 2 : BN);
        // diagnostics bit 64 (results are garbage): every workgroup reads the operands of tile (0, 0) -- an L2-resident operand
        // stream under the product's instruction stream, to tell memory latency under load from issue / clock limits
        const int m0l = (kDebug & 64) ? 0 : m0, n0l = (kDebug & 64) ? 0 : n0;
        a_org = reinterpret_cast<const char*>(ATR ? g.A + m0l : g.A + (int64_t)m0l * g.lda);
        if constexpr (WTR) w_org = reinterpret_cast<const char*>(g.W + n0l);
        else if constexpr (MODE == MODE_GATED || ROPE) w_org = reinterpret_cast<const char*>(g.W);
        else w_org = reinterpret_cast<const char*>(g.W + (int64_t)n0l * g.ldw);
#pragma unroll
        for (int i = 0; i < C::A_SLABS; ++i) {
            if constexpr (ATR) {
                a_off[i] = tr_off(g.lda, wave * C::A_SLABS + i, m0, g.M, BM);
            } else {
                const int r = min((wave * C::A_SLABS + i) * 8 + srow, g.M - 1 - m0);
                a_off[i] = (uint32_t)(r * (int)g.lda + schunk * 8) * 2u;
            }
        }
#pragma unroll
        for (int i = 0; i < C::W_SLABS; ++i) {
            if constexpr (WTR) {
                w_off[i] = tr_off(g.ldw, wave * C::W_SLABS + i, n0, g.N, BN);
            } else {
                const int r = C::w_slab(wave, i) * 8 + srow;
                int wr;
                if constexpr (ROPE) wr = min(rope_col_of_slot(n0 + r, g.rope_cols, g.head_dim), n_rows_w - 1);
                else wr = min(w_row_of_slot<MODE>(n0, r, g.I), n_rows_w - 1) - (MODE == MODE_GATED ? 0 : n0);
                w_off[i] = (uint32_t)(wr * (int)g.ldw + schunk * 8) * 2u;
            }
        }
        if constexpr (ATR) a_rs = make_rsrc(a_org, (int)(((int64_t)g.K * g.lda - m0) * 2));
        if constexpr (WTR) w_rs = make_rsrc(w_org, (int)(((int64_t)g.K * g.ldw - n0) * 2));
    };
    int vt_cur = blockIdx.x;
    set_tile(vt_cur);
    const int64_t a_step = (ATR ? (int64_t)BK * g.lda : BK) * 2, w_step = (WTR ? (int64_t)BK * g.ldw : BK) * 2;
    const uint32_t lds_base =
        __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
    auto a_issue = [&](int i, int kt, char* dst) {
        if constexpr (ATR)
            buf_glds16_asm(a_rs, a_off[i], (int)(kt * a_step), lds_base + (uint32_t)(dst - smem));
        else
            glds16_asm(a_org + kt * a_step, a_off[i], lds_base + (uint32_t)(dst - smem));
    };
    auto w_issue = [&](int i, int kt, char* dst) {
        if constexpr (WTR)
            buf_glds16_asm(w_rs, w_off[i], (int)(kt * w_step), lds_base + (uint32_t)(dst - smem));
        else
            glds16_asm(w_org + kt * w_step, w_off[i], lds_base + (uint32_t)(dst - smem));
    };
    char* sA = smem;                    // [2][A_BYTES]
    char* sW = smem + 2 * C::A_BYTES;   // [2][W_BYTES]

    auto stage = [&](int buf, int kt) {
        if constexpr (kDebug & 1) return;
#pragma unroll
        for (int i = 0; i < C::A_SLABS; ++i)
            a_issue(i, kt, sA + buf * C::A_BYTES + (wave * C::A_SLABS + i) * 1024);
#pragma unroll
        for (int i = 0; i < C::W_SLABS; ++i)
            if (C::W_EVEN || C::w_slab(wave, i) < C::W_TOTAL)
                w_issue(i, kt, sW + buf * C::W_BYTES + C::w_slab(wave, i) * 1024);
    };

    // ---- fragment read addresses ----
    const int wn = wave % C::WN, wm = wave / C::WN;
    const int frow = lane & 15;  // row inside a 16-row sub-tile
    const int fk = lane >> 4;    // 16-B chunk inside a 32-wide k-step
    // byte offset of (row, chunk g) = row*128 + ((g ^ (row&7)) * 16); row&7 == frow&7 here
    const int w_base = (wn * (NI * 16) + frow) * 128;
    const int a_base = (wm * (MI * 16) + frow) * 128;
    const int sw = frow & 7;

    f32x4 acc[NI][MI];
    f32x16 acc2[2][4];   // diagnostics flag 32 only
    // folded RMSNorm (consumer side): 1 / rms of this tile's rows, in LDS behind the staging buffers
    float* rs_lds = reinterpret_cast<float*>(smem + C::LDS_BYTES);
    auto load_rstd = [&](int m_first) {
        if constexpr (MODE != MODE_PLAIN) {
            if (g.nrm_rstd != nullptr && tid < BM) rs_lds[tid] = m_first + tid < g.M ? g.nrm_rstd[m_first + tid] : 0.f;
        }
    };
    if constexpr ((kDebug & 32) != 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc2[i][j][e] = 0.f;
    }

    // transposed image: lane (g4 = lane>>4, li = lane&15) of the 16-column sub-tile `unit` reads rows
    // 32 ks + 8 g4 + (li>>2) (+4 for the upper half) at columns 4 (li&3) .. +4 -> k = 8 g4 + j of column li
    const int tr_li = lane & 15, tr_g = lane >> 4;
    const int tr_key = ((tr_g & 1) << 2) | (tr_li >> 2);
    auto ld_tr = [&](const char* tile, int rowb, int ks, int unit) {
        const char* p = tile + (ks * 32 + 8 * tr_g + (tr_li >> 2)) * rowb + 8 * (tr_li & 3) + ((unit ^ tr_key) << 5);
        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(p));
        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(p + 4 * rowb));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };

    const int nk = (g.K + BK - 1) / BK;  // a partial last k-tile exists only with transposed operands (zero rows)
    // Persistent walk (PERSIST): when this workgroup has another tile, that tile's first k-tile is requested (LDS-DMA into
    // staging buffer 0, free once every wave is past the last barrier of the k-loop) BEFORE the epilogue of the current one,
    // so the fetch latency of a tile's prologue and the HBM write time of its predecessor's epilogue overlap instead of
    // adding up (one workgroup per CU: nothing else overlaps them).  Not for MODE_ROPE, whose epilogue stages the cos / sin
    // rows through the same LDS, nor for the experimental loops.
    constexpr bool PERSIST = (PIPE == 0 || PIPE == 1) && !ROPE;
    bool prefetched = false;
    const int kbeg = 0, kend = nk;
    for (;;) {
    load_rstd(m0);   // read in the epilogue, behind the k-loop's barriers
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (PIPE == 0) {
        // one barrier per k-tile: wait for tile kt, issue tile kt+1's DMA, compute tile kt
        if (!prefetched) stage(0, 0);
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();  // tile kt landed for every wave; buffer buf^1 is free
            if (kt + 1 < nk) stage(buf ^ 1, kt + 1);
            const char* bA = sA + buf * C::A_BYTES + a_base;
            const char* bW = sW + buf * C::W_BYTES + w_base;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int coff = ((ks * 4 + fk) ^ sw) * 16;
                bf16x8 wf[NI], af[MI];
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    if constexpr (WTR) wf[i] = ld_tr(sW + buf * C::W_BYTES, 2 * BN, ks, wn * NI + i);
                    else wf[i] = *reinterpret_cast<const bf16x8*>(bW + i * 2048 + coff);
                }
#pragma unroll
                for (int j = 0; j < MI; ++j) {
                    if constexpr (ATR) af[j] = ld_tr(sA + buf * C::A_BYTES, 2 * BM, ks, wm * MI + j);
                    else af[j] = *reinterpret_cast<const bf16x8*>(bA + j * 2048 + coff);
                }
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < MI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
            }
        }
    } else if constexpr (PIPE == 5) {
        // 4x2-wave tiles of 64 x 144 (256 x 288): six phases of 12 MFMAs per k-tile -- (ks, third of the wave's nine n
        // sub-tiles) -- with the W fragments of phase p+1 (three reads) and the A fragments of the other k-step (four) in
        // flight under phase p's MFMAs, the per-tile barrier in front of the LAST phase and the next tile's first fragments
        // read behind it, the LDS-DMA in two halves as in the 4-phase loop.  200 accumulator + fragment registers.
        static_assert(MI == 4 && NI == 9 && !ATR && !WTR, "six-phase loop: 4x2-wave tiles of 64 x 144, NT operands");
        bf16x8 Wt[2][3], Af2[2][4];
        auto ldW3 = [&](bf16x8(&dst)[3], int buf, int ks, int th) {
            const char* b = sW + buf * C::W_BYTES + w_base + th * (3 * 2048) + ((ks * 4 + fk) ^ sw) * 16;
#pragma unroll
            for (int i = 0; i < 3; ++i) dst[i] = *reinterpret_cast<const bf16x8*>(b + i * 2048);
        };
        auto ldA4 = [&](bf16x8(&dst)[4], int buf, int ks) {
            const char* b = sA + buf * C::A_BYTES + a_base + ((ks * 4 + fk) ^ sw) * 16;
#pragma unroll
            for (int j = 0; j < 4; ++j) dst[j] = *reinterpret_cast<const bf16x8*>(b + j * 2048);
        };
        auto mma3 = [&](const bf16x8(&wf)[3], const bf16x8(&af)[4], auto th) {
            constexpr int TH = decltype(th)::value;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[TH * 3 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[TH * 3 + i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        auto stage_half = [&](int buf, int kt, int half) {
#pragma unroll
            for (int i = 0; i < C::A_SLABS / 2; ++i) {
                const int ii = half * (C::A_SLABS / 2) + i;
                a_issue(ii, kt, sA + buf * C::A_BYTES + (wave * C::A_SLABS + ii) * 1024);
            }
#pragma unroll
            for (int ii = 0; ii < C::W_SLABS; ++ii)
                if ((ii >= C::W_SLABS / 2) == (half != 0) && (C::W_EVEN || C::w_slab(wave, ii) < C::W_TOTAL))
                    w_issue(ii, kt, sW + buf * C::W_BYTES + C::w_slab(wave, ii) * 1024);
        };
        using T0 = std::integral_constant<int, 0>;
        using T1 = std::integral_constant<int, 1>;
        using T2 = std::integral_constant<int, 2>;
        stage(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (nk > 1) stage(1, 1);
        ldW3(Wt[0], 0, 0, 0);
        ldA4(Af2[0], 0, 0);
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            ldW3(Wt[1], buf, 0, 1);
            if (kt >= 1 && kt + 1 < nk) stage_half(buf ^ 1, kt + 1, 1);
            mma3(Wt[0], Af2[0], T0{});
            ldW3(Wt[0], buf, 0, 2);
            ldA4(Af2[1], buf, 1);
            mma3(Wt[1], Af2[0], T1{});
            ldW3(Wt[1], buf, 1, 0);
            mma3(Wt[0], Af2[0], T2{});
            ldW3(Wt[0], buf, 1, 1);
            mma3(Wt[1], Af2[1], T0{});
            ldW3(Wt[1], buf, 1, 2);
            mma3(Wt[0], Af2[1], T1{});
            // every wave holds its last fragments of this tile and its share of tile kt+1 has landed
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (kt + 2 < nk) stage_half(buf, kt + 2, 0);
            if (kt + 1 < nk) {
                ldW3(Wt[0], buf ^ 1, 0, 0);
                ldA4(Af2[0], buf ^ 1, 0);
            }
            mma3(Wt[1], Af2[1], T2{});
        }
    } else {
        // Software-pipelined 4-phase loop (256x256 tile, wave tile 128(m) x 64(n)).  A k-tile is four
        // phases of 16 MFMAs: (ks0,m-half0) (ks0,m-half1) (ks1,m-half0) (ks1,m-half1).  The fragments
        // of phase p+1 are read from LDS while phase p's MFMAs run (two register sets), the next tile's
        // DMA is issued in two halves right after the per-tile barrier, and that barrier sits in front of
        // the LAST phase of a tile so the first fragments of tile kt+1 are prefetched under tile kt.
        static_assert(MI == 8 && (NI == 4 || NI == 3), "pipelined loop is written for the 2x4-wave tiles of 128 x 64 / 128 x 48");
        bf16x8 Wf[2][NI], Af[2][4];
        if constexpr ((kDebug & 2) != 0) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    Af[u][i] = bf16x8{(bf16)1.f, (bf16)0.5f, (bf16)-1.f, (bf16)2.f, (bf16)1.f, (bf16)0.5f, (bf16)-1.f, (bf16)2.f};
#pragma unroll
                for (int i = 0; i < NI; ++i) Wf[u][i] = Af[u][0];
            }
        }
        auto ldW = [&](bf16x8(&dst)[NI], int buf, int ks) {
            if constexpr (kDebug & 2) return;
            if constexpr (WTR) {
#pragma unroll
                for (int i = 0; i < NI; ++i) dst[i] = ld_tr(sW + buf * C::W_BYTES, 2 * BN, ks, wn * NI + i);
            } else {
                const char* b = sW + buf * C::W_BYTES + w_base + ((ks * 4 + fk) ^ sw) * 16;
#pragma unroll
                for (int i = 0; i < NI; ++i) dst[i] = *reinterpret_cast<const bf16x8*>(b + i * 2048);
            }
        };
        auto ldA = [&](bf16x8(&dst)[4], int buf, int ks, int mh) {
            if constexpr (kDebug & 2) return;
            if constexpr (ATR) {
#pragma unroll
                for (int j = 0; j < 4; ++j) dst[j] = ld_tr(sA + buf * C::A_BYTES, 2 * BM, ks, wm * 8 + mh * 4 + j);
            } else {
                const char* b = sA + buf * C::A_BYTES + a_base + mh * 8192 + ((ks * 4 + fk) ^ sw) * 16;
#pragma unroll
                for (int j = 0; j < 4; ++j) dst[j] = *reinterpret_cast<const bf16x8*>(b + j * 2048);
            }
        };
        // diagnostics flag 32 (with 4: results are garbage): the same fragments fed to 32x32x16 MFMAs, half as many for
        // the same FLOPs and matrix-pipe time -- each holds the SIMD's vector issue for 8 of its 32 cycles, a 16x16x32
        // for 8 of its 16 (MI355X_MICROARCH.md): what the loop would gain from the issue slots alone
        auto mma = [&](const bf16x8(&wf)[NI], const bf16x8(&af)[4], auto mh) {
            constexpr int MH = decltype(mh)::value;
            if constexpr ((kDebug & 32) != 0 && NI == 4) {
#pragma unroll
                for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
                    for (int j2 = 0; j2 < 2; ++j2) {
                        acc2[i2][MH * 2 + j2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[2 * i2], af[2 * j2], acc2[i2][MH * 2 + j2], 0, 0, 0);
                        acc2[i2][MH * 2 + j2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[2 * i2 + 1], af[2 * j2 + 1], acc2[i2][MH * 2 + j2], 0, 0, 0);
                    }
                return;
            }
            if (getenv_prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][MH * 4 + j] =
                        __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][MH * 4 + j], 0, 0, 0);
            if (getenv_prio) __builtin_amdgcn_s_setprio(0);
        };
        auto stage_half = [&](int buf, int kt, int half) {
            if constexpr (kDebug & 1) return;
#pragma unroll
            for (int i = 0; i < C::A_SLABS / 2; ++i) {
                const int ii = half * (C::A_SLABS / 2) + i;
                a_issue(ii, kt, sA + buf * C::A_BYTES + (wave * C::A_SLABS + ii) * 1024);
            }
#pragma unroll
            for (int ii = 0; ii < C::W_SLABS; ++ii)   // first half: slabs [0, W_SLABS/2), second: the rest
                if ((ii >= C::W_SLABS / 2) == (half != 0))
                    w_issue(ii, kt, sW + buf * C::W_BYTES + (wave * C::W_SLABS + ii) * 1024);
        };
        using H0 = std::integral_constant<int, 0>;
        using H1 = std::integral_constant<int, 1>;

        if (!prefetched) stage(0, kbeg);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kend - kbeg > 1) stage(1, kbeg + 1);
        ldW(Wf[0], 0, 0);
        ldA(Af[0], 0, 0, 0);
        for (int kt = kbeg; kt < kend; ++kt) {
            const int buf = (kt - kbeg) & 1;
            // phase 1: (ks0, m-half 0)
            ldA(Af[1], buf, 0, 1);
            if (kt >= kbeg + 1 && kt + 1 < kend) stage_half(buf ^ 1, kt + 1, 1);
            mma(Wf[0], Af[0], H0{});
            __builtin_amdgcn_sched_barrier(0);
            // phase 2: (ks0, m-half 1)
            ldW(Wf[1], buf, 1);
            ldA(Af[0], buf, 1, 0);
            mma(Wf[0], Af[1], H1{});
            __builtin_amdgcn_sched_barrier(0);
            // phase 3: (ks1, m-half 0)
            ldA(Af[1], buf, 1, 1);
            mma(Wf[1], Af[0], H0{});
            __builtin_amdgcn_sched_barrier(0);
            // phase 4: (ks1, m-half 1) — every wave has its last fragments of this tile in registers and
            // its share of tile kt+1 has landed: after the barrier buffer `buf` is free for tile kt+2
            if constexpr ((kDebug & 16) == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // 16: timing without the drain
            __syncthreads();
            if (kt + 2 < kend) stage_half(buf, kt + 2, 0);
            if (kt + 1 < kend) {
                ldW(Wf[0], buf ^ 1, 0);
                ldA(Af[0], buf ^ 1, 0, 0);
            }
            mma(Wf[1], Af[1], H1{});
            __builtin_amdgcn_sched_barrier(0);
        }
    }


    // ---- the tile whose accumulators are stored now; then (persistent walk) the next tile's first k-tile is requested ----
    const int m0e = m0, n0e = n0;
    bool more = false;
    if constexpr (PERSIST) {
        const int vt_next = vt_cur + (int)gridDim.x;
        more = vt_next < nwg;
        if (more) {
            __syncthreads();          // every wave has read its last fragments: both staging buffers are free
            vt_cur = vt_next;
            set_tile(vt_next);
            stage(0, 0);
            prefetched = true;
        }
    }
    if constexpr ((kDebug & 4) != 0) {   // diagnostics: no epilogue (one store keeps the accumulators alive)
        float sum = 0.f;
        if constexpr ((kDebug & 32) != 0 && PIPE == 1) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) sum += acc2[i][j][e];
        }
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < MI; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (sum == 12345.678f) g.C[0] = f2bf(sum);
        return;
    }
    // ---- epilogue: lane holds m = lane&15, n = (lane>>4)*4 + reg of each 16x16 sub-tile ----
    const int em = lane & 15, en = (lane >> 4) * 4;
    if constexpr (ROPE) {
        // Linear output rounded to bf16 (what the reference's qkv_proj returns), then q*cos + rotate_half(q)*sin in
        // fp32 and one more rounding -- the arithmetic of vgpt_rope_qk_inplace on the stored tensor, without the store
        // and reload.  N % 16 == 0 (checked on the host), so lane and lane ^ 32 are in range together.
        // The cos / sin rows of this tile's BM tokens are first copied into the (now free) staging buffers with one
        // coalesced LDS-DMA burst: read per lane from global memory they were 2 x MI x NI latency-bound 16-byte loads
        // touching 16 cache lines each.
        const int half = g.head_dim >> 1;
        const int tab_bytes = (BM * half * 4 + 1023) & ~1023;          // one table's rows of this tile, whole 1-KiB pieces
        const bool staged = 2 * tab_bytes <= C::LDS_BYTES;
        if (staged) {
            __syncthreads();                                           // every wave has read its last fragments
            const int64_t row0_bytes = (int64_t)m0e * half * 4;
            // last readable 16 bytes of the table, relative to this tile's first row (rows past M are never used)
            const uint32_t last = (uint32_t)min((int64_t)g.M * half * 4 - row0_bytes - 16, (int64_t)tab_bytes);
            const char* cbase = reinterpret_cast<const char*>(g.rope_cos) + row0_bytes;
            const char* sbase = reinterpret_cast<const char*>(g.rope_sin) + row0_bytes;
            for (int pc = wave; pc * 1024 < tab_bytes; pc += C::NWAVES) {
                const uint32_t off = min((uint32_t)(pc * 1024 + lane * 16), last);
                glds16_asm(cbase, off, lds_base + (uint32_t)(pc * 1024));
                glds16_asm(sbase, off, lds_base + (uint32_t)(tab_bytes + pc * 1024));
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        const bool upper = (lane & 32) != 0;
#pragma unroll
        for (int j = 0; j < MI; ++j) {
            const int ml = wm * (MI * 16) + j * 16 + em;
            const int m = m0e + ml;
            if (m >= g.M) continue;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int gs = n0e + wn * (NI * 16) + i * 16 + en;
                if (gs >= g.N) continue;
                const int n = rope_col_of_slot(gs, g.rope_cols, g.head_dim);
                f32x4 v = acc[i][j];
                if (g.nrm_rstd != nullptr) {
                    const float rs = rs_lds[ml];
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] *= rs;
                }
                bf16x4 o;
                if (gs < g.rope_cols) {
                    const int d = (n % g.head_dim) - (upper ? half : 0);
                    f32x4 cs, sn;
                    if (staged) {
                        cs = *reinterpret_cast<const f32x4*>(smem + (ml * half + d) * 4);
                        sn = *reinterpret_cast<const f32x4*>(smem + tab_bytes + (ml * half + d) * 4);
                    } else {
                        cs = *reinterpret_cast<const f32x4*>(g.rope_cos + (int64_t)m * half + d);
                        sn = *reinterpret_cast<const f32x4*>(g.rope_sin + (int64_t)m * half + d);
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float own = bf2f(f2bf(v[t]));
                        auto sw2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(own), __float_as_uint(own), false, false);
                        const float other = __uint_as_float(upper ? sw2[0] : sw2[1]);
                        // lower half: a*cos - b*sin; upper half: b*cos + a*sin  (rotate_half(x) = [-x2 | x1])
                        o[t] = f2bf(upper ? own * cs[t] + other * sn[t] : own * cs[t] - other * sn[t]);
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) o[t] = f2bf(v[t]);
                }
                store_out4(g.C + (int64_t)m * g.ldc + n, o);
            }
        }
    } else if (MODE == MODE_PLAIN) {
#pragma unroll
        for (int j = 0; j < MI; ++j) {
            const int m = m0e + wm * (MI * 16) + j * 16 + em;
            if (m >= g.M) continue;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int n = n0e + wn * (NI * 16) + i * 16 + en;
                if (n >= g.N) continue;
                f32x4 v = acc[i][j];
                if (g.epi == VGPT_EPI_RESID) {
                    bf16x4 r = *reinterpret_cast<const bf16x4*>(g.extra + (int64_t)m * g.ldr + n);
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] += bf2f(r[t]);
                } else if (g.epi == VGPT_EPI_BIAS) {
                    bf16x4 r = *reinterpret_cast<const bf16x4*>(g.extra + n);
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] += bf2f(r[t]);
                }
                bf16x4 o;
#pragma unroll
                for (int t = 0; t < 4; ++t) o[t] = f2bf(v[t]);
                store_out4(g.C + (int64_t)m * g.ldc + n, o);
            }
        }
    } else {
        // the activation is resolved OUTSIDE the unrolled loops (one instantiation per kind): with the switch inside, the
        // 8 x 4 iterations of the 128 x 128 wave tile exceed the unroller's budget and the accumulators fall into scratch
        auto gated_store = [&](auto actc, auto keepc) {
            constexpr int ACT = decltype(actc)::value;
            constexpr bool KEEP = decltype(keepc)::value;
#pragma unroll
            for (int j = 0; j < MI; ++j) {
                const int m = m0e + wm * (MI * 16) + j * 16 + em;
                if (m >= g.M) continue;
#pragma unroll
                for (int p = 0; p < NI / 2; ++p) {
                    const int n = n0e + (wn * (NI / 2) + p) * 16 + en;  // output column
                    if (n >= g.I) continue;
                    const f32x4 gate = acc[2 * p][j], up = acc[2 * p + 1][j];
                    bf16x4 o;
                    if constexpr (KEEP) {
                        bf16x4 gb, ub;
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            gb[t] = f2bf(gate[t]);
                            ub[t] = f2bf(up[t]);
                            o[t] = f2bf(act_apply(bf2f(gb[t]), ACT) * bf2f(ub[t]));
                        }
                        store_out4(g.gu_out + (int64_t)m * g.ld_gu + n, gb);
                        store_out4(g.gu_out + (int64_t)m * g.ld_gu + g.I + n, ub);
                    } else {
#pragma unroll
                        for (int t = 0; t < 4; ++t) o[t] = f2bf(act_apply(gate[t], ACT) * up[t]);
                    }
                    store_out4(g.C + (int64_t)m * g.ldc + n, o);
                }
            }
        };
        using KT = std::true_type;
        using KF = std::false_type;
        if constexpr (PIPE != 6) {
            // the 8-wave kernels (16 iterations) stay as they were measured: activation selected inside the loops
#pragma unroll
            for (int j = 0; j < MI; ++j) {
                const int m = m0e + wm * (MI * 16) + j * 16 + em;
                if (m >= g.M) continue;
#pragma unroll
                for (int p = 0; p < NI / 2; ++p) {
                    const int n = n0e + (wn * (NI / 2) + p) * 16 + en;  // output column
                    if (n >= g.I) continue;
                    f32x4 gate = acc[2 * p][j], up = acc[2 * p + 1][j];
                    if (g.nrm_rstd != nullptr) {
                        const float rs = rs_lds[wm * (MI * 16) + j * 16 + em];
#pragma unroll
                        for (int t = 0; t < 4; ++t) { gate[t] *= rs; up[t] *= rs; }
                    }
                    bf16x4 o;
                    if (g.gu_out) {
                        bf16x4 gb, ub;
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            gb[t] = f2bf(gate[t]);
                            ub[t] = f2bf(up[t]);
                            o[t] = f2bf(act_apply(bf2f(gb[t]), g.act) * bf2f(ub[t]));
                        }
                        store_out4(g.gu_out + (int64_t)m * g.ld_gu + n, gb);
                        store_out4(g.gu_out + (int64_t)m * g.ld_gu + g.I + n, ub);
                    } else {
#pragma unroll
                        for (int t = 0; t < 4; ++t) o[t] = f2bf(act_apply(gate[t], g.act) * up[t]);
                    }
                    store_out4(g.C + (int64_t)m * g.ldc + n, o);
                }
            }
        } else if (g.act == VGPT_ACT_SILU) {
            if (g.gu_out) gated_store(std::integral_constant<int, VGPT_ACT_SILU>{}, KT{});
            else gated_store(std::integral_constant<int, VGPT_ACT_SILU>{}, KF{});
        } else if (g.act == VGPT_ACT_GELU) {
            if (g.gu_out) gated_store(std::integral_constant<int, VGPT_ACT_GELU>{}, KT{});
            else gated_store(std::integral_constant<int, VGPT_ACT_GELU>{}, KF{});
        } else {
            if (g.gu_out) gated_store(std::integral_constant<int, VGPT_ACT_GELU_TANH>{}, KT{});
            else gated_store(std::integral_constant<int, VGPT_ACT_GELU_TANH>{}, KF{});
        }
    }
    if (!more) break;
    }   // persistent walk
}

// ---------------------------------------------------------------------------------------------------------------------
// Four-wave kernel with the hand-scheduled main loop (gen/gemm_w4_gen.py -> gemm_w4_loop.inc): 256 x (NI * 32) x 64 tiles,
// one wave per SIMD holding a 128 x (NI * 16) accumulator block in AGPRs, operands global -> registers -> LDS (the same
// XOR-swizzled row images as above) with counted waits, one barrier per k-tile.  NT operands, K % 64 == 0, nk >= 2.
// The loop is ONE asm statement; the C++ here computes the tile's addresses (the per-piece row offsets travel in the lanes
// of one VGPR) and runs the epilogue straight from the accumulator registers.
#ifndef VGPT_W4_INC
#define VGPT_W4_INC "gemm_w4_loop.inc"
#endif
#include VGPT_W4_INC
// diagnostics builds only (make gemm-w4-debug-N with bit 16): the loop's stamps leave the asm statement as outputs and go to
// a buffer set through vgpt_gemm_w4_debug_buffer: per (workgroup, wave) {loop cycles, loop time in 10-ns ticks, cycles at the
// barriers, k-tiles}
#ifdef VGPT_W4_STAMPS
#define VGPT_W4_OUTS [cyc] "=s"(st_cyc), [rt] "=s"(st_rt), [bar] "=s"(st_bar)
#define VGPT_W4_OUTS_ VGPT_W4_OUTS,
uint32_t* g_w4_dbg = nullptr;
#else
#define VGPT_W4_OUTS
#define VGPT_W4_OUTS_
#endif

template <int I0, int N0, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I0 < N0) {
        f(std::integral_constant<int, I0>{});
        static_for<I0 + 1, N0>(f);
    }
}

// accumulator quad IDX (registers a[4 IDX .. 4 IDX + 3]) of the asm loop
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

template <int IDX>
__device__ __forceinline__ f32x4 w4_acc() {
    float x0, x1, x2, x3;
    asm volatile("v_accvgpr_read_b32 %0, a%c4\n\tv_accvgpr_read_b32 %1, a%c5\n\tv_accvgpr_read_b32 %2, a%c6\n\tv_accvgpr_read_b32 %3, a%c7"
                 : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3)
                 : "n"(IDX * 4), "n"(IDX * 4 + 1), "n"(IDX * 4 + 2), "n"(IDX * 4 + 3));
    return f32x4{x0, x1, x2, x3};
}

template <int MODE, int NI, bool WTR = false, bool ATR = false>
__global__ __launch_bounds__(256, 1) void gemm_w4_kernel(GemmArgs g) {
    // ATR (with WTR: dW = dY^T X, the reduction runs over the ROWS of both operands): A is staged and read like the transposed W
    // below, a partial last k-tile is zero-filled by the buffer descriptors' range check
    static_assert(!ATR || WTR, "a transposed A comes with a transposed W");
    // WTR: W is stored with the reduction index as its ROW index ([K][N]: dX = dY W of the backward): staged as a
    // [64 reduction rows][256 columns] image (512-byte rows, 32-byte units XOR-swizzled as in gemm_bf16_kernel), fragments by
    // ds_read_b64_tr_b16; a 192-wide tile (NI 6) uses the same image and ignores its last 64 columns
    static_assert(!WTR || (MODE == MODE_PLAIN && NI <= 8), "transposed W: plain kernel, 256- / 192-wide tiles");
    constexpr bool ROPE = MODE == MODE_ROPE;
    constexpr int BM = 256, BN = NI * 32, MI = 8;
    constexpr int A_BYTES = BM * BK * 2;
    // LDS: NI <= 8: A buffers at 0 / 32 KiB, W buffers at 64 / 96 KiB (toggled by XOR 0x8000); NI == 9 (36-KiB W images):
    // [A0 | W0 | A1 | W1], the two halves W4_BUFD bytes apart; behind the staging area the 256 rstd values of the folded RMSNorm
    constexpr int W4_BUFD = A_BYTES + BN * BK * 2;
    constexpr int STAGE_BYTES = NI == 9 ? 2 * W4_BUFD : 4 * A_BYTES;
    constexpr int DUMP_OFF = 3 * A_BYTES;              // NI == 9: the VGPR accumulators (sub-tile column 8) leave through here
    static_assert(MODE != MODE_GATED || NI != 9, "256 x 288 tiles: no gated mode (a wave's 144 slots are 4.5 gate / up pairs)");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lds_base =
        __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
    if (lds_base != 0) __builtin_trap();   // the loop toggles its staging buffers by XOR on absolute LDS addresses
#ifdef VGPT_W4_STAMPS
    const uint32_t st_t0 = (uint32_t)__builtin_amdgcn_s_memrealtime();
#endif

    // ---- tile order: as gemm_bf16_kernel (XCD-aware remap, grouped along m) ----
    const int nwg = g.tiles_m * g.tiles_n;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    constexpr int GROUP = 8;
    const int in_group = GROUP * g.tiles_n;
    const int group_id = bid / in_group;
    const int first_m = group_id * GROUP;
    const int gsz = min(g.tiles_m - first_m, GROUP);
    const int tm = first_m + (bid % in_group) % gsz;
    const int tn = (bid % in_group) / gsz;
    const int m0 = tm * BM;
    const int n0 = tn * (MODE == MODE_GATED ? BN / 2 : BN);
    const int n_rows_w = (MODE == MODE_GATED) ? 2 * g.I : g.N;

    // ---- operands of the loop ----
    // per piece and lane: byte offset of the 16 bytes this lane fetches, relative to the tile's first A row (oa) / to the W
    // origin below (ow); rows clamped per lane into the matrix (rows past M / columns past N are computed and never stored)
    const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
    uint32_t oa[8] = {}, ow[8] = {};
#pragma unroll
    for (int p = 0; p < (NI == 9 || ATR ? 0 : 8); ++p) {
        const int r = min((wave * 8 + p) * 8 + srow, g.M - 1 - m0);
        oa[p] = (uint32_t)(r * (int)g.lda + schunk * 8) * 2u;
    }
    if constexpr (WTR) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {      // piece = two 512-byte rows of the image: lane -> (row, 16-byte chunk)
            const int row = (wave * 8 + p) * 2 + (lane >> 5), t_c = lane & 31;
            const int key = (((row >> 3) & 1) << 2) | (row & 3);
            const int lchunk = ((((t_c >> 1) ^ key) << 1) | (t_c & 1));
            const int col = min(n0 + lchunk * 8, g.N - 8) - n0;
            ow[p] = (uint32_t)(row * (int)g.ldw + col) * 2u;
            if constexpr (ATR) {
                const int cola = min(m0 + lchunk * 8, g.M - 8) - m0;
                oa[p] = (uint32_t)(row * (int)g.lda + cola) * 2u;
            }
        }
    }
#pragma unroll
    for (int p = 0; p < (NI == 9 || WTR ? 0 : 8); ++p) {
        const int sl = (wave * NI + min(p, NI - 1)) * 8 + srow;   // n-slot inside the tile
        int wr;
        if constexpr (ROPE) wr = min(rope_col_of_slot(n0 + sl, g.rope_cols, g.head_dim), n_rows_w - 1);
        else wr = min(w_row_of_slot<MODE>(n0, sl, g.I), n_rows_w - 1) - (MODE == MODE_GATED ? 0 : n0);
        ow[p] = (uint32_t)(wr * (int)g.ldw + schunk * 8) * 2u;
    }
    const bf16* a_org = ATR ? g.A + m0 : g.A + (int64_t)m0 * g.lda;
    const bf16* w_org = WTR ? g.W + n0 : ((MODE == MODE_GATED || ROPE) ? g.W : g.W + (int64_t)n0 * g.ldw);
    auto srd = [](const void* base) {
        const uint64_t b = (uint64_t)(uintptr_t)base;
        v4i32 r;
        r[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)b);
        r[1] = __builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32) & 0xffff);
        r[2] = -1;             // num_records: every offset the loop forms stays inside the matrix by construction
        r[3] = 0x00020000;
        return r;
    };
    v4i32 srdA = srd(a_org), srdW = srd(w_org);
    if constexpr (ATR) {      // reduction rows past K read as zeros (the last k-tile of a token count that is no multiple of 64)
        srdA[2] = __builtin_amdgcn_readfirstlane((int)(((int64_t)g.K * g.lda - m0) * 2));
        srdW[2] = __builtin_amdgcn_readfirstlane((int)(((int64_t)g.K * g.ldw - n0) * 2));
    }
    const int wm = wave >> 1, wn = wave & 1;
    const int frow = lane & 15, fk = lane >> 4, sw = frow & 7;
    constexpr int W_BASE = NI == 9 ? A_BYTES : 2 * A_BYTES;
    uint32_t rdA = (uint32_t)((wm * 128 + frow) * 128 + ((fk ^ sw) * 16));
    uint32_t rdW = (uint32_t)(W_BASE + (wn * NI * 16 + frow) * 128 + ((fk ^ sw) * 16));
    uint32_t wrA = (uint32_t)(wave * 8192 + lane * 16);
    uint32_t wrW = (uint32_t)(W_BASE + wave * (WTR ? 8 : NI) * 1024 + lane * 16);
    const int nk = __builtin_amdgcn_readfirstlane(ATR ? (g.K + BK - 1) / BK : g.K / BK);

    // folded RMSNorm, consumer side: 1 / rms of this tile's 256 rows into LDS behind the staging buffers (thread t: row t);
    // the loop's barriers order it before the epilogue's reads
    float* rs_lds = reinterpret_cast<float*>(smem + STAGE_BYTES);
    if constexpr (MODE != MODE_PLAIN) {
        if (g.nrm_rstd != nullptr) rs_lds[tid] = m0 + tid < g.M ? g.nrm_rstd[m0 + tid] : 0.f;
    }
    [[maybe_unused]] uint32_t st_cyc = 0, st_rt = 0, st_bar = 0;
#ifdef VGPT_W4_STAMPS
    const uint32_t st_t1 = (uint32_t)__builtin_amdgcn_s_memrealtime();
#endif
    if constexpr (WTR) {
        // fragment addresses of the transposed image: lane (g4 = lane >> 4, li = lane & 15) of sub-tile `unit` reads rows
        // 32 ks + 8 g4 + (li >> 2) (+ 4) at columns 4 (li & 3) .. + 4 of the unit's (swizzled) 32-byte column group
        const int tr_li = lane & 15, tr_g = lane >> 4;
        const int tr_key = ((tr_g & 1) << 2) | (tr_li >> 2);
        uint32_t rw0[8], rw1[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int unit = wn * NI + min(i, NI - 1);
            rw0[i] = rw1[i] = (uint32_t)(W_BASE + (8 * tr_g + (tr_li >> 2)) * 512 + 8 * (tr_li & 3) + ((unit ^ tr_key) << 5));
        }
        uint32_t rdA1 = rdA ^ 64u;
        const int wstep = __builtin_amdgcn_readfirstlane((int)(64 * g.ldw * 2));
        if constexpr (ATR) {
            uint32_t ra0[8], ra1[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int unit = wm * 8 + j;
                ra0[j] = ra1[j] = (uint32_t)((8 * tr_g + (tr_li >> 2)) * 512 + 8 * (tr_li & 3) + ((unit ^ tr_key) << 5));
            }
            const int astep = __builtin_amdgcn_readfirstlane((int)(64 * g.lda * 2));
#define VGPT_W4_ATR_OPERANDS(NIW)                                                                                                 \
            : VGPT_W4_OUTS_ [wrA] "+v"(wrA), [wrW] "+v"(wrW),                                                                      \
              [ra0_0] "+v"(ra0[0]), [ra0_1] "+v"(ra0[1]), [ra0_2] "+v"(ra0[2]), [ra0_3] "+v"(ra0[3]), [ra0_4] "+v"(ra0[4]),          \
              [ra0_5] "+v"(ra0[5]), [ra0_6] "+v"(ra0[6]), [ra0_7] "+v"(ra0[7]), [ra1_0] "+v"(ra1[0]), [ra1_1] "+v"(ra1[1]),          \
              [ra1_2] "+v"(ra1[2]), [ra1_3] "+v"(ra1[3]), [ra1_4] "+v"(ra1[4]), [ra1_5] "+v"(ra1[5]), [ra1_6] "+v"(ra1[6]),          \
              [ra1_7] "+v"(ra1[7]),                                                                                               \
              [rw0_0] "+v"(rw0[0]), [rw0_1] "+v"(rw0[1]), [rw0_2] "+v"(rw0[2]), [rw0_3] "+v"(rw0[3]), [rw0_4] "+v"(rw0[4]),          \
              [rw0_5] "+v"(rw0[5]), [rw0_6] "+v"(rw0[6]), [rw0_7] "+v"(rw0[7]), [rw1_0] "+v"(rw1[0]), [rw1_1] "+v"(rw1[1]),          \
              [rw1_2] "+v"(rw1[2]), [rw1_3] "+v"(rw1[3]), [rw1_4] "+v"(rw1[4]), [rw1_5] "+v"(rw1[5]), [rw1_6] "+v"(rw1[6]),          \
              [rw1_7] "+v"(rw1[7])                                                                                                \
            : [srdA] "s"(srdA), [srdW] "s"(srdW), [nk] "s"(nk), [wv] "s"(wave), [wstep] "s"(wstep), [astep] "s"(astep),              \
              [oa0] "v"(oa[0]), [oa1] "v"(oa[1]), [oa2] "v"(oa[2]), [oa3] "v"(oa[3]), [oa4] "v"(oa[4]), [oa5] "v"(oa[5]),            \
              [oa6] "v"(oa[6]), [oa7] "v"(oa[7]), [ow0] "v"(ow[0]), [ow1] "v"(ow[1]), [ow2] "v"(ow[2]), [ow3] "v"(ow[3]),            \
              [ow4] "v"(ow[4]), [ow5] "v"(ow[5]), [ow6] "v"(ow[6]), [ow7] "v"(ow[7])                                                \
            : VGPT_W4_CLOBBERS_WTR
            if constexpr (NI == 8) asm volatile(VGPT_W4_ASM_NI8_ATR VGPT_W4_ATR_OPERANDS(8));
            else asm volatile(VGPT_W4_ASM_NI6_ATR VGPT_W4_ATR_OPERANDS(6));
#undef VGPT_W4_ATR_OPERANDS
        } else if constexpr (NI == 8) {
            asm volatile(VGPT_W4_ASM_NI8_WTR
                         : VGPT_W4_OUTS_ [rdA0] "+v"(rdA), [rdA1] "+v"(rdA1), [wrA] "+v"(wrA), [wrW] "+v"(wrW),
                           [rw0_0] "+v"(rw0[0]), [rw0_1] "+v"(rw0[1]), [rw0_2] "+v"(rw0[2]), [rw0_3] "+v"(rw0[3]), [rw0_4] "+v"(rw0[4]),
                           [rw0_5] "+v"(rw0[5]), [rw0_6] "+v"(rw0[6]), [rw0_7] "+v"(rw0[7]), [rw1_0] "+v"(rw1[0]), [rw1_1] "+v"(rw1[1]),
                           [rw1_2] "+v"(rw1[2]), [rw1_3] "+v"(rw1[3]), [rw1_4] "+v"(rw1[4]), [rw1_5] "+v"(rw1[5]), [rw1_6] "+v"(rw1[6]),
                           [rw1_7] "+v"(rw1[7])
                         : [srdA] "s"(srdA), [srdW] "s"(srdW), [nk] "s"(nk), [wv] "s"(wave), [wstep] "s"(wstep),
                           [oa0] "v"(oa[0]), [oa1] "v"(oa[1]), [oa2] "v"(oa[2]), [oa3] "v"(oa[3]), [oa4] "v"(oa[4]), [oa5] "v"(oa[5]),
                           [oa6] "v"(oa[6]), [oa7] "v"(oa[7]), [ow0] "v"(ow[0]), [ow1] "v"(ow[1]), [ow2] "v"(ow[2]), [ow3] "v"(ow[3]),
                           [ow4] "v"(ow[4]), [ow5] "v"(ow[5]), [ow6] "v"(ow[6]), [ow7] "v"(ow[7])
                         : VGPT_W4_CLOBBERS_WTR);
        } else {
            asm volatile(VGPT_W4_ASM_NI6_WTR
                         : VGPT_W4_OUTS_ [rdA0] "+v"(rdA), [rdA1] "+v"(rdA1), [wrA] "+v"(wrA), [wrW] "+v"(wrW),
                           [rw0_0] "+v"(rw0[0]), [rw0_1] "+v"(rw0[1]), [rw0_2] "+v"(rw0[2]), [rw0_3] "+v"(rw0[3]), [rw0_4] "+v"(rw0[4]),
                           [rw0_5] "+v"(rw0[5]), [rw1_0] "+v"(rw1[0]), [rw1_1] "+v"(rw1[1]), [rw1_2] "+v"(rw1[2]), [rw1_3] "+v"(rw1[3]),
                           [rw1_4] "+v"(rw1[4]), [rw1_5] "+v"(rw1[5])
                         : [srdA] "s"(srdA), [srdW] "s"(srdW), [nk] "s"(nk), [wv] "s"(wave), [wstep] "s"(wstep),
                           [oa0] "v"(oa[0]), [oa1] "v"(oa[1]), [oa2] "v"(oa[2]), [oa3] "v"(oa[3]), [oa4] "v"(oa[4]), [oa5] "v"(oa[5]),
                           [oa6] "v"(oa[6]), [oa7] "v"(oa[7]), [ow0] "v"(ow[0]), [ow1] "v"(ow[1]), [ow2] "v"(ow[2]), [ow3] "v"(ow[3]),
                           [ow4] "v"(ow[4]), [ow5] "v"(ow[5]), [ow6] "v"(ow[6]), [ow7] "v"(ow[7])
                         : VGPT_W4_CLOBBERS_WTR);
        }
    } else if constexpr (NI == 9) {
        // whole tiles only (launch_w4 checks): no per-lane row clamp, one SGPR offset per piece (lane p of `tab`: p < 8 the A
        // pieces, 8 + p the W pieces) on top of one per-lane offset per operand
        uint32_t tab;
        {
            const int pw = min(max(lane - 8, 0), NI - 1);
            const int sl8 = (wave * NI + pw) * 8;
            const int wrow = ROPE ? rope_col_of_slot(n0 + sl8, g.rope_cols, g.head_dim) : sl8;
            tab = lane < 8 ? (uint32_t)((wave * 8 + lane) * 8 * (int)g.lda) * 2u : (uint32_t)(wrow * (int)g.ldw) * 2u;
        }
        const uint32_t va = (uint32_t)(srow * (int)g.lda + schunk * 8) * 2u, vw = (uint32_t)(srow * (int)g.ldw + schunk * 8) * 2u;
        uint32_t rdA1 = rdA ^ 64u, rdW1 = rdW ^ 64u;
        const uint32_t dump = (uint32_t)(DUMP_OFF + wave * 8192 + lane * 16);
        const int bufd = W4_BUFD;
        asm volatile(VGPT_W4_ASM_NI9
                     : VGPT_W4_OUTS_ [rdA0] "+v"(rdA), [rdA1] "+v"(rdA1), [rdW0] "+v"(rdW), [rdW1] "+v"(rdW1), [wrA] "+v"(wrA), [wrW] "+v"(wrW)
                     : [srdA] "s"(srdA), [srdW] "s"(srdW), [nk] "s"(nk), [wv] "s"(wave), [tab] "v"(tab), [va] "v"(va), [vw] "v"(vw),
                       [dump] "v"(dump), [bufd] "s"(bufd)
                     : VGPT_W4_CLOBBERS_NI9);
    } else if constexpr (NI == 8) {
        asm volatile(VGPT_W4_ASM_NI8
                     : VGPT_W4_OUTS
                     : [srdA] "s"(srdA), [srdW] "s"(srdW), [rdA] "v"(rdA), [rdW] "v"(rdW), [wrA] "v"(wrA), [wrW] "v"(wrW), [nk] "s"(nk), [wv] "s"(wave),
                       [oa0] "v"(oa[0]), [oa1] "v"(oa[1]), [oa2] "v"(oa[2]), [oa3] "v"(oa[3]), [oa4] "v"(oa[4]), [oa5] "v"(oa[5]),
                       [oa6] "v"(oa[6]), [oa7] "v"(oa[7]), [ow0] "v"(ow[0]), [ow1] "v"(ow[1]), [ow2] "v"(ow[2]), [ow3] "v"(ow[3]),
                       [ow4] "v"(ow[4]), [ow5] "v"(ow[5]), [ow6] "v"(ow[6]), [ow7] "v"(ow[7])
                     : VGPT_W4_CLOBBERS);
    } else {
        asm volatile(VGPT_W4_ASM_NI6
                     : VGPT_W4_OUTS
                     : [srdA] "s"(srdA), [srdW] "s"(srdW), [rdA] "v"(rdA), [rdW] "v"(rdW), [wrA] "v"(wrA), [wrW] "v"(wrW), [nk] "s"(nk), [wv] "s"(wave),
                       [oa0] "v"(oa[0]), [oa1] "v"(oa[1]), [oa2] "v"(oa[2]), [oa3] "v"(oa[3]), [oa4] "v"(oa[4]), [oa5] "v"(oa[5]),
                       [oa6] "v"(oa[6]), [oa7] "v"(oa[7]), [ow0] "v"(ow[0]), [ow1] "v"(ow[1]), [ow2] "v"(ow[2]), [ow3] "v"(ow[3]),
                       [ow4] "v"(ow[4]), [ow5] "v"(ow[5])
                     : VGPT_W4_CLOBBERS);
    }

#ifdef VGPT_W4_STAMPS
    const uint32_t st_t2 = (uint32_t)__builtin_amdgcn_s_memrealtime();
    if (g.dbg && lane == 0) {
        uint32_t* d = g.dbg + (blockIdx.x * 4 + wave) * 8;
        d[0] = st_cyc; d[1] = st_rt; d[2] = st_bar; d[3] = (uint32_t)nk;
        d[4] = st_t0; d[5] = st_t1; d[6] = st_t2;
    }
#endif
    // ---- epilogue: lane holds m = lane & 15, n = (lane >> 4) * 4 + reg of each 16 x 16 sub-tile (i: n, j: m).  Branch-free:
    //      outputs (and the residual) go through buffer descriptors anchored at the tile's origin, and a lane whose row or
    //      column lies outside the matrix gets an offset past the descriptor's range -- its loads return zeros, its stores are
    //      dropped -- so every row is one basic block (the residual quads of row j + 1 are requested before row j is stored) ----
    const int em = lane & 15, en = (lane >> 4) * 4;
    // accumulator quad of sub-tile (i, j): AGPRs of the asm loop; column 8 of a 288-wide tile comes back from LDS
    auto acc_of = [&](auto ic_, auto jc_) -> f32x4 {
        constexpr int i_ = decltype(ic_)::value, j_ = decltype(jc_)::value;
        if constexpr (i_ < 8) return w4_acc<i_ * 8 + j_>();
        else return *reinterpret_cast<const f32x4*>(smem + DUMP_OFF + wave * 8192 + j_ * 1024 + lane * 16);
    };
    constexpr uint32_t OOB = 0x80000000u;
    auto make_rs = [](const void* base) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000); };
    auto pack4 = [](const f32x4& v) {
        bf16x4 o;
#pragma unroll
        for (int t = 0; t < 4; ++t) o[t] = f2bf(v[t]);
        return __builtin_bit_cast(u32x2_t, o);
    };
    const int nl0 = wn * (NI * 16) + en;                      // this lane's first slot inside the tile
    if constexpr (ROPE) {
        // as gemm_bf16_kernel: product rounded to bf16, rotated in fp32 with the partner column from lane ^ 32, the cos / sin
        // rows of the tile staged through the (now free) LDS
        const int half = g.head_dim >> 1;
        const int tab_bytes = (BM * half * 4 + 1023) & ~1023;
        // (launch_w4 only takes head dims whose two tables fit: w4_ok)
        {
            __syncthreads();
            const int64_t row0_bytes = (int64_t)m0 * half * 4;
            const uint32_t last = (uint32_t)min((int64_t)g.M * half * 4 - row0_bytes - 16, (int64_t)tab_bytes);
            const char* cbase = reinterpret_cast<const char*>(g.rope_cos) + row0_bytes;
            const char* sbase = reinterpret_cast<const char*>(g.rope_sin) + row0_bytes;
            for (int pc = wave; pc * 1024 < tab_bytes; pc += 4) {
                const uint32_t off = min((uint32_t)(pc * 1024 + lane * 16), last);
                glds16_asm(cbase, off, lds_base + (uint32_t)(pc * 1024));
                glds16_asm(sbase, off, lds_base + (uint32_t)(tab_bytes + pc * 1024));
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        // Branch-free over the sub-tiles (one basic block per row: the LDS reads, lane swaps and stores of consecutive sub-tiles
        // overlap; with a branch per sub-tile the compiler drained every store before the next sub-tile's reads -- 72 round
        // trips per lane).  Slot u = gs >> 4 is wave-uniform, the lane part of a slot is en = 4 (lane >> 4): inside the
        // rotated region column n = (u / per) hd + (u % per) 8 + (en & 7) + (en & 8 ? hd / 2 : 0), the partner sits in lane ^ 32
        // (en & 8 <=> upper half-wave), and both read cos / sin at d = (u % per) 8 + (en & 7) -- a valid table index for the
        // plain (V) columns too, whose rotation is computed and discarded.
        // (16-byte stores through v_permlane16_swap of sub-tile pairs, as the gated and plain epilogues have, were measured here
        // and lost: the per-lane head / slot arithmetic and 37-57 spilled SGPRs cost more than the halved store count returned --
        // w4 / eight-wave 0.986 against 0.965 with the 8-byte stores, profiles/r04_w4_check_v7_rope_wide_stores.log)
        const auto rsC = make_rs(g.C + (int64_t)m0 * g.ldc);
        const bool upper = (lane & 32) != 0;
        const int per = g.head_dim >> 4;
        auto rope_store = [&](auto normc) {
            constexpr bool NORM = decltype(normc)::value;
            static_for<0, MI>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const int ml = wm * 128 + j * 16 + em;
                const bool m_ok = m0 + ml < g.M;
                const int mr = m_ok ? ml : 0;
                [[maybe_unused]] const float rs = NORM ? rs_lds[ml] : 1.0f;
                static_for<0, NI>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    const int gs0 = n0 + wn * (NI * 16) + i * 16;         // wave-uniform first slot of the sub-tile
                    const int u = gs0 >> 4;
                    const bool rot = gs0 < g.rope_cols;                    // rope_cols is a multiple of 16
                    const int dcol = (u % per) * 8 + (en & 7);
                    const int n = rot ? (u / per) * g.head_dim + dcol + ((en & 8) ? half : 0) : gs0 + en;
                    f32x4 v = acc_of(ic, jc);
                    if constexpr (NORM) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) v[t] *= rs;
                    }
                    const f32x4 cs = *reinterpret_cast<const f32x4*>(smem + (mr * half + dcol) * 4);
                    const f32x4 sn = *reinterpret_cast<const f32x4*>(smem + tab_bytes + (mr * half + dcol) * 4);
                    f32x4 o;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float own = bf2f(f2bf(v[t]));
                        auto sw2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(own), __float_as_uint(own), false, false);
                        const float other = __uint_as_float(upper ? sw2[0] : sw2[1]);
                        const float r_ = upper ? own * cs[t] + other * sn[t] : own * cs[t] - other * sn[t];
                        o[t] = rot ? r_ : v[t];
                    }
                    const uint32_t off = (m_ok && gs0 + en < g.N) ? (uint32_t)(ml * (int)g.ldc + n) * 2u : OOB;
                    __builtin_amdgcn_raw_buffer_store_b64(pack4(o), rsC, off, 0, 0);
                });
            });
        };
        if (g.nrm_rstd != nullptr) rope_store(std::true_type{});
        else rope_store(std::false_type{});
    } else if constexpr (MODE == MODE_PLAIN) {
        auto plain_store = [&](auto epic, auto widec) {
            constexpr int EPI = decltype(epic)::value;
            // WIDE (N % 8 == 0): 16-byte residual loads and output stores.  v_permlane16_swap on the fp32 accumulators of TWO
            // sub-tiles trades the first one's odd lane rows for the second one's even ones (see the gated epilogue): a lane
            // then holds 8 consecutive columns of one sub-tile row -- half the load and store instructions of the epilogue's
            // burst, whole 16-byte segments.  An odd last sub-tile (NI = 9) and ragged widths keep the 8-byte form.
            constexpr bool WIDE = decltype(widec)::value;
            constexpr int NP = WIDE ? NI / 2 : 0;
            typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
            const auto rsC = make_rs(g.C + (int64_t)m0 * g.ldc + n0);
            const bf16* rbase = EPI == VGPT_EPI_RESID ? g.extra + (int64_t)m0 * g.ldr + n0 : (EPI == VGPT_EPI_BIAS ? g.extra + n0 : g.C);
            const auto rsR = make_rs(rbase);
            const int grp = lane >> 4;
            auto wide_col = [&](int q) { return wn * (NI * 16) + (2 * q + (grp & 1)) * 16 + 4 * (grp & 2); };
            // every residual value of the wave tile is requested up front (MI * NI * 2 registers: the loop's registers are
            // dead by now): with the quads of one row in flight at a time the epilogue was bound by the round trip of a
            // load -- 13.4 us for the 96 KiB of a 256 x 192 tile (in-kernel stamps, profiles/r04_w4_stamps_v2.log)
            u32x4_t rw[MI][NP > 0 ? NP : 1];
            u32x2_t r[MI][NI - 2 * NP > 0 ? NI - 2 * NP : 1];
            if constexpr (EPI != VGPT_EPI_NONE) {
                static_for<0, MI>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    const int ml = wm * 128 + j * 16 + em;
                    const bool row_ok = m0 + ml < g.M;
                    static_for<0, NP>([&](auto qc) {
                        constexpr int q = decltype(qc)::value;
                        const int nl = wide_col(q);
                        const bool ok = row_ok && n0 + nl < g.N;
                        const uint32_t off = EPI == VGPT_EPI_RESID ? (uint32_t)(ml * (int)g.ldr + nl) * 2u : (uint32_t)nl * 2u;
                        rw[j][q] = __builtin_amdgcn_raw_buffer_load_b128(rsR, ok ? off : OOB, 0, 0);
                    });
                    static_for<2 * NP, NI>([&](auto ic) {
                        constexpr int i = decltype(ic)::value;
                        const int nl = nl0 + i * 16;
                        const bool ok = row_ok && n0 + nl < g.N;
                        const uint32_t off = EPI == VGPT_EPI_RESID ? (uint32_t)(ml * (int)g.ldr + nl) * 2u : (uint32_t)nl * 2u;
                        r[j][i - 2 * NP] = __builtin_amdgcn_raw_buffer_load_b64(rsR, ok ? off : OOB, 0, 0);
                    });
                });
            }
            static_for<0, MI>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const int ml = wm * 128 + j * 16 + em;
                const bool row_ok = m0 + ml < g.M;
                float ssq = 0.f;
                static_for<0, NP>([&](auto qc) {
                    constexpr int q = decltype(qc)::value;
                    const f32x4 v0 = acc_of(std::integral_constant<int, 2 * q>{}, jc), v1 = acc_of(std::integral_constant<int, 2 * q + 1>{}, jc);
                    float w8[8];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const auto sw_ = __builtin_amdgcn_permlane16_swap(__float_as_uint(v0[t]), __float_as_uint(v1[t]), false, false);
                        w8[t] = __uint_as_float(sw_[0]);
                        w8[4 + t] = __uint_as_float(sw_[1]);
                    }
                    const int nl = wide_col(q);
                    const bool ok = row_ok && n0 + nl < g.N;
                    if constexpr (EPI != VGPT_EPI_NONE) {
                        const bf16x8 rb = __builtin_bit_cast(bf16x8, rw[j][q]);
#pragma unroll
                        for (int t = 0; t < 8; ++t) w8[t] += bf2f(rb[t]);
                    }
                    bf16x8 o8;
#pragma unroll
                    for (int t = 0; t < 8; ++t) o8[t] = f2bf(w8[t]);
                    if constexpr (EPI == VGPT_EPI_RESID) {
                        float qs = 0.f;
#pragma unroll
                        for (int t = 0; t < 8; ++t) qs += bf2f(o8[t]) * bf2f(o8[t]);
                        ssq += (n0 + nl < g.N) ? qs : 0.f;
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o8), rsC,
                                                           ok ? (uint32_t)(ml * (int)g.ldc + nl) * 2u : OOB, 0, 0);
                });
                static_for<2 * NP, NI>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    const int nl = nl0 + i * 16;
                    const bool ok = row_ok && n0 + nl < g.N;
                    f32x4 v = acc_of(ic, jc);
                    if constexpr (EPI != VGPT_EPI_NONE) {
                        const bf16x4 rb = __builtin_bit_cast(bf16x4, r[j][i - 2 * NP]);
#pragma unroll
                        for (int t = 0; t < 4; ++t) v[t] += bf2f(rb[t]);
                    }
                    const u32x2_t ob = pack4(v);
                    if constexpr (EPI == VGPT_EPI_RESID) {
                        // sum of squares of the ROUNDED outputs (what the next RMSNorm reads), columns past N excluded
                        const bf16x4 o4 = __builtin_bit_cast(bf16x4, ob);
                        float q = 0.f;
#pragma unroll
                        for (int t = 0; t < 4; ++t) q += bf2f(o4[t]) * bf2f(o4[t]);
                        ssq += (n0 + nl < g.N) ? q : 0.f;
                    }
                    __builtin_amdgcn_raw_buffer_store_b64(ob, rsC, ok ? (uint32_t)(ml * (int)g.ldc + nl) * 2u : OOB, 0, 0);
                });
                if constexpr (EPI == VGPT_EPI_RESID) {
                    if (g.ssq_out != nullptr) {      // the four lane groups of a row (lane >> 4), fixed order; lanes 0..15 store
                        ssq += __shfl_xor(ssq, 16, 64);
                        ssq += __shfl_xor(ssq, 32, 64);
                        // write-through (sc1): read by ANOTHER workgroup below, MI355X_MICROARCH.md inter-workgroup visibility
                        if (lane < 16 && row_ok)
                            __hip_atomic_store(g.ssq_out + (int64_t)(tn * 2 + wn) * g.nrm_ld + m0 + ml, ssq, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            });
            if constexpr (EPI == VGPT_EPI_RESID) {
                if (g.ssq_out != nullptr) {
                    // The workgroup that arrives LAST at its 256-row block's counter adds up the block's partial sums -- every
                    // one stored sc1 and drained (vmcnt(0) of every storing wave, then the barrier) before its workgroup's
                    // arrival -- in a fixed order and writes the rows' 1 / rms: the consumers read one float per row.
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    int* flag = reinterpret_cast<int*>(smem + STAGE_BYTES);
                    if (tid == 0) {
                        const int old = __hip_atomic_fetch_add(g.ssq_cnt + tm, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        *flag = old == g.tiles_n - 1;
                        if (old == g.tiles_n - 1) __hip_atomic_store(g.ssq_cnt + tm, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    __syncthreads();
                    if (*flag && m0 + tid < g.M) {
                        const float* pp = g.ssq_out + m0 + tid;
                        const int parts = g.tiles_n * 2;
                        float ss = 0.f;
                        int p_ = 0;
                        for (; p_ + 8 <= parts; p_ += 8) {      // eight loads in flight, added in index order
                            float v8[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u)
                                v8[u] = __hip_atomic_load(pp + (int64_t)(p_ + u) * g.nrm_ld, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                            for (int u = 0; u < 8; ++u) ss += v8[u];
                        }
                        for (; p_ < parts; ++p_)
                            ss += __hip_atomic_load(pp + (int64_t)p_ * g.nrm_ld, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        g.rstd_out[m0 + tid] = rsqrtf(ss * g.nrm_inv_h + g.nrm_eps);
                    }
                }
            }
        };
        const bool wide = (g.N & 7) == 0;
        auto by_width = [&](auto epic) {
            if (wide) plain_store(epic, std::true_type{});
            else plain_store(epic, std::false_type{});
        };
        if (g.epi == VGPT_EPI_RESID) by_width(std::integral_constant<int, VGPT_EPI_RESID>{});
        else if (g.epi == VGPT_EPI_BIAS) by_width(std::integral_constant<int, VGPT_EPI_BIAS>{});
        else by_width(std::integral_constant<int, VGPT_EPI_NONE>{});
    } else {
        auto gated_store = [&](auto actc, auto keepc, auto normc) {
            constexpr int ACT = decltype(actc)::value;
            constexpr bool KEEP = decltype(keepc)::value;
            constexpr bool NORM = decltype(normc)::value;
            const auto rsC = make_rs(g.C + (int64_t)m0 * g.ldc + n0);
            const auto rsG = make_rs(KEEP ? g.gu_out + (int64_t)m0 * g.ld_gu + n0 : g.C);
            const int ol0 = wn * (NI / 2) * 16 + en;          // this lane's first OUTPUT column inside the tile
            static_for<0, MI>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const int ml = wm * 128 + j * 16 + em;
                const bool row_ok = m0 + ml < g.M;
                [[maybe_unused]] const float rs = NORM ? rs_lds[ml] : 1.0f;
                // Inference form: 16-byte stores.  A lane holds 4 consecutive columns of a sub-tile (8 bytes of bf16); lane
                // groups g = lane >> 4 hold columns 4 g.  v_permlane16_swap on the packed outputs X, Y of TWO sub-tiles trades
                // X's odd lane rows for Y's even ones: an even group then holds columns 4 g .. 4 g + 7 of the first sub-tile,
                // an odd group 4 (g - 1) .. 4 g + 3 of the second -- half the store instructions, whole 16-byte segments
                // (the epilogue's burst of stores is issue-bound: MI355X_MICROARCH.md, T21).
                constexpr int NPAIR = (NI / 2) / 2;
                static_for<0, NPAIR>([&](auto qc) {
                    constexpr int q = decltype(qc)::value;
                    u32x2_t ob[2], gb2[2], ub2[2];
                    static_for<0, 2>([&](auto hc) {
                        constexpr int p = 2 * q + decltype(hc)::value;
                        f32x4 gate = acc_of(std::integral_constant<int, 2 * p>{}, jc), up = acc_of(std::integral_constant<int, 2 * p + 1>{}, jc);
                        f32x4 o;
                        if constexpr (KEEP) {      // training forward: [gate | up] rounded to bf16 leave too, the activation is formed
                            bf16x4 gb, ub;         // from the rounded values (the un-fused pair's arithmetic)
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                gb[t] = f2bf(gate[t]);
                                ub[t] = f2bf(up[t]);
                                o[t] = act_apply(bf2f(gb[t]), ACT) * bf2f(ub[t]);
                            }
                            gb2[decltype(hc)::value] = __builtin_bit_cast(u32x2_t, gb);
                            ub2[decltype(hc)::value] = __builtin_bit_cast(u32x2_t, ub);
                        } else {
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                if constexpr (NORM) { gate[t] *= rs; up[t] *= rs; }
                                o[t] = act_apply(gate[t], ACT) * up[t];
                            }
                        }
                        ob[decltype(hc)::value] = pack4(o);
                    });
                    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
                    auto widen = [&](const u32x2_t (&x)[2]) {
                        const auto s0 = __builtin_amdgcn_permlane16_swap(x[0][0], x[1][0], false, false);
                        const auto s1 = __builtin_amdgcn_permlane16_swap(x[0][1], x[1][1], false, false);
                        return u32x4_t{s0[0], s1[0], s0[1], s1[1]};
                    };
                    const int grp = lane >> 4;
                    const int nl = wn * (NI / 2) * 16 + (2 * q + (grp & 1)) * 16 + 4 * (grp & 2);
                    const bool ok = row_ok && n0 + nl < g.I;        // I % 16 == 0: the 8 columns are in or out together
                    if constexpr (KEEP) {
                        const uint32_t og = ok ? (uint32_t)(ml * (int)g.ld_gu + nl) * 2u : OOB;
                        __builtin_amdgcn_raw_buffer_store_b128(widen(gb2), rsG, og, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b128(widen(ub2), rsG, og, g.I * 2, 0);
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(widen(ob), rsC, ok ? (uint32_t)(ml * (int)g.ldc + nl) * 2u : OOB, 0, 0);
                });
                static_for<2 * NPAIR, NI / 2>([&](auto pc) {
                    constexpr int p = decltype(pc)::value;
                    const int nl = ol0 + p * 16;
                    const bool ok = row_ok && n0 + nl < g.I;
                    f32x4 gate = acc_of(std::integral_constant<int, 2 * p>{}, jc), up = acc_of(std::integral_constant<int, 2 * p + 1>{}, jc);
                    if constexpr (NORM) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) { gate[t] *= rs; up[t] *= rs; }
                    }
                    f32x4 o;
                    if constexpr (KEEP) {
                        bf16x4 gb, ub;
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            gb[t] = f2bf(gate[t]);
                            ub[t] = f2bf(up[t]);
                            o[t] = act_apply(bf2f(gb[t]), ACT) * bf2f(ub[t]);
                        }
                        const uint32_t og = ok ? (uint32_t)(ml * (int)g.ld_gu + nl) * 2u : OOB;
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, gb), rsG, og, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, ub), rsG, og, g.I * 2, 0);
                    } else {
#pragma unroll
                        for (int t = 0; t < 4; ++t) o[t] = act_apply(gate[t], ACT) * up[t];
                    }
                    __builtin_amdgcn_raw_buffer_store_b64(pack4(o), rsC, ok ? (uint32_t)(ml * (int)g.ldc + nl) * 2u : OOB, 0, 0);
                });
            });
        };
        using KT = std::true_type;
        using KF = std::false_type;
        // one instantiation per (activation, keeps [gate | up], folded norm) in use: the inference step (SiLU, no keep) with and
        // without the norm, the training forward (keep, no norm); other activations without the norm
        auto by_act = [&](auto keepc, auto normc) {
            if (g.act == VGPT_ACT_SILU) gated_store(std::integral_constant<int, VGPT_ACT_SILU>{}, keepc, normc);
            else if (g.act == VGPT_ACT_GELU) gated_store(std::integral_constant<int, VGPT_ACT_GELU>{}, keepc, normc);
            else gated_store(std::integral_constant<int, VGPT_ACT_GELU_TANH>{}, keepc, normc);
        };
        if (g.gu_out) by_act(KT{}, KF{});
        else if (g.nrm_rstd != nullptr) by_act(KF{}, KT{});
        else by_act(KF{}, KF{});
    }
#ifdef VGPT_W4_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the epilogue's stores have left the CU
    if (g.dbg && lane == 0) g.dbg[(blockIdx.x * 4 + wave) * 8 + 7] = (uint32_t)__builtin_amdgcn_s_memrealtime();
#endif
}

// Persistent walk: OFF unless VGPT_GEMM_PERSIST=1.  Measured in round 3 on one box (bench.py, same process order): sampler
// step 31.996 ms with it against 32.03 without (gate_up 318 vs 322 us), stage-1 step 216.0 ms WITH it against 213.5 without --
// hardware dispatch already starts the next workgroup's prologue while other CUs store, and it balances the ragged last
// m-tile rows of the training shapes dynamically, which a static walk cannot.  Kept as a switch for later A/B runs.
bool persist_enabled() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("VGPT_GEMM_PERSIST");
        v = e ? (atoi(e) != 0) : 0;
    }
    return v != 0;
}

int cu_count() {
    static int v = 0;
    if (v == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
            v = n;
        else
            v = 256;
    }
    return v;
}

template <int MODE, typename C, int PIPE, bool ATR = false, bool WTR = false>
int launch_cfg(GemmArgs g, int64_t n_out, hipStream_t s, const char* name) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_kernel<MODE, C, PIPE, ATR, WTR>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES + C::BM * 4);
        if (e != hipSuccess) {
            vgpt_set_error("%s: hipFuncSetAttribute: %s", name, hipGetErrorString(e));
            return VGPT_ERR_HIP;
        }
        attr_set = true;
    }
    g.tiles_m = (int)cdiv(g.M, C::BM);
    g.tiles_n = (int)cdiv(n_out, MODE == MODE_GATED ? C::BN / 2 : C::BN);
    // persistent walk (kernel: PERSIST): one round of the chip's workgroup slots, each workgroup taking tiles
    // blockIdx.x, + gridDim.x, ...; the slot count is a multiple of 8 (XCD remap).  Off by default (persist_enabled()).
    int grid = g.tiles_m * g.tiles_n;
    if ((PIPE == 0 || PIPE == 1) && MODE != MODE_ROPE && persist_enabled() && g.nrm_rstd == nullptr) {
        const int slots = cu_count() * (C::LDS_BYTES > 80 * 1024 ? 1 : 2);
        if (slots % 8 == 0 && grid > slots) grid = slots;
    }
    hipLaunchKernelGGL((gemm_bf16_kernel<MODE, C, PIPE, ATR, WTR>), dim3(grid), dim3(C::THREADS),
                       C::LDS_BYTES + C::BM * 4, s, g);
    VGPT_CHECK_LAUNCH(name);
    return VGPT_OK;
}

// 0 = heuristic, 128 / 256 / 192 / 288 = forced tile, 257 / 289 = the 256- / 288-wide tile with the simple (non-pipelined) loop
// (VGPT_GEMM_TILE, read once; for A/B measurements)
int forced_tile() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("VGPT_GEMM_TILE");
        v = e ? atoi(e) : 0;
    }
    return v;
}

// Launch plan of a big-tile GEMM: the big-tile kernel (one block per CU, so T tiles take ceil(T/256) rounds) gets either
// everything or — when the last round would be badly filled — only the m-tile rows that make full rounds, and the
// 128x128 kernel (2 blocks/CU, 4x smaller tiles) runs the remaining rows behind it.
// Costs are in units of one 256-row big-tile ROUND.  A round takes the same time whatever the tile's width (1.57 us per
// k-tile with 256 x 192 tiles against 1.62 with 256 x 256: the loop waits on its operand stream, not on the MFMAs), so a
// narrower tile only pays where it removes a badly filled round; a round of the 128-tile kernel (512 tiles) measured at
// 0.62 of a big round (46 us against 75-78 at K = 3072).
struct BigPlan {
    int64_t rows_big;  // rows given to the big-tile kernel (M: no split)
    double cost;
};
// Cost of a launch plan in units of one k-tile of a 256x256 workgroup tile.  Measured at M = 4096 over K = 1024 .. 8192
// (scripts/gemm_k_sweep.py): a round of 256x256 tiles takes 1.0 per k-tile, a round of 256x192 tiles 0.77 (its 48 instead
// of 64 MFMAs per wave), and every round pays about 7 more for its prologue and epilogue (nothing overlaps them with one
// workgroup per CU); a round of the 128x128 kernel (two workgroups per CU) runs 0.62 of a 256x256 round.
BigPlan plan_big(int64_t M, int64_t n_out, int bn_out, int64_t nk) {
    constexpr int CUS = 256;
    constexpr double FIXED = 7.0;
    const double round_cost = (bn_out == 192 ? 0.77 : 1.0) * (double)nk + FIXED;
    const int64_t tiles_n = cdiv(n_out, bn_out), tiles_m = cdiv(M, 256), T = tiles_m * tiles_n;
    const int64_t full = T / CUS, rem = T % CUS;
    auto small = [&](int64_t rows) {
        return (double)cdiv(cdiv(rows, 128) * cdiv(n_out, 128), 2 * CUS) * 0.62 * ((double)nk + FIXED);
    };
    const double whole = (double)cdiv(T, CUS) * round_cost;
    if (full >= 1 && rem > 0 && rem < (CUS * 85) / 100) {
        const int64_t rows_big = (full * CUS) / tiles_n;
        if (rows_big >= 1 && rows_big < tiles_m) {
            const double split = (double)cdiv(rows_big * tiles_n, CUS) * round_cost + small(M - rows_big * 256);
            if (split < 0.97 * whole) return {rows_big * 256, split};
        }
    }
    return {M, whole};
}

// Which kernel family big NT products take: 0 = the four-wave kernel wherever w4_ok() (default), 1 = the eight-wave LDS-DMA
// kernels only (parity tests run both; same-box A/B).  Set through vgpt_gemm_set_family.
int g_family = 0;
bool w4_enabled() { return g_family == 0; }

template <int MODE, int NI, bool WTR = false, bool ATR = false>
int launch_w4_cfg(GemmArgs g, int64_t n_out, hipStream_t s, const char* name) {
    constexpr int LDS = (NI == 9 ? 2 * (256 + 288) * BK * 2 : 128 * 1024) + 1024;   // staging buffers + the 256 rstd values of the folded RMSNorm
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_w4_kernel<MODE, NI, WTR, ATR>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) {
            vgpt_set_error("%s: hipFuncSetAttribute: %s", name, hipGetErrorString(e));
            return VGPT_ERR_HIP;
        }
        attr_set = true;
    }
    constexpr int BN = NI * 32;
#ifdef VGPT_W4_STAMPS
    g.dbg = g_w4_dbg;
#endif
    g.tiles_m = (int)cdiv(g.M, 256);
    g.tiles_n = (int)cdiv(n_out, MODE == MODE_GATED ? BN / 2 : BN);
    hipLaunchKernelGGL((gemm_w4_kernel<MODE, NI, WTR, ATR>), dim3(g.tiles_m * g.tiles_n), dim3(256), LDS, s, g);
    VGPT_CHECK_LAUNCH(name);
    return VGPT_OK;
}

// Shapes the four-wave kernel takes: NT operands, at least two k-tiles, 8 rows / columns to clamp into, every byte offset the
// loop forms below 2 GiB (A: relative to the tile's first row; W: relative to the matrix for the row-permuting modes)
template <int MODE>
bool w4_ok(const GemmArgs& g, int64_t n_out) {
    const int64_t rows_w = MODE == MODE_GATED ? 2 * (int64_t)g.I : g.N;
    if (g.K % BK != 0 || g.K < 2 * BK || g.M < 8 || n_out < 8 || rows_w < 8) return false;
    if (MODE == MODE_GATED && g.I < 8) return false;
    if ((256 + 8) * g.lda * 2 + (int64_t)g.K * 2 >= (1ll << 31)) return false;
    if ((rows_w + 8) * g.ldw * 2 + (int64_t)g.K * 2 >= (1ll << 31)) return false;
    if (g.ldc >= (1 << 21) || g.ldr >= (1 << 21) || g.ld_gu >= (1 << 21)) return false;   // 256 rows x ld x 2 bytes below the descriptors' range
    if (MODE == MODE_ROPE) {   // the epilogue stages the tile's cos / sin rows through LDS (two tables of 256 x head_dim / 2 floats)
        const int tab_bytes = (256 * (g.head_dim >> 1) * 4 + 1023) & ~1023;
        if (2 * tab_bytes > 3 * 256 * BK * 2 || g.rope_cols % 16 != 0) return false;
    }
    return true;
}

// 256- or 192-wide tiles.  Cost of a round of 256 workgroups in units of one k-tile of a 256 x 256 round, fitted to in-step
// kernel times of round 4 (o_proj 67 us and down_proj 152 us: one round of 192-wide tiles at 48 / 128 k-tiles; gate_up 292 us:
// four rounds of 256-wide tiles at 48): a 256-wide k-tile 1.19 us, a 192-wide one 1.06 us (0.89 -- the loop is paced by the
// clock the part holds under the operand stream, not by its MFMA count alone), and 16 us = 13.5 k-tiles per round for
// dispatch, prologue and the epilogue's burst of stores.
template <int MODE>
void w4_costs(const GemmArgs& g, int64_t n_out, double& c256, double& c192) {
    const int64_t nk = g.K / BK, tiles_m = cdiv(g.M, 256), cus = cu_count();
    const int64_t t256 = tiles_m * cdiv(n_out, MODE == MODE_GATED ? 128 : 256), t192 = tiles_m * cdiv(n_out, MODE == MODE_GATED ? 96 : 192);
    const double FIXED = 13.5;
    c256 = (double)cdiv(t256, cus) * ((double)nk + FIXED);
    c192 = (double)cdiv(t192, cus) * (0.89 * (double)nk + FIXED);
}
// the cheaper of the two in units of one 256-wide round
template <int MODE>
double w4_cost_rounds(const GemmArgs& g, int64_t n_out) {
    double c256, c192;
    w4_costs<MODE>(g, n_out, c256, c192);
    return (c192 < c256 ? c192 : c256) / ((double)(g.K / BK) + 13.5);
}

// 256 x 288 tiles (NI = 9: whole tiles only, no gated mode): a k-tile is 144 instead of 128 MFMAs per wave
template <int MODE>
double w4_cost288(const GemmArgs& g, int64_t n_out) {
    if (MODE == MODE_GATED || g.M % 256 != 0 || n_out % 288 != 0) return 1e30;
    const int64_t t288 = (g.M / 256) * (n_out / 288);
    return (double)cdiv(t288, cu_count()) * (1.125 * (double)(g.K / BK) + 13.5);
}

template <int MODE>
int launch_w4(const GemmArgs& g, int64_t n_out, hipStream_t s, const char* name) {
    double c256, c192;
    w4_costs<MODE>(g, n_out, c256, c192);
    const double c288 = w4_cost288<MODE>(g, n_out);
    static int forced = -1;
    if (forced < 0) {
        const char* e = getenv("VGPT_GEMM_W4_NI");   // probes: force the tile width (8 = 256, 6 = 192, 9 = 288 where it applies)
        forced = e ? atoi(e) : 0;
    }
    if constexpr (MODE != MODE_GATED) {
        if (c288 < 1e29 && g.ssq_out == nullptr && (forced == 9 || (forced == 0 && c288 < c256 && c288 < c192)))
            return launch_w4_cfg<MODE, 9>(g, n_out, s, name);
    }
    const bool use192 = forced == 6 || (forced != 8 && c192 < c256);
    if (use192) return launch_w4_cfg<MODE, 6>(g, n_out, s, name);
    return launch_w4_cfg<MODE, 8>(g, n_out, s, name);
}

template <int MODE, bool ATR = false, bool WTR = false>
int launch(const GemmArgs& g, int64_t n_out, hipStream_t s, const char* name) {
    const int f = forced_tile();
    // the 256-tile pays off once the grid fills the chip (>= ~half of the 256 CUs with 256x256 tiles)
    const int64_t tiles_n = cdiv(n_out, MODE == MODE_GATED ? 128 : 256);
    const int64_t tiles_m = cdiv(g.M, 256);
    const int64_t big_tiles = tiles_m * tiles_n;
    const bool use256 = f == 256 || f == 257 || f == 192 || f == 288 || f == 289 || f == 512 || (f != 128 && big_tiles >= 128);
    if (!use256) return launch_cfg<MODE, Cfg128, 0, ATR, WTR>(g, n_out, s, name);
    if constexpr (!ATR && !WTR) {
        if (f == 0 && w4_enabled() && w4_ok<MODE>(g, n_out)) {
            return launch_w4<MODE>(g, n_out, s, name);
        }
    }
    if constexpr (!ATR && WTR && MODE == MODE_PLAIN) {
        // dX = dY W of the backward on the four-wave kernel (fragments of W by ds_read_b64_tr_b16): one launch for all rows --
        // the eight-wave plan splits a 7740-row product at a round boundary and re-streams W for the remainder
        if (f == 0 && w4_enabled() && g.K % BK == 0 && g.K >= 2 * BK && g.M >= 8 && n_out >= 8 &&
            (256 + 8) * g.lda * 2 + (int64_t)g.K * 2 < (1ll << 31) && ((int64_t)g.K + 64) * g.ldw * 2 < (1ll << 31) &&
            g.ldc < (1 << 21) && g.ldr < (1 << 21)) {
            double c256, c192;
            w4_costs<MODE>(g, n_out, c256, c192);
            if (c192 < c256) return launch_w4_cfg<MODE, 6, true>(g, n_out, s, name);
            return launch_w4_cfg<MODE, 8, true>(g, n_out, s, name);
        }
    }
    if constexpr (ATR && WTR && MODE == MODE_PLAIN) {
        // dW = dY^T X of the backward on the four-wave kernel: both images natural [64 reduction rows][256 columns], fragments
        // of both operands by ds_read_b64_tr_b16
        if (f == 0 && w4_enabled() && g.K >= 2 * BK && g.M >= 8 && n_out >= 8 && big_tiles >= 128 &&
            ((int64_t)g.K + 64) * g.lda * 2 < (1ll << 31) && ((int64_t)g.K + 64) * g.ldw * 2 < (1ll << 31) &&
            g.ldc < (1 << 21) && g.ldr < (1 << 21)) {
            double c256, c192;
            w4_costs<MODE>(g, n_out, c256, c192);
            if (c192 < c256) return launch_w4_cfg<MODE, 6, true, true>(g, n_out, s, name);
            return launch_w4_cfg<MODE, 8, true, true>(g, n_out, s, name);
        }
    }
    if (f == 257) return launch_cfg<MODE, Cfg256, 0, ATR, WTR>(g, n_out, s, name);
    if constexpr ((MODE == MODE_PLAIN || MODE == MODE_ROPE) && !ATR && !WTR) {
        if (f == 288) return launch_cfg<MODE, Cfg288, 5, ATR, WTR>(g, n_out, s, name);
        if (f == 289) return launch_cfg<MODE, Cfg288, 0, ATR, WTR>(g, n_out, s, name);
    }
    const int64_t nk = cdiv(g.K, BK);
    BigPlan p256 = plan_big(g.M, n_out, MODE == MODE_GATED ? 128 : 256, nk);
    if (f == 256) p256.rows_big = g.M;
    bool use192 = false;
    BigPlan p = p256;
    if constexpr ((MODE == MODE_PLAIN || MODE == MODE_ROPE) && !ATR && !WTR) {
        // 256 x 192 tiles when their rounds fit the problem better
        BigPlan p192 = plan_big(g.M, n_out, 192, nk);
        if (f == 192) p192.rows_big = g.M;
        use192 = f == 192 || (f == 0 && p192.cost < 0.985 * p256.cost);
        if (use192) p = p192;
    }
    // 256 x 288 tiles (a round costs ~1.3 of a 256 x 256 round at the same K -- 72 instead of 64 MFMAs per wave and k-tile;
    // 1.35 is the simple loop's figure, VGPT_GEMM_TILE=289) where they divide N and save enough rounds -- in practice
    // qkv_proj (N = 9216) of a 4096-row sampler step: two rounds against three of 192-wide tiles, 203 vs 225-230 us with
    // weights from HBM (scripts/gemm_epilogue_probe.py, same box)
    if constexpr ((MODE == MODE_PLAIN || MODE == MODE_ROPE) && !ATR && !WTR) {
        if (f == 0 && n_out % 288 == 0 && p.rows_big >= g.M) {
            const double r288 = (double)cdiv(tiles_m * (n_out / 288), 256) * 1.35;
            const double rcur = (double)cdiv(tiles_m * cdiv(n_out, use192 ? 192 : 256), 256);
            if (r288 < 0.95 * rcur) return launch_cfg<MODE, Cfg288, 5, ATR, WTR>(g, n_out, s, name);
        }
    }
    auto big = [&](const GemmArgs& ga) {
        if constexpr ((MODE == MODE_PLAIN || MODE == MODE_ROPE) && !ATR && !WTR) {
            if (use192) return launch_cfg<MODE, Cfg192, 1, ATR, WTR>(ga, n_out, s, name);
        }
        return launch_cfg<MODE, Cfg256, 1, ATR, WTR>(ga, n_out, s, name);
    };
    if (p.rows_big >= g.M) return big(g);
    GemmArgs g1 = g, g2 = g;
    const int64_t m1 = p.rows_big;
    g1.M = (int)m1;
    g2.M = g.M - (int)m1;
    g2.A = ATR ? g.A + m1 : g.A + m1 * g.lda;  // a transposed A keeps m along its columns
    g2.C = g.C + m1 * g.ldc;
    if (g.epi == VGPT_EPI_RESID) g2.extra = g.extra + m1 * g.ldr;
    if (g.gu_out) g2.gu_out = g.gu_out + m1 * g.ld_gu;
    if (g.nrm_rstd) g2.nrm_rstd = g.nrm_rstd + m1;
    if (MODE == MODE_ROPE) {
        g2.rope_cos = g.rope_cos + m1 * (g.head_dim / 2);
        g2.rope_sin = g.rope_sin + m1 * (g.head_dim / 2);
    }
    int rc = big(g1);
    if (rc != VGPT_OK) return rc;
    return launch_cfg<MODE, Cfg128, 0, ATR, WTR>(g2, n_out, s, name);
}

bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }


}  // namespace

#ifdef VGPT_W4_STAMPS
VGPT_EXPORT int vgpt_gemm_w4_debug_buffer(void* p) { g_w4_dbg = (uint32_t*)p; return 0; }
#endif

VGPT_EXPORT int vgpt_gemm_set_family(int family) {
    const int prev = g_family;
    if (family == 0 || family == 1) g_family = family;
    return prev;
}

/* ---- RMSNorm folded into the GEMMs around it ---- */
namespace {
// partial sums per row for this shape on the four-wave kernel; 0: not a shape it takes
int norm_partials(int64_t M, int64_t N, int64_t K) {
    if (M <= 0 || N <= 0 || K <= 0 || M >= (1 << 30) || N >= (1 << 30) || K >= (1 << 30)) return 0;
    GemmArgs g;
    g.M = (int)M; g.N = (int)N; g.K = (int)K;
    g.lda = K; g.ldw = K; g.ldc = N; g.ldr = N; g.I = 0;
    if (g_family != 0 || forced_tile() != 0 || cdiv(M, 256) * cdiv(N, 256) < 128 || !w4_ok<MODE_PLAIN>(g, N)) return 0;
    double c256, c192;
    w4_costs<MODE_PLAIN>(g, N, c256, c192);
    return 2 * (int)cdiv(N, c192 < c256 ? 192 : 256);
}
int64_t norm_cnt_bytes(int64_t M) { return (cdiv(M, 256) * 4 + 255) / 256 * 256; }
}  // namespace

// bytes of workspace vgpt_gemm_bf16_resid_rstd needs for this shape; 0: the shape is not one the four-wave kernel takes (the
// caller keeps the separate RMSNorm)
VGPT_EXPORT int64_t vgpt_gemm_norm_workspace_bytes(int64_t M, int64_t N, int64_t K) {
    const int parts = norm_partials(M, N, K);
    return parts == 0 ? 0 : norm_cnt_bytes(M) + (int64_t)parts * M * 4;
}

VGPT_EXPORT int vgpt_gemm_bf16_resid_rstd(const void* A, const void* W, void* C, const void* resid, float* rstd_out, void* workspace,
                                          int64_t workspace_bytes, float eps, int64_t M, int64_t N, int64_t K, int64_t lda,
                                          int64_t ldw, int64_t ldc, int64_t ldr, void* stream) {
    VGPT_REQUIRE(A && W && C && resid && rstd_out && workspace, VGPT_ERR_INVALID, "vgpt_gemm_bf16_resid_rstd: null pointer");
    VGPT_REQUIRE(M > 0 && N > 0 && K > 0 && K % BK == 0 && N % 4 == 0 && ldc % 4 == 0 && ldr % 4 == 0 && lda % 8 == 0 && ldw % 8 == 0 &&
                     aligned16(A) && aligned16(W) && ((uintptr_t)C & 7) == 0 && ((uintptr_t)resid & 7) == 0 && eps >= 0.f,
                 VGPT_ERR_UNSUPPORTED, "vgpt_gemm_bf16_resid_rstd: shape / alignment as vgpt_gemm_bf16");
    const int64_t need = vgpt_gemm_norm_workspace_bytes(M, N, K);
    VGPT_REQUIRE(need > 0, VGPT_ERR_UNSUPPORTED,
                 "vgpt_gemm_bf16_resid_rstd: not a shape of the four-wave kernel (vgpt_gemm_norm_workspace_bytes)");
    VGPT_REQUIRE(workspace_bytes >= need && ((uintptr_t)workspace & 255) == 0, VGPT_ERR_INVALID,
                 "vgpt_gemm_bf16_resid_rstd: workspace too small or not 256-byte aligned");
    GemmArgs g;
    g.A = (const bf16*)A; g.W = (const bf16*)W; g.C = (bf16*)C; g.extra = (const bf16*)resid;
    g.M = (int)M; g.N = (int)N; g.K = (int)K;
    g.lda = lda; g.ldw = ldw; g.ldc = ldc; g.ldr = ldr;
    g.epi = VGPT_EPI_RESID; g.act = VGPT_ACT_NONE; g.I = 0;
    g.tiles_m = g.tiles_n = 0;
    g.rope_cos = g.rope_sin = nullptr; g.rope_cols = g.head_dim = 0;
    g.ssq_cnt = (int*)workspace;
    g.ssq_out = (float*)((char*)workspace + norm_cnt_bytes(M));
    g.rstd_out = rstd_out; g.nrm_ld = M; g.nrm_eps = eps; g.nrm_inv_h = 1.0f / (float)N;
    return launch_w4<MODE_PLAIN>(g, N, (hipStream_t)stream, "vgpt_gemm_bf16_resid_rstd");
}

VGPT_EXPORT int vgpt_gemm_bf16(const void* A, const void* W, void* C, const void* extra, int64_t M,
                               int64_t N, int64_t K, int64_t lda, int64_t ldw, int64_t ldc,
                               int64_t ldr, int epilogue, void* stream) {
    VGPT_REQUIRE(A && W && C, VGPT_ERR_INVALID, "vgpt_gemm_bf16: null pointer");
    VGPT_REQUIRE(M >= 0 && N > 0 && K > 0, VGPT_ERR_INVALID, "vgpt_gemm_bf16: bad shape");
    VGPT_REQUIRE(epilogue == VGPT_EPI_NONE || epilogue == VGPT_EPI_RESID || epilogue == VGPT_EPI_BIAS,
                 VGPT_ERR_INVALID, "vgpt_gemm_bf16: unknown epilogue %d", epilogue);
    VGPT_REQUIRE(epilogue == VGPT_EPI_NONE || extra, VGPT_ERR_INVALID,
                 "vgpt_gemm_bf16: epilogue needs `extra`");
    VGPT_REQUIRE(K % BK == 0, VGPT_ERR_UNSUPPORTED, "vgpt_gemm_bf16: K=%ld not a multiple of 64",
                 (long)K);
    VGPT_REQUIRE(N % 4 == 0 && ldc % 4 == 0 && (epilogue != VGPT_EPI_RESID || ldr % 4 == 0),
                 VGPT_ERR_UNSUPPORTED, "vgpt_gemm_bf16: N/ldc/ldr must be multiples of 4");
    VGPT_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && aligned16(A) && aligned16(W) &&
                     ((uintptr_t)C & 7) == 0,
                 VGPT_ERR_UNSUPPORTED, "vgpt_gemm_bf16: operands must be 16-byte aligned rows");
    VGPT_REQUIRE(M < (1 << 30) && N < (1 << 30) && K < (1 << 30), VGPT_ERR_UNSUPPORTED,
                 "vgpt_gemm_bf16: dimension too large");
    if (M == 0) return VGPT_OK;
    GemmArgs g;
    g.A = (const bf16*)A; g.W = (const bf16*)W; g.C = (bf16*)C; g.extra = (const bf16*)extra;
    g.M = (int)M; g.N = (int)N; g.K = (int)K;
    g.lda = lda; g.ldw = ldw; g.ldc = ldc; g.ldr = ldr;
    g.epi = epilogue; g.act = VGPT_ACT_NONE; g.I = 0;
    g.tiles_m = g.tiles_n = 0;
    g.rope_cos = g.rope_sin = nullptr; g.rope_cols = g.head_dim = 0;
    return launch<MODE_PLAIN>(g, N, (hipStream_t)stream, "vgpt_gemm_bf16");
}

VGPT_EXPORT int vgpt_gemm_bf16_tr(const void* A, const void* W, void* C, const void* extra, int64_t M, int64_t N, int64_t K,
                                  int64_t lda, int64_t ldw, int64_t ldc, int64_t ldr, int epilogue, int a_transposed,
                                  int w_transposed, void* stream) {
    if (!a_transposed && !w_transposed)
        return vgpt_gemm_bf16(A, W, C, extra, M, N, K, lda, ldw, ldc, ldr, epilogue, stream);
    VGPT_REQUIRE(A && W && C, VGPT_ERR_INVALID, "vgpt_gemm_bf16_tr: null pointer");
    VGPT_REQUIRE(M >= 0 && N > 0 && K > 0, VGPT_ERR_INVALID, "vgpt_gemm_bf16_tr: bad shape");
    VGPT_REQUIRE(epilogue == VGPT_EPI_NONE || epilogue == VGPT_EPI_RESID || epilogue == VGPT_EPI_BIAS, VGPT_ERR_INVALID,
                 "vgpt_gemm_bf16_tr: unknown epilogue %d", epilogue);
    VGPT_REQUIRE(epilogue == VGPT_EPI_NONE || extra, VGPT_ERR_INVALID, "vgpt_gemm_bf16_tr: epilogue needs `extra`");
    VGPT_REQUIRE(a_transposed || K % BK == 0, VGPT_ERR_UNSUPPORTED,
                 "vgpt_gemm_bf16_tr: K=%ld must be a multiple of 64 unless both operands are transposed", (long)K);
    VGPT_REQUIRE(!a_transposed || w_transposed, VGPT_ERR_UNSUPPORTED,
                 "vgpt_gemm_bf16_tr: a transposed A needs a transposed W (dW = dY^T X)");
    VGPT_REQUIRE(N % 8 == 0 && N >= 8 && (!a_transposed || (M % 8 == 0 && M >= 8)), VGPT_ERR_UNSUPPORTED,
                 "vgpt_gemm_bf16_tr: the width of a transposed operand must be a multiple of 8");
    VGPT_REQUIRE(ldc % 4 == 0 && (epilogue != VGPT_EPI_RESID || ldr % 4 == 0), VGPT_ERR_UNSUPPORTED,
                 "vgpt_gemm_bf16_tr: ldc/ldr must be multiples of 4");
    VGPT_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && aligned16(A) && aligned16(W) && ((uintptr_t)C & 7) == 0,
                 VGPT_ERR_UNSUPPORTED, "vgpt_gemm_bf16_tr: operands must be 16-byte aligned rows");
    VGPT_REQUIRE(M < (1 << 30) && N < (1 << 30) && K < (1 << 30) && K * ldw < (1ll << 30) &&
                     (!a_transposed || K * lda < (1ll << 30)),
                 VGPT_ERR_UNSUPPORTED, "vgpt_gemm_bf16_tr: dimension too large (a transposed operand must stay below 2 GiB)");
    if (M == 0) return VGPT_OK;
    GemmArgs g;
    g.A = (const bf16*)A; g.W = (const bf16*)W; g.C = (bf16*)C; g.extra = (const bf16*)extra;
    g.M = (int)M; g.N = (int)N; g.K = (int)K;
    g.lda = lda; g.ldw = ldw; g.ldc = ldc; g.ldr = ldr;
    g.epi = epilogue; g.act = VGPT_ACT_NONE; g.I = 0;
    g.tiles_m = g.tiles_n = 0;
    g.rope_cos = g.rope_sin = nullptr; g.rope_cols = g.head_dim = 0;
    if (a_transposed) return launch<MODE_PLAIN, true, true>(g, N, (hipStream_t)stream, "vgpt_gemm_bf16_tr");
    return launch<MODE_PLAIN, false, true>(g, N, (hipStream_t)stream, "vgpt_gemm_bf16_tr");
}

static int gated_mlp_impl(const void* A, const void* W_gate_up, void* out, void* gate_up_out, int64_t M, int64_t I, int64_t K,
                          int64_t lda, int64_t ldw, int64_t ldo, int64_t ld_gu, int act, void* stream,
                          const float* nrm_rstd = nullptr);

VGPT_EXPORT int vgpt_gated_mlp_act_fwd(const void* A, const void* W_gate_up, void* out, int64_t M,
                                       int64_t I, int64_t K, int64_t lda, int64_t ldw, int64_t ldo,
                                       int act, void* stream) {
    return gated_mlp_impl(A, W_gate_up, out, nullptr, M, I, K, lda, ldw, ldo, 0, act, stream);
}

VGPT_EXPORT int vgpt_gated_mlp_act_fwd_keep(const void* A, const void* W_gate_up, void* out, void* gate_up_out, int64_t M,
                                            int64_t I, int64_t K, int64_t lda, int64_t ldw, int64_t ldo, int64_t ld_gu,
                                            int act, void* stream) {
    VGPT_REQUIRE(gate_up_out && ld_gu >= 2 * I && ld_gu % 4 == 0 && ((uintptr_t)gate_up_out & 7) == 0, VGPT_ERR_INVALID,
                 "vgpt_gated_mlp_act_fwd_keep: gate_up_out must be an 8-byte aligned (M, >= 2I) buffer, ld_gu a multiple of 4");
    return gated_mlp_impl(A, W_gate_up, out, gate_up_out, M, I, K, lda, ldw, ldo, ld_gu, act, stream);
}

VGPT_EXPORT int vgpt_gated_mlp_act_fwd_prenorm(const void* A, const void* W_gate_up, void* out, const float* rstd, int64_t M,
                                               int64_t I, int64_t K, int64_t lda, int64_t ldw, int64_t ldo, int act, void* stream) {
    VGPT_REQUIRE(rstd, VGPT_ERR_INVALID, "vgpt_gated_mlp_act_fwd_prenorm: needs the rows' 1 / rms");
    return gated_mlp_impl(A, W_gate_up, out, nullptr, M, I, K, lda, ldw, ldo, 0, act, stream, rstd);
}

static int gated_mlp_impl(const void* A, const void* W_gate_up, void* out, void* gate_up_out, int64_t M, int64_t I, int64_t K,
                          int64_t lda, int64_t ldw, int64_t ldo, int64_t ld_gu, int act, void* stream, const float* nrm_rstd) {
    VGPT_REQUIRE(A && W_gate_up && out, VGPT_ERR_INVALID, "vgpt_gated_mlp_act_fwd: null pointer");
    VGPT_REQUIRE(M >= 0 && I > 0 && K > 0, VGPT_ERR_INVALID, "vgpt_gated_mlp_act_fwd: bad shape");
    VGPT_REQUIRE(act >= VGPT_ACT_SILU && act <= VGPT_ACT_GELU_TANH, VGPT_ERR_INVALID,
                 "vgpt_gated_mlp_act_fwd: unknown activation %d", act);
    VGPT_REQUIRE(K % BK == 0, VGPT_ERR_UNSUPPORTED,
                 "vgpt_gated_mlp_act_fwd: K=%ld not a multiple of 64", (long)K);
    VGPT_REQUIRE(I % 16 == 0 && ldo % 4 == 0, VGPT_ERR_UNSUPPORTED,
                 "vgpt_gated_mlp_act_fwd: I must be a multiple of 16, ldo of 4");
    VGPT_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && aligned16(A) && aligned16(W_gate_up) &&
                     ((uintptr_t)out & 7) == 0,
                 VGPT_ERR_UNSUPPORTED, "vgpt_gated_mlp_act_fwd: operands must be 16-byte aligned rows");
    VGPT_REQUIRE(M < (1 << 30) && I < (1 << 29) && K < (1 << 30), VGPT_ERR_UNSUPPORTED,
                 "vgpt_gated_mlp_act_fwd: dimension too large");
    if (M == 0) return VGPT_OK;
    GemmArgs g;
    g.A = (const bf16*)A; g.W = (const bf16*)W_gate_up; g.C = (bf16*)out; g.extra = nullptr;
    g.M = (int)M; g.N = (int)I; g.K = (int)K;
    g.lda = lda; g.ldw = ldw; g.ldc = ldo; g.ldr = 0;
    g.epi = VGPT_EPI_NONE; g.act = act; g.I = (int)I;
    g.tiles_m = g.tiles_n = 0;
    g.rope_cos = g.rope_sin = nullptr; g.rope_cols = g.head_dim = 0;
    g.gu_out = (bf16*)gate_up_out; g.ld_gu = ld_gu;
    g.nrm_rstd = nrm_rstd;
    return launch<MODE_GATED>(g, I, (hipStream_t)stream, "vgpt_gated_mlp_act_fwd");
}

static int gemm_rope_impl(const void* A, const void* W, void* C, const float* cos_t, const float* sin_t, int64_t M, int64_t N,
                         int64_t K, int64_t lda, int64_t ldw, int64_t ldc, int n_rot_heads, int head_dim, void* stream,
                         const float* nrm_rstd) {
    VGPT_REQUIRE(M >= 0 && N > 0 && K > 0 && n_rot_heads > 0 && head_dim > 0, VGPT_ERR_INVALID,
                 "vgpt_gemm_bf16_rope: bad shape");
    VGPT_REQUIRE(M == 0 || (A && W && C && cos_t && sin_t), VGPT_ERR_INVALID, "vgpt_gemm_bf16_rope: null pointer");
    VGPT_REQUIRE(head_dim % 16 == 0 && (int64_t)n_rot_heads * head_dim <= N, VGPT_ERR_UNSUPPORTED,
                 "vgpt_gemm_bf16_rope: head_dim=%d must be a multiple of 16 and the rotated heads must fit in N", head_dim);
    VGPT_REQUIRE(K % BK == 0, VGPT_ERR_UNSUPPORTED, "vgpt_gemm_bf16_rope: K=%ld not a multiple of 64", (long)K);
    VGPT_REQUIRE(N % 16 == 0 && ldc % 4 == 0, VGPT_ERR_UNSUPPORTED, "vgpt_gemm_bf16_rope: N must be a multiple of 16, ldc of 4");
    VGPT_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && aligned16(A) && aligned16(W) && ((uintptr_t)C & 7) == 0 &&
                     aligned16(cos_t) && aligned16(sin_t),
                 VGPT_ERR_UNSUPPORTED, "vgpt_gemm_bf16_rope: operands must be 16-byte aligned rows");
    VGPT_REQUIRE(M < (1 << 30) && N < (1 << 30) && K < (1 << 30), VGPT_ERR_UNSUPPORTED,
                 "vgpt_gemm_bf16_rope: dimension too large");
    if (M == 0) return VGPT_OK;
    GemmArgs g;
    g.A = (const bf16*)A; g.W = (const bf16*)W; g.C = (bf16*)C; g.extra = nullptr;
    g.M = (int)M; g.N = (int)N; g.K = (int)K;
    g.lda = lda; g.ldw = ldw; g.ldc = ldc; g.ldr = 0;
    g.epi = VGPT_EPI_NONE; g.act = VGPT_ACT_NONE; g.I = 0;
    g.tiles_m = g.tiles_n = 0;
    g.rope_cos = cos_t; g.rope_sin = sin_t; g.rope_cols = n_rot_heads * head_dim; g.head_dim = head_dim;
    g.nrm_rstd = nrm_rstd;
    return launch<MODE_ROPE>(g, N, (hipStream_t)stream, "vgpt_gemm_bf16_rope");
}

VGPT_EXPORT int vgpt_gemm_bf16_rope(const void* A, const void* W, void* C, const float* cos_t, const float* sin_t,
                                    int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw, int64_t ldc,
                                    int n_rot_heads, int head_dim, void* stream) {
    return gemm_rope_impl(A, W, C, cos_t, sin_t, M, N, K, lda, ldw, ldc, n_rot_heads, head_dim, stream, nullptr);
}

VGPT_EXPORT int vgpt_gemm_bf16_rope_prenorm(const void* A, const void* W, void* C, const float* cos_t, const float* sin_t,
                                            const float* rstd, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
                                            int64_t ldc, int n_rot_heads, int head_dim, void* stream) {
    VGPT_REQUIRE(rstd, VGPT_ERR_INVALID, "vgpt_gemm_bf16_rope_prenorm: needs the rows' 1 / rms");
    return gemm_rope_impl(A, W, C, cos_t, sin_t, M, N, K, lda, ldw, ldc, n_rot_heads, head_dim, stream, rstd);
}

