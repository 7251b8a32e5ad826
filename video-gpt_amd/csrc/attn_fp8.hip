// Block-masked attention forward with MX-fp8 (e4m3 + E8M0 block scales) operands on the block-scaled MFMA
// (v_mfma_scale_f32_32x32x64_f8f6f4: twice the bf16 rate), head_dim 96 -- the inference-only "fp8 attention" option of
// the cfg-5 rollout.  Same operator as attn_fwd.hip (module.local_attn = SDPA with the additive block
// mask, LVM/transform/sdpa_transform.py:78-86,152; mask of LVM/processor.py:575-731 bit-packed by mask.hip); what
// differs is the arithmetic: Q, K, V and the probabilities P are rounded to e4m3 (3 mantissa bits) in blocks of 32 that
// share a power-of-two scale; scores, softmax statistics and the output accumulator stay fp32.  Tolerance against the
// fp32 oracle is therefore that of fp8 operands (tests/test_attn_fp8_gpu.py states it), not the bf16 path's.
//
// Operand map of the instruction with e4m3 data (scripts/probes/mfma_fp8_layout.hip, checked with exact integers on an
// MI355X): lane l (r = l & 31, h = l >> 5) supplies 32 bytes; bytes 0..15 are k = 16 h + j of row / column r, bytes 16..31
// are k = 32 + 16 h + j.  The E8M0 scale in byte 0 of lane (r, 0)'s scale register applies to k = 0..31 of row r (bytes
// 0..15 of BOTH lane halves), lane (r, 1)'s to k = 32..63.  C/D as every 32x32 MFMA: column = r, rows (i&3)+8(i>>2)+4h.
//
// Two kernels:
//   attn_fp8_quantize_kernel   fused (B, L, .) bf16 Q/K/V (RoPE applied) -> a workspace holding
//        Q8   (B, n_heads, L, 96) e4m3, pre-multiplied by scale * log2(e);  QS (B, n_heads, L, 4) scales of its 3 d-blocks
//        KV   (B, n_kv_heads, ceil(L/64)) records of REC bytes, one per 64-key tile, copied to LDS as they are:
//             K8 [key][96] | KS [key][4] | V8 [d tile][h][d][32] in the P^T operand's key order | VS [d tile][key half][d]
//   attn_fwd_fp8_kernel        planned launch (the work items / summaries / order of vgpt_attn_plan_build): four waves of
//        32 query rows, S^T = K Q^T (2 MFMAs per 32 keys: d 0..63 and d 64..95 + zeros), online softmax in the log2
//        domain, P^T taken straight from the accumulators as the B operand of O^T += V^T P^T (3 MFMAs per 64 keys).
#include "common.h"

namespace {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int D = 96;
constexpr int K8_BYTES = 64 * D;              // 6144
constexpr int KS_OFF = K8_BYTES;              // 64 keys x 4 scale bytes
constexpr int V8_OFF = KS_OFF + 256;          // 6400
constexpr int VS_OFF = V8_OFF + 3 * 2 * 32 * 32;   // 12544
constexpr int REC = 13 * 1024;                // 13312: one tile record, whole 1-KiB DMA pieces
constexpr int P_SCALE_EXP = 8;                // probabilities are stored as p * 2^8 (<= 256 < 448) with block scale 2^-8
static_assert(VS_OFF + 192 <= REC, "record layout");

// power-of-two block scale: smallest e with amax * 2^-e <= 448 (e4m3 maximum); returns the E8M0 byte and 2^-e
__device__ __forceinline__ int block_scale(float amax, float& inv) {
    int e = 0;
    if (amax > 0.f) {
        (void)frexpf(amax * (1.0f / 448.0f), &e);   // amax / 448 = m 2^e with m in [0.5, 1)
        e = max(-120, min(e, 120));
    }
    inv = __builtin_ldexpf(1.0f, -e);
    return e + 127;
}
__device__ __forceinline__ uint32_t pack4_fp8(float a, float b, float c, float d) {
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (uint32_t)w;
}

struct QuantArgs {
    const bf16* q; const bf16* k; const bf16* v;
    uint8_t* q8; uint8_t* qs; uint8_t* kv;
    int B, L, n_heads, n_kv_heads, nkt;
    int kt0;   // first key tile / 64-row query block to (re)quantise: earlier rows keep what the workspace holds
    int64_t q_sb, q_sh, q_ss, k_sb, k_sh, k_ss, v_sb, v_sh, v_ss;
    float q_mul;   // softmax scale * log2(e), folded into Q before rounding
};

// grid.x < B * n_kv_heads * nkt: one K/V tile record; the remaining blocks: 64 query rows of one head each
__global__ __launch_bounds__(256) void attn_fp8_quantize_kernel(QuantArgs a) {
    const int tid = threadIdx.x;
    const int nt = a.nkt - a.kt0;   // tiles / row blocks this launch covers
    const int n_rec = a.B * a.n_kv_heads * nt;
    if ((int)blockIdx.x < n_rec) {
        const int kt = a.kt0 + blockIdx.x % nt, kvh = (blockIdx.x / nt) % a.n_kv_heads, b = blockIdx.x / (nt * a.n_kv_heads);
        uint8_t* rec = a.kv + (((int64_t)b * a.n_kv_heads + kvh) * a.nkt + kt) * REC;
        // the V tile (64 keys x 96 d) goes through LDS: it is read by columns below, and from global memory that would be
        // 2-byte accesses a row stride apart
        __shared__ __attribute__((aligned(16))) bf16 vt[64 * D];
#pragma unroll
        for (int c = tid; c < 64 * 12; c += 256) {
            const int key = c / 12, part = c % 12, r2 = kt * 64 + key;
            bf16x8 t = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (r2 < a.L) t = *reinterpret_cast<const bf16x8*>(a.v + (int64_t)b * a.v_sb + (int64_t)kvh * a.v_sh + (int64_t)r2 * a.v_ss + part * 8);
            *reinterpret_cast<bf16x8*>(vt + key * D + part * 8) = t;
        }
        __syncthreads();
        if (tid < 192) {
            // K: thread = (key, block of 32 d)
            const int key = tid / 3, blk = tid % 3, row = kt * 64 + key;
            float x[32];
            float amax = 0.f;
            if (row < a.L) {
                const bf16* p = a.k + (int64_t)b * a.k_sb + (int64_t)kvh * a.k_sh + (int64_t)row * a.k_ss + blk * 32;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const bf16x8 t = *reinterpret_cast<const bf16x8*>(p + 8 * c);
#pragma unroll
                    for (int j = 0; j < 8; ++j) { x[8 * c + j] = bf2f(t[j]); amax = fmaxf(amax, fabsf(x[8 * c + j])); }
                }
            } else {
#pragma unroll
                for (int j = 0; j < 32; ++j) x[j] = 0.f;
            }
            float inv;
            const int sc = block_scale(amax, inv);
            uint32_t w[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) w[c] = pack4_fp8(x[4 * c] * inv, x[4 * c + 1] * inv, x[4 * c + 2] * inv, x[4 * c + 3] * inv);
            uint4* dst = reinterpret_cast<uint4*>(rec + key * D + blk * 32);
            dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
            dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
            rec[KS_OFF + key * 4 + blk] = (uint8_t)sc;
            if (blk == 0) rec[KS_OFF + key * 4 + 3] = 127;
            // V: thread = (d tile, key half, d): the 32 keys of that half, column d
            const int dt = tid / 64, kb = (tid >> 5) & 1, d = tid & 31;
            amax = 0.f;
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) {
                x[kk] = bf2f(vt[(kb * 32 + kk) * D + dt * 32 + d]);
                amax = fmaxf(amax, fabsf(x[kk]));
            }
            const int sv = block_scale(amax, inv);
            // key kk of the half sits in lane half h = (kk >> 2) & 1 at byte kb * 16 + (kk & 3) + 4 (kk >> 3): the order in
            // which the S^T accumulators hand the probabilities to the P^T operand
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const int h = g & 1, j16 = 4 * (g >> 1);
                *reinterpret_cast<uint32_t*>(rec + V8_OFF + ((dt * 2 + h) * 32 + d) * 32 + kb * 16 + j16) =
                    pack4_fp8(x[4 * g] * inv, x[4 * g + 1] * inv, x[4 * g + 2] * inv, x[4 * g + 3] * inv);
            }
            rec[VS_OFF + (dt * 2 + kb) * 32 + d] = (uint8_t)sv;
        }
        return;
    }
    // Q: block = 64 rows of one (batch, head); thread = (row, block of 32 d) for tid < 192
    if (tid >= 192) return;
    const int qb = blockIdx.x - n_rec;
    const int rb = a.kt0 + qb % nt, head = (qb / nt) % a.n_heads, b = qb / (nt * a.n_heads);
    const int row = rb * 64 + tid / 3, blk = tid % 3;
    if (row >= a.L) return;
    const bf16* p = a.q + (int64_t)b * a.q_sb + (int64_t)head * a.q_sh + (int64_t)row * a.q_ss + blk * 32;
    float x[32];
    float amax = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const bf16x8 t = *reinterpret_cast<const bf16x8*>(p + 8 * c);
#pragma unroll
        for (int j = 0; j < 8; ++j) { x[8 * c + j] = bf2f(t[j]) * a.q_mul; amax = fmaxf(amax, fabsf(x[8 * c + j])); }
    }
    float inv;
    const int sc = block_scale(amax, inv);
    uint32_t w[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) w[c] = pack4_fp8(x[4 * c] * inv, x[4 * c + 1] * inv, x[4 * c + 2] * inv, x[4 * c + 3] * inv);
    const int64_t qrow = ((int64_t)b * a.n_heads + head) * a.L + row;
    uint4* dst = reinterpret_cast<uint4*>(a.q8 + qrow * D + blk * 32);
    dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
    dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
    a.qs[qrow * 4 + blk] = (uint8_t)sc;
    if (blk == 0) a.qs[qrow * 4 + 3] = 127;
}

struct Fp8Args {
    const uint8_t* q8; const uint8_t* qs; const uint8_t* kv;
    bf16* o;
    const uint32_t* bits;
    const int32_t* items;      // n_items x 4: batch, row0, nrows (<= 128), 0
    const uint16_t* isum;      // n_items x nkt: 2 bits per 32-row slab
    const int32_t* order;
    int n_items, L, n_heads, kv_group, n_kv_heads, W, nkt;
    int64_t o_sb, o_sh, o_ss;
};

// LDS-DMA from inline asm (the compiler must not see LDS being written: it would drain vmcnt(0) before later LDS reads)
__device__ __forceinline__ void dma16(const uint8_t* base, uint32_t off, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(off), "s"(base), "s"(lds_dst)
                 : "memory");
}

__global__ __launch_bounds__(256, 2) void attn_fwd_fp8_kernel(Fp8Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 x REC
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    int rank, head;
    if ((a.n_heads & 7) == 0) {   // XCD x (blockIdx & 7) owns n_heads / 8 consecutive heads: their K/V records stay in its L2
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
    const uint8_t* rec0 = a.kv + ((int64_t)b * a.n_kv_heads + kvh) * a.nkt * REC;
    const bool wave_live = row0 + wave * 32 <= row_last;
    const int q_row = row0 + wave * 32 + r;
    const uint32_t lds_base =
        __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem);

    // ---- Q fragments (B operand of S^T = K Q^T): d 16h.. and 32+16h.. | d 64+16h.. and zeros ----
    v8i qa, qb;
    int sq1, sq2;
    {
        const int64_t qrow = ((int64_t)b * a.n_heads + head) * a.L + min(q_row, row_last);
        const uint8_t* qp = a.q8 + qrow * D;
        const v4i x0 = *reinterpret_cast<const v4i*>(qp + 16 * h), x1 = *reinterpret_cast<const v4i*>(qp + 32 + 16 * h),
                  x2 = *reinterpret_cast<const v4i*>(qp + 64 + 16 * h);
        qa = v8i{x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
        qb = v8i{x2[0], x2[1], x2[2], x2[3], 0, 0, 0, 0};
        const uint32_t s4 = *reinterpret_cast<const uint32_t*>(a.qs + qrow * 4);
        sq1 = (int)((s4 >> (8 * h)) & 0xff);
        sq2 = h ? 127 : (int)((s4 >> 16) & 0xff);
    }
    const uint32_t* mrow = a.bits + ((int64_t)b * a.L + min(q_row, row_last)) * a.W;

    f32x16 O[3];
#pragma unroll
    for (int dt = 0; dt < 3; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) O[dt][i] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    auto next_active = [&](int kt) {   // first tile >= kt any slab of the item sees, or nkt
        while (kt < a.nkt && sum16[kt] == 0) ++kt;
        return kt;
    };
    auto stage = [&](int buf, int kt) {
        const uint8_t* src = rec0 + (int64_t)kt * REC;
        for (int p = wave; p < REC / 1024; p += 4)
            dma16(src, (uint32_t)(p * 1024 + lane * 16), lds_base + (uint32_t)(buf * REC + p * 1024));
    };

    int kt = next_active(0), buf = 0;
    if (kt < a.nkt) stage(0, kt);
    while (kt < a.nkt) {
        const int nxt = next_active(kt + 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // tile kt landed in `buf` for every wave; the other buffer is free
        if (nxt < a.nkt) stage(buf ^ 1, nxt);
        const int code = (sum16[kt] >> (2 * wave)) & 3;
        if (wave_live && code != 0) {
            const char* t = smem + buf * REC;
            f32x16 S[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const int key = kb * 32 + r;
                const v4i k0 = *reinterpret_cast<const v4i*>(t + key * D + 16 * h), k1 = *reinterpret_cast<const v4i*>(t + key * D + 32 + 16 * h),
                          k2 = *reinterpret_cast<const v4i*>(t + key * D + 64 + 16 * h);
                const uint32_t s4 = *reinterpret_cast<const uint32_t*>(t + KS_OFF + key * 4);
                const int sk1 = (int)((s4 >> (8 * h)) & 0xff), sk2 = h ? 127 : (int)((s4 >> 16) & 0xff);
                const v8i ka = v8i{k0[0], k0[1], k0[2], k0[3], k1[0], k1[1], k1[2], k1[3]};
                const v8i kc = v8i{k2[0], k2[1], k2[2], k2[3], 0, 0, 0, 0};
                f32x16 z;
#pragma unroll
                for (int i = 0; i < 16; ++i) z[i] = 0.f;
                z = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ka, qa, z, 0, 0, 0, sk1, 0, sq1);
                S[kb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kc, qb, z, 0, 0, 0, sk2, 0, sq2);
            }
            if (code == 2) {   // mixed tile: this row's two mask words (keys 64 kt .. +63)
                const uint32_t w0 = mrow[min(2 * kt, a.W - 1)], w1 = 2 * kt + 1 < a.W ? mrow[2 * kt + 1] : 0u;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    const uint32_t w = (kb ? w1 : w0) >> (4 * h);
#pragma unroll
                    for (int i = 0; i < 16; ++i) S[kb][i] = ((w >> ((i & 3) + 8 * (i >> 2))) & 1u) ? S[kb][i] : -INFINITY;
                }
            }
            float mx = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) mx = fmaxf(mx, S[kb][i]);
            const float m_new = fmaxf(m_run, half_max(mx));
            const float mu = m_new == -INFINITY ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m_run - mu);
            // probabilities carry the operand's 2^8 (p' = 2^8 p = exp2(s - (m - 8))): the row sum accumulates p' as well,
            // so the factor cancels in O / l, and the MFMA's block scale 2^-8 returns O to the plain sum of p v
            const float mu8 = mu - (float)P_SCALE_EXP;
            float rs = 0.f;
            v8i pf;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float p4[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        p4[u] = __builtin_amdgcn_exp2f(S[kb][4 * g + u] - mu8);
                        rs += p4[u];
                    }
                    pf[kb * 4 + g] = (int)pack4_fp8(p4[0], p4[1], p4[2], p4[3]);
                }
            l_run = l_run * alpha + rs;
            m_run = m_new;
            // (a wave-uniform "no maximum moved" branch around this rescale costs 20-25 registers and with them the fourth
            // wave per SIMD: 141 instead of 131 us per cfg-2 layer)
#pragma unroll
            for (int dt = 0; dt < 3; ++dt) {
#pragma unroll
                for (int i = 0; i < 16; ++i) O[dt][i] *= alpha;
                const v4i v0 = *reinterpret_cast<const v4i*>(t + V8_OFF + ((dt * 2 + h) * 32 + r) * 32),
                          v1 = *reinterpret_cast<const v4i*>(t + V8_OFF + ((dt * 2 + h) * 32 + r) * 32 + 16);
                const v8i vf = v8i{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                const int sv = (int)(uint8_t)t[VS_OFF + (dt * 2 + h) * 32 + r];
                O[dt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vf, pf, O[dt], 0, 0, 0, sv, 0, 127 - P_SCALE_EXP);
            }
        }
        buf ^= 1;
        kt = nxt;
    }
    if (!wave_live || q_row > row_last) return;
    const float l_t = half_sum(l_run);   // = 2^8 x the sum of probabilities
    const float inv = l_t > 0.f ? (float)(1 << P_SCALE_EXP) / l_t : 0.f;
    bf16* op = a.o + (int64_t)b * a.o_sb + (int64_t)head * a.o_sh + (int64_t)q_row * a.o_ss;
#pragma unroll
    for (int dt = 0; dt < 3; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            bf16x4 o;
#pragma unroll
            for (int u = 0; u < 4; ++u) o[u] = f2bf(O[dt][4 * g4 + u] * inv);
            *reinterpret_cast<bf16x4*>(op + dt * 32 + 8 * g4 + 4 * h) = o;
        }
}

int64_t align256(int64_t v) { return (v + 255) / 256 * 256; }

}  // namespace

VGPT_EXPORT int64_t vgpt_attn_fp8_workspace_bytes(int64_t B, int64_t L, int n_heads, int n_kv_heads, int head_dim) {
    if (B <= 0 || L <= 0 || n_heads <= 0 || n_kv_heads <= 0 || head_dim != D) return -1;
    return align256(B * n_heads * L * D) + align256(B * n_heads * L * 4) + B * n_kv_heads * cdiv(L, 64) * REC;
}

VGPT_EXPORT int vgpt_attn_fp8_quantize(const void* q, const void* k, const void* v, void* workspace, int64_t B, int64_t L,
                                       int64_t row_begin, int n_heads, int n_kv_heads, int head_dim, int64_t q_sb,
                                       int64_t q_sh, int64_t q_ss, int64_t k_sb, int64_t k_sh, int64_t k_ss, int64_t v_sb,
                                       int64_t v_sh, int64_t v_ss, float scale, void* stream) {
    VGPT_REQUIRE(q && k && v && workspace, VGPT_ERR_INVALID, "vgpt_attn_fp8_quantize: null pointer");
    VGPT_REQUIRE(head_dim == D, VGPT_ERR_UNSUPPORTED, "vgpt_attn_fp8_quantize: head_dim must be 96 (got %d)", head_dim);
    VGPT_REQUIRE(B > 0 && L > 0 && L <= (1 << 22) && n_heads > 0 && n_kv_heads > 0 && n_heads % n_kv_heads == 0,
                 VGPT_ERR_INVALID, "vgpt_attn_fp8_quantize: bad shape");
    VGPT_REQUIRE(scale > 0.f, VGPT_ERR_INVALID, "vgpt_attn_fp8_quantize: scale must be positive");
    VGPT_REQUIRE(row_begin >= 0 && row_begin <= L && row_begin % 64 == 0, VGPT_ERR_INVALID,
                 "vgpt_attn_fp8_quantize: row_begin must be a multiple of 64 in [0, L]");
    if (row_begin >= L) return VGPT_OK;
    const int64_t strides[] = {q_sb, q_sh, q_ss, k_sb, k_sh, k_ss, v_sb, v_sh, v_ss};
    for (int64_t st : strides)
        VGPT_REQUIRE(st % 8 == 0, VGPT_ERR_UNSUPPORTED, "vgpt_attn_fp8_quantize: q/k/v strides must be multiples of 8 elements");
    VGPT_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)workspace) & 15) == 0, VGPT_ERR_UNSUPPORTED,
                 "vgpt_attn_fp8_quantize: q, k, v and the workspace must be 16-byte aligned");
    const int64_t nkt = cdiv(L, 64), nt = nkt - row_begin / 64;
    const int64_t n_rec = B * n_kv_heads * nt, n_q = B * n_heads * nt;
    VGPT_REQUIRE(n_rec + n_q < (1ll << 31), VGPT_ERR_UNSUPPORTED, "vgpt_attn_fp8_quantize: problem too large");
    QuantArgs a;
    a.q = (const bf16*)q; a.k = (const bf16*)k; a.v = (const bf16*)v;
    a.q8 = (uint8_t*)workspace;
    a.qs = a.q8 + align256(B * n_heads * L * D);
    a.kv = a.qs + align256(B * n_heads * L * 4);
    a.B = (int)B; a.L = (int)L; a.n_heads = n_heads; a.n_kv_heads = n_kv_heads; a.nkt = (int)nkt; a.kt0 = (int)(row_begin / 64);
    a.q_sb = q_sb; a.q_sh = q_sh; a.q_ss = q_ss; a.k_sb = k_sb; a.k_sh = k_sh; a.k_ss = k_ss;
    a.v_sb = v_sb; a.v_sh = v_sh; a.v_ss = v_ss;
    a.q_mul = scale * 1.4426950408889634f;
    hipLaunchKernelGGL(attn_fp8_quantize_kernel, dim3((unsigned)(n_rec + n_q)), dim3(256), 0, (hipStream_t)stream, a);
    VGPT_CHECK_LAUNCH("vgpt_attn_fp8_quantize");
    return VGPT_OK;
}

VGPT_EXPORT int vgpt_attn_fwd_plan_fp8(const void* workspace, void* o, const uint32_t* bits, const int32_t* items,
                                       const uint16_t* item_summary, const int32_t* order, int64_t n_items, int64_t B,
                                       int64_t L, int n_heads, int n_kv_heads, int head_dim, int64_t o_sb, int64_t o_sh,
                                       int64_t o_ss, void* stream) {
    VGPT_REQUIRE(workspace && o && bits && items && item_summary && order, VGPT_ERR_INVALID, "vgpt_attn_fwd_plan_fp8: null pointer");
    VGPT_REQUIRE(head_dim == D, VGPT_ERR_UNSUPPORTED, "vgpt_attn_fwd_plan_fp8: head_dim must be 96 (got %d)", head_dim);
    VGPT_REQUIRE(B > 0 && L > 0 && L <= (1 << 22) && n_items >= 0 && n_items < 65536 && n_heads > 0 && n_kv_heads > 0 &&
                     n_heads % n_kv_heads == 0,
                 VGPT_ERR_INVALID, "vgpt_attn_fwd_plan_fp8: bad shape");
    VGPT_REQUIRE(o_sb % 4 == 0 && o_sh % 4 == 0 && o_ss % 4 == 0 && ((uintptr_t)o & 7) == 0, VGPT_ERR_UNSUPPORTED,
                 "vgpt_attn_fwd_plan_fp8: o strides must be multiples of 4 elements, o 8-byte aligned");
    if (n_items == 0) return VGPT_OK;
    Fp8Args a;
    a.q8 = (const uint8_t*)workspace;
    a.qs = a.q8 + align256(B * n_heads * L * D);
    a.kv = a.qs + align256(B * n_heads * L * 4);
    a.o = (bf16*)o; a.bits = bits; a.items = items; a.isum = item_summary; a.order = order;
    a.n_items = (int)n_items; a.L = (int)L; a.n_heads = n_heads; a.kv_group = n_heads / n_kv_heads; a.n_kv_heads = n_kv_heads;
    a.W = (int)cdiv(L, 32); a.nkt = (int)cdiv(L, 64);
    a.o_sb = o_sb; a.o_sh = o_sh; a.o_ss = o_ss;
    hipLaunchKernelGGL(attn_fwd_fp8_kernel, dim3((unsigned)(n_items * n_heads)), dim3(256), 2 * REC, (hipStream_t)stream, a);
    VGPT_CHECK_LAUNCH("vgpt_attn_fwd_plan_fp8");
    return VGPT_OK;
}
