/*
 * vgpt.h — C ABI of libvgpt_hip.so: the MI355X (gfx950) kernels behind the
 * next-clip diffusion hot path of Video-GPT.
 *
 * Every entry point replaces one piece of arithmetic the reference executes
 * through PyTorch ops; the reference site each one stands in for is cited as
 * `path:line` relative to the reference checkout (third-party pins:
 * transformers==4.47.1 Phi3 blocks, diffusers==0.29.0 AutoencoderKL).
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless a parameter is named host_*;
 *   - outputs and workspaces are caller-allocated; nothing here allocates,
 *     frees or synchronises, so every call is capturable into a hipGraph;
 *   - `stream` is a hipStream_t passed as void*;
 *   - return 0 on success, <0 on failure (VGPT_ERR_*); the message of the last
 *     failure on the calling thread is returned by vgpt_last_error();
 *   - bf16 tensors are raw 16-bit bfloat16; "row-major" means last dim contiguous.
 */
#ifndef VGPT_H
#define VGPT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VGPT_OK 0
#define VGPT_ERR_INVALID (-1)     /* bad argument (null pointer, negative size, misaligned) */
#define VGPT_ERR_UNSUPPORTED (-2) /* shape outside what the kernel was built for */
#define VGPT_ERR_HIP (-3)         /* HIP runtime error at launch */

/* activation of the gated MLP (Phi3MLP: down(up * act(gate)); config.hidden_act) */
#define VGPT_ACT_SILU 0
#define VGPT_ACT_GELU 1      /* erf form  */
#define VGPT_ACT_GELU_TANH 2 /* tanh form */
#define VGPT_ACT_NONE 3

/* epilogues of vgpt_gemm_bf16 */
#define VGPT_EPI_NONE 0  /* C = A W^T                       */
#define VGPT_EPI_RESID 1 /* C = A W^T + R   (residual add)  */
#define VGPT_EPI_BIAS 2  /* C = A W^T + bias[n]             */

/* prediction type of the sampler (LVM/scheduler.py:178) */
#define VGPT_PRED_V 0
#define VGPT_PRED_X1 1

const char* vgpt_last_error(void);
/* VGPT_ABI_VERSION of the library that was loaded.  Bumped with EVERY change of an exported signature; a binding written
 * for another value must refuse to call (video-gpt_amd/_lib.py does): with shifted arguments a stale library would read a
 * stream pointer as a scale and fault on the device instead of failing cleanly. */
#define VGPT_ABI_VERSION 5
int vgpt_abi_version(void);

/* ---- transformer block -------------------------------------------------- */

/* Phi3RMSNorm (transformers 4.47.1 modeling_phi3.py Phi3RMSNorm.forward; call
 * sites OmniGen/transformer.py:196-214): y = bf16(bf16(x * rsqrt(mean(x^2)+eps)) * w),
 * fp32 accumulation. x,y: (rows, H) bf16 row-major; w: (H) bf16. H % 8 == 0. */
int vgpt_rmsnorm_fwd(const void* x, const void* w, void* y, int64_t rows, int64_t H, float eps,
                     void* stream);

/* Phi3RotaryEmbedding.forward (cos/sin table): cos/sin[t][i] = scale * f(pos[t] * inv_freq[i]),
 * i < half; rounded to bf16 and stored back as fp32 when round_bf16 != 0 (the
 * reference casts cos/sin to the model dtype, LVM/transform/sdpa_transform.py:52).  scale = 1 for
 * plain RoPE; for rope_scaling "su"/"longrope" checkpoints the caller passes the rescaled inv_freq
 * (short / long factors) and the attention factor as `scale` (HF Phi3LongRoPEScaledRotaryEmbedding). */
int vgpt_rope_table(const int64_t* position_ids, const float* inv_freq, float* cos_out,
                    float* sin_out, int64_t tokens, int half, int round_bf16, float scale, void* stream);

/* apply_rotary_pos_emb on the q and k parts of a fused qkv buffer, in place
 * (LVM/transform/sdpa_transform.py:53). qkv: (tokens, (n_q+2*n_kv)*hd) bf16;
 * cos,sin: (tokens, hd/2) fp32. (hd/2) % 8 == 0. */
int vgpt_rope_qk_inplace(void* qkv, const float* cos_t, const float* sin_t, int64_t tokens,
                         int n_q_heads, int n_kv_heads, int head_dim, void* stream);

/* nn.Linear without bias on MFMA: C[M,N] = A[M,K] W[N,K]^T (+ epilogue), bf16 in,
 * fp32 accumulate, bf16 out. Replaces qkv_proj / o_proj / down_proj
 * (LVM/transform/sdpa_transform.py:39,89; Phi3MLP.down_proj). K % 64 == 0, N % 4 == 0.
 * lda/ldw/ldc/ldr are row strides in elements. `extra` is R (M,N) for EPI_RESID,
 * bias (N) for EPI_BIAS, ignored for EPI_NONE. */
int vgpt_gemm_bf16(const void* A, const void* W, void* C, const void* extra, int64_t M, int64_t N,
                   int64_t K, int64_t lda, int64_t ldw, int64_t ldc, int64_t ldr, int epilogue,
                   void* stream);

/* qkv_proj followed by apply_rotary_pos_emb in one kernel (LVM/transform/sdpa_transform.py:39,52-53): C = A W^T rounded
 * to bf16 (the Linear's output), then columns [0, n_rot_heads * head_dim) -- the q heads followed by the k heads --
 * rotated per token: out[d] = x[d] cos[d] - x[d + hd/2] sin[d], out[d + hd/2] = x[d + hd/2] cos[d] + x[d] sin[d] with
 * cos / sin (M, head_dim / 2) fp32 from vgpt_rope_table (row m = token m of A); the remaining columns (v) are stored
 * unrotated.  Same result as vgpt_gemm_bf16 + vgpt_rope_qk_inplace without writing and re-reading q and k.
 * N % 16 == 0, head_dim % 16 == 0, K % 64 == 0. */
int vgpt_gemm_bf16_rope(const void* A, const void* W, void* C, const float* cos_t, const float* sin_t, int64_t M,
                        int64_t N, int64_t K, int64_t lda, int64_t ldw, int64_t ldc, int n_rot_heads, int head_dim,
                        void* stream);
/* The same product with operands stored transposed, for the backward of the Linear layers without materialising a
 * transpose (the reference gets these from torch.autograd: dX = dY W, dW = dY^T X):
 *   w_transposed: W is stored (K, N) row-major (ldw = its row stride)   -> dX[M,N'] = dY[M,K'] W[K',N']
 *   a_transposed: A is stored (K, M) row-major; needs w_transposed      -> dW[N,K'] = dY^T X with A = dY, W = X
 * Widths of transposed operands must be multiples of 8; K % 64 == 0 unless BOTH operands are transposed (then the
 * rows of a partial last k-tile are zero-filled by the hardware's buffer range check).  A transposed operand must
 * stay below 2 GiB.  With both flags 0 this is vgpt_gemm_bf16. */
int vgpt_gemm_bf16_tr(const void* A, const void* W, void* C, const void* extra, int64_t M, int64_t N, int64_t K,
                      int64_t lda, int64_t ldw, int64_t ldc, int64_t ldr, int epilogue, int a_transposed,
                      int w_transposed, void* stream);

/* Kernel family of the big NT products behind vgpt_gemm_bf16 / vgpt_gemm_bf16_rope / vgpt_gated_mlp_act_fwd[_keep]:
 * 0 (default) = the four-wave kernel with the hand-scheduled register-staged loop wherever it applies (K a multiple of 64
 * with at least two k-tiles, 128 or more 256-row tiles), 1 = the eight-wave LDS-DMA kernels only.  Same operand contract and
 * epilogues; results differ by the order of the fp32 additions inside a k-tile.  Process-wide, not thread-safe: a test and
 * measurement switch, set before the launches it is meant for.  Returns the previous value; other values change nothing. */
int vgpt_gemm_set_family(int family);

/* RMSNorm folded into the GEMMs around it (a decoder layer's two Phi3RMSNorm calls, OmniGen/transformer.py:196-214 through
 * transformers' Phi3DecoderLayer: hidden = residual + attn(input_layernorm(hidden)); hidden = residual +
 * mlp(post_attention_layernorm(hidden))).  The PRODUCER of a residual stream (o_proj / down_proj + residual) also leaves
 * 1 / rms of every output row behind; the CONSUMER (qkv_proj + RoPE, gate_up + activation) reads the raw residual stream as its
 * A operand and a weight with the norm's gain folded in, and multiplies its fp32 accumulators by the row's 1 / rms before
 * anything else:
 *      norm(x) W^T = rstd(x) . (x (W . diag(gain))^T)
 * -- two launches and one activation round trip less per norm.  Differences from the separate kernel: the gain meets the weight
 * (one bf16 rounding of gain * W) instead of the normalised activation (two roundings), and rstd multiplies an fp32 sum; the
 * model-level tolerances (tests/golden/tolerance_calibration.json) hold unchanged.  Deterministic: every workgroup stores the
 * sum of squares of ITS columns of a row, and the last workgroup to arrive at a 256-row block's counter adds the block's
 * partial sums up in a fixed order (no floating-point atomics).
 *   vgpt_gemm_norm_workspace_bytes(M, N, K): bytes of workspace vgpt_gemm_bf16_resid_rstd needs for this shape, 0 if the shape
 *       is not one it takes (then the caller keeps vgpt_rmsnorm_fwd).  The workspace is 256-byte aligned, ZEROED ONCE by the
 *       caller before its first use (it holds arrival counters that every launch leaves at zero again) and may be shared by
 *       all such launches of one stream;
 *   vgpt_gemm_bf16_resid_rstd: C = A W^T + resid (as vgpt_gemm_bf16 with VGPT_EPI_RESID; C may be resid), and
 *       rstd_out[m] = rsqrt(mean_n C[m, n]^2 + eps) on the rounded values;
 *   vgpt_rms_rstd: the same statistic of a matrix x (M, H) no GEMM here produced;
 *   vgpt_fold_norm_gain: W_out (N, K) = bf16(W[n][k] * gain[k]);
 *   vgpt_gemm_bf16_rope_prenorm / vgpt_gated_mlp_act_fwd_prenorm: vgpt_gemm_bf16_rope / vgpt_gated_mlp_act_fwd on
 *       A = the raw stream, W = the folded weight, row m of the product scaled by rstd[m]. */
int64_t vgpt_gemm_norm_workspace_bytes(int64_t M, int64_t N, int64_t K);
int vgpt_gemm_bf16_resid_rstd(const void* A, const void* W, void* C, const void* resid, float* rstd_out, void* workspace,
                              int64_t workspace_bytes, float eps, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
                              int64_t ldc, int64_t ldr, void* stream);
int vgpt_rms_rstd(const void* x, float* rstd_out, int64_t rows, int64_t H, int64_t ldx, float eps, void* stream);
int vgpt_fold_norm_gain(const void* W, const void* gain, void* W_out, int64_t N, int64_t K, void* stream);
int vgpt_gemm_bf16_rope_prenorm(const void* A, const void* W, void* C, const float* cos_t, const float* sin_t, const float* rstd,
                                int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw, int64_t ldc, int n_rot_heads,
                                int head_dim, void* stream);
int vgpt_gated_mlp_act_fwd_prenorm(const void* A, const void* W_gate_up, void* out, const float* rstd, int64_t M, int64_t I,
                                   int64_t K, int64_t lda, int64_t ldw, int64_t ldo, int act, void* stream);

/* Phi3MLP first half, fused: out[M,I] = act(A Wg^T) * (A Wu^T) where
 * W_gate_up (2I, K) = [Wg ; Wu] as stored by Phi3MLP.gate_up_proj.
 * K % 64 == 0, I % 64 == 0. */
int vgpt_gated_mlp_act_fwd(const void* A, const void* W_gate_up, void* out, int64_t M, int64_t I,
                           int64_t K, int64_t lda, int64_t ldw, int64_t ldo, int act, void* stream);
/* The same product for the TRAINING forward (LVMTraining.forward through Phi3MLP, LVM/model.py:752-845): additionally stores
 * gate_up_out (M, ld_gu >= 2I) = [A Wg^T | A Wu^T] rounded to bf16 -- gate_up_proj's own output, which the backward reads --
 * and computes out = act(gate) * up FROM those rounded values: bit for bit vgpt_gemm_bf16 followed by vgpt_silu_mul_fwd,
 * without writing and re-reading the (M, 2I) tensor in between. */
int vgpt_gated_mlp_act_fwd_keep(const void* A, const void* W_gate_up, void* out, void* gate_up_out, int64_t M, int64_t I,
                                int64_t K, int64_t lda, int64_t ldw, int64_t ldo, int64_t ld_gu, int act, void* stream);

/* ---- block-masked attention ---------------------------------------------- */

/* (B,L,L) bool/uint8 mask (1 = visible; LVM/processor.py:682-731) -> bit-packed rows
 * bits[b][q][w] (w < W = ceil(L/32)); bit j of word w = mask[b][q][32w+j]; bits past L are 0. */
int vgpt_mask_pack_bool(const uint8_t* mask, uint32_t* bits, int64_t B, int64_t L, void* stream);
/* Same from the additive float mask the reference hands to SDPA
 * (OmniGen/transformer.py:139-145: 0 visible, finfo.min masked). is_f32: 0 bf16, 1 fp32.
 * mask is (B,1,L,L) contiguous. */
int vgpt_mask_pack_additive(const void* mask, int is_f32, uint32_t* bits, int64_t B, int64_t L,
                            void* stream);
/* The same packed rows generated from per-token attributes, without any (B,L,L) tensor: replaces the host mask
 * painters LVM/processor.py:575-616 (stage 1), :618-680 (frame-block training), :682-731 (next-clip inference) — all
 * three follow one rule over token attributes (video-gpt_amd/layout.py).
 * attr: (B, L, 2) int32, 8-byte aligned; per token t of row b
 *   word 0 = low | seq << 24     low (24 bits): thr of a CLEAN token = first query row that sees this key; sub of a NOISY
 *                                token = sub-group inside its clip (0 everywhere in the collator's layouts; the engine
 *                                numbers the time rows of denoise step s with s + 1 in its per-clip pass); seq: sequence
 *                                id (8 bits)
 *   word 1 = kind | oc << 2 | grp << 4   kind 0 PAD, 1 CLEAN, 2 NOISY, 3 GAP; oc = min(offset in frame block, 2);
 *                                        grp = id of the (sequence, clip) a NOISY token belongs to
 * bit(q,k) = kind[q]==PAD  ||  (kind[k]==CLEAN && seq[q]==seq[k] && q >= thr[k])
 *         ||  (kind[k]==NOISY && kind[q]==NOISY && grp[q]==grp[k] && oc[q] >= oc[k] && (sub[k]==0 || sub[k]==sub[q])). */
int vgpt_mask_build_tokens(const int32_t* attr, uint32_t* bits, int64_t B, int64_t L, void* stream);
/* Tile summary: one byte per (b, 128-row q block, 64-key tile): 2 bits per 32-row
 * q sub-block (0 = all masked, 1 = all visible, 2 = mixed).
 * summary: (B, ceil(L/128), ceil(L/64)) uint8. */
int vgpt_mask_tile_summary(const uint32_t* bits, uint8_t* summary, int64_t B, int64_t L,
                           void* stream);
/* Number of query rows with no visible key (the reference would average over all keys
 * there; this library does not support it). count: 1 x int32, zeroed by the callee. */
int vgpt_mask_count_empty_rows(const uint32_t* bits, int32_t* count, int64_t B, int64_t L,
                               void* stream);

/* softmax(q k^T * scale + mask) v with the mask given as bits + tile summary.
 * Replaces module.local_attn (F.scaled_dot_product_attention,
 * LVM/transform/sdpa_transform.py:78-86,152). q,k,v,o are bf16 with element strides
 * (batch, head, seq); the head_dim axis is contiguous. head_dim == 96 or 128... see
 * vgpt_attn_supported(). n_kv_heads must divide n_heads (repeat_kv).
 * variant: 0 = default (hardware transposed LDS reads), 1 = reference-slow path. */
int vgpt_attn_blockmask_fwd(const void* q, const void* k, const void* v, void* o,
                            const uint32_t* bits, const uint8_t* summary, int64_t B, int64_t L,
                            int n_heads, int n_kv_heads, int head_dim, int64_t q_sb, int64_t q_sh,
                            int64_t q_ss, int64_t k_sb, int64_t k_sh, int64_t k_ss, int64_t v_sb,
                            int64_t v_sh, int64_t v_ss, int64_t o_sb, int64_t o_sh, int64_t o_ss,
                            float scale, int variant, void* stream);
int vgpt_attn_supported(int head_dim);
/* Same, computing only query rows [q_start, L) (q_start % 128 == 0) against ALL L keys: rows before q_start are a
 * cached, step-invariant key/value prefix (condition frames never see the clip being denoised, LVM/processor.py:
 * 682-731, so their K/V need not be recomputed every denoise step as LVM/scheduler.py:174 does).  q, o and the mask
 * are indexed by absolute row.  order: NULL, or the launch order computed by vgpt_attn_qblock_order for the same
 * (summary, q_start): block masks give q blocks very different key counts, and starting the long ones first keeps
 * the launch from ending on a few stragglers (results do not depend on it). */
int vgpt_attn_blockmask_fwd_qrange(const void* q, const void* k, const void* v, void* o, int64_t q_start,
                                   const uint32_t* bits, const uint8_t* summary, const int32_t* order, int64_t B, int64_t L,
                                   int n_heads,
                                   int n_kv_heads, int head_dim, int64_t q_sb, int64_t q_sh, int64_t q_ss, int64_t k_sb,
                                   int64_t k_sh, int64_t k_ss, int64_t v_sb, int64_t v_sh, int64_t v_ss, int64_t o_sb,
                                   int64_t o_sh, int64_t o_ss, float scale, void* stream);

/* ---- planned forward.  A PLAN describes the query side of one mask:
 *   items         (n_items, 4) int32: {batch, row0, nrows (1..item_rows), 0}; disjoint row ranges, cut wherever the
 *                 caller likes -- at the boundaries of packed sequences, so that no item mixes rows with different key
 *                 sets (an aligned block straddling two sequences would walk the key tiles of both);
 *   item_summary  (n_items, ceil(L/64)) uint16: 2 bits per 32-row slab of the item (0 none / 1 all / 2 mixed);
 *   order         (n_items) int32: items sorted longest first.
 * vgpt_attn_plan_build fills item_summary and order from bits and items; both live in ONE caller-allocated workspace of
 * vgpt_attn_plan_workspace_bytes(L, n_items) bytes: item_summary at its start, order at the next multiple of 256 bytes
 * behind n_items * ceil(L/64) * 2.  vgpt_attn_fwd_plan computes exactly the rows the items cover (same result as
 * vgpt_attn_blockmask_fwd up to the rounding of the online softmax); lse may be NULL.  item_rows = the upper bound of the
 * items' row counts, 128 or 256: 256 (head_dim 96 only) runs workgroups of 8 waves on 256 query rows, so one K / V tile
 * staged in LDS serves twice the rows (half the LDS-DMA instructions per wave and half the L2 -> LDS bytes per FLOP). */
int64_t vgpt_attn_plan_workspace_bytes(int64_t L, int64_t n_items);
int vgpt_attn_plan_build(const uint32_t* bits, int64_t B, int64_t L, const int32_t* items, int64_t n_items,
                         uint16_t* item_summary, int32_t* order, void* stream);
int vgpt_attn_fwd_plan(const void* q, const void* k, const void* v, void* o, float* lse, const uint32_t* bits,
                       const int32_t* items, const uint16_t* item_summary, const int32_t* order, int64_t n_items,
                       int64_t B, int64_t L, int n_heads, int n_kv_heads, int head_dim, int64_t q_sb,
                       int64_t q_sh, int64_t q_ss, int64_t k_sb, int64_t k_sh, int64_t k_ss, int64_t v_sb, int64_t v_sh,
                       int64_t v_ss, int64_t o_sb, int64_t o_sh, int64_t o_ss, float scale, int item_rows, void* stream);

/* ---- MX-fp8 attention (inference option of the cfg-5 rollout; head_dim 96).  Same operator as
 * vgpt_attn_fwd_plan with Q, K, V and the probabilities rounded to OCP e4m3 in blocks of 32 sharing a power-of-two (E8M0)
 * scale, products on the block-scaled MFMA (2x the bf16 rate), fp32 scores / statistics / accumulator.  Two calls:
 * vgpt_attn_fp8_quantize turns the (RoPE-rotated) q / k / v views into `workspace`
 * (vgpt_attn_fp8_workspace_bytes(B, L, n_heads, n_kv_heads, head_dim) bytes, 256-byte aligned; the softmax scale is
 * folded into Q there; only rows >= row_begin, a multiple of 64, are (re)written: a cached prefix keeps its bytes),
 * vgpt_attn_fwd_plan_fp8 computes the rows of a plan (items / item_summary / order of
 * vgpt_attn_plan_build) from it.  No LSE output: the training path stays bf16.  Replaces the same reference lines as
 * vgpt_attn_blockmask_fwd (LVM/transform/sdpa_transform.py:78-86,152). */
int64_t vgpt_attn_fp8_workspace_bytes(int64_t B, int64_t L, int n_heads, int n_kv_heads, int head_dim);
int vgpt_attn_fp8_quantize(const void* q, const void* k, const void* v, void* workspace, int64_t B, int64_t L,
                           int64_t row_begin, int n_heads, int n_kv_heads, int head_dim, int64_t q_sb, int64_t q_sh,
                           int64_t q_ss, int64_t k_sb, int64_t k_sh, int64_t k_ss, int64_t v_sb, int64_t v_sh, int64_t v_ss,
                           float scale, void* stream);
int vgpt_attn_fwd_plan_fp8(const void* workspace, void* o, const uint32_t* bits, const int32_t* items,
                           const uint16_t* item_summary, const int32_t* order, int64_t n_items, int64_t B, int64_t L,
                           int n_heads, int n_kv_heads, int head_dim, int64_t o_sb, int64_t o_sh, int64_t o_ss, void* stream);

/* order (B, ceil(L/128) - q_start/128) int32 <- the 128-row q blocks [q_start/128, ...) of every batch item, the one
 * with the most non-empty 64-key tiles first (stable). */
int vgpt_attn_qblock_order(const uint8_t* summary, int64_t B, int64_t L, int64_t q_start, int32_t* order, void* stream);

/* Diagnostics: while `buf` (device memory, 4 x uint64 per workgroup) is set, every forward launch with at most
 * `capacity_workgroups` workgroups records {start, end (100 MHz realtime ticks), XCC_ID<<32|HW_ID, work item<<32 |
 * key tiles processed} per workgroup.  buf = NULL switches it off (the default). */
int vgpt_attn_trace(void* buf, int64_t capacity_workgroups);

/* Tile body of the head-dim-96 attention forward (every vgpt_attn_* forward entry above): 1 (default) = the generated,
 * hand-scheduled, software-pipelined bodies (csrc/gen/attn_p2_gen.py), 0 = the compiler-scheduled body.  Outputs and log-sum-exp
 * are bit-identical.  Process-wide, not thread-safe: a test and measurement switch (the environment variable VGPT_ATTN_P2=0
 * sets the initial value, read once).  Returns the previous value; other values change nothing. */
int vgpt_attn_set_hand_scheduled(int on);

/* ---- model glue ---------------------------------------------------------- */

/* embed_tokens gather (LVM/model.py:430-432): out[r] = table[ids[r]]. H % 8 == 0. */
int vgpt_embed_gather(const int64_t* ids, const void* table, void* out, int64_t rows, int64_t H,
                      int64_t vocab, void* stream);

/* PatchEmbedMR + cropped 2-D sincos position embedding, scattered into the token
 * sequence (LVM/model.py:149-154,268-289,299-306,436-453).
 * x: (n_frames, C, h, w) bf16; Wp: (H, C*p*p) bf16; bias: (H) bf16;
 * pos_embed: (pos_max*pos_max, H) bf16; dst_row[f]: first row of frame f in `seq`
 * ((rows,H) bf16). patch p = 2, C*p*p == 16. */
int vgpt_patch_embed_fwd(const void* x, const void* Wp, const void* bias, const void* pos_embed,
                         const int32_t* dst_row, void* seq, int n_frames, int C, int h, int w,
                         int64_t H, int pos_max, void* stream);

/* TimestepEmbedder.timestep_embedding (LVM/model.py:39-58): out[n] = [cos(t f) | sin(t f)]
 * (dim = 2*half) rounded to bf16. freqs: (half) fp32 computed by the host exactly as
 * the reference does. t: (n) fp32. */
int vgpt_timestep_sinusoid(const float* t, const float* freqs, void* out, int n, int half,
                           void* stream);

/* Small-M Linear: out[m][n] = post(sum_k pre(x[m][k]) W[n][k] + bias[n]), M <= 32.
 * pre/post: VGPT_ACT_NONE or VGPT_ACT_SILU. Replaces the TimestepEmbedder MLP
 * (LVM/model.py:32-36) and FinalLayer.adaLN_modulation (LVM/model.py:74-77).
 * out rows may be scattered: row m is written at out + out_row[m]*ldo when out_row != NULL
 * (time tokens scattered into the sequence, LVM/model.py:442-446). K % 8 == 0. */
int vgpt_linear_small(const void* x, const void* W, const void* bias, void* out,
                      const int32_t* out_row, int M, int64_t N, int64_t K, int64_t ldx, int64_t ldo,
                      int pre_act, int post_act, void* stream);

/* FinalLayer (LayerNorm no-affine eps + modulate + Linear(H -> p*p*C)) + unpatchify
 * (LVM/model.py:79-83,255-265,478-486). hidden: (rows,H) bf16; src_row[f] first row of
 * frame f; mod: (n_frames, 2H) bf16 = [shift | scale]; Wf: (p*p*C, H) bf16; bf: (p*p*C);
 * out: (n_frames, C, h, w) bf16. p = 2, p*p*C == 16. */
int vgpt_final_layer_fwd(const void* hidden, const int32_t* src_row, const void* mod,
                         const void* Wf, const void* bf, void* out, int n_frames, int C, int h,
                         int w, int64_t H, float eps, void* stream);

/* ---- sampler (LVM/scheduler.py:161-208) ---------------------------------- */

/* timesteps[i] = sigma[*step] for i < n (scheduler.py:169).  sigma holds n_steps + 1 entries; with *step outside
 * [0, n_steps] nothing is written (a graph replayed past the end of the table must not read beyond it). */
int vgpt_sampler_set_timesteps(const float* sigma, const int32_t* step, int n_steps, float* timesteps, int n,
                               void* stream);
/* One Euler step with optional x1->v conversion and image CFG (scheduler.py:178-204).
 * z: (n_frames, elems) fp32 state, z_model: same shape bf16 copy handed to the model,
 * pred: (n_frames, elems) bf16. With use_cfg the first half of the frames is the
 * conditional branch, the second half the unconditional one.  sigma holds n_steps + 1 entries; with *step outside
 * [0, n_steps) the state is left untouched. */
int vgpt_euler_cfg_update(float* z, void* z_model, const void* pred, const float* sigma,
                          const int32_t* step, int n_steps, int n_frames, int64_t elems, int pred_type,
                          int use_cfg, float cfg_scale, void* stream);
/* *step += 1 */
int vgpt_sampler_advance(int32_t* step, void* stream);
/* dst[l][0:slab_bytes] = src[*step][l][0:slab_bytes] for l < n_layers (step clamped to [0, n_steps)); all sizes and
 * strides in bytes, multiples of 16.  The engine keeps the post-RoPE q/k/v rows of the time tokens of every denoise
 * step (they depend on the step only: a time row sees `<|diffusion|>` and time columns, never image columns,
 * LVM/processor.py:682-731) and drops the current step's rows into the per-layer qkv buffer; the reference recomputes
 * them inside every model call (LVM/scheduler.py:174 -> LVM/model.py:446-449).  Reads the step from device memory:
 * capturable. */
int vgpt_sampler_copy_step_rows(const void* src, void* dst, const int32_t* step, int n_steps, int n_layers,
                                int64_t slab_bytes, int64_t src_step_stride_bytes, int64_t src_layer_stride_bytes,
                                int64_t dst_layer_stride_bytes, void* stream);
/* z_model = bf16(z) */
int vgpt_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream);

/* ---- VAE conv stack (diffusers==0.29.0 AutoencoderKL, sdxl-vae layout; fp32 like the reference:
 *      LVM/pipeline.py:110-117 encode, :558-590 decode, LVM/train/train_x1_stage1_noiseinput.py:190) ---- */

/* GroupNorm statistics: stats[(n*groups+g)*2 + {0,1}] = {mean, rsqrt(var+eps)} over (C/groups, HW).
 * x: (N, C, HW) fp32. */
int vgpt_groupnorm_stats(const float* x, float* stats, int64_t N, int C, int HW, int groups, float eps,
                         void* stream);

/* Implicit-GEMM convolution on the fp32 MFMA, NCHW fp32:
 *   y = conv(f(x), w) + bias + resid,  f = optional nearest-x2 upsample, then optional
 *   GroupNorm(gn_stats, gamma, beta)[+SiLU] applied on load (zero padding applied after f).
 * ksize 3 (stride 1 pad 1 | stride 2 with Downsample2D's (0,1,0,1) pad) or 1.  w: (Cout, Cin*k*k) with row
 * stride ldw, or stored transposed (Cin*k*k, Cout) when w_transposed; w_batch_stride != 0 selects a
 * per-image weight matrix (attention products).  bias/resid/gn_* may be NULL (gn_groups = 0). */
int vgpt_conv2d_fwd(const float* x, const float* w, const float* bias, const float* resid,
                    const float* gn_stats, const float* gn_gamma, const float* gn_beta, float* y, int N,
                    int Cin, int Hin, int Win, int Cout, int ksize, int stride, int upsample, int gn_groups,
                    int gn_silu, int w_transposed, int64_t ldw, int64_t w_batch_stride, void* stream);

/* 3x3 stride-1 convolution on the bf16 matrix cores at fp32-class accuracy: every operand is split hi + lo (two bf16)
 * and the three leading product terms are accumulated in fp32 (~16 mantissa bits; the reference's torch/cuDNN default
 * for its fp32 VAE is TF32, 10 bits).  Same prologue (nearest x2 upsample, GroupNorm(+SiLU)) and epilogue (bias,
 * residual) as vgpt_conv2d_fwd.  Weights are pre-split once by vgpt_conv_pack_weights_bx3 into LDS-ready images:
 * w (Cout, Cin, 3, 3) fp32 -> `packed`, vgpt_conv_bx3_packed_bytes(Cout, Cin) bytes (one 48-KiB hi + lo image per
 * 64-channel output tile and 16-channel input chunk, ten tap slots of 16 channels per output channel, copied into LDS by
 * LDS-DMA; the format is private to the two functions below). */
int64_t vgpt_conv_bx3_packed_bytes(int Cout, int Cin);
int vgpt_conv_pack_weights_bx3(const float* w, void* packed, int Cout, int Cin, void* stream);
int vgpt_conv2d_bx3_fwd(const float* x, const void* packed, const float* bias, const float* resid, const float* gn_stats,
                        const float* gn_gamma, const float* gn_beta, float* y, int N, int Cin, int Hin, int Win, int Cout,
                        int upsample, int gn_groups, int gn_silu, void* stream);
/* 1x1 convolution (resnet shortcuts, the mid-block attention's projections; diffusers ResnetBlock2D.conv_shortcut /
 * Attention.to_q..to_out, call sites LVM/pipeline.py:565,578) in the same split-bf16 arithmetic and with the same
 * GroupNorm(+SiLU) prologue / bias + residual epilogue; x (N, Cin, HW), y (N, Cout, HW) fp32, Cin % 32 == 0.
 * w (Cout, Cin) fp32 is pre-split once: vgpt_conv1x1_bx3_packed_bytes(Cout, Cin) bytes. */
int64_t vgpt_conv1x1_bx3_packed_bytes(int Cout, int Cin);
int vgpt_conv1x1_pack_weights_bx3(const float* w, void* packed, int Cout, int Cin, void* stream);
int vgpt_conv1x1_bx3_fwd(const float* x, const void* packed, const float* bias, const float* resid, const float* gn_stats,
                         const float* gn_gamma, const float* gn_beta, float* y, int N, int Cin, int HW, int Cout,
                         int gn_groups, int gn_silu, void* stream);

/* In-place softmax over the KEY axis of S^T (N, keys, queries) with pre-scale (mid-block attention). */
int vgpt_col_softmax(float* s, int N, int keys, int queries, float scale, void* stream);

/* DiagonalGaussianDistribution.sample + latent scaling (LVM/pipeline.py:110-115):
 * z = (mean + exp(0.5*clamp(logvar,-30,20)) * noise - shift) * scaling.
 * moments: (N, 2*per_image) = [mean | logvar] per image; noise, z: (N, per_image). */
int vgpt_vae_sample(const float* moments, const float* noise, float* z, int N, int64_t per_image, float shift,
                    float scaling, void* stream);

/* (x*0.5+0.5).clamp(0,1)*255 -> uint8, NCHW -> NHWC (LVM/pipeline.py:585-588). */
int vgpt_vae_postprocess_u8(const float* x, uint8_t* out, int N, int C, int H, int W, void* stream);

/* y = float(x) * mul + add (latent / scaling_factor + shift before decode, LVM/pipeline.py:573-577). */
int vgpt_affine_to_f32(const void* x, int x_is_bf16, float* y, int64_t n, float mul, float add, void* stream);

/* ---- stage-1 pre-training step (LVM/train_helper/loss.py:128-243, LVM/train/train_x1_stage1_noiseinput.py:351-405;
 *      the reference gets these from torch.autograd + DeepSpeed) ---------------------------------------- */

/* Attention forward that also returns the base-2 log-sum-exp of the scaled scores, lse (B, n_heads, L) fp32.
 * order: NULL or the launch order from vgpt_attn_qblock_order(q_start = 0). */
int vgpt_attn_blockmask_fwd_lse(const void* q, const void* k, const void* v, void* o, float* lse,
                                const uint32_t* bits, const uint8_t* summary, const int32_t* order, int64_t B, int64_t L,
                                int n_heads,
                                int n_kv_heads, int head_dim, int64_t q_sb, int64_t q_sh, int64_t q_ss, int64_t k_sb,
                                int64_t k_sh, int64_t k_ss, int64_t v_sb, int64_t v_sh, int64_t v_ss, int64_t o_sb,
                                int64_t o_sh, int64_t o_ss, float scale, void* stream);
/* Attention backward (head_dim 96): dq, dk, dv from q, k, v, o, dout, lse.  strides: 24 x int64 element strides
 * (batch, head, seq) of q, k, v, o, dout, dq, dk, dv in that order.  delta_ws: (B, n_heads, L) fp32 workspace. */
int vgpt_attn_blockmask_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout,
                            const float* lse, float* delta_ws, void* dq, void* dk, void* dv, const uint32_t* bits,
                            const uint8_t* summary, int64_t B, int64_t L, int n_heads, int n_kv_heads, int head_dim,
                            const int64_t* strides, float scale, void* stream);

/* (R, C) bf16 row stride ld_in -> (C, Rp) bf16, columns r >= R zero-filled (operands of the backward NT GEMMs). */
int vgpt_transpose_pad_bf16(const void* in, void* out, int64_t R, int64_t C, int64_t Rp, int64_t ld_in, void* stream);
/* un-fused gated activation: act = f(gate) * up from gate_up (M, 2I); and its backward d(gate_up). */
int vgpt_silu_mul_fwd(const void* gate_up, void* act_out, int64_t M, int64_t I, int act, void* stream);
int vgpt_silu_mul_bwd(const void* gate_up, const void* dact, void* dgate_up, int64_t M, int64_t I, int act, void* stream);
/* y = f(pre);  dx = f'(pre) * dy  (bf16, n elements). */
int vgpt_act_fwd(const void* pre, void* y, int64_t n, int act, void* stream);
int vgpt_act_bwd(const void* pre, const void* dy, void* dx, int64_t n, int act, void* stream);
/* Phi3RMSNorm backward: dx = d(norm)/dx (+ dres if not NULL), dw += sum_rows dy * xhat (fp32, caller zeroes). */
int vgpt_rmsnorm_bwd(const void* x, const void* w, const void* dy, const void* dres, void* dx, float* dw,
                     float* rstd_ws /* rows floats */, int64_t rows, int64_t H, float eps, void* stream);
/* C[m][n] = alpha * sum_k A[m*sa_m + k*sa_k] * B[k*sb_k + n*sb_n] (+ C); *_f32: 0 = bf16, 1 = fp32.  For the
 * small heads (patch embeds, timestep MLPs, adaLN, final Linear) whose backward is not worth an MFMA kernel.
 * splitk_ws (NULL or ws_floats floats): workspace that lets long reductions with few outputs be sliced over the chip
 * (slices are added in a fixed order: deterministic).  vgpt_matmul_generic_workspace_bytes(M, N, K) is the size with which
 * a problem gets all the slices it can use (0: it runs unsliced); a smaller workspace means fewer slices. */
int64_t vgpt_matmul_generic_workspace_bytes(int64_t M, int64_t N, int64_t K);
int vgpt_matmul_generic(const void* A, int a_f32, int64_t sa_m, int64_t sa_k, const void* B, int b_f32, int64_t sb_k,
                        int64_t sb_n, void* C, int c_f32, int64_t sc_m, int64_t sc_n, int64_t M, int64_t N, int64_t K,
                        float alpha, int accumulate, float* splitk_ws, int64_t ws_floats, void* stream);
/* out[c] (+)= sum_r X[r*ld + c]  (bias gradients). */
int vgpt_colsum(const void* X, int x_f32, float* out, int64_t R, int64_t C, int64_t ld, int accumulate, void* stream);
/* xt[f] = t[f]*x1[f] + (1-t[f])*x0[f]  (loss.py:175,186), fp32 in, bf16 out. */
int vgpt_lerp_frames(const float* x1, const float* x0, const float* t, void* out, int n_frames, int64_t elems,
                     void* stream);
/* loss[f] = mean((x1[f]-pred[f])^2) (loss.py:209-218); dpred (optional, bf16) = d(mean_f loss)/dpred. */
int vgpt_mse_frames(const void* pred, const float* x1, float* loss, void* dpred, int n_frames, int64_t elems,
                    void* stream);
/* The same with the mean taken over n_mean >= n_frames terms: the frames' terms are part of a longer loss vector (the
 * input-head terms of input_output_return, loss.py:220-225, appended before .mean()). */
int vgpt_mse_frames_mean(const void* pred, const float* x1, float* loss, void* dpred, int n_frames, int n_mean, int64_t elems,
                         void* stream);
/* FinalLayer split for training: v = LN(x)(1+scale)+shift with xhat / rstd saved; and its backward (dmod fp32
 * (n_frames, 2H) accumulated with atomics — caller zeroes; dhidden rows written at dst_row[f] + t). */
int vgpt_ln_mod_fwd(const void* hidden, const int32_t* src_row, const void* mod, void* v_out, float* xhat_out,
                    float* rstd_out, int n_frames, int ntok, int64_t H, float eps, void* stream);
int vgpt_ln_mod_bwd(const void* dv, const float* xhat, const float* rstd, const void* mod, const int32_t* dst_row,
                    void* dhidden, float* dmod, int n_frames, int ntok, int64_t H, void* stream);
/* embed_tokens backward: dtable[ids[r]] += dseq[r] for rows with keep[r] != 0 (fp32 atomics). */
int vgpt_embed_bwd(const int64_t* ids, const uint8_t* keep, const void* dseq, float* dtable, int64_t rows, int64_t H,
                   int64_t vocab, void* stream);
/* (n_frames, C, h, w) -> (n_frames*ntok, 16) patch vectors; gradient of unpatchify; row-segment gather. */
int vgpt_patchify(const void* x, void* patches, int n_frames, int C, int h, int w, void* stream);
int vgpt_unpatchify_bwd(const void* dpred, void* dy16, int n_frames, int C, int h, int w, void* stream);
int vgpt_gather_rows(const void* in, const int32_t* row0, void* out, int n_seg, int per, int64_t H, void* stream);
/* *out += sum g^2 (deterministic two-stage reduction: replicas must agree bit for bit);  coef = min(1, max_norm/(sqrt(sumsq)+1e-6)) * extra_scale;  AdamW on fp32 master weights
 * (torch.optim.AdamW update; bf16 model copy refreshed; *grad_scale multiplies the gradient). */
int vgpt_sumsq(const void* g, int g_f32, float* out, int64_t n, float* partial_ws /* >= 1024 floats */, void* stream);
int vgpt_clip_coef(const float* sumsq, float* coef, float* norm_out, float max_norm, float extra_scale, void* stream);
int vgpt_adamw_step(float* master, void* param, const void* grad, int grad_f32, float* m, float* v, int64_t n, float lr,
                    float beta1, float beta2, float eps, float weight_decay, int step, const float* grad_scale,
                    void* stream);

/* ---- hipGraph helpers (sampler loop under graph capture) ----------------- */
int vgpt_graph_begin_capture(void* stream);
int vgpt_graph_end_capture(void* stream, void** graph_exec_out);
int vgpt_graph_launch(void* graph_exec, void* stream);
int vgpt_graph_destroy(void* graph_exec);

/* ---- box calibration (bench.py `calibration`; never on the product path) -- */
/* 256 workgroups x 8 waves of nothing but mfma_f32_16x16x32_bf16 on random register operands: the rate the matrix pipes
 * hold at the clock this device grants under load (scripts/probes/mfma_shape_rate.hip is the stand-alone form);
 * vgpt_calib_mfma_flops(iters) = the FLOPs one such launch executes.  vgpt_calib_copy: 16-byte-per-lane streaming copy
 * of n_bytes (n_bytes read + n_bytes written): the HBM rate.  The caller times the launches with events on `stream`. */
int vgpt_calib_mfma(float* out, int iters, void* stream);
double vgpt_calib_mfma_flops(int iters);
int vgpt_calib_copy(const void* src, void* dst, int64_t n_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VGPT_H */
