/* pianobart_hip.h -- C ABI of libpianobart_hip.so (gfx950 / MI355X).
 *
 * The reference (RS2002/PianoBart) has no FFI: its hot path is Python calling ATen ops through
 * `transformers.BartModel`. This header is the boundary a maintainer binds
 * instead (ctypes stub in INTEGRATION.md): one entry per op of SURVEY.md 2.2, plain pointers and
 * sizes, a hipStream_t passed as void*, int status (0 = ok; otherwise pb_last_error() describes it).
 * All pointers are DEVICE pointers unless stated. No global state besides the last-error string.
 *
 * Storage dtype of activations / weight shadows: PB_F32 (exact-f32 parity path, f32-input MFMA)
 * or PB_BF16 (throughput path, bf16 MFMA, f32 accumulate). Statistics, loss, optimizer state,
 * biases and LayerNorm parameters are always f32.
 */
#ifndef PIANOBART_HIP_H
#define PIANOBART_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define PB_F32 0
#define PB_BF16 1
#define PB_F32X3 2   /* pb_gemm only (ABI 8): f32 operands, f32 C and aux, the products as split-bf16 triples a_hi b_hi + a_hi b_lo + a_lo b_hi on the
                        bf16 matrix cores with f32 accumulation (~2^-16 relative per product): the parity-grade instantiation that is not bound by
                        the f32-input MFMA rate (precision="bf16x3"); every other op of that instantiation runs its PB_F32 form */

#define PB_ABI_VERSION 8   /* 8 (round 6): + pb_decoder_sampler_init / launch / wait / logs / seek (device-sampled decode), dtype PB_F32X3 in pb_gemm, pb_flash_*_x3, pb_gemm_reserve_cus; 7 (round 5): + PB_GEMM_ROWDOT / rowdot_out in pb_gemm_desc, delta_rows in pb_flash_bwd1*; 6 (round 5): + pb_flash_bwd1_supported; 5 (round 4): + pb_flash_bwd1*, bh_order in the packed attention calls; 4 (round 3): + pb_decoder_*, pb_nucleus_rows, pb_ids_check */
int pb_abi_version(void);
const char* pb_last_error(void);

/* ---- K3/K5/K6/K7/K8 (+ unfused attention products): C (+)= epi(alpha * A.B^T) ------------------
 * Replaces nn.Linear forward/backward inside transformers BartAttention / Bart*Layer
 * (modeling_bart.py:207,227-228,255,297-302) and model.py:124-125 (MLM heads); PianoBart.py:68,71.
 * A(m,k): a_kcontig ? A[m*lda + k] : A[k*lda + m];  B(n,k): b_kcontig ? B[n*ldb + k] : B[k*ldb + n].
 * Batched over nb1*nb2 problems with element strides s?1 / s?2. */
#define PB_GEMM_ACCUM 1          /* C = C + result                                   */
#define PB_GEMM_C_F32 2          /* C is float regardless of dtype                   */
#define PB_GEMM_GELU 4           /* C = gelu_erf(result); aux_out = gelu_erf'(result): the forward pays 3 extra FMAs per element so that */
#define PB_GEMM_MUL_GELU_GRAD 8  /* C = result * aux_in      ... the backward of the activation (dU = dG * gelu'(U)) is one multiply        */
#define PB_GEMM_FORCE_V1 16      /* use the generic register-staged kernel (tests)    */
#define PB_GEMM_TILE128 32       /* bf16 fast path: force the 128x128 tile             */
#define PB_GEMM_TILE256 64       /* bf16 fast path: prefer the 256x256 tile (default when M >= 2048, N >= 512, no split-K) */
#define PB_GEMM_NO_EPILOGUE 128  /* profiling: main loop only, nothing is stored                                    */
#define PB_GEMM_REG_EPILOGUE 256 /* A/B runs: interior tiles of the 256x256 kernel store straight from the MFMA register layout (64-byte row pieces on
                                    lanes 16 apart) instead of through the row staging (8 rows x 128 contiguous bytes per instruction)            */
#define PB_GEMM_ONE_BARRIER 2048 /* A/B runs: 256x256 tile with the one-barrier kernel instead of the ping-pong one   */
#define PB_GEMM_PLAIN_GRID 4096  /* 256x256 ping-pong kernel as an ordinary grid (one workgroup per work item) instead of the
                                    persistent one-per-CU grid: what to ask for when other kernels (RCCL) hold CUs        */
#define PB_GEMM_TAIL_SPLIT 32768 /* 256x256 kernel: the tiles of a partly filled last round of the grid may be cut into K ranges that
                                    occupy the idle CUs (f32 partials, finished by a second small launch); a cost model decides.
                                    Pays for a caller that runs one GEMM at a time (N = 768 at 26 624 rows: +13-17 %): the training
                                    step asks for it in forward; in backward its second stream already fills those CUs     */
#define PB_GEMM_ROWDOT 131072     /* C = result as usual, and per 64-column group the row sums of C * aux_in go to rowdot_out (see pb_gemm_desc) */
#define PB_GEMM_LEAVE_CUS 262144  /* persistent grids: launch CUs - pb_gemm_reserve_cus() workgroups instead of one per CU (data parallel: the backward
                                    GEMMs that run beside RCCL's resident kernels)                                                       */
#define PB_GEMM_ROW_SPLIT 65536  /* 256x256 kernel: the M tiles of the full rounds of the persistent grid stay with it, the remaining rows go to a
                                    second launch of the 128x128 kernel (no partials). Measured (round 3): -1.8 % on the one-stream step
                                    together with nothing else, +-0 on the shipped two-stream step (its second stream already fills the CUs a
                                    short last round leaves idle, and the forward's K = 768 shapes lose what the split saves to the slower
                                    128x128 tiles): on request only                                                                   */
typedef struct pb_gemm_desc {
    const void* A; const void* B; void* C;
    const float* bias;            /* per-n, may be NULL */
    const void* aux_in; void* aux_out; /* dtype storage, leading dim ldaux */
    int32_t dtype, a_kcontig, b_kcontig, flags;
    int32_t M, N, K, nb1, nb2, _pad;
    int64_t lda, ldb, ldc, ldaux;
    int64_t sA1, sA2, sB1, sB2, sC1, sC2;
    float alpha; float _pad2;
    int32_t splitk; int32_t _pad3;   /* >1: split K over blockIdx.z into f32 slabs (bf16, f32 C, no epilogue) */
    void* slabs;                      /* workspace of splitk*M*N floats when splitk > 1 */
    float* colsum_out;                /* optional: colsum_out[n] += sum_m C[m][n] (the bias gradient of the layer that produced C's
                                         cotangent, e.g. db1 from dU): taken from the epilogue registers of the 256x256 kernel, by a
                                         pb_colsum pass over C otherwise. Needs colsum_ws; single batch, no split-K */
    float* colsum_ws;                 /* workspace of pb_gemm_colsum_ws_floats(M, N) floats */
    float* rowdot_out; int64_t ld_rowdot; /* PB_GEMM_ROWDOT (ABI 7): rowdot_out[(n / 64) * ld_rowdot + m] = sum over columns 64 (n / 64) .. + 63 of
                                         bf16(C[m][.]) * aux_in[m][.], f32 -- with C = dO (the input gradient of the attention output
                                         projection) and aux_in = O this is the delta = rowsum(dO * O) per head that pb_flash_bwd1* reads
                                         (delta_rows). NT layout, M and N multiples of 256, bf16, no other epilogue; refused otherwise */
} pb_gemm_desc;
int pb_gemm(const pb_gemm_desc* d, void* stream);
/* CUs (rounded up to a multiple of 8) that the persistent one-workgroup-per-CU GEMM grids launched with PB_GEMM_LEAVE_CUS leave to other resident kernels -- RCCL's, in a
 * data-parallel job (the reference: nn.DataParallel's reduce_add on the side, pretrain.py:63-65). Process-wide; 0 = none (default). ABI 8. */
int pb_gemm_reserve_cus(int32_t n);
int64_t pb_gemm_colsum_ws_floats(int32_t M, int32_t N);

/* ---- K1/K2: Octuple gather-sum + position + LayerNorm (+dropout) -----------------------------
 * Replaces PianoBart.py:60-71 (8 x Embedding*16 -> cat -> Linear) and modeling_bart.py:520-525 /
 * 648-654 (x + pos[s+2] -> layernorm_embedding -> dropout). P is the projected table
 * P[off_i + v] = 16 * E_i[v] @ W_lin[:, 256 i : 256 i + 256]^T  (1280 x d, f32), built with pb_gemm. */
int pb_ids_to_i16(const int64_t* ids, int16_t* out, int64_t n, void* stream);
/* Range check of (T,8) Octuple ids against the 8 table sizes (device int32[8], classes order): *flag |= 1 (device word) when an id is
 * negative or >= its table size -- what nn.Embedding answers with an IndexError (PianoBart.py:15-16). pb_ids_to_i16 writes -1
 * for a value that does not fit int16, so a checked conversion cannot alias into a valid id. An offending id is REPLACED by 0 in
 * ids16, so that the gather kernels enqueued behind the check never leave their tables; the host reads the word at its next
 * synchronisation point (Engine.check_ids) and raises before any result is used. */
int pb_ids_check(int16_t* ids16, int64_t n, const int32_t* limits8, int32_t* flag, void* stream);
int pb_embed_ln_fwd(const int16_t* ids16 /*(T,8)*/, const float* P, const int32_t* seg_off /*host, 8*/,
                    const float* lin_bias, const float* pos /*(S+2,d)*/, const float* ln_w, const float* ln_b,
                    void* y /*(T,d) dtype*/, float* mean, float* rstd, int32_t T, int32_t S, int32_t d,
                    int32_t dtype, float eps, uint64_t seed, uint32_t site, float p_drop, void* stream);
/* backward: dy -> dbias (d), dgamma/dbeta (d) and either (dz_out == NULL) dP / dpos by f32 atomic scatter-add
 * (exact-f32 path), or (dz_out != NULL) dz (T,d) in dtype for the atomic-free route:
 * dP = Onehot^T dz through pb_gemm (pb_onehot_build) and dpos through pb_batch_sum. */
int pb_embed_ln_bwd(const void* dy, const int16_t* ids16, const float* P, const int32_t* seg_off,
                    const float* lin_bias, const float* pos, const float* ln_w, const float* mean,
                    const float* rstd, float* dP, float* dpos, float* dbias, float* dgamma, float* dbeta,
                    float* partials /*workspace, pb_ln_partials_floats()*/, void* dz_out, int32_t T, int32_t S, int32_t d,
                    int32_t dtype, uint64_t seed, uint32_t site, float p_drop, void* stream);
/* the same two on packed rows (pb_rowmap_build): row r is row row_ids[r] = b*S + s of the padded batch, i.e. it sits at sequence
 * position row_ids[r] % S and draws the dropout bits of that row (a packed step drops exactly what the padded step drops);
 * T need not be a multiple of S. With dz_out the position-table gradient comes from pb_pos_grad_packed. */
int pb_embed_ln_fwd_packed(const int16_t* ids16, const int32_t* row_ids, const float* P, const int32_t* seg_off,
                           const float* lin_bias, const float* pos, const float* ln_w, const float* ln_b, void* y,
                           float* mean, float* rstd, int32_t T, int32_t S, int32_t d, int32_t dtype, float eps,
                           uint64_t seed, uint32_t site, float p_drop, void* stream);
int pb_embed_ln_bwd_packed(const void* dy, const int16_t* ids16, const int32_t* row_ids, const float* P,
                           const int32_t* seg_off, const float* lin_bias, const float* pos, const float* ln_w,
                           const float* mean, const float* rstd, float* dP, float* dpos, float* dbias, float* dgamma,
                           float* dbeta, float* partials, void* dz_out, int32_t T, int32_t S, int32_t d, int32_t dtype,
                           uint64_t seed, uint32_t site, float p_drop, void* stream);
/* onehot (T,V) bf16 with ones at columns seg_off[i] + ids16[t][i] */
int pb_onehot_build(const int16_t* ids16, const int32_t* seg_off /*host 8*/, void* out, int64_t T, int32_t V, void* stream);
/* out[i] += sum_b x[b*Sd + i], i < Sd */
int pb_batch_sum(const void* x, float* out, int32_t B, int64_t Sd, int32_t dtype, void* stream);

/* ---- K5/K6 tail: y = LayerNorm(res + dropout(a)) ------------------------------------------------
 * Replaces dropout + residual + LayerNorm of modeling_bart.py:292-294,300-302,362-364,377-379,386-388. */
int64_t pb_ln_partials_floats(int32_t d);
int pb_add_ln_fwd(const void* res, const void* a, const float* ln_w, const float* ln_b, void* y,
                  float* mean, float* rstd, int32_t T, int32_t d, int32_t dtype, float eps,
                  uint64_t seed, uint32_t site, float p_drop, void* stream);
/* on packed rows: row r draws the dropout bits of row row_ids[r] of the padded batch */
int pb_add_ln_fwd_packed(const void* res, const void* a, const float* ln_w, const float* ln_b, void* y,
                         float* mean, float* rstd, const int32_t* row_ids, int32_t T, int32_t d, int32_t dtype, float eps,
                         uint64_t seed, uint32_t site, float p_drop, void* stream);
/* dres gets dz (accumulated into if accum_dres), da gets dz*dropmask; dgamma/dbeta/dbias_a are ADDED to. */
int pb_add_ln_bwd(const void* dy, const void* res, const void* a, const float* ln_w, const float* mean,
                  const float* rstd, void* dres, void* da, float* dgamma, float* dbeta, float* dbias_a,
                  float* partials, int32_t T, int32_t d, int32_t dtype, int32_t dres_f32, int32_t accum_dres,
                  uint64_t seed, uint32_t site, float p_drop, void* stream);
int pb_add_ln_bwd_packed(const void* dy, const void* res, const void* a, const float* ln_w, const float* mean,
                         const float* rstd, void* dres, void* da, float* dgamma, float* dbeta, float* dbias_a,
                         float* partials, const int32_t* row_ids, int32_t T, int32_t d, int32_t dtype, int32_t dres_f32,
                         int32_t accum_dres, uint64_t seed, uint32_t site, float p_drop, void* stream);

/* ---- bias gradients: out[n] += sum_t dy[t][n] --------------------------------------------------*/
/* partials: workspace of at least pb_colsum_partials_floats(N) floats */
int64_t pb_colsum_partials_floats(int32_t N);
int pb_colsum(const void* dy, int64_t ld, float* out, float* partials, int32_t T, int32_t N, int32_t dtype,
              int32_t src_f32, void* stream);

/* ---- K4 (unfused form, both dtypes): masked softmax over key axis -------------------------------
 * scores (B,H,Sq,Sk) f32 = q.k^T (unscaled); P = softmax(scale*scores + mask); a query row with no
 * visible key gives an all-zero row (transformers 5.x SDPA behaviour, oracle header).
 * key_mask (B,Sk) float (!=0 keeps) or NULL; causal: key j visible to query i iff j <= i. */
int pb_softmax_fwd(const float* scores, const float* key_mask, void* P, int32_t B, int32_t H, int32_t Sq,
                   int32_t Sk, float scale, int32_t causal, int32_t dtype, void* stream);
/* dS = scale * P * (dP - rowsum(dP*P)) written in dtype */
int pb_softmax_bwd(const float* dP, const void* P, void* dS, int64_t rows, int32_t Sk, float scale,
                   int32_t dtype, void* stream);

/* ---- K4 fused (bf16): flash attention forward / backward, head_dim 32/64/128 --------------------
 * Replaces modeling_bart.py:115-140 (eager) / F.scaled_dot_product_attention and its autograd backward.
 * q,k,v,o,dout,dq,dk,dv: bf16, element (b,s,h,c) at ptr[b*sb + s*ss + h*hd + c]; lse, delta: (B,H,Sq) f32.
 * dout must have o's strides. Backward = delta + dKV + dQ kernels (no atomics, deterministic). */
/* kmax (B) int32, optional: 1 + index of the last visible key of each batch row (pb_key_extent); key tiles at or beyond
 * it are skipped (they are masked for every query). NULL = no skipping. Used by the head_dim-64 kernels. */
int pb_key_extent(const float* key_mask, int32_t* kmax, int32_t B, int32_t Sk, void* stream);
int pb_flash_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const float* key_mask, const int32_t* kmax,
                 int32_t B, int32_t H, int32_t Sq, int32_t Sk, int32_t hd, int64_t q_sb, int64_t q_ss,
                 int64_t k_sb, int64_t k_ss, int64_t v_sb, int64_t v_ss, int64_t o_sb, int64_t o_ss,
                 float scale, int32_t causal, void* stream);
int pb_flash_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                 const float* key_mask, const int32_t* kmax, void* dq, void* dk, void* dv, float* delta,
                 int32_t B, int32_t H, int32_t Sq, int32_t Sk, int32_t hd, int64_t q_sb, int64_t q_ss,
                 int64_t k_sb, int64_t k_ss, int64_t v_sb, int64_t v_ss, int64_t o_sb, int64_t o_ss,
                 int64_t dq_sb, int64_t dq_ss, int64_t dk_sb, int64_t dk_ss, int64_t dv_sb, int64_t dv_ss,
                 float scale, int32_t causal,
                 float* dbias_q, float* dbias_k, float* dbias_v, float* dbias_ws /* optional (all or none; head_dim 64/96/128): dbias_x[c] +=
                     column sums of dQ / dK / dV over (batch, position) = the bias gradients of the q / k / v projections, taken from the
                     kernels' epilogue registers; dbias_ws: pb_flash_bias_ws_floats(B, H, Sq, Sk, hd) floats */,
                 void* stream);
int64_t pb_flash_bias_ws_floats(int32_t B, int32_t H, int32_t Sq, int32_t Sk, int32_t hd);

/* ---- K4 on packed rows (dead-row compaction, head_dim 64 / 96 / 128) -----------------------------
 * The PAD tail the reference computes and then masks (PianoBart.py:60-75: attention_mask hides rows as keys only) is dropped: the
 * rows of the batch lie back to back. Batch b's query rows are rows q_off[b] .. q_off[b] + q_len[b] - 1 of q / o / dout / dq
 * (row stride *_ss, element (row, h, c) at ptr[row*ss + h*hd + c]), its key rows k_off[b] .. k_off[b] + k_len[b] - 1 of
 * k / v / dk / dv; the first k_vis[b] key rows of the batch are the visible ones (the rest receive no gradient and are seen
 * by no query). causal: key row j is visible to query row i of the same batch iff j <= i (row indices within the batch) and
 * j < k_vis[b]. Sq_max / Sk_max: maxima of q_len / k_len (grid and LDS sizing); lse and delta are (B, H, Sq_max) f32.
 * dbias_ws: pb_flash_bias_ws_floats(B, H, Sq_max, Sk_max, hd) floats. All five descriptors are device int32 (B).
 * bh_order (device int32, B * H entries, or NULL): the order in which the grid takes the (batch, head) pairs (entry = b * H + h). Costs
 * spread 4x over a packed batch; a caller that lists the pairs longest first, dealt eight at a time (one per XCD), removes the tail
 * of a static grid. It changes the order of the work only: results are bit-identical with and without it. */
int pb_flash_fwd_packed(const void* q, const void* k, const void* v, void* o, float* lse, const int32_t* q_off,
                        const int32_t* q_len, const int32_t* k_off, const int32_t* k_len, const int32_t* k_vis,
                        int32_t B, int32_t H, int32_t Sq_max, int32_t Sk_max, int32_t hd, int64_t q_ss, int64_t k_ss,
                        int64_t v_ss, int64_t o_ss, float scale, int32_t causal, const int32_t* bh_order, void* stream);
int pb_flash_bwd_packed(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                        void* dq, void* dk, void* dv, float* delta, const int32_t* q_off, const int32_t* q_len,
                        const int32_t* k_off, const int32_t* k_len, const int32_t* k_vis, int32_t B, int32_t H,
                        int32_t Sq_max, int32_t Sk_max, int32_t hd, int64_t q_ss, int64_t k_ss, int64_t v_ss, int64_t o_ss,
                        int64_t dq_ss, int64_t dk_ss, int64_t dv_ss, float scale, int32_t causal,
                        float* dbias_q, float* dbias_k, float* dbias_v, float* dbias_ws, const int32_t* bh_order, void* stream);

/* ---- K4b: attention backward in ONE pass (pb_flash1.hip, head_dim 64) -------------------------------
 * Same math, arguments and results (to bf16 rounding) as pb_flash_bwd / pb_flash_bwd_packed, computed key-stationary: a workgroup
 * owns 256 keys of one (batch, head), keeps their dK / dV in accumulator registers and sweeps the query tiles once (5 matrix
 * products and one exp pass per (query, key) pair instead of 7 and 2). dQ is summed over the key blocks without atomics: block j
 * writes its partial into bf16 slab j of dq_ws (pb_flash_bwd1_ws_bytes(rows of the q side, H, hd, Sk_max) bytes; packed rows:
 * q_rows = rows of the q tensor), a second kernel adds a row's slabs in f32 in block order and rounds once. Deterministic.
 * delta_rows (ABI 7, may be NULL): delta = rowsum(dO * O) per head as [H][rows of the q side] (dense: row = b * Sq + s), e.g. written by the
 * PB_GEMM_ROWDOT epilogue of the GEMM that produced dO; NULL = the call computes it itself into `delta` (B, H, Sq) with one more launch.
 * Replaces the autograd backward of tf:modeling_bart.py:115-140 like K4. */
int64_t pb_flash_bwd1_ws_bytes(int64_t rows, int32_t H, int32_t hd, int32_t Sk_max);
/* 1 if the one-pass kernel takes the shape (head_dim 64, Sq_max <= 6144: its per-sequence -lse / -delta tables live in LDS; dQ slabs
 * <= 8 GiB), else 0: call pb_flash_bwd / pb_flash_bwd_packed. rows = rows of the q side (dense: B * Sq). */
int32_t pb_flash_bwd1_supported(int32_t Sq_max, int32_t Sk_max, int32_t hd, int64_t rows, int32_t H);
int pb_flash_bwd1(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                  const float* key_mask, const int32_t* kmax, void* dq, void* dk, void* dv, float* delta,
                  int32_t B, int32_t H, int32_t Sq, int32_t Sk, int32_t hd, int64_t q_sb, int64_t q_ss,
                  int64_t k_sb, int64_t k_ss, int64_t v_sb, int64_t v_ss, int64_t o_sb, int64_t o_ss,
                  int64_t dq_sb, int64_t dq_ss, int64_t dk_sb, int64_t dk_ss, int64_t dv_sb, int64_t dv_ss,
                  float scale, int32_t causal, float* dbias_q, float* dbias_k, float* dbias_v, float* dbias_ws,
                  void* dq_ws, const float* delta_rows, void* stream);
int pb_flash_bwd1_packed(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                         void* dq, void* dk, void* dv, float* delta, const int32_t* q_off, const int32_t* q_len,
                         const int32_t* k_off, const int32_t* k_len, const int32_t* k_vis, int32_t B, int32_t H,
                         int32_t Sq_max, int32_t Sk_max, int32_t hd, int64_t q_ss, int64_t k_ss, int64_t v_ss, int64_t o_ss,
                         int64_t dq_ss, int64_t dk_ss, int64_t dv_ss, float scale, int32_t causal,
                         float* dbias_q, float* dbias_k, float* dbias_v, float* dbias_ws, void* dq_ws, int64_t q_rows,
                         const int32_t* bh_order, const float* delta_rows, void* stream);

/* ---- row maps for the packed step (pb_rowmap.hip) ------------------------------------------------
 * pb_rowmap_count: counts (B,8) int32 = {encoder rows visible as keys (emask != 0), decoder rows visible as keys (dmask != 0),
 *   decoder live rows (visible, or loss_mask (B,S,8) row != 0), 1 iff the visible decoder positions are exactly 0 .. L-1,
 *   decoder rows with a loss term, 0, 0, 0}.
 * pb_rowmap_build: batch b's packed rows off[b] .. off[b] + len[b] - 1 = its positions with mask != 0 (ascending), then those
 *   with a loss term (loss_mask may be NULL), then its first remaining (dead) positions up to len[b]; row_src[r] = b*S + s and
 *   row_pos[r] = s for packed row r, inv (B*S) = packed row of (b, s) or -1. len[b] must cover the first two classes.
 * pb_rowmap_build_sub: the rows of an existing packing (present (B*S) = its `inv`) that the LAST decoder layer's query side needs:
 *   batch b's rows off[b] .. + len[b] - 1 = its positions with a loss term (ascending), then other positions of the packing;
 *   row_src[r] = b*S + s, row_idx[r] = present[b*S + s] (the row of the existing packing). len[b] <= rows of b in that packing.
 * pb_gather_rows16: dst row r = src row row_src[r] (row_bytes a multiple of 16). pb_scatter_rows16: dst row row_dst[r] = src row r.
 * pb_pos_grad_packed: out (S,d) f32 += sum_b x[inv[b][s]] (the position-table gradient; replaces pb_batch_sum). */
int pb_rowmap_count(const float* emask, const float* dmask, const float* loss_mask, int32_t* counts, int32_t B, int32_t S,
                    void* stream);
int pb_rowmap_build(const float* mask, const float* loss_mask, const int32_t* off, const int32_t* len, int32_t* row_src,
                    int32_t* row_pos, int32_t* inv, int32_t B, int32_t S, void* stream);
int pb_rowmap_build_sub(const float* loss_mask, const int32_t* present, const int32_t* off, const int32_t* len,
                        int32_t* row_src, int32_t* row_idx, int32_t B, int32_t S, void* stream);
int pb_gather_rows16(const void* src, const int32_t* row_src, void* dst, int64_t n_rows, int32_t row_bytes, void* stream);
int pb_scatter_rows16(const void* src, const int32_t* row_dst, void* dst, int64_t n_rows, int32_t row_bytes, void* stream);
int pb_pos_grad_packed(const void* x, const int32_t* inv, float* out, int32_t B, int32_t S, int32_t d, int32_t dtype,
                       void* stream);

/* ---- K9: fused 8-segment log-softmax + CE + argmax + masked accuracy (+ dlogits) ----------------
 * Replaces pretrain.py:112-118,163-189 (np.argmax x8, CrossEntropyLoss x8, masked means).
 * logits (T,V) f32 with the 8 heads at column offsets seg_off[i]; target (T,8) int16; loss_mask
 * (T,8) f32. sums (3,8) f32 += {sum ce*m, sum m, sum correct*m}. If dlogits != NULL:
 * dlogits[t, off_i+c] = coef[i] * m[t,i] * (softmax_c - 1[c==target]) in dtype, coef (8) device f32
 * (= w_i / (sum_w * M_i)). argmax_out (T,8) int16 may be NULL. */
int pb_ce_fwd_bwd(const float* logits, const int16_t* target, const float* loss_mask, const int32_t* seg_off /*host 9*/,
                  float* sums, float* partials, const float* coef, void* dlogits, int16_t* argmax_out,
                  int32_t T, int32_t V, int32_t dtype, void* stream);
int64_t pb_ce_partials_floats(void);
/* counts[i] = sum_t loss_mask[t,i]  (f32, 8) -- the M_i of pretrain.py:117 */
int pb_mask_count(const float* loss_mask, float* counts, float* partials /* >= pb_ce_partials_floats() */, int64_t T, void* stream);
/* coef[i] = scale * w[i] / (sum_w * counts[i])   (scale = 1 for pretrain.py:185-189; finetune_generation.py:241-250 uses
 * w_i = weight_i * n_tok_i with the denominator sum(n_tok), i.e. scale = sum_w / sum(n_tok)) */
int pb_loss_coef(const float* counts, const float* w /*device 8*/, float* coef, float scale, void* stream);

/* ---- K10/K11: global grad norm, clip, HF-AdamW, bf16 shadow refresh --------------------------------
 * Replaces clip_grad_norm_(.,3.0) (pretrain.py:195) and transformers.AdamW.step (pretrain.py:76,196;
 * 4.29.2 formula: eps added to sqrt(v) before bias correction, decoupled decay after the update). */
int64_t pb_norm_partials_floats(void);
int pb_grad_sqnorm(const float* g, int64_t n, float* partials, float* out_sq /*1 float, overwritten*/, void* stream);
/* clip_coef = min(1, max_norm / (sqrt(sq * gscale^2) + 1e-6)) * gscale */
int pb_clip_coef(const float* sq, float max_norm, float gscale, float* coef, void* stream);
int pb_adamw_step(float* p, const float* g, float* m, float* v, void* shadow /*bf16 or NULL*/, int64_t n,
                  const float* clip_coef /*device, may be NULL*/, float lr, float beta1, float beta2, float eps,
                  float weight_decay, int32_t step, void* stream);
int pb_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream);
/* x (n f32, n % 8 == 0) -> hi = bf16(x), lo = bf16(x - hi): the two planes of the split-bf16 arithmetic (precision="bf16x3") as separate arrays, for products whose
 * other operand is exact in bf16 (the one-hot matrix of the embedding-table gradient: Onehot^T dz = Onehot^T dz_hi + Onehot^T dz_lo). Replaces nothing in the
 * reference: torch.nn.Embedding's backward (index_add in f32) is what the pair of GEMMs computes. */
int pb_split_bf16(const float* x, void* hi, void* lo, int64_t n, void* stream);
int pb_cast_bf16_to_f32(const void* src, float* dst, int64_t n, void* stream);
int pb_fill_f32(float* dst, float value, int64_t n, void* stream);
/* Transposed copies of the weight matrices inside the flat bf16 shadow: table (device, n_matrices x 4 int32) = {element offset, R, C,
 * index of the matrix's first 64x64 tile}; matrix e (R x C row-major at src + offset) is written C x R at dst + offset; R, C and
 * offset multiples of 8; n_tiles = sum of ceil(R/64) ceil(C/64). The backward's dX = dY W then reads W^T as a K-contiguous operand
 * (the NT form of the GEMM kernel is 7-20 % faster than the NN form). */
int pb_transpose_batch_bf16(const void* src, void* dst, const int32_t* table, int32_t n_matrices, int32_t n_tiles, void* stream);
/* dst[i] = bf16(sum_r f32(src[r*n + i])), r < rows; n % 8 == 0: the f32 accumulation of the bf16 gradient chunks a rank owns in the
 * data-parallel exchange (pianobart_amd/parallel.py; replaces the logits gather + gradient reduce of nn.DataParallel, pretrain.py:63-65) */
int pb_sum_rows_bf16(const void* src, void* dst, int32_t rows, int64_t n, void* stream);

/* ---- device-side corruption for the pre-train step (distributional counterpart of gen_mask's
 * TokenMask n=0 branch, pretrain.py:276-295) and decoder shift-right (pretrain.py:132-139) ---------*/
int pb_shift_right(const int16_t* ids, const int16_t* sos_row /*device 8*/, int16_t* out, int32_t B, int32_t S, void* stream);
/* Replaces Pretrainer.gen_mask (pretrain.py:211-546): one workgroup corrupts one (S,8) sequence in LDS.
 * choice (B) int32 device, 1..5 (other values / NULL: drawn uniformly in-kernel, reported in choice_out if given);
 * out (B,S,8) int16, loss_mask (B,S,8) f32 (per-position mask repeated over the 8 columns, pretrain.py:141-142).
 * pad_row / mask_row / n_tokens: HOST arrays of 8. mask_percent is a double: int(l * p) and round(l * p) must come out as in
 * Python. Same distributions as the reference, Philox instead of MT19937. */
int pb_corrupt(const int16_t* ids, int16_t* out, float* loss_mask, const int32_t* choice, int32_t* choice_out, int32_t B,
               int32_t S, double mask_percent, uint64_t seed, const int16_t* pad_row, const int16_t* mask_row,
               const int32_t* n_tokens, void* stream);
/* The same kernel with its random DECISIONS supplied by the caller instead of drawn from Philox: what is left is the deterministic
 * part of gen_mask, which must reproduce the reference's outputs bit for bit when fed the reference's own decisions
 * (tests/test_corrupt_gpu.py, tests/golden/g6_gen_mask.npz). choice (B) int32 device, required. decisions: (B, dec_stride) int32
 * device, dec_stride >= pb_corrupt_replay_stride(S); per sample, by choice:
 *   1 TokenDeletion        dec[i] != 0: position i is deleted (the shuffled maskpos of pretrain.py:221-226); i < S
 *   2 TokenMask            dec[i] = 0 untouched, 1 in mask80 (MASK row), 2 in rand10 (row i of rand_rows), 3 in cur10; i < S
 *   3 SentencePermutation  dec[bar] = place of bar value `bar` in the shuffled bar order (pretrain.py:385-386)
 *   4 TokenInfilling       dec[attempt * S + step] = -1 (random.random() >= p/3: copy the row) or the np.random.poisson(3) draw of a
 *                          span that starts at this step; attempt < 10, steps in the order the reference's while loop takes them
 *   5 DocumentRotation     dec[0] = the rotation offset (random.randint(0, l-1))
 * rand_rows: (B,S,8) int16 device, the get_rand_tok() rows of choice 2 (may be NULL when no decision is 2). */
int pb_corrupt_replay(const int16_t* ids, int16_t* out, float* loss_mask, const int32_t* choice, int32_t B, int32_t S,
                      double mask_percent, const int32_t* decisions, int64_t dec_stride, const int16_t* rand_rows,
                      const int16_t* pad_row, const int16_t* mask_row, void* stream);
int64_t pb_corrupt_replay_stride(int32_t S);

/* ---- K14: fine-tune heads, exact f32 (model.py:128-143 SelfAttention, :165-218 SequenceClassification, :220-232 Excitation,
 * :236-272 TokenClassification; loss finetune.py:121-129). Their matrix products are pb_gemm (f32) calls.
 * pb_eltwise_fwd: y = dropout(act(x)), op 1 tanh / 2 relu / 3 sigmoid / 4 identity / 5 x * x2; n elements;
 *   dropout by the step's Philox stream (seed, site), p_drop = 0 disables it.
 * pb_eltwise_bwd: dx = dy * mask * act'(.) with the derivative taken from the PRE-dropout output `y` (op 5: y = x, also dx2).
 * pb_softmax_dim1_*: F.softmax(x, dim=1) of x (B, S, R) and its backward (model.py:140).
 * pb_ce_rows: loss[row] = CrossEntropy(logits[row, :C], target[row]) (reduction='none'), argmax[row] (first maximum, may be
 *   NULL) and, if dlogits != NULL, dlogits = (softmax - onehot) * coef[0] * weight[row] (weight / coef may be NULL = 1). */
int pb_eltwise_fwd(int32_t op, const float* x, const float* x2, float* y, int64_t n, uint64_t seed, uint32_t site, float p_drop, void* stream);
int pb_eltwise_bwd(int32_t op, const float* y, const float* x2, const float* dy, float* dx, float* dx2, int64_t n, uint64_t seed,
                   uint32_t site, float p_drop, void* stream);
int pb_softmax_dim1_fwd(const float* x, float* y, int32_t B, int32_t S, int32_t R, void* stream);
int pb_softmax_dim1_bwd(const float* y, const float* dy, float* dx, int32_t B, int32_t S, int32_t R, void* stream);
int pb_ce_rows(const float* logits, const int32_t* target, const float* weight, const float* coef, float* loss, float* dlogits,
               int32_t* argmax, int64_t rows, int32_t C, void* stream);
/* Decoder label-embedding swap of the velocity task (PianoBart.change_decoder_embedding, PianoBart.py:88-91; model.py:242-245):
 * pb_gather_rows: out[t] = table[ids[t]] + bias for a small projected label table (nrows x d, f32); pb_gather_rows_bwd: dtable[r] =
 * sum of dout[t] over ids[t] == r (sequential per element: deterministic). pb_dropout: y = x * mask / (1 - p) in storage dtype
 * (BART drops after layernorm_embedding); the same call on the gradient is its backward. */
int pb_gather_rows(const float* table, const int32_t* ids, const float* bias, float* out, int64_t T, int32_t d, int32_t nrows, void* stream);
int pb_gather_rows_bwd(const float* dout, const int32_t* ids, float* dtable, int64_t T, int32_t d, int32_t nrows, void* stream);
int pb_dropout(const void* x, void* y, int64_t n, int32_t dtype, uint64_t seed, uint32_t site, float p_drop, void* stream);
/* The optional regulariser of the fine-tune loop, `loss += weight * torch.norm(param, p=2)` for every parameter tensor
 * (finetune.py:241-243): *loss_acc += weight * ||p||_2 (if loss_acc != NULL) and g += weight * p / ||p||_2 (if g != NULL; 0 where the
 * norm is 0, as torch's norm backward). p, g: n f32 elements, any alignment; scratch: pb_l2_penalty_scratch_floats() floats. */
int pb_l2_penalty(const float* p, float* g, int64_t n, float weight, float* scratch, float* loss_acc, void* stream);
int64_t pb_l2_penalty_scratch_floats(void);

/* ---- K13: batch-1 KV-cached decode (model.py:28-66) --------------------------------------------------------------
 * pb_gemv: y[n] = act(sum_k W[n][k] x[k] + bias[n]), W (N,K) row-major in dtype, x (K) dtype, y dtype or f32, gelu = exact erf GELU.
 * pb_attn_decode: one query (H*hd) against cached K/V rows (element (j,h,c) at ptr[j*ss + h*hd + c]), keys 0..Sk-1, optional
 * key mask (Sk) float; a row with no visible key gives zeros. head_dim 32, 64, 96 or 128, Sk <= 8192.
 * pb_decode_step: one decoder token through all layers: embed(tok16) + pos[i] -> ND x [self-attn with K/V appended at row i,
 * cross-attn on the cached encoder K/V, FFN] -> logits (vocab) f32. All pointers device pointers, weights in dtype storage,
 * biases / LayerNorm / tables f32. */
int pb_gemv(const void* W, const void* x, const float* bias, void* y, int32_t N, int32_t K, int32_t dtype, int32_t y_f32, int32_t gelu, void* stream);
int pb_attn_decode(const void* q, const void* k_cache, const void* v_cache, void* out, const float* key_mask, int32_t H, int32_t Sk,
                   int32_t hd, int64_t k_ss, int64_t v_ss, float scale, int32_t dtype, void* stream);
#define PB_DECODE_MAX_LAYERS 48
#define PB_DECODE_MAX_SPLITS 16
typedef struct pb_decode_layer {
    const void* wqkv; const float* bqkv; const void* wo; const float* bo; const float* ln1_w; const float* ln1_b;
    const void* wq_c; const float* bq_c; const void* wo_c; const float* bo_c; const float* lnc_w; const float* lnc_b;
    const void* w1; const float* b1; const void* w2; const float* b2; const float* ln2_w; const float* ln2_b;
    void* kv_self;            /* (S, 2d) dtype: k | v rows of the tokens decoded so far */
    const void* kv_cross;     /* (S_enc, 2d) dtype: encoder keys | values of this layer */
} pb_decode_layer;
typedef struct pb_decode_plan {
    int32_t dtype, d, H, ffn, S, S_enc, n_layers, vocab;
    int32_t tab_off[9]; int32_t _pad;
    const int16_t* tok16;     /* (8) current decoder input token */
    const float* ptab; const float* lin_b; const float* pos; const float* lne_w; const float* lne_b; const float* enc_mask;
    void* x; void* y1; void* yc; void* y2; void* q; void* ctx; void* a; void* g;   /* scratch rows: d (g: ffn) elements of dtype */
    float* stat;              /* 8 floats */
    float* attn_part;         /* H * PB_DECODE_MAX_SPLITS * (d / H + 4) floats, 16-byte aligned: per-(head, key split) {max, sum, output} of the single-query
                                 attention, merged by the out-projection GEMV; NULL = one workgroup per head writing ctx (the round-1 form) */
    float* logits;            /* (vocab) f32 */
    const void* head_w; const float* head_b;
    pb_decode_layer layers[PB_DECODE_MAX_LAYERS];
} pb_decode_plan;
int pb_decode_step(const pb_decode_plan* plan, int32_t i, void* stream);

/* Decode as ONE hipGraph replay per token (round 3; replaces the per-position host loop of model.py:42-65 around pb_decode_step for the
 * shapes it covers: bf16, head_dim 64 / 128, d a multiple of 256 up to 1024). The position lives in device memory, so the launches of a
 * token (embed, per layer {self-attention with the q|k|v projections and the pending post-LN fused in, out-projection, cross-attention
 * with its q projection, out-projection, fc1 + GELU, fc2}, LM heads: 6 n_layers + 2) carry no position-dependent argument; the token ids
 * go up and the logits row comes down through copy nodes of the same graph.
 *   pb_decoder_create   0 = created (*dec), 1 = this plan's shape is not covered (keep pb_decode_step), < 0 = error. The plan is copied;
 *                       its buffers (weights, K/V caches, scratch rows, attn_part) must stay alive until pb_decoder_destroy.
 *   pb_decoder_reset    start of a prompt: position -1, ordered behind everything enqueued on `caller_stream` so far (encoder pass, cross
 *                       K/V projections); use_graph = 0 issues every token's launches directly (A/B, debugging).
 *   pb_decoder_step     one token: tok8 (8 ids, host) in, the (vocab) f32 logits row of its position out (host); returns when it landed.
 *   pb_decoder_launches kernels per token; pb_decoder_graph 1 when tokens are graph replays. */
/* Host-side nucleus sampling of one position (model.py:84-98 for the 8 heads): probs (heads, width) f32 softmax rows of lengths n[h],
 * thresholds p[h], u[h] = the uniform draw np.random.choice would consume. out[h] = sampled id; bit h of *tie_mask set (out[h] = -1)
 * when the result would depend on numpy's order of equal probabilities: the caller runs its numpy code for that head. No device work. */
int pb_nucleus_rows(const float* probs, int32_t width, const int32_t* n, const float* p, const double* u, int32_t heads, int32_t* out,
                    int32_t* tie_mask);
int pb_decoder_create(const pb_decode_plan* plan, void** dec);
int pb_decoder_destroy(void* dec);
int pb_decoder_reset(void* dec, void* caller_stream, int32_t use_graph);
int pb_decoder_step(void* dec, const int16_t* tok8, float* logits_out);
int pb_decoder_launches(void* dec);
int pb_decoder_graph(void* dec);
/* Device-sampled decode (round 6; ABI 8): the per-token host round trip of model.py:42-65 (logits row down, sampling(), token up) leaves
 * the critical path. np.random.choice's uniform draws do not depend on the logits (model.py:97), so the host draws the (S, 8) of a whole
 * prompt ahead and uploads them once; a one-workgroup kernel behind the LM-head GEMV then does model.py:68-107 for the 8 heads (y = logit
 * / T, softmax, nucleus with the threshold p and the draw of that position: pb_nucleus_rows' arithmetic order) and writes the next decoder
 * input on the device, so tokens are enqueued back to back, 8 per hipGraph replay. The kernel also writes the raw logits row and its 8 ids
 * to pinned host logs indexed by position: the HOST stays the authority -- it replays each position from the logged row through the
 * reference code path (CPU softmax + nucleus, consuming the global RNG stream exactly as before) and, on the rare position where the
 * device's softmax rounding made another choice, rewinds the decoder (pb_decoder_seek) and continues from its own token. Results are
 * therefore bit-identical to the per-token loop (tests/test_model_gpu.py) whatever the device sampled.
 *   pb_decoder_sampler_init  temperatures, thresholds, class counts and logits offsets of the 8 heads, u = (S, 8) f64 draws (host; copied),
 *                            fault_period > 0 corrupts head 0's id at every fault_period-th position (tests of the rewind path only).
 *   pb_decoder_launch        enqueue the next ntok tokens (first_tok8, 8 host ids or NULL, is copied up as the first decoder input);
 *                            returns a ticket >= 0 for pb_decoder_wait, or < 0.
 *   pb_decoder_wait          block until that run's tokens are decoded and logged.
 *   pb_decoder_logs          the pinned logs: (S, vocab) f32 logits rows, (S, 8) int16 device-sampled ids.
 *   pb_decoder_seek          drain, then: last decoded position = pos, decoder input of position pos + 1 = tok8. */
int pb_decoder_sampler_init(void* dec, const float* temps8, const float* p8, const int32_t* n8, const int32_t* off8, const double* u,
                            int64_t n_u, int32_t fault_period);
int pb_decoder_launch(void* dec, int32_t ntok, const int16_t* first_tok8);
int pb_decoder_wait(void* dec, int32_t ticket);
int pb_decoder_logs(void* dec, float** logits_rows, int16_t** tok_rows);
int pb_decoder_seek(void* dec, int32_t pos, const int16_t* tok8);

/* ---- fused attention of the "bf16x3" parity instantiation (round 6, ABI 8): f32 q / k / v / o, every product a split-bf16 triple on the
 * bf16 matrix cores (see PB_F32X3), softmax in f32 -- instead of the unfused QK^T -> softmax -> PV chain of the exact-f32 path
 * (modeling_bart.py:115-140 through PianoBart.py:76), whose (B, H, S, S) f32 matrices dominate that path's HBM time. Same masks and
 * conventions as pb_flash_fwd / pb_flash_bwd (key padding row per batch, bit 0 of `causal`, zero output row + lse = +inf for a query
 * without a visible key); strides in ELEMENTS, multiples of 4; head_dim 32 / 64 / 128 (pb_flash_x3_supported). kmax (may be NULL): per batch
 * row 1 + the last visible key (pb_key_extent of the mask): key tiles behind it are skipped, results unchanged. delta: (B, H, Sq) f32 scratch,
 * written by the call. */
int pb_flash_x3_supported(int32_t hd);
int pb_flash_fwd_x3(const float* q, const float* k, const float* v, float* o, float* lse, const float* key_mask, const int32_t* kmax, int32_t B, int32_t H, int32_t Sq,
                    int32_t Sk, int32_t hd, int64_t q_sb, int64_t q_ss, int64_t k_sb, int64_t k_ss, int64_t v_sb, int64_t v_ss, int64_t o_sb,
                    int64_t o_ss, float scale, int32_t causal, void* stream);
int pb_flash_bwd_x3(const float* q, const float* k, const float* v, const float* o, const float* dout, const float* lse, const float* key_mask,
                    const int32_t* kmax, float* dq, float* dk, float* dv, float* delta, int32_t B, int32_t H, int32_t Sq, int32_t Sk, int32_t hd, int64_t q_sb,
                    int64_t q_ss, int64_t k_sb, int64_t k_ss, int64_t v_sb, int64_t v_ss, int64_t o_sb, int64_t o_ss, int64_t dq_sb,
                    int64_t dq_ss, int64_t dk_sb, int64_t dk_ss, int64_t dv_sb, int64_t dv_ss, float scale, int32_t causal, void* stream);

/* the same on packed rows (the layout of pb_flash_fwd_packed / pb_flash_bwd_packed: per-sequence q_off / q_len / k_off / k_len and the visible-key count k_vis
 * instead of a key mask; Sq_max / Sk_max = the longest sequence) */
int pb_flash_fwd_x3_packed(const float* q, const float* k, const float* v, float* o, float* lse, const int32_t* q_off, const int32_t* q_len, const int32_t* k_off,
                           const int32_t* k_len, const int32_t* k_vis, int32_t B, int32_t H, int32_t Sq_max, int32_t Sk_max, int32_t hd, int64_t q_ss,
                           int64_t k_ss, int64_t v_ss, int64_t o_ss, float scale, int32_t causal, void* stream);
int pb_flash_bwd_x3_packed(const float* q, const float* k, const float* v, const float* o, const float* dout, const float* lse, float* dq, float* dk, float* dv,
                           float* delta, const int32_t* q_off, const int32_t* q_len, const int32_t* k_off, const int32_t* k_len, const int32_t* k_vis,
                           int32_t B, int32_t H, int32_t Sq_max, int32_t Sk_max, int32_t hd, int64_t q_ss, int64_t k_ss, int64_t v_ss, int64_t o_ss,
                           int64_t dq_ss, int64_t dk_ss, int64_t dv_ss, float scale, int32_t causal, void* stream);

/* ---- K15: deferred parameter-gradient reductions -----------------------------------------------------------------------
 * The bias / LayerNorm-parameter gradients of one backward pass (the `db = grad.sum(0)` of every nn.Linear and nn.LayerNorm autograd
 * node under BartModel, modeling_bart.py:280-390) leave their kernels as per-workgroup partial rows. Between pb_defer_begin and
 * pb_defer_flush (same host thread) pb_add_ln_bwd, pb_gemm (colsum_out) and pb_flash_bwd (dbias_*) keep those rows in `arena`
 * instead of reducing them one small launch at a time, and pb_defer_flush sums all of them in ONE launch, in a fixed order
 * (bit-reproducible). arena: device floats, 16-byte aligned; table: device bytes, table_entries * pb_defer_desc_bytes(). When either
 * is full the calls fall back to the immediate reduction. Outputs are accumulated (+=) exactly as without deferral, so they must
 * not be read, nor written by anything else, before the flush. */
int pb_defer_begin(float* arena, int64_t arena_floats, void* table, int32_t table_entries);
int pb_defer_flush(void* stream);

/* ---- events for ordering two streams of one GPU (Engine's second stream) ---------------------------
 * mode bit 0: hipEventDisableSystemFence, bit 1: hipEventReleaseToDevice (both on top of hipEventDisableTiming): the producer and
 * the consumer are kernels on the same device, so the system-scope write-back / invalidate of a default event is not needed.
 * The handle is an opaque hipEvent_t. */
int pb_event_create(void** ev, int32_t mode);
int pb_event_destroy(void* ev);
int pb_event_record(void* ev, void* stream);
int pb_stream_wait_event(void* stream, void* ev);
int32_t pb_defer_desc_bytes(void);

#ifdef __cplusplus
}
#endif
#endif
