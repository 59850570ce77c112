/* vit_amd.h -- C ABI of libvit_amd.so: the MI355X (gfx950) kernels behind the ViT training hot path of ViskaWei/VIT.
 *
 * The reference has no FFI layer: its boundary is the Python object surface that Lightning calls
 * (SURVEY.md section 8b).  This header is therefore the boundary the build's own host side (the vit_amd Python package, which
 * mirrors that surface) binds through ctypes; INTEGRATION.md shows the stub a maintainer of the reference would add.
 * Each entry point cites the reference code whose arithmetic it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (torch); nothing is allocated inside except the handle;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*); no call synchronises the device;
 *   - return value: VIT_OK or a negative vit_status; vit_last_error() returns a thread-local message;
 *   - `dtype` arguments are vit_dtype codes; row-major tensors; "ld*" = leading dimension in ELEMENTS;
 *   - alignment: every base pointer 16-byte aligned, every leading/inner dimension of a bf16 GEMM operand a multiple
 *     of 8 elements (of an f32 one: 4), checked on the host before any launch (VIT_ERR_ARG otherwise);
 *   - dropout masks are a pure function of (seed, site, row, col): forward and backward regenerate them, nothing is
 *     stored;
 *   - state: the handle (workspace pointer, optional per-step state pointer, launch geometry: vit_handle_set_option) and
 *     the process-wide kernel-selection knobs of vit_set_option (A/B switches between bit-identical kernel forms); nothing
 *     else persists between calls.
 */
#ifndef VIT_AMD_H_
#define VIT_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VIT_AMD_VERSION 100

typedef struct vit_ctx* vit_handle;
typedef void* vit_stream; /* hipStream_t */

typedef enum { VIT_OK = 0, VIT_ERR_ARG = -1, VIT_ERR_HIP = -2, VIT_ERR_UNSUPPORTED = -3, VIT_ERR_WORKSPACE = -4 } vit_status;
typedef enum { VIT_F32 = 0, VIT_BF16 = 1 } vit_dtype;
/* GELU: C = gelu(acc + bias), aux_out (optional) = the pre-activation.  DGELU: C = acc * gelu'(aux_in).
 * GELU_GRAD: as GELU but aux_out = gelu'(pre-activation) -- the forward already holds Phi and exp(-x^2/2), so saving the
 * derivative costs two FMAs there and the backward GEMM (MUL_AUX: C = acc * aux_in) needs no transcendental at all. */
typedef enum { VIT_ACT_NONE = 0, VIT_ACT_GELU = 1, VIT_ACT_DGELU = 2, VIT_ACT_GELU_GRAD = 3, VIT_ACT_MUL_AUX = 4 } vit_act;
typedef enum { VIT_LOSS_MSE = 0, VIT_LOSS_L1 = 1, VIT_LOSS_CE = 2 } vit_loss;

int vit_version(void);
const char* vit_last_error(void);

/* Handle: device id + a caller-owned workspace (split-K slabs, reduction partials). */
int vit_create(vit_handle* out, int device);
int vit_destroy(vit_handle h);
int vit_set_workspace(vit_handle h, void* ws, size_t bytes);
/* Process-wide tuning / diagnostics knobs (never change results beyond rounding order):
 *   "gemm_core": 0 = generic 128x128 core only, 1 = automatic (default): tile-aligned problems (M, N multiples of 256, K of
 *                64) run the 256x256x64 ping-pong core (wave halves one barrier out of phase: LOAD segment beside MFMA
 *                segment, ring of 8 half-tiles, 4 in flight); 5 = the same choice named explicitly.
 *   "attn_split": workgroups per (batch, head) in the resident attention kernels (T <= 256), default 2.
 *   "attn_bwd_fused": attention backward form: non-zero (default 4) = the pair-pipelined single kernel where it fits (head_dim
 *                64, 64 <= T <= 208), the dQ + dK/dV pair elsewhere; 0 = the dQ + dK/dV pair everywhere.  (The values 1 .. 3
 *                named the single-kernel forms of round 2, removed in round 4; they are accepted and mean the default.)
 *   "attn_bwd_dma": 1 (default) = the dQ + dK/dV pair at head_dim 64 stages its LDS images by LDS-DMA in reading order with
 *                per-tile counted waits (the first tile's arithmetic starts when 16 KiB have landed); 0 = register-staged
 *                images, all in before the loop starts.  Same results bit for bit.
 *   "attn32_mfma": 1 (default) = fp32 attention (precision '32') at head_dim 64 runs on the f32-input matrix instructions
 *                (v_mfma_f32_16x16x4_f32: exact f32 products and accumulation); 0 = the one-wave-per-row vector kernels.
 *   "gemm_ngroups": 1 (default) = XCDs 0-3 / 4-7 walk the lower / upper half of the N-tiles when the weights exceed an L2.
 *   "attn_res_max_t": longest sequence the resident attention kernels take (default 592 = what fits the LDS at head_dim
 *                64); longer ones, or everything with 0, go to the tiled kernels.
 *   "gemm_half_tail": 1 = the tiles of a partial last round of a multi-round ping-pong GEMM (bias/dropout -> bf16
 *                and plain dX epilogues; 2 = the GELU epilogue too) run in a second launch as half tiles, two workgroups per
 *                tile; 0 (default since r05) = one launch: the half tiles shorten that product but cost more CU-time, and
 *                inside the power-limited training step the single launch is faster overall.  Same results bit for bit.
 *   "reserve_cus": 0 (default) .. 128 = the one-workgroup-per-CU kernels (ping-pong GEMMs, pair-pipelined attention backward) size
 *                their grids for that many fewer CUs, leaving room for a collective's kernels that overlap them (data-parallel
 *                runs; bench.py --reserve-cus).
 *   "attn_fwd_waves": 12 (default) or 8 = most waves per workgroup of the resident attention forward; only changes the launch
 *                where one workgroup fills the LDS (head_dim 64, T > ~290): 577 tokens run as 2 x 10 waves instead of 3 x 7.
 *   "gemm_split_tail": 1 (default) = a SHORT tail (at most a quarter of the workgroup slots) of a long-K product runs as K-slices of
 *                whole tiles + a small reduce-and-epilogue kernel instead of half tiles (ViT-L: 36 of 292 tiles at K = 3072 /
 *                4096); needs 4 x slices x 65536 bytes of workspace per tail tile, else the half-tile launch is used; 0 = never.
 *   "gemm_balance_wgs": 1 (default) = a multi-round ping-pong GEMM launches ceil(tiles / rounds) workgroups instead of 256
 *                (same makespan in tile-times, idle CUs instead of CUs that idle for the last round); 0 = always 256.
 *   "gemm_pp_slots": 8 (the only value since round 2: the 10-slot ring, 96 KiB of operand loads in flight per CU, measured
 *                0-15 % slower on the ViT-B shapes and was removed with the K-loop rewrite; any other value is VIT_ERR_ARG).
 *   (Timing diagnostics that switch pieces of a kernel off are compile-time variant builds -- python -m vit_amd.build
 *   --defs ... --tag ... -- never a switch of this library: results are meaningless in such a build.)
 *                Returns VIT_ERR_ARG for an unknown name. */
int vit_set_option(const char* name, int value);
/* Per-handle launch geometry: what changes HOW MANY workgroups a call through this handle launches belongs to the handle, so
 * that two engines of one process (training + evaluation, an engine's second-stream handle) do not depend on the order
 * in which they were configured.  "reserve_cus": -1 (default) = follow the process-wide value above, 0 .. 128 = this handle's
 * own.  The remaining vit_set_option knobs select between bit-identical kernel forms for A/B runs and stay process-wide.
 * Reference: none (Lightning's 'ddp' overlaps NCCL with kernels that do not own whole SMs, src/hardware_utils.py:86-95). */
int vit_handle_set_option(vit_handle h, const char* name, int value);

/* Per-step state in device memory, for a training step captured as a hipGraph (HIP streams and graphs instead of a tracing
 * compiler: kernel arguments are frozen at capture, so what changes from step to step must be read from memory).
 * `state`: 32 bytes, 16-byte aligned: { u32 key0, key1; f32 lr, bc1, rsqrt_bc2; u32 step; u32 pad[2] }, caller-owned.
 * While bound (NULL unbinds), every call through this handle that takes (dropout_p, seed, site) XORs the state's keys
 * into its dropout keys at kernel entry: the masks of a replay are those of (seed, site, step).  vit_step_advance (one
 * tiny kernel, first node of the captured step) does step += 1, derives the keys from (base_seed, step) and AdamW's
 * bias corrections from step; `lr` is the host's to write (a scheduler changes it between replays).
 * vit_adamw_step_dyn = vit_adamw_step with lr / bias corrections read from the state.
 * Replaces what Lightning's loop hands torch per step: the generator advance behind nn.Dropout and optimizer.step()'s
 * step count (basemodule.py:230-251). */
int vit_step_state_bind(vit_handle h, void* state);
int vit_step_advance(vit_handle h, uint64_t base_seed, float beta1, float beta2, vit_stream stream);
int vit_adamw_step_dyn(vit_handle h, float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float beta1,
                       float beta2, float eps, float weight_decay, const float* sqnorm, float max_norm, vit_stream stream);

/* ------------------------------------------------------------------------------------------------ GEMM
 * C = epilogue(alpha * op(A) * op(B)), bf16 MFMA (v_mfma_f32_16x16x32_bf16), fp32 accumulate.
 * Replaces every nn.Linear on the path: tokenization.py:41,50 (patch projection); HF ViTSelfAttention
 * query/key/value (restated at vit_with_rope.py:54-56), ViTSelfOutput.dense, ViTIntermediate.dense, ViTOutput.dense
 * and their autograd backward (dX = dY*W, dW = dY^T*X).
 *   a_trans = 0: A stored [M][K] (lda >= K)      a_trans = 1: A stored [K][M] (lda >= M)
 *   b_trans = 0: B stored [N][K] (ldb >= K; the nn.Linear weight layout)   b_trans = 1: B stored [K][N]
 * Epilogue, in this order, on the fp32 accumulator v of element (m, n):
 *   v = alpha*v + bias[n];  ACT_GELU: (aux_out[m,n] = bf16(v) if aux_out), v = gelu_erf(v);
 *   ACT_DGELU: v *= gelu_erf'(aux_in[m,n]);  dropout(p, seed, site) as a function of (out_row, n);
 *   v += residual[out_row, n];  C[out_row, n] = (c_dtype) v
 * with out_row = (m / rows_per_batch) * out_batch_rows + (m % rows_per_batch) + out_row_offset when
 * rows_per_batch > 0 (patch-embed writes token rows 1..N of each sample, embedding.py:87-88), else out_row = m.
 * split_k > 1 accumulates partial products in the handle's workspace and reduces them deterministically (no float
 * atomics: the reference runs with deterministic=True, basemodule.py:250); only alpha and C (f32) apply then.
 */
typedef struct vit_gemm_desc {
  int M, N, K;
  int a_trans, b_trans;
  int ab_dtype;            /* VIT_BF16 (VIT_F32 operands: split-bf16 "x3" mode) */
  const void* A; int64_t lda;
  const void* B; int64_t ldb;
  void* C; int64_t ldc; int c_dtype;
  float alpha;
  const float* bias;       /* [N] or NULL */
  int act;                 /* vit_act */
  void* aux_out;           /* ACT_GELU: optional bf16 [M,N] pre-activation (ld = ldaux) */
  const void* aux_in;      /* ACT_DGELU: bf16 [M,N] pre-activation (ld = ldaux) */
  int64_t ldaux;
  float dropout_p; uint64_t seed; uint64_t site;
  const float* residual; int64_t ldres;   /* f32, indexed by out_row */
  int rows_per_batch, out_batch_rows, out_row_offset;
  int split_k;             /* 0/1 = off; >1 = that many K slices; -1 = choose */
  int accumulate;          /* split_k path: C += result instead of C = result */
  float* colsum_out;       /* optional f32 [N]: column sums of C as stored (a Linear's bias gradient when C is the gradient
                            * of its output); summed inside the epilogue where the kernel can, else by a vit_colsum pass */
  /* Rotary position embedding of the fused QKV projection's output (src/models/vit_with_rope.py:58-60, rope.py:116-131):
   * columns [0, rope_cols) of C are heads of rope_dh columns whose element i pairs with element i + rope_dh / 2, row m is token
   * m % rope_T; cos / sin: f32 [rope_T, rope_dh / 2] (vit_rope_qk's tables).  NULL = off.  The ping-pong core rotates in its
   * epilogue, on the f32 values before the one rounding to bf16 (rope_dh 16 / 32 / 64, rope_cols a multiple of 64); every
   * other case runs vit_rope_qk on C right after the product -- same contract either way, PROVIDED the tables are the
   * reference's: row t holds the angles t * theta_i, linear in t and zero-based (rope.py:36-56).  The rotating epilogue reads
   * eight table rows per tile and steps a lane's later rows by the angle-addition recurrence over table row 8, so a table of
   * other positions (offset, scaled, learned) is NOT supported through this descriptor: rotate with vit_rope_qk, which reads
   * the table per row.  dropout_p must be 0 with rope (the reference rotates before any dropout). */
  const float* rope_cos; const float* rope_sin;
  int rope_T, rope_dh, rope_cols;
} vit_gemm_desc;
int vit_gemm(vit_handle h, const vit_gemm_desc* d, vit_stream stream);
/* Symbol (as rocprofv3 prints it, without the "void vit::" prefix and argument list) of the kernel the calling thread's
 * last vit_gemm launched: lets a benchmark attribute its per-call timings to the kernels a profiler lists. */
const char* vit_last_gemm_kernel(void);

/* Convenience forms named after SURVEY.md section 8b.  x:[M,K] bf16, W:[N,K] bf16 (nn.Linear layout), y:[M,N]. */
int vit_linear_fwd(vit_handle h, const void* x, const void* W, const float* bias, void* y, int y_dtype, int M, int N,
                   int K, int act, void* aux_out, float dropout_p, uint64_t seed, uint64_t site,
                   const float* residual, vit_stream stream);
/* dX[M,K] = dY[M,N] * W[N,K]  (optionally *= gelu'(aux_in)) */
int vit_linear_bwd_dx(vit_handle h, const void* dy, const void* W, void* dx, int dx_dtype, int M, int N, int K,
                      const void* dgelu_aux_in, vit_stream stream);
/* dW[N,K] (f32) (+)= dY[M,N]^T * X[M,K], deterministic split-K over M */
int vit_linear_bwd_dw(vit_handle h, const void* dy, const void* x, float* dW, int M, int N, int K, int accumulate,
                      vit_stream stream);

/* ------------------------------------------------------------------------------------------- LayerNorm
 * HF nn.LayerNorm(hidden, eps=layer_norm_eps=1e-12) (builder.py:250): layernorm_before/after and vit.layernorm.
 * x: f32 [rows, D] (the residual stream is kept in fp32); y: y_dtype [rows, D]; mean/rstd: f32 [rows] (saved for bwd).
 */
int vit_layernorm_fwd(vit_handle h, const float* x, const float* gamma, const float* beta, void* y, int y_dtype,
                      float* mean, float* rstd, int rows, int D, float eps, vit_stream stream);
/* The same LayerNorm over x + delta, where delta (delta_dtype [rows, D]) is the output of the Linear underneath a
 * "dropout(Linear(.)) + residual" (HF ViTLayer: attention_output + hidden_states; ViTOutput: hidden_states + input_tensor),
 * and xsum (f32 [rows, D]) receives the sum: the residual add rides on the pass that reads the stream anyway, so the
 * projection GEMMs write bf16 and never load. */
int vit_layernorm_fwd_residual(vit_handle h, const float* x, const void* delta, int delta_dtype, float* xsum,
                               const float* gamma, const float* beta, void* y, int y_dtype, float* mean, float* rstd,
                               int rows, int D, float eps, vit_stream stream);
/* dx[rows,D] (f32) = LN'(dy) (+ dres if not NULL: the residual branch's gradient); dgamma/dbeta (f32 [D]) are
 * reduced deterministically through the workspace; accumulate!=0 adds into them. dy: dy_dtype [rows, D]. */
int vit_layernorm_bwd(vit_handle h, const void* dy, int dy_dtype, const float* x, const float* gamma,
                      const float* mean, const float* rstd, const float* dres, float* dx, float* dgamma,
                      float* dbeta, int rows, int D, vit_stream stream);

/* Same, fused with what always follows it in this path: also writes dyn (bf16 [rows, D]) = dropout_mask(seed, site) *
 * dx -- the gradient wrt the output of the Linear under "dropout(Linear(.)) + residual" -- and dbias (f32 [D]) = its
 * column sums (that Linear's bias gradient): one pass instead of vit_dropout_bwd_cast + vit_colsum re-reading dx. */
int vit_layernorm_bwd_fused(vit_handle h, const void* dy, int dy_dtype, const float* x, const float* gamma,
                            const float* mean, const float* rstd, const float* dres, float* dx, float* dgamma,
                            float* dbeta, int rows, int D, void* dyn, int dyn_dtype, float* dbias, float dropout_p,
                            uint64_t seed, uint64_t site, vit_stream stream);

/* ------------------------------------------------------------------------------------------- Attention
 * softmax(Q K^T * scale) -> dropout -> * V, per (batch, head); flash-style (scores never reach HBM).
 * Replaces HF ViTSelfAttention's eager arithmetic as restated in vit_with_rope.py:63-81 (no logit clamp exists in
 * the reference: SURVEY.md section 0.4).  qkv: bf16 [B*T, 3*H*dh] token-major, columns [q | k | v], head h at
 * columns h*dh (the fused QKV projection's natural output; equals view(B,T,H,dh).transpose(1,2) of each third).
 * ctx: bf16 [B*T, H*dh] (= context_layer after transpose+view, vit_with_rope.py:75-78); lse: f32 [B*H, T]
 * (log-sum-exp of the scaled scores, saved for backward).
 * io_dtype = VIT_BF16: the MFMA flash kernels (qkv / ctx / dctx / dqkv bf16).  io_dtype = VIT_F32: the same tensors in
 * f32 and exact fp32 arithmetic (the precision='32' path; T <= 4096).
 */
int vit_attention_fwd(vit_handle h, const void* qkv, void* ctx, float* lse, int io_dtype, int B, int H, int T, int dh,
                      float scale, float dropout_p, uint64_t seed, uint64_t site, vit_stream stream);
/* dqkv: bf16 [B*T, 3*H*dh] from dctx: bf16 [B*T, H*dh]; recomputes probabilities from lse. delta: f32 [B*H, T]
 * scratch (rowsum(dctx*ctx)), written by this call. */
int vit_attention_bwd(vit_handle h, const void* qkv, const void* ctx, const void* dctx, const float* lse,
                      float* delta, void* dqkv, int io_dtype, int B, int H, int T, int dh, float scale,
                      float dropout_p, uint64_t seed, uint64_t site, vit_stream stream);
/* Same, and dqkv_colsum (f32 [3*H*dh]) = the column sums of dqkv as stored: the bias gradient of the fused QKV projection
 * (without RoPE; with it the sums must be taken after the inverse rotation).  The resident bf16 kernels sum their own rows
 * on the way out (one partial row per wave through the workspace); other paths run vit_colsum afterwards. */
int vit_attention_bwd_colsum(vit_handle h, const void* qkv, const void* ctx, const void* dctx, const float* lse,
                             float* delta, void* dqkv, int io_dtype, int B, int H, int T, int dh, float scale,
                             float dropout_p, uint64_t seed, uint64_t site, float* dqkv_colsum, vit_stream stream);
/* Training pair with the context kept to ~16 mantissa bits.  ctx_lo (bf16 [B*T, H*dh], may be NULL = the calls above):
 * the forward also stores the rounding residual ctx_exact - bf16(ctx_exact); the backward forms
 * delta = rowsum(dctx * (ctx + ctx_lo)).  Why: the softmax backward is dS = P o (dP - delta); the reference (autocast,
 * vit_with_rope.py:63-81 under basemodule.py:233) takes delta = sum_j P_j dP_j in fp32, which cancels exactly against dP.
 * rowsum(dctx * ctx) with an 8-bit ctx is off by an amount COMMON to a score row, which survives the sum over keys in dQ / dK
 * once token representations share a large common component (deep layers): 5e-2 on ViT-L's late query weights against
 * the reference's own bf16 1.1e-2; with the residual 1e-2 or better (tests/test_parity_deep_gpu.py).  dqkv_colsum may be
 * NULL.  io_dtype = VIT_F32 ignores ctx_lo. */
int vit_attention_fwd_lo(vit_handle h, const void* qkv, void* ctx, void* ctx_lo, float* lse, int io_dtype, int B, int H,
                         int T, int dh, float scale, float dropout_p, uint64_t seed, uint64_t site, vit_stream stream);
int vit_attention_bwd_lo(vit_handle h, const void* qkv, const void* ctx, const void* ctx_lo, const void* dctx,
                         const float* lse, float* delta, void* dqkv, int io_dtype, int B, int H, int T, int dh, float scale,
                         float dropout_p, uint64_t seed, uint64_t site, float* dqkv_colsum, vit_stream stream);
/* Attention probabilities [B, H, T, T] f32 (eval-mode, for output_attentions=True: specvit.py:92-93). */
int vit_attention_probs(vit_handle h, const void* qkv, float* probs, int io_dtype, int B, int H, int T, int dh,
                        float scale, vit_stream stream);

/* ------------------------------------------------------------------------------- Embedding-side kernels
 * x.unfold(1,P,S) (+ zero pad of the ragged tail patch) -> patches [B*N, P] (bf16 or f32)   (tokenization.py:45-49) */
int vit_unfold_cast(vit_handle h, const float* x, void* patches, int out_dtype, int B, int L, int P, int S, int N,
                    vit_stream stream);
/* Backward of vit_unfold_cast wrt the signal, for a trainable input preprocessor (src/models/preprocessor.py:96-111 in
 * front of src/models/tokenization.py:43-69): dx[B, L] (f32) = overlap-add of dpatches[B*N, P] (f32) over the windows that
 * lie inside the signal.  Gather form, no atomics. */
int vit_fold_add(vit_handle h, const float* dpatches, float* dx, int B, int L, int P, int S, int N, vit_stream stream);

/* Training-time noise injection of ViTLModule.training_step (src/vit.py:86-88): out = flux + randn_like(flux) * error *
 * noise_level over n f32 elements (n % 4 == 0; out may alias flux).  The normal variates come from a counter-based
 * generator keyed on (seed, element index); the reference's come from torch's generator, whose stream is
 * implementation-defined, so only the distribution is contractual. */
int vit_add_noise(vit_handle h, const float* flux, const float* error, float* out, long n, float noise_level,
                  uint64_t seed, vit_stream stream);

/* Rotary position embedding of ViTSelfAttentionWithRoPE (src/models/vit_with_rope.py:58-60 -> src/models/rope.py:66-131,
 * model.pos_encoding_type: 'rope'): rotates the query and key thirds of the token-major qkv buffer ([rows, ld >= 3*H*dh],
 * rows = B*T, position = row % T) in place, pairing element i of a head with element i + dh/2 (_rotate_half).
 * cos_half / sin_half: f32 [T, dh/2] -- the first half of the reference's cached tables (rope.py:44-56), built on the
 * host by the reference's own formula.  inverse != 0 rotates by the negative angle: the backward of the rotation,
 * applied to dq / dk before the projection's weight and input gradients. */
int vit_rope_qk(vit_handle h, void* qkv, int dtype, const float* cos_half, const float* sin_half, long rows, int T,
                int H, int dh, long ld, int inverse, vit_stream stream);

/* rows 0 of every sample <- cls_token (embedding.py:87-88); optional "+ position_embeddings" (embedding.py:95-97)
 * and the embedding dropout (embedding.py:100) over the whole [B, T, D] f32 token tensor, in place. */
int vit_embed_finish(vit_handle h, float* tokens, const float* cls, const float* pos, int B, int T, int D,
                     float dropout_p, uint64_t seed, uint64_t site, vit_stream stream);
/* backward of the two above: dtokens [B,T,D] f32 -> dpatch_out bf16 [B*N, D] (dropout mask applied), dcls [D],
 * dpos [T,D] or NULL. */
int vit_embed_finish_bwd(vit_handle h, const float* dtokens, void* dpatch_out, int dpatch_dtype, float* dcls, float* dpos,
                         int B, int T, int D, float dropout_p, uint64_t seed, uint64_t site, int accumulate,
                         vit_stream stream);

/* ------------------------------------------------------------------------------- Elementwise / reductions
 * dy ([rows, cols], bf16 or f32) = dropout_mask(seed,site) * dx (f32): the gradient of "dropout(y) + residual" wrt y */
int vit_dropout_bwd_cast(vit_handle h, const float* dx, void* dy, int dy_dtype, int rows, int cols, float dropout_p,
                         uint64_t seed, uint64_t site, vit_stream stream);
/* out[cols] (f32) (+)= column sums of a [rows, cols] tensor (bias gradients), deterministic two-stage */
int vit_colsum(vit_handle h, const void* a, int a_dtype, int64_t lda, float* out, int rows, int cols, int accumulate,
               vit_stream stream);
/* f32 -> bf16 copy (weights after an optimizer step, inputs) */
int vit_cast_f32_bf16(vit_handle h, const float* src, void* dst, int64_t n, vit_stream stream);
/* dst (f32) = scale * src (bf16): the receive side of the optional bf16 gradient exchange (`train.ddp_grad_dtype: bf16`;
 * the reference's DDP exchanges fp32 gradients, src/hardware_utils.py:86-95 -- SURVEY.md section 5 prices the halved bytes) */
int vit_cast_bf16_f32(vit_handle h, const void* src, float* dst, int64_t n, float scale, vit_stream stream);

/* ------------------------------------------------------------------------------------- Head + loss
 * logits[B, C] = cls_rows * W^T + b, cls_rows = last_hidden[:, 0, :] (specvit.py:78-81); loss (specvit.py:83-89):
 * MSE / L1 over logits.view(-1) vs labels.view(-1) (mean), or cross-entropy over int64 labels (mean).
 * last_hidden: f32 [B, T, D]; loss_out: f32 [1]. labels may be NULL (loss_out untouched). */
int vit_head_loss_fwd(vit_handle h, const float* last_hidden, const float* W, const float* b, const void* labels,
                      float* logits, float* loss_out, int B, int T, int D, int C, int loss_kind, vit_stream stream);
/* d(last_hidden) (zero except the CLS rows), dW [C,D], db [C] given dloss (f32 [1], the upstream scalar gradient). */
int vit_head_loss_bwd(vit_handle h, const float* last_hidden, const float* W, const float* logits, const void* labels,
                      const float* dloss, float* dlast_hidden, float* dW, float* db, int B, int T, int D, int C,
                      int loss_kind, int accumulate, vit_stream stream);

/* ------------------------------------------------------------------------------------- Optimizer
 * Global L2 norm (squared) of a flat f32 gradient buffer -> out[0] (Lightning gradient_clip_val=0.5, norm clipping:
 * basemodule.py:244). Deterministic two-stage reduction. */
int vit_grad_sqnorm(vit_handle h, const float* g, int64_t n, float* out, vit_stream stream);
/* out[0] += the squared norm of g: parameters that live outside the model's flat buffer (a trainable input
 * preprocessor) join the same global clipping norm. */
int vit_grad_sqnorm_acc(vit_handle h, const float* g, int64_t n, float* out, vit_stream stream);
/* torch.optim.AdamW step (opt/optimizer.py:16,108: lr, weight_decay=0 by default) over a flat parameter buffer, with
 * the clip coefficient min(1, max_norm / (sqrt(*sqnorm) + 1e-6)) applied to g on the fly (sqnorm may be NULL).
 * Also refreshes the bf16 shadow copy used by the GEMMs (p_bf16 may be NULL). bias_correction uses `step` (1-based). */
int vit_adamw_step(vit_handle h, float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr,
                   float beta1, float beta2, float eps, float weight_decay, int step, const float* sqnorm,
                   float max_norm, vit_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* VIT_AMD_H_ */
