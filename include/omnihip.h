/* libomnihip - C ABI of the MI355X (gfx950) kernels behind the Omni-Video 2 sampling path.
 *
 * The reference has no FFI layer of its own: its hot path crosses into native code only
 * through Python entry points of third-party wheels (flash_attn, torch/cuDNN/cuBLAS).
 * Each entry point below names the reference call site (path:line under the reference
 * tree) whose device work it replaces.
 *
 * Conventions
 *   - every pointer is a borrowed DEVICE pointer (bf16 tensors are raw uint16_t bit
 *     patterns); nothing is retained after the call returns; no entry point allocates,
 *     frees or synchronises, so all of them may be captured into a HIP graph (the one
 *     exception, marked SET-UP at its declaration: the create / map / destroy functions of
 *     the peer-mapped exchange buffers, ovd_peer_alloc ... ovd_peer_close);
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *   - return value 0 = launched; non-zero = rejected before any launch, with the reason
 *     in ovd_last_error() (thread-local, valid until the next call on that thread);
 *   - strides (`ld*`, `*_bs`) are in ELEMENTS;
 *   - two helpers do no GPU work and are marked HOST at their declaration
 *     (ovd_resize_coeffs fills host tables, ovd_flash_attn_workspace_bytes returns a size).
 */
#ifndef OMNIHIP_H
#define OMNIHIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* ovd_last_error(void);
/* ABI version of this header; bumped on any signature change. */
int ovd_abi_version(void);

/* LayerNorm over the last dim (fp32 statistics, biased variance, eps inside the sqrt),
 * optional bf16 affine (weight, bias both or neither), optional AdaLN modulation
 * y = n * (1 + scale[c]) + shift[c] (fp32 rows of length C, both or neither).
 * round_norm != 0 rounds the normalised value to bf16 before the modulation (the case of
 * a bf16 input tensor).  C must be a multiple of 256.
 * Replaces WanLayerNorm.forward (omnivideo/modules/model.py:92-97) fused with the
 * modulation in WanAttentionBlock.forward (model.py:243,250,252) and Head.forward
 * (model.py:287-289); also nn.LayerNorm in VisualContextAdapter.forward
 * (omnivideo/modules/visual_context_adapter.py:111,114). */
int ovd_layernorm_modulate_bf16(const float* x, int64_t ldx, const float* shift, const float* scale,
                                const uint16_t* weight, const uint16_t* bias, uint16_t* out, int64_t ldo,
                                int M, int C, float eps, int round_norm, void* stream);
int ovd_layernorm_modulate_f32(const float* x, int64_t ldx, const float* shift, const float* scale,
                               const uint16_t* weight, const uint16_t* bias, float* out, int64_t ldo,
                               int M, int C, float eps, int round_norm, void* stream);

/* The same LayerNorm (+ affine) (+ modulation) with the residual update that precedes it in the block
 * applied on the way in: x[m][c] += res[m][c] * res_gate[c] (res_gate NULL = 1), x written back, then
 * normalised.  res is the bf16 output of the layer's last Linear (ovd_gemm_bf16, epilogue 0), so
 * `x = x + y * e` of WanAttentionBlock.forward (omnivideo/modules/model.py:245-246, 250, 253-254) and the
 * norm that follows (:243, :250, :252; Head.forward :287-289) make one pass over x instead of two.
 * Results equal epilogue 3 of ovd_gemm_bf16 followed by ovd_layernorm_modulate_* bit for bit.
 * out_kind: 0 = bf16 out, 1 = fp32 out, 2 = e4m3 out + row_scale[M] (ovd_layernorm_modulate_fp8). */
int ovd_resid_layernorm_modulate(float* x, int64_t ldx, const uint16_t* res, int64_t ldr, const float* res_gate,
                                 const float* shift, const float* scale, const uint16_t* weight, const uint16_t* bias,
                                 void* out, int64_t ldo, int out_kind, float* row_scale, int M, int C, float eps,
                                 int round_norm, void* stream);

/* RMSNorm over the FULL row (all heads) with bf16 weight, then optional 3-axis
 * interleaved-pair rotary embedding, bf16 in / bf16 out (in-place allowed).
 * rope: float2 (cos,sin) table [3][rope_rows][D/2] or NULL for no rotation; the token
 * index of a row is (row % rows_per_sample) + tok_offset, decomposed over the patch grid
 * (gf,gh,gw); rows with token index >= n_rope are not rotated.  weight == NULL skips
 * the normalisation (rotation only).
 * Replaces WanRMSNorm.forward (model.py:76-84) at model.py:137-138,169-170 and
 * unified_model.py:201, plus rope_apply (model.py:37-65; sequence-parallel variant
 * omnivideo/distributed/full_sequence_parallel.py:22-60 via tok_offset). */
int ovd_rmsnorm_rope_bf16(const uint16_t* in, int64_t ldi, const uint16_t* weight, const float* rope,
                          int rope_rows, uint16_t* out, int64_t ldo, int M, int C, int head_dim, float eps,
                          int rows_per_sample, int tok_offset, int n_rope, int gf, int gh, int gw,
                          void* stream);

/* ovd_rmsnorm_rope_bf16 for the q and k thirds of fused projection rows qkv [M, 3C] (q | k | v, row stride
 * ld) in ONE launch, each third with its own RMSNorm weight, both rotated; with copy_v the v third is
 * copied along unchanged.  The result of third p goes to out + p * out_part_stride, its C columns split
 * into blocks of col_block columns that sit out_block_stride elements apart; row m of a block starts at
 * (m / rows_per_sample) * out_sample_stride + (m % rows_per_sample) * ldo.
 *   in place:  out = qkv, ldo = ld, out_part_stride = C, col_block = C, out_sample_stride =
 *              rows_per_sample * ld, copy_v = 0  (what WanSelfAttention.forward does at model.py:137-149);
 *   sequence parallelism: out = the all-to-all send image [3][samples][p ranks][rows][C/p] (col_block =
 *              ldo = C/p, out_block_stride = rows * C/p, out_sample_stride = p * rows * C/p, out_part_stride
 *              = samples * p * rows * C/p, copy_v = 1): the head-scatter of Ulysses attention
 *              (omnivideo/distributed/ulysses.py:41-43, util.py:106-109 chunk + contiguous copies) written by
 *              the kernel that produces q and k, so no separate packing pass exists. */
int ovd_qkv_norm_rope_bf16(const uint16_t* qkv, int64_t ld, const uint16_t* weight_q, const uint16_t* weight_k,
                           const float* rope, int rope_rows, uint16_t* out, int64_t ldo, int64_t out_part_stride,
                           int col_block, int64_t out_block_stride, int64_t out_sample_stride, int copy_v, int M,
                           int C, int head_dim, float eps, int rows_per_sample, int tok_offset, int n_rope, int gf,
                           int gh, int gw, void* stream);

/* dst[m][b * col_block + c] = src[b][m][c], b < nblk, m < M (dst row stride ldd): the head-gather after
 * the second all-to-all of Ulysses attention (ulysses.py:55; util.py:106-109 cat along the head dim) -
 * the attention output of every head group, received one group after another, back into token rows. */
int ovd_gather_col_blocks_bf16(const uint16_t* src, uint16_t* dst, int64_t ldd, int64_t M, int nblk, int col_block,
                               void* stream);

/* The two kernels above as sender and receiver of the sequence-parallel exchange itself, for ranks that
 * have mapped each other's buffers (ovd_peer_* below): no all-to-all call and no copy pass remains.
 * ovd_qkv_norm_rope_scatter_bf16: as ovd_qkv_norm_rope_bf16 with copy_v = 1, but column block b of every
 * third goes to block_ptrs[b] (DEVICE array of C / col_block addresses, element offsets inside each as
 * above with out_block_stride = 0): block_ptrs[j] = rank j's receive buffer [3][samples][p * rows][C/p] as
 * mapped here, advanced to this rank's rows, makes the kernel that normalises and rotates q and k write
 * them straight into every peer's attention operands over xGMI.
 * ovd_gather_col_blocks_from_bf16: dst[m][b * col_block + c] = src_ptrs[b][m * col_block + c] (DEVICE array
 * of nblk addresses); src_ptrs[j] = rank j's attention output for this rank's rows makes the head-gather
 * read straight from the peers.  Blocks are walked starting at block `first` (pass own rank + 1 modulo
 * nblk: the ranks then read from different peers at any time).
 * Replace dist.all_to_all at omnivideo/distributed/util.py:95-121 as used by ulysses.py:41-43 and :55. */
int ovd_qkv_norm_rope_scatter_bf16(const uint16_t* qkv, int64_t ld, const uint16_t* weight_q, const uint16_t* weight_k,
                                   const float* rope, int rope_rows, uint16_t* const* block_ptrs, int64_t ldo,
                                   int64_t out_part_stride, int col_block, int64_t out_sample_stride, int M, int C,
                                   int head_dim, float eps, int rows_per_sample, int tok_offset, int n_rope, int gf,
                                   int gh, int gw, void* stream);
int ovd_gather_col_blocks_from_bf16(const uint16_t* const* src_ptrs, uint16_t* dst, int64_t ldd, int64_t M, int nblk,
                                    int col_block, int first, void* stream);

/* out = epilogue(A[M,K] * W[N,K]^T + bias[N]), bf16 operands, fp32 accumulation.
 * epilogue 0: bf16 out; 1: bf16 out of tanh-GELU(bf16-rounded pre-activation);
 * 2: fp32 out holding the bf16-rounded value; 3: fp32 in-place residual
 * out += bf16-rounded(acc + bias) * gate[n] (gate fp32 [N] or NULL for 1).
 * K % 64 == 0, N % 128 == 0.  bias may be NULL.
 * Replaces nn.Linear under bf16 autocast at model.py:137-139,153,169-171,178,
 * 211-213,386-388, unified_model.py:202, visual_context_adapter.py:112, the Conv3d patch
 * embeddings (model.py:455, visual_context_adapter.py:101) after ovd_patchify, and the
 * gated residual adds of model.py:245-254. */
int ovd_gemm_bf16(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, const uint16_t* bias,
                  void* out, int64_t ldo, const float* gate, int M, int N, int K, int epilogue,
                  void* stream);

/* softmax(q k^T * softmax_scale) v, non-causal, per (batch, head); head_dim must be 128.
 * q [B,Lq,N,D], k/v [B,Lk,N,D], out [B,Lq,N,D], all bf16; *_ld = elements between
 * consecutive tokens, *_bs = elements between batch entries.  Keys at index >= k_len
 * are masked out (0 < k_len <= Lk).
 * Replaces flash_attn(_interface).flash_attn_varlen_func as called from
 * omnivideo/modules/attention.py:96-130 (via model.py:144-149 and model.py:174). */
int ovd_flash_attn_fwd_bf16(const uint16_t* q, const uint16_t* k, const uint16_t* v, uint16_t* out, int B,
                            int Lq, int Lk, int N, int head_dim, int k_len, int64_t q_ld, int64_t k_ld,
                            int64_t v_ld, int64_t o_ld, int64_t q_bs, int64_t k_bs, int64_t v_bs,
                            int64_t o_bs, float softmax_scale, void* stream);

/* As ovd_flash_attn_fwd_bf16, with key k_len - 1 counted last_key_copies (>= 1) times: its
 * probability enters the row sum and P V with that weight.  For cross-attention over a context
 * zero-padded to text_len (model.py:480-485 pads, model.py:174 attends every padded key): all
 * padded rows of K and of V are identical, so rows [n, text_len) can be given as the single
 * key n with last_key_copies = text_len - n and k_len = n + 1; the result equals attending
 * every padded key up to fp32 summation order. */
int ovd_flash_attn_tail_fwd_bf16(const uint16_t* q, const uint16_t* k, const uint16_t* v, uint16_t* out, int B,
                                 int Lq, int Lk, int N, int head_dim, int k_len, int last_key_copies,
                                 int64_t q_ld, int64_t k_ld, int64_t v_ld, int64_t o_ld, int64_t q_bs,
                                 int64_t k_bs, int64_t v_bs, int64_t o_bs, float softmax_scale, void* stream);

/* The same operator with a caller-owned workspace that lets the library balance the last round of
 * workgroups: when the grid of 256-row query blocks x heads x batch leaves at most half of the
 * compute units busy in its final round, the trailing query blocks are computed by two workgroups
 * each (one per half of the keys) and merged by a second small kernel; results then differ from
 * the plain call only by fp32 summation order.  ovd_flash_attn_workspace_bytes (host function, no
 * GPU work) returns the size that makes this possible for a shape (0 = it would not split); a NULL
 * or smaller workspace gives exactly the plain call.  The workspace holds no state between calls.
 * No counterpart in the reference (flash-attn allocates its own split buffers internally). */
int64_t ovd_flash_attn_workspace_bytes(int B, int Lq, int N, int k_len);
int ovd_flash_attn_ws_fwd_bf16(const uint16_t* q, const uint16_t* k, const uint16_t* v, uint16_t* out, int B, int Lq,
                               int Lk, int N, int head_dim, int k_len, int last_key_copies, int64_t q_ld,
                               int64_t k_ld, int64_t v_ld, int64_t o_ld, int64_t q_bs, int64_t k_bs, int64_t v_bs,
                               int64_t o_bs, float softmax_scale, void* workspace, int64_t workspace_bytes,
                               void* stream);

/* ovd_flash_attn_ws_fwd_bf16 without the workspace argument, for callers that cannot allocate device
 * memory (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc): the
 * tail-balancing workspace is a 33 MiB (34 603 008 byte) device array inside the library, which every
 * GPU the library runs on holds until the process ends.  No counterpart in the reference.  The call
 * allocates and frees nothing; results are bit-identical to ovd_flash_attn_ws_fwd_bf16 given a
 * workspace of ovd_flash_attn_workspace_bytes(...) bytes.  A shape whose split needs more than the
 * array holds is rejected (return 1, reason in ovd_last_error()); the array covers every shape on a
 * 256-CU device.  The first call on each device looks the array's device address up (one
 * hipGetSymbolAddress): make it outside graph capture.  All calls on one GPU share that GPU's array:
 * calls that overlap on one GPU (two streams, two host threads) race on it; calls on one stream do not. */
int ovd_flash_attn_static_fwd_bf16(const uint16_t* q, const uint16_t* k, const uint16_t* v, uint16_t* out, int B,
                                   int Lq, int Lk, int N, int head_dim, int k_len, int last_key_copies,
                                   int64_t q_ld, int64_t k_ld, int64_t v_ld, int64_t o_ld, int64_t q_bs,
                                   int64_t k_bs, int64_t v_bs, int64_t o_bs, float softmax_scale, void* stream);

/* [Cin,T,H,W] (fp32, or bf16 when lat_is_bf16) -> bf16 rows [f*h*w, Cin*pt*ph*pw] in
 * (f,h,w) token order and (cin,pt,ph,pw) column order, ready for ovd_gemm_bf16 against
 * the flattened Conv3d weight.  Replaces the im2col half of nn.Conv3d with
 * kernel = stride = patch (model.py:384-385,455-458; visual_context_adapter.py:52-54). */
int ovd_patchify(const void* lat, int lat_is_bf16, uint16_t* rows, int Cin, int T, int H, int W, int pt,
                 int ph, int pw, void* stream);

/* out[0:dim/2] = cos(t * 10000^(-i/(dim/2))), out[dim/2:] = sin(...), evaluated in
 * float64, stored fp32; t = timesteps[step_ptr ? *step_ptr : step] (float64 table: the
 * reference converts its integer timesteps with position.type(torch.float64)).
 * Replaces sinusoidal_embedding_1d (model.py:13-23) at model.py:473-474. */
int ovd_time_sinusoid(const double* timesteps, const int* step_ptr, int step, float* out, int dim,
                      void* stream);

/* y[N] = act(x[K]) . W[N,K] + b[N] in fp32 on bf16 weights; silu_in applies SiLU to x
 * first.  Replaces time_embedding / time_projection (model.py:390-392,472-475), which
 * the reference evaluates under a float32 autocast for every token although the
 * timestep is one scalar per sample. */
int ovd_gemv_f32(const float* x, const uint16_t* W, const uint16_t* b, float* y, int N, int K,
                 int silu_in, void* stream);

/* out[l][i] = float(params[l][i]) + e0[i], i < n6c = 6*C, for every layer l: the
 * per-block modulation add (model.py:237-238).  params = device array of n_layers
 * device pointers. */
int ovd_modulation_table(const uint16_t* const* params, const float* e0, float* out, int n_layers,
                         int n6c, void* stream);
/* out[j][c] = float(mod[j][c]) + e[c], j in {0,1}: Head.forward's modulation
 * (model.py:286). */
int ovd_head_modulation(const uint16_t* mod, const float* e, float* out, int C, void* stream);

/* Head projection in exact fp32 fused with the unpatchify scatter:
 * out[c][f*pt+p][h*ph+q][w*pw+r] = y[tok(f,h,w)] . W[(p,q,r,c)] + bias, with
 * out_dim*pt*ph*pw == 64.  y is the normalised + modulated fp32 activation; its first n_tok
 * rows hold tokens tok_offset .. tok_offset+n_tok-1 (a sequence-parallel rank passes its
 * shard; rows past the grid's f*h*w tokens are skipped).  Only those tokens' outputs are
 * written.
 * Replaces Head.head under float32 autocast (model.py:285-289) and WanModel.unpatchify
 * (model.py:526-549). */
int ovd_head_unpatchify(const float* y, int64_t ldy, const uint16_t* W, const uint16_t* bias, float* out,
                        int n_tok, int C, int Cout, int pt, int ph, int pw, int gf, int gh, int gw,
                        int tok_offset, void* stream);

/* One sampling step on n latent elements:
 *   v = uncond + guide * (cond - uncond)                 (x2x_gen_unified.py:484)
 *   x0 = sample - sigma * v                              (fm_solvers_unipc.py:316-322)
 *   corrector (from step 1 on), history shift, predictor (fm_solvers_unipc.py:690-729)
 * sample / last_sample are in the latent dtype (bf16 when sample_is_bf16, else fp32) and
 * are updated in place; m1/m2 are the fp32 x0 history (newest / older).  coef is the
 * host-built table [steps][16] (row layout: csrc/sampler.hip); the row index is
 * *step_ptr when step_ptr is non-NULL, else `step`. */
int ovd_cfg_unipc_step(const float* cond, const float* uncond, void* sample, void* last_sample, float* m1,
                       float* m2, const float* coef, float guide_scale, const int* step_ptr, int step,
                       int64_t n, int sample_is_bf16, void* stream);
#define OVD_UNIPC_COEF_STRIDE 16

/* One DPM-Solver++ step (order <= 2, midpoint, flow prediction) on n latent elements:
 *   v = uncond + guide * (cond - uncond)                       (x2x_gen_unified.py:484)
 *   x0 = sample - sigma * v                                    (fm_solvers.py:383-385)
 *   next = ratio*sample - b*x0 [ - half_b * inv_r0*(x0 - m1) ] (fm_solvers.py:455-466, :527-560)
 *   m1 = x0                                                    (fm_solvers.py:748-751)
 * sample is bf16 (sample_is_bf16, only ever the initial noise) or fp32; next is fp32 (the
 * reference's prev_sample takes the dtype of the converted model output, fm_solvers.py:793)
 * and may alias an fp32 sample.  coef is the host-built table [steps][8]: sigma, order, ratio,
 * b, half_b, inv_r0; the row index is *step_ptr when step_ptr is non-NULL, else `step`. */
int ovd_cfg_dpm_step(const float* cond, const float* uncond, const void* sample, float* next, float* m1,
                     const float* coef, float guide_scale, const int* step_ptr, int step, int64_t n,
                     int sample_is_bf16, void* stream);
#define OVD_DPM_COEF_STRIDE 8

/* fp32 video -> uint8 frames [nvox, C] (T,H,W,C order): clamp to [lo, hi], (x - lo) / (hi - lo),
 * times 255, truncate - the tensor half of cache_video (omnivideo/utils/utils.py:175-182, with
 * torchvision.utils.make_grid's normalize for a single video).  Element x[v][c] is read at
 * v*voxel_stride + c*channel_stride: (Cs, 1) for channels-last, (1, nvox) for planar [C,T,H,W]. */
int ovd_video_to_uint8(const float* in, uint8_t* out, int64_t voxel_stride, int64_t channel_stride, int C,
                       int64_t nvox, float lo, float hi, void* stream);

/* ---- FP8 (OCP e4m3) mode of the linear layers.  The reference has no FP8 path: it runs these
 * layers in bf16 (omnivideo/modules/model.py:125-154 self-attention projections, :159-179 cross-
 * attention projections, :218-258 feed-forward); this mode is the reduced-precision configuration
 * for the largest model and is checked against this library's own bf16 entries. ---- */

/* bf16 rows [M, K] -> e4m3 rows q [M, K] and scale[m] = absmax(row m) / 448 (1 for an all-zero row),
 * so that x[m][k] ~ q[m][k] * scale[m].  K % 8 == 0, K <= 14336.  Weights [N, K] of nn.Linear get
 * one scale per output channel this way (once per model); activations one scale per token row. */
int ovd_quant_rows_fp8(const void* x, int64_t ldx, uint8_t* q, int64_t ldq, float* scale, int M, int K, void* stream);

/* ovd_layernorm_modulate_bf16 followed by ovd_quant_rows_fp8 of its bf16 result, in one pass. */
int ovd_layernorm_modulate_fp8(const float* x, int64_t ldx, const float* shift, const float* scale, const void* weight,
                               const void* bias, uint8_t* out, int64_t ldo, float* row_scale, int M, int C, float eps,
                               int round_norm, void* stream);

/* out = epilogue( (sum_k A[m][k] * W[n][k]) * a_scale[m] * a_scale_all * w_scale[n] + bias[n] ) on e4m3
 * operands with fp32 accumulation (v_mfma_scale_f32_32x32x64_f8f6f4, unit block scales).  a_scale may
 * be NULL (1 per row).  Epilogues 0-3 as ovd_gemm_bf16; 4 (OVD_EPI_FP8_GELU): out is e4m3 [M, N] holding
 * bf16(gelu(bf16(.))) / out_scale, saturated at +-448, for a following ovd_gemm_fp8 with
 * a_scale_all = out_scale.  N % 256 == 0, K % 64 == 0, K >= 128, lda/ldw % 16 == 0, each operand < 4 GiB. */
#define OVD_EPI_FP8_GELU 4
int ovd_gemm_fp8(const uint8_t* A, int64_t lda, const float* a_scale, float a_scale_all, const uint8_t* W, int64_t ldw,
                 const float* w_scale, const void* bias, void* out, int64_t ldo, const float* gate, float out_scale,
                 int M, int N, int K, int epilogue, void* stream);

/* Reduced-precision self-attention of the same mode: Q K^T on int8 operands (v_mfma_i32_32x32x32_i8, half
 * the matrix time of the bf16 product), softmax in fp32, P and V in bf16 as in ovd_flash_attn_fwd_bf16.
 * No reference counterpart (the reference's attention is flash-attn in bf16, attention.py:96-130); checked
 * against this library's bf16 attention and the exact-softmax oracle at a stated tolerance.
 *
 * ovd_quant_heads_i8: x bf16 [B, L, N, 128] (token stride ldx, batch stride x_bs, elements) -> q int8 of the
 * same shape (ldq, q_bs) and one fp32 scale per rows_per_scale rows and head, x ~ q * scale with
 * scale = absmax / 127 of the block (1 for an all-zero block), q = rint(x * (127 / absmax)):
 *   rows_per_scale = 1  (queries): scale [B, L, N];
 *   rows_per_scale = 64 (keys, one scale per 64-key tile): scale [B, ceil(L / 64), N].
 * ovd_flash_attn_qk8_fwd_bf16: q8 / q_scale and k8 / k_scale in those two forms, v and out bf16; strides in
 * elements; keys >= k_len masked; workspace as in ovd_flash_attn_ws_fwd_bf16 (NULL = none). */
int ovd_quant_heads_i8(const uint16_t* x, int64_t ldx, int64_t x_bs, int8_t* q, int64_t ldq, int64_t q_bs, float* scale,
                       int B, int L, int N, int head_dim, int rows_per_scale, void* stream);
int ovd_flash_attn_qk8_fwd_bf16(const int8_t* q8, const float* q_scale, const int8_t* k8, const float* k_scale,
                                const uint16_t* v, uint16_t* out, int B, int Lq, int Lk, int N, int head_dim, int k_len,
                                int64_t q_ld, int64_t k_ld, int64_t v_ld, int64_t o_ld, int64_t q_bs, int64_t k_bs,
                                int64_t v_bs, int64_t o_bs, float softmax_scale, void* workspace,
                                int64_t workspace_bytes, void* stream);

/* Source-video frames -> VAE encoder input, the device half of transform_frames_to_tensor
 * (tools/inference/generate_omni_v2v.py:306-343: CenterCrop, Resize on a PIL image = Pillow's 8-bit
 * bilinear resample, ToTensor, Normalize).  frames [T,H,W,3] uint8; the crop window (top, left,
 * crop_h, crop_w) is resized to out_h x out_w bit-exactly as Pillow does (horizontal pass into
 * tmp [T,crop_h,out_w,3] uint8, then vertical), and each byte b is written as lut[b] (256 floats;
 * the reference's map is (b/255 - 0.5)/0.5) to out [3,T,out_h,out_w] fp32 - the planar layout
 * Wan2_1_VAE.encode takes.  The tap tables are DEVICE copies of what ovd_resize_coeffs fills. */
int ovd_frames_resize_normalize(const uint8_t* frames, uint8_t* tmp, float* out, const int* bounds_w,
                                const int* coef_w, int ksize_w, const int* bounds_h, const int* coef_h, int ksize_h,
                                const float* lut, int64_t T, int H, int W, int top, int left, int crop_h, int crop_w,
                                int out_h, int out_w, void* stream);

/* HOST function (no GPU work): tap tables of one axis of that resize, in Pillow's arithmetic
 * (third party: Pillow src/libImaging/Resample.c, precompute_coeffs + normalize_coeffs_8bpc,
 * bilinear filter).  bounds_host [out_size][2] = (first tap, taps), coef_host [out_size][ksize_cap].
 * Returns the tap-table width ksize (call with ksize_cap = 0 to learn it; nothing is written when
 * ksize_cap < ksize), or -1 for sizes it does not serve (non-positive, or ksize > 64). */
int ovd_resize_coeffs(int in_size, int out_size, int* bounds_host, int* coef_host, int ksize_cap);

/* umT5 encoder self-attention (omnivideo/modules/t5.py:85-119): no score scaling, additive
 * relative-position bias, fp32 softmax, bf16 tensors between the steps:
 *   s = bf16(bf16(q.k) + pos_embedding[bucket[i][j]][head]);  keys j >= k_lens[b] get the lowest score
 *   p = bf16(softmax_fp32(s));  out = bf16(p.v)
 * qkv [B, L, 3*N*64] bf16 (q | k | v; ld = elements between tokens), pos_embedding [buckets, N]
 * bf16 (the block's T5RelativeEmbedding weight, t5.py:230), bucket [L, L] int32 (t5.py:244-263,
 * built once per length on the host), k_lens [B] int32 or NULL (prefix mask lengths), out
 * [B, L, N*64] bf16.  head_dim must be 64. */
int ovd_t5_attention_bf16(const uint16_t* qkv, int64_t ld, const uint16_t* pos_embedding, const int32_t* bucket,
                          const int32_t* k_lens, uint16_t* out, int64_t ldo, int B, int L, int N, int head_dim,
                          void* stream);

/* out[m][f] = bf16(fc1[m][f] * GELU(gate[m][f])) with h = [gate | fc1] ([M, 2F] bf16, the fused
 * output of both projections); GELU is the reference's tensor-op formula evaluated in bf16
 * arithmetic (t5.py:47-49, :135-136). */
int ovd_t5_gated_gelu_bf16(const uint16_t* h, int64_t ldh, uint16_t* out, int64_t ldo, int M, int F, void* stream);

/* x = bf16(x + y) over n (even) elements: the encoder's bf16 residual adds (t5.py:172-173). */
int ovd_add_bf16(uint16_t* x, const uint16_t* y, int64_t n, void* stream);

/* *step_ptr += 1 on the device (advances the step counter inside a captured graph). */
int ovd_step_advance(int* step_ptr, void* stream);

/* ---- causal 3D VAE, channels-last fp32 activations [T,H,W,C] ------------------------ */

/* Time-causal 3D convolution as an implicit GEMM in exact fp32:
 *   out[t,h,w,co] = bias[co] + sum act(in[t*stride_t+kt-pad_t, h*s+kh-pad_hw, w*s+kw-pad_hw, ci]) * w[kt,kh,kw][co][ci]
 * x [T,H,W,Cin]; frames at time index -1 / -2 (the previous chunk's tail) come from prev1 /
 * prev2 ([H,W,Cin] each, NULL = zeros); spatial out-of-range reads are zero.
 * w is [KT*KH*KW][Cout][Cin] fp32.  Kernel sizes: KT in {1,3}, KH = KW in {1,3}; strides 1|2.
 * Optional input transform act(): per-voxel RMS norm x * rnorm[voxel] * sqrt(Cin) * gamma[ci]
 * (rnorm* from ovd_vae_voxel_rnorm; gamma NULL = off) followed by SiLU when silu != 0.
 * upsample2: the conv reads a nearest-neighbour 2x spatially upsampled view of x.
 * time_interleave: output channel co goes to frame 2t + co/(Cout/2), channel co%(Cout/2).
 * resid (same layout as out) is added when non-NULL.  Cin % 4 == 0.
 * Output extent: To = (T+pad_t-KT)/stride_t+1, Ho/Wo by the usual formula (a stride-2 conv
 * reads one zero row/column past the bottom/right edge).
 * Replaces CausalConv3d.forward (omnivideo/modules/vae2_1.py:27-35) with its feature-cache
 * concat, RMS_norm + SiLU (:50-53,193-197), Upsample + Conv2d and ZeroPad2d + strided
 * Conv2d in Resample (:75-95,137-140), the time_conv frame interleave (:133-136), the
 * strided time_conv (:155-156), the residual add of ResidualBlock (:219) and the 1x1
 * Conv2d layers and matmuls of AttentionBlock (:245-261). */
int ovd_vae_conv3d_f32(const float* x, const float* prev2, const float* prev1, const float* rnorm,
                       const float* rnorm_prev2, const float* rnorm_prev1, const float* gamma, int silu,
                       const float* w, const float* bias, const float* resid, float* out, int T, int H,
                       int W, int Cin, int Cout, int KT, int KH, int KW, int stride_t, int stride_hw,
                       int pad_t, int pad_hw, int upsample2, int time_interleave, void* stream);

/* The same convolution with its weights given in the split form of ovd_vae_split_weights, which is the
 * image the kernel keeps in LDS: staging a weight slice is then a copy instead of a conversion
 * repeated by every workgroup.  Results are bit-identical to ovd_vae_conv3d_f32 on its default
 * (split-bf16) path; the exact-fp32 path (OVD_VAE_EXACT_FP32=1) needs the fp32 weights and rejects
 * this entry.  ovd_vae_split_weights: w fp32 [rows][Cin] (rows = KT*KH*KW*Cout) ->
 * out uint16 [rows][ceil(Cin/32)][32 hi | 32 lo] with hi = bf16(w), lo = bf16(w - hi), zero past Cin. */
int ovd_vae_split_weights(const float* w, uint16_t* out, int64_t rows, int Cin, void* stream);
int ovd_vae_conv3d_sw_f32(const float* x, const float* prev2, const float* prev1, const float* rnorm,
                          const float* rnorm_prev2, const float* rnorm_prev1, const float* gamma, int silu,
                          const uint16_t* w_split, const float* bias, const float* resid, float* out, int T, int H,
                          int W, int Cin, int Cout, int KT, int KH, int KW, int stride_t, int stride_hw, int pad_t,
                          int pad_hw, int upsample2, int time_interleave, void* stream);

/* The input side of a convolution done once per layer instead of once per workgroup that stages a voxel:
 * ovd_vae_act_split: x fp32 [nvox][C] -> out uint16 [nvox][C/32][32 hi | 32 lo], the split-bf16 image (hi =
 * bf16(y), lo = bf16(y - hi)) of y = SiLU(RMS_norm(x)) - F.normalize over the channels * sqrt(C) * gamma
 * (gamma NULL: no norm), SiLU when silu != 0 - i.e. RMS_norm.forward + nn.SiLU in front of the convolutions of
 * ResidualBlock and of the encoder / decoder heads (vae2_1.py:50-53,193-197,311-314,416-419).  C % 32 == 0.
 * ovd_vae_conv3d_ps_f32: the 3x3 (KT = 1 or 3) stride-1 convolution of ovd_vae_conv3d_sw_f32 on such an image
 * (x_split [T][H][W][Cin/32][64]; prev*_split the [H][W][Cin/32][64] frames at time -2 / -1, NULL = zeros):
 * same products, same accumulation, bit-identical results; the kernel's staging is then pure copies.  upsample2 != 0:
 * the convolution runs on the nearest-neighbour 2x spatial upsampling of the image (output [T][2H][2W][Cout]; the
 * Upsample + Conv2d pair of Resample, vae2_1.py:72-84), as in ovd_vae_conv3d_sw_f32. */
int ovd_vae_act_split(const float* x, const float* gamma, int silu, uint16_t* out, int64_t nvox, int C, void* stream);
int ovd_vae_conv3d_ps_f32(const uint16_t* x_split, const uint16_t* prev2_split, const uint16_t* prev1_split,
                          const uint16_t* w_split, const float* bias, const float* resid, float* out, int T, int H, int W,
                          int Cin, int Cout, int KT, int KH, int KW, int pad_t, int pad_hw, int upsample2, void* stream);

/* out[v] = 1 / max(||x[v,:]||_2, 1e-12) for nvox voxels of C channels: the denominator of
 * F.normalize in RMS_norm.forward (vae2_1.py:50-53). */
int ovd_vae_voxel_rnorm(const float* x, float* out, int64_t nvox, int C, void* stream);

/* [C,T,H,W] -> [T,H,W,Cp] (Cp >= C, extra channels zero) with out = (in + pre[c]) * mul[c]
 * + post[c] (each vector may be NULL); and the inverse for the first C of Cs channels with an
 * optional clamp to [-1,1].  Layout change at the VAE boundary plus the latent
 * (de)normalisation of WanVAE_.encode/.decode (vae2_1.py:535-539,546-550) and the output
 * clamp of Wan2_1_VAE.decode (:660). */
int ovd_vae_to_channels_last(const float* in, float* out, const float* pre, const float* mul,
                             const float* post, int C, int Cp, int64_t nvox, void* stream);
int ovd_vae_from_channels_last(const float* in, float* out, const float* pre, const float* mul,
                               const float* post, int C, int Cs, int64_t nvox, int clamp, void* stream);

/* In-place softmax(scale * s[m,:]) over M rows of length N, fp32: the softmax of
 * F.scaled_dot_product_attention in AttentionBlock.forward (vae2_1.py:251-255). */
int ovd_softmax_rows_f32(float* s, int M, int N, float scale, void* stream);

/* ---- Sequence-parallel exchange through peer-mapped device memory --------------------------------
 * The collectives of Ulysses attention without a collective library: every rank (one process per GPU)
 * maps the exchange buffers of its peers once, and the exchange is plain kernels on the caller's
 * stream - p concurrent reads, one per xGMI link of the full mesh - which can be captured into the HIP
 * graph of the sampling step.  Replaces dist.all_to_all inside AllToAll / all_to_all
 * (omnivideo/distributed/util.py:95-121, used by ulysses.py:41-43,55) together with the chunk /
 * contiguous / cat copies around it (util.py:106-109), and the all-gather of the head output
 * (util.py:124-130 via full_sequence_parallel.py:163).
 *
 * SET-UP functions (HOST; they allocate, free or map memory and are the exception to "no entry point
 * allocates": explicit create / destroy, never called from a graph):
 *   ovd_peer_alloc   device buffer that can be exported (zero-filled); fine_grained != 0 for the flag
 *                    arrays (a store from another device must become visible to a kernel that is
 *                    already running), 0 for payload;
 *   ovd_peer_export  64-byte handle of a buffer from ovd_peer_alloc, to be sent to the peers by any
 *                    host channel;  ovd_peer_open maps a peer's handle (in another process; a buffer of
 *                    the same process is used through its own pointer);  ovd_peer_close / ovd_peer_free.
 */
int ovd_peer_alloc(void** ptr, int64_t bytes, int fine_grained);
int ovd_peer_free(void* ptr);
int ovd_peer_export(void* ptr, unsigned char handle[64]);
int ovd_peer_open(const unsigned char handle[64], void** ptr);
int ovd_peer_close(void* ptr);

/* Barrier over the p ranks, on the stream.  my_flags: this rank's fine-grained array of 128 uint32 (zero at
 * start: [j] = epoch reached by rank j, [64 + j] = rank j has given up on a barrier); peer_flags: DEVICE array
 * of p pointers, entry j = rank j's flag array as mapped here (entry `rank` = my_flags); state: 2 uint32 of
 * this rank, zero at start ([0] epoch counter; [1] stays 0 unless the group has failed: 1 + j = rank j did not
 * arrive within timeout_s seconds at a barrier of this rank, 65 + j = rank j reported such a failure of its
 * own).  A failure is told to every peer, so one barrier later state[1] is set on every rank; the kernel
 * returns in any case and the caller must treat everything after as invalid.  When the kernel ends without
 * failure, every kernel that any rank launched on its stream before its own matching barrier has finished.
 * p <= 64. */
int ovd_peer_barrier(uint32_t* my_flags, uint32_t* const* peer_flags, uint32_t* state, int p, int rank,
                     double timeout_s, void* stream);

/* recv[j] = chunk `rank` of rank j's send buffer, j < p: recv [p][chunk_bytes] local, peer_send a DEVICE
 * array of p pointers to the ranks' send buffers [p][chunk_bytes] as mapped here.  chunk_bytes % 16 == 0.
 * Call between two barriers: the one before makes the peers' send buffers complete, a later one must
 * pass before a send buffer is rewritten. */
int ovd_peer_all_to_all(void* recv, const void* const* peer_send, int64_t chunk_bytes, int p, int rank,
                        void* stream);

/* out[i] = sum over j < p of peer_src[j][i], i < n, fp32, added in rank order (identical bits on every
 * rank); peer_src a DEVICE array of p pointers.  `out` must not be one of the published buffers. */
int ovd_peer_sum_f32(float* out, const float* const* peer_src, int64_t n, int p, void* stream);

#ifdef __cplusplus
}
#endif
#endif
