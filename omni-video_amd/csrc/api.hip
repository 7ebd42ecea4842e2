// extern "C" surface of libomnihip: thin argument adapters over the kernel launchers.
#include <stdarg.h>

#include "../../include/omnihip.h"
#include "common.h"

static thread_local char g_err[512] = "";

void ovd_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int ovd_impl_ln_bf16(float*, long, const bf16_t*, long, const float*, const float*, const float*, const bf16_t*,
                     const bf16_t*, bf16_t*, long, int, int, float, int, hipStream_t);
int ovd_impl_ln_f32(float*, long, const bf16_t*, long, const float*, const float*, const float*, const bf16_t*,
                    const bf16_t*, float*, long, int, int, float, int, hipStream_t);
int ovd_impl_rmsnorm_rope(const bf16_t*, long, const bf16_t*, const float*, int, bf16_t*, long, int, int, int, float,
                          int, int, int, int, int, int, hipStream_t);
int ovd_impl_qkv_norm_rope(const bf16_t*, long, const bf16_t*, const bf16_t*, const float*, int, bf16_t*, bf16_t* const*,
                           long, long, int, long, long, int, int, int, int, float, int, int, int, int, int, int,
                           hipStream_t);
int ovd_impl_gather_col_blocks(const bf16_t*, const bf16_t* const*, bf16_t*, long, long, int, int, int, hipStream_t);
int ovd_impl_gemm_bf16(const bf16_t*, long, const bf16_t*, long, const bf16_t*, void*, long, const float*, int, int,
                       int, int, hipStream_t);
int ovd_impl_attn_fwd(const bf16_t*, const bf16_t*, const bf16_t*, bf16_t*, int, int, int, int, int, int, long, long,
                      long, long, long, long, long, long, float, int, void*, long, hipStream_t);
long ovd_impl_attn_workspace_bytes(int, int, int, int);
int ovd_impl_quant_heads_i8(const bf16_t*, long, long, int8_t*, long, long, float*, int, int, int, int, int, hipStream_t);
int ovd_impl_attn_i8_fwd(const int8_t*, const float*, const int8_t*, const float*, const bf16_t*, bf16_t*, int, int, int,
                         int, int, int, long, long, long, long, long, long, long, long, float, void*, long, hipStream_t);
int ovd_impl_attn_fwd_static(const bf16_t*, const bf16_t*, const bf16_t*, bf16_t*, int, int, int, int, int, int, long,
                             long, long, long, long, long, long, long, float, int, hipStream_t);
int ovd_impl_to_uint8(const float*, uint8_t*, long, long, int, long, float, float, hipStream_t);
int ovd_impl_resize_coeffs(int, int, int*, int*, int);
int ovd_impl_quant_rows_fp8(const bf16_t*, long, uint8_t*, long, float*, int, int, hipStream_t);
int ovd_impl_ln_fp8(float*, long, const bf16_t*, long, const float*, const float*, const float*, const bf16_t*,
                    const bf16_t*, uint8_t*, long, float*, int, int, float, int, hipStream_t);
int ovd_impl_gemm_fp8(const uint8_t*, long, const float*, float, const uint8_t*, long, const float*, const bf16_t*, void*,
                      long, const float*, float, int, int, int, int, hipStream_t);
int ovd_impl_frames_resize(const uint8_t*, uint8_t*, float*, const int*, const int*, int, const int*, const int*, int,
                           const float*, long, int, int, int, int, int, int, int, int, hipStream_t);
int ovd_impl_t5_attention(const bf16_t*, long, const bf16_t*, const int*, const int*, bf16_t*, long, int, int, int, int,
                          hipStream_t);
int ovd_impl_t5_gated_gelu(const bf16_t*, long, bf16_t*, long, int, int, hipStream_t);
int ovd_impl_add_bf16(bf16_t*, const bf16_t*, long, hipStream_t);
int ovd_impl_patchify(const void*, int, bf16_t*, int, int, int, int, int, int, int, hipStream_t);
int ovd_impl_time_sinusoid(const double*, const int*, int, float*, int, hipStream_t);
int ovd_impl_gemv_f32(const float*, const bf16_t*, const bf16_t*, float*, int, int, int, hipStream_t);
int ovd_impl_modulation(const bf16_t* const*, const float*, float*, int, int, hipStream_t);
int ovd_impl_head_modulation(const bf16_t*, const float*, float*, int, hipStream_t);
int ovd_impl_head_unpatchify(const float*, long, const bf16_t*, const bf16_t*, float*, int, int, int, int, int, int,
                             int, int, int, int, hipStream_t);
int ovd_impl_cfg_unipc_step(const float*, const float*, void*, void*, float*, float*, const float*, float,
                            const int*, int, long, int, hipStream_t);
int ovd_impl_cfg_dpm_step(const float*, const float*, const void*, float*, float*, const float*, float, const int*,
                          int, long, int, hipStream_t);
int ovd_impl_step_advance(int*, hipStream_t);
int ovd_impl_split_weights(const float*, uint16_t*, long, int, hipStream_t);
int ovd_impl_act_split(const float*, const float*, int, uint16_t*, long, int, hipStream_t);
int ovd_impl_conv3d_cl(const float*, const float*, const float*, const uint16_t*, const uint16_t*, const uint16_t*,
                       const float*, const float*, const float*, const float*,
                       int, const float*, const uint16_t*, const float*, const float*, float*, int, int, int, int, int, int, int, int,
                       int, int, int, int, int, int, hipStream_t);
int ovd_impl_voxel_rnorm(const float*, float*, long, int, hipStream_t);
int ovd_impl_to_cl(const float*, float*, const float*, const float*, const float*, int, int, long, hipStream_t);
int ovd_impl_from_cl(const float*, float*, const float*, const float*, const float*, int, int, long, int, hipStream_t);
int ovd_impl_softmax_rows(float*, int, int, float, hipStream_t);
int ovd_impl_peer_alloc(void**, size_t, int);
int ovd_impl_peer_free(void*);
int ovd_impl_peer_export(void*, unsigned char*);
int ovd_impl_peer_open(const unsigned char*, void**);
int ovd_impl_peer_close(void*);
int ovd_impl_peer_barrier(uint32_t*, uint32_t* const*, uint32_t*, int, int, double, hipStream_t);
int ovd_impl_peer_all_to_all(void*, const void* const*, long, int, int, hipStream_t);
int ovd_impl_peer_sum_f32(float*, const float* const*, long, int, hipStream_t);

#define ST(s) reinterpret_cast<hipStream_t>(s)
#define NONNULL(p) OVD_REQUIRE((p) != nullptr, "%s: null pointer argument '%s'", __func__, #p)

extern "C" {

const char* ovd_last_error(void) { return g_err; }
int ovd_abi_version(void) { return 1; }

int ovd_layernorm_modulate_bf16(const float* x, int64_t ldx, const float* shift, const float* scale,
                                const uint16_t* weight, const uint16_t* bias, uint16_t* out, int64_t ldo, int M,
                                int C, float eps, int round_norm, void* stream) {
    NONNULL(x); NONNULL(out);
    OVD_REQUIRE(M >= 0 && C > 0, "layernorm: bad shape M=%d C=%d", M, C);
    return ovd_impl_ln_bf16(const_cast<float*>(x), ldx, nullptr, 0, nullptr, shift, scale, weight, bias, out, ldo, M, C,
                            eps, round_norm, ST(stream));
}

int ovd_layernorm_modulate_f32(const float* x, int64_t ldx, const float* shift, const float* scale,
                               const uint16_t* weight, const uint16_t* bias, float* out, int64_t ldo, int M, int C,
                               float eps, int round_norm, void* stream) {
    NONNULL(x); NONNULL(out);
    OVD_REQUIRE(M >= 0 && C > 0, "layernorm: bad shape M=%d C=%d", M, C);
    return ovd_impl_ln_f32(const_cast<float*>(x), ldx, nullptr, 0, nullptr, shift, scale, weight, bias, out, ldo, M, C,
                           eps, round_norm, ST(stream));
}

int ovd_rmsnorm_rope_bf16(const uint16_t* in, int64_t ldi, const uint16_t* weight, const float* rope, int rope_rows,
                          uint16_t* out, int64_t ldo, int M, int C, int head_dim, float eps, int rows_per_sample,
                          int tok_offset, int n_rope, int gf, int gh, int gw, void* stream) {
    NONNULL(in); NONNULL(out);
    OVD_REQUIRE(weight != nullptr || rope != nullptr, "rmsnorm: neither weight nor rope given");
    OVD_REQUIRE(M >= 0 && C > 0, "rmsnorm: bad shape M=%d C=%d", M, C);
    return ovd_impl_rmsnorm_rope(in, ldi, weight, rope, rope_rows, out, ldo, M, C, head_dim, eps, rows_per_sample,
                                 tok_offset, n_rope, gf, gh, gw, ST(stream));
}

int ovd_qkv_norm_rope_bf16(const uint16_t* qkv, int64_t ld, const uint16_t* weight_q, const uint16_t* weight_k,
                           const float* rope, int rope_rows, uint16_t* out, int64_t ldo, int64_t out_part_stride,
                           int col_block, int64_t out_block_stride, int64_t out_sample_stride, int copy_v, int M, int C,
                           int head_dim, float eps, int rows_per_sample, int tok_offset, int n_rope, int gf, int gh,
                           int gw, void* stream) {
    NONNULL(qkv); NONNULL(weight_q); NONNULL(weight_k); NONNULL(rope); NONNULL(out);
    OVD_REQUIRE(M >= 0 && C > 0, "qkv_norm_rope: bad shape M=%d C=%d", M, C);
    return ovd_impl_qkv_norm_rope(qkv, ld, weight_q, weight_k, rope, rope_rows, out, nullptr, ldo, out_part_stride,
                                  col_block, out_block_stride, out_sample_stride, copy_v, M, C, head_dim, eps,
                                  rows_per_sample, tok_offset, n_rope, gf, gh, gw, ST(stream));
}

int ovd_qkv_norm_rope_scatter_bf16(const uint16_t* qkv, int64_t ld, const uint16_t* weight_q, const uint16_t* weight_k,
                                   const float* rope, int rope_rows, uint16_t* const* block_ptrs, int64_t ldo,
                                   int64_t out_part_stride, int col_block, int64_t out_sample_stride, int M, int C,
                                   int head_dim, float eps, int rows_per_sample, int tok_offset, int n_rope, int gf,
                                   int gh, int gw, void* stream) {
    NONNULL(qkv); NONNULL(weight_q); NONNULL(weight_k); NONNULL(rope); NONNULL(block_ptrs);
    OVD_REQUIRE(M >= 0 && C > 0, "qkv_norm_rope: bad shape M=%d C=%d", M, C);
    return ovd_impl_qkv_norm_rope(qkv, ld, weight_q, weight_k, rope, rope_rows, nullptr, block_ptrs, ldo,
                                  out_part_stride, col_block, 0, out_sample_stride, 1, M, C, head_dim, eps,
                                  rows_per_sample, tok_offset, n_rope, gf, gh, gw, ST(stream));
}

int ovd_gather_col_blocks_bf16(const uint16_t* src, uint16_t* dst, int64_t ldd, int64_t M, int nblk, int col_block,
                               void* stream) {
    NONNULL(src); NONNULL(dst);
    return ovd_impl_gather_col_blocks(src, nullptr, dst, ldd, M, nblk, col_block, 0, ST(stream));
}

int ovd_gather_col_blocks_from_bf16(const uint16_t* const* src_ptrs, uint16_t* dst, int64_t ldd, int64_t M, int nblk,
                                    int col_block, int first, void* stream) {
    NONNULL(src_ptrs); NONNULL(dst);
    return ovd_impl_gather_col_blocks(nullptr, src_ptrs, dst, ldd, M, nblk, col_block, first, ST(stream));
}

int ovd_gemm_bf16(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, const uint16_t* bias, void* out,
                  int64_t ldo, const float* gate, int M, int N, int K, int epilogue, void* stream) {
    NONNULL(A); NONNULL(W); NONNULL(out);
    OVD_REQUIRE(M >= 0, "gemm: negative M");
    return ovd_impl_gemm_bf16(A, lda, W, ldw, bias, out, ldo, gate, M, N, K, epilogue, ST(stream));
}

int ovd_flash_attn_fwd_bf16(const uint16_t* q, const uint16_t* k, const uint16_t* v, uint16_t* out, int B, int Lq,
                            int Lk, int N, int head_dim, int k_len, int64_t q_ld, int64_t k_ld, int64_t v_ld,
                            int64_t o_ld, int64_t q_bs, int64_t k_bs, int64_t v_bs, int64_t o_bs,
                            float softmax_scale, void* stream) {
    NONNULL(q); NONNULL(k); NONNULL(v); NONNULL(out);
    return ovd_impl_attn_fwd(q, k, v, out, B, Lq, Lk, N, head_dim, k_len, q_ld, k_ld, v_ld, o_ld, q_bs, k_bs, v_bs,
                             o_bs, softmax_scale, 1, nullptr, 0, ST(stream));
}

int ovd_flash_attn_tail_fwd_bf16(const uint16_t* q, const uint16_t* k, const uint16_t* v, uint16_t* out, int B,
                                 int Lq, int Lk, int N, int head_dim, int k_len, int last_key_copies, int64_t q_ld,
                                 int64_t k_ld, int64_t v_ld, int64_t o_ld, int64_t q_bs, int64_t k_bs,
                                 int64_t v_bs, int64_t o_bs, float softmax_scale, void* stream) {
    NONNULL(q); NONNULL(k); NONNULL(v); NONNULL(out);
    return ovd_impl_attn_fwd(q, k, v, out, B, Lq, Lk, N, head_dim, k_len, q_ld, k_ld, v_ld, o_ld, q_bs, k_bs, v_bs,
                             o_bs, softmax_scale, last_key_copies, nullptr, 0, ST(stream));
}

int64_t ovd_flash_attn_workspace_bytes(int B, int Lq, int N, int k_len) {
    if (B <= 0 || Lq <= 0 || N <= 0 || k_len <= 0) return 0;
    return ovd_impl_attn_workspace_bytes(B, Lq, N, k_len);
}

int ovd_flash_attn_ws_fwd_bf16(const uint16_t* q, const uint16_t* k, const uint16_t* v, uint16_t* out, int B, int Lq,
                               int Lk, int N, int head_dim, int k_len, int last_key_copies, int64_t q_ld,
                               int64_t k_ld, int64_t v_ld, int64_t o_ld, int64_t q_bs, int64_t k_bs, int64_t v_bs,
                               int64_t o_bs, float softmax_scale, void* workspace, int64_t workspace_bytes,
                               void* stream) {
    NONNULL(q); NONNULL(k); NONNULL(v); NONNULL(out);
    OVD_REQUIRE(workspace_bytes >= 0 && (workspace != nullptr || workspace_bytes == 0),
                "attention: workspace_bytes=%lld without a workspace", (long long)workspace_bytes);
    return ovd_impl_attn_fwd(q, k, v, out, B, Lq, Lk, N, head_dim, k_len, q_ld, k_ld, v_ld, o_ld, q_bs, k_bs, v_bs,
                             o_bs, softmax_scale, last_key_copies, workspace, workspace_bytes, ST(stream));
}

int ovd_flash_attn_static_fwd_bf16(const uint16_t* q, const uint16_t* k, const uint16_t* v, uint16_t* out, int B,
                                   int Lq, int Lk, int N, int head_dim, int k_len, int last_key_copies, int64_t q_ld,
                                   int64_t k_ld, int64_t v_ld, int64_t o_ld, int64_t q_bs, int64_t k_bs,
                                   int64_t v_bs, int64_t o_bs, float softmax_scale, void* stream) {
    NONNULL(q); NONNULL(k); NONNULL(v); NONNULL(out);
    return ovd_impl_attn_fwd_static(q, k, v, out, B, Lq, Lk, N, head_dim, k_len, q_ld, k_ld, v_ld, o_ld, q_bs, k_bs,
                                    v_bs, o_bs, softmax_scale, last_key_copies, ST(stream));
}

int ovd_quant_heads_i8(const uint16_t* x, int64_t ldx, int64_t x_bs, int8_t* q, int64_t ldq, int64_t q_bs, float* scale,
                       int B, int L, int N, int head_dim, int rows_per_scale, void* stream) {
    NONNULL(x); NONNULL(q); NONNULL(scale);
    return ovd_impl_quant_heads_i8(x, ldx, x_bs, q, ldq, q_bs, scale, B, L, N, head_dim, rows_per_scale, ST(stream));
}

int ovd_flash_attn_qk8_fwd_bf16(const int8_t* q8, const float* q_scale, const int8_t* k8, const float* k_scale,
                                const uint16_t* v, uint16_t* out, int B, int Lq, int Lk, int N, int head_dim, int k_len,
                                int64_t q_ld, int64_t k_ld, int64_t v_ld, int64_t o_ld, int64_t q_bs, int64_t k_bs,
                                int64_t v_bs, int64_t o_bs, float softmax_scale, void* workspace,
                                int64_t workspace_bytes, void* stream) {
    NONNULL(q8); NONNULL(q_scale); NONNULL(k8); NONNULL(k_scale); NONNULL(v); NONNULL(out);
    OVD_REQUIRE(workspace_bytes >= 0 && (workspace != nullptr || workspace_bytes == 0),
                "attention: workspace_bytes=%lld without a workspace", (long long)workspace_bytes);
    return ovd_impl_attn_i8_fwd(q8, q_scale, k8, k_scale, v, out, B, Lq, Lk, N, head_dim, k_len, q_ld, k_ld, v_ld, o_ld,
                                q_bs, k_bs, v_bs, o_bs, softmax_scale, workspace, workspace_bytes, ST(stream));
}

int ovd_patchify(const void* lat, int lat_is_bf16, uint16_t* rows, int Cin, int T, int H, int W, int pt, int ph,
                 int pw, void* stream) {
    NONNULL(lat); NONNULL(rows);
    return ovd_impl_patchify(lat, lat_is_bf16, rows, Cin, T, H, W, pt, ph, pw, ST(stream));
}

int ovd_time_sinusoid(const double* timesteps, const int* step_ptr, int step, float* out, int dim, void* stream) {
    NONNULL(timesteps); NONNULL(out);
    OVD_REQUIRE(step_ptr != nullptr || step >= 0, "time_sinusoid: negative step");
    return ovd_impl_time_sinusoid(timesteps, step_ptr, step, out, dim, ST(stream));
}

int ovd_gemv_f32(const float* x, const uint16_t* W, const uint16_t* b, float* y, int N, int K, int silu_in,
                 void* stream) {
    NONNULL(x); NONNULL(W); NONNULL(y);
    return ovd_impl_gemv_f32(x, W, b, y, N, K, silu_in, ST(stream));
}

int ovd_modulation_table(const uint16_t* const* params, const float* e0, float* out, int n_layers, int n6c,
                         void* stream) {
    NONNULL(params); NONNULL(e0); NONNULL(out);
    OVD_REQUIRE(n_layers >= 0 && n_layers <= 65535 && n6c > 0, "modulation: bad sizes");
    return ovd_impl_modulation(params, e0, out, n_layers, n6c, ST(stream));
}

int ovd_head_modulation(const uint16_t* mod, const float* e, float* out, int C, void* stream) {
    NONNULL(mod); NONNULL(e); NONNULL(out);
    return ovd_impl_head_modulation(mod, e, out, C, ST(stream));
}

int ovd_head_unpatchify(const float* y, int64_t ldy, const uint16_t* W, const uint16_t* bias, float* out, int n_tok,
                        int C, int Cout, int pt, int ph, int pw, int gf, int gh, int gw, int tok_offset,
                        void* stream) {
    NONNULL(y); NONNULL(W); NONNULL(out);
    return ovd_impl_head_unpatchify(y, ldy, W, bias, out, n_tok, C, Cout, pt, ph, pw, gf, gh, gw, tok_offset,
                                    ST(stream));
}

int ovd_cfg_unipc_step(const float* cond, const float* uncond, void* sample, void* last_sample, float* m1, float* m2,
                       const float* coef, float guide_scale, const int* step_ptr, int step, int64_t n,
                       int sample_is_bf16, void* stream) {
    NONNULL(cond); NONNULL(uncond); NONNULL(sample); NONNULL(last_sample); NONNULL(m1); NONNULL(m2); NONNULL(coef);
    OVD_REQUIRE(step_ptr != nullptr || step >= 0, "cfg_unipc: negative step");
    return ovd_impl_cfg_unipc_step(cond, uncond, sample, last_sample, m1, m2, coef, guide_scale, step_ptr, step, n,
                                   sample_is_bf16, ST(stream));
}

int ovd_cfg_dpm_step(const float* cond, const float* uncond, const void* sample, float* next, float* m1,
                     const float* coef, float guide_scale, const int* step_ptr, int step, int64_t n,
                     int sample_is_bf16, void* stream) {
    NONNULL(cond); NONNULL(uncond); NONNULL(sample); NONNULL(next); NONNULL(m1); NONNULL(coef);
    OVD_REQUIRE(step_ptr != nullptr || step >= 0, "cfg_dpm: negative step");
    return ovd_impl_cfg_dpm_step(cond, uncond, sample, next, m1, coef, guide_scale, step_ptr, step, n,
                                 sample_is_bf16, ST(stream));
}

int ovd_video_to_uint8(const float* in, uint8_t* out, int64_t voxel_stride, int64_t channel_stride, int C,
                       int64_t nvox, float lo, float hi, void* stream) {
    NONNULL(in); NONNULL(out);
    return ovd_impl_to_uint8(in, out, voxel_stride, channel_stride, C, nvox, lo, hi, ST(stream));
}

int ovd_quant_rows_fp8(const void* x, int64_t ldx, uint8_t* q, int64_t ldq, float* scale, int M, int K, void* stream) {
    NONNULL(x); NONNULL(q); NONNULL(scale);
    return ovd_impl_quant_rows_fp8((const bf16_t*)x, ldx, q, ldq, scale, M, K, ST(stream));
}

int ovd_layernorm_modulate_fp8(const float* x, int64_t ldx, const float* shift, const float* scale, const void* weight,
                               const void* bias, uint8_t* out, int64_t ldo, float* row_scale, int M, int C, float eps,
                               int round_norm, void* stream) {
    NONNULL(x); NONNULL(out); NONNULL(row_scale);
    return ovd_impl_ln_fp8(const_cast<float*>(x), ldx, nullptr, 0, nullptr, shift, scale, (const bf16_t*)weight,
                           (const bf16_t*)bias, out, ldo, row_scale, M, C, eps, round_norm, ST(stream));
}

int ovd_resid_layernorm_modulate(float* x, int64_t ldx, const uint16_t* res, int64_t ldr, const float* res_gate,
                                 const float* shift, const float* scale, const uint16_t* weight, const uint16_t* bias,
                                 void* out, int64_t ldo, int out_kind, float* row_scale, int M, int C, float eps,
                                 int round_norm, void* stream) {
    NONNULL(x); NONNULL(res); NONNULL(out);
    OVD_REQUIRE(M >= 0 && C > 0, "layernorm: bad shape M=%d C=%d", M, C);
    switch (out_kind) {
        case 0:
            return ovd_impl_ln_bf16(x, ldx, res, ldr, res_gate, shift, scale, weight, bias, (bf16_t*)out, ldo, M, C, eps,
                                    round_norm, ST(stream));
        case 1:
            return ovd_impl_ln_f32(x, ldx, res, ldr, res_gate, shift, scale, weight, bias, (float*)out, ldo, M, C, eps,
                                   round_norm, ST(stream));
        case 2:
            NONNULL(row_scale);
            return ovd_impl_ln_fp8(x, ldx, res, ldr, res_gate, shift, scale, weight, bias, (uint8_t*)out, ldo, row_scale,
                                   M, C, eps, round_norm, ST(stream));
        default:
            OVD_FAIL("layernorm: out_kind %d is not 0 (bf16), 1 (fp32) or 2 (e4m3 + row scales)", out_kind);
    }
}

int ovd_gemm_fp8(const uint8_t* A, int64_t lda, const float* a_scale, float a_scale_all, const uint8_t* W, int64_t ldw,
                 const float* w_scale, const void* bias, void* out, int64_t ldo, const float* gate, float out_scale,
                 int M, int N, int K, int epilogue, void* stream) {
    NONNULL(A); NONNULL(W); NONNULL(w_scale); NONNULL(out);
    return ovd_impl_gemm_fp8(A, lda, a_scale, a_scale_all, W, ldw, w_scale, (const bf16_t*)bias, out, ldo, gate,
                             out_scale, M, N, K, epilogue, ST(stream));
}

int ovd_resize_coeffs(int in_size, int out_size, int* bounds_host, int* coef_host, int ksize_cap) {
    if (in_size <= 0 || out_size <= 0) return -1;
    const int k = ovd_impl_resize_coeffs(in_size, out_size, nullptr, nullptr, 0);
    if (k > 64) return -1;                          // more than 31x reduction: not a video preprocessing case
    return ovd_impl_resize_coeffs(in_size, out_size, bounds_host, coef_host, ksize_cap);
}

int ovd_frames_resize_normalize(const uint8_t* frames, uint8_t* tmp, float* out, const int* bounds_w,
                                const int* coef_w, int ksize_w, const int* bounds_h, const int* coef_h, int ksize_h,
                                const float* lut, int64_t T, int H, int W, int top, int left, int crop_h, int crop_w,
                                int out_h, int out_w, void* stream) {
    NONNULL(frames); NONNULL(tmp); NONNULL(out); NONNULL(bounds_w); NONNULL(coef_w); NONNULL(bounds_h);
    NONNULL(coef_h); NONNULL(lut);
    return ovd_impl_frames_resize(frames, tmp, out, bounds_w, coef_w, ksize_w, bounds_h, coef_h, ksize_h, lut, T, H, W,
                                  top, left, crop_h, crop_w, out_h, out_w, ST(stream));
}

int ovd_t5_attention_bf16(const uint16_t* qkv, int64_t ld, const uint16_t* pos_embedding, const int32_t* bucket,
                          const int32_t* k_lens, uint16_t* out, int64_t ldo, int B, int L, int N, int head_dim,
                          void* stream) {
    NONNULL(qkv); NONNULL(pos_embedding); NONNULL(bucket); NONNULL(out);
    return ovd_impl_t5_attention(qkv, ld, pos_embedding, bucket, k_lens, out, ldo, B, L, N, head_dim, ST(stream));
}

int ovd_t5_gated_gelu_bf16(const uint16_t* h, int64_t ldh, uint16_t* out, int64_t ldo, int M, int F, void* stream) {
    NONNULL(h); NONNULL(out);
    return ovd_impl_t5_gated_gelu(h, ldh, out, ldo, M, F, ST(stream));
}

int ovd_add_bf16(uint16_t* x, const uint16_t* y, int64_t n, void* stream) {
    NONNULL(x); NONNULL(y);
    return ovd_impl_add_bf16(x, y, n, ST(stream));
}

int ovd_step_advance(int* step_ptr, void* stream) {
    NONNULL(step_ptr);
    return ovd_impl_step_advance(step_ptr, ST(stream));
}

int ovd_vae_conv3d_f32(const float* x, const float* prev2, const float* prev1, const float* rnorm,
                       const float* rnorm_prev2, const float* rnorm_prev1, const float* gamma, int silu,
                       const float* w, const float* bias, const float* resid, float* out, int T, int H, int W, int Cin,
                       int Cout, int KT, int KH, int KW, int stride_t, int stride_hw, int pad_t, int pad_hw,
                       int upsample2, int time_interleave, void* stream) {
    NONNULL(x); NONNULL(w); NONNULL(out);
    OVD_REQUIRE((prev2 == nullptr || rnorm == nullptr) || rnorm_prev2 != nullptr, "conv3d: prev2 needs its voxel norms");
    OVD_REQUIRE((prev1 == nullptr || rnorm == nullptr) || rnorm_prev1 != nullptr, "conv3d: prev1 needs its voxel norms");
    return ovd_impl_conv3d_cl(x, prev2, prev1, nullptr, nullptr, nullptr, rnorm, rnorm_prev2, rnorm_prev1, gamma, silu, w,
                              nullptr, bias, resid, out,
                              T, H, W, Cin, Cout, KT, KH, KW, stride_t, stride_hw, pad_t, pad_hw, upsample2,
                              time_interleave, ST(stream));
}

int ovd_vae_split_weights(const float* w, uint16_t* out, int64_t rows, int Cin, void* stream) {
    NONNULL(w); NONNULL(out);
    return ovd_impl_split_weights(w, out, rows, Cin, ST(stream));
}

int ovd_vae_conv3d_sw_f32(const float* x, const float* prev2, const float* prev1, const float* rnorm,
                          const float* rnorm_prev2, const float* rnorm_prev1, const float* gamma, int silu,
                          const uint16_t* w_split, const float* bias, const float* resid, float* out, int T, int H,
                          int W, int Cin, int Cout, int KT, int KH, int KW, int stride_t, int stride_hw, int pad_t,
                          int pad_hw, int upsample2, int time_interleave, void* stream) {
    NONNULL(x); NONNULL(w_split); NONNULL(out);
    OVD_REQUIRE((prev2 == nullptr || rnorm == nullptr) || rnorm_prev2 != nullptr, "conv3d: prev2 needs its voxel norms");
    OVD_REQUIRE((prev1 == nullptr || rnorm == nullptr) || rnorm_prev1 != nullptr, "conv3d: prev1 needs its voxel norms");
    return ovd_impl_conv3d_cl(x, prev2, prev1, nullptr, nullptr, nullptr, rnorm, rnorm_prev2, rnorm_prev1, gamma, silu,
                              nullptr, w_split, bias, resid,
                              out, T, H, W, Cin, Cout, KT, KH, KW, stride_t, stride_hw, pad_t, pad_hw, upsample2,
                              time_interleave, ST(stream));
}

int ovd_vae_act_split(const float* x, const float* gamma, int silu, uint16_t* out, int64_t nvox, int C, void* stream) {
    NONNULL(x); NONNULL(out);
    OVD_REQUIRE(nvox >= 0, "act_split: negative voxel count");
    return ovd_impl_act_split(x, gamma, silu, out, nvox, C, ST(stream));
}

int ovd_vae_conv3d_ps_f32(const uint16_t* x_split, const uint16_t* prev2_split, const uint16_t* prev1_split,
                          const uint16_t* w_split, const float* bias, const float* resid, float* out, int T, int H, int W,
                          int Cin, int Cout, int KT, int KH, int KW, int pad_t, int pad_hw, int upsample2, void* stream) {
    NONNULL(x_split); NONNULL(w_split); NONNULL(out);
    return ovd_impl_conv3d_cl(nullptr, nullptr, nullptr, x_split, prev2_split, prev1_split, nullptr, nullptr, nullptr,
                              nullptr, 0, nullptr, w_split, bias, resid, out, T, H, W, Cin, Cout, KT, KH, KW, 1, 1, pad_t,
                              pad_hw, upsample2, 0, ST(stream));
}

int ovd_vae_voxel_rnorm(const float* x, float* out, int64_t nvox, int C, void* stream) {
    NONNULL(x); NONNULL(out);
    return ovd_impl_voxel_rnorm(x, out, nvox, C, ST(stream));
}

int ovd_vae_to_channels_last(const float* in, float* out, const float* pre, const float* mul, const float* post, int C,
                             int Cp, int64_t nvox, void* stream) {
    NONNULL(in); NONNULL(out);
    return ovd_impl_to_cl(in, out, pre, mul, post, C, Cp, nvox, ST(stream));
}

int ovd_vae_from_channels_last(const float* in, float* out, const float* pre, const float* mul, const float* post,
                               int C, int Cs, int64_t nvox, int clamp, void* stream) {
    NONNULL(in); NONNULL(out);
    return ovd_impl_from_cl(in, out, pre, mul, post, C, Cs, nvox, clamp, ST(stream));
}

int ovd_softmax_rows_f32(float* s, int M, int N, float scale, void* stream) {
    NONNULL(s);
    return ovd_impl_softmax_rows(s, M, N, scale, ST(stream));
}

int ovd_peer_alloc(void** ptr, int64_t bytes, int fine_grained) {
    NONNULL(ptr);
    OVD_REQUIRE(bytes > 0, "peer_alloc: %lld bytes", (long long)bytes);
    return ovd_impl_peer_alloc(ptr, (size_t)bytes, fine_grained);
}

int ovd_peer_free(void* ptr) {
    NONNULL(ptr);
    return ovd_impl_peer_free(ptr);
}

int ovd_peer_export(void* ptr, unsigned char handle[64]) {
    NONNULL(ptr); NONNULL(handle);
    return ovd_impl_peer_export(ptr, handle);
}

int ovd_peer_open(const unsigned char handle[64], void** ptr) {
    NONNULL(handle); NONNULL(ptr);
    return ovd_impl_peer_open(handle, ptr);
}

int ovd_peer_close(void* ptr) {
    NONNULL(ptr);
    return ovd_impl_peer_close(ptr);
}

int ovd_peer_barrier(uint32_t* my_flags, uint32_t* const* peer_flags, uint32_t* state, int p, int rank,
                     double timeout_s, void* stream) {
    NONNULL(my_flags); NONNULL(peer_flags); NONNULL(state);
    return ovd_impl_peer_barrier(my_flags, peer_flags, state, p, rank, timeout_s, ST(stream));
}

int ovd_peer_all_to_all(void* recv, const void* const* peer_send, int64_t chunk_bytes, int p, int rank,
                        void* stream) {
    NONNULL(recv); NONNULL(peer_send);
    return ovd_impl_peer_all_to_all(recv, peer_send, (long)chunk_bytes, p, rank, ST(stream));
}

int ovd_peer_sum_f32(float* out, const float* const* peer_src, int64_t n, int p, void* stream) {
    NONNULL(out); NONNULL(peer_src);
    return ovd_impl_peer_sum_f32(out, peer_src, (long)n, p, ST(stream));
}

}  // extern "C"
