// Row-wise normalisation kernels (HBM-bound): one 64-lane wave owns one token row,
// the row lives in registers between the statistics pass and the write pass, loads
// and stores are 16 bytes per lane.
#include "common.h"

// ---------------------------------------------------------------------------------
// LayerNorm (biased variance, fp32) + optional affine (bf16 params) + optional
// AdaLN modulation  y = n * (1 + scale[c]) + shift[c]  -> bf16 or fp32.
// Reference: WanLayerNorm (omnivideo/modules/model.py:87-97) fused with the
// modulation of WanAttentionBlock.forward (:243,252) and Head.forward (:287-289).
// round_norm: the block-0 case where the LayerNorm input is a bf16 tensor and the
// normalised value is cast back to bf16 before the modulation.
// RES: the pending residual update of the layer before this norm is applied on the way in,
// x[m][c] += res[m][c] * res_gate[c] (res bf16 = the rounded output of that layer's last Linear,
// model.py:245-246,250,253-254), written back to x and then normalised - one pass over x
// instead of a read-modify-write in the GEMM epilogue followed by this kernel's read.
template <int VPL, typename OutT, bool RES>
__global__ void __launch_bounds__(256) ln_mod_kernel(float* __restrict__ x, long ldx,
                                                     const bf16_t* __restrict__ res, long ldr,
                                                     const float* __restrict__ res_gate,
                                                     const float* __restrict__ shift,
                                                     const float* __restrict__ scale,
                                                     const bf16_t* __restrict__ w,
                                                     const bf16_t* __restrict__ b, OutT* __restrict__ out,
                                                     long ldo, int M, int C, float eps, int round_norm) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float4* xr = reinterpret_cast<float4*>(x + (long)row * ldx);
    float4 v[VPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        v[i] = xr[i * 64 + lane];
        if (RES) {
            const uint2 rp = *reinterpret_cast<const uint2*>(res + (long)row * ldr + (i * 64 + lane) * 4);
            float4 g = make_float4(1.f, 1.f, 1.f, 1.f);
            if (res_gate) g = *reinterpret_cast<const float4*>(res_gate + (i * 64 + lane) * 4);
            v[i].x = fmaf(bf2f(rp.x & 0xffff), g.x, v[i].x);
            v[i].y = fmaf(bf2f(rp.x >> 16), g.y, v[i].y);
            v[i].z = fmaf(bf2f(rp.y & 0xffff), g.z, v[i].z);
            v[i].w = fmaf(bf2f(rp.y >> 16), g.w, v[i].w);
            xr[i * 64 + lane] = v[i];
        }
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        float a = v[i].x - mean, bb = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
        q += (a * a + bb * bb) + (c * c + d * d);
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int col = (i * 64 + lane) * 4;
        float y[4] = {(v[i].x - mean) * rstd, (v[i].y - mean) * rstd, (v[i].z - mean) * rstd,
                      (v[i].w - mean) * rstd};
        if (w) {
            const uint2 wp = *reinterpret_cast<const uint2*>(w + col);
            const uint2 bp = *reinterpret_cast<const uint2*>(b + col);
            y[0] = y[0] * bf2f(wp.x & 0xffff) + bf2f(bp.x & 0xffff);
            y[1] = y[1] * bf2f(wp.x >> 16) + bf2f(bp.x >> 16);
            y[2] = y[2] * bf2f(wp.y & 0xffff) + bf2f(bp.y & 0xffff);
            y[3] = y[3] * bf2f(wp.y >> 16) + bf2f(bp.y >> 16);
        }
        if (round_norm) {
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = round_bf(y[j]);
        }
        if (scale) {
            const float4 sc = *reinterpret_cast<const float4*>(scale + col);
            const float4 sh = *reinterpret_cast<const float4*>(shift + col);
            y[0] = y[0] * (1.f + sc.x) + sh.x;
            y[1] = y[1] * (1.f + sc.y) + sh.y;
            y[2] = y[2] * (1.f + sc.z) + sh.z;
            y[3] = y[3] * (1.f + sc.w) + sh.w;
        }
        if constexpr (sizeof(OutT) == 2) {
            uint2 o;
            o.x = pack_bf2(y[0], y[1]);
            o.y = pack_bf2(y[2], y[3]);
            *reinterpret_cast<uint2*>(out + (long)row * ldo + col) = o;
        } else {
            *reinterpret_cast<float4*>(out + (long)row * ldo + col) = make_float4(y[0], y[1], y[2], y[3]);
        }
    }
}

template <typename OutT>
static int launch_ln(float* x, long ldx, const bf16_t* res, long ldr, const float* res_gate, const float* shift,
                     const float* scale, const bf16_t* w, const bf16_t* b, OutT* out, long ldo, int M, int C,
                     float eps, int round_norm, hipStream_t st) {
    OVD_REQUIRE(C % 256 == 0, "layernorm: C=%d must be a multiple of 256", C);
    OVD_REQUIRE(res != nullptr || res_gate == nullptr, "layernorm: residual gate without a residual");
    OVD_REQUIRE(res == nullptr || (ldr % 4 == 0 && ldr >= C), "layernorm: bad residual row stride");
    OVD_REQUIRE((w == nullptr) == (b == nullptr), "layernorm: weight and bias must come together");
    OVD_REQUIRE((scale == nullptr) == (shift == nullptr), "layernorm: scale and shift must come together");
    OVD_REQUIRE(ldx % 4 == 0 && ldo % 4 == 0, "layernorm: row strides must be multiples of 4");
    if (M == 0) return 0;
    dim3 grid(ceil_div(M, 4)), blk(256);
#define LN_CASE(V)                                                                                         \
    case V:                                                                                                \
        if (res)                                                                                           \
            hipLaunchKernelGGL((ln_mod_kernel<V, OutT, true>), grid, blk, 0, st, x, ldx, res, ldr, res_gate,  \
                               shift, scale, w, b, out, ldo, M, C, eps, round_norm);                       \
        else                                                                                               \
            hipLaunchKernelGGL((ln_mod_kernel<V, OutT, false>), grid, blk, 0, st, x, ldx, res, ldr, res_gate, \
                               shift, scale, w, b, out, ldo, M, C, eps, round_norm);                       \
        break;
    switch (C / 256) {
        LN_CASE(1) LN_CASE(2) LN_CASE(3) LN_CASE(4) LN_CASE(6) LN_CASE(8) LN_CASE(12) LN_CASE(16) LN_CASE(20)
        default:
            OVD_FAIL("layernorm: unsupported width C=%d", C);
    }
#undef LN_CASE
    OVD_CHECK_LAUNCH("layernorm");
    return 0;
}

// x is written only when a residual is given
int ovd_impl_ln_bf16(float* x, long ldx, const bf16_t* res, long ldr, const float* res_gate, const float* shift,
                     const float* scale, const bf16_t* w, const bf16_t* b, bf16_t* out, long ldo, int M, int C,
                     float eps, int round_norm, hipStream_t st) {
    return launch_ln<bf16_t>(x, ldx, res, ldr, res_gate, shift, scale, w, b, out, ldo, M, C, eps, round_norm, st);
}
int ovd_impl_ln_f32(float* x, long ldx, const bf16_t* res, long ldr, const float* res_gate, const float* shift,
                    const float* scale, const bf16_t* w, const bf16_t* b, float* out, long ldo, int M, int C,
                    float eps, int round_norm, hipStream_t st) {
    return launch_ln<float>(x, ldx, res, ldr, res_gate, shift, scale, w, b, out, ldo, M, C, eps, round_norm, st);
}

// ---------------------------------------------------------------------------------
// Full-width RMSNorm over all heads + 3-axis interleaved-pair RoPE, bf16 in/out.
// Reference: WanRMSNorm (model.py:68-84) applied to the q/k projections (:137-138)
// followed by rope_apply (:37-65) and the cast to half precision that the attention
// entry performs (attention.py:59-83).  Rounding points follow the reference's bf16
// flow: normalised value -> bf16, times weight -> bf16, rotated value -> bf16.
// rope table: float2 (cos, sin) [3][rope_rows][D/2]; axis a serves the pair range
// it owns inside a head (frame: first D/2 - 2*(D/6) pairs, then height, then width).
// Token index = (row % rows_per_sample) + tok_offset; rows whose token index is
// >= n_rope are normalised but not rotated (zero-padded tail of the sequence).
template <int NCH>
__global__ void __launch_bounds__(256) rmsnorm_rope_kernel(const bf16_t* __restrict__ in, long ldi,
                                                           const bf16_t* __restrict__ weight,
                                                           const float2* __restrict__ rope, int rope_rows,
                                                           bf16_t* __restrict__ out, long ldo, int M, int C,
                                                           int D, float eps, int rows_per_sample,
                                                           int tok_offset, int n_rope, int gh, int gw) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nchunk = C >> 3;
    const uint4* src = reinterpret_cast<const uint4*>(in + (long)row * ldi);
    uint4 raw[NCH];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = i * 64 + lane;
        raw[i] = make_uint4(0, 0, 0, 0);
        if (c < nchunk) raw[i] = src[c];
        const uint32_t u[4] = {raw[i].x, raw[i].y, raw[i].z, raw[i].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a = bf2f(u[j] & 0xffff), b = bf2f(u[j] >> 16);
            ss += a * a + b * b;
        }
    }
    const float rinv = rsqrtf(wave_sum(ss) / (float)C + eps);

    const int tok = (row % rows_per_sample) + tok_offset;
    const bool rotate = rope != nullptr && tok < n_rope;
    int pf = 0, ph = 0, pw = 0;
    if (rotate) {
        pw = tok % gw;
        const int t2 = tok / gw;
        ph = t2 % gh;
        pf = t2 / gh;
    }
    const int half = D >> 1;
    const int n_f = half - 2 * (half / 3), n_h = half / 3;

#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = i * 64 + lane;
        if (c >= nchunk) continue;
        const int col = c * 8;
        const uint32_t u[4] = {raw[i].x, raw[i].y, raw[i].z, raw[i].w};
        uint32_t wu[4] = {0, 0, 0, 0};
        if (weight) {
            const uint4 wq = *reinterpret_cast<const uint4*>(weight + col);
            wu[0] = wq.x; wu[1] = wq.y; wu[2] = wq.z; wu[3] = wq.w;
        }
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a = bf2f(u[j] & 0xffff), b = bf2f(u[j] >> 16);
            if (weight) {   // weight == nullptr: rotation only, no normalisation
                a = round_bf(round_bf(a * rinv) * bf2f(wu[j] & 0xffff));
                b = round_bf(round_bf(b * rinv) * bf2f(wu[j] >> 16));
            }
            if (rotate) {
                const int p = ((col & (D - 1)) >> 1) + j;   // pair index inside the head
                const int axis = (p >= n_f) + (p >= n_f + n_h);
                const int pos = axis == 0 ? pf : (axis == 1 ? ph : pw);
                const float2 cs = rope[((long)axis * rope_rows + pos) * half + p];
                const float ra = a * cs.x - b * cs.y;
                const float rb_ = a * cs.y + b * cs.x;
                a = ra;
                b = rb_;
            }
            o[j] = pack_bf2(a, b);
        }
        *reinterpret_cast<uint4*>(out + (long)row * ldo + col) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

int ovd_impl_rmsnorm_rope(const bf16_t* in, long ldi, const bf16_t* weight, const float* rope, int rope_rows,
                          bf16_t* out, long ldo, int M, int C, int D, float eps, int rows_per_sample,
                          int tok_offset, int n_rope, int gf, int gh, int gw, hipStream_t st) {
    OVD_REQUIRE(C % 8 == 0 && ldi % 8 == 0 && ldo % 8 == 0, "rmsnorm: C and row strides must be multiples of 8");
    OVD_REQUIRE(rows_per_sample > 0, "rmsnorm: rows_per_sample must be positive");
    if (rope) {
        OVD_REQUIRE(D > 0 && (D & (D - 1)) == 0 && C % D == 0, "rmsnorm: head_dim must be a power of two dividing C");
        OVD_REQUIRE(gf > 0 && gh > 0 && gw > 0 && n_rope <= (long)gf * gh * gw,
                    "rmsnorm: rope grid (%d,%d,%d) does not cover %d tokens", gf, gh, gw, n_rope);
        OVD_REQUIRE(gf <= rope_rows && gh <= rope_rows && gw <= rope_rows, "rmsnorm: rope table too short");
    }
    if (M == 0) return 0;
    const int nch = ceil_div(C / 8, 64);
    dim3 grid(ceil_div(M, 4)), blk(256);
#define RN_CASE(V)                                                                                      \
    case V:                                                                                             \
        hipLaunchKernelGGL((rmsnorm_rope_kernel<V>), grid, blk, 0, st, in, ldi, weight,                 \
                           reinterpret_cast<const float2*>(rope), rope_rows, out, ldo, M, C, D, eps,    \
                           rows_per_sample, tok_offset, n_rope, gh, gw);                                \
        break;
    switch (nch) {
        RN_CASE(1) RN_CASE(2) RN_CASE(3) RN_CASE(4) RN_CASE(5) RN_CASE(6) RN_CASE(8) RN_CASE(10)
        default:
            OVD_FAIL("rmsnorm: unsupported width C=%d", C);
    }
#undef RN_CASE
    OVD_CHECK_LAUNCH("rmsnorm_rope");
    return 0;
}
