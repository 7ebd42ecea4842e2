// Row-wise normalisation kernels (HBM-bound): one 64-lane wave owns one token row (wide
// LayerNorm rows: two or four waves share it), the row lives in registers between the
// statistics pass and the write pass, loads are 16 bytes per lane.
#include "ln_row.h"

// ---------------------------------------------------------------------------------
// LayerNorm (biased variance, fp32) + optional affine (bf16 params) + optional
// AdaLN modulation  y = n * (1 + scale[c]) + shift[c]  -> bf16 or fp32.
// Reference: WanLayerNorm (omnivideo/modules/model.py:87-97) fused with the
// modulation of WanAttentionBlock.forward (:243,252) and Head.forward (:287-289).
// round_norm: the block-0 case where the LayerNorm input is a bf16 tensor and the
// normalised value is cast back to bf16 before the modulation.
// Row arithmetic: ln_row() of ln_row.h.
template <int VPL, int WPR, typename OutT, int F>
__global__ void __launch_bounds__(256) ln_mod_kernel(float* __restrict__ x, long ldx,
                                                     const bf16_t* __restrict__ res, long ldr,
                                                     const float* __restrict__ res_gate,
                                                     const float* __restrict__ shift,
                                                     const float* __restrict__ scale,
                                                     const bf16_t* __restrict__ w,
                                                     const bf16_t* __restrict__ b, OutT* __restrict__ out,
                                                     long ldo, int M, int C, float eps, int round_norm) {
    __shared__ float red[2][4];
    float4 y[VPL];
    int row;
    if (!ln_row<VPL, WPR, F>(x, ldx, res, ldr, res_gate, shift, scale, w, b, M, C, eps, round_norm, red, row, y))
        return;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int col = ln_piece<WPR>(i) * 4;
        if constexpr (sizeof(OutT) == 2) {
            uint2 o;
            o.x = pack_bf2(y[i].x, y[i].y);
            o.y = pack_bf2(y[i].z, y[i].w);
            *reinterpret_cast<uint2*>(out + (long)row * ldo + col) = o;
        } else {
            *reinterpret_cast<float4*>(out + (long)row * ldo + col) = y[i];
        }
    }
}

template <int V, int WPR, typename OutT, int F>
static void launch_ln_one(hipStream_t st, float* x, long ldx, const bf16_t* res, long ldr,
                          const float* res_gate, const float* shift, const float* scale, const bf16_t* w,
                          const bf16_t* b, OutT* out, long ldo, int M, int C, float eps, int round_norm) {
    hipLaunchKernelGGL((ln_mod_kernel<V, WPR, OutT, F>), dim3(ceil_div(M, 4 / WPR)), dim3(256), 0, st, x, ldx,
                       res, ldr, res_gate, shift, scale, w, b, out, ldo, M, C, eps, round_norm);
}

template <int V, int WPR, typename OutT, typename... A>
static void launch_ln_flags(int flags, A... a) {
    switch (flags) {
#define LN_F(F) case F: launch_ln_one<V, WPR, OutT, F>(a...); break;
        LN_FOR_EACH_FLAGS(LN_F)
#undef LN_F
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
    const int flags = (res ? LN_RES : 0) | (res_gate ? LN_GATE : 0) | (w ? LN_AFF : 0) | (scale ? LN_MOD : 0);
    // C = 256 * V * WPR: V pieces of 16 bytes per lane, WPR waves per row
#define LN_CASE(V, WPR)                                                                                   \
    case V * WPR:                                                                                         \
        launch_ln_flags<V, WPR, OutT>(flags, st, x, ldx, res, ldr, res_gate, shift, scale, w, b, out, ldo, M, \
                                      C, eps, round_norm);                                                \
        break;
    switch (C / 256) {
        LN_FOR_EACH_SHAPE(LN_CASE)
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
// WEIGHT / ROPE (compile time) say which operands exist; out-of-range chunks and rows that are not
// rotated load from clamped addresses and discard the value, so the kernel has no branch between
// its loads and every load is in flight before the first use.
// WPR waves share one row, as in ln_row.h: chunk i of a lane is 16-byte chunk (i * WPR + wave) * 64 + lane.
// Several row-parts in one launch (blockIdx.y = part): the q | k | v thirds of a fused projection row.
// Part p reads columns [p * in_part_stride, + C) of the input row and uses weight[p]; part 2 (v) is copied
// unchanged.  The output row of part p goes to out + p * out_part_stride, split into column blocks of
// cb8 16-byte chunks that sit out_block_stride elements apart, at row offset
// (row / rows_per_sample) * out_sample_stride + (row % rows_per_sample) * ldo: in place this is the input
// layout itself; for sequence parallelism it is the all-to-all send image [part][sample][dest rank][row][C/p].
// With block_ptrs (a device array, one base address per column block) block b goes to block_ptrs[b] instead
// of out + b * out_block_stride: the column blocks then land in different buffers - the receive buffers of
// the sequence-parallel peers, mapped into this process, so that the kernel that produces q and k is also
// the sender.
struct RnParts {
    const bf16_t* weight[2];
    bf16_t* const* block_ptrs;
    long in_part_stride, out_part_stride, out_block_stride, out_sample_stride;
    int cb8;
};

template <int NCH, int WPR, bool WEIGHT, bool ROPE>
__global__ void __launch_bounds__(256) rmsnorm_rope_kernel(const bf16_t* __restrict__ in, long ldi, RnParts pp,
                                                           const float2* __restrict__ rope, int rope_rows,
                                                           bf16_t* __restrict__ out, long ldo, int M, int C,
                                                           int D, float eps, int rows_per_sample,
                                                           int tok_offset, int n_rope, int gh, int gw) {
    __shared__ float red[4];
    int row = blockIdx.x * (4 / WPR) + (threadIdx.x >> 6) / WPR;
    const bool live = row < M;
    if constexpr (WPR == 1) {
        if (!live) return;
    } else {
        row = live ? row : M - 1;     // keeps the block's barrier matched; stores nothing
    }
    const int part = blockIdx.y;
    const int nchunk = C >> 3;
    auto chunk = [&](int i) -> int { return (i * WPR + (int)((threadIdx.x >> 6) & (WPR - 1))) * 64 + (int)(threadIdx.x & 63); };
    const uint4* src = reinterpret_cast<const uint4*>(in + (long)row * ldi + part * pp.in_part_stride);
    const int srow = row % rows_per_sample;
    const long orow = part * pp.out_part_stride + (long)(row / rows_per_sample) * pp.out_sample_stride + (long)srow * ldo;
    auto ochunk = [&](int c) -> uint4* {      // where 16-byte chunk c of this row goes
        const int blk = pp.cb8 == nchunk ? 0 : c / pp.cb8;
        bf16_t* base = pp.block_ptrs ? pp.block_ptrs[blk] : out + (long)blk * pp.out_block_stride;
        return reinterpret_cast<uint4*>(base + orow + (long)(c - blk * pp.cb8) * 8);
    };
    uint4 raw[NCH], wq[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) raw[i] = src[min(chunk(i), nchunk - 1)];
    if (part == 2) {                  // v: no normalisation, no rotation
        if (live) {
#pragma unroll
            for (int i = 0; i < NCH; ++i)
                if (chunk(i) < nchunk) *ochunk(chunk(i)) = raw[i];
        }
        return;                       // (block-uniform: every wave of the block has the same part)
    }
    if constexpr (WEIGHT) {
        const uint4* wsrc = reinterpret_cast<const uint4*>(pp.weight[part]);
#pragma unroll
        for (int i = 0; i < NCH; ++i) wq[i] = wsrc[min(chunk(i), nchunk - 1)];
    }
    const int tok = srow + tok_offset;
    const bool rotate = ROPE && tok < n_rope;
    float2 cs[NCH][4];
    if constexpr (ROPE) {
        const int t = rotate ? tok : 0;
        const int pos[3] = {t / gw / gh, (t / gw) % gh, t % gw};
        const int half = D >> 1;
        const int n_f = half - 2 * (half / 3), n_h = half / 3;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int col = min(chunk(i), nchunk - 1) * 8;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int p = ((col & (D - 1)) >> 1) + j;   // pair index inside the head
                const int axis = (p >= n_f) + (p >= n_f + n_h);
                cs[i][j] = rope[((long)axis * rope_rows + pos[axis]) * half + p];
            }
        }
    }
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        if (chunk(i) >= nchunk) raw[i] = make_uint4(0, 0, 0, 0);
        const uint32_t u[4] = {raw[i].x, raw[i].y, raw[i].z, raw[i].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a = bf2f(u[j] & 0xffff), b = bf2f(u[j] >> 16);
            ss += a * a + b * b;
        }
    }
    const float rinv = rsqrtf(row_sum<WPR>(ss, red) / (float)C + eps);
    if (!live) return;

#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = chunk(i);
        const uint32_t u[4] = {raw[i].x, raw[i].y, raw[i].z, raw[i].w};
        const uint32_t wu[4] = {wq[i].x, wq[i].y, wq[i].z, wq[i].w};
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a = bf2f(u[j] & 0xffff), b = bf2f(u[j] >> 16);
            if constexpr (WEIGHT) {   // without a weight: rotation only, no normalisation
                a = round_bf(round_bf(a * rinv) * bf2f(wu[j] & 0xffff));
                b = round_bf(round_bf(b * rinv) * bf2f(wu[j] >> 16));
            }
            if constexpr (ROPE) {
                const float ra = a * cs[i][j].x - b * cs[i][j].y;
                const float rb_ = a * cs[i][j].y + b * cs[i][j].x;
                a = rotate ? ra : a;
                b = rotate ? rb_ : b;
            }
            o[j] = pack_bf2(a, b);
        }
        if (c < nchunk) *ochunk(c) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

static int launch_rmsnorm_rope(const bf16_t* in, long ldi, const RnParts& pp, int parts, bool has_weight,
                               const float* rope, int rope_rows, bf16_t* out, long ldo, int M, int C, int D, float eps,
                               int rows_per_sample, int tok_offset, int n_rope, int gf, int gh, int gw,
                               hipStream_t st) {
    OVD_REQUIRE(C % 8 == 0 && ldi % 8 == 0 && ldo % 8 == 0, "rmsnorm: C and row strides must be multiples of 8");
    OVD_REQUIRE(rows_per_sample > 0, "rmsnorm: rows_per_sample must be positive");
    OVD_REQUIRE(has_weight || rope != nullptr, "rmsnorm: neither weight nor rope given");
    if (rope) {
        OVD_REQUIRE(D > 0 && (D & (D - 1)) == 0 && C % D == 0, "rmsnorm: head_dim must be a power of two dividing C");
        OVD_REQUIRE(gf > 0 && gh > 0 && gw > 0 && n_rope <= (long)gf * gh * gw,
                    "rmsnorm: rope grid (%d,%d,%d) does not cover %d tokens", gf, gh, gw, n_rope);
        OVD_REQUIRE(gf <= rope_rows && gh <= rope_rows && gw <= rope_rows, "rmsnorm: rope table too short");
    }
    if (M == 0) return 0;
    // 16-byte chunks per lane when one wave holds the row; above 4 the row is shared by 2 or 4 waves
    const int nch = ceil_div(C / 8, 64);
    OVD_REQUIRE(nch <= 12, "rmsnorm: unsupported width C=%d (at most 6144)", C);
    const int wpr = nch <= 4 ? 1 : (nch <= 8 ? 2 : 4);
    dim3 grid(ceil_div(M, 4 / wpr), parts), blk(256);
#define RN_LAUNCH(V, P, W, R)                                                                           \
    hipLaunchKernelGGL((rmsnorm_rope_kernel<V, P, W, R>), grid, blk, 0, st, in, ldi, pp,                \
                       reinterpret_cast<const float2*>(rope), rope_rows, out, ldo, M, C, D, eps,        \
                       rows_per_sample, tok_offset, n_rope, gh, gw)
#define RN_CASE(V, P)                                                                                   \
    case V * 8 + P:                                                                                     \
        if (has_weight && rope) RN_LAUNCH(V, P, true, true);                                            \
        else if (has_weight) RN_LAUNCH(V, P, true, false);                                              \
        else RN_LAUNCH(V, P, false, true);                                                              \
        break;
    switch (ceil_div(nch, wpr) * 8 + wpr) {
        RN_CASE(1, 1) RN_CASE(2, 1) RN_CASE(3, 1) RN_CASE(4, 1) RN_CASE(3, 2) RN_CASE(4, 2) RN_CASE(3, 4)
        default:
            OVD_FAIL("rmsnorm: unsupported width C=%d", C);
    }
#undef RN_LAUNCH
#undef RN_CASE
    OVD_CHECK_LAUNCH("rmsnorm_rope");
    return 0;
}

int ovd_impl_rmsnorm_rope(const bf16_t* in, long ldi, const bf16_t* weight, const float* rope, int rope_rows,
                          bf16_t* out, long ldo, int M, int C, int D, float eps, int rows_per_sample,
                          int tok_offset, int n_rope, int gf, int gh, int gw, hipStream_t st) {
    RnParts pp;
    pp.weight[0] = pp.weight[1] = weight;
    pp.block_ptrs = nullptr;
    pp.in_part_stride = pp.out_part_stride = pp.out_block_stride = 0;
    pp.out_sample_stride = (long)rows_per_sample * ldo;       // plain rows: row m at m * ldo
    pp.cb8 = C / 8;
    return launch_rmsnorm_rope(in, ldi, pp, 1, weight != nullptr, rope, rope_rows, out, ldo, M, C, D, eps,
                               rows_per_sample, tok_offset, n_rope, gf, gh, gw, st);
}

// q | k | v thirds of fused projection rows [M, 3C] in one launch: RMSNorm (own weight each) + RoPE on q and k;
// v is copied when copy_v (its destination differs from its source).  Output layout: struct RnParts.
// out == nullptr: the column blocks go to block_ptrs[b] (a device array of C / col_block addresses).
int ovd_impl_qkv_norm_rope(const bf16_t* qkv, long ld, const bf16_t* wq, const bf16_t* wk, const float* rope,
                           int rope_rows, bf16_t* out, bf16_t* const* block_ptrs, long ldo, long out_part_stride,
                           int col_block, long out_block_stride, long out_sample_stride, int copy_v, int M, int C,
                           int D, float eps, int rows_per_sample, int tok_offset, int n_rope, int gf, int gh, int gw,
                           hipStream_t st) {
    OVD_REQUIRE((out != nullptr) != (block_ptrs != nullptr), "qkv_norm_rope: exactly one of out / block_ptrs");
    OVD_REQUIRE(col_block > 0 && col_block % 8 == 0 && C % col_block == 0,
                "qkv_norm_rope: col_block=%d must be a multiple of 8 dividing C=%d", col_block, C);
    OVD_REQUIRE(ld >= 3L * C, "qkv_norm_rope: input rows shorter than 3*C");
    OVD_REQUIRE(out_part_stride % 8 == 0 && out_block_stride % 8 == 0 && out_sample_stride % 8 == 0,
                "qkv_norm_rope: output strides must be multiples of 8 elements");
    RnParts pp;
    pp.weight[0] = wq;
    pp.weight[1] = wk;
    pp.block_ptrs = block_ptrs;
    pp.in_part_stride = C;
    pp.out_part_stride = out_part_stride;
    pp.out_block_stride = out_block_stride;
    pp.out_sample_stride = out_sample_stride;
    pp.cb8 = col_block / 8;
    return launch_rmsnorm_rope(qkv, ld, pp, copy_v ? 3 : 2, true, rope, rope_rows, out, ldo, M, C, D, eps,
                               rows_per_sample, tok_offset, n_rope, gf, gh, gw, st);
}

// dst[m][b * cb + c] = src[b][m][c]: column blocks that arrive one after another (the attention output of
// each head group after the sequence-parallel all-to-all) back into rows.  16 bytes per lane.  With src_ptrs
// (device array) block b is read from src_ptrs[b][m][c] instead: the blocks then sit in different buffers -
// the attention outputs of the sequence-parallel peers, mapped into this process - and the gather is also
// the receive.  The blocks are walked starting at block `first`, so that ranks read from different peers.
__global__ void __launch_bounds__(256) gather_col_blocks_kernel(const uint4* __restrict__ src,
                                                                const uint4* const* __restrict__ src_ptrs,
                                                                uint4* __restrict__ dst, long M, int nblk, int cb8,
                                                                long ldd8, int first) {
    const long per = M * cb8;
    const long total = per * nblk;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % cb8);
        const long t = i / cb8;
        const long m = t % M;
        int b = (int)(t / M) + first;
        b = b >= nblk ? b - nblk : b;
        const uint4* s = src_ptrs ? src_ptrs[b] : src + b * per;
        dst[m * ldd8 + (long)b * cb8 + c] = s[m * cb8 + c];
    }
}

int ovd_impl_gather_col_blocks(const bf16_t* src, const bf16_t* const* src_ptrs, bf16_t* dst, long ldd, long M,
                               int nblk, int col_block, int first, hipStream_t st) {
    OVD_REQUIRE((src != nullptr) != (src_ptrs != nullptr), "gather_col_blocks: exactly one of src / src_ptrs");
    OVD_REQUIRE(col_block > 0 && col_block % 8 == 0 && ldd % 8 == 0 && ldd >= (long)nblk * col_block,
                "gather_col_blocks: col_block=%d and the destination stride must be multiples of 8", col_block);
    OVD_REQUIRE(M >= 0 && nblk > 0 && first >= 0 && first < nblk, "gather_col_blocks: bad shape");
    if (M == 0) return 0;
    const long total = M * nblk * (col_block / 8);
    const int blocks = (int)(total / 256 + 1 < 16384 ? total / 256 + 1 : 16384);
    hipLaunchKernelGGL(gather_col_blocks_kernel, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const uint4*>(src),
                       reinterpret_cast<const uint4* const*>(src_ptrs), reinterpret_cast<uint4*>(dst), M, nblk,
                       col_block / 8, ldd / 8, first);
    OVD_CHECK_LAUNCH("gather_col_blocks");
    return 0;
}
