// One row of LayerNorm (+ pending residual) (+ affine) (+ AdaLN modulation), shared by the bf16 / fp32
// kernel of norm.hip and the e4m3 kernel of fp8.hip.
// Reference: WanLayerNorm (omnivideo/modules/model.py:87-97) with the modulation of
// WanAttentionBlock.forward (:243,252) and Head.forward (:287-289); the residual is the gated update
// of :245-246,250,253-254 that precedes the norm.
//
// F (compile time) says which operands exist, so the row is one straight block of code: every load
// of the row phase is issued before the first use and the per-channel vectors of the write phase are
// fetched under the two reductions.  With run-time tests each 16-byte piece became its own
// load -> wait -> store round trip (18 dependent memory latencies per 1536-wide row).
// WPR waves share one row (wide rows: a wave keeps at most 6 x 16 bytes of the row per lane); their
// partial sums meet in LDS and are added in a fixed order.
#pragma once
#include "common.h"

constexpr int LN_RES = 1, LN_GATE = 2, LN_AFF = 4, LN_MOD = 8;

template <int WPR>
__device__ __forceinline__ float row_sum(float s, float* red) {
    s = wave_sum(s);
    if constexpr (WPR == 1) return s;
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[wave] = s;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < WPR; ++k) t += red[(wave & ~(WPR - 1)) + k];
    return t;
}

template <int WPR>
__device__ __forceinline__ float row_max(float s, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s = fmaxf(s, __shfl_xor(s, o, 64));
    if constexpr (WPR == 1) return s;
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[wave] = s;
    __syncthreads();
    float t = red[wave & ~(WPR - 1)];
#pragma unroll
    for (int k = 1; k < WPR; ++k) t = fmaxf(t, red[(wave & ~(WPR - 1)) + k]);
    return t;
}

// float4 index, inside the row, of piece i of this lane (blocks of 256 threads = 4 waves)
template <int WPR>
__device__ __forceinline__ int ln_piece(int i) {
    return (i * WPR + ((threadIdx.x >> 6) & (WPR - 1))) * 64 + (threadIdx.x & 63);
}

// Normalised (and affine / modulated) values of this lane's pieces of row `row` in y[].
// Returns false for a wave whose row lies past M: it has written nothing and must write nothing (with
// WPR > 1 it has recomputed row M-1 so that the barriers of the block stay matched; the caller goes on
// to any further block-wide reduction before leaving).
// red: LDS float[2][4].  round_norm: the normalised value is rounded to bf16 before the modulation (the
// block-0 case where the LayerNorm input is a bf16 tensor).
template <int VPL, int WPR, int F>
__device__ __forceinline__ bool ln_row(float* __restrict__ x, long ldx, const bf16_t* __restrict__ res, long ldr,
                                       const float* __restrict__ res_gate, const float* __restrict__ shift,
                                       const float* __restrict__ scale, const bf16_t* __restrict__ w,
                                       const bf16_t* __restrict__ b, int M, int C, float eps, int round_norm,
                                       float (*red)[4], int& row, float4 (&y)[VPL]) {
    constexpr bool RES = (F & LN_RES) != 0, GATE = (F & LN_GATE) != 0, AFF = (F & LN_AFF) != 0,
                   MOD = (F & LN_MOD) != 0;
    row = blockIdx.x * (4 / WPR) + (threadIdx.x >> 6) / WPR;
    const bool live = row < M;
    if constexpr (WPR == 1) {
        if (!live) return false;
    } else {
        row = live ? row : M - 1;
    }
    float4* xr = reinterpret_cast<float4*>(x + (long)row * ldx);
    float4 v[VPL];
    uint2 rp[VPL];
    float4 g[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) v[i] = xr[ln_piece<WPR>(i)];
    if constexpr (RES) {
#pragma unroll
        for (int i = 0; i < VPL; ++i)
            rp[i] = *reinterpret_cast<const uint2*>(res + (long)row * ldr + ln_piece<WPR>(i) * 4);
    }
    if constexpr (GATE) {
#pragma unroll
        for (int i = 0; i < VPL; ++i) g[i] = *reinterpret_cast<const float4*>(res_gate + ln_piece<WPR>(i) * 4);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        if constexpr (RES) {
            const float4 gg = GATE ? g[i] : make_float4(1.f, 1.f, 1.f, 1.f);
            v[i].x = fmaf(bf2f(rp[i].x & 0xffff), gg.x, v[i].x);
            v[i].y = fmaf(bf2f(rp[i].x >> 16), gg.y, v[i].y);
            v[i].z = fmaf(bf2f(rp[i].y & 0xffff), gg.z, v[i].z);
            v[i].w = fmaf(bf2f(rp[i].y >> 16), gg.w, v[i].w);
            if (live) xr[ln_piece<WPR>(i)] = v[i];
        }
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    // per-channel vectors of the write phase: in flight while the statistics are reduced
    uint2 wp[VPL], bp[VPL];
    float4 sc[VPL], sh[VPL];
    if constexpr (AFF) {
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            wp[i] = *reinterpret_cast<const uint2*>(w + ln_piece<WPR>(i) * 4);
            bp[i] = *reinterpret_cast<const uint2*>(b + ln_piece<WPR>(i) * 4);
        }
    }
    if constexpr (MOD) {
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            sc[i] = *reinterpret_cast<const float4*>(scale + ln_piece<WPR>(i) * 4);
            sh[i] = *reinterpret_cast<const float4*>(shift + ln_piece<WPR>(i) * 4);
        }
    }
    const float mean = row_sum<WPR>(s, red[0]) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        float a = v[i].x - mean, bb = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
        q += (a * a + bb * bb) + (c * c + d * d);
    }
    const float rstd = rsqrtf(row_sum<WPR>(q, red[1]) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        float t[4] = {(v[i].x - mean) * rstd, (v[i].y - mean) * rstd, (v[i].z - mean) * rstd,
                      (v[i].w - mean) * rstd};
        if constexpr (AFF) {
            t[0] = t[0] * bf2f(wp[i].x & 0xffff) + bf2f(bp[i].x & 0xffff);
            t[1] = t[1] * bf2f(wp[i].x >> 16) + bf2f(bp[i].x >> 16);
            t[2] = t[2] * bf2f(wp[i].y & 0xffff) + bf2f(bp[i].y & 0xffff);
            t[3] = t[3] * bf2f(wp[i].y >> 16) + bf2f(bp[i].y >> 16);
        }
        if (round_norm) {
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] = round_bf(t[j]);
        }
        if constexpr (MOD) {
            t[0] = t[0] * (1.f + sc[i].x) + sh[i].x;
            t[1] = t[1] * (1.f + sc[i].y) + sh[i].y;
            t[2] = t[2] * (1.f + sc[i].z) + sh[i].z;
            t[3] = t[3] * (1.f + sc[i].w) + sh[i].w;
        }
        y[i] = make_float4(t[0], t[1], t[2], t[3]);
    }
    return live;
}

// instantiated (pieces per lane, waves per row) splits, C = 256 * pieces * waves, and operand flag sets
// (a gate needs a residual, so LN_GATE never appears without LN_RES)
#define LN_FOR_EACH_SHAPE(X) X(1, 1) X(2, 1) X(3, 1) X(4, 1) X(6, 1) X(4, 2) X(6, 2) X(4, 4) X(5, 4)
#define LN_FOR_EACH_FLAGS(X) X(0) X(1) X(3) X(4) X(5) X(7) X(8) X(9) X(11) X(12) X(13) X(15)
