// Causal 3D VAE kernels, channels-last: activations are [T,H,W,C] fp32 so that one voxel's
// channel vector is contiguous (coalesced 16-byte loads into LDS, 128-byte coalesced
// stores of 32 output channels per voxel).
//
// conv3d_cl_kernel is an implicit GEMM on the matrix cores, fp32 in and out.  Products run
// either as split bf16 (default: each operand = bf16 hi + bf16 lo, three
// v_mfma_f32_32x32x16_bf16 per product, fp32 accumulation, ~2^-16 relative error: finer than
// the TF32 convolutions the reference's fp32 VAE gets from cuDNN by default) or in exact fp32
// (v_mfma_f32_32x32x2_f32: an fmaf chain; OVD_VAE_EXACT_FP32=1):
//   out[t,h,w,co] = bias[co] + sum_{kt,kh,kw,ci} act(in[t*st+kt-pt, h*s+kh-ph, w*s+kw-ph, ci]) * W[kt,kh,kw][co][ci]
// One workgroup = 4 waves = 128 output voxels (an 8x16 patch of one frame, or 128
// consecutive voxels for 1x1 kernels) x 32*NB output channels.  Per (kt, 32-channel
// chunk) the input halo patch is staged once in LDS and reused by all 9 spatial taps as
// shifted windows; per tap the [32*NB x 32] weight slice is staged and each wave issues
// NB*16 MFMAs (K=32).  LDS rows are padded to 36 floats so the 16-byte fragment reads of
// 16 neighbouring voxels / weight rows land on distinct bank slots.
// Fused into the kernel: the time-causal left context (two frame pointers from the
// previous chunk, or zeros), per-voxel RMS norm * gamma and SiLU on the input side,
// nearest-neighbour 2x spatial upsampling of the input, strided (downsampling) convs,
// bias, residual add and the frame interleave of the temporal upsampler on the output
// side.  Reference: CausalConv3d / RMS_norm / Resample / ResidualBlock in
// omnivideo/modules/vae2_1.py:16-219.
#include "common.h"

namespace {

constexpr int LDV = 36;              // floats per staged voxel / weight row (32 + 4 pad)
constexpr int MAX_HALO = 17 * 33;    // stride-2 3x3 conv on an 8x16 tile

struct ConvArgs {
    const float* x; const float* prev2; const float* prev1;      // frames t>=0, t=-2, t=-1
    const float* rn; const float* rn_prev2; const float* rn_prev1;   // per-voxel 1/||x|| (or null)
    const float* gamma; const float* w; const float* bias; const float* resid; float* out;
    const uint16_t* ws;                  // weights already split (ovd_vae_split_weights), or null
    const uint16_t* xs; const uint16_t* prev2s; const uint16_t* prev1s;   // input already activated and split
                                         // (ovd_vae_act_split): [voxel][chunk][32 hi | 32 lo]; x / prev* are then null
    int T, H, W, Cin, Cout, To, Ho, Wo;
    int KT, KH, KW, st, s, pt, ph;       // kernel, temporal stride, spatial stride, paddings
    int ups, til, silu, TW;              // input 2x upsample, time-interleaved output, SiLU, tile width
    float norm_scale;                    // sqrt(Cin) when gamma != null
};

__device__ __forceinline__ float silu_f(float x) {
    return x * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}

// x = hi + lo with hi = bf16(x), lo = bf16(x - hi): 16 mantissa bits in two bf16 values
__device__ __forceinline__ void split_bf16(float x, uint32_t& hi, uint32_t& lo) {
    hi = f2bf(x);
    lo = f2bf(x - bf2f(hi));
}
// four floats -> 4 hi + 4 lo bf16, stored at p (hi) and p + 64 bytes (lo)
__device__ __forceinline__ void store_split4(char* p, const float4& v) {
    uint32_t h[4], l[4];
    split_bf16(v.x, h[0], l[0]);
    split_bf16(v.y, h[1], l[1]);
    split_bf16(v.z, h[2], l[2]);
    split_bf16(v.w, h[3], l[3]);
    *reinterpret_cast<uint2*>(p) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
    *reinterpret_cast<uint2*>(p + 64) = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
}

// SPLIT = false: exact fp32 products (v_mfma_f32_32x32x2_f32).  SPLIT = true: both operands are
// split into bf16 hi + lo while they are staged into LDS (same 144-byte rows: 32 hi | 32 lo | pad)
// and each product is hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation:
// ~2^-16 relative error per product at 3/16 of the fp32 matrix time.
template <int NB, bool SPLIT>
__global__ void __launch_bounds__(256) conv3d_cl_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* halo = lds;                          // [HR*WR][LDV]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int TW = a.TW, TH = 128 / TW;          // TW is a power of two (16 or 128)
    const int tw_shift = __builtin_ctz(TW);
    const int tiles_w = (a.Wo + TW - 1) / TW, tiles_h = (a.Ho + TH - 1) / TH;
    int bid = blockIdx.x;
    const int tw_i = bid % tiles_w; bid /= tiles_w;
    const int th_i = bid % tiles_h;
    const int to = bid / tiles_h;
    const int h0 = th_i * TH, w0 = tw_i * TW;
    const int n0 = blockIdx.y * 32 * NB;

    // halo rectangle in source coordinates
    int hs0, ws0, HR, WR;
    if (a.ups) {
        hs0 = (h0 - 1) >> 1; ws0 = (w0 - 1) >> 1;
        HR = ((h0 + TH) >> 1) - hs0 + 1; WR = ((w0 + TW) >> 1) - ws0 + 1;
    } else {
        hs0 = h0 * a.s - a.ph; ws0 = w0 * a.s - a.ph;
        HR = (TH - 1) * a.s + a.KH; WR = (TW - 1) * a.s + a.KW;
    }
    float* wt = lds + HR * WR * LDV;             // 2 x [32*NB][LDV]

    // this lane's output voxel inside the tile (A-operand row) and its halo index per tap
    const int iv = wave * 32 + (lane & 31);
    const int dh = iv >> tw_shift, dw = iv & (TW - 1);
    const int kq = lane >> 5;                    // which half of the k pairs this lane feeds

    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[nb][j] = 0.f;

    const int nchunk = (a.Cin + 31) >> 5;
    const long frame_elems = (long)a.H * a.W * a.Cin;
    // Halo staging map.  Thread tid stages float4 (tid & 7) of voxels (tid >> 3) + 32 * it; which source pixel
    // that is does not depend on (kt, chunk), so the division by the halo width is done once per workgroup.
    // hpix: source pixel hs * W + ws; -1 outside the frame (zeros are staged); -2 beyond the halo (nothing is).
    constexpr int HIT = 6;                       // covers halos of up to 192 voxels (stride-1 3x3: 10 x 18)
    const int nvox = HR * WR;
    const bool small_halo = nvox <= HIT * 32;
    const int hv0 = tid >> 3, hj = tid & 7;
    int hpix[HIT];
#pragma unroll
    for (int it = 0; it < HIT; ++it) {
        const int v = hv0 + 32 * it;
        const int hr = v / WR, wc_ = v - hr * WR;
        const int hs = hs0 + hr, ws = ws0 + wc_;
        hpix[it] = v >= nvox ? -2 : (hs >= 0 && hs < a.H && ws >= 0 && ws < a.W) ? hs * a.W + ws : -1;
    }
    const bool pre_split = SPLIT && a.ws != nullptr;
    int wrow[NB];           // element offsets: the weight tensors hold far fewer than 2^31 values
    bool wok[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int co = n0 + hv0 + 32 * i;
        wok[i] = co < a.Cout;
        wrow[i] = pre_split ? co * nchunk * 64 + 8 * hj : co * a.Cin + 4 * hj;
    }
    for (int kt = 0; kt < a.KT; ++kt) {
        const int ts = to * a.st + kt - a.pt;
        const float* fp; const float* rp;
        if (ts >= 0) { fp = a.x + ts * frame_elems; rp = a.rn ? a.rn + (long)ts * a.H * a.W : nullptr; }
        else if (ts == -1) { fp = a.prev1; rp = a.rn_prev1; }
        else { fp = a.prev2; rp = a.rn_prev2; }
        if (fp == nullptr) continue;             // zero padding contributes nothing
        for (int cc = 0; cc < nchunk; ++cc) {
            __syncthreads();                      // previous users of halo / wt are done
            if (small_halo) {
                const int ci = cc * 32 + 4 * hj;
                const bool cok = ci < a.Cin;
                float4 g = make_float4(1.f, 1.f, 1.f, 1.f);
                if (a.gamma && cok) g = *reinterpret_cast<const float4*>(a.gamma + ci);
                // two batches of three voxels: the loads of a batch are all issued before the first is used, and
                // only 15 registers hold staged values (a sixth wave per CU is worth more than deeper batches)
#pragma unroll
                for (int b0 = 0; b0 < HIT; b0 += 3) {
                    float4 val[3];
                    float rn[3];
#pragma unroll
                    for (int u = 0; u < 3; ++u) {
                        val[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                        rn[u] = 0.f;
                        if (hpix[b0 + u] >= 0 && cok) {
                            val[u] = *reinterpret_cast<const float4*>(fp + (long)hpix[b0 + u] * a.Cin + ci);
                            if (a.gamma) rn[u] = rp[hpix[b0 + u]];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 3; ++u) {
                        const int it = b0 + u;
                        if (hpix[it] == -2) continue;
                        float4 x = val[u];
                        if (hpix[it] >= 0 && cok) {
                            if (a.gamma) {
                                const float r = rn[u] * a.norm_scale;
                                x.x = x.x * r * g.x; x.y = x.y * r * g.y;
                                x.z = x.z * r * g.z; x.w = x.w * r * g.w;
                            }
                            if (a.silu) { x.x = silu_f(x.x); x.y = silu_f(x.y); x.z = silu_f(x.z); x.w = silu_f(x.w); }
                        }
                        float* dst = halo + (hv0 + 32 * it) * LDV;
                        if (SPLIT) store_split4(reinterpret_cast<char*>(dst) + 8 * hj, x);
                        else *reinterpret_cast<float4*>(dst + 4 * hj) = x;
                    }
                }
            } else
            for (int idx = tid; idx < HR * WR * 8; idx += 256) {
                const int v = idx >> 3, j = idx & 7;
                const int hs = hs0 + v / WR, ws = ws0 + v % WR;
                const int ci = cc * 32 + 4 * j;
                float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
                if (hs >= 0 && hs < a.H && ws >= 0 && ws < a.W && ci < a.Cin) {
                    val = *reinterpret_cast<const float4*>(fp + ((long)hs * a.W + ws) * a.Cin + ci);
                    if (a.gamma) {
                        const float r = rp[(long)hs * a.W + ws] * a.norm_scale;
                        const float4 g = *reinterpret_cast<const float4*>(a.gamma + ci);
                        val.x = val.x * r * g.x; val.y = val.y * r * g.y;
                        val.z = val.z * r * g.z; val.w = val.w * r * g.w;
                    }
                    if (a.silu) {       // x * sigmoid(x) from one v_exp_f32 and one v_rcp_f32 (~2 ulp)
                        val.x = silu_f(val.x); val.y = silu_f(val.y); val.z = silu_f(val.z); val.w = silu_f(val.w);
                    }
                }
                if (SPLIT) store_split4(reinterpret_cast<char*>(halo + v * LDV) + 8 * j, val);
                else *reinterpret_cast<float4*>(halo + v * LDV + 4 * j) = val;
            }
            const int ksteps = (min(32, a.Cin - cc * 32) + 7) >> 3;
            // Weight slices [32*NB x 32] of the taps alternate between two LDS buffers: the slice of
            // tap i + 1 is fetched into registers before the MFMAs of tap i and written to the other
            // buffer after them, so a tap costs one barrier and its global loads fly behind the MFMAs.
            // Each thread moves NB float4s per tap (32*NB rows x 8 float4 / 256 threads).
            const int ntaps = a.KH * a.KW;
            // With pre-split weights (a.ws: [tap][co][chunk][32 hi | 32 lo] bf16) a float4 is 16 raw
            // bytes of that 128-byte row and goes to LDS as is; otherwise it is four fp32 weights,
            // split while being written.
            // thread tid moves float4 (tid & 7) of weight rows (tid >> 3) + 32 * i: the per-thread part of the
            // address is wrow[i] (set up once), the (kt, tap, chunk) part is the same for the whole workgroup
            auto w_ptr = [&](int tap_hw, int i) -> const float* {
                if (!wok[i]) return nullptr;
                const long slice = (long)(kt * ntaps + tap_hw) * a.Cout;
                if (pre_split) return reinterpret_cast<const float*>(a.ws + wrow[i] + (slice * nchunk + cc) * 64);
                if (cc * 32 + 4 * hj >= a.Cin) return nullptr;
                return a.w + wrow[i] + slice * a.Cin + cc * 32;
            };
            auto w_put = [&](float* buf, int i, const float4& val) {
                const int idx = tid + i * 256, r = idx >> 3, j = idx & 7;
                if (pre_split) *reinterpret_cast<float4*>(reinterpret_cast<char*>(buf + r * LDV) + 16 * j) = val;
                else if (SPLIT) store_split4(reinterpret_cast<char*>(buf + r * LDV) + 8 * j, val);
                else *reinterpret_cast<float4*>(buf + r * LDV + 4 * j) = val;
            };
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const float* p = w_ptr(0, i);
                w_put(wt, i, p ? *reinterpret_cast<const float4*>(p) : make_float4(0.f, 0.f, 0.f, 0.f));
            }
            __syncthreads();                      // halo and the first slice are in place
            for (int kh = 0; kh < a.KH; ++kh)
                for (int kw = 0; kw < a.KW; ++kw) {
                    const int tap_hw = kh * a.KW + kw;
                    const float* wcur = wt + (tap_hw & 1) * (32 * NB * LDV);
                    float* wnext = wt + ((tap_hw + 1) & 1) * (32 * NB * LDV);
                    const bool has_next = tap_hw + 1 < ntaps;
                    float4 pre[NB];
                    if (has_next) {
#pragma unroll
                        for (int i = 0; i < NB; ++i) {
                            const float* p = w_ptr(tap_hw + 1, i);
                            pre[i] = p ? *reinterpret_cast<const float4*>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
                        }
                    }
                    int hsrc, wsrc;
                    if (a.ups) { hsrc = ((h0 + dh + kh - 1) >> 1) - hs0; wsrc = ((w0 + dw + kw - 1) >> 1) - ws0; }
                    else { hsrc = dh * a.s + kh; wsrc = dw * a.s + kw; }
                    const float* ap = halo + (hsrc * WR + wsrc) * LDV + 4 * kq;
                    const float* bp = wcur + (lane & 31) * LDV + 4 * kq;
                    if (SPLIT) {
                        // lane (row/col = lane & 31, half kq) feeds k = 16*ks + 8*kq .. +7: 16 bytes
                        // at +32*ks + 16*kq of the hi plane, the lo plane 64 bytes further
                        const char* ac = reinterpret_cast<const char*>(ap);
                        const char* bc = reinterpret_cast<const char*>(bp);
                        // both 16-wide k-steps always: a chunk's rows beyond Cin are staged as zeros, and a loop
                        // with a run-time trip count here makes hipcc move all accumulators out of and back into
                        // the AGPR file around it, once per tap (140 register moves beside 18 MFMAs)
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) {
                            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(ac + 32 * ks);
                            const bf16x8 al = *reinterpret_cast<const bf16x8*>(ac + 64 + 32 * ks);
#pragma unroll
                            for (int nb = 0; nb < NB; ++nb) {
                                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(bc + nb * 32 * LDV * 4 + 32 * ks);
                                const bf16x8 bl = *reinterpret_cast<const bf16x8*>(bc + nb * 32 * LDV * 4 + 64 + 32 * ks);
                                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[nb], 0, 0, 0);
                                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[nb], 0, 0, 0);
                                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[nb], 0, 0, 0);
                            }
                        }
                    } else
                    for (int k8 = 0; k8 < ksteps; ++k8) {
                        const float4 av = *reinterpret_cast<const float4*>(ap + 8 * k8);
                        const float avv[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) {
                            const float4 bv = *reinterpret_cast<const float4*>(bp + nb * 32 * LDV + 8 * k8);
                            const float bvv[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(avv[q], bvv[q], acc[nb], 0, 0, 0);
                        }
                    }
                    // hipcc (ROCm 7.2) copies the accumulators out of the AGPR file at this loop
                    // exit and pads the MFMA -> accvgpr_read hazard only for some of the 16
                    // registers across the branch (the last-written register came back stale with
                    // NB == 1).  32 wait states cover the 16-pass MFMA; < 3 % of a tap's MFMA time.
                    MFMA_DRAIN();
                    if (has_next) {
#pragma unroll
                        for (int i = 0; i < NB; ++i) w_put(wnext, i, pre[i]);
                        __syncthreads();          // next slice visible; this one free for the tap after it
                    }
                }
        }
    }

    // The last MFMA of the loop nest retires its 16 result registers over 64 cycles; with a
    // single accumulator (NB == 1) nothing separates it from the reads below and hipcc's
    // hazard padding does not reach across the nested-loop exit (the last register came back
    // stale).  Pad explicitly.
    MFMA_DRAIN();
    // epilogue: D row = voxel (reg&3)+8*(reg>>2)+4*kq of this wave's 32, column = lane&31 = channel.  The residual
    // values of a 32-channel block are all fetched before its first store (out and resid may be one buffer as far as
    // the compiler knows, so a load placed after a store would wait for it).
    const int C_out_row = a.til ? a.Cout / 2 : a.Cout;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int co = n0 + nb * 32 + (lane & 31);
        const bool cok = co < a.Cout;
        const float bv = (a.bias && cok) ? a.bias[co] : 0.f;
        int frame = to, cch = co;
        if (a.til) { frame = 2 * to + co / C_out_row; cch = co % C_out_row; }
        const long fbase = (long)frame * a.Ho * a.Wo * C_out_row + cch;      // (a frame's offsets fit 32 bits: host check)
#pragma unroll
        for (int half = 0; half < 2; ++half) {      // eight registers at a time: the loads in flight stay within the register budget
            int off[8];
            float rv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int reg = half * 8 + i;
                const int ov = wave * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kq;
                const int ho = h0 + (ov >> tw_shift), wo = w0 + (ov & (TW - 1));
                off[i] = (cok && ho < a.Ho && wo < a.Wo) ? (ho * a.Wo + wo) * C_out_row : -1;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) rv[i] = a.resid ? a.resid[fbase + max(off[i], 0)] : 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (off[i] >= 0) a.out[fbase + off[i]] = acc[nb][half * 8 + i] + bv + rv[i];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Producer / consumer form of the 3x3 stride-1 convolutions (split-bf16 products, pre-split weights): the same
// products and the same accumulation order per output element as conv3d_cl_kernel<NB, true> (x_lo w_hi, x_hi w_lo,
// x_hi w_hi per 32-channel chunk, chunks and taps in the same order), with the two halves of its work given to
// different waves so that they overlap instead of alternating.
//   * 8 consumer waves own two patch rows (2 x 16 voxels) each of a 16 x 16 patch (256 voxels x 32*NB channels per
//     workgroup): per tap they read their operand fragments from LDS and issue the NB*12 MFMAs - nothing else.
//   * 4 producer waves do what conv3d_cl_kernel does between its barriers.  Two of them fetch the halo patch of the
//     NEXT (time tap, 32-channel chunk) group - from the pre-split image (PRE: pure copies), or from the fp32
//     activation tensor, applying voxel norm / gamma / SiLU and the split into bf16 hi | lo - and write it to the
//     other halo buffer, a few voxels per tap; the other two copy the weight slice of the tap after next into a ring of
//     three slots.  Every global load is issued three taps before its value is written to LDS, so a producer does not
//     wait for memory inside a tap; the roles are separate waves because a wave's loads return in order as far as
//     its counter can tell, and the L2-resident weight slices would otherwise queue behind the halo's HBM latency.
// One barrier per tap joins the two sides: after barrier j the slot of tap j and (first tap of a group) the
// group's halo are complete, and the slot of tap j - 1 and the previous group's halo buffer are free.
// A group whose frame is temporal zero padding is skipped (as in conv3d_cl_kernel).
// timing-only ablations of the producers (tools/build_variant.sh ... -DWS_ABL_NO_HALO=1; results are wrong when set)
#ifndef WS_ABL_NO_WEIGHTS
#define WS_ABL_NO_WEIGHTS 0
#endif
#ifndef WS_ABL_NO_HALO
#define WS_ABL_NO_HALO 0
#endif
constexpr int WS_T = 16;              // patch edge
constexpr int WS_CONS = 8, WS_PROD = 4;
constexpr int WS_LDV = 40;            // floats per staged row: 32 (bf16 hi | lo of 32 channels) + 8 pad; at 160 bytes
                                      // the 16-row x four-k-block reads of a 16x16x32 fragment fall on 64 distinct banks
constexpr int WS_HIT = 11;            // halo batches of 32 voxels per producer pass: up to 352 voxels (18 x 18 = 324)

// Barrier between the taps: this wave's LDS traffic has retired, then s_barrier.  (A __syncthreads() also waits
// for every global load the wave has in flight - the producers' fetches for later taps - and would put the
// memory latency back into every tap.)
__device__ __forceinline__ void ws_tap_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// 16 bytes through a buffer descriptor: wave-uniform base and scalar offset, 32-bit per-lane offset (no 64-bit address
// arithmetic per load); an offset beyond the descriptor's size reads zeros.
typedef unsigned int ws_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ws_buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const ws_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ws_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

template <int NB, bool PRE>
__global__ void __launch_bounds__(768) conv3d_ws_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_w = (a.Wo + WS_T - 1) / WS_T, tiles_h = (a.Ho + WS_T - 1) / WS_T;
    int bid = blockIdx.x;
    const int tw_i = bid % tiles_w; bid /= tiles_w;
    const int th_i = bid % tiles_h;
    const int to = bid / tiles_h;
    const int h0 = th_i * WS_T, w0 = tw_i * WS_T;
    const int n0 = blockIdx.y * 32 * NB;
    int hs0, ws0, HR, WR;
    if (a.ups) {
        hs0 = (h0 - 1) >> 1; ws0 = (w0 - 1) >> 1;
        HR = ((h0 + WS_T) >> 1) - hs0 + 1; WR = ((w0 + WS_T) >> 1) - ws0 + 1;
    } else {
        hs0 = h0 - a.ph; ws0 = w0 - a.ph;
        HR = WS_T - 1 + a.KH; WR = WS_T - 1 + a.KW;
    }
    const int nvox = HR * WR;
    const int halo_stride = WS_HIT * 32 * WS_LDV;                      // floats per halo buffer (every producer batch has its rows)
    float* const wt = lds + 2 * halo_stride;                        // 3 x [32*NB][WS_LDV]
    constexpr int WSLOT = 32 * NB * WS_LDV;
    const int nchunk = (a.Cin + 31) >> 5;
    const long frame_elems = (long)a.H * a.W * a.Cin;
    const long frame_vox = (long)a.H * a.W;

    // The time taps whose frame exists: frames before the first of the video are zero padding and contribute
    // nothing.  Source frames ascend with kt and a missing one is always the earliest, so the taps that exist are
    // the last `nact` of the KT.
    // PRE: the input is the pre-split image; fp then points at its 128-byte voxel-chunk rows (as float*)
    auto frame = [&](int kt, const float*& fp, const float*& rp) __attribute__((always_inline)) {
        const int ts = to * a.st + kt - a.pt;
        if (PRE) {
            rp = nullptr;
            if (ts >= 0) fp = reinterpret_cast<const float*>(a.xs + ts * frame_vox * nchunk * 64);
            else fp = reinterpret_cast<const float*>(ts == -1 ? a.prev1s : a.prev2s);
            return;
        }
        if (ts >= 0) { fp = a.x + ts * frame_elems; rp = a.rn ? a.rn + ts * frame_vox : nullptr; }
        else if (ts == -1) { fp = a.prev1; rp = a.rn_prev1; }
        else { fp = a.prev2; rp = a.rn_prev2; }
    };
    int nact = 0;
    for (int kt = 0; kt < a.KT; ++kt) {
        const float* fp; const float* rp;
        frame(kt, fp, rp);
        nact += fp != nullptr ? 1 : 0;
    }
    const int kt_first = a.KT - nact;
    const int G = nact * nchunk;                  // groups; 9 taps each
    const int ntaps = 9;

    if (wave >= WS_CONS) {
        // ------------------------------------------------------------------ producers
        // Two roles, two waves each: waves 8-9 copy weight slices, waves 10-11 stage the halo.  A wave's loads return
        // in order as far as its counter can tell, so a wave that fetched both would hold every (L2-resident) weight
        // slice behind the HBM latency of the halo voxels fetched before it.
        // Every load below is unconditional and every address valid (indices clamped, invalid values replaced
        // by zero after the load, never by a branch around it): with straight-line code the compiler can count
        // the loads in flight and waits only for the one a store needs (a load inside a branch makes it wait for
        // all of them, which would put the memory latency back into every tap).
        const int rtid = tid - 64 * WS_CONS - (wave >= WS_CONS + 2 ? 128 : 0);      // 0..127 inside the role
        const int hv0 = rtid >> 3, hj = rtid & 7;
        // group g -> (frame, voxel norms, time tap, chunk); evaluated once per group, not per use
        struct Grp { const float* fp; const float* rp; int kt, cc, ci; bool cok; };
        auto grp = [&](int g) __attribute__((always_inline)) {
            Grp r;
            const int ga = g / nchunk;
            r.cc = g - ga * nchunk;
            r.kt = kt_first + ga;
            frame(r.kt, r.fp, r.rp);
            if (r.rp == nullptr) r.rp = r.fp;                   // no voxel norms: any readable address
            r.cok = r.cc * 32 + 4 * hj < a.Cin;
            r.ci = min(r.cc * 32 + 4 * hj, a.Cin - 4);
            return r;
        };
        if (wave < WS_CONS + 2) {
            // ---- weight slices: 32*NB rows x 8 pieces of 16 bytes over 128 lanes = 2*NB pieces per lane.  The slice
            // of tap j + 5 is fetched at tap j into one of three register sets and written at tap j + 3 into the slot
            // tap j + 2 has left.  (A register set is a struct of named members returned by value: indexed arrays of
            // float4 ended up in scratch memory here, and scratch accesses share the counter of the global loads.)
            // per-lane BYTE offsets as 32-bit values; the slice's offset travels in a scalar register
            unsigned wrow[2 * NB];  // rows beyond Cout repeat the last one: their products land in columns that are not stored
#pragma unroll
            for (int i = 0; i < 2 * NB; ++i) wrow[i] = (unsigned)(min(n0 + hv0 + 16 * i, a.Cout - 1) * nchunk * 64 + 8 * hj) * 2u;
            const __amdgpu_buffer_rsrc_t wres = ws_rsrc(a.ws, (unsigned)(a.KT * ntaps * a.Cout * nchunk) * 128u);
            struct WSet { float4 v0, v1, v2, v3, v4, v5; };
            auto w_load = [&](const Grp& q, int tap) __attribute__((always_inline)) {
                WSet v;
                const unsigned so = (unsigned)((q.kt * ntaps + tap) * a.Cout * nchunk + q.cc) * 128u;
                v.v0 = ws_buf_load(wres, wrow[0], so);
                v.v1 = ws_buf_load(wres, wrow[1], so);
                v.v2 = v.v3 = v.v4 = v.v5 = v.v0;
                if (NB > 1) {
                    v.v2 = ws_buf_load(wres, wrow[NB > 1 ? 2 : 0], so);
                    v.v3 = ws_buf_load(wres, wrow[NB > 1 ? 3 : 0], so);
                }
                if (NB > 2) {
                    v.v4 = ws_buf_load(wres, wrow[NB > 2 ? 4 : 0], so);
                    v.v5 = ws_buf_load(wres, wrow[NB > 2 ? 5 : 0], so);
                }
                return v;
            };
            auto w_store = [&](int slot, const WSet v) __attribute__((always_inline)) {
                char* dst = reinterpret_cast<char*>(wt + slot * WSLOT + hv0 * WS_LDV) + 16 * hj;
                *reinterpret_cast<float4*>(dst) = v.v0;
                *reinterpret_cast<float4*>(dst + 16 * WS_LDV * 4) = v.v1;
                if (NB > 1) {
                    *reinterpret_cast<float4*>(dst + 32 * WS_LDV * 4) = v.v2;
                    *reinterpret_cast<float4*>(dst + 48 * WS_LDV * 4) = v.v3;
                }
                if (NB > 2) {
                    *reinterpret_cast<float4*>(dst + 64 * WS_LDV * 4) = v.v4;
                    *reinterpret_cast<float4*>(dst + 80 * WS_LDV * 4) = v.v5;
                }
            };
            Grp cur = grp(0);
            WSet wreg0 = w_load(cur, 0), wreg1 = w_load(cur, 1), wreg2;
            w_store(0, wreg0);
            w_store(1, wreg1);
            wreg2 = w_load(cur, 2);
            wreg0 = w_load(cur, 3);
            wreg1 = w_load(cur, 4);
            for (int g = 0; g < G; ++g) {
                const Grp nxt = grp(min(g + 1, G - 1));       // the last group fetches its own slices again (never read)
#define WS_WEIGHT_TAP(T, WSET)                                                                             \
                {                                                                                          \
                    ws_tap_barrier();                                                                      \
                    if (!WS_ABL_NO_WEIGHTS) {                                                              \
                        w_store(((T) + 2) % 3, WSET);                                                      \
                        if ((T) + 5 < 9) WSET = w_load(cur, (T) + 5);                                      \
                        else WSET = w_load(nxt, (T) + 5 - 9);                                              \
                    }                                                                                      \
                }
                WS_WEIGHT_TAP(0, wreg2) WS_WEIGHT_TAP(1, wreg0) WS_WEIGHT_TAP(2, wreg1)
                WS_WEIGHT_TAP(3, wreg2) WS_WEIGHT_TAP(4, wreg0) WS_WEIGHT_TAP(5, wreg1)
                WS_WEIGHT_TAP(6, wreg2) WS_WEIGHT_TAP(7, wreg0) WS_WEIGHT_TAP(8, wreg1)
#undef WS_WEIGHT_TAP
                cur = nxt;
            }
            return;
        }
        // ---- halo: batches of 16 voxels (8 lanes per voxel row of 128 bytes); the next group's halo is fetched four
        // batches per tap in taps 0..5 and each batch is written three taps after its fetch (twelve in registers).
        constexpr int HB = 2 * WS_HIT;      // batches of 16 voxels
        int hpix[HB];                       // source pixel of halo voxel hv0 + 16 * it; -1: outside the frame (zeros)
        unsigned hoff[PRE ? HB : 1];        // PRE: byte offset of this lane's 16 bytes inside the frame's chunk-0 image
#pragma unroll
        for (int it = 0; it < HB; ++it) {
            const int v = min(hv0 + 16 * it, nvox - 1);         // rows beyond the patch repeat its last voxel (never read)
            const int hr = v / WR, wc_ = v - hr * WR;
            const int hs = hs0 + hr, ws = ws0 + wc_;
            hpix[it] = (hs >= 0 && hs < a.H && ws >= 0 && ws < a.W) ? hs * a.W + ws : -1;
            // outside the frame: an offset beyond the descriptor, which reads zeros
            if (PRE) hoff[it] = hpix[it] >= 0 ? (unsigned)hpix[it] * (unsigned)(nchunk * 128) + 16u * hj : 0x80000000u;
        }
        const unsigned frame_bytes = (unsigned)(a.H * a.W * nchunk) * 128u;
        auto h_load = [&](const Grp& q, int it, float4& val, float& rn) __attribute__((always_inline)) {
            if (PRE) {      // 16 raw bytes of the voxel's 128-byte row of this chunk
                val = ws_buf_load(ws_rsrc(q.fp, frame_bytes), hoff[it], (unsigned)q.cc * 128u);
                rn = 0.f;
                return;
            }
            const int pix = max(hpix[it], 0);
            val = *reinterpret_cast<const float4*>(q.fp + (long)pix * a.Cin + q.ci);
            rn = q.rp[pix];
        };
        auto g_load = [&](const Grp& q) __attribute__((always_inline)) {
            if (PRE) return make_float4(1.f, 1.f, 1.f, 1.f);
            return *reinterpret_cast<const float4*>((a.gamma ? a.gamma : q.fp) + q.ci);
        };
        const bool has_norm = a.gamma != nullptr, has_silu = a.silu != 0;
        auto h_store = [&](const Grp& q, int buf, int it, float4 x, float rn, const float4& gm) __attribute__((always_inline)) {
            float* dst = lds + buf * halo_stride + (hv0 + 16 * it) * WS_LDV;
            if (PRE) {      // (out-of-frame voxels arrive as zeros: the descriptor's range check)
                *reinterpret_cast<float4*>(reinterpret_cast<char*>(dst) + 16 * hj) = x;
                return;
            }
            if (has_norm) {
                const float r = rn * a.norm_scale;
                x.x = x.x * r * gm.x; x.y = x.y * r * gm.y; x.z = x.z * r * gm.z; x.w = x.w * r * gm.w;
            }
            if (has_silu) { x.x = silu_f(x.x); x.y = silu_f(x.y); x.z = silu_f(x.z); x.w = silu_f(x.w); }
            if (!(hpix[it] >= 0 && q.cok)) x = make_float4(0.f, 0.f, 0.f, 0.f);
            store_split4(reinterpret_cast<char*>(dst) + 8 * hj, x);
        };
        constexpr int HRING = 12, HPT = 4;            // batches in registers / fetched per tap
        float4 hv[HRING];
        float hn[HRING];
        Grp cur = grp(0);
        {       // prologue: the first group's halo, fetched and written at once
            const float4 gm = g_load(cur);
#pragma unroll
            for (int i0 = 0; i0 < HB; i0 += HRING) {
#pragma unroll
                for (int it = i0; it < i0 + HRING && it < HB; ++it) h_load(cur, it, hv[it - i0], hn[it - i0]);
#pragma unroll
                for (int it = i0; it < i0 + HRING && it < HB; ++it) h_store(cur, 0, it, hv[it - i0], hn[it - i0], gm);
            }
        }
        for (int g = 0; g < G; ++g) {
            const Grp nxt = grp(min(g + 1, G - 1));           // the last group stages its own halo again (never read)
            const float4 gm = g_load(nxt);
            const int nbuf = (g + 1) & 1;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                ws_tap_barrier();
                if (!WS_ABL_NO_HALO && t >= 3) {       // batches fetched at tap t - 3
#pragma unroll
                    for (int it = HPT * (t - 3); it < HPT * (t - 3) + HPT && it < HB; ++it)
                        h_store(nxt, nbuf, it, hv[it % HRING], hn[it % HRING], gm);
                }
                if (!WS_ABL_NO_HALO && t <= 5) {
#pragma unroll
                    for (int it = HPT * t; it < HPT * t + HPT && it < HB; ++it) h_load(nxt, it, hv[it % HRING], hn[it % HRING]);
                }
            }
            cur = nxt;
        }
        return;
    }

    // ---------------------------------------------------------------------- consumers
    // v_mfma_f32_16x16x32_bf16 with the weights as the A operand (rows = 16 output channels) and the voxels as B
    // (columns = 16 voxels = one patch row): lane holds A[row lane % 16][k (lane / 16) * 8 .. + 7], B[k ...][column
    // lane % 16] and D[rows (lane / 16) * 4 .. + 3][column lane % 16], i.e. four consecutive output channels of one voxel.
    // (The same FLOPs issued as 32x32x16 instructions hold a shader clock of 1.91 GHz against 2.2 GHz for these: the
    // kernel runs at the power limit, and the narrow instruction costs less energy per product.)
    const int l16 = lane & 15, kb = lane >> 4;
    constexpr int NCB = 2 * NB;         // blocks of 16 output channels
    f32x4 acc[2][NCB];
#pragma unroll
    for (int vb = 0; vb < 2; ++vb)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) acc[vb][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // Fragments are read ahead of their use in two stages per tap, pinned by scheduling barriers (left alone, hipcc
    // re-uses one register set for successive weight fragments: read, wait, two MFMAs, read, wait ...).  While the
    // first NB channel blocks of tap t are multiplied, the second NB are in flight; while those are multiplied, the
    // voxels and the first NB blocks of tap t + 1 - whose weight slice and halo are complete at tap t's barrier.  The
    // first tap of a group reads its voxels behind the barrier (the halo's last rows are written in the tap before).
    struct XFrag { bf16x8 h[2], l[2]; };
    struct WFrag { bf16x8 h[NB], l[NB]; };
    auto read_x = [&](const float* halo, int t) __attribute__((always_inline)) {
        XFrag x;
        const int kh = t / 3, kw = t % 3;
#pragma unroll
        for (int vb = 0; vb < 2; ++vb) {        // voxel block vb: patch row 2 * wave + vb, columns 0 .. 15
            const int dh = 2 * wave + vb;
            int hsrc, wsrc;
            if (a.ups) { hsrc = ((h0 + dh + kh - 1) >> 1) - hs0; wsrc = ((w0 + l16 + kw - 1) >> 1) - ws0; }
            else { hsrc = dh + kh; wsrc = l16 + kw; }
            const char* xc = reinterpret_cast<const char*>(halo + (hsrc * WR + wsrc) * WS_LDV) + 16 * kb;
            x.h[vb] = *reinterpret_cast<const bf16x8*>(xc);
            x.l[vb] = *reinterpret_cast<const bf16x8*>(xc + 64);
        }
        return x;
    };
    const char* const wlane = reinterpret_cast<const char*>(wt + l16 * WS_LDV) + 16 * kb;
    auto read_w = [&](int slot, int cb0) __attribute__((always_inline)) {
        WFrag w;
        const char* wc = wlane + (slot * WSLOT + cb0 * 16 * WS_LDV) * 4;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            w.h[i] = *reinterpret_cast<const bf16x8*>(wc + i * 16 * WS_LDV * 4);
            w.l[i] = *reinterpret_cast<const bf16x8*>(wc + i * 16 * WS_LDV * 4 + 64);
        }
        return w;
    };
    auto mma = [&](int cb0, const WFrag& w, const XFrag& x) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            // per accumulator: x_lo * w_hi, x_hi * w_lo, x_hi * w_hi (small terms first)
            acc[0][cb0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.h[i], x.l[0], acc[0][cb0 + i], 0, 0, 0);
            acc[1][cb0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.h[i], x.l[1], acc[1][cb0 + i], 0, 0, 0);
            acc[0][cb0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.l[i], x.h[0], acc[0][cb0 + i], 0, 0, 0);
            acc[1][cb0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.l[i], x.h[1], acc[1][cb0 + i], 0, 0, 0);
            acc[0][cb0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.h[i], x.h[0], acc[0][cb0 + i], 0, 0, 0);
            acc[1][cb0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.h[i], x.h[1], acc[1][cb0 + i], 0, 0, 0);
        }
    };
    XFrag x0, x1;
    WFrag wa, wb;
    for (int g = 0; g < G; ++g) {
        const float* halo = lds + (g & 1) * halo_stride;
#define WS_CONSUMER_TAP(T, XCUR, XNEXT)                                                                   \
        {                                                                                                  \
            ws_tap_barrier();                                                                              \
            if ((T) == 0) {                                                                                \
                XCUR = read_x(halo, 0);                                                                    \
                if (g == 0) wa = read_w(0, 0);                                                             \
            }                                                                                              \
            wb = read_w((T) % 3, NB);                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                             \
            mma(0, wa, XCUR);                                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                             \
            if ((T) < 8) XNEXT = read_x(halo, (T) + 1);                                                    \
            wa = read_w(((T) + 1) % 3, 0);       /* tap 8: the next group's first slice (slot 0) */       \
            __builtin_amdgcn_sched_barrier(0);                                                             \
            mma(NB, wb, XCUR);                                                                             \
            __builtin_amdgcn_sched_barrier(0);                                                             \
        }
        WS_CONSUMER_TAP(0, x0, x1) WS_CONSUMER_TAP(1, x1, x0) WS_CONSUMER_TAP(2, x0, x1)
        WS_CONSUMER_TAP(3, x1, x0) WS_CONSUMER_TAP(4, x0, x1) WS_CONSUMER_TAP(5, x1, x0)
        WS_CONSUMER_TAP(6, x0, x1) WS_CONSUMER_TAP(7, x1, x0) WS_CONSUMER_TAP(8, x0, x1)
#undef WS_CONSUMER_TAP
    }
    MFMA_DRAIN();
    // Epilogue, one patch row (16 voxels) at a time: lane holds channels n0 + cb * 16 + kb * 4 .. + 3 of voxel l16.  The
    // residual values of a row are all requested before its first store (out and resid may be one buffer as far as
    // the compiler knows, so a load placed after a store would wait for it: one memory round trip per block).
    const bool vec = (a.Cout & 3) == 0;         // 16-byte accesses need whole groups of four channels
#pragma unroll
    for (int vb = 0; vb < 2; ++vb) {
        const int ho = h0 + 2 * wave + vb, wo = w0 + l16;
        const bool vok = ho < a.Ho && wo < a.Wo;
        const long obase = (((long)to * a.Ho + min(ho, a.Ho - 1)) * a.Wo + min(wo, a.Wo - 1)) * a.Cout;
        if (vec) {
            float4 rv[NCB];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const int co = min(n0 + cb * 16 + kb * 4, a.Cout - 4);
                rv[cb] = a.resid ? *reinterpret_cast<const float4*>(a.resid + obase + co) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const int co = n0 + cb * 16 + kb * 4;
                if (!vok || co >= a.Cout) continue;
                const float4 bv = a.bias ? *reinterpret_cast<const float4*>(a.bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
                float4 y;
                y.x = acc[vb][cb][0] + bv.x + rv[cb].x;
                y.y = acc[vb][cb][1] + bv.y + rv[cb].y;
                y.z = acc[vb][cb][2] + bv.z + rv[cb].z;
                y.w = acc[vb][cb][3] + bv.w + rv[cb].w;
                *reinterpret_cast<float4*>(a.out + obase + co) = y;
            }
        } else {
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int co = n0 + cb * 16 + kb * 4 + i;
                    if (!vok || co >= a.Cout) continue;
                    float y = acc[vb][cb][i] + (a.bias ? a.bias[co] : 0.f);
                    if (a.resid) y += a.resid[obase + co];
                    a.out[obase + co] = y;
                }
        }
    }
}

// x fp32 [nvox][C] -> the convolution's LDS image of its input, written once instead of by every workgroup that
// stages the voxel: out uint16 [nvox][C/32][32 hi | 32 lo], value = act(x) split into bf16 hi + bf16 lo, with
// act = RMS_norm (F.normalize over the channels * sqrt(C) * gamma; gamma null = off) followed by SiLU (silu != 0):
// RMS_norm.forward + nn.SiLU in front of the convolutions of ResidualBlock and of the heads (vae2_1.py:50-53,
// 193-197).  LPV lanes share a voxel (NI float4 each), as in voxel_rnorm_kernel.  C % 32 == 0.
template <int LPV, int NI>
__global__ void __launch_bounds__(256) act_split_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        int silu, uint16_t* __restrict__ out, long nvox, int C,
                                                        float norm_scale) {
    const int lane = threadIdx.x & 63, sub = lane % LPV;
    const long v = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * (64 / LPV) + lane / LPV;
    const bool live = v < nvox;
    const float4* p = reinterpret_cast<const float4*>(x + (live ? v : 0) * C);
    float4 val[NI], gm[NI];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        val[i] = p[sub + i * LPV];
        gm[i] = gamma ? reinterpret_cast<const float4*>(gamma)[sub + i * LPV] : make_float4(1.f, 1.f, 1.f, 1.f);
        ss += val[i].x * val[i].x + val[i].y * val[i].y + val[i].z * val[i].z + val[i].w * val[i].w;
    }
#pragma unroll
    for (int o = LPV / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float r = gamma ? (1.0f / fmaxf(sqrtf(ss), 1e-12f)) * norm_scale : 1.f;
    if (!live) return;
    const int nchunk = C >> 5;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        float4 y = val[i];
        if (gamma) { y.x = y.x * r * gm[i].x; y.y = y.y * r * gm[i].y; y.z = y.z * r * gm[i].z; y.w = y.w * r * gm[i].w; }
        if (silu) { y.x = silu_f(y.x); y.y = silu_f(y.y); y.z = silu_f(y.z); y.w = silu_f(y.w); }
        const int q = sub + i * LPV;                 // float4 index inside the voxel: chunk q / 8, piece q % 8
        store_split4(reinterpret_cast<char*>(out + (v * nchunk + (q >> 3)) * 64) + 8 * (q & 7), y);
    }
}

// 1 / max(||x_v||_2, 1e-12) per voxel (F.normalize's denominator, vae2_1.py:50-53).  LPV lanes share
// a voxel (LPV = the largest power of two dividing C/4, so every lane loads the same number of
// float4s: 8 lanes x 3 for C = 96, 16 x 3 for 192, 32 x 3 for 384) and a wave covers 64/LPV voxels.
template <int LPV>
__global__ void __launch_bounds__(256) voxel_rnorm_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                          long nvox, int C) {
    const int lane = threadIdx.x & 63, sub = lane % LPV;
    const long v = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * (64 / LPV) + lane / LPV;
    float s = 0.f;
    if (v < nvox) {
        const float4* p = reinterpret_cast<const float4*>(x + v * C);
        for (int i = sub; i < C / 4; i += LPV) {
            const float4 q = p[i];
            s += q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
        }
    }
#pragma unroll
    for (int o = LPV / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (v < nvox && sub == 0) out[v] = 1.0f / fmaxf(sqrtf(s), 1e-12f);
}

// fp32 weights [rows][Cin] -> [rows][ceil(Cin/32)][32 hi | 32 lo] bf16 (zero beyond Cin): the image the
// conv kernel puts into LDS, so that staging a weight slice is a copy
__global__ void split_weights_kernel(const float* __restrict__ w, uint16_t* __restrict__ out, long rows, int Cin,
                                     int nchunk) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * nchunk * 32) return;
    const int k = (int)(idx & 31);
    const long rc = idx >> 5;
    const int cc = (int)(rc % nchunk);
    const long r = rc / nchunk;
    const int ci = cc * 32 + k;
    uint32_t hi = 0, lo = 0;
    if (ci < Cin) split_bf16(w[r * Cin + ci], hi, lo);
    out[rc * 64 + k] = (uint16_t)hi;
    out[rc * 64 + 32 + k] = (uint16_t)lo;
}

// [C,T,H,W] -> [T,H,W,Cp] with out = (in + pre[c]) * mul[c] + post[c]; channels >= C are zero
__global__ void to_cl_kernel(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ pre,
                             const float* __restrict__ mul, const float* __restrict__ post, int C, int Cp,
                             long nvox) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nvox * Cp) return;
    const int c = (int)(idx % Cp);
    const long v = idx / Cp;
    float val = 0.f;
    if (c < C) {
        val = in[(long)c * nvox + v];
        if (pre) val += pre[c];
        if (mul) val *= mul[c];
        if (post) val += post[c];
    }
    out[idx] = val;
}

// [T,H,W,Cs] (first C channels) -> [C,T,H,W] with the same affine, optional clamp to [-1,1]
__global__ void from_cl_kernel(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ pre,
                               const float* __restrict__ mul, const float* __restrict__ post, int C, int Cs,
                               long nvox, int clamp) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nvox * C) return;
    const long v = idx % nvox;
    const int c = (int)(idx / nvox);
    float val = in[v * Cs + c];
    if (pre) val += pre[c];
    if (mul) val *= mul[c];
    if (post) val += post[c];
    if (clamp) val = fminf(fmaxf(val, -1.f), 1.f);
    out[idx] = val;
}

// in-place softmax(scale * row) over rows of length N (fp32), one workgroup per row
__global__ void __launch_bounds__(256) softmax_rows_kernel(float* __restrict__ s, int N, float scale) {
    __shared__ float red[8];
    float* row = s + (long)blockIdx.x * N;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float mx = -INFINITY;
    for (int i = tid; i < N; i += 256) mx = fmaxf(mx, row[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int i = tid; i < N; i += 256) {
        const float e = expf((row[i] - mx) * scale);
        row[i] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
    for (int i = tid; i < N; i += 256) row[i] *= inv;
}

}  // namespace

int ovd_impl_conv3d_cl(const float* x, const float* prev2, const float* prev1, const uint16_t* xs,
                       const uint16_t* prev2s, const uint16_t* prev1s, const float* rn, const float* rn_prev2,
                       const float* rn_prev1, const float* gamma, int silu, const float* w, const uint16_t* w_split,
                       const float* bias,
                       const float* resid, float* out, int T, int H, int W, int Cin, int Cout, int KT, int KH, int KW,
                       int stride_t, int stride_hw, int pad_t, int pad_hw, int upsample2, int time_interleave,
                       hipStream_t st) {
    OVD_REQUIRE(T > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "conv3d: empty tensor");
    OVD_REQUIRE(Cin % 4 == 0, "conv3d: Cin=%d must be a multiple of 4 (pad the channels)", Cin);
    OVD_REQUIRE((KT == 1 || KT == 3) && (KH == 1 || KH == 3) && KH == KW, "conv3d: kernel %dx%dx%d not built", KT, KH, KW);
    OVD_REQUIRE(stride_t >= 1 && stride_t <= 2 && stride_hw >= 1 && stride_hw <= 2, "conv3d: stride not built");
    OVD_REQUIRE(pad_t >= 0 && pad_t <= 2 && pad_hw >= 0 && pad_hw <= 1, "conv3d: padding not built");
    OVD_REQUIRE(!(upsample2 && (stride_hw != 1 || KH != 3 || pad_hw != 1)), "conv3d: upsample needs a 3x3 stride-1 pad-1 conv");
    OVD_REQUIRE(!(time_interleave && Cout % 2), "conv3d: time interleave needs an even Cout");
    OVD_REQUIRE((gamma == nullptr) == (rn == nullptr), "conv3d: gamma and the voxel norms come together");
    const int Hin = upsample2 ? 2 * H : H, Win = upsample2 ? 2 * W : W;
    const int Ho = (Hin + 2 * pad_hw - KH + (stride_hw == 2 ? 1 : 0)) / stride_hw + 1 - (stride_hw == 2 ? 0 : 0);
    const int Wo = (Win + 2 * pad_hw - KW + (stride_hw == 2 ? 1 : 0)) / stride_hw + 1;
    // temporal extent: `pad_t` frames of left context (previous-chunk frames or zeros)
    const int To = (T + pad_t - KT) / stride_t + 1;
    OVD_REQUIRE(To > 0 && Ho > 0 && Wo > 0, "conv3d: output would be empty");
    ConvArgs a;
    a.xs = xs; a.prev2s = prev2s; a.prev1s = prev1s;
    a.x = x; a.prev2 = prev2; a.prev1 = prev1; a.rn = rn; a.rn_prev2 = rn_prev2; a.rn_prev1 = rn_prev1;
    a.gamma = gamma; a.w = w; a.ws = w_split; a.bias = bias; a.resid = resid; a.out = out;
    a.T = T; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.To = To; a.Ho = Ho; a.Wo = Wo;
    a.KT = KT; a.KH = KH; a.KW = KW; a.st = stride_t; a.s = stride_hw; a.pt = pad_t; a.ph = pad_hw;
    a.ups = upsample2; a.til = time_interleave; a.silu = silu;
    a.TW = (KH == 1) ? 128 : 16;
    a.norm_scale = sqrtf((float)Cin);
    if (KH == 1) {      // 1x1: voxels are independent, flatten H*W into one axis of 128-voxel tiles
        a.W = H * W; a.H = 1; a.Wo = Ho * Wo; a.Ho = 1;
    }
    const int TW = a.TW, TH = 128 / TW;
    const int tiles = ((a.Wo + TW - 1) / TW) * ((a.Ho + TH - 1) / TH) * To;
    int HR, WR;
    if (upsample2) { HR = TH / 2 + 2; WR = TW / 2 + 2; }
    else { HR = (TH - 1) * stride_hw + KH; WR = (TW - 1) * stride_hw + KW; }
    OVD_REQUIRE(HR * WR <= MAX_HALO, "conv3d: halo too large");
    OVD_REQUIRE((long)Ho * Wo * Cout < (1L << 31), "conv3d: an output frame must stay below 2^31 elements");
    // 32*nb output channels per workgroup: least padding; among equals 96 channels first (measured on
    // the decoder's 384-channel layers: 215 TFLOP/s against 180 with 128 and 181 with 64)
    int nb = 1;
    {
        int best = 1 << 30;
        const int cand[4] = {3, 4, 2, 1};
        for (int i = 0; i < 4; ++i) {
            const int waste = ceil_div(Cout, 32 * cand[i]) * 32 * cand[i] - Cout;
            if (waste < best) { best = waste; nb = cand[i]; }
        }
    }
    // few voxels (the latent-resolution layers: one 60 x 104 frame per chunk): 32-channel workgroups
    // instead, so that the grid covers the chip at least twice (same arithmetic per output element)
    const int cus = ovd_device_cus();
    if (nb > 1 && (long)tiles * ceil_div(Cout, 32 * nb) < 2L * cus) nb = 1;
    {   // OVD_VAE_NB=1|2|3|4 forces the width where it divides Cout (tuning aid)
        static std::atomic<int> forced_cache{INT32_MIN};
        const int forced = ovd_env_int(forced_cache, "OVD_VAE_NB", 0);
        if (forced >= 1 && forced <= 4 && Cout % (32 * forced) == 0) nb = forced;
    }
    const size_t lds = (size_t)(HR * WR + 2 * 32 * nb) * LDV * sizeof(float);
    dim3 grid(tiles, ceil_div(Cout, 32 * nb)), blk(256);
    // split-bf16 products by default; OVD_VAE_EXACT_FP32=1 selects the exact fp32 matrix path
    static std::atomic<int> exact_cache{INT32_MIN};
    const int exact = ovd_env_int(exact_cache, "OVD_VAE_EXACT_FP32", 0) != 0 ? 1 : 0;
#define CONV_CASE(NBV, SPL)                                                                                    \
    case NBV: {                                                                                                \
        static OvdLdsLimit lim;                                                                                \
        if (!lim.ensure(reinterpret_cast<const void*>(&conv3d_cl_kernel<NBV, SPL>), lds))                      \
            OVD_FAIL("conv3d: cannot reserve %zu bytes of LDS", lds);                                          \
        hipLaunchKernelGGL((conv3d_cl_kernel<NBV, SPL>), grid, blk, lds, st, a);                               \
    } break;
    OVD_REQUIRE(w != nullptr || (w_split != nullptr && !exact),
                "conv3d: the exact-fp32 path (OVD_VAE_EXACT_FP32) needs the fp32 weights");
    // 3x3 stride-1 convolutions on pre-split weights: the producer / consumer kernel (OVD_VAE_WS=0 disables)
    static std::atomic<int> ws_cache{INT32_MIN};
    const bool pre = xs != nullptr;
    // (the kernel addresses the split weights, and a pre-split frame, with 32-bit byte offsets)
    const long nchunk_l = (Cin + 31) / 32;
    const bool ws_ok = !exact && w_split != nullptr && KH == 3 && stride_hw == 1 && stride_t == 1 && !time_interleave &&
                       (long)KT * 9 * Cout * nchunk_l * 128 < (1L << 31);
    OVD_REQUIRE(xs == nullptr || (long)H * W * nchunk_l * 128 < (1L << 31), "conv3d: a pre-split frame must stay below 2 GiB");
    OVD_REQUIRE(!pre || (ws_ok && Cin % 32 == 0),
                "conv3d: a pre-split input needs split weights, a 3x3 stride-1 kernel, Cin %% 32 == 0 and the split-bf16 path");
    if (ws_ok && (pre || ovd_env_int(ws_cache, "OVD_VAE_WS", 1) != 0)) {
        const int t16 = ((a.Wo + WS_T - 1) / WS_T) * ((a.Ho + WS_T - 1) / WS_T) * To;
        int nbw = nb == 4 ? 2 : nb;          // (128 channels per workgroup would spill the producers' registers)
        if (nbw > 1 && (long)t16 * ceil_div(Cout, 32 * nbw) < (long)cus) nbw = 1;
        const size_t lds_ws = (size_t)(2 * WS_HIT * 32 + 3 * 32 * nbw) * WS_LDV * sizeof(float);
        dim3 gridw(t16, ceil_div(Cout, 32 * nbw)), blkw(64 * (WS_CONS + WS_PROD));
#define WS_CASE(NBV, PREV)                                                                                   \
    case NBV: {                                                                                              \
        static OvdLdsLimit lim;                                                                              \
        if (!lim.ensure(reinterpret_cast<const void*>(&conv3d_ws_kernel<NBV, PREV>), lds_ws))                \
            OVD_FAIL("conv3d: cannot reserve %zu bytes of LDS", lds_ws);                                     \
        hipLaunchKernelGGL((conv3d_ws_kernel<NBV, PREV>), gridw, blkw, lds_ws, st, a);                       \
    } break;
        if (pre) { switch (nbw) { WS_CASE(1, true) WS_CASE(2, true) WS_CASE(3, true) } }
        else { switch (nbw) { WS_CASE(1, false) WS_CASE(2, false) WS_CASE(3, false) } }
#undef WS_CASE
        OVD_CHECK_LAUNCH("conv3d_ws");
        return 0;
    }
    if (exact) {
        a.ws = nullptr;
        switch (nb) { CONV_CASE(1, false) CONV_CASE(2, false) CONV_CASE(3, false) CONV_CASE(4, false) }
    } else {
        switch (nb) { CONV_CASE(1, true) CONV_CASE(2, true) CONV_CASE(3, true) CONV_CASE(4, true) }
    }
#undef CONV_CASE
    OVD_CHECK_LAUNCH("conv3d_cl");
    return 0;
}

int ovd_impl_voxel_rnorm(const float* x, float* out, long nvox, int C, hipStream_t st) {
    OVD_REQUIRE(C % 4 == 0 && C > 0, "voxel_rnorm: C must be a multiple of 4");
    if (nvox == 0) return 0;
    const int q = C / 4;
#define RN_CASE(L)                                                                                         \
    hipLaunchKernelGGL(voxel_rnorm_kernel<L>, dim3(ceil_div(nvox, 4 * (64 / L))), dim3(256), 0, st, x, out, nvox, C)
    if (q % 64 == 0) RN_CASE(64);
    else if (q % 32 == 0) RN_CASE(32);
    else if (q % 16 == 0) RN_CASE(16);
    else if (q % 8 == 0) RN_CASE(8);
    else if (q % 4 == 0) RN_CASE(4);
    else if (q % 2 == 0) RN_CASE(2);
    else RN_CASE(1);
#undef RN_CASE
    OVD_CHECK_LAUNCH("voxel_rnorm");
    return 0;
}

int ovd_impl_act_split(const float* x, const float* gamma, int silu, uint16_t* out, long nvox, int C, hipStream_t st) {
    OVD_REQUIRE(C > 0 && C % 32 == 0, "act_split: C=%d must be a multiple of 32", C);
    if (nvox == 0) return 0;
    const int q = C / 4;
    const float ns = sqrtf((float)C);
#define AS_CASE(L, N)                                                                                          \
    hipLaunchKernelGGL((act_split_kernel<L, N>), dim3(ceil_div(nvox, 4 * (64 / L))), dim3(256), 0, st, x, gamma, silu, \
                       out, nvox, C, ns)
    // LPV lanes x NI float4 = C / 4 (NI <= 6 values in registers per lane)
    if (q % 32 == 0 && q / 32 <= 6) { switch (q / 32) { case 1: AS_CASE(32, 1); break; case 2: AS_CASE(32, 2); break;
        case 3: AS_CASE(32, 3); break; case 4: AS_CASE(32, 4); break; case 5: AS_CASE(32, 5); break; default: AS_CASE(32, 6); } }
    else if (q % 16 == 0 && q / 16 <= 3) { switch (q / 16) { case 1: AS_CASE(16, 1); break; case 2: AS_CASE(16, 2); break;
        default: AS_CASE(16, 3); } }
    else if (q / 8 <= 3) { switch (q / 8) { case 1: AS_CASE(8, 1); break; case 2: AS_CASE(8, 2); break; default: AS_CASE(8, 3); } }
    else OVD_FAIL("act_split: unsupported channel count C=%d", C);
#undef AS_CASE
    OVD_CHECK_LAUNCH("act_split");
    return 0;
}

int ovd_impl_split_weights(const float* w, uint16_t* out, long rows, int Cin, hipStream_t st) {
    OVD_REQUIRE(rows > 0 && Cin > 0, "split_weights: empty tensor");
    const int nchunk = (Cin + 31) / 32;
    hipLaunchKernelGGL(split_weights_kernel, dim3(ceil_div(rows * nchunk * 32, 256)), dim3(256), 0, st, w, out, rows,
                       Cin, nchunk);
    OVD_CHECK_LAUNCH("split_weights");
    return 0;
}

int ovd_impl_to_cl(const float* in, float* out, const float* pre, const float* mul, const float* post, int C, int Cp,
                   long nvox, hipStream_t st) {
    OVD_REQUIRE(Cp >= C && C > 0, "to_channels_last: padded channel count smaller than C");
    if (nvox == 0) return 0;
    hipLaunchKernelGGL(to_cl_kernel, dim3(ceil_div(nvox * Cp, 256)), dim3(256), 0, st, in, out, pre, mul, post, C, Cp,
                       nvox);
    OVD_CHECK_LAUNCH("to_channels_last");
    return 0;
}

int ovd_impl_from_cl(const float* in, float* out, const float* pre, const float* mul, const float* post, int C, int Cs,
                     long nvox, int clamp, hipStream_t st) {
    OVD_REQUIRE(Cs >= C && C > 0, "from_channels_last: source has fewer channels than requested");
    if (nvox == 0) return 0;
    hipLaunchKernelGGL(from_cl_kernel, dim3(ceil_div(nvox * C, 256)), dim3(256), 0, st, in, out, pre, mul, post, C, Cs,
                       nvox, clamp);
    OVD_CHECK_LAUNCH("from_channels_last");
    return 0;
}

int ovd_impl_softmax_rows(float* s, int M, int N, float scale, hipStream_t st) {
    OVD_REQUIRE(M >= 0 && N > 0, "softmax: bad shape");
    if (M == 0) return 0;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(M), dim3(256), 0, st, s, N, scale);
    OVD_CHECK_LAUNCH("softmax_rows");
    return 0;
}
