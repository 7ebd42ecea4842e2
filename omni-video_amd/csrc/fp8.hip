// FP8 (OCP e4m3) path of the linear layers: the reduced-precision mode for the largest model
// (no counterpart in the reference, which runs these layers in bf16: omnivideo/modules/model.py
// :125-154, :159-179, :218-258 are the layers it serves; results are compared with this build's
// own bf16 path).
//
//   quant_rows_fp8_kernel   bf16 rows -> e4m3 rows + one fp32 scale per row (absmax / 448);
//                           used once for weights ([N, K]: a scale per output channel) and per
//                           call for activations that a bf16 kernel produced
//   ln_mod_fp8_kernel       the LayerNorm + modulation kernel of norm.hip with that same row
//                           quantisation fused into its write pass
//   gemm_fp8_kernel<EPI>    out = epilogue((sum_k A8[m][k] W8[n][k]) * sa[m] * sw[n] + bias[n])
//
// Two GEMM loops on the 256 x 256 tile of gemm256.hip.  K % 128 == 0 (every layer of both models):
// gemm_fp8p_kernel, that file's four-phase schedule on K-tiles of 128 bytes per row - the DMA, the LDS
// image and its swizzle, the counted waits and the one-barrier stagger of the two wave groups are
// byte-identical; a phase is 4 v_mfma_scale_f32_32x32x64_f8f6f4 (unit block scales), the matrix time of
// the 16 bf16 MFMAs it replaces for twice the K.  Other K (multiples of 64): gemm_fp8_kernel, a ring of four
// K-tiles of 64 bytes per row, 8 MFMAs per K-tile and wave, vmcnt(4).
#include "ln_row.h"

namespace {

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((address_space(3))) void* lds_ptr;

constexpr float FP8_MAX = 448.f;

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// four floats -> four e4m3 bytes (round to nearest even, saturating at +-448)
__device__ __forceinline__ uint32_t pack_fp8x4(float a, float b, float c, float d) {
    a = fminf(fmaxf(a, -FP8_MAX), FP8_MAX);
    b = fminf(fmaxf(b, -FP8_MAX), FP8_MAX);
    c = fminf(fmaxf(c, -FP8_MAX), FP8_MAX);
    d = fminf(fmaxf(d, -FP8_MAX), FP8_MAX);
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (uint32_t)w;
}

// ------------------------------------------------------------------------- row quantisation
// one wave per row; the row (K <= 512 * VPL values, K % 8 == 0) stays in registers between the two passes
template <int VPL>
__global__ void __launch_bounds__(256) quant_rows_fp8_kernel(const bf16_t* __restrict__ x, long ldx,
                                                             uint8_t* __restrict__ q, long ldq,
                                                             float* __restrict__ scale, int M, int K) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const uint4* xr = reinterpret_cast<const uint4*>(x + (long)row * ldx);
    uint4 v[VPL];
    float mx = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        v[i] = (i * 64 + lane) * 8 < K ? xr[i * 64 + lane] : make_uint4(0, 0, 0, 0);
        const uint32_t w[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            mx = fmaxf(mx, fmaxf(fabsf(bf2f(w[j] & 0xffff)), fabsf(bf2f(w[j] >> 16))));
    }
    mx = wave_max(mx);
    const float s = mx > 0.f ? mx / FP8_MAX : 1.f;
    const float inv = 1.f / s;
    if (lane == 0) scale[row] = s;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const uint32_t w[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
        uint2 o;
        o.x = pack_fp8x4(bf2f(w[0] & 0xffff) * inv, bf2f(w[0] >> 16) * inv, bf2f(w[1] & 0xffff) * inv,
                         bf2f(w[1] >> 16) * inv);
        o.y = pack_fp8x4(bf2f(w[2] & 0xffff) * inv, bf2f(w[2] >> 16) * inv, bf2f(w[3] & 0xffff) * inv,
                         bf2f(w[3] >> 16) * inv);
        if ((i * 64 + lane) * 8 < K) *reinterpret_cast<uint2*>(q + (long)row * ldq + (i * 64 + lane) * 8) = o;
    }
}

// LayerNorm (+ affine) (+ modulation) as ln_mod_kernel of norm.hip computes it, rounded to bf16 as
// that kernel's output is, then quantised per row.
template <int VPL, int WPR, int F>
__global__ void __launch_bounds__(256) ln_mod_fp8_kernel(float* __restrict__ x, long ldx,
                                                         const bf16_t* __restrict__ res, long ldr,
                                                         const float* __restrict__ res_gate,
                                                         const float* __restrict__ shift,
                                                         const float* __restrict__ scale,
                                                         const bf16_t* __restrict__ w, const bf16_t* __restrict__ b,
                                                         uint8_t* __restrict__ out, long ldo,
                                                         float* __restrict__ row_scale, int M, int C, float eps,
                                                         int round_norm) {
    __shared__ float red[3][4];
    float4 y[VPL];
    int row;
    const bool live =
        ln_row<VPL, WPR, F>(x, ldx, res, ldr, res_gate, shift, scale, w, b, M, C, eps, round_norm, red, row, y);
    if constexpr (WPR == 1) {
        if (!live) return;
    }
    float mx = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        y[i] = make_float4(round_bf(y[i].x), round_bf(y[i].y), round_bf(y[i].z), round_bf(y[i].w));
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(y[i].x), fabsf(y[i].y)), fmaxf(fabsf(y[i].z), fabsf(y[i].w))));
    }
    mx = row_max<WPR>(mx, red[2]);
    if (!live) return;
    const float sq = mx > 0.f ? mx / FP8_MAX : 1.f;
    const float inv = 1.f / sq;
    if ((threadIdx.x & (64 * WPR - 1)) == 0) row_scale[row] = sq;
#pragma unroll
    for (int i = 0; i < VPL; ++i)
        *reinterpret_cast<uint32_t*>(out + (long)row * ldo + ln_piece<WPR>(i) * 4) =
            pack_fp8x4(y[i].x * inv, y[i].y * inv, y[i].z * inv, y[i].w * inv);
}

template <int V, int WPR, typename... A>
static void launch_ln_fp8_flags(int flags, int M, hipStream_t st, A... a) {
    const dim3 grid(ceil_div(M, 4 / WPR)), blk(256);
    switch (flags) {
#define LN_F(F) case F: hipLaunchKernelGGL((ln_mod_fp8_kernel<V, WPR, F>), grid, blk, 0, st, a...); break;
        LN_FOR_EACH_FLAGS(LN_F)
#undef LN_F
    }
}

// ------------------------------------------------------------------------------------- GEMM
constexpr int TM = 256, TN = 256, TKB = 64, NSTAGE = 4;   // TKB: bytes (= fp8 k-values) per K-tile row
constexpr int PART = 256 * TKB;        // 16 KiB: one operand of one K-tile
constexpr int STAGE = 2 * PART;        // A part | W part

enum Epi { EPI_BF16 = 0, EPI_BF16_GELU = 1, EPI_F32 = 2, EPI_RESID = 3, EPI_FP8_GELU = 4 };

struct Src {
    const char* base;
    unsigned off[2];
};

__device__ __forceinline__ void stage_part(const Src& s, long kbyte, char* dst_wave) {
    __builtin_amdgcn_global_load_lds(s.base + kbyte + s.off[0], (lds_ptr)dst_wave, 16, 0, 0);
    __builtin_amdgcn_global_load_lds(s.base + kbyte + s.off[1], (lds_ptr)(dst_wave + 8 * 1024), 16, 0, 0);
}

// fragment of a 32-row block: lane (r = lane & 31, h = lane >> 5) takes bytes 32h .. 32h + 31 of row r
__device__ __forceinline__ i32x8 lds_frag8(const char* blk, int off0, int off1) {
    const i32x4 lo = *reinterpret_cast<const i32x4*>(blk + off0);
    const i32x4 hi = *reinterpret_cast<const i32x4*>(blk + off1);
    return (i32x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

template <bool STAGE_NEXT, int VMCNT>
__device__ __forceinline__ void k_tile8(f32x16 (&acc)[2][4], const char* smem, int kt, const Src& sa, const Src& sw,
                                        int off0, int off1, int a_rd, int w_rd, int wave) {
    const char* st = smem + (kt & (NSTAGE - 1)) * STAGE;
    char* nx = const_cast<char*>(smem) + ((kt + 2) & (NSTAGE - 1)) * STAGE + wave * 1024;
    const long kb = (long)(kt + 2) * TKB;
    i32x8 wf[2], af[4];
#pragma unroll
    for (int t = 0; t < 2; ++t) wf[t] = lds_frag8(st + PART + w_rd + t * 2048, off0, off1);
#pragma unroll
    for (int t = 0; t < 4; ++t) af[t] = lds_frag8(st + a_rd + t * 2048, off0, off1);
    if (STAGE_NEXT) {
        stage_part(sw, kb, nx + PART);
        stage_part(sa, kb, nx);
    }
    if (VMCNT == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    if (VMCNT == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
            acc[nt][mt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf[nt], af[mt], acc[nt][mt], 0, 0, 0,
                                                                          0x7F7F7F7F, 0, 0x7F7F7F7F);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
}

template <int EPI>
__global__ void __launch_bounds__(512, 2)
gemm_fp8_kernel(const uint8_t* __restrict__ A, long lda, const float* __restrict__ a_scale, float a_scale_all,
                const uint8_t* __restrict__ W, long ldw, const float* __restrict__ w_scale,
                const bf16_t* __restrict__ bias, void* __restrict__ out, long ldo, const float* __restrict__ gate,
                float out_inv_scale, int M, int N, int K, int tiles_m, int tiles_n, int GM, int wide_store) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int per_group = GM * tiles_n;
    const int grp = bid / per_group;
    const int first_m = grp * GM;
    const int gsz = min(tiles_m - first_m, GM);
    const int tm = first_m + (bid % per_group) % gsz;
    const int tn = (bid % per_group) / gsz;
    const int row0 = tm * TM, col0 = tn * TN;

    Src sa, sw;
    sa.base = reinterpret_cast<const char*>(A);
    sw.base = reinterpret_cast<const char*>(W);
    {
        const int c = (lane & 3) ^ ((4 - ((lane >> 4) & 3)) & 3);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = (i * 8 + wave) * 16 + (lane >> 2);
            const int ar = min(row0 + r, M - 1);
            sa.off[i] = (unsigned)ar * (unsigned)lda + (unsigned)c * 16u;
            sw.off[i] = (unsigned)(col0 + r) * (unsigned)ldw + (unsigned)c * 16u;
        }
    }
    const int fr = lane & 31, fh = lane >> 5;
    const int sz = (4 - ((fr >> 2) & 3)) & 3;
    const int off0 = fr * 64 + (((2 * fh) ^ sz) << 4), off1 = fr * 64 + (((2 * fh + 1) ^ sz) << 4);
    const int a_rd = wr * 8192, w_rd = wc * 4096;

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = K / TKB;      // >= 2 (checked on the host)
    char* const wdst = smem + wave * 1024;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        stage_part(sw, (long)t * TKB, wdst + t * STAGE + PART);
        stage_part(sa, (long)t * TKB, wdst + t * STAGE);
    }
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();            // second half of the waves: half a phase behind
    int kt = 0;
    for (; kt < nk - 2; ++kt) k_tile8<true, 4>(acc, smem, kt, sa, sw, off0, off1, a_rd, w_rd, wave);
    k_tile8<false, 0>(acc, smem, kt, sa, sw, off0, off1, a_rd, w_rd, wave);
    k_tile8<false, -1>(acc, smem, kt + 1, sa, sw, off0, off1, a_rd, w_rd, wave);
    if (wr == 0) __builtin_amdgcn_s_barrier();            // balances the extra barrier above

    // epilogue: acc[nt][mt] register 4g + j is out[m][n + j], m = row0 + wr*128 + mt*32 + (lane & 31),
    // n = col0 + wc*64 + nt*32 + 8g + 4*(lane >> 5)
    constexpr bool NARROW_OUT = EPI == EPI_BF16 || EPI == EPI_BF16_GELU || EPI == EPI_FP8_GELU;
    constexpr int OB = EPI == EPI_FP8_GELU ? 1 : 2;            // bytes per output element on the staged path
    char* ep = smem + wave * 16384;
    const bool staged = NARROW_OUT && wide_store;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = row0 + wr * 128 + mt * 32 + fr;
        if (!staged && m >= M) continue;
        const float sa_m = (a_scale ? a_scale[min(m, M - 1)] : 1.f) * a_scale_all;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = col0 + wc * 64 + nt * 32 + 8 * g + 4 * fh;
                const float4 ws = *reinterpret_cast<const float4*>(w_scale + n);
                float y[4] = {acc[nt][mt][4 * g] * (sa_m * ws.x), acc[nt][mt][4 * g + 1] * (sa_m * ws.y),
                              acc[nt][mt][4 * g + 2] * (sa_m * ws.z), acc[nt][mt][4 * g + 3] * (sa_m * ws.w)};
                if (bias) {
                    const uint2 bp = *reinterpret_cast<const uint2*>(bias + n);
                    y[0] += bf2f(bp.x & 0xffff);
                    y[1] += bf2f(bp.x >> 16);
                    y[2] += bf2f(bp.y & 0xffff);
                    y[3] += bf2f(bp.y >> 16);
                }
                if (EPI == EPI_BF16 || EPI == EPI_BF16_GELU) {
                    if (EPI == EPI_BF16_GELU) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) y[j] = gelu_tanh_f(round_bf(y[j]));
                    }
                    uint2 o;
                    o.x = pack_bf2(y[0], y[1]);
                    o.y = pack_bf2(y[2], y[3]);
                    if (staged) {      // rows of 128 B, 16-byte chunk c at c ^ ((r >> 1) & 7)
                        const int r = mt * 32 + fr, c = nt * 4 + g;
                        *reinterpret_cast<uint2*>(ep + r * 128 + ((c ^ ((r >> 1) & 7)) << 4) + fh * 8) = o;
                    } else {
                        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(out) + (long)m * ldo + n) = o;
                    }
                } else if (EPI == EPI_FP8_GELU) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) y[j] = round_bf(gelu_tanh_f(round_bf(y[j]))) * out_inv_scale;
                    const uint32_t o = pack_fp8x4(y[0], y[1], y[2], y[3]);
                    if (staged) {      // rows of 64 B, chunk c at c ^ ((r >> 2) & 3)
                        const int r = mt * 32 + fr, c = nt * 2 + (g >> 1);
                        *reinterpret_cast<uint32_t*>(ep + r * 64 + ((c ^ ((r >> 2) & 3)) << 4) + (g & 1) * 8 + fh * 4) = o;
                    } else {
                        *reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(out) + (long)m * ldo + n) = o;
                    }
                } else if (EPI == EPI_F32) {
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + (long)m * ldo + n) =
                        make_float4(round_bf(y[0]), round_bf(y[1]), round_bf(y[2]), round_bf(y[3]));
                } else {
                    float gv[4] = {1.f, 1.f, 1.f, 1.f};
                    if (gate) {
                        const float4 g4 = *reinterpret_cast<const float4*>(gate + n);
                        gv[0] = g4.x; gv[1] = g4.y; gv[2] = g4.z; gv[3] = g4.w;
                    }
                    float4* p = reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + (long)m * ldo + n);
                    float4 x = *p;
                    x.x += round_bf(y[0]) * gv[0];
                    x.y += round_bf(y[1]) * gv[1];
                    x.z += round_bf(y[2]) * gv[2];
                    x.w += round_bf(y[3]) * gv[3];
                    *p = x;
                }
            }
    }
    if (staged) {
        // the wave's 128 x 64 tile back out of LDS, 16 B per lane: whole 128-byte lines (bf16) or
        // 64-byte half lines (e4m3) per row
        constexpr int ROWB = 64 * OB, CPR = ROWB / 16, RPI = 64 / CPR;     // row bytes, chunks per row, rows per pass
        const int rr = lane / CPR, cc = lane % CPR;
        char* ob = reinterpret_cast<char*>(out) + ((long)(row0 + wr * 128) * ldo + col0 + wc * 64) * OB + cc * 16;
#pragma unroll
        for (int i = 0; i < 128 / RPI; ++i) {
            const int r = i * RPI + rr;
            const int sw = OB == 2 ? ((r >> 1) & 7) : ((r >> 2) & 3);
            const uint4 v = *reinterpret_cast<const uint4*>(ep + r * ROWB + ((cc ^ sw) << 4));
            if (row0 + wr * 128 + r < M) *reinterpret_cast<uint4*>(ob + (long)r * ldo * OB) = v;
        }
    }
}


// ------------------------------------------------------------------ GEMM, phase-pipelined loop
// The schedule of gemm256.hip on e4m3 operands, for K % 128 == 0: K-tiles of 128 BYTES per row (the same
// LDS image: two buffers of 64 KiB, A part | W part, 256 rows of 128 B, chunk c of row r at c ^ ((r>>1)&7)),
// four phases per K-tile, one 16 KiB half-tile staged per phase, vmcnt(8), the two wave groups one barrier
// apart.  Wave (wr, wc) owns rows {wr*64 .. +63} of both A halves and weight rows {wc*32 .. +31} of both W
// halves; a phase is 4 v_mfma_scale_f32_32x32x64_f8f6f4 (2 row blocks x 2 k-steps of 64), the matrix time of
// the 16 bf16 MFMAs of that kernel's phase for twice the K.
constexpr int PBUF = 2 * 256 * 128;     // one buffer: A part 32 KiB | W part 32 KiB

struct Src4 {
    const char* base;
    unsigned off[4];
};

template <int N>
__device__ __forceinline__ void wait_vm() {
    if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    if (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    if (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}

template <int HALF>
__device__ __forceinline__ void stage_half(const Src4& s, long kbyte, char* part_wave) {
    __builtin_amdgcn_global_load_lds(s.base + kbyte + s.off[2 * HALF], (lds_ptr)(part_wave + (2 * HALF) * 8192), 16, 0, 0);
    __builtin_amdgcn_global_load_lds(s.base + kbyte + s.off[2 * HALF + 1], (lds_ptr)(part_wave + (2 * HALF + 1) * 8192), 16,
                                     0, 0);
}

template <int WAIT>
__device__ __forceinline__ void phase_open8() {
    if (WAIT >= 0) wait_vm<WAIT>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void phase_close8() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
}

// fragment of a 32-row block for k-step ks: lane (r = lane & 31, h = lane >> 5) takes bytes 64*ks + 32h .. +31
// of row r = chunks 4*ks + 2h and 4*ks + 2h + 1 (o[ks][0], o[ks][1]: swizzled byte offsets inside the block)
struct FragOff {
    int o[2][2];
};
__device__ __forceinline__ i32x8 lds_frag8p(const char* blk, const FragOff& f, int ks) {
    const i32x4 lo = *reinterpret_cast<const i32x4*>(blk + f.o[ks][0]);
    const i32x4 hi = *reinterpret_cast<const i32x4*>(blk + f.o[ks][1]);
    return (i32x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// 4 MFMAs: weight block wf (both k-steps) x the two row blocks of one A half; accumulators acc[NH][MB0], acc[NH][MB0 + 1]
template <int NH, int MB0>
__device__ __forceinline__ void quadrant8(f32x16 (&acc)[2][4], const i32x8 (&wf)[2], const i32x8 (&af)[4]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
            acc[NH][MB0 + mb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf[ks], af[mb * 2 + ks], acc[NH][MB0 + mb], 0,
                                                                              0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
}

// TAIL 0: steady state, 1: next-to-last K-tile, 2: last (see gemm256.hip for the schedule and its hazards)
template <int TAIL>
__device__ __forceinline__ void k_tile128(f32x16 (&acc)[2][4], char* smem, int buf, long kb1, long kb2, const Src4& sa,
                                          const Src4& sw, const FragOff& fo, int a_rd, int w_rd, int wave) {
    const char* cur = smem + buf * PBUF;
    char* own = smem + buf * PBUF + wave * 1024;
    char* oth = smem + (buf ^ 1) * PBUF + wave * 1024;
    const char* ca = cur + a_rd;
    const char* cw = cur + PBUF / 2 + w_rd;
    i32x8 w0[2], w1[2], af[4];       // [k-step]; [row block * 2 + k-step]
    // phase 1
    w0[0] = lds_frag8p(cw, fo, 0);
    w0[1] = lds_frag8p(cw, fo, 1);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        af[b * 2] = lds_frag8p(ca + b * 4096, fo, 0);
        af[b * 2 + 1] = lds_frag8p(ca + b * 4096, fo, 1);
    }
    if (TAIL < 2) stage_half<1>(sw, kb1, oth + PBUF / 2);
    phase_open8<TAIL == 2 ? 2 : 8>();
    quadrant8<0, 0>(acc, w0, af);
    phase_close8();
    // phase 2
    w1[0] = lds_frag8p(cw + 16384, fo, 0);
    w1[1] = lds_frag8p(cw + 16384, fo, 1);
    if (TAIL < 2) stage_half<1>(sa, kb1, oth);
    phase_open8<TAIL == 2 ? 0 : 8>();
    quadrant8<1, 0>(acc, w1, af);
    phase_close8();
    // phase 3
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        af[b * 2] = lds_frag8p(ca + 16384 + b * 4096, fo, 0);
        af[b * 2 + 1] = lds_frag8p(ca + 16384 + b * 4096, fo, 1);
    }
    if (TAIL == 0) stage_half<0>(sw, kb2, own + PBUF / 2);
    phase_open8<-1>();
    quadrant8<1, 2>(acc, w1, af);
    phase_close8();
    // phase 4
    if (TAIL == 0) stage_half<0>(sa, kb2, own);
    phase_open8<TAIL == 0 ? 8 : TAIL == 1 ? 4 : -1>();
    quadrant8<0, 2>(acc, w0, af);
    if (TAIL != 0) {
        // the last K-tiles run straight into the epilogue: without a use here hipcc sinks their MFMAs past the
        // closing barriers to the accumulators' first use (and spills the fragments it then has to keep)
#pragma unroll
        for (int i = 0; i < 2; ++i)
            asm volatile("" : "+v"(acc[i][0]), "+v"(acc[i][1]), "+v"(acc[i][2]), "+v"(acc[i][3]));
    }
    phase_close8();
}

// One 256 x 256 output tile: `bid` is the tile's position in the XCD-ordered walk.
typedef unsigned u32x4_nt __attribute__((ext_vector_type(4)));

template <int EPI>
__device__ __forceinline__ void
gemm_fp8p_tile(const uint8_t* __restrict__ A, long lda, const float* __restrict__ a_scale, float a_scale_all,
               const uint8_t* __restrict__ W, long ldw, const float* __restrict__ w_scale,
               const bf16_t* __restrict__ bias, void* __restrict__ out, long ldo, const float* __restrict__ gate,
               float out_inv_scale, int M, int N, int K, int tiles_m, int tiles_n, int GM, int wide_store, int bid) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // per-lane values are derived again for every tile: hoisted out of the tile loop they outlive the K-loop
    // and hipcc spills them
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    const int per_group = GM * tiles_n;
    const int grp = bid / per_group;
    const int first_m = grp * GM;
    const int gsz = min(tiles_m - first_m, GM);
    const int tm = first_m + (bid % per_group) % gsz;
    const int tn = (bid % per_group) / gsz;
    const int row0 = tm * TM, col0 = tn * TN;

    Src4 sa, sw;
    sa.base = reinterpret_cast<const char*>(A);
    sw.base = reinterpret_cast<const char*>(W);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (i * 8 + wave) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        const int ar = min(row0 + r, M - 1);
        sa.off[i] = (unsigned)ar * (unsigned)lda + (unsigned)c * 16u;
        sw.off[i] = (unsigned)(col0 + r) * (unsigned)ldw + (unsigned)c * 16u;
    }
    const int fr = lane & 31, fh = lane >> 5;
    FragOff fo;
    {
        const int swz = (fr >> 1) & 7;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j) fo.o[ks][j] = fr * 128 + (((4 * ks + 2 * fh + j) ^ swz) << 4);
    }
    const int a_rd = wr * 8192, w_rd = wc * 4096;

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    char* const wdst = smem + wave * 1024;
    const int nk = K / 128;      // >= 2 (checked on the host)
    stage_half<0>(sw, 0, wdst + PBUF / 2);
    stage_half<0>(sa, 0, wdst);
    stage_half<1>(sw, 0, wdst + PBUF / 2);
    stage_half<1>(sa, 0, wdst);
    stage_half<0>(sw, 128, wdst + PBUF + PBUF / 2);
    stage_half<0>(sa, 128, wdst + PBUF);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();            // second half of the waves: one barrier behind
    int t = 0;
    for (; t + 2 < nk; ++t)
        k_tile128<0>(acc, smem, t & 1, (long)(t + 1) * 128, (long)(t + 2) * 128, sa, sw, fo, a_rd, w_rd, wave);
    k_tile128<1>(acc, smem, t & 1, (long)(t + 1) * 128, 0, sa, sw, fo, a_rd, w_rd, wave);
    ++t;
    k_tile128<2>(acc, smem, t & 1, 0, 0, sa, sw, fo, a_rd, w_rd, wave);
    if (wr == 0) __builtin_amdgcn_s_barrier();            // balances the extra barrier above
    asm volatile("" ::: "memory");                        // keeps the epilogue's scale / bias loads out of the last K-tile

    // epilogue: acc[nh][mh*2 + mb] register 4g + j is out[m][n + j],
    //   m = row0 + mh*128 + wr*64 + mb*32 + (lane & 31),  n = col0 + nh*128 + wc*32 + 8g + 4*(lane >> 5)
    constexpr bool NARROW_OUT = EPI == EPI_BF16 || EPI == EPI_BF16_GELU || EPI == EPI_FP8_GELU;
    constexpr int OB = EPI == EPI_FP8_GELU ? 1 : 2;            // bytes per output element on the staged path
    constexpr int ROWB = 256 * OB;                             // bytes per tile row in the staged image
    const bool staged = NARROW_OUT && wide_store;
    // staged: one 128-row half of the tile per LDS buffer, rows of ROWB bytes, 16-byte chunk c of row r at
    // c ^ (r & 15); read back 16 B per lane so that ROWB / 16 lanes store one whole row segment
#pragma unroll
    for (int mh = 0; mh < 2; ++mh) {
        char* img = smem + mh * PBUF;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const int rl = wr * 64 + mb * 32 + fr;              // row inside the half
            const int m = row0 + mh * 128 + rl;
            if (!staged && m >= M) continue;
            const float sa_m = (a_scale ? a_scale[min(m, M - 1)] : 1.f) * a_scale_all;
#pragma unroll
            for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nl = nh * 128 + wc * 32 + 8 * g + 4 * fh;      // column inside the tile
                    const int n = col0 + nl;
                    const f32x16& a16 = acc[nh][mh * 2 + mb];
                    const float4 ws = *reinterpret_cast<const float4*>(w_scale + n);
                    float y[4] = {a16[4 * g] * (sa_m * ws.x), a16[4 * g + 1] * (sa_m * ws.y),
                                  a16[4 * g + 2] * (sa_m * ws.z), a16[4 * g + 3] * (sa_m * ws.w)};
                    if (bias) {
                        const uint2 bp = *reinterpret_cast<const uint2*>(bias + n);
                        y[0] += bf2f(bp.x & 0xffff);
                        y[1] += bf2f(bp.x >> 16);
                        y[2] += bf2f(bp.y & 0xffff);
                        y[3] += bf2f(bp.y >> 16);
                    }
                    if (EPI == EPI_BF16 || EPI == EPI_BF16_GELU) {
                        if (EPI == EPI_BF16_GELU) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) y[j] = gelu_tanh_f(round_bf(y[j]));
                        }
                        uint2 o;
                        o.x = pack_bf2(y[0], y[1]);
                        o.y = pack_bf2(y[2], y[3]);
                        if (staged) {
                            const int c = nl >> 3;
                            *reinterpret_cast<uint2*>(img + rl * ROWB + ((c ^ (rl & 15)) << 4) + fh * 8) = o;
                        } else {
                            *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(out) + (long)m * ldo + n) = o;
                        }
                    } else if (EPI == EPI_FP8_GELU) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) y[j] = round_bf(gelu_tanh_f(round_bf(y[j]))) * out_inv_scale;
                        const uint32_t o = pack_fp8x4(y[0], y[1], y[2], y[3]);
                        if (staged) {
                            const int c = nl >> 4;
                            *reinterpret_cast<uint32_t*>(img + rl * ROWB + ((c ^ (rl & 15)) << 4) + (nl & 15)) = o;
                        } else {
                            *reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(out) + (long)m * ldo + n) = o;
                        }
                    } else if (EPI == EPI_F32) {
                        *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + (long)m * ldo + n) =
                            make_float4(round_bf(y[0]), round_bf(y[1]), round_bf(y[2]), round_bf(y[3]));
                    } else {
                        float gv[4] = {1.f, 1.f, 1.f, 1.f};
                        if (gate) {
                            const float4 g4 = *reinterpret_cast<const float4*>(gate + n);
                            gv[0] = g4.x; gv[1] = g4.y; gv[2] = g4.z; gv[3] = g4.w;
                        }
                        float4* p = reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + (long)m * ldo + n);
                        float4 x = *p;
                        x.x += round_bf(y[0]) * gv[0];
                        x.y += round_bf(y[1]) * gv[1];
                        x.z += round_bf(y[2]) * gv[2];
                        x.w += round_bf(y[3]) * gv[3];
                        *p = x;
                    }
                }
        }
        if (staged) {       // uniform: every thread reaches the barrier
            __syncthreads();
            constexpr int CPR = ROWB / 16, RPI = 512 / CPR;       // chunks per row; rows per pass of the workgroup
            const int rr = tid / CPR, cc = tid % CPR;
            char* ob = reinterpret_cast<char*>(out) + ((long)(row0 + mh * 128) * ldo + col0) * OB + cc * 16;
#pragma unroll
            for (int i = 0; i < 128 / RPI; ++i) {
                const int r = i * RPI + rr;
                const uint4 v = *reinterpret_cast<const uint4*>(img + r * ROWB + ((cc ^ (r & 15)) << 4));
                if (row0 + mh * 128 + r < M)        // streaming store, as in the bf16 kernel
                    __builtin_nontemporal_store((u32x4_nt){v.x, v.y, v.z, v.w},
                                                reinterpret_cast<u32x4_nt*>(ob + (long)r * ldo * OB));
            }
        }
    }
}


// Persistent launch, one workgroup per CU, as gemm256_kernel: the workgroups of an XCD take the tiles of that
// XCD's chunk of the walk round by round.
template <int EPI>
__global__ void __launch_bounds__(512, 2)
gemm_fp8p_kernel(const uint8_t* __restrict__ A, long lda, const float* __restrict__ a_scale, float a_scale_all,
                 const uint8_t* __restrict__ W, long ldw, const float* __restrict__ w_scale,
                 const bf16_t* __restrict__ bias, void* __restrict__ out, long ldo, const float* __restrict__ gate,
                 float out_inv_scale, int M, int N, int K, int tiles_m, int tiles_n, int GM, int wide_store) {
    const int nwg = tiles_m * tiles_n;
    const int xcd = blockIdx.x & 7, q = nwg >> 3, r = nwg & 7;
    const int first = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int count = q + (xcd < r ? 1 : 0);
    const int per_xcd = (gridDim.x + 7) >> 3;
    for (int t = blockIdx.x >> 3; t < count; t += per_xcd) {
        gemm_fp8p_tile<EPI>(A, lda, a_scale, a_scale_all, W, ldw, w_scale, bias, out, ldo, gate, out_inv_scale, M, N, K,
                            tiles_m, tiles_n, GM, wide_store, first + t);
        __syncthreads();          // the LDS buffers are staged again by the next tile
    }
}

}  // namespace

int ovd_impl_quant_rows_fp8(const bf16_t* x, long ldx, uint8_t* q, long ldq, float* scale, int M, int K,
                            hipStream_t st) {
    OVD_REQUIRE(K % 8 == 0 && K > 0 && K <= 512 * 28, "quant_rows_fp8: K=%d must be a multiple of 8, at most 14336", K);
    OVD_REQUIRE(ldx % 8 == 0 && ldq % 8 == 0 && ldx >= K && ldq >= K, "quant_rows_fp8: bad row strides");
    if (M == 0) return 0;
    dim3 grid(ceil_div(M, 4)), blk(256);
    // registers per lane: the smallest instantiated row capacity (512 * V values) that holds K
    const int vpl = ceil_div(K, 512);
#define Q_LAUNCH(V) hipLaunchKernelGGL((quant_rows_fp8_kernel<V>), grid, blk, 0, st, x, ldx, q, ldq, scale, M, K)
    if (vpl <= 1) Q_LAUNCH(1);
    else if (vpl <= 3) Q_LAUNCH(3);
    else if (vpl <= 6) Q_LAUNCH(6);
    else if (vpl <= 10) Q_LAUNCH(10);
    else if (vpl <= 18) Q_LAUNCH(18);
    else Q_LAUNCH(28);
#undef Q_LAUNCH
    OVD_CHECK_LAUNCH("quant_rows_fp8");
    return 0;
}

int ovd_impl_ln_fp8(float* x, long ldx, const bf16_t* res, long ldr, const float* res_gate, const float* shift,
                    const float* scale, const bf16_t* w, const bf16_t* b, uint8_t* out, long ldo, float* row_scale,
                    int M, int C, float eps, int round_norm, hipStream_t st) {
    OVD_REQUIRE(C % 256 == 0, "layernorm fp8: C=%d must be a multiple of 256", C);
    OVD_REQUIRE(res != nullptr || res_gate == nullptr, "layernorm fp8: residual gate without a residual");
    OVD_REQUIRE(res == nullptr || (ldr % 4 == 0 && ldr >= C), "layernorm fp8: bad residual row stride");
    OVD_REQUIRE((w == nullptr) == (b == nullptr), "layernorm fp8: weight and bias must come together");
    OVD_REQUIRE((scale == nullptr) == (shift == nullptr), "layernorm fp8: scale and shift must come together");
    OVD_REQUIRE(ldx % 4 == 0 && ldo % 4 == 0, "layernorm fp8: row strides must be multiples of 4");
    if (M == 0) return 0;
    const int flags = (res ? LN_RES : 0) | (res_gate ? LN_GATE : 0) | (w ? LN_AFF : 0) | (scale ? LN_MOD : 0);
#define LN_CASE(V, WPR)                                                                                       \
    case V * WPR:                                                                                             \
        launch_ln_fp8_flags<V, WPR>(flags, M, st, x, ldx, res, ldr, res_gate, shift, scale, w, b, out, ldo,   \
                                    row_scale, M, C, eps, round_norm);                                        \
        break;
    switch (C / 256) {
        LN_FOR_EACH_SHAPE(LN_CASE)
        default:
            OVD_FAIL("layernorm fp8: unsupported width C=%d", C);
    }
#undef LN_CASE
    OVD_CHECK_LAUNCH("layernorm fp8");
    return 0;
}

bool ovd_gemm_fp8_applicable(long lda, long ldw, int M, int N, int K) {
    return M > 0 && N % TN == 0 && K % TKB == 0 && K >= 2 * TKB && lda % 16 == 0 && ldw % 16 == 0 &&
           (long)M * lda < (1L << 32) && (long)N * ldw < (1L << 32);
}

int ovd_impl_gemm_fp8(const uint8_t* A, long lda, const float* a_scale, float a_scale_all, const uint8_t* W, long ldw,
                      const float* w_scale, const bf16_t* bias, void* out, long ldo, const float* gate,
                      float out_scale, int M, int N, int K, int epilogue, hipStream_t st) {
    OVD_REQUIRE(ovd_gemm_fp8_applicable(lda, ldw, M, N, K),
                "gemm_fp8: shape M=%d N=%d K=%d (lda %ld, ldw %ld) not supported: N %% 256, K %% 64, K >= 128, "
                "strides %% 16, operands < 4 GiB", M, N, K, lda, ldw);
    OVD_REQUIRE(lda >= K && ldw >= K && ldo >= N && ldo % 4 == 0, "gemm_fp8: bad strides");
    OVD_REQUIRE(out_scale > 0.f && a_scale_all > 0.f, "gemm_fp8: scales must be positive");
    const int tiles_m = ceil_div(M, TM), tiles_n = N / TN;
    dim3 grid(tiles_m * tiles_n), blk(512);
    const size_t lds = NSTAGE * STAGE;          // = 2 * PBUF: both loops use 128 KiB
    const float out_inv = 1.f / out_scale;
    // K-tiles of 128 bytes in four phases where K allows it, the ring of 64-byte K-tiles otherwise
    static std::atomic<int> ring_cache{INT32_MIN};
    const bool phased = K % 128 == 0 && K >= 256 && ovd_env_int(ring_cache, "OVD_FP8_RING", 0) == 0;
    // the phased kernel is persistent: one workgroup per CU when there are more tiles than that (a multiple of 8,
    // so that every XCD gets the same number), one per tile otherwise
    static std::atomic<int> persist_cache{INT32_MIN};
    const int cus = ovd_env_int(persist_cache, "OVD_GEMM_PERSIST", 1) ? max(8, ovd_device_cus() / 8 * 8) : 1 << 30;
    const dim3 pgrid(min(tiles_m * tiles_n, cus));
    // 16-byte output stores (bf16 and e4m3 epilogues) need 16-byte aligned rows
    const long row_bytes = ldo * (epilogue == EPI_FP8_GELU ? 1 : 2);
    const int wide = (row_bytes % 16 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0) ? 1 : 0;
#define GEMM_CASE(E)                                                                                      \
    case E: {                                                                                             \
        static OvdLdsLimit lim;                                                                           \
        if (!lim.ensure(reinterpret_cast<const void*>(&gemm_fp8_kernel<E>), lds))                         \
            OVD_FAIL("gemm_fp8: cannot reserve %zu bytes of LDS", lds);                                    \
        static OvdLdsLimit limp;                                                                          \
        if (!limp.ensure(reinterpret_cast<const void*>(&gemm_fp8p_kernel<E>), lds))                       \
            OVD_FAIL("gemm_fp8: cannot reserve %zu bytes of LDS", lds);                                    \
        if (phased)                                                                                       \
            hipLaunchKernelGGL((gemm_fp8p_kernel<E>), pgrid, blk, lds, st, A, lda, a_scale, a_scale_all, W, ldw, \
                               w_scale, bias, out, ldo, gate, out_inv, M, N, K, tiles_m, tiles_n, 4, wide); \
        else                                                                                              \
            hipLaunchKernelGGL((gemm_fp8_kernel<E>), grid, blk, lds, st, A, lda, a_scale, a_scale_all, W, ldw, \
                               w_scale, bias, out, ldo, gate, out_inv, M, N, K, tiles_m, tiles_n, 4, wide); \
    } break;
    switch (epilogue) {
        GEMM_CASE(EPI_BF16) GEMM_CASE(EPI_BF16_GELU) GEMM_CASE(EPI_F32) GEMM_CASE(EPI_RESID) GEMM_CASE(EPI_FP8_GELU)
        default:
            OVD_FAIL("gemm_fp8: unknown epilogue %d", epilogue);
    }
#undef GEMM_CASE
    OVD_CHECK_LAUNCH("gemm_fp8");
    return 0;
}
