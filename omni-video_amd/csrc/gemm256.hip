// Large-shape bf16 GEMM with the epilogues of gemm.hip:
//   out[m][n] = epilogue( sum_k A[m][k] * W[n][k] + bias[n] ),  A [M,K], W [N,K] (nn.Linear).
//
// One workgroup = 8 waves = a 256 x 256 output tile at a time, one (persistent) workgroup per CU.  Wave (wr, wc) = (w >> 2, w & 3)
// owns rows {wr*64 .. +63} of BOTH 128-row halves of the A tile and weight rows {wc*32 .. +31} of both
// halves of the W tile: 8 x 4 tiles of v_mfma_f32_16x16x32_bf16 (128 accumulator registers), computed
// transposed (weight rows on the MFMA M axis) so that a lane ends up with 4 consecutive output columns
// of one row.
//
// Operands stream HBM -> LDS with 16-byte global_load_lds in K-tiles of 64.  LDS rows are 128 B (64
// k-values), so every DMA instruction fetches 8 whole 128-byte lines; two buffers of 64 KiB
// (A part 256 rows x 128 B | W part 256 rows x 128 B).  16-byte chunk c of row r sits at chunk position
// c ^ ((r >> 1) & 7): every 16-lane group of a ds_read_b128 fragment read (either half of the row) hits 16
// distinct bank slots.  The swizzle is applied on the SOURCE address of the DMA (its LDS image is
// lane-linear) and again on the read address.
//
// A K-tile is four phases of 16 MFMAs.  Every phase puts ONE 16 KiB half-tile (128 rows of one operand)
// in flight, 2 DMA instructions per wave, and the DMA queue is never drained inside the loop:
//   phase 1: read W-half0 (4 fragments) and A-half0 (8), MFMA quadrant (A0, W0); stage W-half1 of K-tile t+1
//   phase 2: read W-half1 (4),                           MFMA (A0, W1);            stage A-half1 of t+1
//   phase 3: read A-half1 (8),                           MFMA (A1, W1);            stage W-half0 of t+2
//   phase 4: -,                                          MFMA (A1, W0);            stage A-half0 of t+2
// Because every wave reads a given half-tile in the same phase, the half-tile is free for the K-tile after
// next right behind that phase.  The two waves that share a SIMD (w and w + 4) run one barrier apart:
// waves 4-7 execute one extra s_barrier before the loop (waves 0-3 one after it), so one of them issues its
// LDS reads and DMA while the other one feeds the matrix pipe.
//
// Hazards the schedule is built around:
//   * write-after-read: a half-tile is restaged two or more phases after its last read (the reads are
//     retired by the lgkmcnt(0) behind the phase's first barrier, for the later wave group one barrier on);
//   * read-after-write: a half-tile is read one phase after the counted wait that retires it
//     (vmcnt(8) = every DMA issued four or more phases ago has landed) and the barrier behind that wait;
//   * raw s_barrier + counted waits only inside the loop: a __syncthreads() would drain the DMA queue;
//   * no ordinary global load while DMA is in flight (bias / gate are read ahead of the first DMA).
//
// Measured on the loop (FFN-down shape, K = 8960): matrix pipe busy in 73 % of the kernel's cycles.
// Without any DMA in the loop (timing only) it runs 19 % faster, with every tile streaming the same
// L2-resident operands 9 % faster; waiting for DMA issued two phases ago instead of four costs 1 %
// (latency is covered; what remains is the L2 -> LDS path's throughput).  Raising the waves' priority
// for the MFMA phase costs 1-3 %, handing the matrix pipe over 2-8 MFMAs before the end of a phase
// 4-7 %, retiring the fragment reads before the barrier instead of behind it 1-2 %.
#include "common.h"

namespace {

constexpr int TM = 256, TN = 256;
constexpr int KT = 64;                        // K-tile
constexpr int PAIR = 2 * 256 * KT * 2;        // one buffer: A part 32 KiB | W part 32 KiB
constexpr int NPAIR = 2;                      // buffers

enum Epi { EPI_BF16 = 0, EPI_BF16_GELU = 1, EPI_F32 = 2, EPI_RESID = 3 };

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 lds_frag(const char* p) { return *reinterpret_cast<const bf16x8*>(p); }

struct Src4 {
    const char* base;
    unsigned off[4];       // this lane's byte offsets (clamped row * ld + swizzled chunk) for its four rows
};

template <int N>
__device__ __forceinline__ void wait_vm() {
    if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    if (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    if (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}

// Half-tile HALF (rows HALF*128 .. +127) of one operand: DMA instruction i of wave w covers part rows
// (i*8 + w)*8 .. +7, 8 lanes per 128-byte row.
template <int HALF>
__device__ __forceinline__ void stage_half(const Src4& s, long kbyte, char* part_wave) {
    __builtin_amdgcn_global_load_lds(s.base + kbyte + s.off[2 * HALF], (lds_ptr)(part_wave + (2 * HALF) * 8192), 16, 0, 0);
    __builtin_amdgcn_global_load_lds(s.base + kbyte + s.off[2 * HALF + 1], (lds_ptr)(part_wave + (2 * HALF + 1) * 8192), 16,
                                     0, 0);
}

template <int WAIT>
__device__ __forceinline__ void phase_open() {
    if (WAIT >= 0) wait_vm<WAIT>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void phase_close() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
}

// 16 MFMAs: accumulator columns nt0, nt0 + 1 x rows mt0 .. mt0 + 3, both k-halves; fragments [block * 2 + k-half]
template <int NT0, int MT0>
__device__ __forceinline__ void quadrant(f32x4 (&acc)[4][8], const bf16x8 (&wf)[4], const bf16x8 (&af)[8]) {
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                acc[NT0 + nt][MT0 + mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt * 2 + kh], af[mt * 2 + kh],
                                                                                  acc[NT0 + nt][MT0 + mt], 0, 0, 0);
}

// TAIL 0: steady state, 1: next-to-last K-tile (nothing to stage two tiles ahead), 2: last.
// bias_lane / bias_q (TAIL 2 only): this lane's bias values are requested in the last K-tile, behind its last counted
// wait, and land under its remaining two phases.
template <int TAIL>
__device__ __forceinline__ void k_tile64(f32x4 (&acc)[4][8], char* smem, int buf, long kb1, long kb2, const Src4& sa,
                                         const Src4& sw, int lo0, int lo1, int a_rd, int w_rd, int wave,
                                         const bf16_t* bias_lane = nullptr, unsigned long long* bias_q = nullptr) {
    const char* cur = smem + buf * PAIR;
    char* own = smem + buf * PAIR + wave * 1024;
    char* oth = smem + (buf ^ 1) * PAIR + wave * 1024;
    const char* ca = cur + a_rd;
    const char* cw = cur + PAIR / 2 + w_rd;
    bf16x8 w0[4], w1[4], af[8];
    // phase 1
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        w0[b * 2] = lds_frag(cw + b * 2048 + lo0);
        w0[b * 2 + 1] = lds_frag(cw + b * 2048 + lo1);
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        af[b * 2] = lds_frag(ca + b * 2048 + lo0);
        af[b * 2 + 1] = lds_frag(ca + b * 2048 + lo1);
    }
    if (TAIL < 2) stage_half<1>(sw, kb1, oth + PAIR / 2);
    phase_open<TAIL == 2 ? 2 : 8>();
    quadrant<0, 0>(acc, w0, af);
    phase_close();
    // phase 2
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        w1[b * 2] = lds_frag(cw + 16384 + b * 2048 + lo0);
        w1[b * 2 + 1] = lds_frag(cw + 16384 + b * 2048 + lo1);
    }
    if (TAIL < 2) stage_half<1>(sa, kb1, oth);
    phase_open<TAIL == 2 ? 0 : 8>();
    if (TAIL == 2) {
        // Inline assembly, so that the compiler keeps no books on these loads: one it knows to be pending makes it
        // drain the DMA queue (s_waitcnt vmcnt(0)) wherever its register shuffling touches the destination.  Nothing
        // else is in flight here (the wait above was vmcnt(0)); the epilogue waits for them itself.
        asm volatile("global_load_dwordx2 %0, %4, off\n\tglobal_load_dwordx2 %1, %4, off offset:32\n\t"
                     "global_load_dwordx2 %2, %4, off offset:256\n\tglobal_load_dwordx2 %3, %4, off offset:288"
                     : "=&v"(bias_q[0]), "=&v"(bias_q[1]), "=&v"(bias_q[2]), "=&v"(bias_q[3])
                     : "v"(bias_lane)
                     : "memory");
    }
    quadrant<2, 0>(acc, w1, af);
    phase_close();
    // phase 3
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        af[b * 2] = lds_frag(ca + 16384 + b * 2048 + lo0);
        af[b * 2 + 1] = lds_frag(ca + 16384 + b * 2048 + lo1);
    }
    if (TAIL == 0) stage_half<0>(sw, kb2, own + PAIR / 2);
    phase_open<-1>();
    quadrant<2, 4>(acc, w1, af);
    phase_close();
    // phase 4
    if (TAIL == 0) stage_half<0>(sa, kb2, own);
    phase_open<TAIL == 0 ? 8 : TAIL == 1 ? 4 : -1>();
    quadrant<0, 4>(acc, w0, af);
    phase_close();
}

// One 256 x 256 output tile: `bid` is the tile's position in the XCD-ordered walk.
template <int EPI>
__device__ __forceinline__ void
gemm256_tile(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ W, long ldw,
             const bf16_t* __restrict__ bias, void* __restrict__ out, long ldo, const float* __restrict__ gate,
             int M, int N, int K, int tiles_m, int tiles_n, int GM, int wide_store, int bid) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
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
        sa.off[i] = (unsigned)ar * (unsigned)lda * 2u + (unsigned)c * 16u;
        sw.off[i] = (unsigned)(col0 + r) * (unsigned)ldw * 2u + (unsigned)c * 16u;
    }
    const int lo0 = (lane & 15) * 128 + (((lane >> 4) ^ (((lane & 15) >> 1) & 7)) << 4), lo1 = lo0 ^ 64;
    const int a_rd = wr * 8192, w_rd = wc * 4096;

    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    char* const wdst = smem + wave * 1024;
    const int nk = K / 64;      // >= 2 (checked on the host)
    stage_half<0>(sw, 0, wdst + PAIR / 2);
    stage_half<0>(sa, 0, wdst);
    stage_half<1>(sw, 0, wdst + PAIR / 2);
    stage_half<1>(sa, 0, wdst);
    stage_half<0>(sw, 128, wdst + PAIR + PAIR / 2);
    stage_half<0>(sa, 128, wdst + PAIR);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // the two half-tiles phase 1 reads; the loop's waits retire the rest
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();            // second half of the waves: one barrier behind
    int t = 0;
    for (; t + 2 < nk; ++t)
        k_tile64<0>(acc, smem, t & 1, (long)(t + 1) * 128, (long)(t + 2) * 128, sa, sw, lo0, lo1, a_rd, w_rd, wave);
    k_tile64<1>(acc, smem, t & 1, (long)(t + 1) * 128, 0, sa, sw, lo0, lo1, a_rd, w_rd, wave);
    ++t;
    // bias of this lane's 16 output columns: n = col0 + (nt >> 1) * 128 + wc * 32 + (nt & 1) * 16 + fq * 4
    // (without a bias the loads read the weights instead - any valid address - and the values are dropped: no branch
    // around inline assembly whose results the compiler cannot see arriving)
    unsigned long long bias_q[4];
    k_tile64<2>(acc, smem, t & 1, 0, 0, sa, sw, lo0, lo1, a_rd, w_rd, wave, (bias ? bias : W) + col0 + wc * 32 + fq * 4,
                bias_q);
    if (wr == 0) __builtin_amdgcn_s_barrier();            // balances the extra barrier above
    // the bias values have landed; the registers are operands so that no use of them can be scheduled above the wait
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(bias_q[0]), "+v"(bias_q[1]), "+v"(bias_q[2]), "+v"(bias_q[3])::"memory");

    // epilogue: lane holds out[m][n .. n+3],
    //   m = row0 + (mt >> 2) * 128 + wr * 64 + (mt & 3) * 16 + fr,  n = col0 + (nt >> 1) * 128 + wc * 32 + (nt & 1) * 16 + fq * 4
    float bv[4][4], gv[4][4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const uint32_t blo = bias ? (uint32_t)bias_q[nt] : 0u, bhi = bias ? (uint32_t)(bias_q[nt] >> 32) : 0u;
        bv[nt][0] = bf2f(blo & 0xffff);
        bv[nt][1] = bf2f(blo >> 16);
        bv[nt][2] = bf2f(bhi & 0xffff);
        bv[nt][3] = bf2f(bhi >> 16);
        if (EPI == EPI_RESID && !wide_store && gate) {      // (the row-wise epilogue fetches the gate in its own layout)
            const float4 g = *reinterpret_cast<const float4*>(gate + col0 + (nt >> 1) * 128 + wc * 32 + (nt & 1) * 16 + fq * 4);
            gv[nt][0] = g.x; gv[nt][1] = g.y; gv[nt][2] = g.z; gv[nt][3] = g.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) gv[nt][j] = 1.f;
        }
    }
    if ((EPI == EPI_BF16 || EPI == EPI_BF16_GELU) && wide_store) {
        // bf16 output through LDS, one 128-row half of the tile per ring buffer (64 KiB = 128 rows of 512 B):
        // written in the accumulator layout (8 B per lane), read back 16 B per lane so that 32 lanes store one
        // whole 512-byte row segment.  16-byte chunk c of row r sits at chunk c ^ (r & 15): conflict-free both ways.
        bf16_t* ob = reinterpret_cast<bf16_t*>(out) + (long)row0 * ldo + col0 + (lane & 31) * 8;
        // both halves are staged before the one barrier (the ring's two buffers are free), and the barrier waits for
        // this wave's LDS writes only: __syncthreads() would also wait for global stores in flight
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                float y[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) y[j] = acc[nt][mt][j] + bv[nt][j];
                if (EPI == EPI_BF16_GELU) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) y[j] = gelu_tanh_f(round_bf(y[j]));
                }
                uint2 o;
                o.x = pack_bf2(y[0], y[1]);
                o.y = pack_bf2(y[2], y[3]);
                const int r = wr * 64 + (mt & 3) * 16 + fr;
                const int c = (nt >> 1) * 16 + wc * 4 + (nt & 1) * 2 + (fq >> 1);
                *reinterpret_cast<uint2*>(smem + (mt >> 2) * PAIR + r * 512 + ((c ^ fr) << 4) + (fq & 1) * 8) = o;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const char* img = smem + hf * PAIR;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = i * 16 + wave * 2 + (lane >> 5);
                const uint4 v = *reinterpret_cast<const uint4*>(img + r * 512 + (((lane & 31) ^ (r & 15)) << 4));
                // streaming store: the output is not read again by this kernel and would only push operand
                // tiles out of L2 (1-2 % on the K = 1536 shapes)
                if (row0 + hf * 128 + r < M)
                    __builtin_nontemporal_store((u32x4){v.x, v.y, v.z, v.w},
                                                reinterpret_cast<u32x4*>(ob + (long)(hf * 128 + r) * ldo));
            }
        }
    } else if ((EPI == EPI_F32 || EPI == EPI_RESID) && wide_store) {
        // fp32 output (out = round_bf(acc + bias), or out += gate * that, in place) through LDS as well: the tile is
        // staged as bf16 - the values are bf16-representable - in the image of the bf16 path, both 128-row halves at
        // once (the ring is free), and each wave then walks whole rows: 64 lanes x 16 bytes = one contiguous 1 KiB row
        // segment per load and per store.  In the accumulator layout a wave's load touches sixteen rows with 64 bytes in
        // each, and the 512 KB a tile reads and writes then take longer than half its products.
        // The old values of the first PF rows are requested between the staging and its barrier, the later ones PF rows
        // ahead of their use.
        float* const orow = reinterpret_cast<float*>(out) + col0 + lane * 4;
        float4 g4 = make_float4(1.f, 1.f, 1.f, 1.f);
        if (EPI == EPI_RESID && gate) g4 = *reinterpret_cast<const float4*>(gate + col0 + lane * 4);
        constexpr int PF = 12;
        float4 xo[PF];
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                uint2 o;
                o.x = pack_bf2(acc[nt][mt][0] + bv[nt][0], acc[nt][mt][1] + bv[nt][1]);
                o.y = pack_bf2(acc[nt][mt][2] + bv[nt][2], acc[nt][mt][3] + bv[nt][3]);
                const int r = wr * 64 + (mt & 3) * 16 + fr;
                const int c = (nt >> 1) * 16 + wc * 4 + (nt & 1) * 2 + (fq >> 1);
                *reinterpret_cast<uint2*>(smem + (mt >> 2) * PAIR + r * 512 + ((c ^ fr) << 4) + (fq & 1) * 8) = o;
            }
        asm volatile("" ::: "memory");      // (the accumulators are dead from here: room for the rows in flight)
        if (EPI == EPI_RESID) {
#pragma unroll
            for (int i = 0; i < PF; ++i)
                xo[i] = *reinterpret_cast<const float4*>(orow + (long)min(row0 + wave + 8 * i, M - 1) * ldo);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (not __syncthreads(): the loads above stay in flight)
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int r = wave + 8 * i, rr = r & 127;
            const uint2 v = *reinterpret_cast<const uint2*>(smem + (r >> 7) * PAIR + rr * 512 +
                                                            (((lane >> 1) ^ (rr & 15)) << 4) + (lane & 1) * 8);
            const float t0 = bf2f(v.x & 0xffff), t1 = bf2f(v.x >> 16), t2 = bf2f(v.y & 0xffff), t3 = bf2f(v.y >> 16);
            float4 x;
            if (EPI == EPI_RESID) {
                x = xo[i % PF];
                if (i + PF < 32)
                    xo[i % PF] = *reinterpret_cast<const float4*>(orow + (long)min(row0 + r + 8 * PF, M - 1) * ldo);
                x.x += t0 * g4.x;
                x.y += t1 * g4.y;
                x.z += t2 * g4.z;
                x.w += t3 * g4.w;
            } else {
                x = make_float4(t0, t1, t2, t3);
            }
            if (row0 + r < M) *reinterpret_cast<float4*>(orow + (long)(row0 + r) * ldo) = x;
        }
    } else
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
        const int m = row0 + (mt >> 2) * 128 + wr * 64 + (mt & 3) * 16 + fr;
        if (m >= M) continue;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n = col0 + (nt >> 1) * 128 + wc * 32 + (nt & 1) * 16 + fq * 4;
            float y[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = acc[nt][mt][j] + bv[nt][j];
            if (EPI == EPI_BF16 || EPI == EPI_BF16_GELU) {
                if (EPI == EPI_BF16_GELU) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) y[j] = gelu_tanh_f(round_bf(y[j]));
                }
                uint2 o;
                o.x = pack_bf2(y[0], y[1]);
                o.y = pack_bf2(y[2], y[3]);
                *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(out) + (long)m * ldo + n) = o;
            } else if (EPI == EPI_F32) {
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + (long)m * ldo + n) =
                    make_float4(round_bf(y[0]), round_bf(y[1]), round_bf(y[2]), round_bf(y[3]));
            } else {
                float4* p = reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + (long)m * ldo + n);
                float4 x = *p;
                x.x += round_bf(y[0]) * gv[nt][0];
                x.y += round_bf(y[1]) * gv[nt][1];
                x.z += round_bf(y[2]) * gv[nt][2];
                x.w += round_bf(y[3]) * gv[nt][3];
                *p = x;
            }
        }
    }
}


// Persistent launch: one workgroup per CU.  Workgroups that share an XCD (same id mod 8) own one contiguous
// chunk of the tile grid - walked in groups of GM tile-rows so that A and W tiles are re-used out of that
// XCD's L2 - and take its tiles round by round, the same tiles at the same time as one workgroup per tile
// would, without a dispatch per tile (1.5-2 % on the K = 1536 shapes).
template <int EPI>
__global__ void __launch_bounds__(512, 2)
gemm256_kernel(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ W, long ldw,
               const bf16_t* __restrict__ bias, void* __restrict__ out, long ldo, const float* __restrict__ gate,
               int M, int N, int K, int tiles_m, int tiles_n, int GM, int wide_store) {
    const int nwg = tiles_m * tiles_n;
    const int xcd = blockIdx.x & 7, q = nwg >> 3, r = nwg & 7;
    const int first = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int count = q + (xcd < r ? 1 : 0);
    const int per_xcd = (gridDim.x + 7) >> 3;
    for (int t = blockIdx.x >> 3; t < count; t += per_xcd) {
        gemm256_tile<EPI>(A, lda, W, ldw, bias, out, ldo, gate, M, N, K, tiles_m, tiles_n, GM, wide_store, first + t);
        // the LDS buffers are staged again by the next tile: every wave's LDS reads have retired.  (Not
        // __syncthreads(): that also waits for this tile's global stores, which may drain under the next tile's DMA.)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
}

}  // namespace

// True when the shape fits this kernel (the caller falls back to the 128 x 128 kernel otherwise).
bool ovd_gemm256_applicable(long lda, long ldw, int M, int N, int K) {
    // one workgroup per CU.  From half a round of 256 x 256 tiles the deep pipeline beats the 128 x 128 kernel: at
    // M = 10 140 (240 tiles) 1 220 against 868 TFLOP/s for N = K = 1536 and 1 469 against 1 037 for K = 8960
    const long tiles = (long)ceil_div(M, TM) * (N / TN);
    static std::atomic<int> min_tiles_cache{INT32_MIN};
    const int min_tiles = ovd_env_int(min_tiles_cache, "OVD_GEMM256_MIN_TILES", 128);      // tuning aid
    return N % TN == 0 && K % KT == 0 && K >= 2 * KT && M >= 2 * TM && tiles >= min_tiles &&
           (long)M * lda * 2 < (1L << 32) && (long)N * ldw * 2 < (1L << 32);
}

int ovd_impl_gemm256_bf16(const bf16_t* A, long lda, const bf16_t* W, long ldw, const bf16_t* bias, void* out,
                          long ldo, const float* gate, int M, int N, int K, int epilogue, hipStream_t st) {
    OVD_REQUIRE(ovd_gemm256_applicable(lda, ldw, M, N, K), "gemm256: shape M=%d N=%d K=%d not supported", M, N, K);
    const int tiles_m = ceil_div(M, TM), tiles_n = N / TN;
    // one workgroup per CU, a multiple of 8 so that every XCD gets the same number
    static std::atomic<int> persist_cache{INT32_MIN};
    const int cus = ovd_env_int(persist_cache, "OVD_GEMM_PERSIST", 1) ? max(8, ovd_device_cus() / 8 * 8) : 1 << 30;
    dim3 grid(min(tiles_m * tiles_n, cus)), blk(512);
    const size_t lds = NPAIR * PAIR;
#define GEMM_CASE(E)                                                                                      \
    case E: {                                                                                             \
        static OvdLdsLimit lim;                                                                           \
        if (!lim.ensure(reinterpret_cast<const void*>(&gemm256_kernel<E>), lds))                          \
            OVD_FAIL("gemm256: cannot reserve %zu bytes of LDS", lds);                                     \
        hipLaunchKernelGGL((gemm256_kernel<E>), grid, blk, lds, st, A, lda, W, ldw, bias, out, ldo, gate, \
                           M, N, K, tiles_m, tiles_n, gm, wide);                                          \
    } break;
    // tile-rows per group of the tile walk (OVD_GEMM_GM overrides: tuning aid)
    static std::atomic<int> gm_cache{INT32_MIN};
    int gm = ovd_env_int(gm_cache, "OVD_GEMM_GM", 4);
    if (gm <= 0) gm = 4;
    // 16-byte output stores need 16-byte aligned rows
    const int wide = ((epilogue == EPI_F32 || epilogue == EPI_RESID ? ldo % 4 == 0 : ldo % 8 == 0) &&
                      reinterpret_cast<uintptr_t>(out) % 16 == 0) ? 1 : 0;
    switch (epilogue) { GEMM_CASE(EPI_BF16) GEMM_CASE(EPI_BF16_GELU) GEMM_CASE(EPI_F32) GEMM_CASE(EPI_RESID) }
#undef GEMM_CASE
    OVD_CHECK_LAUNCH("gemm256_bf16");
    return 0;
}
