// Large-shape bf16 GEMM with the epilogues of gemm.hip:
//   out[m][n] = epilogue( sum_k A[m][k] * W[n][k] + bias[n] ),  A [M,K], W [N,K] (nn.Linear).
//
// One workgroup = 8 waves = a 256 x 256 output tile; wave (wr, wc) = (w >> 2, w & 3) owns
// 128 rows x 64 columns = 8 x 4 tiles of v_mfma_f32_16x16x32_bf16 (128 accumulator
// registers), computed transposed (weight rows on the MFMA M axis) so that a lane ends up
// with 4 consecutive output columns of one row.
//
// Operands stream HBM -> LDS with 16-byte global_load_lds; K-tiles of 32 are staged in PAIRS
// (K % 64 == 0).  LDS rows are 128 B (64 k-values), so every DMA instruction
// fetches 8 whole 128-byte lines; the ring holds 2 pairs of 64 KiB (A part 256 rows x 128 B |
// W part 256 rows x 128 B; one workgroup per CU).  16-byte chunk c of row r sits at chunk position
// c ^ ((r >> 1) & 7): every 16-lane group of a ds_read_b128 fragment read (either half of the row)
// hits 16 distinct bank slots.  The swizzle is applied on the SOURCE address of the DMA (its LDS
// image is lane-linear) and again on the read address.  The even K-tile of a pair puts the whole
// next pair in flight (8 DMA instructions per wave) into the other slot, the odd one waits for it.
// (A ring of 4 single K-tiles with 64-byte rows, half a line per row and DMA instruction, ran 1-4 %
// slower.)
//
// A K-tile is one phase per wave: 12 fragment reads, then 32 MFMAs.  The two waves that share a
// SIMD (w and w + 4) run half a period apart: waves 4-7 execute one extra s_barrier before the loop
// (waves 0-3 one after it), so that one of them issues its LDS reads and DMA while the other one
// feeds the matrix pipe.
//
// What bounds the loop (measured): the operand stream into the CU.  At 128 FLOP per staged byte the
// kernel moves ~18 B/clk/CU at 1.25 PFLOP/s; removing half of the DMA (wrong results, timing only)
// ran at 1.59 PFLOP/s, while a deeper ring (5 tiles), DMA issued from inside the MFMA phase, and one
// operand staged through registers + ds_write_b128 all ran within 2 % of the plain loop.
//
// Hazards the schedule is built around:
//   * a wave's fragment reads are complete (lgkmcnt(0)) BEFORE the barrier that ends its read
//     phase, so the slot of the previous pair can be re-staged right behind that barrier;
//   * raw s_barrier + explicit waits only inside the loop: a __syncthreads() would drain the DMA queue;
//   * no ordinary global load while DMA is in flight (bias / gate are read in the epilogue).
#include "common.h"

namespace {

constexpr int TM = 256, TN = 256, TK = 32;
constexpr int PAIR = 2 * 2 * 256 * TK * 2;   // two K-tiles staged together: A part 32 KiB | W part 32 KiB
constexpr int NPAIR = 2;                     // pairs in the ring

enum Epi { EPI_BF16 = 0, EPI_BF16_GELU = 1, EPI_F32 = 2, EPI_RESID = 3 };

typedef __attribute__((address_space(3))) void* lds_ptr;

__device__ __forceinline__ bf16x8 lds_frag(const char* p) { return *reinterpret_cast<const bf16x8*>(p); }

struct Src4 {
    const char* base;
    unsigned off[4];       // this lane's byte offsets (clamped row * ld + swizzled chunk) for its four rows
};

__device__ __forceinline__ void stage_part4(const Src4& s, long kbyte, char* dst_wave) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_global_load_lds(s.base + kbyte + s.off[i], (lds_ptr)(dst_wave + i * 8192), 16, 0, 0);
}

// One K-tile (half of the pair in `slot`; lane_off selects the half).  STAGE_NEXT: put the next pair in
// flight into the other slot; WAIT: retire this wave's share of it.
template <bool STAGE_NEXT, bool WAIT>
__device__ __forceinline__ void k_tile2(f32x4 (&acc)[4][8], char* smem, int slot, long kb_next, const Src4& sa,
                                        const Src4& sw, int lane_off, int a_rd, int w_rd, int wave) {
    const char* st = smem + slot * PAIR;
    bf16x8 wf[4], af[8];
#pragma unroll
    for (int t = 0; t < 4; ++t) wf[t] = lds_frag(st + PAIR / 2 + w_rd + t * 2048 + lane_off);
#pragma unroll
    for (int t = 0; t < 8; ++t) af[t] = lds_frag(st + a_rd + t * 2048 + lane_off);
    if (STAGE_NEXT) {
        char* nx = smem + (slot ^ 1) * PAIR + wave * 1024;
        stage_part4(sw, kb_next, nx + PAIR / 2);
        stage_part4(sa, kb_next, nx);
    }
    if (WAIT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
            acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], af[mt], acc[nt][mt], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
}

template <int EPI>
__global__ void __launch_bounds__(512, 2)
gemm256_kernel(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ W, long ldw,
               const bf16_t* __restrict__ bias, void* __restrict__ out, long ldo, const float* __restrict__ gate,
               int M, int N, int K, int tiles_m, int tiles_n, int GM, int wide_store) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    // blocks that share an XCD (same id mod 8) get one contiguous chunk of the tile grid,
    // walked in groups of 4 tile-rows so that A and W tiles are re-used out of that XCD's L2
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

    // staging: wave w, instruction i covers part rows (i*8 + w)*8 .. +7, 8 lanes per 128-byte row
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
    // fragment reads: lane (r = lane & 15, q = lane >> 4) reads chunk 4*half + q of row r of a 16-row block
    const int lo0 = (lane & 15) * 128 + (((lane >> 4) ^ (((lane & 15) >> 1) & 7)) << 4), lo1 = lo0 ^ 64;
    const int a_rd = wr * 16384, w_rd = wc * 8192;

    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    char* const wdst = smem + wave * 1024;
    const int np = K / 64;      // >= 2 (checked on the host)
    stage_part4(sw, 0, wdst + PAIR / 2);
    stage_part4(sa, 0, wdst);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();            // second half of the waves: half a period behind
    int j = 0;
    for (; j < np - 1; ++j) {
        k_tile2<true, false>(acc, smem, j & 1, (long)(j + 1) * 128, sa, sw, lo0, a_rd, w_rd, wave);
        k_tile2<false, true>(acc, smem, j & 1, 0, sa, sw, lo1, a_rd, w_rd, wave);
    }
    k_tile2<false, false>(acc, smem, j & 1, 0, sa, sw, lo0, a_rd, w_rd, wave);
    k_tile2<false, false>(acc, smem, j & 1, 0, sa, sw, lo1, a_rd, w_rd, wave);
    if (wr == 0) __builtin_amdgcn_s_barrier();            // balances the extra barrier above

    // epilogue: lane holds out[m][n..n+3], m = row0 + wr*128 + mt*16 + fr, n = col0 + wc*64 + nt*16 + fq*4.
    // Rows outermost: the four stores of a lane for one row are the four 32-byte quarters of
    // one 128-byte line (64 bf16 columns), issued back to back so that they merge on the way out.
    const int fr = lane & 15, fq = lane >> 4;
    float bv[4][4], gv[4][4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int n = col0 + wc * 64 + nt * 16 + fq * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bv[nt][j] = 0.f;
            gv[nt][j] = 1.f;
        }
        if (bias) {
            const uint2 bp = *reinterpret_cast<const uint2*>(bias + n);
            bv[nt][0] = bf2f(bp.x & 0xffff);
            bv[nt][1] = bf2f(bp.x >> 16);
            bv[nt][2] = bf2f(bp.y & 0xffff);
            bv[nt][3] = bf2f(bp.y >> 16);
        }
        if (EPI == EPI_RESID && gate) {
            const float4 g4 = *reinterpret_cast<const float4*>(gate + n);
            gv[nt][0] = g4.x; gv[nt][1] = g4.y; gv[nt][2] = g4.z; gv[nt][3] = g4.w;
        }
    }
    if ((EPI == EPI_BF16 || EPI == EPI_BF16_GELU) && wide_store) {
        // bf16 output through LDS: the ring is free (every wave is past the last barrier and every
        // DMA has landed), each wave owns 16 KiB of it for its 128 x 64 tile = 128 rows of 128 B.
        // Written in the accumulator layout (8 B per lane), read back 16 B per lane so that one
        // store instruction covers 8 full 128-byte lines.  16-byte chunk c of row r sits at chunk
        // c ^ ((r >> 1) & 7): the writes are 2 lanes per bank (the minimum for 512 B), the reads
        // conflict-free.
        char* ep = smem + wave * 16384;
        const int rr = lane >> 3, cc = lane & 7;
        bf16_t* ob = reinterpret_cast<bf16_t*>(out) + (long)(row0 + wr * 128) * ldo + col0 + wc * 64 + cc * 8;
        // two halves of 64 rows: the stores of the first half are in flight while the second half is
        // being computed (the GELU is ~40 VALU cycles per element)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
            for (int mt = 4 * hf; mt < 4 * hf + 4; ++mt)
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
                    const int r = mt * 16 + fr, c = nt * 2 + (fq >> 1);
                    *reinterpret_cast<uint2*>(ep + r * 128 + ((c ^ ((r >> 1) & 7)) << 4) + (fq & 1) * 8) = o;
                }
#pragma unroll
            for (int i = 8 * hf; i < 8 * hf + 8; ++i) {
                const int r = i * 8 + rr;
                const uint4 v = *reinterpret_cast<const uint4*>(ep + r * 128 + ((cc ^ ((r >> 1) & 7)) << 4));
                if (row0 + wr * 128 + r < M) *reinterpret_cast<uint4*>(ob + (long)r * ldo) = v;
            }
        }
        return;
    }
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
        const int m = row0 + wr * 128 + mt * 16 + fr;
        if (m >= M) continue;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n = col0 + wc * 64 + nt * 16 + fq * 4;
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

}  // namespace

// True when the shape fits this kernel (the caller falls back to the 128 x 128 kernel otherwise).
bool ovd_gemm256_applicable(long lda, long ldw, int M, int N, int K) {
    // one workgroup per CU: worth it from about two rounds of 256 x 256 tiles over the 256 CUs
    const long tiles = (long)ceil_div(M, TM) * (N / TN);
    return N % TN == 0 && K % (2 * TK) == 0 && K >= 4 * TK && M >= 2 * TM && tiles >= 512 &&
           (long)M * lda * 2 < (1L << 32) && (long)N * ldw * 2 < (1L << 32);
}

int ovd_impl_gemm256_bf16(const bf16_t* A, long lda, const bf16_t* W, long ldw, const bf16_t* bias, void* out,
                          long ldo, const float* gate, int M, int N, int K, int epilogue, hipStream_t st) {
    OVD_REQUIRE(ovd_gemm256_applicable(lda, ldw, M, N, K), "gemm256: shape M=%d N=%d K=%d not supported", M, N, K);
    const int tiles_m = ceil_div(M, TM), tiles_n = N / TN;
    dim3 grid(tiles_m * tiles_n), blk(512);
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
    const int wide = (ldo % 8 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0) ? 1 : 0;
    switch (epilogue) { GEMM_CASE(EPI_BF16) GEMM_CASE(EPI_BF16_GELU) GEMM_CASE(EPI_F32) GEMM_CASE(EPI_RESID) }
#undef GEMM_CASE
    OVD_CHECK_LAUNCH("gemm256_bf16");
    return 0;
}
