// bf16 x bf16 -> fp32-accumulate GEMM on the matrix cores with fused epilogues.
//   out[m][n] = epilogue( sum_k A[m][k] * W[n][k] + bias[n] )
// A is the activation matrix [M,K] (row stride lda), W the nn.Linear weight [N,K]
// (row stride ldw): both operands are K-contiguous, which is the MFMA fragment order.
//
// Tile 128x128x64, 4 waves (2x2), each wave 64x64 = 4x4 tiles of v_mfma_f32_16x16x32_bf16.
// Operands go HBM -> LDS with 16-byte global_load_lds (no VGPR round trip), two LDS
// buffers so the next K-tile streams in under the current tile's MFMAs.  LDS rows are
// 128 B; the 16-B chunk index is XOR-swizzled with (row>>1)&7 on the SOURCE address
// (the LDS image of a global_load_lds is lane-linear) and again on the ds_read_b128
// address, which makes every 16-lane read group hit 16 distinct bank slots.
// The product is computed transposed (weight rows on the MFMA M axis) so each lane
// ends up with 4 consecutive output columns of one row: 8-/16-byte epilogue stores.
// Workgroups are dealt to XCDs in contiguous chunks of the tile grid (each XCD has a
// private L2) and walk the chunk in groups of 8 tile-rows for operand reuse.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;   // 16 KiB per operand tile

enum Epi { EPI_BF16 = 0, EPI_BF16_GELU = 1, EPI_F32 = 2, EPI_RESID = 3 };

typedef __attribute__((address_space(3))) void* lds_ptr;

__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ g, long ld, int row0, int nrows,
                                           int k0, char* lds_tile, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r8 = (i * 4 + wave) * 8;
        const int r = r8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        int gr = row0 + r;
        gr = gr < nrows ? gr : nrows - 1;
        const bf16_t* src = g + (long)gr * ld + k0 + c * 8;
        __builtin_amdgcn_global_load_lds(src, (lds_ptr)(lds_tile + r8 * (BK * 2)), 16, 0, 0);
    }
}

__device__ __forceinline__ bf16x8 read_frag(const char* lds_tile, int r, int chunk) {
    const int off = r * (BK * 2) + ((chunk ^ ((r >> 1) & 7)) << 4);
    return *reinterpret_cast<const bf16x8*>(lds_tile + off);
}

template <int EPI>
__global__ void __launch_bounds__(256) gemm_bf16_kernel(const bf16_t* __restrict__ A, long lda,
                                                        const bf16_t* __restrict__ W, long ldw,
                                                        const bf16_t* __restrict__ bias, void* __restrict__ out,
                                                        long ldo, const float* __restrict__ gate, int M, int N,
                                                        int K, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    // XCD-aware bijective remap of the linear block id, then grouped (8 tile-rows) order
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    constexpr int GM = 8;
    const int per_group = GM * tiles_n;
    const int grp = bid / per_group;
    const int first_m = grp * GM;
    const int gsz = min(tiles_m - first_m, GM);
    const int tm = first_m + (bid % per_group) % gsz;
    const int tn = (bid % per_group) / gsz;
    const int row0 = tm * BM, col0 = tn * BN;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = K / BK;
    stage_tile(A, lda, row0, M, 0, smem, wave, lane);
    stage_tile(W, ldw, col0, N, 0, smem + 2 * TILE_BYTES, wave, lane);

    const int fr = lane & 15, fq = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) {
            stage_tile(A, lda, row0, M, (kt + 1) * BK, smem + (buf ^ 1) * TILE_BYTES, wave, lane);
            stage_tile(W, ldw, col0, N, (kt + 1) * BK, smem + (2 + (buf ^ 1)) * TILE_BYTES, wave, lane);
        }
        const char* sA = smem + buf * TILE_BYTES;
        const char* sW = smem + (2 + buf) * TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 wf[4], af[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                wf[t] = read_frag(sW, wc * 64 + t * 16 + fr, ks * 4 + fq);
                af[t] = read_frag(sA, wr * 64 + t * 16 + fr, ks * 4 + fq);
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], af[mt], acc[nt][mt], 0, 0, 0);
        }
    }

    // epilogue: lane holds out[m][n..n+3], m = row0 + wr*64 + mt*16 + fr, n = col0 + wc*64 + nt*16 + fq*4
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int n = col0 + wc * 64 + nt * 16 + fq * 4;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (bias) {
            const uint2 bp = *reinterpret_cast<const uint2*>(bias + n);
            bv[0] = bf2f(bp.x & 0xffff);
            bv[1] = bf2f(bp.x >> 16);
            bv[2] = bf2f(bp.y & 0xffff);
            bv[3] = bf2f(bp.y >> 16);
        }
        float gv[4] = {1.f, 1.f, 1.f, 1.f};
        if (EPI == EPI_RESID && gate) {
            const float4 g4 = *reinterpret_cast<const float4*>(gate + n);
            gv[0] = g4.x; gv[1] = g4.y; gv[2] = g4.z; gv[3] = g4.w;
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int m = row0 + wr * 64 + mt * 16 + fr;
            if (m >= M) continue;
            float y[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = acc[nt][mt][j] + bv[j];
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
                x.x += round_bf(y[0]) * gv[0];
                x.y += round_bf(y[1]) * gv[1];
                x.z += round_bf(y[2]) * gv[2];
                x.w += round_bf(y[3]) * gv[3];
                *p = x;
            }
        }
    }
}

}  // namespace

// epilogue: 0 = bf16 out, 1 = bf16 out with tanh-GELU (applied to the bf16-rounded
// pre-activation), 2 = fp32 out holding the bf16-rounded value, 3 = fp32 in-place
// residual  out += round_bf16(acc + bias) * gate[n]  (gate may be null = 1).
// Reference call sites: nn.Linear under bf16 autocast throughout
// omnivideo/modules/model.py (q/k/v/o :118-121, ffn :211-213, text_embedding :386-388)
// and the gated residual adds of WanAttentionBlock.forward (:245-254).
bool ovd_gemm256_applicable(long lda, long ldw, int M, int N, int K);
int ovd_impl_gemm256_bf16(const bf16_t* A, long lda, const bf16_t* W, long ldw, const bf16_t* bias, void* out,
                          long ldo, const float* gate, int M, int N, int K, int epilogue, hipStream_t st);

int ovd_impl_gemm_bf16(const bf16_t* A, long lda, const bf16_t* W, long ldw, const bf16_t* bias, void* out,
                       long ldo, const float* gate, int M, int N, int K, int epilogue, hipStream_t st) {
    OVD_REQUIRE(K > 0 && K % BK == 0, "gemm: K=%d must be a positive multiple of %d", K, BK);
    OVD_REQUIRE(N > 0 && N % BN == 0, "gemm: N=%d must be a positive multiple of %d", N, BN);
    OVD_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && ldo % 4 == 0, "gemm: row strides must keep 16-byte alignment");
    OVD_REQUIRE(lda >= K && ldw >= K && ldo >= N, "gemm: row stride shorter than the row");
    OVD_REQUIRE(epilogue >= 0 && epilogue <= 3, "gemm: unknown epilogue %d", epilogue);
    OVD_REQUIRE(gate == nullptr || epilogue == EPI_RESID, "gemm: gate only applies to the residual epilogue");
    if (M == 0) return 0;
    // large shapes go to the 256 x 256 deep-pipelined kernel (OVD_GEMM256=0 disables: tuning aid)
    static std::atomic<int> use256_cache{INT32_MIN};
    const int use256 = ovd_env_int(use256_cache, "OVD_GEMM256", 1);
    if (use256 && ovd_gemm256_applicable(lda, ldw, M, N, K))
        return ovd_impl_gemm256_bf16(A, lda, W, ldw, bias, out, ldo, gate, M, N, K, epilogue, st);
    const int tiles_m = ceil_div(M, BM), tiles_n = N / BN;
    dim3 grid(tiles_m * tiles_n), blk(256);
    const size_t lds = 4 * TILE_BYTES;
#define GEMM_CASE(E)                                                                                      \
    case E: {                                                                                             \
        static OvdLdsLimit lim;                                                                           \
        if (!lim.ensure(reinterpret_cast<const void*>(&gemm_bf16_kernel<E>), lds))                        \
            OVD_FAIL("gemm: cannot reserve %zu bytes of LDS", lds);                                        \
        hipLaunchKernelGGL((gemm_bf16_kernel<E>), grid, blk, lds, st, A, lda, W, ldw, bias, out, ldo, gate, \
                           M, N, K, tiles_m, tiles_n);                                                    \
    } break;
    switch (epilogue) {
        GEMM_CASE(EPI_BF16) GEMM_CASE(EPI_BF16_GELU) GEMM_CASE(EPI_F32) GEMM_CASE(EPI_RESID)
    }
#undef GEMM_CASE
    OVD_CHECK_LAUNCH("gemm_bf16");
    return 0;
}
