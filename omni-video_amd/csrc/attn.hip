// Flash-style attention forward for head_dim 128, bf16 in / fp32 accumulate / bf16 out,
// non-causal, keys limited to k_len (tail masked).  Replaces the flash_attn_varlen_func
// call sites of the reference (omnivideo/modules/attention.py:96-130) for the shapes the
// DiT uses: self-attention (Lq = Lk = padded sequence, k_len = real tokens;
// model.py:144-149) and cross-attention over the padded context (model.py:174).
//
// Work split: one workgroup = 8 waves = 256 query rows of one (batch, head); each wave
// owns 32 query rows.  K/V tiles of 64 keys are staged HBM -> registers -> LDS (two
// buffers; the global loads for tile t+1 are issued before tile t's MFMAs and written
// to LDS after them, so HBM latency hides under compute).
//
// S^T = K Q^T is computed with the key on the MFMA row axis (v_mfma_f32_32x32x16_bf16),
// so every lane holds 32 scores of ONE query column: the row max / row sum are
// in-register reductions plus one cross-half exchange, and the exponentiated scores
// are already laid out as the B operand of O^T += V^T P^T (no LDS round trip for P).
// V^T fragments come from the row-major V tile through ds_read_b64_tr_b16.
// LDS images are XOR-swizzled per 16-byte chunk so both the b128 row reads of K and
// the transposed reads of V are bank-conflict free.
//
// Softmax runs in the exp2 domain with the scale folded into one FMA per score.  The
// running maximum is only raised when a tile's maximum exceeds it by more than
// RESCALE_THR (log2 units): probabilities then stay below 2^THR, which costs no
// relative precision in bf16/fp32, and the 64-register rescale of O is skipped on
// almost every tile.
#include "common.h"

namespace {

constexpr int HD = 128;        // head dim
constexpr int BKV = 64;        // keys per tile
constexpr int KV_TILE_BYTES = BKV * HD * 2;   // 16 KiB
constexpr float RESCALE_THR = 6.0f;

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((address_space(3))) bf16x4_t* lds_b4_ptr;

__device__ __forceinline__ int k_off(int row, int chunk) { return row * 256 + ((chunk ^ (row & 15)) << 4); }
__device__ __forceinline__ int v_off(int row, int chunk) { return row * 256 + ((chunk ^ ((row & 3) << 2)) << 4); }

__device__ __forceinline__ float max3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

// full-wave combine of the two 32-lane halves: every lane gets max(x[lane], x[lane^32])
__device__ __forceinline__ float half_max(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __builtin_fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_sum(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef float f32x2 __attribute__((ext_vector_type(2)));

// One K/V tile of 64 keys against this wave's 32 queries: S^T = K Q^T, online softmax,
// O^T += V^T P^T.  MASK = the tile holds keys at or beyond k_len (only ever the last tile).
template <bool MASK>
__device__ __forceinline__ void tile_step(f32x16 (&o_acc)[4], const bf16x8 (&qf)[8], float& m_c, float& l_run,
                                          const char* sk, const char* sv, int key0, int k_len, float scale_log2,
                                          int lane) {
    const int l31 = lane & 31, hi = lane >> 5;
    // ---- S^T[key][query] for 64 keys: two independent 32-key accumulators, 8 k-steps over d
    f32x16 s0, s1;
#pragma unroll
    for (int j = 0; j < 16; ++j) s0[j] = s1[j] = 0.f;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        const bf16x8 kf0 = *reinterpret_cast<const bf16x8*>(sk + k_off(l31, 2 * ks + hi));
        const bf16x8 kf1 = *reinterpret_cast<const bf16x8*>(sk + k_off(32 + l31, 2 * ks + hi));
        s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf0, qf[ks], s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf1, qf[ks], s1, 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    // s{kb}[r] is the score of key kb*32 + (r&3) + 8*(r>>2) + 4*hi against query l31
    if (MASK) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = key0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (key >= k_len) s0[r] = -1e30f;
            if (key + 32 >= k_len) s1[r] = -1e30f;
        }
    }

    // ---- online softmax (per query = per lane column; halves hold disjoint keys)
    float mx = max3(s0[0], s0[1], s0[2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) mx = max3(mx, s0[r], s0[r + 1]);
    mx = max3(mx, s0[15], s1[0]);
#pragma unroll
    for (int r = 1; r < 15; r += 2) mx = max3(mx, s1[r], s1[r + 1]);
    mx = __builtin_fmaxf(mx, s1[15]);
    mx = half_max(mx) * scale_log2;
    if (__any(mx > m_c + RESCALE_THR)) {
        const float m_new = __builtin_fmaxf(m_c, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_c - m_new);
        m_c = m_new;
        l_run *= alpha;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) o_acc[i][j] *= alpha;
    }
    float rs0 = 0.f, rs1 = 0.f;
    bf16x8 pf[4];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const f32x16& s = kb ? s1 : s0;
        float p[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) p[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], scale_log2, -m_c));
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            rs0 += p[r];
            rs1 += p[r + 1];
        }
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            uint32_t w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = pack_bf2(p[8 * h2 + 2 * j], p[8 * h2 + 2 * j + 1]);
            pf[kb * 2 + h2] = __builtin_bit_cast(bf16x8, make_uint4(w[0], w[1], w[2], w[3]));
        }
    }
    l_run += rs0 + rs1;

    // ---- O^T[d][query] += V^T[d][key] P^T[key][query]; 4 k-steps of 16 keys, 4 d-blocks of 32
    // element j of the P fragment of k-step st is key 16*st + 8*(j>>2) + 4*hi + (j&3)
    const int g = lane >> 4, i16 = lane & 15, qd = i16 >> 2, pp = i16 & 3;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        const int krow = 16 * st + 4 * (g >> 1) + qd;    // this lane's address row (first read)
        bf16x8 vf[4];
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            const int chunk = db * 4 + 2 * (g & 1) + (pp >> 1);
            const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                (lds_b4_ptr)(sv + v_off(krow, chunk) + 8 * (pp & 1)));
            const bf16x4_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                (lds_b4_ptr)(sv + v_off(krow + 8, chunk) + 8 * (pp & 1)));
            const uint2 lo2 = __builtin_bit_cast(uint2, lo), hi2 = __builtin_bit_cast(uint2, hi4);
            vf[db] = __builtin_bit_cast(bf16x8, make_uint4(lo2.x, lo2.y, hi2.x, hi2.y));
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int db = 0; db < 4; ++db)
            o_acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[db], pf[st], o_acc[db], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    }
}

template <int NW>
__device__ __forceinline__ void
attn_fwd_body(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ V,
                bf16_t* __restrict__ O, int Lq, int Lk, int k_len, long q_ld, long k_ld, long v_ld, long o_ld,
                long q_bs, long k_bs, long v_bs, long o_bs, float scale_log2) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // K[2] | V[2]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * (32 * NW) + wave * 32;

    const bf16_t* Qb = Q + b * q_bs + (long)head * HD;
    const bf16_t* Kb = K + b * k_bs + (long)head * HD;
    const bf16_t* Vb = V + b * v_bs + (long)head * HD;

    // Q fragments: B operand of S^T = K Q^T; lane (query l31, half hi) holds d = 16*ks + 8*hi + j
    bf16x8 qf[8];
    {
        int qr = q0 + l31;
        qr = qr < Lq ? qr : Lq - 1;
        const bf16_t* qp = Qb + (long)qr * q_ld + 8 * hi;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + 16 * ks);
    }

    f32x16 o_acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) o_acc[i][j] = 0.f;
    float m_c = -1e30f;     // running maximum, already multiplied by scale_log2
    float l_run = 0.f;      // this half's share of the row sum

    const int nt = (k_len + BKV - 1) / BKV;

    // staging: HBM -> LDS directly (16 B per lane, lane-linear image: one wave instruction
    // fills 4 tile rows); the XOR swizzle is applied to the SOURCE chunk so that the image
    // matches k_off / v_off.  Wave w stages rows 4*(i*NW + w) .. +3 for i < 16/NW.
    // Per-lane element offsets inside a tile are fixed; a tile only moves the uniform base.
    constexpr int NST = 16 / NW;
    int kofs[NST], vofs[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) {
        const int r = (i * NW + wave) * 4 + (lane >> 4), p = lane & 15;
        kofs[i] = r * (int)k_ld + ((p ^ (r & 15)) << 3);
        vofs[i] = r * (int)v_ld + ((p ^ ((r & 3) << 2)) << 3);
    }
#define STAGE_TILE(t, buf)                                                                              \
    do {                                                                                                \
        char* sk_ = smem + (buf) * KV_TILE_BYTES;                                                       \
        char* sv_ = smem + (2 + (buf)) * KV_TILE_BYTES;                                                 \
        if (((t) + 1) * BKV <= Lk) {                                                                    \
            const bf16_t* Kt = Kb + (long)(t) * BKV * k_ld;                                             \
            const bf16_t* Vt = Vb + (long)(t) * BKV * v_ld;                                             \
            _Pragma("unroll") for (int i = 0; i < NST; ++i) {                                           \
                const int r4 = (i * NW + wave) * 4;                                                     \
                __builtin_amdgcn_global_load_lds(Kt + kofs[i], (lds_ptr)(sk_ + r4 * 256), 16, 0, 0);    \
                __builtin_amdgcn_global_load_lds(Vt + vofs[i], (lds_ptr)(sv_ + r4 * 256), 16, 0, 0);    \
            }                                                                                           \
        } else { /* tile reaches past the last row of K/V: clamp the row (those keys are masked) */     \
            _Pragma("unroll") for (int i = 0; i < NST; ++i) {                                           \
                const int r4 = (i * NW + wave) * 4, r = r4 + (lane >> 4), p = lane & 15;                \
                int gr = (t) * BKV + r;                                                                 \
                gr = gr < Lk ? gr : Lk - 1;                                                             \
                __builtin_amdgcn_global_load_lds(Kb + (long)gr * k_ld + ((p ^ (r & 15)) << 3),          \
                                                 (lds_ptr)(sk_ + r4 * 256), 16, 0, 0);                  \
                __builtin_amdgcn_global_load_lds(Vb + (long)gr * v_ld + ((p ^ ((r & 3) << 2)) << 3),    \
                                                 (lds_ptr)(sv_ + r4 * 256), 16, 0, 0);                  \
            }                                                                                           \
        }                                                                                               \
    } while (0)

    STAGE_TILE(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const bool partial = nt * BKV > k_len;
    const int n_plain = partial ? nt - 1 : nt;
    for (int t = 0; t < n_plain; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) STAGE_TILE(t + 1, buf ^ 1);
        tile_step<false>(o_acc, qf, m_c, l_run, smem + buf * KV_TILE_BYTES, smem + (2 + buf) * KV_TILE_BYTES,
                         t * BKV, k_len, scale_log2, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (partial) {
        const int buf = (nt - 1) & 1;
        tile_step<true>(o_acc, qf, m_c, l_run, smem + buf * KV_TILE_BYTES, smem + (2 + buf) * KV_TILE_BYTES,
                        (nt - 1) * BKV, k_len, scale_log2, lane);
    }

#undef STAGE_TILE

    // ---- normalise and store: lane (query l31, half hi) holds d = db*32 + 8*(r>>2) + 4*hi + (r&3)
    const float inv = 1.0f / half_sum(l_run);
    const int qr = q0 + l31;
    if (qr < Lq) {
        bf16_t* op = O + b * o_bs + (long)qr * o_ld + (long)head * HD + 4 * hi;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                uint2 w;
                w.x = pack_bf2(o_acc[db][4 * g4 + 0] * inv, o_acc[db][4 * g4 + 1] * inv);
                w.y = pack_bf2(o_acc[db][4 * g4 + 2] * inv, o_acc[db][4 * g4 + 3] * inv);
                *reinterpret_cast<uint2*>(op + db * 32 + 8 * g4) = w;
            }
    }
}

#define ATTN_KERNEL(NW)                                                                                  \
    __global__ void __launch_bounds__(64 * NW, 2) attn_fwd_kernel_w##NW(                                 \
        const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ V,       \
        bf16_t* __restrict__ O, int Lq, int Lk, int k_len, long q_ld, long k_ld, long v_ld, long o_ld,   \
        long q_bs, long k_bs, long v_bs, long o_bs, float scale_log2) {                                  \
        attn_fwd_body<NW>(Q, K, V, O, Lq, Lk, k_len, q_ld, k_ld, v_ld, o_ld, q_bs, k_bs, v_bs, o_bs,     \
                          scale_log2);                                                                   \
    }
ATTN_KERNEL(4)
ATTN_KERNEL(8)
#undef ATTN_KERNEL

}  // namespace

// q [B,Lq,N,128], k/v [B,Lk,N,128], out [B,Lq,N,128]; *_ld = elements between
// consecutive tokens, *_bs = elements between batch entries (so the fused QKV
// projection buffer can be addressed in place).  Keys at index >= k_len are masked.
int ovd_impl_attn_fwd(const bf16_t* q, const bf16_t* k, const bf16_t* v, bf16_t* out, int B, int Lq, int Lk,
                      int N, int D, int k_len, long q_ld, long k_ld, long v_ld, long o_ld, long q_bs, long k_bs,
                      long v_bs, long o_bs, float softmax_scale, hipStream_t st) {
    OVD_REQUIRE(D == HD, "attention: head_dim %d not supported (only 128)", D);
    OVD_REQUIRE(B > 0 && N > 0 && Lq > 0 && Lk > 0, "attention: empty problem B=%d N=%d Lq=%d Lk=%d", B, N, Lq, Lk);
    OVD_REQUIRE(k_len > 0 && k_len <= Lk, "attention: k_len=%d outside (0, Lk=%d]", k_len, Lk);
    OVD_REQUIRE(q_ld % 8 == 0 && k_ld % 8 == 0 && v_ld % 8 == 0 && o_ld % 4 == 0,
                "attention: token strides must keep 16-byte alignment");
    OVD_REQUIRE(q_ld >= (long)N * D && k_ld >= (long)N * D && v_ld >= (long)N * D && o_ld >= (long)N * D,
                "attention: token stride shorter than N*D");
    OVD_REQUIRE(softmax_scale > 0.f, "attention: softmax_scale must be positive");
    const size_t lds = 4 * KV_TILE_BYTES;
    const float scale_log2 = softmax_scale * 1.4426950408889634f;
    static int variant = -1;
    if (variant < 0) {
        const char* e = getenv("OVD_ATTN_WAVES");
        variant = (e && atoi(e) == 8) ? 8 : 4;
    }
#define ATTN_CASE(NW)                                                                                      \
    {                                                                                                      \
        static bool attr_set = false;                                                                      \
        if (!attr_set) {                                                                                   \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel_w##NW),                   \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)   \
                OVD_FAIL("attention: cannot reserve %zu bytes of LDS", lds);                                \
            attr_set = true;                                                                               \
        }                                                                                                  \
        dim3 grid(ceil_div(Lq, 32 * NW), N, B), blk(64 * NW);                                              \
        hipLaunchKernelGGL(attn_fwd_kernel_w##NW, grid, blk, lds, st, q, k, v, out, Lq, Lk, k_len, q_ld,   \
                           k_ld, v_ld, o_ld, q_bs, k_bs, v_bs, o_bs, scale_log2);                          \
    }
    if (variant == 8) ATTN_CASE(8) else ATTN_CASE(4)
#undef ATTN_CASE
    OVD_CHECK_LAUNCH("attn_fwd");
    return 0;
}
