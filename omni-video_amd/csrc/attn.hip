// Flash-style attention forward for head_dim 128, bf16 in / fp32 accumulate / bf16 out,
// non-causal, keys limited to k_len (tail masked).  Replaces the flash_attn_varlen_func
// call sites of the reference (omnivideo/modules/attention.py:96-130) for the shapes the
// DiT uses: self-attention (Lq = Lk = padded sequence, k_len = real tokens;
// model.py:144-149) and cross-attention over the padded context (model.py:174).
//
// Work split: one workgroup = 8 (or 4) waves, each wave owns 32 query rows of one
// (batch, head).  K/V tiles of 64 keys go HBM -> LDS directly (global_load_lds, 16 B per
// lane; the LDS image of such a load is lane-linear, so the XOR swizzle is applied to the
// SOURCE chunk), two buffers per operand, each tile in flight for a whole step.
//
// S^T = K Q^T is computed with the key on the MFMA row axis (v_mfma_f32_32x32x16_bf16),
// so every lane holds 32 scores of ONE query column: the row max / row sum are
// in-register reductions plus one cross-half exchange, and the exponentiated scores
// are already laid out as the B operand of O^T += V^T P^T (no LDS round trip for P).
// V^T fragments come from the row-major V tile through ds_read_b64_tr_b16.
// LDS images are XOR-swizzled per 16-byte chunk so both the b128 row reads of K and
// the transposed reads of V are bank-conflict free.
//
// The loop is a two-tile software pipeline: the MFMAs of S^T(t+1) issue beside the
// exp / sum / pack of tile t, the P V MFMAs of tile t beside the row maximum of tile t+1
// and the test for a rescale; scheduling fences keep each k-step's reads, MFMAs and
// vector slice together.  Softmax runs in the exp2 domain with the scale folded into one
// FMA per score.  The running maximum is only raised when a tile's maximum exceeds it by
// more than RESCALE_THR (log2 units): probabilities then stay below 2^THR, which costs no
// relative precision in bf16/fp32, and the 64-register rescale of O is skipped on almost
// every tile.
//
// Output: the normalised O^T accumulators are written as bf16 into the wave's (by then
// free) LDS region with a per-row chunk swizzle and read back row-major, so each lane
// stores 16 bytes and a store instruction covers whole 256-byte head rows instead of
// 8-byte pieces in accumulator order.
//
// Reduced-precision form (template parameter I8; the reference has none): Q K^T on int8 operands with
// v_mfma_i32_32x32x32_i8 - half the matrix time of the bf16 product; q carries a scale per token and
// head, k a scale per 64-key tile and head (ovd_impl_quant_heads_i8), both folded into the exp2
// multiplier, so the vector work per score is unchanged.  K tiles are 64 rows of 128 bytes.  The int32
// accumulators start at the bit pattern of 1.5 * 2^23, which makes the finished accumulator the float
// 12582912 + score: no conversion instruction per score.  Softmax, P (bf16) and P V are shared.
//
// Tail balancing: all workgroups of a launch take the same time, so a grid that is not a
// multiple of the CU count ends in a partly filled round.  When attn_split_plan() finds
// such a tail, the trailing query blocks of every XCD are split over the two halves of the
// key range (those workgroups are dispatched last); each half writes its unnormalised O,
// running maximum and sum to a caller-owned workspace and attn_merge_kernel combines the
// two in the exp2 domain.  Launches with few workgroups, or with at most 1024 keys
// (cross-attention), use 4-wave workgroups so more of them fit the chip.
#include "common.h"

namespace {

constexpr int HD = 128;        // head dim
constexpr int BKV = 64;        // keys per tile
constexpr int KV_TILE_BYTES = BKV * HD * 2;   // 16 KiB
constexpr float RESCALE_THR = 6.0f;

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((address_space(3))) bf16x4_t* lds_b4_ptr;

__device__ __forceinline__ int k_off(int row, int chunk) { return row * 256 + ((chunk ^ (row & 15)) << 4); }
// int8 K tile: rows of 128 bytes (8 chunks); chunk c of row r sits at c ^ ((r >> 1) & 7), which spreads every
// 16-lane group of a ds_read_b128 fragment read over 16 distinct 16-byte bank slots
__device__ __forceinline__ int k8_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
// The int8 score accumulators start at the bit pattern of 1.5 * 2^23: adding an integer |s| < 2^22 to that
// pattern gives the bit pattern of the float 12582912 + s, so the finished int32 accumulator IS a float and
// no conversion instruction is spent per score (sum of 128 products of int8: |s| <= 128 * 127^2 < 2^22).
constexpr int I8_MAGIC_BITS = 0x4B400000;
constexpr float I8_MAGIC = 12582912.0f;
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

// ---------------------------------------------------------------------------------------
// Two-tile software pipeline: while the matrix pipe computes S^T of tile t+1, the vector
// pipe finishes the softmax of tile t; the row maximum of tile t+1 is folded in between
// the P V products of tile t.  Scores of two tiles are live (named pairs, swapped by call
// order).  LDS holds K(t+1) and V(t) during step t; K(t+2) and V(t+1) stream in under it.

// raw (unscaled) maximum of this lane's 32 scores
__device__ __forceinline__ float lane_max32(const f32x16& a, const f32x16& b) {
    float mx = max3(a[0], a[1], a[2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) mx = max3(mx, a[r], a[r + 1]);
    mx = max3(mx, a[15], b[0]);
#pragma unroll
    for (int r = 1; r < 15; r += 2) mx = max3(mx, b[r], b[r + 1]);
    return __builtin_fmaxf(mx, b[15]);
}

// Last-tile fix-ups: keys at or beyond k_len are masked; key k_len - 1 gets tail_bias added
// to its raw score (= log2(c) / scale_log2 when that key stands for c identical keys, so that
// its probability, its share of the row sum and of P V all count c times; 0 otherwise).
template <bool MASK>
__device__ __forceinline__ void mask_scores(f32x16& s0, f32x16& s1, int key0, int k_len, float tail_bias, int hi) {
    if (MASK) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = key0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (key == k_len - 1) s0[r] += tail_bias;
            if (key + 32 == k_len - 1) s1[r] += tail_bias;
            if (key >= k_len) s0[r] = -1e30f;
            if (key + 32 >= k_len) s1[r] = -1e30f;
        }
    }
}

// Streams one operand (K or V) HBM -> LDS, one 64-row tile per call: per-lane byte offset of
// this lane's row in the next tile (pass 0), advanced by a tile per call and clamped to the
// operand's last row (tiles that reach past it only feed masked keys).
struct Stream {
    const char* base;    // (batch, head) base of the operand
    unsigned row;        // byte offset of this lane's row of the next tile to stage (pass 0)
    unsigned chunk;      // byte offset of this lane's (swizzled) 16-byte chunk inside the row
    unsigned last;       // byte offset of the operand's last row
    unsigned pass;       // bytes between the rows of consecutive passes (4 * NW rows)
    unsigned tile;       // bytes per 64-row tile
};

template <int NW>
__device__ __forceinline__ void stage_next(Stream& s, char* dst_wave) {
#pragma unroll
    for (int i = 0; i < 16 / NW; ++i) {
        const unsigned ro = min(s.row + i * s.pass, s.last);
        __builtin_amdgcn_global_load_lds(s.base + ro + s.chunk, (lds_ptr)(dst_wave + i * NW * 1024), 16, 0, 0);
    }
    s.row += s.tile;
}

// int8 K tile: 64 rows x 128 bytes; one wave instruction covers 8 rows
template <int NW>
__device__ __forceinline__ void stage_next_k8(Stream& s, char* dst_wave) {
#pragma unroll
    for (int i = 0; i < 8 / NW; ++i) {
        const unsigned ro = min(s.row + i * s.pass, s.last);
        __builtin_amdgcn_global_load_lds(s.base + ro + s.chunk, (lds_ptr)(dst_wave + i * NW * 1024), 16, 0, 0);
    }
    s.row += s.tile;
}

// c0/c1: scores of the current tile; mx = their row maximum (scaled, whole wave), need = some
// row's maximum outgrew the running one.  n0/n1 receive the scores of the next tile (HAS_NEXT),
// read from sk_next; V of the current tile is in sv_cur.  The K tile after next and the next
// V tile are put in flight from inside the first two k-steps.
// I8: Q and K are int8 with the scores' scale folded into sc_cur / sc_next (scale_log2 * the query's scale *
// the key tile's scale; bf16: both are scale_log2); qf then holds 4 fragments of 16 bytes in its first 16
// registers, the K tile has 128-byte rows, and a score register holds the float 12582912 + s.
template <int NW, bool I8, bool HAS_NEXT, bool MASK_NEXT>
__device__ __forceinline__ void pipe_step(f32x16 (&o_acc)[4], const bf16x8 (&qf)[8], float& m_c, float& l_run,
                                          const f32x16& c0, const f32x16& c1, f32x16& n0, f32x16& n1,
                                          float& mx, bool& need, const char* sk_next, const char* sv_cur,
                                          Stream& ks_, char* k_dst, Stream& vs_, char* v_dst, int key0_next,
                                          int k_len, float tail_bias, float sc_cur, float sc_next, int lane) {
    const int l31 = lane & 31, hi = lane >> 5;
    if (need) {
        const float m_new = __builtin_fmaxf(m_c, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_c - m_new);
        m_c = m_new;
        l_run *= alpha;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) o_acc[i][j] *= alpha;
    }
    // ---- S^T of the next tile (MFMA) beside exp / sum / pack of the current one (VALU)
    if (HAS_NEXT) {
#pragma unroll
        for (int j = 0; j < 16; ++j) n0[j] = n1[j] = I8 ? __int_as_float(I8_MAGIC_BITS) : 0.f;
    }
    // p = exp2(score * scale_log2 - m_c); with int8 scores the register holds 12582912 + score
    const float scale_log2 = sc_cur;
    const float m_sub = I8 ? __builtin_fmaf(sc_cur, I8_MAGIC, m_c) : m_c;
    float rs0 = 0.f, rs1 = 0.f;
    uint32_t pw[16];     // packed bf16 pairs of P: word i = scores 2i, 2i+1 (score i<16 -> c0[i], else c1[i-16])
    // exp / sum / pack of two adjacent scores; the empty asm pins the result inside the
    // scheduling region it is written in (pure arithmetic would otherwise sink to its use)
#define SOFTMAX_PAIR(i)                                                                            \
    do {                                                                                           \
        const f32x16& c_ = (i) < 8 ? c0 : c1;                                                      \
        const int r_ = ((i) & 7) * 2;                                                              \
        const float pa_ = __builtin_amdgcn_exp2f(__builtin_fmaf(c_[r_], scale_log2, -m_sub));      \
        const float pb_ = __builtin_amdgcn_exp2f(__builtin_fmaf(c_[r_ + 1], scale_log2, -m_sub));  \
        rs0 += pa_;                                                                                \
        rs1 += pb_;                                                                                \
        pw[i] = pack_bf2(pa_, pb_);                                                                \
        asm volatile("" : "+v"(pw[i]), "+v"(rs0), "+v"(rs1));                                      \
    } while (0)
    const int g = lane >> 4, i16 = lane & 15, qd = i16 >> 2, pp = i16 & 3;
    bf16x8 vf[2][4];
#define READ_V(st, dst)                                                                            \
    _Pragma("unroll") for (int db = 0; db < 4; ++db) {                                             \
        const int krow = 16 * (st) + 4 * (g >> 1) + qd;                                            \
        const int chunk = db * 4 + 2 * (g & 1) + (pp >> 1);                                        \
        const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(                              \
            (lds_b4_ptr)(sv_cur + v_off(krow, chunk) + 8 * (pp & 1)));                             \
        const bf16x4_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(                             \
            (lds_b4_ptr)(sv_cur + v_off(krow + 8, chunk) + 8 * (pp & 1)));                         \
        const uint2 lo2 = __builtin_bit_cast(uint2, lo), hi2 = __builtin_bit_cast(uint2, hi4);     \
        dst[db] = __builtin_bit_cast(bf16x8, make_uint4(lo2.x, lo2.y, hi2.x, hi2.y));              \
    }
    // K fragments are fetched one k-step ahead; the scheduling fences keep each k-step's
    // {2 LDS reads, 2 MFMAs, its share of the softmax} together so that register life times
    // stay short and the vector work sits in the MFMA shadows.  16 of the 32 scores are
    // finished here, the other 16 beside the first three P V steps.
    if constexpr (!I8) {
        bf16x8 kf[2][2];
        if (HAS_NEXT) {
            kf[0][0] = *reinterpret_cast<const bf16x8*>(sk_next + k_off(l31, hi));
            kf[0][1] = *reinterpret_cast<const bf16x8*>(sk_next + k_off(32 + l31, hi));
        }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            if (HAS_NEXT) {
                if (ks < 7) {
                    kf[(ks + 1) & 1][0] = *reinterpret_cast<const bf16x8*>(sk_next + k_off(l31, 2 * ks + 2 + hi));
                    kf[(ks + 1) & 1][1] = *reinterpret_cast<const bf16x8*>(sk_next + k_off(32 + l31, 2 * ks + 2 + hi));
                }
                n0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks & 1][0], qf[ks], n0, 0, 0, 0);
                n1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks & 1][1], qf[ks], n1, 0, 0, 0);
            }
            SOFTMAX_PAIR(ks);           // one pair per k-step (12 of 16 pairs here ran 1.3 % slower than 8)
            if (ks == 0) stage_next<NW>(ks_, k_dst);
            if (ks == 1) stage_next<NW>(vs_, v_dst);
            if (ks == 7) READ_V(0, vf[0]);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        // int8: 4 k-steps of 32 dims, two v_mfma_i32_32x32x32_i8 each (half the matrix time of the bf16 form);
        // 8 of the 16 score pairs are finished here, the rest beside the P V products
        i32x4 kf[2][2];
        const i32x4* q8 = reinterpret_cast<const i32x4*>(&qf[0]);
        if (HAS_NEXT) {
            kf[0][0] = *reinterpret_cast<const i32x4*>(sk_next + k8_off(l31, hi));
            kf[0][1] = *reinterpret_cast<const i32x4*>(sk_next + k8_off(32 + l31, hi));
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (HAS_NEXT) {
                if (ks < 3) {
                    kf[(ks + 1) & 1][0] = *reinterpret_cast<const i32x4*>(sk_next + k8_off(l31, 2 * ks + 2 + hi));
                    kf[(ks + 1) & 1][1] = *reinterpret_cast<const i32x4*>(sk_next + k8_off(32 + l31, 2 * ks + 2 + hi));
                }
                n0 = __builtin_bit_cast(f32x16, __builtin_amdgcn_mfma_i32_32x32x32_i8(
                                                    kf[ks & 1][0], q8[ks], __builtin_bit_cast(i32x16, n0), 0, 0, 0));
                n1 = __builtin_bit_cast(f32x16, __builtin_amdgcn_mfma_i32_32x32x32_i8(
                                                    kf[ks & 1][1], q8[ks], __builtin_bit_cast(i32x16, n1), 0, 0, 0));
            }
            SOFTMAX_PAIR(2 * ks);
            SOFTMAX_PAIR(2 * ks + 1);
            if (ks == 0) stage_next_k8<NW>(ks_, k_dst);
            if (ks == 1) stage_next<NW>(vs_, v_dst);
            if (ks == 3) READ_V(0, vf[0]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (HAS_NEXT) mask_scores<MASK_NEXT>(n0, n1, key0_next, k_len, tail_bias, hi);

    // ---- O^T += V^T P^T of the current tile; beside it the last 8 scores' softmax, then the
    // next tile's lane maximum
    float nmx = -3.0e38f;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        if (st < 3) READ_V(st + 1, vf[(st + 1) & 1]);
        const bf16x8 pfrag =
            __builtin_bit_cast(bf16x8, make_uint4(pw[4 * st], pw[4 * st + 1], pw[4 * st + 2], pw[4 * st + 3]));
#pragma unroll
        for (int db = 0; db < 4; ++db)
            o_acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[st & 1][db], pfrag, o_acc[db], 0, 0, 0);
        // pairs 8-10, 11-13, 14-15 (each four are packed before the step that multiplies them)
        if (st == 0) { SOFTMAX_PAIR(8); SOFTMAX_PAIR(9); SOFTMAX_PAIR(10); }
        if (st == 1) { SOFTMAX_PAIR(11); SOFTMAX_PAIR(12); SOFTMAX_PAIR(13); }
        if (st == 2) { SOFTMAX_PAIR(14); SOFTMAX_PAIR(15); }
        if (st >= 2 && HAS_NEXT) {
            const f32x16& n = st == 2 ? n0 : n1;
#pragma unroll
            for (int r = 0; r < 16; r += 2) nmx = max3(nmx, n[r], n[r + 1]);
            asm volatile("" : "+v"(nmx));
            if (st == 3) {
                mx = (I8 ? half_max(nmx) - I8_MAGIC : half_max(nmx)) * sc_next;
                need = __any(mx > m_c + RESCALE_THR);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    l_run += rs0 + rs1;
#undef READ_V
#undef SOFTMAX_PAIR
}

constexpr int WS_VALUES = 66;      // per lane of a split workgroup: 64 accumulators, row maximum, row sum

// I8: Q and K point at int8 tensors of the same [B, L, N, 128] shape (strides in elements = bytes) with
// q_scale [B, Lq, N] (one per token and head) and k_scale [B, ceil(Lk / 64), N] (one per 64-key tile and
// head): q ~ q8 * q_scale, k ~ k8 * k_scale.
template <int NW, bool I8>
__device__ __forceinline__ void
attn_fwd_pipe(const void* __restrict__ Qv, const void* __restrict__ Kv, const bf16_t* __restrict__ V,
              bf16_t* __restrict__ O, int Lq, int Lk, int k_len, long q_ld, long k_ld, long v_ld, long o_ld,
              long q_bs, long k_bs, long v_bs, long o_bs, float scale_log2, float tail_bias, int n_qblocks,
              int n_heads, int n_batch, float* __restrict__ ws, int n_split, int k_split,
              const float* __restrict__ q_scale, const float* __restrict__ k_scale) {
    constexpr int ESZ = I8 ? 1 : 2;         // bytes per Q / K element
    const char* Q = reinterpret_cast<const char*>(Qv);
    const char* K = reinterpret_cast<const char*>(Kv);
    extern __shared__ __attribute__((aligned(16))) char smem[];   // K[2] | V[2]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    // 1-D grid over (query block, head, batch).  Workgroup ids go round-robin over the 8 XCDs, each
    // with its own L2: when the (batch, head) pairs divide by 8, pair p is served entirely by
    // XCD p % 8 - consecutive workgroups of that XCD walk its query blocks - so a head's K/V
    // stream is fetched into one L2 instead of eight.
    //
    // Tail balancing: every workgroup runs for the same time (all keys), so a grid that is not a
    // multiple of the CU count idles part of the chip during its last round.  The host may ask
    // for the last n_split query blocks of every XCD to be served by two workgroups each, one
    // per half of the keys (split at k_split); those write their unnormalised accumulators, row
    // maximum and row sum to `ws`, and attn_merge_kernel combines the two halves.
    int qb, pair, half = -1, s_idx = 0;
    {
        const int id = blockIdx.x, pairs = n_heads * n_batch;
        if ((pairs & 7) == 0) {
            int j = id >> 3;
            const int n_full = (pairs >> 3) * n_qblocks - n_split;
            if (j >= n_full) {
                const int h = j - n_full;
                half = h & 1;
                s_idx = (id & 7) * n_split + (h >> 1);
                j = n_full + (h >> 1);
            }
            pair = (id & 7) + 8 * (j / n_qblocks);
            qb = j % n_qblocks;
        } else {
            // any other pair count: XCD x (= id % 8) gets one contiguous range of the (pair, query block)
            // space, so its CUs stream the same K/V at the same time (consecutive ids would put all 8 XCDs
            // on one pair, each fetching that pair's K/V for itself: 8.5x the algorithmic bytes at 10 pairs)
            const int total = pairs * n_qblocks, xcd = id & 7, q = total >> 3, r = total & 7;
            const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
            pair = lin / n_qblocks;
            qb = lin % n_qblocks;
        }
    }
    const int head = pair % n_heads, b = pair / n_heads;
    const int q0 = qb * (32 * NW) + wave * 32;

    if (half == 0) {                 // keys [0, k_split): whole tiles, no tail
        k_len = k_split;
        tail_bias = 0.f;
    }
    const int k_begin = half == 1 ? k_split : 0;
    if (half == 1) {                 // keys [k_split, k_len)
        k_len -= k_split;
        Lk -= k_split;
    }
    const char* Qb = Q + (b * q_bs + (long)head * HD) * ESZ;
    const char* Kb = K + (b * k_bs + (long)head * HD + (long)k_begin * k_ld) * ESZ;
    const bf16_t* Vb = V + b * v_bs + (long)head * HD + (long)k_begin * v_ld;

    bf16x8 qf[8];
    float sq = scale_log2;                  // int8: scale_log2 * this lane's query scale
    const float* ksc = nullptr;             // int8: scales of this (batch, head)'s key tiles, tile stride n_heads
    {
        int qr = q0 + l31;
        qr = qr < Lq ? qr : Lq - 1;
        if constexpr (I8) {
            const char* qp = Qb + (long)qr * q_ld + 16 * hi;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + 32 * ks);
            sq *= q_scale[((long)b * Lq + qr) * n_heads + head];
            ksc = k_scale + ((long)b * ((Lk + (half == 1 ? k_split : 0) + BKV - 1) / BKV) + k_begin / BKV) * n_heads + head;
        } else {
            const char* qp = Qb + ((long)qr * q_ld + 8 * hi) * 2;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + 32 * ks);
        }
    }
    // scale of tile t's scores (bf16: scale_log2 for every tile)
    auto tile_scale = [&](int t) -> float { return I8 ? sq * ksc[(long)t * n_heads] : scale_log2; };
    f32x16 o_acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) o_acc[i][j] = 0.f;
    float m_c = -1e30f, l_run = 0.f;
    const int nt = (k_len + BKV - 1) / BKV;
    const bool partial = nt * BKV > k_len || tail_bias != 0.f;     // last tile needs the fix-up pass

    // staging pass i of wave w covers tile rows 4*(i*NW + w) .. +3 (one wave instruction = 4 rows,
    // lane-linear in LDS); the XOR swizzle of k_off / v_off is applied to the SOURCE chunk
    Stream kst, vst;
    {
        const int r = wave * 4 + (lane >> 4), p = lane & 15;
        kst.base = Kb;
        vst.base = reinterpret_cast<const char*>(Vb);
        vst.row = (unsigned)r * (unsigned)v_ld * 2u;
        vst.chunk = (unsigned)((p ^ ((r & 3) << 2)) << 4);
        vst.last = (unsigned)(Lk - 1) * (unsigned)v_ld * 2u;
        vst.pass = 4u * NW * (unsigned)v_ld * 2u;
        vst.tile = (unsigned)BKV * (unsigned)v_ld * 2u;
        if constexpr (I8) {        // 128-byte rows: a wave instruction covers 8 rows, 8 lanes each
            const int r8 = wave * 8 + (lane >> 3), p8 = lane & 7;
            kst.row = (unsigned)r8 * (unsigned)k_ld;
            kst.chunk = (unsigned)((p8 ^ ((r8 >> 1) & 7)) << 4);
            kst.last = (unsigned)(Lk - 1) * (unsigned)k_ld;
            kst.pass = 8u * NW * (unsigned)k_ld;
            kst.tile = (unsigned)BKV * (unsigned)k_ld;
        } else {
            kst.row = (unsigned)r * (unsigned)k_ld * 2u;
            kst.chunk = (unsigned)((p ^ (r & 15)) << 4);
            kst.last = (unsigned)(Lk - 1) * (unsigned)k_ld * 2u;
            kst.pass = 4u * NW * (unsigned)k_ld * 2u;
            kst.tile = (unsigned)BKV * (unsigned)k_ld * 2u;
        }
    }
    char* const wave_lds = smem + wave * 1024;
    auto stage_k = [&](char* dst) {
        if constexpr (I8) stage_next_k8<NW>(kst, dst);
        else stage_next<NW>(kst, dst);
    };
    stage_k(wave_lds);                                               // K(0) -> K buffer 0
    stage_next<NW>(vst, wave_lds + 2 * KV_TILE_BYTES);               // V(0) -> V buffer 0
    stage_k(wave_lds + KV_TILE_BYTES);                               // K(1) -> K buffer 1
    float sc_c = tile_scale(0), sc_n = sc_c;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    f32x16 a0, a1, b0, b1;
    float mx;
    bool need;
    {   // scores of tile 0
        if constexpr (I8) {
#pragma unroll
            for (int j = 0; j < 16; ++j) a0[j] = a1[j] = __int_as_float(I8_MAGIC_BITS);
            const i32x4* q8 = reinterpret_cast<const i32x4*>(&qf[0]);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const i32x4 kf0 = *reinterpret_cast<const i32x4*>(smem + k8_off(l31, 2 * ks + hi));
                const i32x4 kf1 = *reinterpret_cast<const i32x4*>(smem + k8_off(32 + l31, 2 * ks + hi));
                a0 = __builtin_bit_cast(f32x16, __builtin_amdgcn_mfma_i32_32x32x32_i8(kf0, q8[ks], __builtin_bit_cast(i32x16, a0), 0, 0, 0));
                a1 = __builtin_bit_cast(f32x16, __builtin_amdgcn_mfma_i32_32x32x32_i8(kf1, q8[ks], __builtin_bit_cast(i32x16, a1), 0, 0, 0));
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) a0[j] = a1[j] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const bf16x8 kf0 = *reinterpret_cast<const bf16x8*>(smem + k_off(l31, 2 * ks + hi));
                const bf16x8 kf1 = *reinterpret_cast<const bf16x8*>(smem + k_off(32 + l31, 2 * ks + hi));
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf0, qf[ks], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf1, qf[ks], a1, 0, 0, 0);
            }
        }
        if (nt == 1 && partial) mask_scores<true>(a0, a1, 0, k_len, tail_bias, hi);
        const float raw = half_max(lane_max32(a0, a1));
        mx = (I8 ? raw - I8_MAGIC : raw) * sc_c;
        need = __any(mx > m_c + RESCALE_THR);
    }
    __syncthreads();    // every wave is done with K(0) before K(2) overwrites it

    // one step: cur = (C0, C1) holds tile t; (N0, N1) receives tile t + 1; K(t + 2) streams into
    // the K buffer tile t used, V(t + 1) into the V buffer tile t - 1 used
#define PIPE_STEP(HAS_NEXT, MASK_NEXT, C0, C1, N0, N1, t)                                                \
    do {                                                                                                 \
        const int t_ = (t);                                                                              \
        if (I8 && HAS_NEXT) sc_n = tile_scale(t_ + 1);                                                   \
        pipe_step<NW, I8, HAS_NEXT, MASK_NEXT>(o_acc, qf, m_c, l_run, C0, C1, N0, N1, mx, need,          \
                                               smem + ((t_ + 1) & 1) * KV_TILE_BYTES,                    \
                                               smem + (2 + (t_ & 1)) * KV_TILE_BYTES, kst,               \
                                               wave_lds + (t_ & 1) * KV_TILE_BYTES, vst,                 \
                                               wave_lds + (2 + ((t_ + 1) & 1)) * KV_TILE_BYTES,          \
                                               (t_ + 1) * BKV, k_len, tail_bias, sc_c, sc_n, lane);      \
        sc_c = sc_n;                                                                                     \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                 \
        __syncthreads();                                                                                 \
    } while (0)

    // steps whose next tile exists and needs no mask run in the loop (one code path, two
    // roles); a masked next tile and the last tile are peeled behind it
    const int n_pipe = partial ? (nt >= 2 ? nt - 2 : 0) : nt - 1;
    int t = 0;
    for (; t + 1 < n_pipe; t += 2) {
        PIPE_STEP(true, false, a0, a1, b0, b1, t);
        PIPE_STEP(true, false, b0, b1, a0, a1, t + 1);
    }
    if (t < n_pipe) {
        PIPE_STEP(true, false, a0, a1, b0, b1, t);
        a0 = b0;
        a1 = b1;
        ++t;
    }
    if (partial && nt >= 2) {
        PIPE_STEP(true, true, a0, a1, b0, b1, t);
        a0 = b0;
        a1 = b1;
        ++t;
    }
    PIPE_STEP(false, false, a0, a1, b0, b1, t);
#undef PIPE_STEP

    if (half >= 0) {                 // partial result in the register layout, [value][lane]: coalesced rows
        float* w = ws + ((long)(s_idx * 2 + half) * NW + wave) * (WS_VALUES * 64) + lane;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int j = 0; j < 16; ++j) w[(db * 16 + j) * 64] = o_acc[db][j];
        w[64 * 64] = m_c;
        w[65 * 64] = half_sum(l_run);
        return;
    }
    const float inv = 1.0f / half_sum(l_run);
    // Output through LDS (the K/V buffers are free after the last step's barrier; 8 KiB per wave):
    // the accumulators hold 4 consecutive d of one query row per register group, which stored
    // directly are 16-byte fragments per row.  Written as [32 rows][256 B] with 16-byte chunk c of
    // row r at c ^ (r & 15) (two lanes per bank: the minimum for 512 B), read back a quarter row per
    // lane so that every store instruction covers four rows of two full 128-byte lines.
    {
        char* ep = smem + wave * 8192;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                uint2 w;
                w.x = pack_bf2(o_acc[db][4 * g4 + 0] * inv, o_acc[db][4 * g4 + 1] * inv);
                w.y = pack_bf2(o_acc[db][4 * g4 + 2] * inv, o_acc[db][4 * g4 + 3] * inv);
                *reinterpret_cast<uint2*>(ep + l31 * 256 + (((db * 4 + g4) ^ (l31 & 15)) << 4) + hi * 8) = w;
            }
        const int rr = lane >> 4, cc = lane & 15;
        bf16_t* ob = O + b * o_bs + (long)head * HD + cc * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = i * 4 + rr;
            const uint4 w = *reinterpret_cast<const uint4*>(ep + r * 256 + ((cc ^ (r & 15)) << 4));
            if (q0 + r < Lq) *reinterpret_cast<uint4*>(ob + (long)(q0 + r) * o_ld) = w;
        }
    }
}

#define ATTN_KERNEL(NW)                                                                                  \
    __global__ void __launch_bounds__(64 * NW, 2) attn_pipe_kernel_w##NW(                                \
        const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ V,       \
        bf16_t* __restrict__ O, int Lq, int Lk, int k_len, long q_ld, long k_ld, long v_ld, long o_ld,   \
        long q_bs, long k_bs, long v_bs, long o_bs, float scale_log2, float tail_bias, int n_qblocks,    \
        int n_heads, int n_batch, float* __restrict__ ws, int n_split, int k_split) {                    \
        attn_fwd_pipe<NW, false>(Q, K, V, O, Lq, Lk, k_len, q_ld, k_ld, v_ld, o_ld, q_bs, k_bs, v_bs,    \
                                 o_bs, scale_log2, tail_bias, n_qblocks, n_heads, n_batch, ws, n_split,  \
                                 k_split, nullptr, nullptr);                                             \
    }                                                                                                    \
    __global__ void __launch_bounds__(64 * NW, 2) attn_i8_kernel_w##NW(                                  \
        const int8_t* __restrict__ Q, const float* __restrict__ q_scale, const int8_t* __restrict__ K,  \
        const float* __restrict__ k_scale, const bf16_t* __restrict__ V, bf16_t* __restrict__ O, int Lq, \
        int Lk, int k_len, long q_ld, long k_ld, long v_ld, long o_ld, long q_bs, long k_bs, long v_bs,  \
        long o_bs, float scale_log2, int n_qblocks, int n_heads, int n_batch, float* __restrict__ ws,    \
        int n_split, int k_split) {                                                                      \
        attn_fwd_pipe<NW, true>(Q, K, V, O, Lq, Lk, k_len, q_ld, k_ld, v_ld, o_ld, q_bs, k_bs, v_bs,     \
                                o_bs, scale_log2, 0.f, n_qblocks, n_heads, n_batch, ws, n_split,         \
                                k_split, q_scale, k_scale);                                              \
    }
ATTN_KERNEL(4)
ATTN_KERNEL(8)
#undef ATTN_KERNEL

// Combines the two key halves of the split query blocks: one 64-lane block per (split block, wave),
// every lane in the role it had in the attention kernel.  (o, m, l) of a half are relative to its own
// reference maximum m: o = o0 2^(m0-m) + o1 2^(m1-m), l likewise, out = o / l.
__global__ void __launch_bounds__(64) attn_merge_kernel(const float* __restrict__ ws, bf16_t* __restrict__ O, int Lq,
                                                        long o_ld, long o_bs, int n_qblocks, int n_heads, int n_batch,
                                                        int n_split, int nw) {
    const int lane = threadIdx.x, l31 = lane & 31, hi = lane >> 5;
    const int s_idx = blockIdx.x / nw, wave = blockIdx.x % nw;
    const int xcd = s_idx / n_split;
    const int j = ((n_heads * n_batch) >> 3) * n_qblocks - n_split + s_idx % n_split;
    const int pair = xcd + 8 * (j / n_qblocks), qb = j % n_qblocks;
    const int head = pair % n_heads, b = pair / n_heads;
    const int qr = qb * (32 * nw) + wave * 32 + l31;
    const float* w0 = ws + ((long)(s_idx * 2) * nw + wave) * (WS_VALUES * 64) + lane;
    const float* w1 = w0 + (long)nw * (WS_VALUES * 64);
    const float m0 = w0[64 * 64], m1 = w1[64 * 64];
    const float m = fmaxf(m0, m1);
    const float a0 = __builtin_amdgcn_exp2f(m0 - m), a1 = __builtin_amdgcn_exp2f(m1 - m);
    const float inv = 1.0f / (w0[65 * 64] * a0 + w1[65 * 64] * a1);
    if (qr >= Lq) return;
    bf16_t* op = O + b * o_bs + (long)qr * o_ld + (long)head * HD + 4 * hi;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            float y[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = (db * 16 + 4 * g4 + i) * 64;
                y[i] = (w0[r] * a0 + w1[r] * a1) * inv;
            }
            uint2 w;
            w.x = pack_bf2(y[0], y[1]);
            w.y = pack_bf2(y[2], y[3]);
            *reinterpret_cast<uint2*>(op + db * 32 + 8 * g4) = w;
        }
}

// bf16 x [B, L, N, 128] -> int8 q of the same shape and one scale per (RPS rows, head): x ~ q * scale,
// scale = absmax / 127 over the RPS x 128 block (1 when the block is all zero), q = rint(x * (127 / absmax)).
// One block = 64 rows of one (batch, head); thread t holds 32 consecutive dims of row t / 4.
template <int RPS>
__global__ void __launch_bounds__(256) quant_heads_i8_kernel(const bf16_t* __restrict__ x, long ldx, long x_bs,
                                                             int8_t* __restrict__ q, long ldq, long q_bs,
                                                             float* __restrict__ scale, int L, int N) {
    __shared__ float red[4];
    const int tid = threadIdx.x, row = blockIdx.x * 64 + (tid >> 2), quarter = tid & 3;
    const int head = blockIdx.y, b = blockIdx.z;
    const bool live = row < L;
    uint4 raw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) raw[i] = make_uint4(0, 0, 0, 0);
    if (live) {
        const uint4* src = reinterpret_cast<const uint4*>(x + b * x_bs + (long)row * ldx + head * HD + quarter * 32);
#pragma unroll
        for (int i = 0; i < 4; ++i) raw[i] = src[i];
    }
    float v[32];
    float mx = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t u[4] = {raw[i].x, raw[i].y, raw[i].z, raw[i].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[i * 8 + 2 * j] = bf2f(u[j] & 0xffff);
            v[i * 8 + 2 * j + 1] = bf2f(u[j] >> 16);
            mx = __builtin_fmaxf(mx, __builtin_fmaxf(fabsf(v[i * 8 + 2 * j]), fabsf(v[i * 8 + 2 * j + 1])));
        }
    }
    mx = __builtin_fmaxf(mx, __shfl_xor(mx, 1, 64));
    mx = __builtin_fmaxf(mx, __shfl_xor(mx, 2, 64));
    if (RPS > 1) {
#pragma unroll
        for (int o = 4; o < 64; o <<= 1) mx = __builtin_fmaxf(mx, __shfl_xor(mx, o, 64));
        if ((tid & 63) == 0) red[tid >> 6] = mx;
        __syncthreads();
        mx = __builtin_fmaxf(__builtin_fmaxf(red[0], red[1]), __builtin_fmaxf(red[2], red[3]));
    }
    const float inv = mx > 0.f ? 127.0f / mx : 0.f;
    if (RPS > 1) {
        if (tid == 0) scale[((long)b * gridDim.x + blockIdx.x) * N + head] = mx > 0.f ? mx / 127.0f : 1.f;
    } else if (live && quarter == 0) {
        scale[((long)b * L + row) * N + head] = mx > 0.f ? mx / 127.0f : 1.f;
    }
    if (!live) return;
    uint32_t o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint32_t w = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int qi = __float2int_rn(v[i * 4 + j] * inv);
            qi = qi > 127 ? 127 : (qi < -127 ? -127 : qi);
            w |= (uint32_t)(qi & 0xff) << (8 * j);
        }
        o[i] = w;
    }
    uint4* dst = reinterpret_cast<uint4*>(q + b * q_bs + (long)row * ldq + head * HD + quarter * 32);
    dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
    dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
}

}  // namespace

// Plan of the tail balancing for an 8-wave grid: the number of query blocks per XCD to split in two
// (0 = none).  Worth it when the last round of workgroups would fill at most half of the CUs.
static int attn_split_plan(int B, int Lq, int N, int k_len) {
    const int cus = ovd_device_cus();
    const long pairs = (long)B * N, nwg = pairs * ceil_div(Lq, 256);
    if ((pairs & 7) || (cus & 7) || k_len <= 16 * BKV) return 0;      // (at most 1024 keys run on 4-wave workgroups)
    const long rounds = nwg / cus, rem = nwg % cus;
    if (rounds < 1 || rounds > 32 || rem == 0 || rem > cus / 2 || (rem & 7)) return 0;
    return (int)(rem / 8);
}

// bytes of workspace with which ovd_impl_attn_fwd can balance its last round (0: it would not split)
long ovd_impl_attn_workspace_bytes(int B, int Lq, int N, int k_len) {
    if ((long)ceil_div(Lq, 256) * N * B < 256) return 0;
    return (long)attn_split_plan(B, Lq, N, k_len) * 8 * 2 * 8 * (WS_VALUES * 64) * (long)sizeof(float);
}

int ovd_impl_attn_fwd(const bf16_t* q, const bf16_t* k, const bf16_t* v, bf16_t* out, int B, int Lq, int Lk,
                      int N, int D, int k_len, long q_ld, long k_ld, long v_ld, long o_ld, long q_bs, long k_bs,
                      long v_bs, long o_bs, float softmax_scale, int last_key_copies, void* workspace,
                      long workspace_bytes, hipStream_t st);

// ---- allocation-free variant ------------------------------------------------------------------
// The tail-balancing workspace as a device array of the library itself, for callers that cannot
// allocate.  Sized for the largest split the plan above produces on a 256-CU device: 16 query blocks
// per XCD x 8 XCDs x 2 key halves x 8 waves x 66 values x 64 lanes x 4 bytes = 33 MiB, held by every
// device the library runs on until the process ends.  Each device has its own copy at its own
// address; `scratch` itself is only a host-side handle for hipGetSymbolAddress.
constexpr long ATTN_SCRATCH_BYTES = 16L * 8 * 2 * 8 * (WS_VALUES * 64) * (long)sizeof(float);
static __device__ __align__(256) unsigned char scratch[ATTN_SCRATCH_BYTES];

int ovd_impl_attn_fwd_static(const bf16_t* q, const bf16_t* k, const bf16_t* v, bf16_t* out, int B, int Lq, int Lk,
                             int N, int D, int k_len, long q_ld, long k_ld, long v_ld, long o_ld, long q_bs,
                             long k_bs, long v_bs, long o_bs, float softmax_scale, int last_key_copies,
                             hipStream_t st) {
    static std::atomic<void*> dev_ptr[OVD_MAX_DEVICES];      // device address of `scratch`, per device
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= OVD_MAX_DEVICES)
        OVD_FAIL("attention (static workspace): no current device, or device index %d beyond %d", dev,
                 OVD_MAX_DEVICES - 1);
    void* ptr = dev_ptr[dev].load(std::memory_order_acquire);
    if (ptr == nullptr) {       // first call on this device (with it current): look the array up once
        const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || ptr == nullptr)
            OVD_FAIL("attention (static workspace): hipGetSymbolAddress failed on device %d: %s", dev,
                     hipGetErrorString(e));
        dev_ptr[dev].store(ptr, std::memory_order_release);
    }
    OVD_REQUIRE(B > 0 && N > 0 && Lq > 0 && k_len > 0, "attention: empty problem B=%d N=%d Lq=%d k_len=%d", B, N, Lq,
                k_len);
    const long need = ovd_impl_attn_workspace_bytes(B, Lq, N, k_len);
    OVD_REQUIRE(need <= ATTN_SCRATCH_BYTES,
                "attention (static workspace): B=%d Lq=%d N=%d k_len=%d needs %ld bytes of split workspace, the "
                "built-in array holds %ld; use ovd_flash_attn_ws_fwd_bf16 with a caller-owned workspace",
                B, Lq, N, k_len, need, ATTN_SCRATCH_BYTES);
    // every value of the workspace is written by the split workgroups before the merge kernel reads it:
    // nothing has to start at zero, and the previous call's contents are simply overwritten
    return ovd_impl_attn_fwd(q, k, v, out, B, Lq, Lk, N, D, k_len, q_ld, k_ld, v_ld, o_ld, q_bs, k_bs, v_bs, o_bs,
                             softmax_scale, last_key_copies, ptr, ATTN_SCRATCH_BYTES, st);
}

// q [B,Lq,N,128], k/v [B,Lk,N,128], out [B,Lq,N,128]; *_ld = elements between
// consecutive tokens, *_bs = elements between batch entries (so the fused QKV
// projection buffer can be addressed in place).  Keys at index >= k_len are masked; key
// k_len - 1 counts last_key_copies times (a run of identical trailing keys folded into one).
int ovd_impl_attn_fwd(const bf16_t* q, const bf16_t* k, const bf16_t* v, bf16_t* out, int B, int Lq, int Lk,
                      int N, int D, int k_len, long q_ld, long k_ld, long v_ld, long o_ld, long q_bs, long k_bs,
                      long v_bs, long o_bs, float softmax_scale, int last_key_copies, void* workspace,
                      long workspace_bytes, hipStream_t st) {
    OVD_REQUIRE(D == HD, "attention: head_dim %d not supported (only 128)", D);
    OVD_REQUIRE(B > 0 && N > 0 && Lq > 0 && Lk > 0, "attention: empty problem B=%d N=%d Lq=%d Lk=%d", B, N, Lq, Lk);
    OVD_REQUIRE(k_len > 0 && k_len <= Lk, "attention: k_len=%d outside (0, Lk=%d]", k_len, Lk);
    OVD_REQUIRE(q_ld % 8 == 0 && k_ld % 8 == 0 && v_ld % 8 == 0 && o_ld % 8 == 0 && o_bs % 8 == 0 &&
                    reinterpret_cast<uintptr_t>(out) % 16 == 0,
                "attention: token strides and the output pointer must keep 16-byte alignment");
    OVD_REQUIRE(q_ld >= (long)N * D && k_ld >= (long)N * D && v_ld >= (long)N * D && o_ld >= (long)N * D,
                "attention: token stride shorter than N*D");
    OVD_REQUIRE(softmax_scale > 0.f, "attention: softmax_scale must be positive");
    OVD_REQUIRE(last_key_copies >= 1, "attention: last_key_copies=%d must be at least 1", last_key_copies);
    OVD_REQUIRE(((long)Lk + 4 * BKV) * (k_ld > v_ld ? k_ld : v_ld) * 2 < (1L << 32),
                "attention: one (batch, head) slab of K/V must span less than 4 GiB (Lk=%d)", Lk);
    const size_t lds = 4 * KV_TILE_BYTES;
    const float scale_log2 = softmax_scale * 1.4426950408889634f;
    const float tail_bias = last_key_copies > 1 ? log2f((float)last_key_copies) / scale_log2 : 0.f;
    // 8-wave workgroups (256 query rows) unless that grid would leave compute units idle or the keys are few;
    // OVD_ATTN_WAVES=4|8 overrides (tuning aid)
    static std::atomic<int> forced_cache{INT32_MIN};
    const int forced = ovd_env_int(forced_cache, "OVD_ATTN_WAVES", 0);
    // few keys (cross-attention over a short context): two 4-wave workgroups per CU overlap each other's
    // prologue and epilogue, which is most of such a launch (0.09 vs 0.11 ms at 131 keys; even at ~1500)
    const bool small = (long)ceil_div(Lq, 256) * N * B < 256 || k_len <= 1024;
    const int waves = forced == 4 || forced == 8 ? forced : (small ? 4 : 8);
    // tail balancing needs the caller's workspace (ovd_impl_attn_workspace_bytes); without it: plain grid
    int n_split = 0, k_split = 0;
    if (waves == 8 && workspace != nullptr) {
        n_split = attn_split_plan(B, Lq, N, k_len);
        if ((long)n_split * 8 * 2 * 8 * (WS_VALUES * 64) * (long)sizeof(float) > workspace_bytes) n_split = 0;
        k_split = (ceil_div(k_len, BKV) / 2) * BKV;
    }
#define ATTN_CASE(KERNEL, NW)                                                                              \
    {                                                                                                      \
        static OvdLdsLimit lim;                                                                            \
        if (!lim.ensure(reinterpret_cast<const void*>(&KERNEL), lds))                                      \
            OVD_FAIL("attention: cannot reserve %zu bytes of LDS", lds);                                    \
        const int nqb = ceil_div(Lq, 32 * NW);                                                             \
        OVD_REQUIRE((long)nqb * N * B + 8L * n_split < (1L << 31), "attention: grid too large");           \
        dim3 grid(nqb * N * B + 8 * n_split), blk(64 * NW);                                                \
        hipLaunchKernelGGL(KERNEL, grid, blk, lds, st, q, k, v, out, Lq, Lk, k_len, q_ld, k_ld, v_ld,      \
                           o_ld, q_bs, k_bs, v_bs, o_bs, scale_log2, tail_bias, nqb, N, B,                 \
                           (float*)workspace, n_split, k_split);                                           \
        if (n_split > 0)                                                                                   \
            hipLaunchKernelGGL(attn_merge_kernel, dim3(8 * n_split * NW), dim3(64), 0, st,                 \
                               (const float*)workspace, out, Lq, o_ld, o_bs, nqb, N, B, n_split, NW);      \
    }
    if (waves == 8) ATTN_CASE(attn_pipe_kernel_w8, 8) else ATTN_CASE(attn_pipe_kernel_w4, 4)
#undef ATTN_CASE
    OVD_CHECK_LAUNCH("attn_fwd");
    return 0;
}

// x [B, L, N, 128] bf16 -> int8 + scales; rows_per_scale 1 (queries: one scale per token and head) or
// 64 (keys: one per 64-key tile and head; scale [B, ceil(L / 64), N])
int ovd_impl_quant_heads_i8(const bf16_t* x, long ldx, long x_bs, int8_t* q, long ldq, long q_bs, float* scale, int B,
                            int L, int N, int D, int rows_per_scale, hipStream_t st) {
    OVD_REQUIRE(D == HD, "quant_heads_i8: head_dim %d not supported (only 128)", D);
    OVD_REQUIRE(B > 0 && L > 0 && N > 0 && N <= 65535 && B <= 65535, "quant_heads_i8: bad shape B=%d L=%d N=%d", B, L, N);
    OVD_REQUIRE(rows_per_scale == 1 || rows_per_scale == BKV, "quant_heads_i8: rows_per_scale must be 1 or %d", BKV);
    OVD_REQUIRE(ldx % 8 == 0 && x_bs % 8 == 0 && ldq % 16 == 0 && q_bs % 16 == 0 &&
                    reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(q) % 16 == 0,
                "quant_heads_i8: rows must keep 16-byte alignment");
    OVD_REQUIRE(ldx >= (long)N * D && ldq >= (long)N * D, "quant_heads_i8: token stride shorter than N*D");
    dim3 grid(ceil_div(L, 64), N, B), blk(256);
    if (rows_per_scale == 1)
        hipLaunchKernelGGL(quant_heads_i8_kernel<1>, grid, blk, 0, st, x, ldx, x_bs, q, ldq, q_bs, scale, L, N);
    else
        hipLaunchKernelGGL(quant_heads_i8_kernel<BKV>, grid, blk, 0, st, x, ldx, x_bs, q, ldq, q_bs, scale, L, N);
    OVD_CHECK_LAUNCH("quant_heads_i8");
    return 0;
}

// Attention with Q K^T on int8 operands (v_mfma_i32_32x32x32_i8, half the matrix time of the bf16 product);
// softmax, P and P V as in ovd_impl_attn_fwd.  q8 / k8 and their scales: ovd_impl_quant_heads_i8.
int ovd_impl_attn_i8_fwd(const int8_t* q8, const float* q_scale, const int8_t* k8, const float* k_scale,
                         const bf16_t* v, bf16_t* out, int B, int Lq, int Lk, int N, int D, int k_len, long q_ld,
                         long k_ld, long v_ld, long o_ld, long q_bs, long k_bs, long v_bs, long o_bs,
                         float softmax_scale, void* workspace, long workspace_bytes, hipStream_t st) {
    OVD_REQUIRE(D == HD, "attention: head_dim %d not supported (only 128)", D);
    OVD_REQUIRE(B > 0 && N > 0 && Lq > 0 && Lk > 0, "attention: empty problem B=%d N=%d Lq=%d Lk=%d", B, N, Lq, Lk);
    OVD_REQUIRE(k_len > 0 && k_len <= Lk, "attention: k_len=%d outside (0, Lk=%d]", k_len, Lk);
    OVD_REQUIRE(q_ld % 16 == 0 && k_ld % 16 == 0 && v_ld % 8 == 0 && o_ld % 8 == 0 && o_bs % 8 == 0 && q_bs % 16 == 0 &&
                    k_bs % 16 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0 &&
                    reinterpret_cast<uintptr_t>(q8) % 16 == 0 && reinterpret_cast<uintptr_t>(k8) % 16 == 0,
                "attention: token strides and pointers must keep 16-byte alignment");
    OVD_REQUIRE(q_ld >= (long)N * D && k_ld >= (long)N * D && v_ld >= (long)N * D && o_ld >= (long)N * D,
                "attention: token stride shorter than N*D");
    OVD_REQUIRE(softmax_scale > 0.f, "attention: softmax_scale must be positive");
    OVD_REQUIRE(((long)Lk + 4 * BKV) * (k_ld > 2 * v_ld ? k_ld : 2 * v_ld) < (1L << 32),
                "attention: one (batch, head) slab of K/V must span less than 4 GiB (Lk=%d)", Lk);
    const size_t lds = 4 * KV_TILE_BYTES;
    const float scale_log2 = softmax_scale * 1.4426950408889634f;
    const bool small = (long)ceil_div(Lq, 256) * N * B < 256 || k_len <= 1024;
    int n_split = 0, k_split = 0;
    if (!small && workspace != nullptr) {
        n_split = attn_split_plan(B, Lq, N, k_len);
        if ((long)n_split * 8 * 2 * 8 * (WS_VALUES * 64) * (long)sizeof(float) > workspace_bytes) n_split = 0;
        k_split = (ceil_div(k_len, BKV) / 2) * BKV;
    }
#define ATTN_CASE(KERNEL, NW)                                                                              \
    {                                                                                                      \
        static OvdLdsLimit lim;                                                                            \
        if (!lim.ensure(reinterpret_cast<const void*>(&KERNEL), lds))                                      \
            OVD_FAIL("attention: cannot reserve %zu bytes of LDS", lds);                                    \
        const int nqb = ceil_div(Lq, 32 * NW);                                                             \
        OVD_REQUIRE((long)nqb * N * B + 8L * n_split < (1L << 31), "attention: grid too large");           \
        dim3 grid(nqb * N * B + 8 * n_split), blk(64 * NW);                                                \
        hipLaunchKernelGGL(KERNEL, grid, blk, lds, st, q8, q_scale, k8, k_scale, v, out, Lq, Lk, k_len,    \
                           q_ld, k_ld, v_ld, o_ld, q_bs, k_bs, v_bs, o_bs, scale_log2, nqb, N, B,          \
                           (float*)workspace, n_split, k_split);                                           \
        if (n_split > 0)                                                                                   \
            hipLaunchKernelGGL(attn_merge_kernel, dim3(8 * n_split * NW), dim3(64), 0, st,                 \
                               (const float*)workspace, out, Lq, o_ld, o_bs, nqb, N, B, n_split, NW);      \
    }
    if (!small) ATTN_CASE(attn_i8_kernel_w8, 8) else ATTN_CASE(attn_i8_kernel_w4, 4)
#undef ATTN_CASE
    OVD_CHECK_LAUNCH("attn_i8_fwd");
    return 0;
}
