// Kernels of the umT5 text encoder that the DiT kernels do not already cover (its linear
// layers run on the bf16 GEMM, its T5LayerNorm on the RMSNorm kernel).  The encoder runs
// two or three times per video on <= 512 tokens: these kernels are written for fidelity to
// the reference's bf16 tensor arithmetic (omnivideo/modules/t5.py), not for peak rate.
#include "common.h"

namespace {

constexpr int T5_HD = 64;      // head dim of umT5 (dim_attn / num_heads)
constexpr int QROWS = 16;      // query rows per workgroup

// Self-attention with an additive relative-position bias and no score scaling
// (t5.py:85-119): s = bf16(bf16(q.k) + bias), p = bf16(softmax_fp32(s)), o = bf16(p.v);
// keys at or beyond k_len get the lowest score (their probability is exactly 0).
// qkv [B, L, 3*N*64] bf16 (q | k | v, token stride ld), pos [buckets, N] bf16,
// bucket [L, L] int32, out [B, L, N*64] bf16.  One workgroup = 16 query rows of one (b, head).
__global__ void __launch_bounds__(256) t5_attn_kernel(const bf16_t* __restrict__ qkv, long ld,
                                                      const bf16_t* __restrict__ pos,
                                                      const int* __restrict__ bucket,
                                                      const int* __restrict__ k_lens, bf16_t* __restrict__ out,
                                                      long ldo, int L, int N) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* S = sm;                       // [QROWS][L]
    float* Qs = sm + QROWS * L;          // [QROWS][64]
    const int tid = threadIdx.x, r = tid >> 4, c = tid & 15;
    const int i0 = blockIdx.x * QROWS, head = blockIdx.y, b = blockIdx.z;
    const int k_len = k_lens ? k_lens[b] : L;
    const bf16_t* base = qkv + (long)b * L * ld;
    const long qo = (long)head * T5_HD, ko = (long)(N + head) * T5_HD, vo = (long)(2 * N + head) * T5_HD;

    {   // stage the 16 query rows as fp32: thread (r, c) loads 4 values
        const int i = min(i0 + r, L - 1);
        const uint2 u = *reinterpret_cast<const uint2*>(base + (long)i * ld + qo + 4 * c);
        float* d = Qs + r * T5_HD + 4 * c;
        d[0] = bf2f(u.x & 0xffff); d[1] = bf2f(u.x >> 16); d[2] = bf2f(u.y & 0xffff); d[3] = bf2f(u.y >> 16);
    }
    __syncthreads();
    const int i = min(i0 + r, L - 1);
    for (int j = c; j < L; j += 16) {
        const uint4* kp = reinterpret_cast<const uint4*>(base + (long)j * ld + ko);
        const float* q = Qs + r * T5_HD;
        float dot = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const uint4 u = kp[t];
            const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                dot = fmaf(q[8 * t + 2 * e], bf2f(w[e] & 0xffff), dot);
                dot = fmaf(q[8 * t + 2 * e + 1], bf2f(w[e] >> 16), dot);
            }
        }
        float s = round_bf(dot);
        s = round_bf(s + bf2f(pos[(long)bucket[(long)i * L + j] * N + head]));
        if (j >= k_len) s = -3.3895313892515355e38f;      // bf16 lowest (masked_fill, t5.py:109)
        S[r * L + j] = s;
    }
    __syncthreads();
    // softmax over the row in fp32 (16 lanes per row), result rounded to bf16
    float mx = -3.4e38f;
    for (int j = c; j < L; j += 16) mx = fmaxf(mx, S[r * L + j]);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 16));
    float sum = 0.f;
    for (int j = c; j < L; j += 16) {
        const float e = expf(S[r * L + j] - mx);
        S[r * L + j] = e;
        sum += e;
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 16);
    for (int j = c; j < L; j += 16) S[r * L + j] = round_bf(S[r * L + j] / sum);
    __syncthreads();
    // o[r][4c .. 4c+3] = sum_j p[r][j] v[j][4c ..]
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < k_len; ++j) {
        const float p = S[r * L + j];
        const uint2 u = *reinterpret_cast<const uint2*>(base + (long)j * ld + vo + 4 * c);
        acc[0] = fmaf(p, bf2f(u.x & 0xffff), acc[0]);
        acc[1] = fmaf(p, bf2f(u.x >> 16), acc[1]);
        acc[2] = fmaf(p, bf2f(u.y & 0xffff), acc[2]);
        acc[3] = fmaf(p, bf2f(u.y >> 16), acc[3]);
    }
    if (i0 + r < L) {
        uint2 w;
        w.x = pack_bf2(acc[0], acc[1]);
        w.y = pack_bf2(acc[2], acc[3]);
        *reinterpret_cast<uint2*>(out + ((long)b * L + i0 + r) * ldo + (long)head * T5_HD + 4 * c) = w;
    }
}

// T5's GELU written out in tensor ops (t5.py:47-49) on a bf16 tensor: every intermediate is
// rounded to bf16, and pow(x, 3) is x*x*x in bf16 arithmetic (two roundings).
__device__ __forceinline__ float t5_gelu_bf16(float x) {
    const float cube = round_bf(round_bf(x * x) * x);
    const float inner = round_bf(x + round_bf(0.044715f * cube));
    const float t = round_bf(tanhf(round_bf(0.7978845608028654f * inner)));
    return round_bf(round_bf(0.5f * x) * round_bf(1.0f + t));
}

// out[m][f] = bf16(fc1[m][f] * gelu(gate[m][f])), h = [gate | fc1] of width 2F (t5.py:135-136)
__global__ void __launch_bounds__(256) t5_gated_gelu_kernel(const bf16_t* __restrict__ h, long ldh,
                                                            bf16_t* __restrict__ out, long ldo, int M, int F) {
    const long n = (long)M * (F / 2);
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
        const int m = (int)(idx / (F / 2)), f = (int)(idx % (F / 2)) * 2;
        const uint32_t g = *reinterpret_cast<const uint32_t*>(h + (long)m * ldh + f);
        const uint32_t u = *reinterpret_cast<const uint32_t*>(h + (long)m * ldh + F + f);
        const float a = bf2f(u & 0xffff) * t5_gelu_bf16(bf2f(g & 0xffff));
        const float b = bf2f(u >> 16) * t5_gelu_bf16(bf2f(g >> 16));
        *reinterpret_cast<uint32_t*>(out + (long)m * ldo + f) = pack_bf2(a, b);
    }
}

// x = bf16(x + y), bf16 residual stream of the encoder (t5.py:172-173)
__global__ void __launch_bounds__(256) add_bf16_kernel(bf16_t* __restrict__ x, const bf16_t* __restrict__ y, long n2) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long)gridDim.x * blockDim.x) {
        const uint32_t a = reinterpret_cast<const uint32_t*>(x)[i], b = reinterpret_cast<const uint32_t*>(y)[i];
        reinterpret_cast<uint32_t*>(x)[i] =
            pack_bf2(bf2f(a & 0xffff) + bf2f(b & 0xffff), bf2f(a >> 16) + bf2f(b >> 16));
    }
}

}  // namespace

int ovd_impl_t5_attention(const bf16_t* qkv, long ld, const bf16_t* pos, const int* bucket, const int* k_lens,
                          bf16_t* out, long ldo, int B, int L, int N, int head_dim, hipStream_t st) {
    OVD_REQUIRE(head_dim == T5_HD, "t5 attention: head_dim %d not supported (only 64)", head_dim);
    OVD_REQUIRE(B > 0 && L > 0 && N > 0, "t5 attention: empty problem");
    OVD_REQUIRE(ld % 8 == 0 && ld >= 3L * N * T5_HD && ldo % 4 == 0 && ldo >= (long)N * T5_HD,
                "t5 attention: bad row strides");
    const size_t lds = (size_t)(QROWS * L + QROWS * T5_HD) * sizeof(float);
    OVD_REQUIRE(lds <= 160 * 1024, "t5 attention: L=%d too long for one LDS score tile", L);
    static OvdLdsLimit lim;
    if (!lim.ensure(reinterpret_cast<const void*>(&t5_attn_kernel), lds))
        OVD_FAIL("t5 attention: cannot reserve %zu bytes of LDS", lds);
    hipLaunchKernelGGL(t5_attn_kernel, dim3(ceil_div(L, QROWS), N, B), dim3(256), lds, st, qkv, ld, pos, bucket,
                       k_lens, out, ldo, L, N);
    OVD_CHECK_LAUNCH("t5_attention");
    return 0;
}

int ovd_impl_t5_gated_gelu(const bf16_t* h, long ldh, bf16_t* out, long ldo, int M, int F, hipStream_t st) {
    OVD_REQUIRE(F > 0 && F % 2 == 0 && ldh % 2 == 0 && ldo % 2 == 0 && ldh >= 2L * F && ldo >= F,
                "t5 gated gelu: bad shape");
    if (M == 0) return 0;
    int blocks = ceil_div((long)M * (F / 2), 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(t5_gated_gelu_kernel, dim3(blocks), dim3(256), 0, st, h, ldh, out, ldo, M, F);
    OVD_CHECK_LAUNCH("t5_gated_gelu");
    return 0;
}

int ovd_impl_add_bf16(bf16_t* x, const bf16_t* y, long n, hipStream_t st) {
    OVD_REQUIRE(n >= 0 && n % 2 == 0, "add_bf16: element count must be even");
    if (n == 0) return 0;
    int blocks = ceil_div(n / 2, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(add_bf16_kernel, dim3(blocks), dim3(256), 0, st, x, y, n / 2);
    OVD_CHECK_LAUNCH("add_bf16");
    return 0;
}
