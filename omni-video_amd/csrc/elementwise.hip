// Small HBM-/latency-bound kernels around the DiT blocks: patch gather, timestep
// embedding, modulation table, fp32 head projection + unpatchify scatter.
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------
// [Cin,T,H,W] -> bf16 rows [f*h*w, Cin*pt*ph*pw], row order (f,h,w), column order
// (cin,pt,ph,pw) = the flattened Conv3d weight (reference: nn.Conv3d with
// kernel=stride=patch, omnivideo/modules/model.py:384-385,455-458 and
// visual_context_adapter.py:52-54,101-108).
template <typename T>
__global__ void patchify_kernel(const T* __restrict__ lat, bf16_t* __restrict__ rows, int Cin, int T_, int H,
                                int W, int pt, int ph, int pw, int gf, int gh, int gw, long total) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int kdim = Cin * pt * ph * pw;
    const int col = (int)(idx % kdim);
    const long tok = idx / kdim;
    const int r = col % pw, q = (col / pw) % ph, p = (col / (pw * ph)) % pt, c = col / (pw * ph * pt);
    const int w = (int)(tok % gw), h = (int)((tok / gw) % gh), f = (int)(tok / ((long)gw * gh));
    const long src = (((long)c * T_ + (f * pt + p)) * H + (h * ph + q)) * W + (w * pw + r);
    float v;
    if constexpr (sizeof(T) == 2) v = bf2f(lat[src]); else v = lat[src];
    rows[idx] = f2bf(v);
}

// ---------------------------------------------------------------------------------
// Sinusoidal timestep features [cos | sin](t * 10000^(-i/half)) evaluated in float64
// and stored as float32 (model.py:13-23,473-474).  t comes from a device table so a
// captured graph can replay with a device-side step counter.
__global__ void time_sinusoid_kernel(const double* __restrict__ timesteps, const int* __restrict__ step_ptr,
                                     int step, float* __restrict__ out, int dim) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int half = dim / 2;
    if (i >= half) return;
    const int s = step_ptr ? *step_ptr : step;
    const double t = timesteps[s];
    const double ang = t * pow(10000.0, -(double)i / (double)half);
    out[i] = (float)cos(ang);
    out[half + i] = (float)sin(ang);
}

// y[n] = sum_k act(x[k]) * W[n][k] + b[n], fp32 math on bf16 weights (the time MLP
// runs under a float32 autocast: model.py:469-476).  One wave per output row.
__global__ void __launch_bounds__(256) gemv_f32_kernel(const float* __restrict__ x, const bf16_t* __restrict__ W,
                                                       const bf16_t* __restrict__ b, float* __restrict__ y, int N,
                                                       int K, int silu_in) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const bf16_t* wr = W + (long)n * K;
    float acc = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
        const uint2 wp = *reinterpret_cast<const uint2*>(wr + k);
        float4 xv = *reinterpret_cast<const float4*>(x + k);
        if (silu_in) {
            xv.x = xv.x / (1.f + expf(-xv.x));
            xv.y = xv.y / (1.f + expf(-xv.y));
            xv.z = xv.z / (1.f + expf(-xv.z));
            xv.w = xv.w / (1.f + expf(-xv.w));
        }
        acc += xv.x * bf2f(wp.x & 0xffff) + xv.y * bf2f(wp.x >> 16) + xv.z * bf2f(wp.y & 0xffff) +
               xv.w * bf2f(wp.y >> 16);
    }
    acc = wave_sum(acc);
    if (lane == 0) y[n] = acc + (b ? bf2f(b[n]) : 0.f);
}

// mod[l][j][c] = float(modulation_l[j][c]) + e0[j][c]  (model.py:237-238) for every
// layer at once; `params` holds one device pointer per layer.
__global__ void modulation_kernel(const bf16_t* const* __restrict__ params, const float* __restrict__ e0,
                                  float* __restrict__ out, int n6c) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n6c) return;
    const bf16_t* p = params[blockIdx.y];
    out[(long)blockIdx.y * n6c + i] = bf2f(p[i]) + e0[i];
}

// head modulation: out[j][c] = float(mod[j][c]) + e[c], j in {0,1} (model.py:286)
__global__ void head_modulation_kernel(const bf16_t* __restrict__ mod, const float* __restrict__ e,
                                       float* __restrict__ out, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * C) return;
    out[i] = bf2f(mod[i]) + e[i % C];
}

// ---------------------------------------------------------------------------------
// Head projection in exact fp32 (v_mfma_f32_32x32x2_f32, an fmaf chain) fused with the
// unpatchify scatter 'fhwpqrc->cfphqwr' (model.py:288-289,526-549).
// y [M, C] fp32 (already normalised + modulated), W bf16 [NO, C], NO = Cout*pt*ph*pw = 64.
__global__ void __launch_bounds__(256) head_unpatchify_kernel(const float* __restrict__ y, long ldy,
                                                              const bf16_t* __restrict__ W,
                                                              const bf16_t* __restrict__ bias,
                                                              float* __restrict__ out, int n_tok, int C, int Cout,
                                                              int pt, int ph, int pw, int gf, int gh, int gw,
                                                              int tok_offset) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = (blockIdx.x * 4 + wave) * 32;
    if (row0 >= n_tok) return;
    const int i = lane & 31, kh = lane >> 5;
    int ar = row0 + i;
    ar = ar < n_tok ? ar : n_tok - 1;
    const float* ap = y + (long)ar * ldy + 4 * kh;
    const bf16_t* w0 = W + (long)i * C + 4 * kh;
    const bf16_t* w1 = W + (long)(32 + i) * C + 4 * kh;
    f32x16 acc0, acc1;
#pragma unroll
    for (int j = 0; j < 16; ++j) { acc0[j] = 0.f; acc1[j] = 0.f; }
    for (int k0 = 0; k0 < C; k0 += 8) {
        const float4 a = *reinterpret_cast<const float4*>(ap + k0);
        const uint2 b0 = *reinterpret_cast<const uint2*>(w0 + k0);
        const uint2 b1 = *reinterpret_cast<const uint2*>(w1 + k0);
        const float av[4] = {a.x, a.y, a.z, a.w};
        const float b0v[4] = {bf2f(b0.x & 0xffff), bf2f(b0.x >> 16), bf2f(b0.y & 0xffff), bf2f(b0.y >> 16)};
        const float b1v[4] = {bf2f(b1.x & 0xffff), bf2f(b1.x >> 16), bf2f(b1.y & 0xffff), bf2f(b1.y >> 16)};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], b0v[s], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], b1v[s], acc1, 0, 0, 0);
        }
    }
    MFMA_DRAIN();
    const int T_ = gf * pt, H = gh * ph, Wd = gw * pw;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int col = half * 32 + i;            // output feature
        const float bv = bias ? bf2f(bias[col]) : 0.f;
        const int c = col % Cout;
        const int pq = col / Cout;
        const int r = pq % pw, q = (pq / pw) % ph, p = pq / (pw * ph);
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int lr = row0 + (reg & 3) + 8 * (reg >> 2) + 4 * kh;
            const int tok = tok_offset + lr;          // y row lr holds token tok_offset + lr
            if (lr >= n_tok || tok >= gf * gh * gw) continue;
            const int w = tok % gw, h = (tok / gw) % gh, f = tok / (gw * gh);
            const float v = (half ? acc1[reg] : acc0[reg]) + bv;
            out[(((long)c * T_ + (f * pt + p)) * H + (h * ph + q)) * Wd + (w * pw + r)] = v;
        }
    }
}

__global__ void step_advance_kernel(int* step_ptr) { *step_ptr += 1; }

}  // namespace

int ovd_impl_patchify(const void* lat, int lat_is_bf16, bf16_t* rows, int Cin, int T, int H, int W, int pt, int ph,
                      int pw, hipStream_t st) {
    OVD_REQUIRE(pt > 0 && ph > 0 && pw > 0 && Cin > 0, "patchify: bad patch/channels");
    const int gf = T / pt, gh = H / ph, gw = W / pw;
    const long total = (long)gf * gh * gw * Cin * pt * ph * pw;
    if (total == 0) return 0;
    dim3 grid(ceil_div(total, 256)), blk(256);
    if (lat_is_bf16)
        hipLaunchKernelGGL(patchify_kernel<bf16_t>, grid, blk, 0, st, (const bf16_t*)lat, rows, Cin, T, H, W, pt, ph,
                           pw, gf, gh, gw, total);
    else
        hipLaunchKernelGGL(patchify_kernel<float>, grid, blk, 0, st, (const float*)lat, rows, Cin, T, H, W, pt, ph,
                           pw, gf, gh, gw, total);
    OVD_CHECK_LAUNCH("patchify");
    return 0;
}

int ovd_impl_time_sinusoid(const double* timesteps, const int* step_ptr, int step, float* out, int dim,
                           hipStream_t st) {
    OVD_REQUIRE(dim > 0 && dim % 2 == 0, "time_sinusoid: dim must be even");
    hipLaunchKernelGGL(time_sinusoid_kernel, dim3(ceil_div(dim / 2, 128)), dim3(128), 0, st, timesteps, step_ptr,
                       step, out, dim);
    OVD_CHECK_LAUNCH("time_sinusoid");
    return 0;
}

int ovd_impl_gemv_f32(const float* x, const bf16_t* W, const bf16_t* b, float* y, int N, int K, int silu_in,
                      hipStream_t st) {
    OVD_REQUIRE(K % 4 == 0 && N > 0, "gemv: K must be a multiple of 4");
    hipLaunchKernelGGL(gemv_f32_kernel, dim3(ceil_div(N, 4)), dim3(256), 0, st, x, W, b, y, N, K, silu_in);
    OVD_CHECK_LAUNCH("gemv_f32");
    return 0;
}

int ovd_impl_modulation(const bf16_t* const* params, const float* e0, float* out, int n_layers, int n6c,
                        hipStream_t st) {
    if (n_layers == 0) return 0;
    hipLaunchKernelGGL(modulation_kernel, dim3(ceil_div(n6c, 256), n_layers), dim3(256), 0, st, params, e0, out,
                       n6c);
    OVD_CHECK_LAUNCH("modulation");
    return 0;
}

int ovd_impl_head_modulation(const bf16_t* mod, const float* e, float* out, int C, hipStream_t st) {
    hipLaunchKernelGGL(head_modulation_kernel, dim3(ceil_div(2 * C, 256)), dim3(256), 0, st, mod, e, out, C);
    OVD_CHECK_LAUNCH("head_modulation");
    return 0;
}

int ovd_impl_head_unpatchify(const float* y, long ldy, const bf16_t* W, const bf16_t* bias, float* out, int n_tok,
                             int C, int Cout, int pt, int ph, int pw, int gf, int gh, int gw, int tok_offset,
                             hipStream_t st) {
    OVD_REQUIRE(Cout * pt * ph * pw == 64, "head: out_dim*prod(patch) must be 64, got %d", Cout * pt * ph * pw);
    OVD_REQUIRE(C % 8 == 0 && ldy % 4 == 0, "head: C must be a multiple of 8");
    OVD_REQUIRE(n_tok >= 0 && tok_offset >= 0 && tok_offset <= (long)gf * gh * gw,
                "head: rows %d at token offset %d outside grid %dx%dx%d", n_tok, tok_offset, gf, gh, gw);
    if (n_tok == 0) return 0;
    hipLaunchKernelGGL(head_unpatchify_kernel, dim3(ceil_div(n_tok, 128)), dim3(256), 0, st, y, ldy, W, bias, out,
                       n_tok, C, Cout, pt, ph, pw, gf, gh, gw, tok_offset);
    OVD_CHECK_LAUNCH("head_unpatchify");
    return 0;
}

int ovd_impl_step_advance(int* step_ptr, hipStream_t st) {
    hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(1), 0, st, step_ptr);
    OVD_CHECK_LAUNCH("step_advance");
    return 0;
}
