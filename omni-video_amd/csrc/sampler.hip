// Classifier-free-guidance combine + one UniPC predictor/corrector step in a single
// elementwise pass.  Compiled with -ffp-contract=off: every multiply and add rounds
// separately, in the order the reference's tensor expressions evaluate them
// (omnivideo/x2x_gen_unified.py:484; omnivideo/utils/fm_solvers_unipc.py:316-322
// convert_model_output, :610-626 corrector, :467-484 predictor, :706-724 history).
// All step-dependent scalars come from a host-built table (they depend only on the
// sigma schedule), one row per step, so the same launch replays inside a graph with a
// device-side step counter.
#include "common.h"

namespace {

// row layout of the coefficient table (floats)
enum {
    CF_SIGMA = 0,    // sigma_i
    CF_RESERVED,     // unused
    CF_USE_CORR,     // 1 if the corrector runs at this step
    CF_C_ORDER,      // corrector order (1 or 2)
    CF_C_R,          // sigma_t / sigma_s0
    CF_C_A,          // alpha_t * h_phi_1
    CF_C_AB,         // alpha_t * B_h
    CF_C_RK,         // r_k of the older history entry (order 2)
    CF_C_RHO0,       // rho for the history difference (order 2)
    CF_C_RHO1,       // rho for the current difference
    CF_P_ORDER,      // predictor order (1 or 2)
    CF_P_R,
    CF_P_A,
    CF_P_AB,
    CF_P_RK,
    CF_COUNT = 16
};

template <typename T> __device__ __forceinline__ float ldv(const T* p, long i);
template <> __device__ __forceinline__ float ldv<float>(const float* p, long i) { return p[i]; }
template <> __device__ __forceinline__ float ldv<bf16_t>(const bf16_t* p, long i) { return bf2f(p[i]); }
template <typename T> __device__ __forceinline__ float rnd(float v);
template <> __device__ __forceinline__ float rnd<float>(float v) { return v; }
template <> __device__ __forceinline__ float rnd<bf16_t>(float v) { return round_bf(v); }
template <typename T> __device__ __forceinline__ void stv(T* p, long i, float v);
template <> __device__ __forceinline__ void stv<float>(float* p, long i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void stv<bf16_t>(bf16_t* p, long i, float v) { p[i] = f2bf(v); }

template <typename T>
__global__ void __launch_bounds__(256) cfg_unipc_kernel(const float* __restrict__ cond,
                                                        const float* __restrict__ uncond, T* __restrict__ sample,
                                                        T* __restrict__ last_sample, float* __restrict__ m1,
                                                        float* __restrict__ m2, const float* __restrict__ coef, float guide,
                                                        const int* __restrict__ step_ptr, int step, long n) {
    const int s = step_ptr ? *step_ptr : step;
    const float* cf = coef + (long)s * CF_COUNT;
    const float sigma = cf[CF_SIGMA];
    const bool use_corr = cf[CF_USE_CORR] != 0.f;
    const int c_order = (int)cf[CF_C_ORDER], p_order = (int)cf[CF_P_ORDER];
    const float c_r = cf[CF_C_R], c_a = cf[CF_C_A], c_ab = cf[CF_C_AB], c_rk = cf[CF_C_RK];
    const float rho0 = cf[CF_C_RHO0], rho1 = cf[CF_C_RHO1];
    const float p_r = cf[CF_P_R], p_a = cf[CF_P_A], p_ab = cf[CF_P_AB], p_rk = cf[CF_P_RK];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float c = cond[i], u = uncond[i];
        const float v = u + guide * (c - u);
        const float xs = ldv<T>(sample, i);
        const float m_t = xs - sigma * v;
        const float m0 = m1[i];
        float xc = xs;
        if (use_corr) {
            const float t1 = rnd<T>(c_r * ldv<T>(last_sample, i));
            const float x_t_ = t1 - c_a * m0;
            const float d1t = m_t - m0;
            float res;
            if (c_order == 1) {
                res = rho1 * d1t;
            } else {
                const float d1 = (m2[i] - m0) / c_rk;
                res = rho0 * d1 + rho1 * d1t;
            }
            xc = rnd<T>(x_t_ - c_ab * res);
        }
        const float t1 = rnd<T>(p_r * xc);
        float x_t = t1 - p_a * m_t;
        if (p_order == 2) {
            const float d1 = (m0 - m_t) / p_rk;
            x_t = x_t - p_ab * (0.5f * d1);
        }
        stv<T>(last_sample, i, xc);
        stv<T>(sample, i, x_t);
        m2[i] = m0;
        m1[i] = m_t;
    }
}

// Classifier-free-guidance combine + one DPM-Solver++ (order <= 2, midpoint) step
// (omnivideo/utils/fm_solvers.py:383-385 convert_model_output, :455-466 first order,
// :527-560 second order, :748-751 history).  Row layout of its coefficient table:
enum { DP_SIGMA = 0, DP_ORDER, DP_RATIO, DP_B, DP_HALF_B, DP_INV_R0, DP_COUNT = 8 };

template <typename T>
__global__ void __launch_bounds__(256) cfg_dpm_kernel(const float* __restrict__ cond, const float* __restrict__ uncond,
                                                      const T* __restrict__ sample, float* __restrict__ next,
                                                      float* __restrict__ m1, const float* __restrict__ coef,
                                                      float guide, const int* __restrict__ step_ptr, int step,
                                                      long n) {
    const int s = step_ptr ? *step_ptr : step;
    const float* cf = coef + (long)s * DP_COUNT;
    const float sigma = cf[DP_SIGMA], ratio = cf[DP_RATIO], b = cf[DP_B], half_b = cf[DP_HALF_B];
    const float inv_r0 = cf[DP_INV_R0];
    const bool second = cf[DP_ORDER] == 2.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float c = cond[i], u = uncond[i];
        const float v = u + guide * (c - u);
        const float xs = ldv<T>(sample, i);
        const float x0 = xs - sigma * v;
        const float t1 = ratio * xs;
        float x = t1 - b * x0;
        if (second) {
            const float d1 = inv_r0 * (x0 - m1[i]);
            x = x - half_b * d1;
        }
        m1[i] = x0;
        next[i] = x;
    }
}

}  // namespace

int ovd_impl_cfg_dpm_step(const float* cond, const float* uncond, const void* sample, float* next, float* m1,
                          const float* coef, float guide, const int* step_ptr, int step, long n, int sample_is_bf16,
                          hipStream_t st) {
    OVD_REQUIRE(n >= 0, "cfg_dpm: negative size");
    OVD_REQUIRE(!sample_is_bf16 || (const void*)next != sample, "cfg_dpm: a bf16 sample cannot be updated in place");
    if (n == 0) return 0;
    int blocks = ceil_div(n, 256);
    if (blocks > 2048) blocks = 2048;
    if (sample_is_bf16)
        hipLaunchKernelGGL(cfg_dpm_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, cond, uncond, (const bf16_t*)sample,
                           next, m1, coef, guide, step_ptr, step, n);
    else
        hipLaunchKernelGGL(cfg_dpm_kernel<float>, dim3(blocks), dim3(256), 0, st, cond, uncond, (const float*)sample,
                           next, m1, coef, guide, step_ptr, step, n);
    OVD_CHECK_LAUNCH("cfg_dpm_step");
    return 0;
}

int ovd_impl_cfg_unipc_step(const float* cond, const float* uncond, void* sample, void* last_sample, float* m1,
                            float* m2, const float* coef, float guide, const int* step_ptr, int step, long n,
                            int sample_is_bf16, hipStream_t st) {
    OVD_REQUIRE(n >= 0, "cfg_unipc: negative size");
    if (n == 0) return 0;
    int blocks = ceil_div(n, 256);
    if (blocks > 2048) blocks = 2048;
    if (sample_is_bf16)
        hipLaunchKernelGGL(cfg_unipc_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, cond, uncond, (bf16_t*)sample,
                           (bf16_t*)last_sample, m1, m2, coef, guide, step_ptr, step, n);
    else
        hipLaunchKernelGGL(cfg_unipc_kernel<float>, dim3(blocks), dim3(256), 0, st, cond, uncond, (float*)sample,
                           (float*)last_sample, m1, m2, coef, guide, step_ptr, step, n);
    OVD_CHECK_LAUNCH("cfg_unipc_step");
    return 0;
}
