// Frame pre- and post-processing around the VAE: byte and integer work, HBM-bound.
//  - frames -> encoder input: centre crop, Pillow-exact 8-bit bilinear resize (two separable
//    passes with 22-bit fixed-point coefficients and a uint8 intermediate), then the
//    ToTensor/Normalize map through a 256-entry table
//    (reference: tools/inference/generate_omni_v2v.py:306-343)
//  - decoder output -> uint8 frames (reference: omnivideo/utils/utils.py:175-184)
#include "common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ int clip8(int acc) {
    const int v = acc >> PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// Horizontal pass over the crop window: tmp[t][y][xx][c], y in [0, crop_h), xx in [0, out_w).
// bounds[xx] = (first source column inside the crop, tap count), coef[xx][ksize].
__global__ void __launch_bounds__(256) resize_h_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ tmp,
                                                       const int* __restrict__ bounds, const int* __restrict__ coef,
                                                       int ksize, int H, int W, int top, int left, int crop_h,
                                                       int crop_w, int out_w, long total) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % 3);
    long r = idx / 3;
    const int xx = (int)(r % out_w);
    r /= out_w;
    const int y = (int)(r % crop_h);
    const long t = r / crop_h;
    const int x0 = max(bounds[2 * xx], 0), n = min(bounds[2 * xx + 1], min(ksize, crop_w - x0));
    const uint8_t* row = in + ((t * H + top + y) * W + left + x0) * 3 + c;
    const int* k = coef + (long)xx * ksize;
    int acc = 1 << (PRECISION_BITS - 1);
    for (int x = 0; x < n; ++x) acc += (int)row[3 * x] * k[x];
    tmp[idx] = (uint8_t)clip8(acc);
}

// Vertical pass + table lookup: out[c][t][yy][xx] (PLANAR) = lut[clip8(sum_y tmp[t][y0+y][xx][c] * k[y])]
__global__ void __launch_bounds__(256) resize_v_kernel(const uint8_t* __restrict__ tmp, float* __restrict__ out,
                                                       const int* __restrict__ bounds, const int* __restrict__ coef,
                                                       const float* __restrict__ lut, int ksize, int crop_h, int out_h,
                                                       int out_w, long T, long total) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int xx = (int)(idx % out_w);
    long r = idx / out_w;
    const int yy = (int)(r % out_h);
    r /= out_h;
    const long t = r % T;
    const int c = (int)(r / T);
    const int y0 = max(bounds[2 * yy], 0), n = min(bounds[2 * yy + 1], min(ksize, crop_h - y0));
    const uint8_t* col = tmp + ((t * crop_h + y0) * out_w + xx) * 3 + c;
    const int* k = coef + (long)yy * ksize;
    int acc = 1 << (PRECISION_BITS - 1);
    for (int y = 0; y < n; ++y) acc += (int)col[(long)y * out_w * 3] * k[y];
    out[idx] = lut[clip8(acc)];
}

// out[v][c] = uint8(255 * (clamp(x, lo, hi) - lo) / max(hi - lo, 1e-5)), truncating, in the fp32
// operation order of the reference (clamp; make_grid's normalize = sub, div; times 255; cast).
// x[v][c] is read at v*vs + c*cs: channels-last decoder output or a planar [C,T,H,W] tensor.
__global__ void __launch_bounds__(256) to_uint8_kernel(const float* __restrict__ in, uint8_t* __restrict__ out, long vs,
                                                       long cs, int C, long nvox, float lo, float hi) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nvox * C) return;
    const long v = idx / C;
    const int c = (int)(idx % C);
    float val = fminf(fmaxf(in[v * vs + c * cs], lo), hi);
    val = (val - lo) / fmaxf(hi - lo, 1e-5f);
    out[idx] = (uint8_t)(val * 255.f);
}

}  // namespace

// Host-side tables of one resize axis, in Pillow's arithmetic (Resample.c precompute_coeffs with
// the bilinear filter, then normalize_coeffs_8bpc).  bounds [out_size][2] = (first tap, tap
// count), coef [out_size][ksize]; returns ksize, or writes nothing and returns the needed ksize
// when `ksize_cap` is too small (call with ksize_cap = 0 to size the buffers).
int ovd_impl_resize_coeffs(int in_size, int out_size, int* bounds, int* coef, int ksize_cap) {
    const double scale = (double)in_size / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    if (ksize_cap < ksize || !bounds || !coef) return ksize;
    const double ss = 1.0 / filterscale;
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double w[64], ww = 0.0;
        int* k = coef + (long)xx * ksize_cap;
        for (int x = 0; x < ksize_cap; ++x) k[x] = 0;
        if (xmax > 64) xmax = 64;                   // ksize <= 64 is checked by the caller
        for (int x = 0; x < xmax; ++x) {
            double a = (x + xmin - center + 0.5) * ss;
            if (a < 0.0) a = -a;
            w[x] = a < 1.0 ? 1.0 - a : 0.0;
            ww += w[x];
        }
        for (int x = 0; x < xmax; ++x) {
            const double v = ww != 0.0 ? w[x] / ww : w[x];
            k[x] = v < 0 ? (int)(-0.5 + v * (1 << PRECISION_BITS)) : (int)(0.5 + v * (1 << PRECISION_BITS));
        }
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
    return ksize;
}

int ovd_impl_frames_resize(const uint8_t* in, uint8_t* tmp, float* out, const int* bounds_w, const int* coef_w,
                           int ksize_w, const int* bounds_h, const int* coef_h, int ksize_h, const float* lut, long T,
                           int H, int W, int top, int left, int crop_h, int crop_w, int out_h, int out_w,
                           hipStream_t st) {
    OVD_REQUIRE(T >= 0 && H > 0 && W > 0 && crop_h > 0 && crop_w > 0 && out_h > 0 && out_w > 0,
                "frames_resize: empty geometry");
    OVD_REQUIRE(top >= 0 && left >= 0 && top + crop_h <= H && left + crop_w <= W,
                "frames_resize: crop window (%d,%d,%d,%d) outside the %dx%d frame", top, left, crop_h, crop_w, H, W);
    OVD_REQUIRE(ksize_w > 0 && ksize_h > 0, "frames_resize: empty coefficient tables");
    if (T == 0) return 0;
    const long n1 = T * crop_h * (long)out_w * 3, n2 = 3L * T * out_h * out_w;
    hipLaunchKernelGGL(resize_h_kernel, dim3(ceil_div(n1, 256)), dim3(256), 0, st, in, tmp, bounds_w, coef_w, ksize_w, H,
                       W, top, left, crop_h, crop_w, out_w, n1);
    OVD_CHECK_LAUNCH("frames_resize (horizontal)");
    hipLaunchKernelGGL(resize_v_kernel, dim3(ceil_div(n2, 256)), dim3(256), 0, st, tmp, out, bounds_h, coef_h, lut,
                       ksize_h, crop_h, out_h, out_w, T, n2);
    OVD_CHECK_LAUNCH("frames_resize (vertical)");
    return 0;
}

int ovd_impl_to_uint8(const float* in, uint8_t* out, long vs, long cs, int C, long nvox, float lo, float hi,
                      hipStream_t st) {
    OVD_REQUIRE(C > 0 && nvox >= 0 && hi > lo, "to_uint8: bad arguments");
    if (nvox == 0) return 0;
    hipLaunchKernelGGL(to_uint8_kernel, dim3(ceil_div(nvox * C, 256)), dim3(256), 0, st, in, out, vs, cs, C, nvox, lo, hi);
    OVD_CHECK_LAUNCH("to_uint8");
    return 0;
}
