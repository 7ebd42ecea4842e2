// Shared device/host helpers for libomnihip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>

typedef uint16_t bf16_t;  // raw bfloat16 bit pattern

typedef __attribute__((ext_vector_type(8))) short bf16x8;   // 8 bf16 (one MFMA A/B fragment)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// ---- error channel -------------------------------------------------------------
void ovd_set_error(const char* fmt, ...);
#define OVD_FAIL(...)          \
    do {                       \
        ovd_set_error(__VA_ARGS__); \
        return 1;              \
    } while (0)
#define OVD_REQUIRE(cond, ...) \
    do {                       \
        if (!(cond)) OVD_FAIL(__VA_ARGS__); \
    } while (0)
#define OVD_CHECK_LAUNCH(name)                                              \
    do {                                                                    \
        hipError_t e__ = hipGetLastError();                                 \
        if (e__ != hipSuccess) OVD_FAIL("%s: launch failed: %s", name, hipGetErrorString(e__)); \
    } while (0)

// ---- bf16 <-> f32 ---------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// round-to-nearest-even through the hardware convert (v_cvt_pk_bf16_f32); NaN stays NaN
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ float round_bf(float f) { return bf2f(f2bf(f)); }

__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;
    f32x2_ v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float gelu_tanh_f(float x) {
    // 0.5 x (1 + tanh(u)), u = sqrt(2/pi) (x + 0.044715 x^3), written as x / (1 + e^{-2u}) (the same
    // function: 0.5 (1 + tanh u) is the logistic function of 2u): one v_exp_f32, one v_rcp_f32 and five
    // multiply-adds.  e^{-2u} overflows to +inf for u << 0 and the quotient is then x * 0 = -0, the
    // limit of the function; absolute error ~1e-7 |x|, the size of the fp32 roundings of the usual form.
    const float c0 = 2.0f * 0.7978845608028654f * 1.4426950408889634f;      // 2 sqrt(2/pi) log2(e)
    const float c1 = c0 * 0.044715f;
    const float u2 = x * __builtin_fmaf(c1, x * x, c0);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-u2));
}

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// ---- per-device launcher state ----------------------------------------------------
// Launchers may be called from several host threads at once (ctypes releases the GIL) and for
// several devices of one process, so what they cache is atomic and keyed by the current device.
constexpr int OVD_MAX_DEVICES = 64;
static inline int ovd_current_device() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= OVD_MAX_DEVICES) d = 0;
    return d;
}

// A kernel's dynamic-LDS limit (hipFuncAttributeMaxDynamicSharedMemorySize is a per-device attribute):
// raised on the first launch on each device; repeating the call from a racing thread is harmless.
struct OvdLdsLimit {
    std::atomic<size_t> cur[OVD_MAX_DEVICES];
    bool ensure(const void* fn, size_t bytes) {
        std::atomic<size_t>& c = cur[ovd_current_device()];
        size_t old = c.load(std::memory_order_acquire);
        if (old >= bytes) return true;
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return false;
        while (old < bytes && !c.compare_exchange_weak(old, bytes, std::memory_order_release)) {}
        return true;
    }
};

// compute units of the current device (256 on MI355X), looked up once per device
static inline int ovd_device_cus() {
    static std::atomic<int> cus[OVD_MAX_DEVICES];
    const int d = ovd_current_device();
    int c = cus[d].load(std::memory_order_relaxed);
    if (c == 0) {
        // OVD_DEVICE_CUS: plan for another compute-unit count than the device reports (tests of the planning
        // arithmetic on shapes a larger part would produce; every launch stays valid, only the balance changes)
        const char* e = getenv("OVD_DEVICE_CUS");
        if (e != nullptr && atoi(e) > 0) c = atoi(e);
        else if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || c <= 0) c = 256;
        cus[d].store(c, std::memory_order_relaxed);
    }
    return c;
}

// integer tuning switch from the environment, read once per process (`cache` starts at INT32_MIN)
static inline int ovd_env_int(std::atomic<int>& cache, const char* name, int dflt) {
    int v = cache.load(std::memory_order_relaxed);
    if (v == INT32_MIN) {
        const char* e = getenv(name);
        v = e ? atoi(e) : dflt;
        if (v == INT32_MIN) v = dflt;
        cache.store(v, std::memory_order_relaxed);
    }
    return v;
}

// Wait states covering the longest MFMA (16 passes) before its results are read by
// non-MFMA instructions on a path where the compiler's hazard padding may not apply.
#define MFMA_DRAIN()                                        \
    do {                                                    \
        __builtin_amdgcn_sched_barrier(0);                  \
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); \
        __builtin_amdgcn_sched_barrier(0);                  \
    } while (0)
