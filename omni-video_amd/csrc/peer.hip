// Sequence-parallel exchange through peer-mapped device memory (one process per GPU, xGMI full mesh).
//
// Every rank of a sequence-parallel group maps the exchange buffers of all the others into its own
// address space once (HIP IPC), and the exchange itself is ordinary kernels on the caller's stream:
//   * peer_barrier_kernel: a rank tells every peer "everything I launched before this point is done" by
//     storing an epoch number into its slot of the peer's flag array, and waits until its own array shows
//     that epoch for every peer.  The flag arrays live in fine-grained memory (stores by another device
//     are visible to a running kernel); the wait is bounded, a rank that never arrives sets a status word
//     instead of hanging the device.
//   * peer_pull_kernel: the all-to-all of Ulysses attention as p concurrent reads, one per xGMI link -
//     rank r copies chunk r of every peer's send buffer into its receive buffer.
//   * peer_sum_f32_kernel: out = the sum over ranks of a buffer every rank published (summed in rank
//     order on every rank, so all ranks hold the same bits).
// Nothing here calls a collective library, allocates or synchronises with the host, so a whole sampling
// step of a rank - exchanges included - is one HIP graph.
//
// Ordering contract (what makes plain device memory enough for the payload): a payload buffer is written
// by kernels that finish before the writer's next barrier launch, and read by kernels that start after
// the reader's matching barrier launch has finished; a buffer that peers read is not rewritten until its
// owner has passed one more barrier.
#include "common.h"

static_assert(sizeof(hipIpcMemHandle_t) == 64, "the C-ABI passes IPC handles as 64 opaque bytes");

// ------------------------------------------------------------------------------ set-up (host side)
int ovd_impl_peer_alloc(void** ptr, size_t bytes, int fine_grained) {
    OVD_REQUIRE(ptr != nullptr && bytes > 0, "peer_alloc: bad arguments");
    // hipIpcOpenMemHandle (ROCm 7.2, dmabuf IPC) never returns for an allocation whose size has bit 31 set in its
    // low 32 bits - 2 GiB hangs, 2 GiB - 4 KiB and 4 GiB map at once (probed with tools/ipc_size_probe.py) - so
    // such a size is rounded up to the next multiple of 4 GiB.
    if ((bytes & 0xFFFFFFFFull) >= 0x80000000ull) bytes = (bytes + 0xFFFFFFFFull) & ~(size_t)0xFFFFFFFFull;
    void* p = nullptr;
    const hipError_t e = hipExtMallocWithFlags(&p, bytes, fine_grained ? hipDeviceMallocFinegrained : hipDeviceMallocDefault);
    if (e != hipSuccess) OVD_FAIL("peer_alloc: %zu bytes (%s): %s", bytes, fine_grained ? "fine-grained" : "default",
                                  hipGetErrorString(e));
    if (hipMemset(p, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        (void)hipFree(p);
        OVD_FAIL("peer_alloc: clearing the buffer failed");
    }
    *ptr = p;
    return 0;
}

int ovd_impl_peer_free(void* ptr) {
    const hipError_t e = hipFree(ptr);
    if (e != hipSuccess) OVD_FAIL("peer_free: %s", hipGetErrorString(e));
    return 0;
}

int ovd_impl_peer_export(void* ptr, unsigned char* handle) {
    hipIpcMemHandle_t h;
    const hipError_t e = hipIpcGetMemHandle(&h, ptr);
    if (e != hipSuccess) OVD_FAIL("peer_export: hipIpcGetMemHandle: %s", hipGetErrorString(e));
    memcpy(handle, &h, sizeof(h));
    return 0;
}

int ovd_impl_peer_open(const unsigned char* handle, void** ptr) {
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    void* p = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) OVD_FAIL("peer_open: hipIpcOpenMemHandle: %s", hipGetErrorString(e));
    *ptr = p;
    return 0;
}

int ovd_impl_peer_close(void* ptr) {
    const hipError_t e = hipIpcCloseMemHandle(ptr);
    if (e != hipSuccess) OVD_FAIL("peer_close: %s", hipGetErrorString(e));
    return 0;
}

// ------------------------------------------------------------------------------ barrier
// One lane per peer.  A rank's flag array holds 2 x 64 words: [j] = the epoch peer j has reached, [64 + j] =
// non-zero once peer j has given up on a barrier.  state[0] = epoch of the last barrier this rank passed,
// state[1] = 0, or 1 + j when peer j did not arrive in time here, or 65 + j when peer j reported that one of
// ITS barriers failed (sticky, first cause wins).  Epochs only grow and a rank can be at most one barrier
// ahead of a peer, so "flag - epoch >= 0" in wrapping arithmetic is the arrival test.
// A rank whose wait times out goes on unsynchronised, so everything the group computes from then on is
// invalid - on every rank, not only on this one.  It therefore marks itself as failed in every peer's array
// (now and again in every later barrier), and a rank that finds a peer's mark stops waiting for it and records
// the failure too: within one barrier of the first time-out every rank's status word is set.
__global__ void __launch_bounds__(64) peer_barrier_kernel(uint32_t* my_flags, uint32_t* const* peer_flags,
                                                          uint32_t* state, int p, int rank, long long timeout_ticks) {
    __shared__ uint32_t e_sh, bad_sh;
    const int j = threadIdx.x;
    if (j == 0) {
        e_sh = state[0] + 1u;
        bad_sh = state[1];
    }
    __syncthreads();
    const uint32_t e = e_sh;
    const bool failed_before = bad_sh != 0u;
    if (j < p) {
        if (failed_before)
            __hip_atomic_store(&peer_flags[j][64 + rank], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __threadfence_system();
        __hip_atomic_store(&peer_flags[j][rank], e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        const long long t0 = wall_clock64();
        while ((int32_t)(__hip_atomic_load(&my_flags[j], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - e) < 0) {
            if (__hip_atomic_load(&my_flags[64 + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) break;
            if (wall_clock64() - t0 > timeout_ticks) {
                atomicCAS(&state[1], 0u, (uint32_t)(j + 1));
                break;
            }
            __builtin_amdgcn_s_sleep(16);
        }
        // a peer that failed earlier may be ahead of this rank: its epochs are all there, its data are not
        if (__hip_atomic_load(&my_flags[64 + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u)
            atomicCAS(&state[1], 0u, (uint32_t)(65 + j));
    }
    __syncthreads();
    if (j < p && !failed_before && __hip_atomic_load(&state[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)
        __hip_atomic_store(&peer_flags[j][64 + rank], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (j == 0) state[0] = e;
}

int ovd_impl_peer_barrier(uint32_t* my_flags, uint32_t* const* peer_flags, uint32_t* state, int p, int rank,
                          double timeout_s, hipStream_t st) {
    OVD_REQUIRE(p >= 1 && p <= 64 && rank >= 0 && rank < p, "peer_barrier: bad group p=%d rank=%d", p, rank);
    OVD_REQUIRE(timeout_s > 0.0 && timeout_s <= 600.0, "peer_barrier: timeout %g s outside (0, 600]", timeout_s);
    int rate_khz = 0;             // wall_clock64() ticks at this rate (100 MHz on MI355X)
    if (hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, ovd_current_device()) != hipSuccess ||
        rate_khz <= 0)
        rate_khz = 100000;
    const long long ticks = (long long)(timeout_s * 1e3 * (double)rate_khz);
    hipLaunchKernelGGL(peer_barrier_kernel, dim3(1), dim3(64), 0, st, my_flags, peer_flags, state, p, rank, ticks);
    OVD_CHECK_LAUNCH("peer_barrier");
    return 0;
}

// ------------------------------------------------------------------------------ all-to-all (pull)
// recv chunk j <- chunk `rank` of peer j's send buffer; chunks are n16 16-byte words.  blockIdx.y walks the
// peers starting behind this rank, so that at any time the ranks read from different peers.
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
constexpr int PULL_UNROLL = 4;
__global__ void __launch_bounds__(256) peer_pull_kernel(u32x4* __restrict__ recv, const u32x4* const* peer_send,
                                                        long n16, int p, int rank) {
    const int j = (rank + 1 + blockIdx.y) % p;
    const u32x4* __restrict__ src = peer_send[j] + (long)rank * n16;
    u32x4* __restrict__ dst = recv + (long)j * n16;
    const long stride = (long)gridDim.x * blockDim.x;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (PULL_UNROLL - 1) * stride < n16; i += PULL_UNROLL * stride) {
        u32x4 v[PULL_UNROLL];
#pragma unroll
        for (int u = 0; u < PULL_UNROLL; ++u) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < PULL_UNROLL; ++u) dst[i + u * stride] = v[u];
    }
    for (; i < n16; i += stride) dst[i] = __builtin_nontemporal_load(src + i);
}

int ovd_impl_peer_all_to_all(void* recv, const void* const* peer_send, long chunk_bytes, int p, int rank,
                             hipStream_t st) {
    OVD_REQUIRE(p >= 1 && p <= 64 && rank >= 0 && rank < p, "peer_all_to_all: bad group p=%d rank=%d", p, rank);
    OVD_REQUIRE(chunk_bytes > 0 && chunk_bytes % 16 == 0, "peer_all_to_all: chunk of %ld bytes is not a multiple of 16",
                chunk_bytes);
    OVD_REQUIRE(((uintptr_t)recv & 15) == 0, "peer_all_to_all: recv is not 16-byte aligned");
    const long n16 = chunk_bytes / 16;
    long bx = (n16 + 256L * PULL_UNROLL - 1) / (256L * PULL_UNROLL);
    const long cap = (4L * ovd_device_cus() + p - 1) / p;      // about four workgroups per CU over all peers
    if (bx > cap) bx = cap;
    hipLaunchKernelGGL(peer_pull_kernel, dim3((unsigned)bx, (unsigned)p), dim3(256), 0, st, (u32x4*)recv,
                       (const u32x4* const*)peer_send, n16, p, rank);
    OVD_CHECK_LAUNCH("peer_all_to_all");
    return 0;
}

// ------------------------------------------------------------------------------ sum over ranks
__global__ void __launch_bounds__(256) peer_sum_f32_kernel(float* __restrict__ out, const float* const* peer_src, long n,
                                                           int p) {
    const long stride = (long)gridDim.x * blockDim.x;
    const long n4 = n / 4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f32x4 a = __builtin_nontemporal_load((const f32x4*)peer_src[0] + i);
        for (int j = 1; j < p; ++j) a += __builtin_nontemporal_load((const f32x4*)peer_src[j] + i);
        ((f32x4*)out)[i] = a;
    }
    for (long i = n4 * 4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float a = peer_src[0][i];
        for (int j = 1; j < p; ++j) a += peer_src[j][i];
        out[i] = a;
    }
}

int ovd_impl_peer_sum_f32(float* out, const float* const* peer_src, long n, int p, hipStream_t st) {
    OVD_REQUIRE(p >= 1 && p <= 64 && n >= 0, "peer_sum: bad arguments p=%d n=%ld", p, n);
    OVD_REQUIRE(((uintptr_t)out & 15) == 0, "peer_sum: out is not 16-byte aligned");
    if (n == 0) return 0;
    long bx = (n / 4 + 255) / 256;
    if (bx < 1) bx = 1;
    if (bx > 4L * ovd_device_cus()) bx = 4L * ovd_device_cus();
    hipLaunchKernelGGL(peer_sum_f32_kernel, dim3((unsigned)bx), dim3(256), 0, st, out, peer_src, n, p);
    OVD_CHECK_LAUNCH("peer_sum");
    return 0;
}
