// Developer probe (not part of the library): can a decode step live on ONE XCD (32 CUs behind one L2) as a persistent kernel?
//   hipcc -O3 --offload-arch=gfx950 tools/xcd_probe.hip -o gpurun_out/xcd_probe && gpurun_out/xcd_probe
// Measures (a) where workgroup b of a launch lands (XCC_ID register): is it b % 8; (b) a barrier among the workgroups of one XCD
// through its L2 (workgroup-scope atomics execute in the L2, sc0 loads miss the per-CU cache) with and without a one-word
// exchange; (c) a two-level barrier over all XCDs (L2 counter per XCD, the last arrival of an XCD goes to a device counter);
// (d) the rate at which the workgroups of ONE XCD stream weights from HBM. Every spin is bounded.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

__global__ void where_kernel(unsigned* out) {
    if (threadIdx.x == 0) out[blockIdx.x] = xcc_id();
}

#define SPIN_MAX (1u << 20)

// barrier among the `n` participants that share one L2: arrival = L2 atomic, wait = sc0 loads of the same word
// poll: 0 = sc0 load (workgroup scope), 1 = L2 atomic add of 0 (always executes in the L2), 2 = sc1 load (agent scope)
__device__ __forceinline__ unsigned poll_word(unsigned* cnt, int poll) {
    if (poll == 0) return __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (poll == 1) return __hip_atomic_fetch_add(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (poll == 3) {                                                    // an L2 atomic that returns the value (the compiler turns fetch_add(0) into a load)
        unsigned v = 0;
        asm volatile("global_atomic_add %0, %1, %0, off sc0\n\ts_waitcnt vmcnt(0)" : "+v"(v) : "v"(cnt) : "memory");
        return v;
    }
    return __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool l2_barrier(unsigned* cnt, unsigned target, unsigned* err, int poll) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        unsigned spins = 0;
        while (poll_word(cnt, poll) < target) {
            if (++spins > SPIN_MAX) { *err = 1; err[3] = __hip_atomic_fetch_add(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); ok = false; break; }
        }
    }
    __syncthreads();
    return ok;
}

// mode 0: barrier only; 1: + every participant publishes a word (plain store: write-through to the L2) and reads its neighbour's
// with an sc0 load after the barrier; 2: + 16 bytes per lane of weights per phase requested before the barrier
__global__ __launch_bounds__(1024) void one_xcd_kernel(unsigned* cnt, unsigned* err, unsigned* slots, const uint4* W, unsigned* sink, int nph, int mode, int npart, int poll) {
    if (blockIdx.x % 8 != 0) return;
    const unsigned me = blockIdx.x / 8;
    if (threadIdx.x == 0 && xcc_id() != 0) atomicAdd(err + 2, 1u);
    unsigned acc = 0;
    for (int p = 0; p < nph; ++p) {
        uint4 w = {0, 0, 0, 0};
        if (mode == 2) w = W[((size_t)(p % 64) * npart + me) * blockDim.x + threadIdx.x];
        if (mode >= 1 && threadIdx.x == 0) {
            slots[(p & 1) * npart + me] = (unsigned)(p * 7919u + me);
            __builtin_amdgcn_s_waitcnt(0);
        }
        if (!l2_barrier(cnt, (unsigned)(p + 1) * npart, err, poll)) return;
        if (mode >= 1 && threadIdx.x == 0) {
            const unsigned nb = (me + 1) % npart;
            const unsigned got = poll_word(slots + (p & 1) * npart + nb, poll);
            if (got != (unsigned)(p * 7919u + nb)) atomicAdd(err + 1, 1u);
        }
        acc += w.x ^ w.y ^ w.z ^ w.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

__device__ __forceinline__ unsigned plain_load(const unsigned* p) {             // a load with no cache-control bit (volatile would set sc0 sc1)
    unsigned v;
    asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// How may the data another participant published be read after the barrier? xmode 0: sc1 (agent-scope) loads; 1: buffer_inv sc0, then
// plain loads; 2: buffer_inv sc1, then plain loads; 3: plain loads and nothing else (control: must go stale). Every participant reads its
// neighbour's 16-word row BEFORE the barrier as well (plain loads: the line then sits in its per-CU cache) and publishes a fresh row.
__global__ __launch_bounds__(512) void exchange_kernel(unsigned* cnt, unsigned* err, unsigned* rows_, unsigned* sink, int nph, int xmode, int npart) {
    if (blockIdx.x % 8 != 0) return;
    const unsigned me = blockIdx.x / 8, nb = (me + 1) % npart;
    unsigned acc = 0;
    const int poll = xmode >= 4 ? 3 : 2;
    for (int p = 0; p < nph; ++p) {
        unsigned* rows = xmode >= 4 ? rows_ + (size_t)p * npart * 32 : rows_;           // xmode 4 / 5: every phase has its own rows (128-byte lines of their own): nothing of them can sit in a per-CU cache
        if (threadIdx.x < 16) {
            if (xmode < 4) acc += plain_load(rows + nb * 32 + threadIdx.x);             // warm the per-CU cache with the OLD row
            rows[me * 32 + threadIdx.x] = (unsigned)(p * 7919u + me * 16 + threadIdx.x);
        }
        __builtin_amdgcn_s_waitcnt(0);
        if (!l2_barrier(cnt, (unsigned)(p + 1) * npart, err, poll)) return;
        if (xmode == 1) asm volatile("buffer_inv sc0" ::: "memory");
        if (xmode == 2) asm volatile("buffer_inv sc1" ::: "memory");
        if (threadIdx.x < 16) {
            unsigned got;
            if (xmode == 0) got = __hip_atomic_load(rows + nb * 32 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else got = plain_load(rows + nb * 32 + threadIdx.x);
            if (got != (unsigned)(p * 7919u + nb * 16 + threadIdx.x)) atomicAdd(err + 1, 1u);
        }
        if (xmode == 5) continue;                                                       // fresh rows need no second barrier
        if (!l2_barrier(cnt + 64, (unsigned)(p + 1) * npart, err, poll)) return;        // nobody overwrites a row that is still being read
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// two-level barrier over every XCD: cnt_l2[x * 32] is the L2 counter of XCD x, cnt_dev the device counter
__global__ __launch_bounds__(256) void two_level_kernel(unsigned* cnt_l2, unsigned* cnt_dev, unsigned* err, unsigned* slots, int nph, int per_xcd, int mode) {
    const unsigned x = blockIdx.x % 8, nwg = gridDim.x;
    for (int p = 0; p < nph; ++p) {
        if (mode >= 1 && threadIdx.x == 0) {
            __hip_atomic_store(slots + (p & 1) * nwg + blockIdx.x, (unsigned)(p * 7919u + blockIdx.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_s_waitcnt(0);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned old = __hip_atomic_fetch_add(cnt_l2 + x * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (old == (unsigned)(p + 1) * per_xcd - 1) __hip_atomic_fetch_add(cnt_dev, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            while (__hip_atomic_load(cnt_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(p + 1) * 8) {
                if (++spins > SPIN_MAX) { *err = 1; break; }
            }
        }
        __syncthreads();
        if (*(volatile unsigned*)err) return;
        if (mode >= 1 && threadIdx.x == 0) {
            const unsigned nb = (blockIdx.x + 1) % nwg;
            const unsigned got = __hip_atomic_load(slots + (p & 1) * nwg + nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (got != (unsigned)(p * 7919u + nb)) atomicAdd(err + 1, 1u);
        }
    }
}

// streaming: the participants (every 8th workgroup when one_xcd) read `bytes` in 16-byte lanes, UR loads in flight per lane
template <int UR>
__global__ __launch_bounds__(1024) void stream_kernel(const uint4* W, size_t n16, unsigned* sink, int one_xcd) {
    if (one_xcd && blockIdx.x % 8 != 0) return;
    const size_t me = one_xcd ? blockIdx.x / 8 : blockIdx.x, np = one_xcd ? gridDim.x / 8 : gridDim.x;
    unsigned acc = 0;
    const size_t stride = np * blockDim.x;
    for (size_t i = me * blockDim.x + threadIdx.x; i < n16; i += stride * UR) {
        uint4 w[UR];
#pragma unroll
        for (int u = 0; u < UR; ++u) w[u] = (i + u * stride < n16) ? W[i + u * stride] : uint4{0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < UR; ++u) acc += w[u].x ^ w[u].y ^ w[u].z ^ w[u].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
    CK(hipSetDevice(0));
    unsigned *cnt, *err, *slots, *sink, *where; uint4* W;
    const size_t wbytes = (size_t)1 << 30;
    CK(hipMalloc(&cnt, 4096)); CK(hipMalloc(&err, 16)); CK(hipMalloc(&slots, 2 * 4096 * 4)); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&where, 4096 * 4));
    CK(hipMalloc(&W, wbytes)); CK(hipMemset(W, 1, wbytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // (a)
    for (int nb : {256, 512, 2048}) {
        for (int th : {256, 1024}) {
            hipLaunchKernelGGL(where_kernel, dim3(nb), dim3(th), 0, 0, where);
            std::vector<unsigned> h(nb); CK(hipMemcpy(h.data(), where, nb * 4, hipMemcpyDeviceToHost));
            int bad = 0; for (int b = 0; b < nb; ++b) bad += h[b] != (unsigned)(b % 8);
            printf("placement: %4d workgroups of %4d threads: %d not on XCD b %% 8 (first ids %u %u %u %u %u %u %u %u %u)\n", nb, th, bad, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8]);
        }
    }
    // (b)
    for (int poll = 0; poll < 3; ++poll) for (int th : {256, 1024}) for (int npart : {32, 64}) for (int mode = 0; mode < 3; ++mode) {
        const int n = 2000; float best = 1e9f; unsigned herr[4] = {0, 0, 0, 0};
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipMemset(cnt, 0, 4096)); CK(hipMemset(err, 0, 16)); CK(hipMemset(slots, 0xff, 2 * 4096 * 4));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(one_xcd_kernel, dim3(npart * 8), dim3(th), 0, 0, cnt, err, slots, W, sink, n, mode, npart, poll);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            CK(hipMemcpy(herr, err, 16, hipMemcpyDeviceToHost)); if (herr[0]) break;
        }
        printf("one XCD, poll %d, %2d workgroups of %4d threads, mode %d: %6.3f us per phase   lost-arrival %u (counter %u) stale-reads %u off-XCD %u\n", poll, npart, th, mode, best * 1e3f / n, herr[0], herr[3], herr[1], herr[2]);
    }
    // (b2)
    unsigned* bigrows; CK(hipMalloc(&bigrows, (size_t)2000 * 32 * 32 * 4));
    for (int xmode = 0; xmode < 6; ++xmode) {
        const int n = 2000, npart = 32; float best = 1e9f; unsigned herr[4] = {0, 0, 0, 0};
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(cnt, 0, 4096)); CK(hipMemset(err, 0, 16)); CK(hipMemset(bigrows, 0xff, (size_t)2000 * 32 * 32 * 4));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(exchange_kernel, dim3(npart * 8), dim3(512), 0, 0, cnt, err, bigrows, sink, n, xmode, npart);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            CK(hipMemcpy(herr, err, 16, hipMemcpyDeviceToHost)); if (herr[0]) break;
        }
        printf("exchange on one XCD, read method %d (0 sc1 loads, 1 buffer_inv sc0 + plain, 2 buffer_inv sc1 + plain, 3 plain only, 4 fresh rows per phase + plain + L2-atomic polls, 5 = 4 with ONE barrier): %6.3f us per 2 barriers + exchange   lost-arrival %u stale-reads %u\n",
               xmode, best * 1e3f / n, herr[0], herr[1]);
    }
    // (c)
    for (int per : {8, 16, 32}) for (int mode = 0; mode < 2; ++mode) {
        const int n = 2000; float best = 1e9f; unsigned herr[4] = {0, 0, 0, 0};
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipMemset(cnt, 0, 4096)); CK(hipMemset(err, 0, 16)); CK(hipMemset(slots, 0xff, 2 * 4096 * 4));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(two_level_kernel, dim3(per * 8), dim3(256), 0, 0, cnt + 8, cnt, err, slots, n, per, mode);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            CK(hipMemcpy(herr, err, 16, hipMemcpyDeviceToHost)); if (herr[0]) break;
        }
        printf("two-level barrier, %2d workgroups per XCD (%3d in all), mode %d: %6.3f us per phase   lost-arrival %u stale-reads %u\n", per, per * 8, mode, best * 1e3f / n, herr[0], herr[1]);
    }
    // (d)
    for (int one : {1, 0}) for (int th : {256, 512, 1024}) for (int per : {32, 64}) {
        const size_t n16 = (one ? ((size_t)256 << 20) : wbytes) / 16;
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(stream_kernel<8>, dim3(per * 8), dim3(th), 0, 0, W, n16, sink, one);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("stream %s: %3d workgroups of %4d threads, 8 x 16 B in flight per lane: %7.1f GB/s\n", one ? "ONE XCD " : "all XCDs", one ? per : per * 8, th, n16 * 16 / (best * 1e6));
    }
    return 0;
}
