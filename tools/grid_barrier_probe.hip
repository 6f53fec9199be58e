// Developer probe (not part of the library): what does a grid-wide barrier cost on MI355X, and what does one "phase" of a
// persistent decode kernel cost (barrier + a GEMV-sized weight read whose loads were issued BEFORE the barrier)?
//   hipcc -O3 --offload-arch=gfx950 tools/grid_barrier_probe.hip -o gpurun_out/grid_barrier_probe && gpurun_out/grid_barrier_probe
// Every spin is bounded: a lost arrival sets an error word and the kernel leaves instead of hanging the box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned* cnt, unsigned target, unsigned* err) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > (1u << 22)) { *err = 1; ok = false; break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    return ok;
}

// mode 0: barriers only. mode 1: each phase every workgroup publishes one word (agent-scope store) and checks the word its
// left neighbour (another XCD) published in the previous phase (agent-scope load): the exchange a decode phase needs.
// mode 2: mode 1 + a 16-byte-per-lane weight read per phase (W: nph slices of nwg*256*16 bytes), requested before the barrier.
__global__ __launch_bounds__(256) void phases_kernel(unsigned* cnt, unsigned* err, unsigned* slots, const uint4* W, unsigned* sink, int nph, int mode) {
    const unsigned nwg = gridDim.x, wg = blockIdx.x;
    unsigned acc = 0;
    for (int p = 0; p < nph; ++p) {
        uint4 w = {0, 0, 0, 0};
        if (mode == 2) w = W[((size_t)p * nwg + wg) * 256 + threadIdx.x];
        if (mode >= 1 && threadIdx.x == 0)
            __hip_atomic_store(slots + (p & 1) * nwg + wg, (unsigned)(p * 7919u + wg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (mode >= 1) __builtin_amdgcn_s_waitcnt(0);                    // the store has left the CU before the arrival is counted
        if (!grid_barrier(cnt, (unsigned)(p + 1) * nwg, err)) return;
        if (mode >= 1 && threadIdx.x == 0) {
            const unsigned nb = (wg + 1) % nwg;
            const unsigned got = __hip_atomic_load(slots + (p & 1) * nwg + nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (got != (unsigned)(p * 7919u + nb)) atomicAdd(err + 1, 1u);
        }
        acc += w.x ^ w.y ^ w.z ^ w.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
    int dev = 0; CK(hipSetDevice(dev));
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, dev));
    printf("device %s, %d CUs\n", pr.name, pr.multiProcessorCount);
    const int nph = 2000;
    unsigned *cnt, *err, *slots, *sink; uint4* W;
    const size_t wbytes = (size_t)nph * 512 * 256 * 16;                  // up to 512 workgroups: 2 MB per phase, 4 GB total? no: cap below
    const int nph_w = 200;                                               // mode 2 walks 200 distinct slices (419 MB at 512 wgs), then wraps
    CK(hipMalloc(&cnt, 4)); CK(hipMalloc(&err, 8)); CK(hipMalloc(&slots, 2 * 1024 * 4)); CK(hipMalloc(&sink, 4));
    CK(hipMalloc(&W, (size_t)nph_w * 512 * 256 * 16)); CK(hipMemset(W, 1, (size_t)nph_w * 512 * 256 * 16));
    (void)wbytes;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int nwg : {256, 512}) {
        for (int mode = 0; mode < 3; ++mode) {
            const int n = mode == 2 ? nph_w : nph;
            float best = 1e9f; unsigned herr[2] = {0, 0};
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipMemset(cnt, 0, 4)); CK(hipMemset(err, 0, 8)); CK(hipMemset(slots, 0xff, 2 * 1024 * 4));
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(phases_kernel, dim3(nwg), dim3(256), 0, 0, cnt, err, slots, W, sink, n, mode);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
                CK(hipMemcpy(herr, err, 8, hipMemcpyDeviceToHost));
                if (herr[0]) break;
            }
            printf("wgs %3d mode %d: %7.3f us per phase (%d phases)  lost-arrival %u  stale-reads %u\n", nwg, mode, best * 1e3f / n, n, herr[0], herr[1]);
        }
    }
    // reference point: the same number of EMPTY kernel launches back to back, and through a graph
    {
        hipStream_t st; CK(hipStreamCreate(&st));
        const int n = 1000;
        CK(hipMemset(cnt, 0, 4)); CK(hipMemset(err, 0, 8));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < n; ++i) hipLaunchKernelGGL(phases_kernel, dim3(256), dim3(256), 0, st, cnt, err, slots, W, sink, 0, 0);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("empty launches, stream: %7.3f us each\n", ms * 1e3f / n);
        }
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
        for (int i = 0; i < 74; ++i) hipLaunchKernelGGL(phases_kernel, dim3(256), dim3(256), 0, st, cnt, err, slots, W, sink, 0, 0);
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, st));
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("empty launches, graph of 74: %7.3f us each\n", ms * 1e3f / (20 * 74));
        }
    }
    return 0;
}
