// Developer probe (not part of the library): the fp32 atomic-add rate an attention backward with ONE pass over the (key block,
// query block) pairs would need for dQ. Pattern of that kernel: a workgroup owns (batch*head, key block), walks the query
// blocks and adds a 128 x 64 f32 tile per pair into dQ[batch*head][query block]; the workgroups of one (batch, head) sit on one XCD.
//   hipcc -O3 --offload-arch=gfx950 tools/atomic_rate_probe.hip -o /tmp/arp && /tmp/arp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// mode 0: global_atomic_add_f32 (no return). mode 1: plain load + add + store (wrong under sharing; bandwidth reference).
// mode 2: plain store only. mode 3: global_atomic_pk_add_bf16 on half the bytes.
template <int MODE>
__global__ __launch_bounds__(256) void add_tiles(float* dq, int nkb, int nqb, float v) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int kb = slot % nkb, bh = (slot / nkb) * 8 + xcd;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, g = lane >> 4;
    float* base = dq + (size_t)bh * nqb * 128 * 64;
    for (int qi = 0; qi < nqb; ++qi) {
        const int qb = (qi + kb) % nqb;                                   // the key blocks of a head start at different query blocks
        float* tile = base + (size_t)qb * 128 * 64;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float* p = tile + (size_t)(wave * 32 + rt * 16 + g * 4 + r) * 64 + dt * 16 + lr;
                    if (MODE == 0) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else if (MODE == 1) *p = *p + v;
                    else if (MODE == 2) *p = v;
                }
    }
}

int main() {
    CK(hipSetDevice(0));
    const int BH = 384, nqb = 8;
    float* dq; const size_t n = (size_t)BH * nqb * 128 * 64;
    CK(hipMalloc(&dq, n * 4)); CK(hipMemset(dq, 0, n * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int nkb : {8, 16}) {
        for (int mode = 0; mode < 3; ++mode) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipEventRecord(e0));
                const dim3 grid(BH * nkb), blk(256);
                if (mode == 0) hipLaunchKernelGGL(add_tiles<0>, grid, blk, 0, 0, dq, nkb, nqb, 1.0f);
                if (mode == 1) hipLaunchKernelGGL(add_tiles<1>, grid, blk, 0, 0, dq, nkb, nqb, 1.0f);
                if (mode == 2) hipLaunchKernelGGL(add_tiles<2>, grid, blk, 0, 0, dq, nkb, nqb, 1.0f);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const double bytes = (double)n * 4 * nkb;
            printf("key blocks %2d mode %d (%s): %8.1f us, %6.2f TB/s of f32 added (%.0f MB)\n", nkb, mode,
                   mode == 0 ? "atomic add f32" : mode == 1 ? "load+add+store" : "store", best * 1e3, bytes / best * 1e-9, bytes * 1e-6);
        }
    }
    float h[4]; CK(hipMemcpy(h, dq, 16, hipMemcpyDeviceToHost)); printf("check %g\n", h[0]);
    return 0;
}
