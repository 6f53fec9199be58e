// VALU issue-rate probe for gfx950: cycles per wave-instruction for a few ops (8 independent chains per lane, 1 wave per SIMD and 2 waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 4096
template <int OP>
__global__ void k(float* out, float seed) {
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = seed + i * 0.001f + threadIdx.x * 1e-6f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = f2{a[2 * i], a[2 * i + 1]};
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) a[i] = __builtin_amdgcn_exp2f(a[i]);
            else if (OP == 1) a[i] = __builtin_fmaf(a[i], 0.999f, 0.001f);
            else if (OP == 2) a[i] = __builtin_amdgcn_rcpf(a[i]);
            else if (OP == 3) a[i] = __builtin_amdgcn_ldexpf(a[i], 1 - (it & 2));
            else if (OP == 4) a[i] = __builtin_floorf(a[i]) + 0.5f;          // floor + add
            else if (OP == 6) a[i] = __builtin_amdgcn_fractf(a[i]) ;
        }
        if (OP == 5) {
#pragma unroll
            for (int i = 0; i < 4; ++i) p[i] = __builtin_elementwise_fma(p[i], f2{0.999f, 0.999f}, f2{0.001f, 0.001f});
#pragma unroll
            for (int i = 0; i < 4; ++i) p[i] = __builtin_elementwise_fma(p[i], f2{0.998f, 0.998f}, f2{0.002f, 0.002f});
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += p[i][0] + p[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char* name, int threads, float* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<256 * 8, threads>>>(d, 0.5f); hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<256 * 8, threads>>>(d, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // blocks: 2048 over 256 CUs = 8 blocks per CU, sequential-ish; waves per block = threads/64 spread over 4 SIMDs
    const double waves_per_simd_total = 8.0 * (threads / 64) / 4.0;     // wave-executions per SIMD over the launch
    const double insts = (double)ITERS * 8 * waves_per_simd_total;
    printf("%-28s threads/block %4d: %.3f ms -> %.2f cycles per wave-instruction (at 2.4 GHz)\n", name, threads, ms, ms * 1e-3 * 2.4e9 / insts);
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 1024 * sizeof(float));
    for (int threads : {256, 512, 1024}) {
        run<0>("v_exp_f32", threads, d);
        run<1>("v_fma_f32", threads, d);
        run<2>("v_rcp_f32", threads, d);
        run<3>("v_ldexp_f32", threads, d);
        run<4>("v_floor_f32 + v_add_f32 (2)", threads, d);
        run<6>("v_fract_f32", threads, d);
        run<5>("v_pk_fma_f32 (8 per iter)", threads, d);
    }
    return 0;
}
