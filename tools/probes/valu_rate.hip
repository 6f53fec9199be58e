// VALU issue-rate probe (one wave per SIMD and two, as in the GEMM epilogue): v_fma_f32, v_pk_fma_f32, v_exp_f32, v_rcp_f32, 8 independent chains per wave.
// Prints cycles per wave-instruction.  hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(2))) float f32x2;
template <int OP>
__global__ void probe(float* out, int iters, unsigned long long* cyc) {
    float a[8]; f32x2 b[8];
    for (int i = 0; i < 8; ++i) { a[i] = 0.001f * (threadIdx.x + i); b[i] = f32x2{a[i], a[i] + 1.f}; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if constexpr (OP == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
            if constexpr (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(b[i]));
            if constexpr (OP == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            if constexpr (OP == 3) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if constexpr (OP == 4) asm volatile("v_cndmask_b32 %0, %0, %0, vcc" : "+v"(a[i]));
            if constexpr (OP == 5) asm volatile("v_bfi_b32 %0, %0, %0, %0" : "+v"(a[i]));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a[i] + b[i][0] + b[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
    float* out; unsigned long long* cyc; hipMalloc(&out, 256 * 512 * 4); hipMallocManaged(&cyc, 8);
    const int iters = 20000;
    const char* names[] = {"v_fma_f32", "v_pk_fma_f32", "v_exp_f32", "v_rcp_f32", "v_cndmask_b32", "v_bfi_b32"};
    for (int threads : {256, 512}) {
        for (int op = 0; op < 6; ++op) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                switch (op) {
                    case 0: probe<0><<<256, threads>>>(out, iters, cyc); break;
                    case 1: probe<1><<<256, threads>>>(out, iters, cyc); break;
                    case 2: probe<2><<<256, threads>>>(out, iters, cyc); break;
                    case 3: probe<3><<<256, threads>>>(out, iters, cyc); break;
                    case 4: probe<4><<<256, threads>>>(out, iters, cyc); break;
                    case 5: probe<5><<<256, threads>>>(out, iters, cyc); break;
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // s_memtime ticks at 100 MHz-derived constant rate on this part; report wall-derived ns per wave-instruction per SIMD instead
            const double per = ms * 1e6 / ((double)iters * 8 * (threads / 256));
            printf("%d waves/SIMD  %-14s %.2f ns per wave-instruction on a SIMD (%.1f cycles at 2.1 GHz)\n", threads / 256, names[op], per, per * 2.1);
        }
    }
    return 0;
}
