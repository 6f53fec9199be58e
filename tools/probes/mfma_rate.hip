// MFMA issue-rate probe: 16x16x32 bf16 (8 bf16 per lane per operand) against the legacy 16x16x16 bf16 (4 per lane), 16 independent accumulators per
// wave, 4 waves per SIMD. Prints TFLOP/s of each on the whole chip.  hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
template <int K32>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a; s16x4 a4;
    for (int i = 0; i < 8; ++i) a[i] = (__bf16)(float)(threadIdx.x + i);
    for (int i = 0; i < 4; ++i) a4[i] = (short)(threadIdx.x + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if constexpr (K32) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, acc[i], 0, 0, 0);
            else acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, a4, acc[i], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 1024 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, grid = 1024;
    for (int v = 0; v < 2; ++v) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (v) probe<1><<<grid, 256>>>(out, iters); else probe<0><<<grid, 256>>>(out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = (double)grid * 4 * iters * 16 * (v ? 16384.0 : 8192.0);
            if (rep) printf("%s: %.3f ms  %.1f TFLOP/s\n", v ? "16x16x32 bf16" : "16x16x16 bf16 (1k)", ms, flops / ms / 1e9);
        }
    }
    return 0;
}
