// Register budget of a 256 x 384 output tile ("full-row tile ... if the 192-accumulator-VGPR tile does not fit, show
// the code-object metadata"). The main loop of gemm3_kernel (pb_gemm2.hip) with 6 column tiles per wave instead of 4: 8 waves as
// 2 x 4, a wave owns 128 x 96 outputs = 8 x 6 accumulator tiles of 16 x 16 = 192 accumulator VGPRs; a phase multiplies one quadrant
// (4 row tiles x 3 column tiles) over K = 64, so 4 x 2 A fragments + 3 x 2 B fragments of 4 VGPRs = 56 more are live, 248 of the
// 256 a wave may hold at two waves per SIMD before the first address, loop counter or epilogue temporary.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -Rpass-analysis=kernel-resource-usage -c tools/tile384_regs.hip
// The summary of that command is committed as profiles/r03_tile384_resource_usage.txt.
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
constexpr int TNW = 6;

__device__ __forceinline__ bf16x8 frag(const char* lds, int row, int ks, int lane) {
    const int g = lane >> 4, r = row + (lane & 15);
    return *reinterpret_cast<const bf16x8*>(lds + r * 128 + (((ks * 4 + g) ^ ((r >> 1) & 7)) << 4));
}

__global__ __launch_bounds__(512) void tile384_kernel(const __bf16* __restrict__ A, const __bf16* __restrict__ B, __bf16* __restrict__ C, int K, int ldc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), wr = wave >> 2, wc = wave & 3;
    f32x4 acc[8][TNW];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < TNW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < K / 64; ++kt) {
        const char* sA = smem + (kt & 1) * 81920;                 // 2 slots x (A 256 rows + B 384 rows) x 128 B = 160 KiB: the whole LDS
        const char* sB = sA + 32768;
        // stand-in for the DMA of the next K-tile (one 16-byte piece per lane and operand keeps the loads in the loop)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A + (long)(kt + 1) * 64 + t * 8),
                                         (__attribute__((address_space(3))) void*)(smem + ((kt + 1) & 1) * 81920 + wave * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(B + (long)(kt + 1) * 64 + t * 8),
                                         (__attribute__((address_space(3))) void*)(smem + ((kt + 1) & 1) * 81920 + 32768 + wave * 1024), 16, 0, 0);
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int nh = 0; nh < 2; ++nh) {
                bf16x8 a[4][2], b[3][2];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) a[i][ks] = frag(sA, wr * 128 + mh * 64 + i * 16, ks, lane);
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) b[j][ks] = frag(sB, wc * 96 + nh * 48 + j * 16, ks, lane);
                __builtin_amdgcn_s_barrier();
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 3; ++j)
                            acc[mh * 4 + i][nh * 3 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j][ks], a[i][ks], acc[mh * 4 + i][nh * 3 + j], 0, 0, 0);
                __builtin_amdgcn_s_barrier();
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const int lr = lane & 15, lg = lane >> 4;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < TNW; ++j) {
            __bf16* c = C + (long)(blockIdx.x * 256 + wr * 128 + i * 16 + lr) * ldc + wc * 96 + j * 16 + lg * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) c[r] = (__bf16)acc[i][j][r];
        }
}
