// What limits the store tail of a 256 x 256 bf16 tile? One workgroup of 8 waves per CU (as gemm3_kernel), each wave owns 128 rows x 64 columns and writes them
// with 16-byte stores in one of three lane layouts; tiles are walked like the persistent GEMM walks them (M fastest inside groups of 8). No arithmetic.
//   layout 0: the GEMM's own -- an instruction covers 16 rows x 64 contiguous bytes (4 lanes per row), two instructions complete a row's 128 bytes
//   layout 1: 8 rows x 128 contiguous bytes per instruction (8 lanes per row)
//   layout 2: 1 row x 1024 contiguous bytes per instruction (upper bound: not a tile layout the MFMA registers can feed)
//   layout 3: layout 1's bytes per instruction, but a row's eight 16-byte pieces on lanes lr + 16 lg and lr + 8 + 16 lg (what a DPP row_ror:8 exchange gives)
// and with 256, 64 or 32 workgroups (is it the chip or the CU?). Prints us per tile-store and the aggregate rate.
//   hipcc --offload-arch=gfx950 -O3 store_tile.hip -o store_tile && ./store_tile
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <int LAYOUT>
__global__ __launch_bounds__(512) void probe(unsigned short* C, long ldc, int tiles_m, int tiles_n, int per_wg) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wr = wave >> 2, wc = wave & 3;
    const u32x4 v = {(unsigned)t, (unsigned)blockIdx.x, 0x3f803f80u, 0x3f803f80u};
    for (int it = 0; it < per_wg; ++it) {
        const int tile = blockIdx.x + it * gridDim.x;
        if (tile >= tiles_m * tiles_n) break;
        const int tm = tile % tiles_m, tn = tile / tiles_m;
        unsigned short* base = C + ((long)tm * 256 + wr * 128) * ldc + tn * 256 + wc * 64;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                long off;
                if (LAYOUT == 0) off = (long)(i * 16 + (lane & 15)) * ldc + jp * 32 + (lane >> 4) * 8;
                else if (LAYOUT == 1) off = (long)((i * 2 + jp) * 8 + (lane >> 3)) * ldc + (lane & 7) * 8;
                else if (LAYOUT == 2) off = (long)(i * 2 + jp) * 8 * ldc + lane * 8;            // (8 of the wave's 128 rows: same bytes per instruction, one row each)
                else {       // layout 3: the bytes of layout 1 (8 rows x 128 B per instruction) but on the lanes a row_ror:8 exchange of layout 0 leaves them on
                    const int lr = lane & 15, lg = lane >> 4, cb = (lg & 1) ? 16 + (lg - 1) * 4 : lg * 4;
                    off = (long)(i * 16 + jp * 8 + (lr & 7)) * ldc + (lr >> 3) * 32 + cb;
                }
                *reinterpret_cast<u32x4*>(base + off) = v;
            }
    }
}
int main() {
    const int M = 26624, N = 3072;
    unsigned short* C; hipMalloc(&C, (size_t)M * N * 2 + (1 << 20));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int tiles_m = M / 256, tiles_n = N / 256;
    for (int grid : {256, 64, 32}) {
        for (int layout = 0; layout < 4; ++layout) {
            const int per_wg = 4;
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                if (layout == 0) probe<0><<<grid, 512>>>(C, N, tiles_m, tiles_n, per_wg);
                if (layout == 1) probe<1><<<grid, 512>>>(C, N, tiles_m, tiles_n, per_wg);
                if (layout == 2) probe<2><<<grid, 512>>>(C, N, tiles_m, tiles_n, per_wg);
                if (layout == 3) probe<3><<<grid, 512>>>(C, N, tiles_m, tiles_n, per_wg);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double bytes = (double)grid * per_wg * 256 * 256 * 2;
            printf("grid %3d layout %d: %7.1f us per launch of %d tiles per workgroup = %5.2f us per tile, %6.2f TB/s\n", grid, layout, best * 1e3, per_wg, best * 1e3 / per_wg, bytes / best / 1e9);
        }
    }
    return 0;
}
