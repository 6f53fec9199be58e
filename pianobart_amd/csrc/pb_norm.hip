// Row kernels of the PianoBART step (HBM-bound, one wave64 per token row, rows kept in registers):
//   * Octuple gather-sum + position + LayerNorm (+dropout)            [K1/K2]  fwd / bwd
//   * y = LayerNorm(res + dropout(a))                                 [K5/K6]  fwd / bwd
//   * column sums for bias gradients
// A lane owns columns {4*(lane + 64*it) .. +3}; NIT = ceil(d/256) is a template parameter so the
// row lives in registers. Parameter-gradient reductions (dgamma, dbeta, dbias) are accumulated in
// registers over a grid-stride loop of rows, reduced across the 4 waves through LDS, written as
// per-block partials and summed by a second tiny kernel in a fixed order: deterministic, no atomics (the one exception is the
// exact-f32 embedding backward, which scatter-adds into the projected table with f32 atomics).
#include "pb_common.h"
#include "pb_api_internal.h"
#include <algorithm>
#include <cstring>
#include <vector>

namespace {

constexpr int LN_THREADS = 512, LN_WAVES = 8, LN_MAX_BLOCKS = 512;

__device__ __forceinline__ DropCfg make_drop(uint64_t seed, uint32_t site, float p) {
    DropCfg d;
    d.seed_lo = (uint32_t)seed; d.seed_hi = (uint32_t)(seed >> 32); d.site = site;
    d.thresh = p > 0.f ? (uint32_t)fminf(p * 4294967296.0f, 4294967295.0f) : 0u;
    d.scale = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    return d;
}

// ------------------------------------------------------------------ LayerNorm core (registers)
template <int NIT>
__device__ __forceinline__ void ln_stats(const f32x4 (&z)[NIT], int lane, int d4, int d, float eps, float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it)
        if (lane + 64 * it < d4) s += z[it][0] + z[it][1] + z[it][2] + z[it][3];
    mean = wave_sum(s) / d;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it)
        if (lane + 64 * it < d4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float c = z[it][j] - mean; q += c * c; }
        }
    rstd = rsqrtf(wave_sum(q) / d + eps);
}

// ------------------------------------------------------------------ y = LN(res + drop(a))
template <typename T, int NIT>
__global__ __launch_bounds__(LN_THREADS) void add_ln_fwd_kernel(const T* __restrict__ res, const T* __restrict__ a,
        const float* __restrict__ w, const float* __restrict__ b, T* __restrict__ y, float* __restrict__ mean_o,
        float* __restrict__ rstd_o, int rows, int d, float eps, uint64_t seed, uint32_t site, float p, const int* __restrict__ row_ids) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, d4 = d >> 2;
    const DropCfg dc = make_drop(seed, site, p);
    for (long row = (long)blockIdx.x * LN_WAVES + wave; row < rows; row += (long)gridDim.x * LN_WAVES) {
        const long rid = row_ids ? row_ids[row] : row;              // packed rows draw the dropout bits of their row in the padded batch
        f32x4 z[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c4 = lane + 64 * it;
            if (c4 < d4) {
                const f32x4 r = load4(res + row * d + 4 * c4);
                const f32x4 x = load4(a + row * d + 4 * c4);
                const f32x4 m = drop_mask4(dc, (uint32_t)(rid * d4 + c4));
                z[it] = r + x * m;
            }
        }
        float mean, rstd;
        ln_stats<NIT>(z, lane, d4, d, eps, mean, rstd);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c4 = lane + 64 * it;
            if (c4 < d4) {
                const f32x4 g = load4(w + 4 * c4), be = load4(b + 4 * c4);
                store4(y + row * d + 4 * c4, (z[it] - mean) * rstd * g + be);
            }
        }
        if (lane == 0) { mean_o[row] = mean; rstd_o[row] = rstd; }
    }
}

// Cross-wave reduction of NACC per-lane accumulators and partial write: partials[blk][k][d].
template <int NIT, int NACC>
__device__ __forceinline__ void write_partials(f32x4 (&acc)[NACC][NIT], float* __restrict__ partials, int d, float* lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, d4 = d >> 2;
#pragma unroll
    for (int k = 0; k < NACC; ++k) {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c4 = lane + 64 * it;
            if (c4 < d4) *reinterpret_cast<f32x4*>(lds + (size_t)wave * d + 4 * c4) = acc[k][it];
        }
        __syncthreads();
        for (int c = threadIdx.x; c < d; c += LN_THREADS) {
            float s = 0.f;
#pragma unroll
            for (int wv = 0; wv < LN_WAVES; ++wv) s += lds[(size_t)wv * d + c];
            partials[((size_t)blockIdx.x * NACC + k) * d + c] = s;
        }
    }
}

// Register budget by row length: up to d = 768 (NIT <= 3) 128 VGPRs = 16 waves per CU (two 8-wave workgroups, tuned for cfg 2); a d = 1024 /
// 2048 row (NIT 4 / 8: cfg 5, the reference's CLI default) keeps 4 / 8 float4 of xhat, g, mask and three accumulators per lane and
// gets the 256-register budget of one workgroup per CU instead of spilling to scratch.
template <typename T, typename TR, int NIT>
__global__ __launch_bounds__(LN_THREADS) __attribute__((amdgpu_waves_per_eu(NIT <= 3 ? 4 : 2, NIT <= 3 ? 4 : 2))) void add_ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ res,
        const T* __restrict__ a, const float* __restrict__ w, const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
        TR* __restrict__ dres, T* __restrict__ da, float* __restrict__ partials, int rows, int d, int accum_dres,
        uint64_t seed, uint32_t site, float p, const int* __restrict__ row_ids) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* lds = reinterpret_cast<float*>(smem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, d4 = d >> 2;
    const DropCfg dc = make_drop(seed, site, p);
    f32x4 acc[3][NIT];   // 0: dgamma, 1: dbeta, 2: dbias_a (column sum of da)
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int it = 0; it < NIT; ++it) acc[k][it] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (long row = (long)blockIdx.x * LN_WAVES + wave; row < rows; row += (long)gridDim.x * LN_WAVES) {
        const float mean = mean_i[row], rstd = rstd_i[row];
        const long rid = row_ids ? row_ids[row] : row;
        f32x4 xh[NIT], g[NIT], msk[NIT];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c4 = lane + 64 * it;
            if (c4 < d4) {
                const f32x4 r = load4(res + row * d + 4 * c4);
                const f32x4 x = load4(a + row * d + 4 * c4);
                msk[it] = drop_mask4(dc, (uint32_t)(rid * d4 + c4));
                xh[it] = (r + x * msk[it] - mean) * rstd;
                const f32x4 dyv = load4(dy + row * d + 4 * c4);
                g[it] = dyv * load4(w + 4 * c4);
                acc[0][it] += dyv * xh[it];
                acc[1][it] += dyv;
#pragma unroll
                for (int j = 0; j < 4; ++j) { s1 += g[it][j]; s2 += g[it][j] * xh[it][j]; }
            }
        }
        s1 = wave_sum(s1) / d; s2 = wave_sum(s2) / d;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c4 = lane + 64 * it;
            if (c4 < d4) {
                const f32x4 dz = (g[it] - s1 - xh[it] * s2) * rstd;
                const f32x4 dav = dz * msk[it];
                acc[2][it] += dav;
                f32x4 o = dz;
                if (accum_dres) o += load4(dres + row * d + 4 * c4);
                store4(dres + row * d + 4 * c4, o);
                if (da) store4(da + row * d + 4 * c4, dav);
            }
        }
    }
    write_partials<NIT, 3>(acc, partials, d, lds);
}

// out_k[c] += sum_blk partials[blk][k][c]   for k < nacc (NULL outputs skipped).
// block = 8 column quads (32 columns) x 128 row groups, grid (ceil(d/32), nacc): with <= 512 partial rows a thread has at most four
// 16-byte loads, all in flight at once (the kernel is pure latency), and every output element is summed by ONE workgroup in a fixed
// order (strided chain, then a fixed LDS tree) -- no atomics, so parameter gradients are bit-reproducible run to run.
constexpr int FIN_GROUPS = 128;
template <int V> struct FinVec;
template <> struct FinVec<4> { using T = f32x4; static __device__ T ld(const float* p) { return load4(p); } static __device__ void st(float* p, T v) { store4(p, v); } };
template <> struct FinVec<1> { using T = float; static __device__ T ld(const float* p) { return *p; } static __device__ void st(float* p, T v) { *p = v; } };

template <int V>   // V = 4: 16-byte accesses (d % 4 == 0, 16-byte aligned pointers); V = 1: any d / alignment
__device__ __forceinline__ void finalize_block(const float* __restrict__ partials, int nblk, int nacc, int d, float* o, int k, int bx) {
    using F = FinVec<V>;
    using T = typename F::T;
    __shared__ T red[FIN_GROUPS][8];
    const int cq = threadIdx.x & 7, rg = threadIdx.x >> 3;
    const int c = V * (bx * 8 + cq);
    const bool live = c < d;
    T s = T(0.f);
    if (live) {
        const float* src = partials + (size_t)k * d + c;
        const size_t st = (size_t)nacc * d;
        int b = rg;
        for (; b + 3 * FIN_GROUPS < nblk; b += 4 * FIN_GROUPS) {
            const T v0 = F::ld(src + (size_t)b * st), v1 = F::ld(src + (size_t)(b + FIN_GROUPS) * st);
            const T v2 = F::ld(src + (size_t)(b + 2 * FIN_GROUPS) * st), v3 = F::ld(src + (size_t)(b + 3 * FIN_GROUPS) * st);
            s += (v0 + v1) + (v2 + v3);
        }
        for (; b < nblk; b += FIN_GROUPS) s += F::ld(src + (size_t)b * st);
    }
    red[rg][cq] = s;
    __syncthreads();
    if (rg < 32) red[rg][cq] = (red[rg][cq] + red[rg + 32][cq]) + (red[rg + 64][cq] + red[rg + 96][cq]);
    __syncthreads();
    if (rg < 8) red[rg][cq] = (red[rg][cq] + red[rg + 8][cq]) + (red[rg + 16][cq] + red[rg + 24][cq]);
    __syncthreads();
    if (rg == 0 && live) {
        const T t = ((red[0][cq] + red[1][cq]) + (red[2][cq] + red[3][cq])) + ((red[4][cq] + red[5][cq]) + (red[6][cq] + red[7][cq]));
        F::st(o + c, F::ld(o + c) + t);
    }
}

template <int V>
__global__ __launch_bounds__(8 * FIN_GROUPS) void finalize_partials_kernel(const float* __restrict__ partials, int nblk, int nacc, int d,
                                                                           float* o0, float* o1, float* o2, float* o3) {
    const int k = blockIdx.y;
    float* o = k == 0 ? o0 : k == 1 ? o1 : k == 2 ? o2 : o3;
    if (!o) return;
    finalize_block<V>(partials, nblk, nacc, d, o, k, blockIdx.x);
}

// Deferred form: one launch sums MANY partial sets (blockIdx.z walks a descriptor table). The ~160 bias / LayerNorm-parameter
// reductions of one backward pass are 6 us of pure launch + latency each when issued one by one behind their producers; collected
// in a table they are a single ~0.1 ms launch at the end of backward (pb_defer_begin / pb_defer_flush).
struct FinDesc { const float* partials; float* o[3]; int nblk, nacc, d, pad; };
__global__ __launch_bounds__(8 * FIN_GROUPS) void finalize_batch_kernel(const FinDesc* __restrict__ table) {
    const FinDesc D = table[blockIdx.z];
    const int k = blockIdx.y;
    if (k >= D.nacc || (int)blockIdx.x * 32 >= D.d) return;
    float* o = k == 0 ? D.o[0] : k == 1 ? D.o[1] : D.o[2];
    if (!o) return;
    finalize_block<4>(D.partials, D.nblk, D.nacc, D.d, o, k, blockIdx.x);
}

// ------------------------------------------------------------------ Octuple embed + pos + LN
struct SegOff { int off[8]; };

template <int NIT>
__device__ __forceinline__ void embed_row(f32x4 (&z)[NIT], const int16_t* __restrict__ ids16, const float* __restrict__ P,
                                          const SegOff& so, const float* __restrict__ lin_bias, const float* __restrict__ pos,
                                          long row, int S, int d, int lane, const int* __restrict__ row_ids) {
    const int d4 = d >> 2;
    // one 16-byte load of the 8 int16 ids of this token, broadcast over the wave
    const uint4 raw = *reinterpret_cast<const uint4*>(ids16 + row * 8);
    int id[8];
    id[0] = (int)(raw.x & 0xffff); id[1] = (int)(raw.x >> 16); id[2] = (int)(raw.y & 0xffff); id[3] = (int)(raw.y >> 16);
    id[4] = (int)(raw.z & 0xffff); id[5] = (int)(raw.z >> 16); id[6] = (int)(raw.w & 0xffff); id[7] = (int)(raw.w >> 16);
    const int s = (int)((row_ids ? (long)row_ids[row] : row) % S);      // packed rows carry their row number in the padded batch
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c4 = lane + 64 * it;
        if (c4 < d4) {
            f32x4 v = load4(lin_bias + 4 * c4) + load4(pos + (size_t)(s + 2) * d + 4 * c4);
#pragma unroll
            for (int i = 0; i < 8; ++i) v += load4(P + (size_t)(so.off[i] + id[i]) * d + 4 * c4);
            z[it] = v;
        }
    }
}

template <typename T, int NIT>
__global__ __launch_bounds__(LN_THREADS) void embed_ln_fwd_kernel(const int16_t* __restrict__ ids16, const float* __restrict__ P,
        const SegOff so, const float* __restrict__ lin_bias, const float* __restrict__ pos, const float* __restrict__ w,
        const float* __restrict__ b, T* __restrict__ y, float* __restrict__ mean_o, float* __restrict__ rstd_o,
        int rows, int S, int d, float eps, uint64_t seed, uint32_t site, float p, const int* __restrict__ row_ids) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, d4 = d >> 2;
    const DropCfg dc = make_drop(seed, site, p);
    for (long row = (long)blockIdx.x * LN_WAVES + wave; row < rows; row += (long)gridDim.x * LN_WAVES) {
        f32x4 z[NIT];
        embed_row<NIT>(z, ids16, P, so, lin_bias, pos, row, S, d, lane, row_ids);
        float mean, rstd;
        ln_stats<NIT>(z, lane, d4, d, eps, mean, rstd);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c4 = lane + 64 * it;
            if (c4 < d4) {
                const f32x4 g = load4(w + 4 * c4), be = load4(b + 4 * c4);
                const f32x4 m = drop_mask4(dc, (uint32_t)((row_ids ? (long)row_ids[row] : row) * d4 + c4));     // dropout AFTER the LN here
                store4(y + row * d + 4 * c4, ((z[it] - mean) * rstd * g + be) * m);
            }
        }
        if (lane == 0) { mean_o[row] = mean; rstd_o[row] = rstd; }
    }
}

// dz_out == NULL: scatter-add dz into dP / dpos with f32 atomics (exact-f32 path).
// dz_out != NULL: write dz (T,d) instead; dP and dpos are then produced without atomics by a one-hot MFMA GEMM
// (dP = Onehot^T dz) and pb_batch_sum (bf16 throughput path).
template <typename T, int NIT>
__global__ __launch_bounds__(LN_THREADS) void embed_ln_bwd_kernel(const T* __restrict__ dy, const int16_t* __restrict__ ids16,
        const float* __restrict__ P, const SegOff so, const float* __restrict__ lin_bias, const float* __restrict__ pos,
        const float* __restrict__ w, const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
        float* __restrict__ dP, float* __restrict__ dpos, float* __restrict__ partials, T* __restrict__ dz_out,
        int rows, int S, int d, uint64_t seed, uint32_t site, float p, const int* __restrict__ row_ids) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* lds = reinterpret_cast<float*>(smem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, d4 = d >> 2;
    const DropCfg dc = make_drop(seed, site, p);
    f32x4 acc[3][NIT];   // 0: dgamma, 1: dbeta, 2: dbias (column sum of dz)
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int it = 0; it < NIT; ++it) acc[k][it] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (long row = (long)blockIdx.x * LN_WAVES + wave; row < rows; row += (long)gridDim.x * LN_WAVES) {
        const float mean = mean_i[row], rstd = rstd_i[row];
        f32x4 z[NIT], g[NIT];
        embed_row<NIT>(z, ids16, P, so, lin_bias, pos, row, S, d, lane, row_ids);
        float s1 = 0.f, s2 = 0.f, nz = 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c4 = lane + 64 * it;
            if (c4 < d4) {
                z[it] = (z[it] - mean) * rstd;                                   // xhat
                const f32x4 m = drop_mask4(dc, (uint32_t)((row_ids ? (long)row_ids[row] : row) * d4 + c4));
                const f32x4 dyv = load4(dy + row * d + 4 * c4) * m;
                g[it] = dyv * load4(w + 4 * c4);
                acc[0][it] += dyv * z[it];
                acc[1][it] += dyv;
#pragma unroll
                for (int j = 0; j < 4; ++j) { s1 += g[it][j]; s2 += g[it][j] * z[it][j]; nz += fabsf(dyv[j]); }
            }
        }
        s1 = wave_sum(s1) / d; s2 = wave_sum(s2) / d; nz = wave_sum(nz);
        if (dz_out) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int c4 = lane + 64 * it;
                if (c4 < d4) {
                    const f32x4 dz = (g[it] - s1 - z[it] * s2) * rstd;
                    acc[2][it] += dz;
                    store4(dz_out + row * d + 4 * c4, dz);
                }
            }
            continue;
        }
        if (nz == 0.f) continue;        // rows that received no gradient (PAD tail) add exact zeros: skip the atomics
        const uint4 raw = *reinterpret_cast<const uint4*>(ids16 + row * 8);
        int id[8];
        id[0] = (int)(raw.x & 0xffff); id[1] = (int)(raw.x >> 16); id[2] = (int)(raw.y & 0xffff); id[3] = (int)(raw.y >> 16);
        id[4] = (int)(raw.z & 0xffff); id[5] = (int)(raw.z >> 16); id[6] = (int)(raw.w & 0xffff); id[7] = (int)(raw.w >> 16);
        const int s = (int)((row_ids ? (long)row_ids[row] : row) % S);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c4 = lane + 64 * it;
            if (c4 < d4) {
                const f32x4 dz = (g[it] - s1 - z[it] * s2) * rstd;
                acc[2][it] += dz;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    atomicAdd(dpos + (size_t)(s + 2) * d + 4 * c4 + j, dz[j]);
#pragma unroll
                    for (int i = 0; i < 8; ++i) atomicAdd(dP + (size_t)(so.off[i] + id[i]) * d + 4 * c4 + j, dz[j]);
                }
            }
        }
    }
    write_partials<NIT, 3>(acc, partials, d, lds);
}

// ------------------------------------------------------------------ column sums (bias grads)
// grid (ceil(N/256), nrb): a lane owns 4 adjacent columns (8/16-byte loads), the 4 waves of a block take
// interleaved rows of the block's row range, 4 rows in flight per lane; waves are combined through LDS.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ dy, long ld, float* __restrict__ partials, int rows, int N, int rows_per_blk) {
    __shared__ float red[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = (blockIdx.x * 64 + lane) * 4;
    const int r0 = blockIdx.y * rows_per_blk, r1 = min(rows, r0 + rows_per_blk);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (c < N) {
        int r = r0 + wave;
        for (; r + 12 < r1; r += 16) {
            const f32x4 a = load4(dy + (long)r * ld + c), b = load4(dy + (long)(r + 4) * ld + c);
            const f32x4 e = load4(dy + (long)(r + 8) * ld + c), f = load4(dy + (long)(r + 12) * ld + c);
            s += (a + b) + (e + f);
        }
        for (; r < r1; r += 4) s += load4(dy + (long)r * ld + c);
    }
    *reinterpret_cast<f32x4*>(&red[wave][lane * 4]) = s;
    __syncthreads();
    const int cc = blockIdx.x * 256 + threadIdx.x;
    if (cc < N) partials[(size_t)blockIdx.y * N + cc] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

int ln_grid(int rows) { return max(1, min((rows + LN_WAVES - 1) / LN_WAVES, LN_MAX_BLOCKS)); }        // kernels that write per-block partials
int ln_grid_fwd(int rows) { return max(1, min((rows + LN_WAVES - 1) / LN_WAVES, 2048)); }

}  // namespace

#define PB_LN_DISPATCH(NITV, ...)                        \
    switch (NITV) {                                      \
        case 1: { constexpr int NIT = 1; __VA_ARGS__; } break; \
        case 2: { constexpr int NIT = 2; __VA_ARGS__; } break; \
        case 3: { constexpr int NIT = 3; __VA_ARGS__; } break; \
        case 4: { constexpr int NIT = 4; __VA_ARGS__; } break; \
        default: { constexpr int NIT = 8; __VA_ARGS__; } break; \
    }

static int check_ln_dims(const char* who, int T, int d) {
    PB_REQUIRE(T >= 0 && d > 0 && d % 4 == 0 && d <= 2048, "%s: d=%d must be a multiple of 4 and <= 2048", who, d);
    return 0;
}
static int nit_for(int d) { const int n = (d / 4 + 63) / 64; return n <= 4 ? n : 8; }

extern "C" int64_t pb_ln_partials_floats(int32_t d) { return (int64_t)LN_MAX_BLOCKS * 3 * d; }

static int add_ln_fwd_impl(const void* res, const void* a, const float* ln_w, const float* ln_b, void* y, float* mean,
                           float* rstd, int32_t T, int32_t d, int32_t dtype, float eps, uint64_t seed, uint32_t site,
                           float p_drop, void* stream_, const int32_t* row_ids) {
    hipStream_t stream = (hipStream_t)stream_;
    if (check_ln_dims("pb_add_ln_fwd", T, d)) return -2;
    if (T == 0) return 0;
    const int grid = ln_grid_fwd(T);
    PB_LN_DISPATCH(nit_for(d),
        if (dtype == PB_BF16)
            hipLaunchKernelGGL((add_ln_fwd_kernel<bf16_t, NIT>), dim3(grid), dim3(LN_THREADS), 0, stream, (const bf16_t*)res,
                               (const bf16_t*)a, ln_w, ln_b, (bf16_t*)y, mean, rstd, T, d, eps, seed, site, p_drop, row_ids);
        else
            hipLaunchKernelGGL((add_ln_fwd_kernel<float, NIT>), dim3(grid), dim3(LN_THREADS), 0, stream, (const float*)res,
                               (const float*)a, ln_w, ln_b, (float*)y, mean, rstd, T, d, eps, seed, site, p_drop, row_ids));
    PB_LAUNCH_CHECK();
    return 0;
}
extern "C" int pb_add_ln_fwd(const void* res, const void* a, const float* ln_w, const float* ln_b, void* y, float* mean,
                             float* rstd, int32_t T, int32_t d, int32_t dtype, float eps, uint64_t seed, uint32_t site,
                             float p_drop, void* stream_) {
    return add_ln_fwd_impl(res, a, ln_w, ln_b, y, mean, rstd, T, d, dtype, eps, seed, site, p_drop, stream_, nullptr);
}
extern "C" int pb_add_ln_fwd_packed(const void* res, const void* a, const float* ln_w, const float* ln_b, void* y, float* mean,
                                    float* rstd, const int32_t* row_ids, int32_t T, int32_t d, int32_t dtype, float eps, uint64_t seed,
                                    uint32_t site, float p_drop, void* stream_) {
    PB_REQUIRE(row_ids != nullptr, "pb_add_ln_fwd_packed: row_ids is required");
    return add_ln_fwd_impl(res, a, ln_w, ln_b, y, mean, rstd, T, d, dtype, eps, seed, site, p_drop, stream_, row_ids);
}

// ---- deferred reductions (see finalize_batch_kernel). The open window (arena cursor, descriptor list) is per host thread: a backward
// pass is issued by one thread. The device table itself belongs to the caller and may be shared between threads (autograd's device
// thread and the main thread use the same workspace) or be re-allocated at the same address, so nothing is remembered about its
// contents: every flush uploads its descriptors (~10 KB) with a stream-ordered copy from a small ring of pinned staging slots.
struct DeferState {
    bool active = false;
    float* arena = nullptr; size_t cap = 0, used = 0;       // device floats handed out as partial-sum storage
    FinDesc* table = nullptr; int table_cap = 0, n = 0, maxd = 0;
    std::vector<FinDesc> host;                             // descriptors of this pass
    static constexpr int NSLOT = 4;
    FinDesc* pin[NSLOT] = {}; size_t pin_cap[NSLOT] = {}; hipEvent_t ev[NSLOT] = {}; int next = 0;
};
static thread_local DeferState g_defer;

float* pb_defer_alloc(size_t nfloats) {
    DeferState& S = g_defer;
    nfloats = (nfloats + 3) & ~(size_t)3;
    if (!S.active || S.n >= S.table_cap || S.used + nfloats > S.cap) return nullptr;
    float* p = S.arena + S.used;
    S.used += nfloats;
    return p;
}

static bool defer_push(const float* partials, int nblk, int nacc, int d, float* o0, float* o1, float* o2, float* o3) {
    DeferState& S = g_defer;
    if (!S.active || partials < S.arena || partials >= S.arena + S.cap || S.n >= S.table_cap || o3 || nacc > 3) return false;
    const uintptr_t al = (uintptr_t)partials | (uintptr_t)o0 | (uintptr_t)o1 | (uintptr_t)o2;
    if (d % 4 || al % 16) return false;
    FinDesc D = {partials, {o0, o1, o2}, nblk, nacc, d, 0};
    S.host.push_back(D);
    S.n++;
    S.maxd = std::max(S.maxd, d);
    return true;
}

extern "C" int pb_defer_begin(float* arena, int64_t arena_floats, void* table, int32_t table_entries) {
    DeferState& S = g_defer;
    PB_REQUIRE(arena && table && arena_floats > 0 && table_entries > 0 && (uintptr_t)arena % 16 == 0, "pb_defer_begin: bad arena / table");
    S.active = true; S.arena = arena; S.cap = (size_t)arena_floats; S.used = 0;
    S.table = (FinDesc*)table; S.table_cap = table_entries; S.n = 0; S.maxd = 0;
    S.host.clear();
    return 0;
}

extern "C" int32_t pb_defer_desc_bytes(void) { return (int32_t)sizeof(FinDesc); }

extern "C" int pb_defer_flush(void* stream_) {
    DeferState& S = g_defer;
    if (!S.active) return 0;
    S.active = false;
    if (S.n == 0) return 0;
    hipStream_t stream = (hipStream_t)stream_;
    // stage the descriptors in a pinned slot (waiting, once in NSLOT flushes at most, for the copy that last used it) and copy
    // them to the caller's table in stream order: no host synchronisation, nothing cached across calls
    const int slot = S.next;
    S.next = (S.next + 1) % DeferState::NSLOT;
    const size_t bytes = S.host.size() * sizeof(FinDesc);
    if (S.ev[slot]) PB_CHECK_HIP(hipEventSynchronize(S.ev[slot]));
    else PB_CHECK_HIP(hipEventCreateWithFlags(&S.ev[slot], hipEventDisableTiming));
    if (S.pin_cap[slot] < bytes) {
        if (S.pin[slot]) PB_CHECK_HIP(hipHostFree(S.pin[slot]));
        S.pin[slot] = nullptr; S.pin_cap[slot] = 0;
        PB_CHECK_HIP(hipHostMalloc((void**)&S.pin[slot], bytes, hipHostMallocDefault));
        S.pin_cap[slot] = bytes;
    }
    memcpy(S.pin[slot], S.host.data(), bytes);
    PB_CHECK_HIP(hipMemcpyAsync(S.table, S.pin[slot], bytes, hipMemcpyHostToDevice, stream));
    PB_CHECK_HIP(hipEventRecord(S.ev[slot], stream));
    hipLaunchKernelGGL(finalize_batch_kernel, dim3((S.maxd + 31) / 32, 3, S.n), dim3(8 * FIN_GROUPS), 0, stream, S.table);
    PB_LAUNCH_CHECK();
    return 0;
}

static int launch_finalize(const float* partials, int nblk, int nacc, int d, float* o0, float* o1, float* o2, float* o3,
                           hipStream_t stream) {
    if (defer_push(partials, nblk, nacc, d, o0, o1, o2, o3)) return 0;
    const uintptr_t al = (uintptr_t)partials | (uintptr_t)o0 | (uintptr_t)o1 | (uintptr_t)o2 | (uintptr_t)o3;
    if (d % 4 == 0 && al % 16 == 0)
        hipLaunchKernelGGL(finalize_partials_kernel<4>, dim3((d + 31) / 32, nacc), dim3(8 * FIN_GROUPS), 0, stream, partials, nblk, nacc, d, o0, o1, o2, o3);
    else
        hipLaunchKernelGGL(finalize_partials_kernel<1>, dim3((d + 7) / 8, nacc), dim3(8 * FIN_GROUPS), 0, stream, partials, nblk, nacc, d, o0, o1, o2, o3);
    PB_LAUNCH_CHECK();
    return 0;
}

int pb_finalize_rows(const float* partials, int nblk, int d, float* out, void* stream, int nacc, float* out1) {
    return launch_finalize(partials, nblk, nacc, d, out, out1, nullptr, nullptr, (hipStream_t)stream);
}

static int add_ln_bwd_impl(const void* dy, const void* res, const void* a, const float* ln_w, const float* mean,
                           const float* rstd, void* dres, void* da, float* dgamma, float* dbeta, float* dbias_a,
                           float* partials, int32_t T, int32_t d, int32_t dtype, int32_t dres_f32, int32_t accum_dres,
                           uint64_t seed, uint32_t site, float p_drop, void* stream_, const int32_t* row_ids) {
    hipStream_t stream = (hipStream_t)stream_;
    if (check_ln_dims("pb_add_ln_bwd", T, d)) return -2;
    if (T == 0) return 0;
    const int grid = ln_grid(T);
    const size_t lds = (size_t)LN_WAVES * d * sizeof(float);
    if (float* slice = pb_defer_alloc((size_t)grid * 3 * d)) partials = slice;     // deferred reduction: the partial rows must outlive this call
    PB_LN_DISPATCH(nit_for(d),
        if (dtype == PB_BF16) {
            if (dres_f32)
                hipLaunchKernelGGL((add_ln_bwd_kernel<bf16_t, float, NIT>), dim3(grid), dim3(LN_THREADS), lds, stream,
                                   (const bf16_t*)dy, (const bf16_t*)res, (const bf16_t*)a, ln_w, mean, rstd, (float*)dres,
                                   (bf16_t*)da, partials, T, d, accum_dres, seed, site, p_drop, row_ids);
            else
                hipLaunchKernelGGL((add_ln_bwd_kernel<bf16_t, bf16_t, NIT>), dim3(grid), dim3(LN_THREADS), lds, stream,
                                   (const bf16_t*)dy, (const bf16_t*)res, (const bf16_t*)a, ln_w, mean, rstd, (bf16_t*)dres,
                                   (bf16_t*)da, partials, T, d, accum_dres, seed, site, p_drop, row_ids);
        } else {
            hipLaunchKernelGGL((add_ln_bwd_kernel<float, float, NIT>), dim3(grid), dim3(LN_THREADS), lds, stream,
                               (const float*)dy, (const float*)res, (const float*)a, ln_w, mean, rstd, (float*)dres,
                               (float*)da, partials, T, d, accum_dres, seed, site, p_drop, row_ids);
        });
    PB_LAUNCH_CHECK();
    return launch_finalize(partials, grid, 3, d, dgamma, dbeta, dbias_a, nullptr, stream);
}
extern "C" int pb_add_ln_bwd(const void* dy, const void* res, const void* a, const float* ln_w, const float* mean,
                             const float* rstd, void* dres, void* da, float* dgamma, float* dbeta, float* dbias_a,
                             float* partials, int32_t T, int32_t d, int32_t dtype, int32_t dres_f32, int32_t accum_dres,
                             uint64_t seed, uint32_t site, float p_drop, void* stream_) {
    return add_ln_bwd_impl(dy, res, a, ln_w, mean, rstd, dres, da, dgamma, dbeta, dbias_a, partials, T, d, dtype, dres_f32, accum_dres, seed, site,
                           p_drop, stream_, nullptr);
}
extern "C" int pb_add_ln_bwd_packed(const void* dy, const void* res, const void* a, const float* ln_w, const float* mean,
                                    const float* rstd, void* dres, void* da, float* dgamma, float* dbeta, float* dbias_a,
                                    float* partials, const int32_t* row_ids, int32_t T, int32_t d, int32_t dtype, int32_t dres_f32,
                                    int32_t accum_dres, uint64_t seed, uint32_t site, float p_drop, void* stream_) {
    PB_REQUIRE(row_ids != nullptr, "pb_add_ln_bwd_packed: row_ids is required");
    return add_ln_bwd_impl(dy, res, a, ln_w, mean, rstd, dres, da, dgamma, dbeta, dbias_a, partials, T, d, dtype, dres_f32, accum_dres, seed, site,
                           p_drop, stream_, row_ids);
}

static int embed_ln_fwd_impl(const int16_t* ids16, const float* P, const int32_t* seg_off, const float* lin_bias,
                             const float* pos, const float* ln_w, const float* ln_b, void* y, float* mean, float* rstd,
                             int32_t T, int32_t S, int32_t d, int32_t dtype, float eps, uint64_t seed, uint32_t site,
                             float p_drop, void* stream_, const int32_t* row_ids) {
    hipStream_t stream = (hipStream_t)stream_;
    if (check_ln_dims("pb_embed_ln_fwd", T, d)) return -2;
    PB_REQUIRE(S > 0 && (row_ids || T % S == 0), "pb_embed_ln_fwd: T=%d not a multiple of S=%d", T, S);
    if (T == 0) return 0;
    SegOff so;
    for (int i = 0; i < 8; ++i) so.off[i] = seg_off[i];
    const int grid = ln_grid_fwd(T);
    PB_LN_DISPATCH(nit_for(d),
        if (dtype == PB_BF16)
            hipLaunchKernelGGL((embed_ln_fwd_kernel<bf16_t, NIT>), dim3(grid), dim3(LN_THREADS), 0, stream, ids16, P, so, lin_bias,
                               pos, ln_w, ln_b, (bf16_t*)y, mean, rstd, T, S, d, eps, seed, site, p_drop, row_ids);
        else
            hipLaunchKernelGGL((embed_ln_fwd_kernel<float, NIT>), dim3(grid), dim3(LN_THREADS), 0, stream, ids16, P, so, lin_bias,
                               pos, ln_w, ln_b, (float*)y, mean, rstd, T, S, d, eps, seed, site, p_drop, row_ids));
    PB_LAUNCH_CHECK();
    return 0;
}
extern "C" int pb_embed_ln_fwd(const int16_t* ids16, const float* P, const int32_t* seg_off, const float* lin_bias,
                               const float* pos, const float* ln_w, const float* ln_b, void* y, float* mean, float* rstd,
                               int32_t T, int32_t S, int32_t d, int32_t dtype, float eps, uint64_t seed, uint32_t site,
                               float p_drop, void* stream_) {
    return embed_ln_fwd_impl(ids16, P, seg_off, lin_bias, pos, ln_w, ln_b, y, mean, rstd, T, S, d, dtype, eps, seed, site, p_drop, stream_, nullptr);
}
extern "C" int pb_embed_ln_fwd_packed(const int16_t* ids16, const int32_t* row_ids, const float* P, const int32_t* seg_off, const float* lin_bias,
                                      const float* pos, const float* ln_w, const float* ln_b, void* y, float* mean, float* rstd,
                                      int32_t T, int32_t S, int32_t d, int32_t dtype, float eps, uint64_t seed, uint32_t site,
                                      float p_drop, void* stream_) {
    PB_REQUIRE(row_ids != nullptr, "pb_embed_ln_fwd_packed: row_ids is required");
    return embed_ln_fwd_impl(ids16, P, seg_off, lin_bias, pos, ln_w, ln_b, y, mean, rstd, T, S, d, dtype, eps, seed, site, p_drop, stream_, row_ids);
}

static int embed_ln_bwd_impl(const void* dy, const int16_t* ids16, const float* P, const int32_t* seg_off,
                             const float* lin_bias, const float* pos, const float* ln_w, const float* mean,
                             const float* rstd, float* dP, float* dpos, float* dbias, float* dgamma, float* dbeta,
                             float* partials, void* dz_out, int32_t T, int32_t S, int32_t d, int32_t dtype, uint64_t seed,
                             uint32_t site, float p_drop, void* stream_, const int32_t* row_ids) {
    hipStream_t stream = (hipStream_t)stream_;
    if (check_ln_dims("pb_embed_ln_bwd", T, d)) return -2;
    PB_REQUIRE(S > 0 && (row_ids || T % S == 0), "pb_embed_ln_bwd: T=%d not a multiple of S=%d", T, S);
    if (T == 0) return 0;
    SegOff so;
    for (int i = 0; i < 8; ++i) so.off[i] = seg_off[i];
    const int grid = ln_grid(T);
    const size_t lds = (size_t)LN_WAVES * d * sizeof(float);
    PB_LN_DISPATCH(nit_for(d),
        if (dtype == PB_BF16)
            hipLaunchKernelGGL((embed_ln_bwd_kernel<bf16_t, NIT>), dim3(grid), dim3(LN_THREADS), lds, stream, (const bf16_t*)dy,
                               ids16, P, so, lin_bias, pos, ln_w, mean, rstd, dP, dpos, partials, (bf16_t*)dz_out, T, S, d, seed, site, p_drop, row_ids);
        else
            hipLaunchKernelGGL((embed_ln_bwd_kernel<float, NIT>), dim3(grid), dim3(LN_THREADS), lds, stream, (const float*)dy,
                               ids16, P, so, lin_bias, pos, ln_w, mean, rstd, dP, dpos, partials, (float*)dz_out, T, S, d, seed, site, p_drop, row_ids));
    PB_LAUNCH_CHECK();
    return launch_finalize(partials, grid, 3, d, dgamma, dbeta, dbias, nullptr, stream);
}
extern "C" int pb_embed_ln_bwd(const void* dy, const int16_t* ids16, const float* P, const int32_t* seg_off,
                               const float* lin_bias, const float* pos, const float* ln_w, const float* mean,
                               const float* rstd, float* dP, float* dpos, float* dbias, float* dgamma, float* dbeta,
                               float* partials, void* dz_out, int32_t T, int32_t S, int32_t d, int32_t dtype, uint64_t seed,
                               uint32_t site, float p_drop, void* stream_) {
    return embed_ln_bwd_impl(dy, ids16, P, seg_off, lin_bias, pos, ln_w, mean, rstd, dP, dpos, dbias, dgamma, dbeta, partials, dz_out, T, S, d,
                             dtype, seed, site, p_drop, stream_, nullptr);
}
extern "C" int pb_embed_ln_bwd_packed(const void* dy, const int16_t* ids16, const int32_t* row_ids, const float* P, const int32_t* seg_off,
                                      const float* lin_bias, const float* pos, const float* ln_w, const float* mean,
                                      const float* rstd, float* dP, float* dpos, float* dbias, float* dgamma, float* dbeta,
                                      float* partials, void* dz_out, int32_t T, int32_t S, int32_t d, int32_t dtype, uint64_t seed,
                                      uint32_t site, float p_drop, void* stream_) {
    PB_REQUIRE(row_ids != nullptr, "pb_embed_ln_bwd_packed: row_ids is required");
    return embed_ln_bwd_impl(dy, ids16, P, seg_off, lin_bias, pos, ln_w, mean, rstd, dP, dpos, dbias, dgamma, dbeta, partials, dz_out, T, S, d,
                             dtype, seed, site, p_drop, stream_, row_ids);
}

extern "C" int64_t pb_colsum_partials_floats(int32_t N) { return (int64_t)256 * N; }

extern "C" int pb_colsum(const void* dy, int64_t ld, float* out, float* partials, int32_t T, int32_t N, int32_t dtype,
                         int32_t src_f32, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    PB_REQUIRE(T >= 0 && N > 0 && N % 4 == 0 && ld % 4 == 0, "pb_colsum: N and ld must be multiples of 4");
    if (T == 0) return 0;
    const int nrb = max(1, min(128, (T + 63) / 64));
    const int rpb = (T + nrb - 1) / nrb;
    dim3 grid((N + 255) / 256, nrb);
    if (src_f32 || dtype == PB_F32)
        hipLaunchKernelGGL((colsum_kernel<float>), grid, dim3(256), 0, stream, (const float*)dy, (long)ld, partials, T, N, rpb);
    else
        hipLaunchKernelGGL((colsum_kernel<bf16_t>), grid, dim3(256), 0, stream, (const bf16_t*)dy, (long)ld, partials, T, N, rpb);
    PB_LAUNCH_CHECK();
    return launch_finalize(partials, nrb, 1, N, out, nullptr, nullptr, nullptr, stream);
}

extern "C" int pb_ids_to_i16(const int64_t* ids, int16_t* out, int64_t n, void* stream_);
namespace {
// onehot (T, V) bf16: row t has ones at the 8 columns off_i + id_i[t]. One wave per row, 16-byte stores.
__global__ __launch_bounds__(256) void onehot_kernel(const int16_t* __restrict__ ids16, const SegOff so, bf16_t* __restrict__ out, long T, int V) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long row = (long)blockIdx.x * 4 + wave; row < T; row += (long)gridDim.x * 4) {
        const uint4 raw = *reinterpret_cast<const uint4*>(ids16 + row * 8);
        int col[8];
        col[0] = so.off[0] + (int)(raw.x & 0xffff); col[1] = so.off[1] + (int)(raw.x >> 16);
        col[2] = so.off[2] + (int)(raw.y & 0xffff); col[3] = so.off[3] + (int)(raw.y >> 16);
        col[4] = so.off[4] + (int)(raw.z & 0xffff); col[5] = so.off[5] + (int)(raw.z >> 16);
        col[6] = so.off[6] + (int)(raw.w & 0xffff); col[7] = so.off[7] + (int)(raw.w >> 16);
        for (int c8 = lane; c8 * 8 < V; c8 += 64) {
            bf16x8 v = {};
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if ((col[i] >> 3) == c8) v[col[i] & 7] = (bf16_t)1.0f;
            *reinterpret_cast<bf16x8*>(out + row * V + c8 * 8) = v;
        }
    }
}
// out[s][c] += sum_b x[(b*S + s)*d + c]   (position-table gradient)
template <typename T>
__global__ __launch_bounds__(256) void batch_sum_kernel(const T* __restrict__ x, float* __restrict__ out, int B, long Sd) {
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < Sd; i += (long)gridDim.x * 1024) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int b = 0; b < B; ++b) s += load4(x + b * Sd + i);
        store4(out + i, load4(out + i) + s);
    }
}
__global__ void ids_to_i16_kernel(const int64_t* __restrict__ ids, int16_t* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int64_t v = ids[i];
        out[i] = (v < 0 || v > 32767) ? (int16_t)-1 : (int16_t)v;        // not an int16: stays recognisable for pb_ids_check
    }
}
// nn.Embedding raises IndexError on an id outside its table (PianoBart.py:15-16); a kernel cannot, so it leaves a mark
__global__ void ids_check_kernel(int16_t* __restrict__ ids, long n, const int* __restrict__ limits, int* __restrict__ flag) {
    bool bad = false;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int v = ids[i];
        if (v < 0 || v >= limits[i & 7]) { bad = true; ids[i] = 0; }     // the gathers enqueued behind this kernel stay inside their tables
    }
    if (__builtin_amdgcn_ballot_w64(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}
__global__ void shift_right_kernel(const int16_t* __restrict__ ids, const int16_t* __restrict__ sos, int16_t* __restrict__ out, int B, int S) {
    const long n = (long)B * S * 8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i & 7);
        const long tok = i >> 3;
        const int s = (int)(tok % S);
        out[i] = s == 0 ? sos[c] : ids[i - 8];
    }
}
}  // namespace
extern "C" int pb_ids_to_i16(const int64_t* ids, int16_t* out, int64_t n, void* stream_) {
    if (n <= 0) return 0;
    const int grid = (int)min((long)2048, (long)((n + 255) / 256));
    hipLaunchKernelGGL(ids_to_i16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream_, ids, out, (long)n);
    PB_LAUNCH_CHECK();
    return 0;
}
extern "C" int pb_ids_check(int16_t* ids16, int64_t n, const int32_t* limits8, int32_t* flag, void* stream_) {
    PB_REQUIRE(ids16 && limits8 && flag && n % 8 == 0, "pb_ids_check: null argument or n not a multiple of 8");
    if (n <= 0) return 0;
    const int grid = (int)min((long)1024, (long)((n + 255) / 256));
    hipLaunchKernelGGL(ids_check_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream_, ids16, (long)n, limits8, flag);
    PB_LAUNCH_CHECK();
    return 0;
}
extern "C" int pb_shift_right(const int16_t* ids, const int16_t* sos_row, int16_t* out, int32_t B, int32_t S, void* stream_) {
    const long n = (long)B * S * 8;
    if (n <= 0) return 0;
    const int grid = (int)min((long)2048, (n + 255) / 256);
    hipLaunchKernelGGL(shift_right_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream_, ids, sos_row, out, B, S);
    PB_LAUNCH_CHECK();
    return 0;
}

extern "C" int pb_onehot_build(const int16_t* ids16, const int32_t* seg_off, void* out, int64_t T, int32_t V, void* stream_) {
    PB_REQUIRE(V % 8 == 0, "pb_onehot_build: V must be a multiple of 8");
    if (T <= 0) return 0;
    SegOff so;
    for (int i = 0; i < 8; ++i) so.off[i] = seg_off[i];
    const int grid = (int)min((long)4096, (long)((T + 3) / 4));
    hipLaunchKernelGGL(onehot_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream_, ids16, so, (bf16_t*)out, (long)T, V);
    PB_LAUNCH_CHECK();
    return 0;
}
extern "C" int pb_batch_sum(const void* x, float* out, int32_t B, int64_t Sd, int32_t dtype, void* stream_) {
    PB_REQUIRE(Sd % 4 == 0, "pb_batch_sum: S*d must be a multiple of 4");
    if (B <= 0 || Sd <= 0) return 0;
    const int grid = (int)min((long)2048, (long)((Sd / 4 + 255) / 256));
    if (dtype == PB_BF16) hipLaunchKernelGGL((batch_sum_kernel<bf16_t>), dim3(grid), dim3(256), 0, (hipStream_t)stream_, (const bf16_t*)x, out, B, (long)Sd);
    else hipLaunchKernelGGL((batch_sum_kernel<float>), dim3(grid), dim3(256), 0, (hipStream_t)stream_, (const float*)x, out, B, (long)Sd);
    PB_LAUNCH_CHECK();
    return 0;
}
