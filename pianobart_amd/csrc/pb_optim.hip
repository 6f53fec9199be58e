// K10/K11: multi-tensor (flat-buffer) gradient norm, clip coefficient, HF-AdamW step with bf16
// shadow refresh, and casts. All HBM-bound streaming kernels over ONE flat f32 buffer per role
// (params / grads / exp_avg / exp_avg_sq), float4 per lane, grid-stride, <= 2048 workgroups.
// HF AdamW (transformers 4.29.2 optimization.py, pretrain.py:76,196), per element:
//   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= lr*sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v)+eps)
//   p -= lr * wd * p        (decoupled decay AFTER the update)
#include "pb_common.h"
#include "pb_api_internal.h"
#include <algorithm>
#include <cmath>

namespace {

constexpr int OPT_BLOCKS = 2048, OPT_THREADS = 256;

__global__ __launch_bounds__(OPT_THREADS) void sqnorm_kernel(const float* __restrict__ g, long n, float* __restrict__ partials) {
    __shared__ float red[OPT_THREADS / 64];
    float s = 0.f;
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * OPT_THREADS + threadIdx.x; i < n4; i += (long)gridDim.x * OPT_THREADS) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(g + 4 * i);
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[(n4 << 2) + threadIdx.x]; s += v * v; }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void sqnorm_finalize_kernel(const float* __restrict__ partials, int nblk, float* __restrict__ out) {
    __shared__ double red[256];
    double s = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 256) s += (double)partials[b];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)red[0];
}

__global__ void clip_coef_kernel(const float* __restrict__ sq, float max_norm, float gscale, float* __restrict__ coef) {
    if (threadIdx.x == 0) {
        const float total = sqrtf(sq[0]) * gscale;                       // norm of the (scaled) gradient
        const float r = max_norm / (total + 1e-6f);                      // torch clip_grad_norm_ rule: clamp(r, max=1), and clamp hands a NaN on
        coef[0] = (r != r ? r : fminf(1.0f, r)) * gscale;                // (a NaN gradient norm poisons every parameter in the reference; fminf would drop it)
    }
}

__global__ __launch_bounds__(OPT_THREADS) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
        float* __restrict__ v, bf16_t* __restrict__ shadow, long n, const float* __restrict__ clip_coef, float step_size,
        float b1, float b2, float eps, float decay) {
    const float gs = clip_coef ? clip_coef[0] : 1.0f;
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * OPT_THREADS + threadIdx.x; i < n4; i += (long)gridDim.x * OPT_THREADS) {
        f32x4 pv = *reinterpret_cast<f32x4*>(p + 4 * i);
        const f32x4 gv = *reinterpret_cast<const f32x4*>(g + 4 * i) * gs;
        f32x4 mv = *reinterpret_cast<f32x4*>(m + 4 * i);
        f32x4 vv = *reinterpret_cast<f32x4*>(v + 4 * i);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            mv[j] = b1 * mv[j] + (1.0f - b1) * gv[j];
            vv[j] = b2 * vv[j] + (1.0f - b2) * gv[j] * gv[j];
            pv[j] -= step_size * mv[j] / (sqrtf(vv[j]) + eps);
            pv[j] -= decay * pv[j];
        }
        *reinterpret_cast<f32x4*>(p + 4 * i) = pv;
        *reinterpret_cast<f32x4*>(m + 4 * i) = mv;
        *reinterpret_cast<f32x4*>(v + 4 * i) = vv;
        if (shadow) store4(shadow + 4 * i, pv);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long i = (n4 << 2) + threadIdx.x;
        const float gg = g[i] * gs;
        const float mm = b1 * m[i] + (1.0f - b1) * gg;
        const float vv = b2 * v[i] + (1.0f - b2) * gg * gg;
        float pp = p[i] - step_size * mm / (sqrtf(vv) + eps);
        pp -= decay * pp;
        p[i] = pp; m[i] = mm; v[i] = vv;
        if (shadow) shadow[i] = (bf16_t)pp;
    }
}

template <typename S, typename D>
__global__ __launch_bounds__(OPT_THREADS) void cast_kernel(const S* __restrict__ src, D* __restrict__ dst, long n) {
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * OPT_THREADS + threadIdx.x; i < n4; i += (long)gridDim.x * OPT_THREADS)
        store4(dst + 4 * i, load4(src + 4 * i));
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) dst[(n4 << 2) + threadIdx.x] = from_f<D>(to_f(src[(n4 << 2) + threadIdx.x]));
}

__global__ __launch_bounds__(OPT_THREADS) void fill_kernel(float* __restrict__ dst, float v, long n) {
    for (long i = (long)blockIdx.x * OPT_THREADS + threadIdx.x; i < n; i += (long)gridDim.x * OPT_THREADS) dst[i] = v;
}

// dst[i] = bf16( sum_r f32(src[r][i]) ): the owner's f32 accumulation of the bf16 gradient chunks its peers sent (parallel.py)
__global__ __launch_bounds__(OPT_THREADS) void sum_rows_bf16_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int R, long n) {
    const long n8 = n >> 3;
    for (long i = (long)blockIdx.x * OPT_THREADS + threadIdx.x; i < n8; i += (long)gridDim.x * OPT_THREADS) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < R; ++r) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(src + (size_t)r * n + 8 * i);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
        }
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16_t)acc[j];
        *reinterpret_cast<bf16x8*>(dst + 8 * i) = o;
    }
}

int opt_grid(long n4) { return (int)std::max(1L, std::min((long)OPT_BLOCKS, (n4 + OPT_THREADS - 1) / OPT_THREADS)); }

// Batched bf16 transpose inside one flat buffer: table entry e = {element offset, R, C, first tile}; matrix e (R x C, row-major at
// src + offset) is written C x R at dst + offset. One workgroup per 64 x 64 tile, through LDS. Dimensions multiples of 8.
__global__ __launch_bounds__(256) void transpose_batch_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, const int4* __restrict__ table, int n) {
    __shared__ __attribute__((aligned(16))) short tile[64][72];        // 144-byte rows: 16-byte aligned pieces, and a column walk (stride 36 words) spreads over the banks
    __shared__ int4 ent;
    const int t = threadIdx.x, b = blockIdx.x;
    // which matrix owns tile b: every thread tests one table entry (a walk of the table by one thread was ~85 dependent loads per workgroup: 16 us of the
    // workgroup's life, the whole kernel's 200 us)
    for (int e = t; e < n; e += 256) {
        const int4 cur = table[e];
        if (cur.w <= b && (e + 1 == n || table[e + 1].w > b)) ent = cur;
    }
    __syncthreads();
    const long off = ent.x;
    const int R = ent.y, C = ent.z, lt = b - ent.w, tc = (C + 63) / 64;
    const int r0 = (lt / tc) * 64, c0 = (lt % tc) * 64;
    const short* S = reinterpret_cast<const short*>(src) + off;
    short* D = reinterpret_cast<short*>(dst) + off;
    // a row of the tile is 128 bytes = the 16-byte pieces of 8 ADJACENT lanes (the memory pipeline merges adjacent lanes only: 16-byte pieces 32 bytes
    // apart, as in the first form of this kernel, are separate requests), 32 rows per instruction, two instructions per side
    {
        const int r = t >> 3, c = (t & 7) * 8;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int rr = r + 32 * h;
            if (r0 + rr < R && c0 + c < C) {
                *reinterpret_cast<uint4*>(&tile[rr][c]) = *reinterpret_cast<const uint4*>(S + (long)(r0 + rr) * C + c0 + c);
            }
        }
    }
    __syncthreads();
    {
        const int orow = t >> 3, oc = (t & 7) * 8;               // output row = source column
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int oo = orow + 32 * h;
            if (c0 + oo < C && r0 + oc < R) {
                uint4 v;
                short* e = reinterpret_cast<short*>(&v);
#pragma unroll
                for (int j = 0; j < 8; ++j) e[j] = tile[oc + j][oo];
                *reinterpret_cast<uint4*>(D + (long)(c0 + oo) * R + r0 + oc) = v;
            }
        }
    }
}

}  // namespace

// ---- fine-tune regulariser `loss += weight * torch.norm(param, p=2)` per parameter tensor (finetune.py:241-243)
constexpr int L2_BLOCKS = 256;
__global__ __launch_bounds__(256) void l2_partial_kernel(const float* __restrict__ p, long n, float* __restrict__ part) {
    __shared__ float red[4];
    float s = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) s += p[i] * p[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// every block re-derives ||p||^2 from the (<= 256) partials in the same fixed order, then g += weight * p / ||p||
// (0 where ||p|| = 0, torch's norm backward); block 0 adds weight * ||p|| to the loss accumulator.
__global__ __launch_bounds__(256) void l2_apply_kernel(const float* __restrict__ p, float* __restrict__ g, long n, const float* __restrict__ part,
                                                       int nblk, float weight, float* __restrict__ loss_acc) {
    __shared__ float red[4];
    float s = (int)threadIdx.x < nblk ? part[threadIdx.x] : 0.f;
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float sq = (red[0] + red[1]) + (red[2] + red[3]);
    const float nrm = sqrtf(sq);
    if (blockIdx.x == 0 && threadIdx.x == 0 && loss_acc) *loss_acc += weight * nrm;
    if (!g || sq <= 0.f) return;
    const float k = weight / nrm;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) g[i] += k * p[i];
}

extern "C" int64_t pb_l2_penalty_scratch_floats(void) { return L2_BLOCKS; }

extern "C" int pb_l2_penalty(const float* p, float* g, int64_t n, float weight, float* scratch, float* loss_acc, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    PB_REQUIRE(n >= 0 && p && scratch, "pb_l2_penalty: p and scratch are required");
    if (n == 0) return 0;
    const int grid = (int)std::max<long>(1, std::min<long>(L2_BLOCKS, (n + 1023) / 1024));
    hipLaunchKernelGGL(l2_partial_kernel, dim3(grid), dim3(256), 0, stream, p, (long)n, scratch);
    PB_LAUNCH_CHECK();
    hipLaunchKernelGGL(l2_apply_kernel, dim3(g ? grid : 1), dim3(256), 0, stream, p, g, (long)n, scratch, grid, weight, loss_acc);
    PB_LAUNCH_CHECK();
    return 0;
}

extern "C" int64_t pb_norm_partials_floats(void) { return OPT_BLOCKS; }

extern "C" int pb_grad_sqnorm(const float* g, int64_t n, float* partials, float* out_sq, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    PB_REQUIRE(n >= 0 && ((uintptr_t)g % 16 == 0), "pb_grad_sqnorm: buffer must be 16-byte aligned");
    const int grid = opt_grid(n >> 2);
    hipLaunchKernelGGL(sqnorm_kernel, dim3(grid), dim3(OPT_THREADS), 0, stream, g, (long)n, partials);
    PB_LAUNCH_CHECK();
    hipLaunchKernelGGL(sqnorm_finalize_kernel, dim3(1), dim3(256), 0, stream, partials, grid, out_sq);
    PB_LAUNCH_CHECK();
    return 0;
}

extern "C" int pb_clip_coef(const float* sq, float max_norm, float gscale, float* coef, void* stream_) {
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream_, sq, max_norm, gscale, coef);
    PB_LAUNCH_CHECK();
    return 0;
}

extern "C" int pb_adamw_step(float* p, const float* g, float* m, float* v, void* shadow, int64_t n, const float* clip_coef,
                             float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step, void* stream_) {
    PB_REQUIRE(step >= 1, "pb_adamw_step: step must be >= 1");
    PB_REQUIRE(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && ((uintptr_t)m % 16 == 0) && ((uintptr_t)v % 16 == 0),
               "pb_adamw_step: buffers must be 16-byte aligned");
    if (n <= 0) return 0;
    const double bc1 = 1.0 - std::pow((double)beta1, (double)step), bc2 = 1.0 - std::pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr * std::sqrt(bc2) / bc1);
    hipLaunchKernelGGL(adamw_kernel, dim3(opt_grid(n >> 2)), dim3(OPT_THREADS), 0, (hipStream_t)stream_, p, g, m, v, (bf16_t*)shadow,
                       (long)n, clip_coef, step_size, beta1, beta2, eps, lr * weight_decay);
    PB_LAUNCH_CHECK();
    return 0;
}

extern "C" int pb_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream_) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL((cast_kernel<float, bf16_t>), dim3(opt_grid(n >> 2)), dim3(OPT_THREADS), 0, (hipStream_t)stream_, src, (bf16_t*)dst, (long)n);
    PB_LAUNCH_CHECK();
    return 0;
}
extern "C" int pb_cast_bf16_to_f32(const void* src, float* dst, int64_t n, void* stream_) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL((cast_kernel<bf16_t, float>), dim3(opt_grid(n >> 2)), dim3(OPT_THREADS), 0, (hipStream_t)stream_, (const bf16_t*)src, dst, (long)n);
    PB_LAUNCH_CHECK();
    return 0;
}
extern "C" int pb_transpose_batch_bf16(const void* src, void* dst, const int32_t* table, int32_t n_matrices, int32_t n_tiles, void* stream_) {
    PB_REQUIRE(src && dst && table && src != dst, "pb_transpose_batch_bf16: NULL or aliased argument");
    if (n_matrices <= 0 || n_tiles <= 0) return 0;
    hipLaunchKernelGGL(transpose_batch_kernel, dim3(n_tiles), dim3(256), 0, (hipStream_t)stream_, (const bf16_t*)src, (bf16_t*)dst,
                       reinterpret_cast<const int4*>(table), n_matrices);
    PB_LAUNCH_CHECK();
    return 0;
}
extern "C" int pb_fill_f32(float* dst, float value, int64_t n, void* stream_) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(fill_kernel, dim3(opt_grid(n >> 2)), dim3(OPT_THREADS), 0, (hipStream_t)stream_, dst, value, (long)n);
    PB_LAUNCH_CHECK();
    return 0;
}
extern "C" int pb_sum_rows_bf16(const void* src, void* dst, int32_t rows, int64_t n, void* stream_) {
    PB_REQUIRE(rows >= 1 && n >= 0 && n % 8 == 0 && ((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 16 == 0),
               "pb_sum_rows_bf16: n must be a multiple of 8 and the buffers 16-byte aligned");
    if (n == 0) return 0;
    hipLaunchKernelGGL(sum_rows_bf16_kernel, dim3(opt_grid(n >> 3)), dim3(OPT_THREADS), 0, (hipStream_t)stream_, (const bf16_t*)src, (bf16_t*)dst,
                       rows, (long)n);
    PB_LAUNCH_CHECK();
    return 0;
}
