// K9: fused 8-head (segmented) log-softmax + cross-entropy + argmax + masked accuracy, with the
// gradient w.r.t. the logits produced in the same pass. Replaces pretrain.py:163-189: 8 full-logit
// D2H copies + np.argmax, 8 x (permute + CrossEntropyLoss(reduction='none') * mask, sum/sum).
// One wave64 per token row of V = sum n_i logits (f32); lane i < 8 carries head i's scalars.
#include "pb_common.h"
#include "pb_api_internal.h"
#include <algorithm>

namespace {

struct Seg9 { int off[9]; };
constexpr int CE_MAX_BLOCKS = 2048;

template <typename T>
__global__ __launch_bounds__(256) void ce_kernel(const float* __restrict__ logits, const int16_t* __restrict__ target,
        const float* __restrict__ loss_mask, const Seg9 so, float* __restrict__ partials, const float* __restrict__ coef,
        T* __restrict__ dlogits, int16_t* __restrict__ argmax_out, int rows, int V) {
    __shared__ float red[4][24];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float a_ce = 0.f, a_m = 0.f, a_ok = 0.f;       // lane i<8: running sums of head i
    for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
        const float* x = logits + row * V;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int o = so.off[i], n = so.off[i + 1] - o;
            const int tgt = target[row * 8 + i];
            const float m = loss_mask[row * 8 + i];
            // max + first-argmax (np.argmax tie rule: lowest index)
            float mx = -INFINITY; int am = 0x7fffffff;
            for (int c = lane; c < n; c += 64) {
                const float v = x[o + c];
                if (v > mx) { mx = v; am = c; }
            }
#pragma unroll
            for (int sft = 32; sft > 0; sft >>= 1) {
                const float ov = __shfl_xor(mx, sft, 64);
                const int oa = __shfl_xor(am, sft, 64);
                if (ov > mx || (ov == mx && oa < am)) { mx = ov; am = oa; }
            }
            float se = 0.f;
            for (int c = lane; c < n; c += 64) se += __expf(x[o + c] - mx);
            se = wave_sum(se);
            const float xt = (tgt >= 0 && tgt < n) ? x[o + tgt] : 0.f;
            const float ce = __logf(se) + mx - xt;
            if (lane == i) {
                a_ce += ce * m; a_m += m; a_ok += (am == tgt ? 1.f : 0.f) * m;
                if (argmax_out) argmax_out[row * 8 + i] = (int16_t)am;
            }
            if (dlogits) {
                const float k = coef[i] * m;                                 // empty head: inf * 0 = NaN, as in pretrain.py:117
                const float inv = 1.0f / se;
                T* g = dlogits + row * V + o;
                for (int c = lane; c < n; c += 64) {
                    const float pr = __expf(x[o + c] - mx) * inv;
                    g[c] = from_f<T>(k * (pr - (c == tgt ? 1.f : 0.f)));
                }
            }
        }
    }
    if (lane < 8) { red[wave][lane] = a_ce; red[wave][8 + lane] = a_m; red[wave][16 + lane] = a_ok; }
    __syncthreads();
    if (threadIdx.x < 24)
        partials[(size_t)blockIdx.x * 24 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// Register-resident form for heads of at most 64 * CE_K classes (the Octuple heads: <= 262): the row's logits are loaded ONCE, all
// loads of a row in flight together, lane l holding classes l, l + 64, ... of every head; max / first-argmax / sum-exp are VALU-only
// wave reductions (DPP + permlane swaps). The generic kernel above re-reads the row three times behind dependent LDS-routed
// shuffles and was pure latency: 310 us for the 168 MB of cfg-2 logits (0.8 TB/s). Same per-lane summation order -> same bits.
constexpr int CE_K = 5;
template <typename T>
__global__ __launch_bounds__(256) void ce_rows_reg_kernel(const float* __restrict__ logits, const int16_t* __restrict__ target,
        const float* __restrict__ loss_mask, const Seg9 so, float* __restrict__ partials, const float* __restrict__ coef,
        T* __restrict__ dlogits, int16_t* __restrict__ argmax_out, int rows, int V) {
    __shared__ float red[4][24];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float a_ce = 0.f, a_m = 0.f, a_ok = 0.f;       // lane i<8: running sums of head i
    for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
        const float* x = logits + row * V;
        float v[8][CE_K];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int o = so.off[i], n = so.off[i + 1] - o;
#pragma unroll
            for (int k = 0; k < CE_K; ++k) {
                const int c = lane + 64 * k;
                v[i][k] = c < n ? x[o + c] : -INFINITY;
            }
        }
        const int my_t = lane < 8 ? (int)target[row * 8 + lane] : 0;
        const float my_m = lane < 8 ? loss_mask[row * 8 + lane] : 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int o = so.off[i], n = so.off[i + 1] - o;
            const int tgt = __builtin_amdgcn_readlane(my_t, i);
            const float m = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_m), i));
            float mx = v[i][0];
#pragma unroll
            for (int k = 1; k < CE_K; ++k) mx = fmaxf(mx, v[i][k]);
            mx = wave_max(mx);
            // first index attaining the maximum (np.argmax tie rule): class ids < 2^24 are exact in f32
            float neg_first = -16777216.f;
#pragma unroll
            for (int k = CE_K - 1; k >= 0; --k) neg_first = v[i][k] == mx ? -(float)(lane + 64 * k) : neg_first;
            const int am = (int)(-wave_max(neg_first));
            float e[CE_K], se = 0.f;
#pragma unroll
            for (int k = 0; k < CE_K; ++k) {
                e[k] = lane + 64 * k < n ? __expf(v[i][k] - mx) : 0.f;
                se += e[k];                                         // same order as the generic kernel's strided loop
            }
            se = wave_sum(se);
            float xt = 0.f;
            if (tgt >= 0 && tgt < n) {                              // wave-uniform
                const int tk = tgt >> 6, tl = tgt & 63;
                float sel = v[i][0];
#pragma unroll
                for (int k = 1; k < CE_K; ++k) sel = tk == k ? v[i][k] : sel;
                xt = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sel), tl));
            }
            const float ce = __logf(se) + mx - xt;
            if (lane == i) {
                a_ce += ce * m; a_m += m; a_ok += (am == tgt ? 1.f : 0.f) * m;
                if (argmax_out) argmax_out[row * 8 + i] = (int16_t)am;
            }
            if (dlogits) {
                const float kk = coef[i] * m;                           // a head with no loss position in the batch: coef = inf, inf * 0 = NaN, as in pretrain.py:117
                const float inv = 1.0f / se;
                T* g = dlogits + row * V + o;
#pragma unroll
                for (int k = 0; k < CE_K; ++k) {
                    const int c = lane + 64 * k;
                    if (c < n) g[c] = from_f<T>(kk * (e[k] * inv - (c == tgt ? 1.f : 0.f)));
                }
            }
        }
    }
    if (lane < 8) { red[wave][lane] = a_ce; red[wave][8 + lane] = a_m; red[wave][16 + lane] = a_ok; }
    __syncthreads();
    if (threadIdx.x < 24)
        partials[(size_t)blockIdx.x * 24 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// 24 sums over the block partials: 32 row groups x 32 columns, 8 independent loads in flight per thread, fixed-order LDS tree
__global__ __launch_bounds__(1024) void ce_finalize_kernel(const float* __restrict__ partials, int nblk, float* __restrict__ sums) {
    __shared__ float red[32][32];
    const int k = threadIdx.x & 31, rg = threadIdx.x >> 5;
    float s = 0.f;
    if (k < 24) {
        int b = rg;
        for (; b + 7 * 32 < nblk; b += 8 * 32) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = partials[(size_t)(b + 32 * u) * 24 + k];
            s += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
        }
        for (; b < nblk; b += 32) s += partials[(size_t)b * 24 + k];
    }
    red[rg][k] = s;
    __syncthreads();
    if (rg < 8) red[rg][k] = (red[rg][k] + red[rg + 8][k]) + (red[rg + 16][k] + red[rg + 24][k]);
    __syncthreads();
    if (rg == 0 && k < 24) sums[k] += ((red[0][k] + red[1][k]) + (red[2][k] + red[3][k])) + ((red[4][k] + red[5][k]) + (red[6][k] + red[7][k]));
}

// counts: stage 1 = per-block column sums of a row range, stage 2 = sum of the block partials (deterministic)
constexpr int MC_BLOCKS = 128;
__global__ __launch_bounds__(256) void mask_count_kernel(const float* __restrict__ loss_mask, float* __restrict__ part, long T) {
    __shared__ float red[256];
    const int c = threadIdx.x & 7, r0 = threadIdx.x >> 3;
    float s = 0.f;
    for (long r = (long)blockIdx.x * 32 + r0; r < T; r += (long)gridDim.x * 32) s += loss_mask[r * 8 + c];
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < 8) {
        float t = 0.f;
        for (int k = 0; k < 32; ++k) t += red[k * 8 + threadIdx.x];
        part[blockIdx.x * 8 + threadIdx.x] = t;
    }
}
__global__ void mask_count_finalize_kernel(const float* __restrict__ part, int nblk, float* __restrict__ counts) {
    if (threadIdx.x < 8) {
        float t = 0.f;
        for (int b = 0; b < nblk; ++b) t += part[b * 8 + threadIdx.x];
        counts[threadIdx.x] = t;
    }
}

__global__ void loss_coef_kernel(const float* __restrict__ counts, const float* __restrict__ w, float* __restrict__ coef, float scale) {
    const int i = threadIdx.x;
    if (i >= 8) return;
    float sw = 0.f;
    for (int k = 0; k < 8; ++k) sw += w[k];
    coef[i] = scale * w[i] / (sw * counts[i]);
}

}  // namespace

extern "C" int64_t pb_ce_partials_floats(void) { return (int64_t)CE_MAX_BLOCKS * 24; }

extern "C" int pb_ce_fwd_bwd(const float* logits, const int16_t* target, const float* loss_mask, const int32_t* seg_off,
                             float* sums, float* partials, const float* coef, void* dlogits, int16_t* argmax_out, int32_t T,
                             int32_t V, int32_t dtype, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (T <= 0) return 0;
    PB_REQUIRE(seg_off[8] == V && seg_off[0] == 0, "pb_ce_fwd_bwd: segment offsets do not cover V=%d", V);
    PB_REQUIRE(dlogits == nullptr || coef != nullptr, "pb_ce_fwd_bwd: dlogits needs coef");
    Seg9 so;
    for (int i = 0; i < 9; ++i) so.off[i] = seg_off[i];
    const int grid = max(1, min(CE_MAX_BLOCKS, (T + 3) / 4));
    bool small_heads = true;
    for (int i = 0; i < 8; ++i) small_heads = small_heads && (seg_off[i + 1] - seg_off[i]) <= 64 * CE_K;
    if (small_heads) {
        if (dtype == PB_BF16)
            hipLaunchKernelGGL((ce_rows_reg_kernel<bf16_t>), dim3(grid), dim3(256), 0, stream, logits, target, loss_mask, so, partials, coef, (bf16_t*)dlogits, argmax_out, T, V);
        else
            hipLaunchKernelGGL((ce_rows_reg_kernel<float>), dim3(grid), dim3(256), 0, stream, logits, target, loss_mask, so, partials, coef, (float*)dlogits, argmax_out, T, V);
    } else if (dtype == PB_BF16)
        hipLaunchKernelGGL((ce_kernel<bf16_t>), dim3(grid), dim3(256), 0, stream, logits, target, loss_mask, so, partials, coef, (bf16_t*)dlogits, argmax_out, T, V);
    else
        hipLaunchKernelGGL((ce_kernel<float>), dim3(grid), dim3(256), 0, stream, logits, target, loss_mask, so, partials, coef, (float*)dlogits, argmax_out, T, V);
    PB_LAUNCH_CHECK();
    hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(1024), 0, stream, partials, grid, sums);
    PB_LAUNCH_CHECK();
    return 0;
}

extern "C" int pb_mask_count(const float* loss_mask, float* counts, float* partials, int64_t T, void* stream_) {
    const int nblk = (int)std::max(1L, std::min((long)MC_BLOCKS, (long)((T + 31) / 32)));
    hipLaunchKernelGGL(mask_count_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream_, loss_mask, partials, (long)T);
    PB_LAUNCH_CHECK();
    hipLaunchKernelGGL(mask_count_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream_, partials, nblk, counts);
    PB_LAUNCH_CHECK();
    return 0;
}

extern "C" int pb_loss_coef(const float* counts, const float* w, float* coef, float scale, void* stream_) {
    hipLaunchKernelGGL(loss_coef_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream_, counts, w, coef, scale);
    PB_LAUNCH_CHECK();
    return 0;
}
