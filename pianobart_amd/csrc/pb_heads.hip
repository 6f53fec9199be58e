// K14: the small f32 kernels of the fine-tune heads (model.py:128-143 SelfAttention, :165-218 SequenceClassification,
// :220-232 Excitation, :236-272 TokenClassification; finetune.py:121-129 loss). The heads are < 0.1 % of a fine-tune
// step (d x 128, 128 x 4, 4d x 256, 256 x classes on top of the backbone), so they run in exact f32: their matrix
// products go through pb_gemm's f32 MFMA kernel, and what is left is element-wise / per-column / per-row streaming work.
#include "pb_common.h"
#include "pb_api_internal.h"
#include <algorithm>

namespace {

// op: 1 tanh, 2 relu, 3 sigmoid, 4 identity (dropout only), 5 multiply by a second operand (Excitation's x * y)
__device__ __forceinline__ float act_f(int op, float x) {
    switch (op) {
        case 1: return tanhf(x);
        case 2: return fmaxf(x, 0.f);
        case 3: return 1.0f / (1.0f + __expf(-x));
        default: return x;
    }
}
// derivative expressed through the OUTPUT y = act(x) (tanh: 1 - y^2, relu: y > 0, sigmoid: y (1 - y))
__device__ __forceinline__ float act_grad_from_y(int op, float y) {
    switch (op) {
        case 1: return 1.0f - y * y;
        case 2: return y > 0.f ? 1.0f : 0.f;
        case 3: return y * (1.0f - y);
        default: return 1.0f;
    }
}

__global__ __launch_bounds__(256) void eltwise_fwd_kernel(int op, const float* __restrict__ x, const float* __restrict__ x2, float* __restrict__ y,
                                                         long n, uint64_t seed, uint32_t site, float p) {
    DropCfg dc;
    dc.seed_lo = (uint32_t)seed; dc.seed_hi = (uint32_t)(seed >> 32); dc.site = site;
    dc.thresh = p > 0.f ? (uint32_t)fminf(p * 4294967296.0f, 4294967295.0f) : 0u;
    dc.scale = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        f32x4 v = load4(x + 4 * i);
        if (op == 5) v = v * load4(x2 + 4 * i);
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = act_f(op, v[e]);
        }
        store4(y + 4 * i, v * drop_mask4(dc, (uint32_t)i));
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && (n & 3)) {                    // ragged tail (n % 4 elements)
        const f32x4 m = drop_mask4(dc, (uint32_t)n4);
        for (int e = 0; e < (int)(n & 3); ++e) {
            const long i = 4 * n4 + e;
            y[i] = (op == 5 ? x[i] * x2[i] : act_f(op, x[i])) * m[e];
        }
    }
}

// dx = dy * mask * act'(y_pre_dropout). For op 5: dx = dy * x2 and dx2 = dy * x (x passed as `y`).
__global__ __launch_bounds__(256) void eltwise_bwd_kernel(int op, const float* __restrict__ y, const float* __restrict__ x2, const float* __restrict__ dy,
                                                         float* __restrict__ dx, float* __restrict__ dx2, long n, uint64_t seed, uint32_t site, float p) {
    DropCfg dc;
    dc.seed_lo = (uint32_t)seed; dc.seed_hi = (uint32_t)(seed >> 32); dc.site = site;
    dc.thresh = p > 0.f ? (uint32_t)fminf(p * 4294967296.0f, 4294967295.0f) : 0u;
    dc.scale = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    const long n4 = n >> 2;
    if (blockIdx.x == 0 && threadIdx.x == 0 && (n & 3)) {
        const f32x4 m = drop_mask4(dc, (uint32_t)n4);
        for (int e = 0; e < (int)(n & 3); ++e) {
            const long i = 4 * n4 + e;
            const float g = dy[i] * m[e];
            if (op == 5) { dx[i] = g * x2[i]; dx2[i] = g * y[i]; } else dx[i] = g * act_grad_from_y(op, y[i]);
        }
    }
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 g = load4(dy + 4 * i) * drop_mask4(dc, (uint32_t)i);
        const f32x4 yv = load4(y + 4 * i);
        if (op == 5) {
            store4(dx + 4 * i, g * load4(x2 + 4 * i));
            store4(dx2 + 4 * i, g * yv);
        } else {
            f32x4 r;
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = g[e] * act_grad_from_y(op, yv[e]);
            store4(dx + 4 * i, r);
        }
    }
}

// softmax over the SEQUENCE axis of x (B, S, R): one wave per (b, j) column (R is 4 in the reference), F.softmax(dim=1)
__global__ __launch_bounds__(64) void softmax_dim1_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int S, int R) {
    const int b = blockIdx.x / R, j = blockIdx.x % R, lane = threadIdx.x;
    const float* xb = x + (long)b * S * R + j;
    float* yb = y + (long)b * S * R + j;
    float mx = -INFINITY;
    for (int s = lane; s < S; s += 64) mx = fmaxf(mx, xb[(long)s * R]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int s = lane; s < S; s += 64) sum += __expf(xb[(long)s * R] - mx);
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int s = lane; s < S; s += 64) yb[(long)s * R] = __expf(xb[(long)s * R] - mx) * inv;
}
// dx = y * (dy - sum_s y dy)
__global__ __launch_bounds__(64) void softmax_dim1_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dx, int S, int R) {
    const int b = blockIdx.x / R, j = blockIdx.x % R, lane = threadIdx.x;
    const long base = (long)b * S * R + j;
    float dot = 0.f;
    for (int s = lane; s < S; s += 64) dot = fmaf(y[base + (long)s * R], dy[base + (long)s * R], dot);
    dot = wave_sum(dot);
    for (int s = lane; s < S; s += 64) dx[base + (long)s * R] = y[base + (long)s * R] * (dy[base + (long)s * R] - dot);
}

// nn.CrossEntropyLoss(reduction='none') on rows of C <= 1024 classes + its gradient scaled by coef[0] * weight[row]:
// loss[row] = logsumexp - logit[target]; dlogits = (softmax - onehot) * w. One wave per row. argmax (first maximum) too.
__global__ __launch_bounds__(256) void ce_rows_kernel(const float* __restrict__ logits, const int32_t* __restrict__ target, const float* __restrict__ weight,
                                                     const float* __restrict__ coef, float* __restrict__ loss, float* __restrict__ dlogits,
                                                     int32_t* __restrict__ argmax, long rows, int C) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* lr = logits + row * C;
    float mx = -INFINITY; int am = 0x7fffffff;
    for (int c = lane; c < C; c += 64) { const float v = lr[c]; if (v > mx) { mx = v; am = c; } }
    const float wm = wave_max(mx);
    int cand = (mx == wm) ? am : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
    float sum = 0.f;
    for (int c = lane; c < C; c += 64) sum += __expf(lr[c] - wm);
    sum = wave_sum(sum);
    const int t = target[row];
    const float w = (weight ? weight[row] : 1.0f);
    if (lane == 0) {
        loss[row] = (t >= 0 && t < C) ? (wm + __logf(sum) - lr[t]) : 0.f;
        if (argmax) argmax[row] = cand;
    }
    if (dlogits) {
        const float k = (coef ? coef[0] : 1.0f) * w / sum;
        const float kt = (coef ? coef[0] : 1.0f) * w;
        for (int c = lane; c < C; c += 64) dlogits[row * C + c] = __expf(lr[c] - wm) * k - (c == t ? kt : 0.f);
    }
}

// dropout of an activation tensor in storage dtype (the decoder label-embedding path: BART drops AFTER layernorm_embedding);
// the same call with the same (seed, site) applied to the incoming gradient is its backward.
template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, long n4, uint64_t seed, uint32_t site, float p) {
    DropCfg dc;
    dc.seed_lo = (uint32_t)seed; dc.seed_hi = (uint32_t)(seed >> 32); dc.site = site;
    dc.thresh = p > 0.f ? (uint32_t)fminf(p * 4294967296.0f, 4294967295.0f) : 0u;
    dc.scale = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) store4(y + 4 * i, load4(x + 4 * i) * drop_mask4(dc, (uint32_t)i));
}

// out[t][:] = table[ids[t]][:] + bias (row gather of a small projected label table, model.py:242-245 + PianoBart.py:65-66,71)
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ table, const int32_t* __restrict__ ids, const float* __restrict__ bias,
                                                         float* __restrict__ out, long T, int d, int nrows) {
    const int d4 = d >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < T * d4; i += (long)gridDim.x * 256) {
        const long t = i / d4; const int c = (int)(i - t * d4) * 4;
        const int r = min(max(ids[t], 0), nrows - 1);
        f32x4 v = load4(table + (long)r * d + c);
        if (bias) v += load4(bias + c);
        store4(out + t * d + c, v);
    }
}
// dtable[r][c] = sum over t with ids[t] == r of dout[t][c]: one thread per (r, c), sequential over t -- deterministic; the table
// has a handful of rows (class labels), so T * nrows id reads are nothing.
__global__ __launch_bounds__(256) void gather_rows_bwd_kernel(const float* __restrict__ dout, const int32_t* __restrict__ ids, float* __restrict__ dtable,
                                                             long T, int d, int nrows) {
    const int c = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
    if (c >= d) return;
    float acc = 0.f;
    for (long t = 0; t < T; ++t) if (min(max(ids[t], 0), nrows - 1) == r) acc += dout[t * d + c];
    dtable[(long)r * d + c] = acc;
}

}  // namespace

extern "C" int pb_eltwise_fwd(int32_t op, const float* x, const float* x2, float* y, int64_t n, uint64_t seed, uint32_t site, float p_drop, void* stream_) {
    PB_REQUIRE(op >= 1 && op <= 5 && n >= 0 && (op != 5 || x2), "pb_eltwise_fwd: op=%d n=%ld", op, (long)n);
    if (n == 0) return 0;
    const int grid = (int)std::max<long>(1, std::min<long>(4096, (n / 4 + 255) / 256));
    hipLaunchKernelGGL(eltwise_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream_, op, x, x2, y, (long)n, seed, site, p_drop);
    PB_LAUNCH_CHECK();
    return 0;
}

extern "C" int pb_eltwise_bwd(int32_t op, const float* y, const float* x2, const float* dy, float* dx, float* dx2, int64_t n, uint64_t seed, uint32_t site,
                              float p_drop, void* stream_) {
    PB_REQUIRE(op >= 1 && op <= 5 && n >= 0 && (op != 5 || (x2 && dx2)), "pb_eltwise_bwd: op=%d n=%ld", op, (long)n);
    if (n == 0) return 0;
    const int grid = (int)std::max<long>(1, std::min<long>(4096, (n / 4 + 255) / 256));
    hipLaunchKernelGGL(eltwise_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream_, op, y, x2, dy, dx, dx2, (long)n, seed, site, p_drop);
    PB_LAUNCH_CHECK();
    return 0;
}

extern "C" int pb_softmax_dim1_fwd(const float* x, float* y, int32_t B, int32_t S, int32_t R, void* stream_) {
    PB_REQUIRE(B > 0 && S > 0 && R > 0, "pb_softmax_dim1_fwd: B=%d S=%d R=%d", B, S, R);
    hipLaunchKernelGGL(softmax_dim1_fwd_kernel, dim3(B * R), dim3(64), 0, (hipStream_t)stream_, x, y, S, R);
    PB_LAUNCH_CHECK();
    return 0;
}

extern "C" int pb_softmax_dim1_bwd(const float* y, const float* dy, float* dx, int32_t B, int32_t S, int32_t R, void* stream_) {
    PB_REQUIRE(B > 0 && S > 0 && R > 0, "pb_softmax_dim1_bwd: B=%d S=%d R=%d", B, S, R);
    hipLaunchKernelGGL(softmax_dim1_bwd_kernel, dim3(B * R), dim3(64), 0, (hipStream_t)stream_, y, dy, dx, S, R);
    PB_LAUNCH_CHECK();
    return 0;
}

extern "C" int pb_ce_rows(const float* logits, const int32_t* target, const float* weight, const float* coef, float* loss, float* dlogits,
                          int32_t* argmax, int64_t rows, int32_t C, void* stream_) {
    PB_REQUIRE(rows >= 0 && C > 0 && loss, "pb_ce_rows: rows=%ld C=%d", (long)rows, C);
    if (rows == 0) return 0;
    hipLaunchKernelGGL(ce_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream_, logits, target, weight, coef, loss, dlogits, argmax, rows, C);
    PB_LAUNCH_CHECK();
    return 0;
}

extern "C" int pb_dropout(const void* x, void* y, int64_t n, int32_t dtype, uint64_t seed, uint32_t site, float p_drop, void* stream_) {
    PB_REQUIRE(n >= 0 && n % 4 == 0, "pb_dropout: n=%ld must be a multiple of 4", (long)n);
    if (n == 0) return 0;
    const long n4 = n / 4;
    const int grid = (int)std::min<long>(4096, (n4 + 255) / 256);
    if (dtype == PB_BF16) hipLaunchKernelGGL((dropout_kernel<bf16_t>), dim3(grid), dim3(256), 0, (hipStream_t)stream_, (const bf16_t*)x, (bf16_t*)y, n4, seed, site, p_drop);
    else hipLaunchKernelGGL((dropout_kernel<float>), dim3(grid), dim3(256), 0, (hipStream_t)stream_, (const float*)x, (float*)y, n4, seed, site, p_drop);
    PB_LAUNCH_CHECK();
    return 0;
}

extern "C" int pb_gather_rows(const float* table, const int32_t* ids, const float* bias, float* out, int64_t T, int32_t d, int32_t nrows, void* stream_) {
    PB_REQUIRE(T >= 0 && d > 0 && d % 4 == 0 && nrows > 0, "pb_gather_rows: T=%ld d=%d nrows=%d", (long)T, d, nrows);
    if (T == 0) return 0;
    const int grid = (int)std::min<long>(4096, (T * (d / 4) + 255) / 256);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream_, table, ids, bias, out, (long)T, d, nrows);
    PB_LAUNCH_CHECK();
    return 0;
}

extern "C" int pb_gather_rows_bwd(const float* dout, const int32_t* ids, float* dtable, int64_t T, int32_t d, int32_t nrows, void* stream_) {
    PB_REQUIRE(T >= 0 && d > 0 && nrows > 0 && nrows <= 65535, "pb_gather_rows_bwd: T=%ld d=%d nrows=%d", (long)T, d, nrows);
    hipLaunchKernelGGL(gather_rows_bwd_kernel, dim3((d + 255) / 256, nrows), dim3(256), 0, (hipStream_t)stream_, dout, ids, dtable, (long)T, d, nrows);
    PB_LAUNCH_CHECK();
    return 0;
}
