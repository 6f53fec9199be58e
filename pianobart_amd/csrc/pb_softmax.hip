// Masked softmax over the key axis for the UNFUSED attention form (exact-f32 parity path and the
// fallback for head sizes the flash kernel does not cover). One wave64 per (b,h,query) row.
// Visibility rule (oracle header / transformers 5.x SDPA): key j is visible to query i iff
// (key_mask[b][j] != 0 or key_mask == NULL) and (j <= i or !causal); a row with no visible key
// yields all zeros.
#include "pb_common.h"
#include "pb_api_internal.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ scores, const float* __restrict__ key_mask,
        T* __restrict__ P, long rows, int H, int Sq, int Sk, float scale, int causal) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
        const int i = (int)(row % Sq);
        const long b = row / ((long)H * Sq);
        const float* s = scores + row * Sk;
        const float* km = key_mask ? key_mask + b * Sk : nullptr;
        const int jend = causal ? min(Sk, i + 1) : Sk;
        float mx = -INFINITY;
        for (int j = lane; j < jend; j += 64)
            if (!km || km[j] != 0.f) mx = fmaxf(mx, s[j] * scale);
        mx = wave_max(mx);
        float sum = 0.f;
        if (mx != -INFINITY)
            for (int j = lane; j < jend; j += 64)
                if (!km || km[j] != 0.f) sum += __expf(s[j] * scale - mx);
        sum = wave_sum(sum);
        const float inv = sum > 0.f ? 1.0f / sum : 0.f;
        T* p = P + row * Sk;
        for (int j = lane; j < Sk; j += 64) {
            const bool vis = j < jend && (!km || km[j] != 0.f) && mx != -INFINITY;
            p[j] = from_f<T>(vis ? __expf(s[j] * scale - mx) * inv : 0.f);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ dP, const T* __restrict__ P, T* __restrict__ dS,
                                                          long rows, int Sk, float scale) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
        const float* dp = dP + row * Sk;
        const T* p = P + row * Sk;
        float dot = 0.f;
        for (int j = lane; j < Sk; j += 64) dot += dp[j] * to_f(p[j]);
        dot = wave_sum(dot);
        T* ds = dS + row * Sk;
        for (int j = lane; j < Sk; j += 64) ds[j] = from_f<T>(scale * to_f(p[j]) * (dp[j] - dot));
    }
}

}  // namespace

extern "C" int pb_softmax_fwd(const float* scores, const float* key_mask, void* P, int32_t B, int32_t H, int32_t Sq, int32_t Sk,
                              float scale, int32_t causal, int32_t dtype, void* stream_) {
    const long rows = (long)B * H * Sq;
    if (rows <= 0 || Sk <= 0) return 0;
    const int grid = (int)min((long)4096, (rows + 3) / 4);
    if (dtype == PB_BF16)
        hipLaunchKernelGGL((softmax_fwd_kernel<bf16_t>), dim3(grid), dim3(256), 0, (hipStream_t)stream_, scores, key_mask, (bf16_t*)P, rows, H, Sq, Sk, scale, causal);
    else
        hipLaunchKernelGGL((softmax_fwd_kernel<float>), dim3(grid), dim3(256), 0, (hipStream_t)stream_, scores, key_mask, (float*)P, rows, H, Sq, Sk, scale, causal);
    PB_LAUNCH_CHECK();
    return 0;
}

extern "C" int pb_softmax_bwd(const float* dP, const void* P, void* dS, int64_t rows, int32_t Sk, float scale, int32_t dtype, void* stream_) {
    if (rows <= 0 || Sk <= 0) return 0;
    const int grid = (int)min((long)4096, (long)((rows + 3) / 4));
    if (dtype == PB_BF16)
        hipLaunchKernelGGL((softmax_bwd_kernel<bf16_t>), dim3(grid), dim3(256), 0, (hipStream_t)stream_, dP, (const bf16_t*)P, (bf16_t*)dS, (long)rows, Sk, scale);
    else
        hipLaunchKernelGGL((softmax_bwd_kernel<float>), dim3(grid), dim3(256), 0, (hipStream_t)stream_, dP, (const float*)P, (float*)dS, (long)rows, Sk, scale);
    PB_LAUNCH_CHECK();
    return 0;
}
