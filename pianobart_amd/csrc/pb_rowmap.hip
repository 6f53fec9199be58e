// Dead-row compaction ("packed rows") of a pre-train batch.
//
// The reference pads every sequence to S = 1024 tokens (Data/data_generation/convert.py:560-565) and its corruption shortens the
// encoder side further (pretrain.py:332-430: deletion / infilling leave a PAD tail), yet every PAD row still runs through all
// layers (PianoBart.py:60-75: attention_mask only hides the rows as KEYS). A row is dead when nothing reads what is computed for it:
//   encoder row: not visible as a key (encoder_attention_mask == 0); its output is read by nobody
//   decoder row: not visible as a key (decoder_attention_mask == 0) AND no loss term (loss_mask row == 0)
// Dead rows have exactly zero gradient and contribute nothing to any live row, so dropping them changes no result.
//
// Packed layout of one side: batch b owns rows off[b] .. off[b] + len[b] - 1, holding its visible positions (ascending), then its
// invisible positions that carry a loss term, then as many dead positions as the host asked for (fillers that round the total up
// to the GEMM tile; being ordinary dead rows they need no special casing anywhere). The visible rows are a prefix of the batch's
// rows, which is what the packed attention kernels need (keys < k_vis[b] are the visible ones); for the causal decoder the
// visible positions must in addition be the prefix 0 .. L-1 of the sequence, so that packed index == position there
// (pb_rowmap_count reports whether that holds; the caller stays dense otherwise).
#include "pb_api_internal.h"
#include "pb_common.h"

namespace {

__device__ __forceinline__ bool row_has_loss(const float* __restrict__ loss_mask, long row) {
    if (!loss_mask) return false;
    const f32x4 a = load4(loss_mask + row * 8), b = load4(loss_mask + row * 8 + 4);
    return a[0] != 0.f || a[1] != 0.f || a[2] != 0.f || a[3] != 0.f || b[0] != 0.f || b[1] != 0.f || b[2] != 0.f || b[3] != 0.f;
}

__global__ __launch_bounds__(256) void rowmap_count_kernel(const float* __restrict__ emask, const float* __restrict__ dmask,
                                                           const float* __restrict__ loss_mask, int* __restrict__ counts, int S) {
    const int b = blockIdx.x, t = threadIdx.x;
    int ev = 0, dv = 0, last = 0, live = 0, nloss = 0;
    for (int s = t; s < S; s += 256) {
        const long row = (long)b * S + s;
        ev += emask[row] != 0.f;
        const bool v = dmask[row] != 0.f, hl = row_has_loss(loss_mask, row);
        dv += v;
        if (v) last = max(last, s + 1);
        live += v || hl;
        nloss += hl;
    }
    __shared__ int red[5][256];
    red[0][t] = ev; red[1][t] = dv; red[2][t] = last; red[3][t] = live; red[4][t] = nloss;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) {
            red[0][t] += red[0][t + o]; red[1][t] += red[1][t + o]; red[2][t] = max(red[2][t], red[2][t + o]); red[3][t] += red[3][t + o];
            red[4][t] += red[4][t + o];
        }
        __syncthreads();
    }
    if (t == 0) {
        counts[8 * b + 0] = red[0][0];
        counts[8 * b + 1] = red[1][0];
        counts[8 * b + 2] = red[3][0];
        counts[8 * b + 3] = red[1][0] == red[2][0];           // the visible decoder positions are exactly 0 .. L-1
        counts[8 * b + 4] = red[4][0];                        // decoder rows that carry a loss term
        counts[8 * b + 5] = counts[8 * b + 6] = counts[8 * b + 7] = 0;
    }
}

__global__ __launch_bounds__(256) void rowmap_build_kernel(const float* __restrict__ mask, const float* __restrict__ loss_mask,
                                                           const int* __restrict__ off, const int* __restrict__ len,
                                                           int* __restrict__ row_src, int* __restrict__ row_pos, int* __restrict__ inv, int S,
                                                           const int* __restrict__ present) {
    const int b = blockIdx.x, t = threadIdx.x;
    const int per = (S + 255) / 256, s0 = min(S, t * per), s1 = min(S, s0 + per);
    // present == NULL: visible / loss-only / dead. present != NULL (subset of an existing packing, mask unused): rows with a loss
    // term, then other rows of that packing (present[row] >= 0), never the rest; row_pos then receives present[row].
    auto cls = [&](int s) {
        const long row = (long)b * S + s;
        if (present) return row_has_loss(loss_mask, row) ? 0 : (present[row] >= 0 ? 1 : 2);
        return mask[row] != 0.f ? 0 : (row_has_loss(loss_mask, row) ? 1 : 2);
    };
    int c[3] = {0, 0, 0};
    for (int s = s0; s < s1; ++s) {
        const int k = cls(s);
        c[0] += k == 0; c[1] += k == 1; c[2] += k == 2;
    }
    __shared__ int base[3][257];
    for (int k = 0; k < 3; ++k) base[k][t + 1] = c[k];
    __syncthreads();
    if (t < 3) {
        base[t][0] = 0;
        for (int i = 0; i < 256; ++i) base[t][i + 1] += base[t][i];
    }
    __syncthreads();
    const int n0 = base[0][256], n1 = base[1][256], L = len[b], o = off[b];
    int r[3] = {base[0][t], n0 + base[1][t], n0 + n1 + base[2][t]};
    for (int s = s0; s < s1; ++s) {
        const int idx = r[cls(s)]++;
        const long row = (long)b * S + s;
        if (idx < L) {
            row_src[o + idx] = (int)row;
            row_pos[o + idx] = present ? present[row] : s;
            if (inv) inv[row] = o + idx;
        } else if (inv) {
            inv[row] = -1;
        }
    }
}

__global__ __launch_bounds__(256) void gather_rows16_kernel(const uint4* __restrict__ src, const int* __restrict__ row_src, uint4* __restrict__ dst,
                                                            long n, int q) {      // q = 16-byte words per row
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n * q; i += (long)gridDim.x * 256) {
        const long r = i / q;
        const int w = (int)(i - r * q);
        dst[i] = src[(long)row_src[r] * q + w];
    }
}

__global__ __launch_bounds__(256) void scatter_rows16_kernel(const uint4* __restrict__ src, const int* __restrict__ row_dst, uint4* __restrict__ dst,
                                                             long n, int q) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n * q; i += (long)gridDim.x * 256) {
        const long r = i / q;
        const int w = (int)(i - r * q);
        dst[(long)row_dst[r] * q + w] = src[i];
    }
}

// out[s][c] += sum over the batch rows b that kept position s of x[inv[b][s]][c]   (position-table gradient from packed rows)
template <typename T>
__global__ __launch_bounds__(256) void pos_grad_packed_kernel(const T* __restrict__ x, const int* __restrict__ inv, float* __restrict__ out, int B, int S, int d) {
    const int d4 = d >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)S * d4; i += (long)gridDim.x * 256) {
        const int s = (int)(i / d4), c4 = (int)(i - (long)s * d4);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int b = 0; b < B; ++b) {
            const int r = inv[(long)b * S + s];
            if (r >= 0) acc += load4(x + (long)r * d + 4 * c4);
        }
        store4(out + (long)s * d + 4 * c4, load4(out + (long)s * d + 4 * c4) + acc);
    }
}

}  // namespace

extern "C" int pb_rowmap_count(const float* emask, const float* dmask, const float* loss_mask, int32_t* counts, int32_t B, int32_t S, void* stream_) {
    PB_REQUIRE(emask && dmask && counts, "pb_rowmap_count: emask, dmask and counts are required");
    if (B <= 0 || S <= 0) return 0;
    hipLaunchKernelGGL(rowmap_count_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream_, emask, dmask, loss_mask, counts, S);
    PB_LAUNCH_CHECK();
    return 0;
}

extern "C" int pb_rowmap_build(const float* mask, const float* loss_mask, const int32_t* off, const int32_t* len, int32_t* row_src, int32_t* row_pos,
                               int32_t* inv, int32_t B, int32_t S, void* stream_) {
    PB_REQUIRE(mask && off && len && row_src && row_pos && inv, "pb_rowmap_build: NULL argument");
    if (B <= 0 || S <= 0) return 0;
    hipLaunchKernelGGL(rowmap_build_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream_, mask, loss_mask, off, len, row_src, row_pos, inv, S,
                       (const int*)nullptr);
    PB_LAUNCH_CHECK();
    return 0;
}

extern "C" int pb_rowmap_build_sub(const float* loss_mask, const int32_t* present, const int32_t* off, const int32_t* len, int32_t* row_src,
                                   int32_t* row_idx, int32_t B, int32_t S, void* stream_) {
    PB_REQUIRE(loss_mask && present && off && len && row_src && row_idx, "pb_rowmap_build_sub: NULL argument");
    if (B <= 0 || S <= 0) return 0;
    hipLaunchKernelGGL(rowmap_build_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream_, (const float*)nullptr, loss_mask, off, len, row_src, row_idx,
                       (int*)nullptr, S, present);
    PB_LAUNCH_CHECK();
    return 0;
}

extern "C" int pb_scatter_rows16(const void* src, const int32_t* row_dst, void* dst, int64_t n_rows, int32_t row_bytes, void* stream_) {
    PB_REQUIRE(row_bytes > 0 && row_bytes % 16 == 0, "pb_scatter_rows16: row_bytes=%d is not a multiple of 16", row_bytes);
    if (n_rows <= 0) return 0;
    const int q = row_bytes / 16;
    const int grid = (int)min((long)2048, (long)((n_rows * q + 255) / 256));
    hipLaunchKernelGGL(scatter_rows16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream_, (const uint4*)src, row_dst, (uint4*)dst, (long)n_rows, q);
    PB_LAUNCH_CHECK();
    return 0;
}

extern "C" int pb_gather_rows16(const void* src, const int32_t* row_src, void* dst, int64_t n_rows, int32_t row_bytes, void* stream_) {
    PB_REQUIRE(row_bytes > 0 && row_bytes % 16 == 0, "pb_gather_rows16: row_bytes=%d is not a multiple of 16", row_bytes);
    if (n_rows <= 0) return 0;
    const int q = row_bytes / 16;
    const int grid = (int)min((long)2048, (long)((n_rows * q + 255) / 256));
    hipLaunchKernelGGL(gather_rows16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream_, (const uint4*)src, row_src, (uint4*)dst, (long)n_rows, q);
    PB_LAUNCH_CHECK();
    return 0;
}

extern "C" int pb_pos_grad_packed(const void* x, const int32_t* inv, float* out, int32_t B, int32_t S, int32_t d, int32_t dtype, void* stream_) {
    PB_REQUIRE(d % 4 == 0, "pb_pos_grad_packed: d must be a multiple of 4");
    if (B <= 0 || S <= 0) return 0;
    const int grid = (int)min((long)2048, (long)(((long)S * (d / 4) + 255) / 256));
    if (dtype == PB_BF16) hipLaunchKernelGGL((pos_grad_packed_kernel<bf16_t>), dim3(grid), dim3(256), 0, (hipStream_t)stream_, (const bf16_t*)x, inv, out, B, S, d);
    else hipLaunchKernelGGL((pos_grad_packed_kernel<float>), dim3(grid), dim3(256), 0, (hipStream_t)stream_, (const float*)x, inv, out, B, S, d);
    PB_LAUNCH_CHECK();
    return 0;
}
