// K13: batch-1 KV-cached decode (PianoBartLM.forward(generate=True), /root/reference/model.py:28-66).
// The reference re-runs the full encoder + decoder over all S positions for every generated position; here one decoder
// token goes through the layers against cached keys/values. At batch 1 every op is a weight-streaming GEMV (HBM-bound:
// ~203 MB of bf16 decoder weights per token at cfg 2) or a tiny row op, so the kernels are:
//   * gemv_kernel    y = act(W x + b): 2 output rows per workgroup, K split over its 4 waves, 16-byte weight loads straight
//                    to VGPRs (no LDS: the operand is streamed once and not shared, cdna_hip_programming.md "GEMV" row);
//   * attn_decode    single-query attention, one 16-wave workgroup per head, coalesced row-chunk loads of the cached K/V
//                    (the stand-alone pb_attn_decode op); inside pb_decode_step the keys of a head are split over up to 16
//                    workgroups (attn_split_kernel) whose partial {max, sum, output} records the out-projection GEMV merges
//                    in its prologue -- no merge launch, no inter-workgroup hand-off;
//   * pb_decode_step a native host function that issues the 8*ND + 2 launches of one token (embed -> ND x [q|k|v (+LN2 of the
//                    layer below), self-attn, out, q_c (+LN1), cross-attn, out_c, fc1+GELU (+LNc), fc2] -> heads (+LN2))
//                    without Python between; the post-LNs ride in the prologue of the GEMV that consumes them.
#include "pb_common.h"
#include "pb_api_internal.h"

namespace {

// ---------------------------------------------------------------- GEMV: y[n] = act(sum_k W[n][k] x'[k] + b[n])
// One workgroup = 4 waves = 2 output rows; the 4 waves split K (16-byte loads, each weight byte read once), partial sums meet in
// LDS. N/2 workgroups keep every CU streaming even at N = 768. Rows n >= n_split go to y2 (the K|V cache row) instead of y.
// With `res` set the input is the post-LN residual row x' = LayerNorm(res + x) * gamma + beta, recomputed by every workgroup
// (d reads from L2, two block reductions) so that the BART post-LN needs no launch of its own; workgroup 0 also stores x'
// to ln_out, where the next residual add finds it.
__device__ __forceinline__ float block_sum4(float v, float* red, int lane, int wave) {
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// With `part` set the input vector is the attention context that attn_split_kernel left as per-key-split partials
// part[h][s] = {m, l, -, -, o[hd]} (f32): every workgroup merges the splits of the heads its K range touches on the fly
// (ctx = sum_s o_s e^(m_s - M) / sum_s l_s e^(m_s - M), rounded to T like the stored context of the one-kernel form), so the
// split needs neither a merge launch nor any inter-workgroup hand-off inside the attention kernel.
struct MergeIn { const float* part; int nsplit, hd, stride; };       // stride = floats per (head, split) record

// NCH = chunks of EPV elements a thread owns along K (K <= NCH * 256 * EPV). Everything a thread will ever read -- its weight chunks
// of both rows, x, the residual, gamma / beta, the split records' maxima -- is requested up front, so the kernel is ONE memory round
// trip deep (plus the two block reductions of the LayerNorm statistics): at batch 1 these kernels are latency, not bandwidth.
template <typename T> struct VecOf;
template <> struct VecOf<bf16_t> { typedef bf16x8 type; };
template <> struct VecOf<float> { typedef f32x4 type; };

template <typename T, typename TO, int NCH, bool MERGE>
__global__ __launch_bounds__(256) void gemv_kernel(const T* __restrict__ W, const T* __restrict__ x, const float* __restrict__ bias,
                                                   TO* __restrict__ y, TO* __restrict__ y2, int n_split, int N, int K, int gelu,
                                                   const T* __restrict__ res, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, T* __restrict__ ln_out, float eps, const MergeIn mg) {
    constexpr int EPV = 16 / sizeof(T);
    __shared__ float red[4][2];
    __shared__ float red1[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = blockIdx.x * 2;
    const bool two = n0 + 1 < N;
    const T* w0 = W + (long)n0 * K;
    const T* w1 = W + (long)(two ? n0 + 1 : n0) * K;
    typedef typename VecOf<T>::type V;                                     // EPV elements = 16 bytes, kept as a register vector (no address taken)
    const V zero4 = V{};
    const float bias_v = (threadIdx.x < 2 && (threadIdx.x == 0 || two) && bias) ? bias[n0 + threadIdx.x] : 0.f;   // requested with everything else
    V u0[NCH], u1[NCH], xr[NCH], rr[NCH];
    f32x4 gm[NCH][EPV / 4], bt[NCH][EPV / 4];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = (threadIdx.x + 256 * i) * EPV;
        const bool in = c < K;
        u0[i] = in ? *reinterpret_cast<const V*>(w0 + c) : zero4;
        u1[i] = in ? *reinterpret_cast<const V*>(w1 + c) : zero4;
        xr[i] = (in && !MERGE) ? *reinterpret_cast<const V*>(x + c) : zero4;
        rr[i] = (in && res) ? *reinterpret_cast<const V*>(res + c) : zero4;
#pragma unroll
        for (int v4 = 0; v4 < EPV / 4; ++v4) {
            gm[i][v4] = (in && res) ? *reinterpret_cast<const f32x4*>(gamma + c + 4 * v4) : f32x4{0.f, 0.f, 0.f, 0.f};
            bt[i][v4] = (in && res) ? *reinterpret_cast<const f32x4*>(beta + c + 4 * v4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    float xf[NCH][EPV];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = (threadIdx.x + 256 * i) * EPV;
        if (MERGE && c < K) {
            const int h = c / mg.hd, off = c % mg.hd;                      // EPV columns of one head (hd is a multiple of EPV)
            const float* rec = mg.part + (size_t)h * mg.nsplit * mg.stride;
            // all loads of the <= PB_DECODE_MAX_SPLITS records are issued before the first use: one L2 round trip, not one per split
            float ms[PB_DECODE_MAX_SPLITS], ls[PB_DECODE_MAX_SPLITS];
            f32x4 oa[PB_DECODE_MAX_SPLITS][EPV / 4];
#pragma unroll
            for (int sp = 0; sp < PB_DECODE_MAX_SPLITS; ++sp) {
                const float* r = rec + (size_t)(sp < mg.nsplit ? sp : 0) * mg.stride;
                ms[sp] = sp < mg.nsplit ? r[0] : -INFINITY;
                ls[sp] = r[1];
#pragma unroll
                for (int v4 = 0; v4 < EPV / 4; ++v4) oa[sp][v4] = *reinterpret_cast<const f32x4*>(r + 4 + off + 4 * v4);
            }
            float M = -INFINITY;
#pragma unroll
            for (int sp = 0; sp < PB_DECODE_MAX_SPLITS; ++sp) M = fmaxf(M, ms[sp]);
            float L = 0.f, o[EPV];
#pragma unroll
            for (int j = 0; j < EPV; ++j) o[j] = 0.f;
#pragma unroll
            for (int sp = 0; sp < PB_DECODE_MAX_SPLITS; ++sp) {
                const float wgt = (M == -INFINITY || ms[sp] == -INFINITY) ? 0.f : __expf(ms[sp] - M);   // no visible key in the split (or at all): weight 0
                L = fmaf(ls[sp], wgt, L);
#pragma unroll
                for (int j = 0; j < EPV; ++j) o[j] = fmaf(oa[sp][j >> 2][j & 3], wgt, o[j]);
            }
            const float inv = L > 0.f ? 1.0f / L : 0.f;                    // nothing visible -> zero row (oracle header)
#pragma unroll
            for (int j = 0; j < EPV; ++j) xf[i][j] = to_f(from_f<T>(o[j] * inv));     // rounded to T like the stored context of the one-kernel form
        } else {
#pragma unroll
            for (int j = 0; j < EPV; ++j) xf[i][j] = to_f(xr[i][j]);
        }
    }
    if (res) {
        // x' = LayerNorm(res + x) * gamma + beta, two-pass statistics from the registers
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
#pragma unroll
            for (int j = 0; j < EPV; ++j) { xf[i][j] += to_f(rr[i][j]); s += xf[i][j]; }       // lanes beyond K hold zeros
        }
        const float mean = block_sum4(s, red1, lane, wave) / (float)K;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const bool in = (threadIdx.x + 256 * i) * EPV < K;
#pragma unroll
            for (int j = 0; j < EPV; ++j) { const float z = xf[i][j] - mean; q = in ? fmaf(z, z, q) : q; }
        }
        const float rstd = rsqrtf(block_sum4(q, red1, lane, wave) / (float)K + eps);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = (threadIdx.x + 256 * i) * EPV;
            V xo;
#pragma unroll
            for (int j = 0; j < EPV; ++j) {       // rounded to T exactly like the stored LayerNorm output the unfused path would read back
                xo[j] = from_f<T>((xf[i][j] - mean) * rstd * gm[i][j >> 2][j & 3] + bt[i][j >> 2][j & 3]);
                xf[i][j] = to_f(xo[j]);
            }
            if (blockIdx.x == 0 && c < K) *reinterpret_cast<V*>(ln_out + c) = xo;
        }
    }
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
#pragma unroll
        for (int j = 0; j < EPV; ++j) { a0 = fmaf(to_f(u0[i][j]), xf[i][j], a0); a1 = fmaf(to_f(u1[i][j]), xf[i][j], a1); }    // weights beyond K are zeros
    }
    a0 = wave_sum(a0); a1 = wave_sum(a1);
    if (lane == 0) { red[wave][0] = a0; red[wave][1] = a1; }
    __syncthreads();
    if (threadIdx.x < 2 && (threadIdx.x == 0 || two)) {
        const int n = n0 + threadIdx.x;
        float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x] + bias_v;
        if (gelu) v = gelu_f(v);
        if (n < n_split) y[n] = from_f<TO>(v); else y2[n - n_split] = from_f<TO>(v);
    }
}

// ---------------------------------------------------------------- single-query attention over a K/V cache
// One workgroup of 16 waves per head. A key row (hd elements) is CPR = hd*sizeof(T)/16 consecutive 16-byte chunks, one per
// lane, so one wave load covers 64/CPR whole rows as fully used 128/256-byte segments (a row-per-thread layout touched 64
// cache lines per instruction and took 17 us at Sk = 1024). Splitting the keys of a head over several workgroups was
// measured too: the agent-scope release/acquire its last-block merge needs costs an L2 write-back + invalidate per launch
// on this multi-XCD part (20 us), more than the parallelism returns at these sizes.
constexpr int AD_WAVES = 16;

// CPR = lanes per key row (a power of two), CR = 16-byte chunks a row really has (head_dim 96: 12 of 16 bf16 / 24 of 32 f32 lanes
// carry data, the others hold zeros and load nothing).
template <typename T, int CPR, int CR = CPR>
__global__ __launch_bounds__(AD_WAVES * 64) void attn_decode_kernel(const T* __restrict__ q, const T* __restrict__ kc,
                                                                    const T* __restrict__ vc, T* __restrict__ out,
                                                                    const float* __restrict__ key_mask, int Sk, long k_ss, long v_ss,
                                                                    float scale) {
    constexpr int EPV = 16 / sizeof(T), HD = CR * EPV, KPW = 64 / CPR;        // keys per wave-wide load
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sc = reinterpret_cast<float*>(smem);            // [Sk] scores -> probabilities
    float* red = sc + ((Sk + 3) & ~3);                     // [16] reductions, then [16][HD] partial outputs
    const int h = blockIdx.x;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, sub = lane % CPR, grp = lane / CPR;
    const bool live = CR == CPR || sub < CR;
    float qv[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) qv[e] = 0.f;
    if (live) {
        T qq[EPV];
        *reinterpret_cast<uint4*>(qq) = *reinterpret_cast<const uint4*>(q + h * HD + sub * EPV);
#pragma unroll
        for (int e = 0; e < EPV; ++e) qv[e] = to_f(qq[e]) * scale;
    }
    float mx = -INFINITY;
    constexpr int UR = 8, STEP = AD_WAVES * KPW;           // UR row-chunk loads in flight per lane before the first use
    for (int jb = wave * KPW; jb < Sk; jb += UR * STEP) {
        uint4 kraw[UR];
#pragma unroll
        for (int r = 0; r < UR; ++r) {
            const int j = jb + r * STEP + grp;
            kraw[r] = uint4{0u, 0u, 0u, 0u};
            if (j < Sk && live) kraw[r] = *reinterpret_cast<const uint4*>(kc + (long)j * k_ss + h * HD + sub * EPV);
        }
#pragma unroll
        for (int r = 0; r < UR; ++r) {
            const int j = jb + r * STEP + grp;
            float a = 0.f;
            if (j < Sk) {
                const T* kv = reinterpret_cast<const T*>(&kraw[r]);
#pragma unroll
                for (int e = 0; e < EPV; ++e) a = fmaf(to_f(kv[e]), qv[e], a);
            }
#pragma unroll
            for (int o = 1; o < CPR; o <<= 1) a += __shfl_xor(a, o, 64);
            if (j < Sk) {
                const float sv = (!key_mask || key_mask[j] != 0.f) ? a : -INFINITY;
                if (sub == 0) sc[j] = sv;
                mx = fmaxf(mx, sv);
            }
        }
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int w2 = 1; w2 < AD_WAVES; ++w2) mx = fmaxf(mx, red[w2]);
    __syncthreads();
    float sum = 0.f;
    if (mx != -INFINITY)
        for (int j = t; j < Sk; j += AD_WAVES * 64) { const float e = __expf(sc[j] - mx); sc[j] = e; sum += e; }
    sum = wave_sum(sum);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    sum = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < AD_WAVES; ++w2) sum += red[w2];
    const float inv = (mx != -INFINITY && sum > 0.f) ? 1.0f / sum : 0.f;       // nothing visible -> zero row (oracle header)
    __syncthreads();
    // o[c] = sum_j p_j V[j][c]: same row-chunk ownership; a lane keeps the EPV columns of its chunk
    float acc[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) acc[e] = 0.f;
    if (inv > 0.f)
        for (int jb = wave * KPW; jb < Sk; jb += UR * STEP) {
            uint4 vraw[UR];
#pragma unroll
            for (int r = 0; r < UR; ++r) {
                const int j = jb + r * STEP + grp;
                vraw[r] = uint4{0u, 0u, 0u, 0u};
                if (j < Sk && live) vraw[r] = *reinterpret_cast<const uint4*>(vc + (long)j * v_ss + h * HD + sub * EPV);
            }
#pragma unroll
            for (int r = 0; r < UR; ++r) {
                const int j = jb + r * STEP + grp;
                if (j < Sk) {
                    const T* vv = reinterpret_cast<const T*>(&vraw[r]);
                    const float pj = sc[j];
#pragma unroll
                    for (int e = 0; e < EPV; ++e) acc[e] = fmaf(pj, to_f(vv[e]), acc[e]);
                }
            }
        }
#pragma unroll
    for (int e = 0; e < EPV; ++e)
#pragma unroll
        for (int o = CPR; o < 64; o <<= 1) acc[e] += __shfl_xor(acc[e], o, 64);
    if (grp == 0 && live)
#pragma unroll
        for (int e = 0; e < EPV; ++e) red[wave * HD + sub * EPV + e] = acc[e];
    __syncthreads();
    if (t < HD) {
        float o = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < AD_WAVES; ++w2) o += red[w2 * HD + t];
        out[h * HD + t] = from_f<T>(o * inv);
    }
}


// ---------------------------------------------------------------- single-query attention, keys split over workgroups
// The one-workgroup-per-head form above keeps 12 of 256 CUs busy at cfg 2 (10.7 us per call, 24 calls per token = 40 % of the
// decode step's GPU time). Here a (head, key-split) pair is one 4-wave workgroup: it computes the scores of its <= `chunk` keys,
// their local maximum m, l = sum e^(s - m) and o = sum e^(s - m) v (unnormalised) and stores {m, l, -, -, o[hd]} as f32; the consumer
// GEMV merges the splits in its prologue (gemv_kernel / MergeIn). Same row-chunk ownership of the cached K / V rows as above.
constexpr int AS_WAVES = 4;
template <typename T, int CPR, int CR = CPR>
__global__ __launch_bounds__(AS_WAVES * 64) void attn_split_kernel(const T* __restrict__ q, const T* __restrict__ kc, const T* __restrict__ vc,
                                                                   float* __restrict__ part, const float* __restrict__ key_mask, int Sk, int chunk,
                                                                   long k_ss, long v_ss, float scale) {
    constexpr int EPV = 16 / sizeof(T), HD = CR * EPV, KPW = 64 / CPR, STEP = AS_WAVES * KPW, UR = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sc = reinterpret_cast<float*>(smem);            // [chunk] scores -> probabilities
    float* red = sc + ((chunk + 3) & ~3);                  // [AS_WAVES] reductions, then [AS_WAVES][HD] partial outputs
    const int h = blockIdx.x, sp = blockIdx.y, nsplit = gridDim.y;
    const int j0 = sp * chunk, j1 = min(Sk, j0 + chunk);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, sub = lane % CPR, grp = lane / CPR;
    const bool live = CR == CPR || sub < CR;
    float qv[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) qv[e] = 0.f;
    if (live) {
        T qq[EPV];
        *reinterpret_cast<uint4*>(qq) = *reinterpret_cast<const uint4*>(q + h * HD + sub * EPV);
#pragma unroll
        for (int e = 0; e < EPV; ++e) qv[e] = to_f(qq[e]) * scale;
    }
    float mx = -INFINITY;
    const int jfirst = j0 + wave * KPW;
    uint4 vpre[UR];                                        // V rows of the first (usually the only) block, requested together with its K rows
    for (int jb = jfirst; jb < j1; jb += UR * STEP) {
        uint4 kraw[UR];
#pragma unroll
        for (int r = 0; r < UR; ++r) {
            const int j = jb + r * STEP + grp;
            kraw[r] = uint4{0u, 0u, 0u, 0u};
            if (j < j1 && live) kraw[r] = *reinterpret_cast<const uint4*>(kc + (long)j * k_ss + h * HD + sub * EPV);
        }
        if (jb == jfirst) {
#pragma unroll
            for (int r = 0; r < UR; ++r) {
                const int j = jb + r * STEP + grp;
                vpre[r] = uint4{0u, 0u, 0u, 0u};
                if (j < j1 && live) vpre[r] = *reinterpret_cast<const uint4*>(vc + (long)j * v_ss + h * HD + sub * EPV);
            }
        }
#pragma unroll
        for (int r = 0; r < UR; ++r) {
            const int j = jb + r * STEP + grp;
            float a = 0.f;
            if (j < j1) {
                const T* kv = reinterpret_cast<const T*>(&kraw[r]);
#pragma unroll
                for (int e = 0; e < EPV; ++e) a = fmaf(to_f(kv[e]), qv[e], a);
            }
#pragma unroll
            for (int o = 1; o < CPR; o <<= 1) a += __shfl_xor(a, o, 64);
            if (j < j1) {
                const float sv = (!key_mask || key_mask[j] != 0.f) ? a : -INFINITY;
                if (sub == 0) sc[j - j0] = sv;
                mx = fmaxf(mx, sv);
            }
        }
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    if (mx != -INFINITY)
        for (int j = t; j < j1 - j0; j += AS_WAVES * 64) { const float e = __expf(sc[j] - mx); sc[j] = e; sum += e; }
    sum = wave_sum(sum);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    sum = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    float acc[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) acc[e] = 0.f;
    if (mx != -INFINITY)
        for (int jb = jfirst; jb < j1; jb += UR * STEP) {
            uint4 vraw[UR];
#pragma unroll
            for (int r = 0; r < UR; ++r) {
                const int j = jb + r * STEP + grp;
                vraw[r] = vpre[r];
                if (jb != jfirst) {
                    vraw[r] = uint4{0u, 0u, 0u, 0u};
                    if (j < j1 && live) vraw[r] = *reinterpret_cast<const uint4*>(vc + (long)j * v_ss + h * HD + sub * EPV);
                }
            }
#pragma unroll
            for (int r = 0; r < UR; ++r) {
                const int j = jb + r * STEP + grp;
                if (j < j1) {
                    const T* vv = reinterpret_cast<const T*>(&vraw[r]);
                    const float pj = sc[j - j0];
#pragma unroll
                    for (int e = 0; e < EPV; ++e) acc[e] = fmaf(pj, to_f(vv[e]), acc[e]);
                }
            }
        }
#pragma unroll
    for (int e = 0; e < EPV; ++e)
#pragma unroll
        for (int o = CPR; o < 64; o <<= 1) acc[e] += __shfl_xor(acc[e], o, 64);
    if (grp == 0 && live)
#pragma unroll
        for (int e = 0; e < EPV; ++e) red[wave * HD + sub * EPV + e] = acc[e];
    __syncthreads();
    float* rec = part + ((size_t)h * nsplit + sp) * (HD + 4);          // {m, l, -, -, o[HD]}: o starts 16-byte aligned
    if (t == 0) { rec[0] = mx; rec[1] = sum; }
    if (t < HD) rec[4 + t] = (red[t] + red[HD + t]) + (red[2 * HD + t] + red[3 * HD + t]);
}

}  // namespace

struct LnIn { const void* res; const float* gamma; const float* beta; void* out; };

static int gemv_launch(const void* W, const void* x, const float* bias, void* y, void* y2, int n_split, int N, int K, int dtype, int y_f32,
                       int gelu, hipStream_t stream, LnIn ln = LnIn{nullptr, nullptr, nullptr, nullptr}, MergeIn mg = MergeIn{nullptr, 0, 0, 0}) {
    const int epv = dtype == PB_BF16 ? 8 : 4;
    PB_REQUIRE(N > 0 && K > 0 && K % epv == 0, "pb_gemv: K=%d must be a multiple of %d", K, epv);
    PB_REQUIRE(((uintptr_t)W % 16 == 0) && ((uintptr_t)x % 16 == 0), "pb_gemv: operands must be 16-byte aligned");
    dim3 grid((N + 1) / 2), block(256);
    const float eps = 1e-5f;
    const int nch = (K + 256 * epv - 1) / (256 * epv);
    PB_REQUIRE(nch <= 4, "pb_gemv: K=%d exceeds %d", K, 4 * 256 * epv);
#define PB_GEMV_GO(TT, TO, NCH_, MG_)                                                                                                      \
    hipLaunchKernelGGL((gemv_kernel<TT, TO, NCH_, MG_>), grid, block, 0, stream, (const TT*)W, (const TT*)x, bias, (TO*)y, (TO*)y2, n_split, N, K, \
                       gelu, (const TT*)ln.res, ln.gamma, ln.beta, (TT*)ln.out, eps, mg)
#define PB_GEMV_NCH(TT, TO)                                                                                   \
    do {                                                                                                      \
        if (mg.part) { PB_REQUIRE(nch <= 1, "pb_gemv: the split-merge prologue needs K <= %d", 256 * epv); PB_GEMV_GO(TT, TO, 1, true); } \
        else if (nch <= 1) PB_GEMV_GO(TT, TO, 1, false);                                                      \
        else if (nch == 2) PB_GEMV_GO(TT, TO, 2, false);                                                      \
        else PB_GEMV_GO(TT, TO, 4, false);                                                                    \
    } while (0)
    if (dtype == PB_BF16) {
        if (y_f32) PB_GEMV_NCH(bf16_t, float); else PB_GEMV_NCH(bf16_t, bf16_t);
    } else {
        PB_GEMV_NCH(float, float);
    }
#undef PB_GEMV_NCH
#undef PB_GEMV_GO
    PB_LAUNCH_CHECK();
    return 0;
}

extern "C" int pb_gemv(const void* W, const void* x, const float* bias, void* y, int32_t N, int32_t K, int32_t dtype, int32_t y_f32,
                       int32_t gelu, void* stream_) {
    return gemv_launch(W, x, bias, y, nullptr, N, N, K, dtype, y_f32, gelu, (hipStream_t)stream_);
}

template <typename T, int CPR, int CR = CPR>
static void attn_decode_launch(const void* q, const void* kc, const void* vc, void* out, const float* key_mask, int H, int Sk, long k_ss,
                               long v_ss, float scale, hipStream_t stream) {
    constexpr int HD = CR * (16 / (int)sizeof(T));
    const size_t lds = (size_t)(((Sk + 3) & ~3) + AD_WAVES * HD) * sizeof(float);
    hipLaunchKernelGGL((attn_decode_kernel<T, CPR, CR>), dim3(H), dim3(AD_WAVES * 64), lds, stream, (const T*)q, (const T*)kc, (const T*)vc, (T*)out,
                       key_mask, Sk, k_ss, v_ss, scale);
}

extern "C" int pb_attn_decode(const void* q, const void* k_cache, const void* v_cache, void* out, const float* key_mask, int32_t H, int32_t Sk,
                              int32_t hd, int64_t k_ss, int64_t v_ss, float scale, int32_t dtype, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    PB_REQUIRE(H > 0 && Sk > 0 && Sk <= 8192 && (hd == 32 || hd == 64 || hd == 96 || hd == 128),
               "pb_attn_decode: H=%d Sk=%d hd=%d (head_dim must be 32, 64, 96 or 128; Sk <= 8192)", H, Sk, hd);
    const int epv = dtype == PB_BF16 ? 8 : 4;
    PB_REQUIRE(k_ss % epv == 0 && v_ss % epv == 0 && ((uintptr_t)k_cache % 16 == 0) && ((uintptr_t)v_cache % 16 == 0) && ((uintptr_t)q % 16 == 0),
               "pb_attn_decode: rows must be 16-byte aligned");
    if (dtype == PB_BF16) {
        if (hd == 32) attn_decode_launch<bf16_t, 4>(q, k_cache, v_cache, out, key_mask, H, Sk, k_ss, v_ss, scale, stream);
        else if (hd == 64) attn_decode_launch<bf16_t, 8>(q, k_cache, v_cache, out, key_mask, H, Sk, k_ss, v_ss, scale, stream);
        else if (hd == 96) attn_decode_launch<bf16_t, 16, 12>(q, k_cache, v_cache, out, key_mask, H, Sk, k_ss, v_ss, scale, stream);
        else attn_decode_launch<bf16_t, 16>(q, k_cache, v_cache, out, key_mask, H, Sk, k_ss, v_ss, scale, stream);
    } else {
        if (hd == 32) attn_decode_launch<float, 8>(q, k_cache, v_cache, out, key_mask, H, Sk, k_ss, v_ss, scale, stream);
        else if (hd == 64) attn_decode_launch<float, 16>(q, k_cache, v_cache, out, key_mask, H, Sk, k_ss, v_ss, scale, stream);
        else if (hd == 96) attn_decode_launch<float, 32, 24>(q, k_cache, v_cache, out, key_mask, H, Sk, k_ss, v_ss, scale, stream);
        else attn_decode_launch<float, 32>(q, k_cache, v_cache, out, key_mask, H, Sk, k_ss, v_ss, scale, stream);
    }
    PB_LAUNCH_CHECK();
    return 0;
}

// key-split single-query attention into `part` (H * nsplit records of hd + 4 floats); returns the number of splits used
template <typename T, int CPR, int CR = CPR>
static void attn_split_launch(const void* q, const void* kc, const void* vc, float* part, const float* key_mask, int H, int Sk, int chunk, int nsplit,
                              long k_ss, long v_ss, float scale, hipStream_t stream) {
    constexpr int HD = CR * (16 / (int)sizeof(T));
    const size_t lds = (size_t)(((chunk + 3) & ~3) + AS_WAVES * HD) * sizeof(float);
    hipLaunchKernelGGL((attn_split_kernel<T, CPR, CR>), dim3(H, nsplit), dim3(AS_WAVES * 64), lds, stream, (const T*)q, (const T*)kc, (const T*)vc, part,
                       key_mask, Sk, chunk, k_ss, v_ss, scale);
}
static int attn_split(const void* q, const void* kc, const void* vc, float* part, const float* key_mask, int H, int Sk, int hd, long k_ss, long v_ss,
                      float scale, int dtype, hipStream_t stream, int& nsplit) {
    // <= PB_DECODE_MAX_SPLITS splits of >= 64 keys: H * nsplit workgroups cover the chip at cfg 2 from 512 keys on
    int chunk = (Sk + PB_DECODE_MAX_SPLITS - 1) / PB_DECODE_MAX_SPLITS;
    chunk = chunk < 64 ? 64 : (chunk + 15) & ~15;
    nsplit = (Sk + chunk - 1) / chunk;
    if (dtype == PB_BF16) {
        if (hd == 32) attn_split_launch<bf16_t, 4>(q, kc, vc, part, key_mask, H, Sk, chunk, nsplit, k_ss, v_ss, scale, stream);
        else if (hd == 64) attn_split_launch<bf16_t, 8>(q, kc, vc, part, key_mask, H, Sk, chunk, nsplit, k_ss, v_ss, scale, stream);
        else if (hd == 96) attn_split_launch<bf16_t, 16, 12>(q, kc, vc, part, key_mask, H, Sk, chunk, nsplit, k_ss, v_ss, scale, stream);
        else attn_split_launch<bf16_t, 16>(q, kc, vc, part, key_mask, H, Sk, chunk, nsplit, k_ss, v_ss, scale, stream);
    } else {
        if (hd == 32) attn_split_launch<float, 8>(q, kc, vc, part, key_mask, H, Sk, chunk, nsplit, k_ss, v_ss, scale, stream);
        else if (hd == 64) attn_split_launch<float, 16>(q, kc, vc, part, key_mask, H, Sk, chunk, nsplit, k_ss, v_ss, scale, stream);
        else if (hd == 96) attn_split_launch<float, 32, 24>(q, kc, vc, part, key_mask, H, Sk, chunk, nsplit, k_ss, v_ss, scale, stream);
        else attn_split_launch<float, 32>(q, kc, vc, part, key_mask, H, Sk, chunk, nsplit, k_ss, v_ss, scale, stream);
    }
    PB_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------- one decoder token, natively sequenced
extern "C" int pb_decode_step(const pb_decode_plan* p, int32_t i, void* stream) {
    PB_REQUIRE(p && p->n_layers > 0 && p->n_layers <= PB_DECODE_MAX_LAYERS, "pb_decode_step: bad plan");
    PB_REQUIRE(i >= 0 && i < p->S, "pb_decode_step: step %d outside 0..%d", i, p->S - 1);
    const int d = p->d, H = p->H, hd = d / H, f = p->ffn, dt = p->dtype;
    const size_t esz = dt == PB_BF16 ? 2 : 4;
    const float scale = 1.0f / sqrtf((float)hd);
    const int32_t* seg = p->tab_off;
    char* x = (char*)p->x; char* alt = (char*)p->y2;
    // token embedding + position i + LayerNorm (S = 1 with the position table advanced by i rows)
    if (pb_embed_ln_fwd(p->tok16, p->ptab, seg, p->lin_b, p->pos + (size_t)i * d, p->lne_w, p->lne_b, x, p->stat, p->stat + 1, 1, 1, d, dt,
                        1e-5f, 0, 0, 0.f, stream)) return -1;
    hipStream_t st = (hipStream_t)stream;
    LnIn ln{nullptr, nullptr, nullptr, nullptr};            // pending post-LN of the previous sub-layer, applied by the next GEMV
    char* h = x;
    for (int l = 0; l < p->n_layers; ++l) {
        const pb_decode_layer& L = p->layers[l];
        char* kvs = (char*)L.kv_self;
        // q | k|v in one launch; k|v land in row i of the self-attention cache. Input: h (layer 0) or LN2 of the layer below.
        if (gemv_launch(L.wqkv, ln.res ? (const void*)p->a : (const void*)h, L.bqkv, p->q, kvs + (size_t)i * 2 * d * esz, d, 3 * d, d, dt, 0, 0, st, ln)) return -1;
        if (ln.res) h = alt;
        int ns = 0;
        if (p->attn_part) {                                // keys split over workgroups, merged in the out-projection's prologue
            if (attn_split(p->q, kvs, kvs + (size_t)d * esz, p->attn_part, nullptr, H, i + 1, hd, 2 * d, 2 * d, scale, dt, st, ns)) return -1;
            if (gemv_launch(L.wo, p->ctx, L.bo, p->a, nullptr, d, d, d, dt, 0, 0, st, LnIn{nullptr, nullptr, nullptr, nullptr}, MergeIn{p->attn_part, ns, hd, hd + 4})) return -1;
        } else {
            if (pb_attn_decode(p->q, kvs, kvs + (size_t)d * esz, p->ctx, nullptr, H, i + 1, hd, 2 * d, 2 * d, scale, dt, stream)) return -1;
            if (pb_gemv(L.wo, p->ctx, L.bo, p->a, d, d, dt, 0, 0, stream)) return -1;
        }
        // cross attention against the cached encoder K/V; the q projection applies LN1(h + a) -> y1
        ln = LnIn{h, L.ln1_w, L.ln1_b, p->y1};
        if (gemv_launch(L.wq_c, p->a, L.bq_c, p->q, nullptr, d, d, d, dt, 0, 0, st, ln)) return -1;
        if (p->attn_part) {
            if (attn_split(p->q, L.kv_cross, (const char*)L.kv_cross + (size_t)d * esz, p->attn_part, p->enc_mask, H, p->S_enc, hd, 2 * d, 2 * d, scale, dt, st, ns)) return -1;
            if (gemv_launch(L.wo_c, p->ctx, L.bo_c, p->a, nullptr, d, d, d, dt, 0, 0, st, LnIn{nullptr, nullptr, nullptr, nullptr}, MergeIn{p->attn_part, ns, hd, hd + 4})) return -1;
        } else {
            if (pb_attn_decode(p->q, L.kv_cross, (const char*)L.kv_cross + (size_t)d * esz, p->ctx, p->enc_mask, H, p->S_enc, hd, 2 * d, 2 * d, scale, dt, stream)) return -1;
            if (pb_gemv(L.wo_c, p->ctx, L.bo_c, p->a, d, d, dt, 0, 0, stream)) return -1;
        }
        // FFN; fc1 applies LNc(y1 + a) -> yc
        ln = LnIn{p->y1, L.lnc_w, L.lnc_b, p->yc};
        if (gemv_launch(L.w1, p->a, L.b1, p->g, nullptr, f, f, d, dt, 0, 1, st, ln)) return -1;
        if (pb_gemv(L.w2, p->g, L.b2, p->a, d, f, dt, 0, 0, stream)) return -1;
        ln = LnIn{p->yc, L.ln2_w, L.ln2_b, alt};            // LN2(yc + a) -> next layer's h, applied by its q|k|v GEMV (or the heads)
    }
    return gemv_launch(p->head_w, p->a, p->head_b, p->logits, nullptr, p->vocab, p->vocab, d, dt, 1, 0, st, ln);
}
