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

// =====================================================================================================================
// Decode, second form (round 3): one token = ONE hipGraph replay of 6 launches per decoder layer (+ embed + heads).
// What changed against pb_decode_step above:
//   * the position i lives in DEVICE memory (incremented by the first kernel of a step), so the launches of a step have no
//     position-dependent argument or grid and one captured graph serves every position: the host's ~3.5 us of enqueue per launch
//     (host-bound with kernels this short) become one hipGraphLaunch per token; the token ids go up and the logits row comes down
//     through two copy nodes of the same graph (pinned host buffers owned by the decoder);
//   * the q projection is fused into the single-query attention (dec_attn_kernel): a (head, key split) workgroup needs q of ITS
//     head only (hd rows of W_q, 98 KB at cfg 2: re-read by the <= 16 splits of a head from L2), so it applies the pending post-LN
//     itself, projects q_h, and goes on to its keys -- the q|k|v GEMV launch and its all-to-all seam are gone. For the self-attention
//     one more workgroup per head projects k_h and v_h of the new token as well, writes them to the cache row and contributes
//     the new token's own {score, 1, v} record, so the regular splits only read rows that older launches wrote;
//   * per layer: self-attn(+LN2 of the layer below, q|k|v) -> out-proj (merges the split records) -> cross-attn(+LN1, q_c) ->
//     out_c (merge) -> fc1 (+LNc, GELU) -> fc2. Every remaining boundary is a real all-to-all seam (each output needs the whole
//     input vector, produced by all workgroups of the launch before): cdna_hip_programming.md 5.6 prices a grid barrier above a
//     kernel boundary, so they stay launches.
// bf16, head_dim 64 or 128, d a multiple of 256 up to 1024; anything else keeps pb_decode_step.
namespace {

__device__ __forceinline__ float half_sum(float v) {            // sum over the 32 lanes of a half-wave, in every lane of it
    v += PB_DPP_F(v, 0xb1);
    v += PB_DPP_F(v, 0x4e);
    v += PB_DPP_F(v, 0x141);
    v += PB_DPP_F(v, 0x140);
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(a[0]) + __uint_as_float(a[1]);
}

struct DecAttnArgs {
    const bf16_t* x_in;                                          // the input vector (d) when res == NULL
    const bf16_t* res; const bf16_t* add; const float* gamma; const float* beta; bf16_t* ln_out;   // x' = LN(res + add) gamma + beta
    const bf16_t* Wq; const float* bq;                           // q projection rows [d][d] (+ bias); head h owns rows h HD ..
    const bf16_t* Wk; const float* bk; const bf16_t* Wv; const float* bv;      // SELF: the new token's k / v rows
    bf16_t* kc; bf16_t* vc; long kv_ss;                          // cached rows: kc + j kv_ss + h HD
    const float* key_mask;                                       // cross: [Sk] (0 = masked) or NULL
    int* pos; int Sk_fixed;                                      // SELF: keys cached so far = *pos, row *pos is written; cross: Sk_fixed keys
    int d, nreg, ck_fixed;                                       // regular key splits; cross: keys per split
    float scale, eps;
    float* part;                                                 // [H][gridDim.y][HD + 4] records {m, l, -, -, o[HD]}
};

// Threads: 256; the self-attention form launches 768 so that the new token's workgroup projects q, k and v side by side (wave
// groups 0 / 1 / 2, every weight load of a group in flight at once: ONE memory round trip instead of six dependent ones -- these
// kernels are pure latency); in the other workgroups of that launch waves 4 .. 11 leave at once.
template <int NC, int HD, bool SELF>
__global__ __launch_bounds__(SELF ? 768 : 256) void dec_attn_kernel(const DecAttnArgs a) {
    constexpr int CPR = HD / 8, KPW = 64 / CPR, STEP = 4 * KPW, UR = 4, RPW = HD / 4, NP = RPW / 2;
    constexpr int PBATCH = (NC <= 3 && !SELF) || NC <= 2 ? (NP < 8 ? NP : 8) : 4;       // passes of weight rows in flight per lane (registers)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* qs = reinterpret_cast<float*>(smem);                  // [HD] q of this head, rounded to bf16 like a stored q row, times the softmax scale
    float* red = qs + HD;                                        // [4 HD] reductions / per-wave partial outputs (new-token workgroup: k | v)
    float* sc = red + 4 * HD;                                    // [keys per split] scores -> probabilities
    const int h = blockIdx.x, sp = blockIdx.y, nrec = gridDim.y;
    const int t = threadIdx.x, lane = t & 63, l32 = lane & 31, half = lane >> 5;
    const int wgrp = SELF ? (int)(t >> 8) : 0;                   // 0: q (and the attention), 1: k, 2: v of the new token
    const int wave = (t >> 6) & 3;
    const int d = a.d;
    const int Sk = SELF ? *a.pos : a.Sk_fixed;
    const bool is_new = SELF && sp == a.nreg;
    if (SELF && wgrp > 0 && !is_new) return;                     // only the new token's workgroup uses the other two wave groups
    int ck = a.ck_fixed;
    if (SELF) { ck = (Sk + a.nreg - 1) / a.nreg; ck = ck < 64 ? 64 : (ck + 15) & ~15; }
    const int j0 = sp * ck, j1 = min(Sk, j0 + ck);
    float* rec = a.part + ((size_t)h * nrec + sp) * (HD + 4);
    if (!is_new && j0 >= Sk) {                                   // no key in this split: a record of weight zero
        if (t == 0) { rec[0] = -INFINITY; rec[1] = 0.f; }
        if (t < HD) rec[4 + t] = 0.f;
        return;
    }
    const int tq = t & 255;                                       // thread index inside its wave group
    // cached rows of the first block of this split: requested before anything else (they do not depend on q)
    const int sub = lane % CPR, grp = lane / CPR;
    const int jfirst = j0 + wave * KPW;
    uint4 kpre[UR], vpre[UR];
#pragma unroll
    for (int r = 0; r < UR; ++r) {
        const int j = jfirst + r * STEP + grp;
        kpre[r] = uint4{0u, 0u, 0u, 0u}; vpre[r] = uint4{0u, 0u, 0u, 0u};
        if (!is_new && j < j1) {
            kpre[r] = *reinterpret_cast<const uint4*>(a.kc + (long)j * a.kv_ss + h * HD + sub * 8);
            vpre[r] = *reinterpret_cast<const uint4*>(a.vc + (long)j * a.kv_ss + h * HD + sub * 8);
        }
    }
    // the input vector, whole, in every half-wave: lane l32 holds the 8 elements of chunks l32 + 32 c
    float xf[NC][8];
    if (a.res) {
        bf16x8 rr[NC], aa[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            rr[c] = *reinterpret_cast<const bf16x8*>(a.res + (l32 + 32 * c) * 8);
            aa[c] = *reinterpret_cast<const bf16x8*>(a.add + (l32 + 32 * c) * 8);
        }
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) { xf[c][j] = (float)rr[c][j] + (float)aa[c][j]; s += xf[c][j]; }
        const float mean = half_sum(s) / (float)d;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float z = xf[c][j] - mean; q = fmaf(z, z, q); }
        const float rstd = rsqrtf(half_sum(q) / (float)d + a.eps);
        const bool store_ln = h == 0 && wgrp == 0 && wave == 0 && half == 0 && (SELF ? is_new : sp == 0);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int e0 = (l32 + 32 * c) * 8;
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(a.gamma + e0), g1 = *reinterpret_cast<const f32x4*>(a.gamma + e0 + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.beta + e0), b1 = *reinterpret_cast<const f32x4*>(a.beta + e0 + 4);
            bf16x8 xo;
#pragma unroll
            for (int j = 0; j < 8; ++j) {                        // rounded to bf16 like the stored LayerNorm output the residual path reads back
                xo[j] = (bf16_t)((xf[c][j] - mean) * rstd * (j < 4 ? g0[j & 3] : g1[j & 3]) + (j < 4 ? b0[j & 3] : b1[j & 3]));
                xf[c][j] = (float)xo[j];
            }
            if (store_ln) *reinterpret_cast<bf16x8*>(a.ln_out + e0) = xo;
        }
    } else {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const bf16x8 xv = *reinterpret_cast<const bf16x8*>(a.x_in + (l32 + 32 * c) * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) xf[c][j] = (float)xv[j];
        }
    }
    // HD rows of a projection: a half-wave per row (2 rows per pass and wave), 4 passes of weight loads in flight
    auto project = [&](const bf16_t* __restrict__ W, const float* __restrict__ bias, float* out, float mul) {
#pragma unroll 1
        for (int pb = 0; pb < NP; pb += PBATCH) {
            bf16x8 w[PBATCH][NC];
            float bv[PBATCH];
#pragma unroll
            for (int p = 0; p < PBATCH; ++p) {
                const int row = h * HD + wave * RPW + 2 * (pb + p) + half;
                bv[p] = bias[row];
#pragma unroll
                for (int c = 0; c < NC; ++c) w[p][c] = *reinterpret_cast<const bf16x8*>(W + (size_t)row * d + (l32 + 32 * c) * 8);
            }
#pragma unroll
            for (int p = 0; p < PBATCH; ++p) {
                float acc = 0.f;
#pragma unroll
                for (int c = 0; c < NC; ++c)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc = fmaf((float)w[p][c][j], xf[c][j], acc);
                acc = half_sum(acc);
                if (l32 == 0) out[wave * RPW + 2 * (pb + p) + half] = (float)(bf16_t)(acc + bv[p]) * mul;
            }
        }
    };
    if (is_new) {
        float* ks = red; float* vs = red + HD;
        if (wgrp == 0) project(a.Wq, a.bq, qs, a.scale);
        else if (wgrp == 1) project(a.Wk, a.bk, ks, 1.f);
        else project(a.Wv, a.bv, vs, 1.f);
        __syncthreads();
        if (t < HD) {
            a.kc[(long)Sk * a.kv_ss + h * HD + t] = (bf16_t)ks[t];
            a.vc[(long)Sk * a.kv_ss + h * HD + t] = (bf16_t)vs[t];
            rec[4 + t] = vs[t];
        }
        if (t < 64) {
            float p = 0.f;
#pragma unroll
            for (int e = lane; e < HD; e += 64) p = fmaf(qs[e], ks[e], p);
            p = wave_sum(p);
            if (lane == 0) { rec[0] = p; rec[1] = 1.f; }
        }
        return;
    }
    project(a.Wq, a.bq, qs, a.scale);
    __syncthreads();
    float qv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) qv[e] = qs[sub * 8 + e];
    float mx = -INFINITY;
    for (int jb = jfirst; jb < j1; jb += UR * STEP) {
        uint4 kraw[UR];
#pragma unroll
        for (int r = 0; r < UR; ++r) {
            const int j = jb + r * STEP + grp;
            kraw[r] = kpre[r];
            if (jb != jfirst) {
                kraw[r] = uint4{0u, 0u, 0u, 0u};
                if (j < j1) kraw[r] = *reinterpret_cast<const uint4*>(a.kc + (long)j * a.kv_ss + h * HD + sub * 8);
            }
        }
#pragma unroll
        for (int r = 0; r < UR; ++r) {
            const int j = jb + r * STEP + grp;
            float s = 0.f;
            if (j < j1) {
                const bf16_t* kv = reinterpret_cast<const bf16_t*>(&kraw[r]);
#pragma unroll
                for (int e = 0; e < 8; ++e) s = fmaf((float)kv[e], qv[e], s);
            }
#pragma unroll
            for (int o = 1; o < CPR; o <<= 1) s += __shfl_xor(s, o, 64);
            if (j < j1) {
                const float sv = (!a.key_mask || a.key_mask[j] != 0.f) ? s : -INFINITY;
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
        for (int j = tq; j < j1 - j0; j += 256) { const float e = __expf(sc[j] - mx); sc[j] = e; sum += e; }
    sum = wave_sum(sum);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    sum = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    if (mx != -INFINITY)
        for (int jb = jfirst; jb < j1; jb += UR * STEP) {
            uint4 vraw[UR];
#pragma unroll
            for (int r = 0; r < UR; ++r) {
                const int j = jb + r * STEP + grp;
                vraw[r] = vpre[r];
                if (jb != jfirst) {
                    vraw[r] = uint4{0u, 0u, 0u, 0u};
                    if (j < j1) vraw[r] = *reinterpret_cast<const uint4*>(a.vc + (long)j * a.kv_ss + h * HD + sub * 8);
                }
            }
#pragma unroll
            for (int r = 0; r < UR; ++r) {
                const int j = jb + r * STEP + grp;
                if (j < j1) {
                    const bf16_t* vv = reinterpret_cast<const bf16_t*>(&vraw[r]);
                    const float pj = sc[j - j0];
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] = fmaf(pj, (float)vv[e], acc[e]);
                }
            }
        }
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int o = CPR; o < 64; o <<= 1) acc[e] += __shfl_xor(acc[e], o, 64);
    if (grp == 0)
#pragma unroll
        for (int e = 0; e < 8; ++e) red[wave * HD + sub * 8 + e] = acc[e];
    __syncthreads();
    if (t == 0) { rec[0] = mx; rec[1] = sum; }
    if (t < HD) rec[4 + t] = (red[t] + red[HD + t]) + (red[2 * HD + t] + red[3 * HD + t]);
}

// token embedding + learned position + LayerNorm of ONE decoder token at the position kept in device memory: i = ++*pos
// (PianoBart.py:60-71 through the projected table, modeling_bart.py positions offset 2; same sums as embed_ln_fwd_kernel)
struct SegOff9 { int off[9]; };
__global__ __launch_bounds__(256) void dec_embed_kernel(const int16_t* __restrict__ tok16, const float* __restrict__ P, const SegOff9 so,
                                                        const float* __restrict__ lin_b, const float* __restrict__ pos_tab,
                                                        const float* __restrict__ w, const float* __restrict__ b, bf16_t* __restrict__ y,
                                                        int* __restrict__ pos, int d, float eps) {
    __shared__ float red1[4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, d4 = d >> 2;
    const int i = *pos + 1;
    const uint4 raw = *reinterpret_cast<const uint4*>(tok16);
    int id[8];
    id[0] = (int)(raw.x & 0xffff); id[1] = (int)(raw.x >> 16); id[2] = (int)(raw.y & 0xffff); id[3] = (int)(raw.y >> 16);
    id[4] = (int)(raw.z & 0xffff); id[5] = (int)(raw.z >> 16); id[6] = (int)(raw.w & 0xffff); id[7] = (int)(raw.w >> 16);
    const bool in = t < d4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (in) {
        v = load4(lin_b + 4 * t) + load4(pos_tab + (size_t)(i + 2) * d + 4 * t);
#pragma unroll
        for (int k = 0; k < 8; ++k) v += load4(P + (size_t)(so.off[k] + id[k]) * d + 4 * t);
    }
    const float mean = block_sum4(in ? v[0] + v[1] + v[2] + v[3] : 0.f, red1, lane, wave) / (float)d;
    float q = 0.f;
    if (in) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float c = v[j] - mean; q += c * c; }
    }
    const float rstd = rsqrtf(block_sum4(q, red1, lane, wave) / (float)d + eps);
    if (in) store4(y + 4 * t, (v - mean) * rstd * load4(w + 4 * t) + load4(b + 4 * t));
    if (t == 0) *pos = i;                                        // every thread has read *pos (two barriers ago); later launches see i
}

// ---------------------------------------------------------------- device-side nucleus sampling (round 6)
// model.py:68-107 for the 8 heads of ONE position, on the device: logits row -> y = logit / T[h] -> softmax -> nucleus(p[h]) with
// the uniform draw u[pos][h] the host drew AHEAD for this position (the draws do not depend on the logits, model.py:97 /
// np.random.choice) -> the 8 ids of the next decoder input, written to tok_dev for the next step's embedding kernel, and -- with the
// raw logits row -- to pinned host logs indexed by position. The arithmetic follows pb_nucleus_rows above step for step (probs /= (sum +
// 1e-5), descending order with ties by index, candidates up to the first cumsum > p, q = cand / sum(cand), f64 cdf / cdf[-1] > u) with
// wave-parallel prefix sums in place of numpy's left-to-right ones (33.6 -> 19.3 us per token under the profiler); it cannot reproduce the host path bit
// for bit anyway (torch's vectorised CPU exp and its summation order are 1 ulp apart), so the HOST remains the authority: it replays every position
// from the logged logits row with the reference code path and rolls the decoder back on the (rare) position where the device chose
// differently (Engine.generate). The device result is a PREDICTION that lets the next token start without a host round trip.
// One workgroup of 512 threads: wave h = head h for the softmax and the scans; the rank counting of the heads with p < 1 uses one
// thread per (head, class). fault_period > 0 (tests only) corrupts head 0's id at every fault_period-th position.
struct SampleArgs {
    const float* logits;                  // (vocab) f32 row of the position just decoded
    const double* u;                      // (S, 8) uniform draws, device
    const int* pos;                       // device: the position the row belongs to
    int16_t* tok_dev;                     // (8) next decoder input
    float* log_logits;                    // pinned host (S, vocab)
    int16_t* log_tok;                     // pinned host (S, 8)
    int vocab, fault_period;
    int off[8], n[8];
    float temp[8], p[8];
};
constexpr int SMP_W = 272;                // >= the largest head (262), multiple of 16
// inclusive prefix sums over a wave (lane order), by shuffles
__device__ __forceinline__ float wave_scan_f(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const float u = __shfl_up(v, o, 64); if (lane >= o) v += u; }
    return v;
}
__global__ __launch_bounds__(512) void dec_sample_kernel(const SampleArgs a) {
    __shared__ __attribute__((aligned(16))) float pn[8][SMP_W];     // normalised probabilities, class order
    __shared__ float sp[8][SMP_W + 64];   // ... in descending order (heads with p < 1), zero tail
    __shared__ int si[8][SMP_W];          // class of each sorted entry
    __shared__ int htok[8];
    const int t = threadIdx.x, lane = t & 63, h = t >> 6;
    const int pos = *a.pos;
    const int n = a.n[h], off = a.off[h];
    const float T = a.temp[h];
    const double u_draw = a.u[(size_t)pos * 8 + h];            // requested now: a load that depends on *pos would otherwise sit at the end of the chain
    // softmax(logit / T) of head h (torch.softmax(logit / t, dim=-1), model.py:103-104), then probs /= (sum(probs) + 1e-5) (model.py:85).
    // The sums here are wave reductions, not numpy's left-to-right ones: a common divisor that differs in its last bit moves every
    // probability alike, so the order and (but for a 1e-7 neighbourhood of a threshold) the choice stay -- the host checks every position.
    float y[5], e[5];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int c = lane + 64 * k;
        const bool in = c < n;
        const float lg = in ? a.logits[off + c] : 0.f;
        if (in) a.log_logits[(size_t)pos * a.vocab + off + c] = lg;
        y[k] = in ? lg / T : -INFINITY;
        mx = fmaxf(mx, y[k]);
    }
    mx = wave_max(mx);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 5; ++k) { e[k] = (lane + 64 * k < n) ? expf(y[k] - mx) : 0.f; s += e[k]; }
    s = wave_sum(s);
    // probs = e / s, then probs /= (sum(probs) + 1e-5) with sum(probs) = 1 to rounding: one division by s (1 + 1e-5). A common factor a few ulps off
    // numpy's moves every probability alike (same order, same candidates but for a 1e-6 neighbourhood of the threshold): the host checks.
    const float inv = 1.0f / (s * 1.00001f);
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int c = lane + 64 * k;
        if (c < SMP_W) pn[h][c] = c < n ? e[k] * inv : -1.f;     // -1 behind the head's classes: never ranked in front of a class
    }
    __syncthreads();
    // descending order of the heads with p < 1 by rank counting, one thread per (head, class): ties by class index
    {
        int hh = -1, c = t;
        for (int q = 0; q < 8; ++q) {
            if (a.p[q] < 1.0f) {
                if (hh < 0 && c < a.n[q]) hh = q;
                if (hh < 0) c -= a.n[q];
            }
        }
        if (hh >= 0) {
            const float v = pn[hh][c];
            int rank = 0;
#pragma unroll 17                                              // fixed trip count (the tail holds -1: never in front of a class), four independent LDS reads in flight
            for (int j4 = 0; j4 < SMP_W / 4; ++j4) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(&pn[hh][4 * j4]);
#pragma unroll
                for (int r = 0; r < 4; ++r) rank += (w[r] > v || (w[r] == v && 4 * j4 + r < c)) ? 1 : 0;
            }
            sp[hh][rank] = v; si[hh][rank] = c;
        }
        if (t < 8 * 64) sp[t >> 6][SMP_W + (t & 63)] = 0.f;
    }
    __syncthreads();
    const float ph = a.p[h];
    if (ph < 1.0f) {                                            // wave-uniform; no barrier below
        // lane l owns sorted entries 5 l .. 5 l + 4 (0 behind the head's classes): prefix sums in sorted order
        float v5[5], pre[5];
        float run = 0.f;
#pragma unroll
        for (int k = 0; k < 5; ++k) { const int i = 5 * lane + k; v5[k] = i < n ? sp[h][i] : 0.f; run += v5[k]; pre[k] = run; }
        const float base = wave_scan_f(run, lane) - run;
        int first = 0x7fffffff;                                  // candidates: up to and including the first cumsum > p; none -> top 1
#pragma unroll
        for (int k = 4; k >= 0; --k) if (5 * lane + k < n && base + pre[k] > ph) first = 5 * lane + k;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) first = min(first, __shfl_xor(first, o, 64));
        const int kc = first == 0x7fffffff ? 1 : first + 1;
        // q = cand / sum(cand), cdf = cumsum(q) / cdf[-1] > u  <=>  prefix_i > u * prefix_{kc - 1}: the prefix sums of the threshold test serve again
        // (numpy renormalises in f32 and accumulates the cdf in f64: a relative 1e-7 against a uniform u)
        float myqs = 0.f;
#pragma unroll
        for (int k = 0; k < 5; ++k) if (5 * lane + k == kc - 1) myqs = base + pre[k];
        const float qs = wave_sum(myqs);                         // exactly one lane holds a non-zero term
        const float thr = (float)(u_draw * (double)qs);
        int best = kc - 1;                                       // first candidate whose running sum exceeds u times the candidates' total
#pragma unroll
        for (int k = 4; k >= 0; --k) if (5 * lane + k < kc && base + pre[k] > thr) best = 5 * lane + k;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) best = min(best, __shfl_xor(best, o, 64));
        if (lane == 0) htok[h] = si[h][best];
    } else {                                                     // p = 1: the cumsum never exceeds it -> the largest probability (lowest class among equals)
        float bv = -1.f; int bi = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int c = lane + 64 * k;
            if (c < n) { const float v = pn[h][c]; if (v > bv) { bv = v; bi = c; } }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) htok[h] = bi;
    }
    __syncthreads();
    if (t < 8) {
        int id = htok[t];
        if (a.fault_period > 0 && t == 0 && (pos % a.fault_period) == a.fault_period - 1) id = (id + 1) % a.n[0];
        a.tok_dev[t] = (int16_t)id;
        a.log_tok[(size_t)pos * 8 + t] = (int16_t)id;
    }
}

constexpr int SPEC_K = 8;                  // tokens per graph replay of the device-sampled decode
constexpr int SPEC_EVENTS = 8;

struct Decoder {
    pb_decode_plan plan;
    hipStream_t stream = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    // device-sampled ("speculative") decode, pb_decoder_sampler_init .. pb_decoder_seek
    bool sampler = false;
    SampleArgs sa{};
    double* u_dev = nullptr;
    float* log_logits = nullptr;           // pinned (S, vocab)
    int16_t* log_tok = nullptr;            // pinned (S, 8)
    hipGraph_t graph1 = nullptr, graphK = nullptr;
    hipGraphExec_t exec1 = nullptr, execK = nullptr;
    hipEvent_t evs[SPEC_EVENTS] = {};
    int next_ev = 0;
    int* pos = nullptr;                    // device: position of the token being decoded
    int16_t* tok_dev = nullptr;            // device copy of the current token (8 ids)
    int16_t* tok_host = nullptr;           // pinned
    float* logits_host = nullptr;          // pinned
    hipEvent_t ev = nullptr;
    int launches = 0, use_graph = 1, ck_cross = 0, nsplit_cross = 0;
    int ns_self = PB_DECODE_MAX_SPLITS, ns_cross = PB_DECODE_MAX_SPLITS;    // workgroups per head of the two attention launches (records per head)
    int steps = 0;                         // tokens decoded since the last reset: the device position must stay inside the caches (plan.S rows)
    size_t lds_attn = 0;
};

template <int NC, int HD>
static void dec_attn_go(const DecAttnArgs& a, bool self, int H, int nrec, size_t lds, hipStream_t st) {
    if (self) hipLaunchKernelGGL((dec_attn_kernel<NC, HD, true>), dim3(H, nrec), dim3(768), lds, st, a);
    else hipLaunchKernelGGL((dec_attn_kernel<NC, HD, false>), dim3(H, nrec), dim3(256), lds, st, a);
}
static int dec_attn_launch(const DecAttnArgs& a, bool self, int H, int hd, int nrec, size_t lds, hipStream_t st) {
    const int nc = a.d / 256;
#define PB_DA(NC_) do { if (hd == 64) dec_attn_go<NC_, 64>(a, self, H, nrec, lds, st); else dec_attn_go<NC_, 128>(a, self, H, nrec, lds, st); } while (0)
    if (nc == 1) PB_DA(1); else if (nc == 2) PB_DA(2); else if (nc == 3) PB_DA(3); else PB_DA(4);
#undef PB_DA
    PB_LAUNCH_CHECK();
    return 0;
}

// the launches of one token on `st` (captured once, or issued directly when capture is unavailable); returns their number in *count
static int decoder_issue(Decoder* D, hipStream_t st, int* count) {
    const pb_decode_plan* p = &D->plan;
    const int d = p->d, H = p->H, hd = d / H, f = p->ffn, dt = p->dtype;
    const float scale = 1.0f / sqrtf((float)hd);
    int n = 0;
    SegOff9 so;
    for (int k = 0; k < 9; ++k) so.off[k] = p->tab_off[k];
    hipLaunchKernelGGL(dec_embed_kernel, dim3(1), dim3(256), 0, st, D->tok_dev, p->ptab, so, p->lin_b, p->pos, p->lne_w, p->lne_b, (bf16_t*)p->x, D->pos, d, 1e-5f);
    PB_LAUNCH_CHECK(); ++n;
    char* x = (char*)p->x; char* alt = (char*)p->y2;
    char* h = x;
    LnIn ln{nullptr, nullptr, nullptr, nullptr};
    const MergeIn mg_self{p->attn_part, D->ns_self, hd, hd + 4}, mg_cross{p->attn_part, D->ns_cross, hd, hd + 4};
    for (int l = 0; l < p->n_layers; ++l) {
        const pb_decode_layer& L = p->layers[l];
        DecAttnArgs a{};
        a.d = d; a.scale = scale; a.eps = 1e-5f; a.part = p->attn_part; a.pos = D->pos; a.kv_ss = 2 * d;
        // self-attention: LN2 of the layer below (or the embedding row), q|k|v of this token, keys 0 .. i
        a.x_in = (const bf16_t*)h; a.res = (const bf16_t*)ln.res; a.add = (const bf16_t*)p->a; a.gamma = ln.gamma; a.beta = ln.beta; a.ln_out = (bf16_t*)ln.out;
        a.Wq = (const bf16_t*)L.wqkv; a.bq = L.bqkv;
        a.Wk = a.Wq + (size_t)d * d; a.bk = L.bqkv + d; a.Wv = a.Wq + (size_t)2 * d * d; a.bv = L.bqkv + 2 * d;
        a.kc = (bf16_t*)L.kv_self; a.vc = a.kc + d; a.key_mask = nullptr; a.nreg = D->ns_self - 1; a.ck_fixed = 0; a.Sk_fixed = 0;
        if (dec_attn_launch(a, true, H, hd, D->ns_self, D->lds_attn, st)) return -1;
        ++n;
        if (ln.res) h = alt;
        if (gemv_launch(L.wo, p->ctx, L.bo, p->a, nullptr, d, d, d, dt, 0, 0, st, LnIn{nullptr, nullptr, nullptr, nullptr}, mg_self)) return -1;
        ++n;
        // cross-attention: LN1(h + a) -> y1, q_c, the cached encoder keys
        DecAttnArgs c{};
        c.d = d; c.scale = scale; c.eps = 1e-5f; c.part = p->attn_part; c.pos = D->pos; c.kv_ss = 2 * d;
        c.x_in = nullptr; c.res = (const bf16_t*)h; c.add = (const bf16_t*)p->a; c.gamma = L.ln1_w; c.beta = L.ln1_b; c.ln_out = (bf16_t*)p->y1;
        c.Wq = (const bf16_t*)L.wq_c; c.bq = L.bq_c;
        c.kc = (bf16_t*)const_cast<void*>(L.kv_cross); c.vc = c.kc + d; c.key_mask = p->enc_mask; c.nreg = D->nsplit_cross; c.ck_fixed = D->ck_cross; c.Sk_fixed = p->S_enc;
        if (dec_attn_launch(c, false, H, hd, D->ns_cross, D->lds_attn, st)) return -1;
        ++n;
        if (gemv_launch(L.wo_c, p->ctx, L.bo_c, p->a, nullptr, d, d, d, dt, 0, 0, st, LnIn{nullptr, nullptr, nullptr, nullptr}, mg_cross)) return -1;
        ++n;
        // FFN: fc1 applies LNc(y1 + a) -> yc
        if (gemv_launch(L.w1, p->a, L.b1, p->g, nullptr, f, f, d, dt, 0, 1, st, LnIn{p->y1, L.lnc_w, L.lnc_b, p->yc})) return -1;
        ++n;
        if (gemv_launch(L.w2, p->g, L.b2, p->a, nullptr, d, d, f, dt, 0, 0, st)) return -1;
        ++n;
        ln = LnIn{p->yc, L.ln2_w, L.ln2_b, alt};
    }
    if (gemv_launch(p->head_w, p->a, p->head_b, p->logits, nullptr, p->vocab, p->vocab, d, dt, 1, 0, st, ln)) return -1;
    ++n;
    *count = n;
    return 0;
}

}  // namespace

extern "C" int pb_decoder_create(const pb_decode_plan* plan, void** out) {
    PB_REQUIRE(plan && out, "pb_decoder_create: null argument");
    *out = nullptr;
    const int d = plan->d, H = plan->H, hd = H > 0 ? d / H : 0;
    // shapes the fused kernels cover; anything else keeps pb_decode_step (return 1 = declined, not an error)
    if (plan->dtype != PB_BF16 || H <= 0 || d % H != 0 || (hd != 64 && hd != 128) || d % 256 != 0 || d > 1024 || !plan->attn_part ||
        plan->n_layers <= 0 || plan->n_layers > PB_DECODE_MAX_LAYERS || plan->ffn % 8 != 0 || plan->ffn > 8192 || plan->S_enc <= 0) return 1;
    Decoder* D = new Decoder();
    D->plan = *plan;
    if (hipStreamCreateWithFlags(&D->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&D->ev, hipEventDisableTiming) != hipSuccess ||
        hipMalloc(&D->pos, 64) != hipSuccess || hipMalloc(&D->tok_dev, 64) != hipSuccess ||
        hipHostMalloc(&D->tok_host, 64, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc(&D->logits_host, sizeof(float) * (size_t)plan->vocab, hipHostMallocDefault) != hipSuccess) {
        pb_set_error("pb_decoder_create: allocation failed: %s", hipGetErrorString(hipGetLastError()));
        pb_decoder_destroy(D);
        return -1;
    }
    // cross-attention: <= 16 splits of >= 64 keys over the visible encoder positions (fixed for the prompt)
    // Every (head, split) workgroup projects q of its head itself (hd rows of W_q: 98 KB at cfg 2), so the split count trades K / V rows
    // per workgroup against re-reads of W_q: PB_DECODE_SPLITS_SELF / _CROSS (2 .. 16; developer A/B, profiles/r06_decode_splits_ab.txt)
    auto env_splits = [](const char* name, int dflt) { const char* e = getenv(name); int v = e ? atoi(e) : dflt; return v < 2 ? 2 : (v > PB_DECODE_MAX_SPLITS ? PB_DECODE_MAX_SPLITS : v); };
    D->ns_self = env_splits("PB_DECODE_SPLITS_SELF", PB_DECODE_MAX_SPLITS);
    D->ns_cross = env_splits("PB_DECODE_SPLITS_CROSS", PB_DECODE_MAX_SPLITS);
    int ck = (plan->S_enc + D->ns_cross - 1) / D->ns_cross;
    ck = ck < 64 ? 64 : (ck + 15) & ~15;
    D->ck_cross = ck; D->nsplit_cross = D->ns_cross;
    int ck_self = (plan->S + D->ns_self - 2) / (D->ns_self - 1);
    ck_self = ck_self < 64 ? 64 : (ck_self + 15) & ~15;
    D->lds_attn = sizeof(float) * (size_t)(5 * hd + (ck > ck_self ? ck : ck_self) + 16);
    *out = D;
    return 0;
}

extern "C" int pb_decoder_destroy(void* dec) {
    Decoder* D = (Decoder*)dec;
    if (!D) return 0;
    if (D->stream) (void)hipStreamSynchronize(D->stream);
    if (D->exec) (void)hipGraphExecDestroy(D->exec);
    if (D->graph) (void)hipGraphDestroy(D->graph);
    if (D->exec1) (void)hipGraphExecDestroy(D->exec1);
    if (D->graph1) (void)hipGraphDestroy(D->graph1);
    if (D->execK) (void)hipGraphExecDestroy(D->execK);
    if (D->graphK) (void)hipGraphDestroy(D->graphK);
    for (int i = 0; i < SPEC_EVENTS; ++i) if (D->evs[i]) (void)hipEventDestroy(D->evs[i]);
    if (D->u_dev) (void)hipFree(D->u_dev);
    if (D->log_logits) (void)hipHostFree(D->log_logits);
    if (D->log_tok) (void)hipHostFree(D->log_tok);
    if (D->ev) (void)hipEventDestroy(D->ev);
    if (D->pos) (void)hipFree(D->pos);
    if (D->tok_dev) (void)hipFree(D->tok_dev);
    if (D->tok_host) (void)hipHostFree(D->tok_host);
    if (D->logits_host) (void)hipHostFree(D->logits_host);
    if (D->stream) (void)hipStreamDestroy(D->stream);
    delete D;
    return 0;
}

// Start of a prompt: the decoder's stream waits for everything already enqueued on the caller's stream (encoder pass, cross K/V
// projections), the position counter goes to -1. use_graph = 0 issues the launches of every token directly (A/B, debugging).
extern "C" int pb_decoder_reset(void* dec, void* caller_stream, int32_t use_graph) {
    Decoder* D = (Decoder*)dec;
    PB_REQUIRE(D, "pb_decoder_reset: null decoder");
    PB_CHECK_HIP(hipEventRecord(D->ev, (hipStream_t)caller_stream));
    PB_CHECK_HIP(hipStreamWaitEvent(D->stream, D->ev, 0));
    PB_CHECK_HIP(hipMemsetAsync(D->pos, 0xff, 4, D->stream));           // -1
    D->steps = 0;
    D->use_graph = use_graph;
    if (use_graph && !D->exec) {
        PB_CHECK_HIP(hipStreamSynchronize(D->stream));
        int n = 0;
        hipError_t e = hipStreamBeginCapture(D->stream, hipStreamCaptureModeRelaxed);
        if (e == hipSuccess) {
            int rc = 0;
            if (hipMemcpyAsync(D->tok_dev, D->tok_host, 16, hipMemcpyHostToDevice, D->stream) != hipSuccess) rc = -1;
            if (!rc) rc = decoder_issue(D, D->stream, &n);
            if (!rc && hipMemcpyAsync(D->logits_host, D->plan.logits, sizeof(float) * (size_t)D->plan.vocab, hipMemcpyDeviceToHost, D->stream) != hipSuccess) rc = -1;
            e = hipStreamEndCapture(D->stream, &D->graph);
            if (rc || e != hipSuccess || !D->graph || hipGraphInstantiate(&D->exec, D->graph, nullptr, nullptr, 0) != hipSuccess) {
                (void)hipGetLastError();
                if (D->graph) { (void)hipGraphDestroy(D->graph); D->graph = nullptr; }
                D->exec = nullptr;
            }
        } else {
            (void)hipGetLastError();
        }
        if (!D->exec) D->use_graph = 0;                                  // capture unavailable: direct launches
        else D->launches = n;
    }
    return 0;
}

// One token: tok8 (the decoder input of this position, 8 ids) goes up, the (vocab) f32 logits row of the position comes back into
// logits_out (host memory). Blocks until the row has landed.
extern "C" int pb_decoder_step(void* dec, const int16_t* tok8, float* logits_out) {
    Decoder* D = (Decoder*)dec;
    PB_REQUIRE(D && tok8 && logits_out, "pb_decoder_step: null argument");
    PB_REQUIRE(D->steps < D->plan.S, "pb_decoder_step: position %d is outside the K/V caches and the position table (S = %d): call pb_decoder_reset for a new prompt",
               D->steps, D->plan.S);
    ++D->steps;
    for (int k = 0; k < 8; ++k) D->tok_host[k] = tok8[k];
    if (D->use_graph && D->exec) {
        PB_CHECK_HIP(hipGraphLaunch(D->exec, D->stream));
    } else {
        PB_CHECK_HIP(hipMemcpyAsync(D->tok_dev, D->tok_host, 16, hipMemcpyHostToDevice, D->stream));
        int n = 0;
        if (decoder_issue(D, D->stream, &n)) return -1;
        D->launches = n;
        PB_CHECK_HIP(hipMemcpyAsync(D->logits_host, D->plan.logits, sizeof(float) * (size_t)D->plan.vocab, hipMemcpyDeviceToHost, D->stream));
    }
    PB_CHECK_HIP(hipStreamSynchronize(D->stream));
    for (int k = 0; k < D->plan.vocab; ++k) logits_out[k] = D->logits_host[k];
    return 0;
}

// ---- device-sampled decode: the sampler's constants and the uniform draws of the whole prompt go up once; tokens are then enqueued in
// runs (pb_decoder_launch: SPEC_K tokens = one graph replay) without waiting for the host; the host follows behind through the pinned
// logs (pb_decoder_wait + pb_decoder_logs), and pb_decoder_seek rewinds after a position where it disagrees with the device's choice.
extern "C" int pb_decoder_sampler_init(void* dec, const float* temps8, const float* p8, const int32_t* n8, const int32_t* off8,
                                       const double* u, int64_t n_u, int32_t fault_period) {
    Decoder* D = (Decoder*)dec;
    PB_REQUIRE(D && temps8 && p8 && n8 && off8 && u, "pb_decoder_sampler_init: null argument");
    PB_REQUIRE(n_u >= (int64_t)D->plan.S * 8, "pb_decoder_sampler_init: %lld draws for %d positions x 8 heads", (long long)n_u, D->plan.S);
    for (int h = 0; h < 8; ++h) {
        PB_REQUIRE(n8[h] > 0 && n8[h] <= SMP_W && n8[h] <= 320 && off8[h] >= 0 && off8[h] + n8[h] <= D->plan.vocab && temps8[h] > 0.f,
                   "pb_decoder_sampler_init: head %d (n %d, offset %d, temperature %g)", h, n8[h], off8[h], (double)temps8[h]);
        D->sa.n[h] = n8[h]; D->sa.off[h] = off8[h]; D->sa.temp[h] = temps8[h]; D->sa.p[h] = p8[h];
    }
    int sorted = 0;
    for (int h = 0; h < 8; ++h) if (p8[h] < 1.0f) sorted += n8[h];
    PB_REQUIRE(sorted <= 512, "pb_decoder_sampler_init: %d classes under heads with p < 1 (one thread each, 512 threads)", sorted);
    const size_t S = (size_t)D->plan.S;
    if (!D->u_dev) {
        if (hipMalloc(&D->u_dev, sizeof(double) * S * 8) != hipSuccess ||
            hipHostMalloc(&D->log_logits, sizeof(float) * S * (size_t)D->plan.vocab, hipHostMallocDefault) != hipSuccess ||
            hipHostMalloc(&D->log_tok, sizeof(int16_t) * S * 8, hipHostMallocDefault) != hipSuccess) {
            pb_set_error("pb_decoder_sampler_init: allocation failed: %s", hipGetErrorString(hipGetLastError()));
            return -1;
        }
        for (int i = 0; i < SPEC_EVENTS; ++i) PB_CHECK_HIP(hipEventCreateWithFlags(&D->evs[i], hipEventDisableTiming));
    }
    PB_CHECK_HIP(hipMemcpyAsync(D->u_dev, u, sizeof(double) * S * 8, hipMemcpyHostToDevice, D->stream));
    PB_CHECK_HIP(hipStreamSynchronize(D->stream));                     // `u` may be pageable: the copy is done when we return
    D->sa.logits = D->plan.logits; D->sa.u = D->u_dev; D->sa.pos = D->pos; D->sa.tok_dev = D->tok_dev;
    D->sa.log_logits = D->log_logits; D->sa.log_tok = D->log_tok; D->sa.vocab = D->plan.vocab; D->sa.fault_period = fault_period;
    D->sampler = true;
    return 0;
}

static int spec_issue(Decoder* D, hipStream_t st, int ntok, int* count) {
    int n = 0;
    for (int k = 0; k < ntok; ++k) {
        if (decoder_issue(D, st, &n)) return -1;
        hipLaunchKernelGGL(dec_sample_kernel, dim3(1), dim3(512), 0, st, D->sa);
        PB_LAUNCH_CHECK();
    }
    *count = n + 1;
    return 0;
}

static bool spec_capture(Decoder* D, int ntok, hipGraph_t* g, hipGraphExec_t* x) {
    int n = 0;
    if (hipStreamBeginCapture(D->stream, hipStreamCaptureModeRelaxed) != hipSuccess) { (void)hipGetLastError(); return false; }
    const int rc = spec_issue(D, D->stream, ntok, &n);
    const hipError_t e = hipStreamEndCapture(D->stream, g);
    if (rc || e != hipSuccess || !*g || hipGraphInstantiate(x, *g, nullptr, nullptr, 0) != hipSuccess) {
        (void)hipGetLastError();
        if (*g) { (void)hipGraphDestroy(*g); *g = nullptr; }
        *x = nullptr;
        return false;
    }
    D->launches = n;
    return true;
}

// Enqueue the next `ntok` tokens (decoder input of the first = tok_dev as the previous step's sampler, pb_decoder_seek or `first_tok8`
// left it). Returns a ticket >= 0 for pb_decoder_wait, < 0 on error. first_tok8 (8 ids, may be NULL) is copied up in front of the run.
extern "C" int pb_decoder_launch(void* dec, int32_t ntok, const int16_t* first_tok8) {
    Decoder* D = (Decoder*)dec;
    PB_REQUIRE(D && D->sampler, "pb_decoder_launch: pb_decoder_sampler_init first");
    PB_REQUIRE(ntok > 0 && D->steps + ntok <= D->plan.S, "pb_decoder_launch: %d tokens from position %d leave the K/V caches (S = %d)", ntok, D->steps, D->plan.S);
    if (first_tok8) {
        PB_CHECK_HIP(hipStreamSynchronize(D->stream));                 // tok_host is about to be rewritten: no copy of it may be in flight
        for (int k = 0; k < 8; ++k) D->tok_host[k] = first_tok8[k];
        PB_CHECK_HIP(hipMemcpyAsync(D->tok_dev, D->tok_host, 16, hipMemcpyHostToDevice, D->stream));
    }
    if (D->use_graph && !D->exec1) {
        PB_CHECK_HIP(hipStreamSynchronize(D->stream));
        if (!spec_capture(D, 1, &D->graph1, &D->exec1) || !spec_capture(D, SPEC_K, &D->graphK, &D->execK)) D->use_graph = 0;
    }
    int left = ntok;
    while (left > 0) {
        if (D->use_graph && left >= SPEC_K) { PB_CHECK_HIP(hipGraphLaunch(D->execK, D->stream)); left -= SPEC_K; }
        else if (D->use_graph) { PB_CHECK_HIP(hipGraphLaunch(D->exec1, D->stream)); left -= 1; }
        else { int n = 0; if (spec_issue(D, D->stream, 1, &n)) return -1; D->launches = n; left -= 1; }
    }
    D->steps += ntok;
    const int tk = D->next_ev;
    D->next_ev = (D->next_ev + 1) % SPEC_EVENTS;
    PB_CHECK_HIP(hipEventRecord(D->evs[tk], D->stream));
    return tk;
}
extern "C" int pb_decoder_wait(void* dec, int32_t ticket) {
    Decoder* D = (Decoder*)dec;
    PB_REQUIRE(D && D->sampler && ticket >= 0 && ticket < SPEC_EVENTS, "pb_decoder_wait: bad ticket %d", ticket);
    PB_CHECK_HIP(hipEventSynchronize(D->evs[ticket]));
    return 0;
}
extern "C" int pb_decoder_logs(void* dec, float** logits_rows, int16_t** tok_rows) {
    Decoder* D = (Decoder*)dec;
    PB_REQUIRE(D && D->sampler && logits_rows && tok_rows, "pb_decoder_logs: no sampler");
    *logits_rows = D->log_logits; *tok_rows = D->log_tok;
    return 0;
}
// Rewind: drain what is enqueued, make `pos` the last decoded position and tok8 the decoder input of position pos + 1.
extern "C" int pb_decoder_seek(void* dec, int32_t pos, const int16_t* tok8) {
    Decoder* D = (Decoder*)dec;
    PB_REQUIRE(D && tok8 && pos >= -1 && pos < D->plan.S, "pb_decoder_seek: position %d", pos);
    PB_CHECK_HIP(hipStreamSynchronize(D->stream));
    for (int k = 0; k < 8; ++k) D->tok_host[k] = tok8[k];
    PB_CHECK_HIP(hipMemcpyAsync(D->tok_dev, D->tok_host, 16, hipMemcpyHostToDevice, D->stream));
    PB_CHECK_HIP(hipMemcpyAsync(D->pos, &pos, 4, hipMemcpyHostToDevice, D->stream));
    PB_CHECK_HIP(hipStreamSynchronize(D->stream));
    D->steps = pos + 1;
    return 0;
}

extern "C" int pb_decoder_launches(void* dec) { return dec ? ((Decoder*)dec)->launches : 0; }
extern "C" int pb_decoder_graph(void* dec) { return dec ? (((Decoder*)dec)->use_graph && ((Decoder*)dec)->exec ? 1 : 0) : 0; }
