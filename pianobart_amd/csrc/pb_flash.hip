// K4: fused (flash-style) attention for gfx950, bf16 in / f32 accumulate, head_dim 32 / 64 / 128.
// Replaces matmul*scale + mask -> softmax -> matmul (transformers modeling_bart.py:115-140 / SDPA) and
// its backward. Masks: key padding (per-batch float row, != 0 keeps), optional causal (key <= query);
// a query row with no visible key produces an all-zero output row (oracle header).
//
// Layout trick used everywhere below (cdna_hip_programming.md 3, "An accumulator tile as the next
// MFMA's operand"): a 16x16 f32 accumulator tile X has its COLUMN on lane&15 and rows 4*(lane>>4)+r in
// its 4 registers. Two stacked tiles (32 rows) converted to bf16 are directly the A- or B-operand
// fragment of the next v_mfma_f32_16x16x32_bf16 that contracts over X's ROW index, provided the other
// operand enumerates k in the same permuted order:  k(j) = 16*(j>>2) + 4*(lane>>4) + (j&3), j = 0..7,
// i.e. two 8-byte reads from a k-contiguous LDS image. So S/P never touch LDS.
//
//   forward  (block = 64 queries, wave = 16 queries; loop over 64-key tiles):
//       S^T[key][q] = K Q^T          A = K rows (LDS [key][d]),   B = Q rows (registers)
//       online softmax on lane-local columns (q = lane&15; 16 keys per lane; 2 shuffles per reduction)
//       O^T[d][q]  += V^T P^T        A = V^T (LDS [d][key], transposed while staging), B = P^T (registers)
//   backward = two kernels, no atomics, deterministic:
//       dKV (block = 64 keys, wave = 16 keys; loop over 64-query tiles)
//            S = Q K^T, dP = dO V^T (B = K / V rows in registers), dV += P^T dO, dK += dS^T Q
//       dQ  (block = 64 queries, wave = 16 queries; loop over 64-key tiles)
//            S^T = K Q^T, dP^T = V dO^T, dQ^T += K^T dS^T
#include "pb_common.h"
#include "pb_api_internal.h"

namespace {

constexpr int FA_THREADS = 256;
constexpr int TQ = 64, TK = 64;           // tile sizes (queries / keys per block-iteration)
constexpr float LOG2E = 1.4426950408889634f;

// ---------------------------------------------------------------- LDS images
// "rows" image: [64 rows][HD] bf16, row = HD*2 bytes, 16-B chunks XOR-swizzled so that a ds_read_b128 of 16
// consecutive rows at one logical chunk is bank-conflict free.
template <int HD> __device__ __forceinline__ int rows_off(int row, int chunk) {
    constexpr int RB = HD * 2, NCH = HD / 8, RPB = 256 / RB;      // RB <= 256
    return row * RB + (((chunk ^ ((row / RPB) % NCH))) << 4);
}
// "transposed" image: [HD rows][64] bf16 (128-B rows): element (r, k) = source (k, r).
__device__ __forceinline__ int tswz(int row) { return ((row >> 1) & 7) ^ ((row >> 4) & 7); }
__device__ __forceinline__ int tr_off(int row, int k) {         // k multiple of 4; returns byte offset of 4 consecutive k
    return row * 128 + ((((k >> 3)) ^ tswz(row)) << 4) + (((k >> 2) & 1) << 3);
}

// Stage a [64][HD] tile (rows >= nvalid read as zero) from global (row stride `ld` elements) into the rows image.
template <int HD>
__device__ __forceinline__ void stage_rows(char* lds, const bf16_t* __restrict__ g, long ld, int nvalid, int t) {
    constexpr int NCH = HD / 8;
    for (int v = t; v < 64 * NCH; v += FA_THREADS) {
        const int row = v / NCH, ch = v % NCH;
        uint4 q = make_uint4(0u, 0u, 0u, 0u);
        if (row < nvalid) q = *reinterpret_cast<const uint4*>(g + (long)row * ld + ch * 8);
        *reinterpret_cast<uint4*>(lds + rows_off<HD>(row, ch)) = q;
    }
}
// Stage the same tile transposed: thread owns a 4(src rows) x 8(cols) block, transposes it with v_perm.
template <int HD>
__device__ __forceinline__ void stage_transposed(char* lds, const bf16_t* __restrict__ g, long ld, int nvalid, int t) {
    constexpr int NCC = HD / 8;                                   // column chunks
    for (int blk = t; blk < 16 * NCC; blk += FA_THREADS) {
        const int cc = blk % NCC, kg = blk / NCC;
        uint4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = kg * 4 + i;
            v[i] = make_uint4(0u, 0u, 0u, 0u);
            if (row < nvalid) v[i] = *reinterpret_cast<const uint4*>(g + (long)row * ld + cc * 8);
        }
        const uint32_t* w0 = reinterpret_cast<const uint32_t*>(&v[0]);
        const uint32_t* w1 = reinterpret_cast<const uint32_t*>(&v[1]);
        const uint32_t* w2 = reinterpret_cast<const uint32_t*>(&v[2]);
        const uint32_t* w3 = reinterpret_cast<const uint32_t*>(&v[3]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int w = j >> 1;
            const uint32_t sel = (j & 1) ? 0x07060302u : 0x05040100u;
            const uint32_t lo = __builtin_amdgcn_perm(w1[w], w0[w], sel);
            const uint32_t hi = __builtin_amdgcn_perm(w3[w], w2[w], sel);
            *reinterpret_cast<uint2*>(lds + tr_off(cc * 8 + j, kg * 4)) = make_uint2(lo, hi);
        }
    }
}

// Operand fragments.
// (a) natural: 8 consecutive k (= 32*ks + 8*g + j) of row (tile*16 + lane&15) from a rows image.
template <int HD>
__device__ __forceinline__ bf16x8 frag_rows(const char* lds, int row, int ks, int g) {
    return *reinterpret_cast<const bf16x8*>(lds + rows_off<HD>(row, ks * 4 + g));
}
// (b) permuted-k (matches an accumulator pair used as the other operand): k(j) = 32*s + 16*(j>>2) + 4*g + (j&3).
__device__ __forceinline__ bf16x8 frag_perm(const char* lds, int row, int s, int g) {
    const bf16x4 a = *reinterpret_cast<const bf16x4*>(lds + tr_off(row, 32 * s + 4 * g));
    const bf16x4 b = *reinterpret_cast<const bf16x4*>(lds + tr_off(row, 32 * s + 16 + 4 * g));
    bf16x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return r;
}
// accumulator pair -> operand fragment
__device__ __forceinline__ bf16x8 pack_pair(const f32x4& lo, const f32x4& hi) {
    bf16x8 r = {(bf16_t)lo[0], (bf16_t)lo[1], (bf16_t)lo[2], (bf16_t)lo[3], (bf16_t)hi[0], (bf16_t)hi[1], (bf16_t)hi[2], (bf16_t)hi[3]};
    return r;
}
// global rows straight to an operand fragment (row-major source, 8 consecutive elements)
__device__ __forceinline__ bf16x8 frag_global(const bf16_t* __restrict__ g, long ld, int row, int nvalid, int col) {
    bf16x8 z = {};
    if (row < nvalid) z = *reinterpret_cast<const bf16x8*>(g + (long)row * ld + col);
    return z;
}

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

struct FaArgs {
    const bf16_t *q, *k, *v, *o, *dout;
    bf16_t *out, *dq, *dk, *dv;
    float* lse; const float* delta; const float* key_mask;
    int B, H, Sq, Sk;
    long q_sb, q_ss, k_sb, k_ss, v_sb, v_ss, o_sb, o_ss, dq_sb, dq_ss, dk_sb, dk_ss, dv_sb, dv_ss;
    float scale; int causal;
};

// group reductions over the 4 lane groups that share lane&15 (lanes l, l^16, l^32, l^48)
__device__ __forceinline__ float grp_max(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ float grp_sum(float v) { v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); }

// ================================================================== forward
template <int HD>
__global__ __launch_bounds__(FA_THREADS) void fa_fwd_kernel(const FaArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ldsK = smem;                         // rows image   [64][HD]
    char* ldsV = smem + 64 * HD * 2;           // transposed   [HD][64]
    float* ldsB = reinterpret_cast<float*>(smem + 2 * 64 * HD * 2);   // key bias [64]: 0 or -inf
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, lr = lane & 15, g = lane >> 4;
    int rb_, h, b;
    grid_map3(rb_, h, b);
    const int q0 = rb_ * TQ;
    const bf16_t* Q = p.q + b * p.q_sb + h * HD;
    const bf16_t* K = p.k + b * p.k_sb + h * HD;
    const bf16_t* V = p.v + b * p.v_sb + h * HD;
    const int myq = q0 + wave * 16 + lr;                  // this lane's query (column of S^T / O^T)
    constexpr int KS = HD / 32, DT = HD / 16;
    bf16x8 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = frag_global(Q, p.q_ss, myq, p.Sq, ks * 32 + g * 8);
    f32x4 oacc[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i) oacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, l = 0.f;
    const float c = p.scale * LOG2E;
    const int kend = p.causal ? min(p.Sk, q0 + TQ) : p.Sk;
    for (int k0 = 0; k0 < kend; k0 += TK) {
        __syncthreads();
        stage_rows<HD>(ldsK, K + (long)k0 * p.k_ss, p.k_ss, p.Sk - k0, t);
        stage_transposed<HD>(ldsV, V + (long)k0 * p.v_ss, p.v_ss, p.Sk - k0, t);
        if (t < TK) {
            const int key = k0 + t;
            const bool vis = key < p.Sk && (!p.key_mask || p.key_mask[(long)b * p.Sk + key] != 0.f);
            ldsB[t] = vis ? 0.f : -INFINITY;
        }
        __syncthreads();
        // S^T tiles: keys 16*kt + (4g + r), query = lane&15
        f32x4 s[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) s[kt] = MFMA16(frag_rows<HD>(ldsK, kt * 16 + lr, ks, g), qf[ks], s[kt]);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const f32x4 bias = *reinterpret_cast<const f32x4*>(ldsB + kt * 16 + g * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = s[kt][r] * c + bias[r];
                if (p.causal && (k0 + kt * 16 + g * 4 + r) > myq) x = -INFINITY;
                s[kt][r] = x;
                mx = fmaxf(mx, x);
            }
        }
        mx = grp_max(mx);
        const float mnew = fmaxf(m, mx);
        const float muse = mnew == -INFINITY ? 0.f : mnew;
        const float alpha = exp2f(m - muse);               // m = -inf -> 0
        float rs = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float e = exp2f(s[kt][r] - muse); s[kt][r] = e; rs += e; }
        rs = grp_sum(rs);
        l = l * alpha + rs;
        m = mnew;
#pragma unroll
        for (int i = 0; i < DT; ++i) oacc[i] *= alpha;
        const bf16x8 p0 = pack_pair(s[0], s[1]), p1 = pack_pair(s[2], s[3]);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            oacc[dt] = MFMA16(frag_perm(ldsV, dt * 16 + lr, 0, g), p0, oacc[dt]);
            oacc[dt] = MFMA16(frag_perm(ldsV, dt * 16 + lr, 1, g), p1, oacc[dt]);
        }
    }
    // epilogue: O[q][16 dt + 4 g + r] = oacc[dt][r] / l ; lse in natural-log units of (scale * s)
    if (myq < p.Sq) {
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        bf16_t* O = p.out + b * p.o_sb + (long)myq * p.o_ss + h * HD;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            bf16x4 r = {(bf16_t)(oacc[dt][0] * inv), (bf16_t)(oacc[dt][1] * inv), (bf16_t)(oacc[dt][2] * inv), (bf16_t)(oacc[dt][3] * inv)};
            *reinterpret_cast<bf16x4*>(O + dt * 16 + g * 4) = r;
        }
        if (g == 0) p.lse[((long)b * p.H + h) * p.Sq + myq] = l > 0.f ? (m + log2f(l)) / LOG2E : INFINITY;
    }
}

// ================================================================== delta = rowsum(dO * O)
template <int HD>
__global__ void fa_delta_kernel(const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout, float* __restrict__ delta,
                                int B, int H, int Sq, long o_sb, long o_ss) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;     // (b, h, q)
    if (idx >= (long)B * H * Sq) return;
    const int q = (int)(idx % Sq), h = (int)((idx / Sq) % H), b = (int)(idx / ((long)Sq * H));
    const bf16_t* op = o + b * o_sb + (long)q * o_ss + h * HD;
    const bf16_t* dp = dout + b * o_sb + (long)q * o_ss + h * HD;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < HD; c += 8) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(op + c), d = *reinterpret_cast<const bf16x8*>(dp + c);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += (float)a[j] * (float)d[j];
    }
    delta[idx] = s;
}

// ================================================================== backward: dK, dV
template <int HD>
__global__ __launch_bounds__(FA_THREADS) void fa_bwd_dkv_kernel(const FaArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TB = 64 * HD * 2;
    char* ldsQ = smem;                  // rows image [64 q][HD]
    char* ldsQT = smem + TB;            // transposed [HD][64 q]
    char* ldsO = smem + 2 * TB;         // dO rows image
    char* ldsOT = smem + 3 * TB;        // dO transposed
    float* ldsL = reinterpret_cast<float*>(smem + 4 * TB);        // lse[64] (log2 units), delta[64]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, lr = lane & 15, g = lane >> 4;
    int rb_, h, b;
    grid_map3(rb_, h, b);
    const int k0 = rb_ * TK;
    const bf16_t* Q = p.q + b * p.q_sb + h * HD;
    const bf16_t* K = p.k + b * p.k_sb + h * HD;
    const bf16_t* V = p.v + b * p.v_sb + h * HD;
    const bf16_t* DO = p.dout + b * p.o_sb + h * HD;
    constexpr int KS = HD / 32, DT = HD / 16;
    const int mykey = k0 + wave * 16 + lr;                // this lane's key (column of S / dP tiles)
    bf16x8 kf[KS], vf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        kf[ks] = frag_global(K, p.k_ss, mykey, p.Sk, ks * 32 + g * 8);
        vf[ks] = frag_global(V, p.v_ss, mykey, p.Sk, ks * 32 + g * 8);
    }
    const bool kvis = mykey < p.Sk && (!p.key_mask || p.key_mask[(long)b * p.Sk + mykey] != 0.f);
    f32x4 dk[DT], dv[DT];                                  // [key = 4g+r][d = 16 dt + lane&15]
#pragma unroll
    for (int i = 0; i < DT; ++i) { dk[i] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const float c = p.scale * LOG2E;
    const int qstart = p.causal ? (k0 / TQ) * TQ : 0;
    for (int q0 = qstart; q0 < p.Sq; q0 += TQ) {
        __syncthreads();
        stage_rows<HD>(ldsQ, Q + (long)q0 * p.q_ss, p.q_ss, p.Sq - q0, t);
        stage_transposed<HD>(ldsQT, Q + (long)q0 * p.q_ss, p.q_ss, p.Sq - q0, t);
        stage_rows<HD>(ldsO, DO + (long)q0 * p.o_ss, p.o_ss, p.Sq - q0, t);
        stage_transposed<HD>(ldsOT, DO + (long)q0 * p.o_ss, p.o_ss, p.Sq - q0, t);
        if (t < TQ) {
            const int q = q0 + t;
            const long li = ((long)b * p.H + h) * p.Sq + q;
            ldsL[t] = q < p.Sq ? p.lse[li] * LOG2E : INFINITY;
            ldsL[64 + t] = q < p.Sq ? p.delta[li] : 0.f;
        }
        __syncthreads();
        // S[q][key], dP[q][key]: rows q = 16 qt + 4g + r, column key = lane&15
        f32x4 s[4], dp[4];
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
            s[qt] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                s[qt] = MFMA16(frag_rows<HD>(ldsQ, qt * 16 + lr, ks, g), kf[ks], s[qt]);
                dp[qt] = MFMA16(frag_rows<HD>(ldsO, qt * 16 + lr, ks, g), vf[ks], dp[qt]);
            }
        }
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
            const f32x4 lse = *reinterpret_cast<const f32x4*>(ldsL + qt * 16 + g * 4);
            const f32x4 dl = *reinterpret_cast<const f32x4*>(ldsL + 64 + qt * 16 + g * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = q0 + qt * 16 + g * 4 + r;
                const bool vis = kvis && (!p.causal || mykey <= q);
                const float pr = vis ? exp2f(s[qt][r] * c - lse[r]) : 0.f;
                s[qt][r] = pr;                                        // P
                dp[qt][r] = pr * (dp[qt][r] - dl[r]) * p.scale;       // dS
            }
        }
        const bf16x8 p0 = pack_pair(s[0], s[1]), p1 = pack_pair(s[2], s[3]);
        const bf16x8 d0 = pack_pair(dp[0], dp[1]), d1 = pack_pair(dp[2], dp[3]);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            // dV[key][d] += sum_q P[q][key] dO[q][d]   (A = P^T via accumulator pair, B = dO^T image rows d)
            dv[dt] = MFMA16(p0, frag_perm(ldsOT, dt * 16 + lr, 0, g), dv[dt]);
            dv[dt] = MFMA16(p1, frag_perm(ldsOT, dt * 16 + lr, 1, g), dv[dt]);
            dk[dt] = MFMA16(d0, frag_perm(ldsQT, dt * 16 + lr, 0, g), dk[dt]);
            dk[dt] = MFMA16(d1, frag_perm(ldsQT, dt * 16 + lr, 1, g), dk[dt]);
        }
    }
    // write: rows key = k0 + 16 wave + 4g + r, col d = 16 dt + lane&15
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int key = k0 + wave * 16 + g * 4 + r;
        if (key < p.Sk) {
            bf16_t* DK = p.dk + b * p.dk_sb + (long)key * p.dk_ss + h * HD;
            bf16_t* DV = p.dv + b * p.dv_sb + (long)key * p.dv_ss + h * HD;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) { DK[dt * 16 + lr] = (bf16_t)dk[dt][r]; DV[dt * 16 + lr] = (bf16_t)dv[dt][r]; }
        }
    }
}

// ================================================================== backward: dQ
template <int HD>
__global__ __launch_bounds__(FA_THREADS) void fa_bwd_dq_kernel(const FaArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TB = 64 * HD * 2;
    char* ldsK = smem;                  // rows image [64 keys][HD]
    char* ldsKT = smem + TB;            // transposed [HD][64 keys]
    char* ldsV = smem + 2 * TB;         // rows image
    float* ldsB = reinterpret_cast<float*>(smem + 3 * TB);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, lr = lane & 15, g = lane >> 4;
    int rb_, h, b;
    grid_map3(rb_, h, b);
    const int q0 = rb_ * TQ;
    const bf16_t* Q = p.q + b * p.q_sb + h * HD;
    const bf16_t* K = p.k + b * p.k_sb + h * HD;
    const bf16_t* V = p.v + b * p.v_sb + h * HD;
    const bf16_t* DO = p.dout + b * p.o_sb + h * HD;
    constexpr int KS = HD / 32, DT = HD / 16;
    const int myq = q0 + wave * 16 + lr;
    bf16x8 qf[KS], of[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        qf[ks] = frag_global(Q, p.q_ss, myq, p.Sq, ks * 32 + g * 8);
        of[ks] = frag_global(DO, p.o_ss, myq, p.Sq, ks * 32 + g * 8);
    }
    const long li = ((long)b * p.H + h) * p.Sq + myq;
    const float lse = myq < p.Sq ? p.lse[li] * LOG2E : INFINITY;
    const float dl = myq < p.Sq ? p.delta[li] : 0.f;
    f32x4 dq[DT];                                              // dQ^T[d = 16 dt + 4g + r][q = lane&15]
#pragma unroll
    for (int i = 0; i < DT; ++i) dq[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float c = p.scale * LOG2E;
    const int kend = p.causal ? min(p.Sk, q0 + TQ) : p.Sk;
    for (int k0 = 0; k0 < kend; k0 += TK) {
        __syncthreads();
        stage_rows<HD>(ldsK, K + (long)k0 * p.k_ss, p.k_ss, p.Sk - k0, t);
        stage_transposed<HD>(ldsKT, K + (long)k0 * p.k_ss, p.k_ss, p.Sk - k0, t);
        stage_rows<HD>(ldsV, V + (long)k0 * p.v_ss, p.v_ss, p.Sk - k0, t);
        if (t < TK) {
            const int key = k0 + t;
            ldsB[t] = (key < p.Sk && (!p.key_mask || p.key_mask[(long)b * p.Sk + key] != 0.f)) ? 1.f : 0.f;
        }
        __syncthreads();
        f32x4 s[4], dp[4];                                     // [key = 16 kt + 4g + r][q = lane&15]
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            s[kt] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                s[kt] = MFMA16(frag_rows<HD>(ldsK, kt * 16 + lr, ks, g), qf[ks], s[kt]);
                dp[kt] = MFMA16(frag_rows<HD>(ldsV, kt * 16 + lr, ks, g), of[ks], dp[kt]);
            }
        }
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const f32x4 vis4 = *reinterpret_cast<const f32x4*>(ldsB + kt * 16 + g * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = k0 + kt * 16 + g * 4 + r;
                const bool vis = vis4[r] != 0.f && (!p.causal || key <= myq);
                const float pr = vis ? exp2f(s[kt][r] * c - lse) : 0.f;
                dp[kt][r] = pr * (dp[kt][r] - dl) * p.scale;      // dS^T
            }
        }
        const bf16x8 d0 = pack_pair(dp[0], dp[1]), d1 = pack_pair(dp[2], dp[3]);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            dq[dt] = MFMA16(frag_perm(ldsKT, dt * 16 + lr, 0, g), d0, dq[dt]);
            dq[dt] = MFMA16(frag_perm(ldsKT, dt * 16 + lr, 1, g), d1, dq[dt]);
        }
    }
    if (myq < p.Sq) {
        bf16_t* DQ = p.dq + b * p.dq_sb + (long)myq * p.dq_ss + h * HD;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            bf16x4 r = {(bf16_t)dq[dt][0], (bf16_t)dq[dt][1], (bf16_t)dq[dt][2], (bf16_t)dq[dt][3]};
            *reinterpret_cast<bf16x4*>(DQ + dt * 16 + g * 4) = r;
        }
    }
}

int check_common(const char* who, int hd, long a, long b2, long c2, long d2) {
    PB_REQUIRE(hd == 32 || hd == 64 || hd == 96 || hd == 128, "%s: head_dim %d not supported by the flash kernels (32/64/96/128)", who, hd);
    PB_REQUIRE(a % 8 == 0 && b2 % 8 == 0 && c2 % 8 == 0 && d2 % 8 == 0, "%s: strides must be multiples of 8 elements", who);
    return 0;
}

}  // namespace

int pb_flash64_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const float* key_mask, const int* kmax, int B, int H, int Sq, int Sk, int hd,
                   long q_sb, long q_ss, long k_sb, long k_ss, long v_sb, long v_ss, long o_sb, long o_ss, float scale, int causal, hipStream_t stream,
                   const int* const* vl = nullptr);
int pb_flash64_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse, float* delta, const float* key_mask,
                   const int* kmax, void* dq, void* dk, void* dv, int B, int H, int Sq, int Sk, int hd, long q_sb, long q_ss, long k_sb, long k_ss, long v_sb,
                   long v_ss, long o_sb, long o_ss, long dq_sb, long dq_ss, long dk_sb, long dk_ss, long dv_sb, long dv_ss, float scale,
                   int causal, float* dbias_q, float* dbias_k, float* dbias_v, float* dbias_ws, hipStream_t stream, const int* const* vl = nullptr);

namespace {
__global__ void key_extent_kernel(const float* __restrict__ key_mask, int* __restrict__ kmax, int Sk) {
    const int b = blockIdx.x;
    int m = 0;
    for (int j = threadIdx.x; j < Sk; j += blockDim.x) if (key_mask[(long)b * Sk + j] != 0.f) m = max(m, j + 1);
    __shared__ int red[256];
    red[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] = max(red[threadIdx.x], red[threadIdx.x + o]); __syncthreads(); }
    if (threadIdx.x == 0) kmax[b] = red[0];
}
}  // namespace

extern "C" int pb_key_extent(const float* key_mask, int32_t* kmax, int32_t B, int32_t Sk, void* stream_) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(key_extent_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream_, key_mask, kmax, Sk);
    PB_LAUNCH_CHECK();
    return 0;
}

#define FA_DISPATCH(HDV, ...)                                   \
    switch (HDV) {                                              \
        case 32: { constexpr int HD = 32; __VA_ARGS__; } break; \
        case 64: { constexpr int HD = 64; __VA_ARGS__; } break; \
        default: { constexpr int HD = 128; __VA_ARGS__; } break; \
    }

extern "C" int pb_flash_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const float* key_mask, const int32_t* kmax, int32_t B,
                            int32_t H, int32_t Sq, int32_t Sk, int32_t hd, int64_t q_sb, int64_t q_ss, int64_t k_sb, int64_t k_ss,
                            int64_t v_sb, int64_t v_ss, int64_t o_sb, int64_t o_ss, float scale, int32_t causal, void* stream_) {
    if (check_common("pb_flash_fwd", hd, q_ss, k_ss, v_ss, o_ss)) return -2;
    PB_REQUIRE(q_sb % 8 == 0 && k_sb % 8 == 0 && v_sb % 8 == 0 && o_sb % 8 == 0, "pb_flash_fwd: batch strides must be multiples of 8");
    if (B <= 0 || H <= 0 || Sq <= 0) return 0;
    PB_REQUIRE(hd != 96 || !(causal & 2), "pb_flash_fwd: head_dim 96 exists in the pipelined kernel family only");
    if ((hd == 64 || hd == 96 || hd == 128) && !(causal & 2))      // bit 1 of `causal` forces the generic kernel (tests)
        return pb_flash64_fwd(q, k, v, o, lse, key_mask, kmax, B, H, Sq, Sk, hd, q_sb, q_ss, k_sb, k_ss, v_sb, v_ss, o_sb, o_ss, scale, causal & 1, (hipStream_t)stream_);
    FaArgs a = {};
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.out = (bf16_t*)o; a.lse = lse; a.key_mask = key_mask;
    a.B = B; a.H = H; a.Sq = Sq; a.Sk = Sk; a.q_sb = q_sb; a.q_ss = q_ss; a.k_sb = k_sb; a.k_ss = k_ss; a.v_sb = v_sb; a.v_ss = v_ss;
    a.o_sb = o_sb; a.o_ss = o_ss; a.scale = scale; a.causal = causal & 1;
    dim3 grid((Sq + TQ - 1) / TQ, H, B);
    FA_DISPATCH(hd, hipLaunchKernelGGL((fa_fwd_kernel<HD>), grid, dim3(FA_THREADS), 2 * 64 * HD * 2 + 256, (hipStream_t)stream_, a));
    PB_LAUNCH_CHECK();
    return 0;
}

extern "C" int pb_flash_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                            const float* key_mask, const int32_t* kmax, void* dq, void* dk, void* dv, float* delta, int32_t B, int32_t H, int32_t Sq,
                            int32_t Sk, int32_t hd, int64_t q_sb, int64_t q_ss, int64_t k_sb, int64_t k_ss, int64_t v_sb,
                            int64_t v_ss, int64_t o_sb, int64_t o_ss, int64_t dq_sb, int64_t dq_ss, int64_t dk_sb, int64_t dk_ss,
                            int64_t dv_sb, int64_t dv_ss, float scale, int32_t causal, float* dbias_q, float* dbias_k, float* dbias_v, float* dbias_ws,
                            void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (check_common("pb_flash_bwd", hd, q_ss, k_ss, v_ss, o_ss)) return -2;
    PB_REQUIRE(dq_ss % 4 == 0 && q_sb % 8 == 0 && k_sb % 8 == 0 && v_sb % 8 == 0 && o_sb % 8 == 0, "pb_flash_bwd: bad strides");
    if (B <= 0 || H <= 0 || Sq <= 0) return 0;
    FaArgs a = {};
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.o = (const bf16_t*)o; a.dout = (const bf16_t*)dout;
    a.dq = (bf16_t*)dq; a.dk = (bf16_t*)dk; a.dv = (bf16_t*)dv; a.lse = const_cast<float*>(lse); a.delta = delta; a.key_mask = key_mask;
    a.B = B; a.H = H; a.Sq = Sq; a.Sk = Sk; a.q_sb = q_sb; a.q_ss = q_ss; a.k_sb = k_sb; a.k_ss = k_ss; a.v_sb = v_sb; a.v_ss = v_ss;
    a.o_sb = o_sb; a.o_ss = o_ss; a.dq_sb = dq_sb; a.dq_ss = dq_ss; a.dk_sb = dk_sb; a.dk_ss = dk_ss; a.dv_sb = dv_sb; a.dv_ss = dv_ss;
    a.scale = scale; a.causal = causal & 1;
    const long nrow = (long)B * H * Sq;
    PB_REQUIRE(hd != 96 || !(causal & 2), "pb_flash_bwd: head_dim 96 exists in the pipelined kernel family only");
    if ((hd == 64 || hd == 96 || hd == 128) && !(causal & 2))          // the pipelined family computes delta inside its dQ kernel
        return pb_flash64_bwd(q, k, v, o, dout, lse, delta, key_mask, kmax, dq, dk, dv, B, H, Sq, Sk, hd, q_sb, q_ss, k_sb, k_ss, v_sb, v_ss, o_sb, o_ss,
                              dq_sb, dq_ss, dk_sb, dk_ss, dv_sb, dv_ss, scale, causal & 1, dbias_q, dbias_k, dbias_v, dbias_ws, stream);
    FA_DISPATCH(hd, hipLaunchKernelGGL((fa_delta_kernel<HD>), dim3((unsigned)((nrow + 255) / 256)), dim3(256), 0, stream, a.o, a.dout, delta, B, H, Sq, o_sb, o_ss));
    PB_LAUNCH_CHECK();
    PB_REQUIRE(!dbias_q, "pb_flash_bwd: fused bias gradients exist in the pipelined kernels (head_dim 64 / 96 / 128) only");
    dim3 gk((Sk + TK - 1) / TK, H, B), gq((Sq + TQ - 1) / TQ, H, B);
    if (hd == 128) {   // 4 x 16 KiB tiles exceed the default 64 KiB dynamic-LDS limit
        PB_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&fa_bwd_dkv_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 64 * 128 * 2 + 512));
    }
    FA_DISPATCH(hd, hipLaunchKernelGGL((fa_bwd_dkv_kernel<HD>), gk, dim3(FA_THREADS), 4 * 64 * HD * 2 + 512, stream, a));
    PB_LAUNCH_CHECK();
    FA_DISPATCH(hd, hipLaunchKernelGGL((fa_bwd_dq_kernel<HD>), gq, dim3(FA_THREADS), 3 * 64 * HD * 2 + 256, stream, a));
    PB_LAUNCH_CHECK();
    return 0;
}

// ---- packed rows ("varlen"): the batch rows of different lengths lie back to back; see include/pianobart_hip.h
extern "C" int pb_flash_fwd_packed(const void* q, const void* k, const void* v, void* o, float* lse, const int32_t* q_off, const int32_t* q_len,
                                   const int32_t* k_off, const int32_t* k_len, const int32_t* k_vis, int32_t B, int32_t H, int32_t Sq_max,
                                   int32_t Sk_max, int32_t hd, int64_t q_ss, int64_t k_ss, int64_t v_ss, int64_t o_ss, float scale,
                                   int32_t causal, const int32_t* bh_order, void* stream_) {
    if (check_common("pb_flash_fwd_packed", hd, q_ss, k_ss, v_ss, o_ss)) return -2;
    PB_REQUIRE(hd == 64 || hd == 96 || hd == 128, "pb_flash_fwd_packed: head_dim %d (64 / 96 / 128 only)", hd);
    PB_REQUIRE(q_off && q_len && k_off && k_len && k_vis, "pb_flash_fwd_packed: the five row descriptors are required");
    if (B <= 0 || H <= 0 || Sq_max <= 0) return 0;
    const int* vl[5] = {q_off, q_len, k_off, k_len, bh_order};
    return pb_flash64_fwd(q, k, v, o, lse, nullptr, k_vis, B, H, Sq_max, Sk_max, hd, 0, q_ss, 0, k_ss, 0, v_ss, 0, o_ss, scale, causal & 1,
                          (hipStream_t)stream_, vl);
}

extern "C" int pb_flash_bwd_packed(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse, void* dq,
                                   void* dk, void* dv, float* delta, const int32_t* q_off, const int32_t* q_len, const int32_t* k_off,
                                   const int32_t* k_len, const int32_t* k_vis, int32_t B, int32_t H, int32_t Sq_max, int32_t Sk_max, int32_t hd,
                                   int64_t q_ss, int64_t k_ss, int64_t v_ss, int64_t o_ss, int64_t dq_ss, int64_t dk_ss, int64_t dv_ss,
                                   float scale, int32_t causal, float* dbias_q, float* dbias_k, float* dbias_v, float* dbias_ws, const int32_t* bh_order,
                                   void* stream_) {
    if (check_common("pb_flash_bwd_packed", hd, q_ss, k_ss, v_ss, o_ss)) return -2;
    PB_REQUIRE(hd == 64 || hd == 96 || hd == 128, "pb_flash_bwd_packed: head_dim %d (64 / 96 / 128 only)", hd);
    PB_REQUIRE(q_off && q_len && k_off && k_len && k_vis, "pb_flash_bwd_packed: the five row descriptors are required");
    PB_REQUIRE(dq_ss % 4 == 0, "pb_flash_bwd_packed: bad strides");
    if (B <= 0 || H <= 0 || Sq_max <= 0) return 0;
    const int* vl[5] = {q_off, q_len, k_off, k_len, bh_order};
    return pb_flash64_bwd(q, k, v, o, dout, lse, delta, nullptr, k_vis, dq, dk, dv, B, H, Sq_max, Sk_max, hd, 0, q_ss, 0, k_ss, 0, v_ss, 0, o_ss,
                          0, dq_ss, 0, dk_ss, 0, dv_ss, scale, causal & 1, dbias_q, dbias_k, dbias_v, dbias_ws, (hipStream_t)stream_, vl);
}
