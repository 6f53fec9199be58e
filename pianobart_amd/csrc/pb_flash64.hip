// K4, second generation for head_dim = 64 (cfg 2 / cfg 5 head size): same math and the same
// accumulator-as-operand layout trick as pb_flash.hip, restructured for the CDNA4 memory system:
//   * K, V, Q, dO tiles go global -> LDS with global_load_lds_dwordx4 in their NATURAL [row][64] layout
//     (128-B rows, chunk swizzle on the source address); no register staging, no transposed copies;
//   * the operand that must be contracted over the image's ROW index (V for O^T = V^T P^T, dO / Q for
//     dV / dK, K for dQ^T) is read with ds_read_b64_tr_b16 (hardware transpose) from the same image that
//     serves the ds_read_b128 row reads: one swizzle, conflict-free for both kinds of read;
//   * 2 LDS stages: the DMA of tile i+1 is in flight while tile i is being multiplied (one barrier/tile);
//   * every wave owns 32 rows (2 MFMA tiles), halving LDS fragment reads per MFMA.
#include "pb_common.h"
#include "pb_api_internal.h"

namespace {

constexpr int HD = 64, FT = 256;
constexpr float LOG2E = 1.4426950408889634f;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

struct Fa64Args {
    const bf16_t *q, *k, *v, *o, *dout;
    bf16_t *out, *dq, *dk, *dv;
    float* lse; const float* delta; const float* key_mask; const int* kmax;
    int B, H, Sq, Sk;
    long q_sb, q_ss, k_sb, k_ss, v_sb, v_ss, o_sb, o_ss, dq_sb, dq_ss, dk_sb, dk_ss, dv_sb, dv_ss;
    float scale; int causal;
};

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

// image [64 rows][128 B]; f(row): 8 distinct values over (row>>1)&7, even values over an aligned group of 8 rows
__device__ __forceinline__ int fsw(int row) { return (((row >> 1) & 3) << 1) | ((row >> 3) & 1); }

__device__ __forceinline__ void glds16(const bf16_t* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// stage rows r0..r0+63 (clamped to nrows-1) of a [*, 64] bf16 matrix with row stride ld: 8 KiB = 8 DMA pieces, 2 per wave
__device__ __forceinline__ void stage64(const bf16_t* __restrict__ base, long ld, int r0, int nrows, char* lds, int wave, int lane) {
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int inst = wave * 2 + n;
        const int row = inst * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ fsw(row);
        const int gr = min(r0 + row, nrows - 1);
        glds16(base + (long)gr * ld + chunk * 8, lds + inst * 1024);
    }
}
// natural fragment: 8 consecutive columns (32 ks + 8 g ..) of image row `row`
__device__ __forceinline__ bf16x8 frag_row(const char* lds, int row, int ks, int g) {
    return *reinterpret_cast<const bf16x8*>(lds + row * 128 + (((ks * 4 + g) ^ fsw(row)) << 4));
}
// transposed, permuted-k fragment: element j = image[row 32 s + 16 (j>>2) + 4 g + (j&3)][column c0 + (lane&15)]
__device__ __forceinline__ bf16x8 frag_tr(const char* lds, int c0, int s, int lane) {
    const int lr = lane & 15, g = lane >> 4, qq = lr >> 2, pp = lr & 3;
    const int chunk = (c0 >> 3) + (pp >> 1);
    const int r0 = 32 * s + 4 * g + qq, r1 = r0 + 16;
    const int o0 = r0 * 128 + ((chunk ^ fsw(r0)) << 4) + ((pp & 1) << 3);
    const int o1 = r1 * 128 + ((chunk ^ fsw(r1)) << 4) + ((pp & 1) << 3);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + o0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + o1));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ bf16x8 pack_pair(const f32x4& lo, const f32x4& hi) {
    bf16x8 r = {(bf16_t)lo[0], (bf16_t)lo[1], (bf16_t)lo[2], (bf16_t)lo[3], (bf16_t)hi[0], (bf16_t)hi[1], (bf16_t)hi[2], (bf16_t)hi[3]};
    return r;
}
__device__ __forceinline__ bf16x8 frag_global(const bf16_t* __restrict__ g, long ld, int row, int nvalid, int col) {
    bf16x8 z = {};
    if (row < nvalid) z = *reinterpret_cast<const bf16x8*>(g + (long)row * ld + col);
    return z;
}
__device__ __forceinline__ float grp_max(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ float grp_sum(float v) { v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); }

constexpr int STG = 2 * 8192 + 512;        // one stage: two 8 KiB images + 128 floats

// 1-D grid -> (row block, head, batch). Workgroups are dealt round-robin over the 8 XCDs (id % 8): all row blocks of one
// (batch, head) are given to ONE XCD, back to back, so the K/V (or Q/dO) tiles they all stream stay in that XCD's 4 MiB
// L2 (measured before: FETCH_SIZE 705 MB per forward launch = K/V re-fetched from beyond L2 by each of the 8 q-blocks).
__device__ __forceinline__ void block_map(int nrb, int H, int B, int& rb, int& h, int& b) {
    const int L = blockIdx.x, BH = H * B;
    int bh;
    if ((BH & 7) == 0) { const int x = L & 7, slot = L >> 3; bh = (slot / nrb) * 8 + x; rb = slot % nrb; }
    else { bh = L / nrb; rb = L % nrb; }
    h = bh % H; b = bh / H;
}

// ================================================================== forward: block = 128 queries
__global__ __launch_bounds__(FT) void fa64_fwd_kernel(const Fa64Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, lr = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    int rb, h, b;
    block_map((p.Sq + 127) / 128, p.H, p.B, rb, h, b);
    const int q0 = rb * 128;
    const bf16_t* Q = p.q + b * p.q_sb + h * HD;
    const bf16_t* K = p.k + b * p.k_sb + h * HD;
    const bf16_t* V = p.v + b * p.v_sb + h * HD;
    int myq[2];
    bf16x8 qf[2][2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        myq[qt] = q0 + wave * 32 + qt * 16 + lr;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf[qt][ks] = frag_global(Q, p.q_ss, myq[qt], p.Sq, ks * 32 + g * 8);
    }
    f32x4 oacc[2][4];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int i = 0; i < 4; ++i) oacc[qt][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m[2] = {-INFINITY, -INFINITY}, l[2] = {0.f, 0.f};
    const float c = p.scale * LOG2E;
    // keys at and beyond kmax[b] (1 + last visible key of this batch row: the PAD tail) are masked for every query: skip their tiles
    const int kvis_end = p.kmax ? min(p.Sk, p.kmax[b]) : p.Sk;
    const int kend = p.causal ? min(kvis_end, q0 + 128) : kvis_end;
    const int nt = (kend + 63) / 64;
    auto stage = [&](int it, int sidx) {
        char* st = smem + sidx * STG;
        stage64(K, p.k_ss, it * 64, p.Sk, st, wave, lane);
        stage64(V, p.v_ss, it * 64, p.Sk, st + 8192, wave, lane);
        if (t < 64) {
            const int key = it * 64 + t;
            const bool vis = key < p.Sk && (!p.key_mask || p.key_mask[(long)b * p.Sk + key] != 0.f);
            reinterpret_cast<float*>(st + 16384)[t] = vis ? 0.f : -INFINITY;
        }
    };
    if (nt > 0) stage(0, 0);
    __syncthreads();
    for (int it = 0; it < nt; ++it) {
        const char* st = smem + (it & 1) * STG;
        if (it + 1 < nt) stage(it + 1, (it + 1) & 1);
        const char* ldsK = st; const char* ldsV = st + 8192;
        const float* ldsB = reinterpret_cast<const float*>(st + 16384);
        const int k0 = it * 64;
        const bool diag = p.causal && (k0 + 63 > q0 + wave * 32);        // wave-uniform: only tiles that touch the diagonal compare
        f32x4 s[2][4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const bf16x8 ka = frag_row(ldsK, kt * 16 + lr, 0, g), kb = frag_row(ldsK, kt * 16 + lr, 1, g);
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                s[qt][kt] = MFMA16(ka, qf[qt][0], (f32x4{0.f, 0.f, 0.f, 0.f}));
                s[qt][kt] = MFMA16(kb, qf[qt][1], s[qt][kt]);
            }
        }
        bf16x8 pf[2][2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const f32x4 bias = *reinterpret_cast<const f32x4*>(ldsB + kt * 16 + g * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x = fmaf(s[qt][kt][r], c, bias[r]);
                    if (diag && (k0 + kt * 16 + g * 4 + r) > myq[qt]) x = -INFINITY;
                    s[qt][kt][r] = x;
                    mx = fmaxf(mx, x);
                }
            }
            mx = grp_max(mx);
            const float mnew = fmaxf(m[qt], mx);
            const float muse = mnew == -INFINITY ? 0.f : mnew;
            const float alpha = __builtin_amdgcn_exp2f(m[qt] - muse);
            float rs = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float e = __builtin_amdgcn_exp2f(s[qt][kt][r] - muse); s[qt][kt][r] = e; rs += e; }
            rs = grp_sum(rs);
            l[qt] = l[qt] * alpha + rs;
            m[qt] = mnew;
#pragma unroll
            for (int i = 0; i < 4; ++i) oacc[qt][i] *= alpha;
            pf[qt][0] = pack_pair(s[qt][0], s[qt][1]);
            pf[qt][1] = pack_pair(s[qt][2], s[qt][3]);
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const bf16x8 v0 = frag_tr(ldsV, dt * 16, 0, lane), v1 = frag_tr(ldsV, dt * 16, 1, lane);
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                oacc[qt][dt] = MFMA16(v0, pf[qt][0], oacc[qt][dt]);
                oacc[qt][dt] = MFMA16(v1, pf[qt][1], oacc[qt][dt]);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        if (myq[qt] < p.Sq) {
            const float inv = l[qt] > 0.f ? 1.0f / l[qt] : 0.f;
            bf16_t* O = p.out + b * p.o_sb + (long)myq[qt] * p.o_ss + h * HD;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                bf16x4 r = {(bf16_t)(oacc[qt][dt][0] * inv), (bf16_t)(oacc[qt][dt][1] * inv), (bf16_t)(oacc[qt][dt][2] * inv), (bf16_t)(oacc[qt][dt][3] * inv)};
                *reinterpret_cast<bf16x4*>(O + dt * 16 + g * 4) = r;
            }
            if (g == 0) p.lse[((long)b * p.H + h) * p.Sq + myq[qt]] = l[qt] > 0.f ? (m[qt] + log2f(l[qt])) / LOG2E : INFINITY;
        }
    }
}

// ================================================================== backward dK, dV: block = 128 keys
__global__ __launch_bounds__(FT) void fa64_bwd_dkv_kernel(const Fa64Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, lr = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    int rb, h, b;
    block_map((p.Sk + 127) / 128, p.H, p.B, rb, h, b);
    const int k0 = rb * 128;
    const bf16_t* Q = p.q + b * p.q_sb + h * HD;
    const bf16_t* K = p.k + b * p.k_sb + h * HD;
    const bf16_t* V = p.v + b * p.v_sb + h * HD;
    const bf16_t* DO = p.dout + b * p.o_sb + h * HD;
    int mykey[2]; bool kvis[2];
    bf16x8 kf[2][2], vf[2][2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        mykey[kt] = k0 + wave * 32 + kt * 16 + lr;
        kvis[kt] = mykey[kt] < p.Sk && (!p.key_mask || p.key_mask[(long)b * p.Sk + mykey[kt]] != 0.f);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            kf[kt][ks] = frag_global(K, p.k_ss, mykey[kt], p.Sk, ks * 32 + g * 8);
            vf[kt][ks] = frag_global(V, p.v_ss, mykey[kt], p.Sk, ks * 32 + g * 8);
        }
    }
    f32x4 dk[2][4], dv[2][4];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int i = 0; i < 4; ++i) { dk[kt][i] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[kt][i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const float c = p.scale * LOG2E;
    const int it0 = p.causal ? k0 / 64 : 0;
    // a key block that lies entirely in the masked tail receives no gradient: skip its whole query loop (zeros are written)
    const int nt = (p.kmax && k0 >= p.kmax[b]) ? 0 : (p.Sq + 63) / 64;
    auto stage = [&](int it, int sidx) {
        char* st = smem + sidx * STG;
        stage64(Q, p.q_ss, it * 64, p.Sq, st, wave, lane);
        stage64(DO, p.o_ss, it * 64, p.Sq, st + 8192, wave, lane);
        if (t < 64) {
            const int q = it * 64 + t;
            const long li = ((long)b * p.H + h) * p.Sq + q;
            float* f = reinterpret_cast<float*>(st + 16384);
            f[t] = q < p.Sq ? p.lse[li] * LOG2E : INFINITY;
            f[64 + t] = q < p.Sq ? p.delta[li] : 0.f;
        }
    };
    if (it0 < nt) stage(it0, 0);
    __syncthreads();
    for (int it = it0; it < nt; ++it) {
        const int par = (it - it0) & 1;
        const char* st = smem + par * STG;
        if (it + 1 < nt) stage(it + 1, par ^ 1);
        const char* ldsQ = st; const char* ldsO = st + 8192;
        const float* ldsL = reinterpret_cast<const float*>(st + 16384);
        const int q0 = it * 64;
        const bool diag = p.causal && (k0 + wave * 32 + 31 > q0);          // wave-uniform: this q tile can be below some of the wave's keys
        bf16x8 pf[2][2], df[2][2];
#pragma unroll
        for (int half = 0; half < 2; ++half) {            // q tiles (2 half, 2 half + 1) -> one k-step of the dV/dK products
            f32x4 s[2][2], dp[2][2];
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                const int qt = half * 2 + qq;
                const bf16x8 qa = frag_row(ldsQ, qt * 16 + lr, 0, g), qb = frag_row(ldsQ, qt * 16 + lr, 1, g);
                const bf16x8 oa = frag_row(ldsO, qt * 16 + lr, 0, g), ob = frag_row(ldsO, qt * 16 + lr, 1, g);
                const f32x4 lse = *reinterpret_cast<const f32x4*>(ldsL + qt * 16 + g * 4);
                const f32x4 dl = *reinterpret_cast<const f32x4*>(ldsL + 64 + qt * 16 + g * 4);
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    f32x4 sv = MFMA16(qa, kf[kt][0], (f32x4{0.f, 0.f, 0.f, 0.f}));
                    sv = MFMA16(qb, kf[kt][1], sv);
                    f32x4 dv_ = MFMA16(oa, vf[kt][0], (f32x4{0.f, 0.f, 0.f, 0.f}));
                    dv_ = MFMA16(ob, vf[kt][1], dv_);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int q = q0 + qt * 16 + g * 4 + r;
                        const bool vis = kvis[kt] && (!diag || mykey[kt] <= q);
                        const float pr = vis ? __builtin_amdgcn_exp2f(fmaf(sv[r], c, -lse[r])) : 0.f;
                        sv[r] = pr;
                        dv_[r] = pr * (dv_[r] - dl[r]) * p.scale;
                    }
                    s[kt][qq] = sv; dp[kt][qq] = dv_;
                }
            }
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) { pf[kt][half] = pack_pair(s[kt][0], s[kt][1]); df[kt][half] = pack_pair(dp[kt][0], dp[kt][1]); }
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
            for (int sidx = 0; sidx < 2; ++sidx) {
                const bf16x8 ot = frag_tr(ldsO, dt * 16, sidx, lane), qtf = frag_tr(ldsQ, dt * 16, sidx, lane);
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    dv[kt][dt] = MFMA16(pf[kt][sidx], ot, dv[kt][dt]);
                    dk[kt][dt] = MFMA16(df[kt][sidx], qtf, dk[kt][dt]);
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = k0 + wave * 32 + kt * 16 + g * 4 + r;
            if (key < p.Sk) {
                bf16_t* DK = p.dk + b * p.dk_sb + (long)key * p.dk_ss + h * HD;
                bf16_t* DV = p.dv + b * p.dv_sb + (long)key * p.dv_ss + h * HD;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) { DK[dt * 16 + lr] = (bf16_t)dk[kt][dt][r]; DV[dt * 16 + lr] = (bf16_t)dv[kt][dt][r]; }
            }
        }
}

// ================================================================== backward dQ: block = 128 queries
__global__ __launch_bounds__(FT) void fa64_bwd_dq_kernel(const Fa64Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, lr = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    int rb, h, b;
    block_map((p.Sq + 127) / 128, p.H, p.B, rb, h, b);
    const int q0 = rb * 128;
    const bf16_t* Q = p.q + b * p.q_sb + h * HD;
    const bf16_t* K = p.k + b * p.k_sb + h * HD;
    const bf16_t* V = p.v + b * p.v_sb + h * HD;
    const bf16_t* DO = p.dout + b * p.o_sb + h * HD;
    int myq[2]; float lse[2], dl[2];
    bf16x8 qf[2][2], of[2][2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        myq[qt] = q0 + wave * 32 + qt * 16 + lr;
        const long li = ((long)b * p.H + h) * p.Sq + myq[qt];
        lse[qt] = myq[qt] < p.Sq ? p.lse[li] * LOG2E : INFINITY;
        dl[qt] = myq[qt] < p.Sq ? p.delta[li] : 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            qf[qt][ks] = frag_global(Q, p.q_ss, myq[qt], p.Sq, ks * 32 + g * 8);
            of[qt][ks] = frag_global(DO, p.o_ss, myq[qt], p.Sq, ks * 32 + g * 8);
        }
    }
    f32x4 dq[2][4];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int i = 0; i < 4; ++i) dq[qt][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float c = p.scale * LOG2E;
    const int kvis_end = p.kmax ? min(p.Sk, p.kmax[b]) : p.Sk;
    const int kend = p.causal ? min(kvis_end, q0 + 128) : kvis_end;
    const int nt = (kend + 63) / 64;
    auto stage = [&](int it, int sidx) {
        char* st = smem + sidx * STG;
        stage64(K, p.k_ss, it * 64, p.Sk, st, wave, lane);
        stage64(V, p.v_ss, it * 64, p.Sk, st + 8192, wave, lane);
        if (t < 64) {
            const int key = it * 64 + t;
            reinterpret_cast<float*>(st + 16384)[t] = (key < p.Sk && (!p.key_mask || p.key_mask[(long)b * p.Sk + key] != 0.f)) ? 1.f : 0.f;
        }
    };
    if (nt > 0) stage(0, 0);
    __syncthreads();
    for (int it = 0; it < nt; ++it) {
        const char* st = smem + (it & 1) * STG;
        if (it + 1 < nt) stage(it + 1, (it + 1) & 1);
        const char* ldsK = st; const char* ldsV = st + 8192;
        const float* ldsB = reinterpret_cast<const float*>(st + 16384);
        const int k0 = it * 64;
        const bool diag = p.causal && (k0 + 63 > q0 + wave * 32);
        bf16x8 df[2][2];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x4 ds_[2][2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int kt = half * 2 + kk;
                const bf16x8 ka = frag_row(ldsK, kt * 16 + lr, 0, g), kb = frag_row(ldsK, kt * 16 + lr, 1, g);
                const bf16x8 va = frag_row(ldsV, kt * 16 + lr, 0, g), vb = frag_row(ldsV, kt * 16 + lr, 1, g);
                const f32x4 vis4 = *reinterpret_cast<const f32x4*>(ldsB + kt * 16 + g * 4);
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) {
                    f32x4 sv = MFMA16(ka, qf[qt][0], (f32x4{0.f, 0.f, 0.f, 0.f}));
                    sv = MFMA16(kb, qf[qt][1], sv);
                    f32x4 dpv = MFMA16(va, of[qt][0], (f32x4{0.f, 0.f, 0.f, 0.f}));
                    dpv = MFMA16(vb, of[qt][1], dpv);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = k0 + kt * 16 + g * 4 + r;
                        const bool vis = vis4[r] != 0.f && (!diag || key <= myq[qt]);
                        const float pr = vis ? __builtin_amdgcn_exp2f(fmaf(sv[r], c, -lse[qt])) : 0.f;
                        dpv[r] = pr * (dpv[r] - dl[qt]) * p.scale;
                    }
                    ds_[qt][kk] = dpv;
                }
            }
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) df[qt][half] = pack_pair(ds_[qt][0], ds_[qt][1]);
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const bf16x8 k0f = frag_tr(ldsK, dt * 16, 0, lane), k1f = frag_tr(ldsK, dt * 16, 1, lane);
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                dq[qt][dt] = MFMA16(k0f, df[qt][0], dq[qt][dt]);
                dq[qt][dt] = MFMA16(k1f, df[qt][1], dq[qt][dt]);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
        if (myq[qt] < p.Sq) {
            bf16_t* DQ = p.dq + b * p.dq_sb + (long)myq[qt] * p.dq_ss + h * HD;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                bf16x4 r = {(bf16_t)dq[qt][dt][0], (bf16_t)dq[qt][dt][1], (bf16_t)dq[qt][dt][2], (bf16_t)dq[qt][dt][3]};
                *reinterpret_cast<bf16x4*>(DQ + dt * 16 + g * 4) = r;
            }
        }
}

}  // namespace

// entry points used by pb_flash.hip's dispatch (same argument meaning as pb_flash_fwd / pb_flash_bwd, hd == 64)
int pb_flash64_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const float* key_mask, const int* kmax, int B, int H, int Sq, int Sk,
                   long q_sb, long q_ss, long k_sb, long k_ss, long v_sb, long v_ss, long o_sb, long o_ss, float scale, int causal, hipStream_t stream) {
    Fa64Args a = {};
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.out = (bf16_t*)o; a.lse = lse; a.key_mask = key_mask; a.kmax = kmax;
    a.B = B; a.H = H; a.Sq = Sq; a.Sk = Sk; a.q_sb = q_sb; a.q_ss = q_ss; a.k_sb = k_sb; a.k_ss = k_ss; a.v_sb = v_sb; a.v_ss = v_ss;
    a.o_sb = o_sb; a.o_ss = o_ss; a.scale = scale; a.causal = causal;
    hipLaunchKernelGGL(fa64_fwd_kernel, dim3(((Sq + 127) / 128) * H * B), dim3(FT), 2 * STG, stream, a);
    PB_LAUNCH_CHECK();
    return 0;
}

int pb_flash64_bwd(const void* q, const void* k, const void* v, const void* dout, const float* lse, const float* delta, const float* key_mask,
                   const int* kmax, void* dq, void* dk, void* dv, int B, int H, int Sq, int Sk, long q_sb, long q_ss, long k_sb, long k_ss, long v_sb,
                   long v_ss, long o_sb, long o_ss, long dq_sb, long dq_ss, long dk_sb, long dk_ss, long dv_sb, long dv_ss, float scale,
                   int causal, hipStream_t stream) {
    Fa64Args a = {};
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.dout = (const bf16_t*)dout;
    a.dq = (bf16_t*)dq; a.dk = (bf16_t*)dk; a.dv = (bf16_t*)dv; a.lse = const_cast<float*>(lse); a.delta = delta; a.key_mask = key_mask; a.kmax = kmax;
    a.B = B; a.H = H; a.Sq = Sq; a.Sk = Sk; a.q_sb = q_sb; a.q_ss = q_ss; a.k_sb = k_sb; a.k_ss = k_ss; a.v_sb = v_sb; a.v_ss = v_ss;
    a.o_sb = o_sb; a.o_ss = o_ss; a.dq_sb = dq_sb; a.dq_ss = dq_ss; a.dk_sb = dk_sb; a.dk_ss = dk_ss; a.dv_sb = dv_sb; a.dv_ss = dv_ss;
    a.scale = scale; a.causal = causal;
    hipLaunchKernelGGL(fa64_bwd_dkv_kernel, dim3(((Sk + 127) / 128) * H * B), dim3(FT), 2 * STG, stream, a);
    PB_LAUNCH_CHECK();
    hipLaunchKernelGGL(fa64_bwd_dq_kernel, dim3(((Sq + 127) / 128) * H * B), dim3(FT), 2 * STG, stream, a);
    PB_LAUNCH_CHECK();
    return 0;
}
