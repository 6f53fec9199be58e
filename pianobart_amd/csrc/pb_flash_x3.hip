// Fused attention of the "bf16x3" parity instantiation (round 6): f32 q / k / v / o in HBM, every product on the bf16 matrix cores
// as a split-bf16 triple (a_hi b_hi + a_hi b_lo + a_lo b_hi, f32 accumulation; pb_gemm_x3.hip has the arithmetic), softmax in f32.
// Replaces, for that instantiation, the unfused QK^T -> masked softmax -> PV chain of the exact-f32 path (modeling_bart.py:115-140
// as used by PianoBart.py:76) -- whose (B, H, S, S) f32 score / probability matrices cost more HBM time than all its GEMMs -- with
// the tile structure of pb_flash.hip:
//   forward  (block = 64 queries, wave = 16 queries; loop over 64-key tiles):  S^T = K Q^T, online softmax, O^T += V^T P^T
//   backward = two kernels, no atomics:  dKV (block = 64 keys; loop over query tiles)  and  dQ (block = 64 queries; loop over key tiles)
// What differs from pb_flash.hip: tiles are read as f32 and cut into a hi and a lo bf16 image in LDS while they are staged; register
// operands (Q rows, K / V rows, dO rows) are cut when loaded; an accumulator pair that becomes the next MFMA's operand (P, dS) is cut
// into (hi, lo) in registers; every MFMA of pb_flash.hip is three. Outputs (O, dQ, dK, dV) and lse / delta are f32.
// Masks and conventions are pb_flash.hip's: key padding (per-batch float row, != 0 keeps), optional causal, a query without a visible
// key gives a zero output row and lse = +inf. head_dim 32 / 64 / 128; row strides multiples of 4 elements.
#include "pb_common.h"
#include "pb_api_internal.h"

namespace {

constexpr int FX_THREADS = 256;
constexpr int XQ = 64, XK = 64;
constexpr float XLOG2E = 1.4426950408889634f;

template <int HD> __device__ __forceinline__ int xrows_off(int row, int chunk) {
    constexpr int RB = HD * 2, NCH = HD / 8, RPB = 256 / RB;
    return row * RB + (((chunk ^ ((row / RPB) % NCH))) << 4);
}
__device__ __forceinline__ int xtswz(int row) { return ((row >> 1) & 7) ^ ((row >> 4) & 7); }
__device__ __forceinline__ int xtr_off(int row, int k) { return row * 128 + ((((k >> 3)) ^ xtswz(row)) << 4) + (((k >> 2) & 1) << 3); }

__device__ __forceinline__ void cut8(const float* x, bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { const bf16_t h = (bf16_t)x[j]; hi[j] = h; lo[j] = (bf16_t)(x[j] - (float)h); }
}
__device__ __forceinline__ void load8f(const float* __restrict__ p, float* x) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { x[j] = a[j]; x[4 + j] = b[j]; }
}

// [64][HD] f32 tile (rows >= nvalid read as zero) -> hi / lo rows images
template <int HD>
__device__ __forceinline__ void xstage_rows(char* hi_img, char* lo_img, const float* __restrict__ g, long ld, int nvalid, int t) {
    constexpr int NCH = HD / 8;
    for (int v = t; v < 64 * NCH; v += FX_THREADS) {
        const int row = v / NCH, ch = v % NCH;
        float x[8];
        if (row < nvalid) load8f(g + (long)row * ld + ch * 8, x);
        else {
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = 0.f;
        }
        bf16x8 h, l;
        cut8(x, h, l);
        *reinterpret_cast<bf16x8*>(hi_img + xrows_off<HD>(row, ch)) = h;
        *reinterpret_cast<bf16x8*>(lo_img + xrows_off<HD>(row, ch)) = l;
    }
}
// the same tile transposed: thread owns 4 (source rows) x 8 (columns), transposed with v_perm, once for hi and once for lo
template <int HD>
__device__ __forceinline__ void xstage_transposed(char* hi_img, char* lo_img, const float* __restrict__ g, long ld, int nvalid, int t) {
    constexpr int NCC = HD / 8;
    for (int blk = t; blk < 16 * NCC; blk += FX_THREADS) {
        const int cc = blk % NCC, kg = blk / NCC;
        bf16x8 vh[4], vl[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = kg * 4 + i;
            float x[8];
            if (row < nvalid) load8f(g + (long)row * ld + cc * 8, x);
            else {
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = 0.f;
            }
            cut8(x, vh[i], vl[i]);
        }
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            char* img = part ? lo_img : hi_img;
            uint4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = __builtin_bit_cast(uint4, part ? vl[i] : vh[i]);
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
                *reinterpret_cast<uint2*>(img + xtr_off(cc * 8 + j, kg * 4)) = make_uint2(lo, hi);
            }
        }
    }
}
// rows image AND transposed image of one tile from ONE pass over the f32 source (the dKV / dQ kernels need both of Q, dO / K): the
// thread that owns a 4 x 8 block for the transpose also writes its four row chunks
template <int HD>
__device__ __forceinline__ void xstage_both(char* r_hi, char* r_lo, char* t_hi, char* t_lo, const float* __restrict__ g, long ld, int nvalid, int t) {
    constexpr int NCC = HD / 8;
    for (int blk = t; blk < 16 * NCC; blk += FX_THREADS) {
        const int cc = blk % NCC, kg = blk / NCC;
        bf16x8 vh[4], vl[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = kg * 4 + i;
            float x[8];
            if (row < nvalid) load8f(g + (long)row * ld + cc * 8, x);
            else {
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = 0.f;
            }
            cut8(x, vh[i], vl[i]);
            *reinterpret_cast<bf16x8*>(r_hi + xrows_off<HD>(row, cc)) = vh[i];
            *reinterpret_cast<bf16x8*>(r_lo + xrows_off<HD>(row, cc)) = vl[i];
        }
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            char* img = part ? t_lo : t_hi;
            uint4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = __builtin_bit_cast(uint4, part ? vl[i] : vh[i]);
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
                *reinterpret_cast<uint2*>(img + xtr_off(cc * 8 + j, kg * 4)) = make_uint2(lo, hi);
            }
        }
    }
}
template <int HD> __device__ __forceinline__ bf16x8 xfrag_rows(const char* lds, int row, int ks, int g) {
    return *reinterpret_cast<const bf16x8*>(lds + xrows_off<HD>(row, ks * 4 + g));
}
__device__ __forceinline__ bf16x8 xfrag_perm(const char* lds, int row, int s, int g) {
    const bf16x4 a = *reinterpret_cast<const bf16x4*>(lds + xtr_off(row, 32 * s + 4 * g));
    const bf16x4 b = *reinterpret_cast<const bf16x4*>(lds + xtr_off(row, 32 * s + 16 + 4 * g));
    bf16x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return r;
}
__device__ __forceinline__ void xpack_pair(const f32x4& a, const f32x4& b, bf16x8& hi, bf16x8& lo) {
    float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    cut8(x, hi, lo);
}
__device__ __forceinline__ void xfrag_global(const float* __restrict__ g, long ld, int row, int nvalid, int col, bf16x8& hi, bf16x8& lo) {
    float x[8];
    if (row < nvalid) load8f(g + (long)row * ld + col, x);
    else {
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = 0.f;
    }
    cut8(x, hi, lo);
}

#define XM(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
// c += a b with a = ah + al, b = bh + bl, the al bl term dropped; the two small terms first
#define XM3(ah, al, bh, bl, c) XM((ah), (bh), XM((ah), (bl), XM((al), (bh), (c))))

struct FxArgs {
    const float *q, *k, *v, *o, *dout;
    float *out, *dq, *dk, *dv;
    float* lse; const float* delta; const float* key_mask;
    const int* kmax;                                          // per batch row: 1 + last visible key (keys at and behind it are masked for every query), or NULL
    const int *vq_off, *vq_len, *vk_off, *vk_len, *vk_vis;    // packed rows (all NULL, or all set): batch row b owns query rows vq_off[b] .. + vq_len[b] - 1 and key rows
                                                              // vk_off[b] .. + vk_len[b] - 1 of ONE (rows, H HD) tensor, its first vk_vis[b] keys visible; Sq / Sk are the maxima
    int B, H, Sq, Sk;
    long q_sb, q_ss, k_sb, k_ss, v_sb, v_ss, o_sb, o_ss, dq_sb, dq_ss, dk_sb, dk_ss, dv_sb, dv_ss;
    float scale; int causal;
};
// the view of batch row b: dense (batch strides, the call's Sq / Sk, key mask) or packed (row offsets, this row's own lengths, visible prefix)
struct XView { long qrow, krow; int Sq, Sk, kvis; bool packed; };
__device__ __forceinline__ XView xview(const FxArgs& p, int b) {
    XView v;
    v.packed = p.vq_off != nullptr;
    if (v.packed) { v.qrow = p.vq_off[b]; v.krow = p.vk_off[b]; v.Sq = p.vq_len[b]; v.Sk = p.vk_len[b]; v.kvis = min(p.vk_vis[b], v.Sk); }
    else { v.qrow = 0; v.krow = 0; v.Sq = p.Sq; v.Sk = p.Sk; v.kvis = p.kmax ? min(p.kmax[b], p.Sk) : p.Sk; }
    return v;
}
// dS = p (dP - delta) scale with the operation order pinned (subtract, multiply, multiply): left to itself the compiler forms different multiply-adds in the one- and the
// two-tile kernels (seen at head_dim 32: dQ differed in the last bits), and the two forms of a kernel must round alike
__device__ __forceinline__ float xds(float pr, float dp, float dl, float scale) {
    float t = dp - dl;
    asm volatile("" : "+v"(t));
    t = pr * t;
    asm volatile("" : "+v"(t));
    t = t * scale;
    asm volatile("" : "+v"(t));                 // ... and the product is ROUNDED before it is cut: fused into the cut's subtraction (x - hi) it changes which way lo rounds
    return t;
}
__device__ __forceinline__ float xgrp_max(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ float xgrp_sum(float v) { v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); }

// ================================================================== forward
template <int HD>
__global__ __launch_bounds__(FX_THREADS) void fx_fwd_kernel(const FxArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TB = 64 * HD * 2;
    char* Kh = smem; char* Kl = smem + TB;                  // rows images [64 keys][HD]
    char* Vh = smem + 2 * TB; char* Vl = smem + 3 * TB;     // transposed [HD][64 keys]
    float* ldsB = reinterpret_cast<float*>(smem + 4 * TB);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, lr = lane & 15, g = lane >> 4;
    int rb_, h, b;
    grid_map3(rb_, h, b);
    const int q0 = rb_ * XQ;
    const XView w = xview(p, b);
    if (q0 >= w.Sq) return;                                    // packed rows: a shorter sequence than the longest one
    const float* Q = p.q + b * p.q_sb + w.qrow * p.q_ss + h * HD;
    const float* K = p.k + b * p.k_sb + w.krow * p.k_ss + h * HD;
    const float* V = p.v + b * p.v_sb + w.krow * p.v_ss + h * HD;
    const int myq = q0 + wave * 16 + lr;
    constexpr int KS = HD / 32, DT = HD / 16;
    bf16x8 qh[KS], ql[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xfrag_global(Q, p.q_ss, myq, w.Sq, ks * 32 + g * 8, qh[ks], ql[ks]);
    f32x4 oacc[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i) oacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, l = 0.f;
    const float c = p.scale * XLOG2E;
    const int kend = min(p.causal ? min(w.Sk, q0 + XQ) : w.Sk, w.kvis);     // key tiles behind the last visible key contribute exact zeros: skipped
    for (int k0 = 0; k0 < kend; k0 += XK) {
        __syncthreads();
        xstage_rows<HD>(Kh, Kl, K + (long)k0 * p.k_ss, p.k_ss, w.Sk - k0, t);
        xstage_transposed<HD>(Vh, Vl, V + (long)k0 * p.v_ss, p.v_ss, w.Sk - k0, t);
        if (t < XK) {
            const int key = k0 + t;
            const bool vis = key < w.kvis && (w.packed || !p.key_mask || p.key_mask[(long)b * p.Sk + key] != 0.f);
            ldsB[t] = vis ? 0.f : -INFINITY;
        }
        __syncthreads();
        f32x4 s[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                s[kt] = XM3(xfrag_rows<HD>(Kh, kt * 16 + lr, ks, g), xfrag_rows<HD>(Kl, kt * 16 + lr, ks, g), qh[ks], ql[ks], s[kt]);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const f32x4 bias = *reinterpret_cast<const f32x4*>(ldsB + kt * 16 + g * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = __builtin_fmaf(s[kt][r], c, bias[r]);            // explicit: the one- and two-tile kernels must round alike whatever the compiler would contract
                if (p.causal && (k0 + kt * 16 + g * 4 + r) > myq) x = -INFINITY;
                s[kt][r] = x;
                mx = fmaxf(mx, x);
            }
        }
        mx = xgrp_max(mx);
        const float mnew = fmaxf(m, mx);
        const float muse = mnew == -INFINITY ? 0.f : mnew;
        const float alpha = __builtin_amdgcn_exp2f(m - muse);
        float rs = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float e = __builtin_amdgcn_exp2f(s[kt][r] - muse); s[kt][r] = e; rs += e; }
        rs = xgrp_sum(rs);
        l = __builtin_fmaf(l, alpha, rs);
        m = mnew;
#pragma unroll
        for (int i = 0; i < DT; ++i) oacc[i] *= alpha;
        bf16x8 p0h, p0l, p1h, p1l;
        xpack_pair(s[0], s[1], p0h, p0l);
        xpack_pair(s[2], s[3], p1h, p1l);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            oacc[dt] = XM3(xfrag_perm(Vh, dt * 16 + lr, 0, g), xfrag_perm(Vl, dt * 16 + lr, 0, g), p0h, p0l, oacc[dt]);
            oacc[dt] = XM3(xfrag_perm(Vh, dt * 16 + lr, 1, g), xfrag_perm(Vl, dt * 16 + lr, 1, g), p1h, p1l, oacc[dt]);
        }
    }
    if (myq < w.Sq) {
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        float* O = p.out + b * p.o_sb + (w.qrow + myq) * p.o_ss + h * HD;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) *reinterpret_cast<f32x4*>(O + dt * 16 + g * 4) = oacc[dt] * inv;
        if (g == 0) p.lse[((long)b * p.H + h) * p.Sq + myq] = l > 0.f ? (m + log2f(l)) / XLOG2E : INFINITY;
    }
}

// ================================================================== delta = rowsum(dO * O), f32
template <int HD>
__global__ void fx_delta_kernel(const float* __restrict__ o, const float* __restrict__ dout, float* __restrict__ delta, int B, int H, int Sq, long o_sb, long o_ss,
                                const int* __restrict__ q_off, const int* __restrict__ q_len) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * H * Sq) return;
    const int q = (int)(idx % Sq), h = (int)((idx / Sq) % H), b = (int)(idx / ((long)Sq * H));
    if (q_off && q >= q_len[b]) { delta[idx] = 0.f; return; }
    const long row = q_off ? (long)q_off[b] + q : q;
    const float* op = o + b * o_sb + row * o_ss + h * HD;
    const float* dp = dout + b * o_sb + row * o_ss + h * HD;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < HD; c += 4) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(op + c), d = *reinterpret_cast<const f32x4*>(dp + c);
        s += a[0] * d[0] + a[1] * d[1] + a[2] * d[2] + a[3] * d[3];
    }
    delta[idx] = s;
}

// ================================================================== backward: dK, dV
template <int HD>
__global__ __launch_bounds__(FX_THREADS) void fx_bwd_dkv_kernel(const FxArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TB = 64 * HD * 2;
    char* Qh = smem;            char* Ql = smem + TB;            // rows images [64 q][HD]
    char* QTh = smem + 2 * TB;  char* QTl = smem + 3 * TB;       // transposed [HD][64 q]
    char* Oh = smem + 4 * TB;   char* Ol = smem + 5 * TB;        // dO rows
    char* OTh = smem + 6 * TB;  char* OTl = smem + 7 * TB;       // dO transposed
    float* ldsL = reinterpret_cast<float*>(smem + 8 * TB);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, lr = lane & 15, g = lane >> 4;
    int rb_, h, b;
    grid_map3(rb_, h, b);
    const int k0 = rb_ * XK;
    const XView w = xview(p, b);
    if (k0 >= w.Sk) return;                                    // packed rows: a shorter sequence than the longest one
    const float* Q = p.q + b * p.q_sb + w.qrow * p.q_ss + h * HD;
    const float* K = p.k + b * p.k_sb + w.krow * p.k_ss + h * HD;
    const float* V = p.v + b * p.v_sb + w.krow * p.v_ss + h * HD;
    const float* DO = p.dout + b * p.o_sb + w.qrow * p.o_ss + h * HD;
    float* DKb = p.dk + b * p.dk_sb + w.krow * p.dk_ss + h * HD;
    float* DVb = p.dv + b * p.dv_sb + w.krow * p.dv_ss + h * HD;
    constexpr int KS = HD / 32, DT = HD / 16;
    const int mykey = k0 + wave * 16 + lr;
    bf16x8 kh[KS], kl[KS], vh[KS], vl[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        xfrag_global(K, p.k_ss, mykey, w.Sk, ks * 32 + g * 8, kh[ks], kl[ks]);
        xfrag_global(V, p.v_ss, mykey, w.Sk, ks * 32 + g * 8, vh[ks], vl[ks]);
    }
    const bool kvis = mykey < w.kvis && (w.packed || !p.key_mask || p.key_mask[(long)b * p.Sk + mykey] != 0.f);
    if (k0 >= w.kvis) {                                        // no visible key in this block: its keys receive zero gradient
        for (int i = t; i < XK * (HD / 4); i += FX_THREADS) {
            const int key = k0 + i / (HD / 4), c4 = (i % (HD / 4)) * 4;
            if (key < w.Sk) {
                *reinterpret_cast<f32x4*>(DKb + (long)key * p.dk_ss + c4) = f32x4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4*>(DVb + (long)key * p.dv_ss + c4) = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        return;
    }
    f32x4 dk[DT], dv[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i) { dk[i] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const float c = p.scale * XLOG2E;
    const int qstart = p.causal ? (k0 / XQ) * XQ : 0;
    for (int q0 = qstart; q0 < w.Sq; q0 += XQ) {
        __syncthreads();
        constexpr int NB = 16 * (HD / 8);                        // 4 x 8 blocks per tile: head_dim <= 64 stages Q and dO side by side
        if (2 * NB <= FX_THREADS) {
            if (t < NB) xstage_both<HD>(Qh, Ql, QTh, QTl, Q + (long)q0 * p.q_ss, p.q_ss, w.Sq - q0, t);
            else if (t < 2 * NB) xstage_both<HD>(Oh, Ol, OTh, OTl, DO + (long)q0 * p.o_ss, p.o_ss, w.Sq - q0, t - NB);
        } else {
            xstage_both<HD>(Qh, Ql, QTh, QTl, Q + (long)q0 * p.q_ss, p.q_ss, w.Sq - q0, t);
            xstage_both<HD>(Oh, Ol, OTh, OTl, DO + (long)q0 * p.o_ss, p.o_ss, w.Sq - q0, t);
        }
        if (t < XQ) {
            const int q = q0 + t;
            const long li = ((long)b * p.H + h) * p.Sq + q;
            ldsL[t] = q < w.Sq ? p.lse[li] * XLOG2E : INFINITY;
            ldsL[64 + t] = q < w.Sq ? p.delta[li] : 0.f;
        }
        __syncthreads();
        f32x4 s[4], dp[4];
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
            s[qt] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                s[qt] = XM3(xfrag_rows<HD>(Qh, qt * 16 + lr, ks, g), xfrag_rows<HD>(Ql, qt * 16 + lr, ks, g), kh[ks], kl[ks], s[qt]);
                dp[qt] = XM3(xfrag_rows<HD>(Oh, qt * 16 + lr, ks, g), xfrag_rows<HD>(Ol, qt * 16 + lr, ks, g), vh[ks], vl[ks], dp[qt]);
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
                const float pr = vis ? __builtin_amdgcn_exp2f(__builtin_fmaf(s[qt][r], c, -lse[r])) : 0.f;
                s[qt][r] = pr;
                dp[qt][r] = xds(pr, dp[qt][r], dl[r], p.scale);
            }
        }
        bf16x8 p0h, p0l, p1h, p1l, d0h, d0l, d1h, d1l;
        xpack_pair(s[0], s[1], p0h, p0l); xpack_pair(s[2], s[3], p1h, p1l);
        xpack_pair(dp[0], dp[1], d0h, d0l); xpack_pair(dp[2], dp[3], d1h, d1l);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            dv[dt] = XM3(p0h, p0l, xfrag_perm(OTh, dt * 16 + lr, 0, g), xfrag_perm(OTl, dt * 16 + lr, 0, g), dv[dt]);
            dv[dt] = XM3(p1h, p1l, xfrag_perm(OTh, dt * 16 + lr, 1, g), xfrag_perm(OTl, dt * 16 + lr, 1, g), dv[dt]);
            dk[dt] = XM3(d0h, d0l, xfrag_perm(QTh, dt * 16 + lr, 0, g), xfrag_perm(QTl, dt * 16 + lr, 0, g), dk[dt]);
            dk[dt] = XM3(d1h, d1l, xfrag_perm(QTh, dt * 16 + lr, 1, g), xfrag_perm(QTl, dt * 16 + lr, 1, g), dk[dt]);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int key = k0 + wave * 16 + g * 4 + r;
        if (key < w.Sk) {
            float* DK = DKb + (long)key * p.dk_ss;
            float* DV = DVb + (long)key * p.dv_ss;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) { DK[dt * 16 + lr] = dk[dt][r]; DV[dt * 16 + lr] = dv[dt][r]; }
        }
    }
}

// ================================================================== backward: dQ
template <int HD>
__global__ __launch_bounds__(FX_THREADS) void fx_bwd_dq_kernel(const FxArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TB = 64 * HD * 2;
    char* Kh = smem;           char* Kl = smem + TB;
    char* KTh = smem + 2 * TB; char* KTl = smem + 3 * TB;
    char* Vh = smem + 4 * TB;  char* Vl = smem + 5 * TB;
    float* ldsB = reinterpret_cast<float*>(smem + 6 * TB);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, lr = lane & 15, g = lane >> 4;
    int rb_, h, b;
    grid_map3(rb_, h, b);
    const int q0 = rb_ * XQ;
    const XView w = xview(p, b);
    if (q0 >= w.Sq) return;                                    // packed rows: a shorter sequence than the longest one
    const float* Q = p.q + b * p.q_sb + w.qrow * p.q_ss + h * HD;
    const float* K = p.k + b * p.k_sb + w.krow * p.k_ss + h * HD;
    const float* V = p.v + b * p.v_sb + w.krow * p.v_ss + h * HD;
    const float* DO = p.dout + b * p.o_sb + w.qrow * p.o_ss + h * HD;
    constexpr int KS = HD / 32, DT = HD / 16;
    const int myq = q0 + wave * 16 + lr;
    bf16x8 qh[KS], ql[KS], oh[KS], ol[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        xfrag_global(Q, p.q_ss, myq, w.Sq, ks * 32 + g * 8, qh[ks], ql[ks]);
        xfrag_global(DO, p.o_ss, myq, w.Sq, ks * 32 + g * 8, oh[ks], ol[ks]);
    }
    const long li = ((long)b * p.H + h) * p.Sq + myq;
    const float lse = myq < w.Sq ? p.lse[li] * XLOG2E : INFINITY;
    const float dl = myq < w.Sq ? p.delta[li] : 0.f;
    f32x4 dq[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i) dq[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float c = p.scale * XLOG2E;
    const int kend = min(p.causal ? min(w.Sk, q0 + XQ) : w.Sk, w.kvis);     // key tiles behind the last visible key contribute exact zeros: skipped
    for (int k0 = 0; k0 < kend; k0 += XK) {
        __syncthreads();
        xstage_both<HD>(Kh, Kl, KTh, KTl, K + (long)k0 * p.k_ss, p.k_ss, w.Sk - k0, t);
        xstage_rows<HD>(Vh, Vl, V + (long)k0 * p.v_ss, p.v_ss, w.Sk - k0, t);
        if (t < XK) {
            const int key = k0 + t;
            ldsB[t] = (key < w.kvis && (w.packed || !p.key_mask || p.key_mask[(long)b * p.Sk + key] != 0.f)) ? 1.f : 0.f;
        }
        __syncthreads();
        f32x4 s[4], dp[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            s[kt] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                s[kt] = XM3(xfrag_rows<HD>(Kh, kt * 16 + lr, ks, g), xfrag_rows<HD>(Kl, kt * 16 + lr, ks, g), qh[ks], ql[ks], s[kt]);
                dp[kt] = XM3(xfrag_rows<HD>(Vh, kt * 16 + lr, ks, g), xfrag_rows<HD>(Vl, kt * 16 + lr, ks, g), oh[ks], ol[ks], dp[kt]);
            }
        }
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const f32x4 vis4 = *reinterpret_cast<const f32x4*>(ldsB + kt * 16 + g * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = k0 + kt * 16 + g * 4 + r;
                const bool vis = vis4[r] != 0.f && (!p.causal || key <= myq);
                const float pr = vis ? __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][r], c, -lse)) : 0.f;
                dp[kt][r] = xds(pr, dp[kt][r], dl, p.scale);
            }
        }
        bf16x8 d0h, d0l, d1h, d1l;
        xpack_pair(dp[0], dp[1], d0h, d0l); xpack_pair(dp[2], dp[3], d1h, d1l);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            dq[dt] = XM3(xfrag_perm(KTh, dt * 16 + lr, 0, g), xfrag_perm(KTl, dt * 16 + lr, 0, g), d0h, d0l, dq[dt]);
            dq[dt] = XM3(xfrag_perm(KTh, dt * 16 + lr, 1, g), xfrag_perm(KTl, dt * 16 + lr, 1, g), d1h, d1l, dq[dt]);
        }
    }
    if (myq < w.Sq) {
        float* DQ = p.dq + b * p.dq_sb + (w.qrow + myq) * p.dq_ss + h * HD;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) *reinterpret_cast<f32x4*>(DQ + dt * 16 + g * 4) = dq[dt];
    }
}

// ================================================================== forward and dQ with NT = 2 tiles of 16 queries per wave (round 6, late)
// With one tile per wave the staged K / V fragments are read from LDS once per 16 queries, and the LDS port, not the matrix pipe, sets the pace (32 KiB of fragment
// reads per wave and key tile against 48 MFMAs); two tiles per wave -- blocks of 128 queries -- use every fragment twice: forward 330 -> 277 us, backward 1 108 -> 995 us at
// B = 16 (tools/flash_x3_bench.py). Taken for head_dim <= 64 when blocks of 128 queries still fill the chip twice over (fx_nt); a query's arithmetic is the same in both forms
// (test_flash_attention_x3_two_tiles_per_wave_changes_no_bit). The key-stationary dK / dV kernel stays at one tile: K, V fragments and two accumulator sets per tile need 376
// registers for two (one wave per SIMD; measured slower).
template <int HD, int NT>
__global__ __launch_bounds__(FX_THREADS) void fx_fwd2_kernel(const FxArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TB = 64 * HD * 2, XQB = XQ * NT;
    char* Kh = smem; char* Kl = smem + TB;                  // rows images [64 keys][HD]
    char* Vh = smem + 2 * TB; char* Vl = smem + 3 * TB;     // transposed [HD][64 keys]
    float* ldsB = reinterpret_cast<float*>(smem + 4 * TB);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, lr = lane & 15, g = lane >> 4;
    int rb_, h, b;
    grid_map3(rb_, h, b);
    const int q0 = rb_ * XQB;
    const XView w = xview(p, b);
    if (q0 >= w.Sq) return;                                    // packed rows: a shorter sequence than the longest one
    const float* Q = p.q + b * p.q_sb + w.qrow * p.q_ss + h * HD;
    const float* K = p.k + b * p.k_sb + w.krow * p.k_ss + h * HD;
    const float* V = p.v + b * p.v_sb + w.krow * p.v_ss + h * HD;
    constexpr int KS = HD / 32, DT = HD / 16;
    int myq[NT];
    bf16x8 qh[NT][KS], ql[NT][KS];
    f32x4 oacc[NT][DT];
    float m[NT], l[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        myq[n] = q0 + (wave * NT + n) * 16 + lr;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xfrag_global(Q, p.q_ss, myq[n], w.Sq, ks * 32 + g * 8, qh[n][ks], ql[n][ks]);
#pragma unroll
        for (int i = 0; i < DT; ++i) oacc[n][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        m[n] = -INFINITY; l[n] = 0.f;
    }
    const float c = p.scale * XLOG2E;
    const int kend = min(p.causal ? min(w.Sk, q0 + XQB) : w.Sk, w.kvis);    // key tiles behind the last visible key contribute exact zeros: skipped
    for (int k0 = 0; k0 < kend; k0 += XK) {
        __syncthreads();
        xstage_rows<HD>(Kh, Kl, K + (long)k0 * p.k_ss, p.k_ss, w.Sk - k0, t);
        xstage_transposed<HD>(Vh, Vl, V + (long)k0 * p.v_ss, p.v_ss, w.Sk - k0, t);
        if (t < XK) {
            const int key = k0 + t;
            const bool vis = key < w.kvis && (w.packed || !p.key_mask || p.key_mask[(long)b * p.Sk + key] != 0.f);
            ldsB[t] = vis ? 0.f : -INFINITY;
        }
        __syncthreads();
        f32x4 s[NT][4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
            for (int n = 0; n < NT; ++n) s[n][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 kfh = xfrag_rows<HD>(Kh, kt * 16 + lr, ks, g), kfl = xfrag_rows<HD>(Kl, kt * 16 + lr, ks, g);
#pragma unroll
                for (int n = 0; n < NT; ++n) s[n][kt] = XM3(kfh, kfl, qh[n][ks], ql[n][ks], s[n][kt]);
            }
        }
        bf16x8 p0h[NT], p0l[NT], p1h[NT], p1l[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const f32x4 bias = *reinterpret_cast<const f32x4*>(ldsB + kt * 16 + g * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x = __builtin_fmaf(s[n][kt][r], c, bias[r]);
                    if (p.causal && (k0 + kt * 16 + g * 4 + r) > myq[n]) x = -INFINITY;
                    s[n][kt][r] = x;
                    mx = fmaxf(mx, x);
                }
            }
            mx = xgrp_max(mx);
            const float mnew = fmaxf(m[n], mx);
            const float muse = mnew == -INFINITY ? 0.f : mnew;
            const float alpha = __builtin_amdgcn_exp2f(m[n] - muse);
            float rs = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float e = __builtin_amdgcn_exp2f(s[n][kt][r] - muse); s[n][kt][r] = e; rs += e; }
            rs = xgrp_sum(rs);
            l[n] = __builtin_fmaf(l[n], alpha, rs);
            m[n] = mnew;
#pragma unroll
            for (int i = 0; i < DT; ++i) oacc[n][i] *= alpha;
            xpack_pair(s[n][0], s[n][1], p0h[n], p0l[n]);
            xpack_pair(s[n][2], s[n][3], p1h[n], p1l[n]);
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const bf16x8 v0h = xfrag_perm(Vh, dt * 16 + lr, 0, g), v0l = xfrag_perm(Vl, dt * 16 + lr, 0, g);
            const bf16x8 v1h = xfrag_perm(Vh, dt * 16 + lr, 1, g), v1l = xfrag_perm(Vl, dt * 16 + lr, 1, g);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                oacc[n][dt] = XM3(v0h, v0l, p0h[n], p0l[n], oacc[n][dt]);
                oacc[n][dt] = XM3(v1h, v1l, p1h[n], p1l[n], oacc[n][dt]);
            }
        }
    }
#pragma unroll
    for (int n = 0; n < NT; ++n)
        if (myq[n] < w.Sq) {
            const float inv = l[n] > 0.f ? 1.0f / l[n] : 0.f;
            float* O = p.out + b * p.o_sb + (w.qrow + myq[n]) * p.o_ss + h * HD;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) *reinterpret_cast<f32x4*>(O + dt * 16 + g * 4) = oacc[n][dt] * inv;
            if (g == 0) p.lse[((long)b * p.H + h) * p.Sq + myq[n]] = l[n] > 0.f ? (m[n] + log2f(l[n])) / XLOG2E : INFINITY;
        }
}

template <int HD, int NT>
__global__ __launch_bounds__(FX_THREADS) void fx_bwd_dq2_kernel(const FxArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TB = 64 * HD * 2, XQB = XQ * NT;              // NT query tiles of 16 per wave (see fx_fwd2_kernel)
    char* Kh = smem;           char* Kl = smem + TB;
    char* KTh = smem + 2 * TB; char* KTl = smem + 3 * TB;
    char* Vh = smem + 4 * TB;  char* Vl = smem + 5 * TB;
    float* ldsB = reinterpret_cast<float*>(smem + 6 * TB);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, lr = lane & 15, g = lane >> 4;
    int rb_, h, b;
    grid_map3(rb_, h, b);
    const int q0 = rb_ * XQB;
    const XView w = xview(p, b);
    if (q0 >= w.Sq) return;                                    // packed rows: a shorter sequence than the longest one
    const float* Q = p.q + b * p.q_sb + w.qrow * p.q_ss + h * HD;
    const float* K = p.k + b * p.k_sb + w.krow * p.k_ss + h * HD;
    const float* V = p.v + b * p.v_sb + w.krow * p.v_ss + h * HD;
    const float* DO = p.dout + b * p.o_sb + w.qrow * p.o_ss + h * HD;
    constexpr int KS = HD / 32, DT = HD / 16;
    int myq[NT];
    bf16x8 qh[NT][KS], ql[NT][KS], oh[NT][KS], ol[NT][KS];
    float lse[NT], dl[NT];
    f32x4 dq[NT][DT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        myq[n] = q0 + (wave * NT + n) * 16 + lr;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            xfrag_global(Q, p.q_ss, myq[n], w.Sq, ks * 32 + g * 8, qh[n][ks], ql[n][ks]);
            xfrag_global(DO, p.o_ss, myq[n], w.Sq, ks * 32 + g * 8, oh[n][ks], ol[n][ks]);
        }
        const long li = ((long)b * p.H + h) * p.Sq + myq[n];
        lse[n] = myq[n] < w.Sq ? p.lse[li] * XLOG2E : INFINITY;
        dl[n] = myq[n] < w.Sq ? p.delta[li] : 0.f;
#pragma unroll
        for (int i = 0; i < DT; ++i) dq[n][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float c = p.scale * XLOG2E;
    const int kend = min(p.causal ? min(w.Sk, q0 + XQB) : w.Sk, w.kvis);    // key tiles behind the last visible key contribute exact zeros: skipped
    for (int k0 = 0; k0 < kend; k0 += XK) {
        __syncthreads();
        xstage_both<HD>(Kh, Kl, KTh, KTl, K + (long)k0 * p.k_ss, p.k_ss, w.Sk - k0, t);
        xstage_rows<HD>(Vh, Vl, V + (long)k0 * p.v_ss, p.v_ss, w.Sk - k0, t);
        if (t < XK) {
            const int key = k0 + t;
            ldsB[t] = (key < w.kvis && (w.packed || !p.key_mask || p.key_mask[(long)b * p.Sk + key] != 0.f)) ? 1.f : 0.f;
        }
        __syncthreads();
        f32x4 s[NT][4], dp[NT][4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
            for (int n = 0; n < NT; ++n) { s[n][kt] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[n][kt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 kfh = xfrag_rows<HD>(Kh, kt * 16 + lr, ks, g), kfl = xfrag_rows<HD>(Kl, kt * 16 + lr, ks, g);
                const bf16x8 vfh = xfrag_rows<HD>(Vh, kt * 16 + lr, ks, g), vfl = xfrag_rows<HD>(Vl, kt * 16 + lr, ks, g);
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    s[n][kt] = XM3(kfh, kfl, qh[n][ks], ql[n][ks], s[n][kt]);
                    dp[n][kt] = XM3(vfh, vfl, oh[n][ks], ol[n][ks], dp[n][kt]);
                }
            }
        }
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const f32x4 vis4 = *reinterpret_cast<const f32x4*>(ldsB + kt * 16 + g * 4);
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = k0 + kt * 16 + g * 4 + r;
                    const bool vis = vis4[r] != 0.f && (!p.causal || key <= myq[n]);
                    const float pr = vis ? __builtin_amdgcn_exp2f(__builtin_fmaf(s[n][kt][r], c, -lse[n])) : 0.f;
                    dp[n][kt][r] = xds(pr, dp[n][kt][r], dl[n], p.scale);
                }
        }
        bf16x8 d0h[NT], d0l[NT], d1h[NT], d1l[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) { xpack_pair(dp[n][0], dp[n][1], d0h[n], d0l[n]); xpack_pair(dp[n][2], dp[n][3], d1h[n], d1l[n]); }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const bf16x8 k0h = xfrag_perm(KTh, dt * 16 + lr, 0, g), k0l = xfrag_perm(KTl, dt * 16 + lr, 0, g);
            const bf16x8 k1h = xfrag_perm(KTh, dt * 16 + lr, 1, g), k1l = xfrag_perm(KTl, dt * 16 + lr, 1, g);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                dq[n][dt] = XM3(k0h, k0l, d0h[n], d0l[n], dq[n][dt]);
                dq[n][dt] = XM3(k1h, k1l, d1h[n], d1l[n], dq[n][dt]);
            }
        }
    }
#pragma unroll
    for (int n = 0; n < NT; ++n)
        if (myq[n] < w.Sq) {
            float* DQ = p.dq + b * p.dq_sb + (w.qrow + myq[n]) * p.dq_ss + h * HD;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) *reinterpret_cast<f32x4*>(DQ + dt * 16 + g * 4) = dq[n][dt];
        }
}

int xcheck(const char* who, int hd, const long* st, int n) {
    PB_REQUIRE(hd == 32 || hd == 64 || hd == 128, "%s: head_dim %d (32 / 64 / 128)", who, hd);
    for (int i = 0; i < n; ++i) PB_REQUIRE(st[i] % 4 == 0, "%s: strides must be multiples of 4 elements (16-byte f32 rows)", who);
    return 0;
}

#define FX_DISPATCH(HDV, ...)                                   \
    switch (HDV) {                                              \
        case 32: { constexpr int HD = 32; __VA_ARGS__; } break; \
        case 64: { constexpr int HD = 64; __VA_ARGS__; } break; \
        default: { constexpr int HD = 128; __VA_ARGS__; } break; \
    }

static int fx_nt(int hd, int B, int H, int Sq) { return (hd <= 64 && (long)B * H * ((Sq + 2 * XQ - 1) / (2 * XQ)) >= 1024) ? 2 : 1; }

template <class F> int opt_in_lds(F fn, size_t bytes) {
    PB_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return 0;
}

}  // namespace

extern "C" int pb_flash_x3_supported(int32_t hd) { return (hd == 32 || hd == 64 || hd == 128) ? 1 : 0; }

static int fx_fwd_impl(const float* q, const float* k, const float* v, float* o, float* lse, const float* key_mask, const int32_t* kmax, int32_t B, int32_t H, int32_t Sq,
                       int32_t Sk, int32_t hd, int64_t q_sb, int64_t q_ss, int64_t k_sb, int64_t k_ss, int64_t v_sb, int64_t v_ss, int64_t o_sb,
                       int64_t o_ss, float scale, int32_t causal, void* stream_, const int* const* vl) {
    const long st[8] = {q_sb, q_ss, k_sb, k_ss, v_sb, v_ss, o_sb, o_ss};
    if (xcheck("pb_flash_fwd_x3", hd, st, 8)) return -2;
    PB_REQUIRE(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) % 16 == 0, "pb_flash_fwd_x3: operands must be 16-byte aligned");
    if (B <= 0 || H <= 0 || Sq <= 0) return 0;
    FxArgs a = {};
    a.q = q; a.k = k; a.v = v; a.out = o; a.lse = lse; a.key_mask = key_mask; a.kmax = key_mask ? kmax : nullptr;
    if (vl) { a.vq_off = vl[0]; a.vq_len = vl[1]; a.vk_off = vl[2]; a.vk_len = vl[3]; a.vk_vis = vl[4]; }
    a.B = B; a.H = H; a.Sq = Sq; a.Sk = Sk; a.q_sb = q_sb; a.q_ss = q_ss; a.k_sb = k_sb; a.k_ss = k_ss; a.v_sb = v_sb; a.v_ss = v_ss;
    a.o_sb = o_sb; a.o_ss = o_ss; a.scale = scale; a.causal = causal & 1;
    const int nt = fx_nt(hd, B, H, Sq);
    dim3 grid((Sq + XQ * nt - 1) / (XQ * nt), H, B);
    const size_t lds = 4 * 64 * (size_t)hd * 2 + 256;
    if (nt == 2) {
        if (hd == 32) { if (opt_in_lds(&fx_fwd2_kernel<32, 2>, lds)) return -1; hipLaunchKernelGGL((fx_fwd2_kernel<32, 2>), grid, dim3(FX_THREADS), lds, (hipStream_t)stream_, a); }
        else { if (opt_in_lds(&fx_fwd2_kernel<64, 2>, lds)) return -1; hipLaunchKernelGGL((fx_fwd2_kernel<64, 2>), grid, dim3(FX_THREADS), lds, (hipStream_t)stream_, a); }
    } else {
        FX_DISPATCH(hd, if (opt_in_lds(&fx_fwd_kernel<HD>, lds)) return -1; hipLaunchKernelGGL((fx_fwd_kernel<HD>), grid, dim3(FX_THREADS), lds, (hipStream_t)stream_, a));
    }
    PB_LAUNCH_CHECK();
    return 0;
}

static int fx_bwd_impl(const float* q, const float* k, const float* v, const float* o, const float* dout, const float* lse, const float* key_mask,
                       const int32_t* kmax, float* dq, float* dk, float* dv, float* delta, int32_t B, int32_t H, int32_t Sq, int32_t Sk, int32_t hd, int64_t q_sb,
                       int64_t q_ss, int64_t k_sb, int64_t k_ss, int64_t v_sb, int64_t v_ss, int64_t o_sb, int64_t o_ss, int64_t dq_sb,
                       int64_t dq_ss, int64_t dk_sb, int64_t dk_ss, int64_t dv_sb, int64_t dv_ss, float scale, int32_t causal, void* stream_, const int* const* vl) {
    hipStream_t stream = (hipStream_t)stream_;
    const long st[14] = {q_sb, q_ss, k_sb, k_ss, v_sb, v_ss, o_sb, o_ss, dq_sb, dq_ss, dk_sb, dk_ss, dv_sb, dv_ss};
    if (xcheck("pb_flash_bwd_x3", hd, st, 14)) return -2;
    PB_REQUIRE(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o | (uintptr_t)dout | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) % 16 == 0,
               "pb_flash_bwd_x3: operands must be 16-byte aligned");
    if (B <= 0 || H <= 0 || Sq <= 0) return 0;
    FxArgs a = {};
    a.q = q; a.k = k; a.v = v; a.o = o; a.dout = dout; a.dq = dq; a.dk = dk; a.dv = dv; a.lse = const_cast<float*>(lse); a.delta = delta; a.key_mask = key_mask; a.kmax = key_mask ? kmax : nullptr;
    if (vl) { a.vq_off = vl[0]; a.vq_len = vl[1]; a.vk_off = vl[2]; a.vk_len = vl[3]; a.vk_vis = vl[4]; }
    a.B = B; a.H = H; a.Sq = Sq; a.Sk = Sk; a.q_sb = q_sb; a.q_ss = q_ss; a.k_sb = k_sb; a.k_ss = k_ss; a.v_sb = v_sb; a.v_ss = v_ss;
    a.o_sb = o_sb; a.o_ss = o_ss; a.dq_sb = dq_sb; a.dq_ss = dq_ss; a.dk_sb = dk_sb; a.dk_ss = dk_ss; a.dv_sb = dv_sb; a.dv_ss = dv_ss;
    a.scale = scale; a.causal = causal & 1;
    const long nrow = (long)B * H * Sq;
    FX_DISPATCH(hd, hipLaunchKernelGGL((fx_delta_kernel<HD>), dim3((unsigned)((nrow + 255) / 256)), dim3(256), 0, stream, o, dout, delta, B, H, Sq, (long)o_sb, (long)o_ss, vl ? vl[0] : nullptr, vl ? vl[1] : nullptr));
    PB_LAUNCH_CHECK();
    const int nt = fx_nt(hd, B, H, Sq);
    dim3 gk((Sk + XK - 1) / XK, H, B), gq((Sq + XQ * nt - 1) / (XQ * nt), H, B);
    const size_t lds_kv = 8 * 64 * (size_t)hd * 2 + 512, lds_q = 6 * 64 * (size_t)hd * 2 + 256;
    FX_DISPATCH(hd, if (opt_in_lds(&fx_bwd_dkv_kernel<HD>, lds_kv)) return -1; hipLaunchKernelGGL((fx_bwd_dkv_kernel<HD>), gk, dim3(FX_THREADS), lds_kv, stream, a));
    PB_LAUNCH_CHECK();
    if (nt == 2) {
        if (hd == 32) { if (opt_in_lds(&fx_bwd_dq2_kernel<32, 2>, lds_q)) return -1; hipLaunchKernelGGL((fx_bwd_dq2_kernel<32, 2>), gq, dim3(FX_THREADS), lds_q, stream, a); }
        else { if (opt_in_lds(&fx_bwd_dq2_kernel<64, 2>, lds_q)) return -1; hipLaunchKernelGGL((fx_bwd_dq2_kernel<64, 2>), gq, dim3(FX_THREADS), lds_q, stream, a); }
    } else {
        FX_DISPATCH(hd, if (opt_in_lds(&fx_bwd_dq_kernel<HD>, lds_q)) return -1; hipLaunchKernelGGL((fx_bwd_dq_kernel<HD>), gq, dim3(FX_THREADS), lds_q, stream, a));
    }
    PB_LAUNCH_CHECK();
    return 0;
}

extern "C" int pb_flash_fwd_x3(const float* q, const float* k, const float* v, float* o, float* lse, const float* key_mask, const int32_t* kmax, int32_t B, int32_t H, int32_t Sq,
                               int32_t Sk, int32_t hd, int64_t q_sb, int64_t q_ss, int64_t k_sb, int64_t k_ss, int64_t v_sb, int64_t v_ss, int64_t o_sb,
                               int64_t o_ss, float scale, int32_t causal, void* stream_) {
    return fx_fwd_impl(q, k, v, o, lse, key_mask, kmax, B, H, Sq, Sk, hd, q_sb, q_ss, k_sb, k_ss, v_sb, v_ss, o_sb, o_ss, scale, causal, stream_, nullptr);
}
extern "C" int pb_flash_bwd_x3(const float* q, const float* k, const float* v, const float* o, const float* dout, const float* lse, const float* key_mask,
                               const int32_t* kmax, float* dq, float* dk, float* dv, float* delta, int32_t B, int32_t H, int32_t Sq, int32_t Sk, int32_t hd, int64_t q_sb,
                               int64_t q_ss, int64_t k_sb, int64_t k_ss, int64_t v_sb, int64_t v_ss, int64_t o_sb, int64_t o_ss, int64_t dq_sb,
                               int64_t dq_ss, int64_t dk_sb, int64_t dk_ss, int64_t dv_sb, int64_t dv_ss, float scale, int32_t causal, void* stream_) {
    return fx_bwd_impl(q, k, v, o, dout, lse, key_mask, kmax, dq, dk, dv, delta, B, H, Sq, Sk, hd, q_sb, q_ss, k_sb, k_ss, v_sb, v_ss, o_sb, o_ss, dq_sb, dq_ss, dk_sb, dk_ss,
                       dv_sb, dv_ss, scale, causal, stream_, nullptr);
}
// packed rows (see pb_flash_fwd_packed): the rows of the batch's sequences lie back to back in (rows, H hd) tensors; q_off / q_len / k_off / k_len / k_vis are
// device int32 (B) arrays, Sq_max / Sk_max the longest sequence (grid, lse / delta row length)
extern "C" int pb_flash_fwd_x3_packed(const float* q, const float* k, const float* v, float* o, float* lse, const int32_t* q_off, const int32_t* q_len, const int32_t* k_off,
                                      const int32_t* k_len, const int32_t* k_vis, int32_t B, int32_t H, int32_t Sq_max, int32_t Sk_max, int32_t hd, int64_t q_ss,
                                      int64_t k_ss, int64_t v_ss, int64_t o_ss, float scale, int32_t causal, void* stream_) {
    PB_REQUIRE(q_off && q_len && k_off && k_len && k_vis, "pb_flash_fwd_x3_packed: the five row descriptors are required");
    const int* vl[5] = {q_off, q_len, k_off, k_len, k_vis};
    return fx_fwd_impl(q, k, v, o, lse, nullptr, nullptr, B, H, Sq_max, Sk_max, hd, 0, q_ss, 0, k_ss, 0, v_ss, 0, o_ss, scale, causal, stream_, vl);
}
extern "C" int pb_flash_bwd_x3_packed(const float* q, const float* k, const float* v, const float* o, const float* dout, const float* lse, float* dq, float* dk, float* dv,
                                      float* delta, const int32_t* q_off, const int32_t* q_len, const int32_t* k_off, const int32_t* k_len, const int32_t* k_vis,
                                      int32_t B, int32_t H, int32_t Sq_max, int32_t Sk_max, int32_t hd, int64_t q_ss, int64_t k_ss, int64_t v_ss, int64_t o_ss,
                                      int64_t dq_ss, int64_t dk_ss, int64_t dv_ss, float scale, int32_t causal, void* stream_) {
    PB_REQUIRE(q_off && q_len && k_off && k_len && k_vis, "pb_flash_bwd_x3_packed: the five row descriptors are required");
    const int* vl[5] = {q_off, q_len, k_off, k_len, k_vis};
    return fx_bwd_impl(q, k, v, o, dout, lse, nullptr, nullptr, dq, dk, dv, delta, B, H, Sq_max, Sk_max, hd, 0, q_ss, 0, k_ss, 0, v_ss, 0, o_ss, 0, dq_ss, 0, dk_ss, 0, dv_ss,
                       scale, causal, stream_, vl);
}
