// Shared device helpers for the PianoBART gfx950 kernels (CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PB_WAVE 64

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// ---- error reporting (host) -------------------------------------------------
extern "C" const char* pb_last_error(void);
void pb_set_error(const char* fmt, ...);
#define PB_CHECK_HIP(expr)                                                     \
    do {                                                                       \
        hipError_t _e = (expr);                                                \
        if (_e != hipSuccess) {                                                \
            pb_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return -1;                                                         \
        }                                                                      \
    } while (0)
#define PB_REQUIRE(cond, ...)                                                  \
    do {                                                                       \
        if (!(cond)) { pb_set_error(__VA_ARGS__); return -2; }                 \
    } while (0)
#define PB_LAUNCH_CHECK() PB_CHECK_HIP(hipGetLastError())

// ---- compile-time loop: f(IntTag<I>{}) for I = I0 .. N-1 (literal register numbers, immediate wait counts) -------
template <int V> struct IntTag { static constexpr int value = V; };
template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(IntTag<I>{}); static_for<I + 1, N>(f); }
}

// ---- scalar conversions -------------------------------------------------------
__device__ __forceinline__ float to_f(float x) { return x; }
__device__ __forceinline__ float to_f(bf16_t x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f(float x);
template <> __device__ __forceinline__ float from_f<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f<bf16_t>(float x) { return (bf16_t)x; }  // v_cvt_pk_bf16_f32, RNE, NaN-safe

// ---- 4-element vector load/store as float (8 B for bf16, 16 B for f32) -----------
__device__ __forceinline__ f32x4 load4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 load4(const bf16_t* p) {
    bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    f32x4 r = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    return r;
}
__device__ __forceinline__ void store4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void store4(bf16_t* p, f32x4 v) {
    bf16x4 r = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    *reinterpret_cast<bf16x4*>(p) = r;
}

// non-temporal variants (streaming outputs that are not re-read by this kernel)
__device__ __forceinline__ void store4_nt(float* p, f32x4 v) { __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p)); }
__device__ __forceinline__ void store4_nt(bf16_t* p, f32x4 v) {
    bf16x4 r = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    __builtin_nontemporal_store(r, reinterpret_cast<bf16x4*>(p));
}

// ---- wave64 reductions ------------------------------------------------------------
// All-lanes butterfly without LDS: quad permutes and row mirrors (DPP) inside each 16-lane row, v_permlane16/32_swap across the
// rows (after swap(v, v) one result is the lane's own value and the other its partner's). __shfl_xor compiles to ds_bpermute,
// six LDS round trips per reduction on the latency path of every row kernel.
#define PB_DPP_F(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xf, 0xf, true))
__device__ __forceinline__ float wave_sum(float v) {
    v += PB_DPP_F(v, 0xb1);      // quad_perm [1,0,3,2]
    v += PB_DPP_F(v, 0x4e);      // quad_perm [2,3,0,1]
    v += PB_DPP_F(v, 0x141);     // row_half_mirror
    v += PB_DPP_F(v, 0x140);     // row_mirror
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, PB_DPP_F(v, 0xb1));
    v = fmaxf(v, PB_DPP_F(v, 0x4e));
    v = fmaxf(v, PB_DPP_F(v, 0x141));
    v = fmaxf(v, PB_DPP_F(v, 0x140));
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// (row block, head, batch row) of a workgroup of a (row blocks, H, B) grid -- the mapping of pb_fa_tiles.h's block_map for the kernels launched on 3-D grids:
// the heads of an XCD (linear workgroup id & 7) keep their row blocks together (their K / V stay in ITS L2), and the row block is rotated by the head's position so
// that no XCD and no shader engine collects the long row blocks of causal calls (x fastest put query blocks {x, x + 8} of EVERY head on XCD x: 10 : 24 units of causal work).
__device__ __forceinline__ void grid_map3(int& rb, int& h, int& b) {
    const int nrb = gridDim.x, H = gridDim.y, BH = gridDim.y * gridDim.z;
    const int L = blockIdx.x + nrb * (blockIdx.y + H * blockIdx.z);
    int bh;
    if ((BH & 7) == 0) { const int x = L & 7, slot = L >> 3, j = slot / nrb; bh = j * 8 + x; rb = (slot - j * nrb + j) % nrb; }
    else { bh = L / nrb; rb = (L - bh * nrb + bh) % nrb; }
    h = bh % H; b = bh / H;
}

// ---- exact-erf GELU (activation_function="gelu") -----------------------------------
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// Fast variants for the bf16 throughput path (the exact-f32 path keeps erff): erf by Abramowitz-Stegun 7.1.26
// (|error| <= 1.5e-7, far below bf16 resolution): one v_exp, one v_rcp; the derivative shares the exponential. Written on
// h(x) = 0.5 erfc(|x| / sqrt2) = 0.5 t P(t) exp(-x^2/2), t = 1 / (1 + 0.3275911 |x| / sqrt2): cdf = 0.5 + copysign(0.5 - h, x) -- a v_bfi instead of a
// compare + select (20 cycles per wave-instruction behind the SGPR mask on gfx950: tools/probes/valu_rate.hip), the 0.5 folded into the coefficients.
// gelu_pair2 below is the same arithmetic on two values in packed f32 instructions.
constexpr float PB_GELU_A2 = 0.72134752044448170f;               // log2(e) / 2: exp(-x^2 / 2) = exp2(-A2 x^2)
constexpr float PB_GELU_C1 = 0.23164188861847f;                  // 0.3275911 / sqrt(2)
constexpr float PB_GELU_H1 = 0.127414796f, PB_GELU_H2 = -0.142248368f, PB_GELU_H3 = 0.7107068705f, PB_GELU_H4 = -0.7265760135f, PB_GELU_H5 = 0.5307027145f;
__device__ __forceinline__ void gelu_parts_fast(float x, float& cdf, float& ex) {
    const float t = __builtin_amdgcn_rcpf(fmaf(fabsf(x), PB_GELU_C1, 1.0f));
    ex = __builtin_amdgcn_exp2f(-((x * x) * PB_GELU_A2));                          // exp(-x^2/2)
    const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, PB_GELU_H5, PB_GELU_H4), PB_GELU_H3), PB_GELU_H2), PB_GELU_H1);
    cdf = 0.5f + __builtin_copysignf(0.5f - poly * ex, x);
}
typedef __attribute__((ext_vector_type(2))) float pb_f32x2;
__device__ __forceinline__ void gelu_pair2(pb_f32x2 x, pb_f32x2& y, pb_f32x2& dy) {
    const pb_f32x2 q = (x * x) * PB_GELU_A2;
    pb_f32x2 e, t;
    e[0] = __builtin_amdgcn_exp2f(-q[0]); e[1] = __builtin_amdgcn_exp2f(-q[1]);
    t[0] = __builtin_amdgcn_rcpf(fmaf(fabsf(x[0]), PB_GELU_C1, 1.0f)); t[1] = __builtin_amdgcn_rcpf(fmaf(fabsf(x[1]), PB_GELU_C1, 1.0f));
    pb_f32x2 pl = t * PB_GELU_H5 + PB_GELU_H4;
    pl = pl * t + PB_GELU_H3; pl = pl * t + PB_GELU_H2; pl = pl * t + PB_GELU_H1; pl = pl * t;
    const pb_f32x2 r = 0.5f - pl * e;
    pb_f32x2 rs; rs[0] = __builtin_copysignf(r[0], x[0]); rs[1] = __builtin_copysignf(r[1], x[1]);
    const pb_f32x2 cdf = rs + 0.5f;
    y = x * cdf;
    dy = (x * 0.3989422804014327f) * e + cdf;
}
__device__ __forceinline__ float gelu_fast(float x) { float c, e; gelu_parts_fast(x, c, e); return x * c; }
__device__ __forceinline__ void gelu_both_fast(float x, float& y, float& dy) { float c, e; gelu_parts_fast(x, c, e); y = x * c; dy = c + x * 0.3989422804014327f * e; }
__device__ __forceinline__ float gelu_grad_fast(float x) { float c, e; gelu_parts_fast(x, c, e); return c + x * 0.3989422804014327f * e; }

// ---- counter-based RNG for dropout: Philox4x32-7 -----------------------------------
// One call yields 4 x 32 random bits for the 4 consecutive elements [4*idx4, 4*idx4+3] of a
// dropout site. Forward and backward regenerate the same mask from (seed, site, idx4).
__device__ __forceinline__ uint4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 7; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
}
struct DropCfg {
    uint32_t seed_lo, seed_hi;   // per-step seed
    uint32_t site;               // unique id of the dropout site inside a step
    uint32_t thresh;             // drop if rnd < thresh  (thresh = p * 2^32); 0 => dropout off
    float scale;                 // 1/(1-p)
};
__device__ __forceinline__ f32x4 drop_mask4(const DropCfg& d, uint32_t idx4) {
    f32x4 m = {1.f, 1.f, 1.f, 1.f};
    if (d.thresh == 0u) return m;
    const uint4 r = philox4x32(idx4, d.site, 0x5EEDu, 0u, d.seed_lo, d.seed_hi);
    m[0] = r.x < d.thresh ? 0.f : d.scale;
    m[1] = r.y < d.thresh ? 0.f : d.scale;
    m[2] = r.z < d.thresh ? 0.f : d.scale;
    m[3] = r.w < d.thresh ? 0.f : d.scale;
    return m;
}
