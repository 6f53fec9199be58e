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
#include "pb_fa_tiles.h"

namespace {


// ================================================================== forward: block = 128 queries
// LDS: ring of {K tile, V tile} | key bias (0 / -inf) of every key this block visits | one "has a masked key" word per tile.
// Pipeline: the DMA of tile it+NS-1 is issued at the top of tile it and waited for, with a counted vmcnt, at the bottom of
// tile it+NS-2 in front of a raw s_barrier; nothing in the loop drains it (no ordinary global load, no __syncthreads(), no
// ds_read_tr builtin). V's transposed fragments are requested before the softmax and collected after.
// Softmax with a LAZY reference maximum: Q is prescaled by scale * log2(e) and each query's current reference m_ref rides into
// the S chains as their INITIAL ACCUMULATOR, so the MFMAs deliver s' = s - m_ref and the common tile costs one v_exp_f32 per
// score -- no subtraction, no cross-lane maximum, no rescale of O. The reference only moves when a tile holds a score more than
// LAZY_THR (log2 units) above it, or the row has none yet; then, and on masked / causal-diagonal tiles, the exact path runs
// (true row maximum, alpha rescale of O and l, new reference). p <= 2^LAZY_THR is harmless: bf16 has f32's exponent range and l
// is summed from the same bf16 p the PV product sees.
constexpr float LAZY_THR = 8.0f;
template <int HD>
__global__ __launch_bounds__(FT) void fa64_fwd_kernel(const Fa64Args pin) {
    constexpr int NB = (HD + 63) / 64, KS = HD / 32, DT = HD / 16, HDT = HD;
    using C = FaCfg<NB>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, lr = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    int rb, h, b;
    block_map((pin.Sq + 127) / 128, pin.H, pin.B, rb, h, b, pin.bh_order);
    if (pin.causal) rb = (pin.Sq + 127) / 128 - 1 - rb;        // causal: a later row block visits more keys -- the long blocks of a (batch, head) are dispatched first
    const int q0 = rb * 128;
    Fa64Args p = pin;
    varlen_localize(p, b);
    const int lse_ld = pin.Sq;                                            // row length of lse (B, H, Sq_max)
    if (q0 >= p.Sq) return;                                               // packed rows: this batch has no such row block
    const bf16_t* Q = p.q + b * p.q_sb + h * HDT;
    const bf16_t* K = p.k + b * p.k_sb + h * HDT;
    const bf16_t* V = p.v + b * p.v_sb + h * HDT;
    const float c = p.scale * LOG2E;
    // keys at and beyond kmax[b] (1 + last visible key of this batch row: the PAD tail) are masked for every query: skip their tiles
    const int kvis_end = p.kmax ? min(p.Sk, p.kmax[b]) : p.Sk;
    const int kend = p.causal ? min(kvis_end, q0 + 128) : kvis_end;
    const int nt = (kend + 63) / 64;
    float* ldsBias = reinterpret_cast<float*>(smem + C::NS * C::STB);
    unsigned* ldsFlag = reinterpret_cast<unsigned*>(ldsBias + ((p.Sk + 63) / 64) * 64);
    for (int tile = wave; tile < nt; tile += 4) {                        // one wave = one tile of keys
        const int key = tile * 64 + lane;
        const bool vis = key < (pin.vl_q_off ? kvis_end : p.Sk) && (!p.key_mask || p.key_mask[(long)b * p.Sk + key] != 0.f);
        ldsBias[key] = vis ? 0.f : -INFINITY;
        const bool allvis = __builtin_amdgcn_ballot_w64(vis) == ~0ull;
        if (lane == 0) ldsFlag[tile] = allvis ? 0u : 1u;
    }
    int myq[2];
    bf16x8 qf[2][KS];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        myq[qt] = q0 + wave * 32 + qt * 16 + lr;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[qt][ks] = scale_frag(frag_global(Q, p.q_ss, myq[qt], p.Sq, ks * 32 + g * 8), c);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[0][ks]), "+v"(qf[1][ks]));   // ordinary loads are done before the first DMA
    f32x4 oacc[2][DT];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int i = 0; i < DT; ++i) oacc[qt][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m[2] = {-INFINITY, -INFINITY};                 // reference maximum of the row in log2 units (-inf: no visible key seen yet)
    f32x4 cinit[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};   // -m (0 while m = -inf): initial accumulator of the S chains
    // row sums ride on the MFMA pipe: l^T += 1 P^T with an all-ones A operand gives sum_k p[q][k] (of the bf16 p the PV product
    // sees) in every register of the accumulator -- 4 MFMAs per tile instead of 32 v_add + 2 cross-row reductions per lane
    f32x4 lacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const bf16_t one = (bf16_t)1.0f;
    const bf16x8 ones = {one, one, one, one, one, one, one, one};
    unsigned voff[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) voff[dt] = lds_u32(smem) + NB * 8192 + tr_lane_off64(dt * 16, lane);
    const StageOff so_k = stage_off(p.k_ss, wave, lane), so_v = stage_off(p.v_ss, wave, lane);
    auto stage = [&](int it, int sidx) {
        char* st = smem + sidx * C::STB;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            stage64(K + nb * 64, p.k_ss, it * 64, p.Sk, st + nb * 8192, wave, lane, so_k, min(8, (HD - nb * 64) / 8), p.zeros);
            stage64(V + nb * 64, p.v_ss, it * 64, p.Sk, st + (NB + nb) * 8192, wave, lane, so_v, min(8, (HD - nb * 64) / 8), p.zeros);
        }
    };
    if (nt > 0) stage(0, 0);
    if (C::NS == 3 && nt > 1) { stage(1, 1); wait_vm<C::PCS>(); } else { wait_vm<0>(); }
    __builtin_amdgcn_s_waitcnt(0xc07f);                                    // this wave's bias / flag stores
    __builtin_amdgcn_s_barrier();
    int sidx = 0;
    for (int it = 0; it < nt; ++it) {
        const char* st = smem + sidx * C::STB;
        const int nidx = sidx == 0 ? C::NS - 1 : sidx - 1;                 // (it + NS - 1) % NS
        if (it + C::NS - 1 < nt) stage(it + C::NS - 1, nidx);
        const float* ldsB = ldsBias + it * 64;
        const int k0 = it * 64;
        const bool diag = p.causal && (k0 + 63 > q0 + wave * 32);        // wave-uniform: only tiles that touch the diagonal compare
        const bool masked = diag || __builtin_amdgcn_readfirstlane(ldsFlag[it]) != 0u;
        f32x4 s[2][4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) s[qt][kt] = cinit[qt];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2)
                    if (2 * nb + k2 < KS) {
                        const bf16x8 kk = frag_row(st + nb * 8192, kt * 16 + lr, k2, g);
#pragma unroll
                        for (int qt = 0; qt < 2; ++qt) s[qt][kt] = MFMA16(kk, qf[qt][2 * nb + k2], s[qt][kt]);
                    }
            }
        }
        s16x4 tv[4][2][2];                                                // V image 0: requested before the softmax, collected after it
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) if (dt < DT) ds_tr_block(tv[dt], voff[dt] + (unsigned)(sidx * C::STB));
        bf16x8 pf[2][2];
        auto finish_row = [&](int qt) {                                   // p -> bf16 operand; row sums on the MFMA pipe
            pf[qt][0] = pack_pair(s[qt][0], s[qt][1]);
            pf[qt][1] = pack_pair(s[qt][2], s[qt][3]);
            lacc[qt] = MFMA16(ones, pf[qt][1], MFMA16(ones, pf[qt][0], lacc[qt]));
        };
        float resc[2] = {1.f, 1.f};                                       // rescale of O, applied behind the paths' merge point (in place)
        // exact path: s' + m_ref (+ key bias, causal compare) -> true row maximum -> rescale -> new reference
        auto exact_tile = [&](auto tag) {
            constexpr bool MASKED = decltype(tag)::value;
            asm volatile("; exact softmax path" ::: "memory");          // keeps hipcc from speculating this (rare) path into the common one
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                const float mold = m[qt] == -INFINITY ? 0.f : m[qt];      // what the chains subtracted
                float mx = -INFINITY;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    f32x4 bias = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (MASKED) bias = *reinterpret_cast<const f32x4*>(ldsB + kt * 16 + g * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float x = s[qt][kt][r] + mold;
                        if constexpr (MASKED) {
                            x += bias[r];
                            if (diag && (k0 + kt * 16 + g * 4 + r) > myq[qt]) x = -INFINITY;
                        }
                        s[qt][kt][r] = x;
                        mx = fmaxf(mx, x);
                    }
                }
                mx = grp_max(mx);
                const float mnew = fmaxf(m[qt], mx);
                const float muse = mnew == -INFINITY ? 0.f : mnew;
                const float alpha = __builtin_amdgcn_exp2f(m[qt] - muse);
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[qt][kt][r] = __builtin_amdgcn_exp2f(s[qt][kt][r] - muse);
                lacc[qt] *= alpha;
                m[qt] = mnew;
                cinit[qt] = f32x4{-muse, -muse, -muse, -muse};
                resc[qt] = alpha;
                finish_row(qt);
            }
        };
        bool exact = masked;
        if (masked) {
            exact_tile(BoolTag<true>{});
        } else {
            // lazy path: does any score of the tile sit more than LAZY_THR above its row's reference (or has a row no reference)?
            float t0 = m[0] == -INFINITY ? INFINITY : -INFINITY, t1 = m[1] == -INFINITY ? INFINITY : -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { t0 = fmaxf(t0, s[0][kt][r]); t1 = fmaxf(t1, s[1][kt][r]); }
            if (__builtin_amdgcn_ballot_w64(fmaxf(t0, t1) > LAZY_THR) == 0ull) {
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) s[qt][kt][r] = __builtin_amdgcn_exp2f(s[qt][kt][r]);
                    finish_row(qt);
                }
            } else {
                exact = true;
                exact_tile(BoolTag<false>{});
            }
        }
        if (exact) {
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int i = 0; i < DT; ++i) oacc[qt][i] *= resc[qt];
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            if (nb > 0) {                                                 // next V image into the same registers (the compiler keeps the anti-dependence)
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) if (nb * 4 + dt < DT) ds_tr_block(tv[dt], voff[dt] + (unsigned)(sidx * C::STB + nb * 8192));
            }
            tr_wait4(tv);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) if (nb * 4 + dt < DT) {
                const bf16x8 v0 = tr_join(tv[dt][0][0], tv[dt][0][1]), v1 = tr_join(tv[dt][1][0], tv[dt][1][1]);
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) oacc[qt][nb * 4 + dt] = MFMA16(v1, pf[qt][1], MFMA16(v0, pf[qt][0], oacc[qt][nb * 4 + dt]));
            }
        }
        if (C::NS == 3 && it + 2 < nt) { wait_vm<C::PCS>(); } else { wait_vm<0>(); }
        __builtin_amdgcn_s_barrier();
        sidx = sidx == C::NS - 1 ? 0 : sidx + 1;
    }
    if constexpr (HD == 64) {
        // O leaves through LDS (round 5, late): the accumulator layout gives a lane 4 columns (8 bytes) of one row with the row's pieces on lanes 16 apart --
        // 8 store instructions of 64 separate 8-byte writes per wave, 5 - 7 % of the kernel (a build without them: 166 -> 156 us dense; the memory pipeline
        // merges adjacent lanes only). The wave's 32 x 64 tile goes into a 4 KiB image of the stage ring (free behind the loop's last barrier; 16-byte slot
        // XOR row & 7) and out as 4 stores of 8 rows x 128 contiguous bytes, 8 adjacent lanes per row.
        typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
        const unsigned obase = lds_u32(smem) + (unsigned)(wave * 4096);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const float lq = lacc[qt][0];
            const float inv = lq > 0.f ? 1.0f / lq : 0.f;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const bf16x4 r = {(bf16_t)(oacc[qt][dt][0] * inv), (bf16_t)(oacc[qt][dt][1] * inv), (bf16_t)(oacc[qt][dt][2] * inv), (bf16_t)(oacc[qt][dt][3] * inv)};
                *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(obase + (unsigned)(qt * 2048 + lr * 128 + ((((dt * 2 + (g >> 1)) ^ (lr & 7)) << 4) | ((g & 1) << 3)))) =
                    __builtin_bit_cast(u32x2, r);
            }
            if (myq[qt] < p.Sq && g == 0) p.lse[((long)b * p.H + h) * lse_ld + myq[qt]] = lq > 0.f ? (m[qt] + log2f(lq)) / LOG2E : INFINITY;   // m = the reference l was summed against
        }
        const int orow = lane >> 3, ochunk = lane & 7;
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int r = hh * 8 + orow, q = q0 + wave * 32 + qt * 16 + r;
                const u32x4 v = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(obase + (unsigned)(qt * 2048 + r * 128 + ((ochunk ^ (r & 7)) << 4)));
                if (q < p.Sq) *reinterpret_cast<u32x4*>(p.out + b * p.o_sb + (long)q * p.o_ss + h * HDT + ochunk * 8) = v;
            }
    } else {
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        if (myq[qt] < p.Sq) {
            const float lq = lacc[qt][0];
            const float inv = lq > 0.f ? 1.0f / lq : 0.f;
            bf16_t* O = p.out + b * p.o_sb + (long)myq[qt] * p.o_ss + h * HDT;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                bf16x4 r = {(bf16_t)(oacc[qt][dt][0] * inv), (bf16_t)(oacc[qt][dt][1] * inv), (bf16_t)(oacc[qt][dt][2] * inv), (bf16_t)(oacc[qt][dt][3] * inv)};
                *reinterpret_cast<bf16x4*>(O + dt * 16 + g * 4) = r;
            }
            if (g == 0) p.lse[((long)b * p.H + h) * lse_ld + myq[qt]] = lq > 0.f ? (m[qt] + log2f(lq)) / LOG2E : INFINITY;   // m = the reference l was summed against
        }
    }
    }
}

// ================================================================== backward dK, dV: block = 64 KT keys
// Per score the backward needs p = exp(s - lse) and ds = p (dp - delta). With K prescaled by c = scale * log2(e) and the
// row constants -lse * log2(e) and -delta loaded as the INITIAL ACCUMULATORS of the S and dP MFMA chains, that is one
// v_exp_f32 and one multiply: no fma, no subtract, no per-score mask (a masked key only dirties its own dK / dV rows, which
// are zeroed in the epilogue; a fully masked or out-of-range query row has -lse = -inf, so p = 0), and the softmax scale is
// applied once to dK. Only tiles on the causal diagonal compare. Pipeline as in the forward: DMA ring of {Q, dO}
// tiles behind a counted vmcnt, raw barriers, transposed fragments by asm; -lse, -delta of the whole row sit in LDS.
template <int HD>
__global__ __launch_bounds__(FT) void fa64_bwd_dkv_kernel(const Fa64Args pin) {
    constexpr int NB = (HD + 63) / 64, KS = HD / 32, DT = HD / 16, HDT = HD;
    using C = FaCfg<NB>;
    constexpr int KT = C::KT, BK_ = 64 * KT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, lr = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    int rb, h, b;
    block_map((pin.Sk + BK_ - 1) / BK_, pin.H, pin.B, rb, h, b, pin.bh_order);
    const int k0 = rb * BK_;
    Fa64Args p = pin;
    varlen_localize(p, b);
    const int lse_ld = pin.Sq;
    if (k0 >= p.Sk) {                                                     // packed rows: no such key block in this batch; its bias-gradient partials are zeros
        if (p.cs_kv && t < 2 * HDT) {
            const int nkb0 = (pin.Sk + BK_ - 1) / BK_, dm = p.H * HDT;
            float* row = p.cs_kv + (long)(b * nkb0 + rb) * 2 * dm + h * HDT;
            row[t < HDT ? t : dm + t - HDT] = 0.f;
        }
        return;
    }
    const bf16_t* Q = p.q + b * p.q_sb + h * HDT;
    const bf16_t* K = p.k + b * p.k_sb + h * HDT;
    const bf16_t* V = p.v + b * p.v_sb + h * HDT;
    const bf16_t* DO = p.dout + b * p.o_sb + h * HDT;
    const float c = p.scale * LOG2E;
    const int it0 = p.causal ? k0 / 64 : 0;
    // a key block that lies entirely in the masked tail receives no gradient: skip its whole query loop (zeros are written)
    const int nt = (p.kmax && k0 >= p.kmax[b]) ? 0 : (p.Sq + 63) / 64;
    const int sqp = ((p.Sq + 63) / 64) * 64;
    float* ldsNL = reinterpret_cast<float*>(smem + C::NS * C::STB);       // -lse * log2(e) per query (-inf: row contributes nothing)
    float* ldsND = ldsNL + sqp;                                          // -delta per query
    {   // all loads of a chunk of 4 x 256 queries are requested before the first is stored (one memory round trip per chunk, not per 256)
        const long li0 = ((long)b * p.H + h) * lse_ld;
        for (int qb = it0 * 64; qb < nt * 64; qb += 4 * FT) {
            float tl[4], td[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int q = qb + t + i * FT;
                tl[i] = q < p.Sq ? p.lse[li0 + q] : INFINITY;
                td[i] = q < p.Sq ? p.delta[li0 + q] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int q = qb + t + i * FT;
                if (q < nt * 64) { ldsNL[q] = tl[i] == INFINITY ? -INFINITY : -tl[i] * LOG2E; ldsND[q] = -td[i]; }
            }
        }
    }
    int mykey[KT];
    bf16x8 kf[KT][KS], vf[KT][KS];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        mykey[kt] = k0 + wave * (16 * KT) + kt * 16 + lr;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            kf[kt][ks] = scale_frag(frag_global(K, p.k_ss, mykey[kt], p.Sk, ks * 32 + g * 8), c);
            vf[kt][ks] = frag_global(V, p.v_ss, mykey[kt], p.Sk, ks * 32 + g * 8);
        }
    }
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(kf[kt][ks]), "+v"(vf[kt][ks]));
    f32x4 dk[KT][DT], dv[KT][DT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int i = 0; i < DT; ++i) { dk[kt][i] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[kt][i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    unsigned qoff[4], ooff[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { qoff[dt] = lds_u32(smem) + tr_lane_off64(dt * 16, lane); ooff[dt] = qoff[dt] + NB * 8192; }
    const StageOff so_q = stage_off(p.q_ss, wave, lane), so_o = stage_off(p.o_ss, wave, lane);
    auto stage = [&](int it, int sidx) {
        char* st = smem + sidx * C::STB;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            stage64(Q + nb * 64, p.q_ss, it * 64, p.Sq, st + nb * 8192, wave, lane, so_q, min(8, (HD - nb * 64) / 8), p.zeros);
            stage64(DO + nb * 64, p.o_ss, it * 64, p.Sq, st + (NB + nb) * 8192, wave, lane, so_o, min(8, (HD - nb * 64) / 8), p.zeros);
        }
    };
    if (it0 < nt) stage(it0, 0);
    if (C::NS == 3 && it0 + 1 < nt) { stage(it0 + 1, 1); wait_vm<C::PCS>(); } else { wait_vm<0>(); }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
    int sidx = 0;
    for (int it = it0; it < nt; ++it) {
        const char* st = smem + sidx * C::STB;
        const int nidx = sidx == 0 ? C::NS - 1 : sidx - 1;
        if (it + C::NS - 1 < nt) stage(it + C::NS - 1, nidx);
        const int q0 = it * 64;
        const bool diag = p.causal && (k0 + wave * (16 * KT) + 16 * KT - 1 > q0);   // wave-uniform: this q tile can be below some of the wave's keys
        bf16x8 pf[KT][2], df[KT][2];
#pragma unroll
        for (int half = 0; half < 2; ++half) {            // q tiles (2 half, 2 half + 1) -> one k-step of the dV/dK products
            f32x4 sv[KT][2], dp[KT][2];
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                const int qt = half * 2 + qq;
                const f32x4 nl = *reinterpret_cast<const f32x4*>(ldsNL + q0 + qt * 16 + g * 4);
                const f32x4 nd = *reinterpret_cast<const f32x4*>(ldsND + q0 + qt * 16 + g * 4);
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) { sv[kt][qq] = nl; dp[kt][qq] = nd; }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const char* ldsQ = st + nb * 8192; const char* ldsO = st + (NB + nb) * 8192;
#pragma unroll
                    for (int k2 = 0; k2 < 2; ++k2)
                        if (2 * nb + k2 < KS) {
                            const bf16x8 qa = frag_row(ldsQ, qt * 16 + lr, k2, g), oa = frag_row(ldsO, qt * 16 + lr, k2, g);
#pragma unroll
                            for (int kt = 0; kt < KT; ++kt) {
                                sv[kt][qq] = MFMA16(qa, kf[kt][2 * nb + k2], sv[kt][qq]);
                                dp[kt][qq] = MFMA16(oa, vf[kt][2 * nb + k2], dp[kt][qq]);
                            }
                        }
                }
            }
            if (diag) {
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                    for (int qq = 0; qq < 2; ++qq)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int q = q0 + (half * 2 + qq) * 16 + g * 4 + r;
                            const float pr = mykey[kt] <= q ? __builtin_amdgcn_exp2f(sv[kt][qq][r]) : 0.f;
                            sv[kt][qq][r] = pr; dp[kt][qq][r] *= pr;
                        }
            } else {
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                    for (int qq = 0; qq < 2; ++qq)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float pr = __builtin_amdgcn_exp2f(sv[kt][qq][r]);
                            sv[kt][qq][r] = pr; dp[kt][qq][r] *= pr;
                        }
            }
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) { pf[kt][half] = pack_pair(sv[kt][0], sv[kt][1]); df[kt][half] = pack_pair(dp[kt][0], dp[kt][1]); }
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int si = 0; si < 2; ++si) {                                   // k-step si of the dV / dK products = q rows 32 si .. 32 si + 31
            s16x4 to[4][2], tq[4][2];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) if (nb * 4 + dt < DT) {
                const unsigned ao = ooff[dt] + (unsigned)(sidx * C::STB + nb * 8192), aq = qoff[dt] + (unsigned)(sidx * C::STB + nb * 8192);
                if (si == 0) { ds_tr<0>(to[dt][0], ao); ds_tr<2048>(to[dt][1], ao); ds_tr<0>(tq[dt][0], aq); ds_tr<2048>(tq[dt][1], aq); }
                else { ds_tr<4096>(to[dt][0], ao); ds_tr<6144>(to[dt][1], ao); ds_tr<4096>(tq[dt][0], aq); ds_tr<6144>(tq[dt][1], aq); }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(to[0][0]), "+v"(to[0][1]), "+v"(to[1][0]), "+v"(to[1][1]), "+v"(to[2][0]), "+v"(to[2][1]), "+v"(to[3][0]), "+v"(to[3][1]),
                                                  "+v"(tq[0][0]), "+v"(tq[0][1]), "+v"(tq[1][0]), "+v"(tq[1][1]), "+v"(tq[2][0]), "+v"(tq[2][1]), "+v"(tq[3][0]), "+v"(tq[3][1]));
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) if (nb * 4 + dt < DT) {
                const bf16x8 ot = tr_join(to[dt][0], to[dt][1]), qtf = tr_join(tq[dt][0], tq[dt][1]);
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    // A = the transposed dO / Q fragment (row = column), B = P^T / dS^T (column = key): a lane ends up with 4 consecutive
                    // columns of ONE key per tile = an 8-byte store (the other operand order leaves 4 keys of one column: 2-byte stores,
                    // and the store tail is paid per instruction, not per byte)
                    dv[kt][nb * 4 + dt] = MFMA16(ot, pf[kt][si], dv[kt][nb * 4 + dt]);
                    dk[kt][nb * 4 + dt] = MFMA16(qtf, df[kt][si], dk[kt][nb * 4 + dt]);
                }
            }
        }
        if (C::NS == 3 && it + 2 < nt) { wait_vm<C::PCS>(); } else { wait_vm<0>(); }
        __builtin_amdgcn_s_barrier();
        sidx = sidx == C::NS - 1 ? 0 : sidx + 1;
    }
    // tile (kt, dt): lane = key 16 kt + lr of this wave, registers = columns 16 dt + 4 g + r
    f32x4 csk[DT], csv[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) { csk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; csv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    if constexpr (HD == 64) {
        // dK / dV rows leave through LDS images like the forward's O (8 rows x 128 contiguous bytes per store instead of 8-byte pieces on lanes 16 apart);
        // the images lie behind the 4 KiB the column-sum reduction below uses
        typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
        const unsigned obase = lds_u32(smem) + 4096u + (unsigned)(wave * (KT * 4096));
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const int key = k0 + wave * (16 * KT) + kt * 16 + lr;
            const bool kvis = pin.vl_q_off ? key < (p.kmax ? p.kmax[b] : p.Sk)
                                           : (!p.key_mask || p.key_mask[(long)b * p.Sk + (key < p.Sk ? key : 0)] != 0.f);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                f32x4 vk = dk[kt][dt] * p.scale, vv = dv[kt][dt];
                if (!kvis) { vk = f32x4{0.f, 0.f, 0.f, 0.f}; vv = f32x4{0.f, 0.f, 0.f, 0.f}; }
                const bf16x4 rk = {(bf16_t)vk[0], (bf16_t)vk[1], (bf16_t)vk[2], (bf16_t)vk[3]}, rv = {(bf16_t)vv[0], (bf16_t)vv[1], (bf16_t)vv[2], (bf16_t)vv[3]};
                const unsigned slot = (unsigned)(lr * 128 + ((((dt * 2 + (g >> 1)) ^ (lr & 7)) << 4) | ((g & 1) << 3)));
                *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(obase + (unsigned)(kt * 4096) + slot) = __builtin_bit_cast(u32x2, rk);
                *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(obase + (unsigned)(kt * 4096 + 2048) + slot) = __builtin_bit_cast(u32x2, rv);
                if (key < p.Sk) { csk[dt] += vk; csv[dt] += vv; }
            }
        }
        const int orow = lane >> 3, ochunk = lane & 7;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int r = hh * 8 + orow, key = k0 + wave * (16 * KT) + kt * 16 + r;
                const unsigned slot = (unsigned)(r * 128 + ((ochunk ^ (r & 7)) << 4));
                const u32x4 vk = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(obase + (unsigned)(kt * 4096) + slot);
                const u32x4 vv = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(obase + (unsigned)(kt * 4096 + 2048) + slot);
                if (key < p.Sk) {
                    *reinterpret_cast<u32x4*>(p.dk + b * p.dk_sb + (long)key * p.dk_ss + h * HDT + ochunk * 8) = vk;
                    *reinterpret_cast<u32x4*>(p.dv + b * p.dv_sb + (long)key * p.dv_ss + h * HDT + ochunk * 8) = vv;
                }
            }
    } else {
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        const int key = k0 + wave * (16 * KT) + kt * 16 + lr;
        const bool kvis = pin.vl_q_off ? key < (p.kmax ? p.kmax[b] : p.Sk)                       // packed rows: the visible keys are a prefix
                                       : (!p.key_mask || p.key_mask[(long)b * p.Sk + (key < p.Sk ? key : 0)] != 0.f);    // a masked key receives no gradient
        bf16_t* DK = p.dk + b * p.dk_sb + (long)key * p.dk_ss + h * HDT + g * 4;
        bf16_t* DV = p.dv + b * p.dv_sb + (long)key * p.dv_ss + h * HDT + g * 4;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            f32x4 vk = dk[kt][dt] * p.scale, vv = dv[kt][dt];
            if (!kvis) { vk = f32x4{0.f, 0.f, 0.f, 0.f}; vv = f32x4{0.f, 0.f, 0.f, 0.f}; }
            if (key < p.Sk) {
                bf16x4 rk = {(bf16_t)vk[0], (bf16_t)vk[1], (bf16_t)vk[2], (bf16_t)vk[3]}, rv = {(bf16_t)vv[0], (bf16_t)vv[1], (bf16_t)vv[2], (bf16_t)vv[3]};
                *reinterpret_cast<bf16x4*>(DK + dt * 16) = rk;
                *reinterpret_cast<bf16x4*>(DV + dt * 16) = rv;
                csk[dt] += vk; csv[dt] += vv;
            }
        }
    }
    }
    if (p.cs_kv) {      // bias gradients: column sums of the block's dK / dV rows -> partial row (b, key block), head h's columns
        const int nkb = (pin.Sk + BK_ - 1) / BK_, d_model = p.H * HDT;
        float* red = reinterpret_cast<float*>(smem);                      // [4 waves][2 HDT]: the tile ring is free after the last barrier
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {                                 // sum over the 16 keys of a DPP row: row_mirror, row_half_mirror, quad_perm
                float v = csk[dt][e], u = csv[dt][e];
                v += PB_DPP_F(v, 0x140); u += PB_DPP_F(u, 0x140);
                v += PB_DPP_F(v, 0x141); u += PB_DPP_F(u, 0x141);
                v += PB_DPP_F(v, 0x4e); u += PB_DPP_F(u, 0x4e);
                v += PB_DPP_F(v, 0xb1); u += PB_DPP_F(u, 0xb1);
                csk[dt][e] = v; csv[dt][e] = u;
            }
            if (lr == 0) {
                *reinterpret_cast<f32x4*>(red + wave * 2 * HDT + dt * 16 + g * 4) = csk[dt];
                *reinterpret_cast<f32x4*>(red + wave * 2 * HDT + HDT + dt * 16 + g * 4) = csv[dt];
            }
        }
        __syncthreads();
        if (t < 2 * HDT) {
            float* row = p.cs_kv + (long)(b * nkb + rb) * 2 * d_model + h * HDT;
            const float v = red[t] + red[2 * HDT + t] + red[4 * HDT + t] + red[6 * HDT + t];
            row[t < HDT ? t : d_model + t - HDT] = v;
        }
    }
}

// ================================================================== backward dQ: block = 128 queries
// Same arithmetic with the roles swapped: Q (prescaled by c) and dO of the wave's 32 queries stay in registers, -lse * log2(e)
// and -delta are per-lane constants splatted into the initial accumulators, K / V tiles stream through the DMA ring, and a
// tile that holds a masked key adds the 0 / -inf key bias before the exp2 (no select).
template <int HD>
__global__ __launch_bounds__(FT) void fa64_bwd_dq_kernel(const Fa64Args pin) {
    constexpr int NB = (HD + 63) / 64, KS = HD / 32, DT = HD / 16, HDT = HD;
    using C = FaCfg<NB>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, lr = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    int rb, h, b;
    block_map((pin.Sq + 127) / 128, pin.H, pin.B, rb, h, b, pin.bh_order);
    if (pin.causal) rb = (pin.Sq + 127) / 128 - 1 - rb;        // causal: a later row block visits more keys -- the long blocks of a (batch, head) are dispatched first
    const int q0 = rb * 128;
    Fa64Args p = pin;
    varlen_localize(p, b);
    const int lse_ld = pin.Sq;
    if (q0 >= p.Sq) {                                                     // packed rows: no such row block in this batch; its bias-gradient partial is zero
        if (p.cs_q && t < HDT) p.cs_q[(long)(b * ((pin.Sq + 127) / 128) + rb) * (p.H * HDT) + h * HDT + t] = 0.f;
        return;
    }
    const bf16_t* Q = p.q + b * p.q_sb + h * HDT;
    const bf16_t* K = p.k + b * p.k_sb + h * HDT;
    const bf16_t* V = p.v + b * p.v_sb + h * HDT;
    const bf16_t* DO = p.dout + b * p.o_sb + h * HDT;
    const float c = p.scale * LOG2E;
    const int kvis_end = p.kmax ? min(p.Sk, p.kmax[b]) : p.Sk;
    const int kend = p.causal ? min(kvis_end, q0 + 128) : kvis_end;
    const int nt = (kend + 63) / 64;
    float* ldsBias = reinterpret_cast<float*>(smem + C::NS * C::STB);
    unsigned* ldsFlag = reinterpret_cast<unsigned*>(ldsBias + ((p.Sk + 63) / 64) * 64);
    for (int tile = wave; tile < nt; tile += 4) {
        const int key = tile * 64 + lane;
        const bool vis = key < (pin.vl_q_off ? kvis_end : p.Sk) && (!p.key_mask || p.key_mask[(long)b * p.Sk + key] != 0.f);
        ldsBias[key] = vis ? 0.f : -INFINITY;
        const bool allvis = __builtin_amdgcn_ballot_w64(vis) == ~0ull;
        if (lane == 0) ldsFlag[tile] = allvis ? 0u : 1u;
    }
    int myq[2];
    bf16x8 qf[2][KS], of[2][KS];
    float nl[2], nd[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        myq[qt] = q0 + wave * 32 + qt * 16 + lr;
        const long li = ((long)b * p.H + h) * lse_ld + myq[qt];
        const float ls = myq[qt] < p.Sq ? p.lse[li] : INFINITY;
        nl[qt] = ls == INFINITY ? -INFINITY : -ls * LOG2E;
        // delta = rowsum(dO . O) of this lane's query, computed here (this kernel owns whole query rows and already holds dO) and
        // published for the dK/dV kernel, which is launched after this one
        float dl = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qf[qt][ks] = scale_frag(frag_global(Q, p.q_ss, myq[qt], p.Sq, ks * 32 + g * 8), c);
            of[qt][ks] = frag_global(DO, p.o_ss, myq[qt], p.Sq, ks * 32 + g * 8);
            const bf16x8 ov = frag_global(p.o + b * p.o_sb + h * HDT, p.o_ss, myq[qt], p.Sq, ks * 32 + g * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) dl = fmaf((float)of[qt][ks][e], (float)ov[e], dl);
        }
        dl = grp_sum(dl);
        nd[qt] = -dl;
        if (g == 0 && myq[qt] < p.Sq) const_cast<float*>(p.delta)[li] = dl;
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[qt][ks]), "+v"(of[qt][ks]));
        asm volatile("" : "+v"(nl[qt]), "+v"(nd[qt]));
    }
    f32x4 dq[2][DT];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int i = 0; i < DT; ++i) dq[qt][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned koff[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) koff[dt] = lds_u32(smem) + tr_lane_off64(dt * 16, lane);
    const StageOff so_k = stage_off(p.k_ss, wave, lane), so_v = stage_off(p.v_ss, wave, lane);
    auto stage = [&](int it, int sidx) {
        char* st = smem + sidx * C::STB;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            stage64(K + nb * 64, p.k_ss, it * 64, p.Sk, st + nb * 8192, wave, lane, so_k, min(8, (HD - nb * 64) / 8), p.zeros);
            stage64(V + nb * 64, p.v_ss, it * 64, p.Sk, st + (NB + nb) * 8192, wave, lane, so_v, min(8, (HD - nb * 64) / 8), p.zeros);
        }
    };
    if (nt > 0) stage(0, 0);
    if (C::NS == 3 && nt > 1) { stage(1, 1); wait_vm<C::PCS>(); } else { wait_vm<0>(); }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
    int sidx = 0;
    for (int it = 0; it < nt; ++it) {
        const char* st = smem + sidx * C::STB;
        const int nidx = sidx == 0 ? C::NS - 1 : sidx - 1;
        if (it + C::NS - 1 < nt) stage(it + C::NS - 1, nidx);
        const float* ldsB = ldsBias + it * 64;
        const int k0 = it * 64;
        const bool diag = p.causal && (k0 + 63 > q0 + wave * 32);
        const bool masked = __builtin_amdgcn_readfirstlane(ldsFlag[it]) != 0u;
        bf16x8 df[2][2];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x4 ds_[2][2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int kt = half * 2 + kk;
                f32x4 sv[2], dpv[2];
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) { sv[qt] = f32x4{nl[qt], nl[qt], nl[qt], nl[qt]}; dpv[qt] = f32x4{nd[qt], nd[qt], nd[qt], nd[qt]}; }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const char* ldsK = st + nb * 8192; const char* ldsV = st + (NB + nb) * 8192;
#pragma unroll
                    for (int k2 = 0; k2 < 2; ++k2)
                        if (2 * nb + k2 < KS) {
                            const bf16x8 ka = frag_row(ldsK, kt * 16 + lr, k2, g), va = frag_row(ldsV, kt * 16 + lr, k2, g);
#pragma unroll
                            for (int qt = 0; qt < 2; ++qt) {
                                sv[qt] = MFMA16(ka, qf[qt][2 * nb + k2], sv[qt]);
                                dpv[qt] = MFMA16(va, of[qt][2 * nb + k2], dpv[qt]);
                            }
                        }
                }
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) {
                    if (masked) sv[qt] += *reinterpret_cast<const f32x4*>(ldsB + kt * 16 + g * 4);
                    if (diag) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) if (k0 + kt * 16 + g * 4 + r > myq[qt]) sv[qt][r] = -INFINITY;
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) dpv[qt][r] *= __builtin_amdgcn_exp2f(sv[qt][r]);
                    ds_[qt][kk] = dpv[qt];
                }
            }
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) df[qt][half] = pack_pair(ds_[qt][0], ds_[qt][1]);
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            s16x4 tk[4][2][2];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) if (nb * 4 + dt < DT) ds_tr_block(tk[dt], koff[dt] + (unsigned)(sidx * C::STB + nb * 8192));
            tr_wait4(tk);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) if (nb * 4 + dt < DT) {
                const bf16x8 k0f = tr_join(tk[dt][0][0], tk[dt][0][1]), k1f = tr_join(tk[dt][1][0], tk[dt][1][1]);
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) dq[qt][nb * 4 + dt] = MFMA16(k1f, df[qt][1], MFMA16(k0f, df[qt][0], dq[qt][nb * 4 + dt]));
            }
        }
        if (C::NS == 3 && it + 2 < nt) { wait_vm<C::PCS>(); } else { wait_vm<0>(); }
        __builtin_amdgcn_s_barrier();
        sidx = sidx == C::NS - 1 ? 0 : sidx + 1;
    }
    f32x4 csq[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) csq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (HD == 64) {
        // dQ rows through LDS images, as the forward's O
        typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
        const unsigned obase = lds_u32(smem) + 4096u + (unsigned)(wave * 4096);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const f32x4 v = dq[qt][dt] * p.scale;
                const bf16x4 r = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(obase + (unsigned)(qt * 2048 + lr * 128 + ((((dt * 2 + (g >> 1)) ^ (lr & 7)) << 4) | ((g & 1) << 3)))) =
                    __builtin_bit_cast(u32x2, r);
                if (myq[qt] < p.Sq) csq[dt] += v;
            }
        const int orow = lane >> 3, ochunk = lane & 7;
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int r = hh * 8 + orow, q = q0 + wave * 32 + qt * 16 + r;
                const u32x4 v = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(obase + (unsigned)(qt * 2048 + r * 128 + ((ochunk ^ (r & 7)) << 4)));
                if (q < p.Sq) *reinterpret_cast<u32x4*>(p.dq + b * p.dq_sb + (long)q * p.dq_ss + h * HDT + ochunk * 8) = v;
            }
    } else {
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
        if (myq[qt] < p.Sq) {
            bf16_t* DQ = p.dq + b * p.dq_sb + (long)myq[qt] * p.dq_ss + h * HDT;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const f32x4 v = dq[qt][dt] * p.scale;
                bf16x4 r = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                *reinterpret_cast<bf16x4*>(DQ + dt * 16 + g * 4) = r;
                csq[dt] += v;
            }
        }
    }
    if (p.cs_q) {       // bias gradient of the q projection: the 16 lanes of a DPP row hold 16 queries of the same 4 columns
        const int nqb = (pin.Sq + 127) / 128, d_model = p.H * HDT;
        float* red = reinterpret_cast<float*>(smem);                      // [4 waves][HDT]
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = csq[dt][e];
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));   // row_mirror
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));   // row_half_mirror
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4e, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xb1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
                csq[dt][e] = v;
            }
            if (lr == 0) *reinterpret_cast<f32x4*>(red + wave * HDT + dt * 16 + g * 4) = csq[dt];
        }
        __syncthreads();
        if (t < HDT) p.cs_q[(long)(b * nqb + rb) * d_model + h * HDT + t] = red[t] + red[HDT + t] + red[2 * HDT + t] + red[3 * HDT + t];
    }
}

}  // namespace

// 256 bytes of device zeros per device, allocated on first use (never freed: lives as long as the process' HIP context)
static const bf16_t* fa_zero_page() {
    static void* pages[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!pages[dev]) {
        void* p = nullptr;
        if (hipMalloc(&p, 256) != hipSuccess || hipMemset(p, 0, 256) != hipSuccess) return nullptr;
        pages[dev] = p;
    }
    return (const bf16_t*)pages[dev];
}

// entry points used by pb_flash.hip's dispatch (same argument meaning as pb_flash_fwd / pb_flash_bwd; hd = 64, 96 or 128)
template <int HD>
static int fa64_fwd_launch(const Fa64Args& a, hipStream_t stream) {
    using C = FaCfg<(HD + 63) / 64>;
    const size_t lds = (size_t)C::NS * C::STB + (size_t)((a.Sk + 63) / 64) * (64 * 4 + 4);
    PB_REQUIRE(lds <= 160 * 1024, "pb_flash_fwd: Sk=%d needs %zu bytes of LDS", a.Sk, lds);
    if (lds > 65536) hipFuncSetAttribute(reinterpret_cast<const void*>(fa64_fwd_kernel<HD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(fa64_fwd_kernel<HD>, dim3(((a.Sq + 127) / 128) * a.H * a.B), dim3(FT), lds, stream, a);
    PB_LAUNCH_CHECK();
    return 0;
}

int pb_flash64_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const float* key_mask, const int* kmax, int B, int H, int Sq, int Sk, int hd,
                   long q_sb, long q_ss, long k_sb, long k_ss, long v_sb, long v_ss, long o_sb, long o_ss, float scale, int causal, hipStream_t stream,
                   const int* const* vl) {
    Fa64Args a = {};
    if (vl) { a.vl_q_off = vl[0]; a.vl_q_len = vl[1]; a.vl_k_off = vl[2]; a.vl_k_len = vl[3]; a.bh_order = vl[4]; }
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.out = (bf16_t*)o; a.lse = lse; a.key_mask = key_mask; a.kmax = kmax;
    a.B = B; a.H = H; a.Sq = Sq; a.Sk = Sk; a.q_sb = q_sb; a.q_ss = q_ss; a.k_sb = k_sb; a.k_ss = k_ss; a.v_sb = v_sb; a.v_ss = v_ss;
    a.o_sb = o_sb; a.o_ss = o_ss; a.scale = scale; a.causal = causal;
    a.zeros = fa_zero_page();
    PB_REQUIRE(a.zeros != nullptr, "pb_flash_fwd: cannot allocate the zero page");
    return hd == 128 ? fa64_fwd_launch<128>(a, stream) : hd == 96 ? fa64_fwd_launch<96>(a, stream) : fa64_fwd_launch<64>(a, stream);
}

template <int HD>
static int fa64_bwd_launch(const Fa64Args& a, hipStream_t stream) {
    using C = FaCfg<(HD + 63) / 64>;
    const size_t lds_dkv = (size_t)C::NS * C::STB + (size_t)((a.Sq + 63) / 64) * 64 * 8;
    const size_t lds_dq = (size_t)C::NS * C::STB + (size_t)((a.Sk + 63) / 64) * (64 * 4 + 4);
    PB_REQUIRE(lds_dkv <= 160 * 1024 && lds_dq <= 160 * 1024, "pb_flash_bwd: Sq=%d Sk=%d need %zu / %zu bytes of LDS", a.Sq, a.Sk, lds_dkv, lds_dq);
    if (lds_dkv > 65536) hipFuncSetAttribute(reinterpret_cast<const void*>(fa64_bwd_dkv_kernel<HD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dkv);
    if (lds_dq > 65536) hipFuncSetAttribute(reinterpret_cast<const void*>(fa64_bwd_dq_kernel<HD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dq);
    constexpr int BK_ = 64 * C::KT;
    hipLaunchKernelGGL(fa64_bwd_dq_kernel<HD>, dim3(((a.Sq + 127) / 128) * a.H * a.B), dim3(FT), lds_dq, stream, a);      // also writes delta
    PB_LAUNCH_CHECK();
    hipLaunchKernelGGL(fa64_bwd_dkv_kernel<HD>, dim3(((a.Sk + BK_ - 1) / BK_) * a.H * a.B), dim3(FT), lds_dkv, stream, a);
    PB_LAUNCH_CHECK();
    return 0;
}

int pb_flash64_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse, float* delta, const float* key_mask,
                   const int* kmax, void* dq, void* dk, void* dv, int B, int H, int Sq, int Sk, int hd, long q_sb, long q_ss, long k_sb, long k_ss, long v_sb,
                   long v_ss, long o_sb, long o_ss, long dq_sb, long dq_ss, long dk_sb, long dk_ss, long dv_sb, long dv_ss, float scale,
                   int causal, float* dbias_q, float* dbias_k, float* dbias_v, float* dbias_ws, hipStream_t stream, const int* const* vl) {
    Fa64Args a = {};
    if (vl) { a.vl_q_off = vl[0]; a.vl_q_len = vl[1]; a.vl_k_off = vl[2]; a.vl_k_len = vl[3]; a.bh_order = vl[4]; }
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.o = (const bf16_t*)o; a.dout = (const bf16_t*)dout;
    a.dq = (bf16_t*)dq; a.dk = (bf16_t*)dk; a.dv = (bf16_t*)dv; a.lse = const_cast<float*>(lse); a.delta = delta; a.key_mask = key_mask; a.kmax = kmax;
    a.B = B; a.H = H; a.Sq = Sq; a.Sk = Sk; a.q_sb = q_sb; a.q_ss = q_ss; a.k_sb = k_sb; a.k_ss = k_ss; a.v_sb = v_sb; a.v_ss = v_ss;
    a.o_sb = o_sb; a.o_ss = o_ss; a.dq_sb = dq_sb; a.dq_ss = dq_ss; a.dk_sb = dk_sb; a.dk_ss = dk_ss; a.dv_sb = dv_sb; a.dv_ss = dv_ss;
    a.scale = scale; a.causal = causal;
    a.zeros = fa_zero_page();
    PB_REQUIRE(a.zeros != nullptr, "pb_flash_bwd: cannot allocate the zero page");
    const int kt_ = hd == 64 ? 2 : 1, nkb = (Sk + 64 * kt_ - 1) / (64 * kt_), nqb = (Sq + 127) / 128, d_model = H * hd;
    if (dbias_q) {
        PB_REQUIRE(dbias_k && dbias_v && dbias_ws, "pb_flash_bwd: dbias_q/k/v and dbias_ws go together");
        const size_t n_kv = (size_t)B * nkb * 2 * d_model;
        if (float* slice = pb_defer_alloc(n_kv + (size_t)B * nqb * d_model)) dbias_ws = slice;     // deferred reduction: the partial rows must outlive this call
        a.cs_kv = dbias_ws; a.cs_q = dbias_ws + n_kv;
    }
    const int rc = hd == 128 ? fa64_bwd_launch<128>(a, stream) : hd == 96 ? fa64_bwd_launch<96>(a, stream) : fa64_bwd_launch<64>(a, stream);
    if (rc || !dbias_q) return rc;
    if (pb_finalize_rows(a.cs_kv, B * nkb, d_model, dbias_k, stream, 2, dbias_v)) return -1;
    return pb_finalize_rows(a.cs_q, B * nqb, d_model, dbias_q, stream);
}

// floats of workspace for the fused bias gradients of pb_flash_bwd (head_dim 64 / 96 / 128)
extern "C" int64_t pb_flash_bias_ws_floats(int32_t B, int32_t H, int32_t Sq, int32_t Sk, int32_t hd) {
    const int kt_ = hd == 64 ? 2 : 1, nkb = (Sk + 64 * kt_ - 1) / (64 * kt_), nqb = (Sq + 63) / 64;      // 64-row query chunks: the one-pass backward's partial rows (pb_flash1.hip)
    return (int64_t)B * H * hd * (2 * nkb + nqb);
}
