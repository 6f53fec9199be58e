// K4b: attention backward in ONE pass over the (key block, query tile) pairs, head_dim 64 (cfg 2 / cfg 5 head size).
// Math: tf:modeling_bart.py:115-140 (softmax(Q K^T / sqrt(hd) + mask) V) differentiated; masks /root/reference/PianoBart.py:76.
//
// The two-kernel backward of pb_flash64.hip recomputes S = Q K^T and dP = dO V^T twice (7 matrix products, 2 exp passes). Here a
// workgroup is KEY-STATIONARY: 4 waves x 64 keys = 256 keys of one (batch, head); each wave keeps dK^T and dV^T of its 64 keys in
// 128 ACCUMULATOR registers (AGPRs, one wave per SIMD: the whole 512-register file) while the workgroup sweeps the query tiles:
//   S^T, dP^T   : key on the MFMA lane, so their accumulators are the A operands of the dV / dK products as they stand
//                 (permuted-k enumeration, pb_fa_tiles.h); -lse and -delta ride in as the chains' initial accumulators;
//   dS          : crosses LDS once ([key][16 q] sub-images, 32-B rows, slot swizzle: conflict-free 8-byte writes and
//                 ds_read_b64_tr_b16 reads) for dQ^T = K^T dS^T, whose 256-key contraction is split over the 4 waves by output tile;
//   dQ          : every key block writes its partial as bf16 into ITS OWN slab (no atomics, no ordering between workgroups);
//                 fa1_reduce_kernel sums a row's slabs in f32 in key-block order, rounds once and emits the q-bias gradient partials.
// 5 products and one exp pass; K and V fragments never leave the registers; Q / dO tiles arrive by LDS-DMA in a 3-deep ring.
// delta = rowsum(dO . O) comes from fa1_delta_kernel (one streaming pass in front).
#include "pb_common.h"
#include "pb_fa_tiles.h"

namespace {

constexpr int KB1 = 256;                       // keys per workgroup
constexpr int RING1 = 3, STB1 = 2 * 8192;      // ring of {Q tile, dO tile} images, 64 queries each
constexpr int OFF_K = RING1 * STB1;            // 4 K images [64 keys][128 B] (source of the K^T fragments of dQ)
constexpr int OFF_DS = OFF_K + 4 * 8192;       // dS: [buffer 2][q tile 2][256 keys][32 B]
constexpr int OFF_TAB = OFF_DS + 2 * 2 * 8192; // -lse * log2(e) and -delta of every query of the sequence

struct Fa1Args {
    Fa64Args a;
    bf16_t* slab;                              // dQ partial of key block j: slab + j * slab_stride, rows as the q rows, row stride H * 64
    long slab_stride, slab_sb;                 // slab_sb: batch stride of a slab (dense layout; unused with packed rows)
};

// dK^T / dV^T accumulate chains live in AGPRs: nothing but MFMAs touches them before the epilogue (VALU cannot address AGPRs, so the
// compiler could not keep them there on its own without copies)
__device__ __forceinline__ void mfma_agpr(f32x4& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ bf16x4 to_bf4(const f32x4& v) {
    bf16x4 r = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    return r;
}
template <int V> struct IntTag { static constexpr int value = V; };
__device__ __forceinline__ bf16x8 join4(const bf16x4& lo, const bf16x4& hi) { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); }

__global__ __launch_bounds__(FT) void fa1_bwd_kernel(const Fa1Args pin) {
    constexpr int HDT = 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, lr = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nkb0 = (pin.a.Sk + KB1 - 1) / KB1;
    int rb, h, b;
    block_map(nkb0, pin.a.H, pin.a.B, rb, h, b);
    const int k0 = rb * KB1;
    Fa64Args p = pin.a;
    varlen_localize(p, b);
    const int lse_ld = pin.a.Sq, d_model = p.H * HDT;
    const int kvis_end = p.kmax ? min(p.Sk, p.kmax[b]) : p.Sk;
    const int nt = (p.Sq + 63) / 64;
    const int it0 = p.causal ? k0 / 64 : 0;
    const bf16_t* K = p.k + b * p.k_sb + h * HDT;
    const bf16_t* V = p.v + b * p.v_sb + h * HDT;
    if (k0 >= kvis_end || it0 >= nt) {
        // no visible key in this block, or no query that sees it: its keys receive zero gradient, its dQ slab is not read
        for (int i = t; i < KB1 * 8; i += FT) {
            const int key = k0 + (i >> 3), ch = i & 7;
            if (key < p.Sk) {
                const bf16x8 z = {};
                *reinterpret_cast<bf16x8*>(p.dk + b * p.dk_sb + (long)key * p.dk_ss + h * HDT + ch * 8) = z;
                *reinterpret_cast<bf16x8*>(p.dv + b * p.dv_sb + (long)key * p.dv_ss + h * HDT + ch * 8) = z;
            }
        }
        if (p.cs_kv && t < 2 * HDT) {
            float* row = p.cs_kv + (long)(b * nkb0 + rb) * 2 * d_model + h * HDT;
            row[t < HDT ? t : d_model + t - HDT] = 0.f;
        }
        return;
    }
    const bf16_t* Q = p.q + b * p.q_sb + h * HDT;
    const bf16_t* DO = p.dout + b * p.o_sb + h * HDT;
    bf16_t* slab = pin.slab + (long)rb * pin.slab_stride + (pin.a.vl_q_off ? (long)pin.a.vl_q_off[b] * d_model : (long)b * pin.slab_sb) + h * HDT;
    const float c = p.scale * LOG2E;
    float* ldsNL = reinterpret_cast<float*>(smem + OFF_TAB);
    float* ldsND = ldsNL + nt * 64;
    for (int q = it0 * 64 + t; q < nt * 64; q += FT) {
        const long li = ((long)b * p.H + h) * lse_ld + q;
        const float ls = q < p.Sq ? p.lse[li] : INFINITY;
        ldsNL[q] = ls == INFINITY ? -INFINITY : -ls * LOG2E;
        ldsND[q] = q < p.Sq ? -p.delta[li] : 0.f;
    }
    // this wave's 64 keys: K (prescaled: S comes out of the MFMA in log2 units) and V fragments stay in registers for the whole sweep
    int mykey[4];
    float kb[4];
    bf16x8 kf[4][2], vf[4][2];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        mykey[kt] = k0 + wave * 64 + kt * 16 + lr;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            kf[kt][ks] = scale_frag(frag_global(K, p.k_ss, mykey[kt], p.Sk, ks * 32 + g * 8), c);
            vf[kt][ks] = frag_global(V, p.v_ss, mykey[kt], p.Sk, ks * 32 + g * 8);
        }
        const bool vis = mykey[kt] < kvis_end && (!p.key_mask || p.key_mask[(long)b * p.Sk + mykey[kt]] != 0.f);
        kb[kt] = vis ? 0.f : -INFINITY;                                   // key bias: a masked key's p is exactly 0 for every query
    }
    const bool anymask = __builtin_amdgcn_ballot_w64(kb[0] != 0.f || kb[1] != 0.f || kb[2] != 0.f || kb[3] != 0.f) != 0ull;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) asm volatile("" : "+v"(kf[kt][ks]), "+v"(vf[kt][ks]));     // ordinary loads are done before the first DMA
        asm volatile("" : "+v"(kb[kt]));
    }
    f32x4 dk[4][4], dv[4][4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int i = 0; i < 4; ++i) { dk[kt][i] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[kt][i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    // ---- DMA: the 4 K images of this block (for the K^T fragments), then the first two {Q, dO} tiles
    const StageOff so_k = stage_off(p.k_ss, wave, lane), so_q = stage_off(p.q_ss, wave, lane), so_o = stage_off(p.o_ss, wave, lane);
#pragma unroll
    for (int m = 0; m < 4; ++m) stage64(K, p.k_ss, k0 + 64 * m, p.Sk, smem + OFF_K + m * 8192, wave, lane, so_k);
    auto stage = [&](int it, int slot) {
        char* st = smem + slot * STB1;
        stage64(Q, p.q_ss, it * 64, p.Sq, st, wave, lane, so_q);
        stage64(DO, p.o_ss, it * 64, p.Sq, st + 8192, wave, lane, so_o);
    };
    stage(it0, 0);
    if (it0 + 1 < nt) { stage(it0 + 1, 1); wait_vm<8>(); } else { wait_vm<4>(); }
    __builtin_amdgcn_s_waitcnt(0xc07f);                                    // this wave's table stores
    __builtin_amdgcn_s_barrier();                                          // K images and tables are in LDS
    // K^T fragments of this wave's share of dQ^T = K^T dS^T: query tile (wave & 1) of a 32-query step x column tiles 2 (wave >> 1), + 1
    const int qsel = wave & 1, cpair = wave >> 1;
    bf16x8 kT[2][8];
    {
        s16x4 tk[2][4][2][2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int m = 0; m < 4; ++m) ds_tr_block(tk[ct][m], lds_u32(smem) + OFF_K + m * 8192 + tr_lane_off64((cpair * 2 + ct) * 16, lane));
        tr_wait8(tk[0], tk[1]);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                kT[ct][2 * m] = tr_join(tk[ct][m][0][0], tk[ct][m][0][1]);
                kT[ct][2 * m + 1] = tr_join(tk[ct][m][1][0], tk[ct][m][1][1]);
            }
    }
    unsigned qoff[4], ooff[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { qoff[dt] = lds_u32(smem) + tr_lane_off64(dt * 16, lane); ooff[dt] = qoff[dt] + 8192; }
    // dS image addressing: row = key within the block, 32-B rows of one 16-query tile, 8-byte slot (4 queries) XOR (row >> 2) & 3
    const unsigned ds_wr = lds_u32(smem) + OFF_DS + (unsigned)((wave * 64 + lr) * 32 + ((g ^ (lr >> 2)) << 3));     // + half * 16384 + qq * 8192 + kt * 512
    const unsigned ds_rd = lds_u32(smem) + OFF_DS + (unsigned)(qsel * 8192 + (4 * g + (lr >> 2)) * 32 + (((lr & 3) ^ g) << 3));   // + half * 16384 + s' * 1024 (+ 512)
    if (it0 + 1 < nt) { wait_vm<4>(); } else { wait_vm<0>(); }
    __builtin_amdgcn_s_barrier();                                          // tile it0 has landed for every wave
    if (it0 + 2 < nt) stage(it0 + 2, 2);

    // ---- first half of a 32-query step: S^T, dP^T, p, dS -> LDS, dV^T += P^T dO, dK^T += dS^T Q
    auto pre = [&](int T, int slot, auto halftag) {
        constexpr int half = decltype(halftag)::value;
        const char* ldsQ = smem + slot * STB1;
        const char* ldsO = ldsQ + 8192;
        const int q0 = T * 64;
        const bool diag = p.causal && (k0 + wave * 64 + 63 > q0 + half * 32);          // wave-uniform: some key of this wave lies behind some query of the step
        bf16x4 p4[4][2], d4[4][2];
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
            const int qt = half * 2 + qq;
            const f32x4 nl = *reinterpret_cast<const f32x4*>(ldsNL + q0 + qt * 16 + g * 4);
            const f32x4 nd = *reinterpret_cast<const f32x4*>(ldsND + q0 + qt * 16 + g * 4);
            f32x4 sv[4], dp[4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) { sv[kt] = anymask ? nl + kb[kt] : nl; dp[kt] = nd; }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 qa = frag_row(ldsQ, qt * 16 + lr, ks, g), oa = frag_row(ldsO, qt * 16 + lr, ks, g);
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    sv[kt] = MFMA16(qa, kf[kt][ks], sv[kt]);
                    dp[kt] = MFMA16(oa, vf[kt][ks], dp[kt]);
                }
            }
            if (diag) {
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pr = mykey[kt] <= q0 + qt * 16 + g * 4 + r ? __builtin_amdgcn_exp2f(sv[kt][r]) : 0.f;
                        sv[kt][r] = pr; dp[kt][r] *= pr;
                    }
            } else {
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pr = __builtin_amdgcn_exp2f(sv[kt][r]);
                        sv[kt][r] = pr; dp[kt][r] *= pr;
                    }
            }
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                p4[kt][qq] = to_bf4(sv[kt]);
                d4[kt][qq] = to_bf4(dp[kt]);
                *reinterpret_cast<__attribute__((address_space(3))) bf16x4*>(ds_wr + half * 16384 + qq * 8192 + kt * 512) = d4[kt][qq];
            }
        }
        bf16x8 pf[4], df[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) { pf[kt] = join4(p4[kt][0], p4[kt][1]); df[kt] = join4(d4[kt][0], d4[kt][1]); }
        s16x4 to[4][2], tq[4][2];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const unsigned ao = ooff[dt] + (unsigned)(slot * STB1), aq = qoff[dt] + (unsigned)(slot * STB1);
            if (half == 0) { ds_tr<0>(to[dt][0], ao); ds_tr<2048>(to[dt][1], ao); ds_tr<0>(tq[dt][0], aq); ds_tr<2048>(tq[dt][1], aq); }
            else { ds_tr<4096>(to[dt][0], ao); ds_tr<6144>(to[dt][1], ao); ds_tr<4096>(tq[dt][0], aq); ds_tr<6144>(tq[dt][1], aq); }
        }
        // one statement: the transposed reads have landed, and the packed operands (VALU results) are two wait states old for the asm MFMAs
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 1" : "+v"(to[0][0]), "+v"(to[0][1]), "+v"(to[1][0]), "+v"(to[1][1]), "+v"(to[2][0]), "+v"(to[2][1]), "+v"(to[3][0]), "+v"(to[3][1]),
                                                  "+v"(tq[0][0]), "+v"(tq[0][1]), "+v"(tq[1][0]), "+v"(tq[1][1]), "+v"(tq[2][0]), "+v"(tq[2][1]), "+v"(tq[3][0]), "+v"(tq[3][1]),
                                                  "+v"(pf[0]), "+v"(pf[1]), "+v"(pf[2]), "+v"(pf[3]), "+v"(df[0]), "+v"(df[1]), "+v"(df[2]), "+v"(df[3]));
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const bf16x8 ot = tr_join(to[dt][0], to[dt][1]), qtf = tr_join(tq[dt][0], tq[dt][1]);
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                mfma_agpr(dv[kt][dt], pf[kt], ot);
                mfma_agpr(dk[kt][dt], df[kt], qtf);
            }
        }
    };
    // ---- second half, behind the barrier: this wave's two tiles of dQ^T = K^T dS^T over the block's 256 keys -> its slab rows
    auto post = [&](int T, auto halftag) {
        constexpr int half = decltype(halftag)::value;
        f32x4 dq0 = {0.f, 0.f, 0.f, 0.f}, dq1 = {0.f, 0.f, 0.f, 0.f};
        const unsigned base = ds_rd + half * 16384;
        s16x4 sb[8][2];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (s == 0) { ds_tr<0>(sb[0][0], base); ds_tr<512>(sb[0][1], base); }
            if (s == 1) { ds_tr<1024>(sb[1][0], base); ds_tr<1536>(sb[1][1], base); }
            if (s == 2) { ds_tr<2048>(sb[2][0], base); ds_tr<2560>(sb[2][1], base); }
            if (s == 3) { ds_tr<3072>(sb[3][0], base); ds_tr<3584>(sb[3][1], base); }
            if (s == 4) { ds_tr<4096>(sb[4][0], base); ds_tr<4608>(sb[4][1], base); }
            if (s == 5) { ds_tr<5120>(sb[5][0], base); ds_tr<5632>(sb[5][1], base); }
            if (s == 6) { ds_tr<6144>(sb[6][0], base); ds_tr<6656>(sb[6][1], base); }
            if (s == 7) { ds_tr<7168>(sb[7][0], base); ds_tr<7680>(sb[7][1], base); }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sb[0][0]), "+v"(sb[0][1]), "+v"(sb[1][0]), "+v"(sb[1][1]), "+v"(sb[2][0]), "+v"(sb[2][1]), "+v"(sb[3][0]), "+v"(sb[3][1]),
                                              "+v"(sb[4][0]), "+v"(sb[4][1]), "+v"(sb[5][0]), "+v"(sb[5][1]), "+v"(sb[6][0]), "+v"(sb[6][1]), "+v"(sb[7][0]), "+v"(sb[7][1]));
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const bf16x8 bs = tr_join(sb[s][0], sb[s][1]);
            dq0 = MFMA16(kT[0][s], bs, dq0);
            dq1 = MFMA16(kT[1][s], bs, dq1);
        }
        const int q = T * 64 + half * 32 + qsel * 16 + lr;
        if (q < p.Sq) {
            bf16_t* row = slab + (long)q * d_model + cpair * 32 + g * 4;
            *reinterpret_cast<bf16x4*>(row) = to_bf4(dq0 * p.scale);
            *reinterpret_cast<bf16x4*>(row + 16) = to_bf4(dq1 * p.scale);
        }
    };
    auto sync = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // this wave's dS stores
        __builtin_amdgcn_s_barrier();
    };

    int slot = 0;
    pre(it0, slot, IntTag<0>{});
    sync();
    for (int T = it0;; ++T) {
        post(T, IntTag<0>{});
        pre(T, slot, IntTag<1>{});
        if (T + 1 < nt) { if (T + 2 < nt) { wait_vm<4>(); } else { wait_vm<0>(); } }      // tile T + 1 has landed (the 4 youngest pieces are tile T + 2's)
        sync();
        post(T, IntTag<1>{});
        if (T + 1 >= nt) break;
        if (T + 3 < nt) stage(T + 3, slot);                                   // tile T's slot: every wave has passed its last read of it
        slot = slot == RING1 - 1 ? 0 : slot + 1;
        pre(T + 1, slot, IntTag<0>{});
        sync();
    }

    // ---- epilogue: dK (x scale), dV rows of this block; masked keys receive zeros; column sums = k / v bias-gradient partials
    asm volatile("s_nop 15\n\ts_nop 7" : "+a"(dk[0][0]), "+a"(dk[0][1]), "+a"(dk[0][2]), "+a"(dk[0][3]), "+a"(dk[1][0]), "+a"(dk[1][1]), "+a"(dk[1][2]), "+a"(dk[1][3]),
                                         "+a"(dk[2][0]), "+a"(dk[2][1]), "+a"(dk[2][2]), "+a"(dk[2][3]), "+a"(dk[3][0]), "+a"(dk[3][1]), "+a"(dk[3][2]), "+a"(dk[3][3]));
    asm volatile("s_nop 0" : "+a"(dv[0][0]), "+a"(dv[0][1]), "+a"(dv[0][2]), "+a"(dv[0][3]), "+a"(dv[1][0]), "+a"(dv[1][1]), "+a"(dv[1][2]), "+a"(dv[1][3]),
                             "+a"(dv[2][0]), "+a"(dv[2][1]), "+a"(dv[2][2]), "+a"(dv[2][3]), "+a"(dv[3][0]), "+a"(dv[3][1]), "+a"(dv[3][2]), "+a"(dv[3][3]));
    float csk[4], csv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { csk[dt] = 0.f; csv[dt] = 0.f; }
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = k0 + wave * 64 + kt * 16 + g * 4 + r;
            if (key < p.Sk) {
                const bool kvis = key < kvis_end && (!p.key_mask || p.key_mask[(long)b * p.Sk + key] != 0.f);
                bf16_t* DK = p.dk + b * p.dk_sb + (long)key * p.dk_ss + h * HDT;
                bf16_t* DV = p.dv + b * p.dv_sb + (long)key * p.dv_ss + h * HDT;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const float vk = kvis ? dk[kt][dt][r] * p.scale : 0.f, vv = kvis ? dv[kt][dt][r] : 0.f;
                    DK[dt * 16 + lr] = (bf16_t)vk;
                    DV[dt * 16 + lr] = (bf16_t)vv;
                    csk[dt] += vk; csv[dt] += vv;
                }
            }
        }
    if (p.cs_kv) {
        float* red = reinterpret_cast<float*>(smem);                      // [4 waves][2 HDT]: the ring is free (every wave is past the last barrier's reads)
        __syncthreads();
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const float sk = grp_sum(csk[dt]), sv_ = grp_sum(csv[dt]);
            if (g == 0) { red[wave * 2 * HDT + dt * 16 + lr] = sk; red[wave * 2 * HDT + HDT + dt * 16 + lr] = sv_; }
        }
        __syncthreads();
        if (t < 2 * HDT) {
            float* row = p.cs_kv + (long)(b * nkb0 + rb) * 2 * d_model + h * HDT;
            const float v = red[t] + red[2 * HDT + t] + red[4 * HDT + t] + red[6 * HDT + t];
            row[t < HDT ? t : d_model + t - HDT] = v;
        }
    }
}

// delta[b][h][s] = sum_c dO[row][h 64 + c] O[row][h 64 + c] (f32): one 8-lane group per (row, head), 16 bytes per lane and tensor
__global__ __launch_bounds__(256) void fa1_delta_kernel(const Fa64Args pin, int rows_per_block) {
    const int b = blockIdx.y;
    Fa64Args p = pin;
    varlen_localize(p, b);
    const int lse_ld = pin.Sq, H = p.H, G = H * 8;                          // G lanes cover one row
    const int s0 = blockIdx.x * rows_per_block, s1 = min(p.Sq, s0 + rows_per_block);
    for (int i = threadIdx.x; i < (s1 - s0) * G; i += 256) {
        const int s = s0 + i / G, j = i % G;                                // j = head * 8 + chunk
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(p.dout + b * p.o_sb + (long)s * p.o_ss + j * 8);
        const bf16x8 o = *reinterpret_cast<const bf16x8*>(p.o + b * p.o_sb + (long)s * p.o_ss + j * 8);
        float acc = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc = fmaf((float)a[e], (float)o[e], acc);
        acc += PB_DPP_F(acc, 0xb1);      // lanes ^1
        acc += PB_DPP_F(acc, 0x4e);      // lanes ^2
        acc += PB_DPP_F(acc, 0x141);     // row_half_mirror: lanes 7 - l of each 8
        if ((j & 7) == 0) const_cast<float*>(p.delta)[((long)b * H + (j >> 3)) * lse_ld + s] = acc;
    }
}

// dq row = sum of the row's valid slabs (key blocks 0 .. n - 1 in order, f32), rounded once; per (batch, 128-row chunk) column sums
// -> partial row of the q-bias gradient. Grid (chunks of 128 rows, B); threads = H * 8 column groups x rows in flight.
__global__ __launch_bounds__(256) void fa1_reduce_kernel(const Fa1Args pin, int nrow_lanes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.y, chunk = blockIdx.x;
    Fa64Args p = pin.a;
    varlen_localize(p, b);
    const int d_model = p.H * 64, G = p.H * 8;
    const int nqb = (pin.a.Sq + 127) / 128;
    const int kvis_end = p.kmax ? min(p.Sk, p.kmax[b]) : p.Sk;
    const int nvalid = (kvis_end + KB1 - 1) / KB1;                           // key blocks that wrote their slab
    const bf16_t* slab = pin.slab + (pin.a.vl_q_off ? (long)pin.a.vl_q_off[b] * d_model : (long)b * pin.slab_sb);
    const int cg = threadIdx.x % G, rl = threadIdx.x / G;
    float cs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) cs[e] = 0.f;
    const int s1 = min(p.Sq, chunk * 128 + 128);
    if (rl < nrow_lanes)
        for (int s = chunk * 128 + rl; s < s1; s += nrow_lanes) {
            const int nb = p.causal ? min(nvalid, s / KB1 + 1) : nvalid;
            float acc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = 0.f;
            for (int j = 0; j < nb; ++j) {
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(slab + (long)j * pin.slab_stride + (long)s * d_model + cg * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
            }
            bf16x8 r;
#pragma unroll
            for (int e = 0; e < 8; ++e) { r[e] = (bf16_t)acc[e]; cs[e] += (float)r[e]; }
            *reinterpret_cast<bf16x8*>(p.dq + b * p.dq_sb + (long)s * p.dq_ss + cg * 8) = r;
        }
    if (p.cs_q) {
        float* red = reinterpret_cast<float*>(smem);                      // [nrow_lanes][d_model]
        if (rl < nrow_lanes)
#pragma unroll
            for (int e = 0; e < 8; ++e) red[rl * d_model + cg * 8 + e] = cs[e];
        __syncthreads();
        for (int col = threadIdx.x; col < d_model; col += blockDim.x) {
            float v = 0.f;
            for (int r = 0; r < nrow_lanes; ++r) v += red[r * d_model + col];
            p.cs_q[(long)(b * nqb + chunk) * d_model + col] = v;
        }
    }
}

}  // namespace

static const bf16_t* fa1_zero_page() {
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

// bytes of dQ slab workspace for `rows` query rows in all (packed: the row count of the q side; dense: B * Sq)
extern "C" int64_t pb_flash_bwd1_ws_bytes(int64_t rows, int32_t H, int32_t hd, int32_t Sk_max) {
    if (hd != 64) return 0;
    return (int64_t)((Sk_max + KB1 - 1) / KB1) * rows * H * hd * 2;
}

// One-pass backward, head_dim 64. vl = {q_off, q_len, k_off, k_len} (packed rows) or NULL (dense: batch strides). Same contract as
// pb_flash64_bwd plus the slab workspace `ws` (pb_flash_bwd1_ws_bytes) and q_rows = rows of the q side (packed) / B * Sq (dense).
int pb_flash1_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse, float* delta, const float* key_mask,
                  const int* kmax, void* dq, void* dk, void* dv, int B, int H, int Sq, int Sk, long q_sb, long q_ss, long k_sb, long k_ss, long v_sb,
                  long v_ss, long o_sb, long o_ss, long dq_sb, long dq_ss, long dk_sb, long dk_ss, long dv_sb, long dv_ss, float scale,
                  int causal, float* dbias_q, float* dbias_k, float* dbias_v, float* dbias_ws, void* ws, long q_rows, hipStream_t stream, const int* const* vl) {
    Fa1Args A = {};
    Fa64Args& a = A.a;
    if (vl) { a.vl_q_off = vl[0]; a.vl_q_len = vl[1]; a.vl_k_off = vl[2]; a.vl_k_len = vl[3]; }
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.o = (const bf16_t*)o; a.dout = (const bf16_t*)dout;
    a.dq = (bf16_t*)dq; a.dk = (bf16_t*)dk; a.dv = (bf16_t*)dv; a.lse = const_cast<float*>(lse); a.delta = delta; a.key_mask = key_mask; a.kmax = kmax;
    a.B = B; a.H = H; a.Sq = Sq; a.Sk = Sk; a.q_sb = q_sb; a.q_ss = q_ss; a.k_sb = k_sb; a.k_ss = k_ss; a.v_sb = v_sb; a.v_ss = v_ss;
    a.o_sb = o_sb; a.o_ss = o_ss; a.dq_sb = dq_sb; a.dq_ss = dq_ss; a.dk_sb = dk_sb; a.dk_ss = dk_ss; a.dv_sb = dv_sb; a.dv_ss = dv_ss;
    a.scale = scale; a.causal = causal;
    a.zeros = fa1_zero_page();
    PB_REQUIRE(a.zeros != nullptr, "pb_flash_bwd1: cannot allocate the zero page");
    PB_REQUIRE(ws != nullptr, "pb_flash_bwd1: the dQ slab workspace is required (pb_flash_bwd1_ws_bytes)");
    const int d_model = H * 64, nkb = (Sk + KB1 - 1) / KB1, nqb = (Sq + 127) / 128;
    A.slab = (bf16_t*)ws; A.slab_stride = q_rows * d_model; A.slab_sb = (long)Sq * d_model;
    if (dbias_q) {
        PB_REQUIRE(dbias_k && dbias_v && dbias_ws, "pb_flash_bwd1: dbias_q/k/v and dbias_ws go together");
        const size_t n_kv = (size_t)B * nkb * 2 * d_model;
        if (float* slice = pb_defer_alloc(n_kv + (size_t)B * nqb * d_model)) dbias_ws = slice;
        a.cs_kv = dbias_ws; a.cs_q = dbias_ws + n_kv;
    }
    const size_t lds = (size_t)OFF_TAB + (size_t)((Sq + 63) / 64) * 64 * 8;
    PB_REQUIRE(lds <= 160 * 1024, "pb_flash_bwd1: Sq=%d needs %zu bytes of LDS", Sq, lds);
    static bool attr_set = false;
    if (!attr_set) {
        PB_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fa1_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    const int rpb = 64;
    hipLaunchKernelGGL(fa1_delta_kernel, dim3((Sq + rpb - 1) / rpb, B), dim3(256), 0, stream, a, rpb);
    PB_LAUNCH_CHECK();
    hipLaunchKernelGGL(fa1_bwd_kernel, dim3(nkb * H * B), dim3(FT), lds, stream, A);
    PB_LAUNCH_CHECK();
    const int G = H * 8;
    PB_REQUIRE(G <= 256, "pb_flash_bwd1: more than 32 heads");
    const int nrl = 256 / G;
    hipLaunchKernelGGL(fa1_reduce_kernel, dim3(nqb, B), dim3(256), dbias_q ? (size_t)nrl * d_model * 4 : 0, stream, A, nrl);
    PB_LAUNCH_CHECK();
    if (!dbias_q) return 0;
    if (pb_finalize_rows(a.cs_kv, B * nkb, d_model, dbias_k, stream, 2, dbias_v)) return -1;
    return pb_finalize_rows(a.cs_q, B * nqb, d_model, dbias_q, stream);
}

extern "C" int pb_flash_bwd1(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                             const float* key_mask, const int32_t* kmax, void* dq, void* dk, void* dv, float* delta, int32_t B, int32_t H, int32_t Sq,
                             int32_t Sk, int32_t hd, int64_t q_sb, int64_t q_ss, int64_t k_sb, int64_t k_ss, int64_t v_sb,
                             int64_t v_ss, int64_t o_sb, int64_t o_ss, int64_t dq_sb, int64_t dq_ss, int64_t dk_sb, int64_t dk_ss,
                             int64_t dv_sb, int64_t dv_ss, float scale, int32_t causal, float* dbias_q, float* dbias_k, float* dbias_v, float* dbias_ws,
                             void* dq_ws, void* stream_) {
    PB_REQUIRE(hd == 64, "pb_flash_bwd1: head_dim %d (64 only)", hd);
    PB_REQUIRE(q_ss % 8 == 0 && k_ss % 8 == 0 && v_ss % 8 == 0 && o_ss % 8 == 0 && dq_ss % 8 == 0 && dk_ss % 8 == 0 && dv_ss % 8 == 0 &&
               q_sb % 8 == 0 && k_sb % 8 == 0 && v_sb % 8 == 0 && o_sb % 8 == 0 && dq_sb % 8 == 0, "pb_flash_bwd1: strides must be multiples of 8 elements");
    if (B <= 0 || H <= 0 || Sq <= 0 || Sk <= 0) return 0;
    return pb_flash1_bwd(q, k, v, o, dout, lse, delta, key_mask, kmax, dq, dk, dv, B, H, Sq, Sk, q_sb, q_ss, k_sb, k_ss, v_sb, v_ss, o_sb, o_ss,
                         dq_sb, dq_ss, dk_sb, dk_ss, dv_sb, dv_ss, scale, causal & 1, dbias_q, dbias_k, dbias_v, dbias_ws, dq_ws, (long)B * Sq,
                         (hipStream_t)stream_, nullptr);
}

extern "C" int pb_flash_bwd1_packed(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse, void* dq,
                                    void* dk, void* dv, float* delta, const int32_t* q_off, const int32_t* q_len, const int32_t* k_off,
                                    const int32_t* k_len, const int32_t* k_vis, int32_t B, int32_t H, int32_t Sq_max, int32_t Sk_max, int32_t hd,
                                    int64_t q_ss, int64_t k_ss, int64_t v_ss, int64_t o_ss, int64_t dq_ss, int64_t dk_ss, int64_t dv_ss,
                                    float scale, int32_t causal, float* dbias_q, float* dbias_k, float* dbias_v, float* dbias_ws,
                                    void* dq_ws, int64_t q_rows, void* stream_) {
    PB_REQUIRE(hd == 64, "pb_flash_bwd1_packed: head_dim %d (64 only)", hd);
    PB_REQUIRE(q_ss % 8 == 0 && k_ss % 8 == 0 && v_ss % 8 == 0 && o_ss % 8 == 0 && dq_ss % 8 == 0 && dk_ss % 8 == 0 && dv_ss % 8 == 0,
               "pb_flash_bwd1_packed: strides must be multiples of 8 elements");
    PB_REQUIRE(q_off && q_len && k_off && k_len && k_vis, "pb_flash_bwd1_packed: the five row descriptors are required");
    if (B <= 0 || H <= 0 || Sq_max <= 0 || Sk_max <= 0) return 0;
    const int* vl[4] = {q_off, q_len, k_off, k_len};
    return pb_flash1_bwd(q, k, v, o, dout, lse, delta, nullptr, k_vis, dq, dk, dv, B, H, Sq_max, Sk_max, 0, q_ss, 0, k_ss, 0, v_ss, 0, o_ss,
                         0, dq_ss, 0, dk_ss, 0, dv_ss, scale, causal & 1, dbias_q, dbias_k, dbias_v, dbias_ws, dq_ws, q_rows, (hipStream_t)stream_, vl);
}
