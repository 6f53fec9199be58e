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

// ---- the accumulator half of the register file is OURS (one wave per SIMD: 256 VGPRs + 256 AGPRs), addressed literally:
//   a[0:63]    dK^T tiles [key tile 4][column tile 4]        a[64:127]   dV^T tiles
//   a[128:159] K fragments (prescaled) [key tile 4][k-step 2] a[160:191]  V fragments
//   a[192:255] K^T fragments of this wave's two dQ^T column tiles [2][k-step 8 over the block's 256 keys]
// Every MFMA is its own asm statement, in the order written (volatile); the compiler places the vector, LDS and scalar work
// around them and owns the VGPRs. What it is NOT told: the MFMAs' latency. The rules kept by construction (cdna_hip_programming.md
// 5.7): a VGPR result is read by vector code only behind later MFMA statements or an s_nop fence; the compiler must not use any
// AGPR itself (audit: 0 spills, no v_accvgpr in compiler code: tools/check_fa1_regs.py).
#define PB_U10(p) p "0", p "1", p "2", p "3", p "4", p "5", p "6", p "7", p "8", p "9"
#define PB_ALL_AGPRS PB_U10("a"), PB_U10("a1"), PB_U10("a2"), PB_U10("a3"), PB_U10("a4"), PB_U10("a5"), PB_U10("a6"), PB_U10("a7"), PB_U10("a8"), PB_U10("a9"), \
    PB_U10("a10"), PB_U10("a11"), PB_U10("a12"), PB_U10("a13"), PB_U10("a14"), PB_U10("a15"), PB_U10("a16"), PB_U10("a17"), PB_U10("a18"), PB_U10("a19"), \
    PB_U10("a20"), PB_U10("a21"), PB_U10("a22"), PB_U10("a23"), PB_U10("a24"), "a250", "a251", "a252", "a253", "a254", "a255"
constexpr int A_DK = 0, A_DV = 64, A_KF = 128, A_VF = 160, A_KT = 192;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int V> struct IntTag { static constexpr int value = V; };
template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(IntTag<I>{}); static_for<I + 1, N>(f); }
}
template <int R> __device__ __forceinline__ void agpr_zero() { asm volatile("v_accvgpr_write_b32 a%c0, 0" :: "i"(R)); }
template <int R> __device__ __forceinline__ void agpr_put(const bf16x8& v) {                 // a[R:R+3] = v
    const u32x4 u = __builtin_bit_cast(u32x4, v);
    asm volatile("v_accvgpr_write_b32 a%c4, %0\n\tv_accvgpr_write_b32 a%c5, %1\n\tv_accvgpr_write_b32 a%c6, %2\n\tv_accvgpr_write_b32 a%c7, %3"
                 :: "v"(u[0]), "v"(u[1]), "v"(u[2]), "v"(u[3]), "i"(R), "i"(R + 1), "i"(R + 2), "i"(R + 3));
}
template <int R> __device__ __forceinline__ f32x4 agpr_get() {
    f32x4 r;
    asm volatile("v_accvgpr_read_b32 %0, a%c4\n\tv_accvgpr_read_b32 %1, a%c5\n\tv_accvgpr_read_b32 %2, a%c6\n\tv_accvgpr_read_b32 %3, a%c7"
                 : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]) : "i"(R), "i"(R + 1), "i"(R + 2), "i"(R + 3));
    return r;
}
// d = A(VGPR) x B(a[RB:RB+3]) + c    (first k-step of an S / dP chain: the row constants ride in as C)
template <int RB> __device__ __forceinline__ void mfma_vab_c(f32x4& d, const bf16x8& a, const f32x4& c) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, a[%c3:%c4], %2" : "=&v"(d) : "v"(a), "v"(c), "i"(RB), "i"(RB + 3) : PB_ALL_AGPRS);
}
template <int RB> __device__ __forceinline__ void mfma_vab(f32x4& d, const bf16x8& a) {       // d += A(VGPR) x B(AGPR)
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, a[%c2:%c3], %0" : "+v"(d) : "v"(a), "i"(RB), "i"(RB + 3) : PB_ALL_AGPRS);
}
template <int RA> __device__ __forceinline__ void mfma_aav(f32x4& d, const bf16x8& b) {       // d += A(AGPR) x B(VGPR)
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, a[%c2:%c3], %1, %0" : "+v"(d) : "v"(b), "i"(RA), "i"(RA + 3) : PB_ALL_AGPRS);
}
template <int RA> __device__ __forceinline__ void mfma_aav_z(f32x4& d, const bf16x8& b) {     // d = A(AGPR) x B(VGPR): C is the constant 0 (a vector
    // instruction that zeroes d right in front of the MFMA would need two wait states that nobody inserts for an asm statement)
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, a[%c2:%c3], %1, 0" : "=&v"(d) : "v"(b), "i"(RA), "i"(RA + 3) : PB_ALL_AGPRS);
}
template <int RD> __device__ __forceinline__ void mfma_acc(const bf16x8& a, const bf16x8& b) { // a[RD:RD+3] += A(VGPR) x B(VGPR)
    asm volatile("v_mfma_f32_16x16x32_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" :: "v"(a), "v"(b), "i"(RD), "i"(RD + 3) : PB_ALL_AGPRS);
}
__device__ __forceinline__ bf16x4 to_bf4(const f32x4& v) {
    bf16x4 r = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    return r;
}
__device__ __forceinline__ bf16x8 join4(const bf16x4& lo, const bf16x4& hi) { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); }
#define PB_PIN() __builtin_amdgcn_sched_barrier(0)

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
    asm volatile("" ::: PB_ALL_AGPRS);                                     // the kernel descriptor allocates all 256 accumulator registers
    const bf16_t* Q = p.q + b * p.q_sb + h * HDT;
    const bf16_t* DO = p.dout + b * p.o_sb + h * HDT;
    bf16_t* slab = pin.slab + (long)rb * pin.slab_stride + (pin.a.vl_q_off ? (long)pin.a.vl_q_off[b] * d_model : (long)b * pin.slab_sb) + h * HDT;
    const float c = p.scale * LOG2E;
    float* ldsNL = reinterpret_cast<float*>(smem + OFF_TAB);
    float* ldsND = ldsNL + nt * 64;
    float* ldsVis = reinterpret_cast<float*>(smem + OFF_DS);               // 1 / 0 per key of the block (the dS buffers are not in use yet)
    for (int q = it0 * 64 + t; q < nt * 64; q += FT) {
        const long li = ((long)b * p.H + h) * lse_ld + q;
        const float ls = q < p.Sq ? p.lse[li] : INFINITY;
        ldsNL[q] = ls == INFINITY ? -INFINITY : -ls * LOG2E;
        ldsND[q] = q < p.Sq ? -p.delta[li] : 0.f;
    }
    {
        const int key = k0 + t;                                            // FT = KB1 = 256: one key per thread
        ldsVis[t] = (key < kvis_end && (!p.key_mask || p.key_mask[(long)b * p.Sk + key] != 0.f)) ? 1.f : 0.f;
    }
    // this wave's 64 keys: K (prescaled: S comes out of the MFMA in log2 units) and V fragments go to AGPRs for the whole sweep. A MASKED
    // key's K fragments are zeros here and in the K^T fragments below: its scores are then -lse (p finite), its dS meets a zero K row in
    // dQ, and its own dK / dV rows are zeroed in the epilogue -- the sweep itself never looks at a key mask.
    static_for<0, 4>([&](auto ktt) {
        constexpr int kt = decltype(ktt)::value;
        const int key = k0 + wave * 64 + kt * 16 + lr;
        const bool vis = key < kvis_end && (!p.key_mask || p.key_mask[(long)b * p.Sk + key] != 0.f);
        static_for<0, 2>([&](auto kss) {
            constexpr int ks = decltype(kss)::value;
            bf16x8 kf = scale_frag(frag_global(K, p.k_ss, key, p.Sk, ks * 32 + g * 8), c);
            const bf16x8 vf = frag_global(V, p.v_ss, key, p.Sk, ks * 32 + g * 8);
            const bf16x8 z = {};
            if (!vis) kf = z;
            agpr_put<A_KF + 4 * (kt * 2 + ks)>(kf);
            agpr_put<A_VF + 4 * (kt * 2 + ks)>(vf);
        });
    });
    static_for<0, 128>([&](auto r) { agpr_zero<decltype(r)::value>(); });
    // ---- DMA: the 4 K images of this block (for the K^T fragments), then the first three {Q, dO} tiles
    const StageOff so_k = stage_off(p.k_ss, wave, lane), so_q = stage_off(p.q_ss, wave, lane), so_o = stage_off(p.o_ss, wave, lane);
#pragma unroll
    for (int m = 0; m < 4; ++m) stage64(K, p.k_ss, k0 + 64 * m, p.Sk, smem + OFF_K + m * 8192, wave, lane, so_k);
    auto stage = [&](int it, int slot) {
        char* st = smem + slot * STB1;
        stage64(Q, p.q_ss, it * 64, p.Sq, st, wave, lane, so_q);
        stage64(DO, p.o_ss, it * 64, p.Sq, st + 8192, wave, lane, so_o);
    };
    const int npre = min(3, nt - it0);                                     // tiles requested up front (the ring is empty)
    stage(it0, 0);
    if (npre > 1) stage(it0 + 1, 1);
    if (npre > 2) stage(it0 + 2, 2);
    if (npre == 3) { wait_vm<12>(); } else if (npre == 2) { wait_vm<8>(); } else { wait_vm<4>(); }
    __builtin_amdgcn_s_waitcnt(0xc07f);                                    // this wave's table stores
    __builtin_amdgcn_s_barrier();                                          // K images and tables are in LDS
    // K^T fragments of this wave's share of dQ^T = K^T dS^T: query tile (wave & 1) of a 32-query step x column tiles 2 (wave >> 1), + 1
    const int qsel = wave & 1, cpair = wave >> 1;
    {
        s16x4 tk[2][4][2][2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int m = 0; m < 4; ++m) ds_tr_block(tk[ct][m], lds_u32(smem) + OFF_K + m * 8192 + tr_lane_off64((cpair * 2 + ct) * 16, lane));
        tr_wait8(tk[0], tk[1]);
        static_for<0, 8>([&](auto ss) {
            constexpr int s = decltype(ss)::value;                         // k-step s = keys 32 s .. 32 s + 31 of the block, element j <-> key 32 s + 16 (j >> 2) + 4 g + (j & 3)
            const f32x4 m0 = *reinterpret_cast<const f32x4*>(ldsVis + 32 * s + 4 * g), m1 = *reinterpret_cast<const f32x4*>(ldsVis + 32 * s + 16 + 4 * g);
            static_for<0, 2>([&](auto ctt) {
                constexpr int ct = decltype(ctt)::value;
                bf16x8 f = tr_join(tk[ct][s >> 1][s & 1][0], tk[ct][s >> 1][s & 1][1]);
#pragma unroll
                for (int j = 0; j < 4; ++j) { if (m0[j] == 0.f) f[j] = (bf16_t)0.f; if (m1[j] == 0.f) f[4 + j] = (bf16_t)0.f; }
                agpr_put<A_KT + 4 * (ct * 8 + s)>(f);
            });
        });
    }
    unsigned qoff[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) qoff[dt] = lds_u32(smem) + tr_lane_off64(dt * 16, lane);
    // dS image addressing: row = key within the block, 32-B rows of one 16-query tile, 8-byte slot (4 queries) XOR (row >> 2) & 3
    const unsigned ds_wr = lds_u32(smem) + OFF_DS + (unsigned)((wave * 64 + lr) * 32 + ((g ^ (lr >> 2)) << 3));     // + half * 16384 + qq * 8192 + kt * 512
    const unsigned ds_rd = lds_u32(smem) + OFF_DS + (unsigned)(qsel * 8192 + (4 * g + (lr >> 2)) * 32 + (((lr & 3) ^ g) << 3));   // + half * 16384 + s' * 1024 (+ 512)
    if (npre == 3) { wait_vm<8>(); } else if (npre == 2) { wait_vm<4>(); } else { wait_vm<0>(); }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();                                          // tile it0 has landed for every wave; ldsVis has been read (the dS buffers may be written)

    // state carried from one 32-query step to the next: the packed P^T / dS^T operands and the transposed dO / Q fragments of the
    // previous step (its dV / dK products run beside this step's softmax), the pending dQ^T tiles
    bf16x8 pfp[4], dfp[4];
    s16x4 to[2][2], tq[2][2];
    const int mykey0 = k0 + wave * 64 + lr;

    // One step = 32 queries (half HALF of the 64-query tile T in ring slot `slot`). MFMA stream: dV/dK(previous step) column tiles 0, 1 |
    // S^T, dP^T of query tile A | S^T, dP^T of query tile B beside exp / dS of A | dV/dK(previous) column tiles 2, 3 and dQ^T(previous)
    // beside exp / dS of B. One barrier per step: behind it dS(this step) is visible and the previous buffer is free.
    auto step = [&](int T, int slot, int pslot, auto halftag, auto firsttag, auto diagtag) {
        constexpr int HALF = decltype(halftag)::value;
        constexpr bool FIRST = decltype(firsttag)::value != 0, DIAG = decltype(diagtag)::value != 0;
        constexpr int PH = 1 - HALF;                                       // half (and dS buffer) of the previous step
        const char* ldsQ = smem + slot * STB1;
        const char* ldsO = ldsQ + 8192;
        const int q0 = T * 64;
        // ---- this step's first query tile
        constexpr int qtA = HALF * 2, qtB = HALF * 2 + 1;
        const f32x4 nlA = *reinterpret_cast<const f32x4*>(ldsNL + q0 + qtA * 16 + g * 4), ndA = *reinterpret_cast<const f32x4*>(ldsND + q0 + qtA * 16 + g * 4);
        const bf16x8 qaA0 = frag_row(ldsQ, qtA * 16 + lr, 0, g), oaA0 = frag_row(ldsO, qtA * 16 + lr, 0, g);
        const bf16x8 qaA1 = frag_row(ldsQ, qtA * 16 + lr, 1, g), oaA1 = frag_row(ldsO, qtA * 16 + lr, 1, g);
        PB_PIN();
        // ---- dV^T, dK^T of the previous step, column tiles 0, 1 (operands in registers since before the barrier)
        if constexpr (!FIRST) {
            static_for<0, 2>([&](auto dtt) {
                constexpr int dt = decltype(dtt)::value;
                const bf16x8 ot = tr_join(to[dt][0], to[dt][1]), qt = tr_join(tq[dt][0], tq[dt][1]);
                static_for<0, 4>([&](auto ktt) {
                    constexpr int kt = decltype(ktt)::value;
                    mfma_acc<A_DV + 4 * (kt * 4 + dt)>(pfp[kt], ot);
                    mfma_acc<A_DK + 4 * (kt * 4 + dt)>(dfp[kt], qt);
                });
            });
        }
        PB_PIN();
        // ---- S^T, dP^T of query tile A
        f32x4 svA[4], dpA[4], svB[4], dpB[4];
        static_for<0, 4>([&](auto ktt) {
            constexpr int kt = decltype(ktt)::value;
            mfma_vab_c<A_KF + 4 * (kt * 2)>(svA[kt], qaA0, nlA);
            mfma_vab_c<A_VF + 4 * (kt * 2)>(dpA[kt], oaA0, ndA);
        });
        const f32x4 nlB = *reinterpret_cast<const f32x4*>(ldsNL + q0 + qtB * 16 + g * 4), ndB = *reinterpret_cast<const f32x4*>(ldsND + q0 + qtB * 16 + g * 4);
        const bf16x8 qaB0 = frag_row(ldsQ, qtB * 16 + lr, 0, g), oaB0 = frag_row(ldsO, qtB * 16 + lr, 0, g);
        const bf16x8 qaB1 = frag_row(ldsQ, qtB * 16 + lr, 1, g), oaB1 = frag_row(ldsO, qtB * 16 + lr, 1, g);
        PB_PIN();
        static_for<0, 4>([&](auto ktt) {
            constexpr int kt = decltype(ktt)::value;
            mfma_vab<A_KF + 4 * (kt * 2 + 1)>(svA[kt], qaA1);
            mfma_vab<A_VF + 4 * (kt * 2 + 1)>(dpA[kt], oaA1);
        });
        PB_PIN();
        // exp / dS of one key tile of one query tile: p = exp2(s'), ds = p dp; pack; the dS piece goes to LDS for dQ^T. The multiplies and
        // the packing are single instructions by name: left to itself hipcc pairs the multiplies into v_pk_mul_f32 (an anti-lever beside
        // MFMAs, MI355X_MICROARCH.md) and then rebuilds the bf16 pairs with v_perm / v_alignbit
        unsigned pw[4][2][2], dw[4][2][2];                                 // [key tile][query tile][dword]: bf16 pairs of p and of dS
        auto soft = [&](f32x4& sv, f32x4& dp, auto ktt, auto qqt) {
            constexpr int kt = decltype(ktt)::value, qq = decltype(qqt)::value;
            float pr[4], ds[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pr[r] = __builtin_amdgcn_exp2f(sv[r]);
                if constexpr (DIAG) pr[r] = (mykey0 + kt * 16 <= q0 + (HALF * 2 + qq) * 16 + g * 4 + r) ? pr[r] : 0.f;
            }
            // v_exp_f32 -> a vector instruction that reads its result needs a wait state (trans forwarding); hipcc pads its own code, not an
            // asm statement's operands: one fence behind the four exps covers the multiplies and the packs
            asm volatile("s_nop 0" : "+v"(pr[0]), "+v"(pr[1]), "+v"(pr[2]), "+v"(pr[3]));
#pragma unroll
            for (int r = 0; r < 4; ++r) asm("v_mul_f32 %0, %1, %2" : "=v"(ds[r]) : "v"(pr[r]), "v"(dp[r]));
            asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pw[kt][qq][0]) : "v"(pr[0]), "v"(pr[1]));
            asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pw[kt][qq][1]) : "v"(pr[2]), "v"(pr[3]));
            asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(dw[kt][qq][0]) : "v"(ds[0]), "v"(ds[1]));
            asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(dw[kt][qq][1]) : "v"(ds[2]), "v"(ds[3]));
            typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
            const u32x2 w = {dw[kt][qq][0], dw[kt][qq][1]};
            *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(ds_wr + HALF * 16384 + qq * 8192 + kt * 512) = w;
        };
        // ---- S^T, dP^T of query tile B beside the softmax of A: two MFMAs, half a key tile's vector work, ...
        static_for<0, 4>([&](auto ktt) {
            constexpr int kt = decltype(ktt)::value;
            mfma_vab_c<A_KF + 4 * (kt * 2)>(svB[kt], qaB0, nlB);
            mfma_vab_c<A_VF + 4 * (kt * 2)>(dpB[kt], oaB0, ndB);
            PB_PIN();
            soft(svA[kt], dpA[kt], ktt, IntTag<0>{});
            PB_PIN();
            mfma_vab<A_KF + 4 * (kt * 2 + 1)>(svB[kt], qaB1);
            mfma_vab<A_VF + 4 * (kt * 2 + 1)>(dpB[kt], oaB1);
            PB_PIN();
        });
        // ---- dV^T, dK^T of the previous step, column tiles 2, 3, and its dQ^T, beside the softmax of B
        f32x4 dqa, dqb;
        if constexpr (!FIRST) {
            // ... and the transposed dO / Q fragments of its column tiles 2, 3 (its tile is still in the ring: slot `pslot` when this is the
            // first half of a new tile)
            s16x4 to2[2][2], tq2[2][2];
#pragma unroll
            for (int d2 = 0; d2 < 2; ++d2) {
                const unsigned aq = qoff[2 + d2] + (unsigned)((HALF ? slot : pslot) * STB1), ao = aq + 8192;
                if constexpr (PH == 0) { ds_tr<0>(to2[d2][0], ao); ds_tr<2048>(to2[d2][1], ao); ds_tr<0>(tq2[d2][0], aq); ds_tr<2048>(tq2[d2][1], aq); }
                else { ds_tr<4096>(to2[d2][0], ao); ds_tr<6144>(to2[d2][1], ao); ds_tr<4096>(tq2[d2][0], aq); ds_tr<6144>(tq2[d2][1], aq); }
            }
            // dS^T of the previous step (visible since the barrier), requested here: its first use is a slot's length away
            s16x4 sb[8][2];
            const unsigned base = ds_rd + PH * 16384;
            ds_tr<0>(sb[0][0], base); ds_tr<512>(sb[0][1], base); ds_tr<1024>(sb[1][0], base); ds_tr<1536>(sb[1][1], base);
            ds_tr<2048>(sb[2][0], base); ds_tr<2560>(sb[2][1], base); ds_tr<3072>(sb[3][0], base); ds_tr<3584>(sb[3][1], base);
            ds_tr<4096>(sb[4][0], base); ds_tr<4608>(sb[4][1], base); ds_tr<5120>(sb[5][0], base); ds_tr<5632>(sb[5][1], base);
            ds_tr<6144>(sb[6][0], base); ds_tr<6656>(sb[6][1], base);
            // the 8 reads in front of these 14 have landed (lgkmcnt counts to 15: the last two dS reads follow the wait)
            asm volatile("s_waitcnt lgkmcnt(14)" : "+v"(to2[0][0]), "+v"(to2[0][1]), "+v"(to2[1][0]), "+v"(to2[1][1]), "+v"(tq2[0][0]), "+v"(tq2[0][1]), "+v"(tq2[1][0]), "+v"(tq2[1][1]));
            ds_tr<7168>(sb[7][0], base); ds_tr<7680>(sb[7][1], base);
            static_for<0, 4>([&](auto ktt) {
                constexpr int kt = decltype(ktt)::value;                   // here: a slot index; the MFMAs of the slot are fixed below
                constexpr int dt = 2 + (kt >> 1);
                const bf16x8 ot = tr_join(to2[dt - 2][0], to2[dt - 2][1]), qt = tr_join(tq2[dt - 2][0], tq2[dt - 2][1]);
                static_for<0, 2>([&](auto hh) {
                    constexpr int k2 = (kt & 1) * 2 + decltype(hh)::value;
                    mfma_acc<A_DV + 4 * (k2 * 4 + dt)>(pfp[k2], ot);
                    mfma_acc<A_DK + 4 * (k2 * 4 + dt)>(dfp[k2], qt);
                });
                PB_PIN();
                soft(svB[kt], dpB[kt], ktt, IntTag<1>{});
                if constexpr (kt == 0)
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sb[0][0]), "+v"(sb[0][1]), "+v"(sb[1][0]), "+v"(sb[1][1]), "+v"(sb[2][0]), "+v"(sb[2][1]), "+v"(sb[3][0]), "+v"(sb[3][1]),
                                                          "+v"(sb[4][0]), "+v"(sb[4][1]), "+v"(sb[5][0]), "+v"(sb[5][1]), "+v"(sb[6][0]), "+v"(sb[6][1]), "+v"(sb[7][0]), "+v"(sb[7][1]));
                PB_PIN();
                static_for<0, 2>([&](auto hh) {
                    constexpr int s = kt * 2 + decltype(hh)::value;
                    const bf16x8 bs = tr_join(sb[s][0], sb[s][1]);
                    if constexpr (s == 0) { mfma_aav_z<A_KT>(dqa, bs); mfma_aav_z<A_KT + 32>(dqb, bs); }
                    else { mfma_aav<A_KT + 4 * s>(dqa, bs); mfma_aav<A_KT + 4 * (8 + s)>(dqb, bs); }
                });
                PB_PIN();
            });
        } else {
            static_for<0, 4>([&](auto ktt) { soft(svB[decltype(ktt)::value], dpB[decltype(ktt)::value], ktt, IntTag<1>{}); });
        }
        // ---- hand-over to the next step: its dV / dK operands (this step's P^T, dS^T and the transposed dO / Q rows of these 32 queries)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const u32x4 pu = {pw[kt][0][0], pw[kt][0][1], pw[kt][1][0], pw[kt][1][1]}, du = {dw[kt][0][0], dw[kt][0][1], dw[kt][1][0], dw[kt][1][1]};
            pfp[kt] = __builtin_bit_cast(bf16x8, pu); dfp[kt] = __builtin_bit_cast(bf16x8, du);
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const unsigned aq = qoff[dt] + (unsigned)(slot * STB1), ao = aq + 8192;
            if constexpr (HALF == 0) { ds_tr<0>(to[dt][0], ao); ds_tr<2048>(to[dt][1], ao); ds_tr<0>(tq[dt][0], aq); ds_tr<2048>(tq[dt][1], aq); }
            else { ds_tr<4096>(to[dt][0], ao); ds_tr<6144>(to[dt][1], ao); ds_tr<4096>(tq[dt][0], aq); ds_tr<6144>(tq[dt][1], aq); }
        }
        // this wave's dQ^T tiles of the previous step -> its slab rows (the fence: the tiles' last MFMAs are 8 cycles old at least)
        if constexpr (!FIRST) {
            asm volatile("s_nop 15" : "+v"(dqa), "+v"(dqb));
            const int q = (HALF ? T * 64 : T * 64 - 32) + qsel * 16 + lr;
            if (q < p.Sq) {
                bf16_t* row = slab + (long)q * d_model + cpair * 32 + g * 4;
                *reinterpret_cast<bf16x4*>(row) = to_bf4(dqa * p.scale);
                *reinterpret_cast<bf16x4*>(row + 16) = to_bf4(dqb * p.scale);
            }
        }
    };
    auto sync = [&]() {
        // this wave's dS stores and the transposed reads for the next step have landed; behind the barrier every wave's have
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(to[0][0]), "+v"(to[0][1]), "+v"(to[1][0]), "+v"(to[1][1]), "+v"(tq[0][0]), "+v"(tq[0][1]), "+v"(tq[1][0]), "+v"(tq[1][1]) :: "memory");
        __builtin_amdgcn_s_barrier();
    };
    auto run = [&](int T, int slot, int pslot, auto halftag, auto firsttag) {
        constexpr int HALF = decltype(halftag)::value;
        const bool diag = p.causal && (k0 + wave * 64 + 63 > T * 64 + HALF * 32);     // wave-uniform: some key of this wave lies behind some query of the step
        if (diag) step(T, slot, pslot, halftag, firsttag, IntTag<1>{}); else step(T, slot, pslot, halftag, firsttag, IntTag<0>{});
    };

    int slot = 0, pslot = 0;
    run(it0, slot, pslot, IntTag<0>{}, IntTag<1>{});
    sync();
    for (int T = it0;; ++T) {
        if (T > it0 && T + 2 < nt) stage(T + 2, pslot);                    // the slot of tile T - 1: every wave is past its last read of it
        run(T, slot, pslot, IntTag<1>{}, IntTag<0>{});
        if (T + 1 < nt) { if (T + 2 < nt) { wait_vm<4>(); } else { wait_vm<0>(); } }      // tile T + 1 has landed (the 4 youngest pieces may be tile T + 2's)
        sync();
        if (T + 1 >= nt) break;
        pslot = slot;
        slot = slot == RING1 - 1 ? 0 : slot + 1;
        run(T + 1, slot, pslot, IntTag<0>{}, IntTag<0>{});
        sync();
    }
    // ---- the last step's dV / dK / dQ
    {
        const int T = nt - 1;
        s16x4 sb[8][2];
        const unsigned base = ds_rd + 16384;
        ds_tr<0>(sb[0][0], base); ds_tr<512>(sb[0][1], base); ds_tr<1024>(sb[1][0], base); ds_tr<1536>(sb[1][1], base);
        ds_tr<2048>(sb[2][0], base); ds_tr<2560>(sb[2][1], base); ds_tr<3072>(sb[3][0], base); ds_tr<3584>(sb[3][1], base);
        ds_tr<4096>(sb[4][0], base); ds_tr<4608>(sb[4][1], base); ds_tr<5120>(sb[5][0], base); ds_tr<5632>(sb[5][1], base);
        ds_tr<6144>(sb[6][0], base); ds_tr<6656>(sb[6][1], base); ds_tr<7168>(sb[7][0], base); ds_tr<7680>(sb[7][1], base);
        s16x4 to2[2][2], tq2[2][2];
#pragma unroll
        for (int d2 = 0; d2 < 2; ++d2) {
            const unsigned aq = qoff[2 + d2] + (unsigned)(slot * STB1), ao = aq + 8192;
            ds_tr<4096>(to2[d2][0], ao); ds_tr<6144>(to2[d2][1], ao); ds_tr<4096>(tq2[d2][0], aq); ds_tr<6144>(tq2[d2][1], aq);
        }
        static_for<0, 2>([&](auto dtt) {
            constexpr int dt = decltype(dtt)::value;
            const bf16x8 ot = tr_join(to[dt][0], to[dt][1]), qt = tr_join(tq[dt][0], tq[dt][1]);
            static_for<0, 4>([&](auto ktt) {
                constexpr int kt = decltype(ktt)::value;
                mfma_acc<A_DV + 4 * (kt * 4 + dt)>(pfp[kt], ot);
                mfma_acc<A_DK + 4 * (kt * 4 + dt)>(dfp[kt], qt);
            });
        });
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(to2[0][0]), "+v"(to2[0][1]), "+v"(to2[1][0]), "+v"(to2[1][1]), "+v"(tq2[0][0]), "+v"(tq2[0][1]), "+v"(tq2[1][0]), "+v"(tq2[1][1]));
        static_for<0, 2>([&](auto dtt) {
            constexpr int dt = 2 + decltype(dtt)::value;
            const bf16x8 ot = tr_join(to2[dt - 2][0], to2[dt - 2][1]), qt = tr_join(tq2[dt - 2][0], tq2[dt - 2][1]);
            static_for<0, 4>([&](auto ktt) {
                constexpr int kt = decltype(ktt)::value;
                mfma_acc<A_DV + 4 * (kt * 4 + dt)>(pfp[kt], ot);
                mfma_acc<A_DK + 4 * (kt * 4 + dt)>(dfp[kt], qt);
            });
        });
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sb[0][0]), "+v"(sb[0][1]), "+v"(sb[1][0]), "+v"(sb[1][1]), "+v"(sb[2][0]), "+v"(sb[2][1]), "+v"(sb[3][0]), "+v"(sb[3][1]),
                                              "+v"(sb[4][0]), "+v"(sb[4][1]), "+v"(sb[5][0]), "+v"(sb[5][1]), "+v"(sb[6][0]), "+v"(sb[6][1]), "+v"(sb[7][0]), "+v"(sb[7][1]));
        f32x4 dqa, dqb;
        static_for<0, 8>([&](auto ss) {
            constexpr int s = decltype(ss)::value;
            const bf16x8 bs = tr_join(sb[s][0], sb[s][1]);
            if constexpr (s == 0) { mfma_aav_z<A_KT>(dqa, bs); mfma_aav_z<A_KT + 32>(dqb, bs); }
            else { mfma_aav<A_KT + 4 * s>(dqa, bs); mfma_aav<A_KT + 4 * (8 + s)>(dqb, bs); }
        });
        asm volatile("s_nop 15" : "+v"(dqa), "+v"(dqb));
        const int q = T * 64 + 32 + qsel * 16 + lr;
        if (q < p.Sq) {
            bf16_t* row = slab + (long)q * d_model + cpair * 32 + g * 4;
            *reinterpret_cast<bf16x4*>(row) = to_bf4(dqa * p.scale);
            *reinterpret_cast<bf16x4*>(row + 16) = to_bf4(dqb * p.scale);
        }
    }

    // ---- epilogue: dK (x scale), dV rows of this block; masked keys receive zeros; column sums = k / v bias-gradient partials
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    float csk[4], csv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { csk[dt] = 0.f; csv[dt] = 0.f; }
    static_for<0, 4>([&](auto ktt) {
        constexpr int kt = decltype(ktt)::value;
        f32x4 dkr[4], dvr[4];
        static_for<0, 4>([&](auto dtt) {
            constexpr int dt = decltype(dtt)::value;
            dkr[dt] = agpr_get<A_DK + 4 * (kt * 4 + dt)>();
            dvr[dt] = agpr_get<A_DV + 4 * (kt * 4 + dt)>();
        });
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = k0 + wave * 64 + kt * 16 + g * 4 + r;
            if (key < p.Sk) {
                const bool kvis = key < kvis_end && (!p.key_mask || p.key_mask[(long)b * p.Sk + key] != 0.f);
                bf16_t* DK = p.dk + b * p.dk_sb + (long)key * p.dk_ss + h * HDT;
                bf16_t* DV = p.dv + b * p.dv_sb + (long)key * p.dv_ss + h * HDT;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const float vk = kvis ? dkr[dt][r] * p.scale : 0.f, vv = kvis ? dvr[dt][r] : 0.f;
                    DK[dt * 16 + lr] = (bf16_t)vk;
                    DV[dt * 16 + lr] = (bf16_t)vv;
                    csk[dt] += vk; csv[dt] += vv;
                }
            }
        }
    });
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
