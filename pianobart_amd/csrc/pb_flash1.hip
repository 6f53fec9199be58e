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
#include <cstdlib>
#include <algorithm>

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
    const float* delta_rows; long delta_ld;    // optional: delta[h * delta_ld + row of the q side] (made by pb_gemm's PB_GEMM_ROWDOT epilogue) instead of a.delta (B, H, Sq)
    unsigned* stamps;                          // diagnostic build only (-DPB_FA1_STAMPS): 16 per-phase cycle sums + 1 step count, added by every wave
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
template <int RD> __device__ __forceinline__ void mfma_acc(const bf16x8& b, const bf16x8& a) { // a[RD:RD+3] += A(VGPR) x B(VGPR); called as (P^T or dS^T, transposed dO or Q)
    asm volatile("v_mfma_f32_16x16x32_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" :: "v"(a), "v"(b), "i"(RD), "i"(RD + 3) : PB_ALL_AGPRS);
}
// Grouped forms: hipcc closes every asm statement that writes a register with an s_nop (one issue slot of a wave that has its SIMD
// to itself), so the MFMAs that always travel together share a statement.
#define MF "v_mfma_f32_16x16x32_bf16 "
// dV / dK tiles of column tile DT for key tiles K0, K0 + 1 (4 MFMAs). A = the transposed dO / Q fragment (row = column c), B = P^T / dS^T
// (column = key): the accumulator tile holds 4 consecutive columns of ONE key per lane, so the epilogue stores 8 bytes per lane and tile
// (with the operands the other way round a lane would hold 4 keys of one column: 2-byte stores, 29 000 cycles of epilogue per workgroup)
template <int DT, int K0> __device__ __forceinline__ void mfma_dvdk4(const bf16x8& p0, const bf16x8& d0, const bf16x8& p1, const bf16x8& d1, const bf16x8& ot, const bf16x8& qt) {
    asm volatile(MF "a[%c6:%c7], %4, %0, a[%c6:%c7]\n\t" MF "a[%c8:%c9], %5, %1, a[%c8:%c9]\n\t" MF "a[%c10:%c11], %4, %2, a[%c10:%c11]\n\t" MF "a[%c12:%c13], %5, %3, a[%c12:%c13]"
                 :: "v"(p0), "v"(d0), "v"(p1), "v"(d1), "v"(ot), "v"(qt),
                    "i"(A_DV + 4 * (K0 * 4 + DT)), "i"(A_DV + 4 * (K0 * 4 + DT) + 3), "i"(A_DK + 4 * (K0 * 4 + DT)), "i"(A_DK + 4 * (K0 * 4 + DT) + 3),
                    "i"(A_DV + 4 * ((K0 + 1) * 4 + DT)), "i"(A_DV + 4 * ((K0 + 1) * 4 + DT) + 3), "i"(A_DK + 4 * ((K0 + 1) * 4 + DT)), "i"(A_DK + 4 * ((K0 + 1) * 4 + DT) + 3)
                 : PB_ALL_AGPRS);
}
// S^T and dP^T of key tile KT, first k-step: d = A x a[K or V fragment] + c
template <int KT> __device__ __forceinline__ void mfma_sdp0(f32x4& sv, f32x4& dp, const bf16x8& qa, const bf16x8& oa, const f32x4& nl, const f32x4& nd) {
    asm volatile(MF "%0, %2, a[%c6:%c7], %4\n\t" MF "%1, %3, a[%c8:%c9], %5"
                 : "=&v"(sv), "=&v"(dp) : "v"(qa), "v"(oa), "v"(nl), "v"(nd),
                   "i"(A_KF + 4 * (KT * 2)), "i"(A_KF + 4 * (KT * 2) + 3), "i"(A_VF + 4 * (KT * 2)), "i"(A_VF + 4 * (KT * 2) + 3) : PB_ALL_AGPRS);
}
template <int KT> __device__ __forceinline__ void mfma_sdp1(f32x4& sv, f32x4& dp, const bf16x8& qa, const bf16x8& oa) {     // second k-step
    asm volatile(MF "%0, %2, a[%c4:%c5], %0\n\t" MF "%1, %3, a[%c6:%c7], %1"
                 : "+v"(sv), "+v"(dp) : "v"(qa), "v"(oa),
                   "i"(A_KF + 4 * (KT * 2 + 1)), "i"(A_KF + 4 * (KT * 2 + 1) + 3), "i"(A_VF + 4 * (KT * 2 + 1)), "i"(A_VF + 4 * (KT * 2 + 1) + 3) : PB_ALL_AGPRS);
}
// dQ^T: k-steps S0 (chain 0) and S0 + 1 (chain 1) of both column tiles; ZERO: the chains start here (C = the constant 0)
template <int S0, bool ZERO> __device__ __forceinline__ void mfma_dq4(f32x4& a0, f32x4& b0, f32x4& a1, f32x4& b1, const bf16x8& bs0, const bf16x8& bs1) {
    if constexpr (ZERO)
        asm volatile(MF "%0, a[%c6:%c7], %4, 0\n\t" MF "%1, a[%c8:%c9], %4, 0\n\t" MF "%2, a[%c10:%c11], %5, 0\n\t" MF "%3, a[%c12:%c13], %5, 0"
                     : "=&v"(a0), "=&v"(b0), "=&v"(a1), "=&v"(b1) : "v"(bs0), "v"(bs1),
                       "i"(A_KT + 4 * S0), "i"(A_KT + 4 * S0 + 3), "i"(A_KT + 4 * (8 + S0)), "i"(A_KT + 4 * (8 + S0) + 3),
                       "i"(A_KT + 4 * (S0 + 1)), "i"(A_KT + 4 * (S0 + 1) + 3), "i"(A_KT + 4 * (9 + S0)), "i"(A_KT + 4 * (9 + S0) + 3) : PB_ALL_AGPRS);
    else
        asm volatile(MF "%0, a[%c6:%c7], %4, %0\n\t" MF "%1, a[%c8:%c9], %4, %1\n\t" MF "%2, a[%c10:%c11], %5, %2\n\t" MF "%3, a[%c12:%c13], %5, %3"
                     : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1) : "v"(bs0), "v"(bs1),
                       "i"(A_KT + 4 * S0), "i"(A_KT + 4 * S0 + 3), "i"(A_KT + 4 * (8 + S0)), "i"(A_KT + 4 * (8 + S0) + 3),
                       "i"(A_KT + 4 * (S0 + 1)), "i"(A_KT + 4 * (S0 + 1) + 3), "i"(A_KT + 4 * (9 + S0)), "i"(A_KT + 4 * (9 + S0) + 3) : PB_ALL_AGPRS);
}
#undef MF
// LDS reads the compiler must not count (it would drain what we keep in flight): destinations are valid behind a wait statement that
// names them
template <int OFF, class T> __device__ __forceinline__ void ds_rd128(T& d, unsigned addr) {
    static_assert(sizeof(T) == 16, "16-byte fragment");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "i"(OFF));
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
#ifdef PB_FA1_STAMPS
    const unsigned long long st_in = __builtin_amdgcn_s_memrealtime();      // workgroup trace: entry / exit times (the 100 MHz counter all XCDs share; s_memtime is per XCD at best) and the CU it ran on (tools/flash1_stamps.py --gaps)
#endif
    const int nkb0 = (pin.a.Sk + KB1 - 1) / KB1;
    int rb, h, b;
    block_map(nkb0, pin.a.H, pin.a.B, rb, h, b, pin.a.bh_order);
    const int k0 = rb * KB1;
    Fa64Args p = pin.a;
    const long qrow0 = pin.a.vl_q_off ? (long)pin.a.vl_q_off[b] : (long)b * pin.a.Sq;     // first row of this sequence on the q side: read ONCE, beside varlen_localize's own
    varlen_localize(p, b);                                                                  // scalar loads (a second, vector load of it further down cost a memory round trip each)
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
#ifdef PB_FA1_STAMPS
    const unsigned st_t0 = (unsigned)__builtin_amdgcn_s_memtime();
    unsigned st_p[6] = {}, st_q[4] = {};
#define PSTAMP(i) st_p[i] = (unsigned)__builtin_amdgcn_s_memtime();
#define QSTAMP(i) st_q[i] = (unsigned)__builtin_amdgcn_s_memtime();
#else
#define PSTAMP(i)
#define QSTAMP(i)
#endif
    const bf16_t* Q = p.q + b * p.q_sb + h * HDT;
    const bf16_t* DO = p.dout + b * p.o_sb + h * HDT;
    bf16_t* slab = pin.slab + (long)rb * pin.slab_stride + (pin.a.vl_q_off ? qrow0 * d_model : (long)b * pin.slab_sb) + h * HDT;
    const float c = p.scale * LOG2E;
    // ---- prologue. Everything is requested before anything is waited for (one memory round trip, not one per loop iteration): the
    // ordinary loads first -- K / V fragments of this wave's 64 keys, the -lse / -delta rows of the sequence -- then the DMA of the 4 K
    // images and of the first three {Q, dO} tiles; then the loads are consumed.
    bf16x8 kfr[4][2], vfr[4][2];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        const int key = k0 + wave * 64 + kt * 16 + lr;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            kfr[kt][ks] = frag_global(K, p.k_ss, key, p.Sk, ks * 32 + g * 8);
            vfr[kt][ks] = frag_global(V, p.v_ss, key, p.Sk, ks * 32 + g * 8);
        }
    }
    QSTAMP(0);
    float* ldsNL = reinterpret_cast<float*>(smem + OFF_TAB);
    float* ldsND = ldsNL + nt * 64;
    float* ldsVis = reinterpret_cast<float*>(smem + OFF_DS);               // 1 / 0 per key of the block (the dS buffers are not in use yet)
    const long li0 = ((long)b * p.H + h) * lse_ld;
    const float* dsrc = pin.delta_rows ? pin.delta_rows + (long)h * pin.delta_ld + qrow0 : p.delta + li0;
    float tl[4], td[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = it0 * 64 + t + i * FT;
        tl[i] = q < p.Sq ? p.lse[li0 + q] : INFINITY;
        td[i] = q < p.Sq ? dsrc[q] : 0.f;
    }
    QSTAMP(1);
    const int key_t = k0 + t;                                              // FT = KB1 = 256: one key per thread
    const float vis_t = (key_t < kvis_end && (!p.key_mask || p.key_mask[(long)b * p.Sk + key_t] != 0.f)) ? 1.f : 0.f;
    // {K, Q, dO} tiles by buffer_load ... lds: a descriptor per operand that ends with this sequence's last row (rows beyond it read as
    // zeros: no clamping, no ragged path), a 32-bit lane offset per DMA piece (row within the tile, swizzled chunk), the tile in the scalar offset
    const StageOff so_k = stage_off(p.k_ss, wave, lane), so_q = stage_off(p.q_ss, wave, lane), so_o = stage_off(p.o_ss, wave, lane);
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(K), 0, (int)(((long)(p.Sk - 1) * p.k_ss + HDT) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Q), 0, (int)(((long)(p.Sq - 1) * p.q_ss + HDT) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(DO), 0, (int)(((long)(p.Sq - 1) * p.o_ss + HDT) * 2), 0x00020000);
    // the slab rows of this (sequence, head): rows beyond the sequence fall outside the descriptor and are dropped by the hardware
    const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc(slab, 0, (int)(((long)(p.Sq - 1) * d_model + HDT) * 2), 0x00020000);
    // dQ slab stores (round 5, late): a wave's dQ^T piece of a step is 16 query rows x 32 columns, and the MFMA layout leaves a row's 8-byte pieces on lanes 16
    // apart -- two store instructions of 64 separate 8-byte writes each (the memory pipeline merges adjacent lanes only; without the stores the whole backward
    // measured 7 % shorter). The piece goes through a 1 KiB LDS image of the wave (in the K images, dead since the K^T fragments were read: [16 rows][64 B],
    // 16-byte slot XOR (row >> 1) & 3) and leaves as ONE store of 16 bytes per lane, 4 adjacent lanes per 64-byte row; the read-back is waited for by the
    // lgkmcnt(0) every step ends with (sync), and the store is issued behind it.
    const unsigned stg = lds_u32(smem) + OFF_K + (unsigned)(wave * 1024);
    const unsigned stg_wa = stg + (unsigned)(lr * 64 + ((((g >> 1) ^ ((lr >> 1) & 3)) << 4) | ((g & 1) << 3)));            // columns 4 g .. of tile A; tile B: 16-byte slots 2, 3
    const unsigned stg_wb = stg + (unsigned)(lr * 64 + (((2 + (g >> 1)) ^ ((lr >> 1) & 3)) << 4 | ((g & 1) << 3)));
    const int srow = lane >> 2, schunk = lane & 3;
    const unsigned stg_rd = stg + (unsigned)(srow * 64 + ((schunk ^ ((srow >> 1) & 3)) << 4));
    const unsigned slab_vo = (unsigned)((((wave & 1) * 16 + srow) * d_model + (wave >> 1) * 32 + schunk * 8) * 2);
    u32x4 slab_v = {0u, 0u, 0u, 0u};
    int slab_qs = 0;
    bool slab_pending = false;
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
    auto slab_put = [&](const f32x4& a, const f32x4& b_, int qs) {          // the two pieces in, the row-major 16 bytes requested
        *reinterpret_cast<__attribute__((address_space(3))) u32x2_t*>(stg_wa) = __builtin_bit_cast(u32x2_t, to_bf4(a));
        *reinterpret_cast<__attribute__((address_space(3))) u32x2_t*>(stg_wb) = __builtin_bit_cast(u32x2_t, to_bf4(b_));
        ds_rd128<0>(slab_v, stg_rd);
        slab_qs = qs; slab_pending = true;
    };
    auto slab_flush = [&]() {                                               // behind a wait that names slab_v
        if (slab_pending) { __builtin_amdgcn_raw_buffer_store_b128(slab_v, rsS, slab_vo, slab_qs, 0); slab_pending = false; }
    };
    const unsigned voQ0 = so_q.off[0] * 2, voQ1 = so_q.off[1] * 2, voO0 = so_o.off[0] * 2, voO1 = so_o.off[1] * 2;
    typedef __attribute__((address_space(3))) void* lds_t;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int sk = (k0 + 64 * m) * (int)p.k_ss * 2;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (lds_t)(smem + OFF_K + m * 8192 + wave * 2048), 16, so_k.off[0] * 2, sk, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (lds_t)(smem + OFF_K + m * 8192 + wave * 2048 + 1024), 16, so_k.off[1] * 2, sk, 0, 0);
    }
    auto stage = [&](int it, int slot) {
        char* st = smem + slot * STB1 + wave * 2048;
        const int sq = it * 64 * (int)p.q_ss * 2, so = it * 64 * (int)p.o_ss * 2;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (lds_t)st, 16, voQ0, sq, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (lds_t)(st + 1024), 16, voQ1, sq, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsO, (lds_t)(st + 8192), 16, voO0, so, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsO, (lds_t)(st + 8192 + 1024), 16, voO1, so, 0, 0);
    };
    QSTAMP(2);
    const int npre = min(3, nt - it0);                                     // tiles requested up front (the ring is empty)
    stage(it0, 0);
    if (npre > 1) stage(it0 + 1, 1);
    if (npre > 2) stage(it0 + 2, 2);
    PSTAMP(0);
    // ---- consume: tables
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = it0 * 64 + t + i * FT;
        if (q < nt * 64) { ldsNL[q] = tl[i] == INFINITY ? -INFINITY : -tl[i] * LOG2E; ldsND[q] = -td[i]; }
    }
    for (int q = it0 * 64 + t + 4 * FT; q < nt * 64; q += FT) {            // sequences beyond 1024 queries: the rest, one round trip per 256
        const float ls = q < p.Sq ? p.lse[li0 + q] : INFINITY;
        ldsNL[q] = ls == INFINITY ? -INFINITY : -ls * LOG2E;
        ldsND[q] = q < p.Sq ? -dsrc[q] : 0.f;
    }
    ldsVis[t] = vis_t;
    // A masked key's score is exactly -lse (its K fragment is zero): p = exp2(-lse log2e) overflows for a row whose log-sum-exp lies below about
    // -88, and inf x 0 in dQ^T = K^T dS^T is a NaN. Such rows are found here, once: if the sequence has one, the waves that hold a
    // masked key run their steps through the compare path below, whose exponent is clamped at 0 (a true log-probability is <= 0).
    bool hot = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) hot |= (tl[i] != INFINITY && -tl[i] * LOG2E > 100.f);
    for (int q = it0 * 64 + t + 4 * FT; q < p.Sq; q += FT) hot |= (-p.lse[li0 + q] * LOG2E > 100.f);
    unsigned* ldsHot = reinterpret_cast<unsigned*>(smem + OFF_DS + 1024);   // 4 words behind ldsVis
    const unsigned hotw = __builtin_amdgcn_ballot_w64(hot) != 0ull ? 1u : 0u;
    if (lane == 0) ldsHot[wave] = hotw;
    PSTAMP(1);
    // this wave's 64 keys: K (prescaled: S comes out of the MFMA in log2 units) and V fragments go to AGPRs for the whole sweep. A MASKED
    // key's K fragments are zeros here and in the K^T fragments below: its scores are then -lse (p finite), its dS meets a zero K row in
    // dQ, and its own dK / dV rows are zeroed in the epilogue -- the sweep itself never looks at a key mask.
    bool allvis = true;
    static_for<0, 4>([&](auto ktt) {
        constexpr int kt = decltype(ktt)::value;
        const int key = k0 + wave * 64 + kt * 16 + lr;
        const bool vis = key < kvis_end && (!p.key_mask || p.key_mask[(long)b * p.Sk + key] != 0.f);
        allvis &= vis;
        static_for<0, 2>([&](auto kss) {
            constexpr int ks = decltype(kss)::value;
            bf16x8 kf = scale_frag(kfr[kt][ks], c);
            const bf16x8 z = {};
            if (!vis) kf = z;
            agpr_put<A_KF + 4 * (kt * 2 + ks)>(kf);
            agpr_put<A_VF + 4 * (kt * 2 + ks)>(vfr[kt][ks]);
        });
    });
    static_for<0, 128>([&](auto r) { agpr_zero<decltype(r)::value>(); });
    PSTAMP(2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PSTAMP(3);
    __builtin_amdgcn_s_waitcnt(0xc07f);                                    // this wave's table stores
    __builtin_amdgcn_s_barrier();                                          // K images and tables are in LDS
    const bool careful = __builtin_amdgcn_ballot_w64(!allvis) != 0ull &&
                         __builtin_amdgcn_readfirstlane(ldsHot[0] | ldsHot[1] | ldsHot[2] | ldsHot[3]) != 0u;      // wave-uniform
    PSTAMP(4);
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
    PSTAMP(5);
    unsigned qoff[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) qoff[dt] = lds_u32(smem) + tr_lane_off64(dt * 16, lane);
    // row fragments of a {Q, dO} tile: row 16 qt + lr, k-step ks -> byte offset qt * 2048 + frow[ks] (+ 8192 for dO): the swizzle only sees lr
    unsigned frow[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) frow[ks] = lds_u32(smem) + (unsigned)(lr * 128 + (((ks * 4 + g) ^ fsw(lr)) << 4));
    const unsigned tab_nl = lds_u32(smem) + OFF_TAB + (unsigned)(g * 16), tab_nd = tab_nl + (unsigned)(nt * 256);     // + q * 4
    // dS image addressing: row = key within the block, 32-B rows of one 16-query tile, 8-byte slot (4 queries) XOR (row >> 2) & 3
    const unsigned ds_wr = lds_u32(smem) + OFF_DS + (unsigned)((wave * 64 + lr) * 32 + ((g ^ (lr >> 2)) << 3));     // + half * 16384 + qq * 8192 + kt * 512
    const unsigned ds_rd = lds_u32(smem) + OFF_DS + (unsigned)(qsel * 8192 + (4 * g + (lr >> 2)) * 32 + (((lr & 3) ^ g) << 3));   // + half * 16384 + s' * 1024 (+ 512)
    if (npre == 3) { wait_vm<8>(); } else if (npre == 2) { wait_vm<4>(); } else { wait_vm<0>(); }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();                                          // tile it0 has landed for every wave; ldsVis has been read (the dS buffers may be written)

    // state carried from one 32-query step to the next: the packed P^T / dS^T operands and the transposed dO / Q fragments (column
    // tiles 0, 1) of the previous step: its dV / dK / dQ products run beside this step's softmax
    bf16x8 pfp[4], dfp[4];
    s16x4 to[2][2], tq[2][2];
    const int mykey0 = k0 + wave * 64 + lr;

    // One step = 32 queries (half HALF of the 64-query tile T in ring slot `slot`). The wave has its SIMD to itself, so every LDS read is
    // requested a phase ahead of its use and waited for by count; MFMA stream of a step (80): dV/dK(previous step) column tiles 0, 1 |
    // S^T, dP^T of query tile A | S^T, dP^T of B beside exp / dS of A | dV/dK(previous) column tiles 2, 3 and dQ^T(previous) beside exp /
    // dS of B. One barrier per step: behind it dS(this step) is visible and the other dS buffer is free.
#ifdef PB_FA1_STAMPS
    unsigned st_acc[12] = {}, st_last = 0, st_steps = 0;
#endif
#if defined(PB_FA1_STAMPS) && PB_FA1_STAMPS > 1
#define STAMP(i) { const unsigned now_ = (unsigned)__builtin_amdgcn_s_memtime(); st_acc[i] += now_ - st_last; st_last = now_; }
#else
#define STAMP(i)
#endif
    auto step = [&](int T, int slot, int pslot, int stage_tile, auto halftag, auto firsttag, auto diagtag) {
        constexpr int HALF = decltype(halftag)::value;
        constexpr bool FIRST = decltype(firsttag)::value != 0, DIAG = decltype(diagtag)::value != 0;
        constexpr int PH = 1 - HALF;                                       // half (and dS buffer) of the previous step
        constexpr int qtA = HALF * 2, qtB = HALF * 2 + 1;
        const int q0 = T * 64;
        const unsigned fq0 = frow[0] + (unsigned)(slot * STB1), fq1 = frow[1] + (unsigned)(slot * STB1);
        const unsigned tnl = tab_nl + (unsigned)(q0 * 4), tnd = tab_nd + (unsigned)(q0 * 4);
        // ---- requests right behind the barrier: query tile A's row fragments and constants (6), the previous step's transposed dO / Q
        // fragments of column tiles 2, 3 (8; its tile is still in the ring: slot `pslot` when this step opens a new tile)
        f32x4 nlA, ndA, nlB, ndB;
        bf16x8 qaA0, oaA0, qaA1, oaA1, qaB0, oaB0, qaB1, oaB1;
        STAMP(0);                                                          // barrier exit -> here
        ds_rd128<qtA * 64>(nlA, tnl); ds_rd128<qtA * 64>(ndA, tnd);
        ds_rd128<qtA * 2048>(qaA0, fq0); ds_rd128<qtA * 2048 + 8192>(oaA0, fq0); ds_rd128<qtA * 2048>(qaA1, fq1); ds_rd128<qtA * 2048 + 8192>(oaA1, fq1);
        s16x4 to2[2][2], tq2[2][2];
        if constexpr (!FIRST) {
            // ---- dV^T, dK^T of the previous step, column tiles 0, 1 (operands in registers since before the barrier); between the MFMA
            // groups: the transposed requests and, when a ring slot has come free, the DMA of the tile after next
            const bf16x8 ot0 = tr_join(to[0][0], to[0][1]), qt0 = tr_join(tq[0][0], tq[0][1]);
            mfma_dvdk4<0, 0>(pfp[0], dfp[0], pfp[1], dfp[1], ot0, qt0);
#pragma unroll
            for (int d2 = 0; d2 < 2; ++d2) {
                const unsigned aq = qoff[2 + d2] + (unsigned)((HALF ? slot : pslot) * STB1), ao = aq + 8192;
                if constexpr (PH == 0) { ds_tr<0>(to2[d2][0], ao); ds_tr<2048>(to2[d2][1], ao); ds_tr<0>(tq2[d2][0], aq); ds_tr<2048>(tq2[d2][1], aq); }
                else { ds_tr<4096>(to2[d2][0], ao); ds_tr<6144>(to2[d2][1], ao); ds_tr<4096>(tq2[d2][0], aq); ds_tr<6144>(tq2[d2][1], aq); }
            }
            mfma_dvdk4<0, 2>(pfp[2], dfp[2], pfp[3], dfp[3], ot0, qt0);
            if (stage_tile >= 0) stage(stage_tile, pslot);                 // HALF == 1 steps only: the slot of tile T - 1, every wave is past its last read of it
            const bf16x8 ot1 = tr_join(to[1][0], to[1][1]), qt1 = tr_join(tq[1][0], tq[1][1]);
            mfma_dvdk4<1, 0>(pfp[0], dfp[0], pfp[1], dfp[1], ot1, qt1);
            mfma_dvdk4<1, 2>(pfp[2], dfp[2], pfp[3], dfp[3], ot1, qt1);
            STAMP(1);                                                      // requests + 16 MFMAs
            asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(nlA), "+v"(ndA), "+v"(qaA0), "+v"(oaA0), "+v"(qaA1), "+v"(oaA1));
            STAMP(2);                                                      // wait: query tile A
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(nlA), "+v"(ndA), "+v"(qaA0), "+v"(oaA0), "+v"(qaA1), "+v"(oaA1));
        }
        // ---- S^T, dP^T of query tile A; query tile B's fragments are requested between its two k-steps
        f32x4 svA[4], dpA[4], svB[4], dpB[4];
        static_for<0, 4>([&](auto ktt) { constexpr int kt = decltype(ktt)::value; mfma_sdp0<kt>(svA[kt], dpA[kt], qaA0, oaA0, nlA, ndA); });
        ds_rd128<qtB * 64>(nlB, tnl); ds_rd128<qtB * 64>(ndB, tnd);
        ds_rd128<qtB * 2048>(qaB0, fq0); ds_rd128<qtB * 2048 + 8192>(oaB0, fq0); ds_rd128<qtB * 2048>(qaB1, fq1); ds_rd128<qtB * 2048 + 8192>(oaB1, fq1);
        static_for<0, 4>([&](auto ktt) { constexpr int kt = decltype(ktt)::value; mfma_sdp1<kt>(svA[kt], dpA[kt], qaA1, oaA1); });
        STAMP(3);                                                          // 16 MFMAs of A + requests of B
        if constexpr (!FIRST)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(nlB), "+v"(ndB), "+v"(qaB0), "+v"(oaB0), "+v"(qaB1), "+v"(oaB1),
                                                  "+v"(to2[0][0]), "+v"(to2[0][1]), "+v"(to2[1][0]), "+v"(to2[1][1]), "+v"(tq2[0][0]), "+v"(tq2[0][1]), "+v"(tq2[1][0]), "+v"(tq2[1][1]));
        else
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(nlB), "+v"(ndB), "+v"(qaB0), "+v"(oaB0), "+v"(qaB1), "+v"(oaB1));
        // exp / dS of one key tile of one query tile: p = exp2(s'), ds = p dp; pack; the dS piece goes to LDS for dQ^T. The multiplies and
        // the packing are single instructions by name: left to itself hipcc pairs the multiplies into v_pk_mul_f32 (an anti-lever beside
        // MFMAs, MI355X_MICROARCH.md) and then rebuilds the bf16 pairs with v_perm / v_alignbit
        auto next_tr = [&]() {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const unsigned aq = qoff[dt] + (unsigned)(slot * STB1), ao = aq + 8192;
                if constexpr (HALF == 0) { ds_tr<0>(to[dt][0], ao); ds_tr<2048>(to[dt][1], ao); ds_tr<0>(tq[dt][0], aq); ds_tr<2048>(tq[dt][1], aq); }
                else { ds_tr<4096>(to[dt][0], ao); ds_tr<6144>(to[dt][1], ao); ds_tr<4096>(tq[dt][0], aq); ds_tr<6144>(tq[dt][1], aq); }
            }
        };
        unsigned pw[4][2][2], dw[4][2][2];                                 // [key tile][query tile][dword]: bf16 pairs of p and of dS
        STAMP(4);                                                          // wait: query tile B, transposed fragments
        const unsigned wr = ds_wr + HALF * 16384;
        auto soft = [&](f32x4& sv, f32x4& dp, auto ktt, auto qqt) {
            constexpr int kt = decltype(ktt)::value, qq = decltype(qqt)::value;
            if constexpr (!DIAG) {
                // one statement (hipcc pads each asm statement that writes a register): 4 exp | the wait state a v_exp result needs before a
                // vector instruction reads it | pack p | 4 multiplies | pack dS
                float e0, e1, e2, e3;
                asm("v_exp_f32 %4, %8\n\tv_exp_f32 %5, %9\n\tv_exp_f32 %6, %10\n\tv_exp_f32 %7, %11\n\ts_nop 0\n\t"
                    "v_cvt_pk_bf16_f32 %0, %4, %5\n\tv_cvt_pk_bf16_f32 %1, %6, %7\n\t"
                    "v_mul_f32 %4, %4, %12\n\tv_mul_f32 %5, %5, %13\n\tv_mul_f32 %6, %6, %14\n\tv_mul_f32 %7, %7, %15\n\t"
                    "v_cvt_pk_bf16_f32 %2, %4, %5\n\tv_cvt_pk_bf16_f32 %3, %6, %7"
                    : "=&v"(pw[kt][qq][0]), "=&v"(pw[kt][qq][1]), "=&v"(dw[kt][qq][0]), "=&v"(dw[kt][qq][1]), "=&v"(e0), "=&v"(e1), "=&v"(e2), "=&v"(e3)
                    : "v"(sv[0]), "v"(sv[1]), "v"(sv[2]), "v"(sv[3]), "v"(dp[0]), "v"(dp[1]), "v"(dp[2]), "v"(dp[3]));
            } else {
                float pr[4], ds[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pr[r] = __builtin_amdgcn_exp2f(fminf(sv[r], 0.f));                    // clamped: a masked key's -lse cannot overflow (see `careful`)
                    pr[r] = (!p.causal || mykey0 + kt * 16 <= q0 + (HALF * 2 + qq) * 16 + g * 4 + r) ? pr[r] : 0.f;
                }
                asm volatile("s_nop 0" : "+v"(pr[0]), "+v"(pr[1]), "+v"(pr[2]), "+v"(pr[3]));
#pragma unroll
                for (int r = 0; r < 4; ++r) asm("v_mul_f32 %0, %1, %2" : "=v"(ds[r]) : "v"(pr[r]), "v"(dp[r]));
                asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pw[kt][qq][0]) : "v"(pr[0]), "v"(pr[1]));
                asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pw[kt][qq][1]) : "v"(pr[2]), "v"(pr[3]));
                asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(dw[kt][qq][0]) : "v"(ds[0]), "v"(ds[1]));
                asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(dw[kt][qq][1]) : "v"(ds[2]), "v"(ds[3]));
            }
            typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
            const u32x2 w = {dw[kt][qq][0], dw[kt][qq][1]};
            *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(wr + qq * 8192 + kt * 512) = w;
        };
        // ---- S^T, dP^T of query tile B beside the softmax of A: two MFMAs, a key tile's vector work, two MFMAs
        s16x4 sb[8][2];
        const unsigned sbase = ds_rd + PH * 16384;
        static_for<0, 4>([&](auto ktt) {
            constexpr int kt = decltype(ktt)::value;
            mfma_sdp0<kt>(svB[kt], dpB[kt], qaB0, oaB0, nlB, ndB);
            PB_PIN();
            soft(svA[kt], dpA[kt], ktt, IntTag<0>{});
            PB_PIN();
            mfma_sdp1<kt>(svB[kt], dpB[kt], qaB1, oaB1);
            if constexpr (!FIRST && kt == 2) {                             // dS^T of the previous step, k-steps 0 .. 3: first use half a phase away
                ds_tr<0>(sb[0][0], sbase); ds_tr<512>(sb[0][1], sbase); ds_tr<1024>(sb[1][0], sbase); ds_tr<1536>(sb[1][1], sbase);
                ds_tr<2048>(sb[2][0], sbase); ds_tr<2560>(sb[2][1], sbase); ds_tr<3072>(sb[3][0], sbase); ds_tr<3584>(sb[3][1], sbase);
            }
            PB_PIN();
        });
        // ---- dV^T, dK^T of the previous step, column tiles 2, 3, and its dQ^T (two chains per column tile), beside the softmax of B
        f32x4 dqa0, dqb0, dqa1, dqb1;
        STAMP(5);                                                          // B's 16 MFMAs beside A's softmax
        if constexpr (!FIRST) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sb[0][0]), "+v"(sb[0][1]), "+v"(sb[1][0]), "+v"(sb[1][1]), "+v"(sb[2][0]), "+v"(sb[2][1]), "+v"(sb[3][0]), "+v"(sb[3][1]));
            ds_tr<4096>(sb[4][0], sbase); ds_tr<4608>(sb[4][1], sbase); ds_tr<5120>(sb[5][0], sbase); ds_tr<5632>(sb[5][1], sbase);
            ds_tr<6144>(sb[6][0], sbase); ds_tr<6656>(sb[6][1], sbase); ds_tr<7168>(sb[7][0], sbase); ds_tr<7680>(sb[7][1], sbase);
            static_for<0, 4>([&](auto ktt) {
                constexpr int kt = decltype(ktt)::value;                   // here: a slot index; the MFMAs of the slot are fixed below
                constexpr int d2 = kt >> 1, k2 = (kt & 1) * 2;
                const bf16x8 ot = tr_join(to2[d2][0], to2[d2][1]), qt = tr_join(tq2[d2][0], tq2[d2][1]);
                mfma_dvdk4<2 + d2, k2>(pfp[k2], dfp[k2], pfp[k2 + 1], dfp[k2 + 1], ot, qt);
                PB_PIN();
                soft(svB[kt], dpB[kt], ktt, IntTag<1>{});
                if constexpr (kt == 1) {
                    // k-steps 4 .. 7 of dS^T have landed (requested a slot and a half ago); the transposed dO / Q fragments of THIS step's
                    // 32 queries (column tiles 0, 1) are requested for the next step (the registers of the previous step's are free since
                    // the first MFMA group): they land long before the barrier
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sb[4][0]), "+v"(sb[4][1]), "+v"(sb[5][0]), "+v"(sb[5][1]), "+v"(sb[6][0]), "+v"(sb[6][1]), "+v"(sb[7][0]), "+v"(sb[7][1]));
                    next_tr();
                }
                PB_PIN();
                const bf16x8 bs0 = tr_join(sb[2 * kt][0], sb[2 * kt][1]), bs1 = tr_join(sb[2 * kt + 1][0], sb[2 * kt + 1][1]);
                mfma_dq4<2 * kt, kt == 0>(dqa0, dqb0, dqa1, dqb1, bs0, bs1);
                PB_PIN();
            });
        } else {
            static_for<0, 4>([&](auto ktt) { soft(svB[decltype(ktt)::value], dpB[decltype(ktt)::value], ktt, IntTag<1>{}); });
        }
        STAMP(6);                                                          // 32 MFMAs beside B's softmax
        // ---- hand-over to the next step: its dV / dK operands (this step's P^T, dS^T; the transposed dO / Q rows of these 32 queries are
        // on their way)
        if constexpr (FIRST) next_tr();
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const u32x4 pu = {pw[kt][0][0], pw[kt][0][1], pw[kt][1][0], pw[kt][1][1]}, du = {dw[kt][0][0], dw[kt][0][1], dw[kt][1][0], dw[kt][1][1]};
            pfp[kt] = __builtin_bit_cast(bf16x8, pu); dfp[kt] = __builtin_bit_cast(bf16x8, du);
        }
        // this wave's dQ^T tiles of the previous step -> its slab rows (the fence: the chains' last MFMAs are 8 instructions old at least)
        if constexpr (!FIRST) {
            asm volatile("s_nop 15" : "+v"(dqa0), "+v"(dqb0), "+v"(dqa1), "+v"(dqb1));
            typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
            const int qs = (HALF ? T * 64 : T * 64 - 32) * d_model * 2;    // first row of the previous step, in bytes
            slab_put((dqa0 + dqa1) * p.scale, (dqb0 + dqb1) * p.scale, qs);
        }
        STAMP(7);                                                          // hand-over, dQ stores
#ifdef PB_FA1_STAMPS
        ++st_steps;
#endif
    };
    auto sync = [&]() {
        // this wave's dS stores and the transposed reads for the next step have landed; behind the barrier every wave's have
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(to[0][0]), "+v"(to[0][1]), "+v"(to[1][0]), "+v"(to[1][1]), "+v"(tq[0][0]), "+v"(tq[0][1]), "+v"(tq[1][0]), "+v"(tq[1][1]), "+v"(slab_v) :: "memory");
        slab_flush();
        STAMP(8);                                                          // DMA wait + LDS wait in front of the barrier
        __builtin_amdgcn_s_barrier();
        STAMP(9);                                                          // barrier
    };
    auto run = [&](int T, int slot, int pslot, int stage_tile, auto halftag, auto firsttag) {
        constexpr int HALF = decltype(halftag)::value;
        const bool diag = careful || (p.causal && (k0 + wave * 64 + 63 > T * 64 + HALF * 32));     // wave-uniform: some key of this wave lies behind some query of the step (or: clamp)
        if (diag) step(T, slot, pslot, stage_tile, halftag, firsttag, IntTag<1>{}); else step(T, slot, pslot, stage_tile, halftag, firsttag, IntTag<0>{});
    };

#ifdef PB_FA1_STAMPS
    st_last = (unsigned)__builtin_amdgcn_s_memtime();
    const unsigned st_t1 = st_last;
#endif
    int slot = 0, pslot = 0;
    run(it0, slot, pslot, -1, IntTag<0>{}, IntTag<1>{});
    sync();
    for (int T = it0;; ++T) {
        run(T, slot, pslot, (T > it0 && T + 2 < nt) ? T + 2 : -1, IntTag<1>{}, IntTag<0>{});
        if (T + 1 < nt) { if (T + 2 < nt) { wait_vm<4>(); } else { wait_vm<0>(); } }      // tile T + 1 has landed (the 4 youngest pieces may be tile T + 2's)
        sync();
        if (T + 1 >= nt) break;
        pslot = slot;
        slot = slot == RING1 - 1 ? 0 : slot + 1;
        run(T + 1, slot, pslot, -1, IntTag<0>{}, IntTag<0>{});
        sync();
    }
#ifdef PB_FA1_STAMPS
    const unsigned st_t2 = (unsigned)__builtin_amdgcn_s_memtime();
#endif
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
        typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
        const int qs = (T * 64 + 32) * d_model * 2;
        slab_put(dqa * p.scale, dqb * p.scale, qs);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(slab_v) :: "memory");
        slab_flush();
    }

    // ---- epilogue: dK (x scale), dV rows of this block; masked keys receive zeros; column sums = k / v bias-gradient partials
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
#ifdef PB_FA1_STAMPS
    const unsigned st_e0 = (unsigned)__builtin_amdgcn_s_memtime();
#endif
    // tile (kt, dt): lane = key 16 kt + lr of this wave, registers = columns 16 dt + 4 g + r
    f32x4 csk[4], csv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { csk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; csv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    static_for<0, 4>([&](auto ktt) {
        constexpr int kt = decltype(ktt)::value;
        const int key = k0 + wave * 64 + kt * 16 + lr;
        const bool kvis = key < kvis_end && (!p.key_mask || p.key_mask[(long)b * p.Sk + (key < p.Sk ? key : 0)] != 0.f);
        // 16 bytes per lane and store: the lanes of the four 16-lane rows hold columns 4 g .. 4 g + 3 of a tile; one v_permlane16_swap per dword
        // hands rows 0 / 2 their odd neighbour's half of column tile dt0 and rows 1 / 3 their even neighbour's half of column tile dt1, so a lane
        // ends with 8 consecutive columns (the store tail is paid per instruction: 16 stores per lane instead of 32)
        bf16_t* DK = p.dk + b * p.dk_sb + (long)key * p.dk_ss + h * HDT + (g >> 1) * 8;
        bf16_t* DV = p.dv + b * p.dv_sb + (long)key * p.dv_ss + h * HDT + (g >> 1) * 8;
        static_for<0, 2>([&](auto dpp) {
            constexpr int dt0 = 2 * decltype(dpp)::value, dt1 = dt0 + 1;
            f32x4 vk0 = agpr_get<A_DK + 4 * (kt * 4 + dt0)>() * p.scale, vv0 = agpr_get<A_DV + 4 * (kt * 4 + dt0)>();
            f32x4 vk1 = agpr_get<A_DK + 4 * (kt * 4 + dt1)>() * p.scale, vv1 = agpr_get<A_DV + 4 * (kt * 4 + dt1)>();
            if (!kvis) { vk0 = vv0 = vk1 = vv1 = f32x4{0.f, 0.f, 0.f, 0.f}; }
            csk[dt0] += vk0; csv[dt0] += vv0; csk[dt1] += vk1; csv[dt1] += vv1;
            typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
            auto pair16 = [&](const f32x4& x, const f32x4& y) {
                const u32x2 xu = __builtin_bit_cast(u32x2, to_bf4(x)), yu = __builtin_bit_cast(u32x2, to_bf4(y));
                const auto s0 = __builtin_amdgcn_permlane16_swap(xu[0], yu[0], false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(xu[1], yu[1], false, false);
                const u32x4 r = {s0[0], s1[0], s0[1], s1[1]};
                return r;
            };
            const u32x4 rk = pair16(vk0, vk1), rv = pair16(vv0, vv1);
            if (key < p.Sk) {
                const int col = ((g & 1) ? dt1 : dt0) * 16;
                *reinterpret_cast<u32x4*>(DK + col) = rk;
                *reinterpret_cast<u32x4*>(DV + col) = rv;
            }
        });
    });
#ifdef PB_FA1_STAMPS
    const unsigned st_e1 = (unsigned)__builtin_amdgcn_s_memtime();
#endif
    if (p.cs_kv) {
        float* red = reinterpret_cast<float*>(smem);                      // [4 waves][2 HDT]: the ring is free (every wave is past the last barrier's reads)
        __syncthreads();
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
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
            float* row = p.cs_kv + (long)(b * nkb0 + rb) * 2 * d_model + h * HDT;
            const float v = red[t] + red[2 * HDT + t] + red[4 * HDT + t] + red[6 * HDT + t];
            row[t < HDT ? t : d_model + t - HDT] = v;
        }
    }
#ifdef PB_FA1_STAMPS
    if (pin.stamps && lane == 0) {
        const unsigned st_t3 = (unsigned)__builtin_amdgcn_s_memtime();
        const unsigned ntrace = pin.stamps[32];                            // word 32 != 0: TRACE mode -- one record per workgroup behind word 64, no sums (28 atomics per wave on 32 words slow the kernel 4x)
        if (ntrace) {
            if (wave == 0 && blockIdx.x < ntrace) {
                unsigned hw, xcc;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                const unsigned long long st_out = __builtin_amdgcn_s_memrealtime();
                unsigned long long* rec = reinterpret_cast<unsigned long long*>(pin.stamps + 64) + 4 * (size_t)blockIdx.x;
                rec[0] = ((unsigned long long)(xcc & 15) << 32) | hw; rec[1] = st_in; rec[2] = st_out;
                rec[3] = ((unsigned long long)(st_t1 - st_t0) << 32) | (unsigned)(st_t3 - st_t2);          // prologue | epilogue (cycles)
            }
        } else {
        for (int i = 0; i < 10; ++i) atomicAdd(pin.stamps + i, st_acc[i] >> 4);
        atomicAdd(pin.stamps + 16, st_steps);
        atomicAdd(pin.stamps + 10, (st_t1 - st_t0) >> 4); atomicAdd(pin.stamps + 11, (st_t2 - st_t1) >> 4); atomicAdd(pin.stamps + 12, (st_t3 - st_t2) >> 4);
        atomicAdd(pin.stamps + 17, 1u);
        atomicAdd(pin.stamps + 18, (st_p[0] - st_t0) >> 4); atomicAdd(pin.stamps + 19, (st_p[1] - st_p[0]) >> 4); atomicAdd(pin.stamps + 20, (st_p[2] - st_p[1]) >> 4);
        atomicAdd(pin.stamps + 21, (st_p[3] - st_p[2]) >> 4); atomicAdd(pin.stamps + 22, (st_p[4] - st_p[3]) >> 4); atomicAdd(pin.stamps + 23, (st_p[5] - st_p[4]) >> 4);
        atomicAdd(pin.stamps + 24, (st_t1 - st_p[5]) >> 4);
        atomicAdd(pin.stamps + 28, (st_q[0] - st_t0) >> 4); atomicAdd(pin.stamps + 29, (st_q[1] - st_q[0]) >> 4); atomicAdd(pin.stamps + 30, (st_q[2] - st_q[1]) >> 4); atomicAdd(pin.stamps + 31, (st_p[0] - st_q[2]) >> 4);
        atomicAdd(pin.stamps + 25, (st_e0 - st_t2) >> 4); atomicAdd(pin.stamps + 26, (st_e1 - st_e0) >> 4); atomicAdd(pin.stamps + 27, (st_t3 - st_e1) >> 4);
        }
    }
#endif
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

// dq row = sum of the row's valid slabs (key blocks 0 .. n - 1 in order, f32), rounded once; per (batch, 64-row chunk) column sums
// -> partial row of the q-bias gradient. Grid (chunks of 64 rows, B); threads = H * 8 column groups x RL rows in flight, four rows of
// loads requested before the first is summed.
constexpr int RCH = 64;
__global__ __launch_bounds__(1024) void fa1_reduce_kernel(const Fa1Args pin, int RL) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.y, chunk = blockIdx.x;
    Fa64Args p = pin.a;
    varlen_localize(p, b);
    const int d_model = p.H * 64, G = p.H * 8;
    const int nch = (pin.a.Sq + RCH - 1) / RCH;
    const int kvis_end = p.kmax ? min(p.Sk, p.kmax[b]) : p.Sk;
    const int nvalid = (kvis_end + KB1 - 1) / KB1;                           // key blocks that wrote their slab
    const bf16_t* slab = pin.slab + (pin.a.vl_q_off ? (long)pin.a.vl_q_off[b] * d_model : (long)b * pin.slab_sb);
    const int cg = threadIdx.x % G, rl = threadIdx.x / G;
    float cs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) cs[e] = 0.f;
    const int s1 = min(p.Sq, chunk * RCH + RCH);
    auto row = [&](int s) {
        const int nb = p.causal ? min(nvalid, s / KB1 + 1) : nvalid;
        bf16x8 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) if (j < nb) v[j] = *reinterpret_cast<const bf16x8*>(slab + (long)j * pin.slab_stride + (long)s * d_model + cg * 8);
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) if (j < nb) {
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += (float)v[j][e];
        }
        for (int j = 4; j < nb; ++j) {                                    // more than 1024 keys: the further blocks one at a time
            const bf16x8 w = *reinterpret_cast<const bf16x8*>(slab + (long)j * pin.slab_stride + (long)s * d_model + cg * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += (float)w[e];
        }
        bf16x8 r;
#pragma unroll
        for (int e = 0; e < 8; ++e) { r[e] = (bf16_t)acc[e]; cs[e] += (float)r[e]; }
        *reinterpret_cast<bf16x8*>(p.dq + b * p.dq_sb + (long)s * p.dq_ss + cg * 8) = r;
    };
    if (rl < RL) {
        int s = chunk * RCH + rl;
        for (; s + RL < s1; s += 2 * RL) { row(s); row(s + RL); }
        if (s < s1) row(s);
    }
    if (p.cs_q) {
        float* red = reinterpret_cast<float*>(smem);                      // [RL][d_model]
        if (rl < RL)
#pragma unroll
            for (int e = 0; e < 8; ++e) red[rl * d_model + cg * 8 + e] = cs[e];
        __syncthreads();
        for (int col = threadIdx.x; col < d_model; col += blockDim.x) {
            float v = 0.f;
            for (int r = 0; r < RL; ++r) v += red[r * d_model + col];
            p.cs_q[(long)(b * nch + chunk) * d_model + col] = v;
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

// 1 when pb_flash_bwd1 / pb_flash_bwd1_packed can take the shape: head_dim 64, the per-sequence -lse / -delta tables of Sq_max queries fit the
// 160 KiB of LDS beside the tile ring (Sq_max <= 6144) and the bf16 dQ slabs stay under 8 GiB; else 0: the caller keeps the dQ + dK/dV pair.
extern "C" int32_t pb_flash_bwd1_supported(int32_t Sq_max, int32_t Sk_max, int32_t hd, int64_t rows, int32_t H) {
    if (hd != 64 || Sq_max <= 0 || Sk_max <= 0 || rows <= 0 || H <= 0 || H > 128) return 0;
    if ((size_t)OFF_TAB + (size_t)((Sq_max + 63) / 64) * 64 * 8 > 160 * 1024) return 0;
    return pb_flash_bwd1_ws_bytes(rows, H, hd, Sk_max) <= ((int64_t)8 << 30) ? 1 : 0;
}

// One-pass backward, head_dim 64. vl = {q_off, q_len, k_off, k_len} (packed rows) or NULL (dense: batch strides). Same contract as
// pb_flash64_bwd plus the slab workspace `ws` (pb_flash_bwd1_ws_bytes) and q_rows = rows of the q side (packed) / B * Sq (dense).
int pb_flash1_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse, float* delta, const float* key_mask,
                  const int* kmax, void* dq, void* dk, void* dv, int B, int H, int Sq, int Sk, long q_sb, long q_ss, long k_sb, long k_ss, long v_sb,
                  long v_ss, long o_sb, long o_ss, long dq_sb, long dq_ss, long dk_sb, long dk_ss, long dv_sb, long dv_ss, float scale,
                  int causal, float* dbias_q, float* dbias_k, float* dbias_v, float* dbias_ws, void* ws, long q_rows, hipStream_t stream, const int* const* vl,
                  const float* delta_rows) {
    Fa1Args A = {};
    A.delta_rows = delta_rows; A.delta_ld = q_rows;
    Fa64Args& a = A.a;
    if (vl) { a.vl_q_off = vl[0]; a.vl_q_len = vl[1]; a.vl_k_off = vl[2]; a.vl_k_len = vl[3]; a.bh_order = vl[4]; }
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.o = (const bf16_t*)o; a.dout = (const bf16_t*)dout;
    a.dq = (bf16_t*)dq; a.dk = (bf16_t*)dk; a.dv = (bf16_t*)dv; a.lse = const_cast<float*>(lse); a.delta = delta; a.key_mask = key_mask; a.kmax = kmax;
    a.B = B; a.H = H; a.Sq = Sq; a.Sk = Sk; a.q_sb = q_sb; a.q_ss = q_ss; a.k_sb = k_sb; a.k_ss = k_ss; a.v_sb = v_sb; a.v_ss = v_ss;
    a.o_sb = o_sb; a.o_ss = o_ss; a.dq_sb = dq_sb; a.dq_ss = dq_ss; a.dk_sb = dk_sb; a.dk_ss = dk_ss; a.dv_sb = dv_sb; a.dv_ss = dv_ss;
    a.scale = scale; a.causal = causal;
    a.zeros = fa1_zero_page();
    PB_REQUIRE(a.zeros != nullptr, "pb_flash_bwd1: cannot allocate the zero page");
    PB_REQUIRE(ws != nullptr, "pb_flash_bwd1: the dQ slab workspace is required (pb_flash_bwd1_ws_bytes)");
    const int d_model = H * 64, nkb = (Sk + KB1 - 1) / KB1, nqb = (Sq + RCH - 1) / RCH;
    A.slab = (bf16_t*)ws; A.slab_stride = q_rows * d_model; A.slab_sb = (long)Sq * d_model;
#ifdef PB_FA1_STAMPS
    if (const char* e = getenv("PB_FA1_STAMP_PTR")) A.stamps = (unsigned*)strtoull(e, nullptr, 0);
#endif
    if (dbias_q) {
        PB_REQUIRE(dbias_k && dbias_v && dbias_ws, "pb_flash_bwd1: dbias_q/k/v and dbias_ws go together");
        // partial rows: 2 per (batch, 256-key block) + 1 per (batch, 64-query chunk) <= what pb_flash_bias_ws_floats provides
        const size_t n_kv = (size_t)B * nkb * 2 * d_model;
        if (float* slice = pb_defer_alloc(n_kv + (size_t)B * nqb * d_model)) dbias_ws = slice;
        a.cs_kv = dbias_ws; a.cs_q = dbias_ws + n_kv;
    }
    const size_t lds = (size_t)OFF_TAB + (size_t)((Sq + 63) / 64) * 64 * 8;
    PB_REQUIRE(lds <= 160 * 1024, "pb_flash_bwd1: Sq=%d needs %zu bytes of LDS", Sq, lds);
    // every launch, like the other kernels: the attribute is per device, and a process may drive several
    PB_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fa1_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int rpb = 64;
    if (!delta_rows) {                                                     // the caller's GEMM epilogue has not made the row sums: one pass over dO and O
        hipLaunchKernelGGL(fa1_delta_kernel, dim3((Sq + rpb - 1) / rpb, B), dim3(256), 0, stream, a, rpb);
        PB_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(fa1_bwd_kernel, dim3(nkb * H * B), dim3(FT), lds, stream, A);
    PB_LAUNCH_CHECK();
    const int G = H * 8;
    PB_REQUIRE(G <= 1024, "pb_flash_bwd1: more than 128 heads");
    const int nrl = std::min(8, 1024 / G);
    hipLaunchKernelGGL(fa1_reduce_kernel, dim3(nqb, B), dim3(G * nrl), dbias_q ? (size_t)nrl * d_model * 4 : 0, stream, A, nrl);
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
                             void* dq_ws, const float* delta_rows, void* stream_) {
    PB_REQUIRE(hd == 64, "pb_flash_bwd1: head_dim %d (64 only)", hd);
    PB_REQUIRE(q_ss % 8 == 0 && k_ss % 8 == 0 && v_ss % 8 == 0 && o_ss % 8 == 0 && dq_ss % 8 == 0 && dk_ss % 8 == 0 && dv_ss % 8 == 0 &&
               q_sb % 8 == 0 && k_sb % 8 == 0 && v_sb % 8 == 0 && o_sb % 8 == 0 && dq_sb % 8 == 0, "pb_flash_bwd1: strides must be multiples of 8 elements");
    if (B <= 0 || H <= 0 || Sq <= 0 || Sk <= 0) return 0;
    return pb_flash1_bwd(q, k, v, o, dout, lse, delta, key_mask, kmax, dq, dk, dv, B, H, Sq, Sk, q_sb, q_ss, k_sb, k_ss, v_sb, v_ss, o_sb, o_ss,
                         dq_sb, dq_ss, dk_sb, dk_ss, dv_sb, dv_ss, scale, causal & 1, dbias_q, dbias_k, dbias_v, dbias_ws, dq_ws, (long)B * Sq,
                         (hipStream_t)stream_, nullptr, delta_rows);
}

extern "C" int pb_flash_bwd1_packed(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse, void* dq,
                                    void* dk, void* dv, float* delta, const int32_t* q_off, const int32_t* q_len, const int32_t* k_off,
                                    const int32_t* k_len, const int32_t* k_vis, int32_t B, int32_t H, int32_t Sq_max, int32_t Sk_max, int32_t hd,
                                    int64_t q_ss, int64_t k_ss, int64_t v_ss, int64_t o_ss, int64_t dq_ss, int64_t dk_ss, int64_t dv_ss,
                                    float scale, int32_t causal, float* dbias_q, float* dbias_k, float* dbias_v, float* dbias_ws,
                                    void* dq_ws, int64_t q_rows, const int32_t* bh_order, const float* delta_rows, void* stream_) {
    PB_REQUIRE(hd == 64, "pb_flash_bwd1_packed: head_dim %d (64 only)", hd);
    PB_REQUIRE(q_ss % 8 == 0 && k_ss % 8 == 0 && v_ss % 8 == 0 && o_ss % 8 == 0 && dq_ss % 8 == 0 && dk_ss % 8 == 0 && dv_ss % 8 == 0,
               "pb_flash_bwd1_packed: strides must be multiples of 8 elements");
    PB_REQUIRE(q_off && q_len && k_off && k_len && k_vis, "pb_flash_bwd1_packed: the five row descriptors are required");
    if (B <= 0 || H <= 0 || Sq_max <= 0 || Sk_max <= 0) return 0;
    const int* vl[5] = {q_off, q_len, k_off, k_len, bh_order};
    return pb_flash1_bwd(q, k, v, o, dout, lse, delta, nullptr, k_vis, dq, dk, dv, B, H, Sq_max, Sk_max, 0, q_ss, 0, k_ss, 0, v_ss, 0, o_ss,
                         0, dq_ss, 0, dk_ss, 0, dv_ss, scale, causal & 1, dbias_q, dbias_k, dbias_v, dbias_ws, dq_ws, q_rows, (hipStream_t)stream_, vl, delta_rows);
}
