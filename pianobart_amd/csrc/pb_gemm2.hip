// bf16 GEMM, second generation (gfx950): direct-to-LDS staging + transposed LDS reads + split-K.
//
//   C[m][n] (+)= epilogue( alpha * sum_k A(m,k) * B(n,k) )            bf16 in, f32 accumulate
//
// Differences from pb_gemm.hip (which stays as the exact-f32 path and the ragged/unaligned fallback):
//   * both operands go global -> LDS with global_load_lds_dwordx4 (no VGPR staging, no ds_write): the LDS
//     image is lane-linear per wave-instruction, so the bank swizzle is applied to the per-lane SOURCE
//     address and again on the read (cdna_hip_programming.md 5.4 rule 21);
//   * an operand that is NOT K-contiguous in memory (dgrad's W[K][N], wgrad's dY[T][M] and X[T][N]) is
//     staged in its natural [k][r] layout (256-B rows) and its MFMA fragments are fetched with
//     ds_read_b64_tr_b16 (hardware transpose), two reads per 8-k fragment, bank-conflict free;
//   * split-K over blockIdx.z into f32 slabs (wgrad has K = B*S tokens but only 36..144 output tiles),
//     summed by pb_reduce_slabs: deterministic, no atomics;
//   * the epilogue stays in registers: operand-swapped MFMAs + v_permlane16_swap give every lane 8 consecutive
//     output columns, stored as 16-byte row segments (epilogue_regs).
// Two kernels:
//   gemm3_kernel  256x256x64 tile, 8 waves, persistent, ping-pong schedule with the DMA prefetch in flight across barriers
//                 (every output at least 512 wide, and the split-K wgrads);
//   gemm2_kernel  128x128x64 tile, 4 waves (2x2 of 64x64), 2 LDS stages, one barrier per K tile, 2 workgroups per CU
//                 (narrow outputs, the 768x768 split-K wgrads; also the 256x256 one-barrier form kept for A/B runs).
// Requirements (else pb_gemm falls back to pb_gemm.hip): K % 64 == 0 per split, 16-byte aligned rows,
// contiguous dims multiples of 8.
#include "pb_common.h"
#include "pb_api_internal.h"
#include <algorithm>
#include <cstdlib>

namespace {

constexpr int BK = 64;
typedef __attribute__((ext_vector_type(4))) short s16x4;

struct Gemm2Args {
    const bf16_t* A; const bf16_t* B; void* C;
    const float* bias; const bf16_t* aux_in; bf16_t* aux_out;
    int M, N, K, Kc;                      // Kc = K range per split (multiple of 64)
    long lda, ldb, ldc, ldaux;
    int nb2; long sA1, sA2, sB1, sB2, sC1, sC2, sCz;
    float alpha; int flags; int tiles_m, tiles_n, nsplit;
    float* cs_ws;                         // column-sum partials (2 tiles_m rows of N floats) or NULL
    float* rowdot; long ld_rowdot;        // PB_GEMM_ROWDOT: rowdot[(n / 64) * ld_rowdot + m] = sum over the 64-column group of C[m][.] * aux_in[m][.]
    // tail split (gemm3_kernel, nsplit == 1): work items 0 .. n_full - 1 are whole tiles; the tiles of the last, partly filled
    // round of the persistent grid are cut into tail_split K ranges each, one work item per range, whose f32 partial tiles go
    // to tail_slabs (item-major, 256 x BN floats each) and are summed, finished and stored by tail_finish_kernel.
    int n_full, tail_split, tail_kc;
    float* tail_slabs;
#ifdef PB_G3_STAMPS
    unsigned long long* stamps;           // diagnostic build only (1024 x 8 x 16 u32 slots): per-wave cycle sums of the persistent loop's sections (tools/gemm_stamps.py)
#endif
};

__device__ __forceinline__ int kswz(int row) { return ((row >> 1) & 7) ^ ((row >> 4) & 7); }
__device__ __forceinline__ int rswz(int krow) { return ((krow & 3) | ((krow >> 1) & 4)) << 1; }

__device__ __forceinline__ void glds16(const bf16_t* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// The same DMA with the address split into a wave-uniform 64-bit base in SGPRs and one 32-bit lane offset (global_load_lds_dwordx4 voff, s[base]): the
// builtin above takes a 64-bit pointer per lane, which in a K loop that advances eight pieces per K-tile keeps two or three 64-bit lane terms per piece
// in registers (the ping-pong kernel's K-contiguous loop held 72 VGPRs besides accumulators and fragments, most of them these). M0 = the LDS address of
// the piece; it is compiler-reserved, so it is saved and restored around the instruction (cdna_hip_programming.md 5.7).
__device__ __forceinline__ void glds16_s(unsigned voff, const char* sbase, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// K-contiguous operand tile: rows r0..r0+RT-1 (clamped to R-1), k0..k0+63. Image [RT][128 B], chunk-swizzled.
// RT*128 bytes = RT/8 DMA pieces of 1 KiB, dealt round-robin to the NW waves.
template <int RT, int NW>
__device__ __forceinline__ void stage_kc(const bf16_t* __restrict__ base, long ld, int r0, int k0, int R, char* lds, int wave, int lane) {
    constexpr int PPW = (RT / 8) / NW;
#pragma unroll
    for (int n = 0; n < PPW; ++n) {
        const int inst = wave * PPW + n;
        const int row = inst * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ kswz(row);
        const int gr = min(r0 + row, R - 1);
        glds16(base + (long)gr * ld + k0 + chunk * 8, lds + inst * 1024);
    }
}
// Row-contiguous operand tile ([K][R] in memory): k rows k0..k0+63, columns r0..r0+RT-1 (clamped). Image [64][RT*2 B].
template <int RT, int NW>
__device__ __forceinline__ void stage_rc(const bf16_t* __restrict__ base, long ld, int r0, int k0, int R, char* lds, int wave, int lane) {
    constexpr int PPW = (RT / 8) / NW;            // pieces per wave (64 * RT * 2 / 1024 = RT / 8 pieces)
    constexpr int RPP = 512 / RT;                 // k-rows per piece
    constexpr int LPR = 64 / RPP;                 // lanes (16-B chunks) per row
#pragma unroll
    for (int n = 0; n < PPW; ++n) {
        const int inst = wave * PPW + n;
        const int krow = inst * RPP + lane / LPR;
        const int chunk = (lane % LPR) ^ rswz(krow);
        const int gc = min(r0 + chunk * 8, R - 8);
        glds16(base + (long)(k0 + krow) * ld + gc, lds + inst * 1024);
    }
}
template <bool KC, int RT, int NW>
__device__ __forceinline__ void stage(const bf16_t* base, long ld, int r0, int k0, int R, char* lds, int wave, int lane) {
    if constexpr (KC) stage_kc<RT, NW>(base, ld, r0, k0, R, lds, wave, lane);
    else stage_rc<RT, NW>(base, ld, r0, k0, R, lds, wave, lane);
}

// MFMA fragment (8 k-values 32 ks + 8 g + j of row `row16 + lane&15`).
template <bool KC, int RT>
__device__ __forceinline__ bf16x8 frag(const char* lds, int rbase /*multiple of 16*/, int ks, int lane) {
    const int lr = lane & 15, g = lane >> 4;
    if constexpr (KC) {
        const int row = rbase + lr;
        return *reinterpret_cast<const bf16x8*>(lds + row * 128 + (((ks * 4 + g) ^ kswz(row)) << 4));
    } else {
        // two transposed 4(k) x 16(r) block reads: lane supplies row kb + q, columns rbase + 4p .. +3
        const int q = lr >> 2, pp = lr & 3;
        const int kb = ks * 32 + g * 8;
        const int chunk = (rbase >> 3) + (pp >> 1);
        const int off0 = (kb + q) * (RT * 2) + ((chunk ^ rswz(kb + q)) << 4) + ((pp & 1) << 3);
        const int off1 = (kb + 4 + q) * (RT * 2) + ((chunk ^ rswz(kb + 4 + q)) << 4) + ((pp & 1) << 3);
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + off0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + off1));
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    }
}

__device__ __forceinline__ void store8(bf16_t* p, f32x4 a, f32x4 b) {
    bf16x8 r = {(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3], (bf16_t)b[0], (bf16_t)b[1], (bf16_t)b[2], (bf16_t)b[3]};
    *reinterpret_cast<bf16x8*>(p) = r;
}
// streaming variants (nt): written / read exactly once before the other pass of the step, so they should not push the operands
// the NEXT kernel re-reads (g for fc2, dU for the fc1 weight / input gradients) out of L2 and the Infinity Cache
__device__ __forceinline__ void store8_nt(bf16_t* p, f32x4 a, f32x4 b) {
    bf16x8 r = {(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3], (bf16_t)b[0], (bf16_t)b[1], (bf16_t)b[2], (bf16_t)b[3]};
    __builtin_nontemporal_store(r, reinterpret_cast<bf16x8*>(p));
}
__device__ __forceinline__ void load8_nt(const bf16_t* p, f32x4& a, f32x4& b) {
    const bf16x8 v = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p));
    a = f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    b = f32x4{(float)v[4], (float)v[5], (float)v[6], (float)v[7]};
}
__device__ __forceinline__ void load8(const bf16_t* p, f32x4& a, f32x4& b) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
    a = f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    b = f32x4{(float)v[4], (float)v[5], (float)v[6], (float)v[7]};
}

// 1-D grid over (split, tile) -> (m0, n0, split).
template <int BM, int BN>
__device__ __forceinline__ void block_tile(const Gemm2Args& p, int L, int& m0, int& n0, int& zs) {
    const int ntiles = p.tiles_m * p.tiles_n;
    // 1-D grid over (split, tile). XCD x (= id % 8) takes a CONTIGUOUS chunk of the (split-major, tile-minor) list, so with
    // split-K the workgroups of one XCD work on (nearly) one K slice: its A/B rows are fetched into that L2 once and
    // shared by all its tiles (wgrad measured ~2x its algorithmic bytes from beyond L2 with the tile-only remap).
    int bid;
    {
        const int total = ntiles * p.nsplit;
        const int q = total >> 3, r = total & 7, x = L & 7, idx = L >> 3;
        const int lin = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + idx;
        zs = lin / ntiles; bid = lin - zs * ntiles;
    }
    // grouped tile order inside an XCD's chunk: super-rows of GM M-tiles x all N-tiles, M fastest. The ~64 workgroups an
    // XCD runs at once then cover ~8 x 8 tiles: 8 A row-panels + 8 B panels are fetched into its L2 and shared, instead
    // of 64 A panels + 1 B panel (measured: FETCH_SIZE 1.2 GB -> per launch at the fc1 shape with the M-fastest order,
    // i.e. the whole 50 MB A matrix re-read from beyond L2 for every one of the 24 N-tiles).
    constexpr int GM = 8;
    const int gsz = GM * p.tiles_n, grp = bid / gsz, first_m = grp * GM;
    const int gm = min(GM, p.tiles_m - first_m), rem = bid - grp * gsz;
    const int tm = first_m + rem % gm, tn = rem / gm;
    m0 = tm * BM; n0 = tn * BN;
}

// Register epilogue. Every kernel here issues its MFMAs operand-swapped (mfma(b, a)), so the lane (lr = lane & 15,
// lg = lane >> 4) holds C[row i*16 + lr][columns j*16 + lg*4 .. +3]: four CONSECUTIVE columns. One v_permlane16_swap per
// dword then hands the two column tiles (j, j+1) of a pair to the even / odd 16-lane rows: every lane owns 8 consecutive
// columns of one output row, i.e. one 16-byte bf16 store (two for f32), 64 (128) contiguous bytes per row and
// instruction, with bias / GELU / GELU-grad / accumulate applied in f32 in between. No LDS and no barrier: the f32 LDS
// panel this replaces cost about a third of the store tail of a 256 x 256 tile, and its 8-byte stores another 6 %.
template <int TM, int TNW = 4>
__device__ __forceinline__ void epilogue_regs(const Gemm2Args& p, f32x4 (&acc)[TM][TNW], int mw /*wave's first row*/, int nw /*first column*/, long coff, int lane,
                                              const float* lds_bias = nullptr /*bias[nw ..] staged in LDS by the caller*/,
                                              float* cs_row = nullptr /*this wave row's column-sum partials: N floats*/) {
    asm volatile("" : "+v"(lane));        // opaque (see epilogue_pf): the K loop needs every register
    const int lr = lane & 15, lg = lane >> 4;
    const bool accum = p.flags & PB_GEMM_ACCUM, c32 = p.flags & PB_GEMM_C_F32;
    const bool do_gelu = p.flags & PB_GEMM_GELU, mul_gg = p.flags & PB_GEMM_MUL_GELU_GRAD;
    float* C32 = reinterpret_cast<float*>(p.C) + coff;
    bf16_t* CT = reinterpret_cast<bf16_t*>(p.C) + coff;
    // even 16-lane rows end up with tile j of the pair, columns lg*4 .. +7; odd rows with tile j+1, columns (lg-1)*4 .. +7
    const int cb = (lg & 1) ? 16 + (lg - 1) * 4 : lg * 4;
    f32x4 bv[2][2];
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
        const int c = nw + jp * 32 + cb;
        const bool has = p.bias && c < p.N;
        if (lds_bias) {
            bv[jp][0] = p.bias ? *reinterpret_cast<const f32x4*>(lds_bias + jp * 32 + cb) : f32x4{0.f, 0.f, 0.f, 0.f};
            bv[jp][1] = p.bias ? *reinterpret_cast<const f32x4*>(lds_bias + jp * 32 + cb + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
            bv[jp][0] = has ? *reinterpret_cast<const f32x4*>(p.bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
            bv[jp][1] = has ? *reinterpret_cast<const f32x4*>(p.bias + c + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    f32x4 cs[2][2] = {{f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}};
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = mw + i * 16 + lr;
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
            f32x4 v0, v1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // TNW = 3: the last column tile has no partner; its even 16-lane rows still end up with columns lg*4 .. +7
                const float xa = acc[i][2 * jp][r], xb = (2 * jp + 1 < TNW) ? acc[i][(2 * jp + 1 < TNW) ? 2 * jp + 1 : 0][r] : 0.f;     // (a bit_cast applied directly to a vector element reads element 0)
                auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(xa), __float_as_uint(xb), false, false);
                v0[r] = __uint_as_float(sw[0]);
                v1[r] = __uint_as_float(sw[1]);
            }
            const int col = nw + jp * 32 + cb;
            const bool lane_has = (2 * jp + 1 < TNW) || !(lg & 1);              // odd rows of an unpaired tile hold nothing
            if (lane_has && row < p.M && col < p.N) {
                v0 = v0 * p.alpha + bv[jp][0];
                v1 = v1 * p.alpha + bv[jp][1];
                if (do_gelu) {                                           // C = gelu(v), aux_out = gelu'(v): both from one exp + one rcp
                    f32x4 d0, d1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float y, dy;
                        gelu_both_fast(v0[e], y, dy); v0[e] = y; d0[e] = dy;
                        gelu_both_fast(v1[e], y, dy); v1[e] = y; d1[e] = dy;
                    }
                    store8_nt(p.aux_out + (long)row * p.ldaux + col, d0, d1);
                }
                if (mul_gg) {                                            // aux_in holds gelu'(pre-activation) from the forward
                    f32x4 u0, u1;
                    load8_nt(p.aux_in + (long)row * p.ldaux + col, u0, u1);
                    v0 *= u0; v1 *= u1;
                }
                const long ci = (long)row * p.ldc + col;
                if (c32) {
                    if (accum) { v0 += *reinterpret_cast<const f32x4*>(C32 + ci); v1 += *reinterpret_cast<const f32x4*>(C32 + ci + 4); }
                    *reinterpret_cast<f32x4*>(C32 + ci) = v0;
                    *reinterpret_cast<f32x4*>(C32 + ci + 4) = v1;
                } else {
                    if (accum) { f32x4 c0, c1; load8(CT + ci, c0, c1); v0 += c0; v1 += c1; }
                    store8(CT + ci, v0, v1);
                }
                if (cs_row) { cs[jp][0] += v0; cs[jp][1] += v1; }
            }
        }
    }
    if (cs_row) {
        // column sums of the wave's TM x 16 rows: the 16 lanes of a DPP row hold 16 different rows of the same 8 columns ->
        // mirror / half-mirror / quad permutes add them up inside the row; lane lr = 0 of each row stores its 2 x 8 sums
#pragma unroll
        for (int jp = 0; jp < 2; ++jp)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = cs[jp][hh][e];
                    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));   // row_mirror
                    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));   // row_half_mirror
                    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4e, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
                    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xb1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
                    cs[jp][hh][e] = v;
                }
        if (lr == 0) {
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                const int col = nw + jp * 32 + cb;
                if (col < p.N && ((2 * jp + 1 < TNW) || !(lg & 1))) {
                    *reinterpret_cast<f32x4*>(cs_row + col) = cs[jp][0];
                    *reinterpret_cast<f32x4*>(cs_row + col + 4) = cs[jp][1];
                }
            }
        }
    }
}

// Read-modify-write epilogues of the 256 x 256 ping-pong kernel (round 5): dU = (dY W2) * gelu'(U) reads a bf16 operand tile, the residual
// accumulation C += (dY W) reads C itself. Left to hipcc, every 16-byte read of epilogue_regs is followed by s_waitcnt vmcnt(0): 16 dependent
// memory round trips per wave and tile, each of which also drains the stores in front of it (tools/gemm_stamps.py: the epilogue of the dfc2
// tile 18.0 us against 5.1 us for a store-only one, 9.5 us for +=). Here the reads are inline-asm loads the compiler does not track, PFD of
// them in flight ahead of their use (into the A / B fragment registers, dead by now), the stores are asm too, and every wait is a COUNTED
// vmcnt: loads, stores and the next item's DMA pieces retire in issue order, so "all but the N youngest" names exactly the load a chunk
// needs. Interior tiles only (no bounds), bf16 C; everything else keeps epilogue_regs.
//   MODE 1: C = (alpha acc + bias) * aux_in (+ column sums of the stored values)      MODE 2: C += alpha acc + bias
//   MODE 3: C = alpha acc + bias, and rowdot[column group of 64][row] = sum over the group of bf16(C) * aux_in: the delta = rowsum(dO * O) per
//           head of the attention backward, taken where dO is made (the out-projection's input gradient) instead of by a pass over dO and O
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
// Row staging of the asm-store epilogues (round 5, late): the MFMA registers leave a lane with 8 consecutive columns of ONE row, and the four lanes that
// complete 64 bytes of that row are 16 lane numbers apart -- the memory pipeline merges ADJACENT lanes only, so such a store instruction is 64 separate
// 16-byte writes and a tile's 128 KiB take a CU 5.3 us whatever the rest of the chip does (tools/probes/store_tile.hip; the same bytes with 8 adjacent
// lanes per 128-byte row: 2.3 us). Each 16-row group therefore takes a trip through a wave-private 2 KiB LDS image [16 rows][128 B] (written as the
// registers lie, 16-byte slot XOR row & 7; read back 8 lanes per row; no barrier: a wave's LDS operations execute in order) and leaves as two stores of
// 8 rows x 128 contiguous bytes.
constexpr int G3_STAGE_OFF = 131072 + 2048, G3_STAGE_BYTES = 8 * 2048;
struct RowStage {
    unsigned wr0, wr1, rd;          // LDS byte addresses: this lane's two chunks (columns cb .. +7 and 32 + cb .. +7 of row lr); its read slot (row lane >> 3, chunk lane & 7; + 1024: row + 8)
    int row, chunk;                 // of the read side
};
__device__ __forceinline__ RowStage row_stage(unsigned lds_base /* of the kernel's dynamic LDS */, int wave, int lane) {
    const int lr = lane & 15, lg = lane >> 4, q = ((lg & 1) << 1) | (lg >> 1);            // q = cb / 8: lg 0, 2, 1, 3 -> chunk 0, 1, 2, 3 of the 32-column half
    const unsigned base = lds_base + (unsigned)(G3_STAGE_OFF + wave * 2048);
    RowStage r;
    r.wr0 = base + (unsigned)(lr * 128 + ((q ^ (lr & 7)) << 4));
    r.wr1 = base + (unsigned)(lr * 128 + (((4 + q) ^ (lr & 7)) << 4));
    r.row = lane >> 3; r.chunk = lane & 7;
    r.rd = base + (unsigned)(r.row * 128 + ((r.chunk ^ r.row) << 4));
    return r;
}
__device__ __forceinline__ void row_stage_put(const RowStage& r, u32x4 c0, u32x4 c1) {
    *reinterpret_cast<__attribute__((address_space(3))) u32x4*>(r.wr0) = c0;
    *reinterpret_cast<__attribute__((address_space(3))) u32x4*>(r.wr1) = c1;
}
__device__ __forceinline__ void row_stage_get(const RowStage& r, u32x4& lo, u32x4& hi) {     // rows 0 .. 7 / 8 .. 15 of the group
    lo = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(r.rd);
    hi = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(r.rd + 1024);
}
// the other direction, for an operand tile that was LOADED 8 rows x 128 bytes per instruction: in as it was loaded, out as the registers want it
__device__ __forceinline__ void row_stage_put_rows(const RowStage& r, u32x4 lo, u32x4 hi) {
    *reinterpret_cast<__attribute__((address_space(3))) u32x4*>(r.rd) = lo;
    *reinterpret_cast<__attribute__((address_space(3))) u32x4*>(r.rd + 1024) = hi;
}
__device__ __forceinline__ void row_stage_get_chunks(const RowStage& r, u32x4& c0, u32x4& c1) {
    c0 = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(r.wr0);
    c1 = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(r.wr1);
}
constexpr int PFD = 6;                            // 8 would spill 2 VGPRs next to the column-sum accumulators (a scratch access is a vmcnt operation too)
// address = wave-uniform 64-bit base in SGPRs + one 32-bit lane offset + an immediate: no 64-bit vector arithmetic per chunk
template <int IMM> __device__ __forceinline__ void pf_load(u32x4& d, unsigned voff, const char* sbase, bool nt) {
    if (nt) asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3 nt" : "=v"(d) : "v"(voff), "s"(sbase), "n"(IMM) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(d) : "v"(voff), "s"(sbase), "n"(IMM) : "memory");
}
template <int IMM> __device__ __forceinline__ void pf_store(unsigned voff, const char* sbase, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3\n\ts_nop 1" :: "v"(voff), "v"(v), "s"(sbase), "n"(IMM) : "memory");   // s_nop: a vector write of the data registers needs 2 wait states behind a 128-bit store (hipcc pads only its own)
}
__device__ __forceinline__ const char* sgpr_ptr(const void* q) {
    const unsigned long long a = (unsigned long long)q;
    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    // an SGPR fresh from v_readfirstlane needs 5 wait states before a global_* instruction reads it as its base; hipcc pads that for
    // its own instructions only, not for the ones inside an asm string (cdna_hip_programming.md 5.7 item 2): without the nops the first
    // load of the epilogue went to a stale address (memory access fault)
    asm volatile("s_nop 4" : "+s"(lo), "+s"(hi));
    return reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
}
// Since the row staging both the operand tile and the result move 8 rows x 128 contiguous bytes per instruction (adjacent lanes: see row_stage above): a 16-row
// group = two loads (rows 0 .. 7 | 8 .. 15) that go through the wave's LDS image INTO the register layout, and two stores that go through it OUT of it.
// Schedule: L(0) .. L(2 PFG - 1) | for every group g: { wait for its two loads; compute; S(2g), S(2g + 1) (+ the row sums of MODE 3); L(2 (g + PFG)), L(.. + 1) }.
constexpr int PFG = PFD / 2;                      // groups of loads in flight
constexpr int pfg_younger(int g, int mode) {      // operations issued after the second load of group g when the group is about to consume them
    const int S = 2 + (mode == 3 ? 1 : 0);
    int n = 0;
    if (g < PFG) { n += 2 * (PFG - 1 - g); for (int k = 0; k < g; ++k) n += S + (k + PFG < 8 ? 2 : 0); }
    else for (int k = g - PFG + 1; k < g; ++k) n += S + (k + PFG < 8 ? 2 : 0);
    return n;
}
template <int N> __device__ __forceinline__ void pf_wait2(u32x4& d0, u32x4& d1) { asm volatile("s_waitcnt vmcnt(%2)" : "+v"(d0), "+v"(d1) : "n"(N) : "memory"); }
__device__ __forceinline__ f32x4 bf_lo4(const u32x4& w) { return f32x4{__uint_as_float(w[0] << 16), __uint_as_float(w[0] & 0xffff0000u), __uint_as_float(w[1] << 16), __uint_as_float(w[1] & 0xffff0000u)}; }
__device__ __forceinline__ f32x4 bf_hi4(const u32x4& w) { return f32x4{__uint_as_float(w[2] << 16), __uint_as_float(w[2] & 0xffff0000u), __uint_as_float(w[3] << 16), __uint_as_float(w[3] & 0xffff0000u)}; }
template <int MODE, bool CS>
__device__ __forceinline__ void epilogue_pf(const Gemm2Args& p, f32x4 (&acc)[8][4], int mw, int nw, long coff, int lane, const float* lds_bias, float* cs_row, unsigned lds_base, int wave) {
    asm volatile("" : "+v"(lane));        // opaque: what the epilogue derives from the lane number is computed HERE, not hoisted in front of the K loop and kept in registers across it
    const int lr = lane & 15, lg = lane >> 4;
    const int cb = (lg & 1) ? 16 + (lg - 1) * 4 : lg * 4;
    const RowStage rs = row_stage(lds_base, wave, lane);
    // memory side: lane = (row lane >> 3 of an 8-row half group, 16-byte chunk lane & 7 of the wave's 64 columns)
    const char* cbase = sgpr_ptr(reinterpret_cast<bf16_t*>(p.C) + coff + (long)mw * p.ldc + nw);
    const char* sbase = MODE != 2 ? sgpr_ptr(p.aux_in + (long)mw * p.ldaux + nw) : cbase;
    const unsigned cvoff = (unsigned)(rs.row * (int)p.ldc + rs.chunk * 8) * 2u, svoff = MODE != 2 ? (unsigned)(rs.row * (int)p.ldaux + rs.chunk * 8) * 2u : cvoff;
    const long sstep = 16 * (MODE != 2 ? p.ldaux : p.ldc), cstep = 16 * p.ldc;          // bytes per 8 rows
    f32x4 bv[2][2];
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
        bv[jp][0] = p.bias ? *reinterpret_cast<const f32x4*>(lds_bias + jp * 32 + cb) : f32x4{0.f, 0.f, 0.f, 0.f};
        bv[jp][1] = p.bias ? *reinterpret_cast<const f32x4*>(lds_bias + jp * 32 + cb + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    u32x4 pf[PFD];
    const char* rbase = MODE == 3 ? sgpr_ptr(p.rowdot + (long)(nw >> 6) * p.ld_rowdot + mw) : nullptr;      // + 16 rows = 64 bytes per row group
    const unsigned rvoff = (unsigned)lr * 4u;
    static_for<0, PFD>([&](auto cc) { constexpr int c = decltype(cc)::value; pf_load<0>(pf[c], svoff, sbase + c * sstep, MODE != 2); });
    f32x4 cs[2][2] = {{f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}};
    static_for<0, 8>([&](auto gg) {
        constexpr int g = decltype(gg)::value, s0 = (2 * g) % PFD, s1 = (2 * g + 1) % PFD;
        pf_wait2<pfg_younger(g, MODE)>(pf[s0], pf[s1]);
        row_stage_put_rows(rs, pf[s0], pf[s1]);
        u32x4 w[2];
        row_stage_get_chunks(rs, w[0], w[1]);
        u32x4 out[2];
        float rd = 0.f;
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
            const f32x4 u0 = bf_lo4(w[jp]), u1 = bf_hi4(w[jp]);
            f32x4 v0, v1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float xa = acc[g][2 * jp][r], xb = acc[g][2 * jp + 1][r];
                auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(xa), __float_as_uint(xb), false, false);
                v0[r] = __uint_as_float(sw[0]);
                v1[r] = __uint_as_float(sw[1]);
            }
            v0 = v0 * p.alpha + bv[jp][0];
            v1 = v1 * p.alpha + bv[jp][1];
            if constexpr (MODE == 1) { v0 *= u0; v1 *= u1; } else if constexpr (MODE == 2) { v0 += u0; v1 += u1; }
            const bf16x8 r = {(bf16_t)v0[0], (bf16_t)v0[1], (bf16_t)v0[2], (bf16_t)v0[3], (bf16_t)v1[0], (bf16_t)v1[1], (bf16_t)v1[2], (bf16_t)v1[3]};
            out[jp] = __builtin_bit_cast(u32x4, r);
            if constexpr (MODE == 3) {
                // the products use the ROUNDED values, as a pass over the stored tensor would; a row's 64 columns lie in the two chunks of 4 lanes
                float s8 = 0.f;                                                  // (per chunk from zero, then added: the association of the per-chunk form, bit for bit)
#pragma unroll
                for (int e = 0; e < 4; ++e) { s8 = fmaf((float)r[e], u0[e], s8); s8 = fmaf((float)r[4 + e], u1[e], s8); }
                rd = jp == 0 ? s8 : rd + s8;
            }
            if constexpr (CS) { cs[jp][0] += v0; cs[jp][1] += v1; }
        }
        u32x4 lo, hi;
        row_stage_put(rs, out[0], out[1]);
        row_stage_get(rs, lo, hi);
        pf_store<0>(cvoff, cbase + (2 * g) * cstep, lo);
        pf_store<0>(cvoff, cbase + (2 * g + 1) * cstep, hi);
        if constexpr (MODE == 3) {
            float s = rd;
            auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(s), __float_as_uint(s), false, false);
            s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
            auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
            s = __uint_as_float(b[0]) + __uint_as_float(b[1]);
            asm volatile("global_store_dword %0, %1, %2" :: "v"(rvoff), "v"(s), "s"(rbase + g * 64) : "memory");    // every lane of a row holds the sum: 4 lanes write the same word
        }
        if constexpr (g + PFG < 8) {
            pf_load<0>(pf[s0], svoff, sbase + (2 * (g + PFG)) * sstep, MODE != 2);
            pf_load<0>(pf[s1], svoff, sbase + (2 * (g + PFG) + 1) * sstep, MODE != 2);
        }
    });
    if constexpr (CS) {
#pragma unroll
        for (int jp = 0; jp < 2; ++jp)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = cs[jp][hh][e];
                    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));   // row_mirror
                    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));   // row_half_mirror
                    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4e, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
                    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xb1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
                    cs[jp][hh][e] = v;
                }
        if (lr == 0) {
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                *reinterpret_cast<f32x4*>(cs_row + nw + jp * 32 + cb) = cs[jp][0];
                *reinterpret_cast<f32x4*>(cs_row + nw + jp * 32 + cb + 4) = cs[jp][1];
            }
        }
    }
}

// The forward fc1 epilogue of an interior tile: C = gelu(alpha acc + bias), aux_out = gelu'(...), both bf16. The tile's 2 x 128 KiB of stores are not what
// it waits for: 128 values per lane x (exp + rcp + ~15 VALU operations) with two waves sharing each SIMD made 12.7 us of a 33 us item (profiles/
// r05_gemm_item_anatomy.txt). Here the arithmetic runs two values per instruction (gelu_pair2: v_pk_mul / v_pk_fma, 5.5 cycles per wave-instruction
// against 4.1 for the scalar ones, tools/probes/valu_rate.hip), the sign select is a v_bfi, and the addresses are an SGPR base + one lane offset per
// operand (epilogue_regs: a 64-bit multiply-add per row and store). Exactly 32 stores per wave (the caller's `pend`).
template <int IMM> __device__ __forceinline__ void pf_store_nt(unsigned voff, const char* sbase, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3 nt\n\ts_nop 1" :: "v"(voff), "v"(v), "s"(sbase), "n"(IMM) : "memory");
}
__device__ __forceinline__ void epilogue_gelu(const Gemm2Args& p, f32x4 (&acc)[8][4], int mw, int nw, long coff, int lane, const float* lds_bias, unsigned lds_base, int wave) {
    asm volatile("" : "+v"(lane));
    const int lr = lane & 15, lg = lane >> 4;
    const int cb = (lg & 1) ? 16 + (lg - 1) * 4 : lg * 4;
    const RowStage rs = row_stage(lds_base, wave, lane);
    const char* cbase = sgpr_ptr(reinterpret_cast<bf16_t*>(p.C) + coff + (long)mw * p.ldc + nw);
    const char* abase = sgpr_ptr(p.aux_out + (long)mw * p.ldaux + nw);
    const unsigned cvoff = (unsigned)(rs.row * (int)p.ldc + rs.chunk * 8) * 2u, avoff = (unsigned)(rs.row * (int)p.ldaux + rs.chunk * 8) * 2u;
    const long cstep = 32 * p.ldc, astep = 32 * p.ldaux;                                // bytes per 16 rows
    f32x4 bv[2][2];
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
        bv[jp][0] = p.bias ? *reinterpret_cast<const f32x4*>(lds_bias + jp * 32 + cb) : f32x4{0.f, 0.f, 0.f, 0.f};
        bv[jp][1] = p.bias ? *reinterpret_cast<const f32x4*>(lds_bias + jp * 32 + cb + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    static_for<0, 8>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
        u32x4 cy[2], cd[2];
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
            f32x4 v0, v1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float xa = acc[i][2 * jp][r], xb = acc[i][2 * jp + 1][r];
                auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(xa), __float_as_uint(xb), false, false);
                v0[r] = __uint_as_float(sw[0]);
                v1[r] = __uint_as_float(sw[1]);
            }
            v0 = v0 * p.alpha + bv[jp][0];
            v1 = v1 * p.alpha + bv[jp][1];
            pb_f32x2 y[4], dy[4];
            gelu_pair2(pb_f32x2{v0[0], v0[1]}, y[0], dy[0]);
            gelu_pair2(pb_f32x2{v0[2], v0[3]}, y[1], dy[1]);
            gelu_pair2(pb_f32x2{v1[0], v1[1]}, y[2], dy[2]);
            gelu_pair2(pb_f32x2{v1[2], v1[3]}, y[3], dy[3]);
            const bf16x8 ry = {(bf16_t)y[0][0], (bf16_t)y[0][1], (bf16_t)y[1][0], (bf16_t)y[1][1], (bf16_t)y[2][0], (bf16_t)y[2][1], (bf16_t)y[3][0], (bf16_t)y[3][1]};
            const bf16x8 rd = {(bf16_t)dy[0][0], (bf16_t)dy[0][1], (bf16_t)dy[1][0], (bf16_t)dy[1][1], (bf16_t)dy[2][0], (bf16_t)dy[2][1], (bf16_t)dy[3][0], (bf16_t)dy[3][1]};
            cy[jp] = __builtin_bit_cast(u32x4, ry); cd[jp] = __builtin_bit_cast(u32x4, rd);
        }
        u32x4 lo, hi;
        row_stage_put(rs, cd[0], cd[1]);
        row_stage_get(rs, lo, hi);
        pf_store_nt<0>(avoff, abase + i * astep, lo);
        pf_store_nt<0>(avoff, abase + i * astep + 16 * p.ldaux, hi);                    // + 8 rows
        row_stage_put(rs, cy[0], cy[1]);
        row_stage_get(rs, lo, hi);
        pf_store<0>(cvoff, cbase + i * cstep, lo);
        pf_store<0>(cvoff, cbase + i * cstep + 16 * p.ldc, hi);
    });
}

// The store-only epilogue of an interior tile (C = alpha acc + bias, bf16) through the row staging: 16 stores of 8 rows x 128 bytes per wave.
__device__ __forceinline__ void epilogue_plain(const Gemm2Args& p, f32x4 (&acc)[8][4], int mw, int nw, long coff, int lane, const float* lds_bias, unsigned lds_base, int wave) {
    asm volatile("" : "+v"(lane));
    const int lr = lane & 15, lg = lane >> 4;
    const int cb = (lg & 1) ? 16 + (lg - 1) * 4 : lg * 4;
    const RowStage rs = row_stage(lds_base, wave, lane);
    const char* cbase = sgpr_ptr(reinterpret_cast<bf16_t*>(p.C) + coff + (long)mw * p.ldc + nw);
    const unsigned cvoff = (unsigned)(rs.row * (int)p.ldc + rs.chunk * 8) * 2u;
    const long cstep = 32 * p.ldc;
    f32x4 bv[2][2];
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
        bv[jp][0] = p.bias ? *reinterpret_cast<const f32x4*>(lds_bias + jp * 32 + cb) : f32x4{0.f, 0.f, 0.f, 0.f};
        bv[jp][1] = p.bias ? *reinterpret_cast<const f32x4*>(lds_bias + jp * 32 + cb + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    static_for<0, 8>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
        u32x4 cy[2];
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
            f32x4 v0, v1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float xa = acc[i][2 * jp][r], xb = acc[i][2 * jp + 1][r];
                auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(xa), __float_as_uint(xb), false, false);
                v0[r] = __uint_as_float(sw[0]);
                v1[r] = __uint_as_float(sw[1]);
            }
            v0 = v0 * p.alpha + bv[jp][0];
            v1 = v1 * p.alpha + bv[jp][1];
            const bf16x8 ry = {(bf16_t)v0[0], (bf16_t)v0[1], (bf16_t)v0[2], (bf16_t)v0[3], (bf16_t)v1[0], (bf16_t)v1[1], (bf16_t)v1[2], (bf16_t)v1[3]};
            cy[jp] = __builtin_bit_cast(u32x4, ry);
        }
        u32x4 lo, hi;
        row_stage_put(rs, cy[0], cy[1]);
        row_stage_get(rs, lo, hi);
        pf_store<0>(cvoff, cbase + i * cstep, lo);
        pf_store<0>(cvoff, cbase + i * cstep + 16 * p.ldc, hi);
    });
}

// The same for an f32 C without bias or accumulation (the split-K slabs of the weight gradients): a row of the wave's 64 columns is 256 bytes, so a 16-row
// group goes through the 2 KiB image one 32-column half at a time (a lane's f32x4 of column tile 2 jh is chunk lg, of tile 2 jh + 1 chunk 4 + lg) and
// leaves as 8 rows x 128 contiguous bytes per store: 32 stores per wave, as before.
__device__ __forceinline__ void epilogue_f32_plain(const Gemm2Args& p, f32x4 (&acc)[8][4], int mw, int nw, long coff, int lane, unsigned lds_base, int wave) {
    asm volatile("" : "+v"(lane));
    const int lr = lane & 15, lg = lane >> 4;
    RowStage rs = row_stage(lds_base, wave, lane);
    const unsigned base = lds_base + (unsigned)(G3_STAGE_OFF + wave * 2048);
    rs.wr0 = base + (unsigned)(lr * 128 + ((lg ^ (lr & 7)) << 4));
    rs.wr1 = base + (unsigned)(lr * 128 + (((4 + lg) ^ (lr & 7)) << 4));
    const char* cbase = sgpr_ptr(reinterpret_cast<float*>(p.C) + coff + (long)mw * p.ldc + nw);
    const unsigned cvoff = (unsigned)(rs.row * (int)p.ldc + rs.chunk * 4) * 4u;
    const long cstep = 64 * p.ldc;                                                       // bytes per 16 rows
    static_for<0, 16>([&](auto cc) {
        constexpr int c = decltype(cc)::value, i = c >> 1, jh = c & 1;
        const f32x4 a0 = acc[i][2 * jh] * p.alpha, a1 = acc[i][2 * jh + 1] * p.alpha;
        u32x4 lo, hi;
        row_stage_put(rs, __builtin_bit_cast(u32x4, a0), __builtin_bit_cast(u32x4, a1));
        row_stage_get(rs, lo, hi);
        pf_store<jh * 128>(cvoff, cbase + i * cstep, lo);
        pf_store<jh * 128>(cvoff, cbase + i * cstep + 32 * p.ldc, hi);                   // + 8 rows
    });
}

// One-barrier kernel. WM x WN waves, each TM x TN MFMA tiles of 16x16: block tile BM = 16*WM*TM by BN = 16*WN*TN.
// Measured: the 128x128 main loop is bound by the L2 -> LDS load path (64 FLOP per loaded byte, ~1 PF ceiling); with two
// workgroups per CU one's store tail overlaps the other's main loop, which is why it still serves the narrow outputs.
template <bool A_KC, bool B_KC, int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(WM * WN * 64) void gemm2_kernel(const Gemm2Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = WM * WN, BM = 16 * WM * TM, BN = 16 * WN * TN;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE_BYTES = A_BYTES + B_BYTES;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WN, wn = wave % WN;
    int m0, n0, zs;
    block_tile<BM, BN>(p, blockIdx.x, m0, n0, zs);
    const int z = blockIdx.y;                                           // the division runs on the vector unit: back to scalars, or the
    const int z1 = __builtin_amdgcn_readfirstlane(z / p.nb2), z2 = __builtin_amdgcn_readfirstlane(z % p.nb2);   // 64-bit batch offsets live in VGPRs across the K loop (2 spilled)
    const bf16_t* A = p.A + z1 * p.sA1 + z2 * p.sA2;
    const bf16_t* B = p.B + z1 * p.sB1 + z2 * p.sB2;
    const long coff = z1 * p.sC1 + z2 * p.sC2 + zs * p.sCz;
    const int kbeg = zs * p.Kc, kend = min(p.K, kbeg + p.Kc);
    const int nk = max(0, kend - kbeg) / BK;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (nk > 0) {
        stage<A_KC, BM, NW>(A, p.lda, m0, kbeg, p.M, smem, wave, lane);
        stage<B_KC, BN, NW>(B, p.ldb, n0, kbeg, p.N, smem + A_BYTES, wave, lane);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const char* cur = smem + (kt & 1) * STAGE_BYTES;
        char* nxt = smem + ((kt + 1) & 1) * STAGE_BYTES;
        if (kt + 1 < nk) {
            stage<A_KC, BM, NW>(A, p.lda, m0, kbeg + (kt + 1) * BK, p.M, nxt, wave, lane);
            stage<B_KC, BN, NW>(B, p.ldb, n0, kbeg + (kt + 1) * BK, p.N, nxt + A_BYTES, wave, lane);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = frag<A_KC, BM>(cur, (wm * TM + i) * 16, ks, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = frag<B_KC, BN>(cur + A_BYTES, (wn * TN + j) * 16, ks, lane);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);   // swapped: see epilogue_regs
        }
        __syncthreads();          // hipcc adds s_waitcnt vmcnt(0) here: next tile has landed, current one is free
    }
    if (p.flags & 128) return;                                      // bit 7: profiling build without the epilogue

    static_assert(TN == 4, "epilogue_regs pairs the 4 column tiles of a wave");
    epilogue_regs<TM>(p, acc, m0 + wm * (16 * TM), n0 + wn * (16 * TN), coff, lane);
}

// ---------------------------------------------------------------------------------------------------------------------
// 256 x 256 x 64 tile, 8 waves (2 x 4, 128 x 64 per wave), ping-pong schedule with the DMA prefetch in flight across
// barriers (cdna_hip_programming.md 5, "What does break it"): one workgroup per CU, 128 KiB of LDS = 2 K-tile slots x
// {A0, A1, B0, B1} half-tiles of 16 KiB. Half h of an operand holds, for every wave row/column group, the h-th half of
// that group's rows (A: 64 of its 128 rows, B: 32 of its 64 columns), so a K-tile is consumed in 4 phases of one C
// quadrant each -- (A0,B0) (A0,B1) (A1,B1) (A1,B0): 16 MFMAs on K = 64 after 12 / 4 / 8 / 4 fragment reads -- and each
// phase re-stages one half-tile (2 DMA instructions per wave) that went dead two phases earlier:
//     phase 4t+0: A1(t+1)   4t+1: B0(t+1)   4t+2: A0(t+2)   4t+3: B1(t+2), then s_waitcnt vmcnt(4)
// (everything but the two newest half-tiles has landed: K-tile t+1 is complete one phase before its first read). The two
// wave rows run one barrier interval apart (the wr = 1 waves take one extra s_barrier up front, the wr = 0 waves one at
// the end), and every phase is  reads + DMA | s_barrier | 16 MFMA | s_barrier : while one wave of a SIMD issues its
// MFMAs the other one issues its LDS reads and DMA. Barriers are raw s_barrier (a __syncthreads() would drain vmcnt).
// RAW: DMA data is read one phase after the counted wait (wait -> barrier -> barrier of the staggered group -> read);
// WAR: a half-tile is re-staged two phases after its last read, whose lgkmcnt(0) sits one interval before.
template <bool KC, int GS, int GSTRIDE = 2 * GS, int N0 = 0, int N1 = 2>
__device__ __forceinline__ void stage_half(const bf16_t* __restrict__ base, long ld, int r0, int k0, int R, char* lds, int h, int wave, int lane) {
#pragma unroll
    for (int n = N0; n < N1; ++n) {
        const int inst = wave * 2 + n;
        if constexpr (KC) {
            const int hr = inst * 8 + (lane >> 3);                                  // row of the half-tile image [128][128 B]
            const int chunk = (lane & 7) ^ kswz(hr);
            const int gr = min(r0 + (hr / GS) * GSTRIDE + h * GS + (hr % GS), R - 1);
            glds16(base + (long)gr * ld + k0 + chunk * 8, lds + inst * 1024);
        } else {
            const int krow = inst * 4 + (lane >> 4);                                // image [64 k][128 columns = 256 B]
            const int chunk = (lane & 15) ^ rswz(krow);
            const int hc = chunk * 8;
            const int gc = min(r0 + (hc / GS) * GSTRIDE + h * GS + (hc % GS), R - 8);
            glds16(base + (long)(k0 + krow) * ld + gc, lds + inst * 1024);
        }
    }
}

// Transposed fragment reads of the ping-pong kernel go through inline asm: for the ds_read_tr builtin hipcc (ROCm 7.2) cannot
// tell the read apart from the LDS-DMA writes in flight and puts s_waitcnt vmcnt(0) in front of the first one of every
// K-tile, which drains the prefetch this kernel exists for (+40 % wave cycles on the NN / TN layouts, found in the .s).
// Form (ii) of cdna_hip_programming.md 5.7: "=v" loads, then ONE wait statement naming every destination "+v" before the
// first consumer; the per-tile lane offset is computed once, (ks, +4 rows) are immediate offsets.
template <int OFF>
__device__ __forceinline__ void ds_tr(s16x4& d, unsigned addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "i"(OFF));
}
__device__ __forceinline__ void ds_tr_frags(s16x4 (&d)[2][2], unsigned addr) {       // [ks][lo/hi] of one 16-row tile, image rows of 256 B
    ds_tr<0>(d[0][0], addr); ds_tr<1024>(d[0][1], addr); ds_tr<8192>(d[1][0], addr); ds_tr<9216>(d[1][1], addr);
}
__device__ __forceinline__ unsigned tr_lane_off(int rbase, int lane) {                // frag<false, 128> minus (ks, +4 rows)
    const int lr = lane & 15, g = lane >> 4, q = lr >> 2, pp = lr & 3;
    const int chunk = (rbase >> 3) + (pp >> 1);
    return (unsigned)((g * 8 + q) * 256 + ((chunk ^ rswz(g * 8 + q)) << 4) + ((pp & 1) << 3));
}
__device__ __forceinline__ bf16x8 tr_join(s16x4 lo, s16x4 hi) {
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}
#define TRW4(X) "+v"(X[0][0]), "+v"(X[0][1]), "+v"(X[1][0]), "+v"(X[1][1])
__device__ __forceinline__ void tr_wait_a(s16x4 (&ta)[4][2][2]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : TRW4(ta[0]), TRW4(ta[1]), TRW4(ta[2]), TRW4(ta[3]));
}
__device__ __forceinline__ void tr_wait_b(s16x4 (&tb)[2][2][2]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : TRW4(tb[0]), TRW4(tb[1]));
}
__device__ __forceinline__ void tr_wait_ab(s16x4 (&ta)[4][2][2], s16x4 (&tb)[2][2][2]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : TRW4(ta[0]), TRW4(ta[1]), TRW4(ta[2]), TRW4(ta[3]), TRW4(tb[0]), TRW4(tb[1]));
}
#undef TRW4

#ifdef PB_G3_STAMPS
#define G3_STAMP(i) { __builtin_amdgcn_sched_barrier(0); const unsigned now_ = (unsigned)__builtin_amdgcn_s_memtime(); g3_acc[i] += now_ - g3_last; g3_last = now_; __builtin_amdgcn_sched_barrier(0); }
#else
#define G3_STAMP(i)
#endif
template <bool A_KC, bool B_KC, int TNW>
__global__ __launch_bounds__(512) void gemm3_kernel(const Gemm2Args p) {
    // TNW = column tiles (16 wide) per wave: 4 -> 256 x 256 block tile; 3 -> 256 x 192 (N = 768: 512 tiles = 2 full rounds of 256 CUs
    // where 256 x 256 gives 384 = 1.5). With TNW = 3 the second B half still stages 32 columns per wave column (16 of them belong
    // to the neighbour), so the DMA piece counts and with them every vmcnt stay as they are; its phases run 8 MFMAs instead of 16.
    constexpr int BNT = 64 * TNW, GSB = 16 * TNW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int HALF = 16384, SLOT = 4 * HALF;
    // LDS map of the stage ring: [A0 slot 0 | A0 slot 1 | A1 slot 0 | A1 slot 1 | B0 s0 | B0 s1 | B1 s0 | B1 s1], 16 KiB each. The (half, slot) of an operand is
    // then one of four offsets below 64 KiB, i.e. an IMMEDIATE of the ds_read: a fragment address needs one register per (row tile, k-step), not one more
    // per slot (with the slots 64 KiB apart the second slot lay beyond the 16-bit offset field).
#define G3_OFF_A(SL, H) ((H) * 2 * HALF + (SL) * HALF)
#define G3_OFF_B(SL, H) (SLOT + (H) * 2 * HALF + (SL) * HALF)
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    // Persistent over the (split, tile) work list: workgroup b takes items b, b + grid, ... (grid = a multiple of 8, so an item's
    // XCD chunk in block_tile stays the workgroup's XCD). The first DMA pieces of the NEXT item are issued right after the last
    // barrier of the current one, i.e. before its epilogue: the ~2 us from first DMA to first MFMA hide under the store tail.
    const int ntile = p.tiles_m * p.tiles_n * p.nsplit;
    const int total = p.tail_split > 1 ? p.n_full + (ntile - p.n_full) * p.tail_split : ntile;
    int L = blockIdx.x;
    int m0, n0, zs, kbeg, nk;
    auto item = [&](int l) {
        if (!(p.tail_split > 1 && l >= p.n_full)) {
            block_tile<256, BNT>(p, l, m0, n0, zs);
            kbeg = zs * p.Kc;
            nk = max(0, min(p.K, kbeg + p.Kc) - kbeg) / BK;
        } else {
            const int u = l - p.n_full, ks = u % p.tail_split;
            block_tile<256, BNT>(p, p.n_full + u / p.tail_split, m0, n0, zs);
            kbeg = ks * p.tail_kc;
            nk = max(0, min(p.K, kbeg + p.tail_kc) - kbeg) / BK;
        }
    };
    item(L);
    const int z = blockIdx.y;                                           // the division runs on the vector unit: back to scalars, or the
    const int z1 = __builtin_amdgcn_readfirstlane(z / p.nb2), z2 = __builtin_amdgcn_readfirstlane(z % p.nb2);   // 64-bit batch offsets live in VGPRs across the K loop (2 spilled)
    const bf16_t* A = p.A + z1 * p.sA1 + z2 * p.sA2;
    const bf16_t* B = p.B + z1 * p.sB1 + z2 * p.sB2;

    // K-contiguous operands (NT): the K-tile is consumed BY K-STEP instead of by quadrant (round 5) -- see the loop
    constexpr bool KSPLIT = A_KC == B_KC && TNW == 4;                  // (round 5, later: the weight-gradient layout TN too -- both operands row-contiguous, transposed reads)
    constexpr bool KS_NT = KSPLIT && A_KC, KS_TN = KSPLIT && !A_KC;
    f32x4 acc[8][TNW];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < TNW; ++j) { acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; if constexpr (KS_TN) asm volatile("" : "+v"(acc[i][j])); }   // TN: opaque (no second copy of the pinned-register loop); NT: the first K-tile's MFMAs take C = 0 and the 128 v_mov go away
    bf16x8 a[4][2], b[2][2];
    s16x4 ta[4][2][2], tb[2][2][2];                                   // asm destinations of the transposed reads
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    unsigned aoff[4], boff[2];                                         // transposed reads: lane offsets of the row / column tiles (boff: relative to the B region in the k-step loop)
    if constexpr (!KS_TN) {
#pragma unroll
        for (int i = 0; i < 4; ++i) aoff[i] = lds0 + tr_lane_off(wr * 64 + i * 16, lane);
#pragma unroll
        for (int j = 0; j < 2; ++j) boff[j] = lds0 + tr_lane_off(wc * 32 + j * 16, lane);
    }
    // K-contiguous fragment addresses of the k-step loop (frag<true, 128>'s formula). Recomputed from the (opaque) lane number in front of every item's K loop:
    // 12 registers the epilogue in between does not have to keep
    unsigned fa[4][2], fb[2][2];
    auto frag_setup = [&]() {
        if constexpr (KS_TN) {
            int lq = lane; asm volatile("" : "+v"(lq));
#pragma unroll
            for (int i = 0; i < 4; ++i) { aoff[i] = lds0 + tr_lane_off(wr * 64 + i * 16, lq); asm volatile("" : "+v"(aoff[i])); }
#pragma unroll
            for (int j = 0; j < 2; ++j) { boff[j] = lds0 + (unsigned)SLOT + tr_lane_off(wc * 32 + j * 16, lq); asm volatile("" : "+v"(boff[j])); }
        }
        if constexpr (A_KC && B_KC) {
            int lq = lane; asm volatile("" : "+v"(lq));
            const int lr_ = lq & 15, g_ = lq >> 4;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { const int row = wr * 64 + i * 16 + lr_; fa[i][ks] = lds0 + (unsigned)(row * 128 + (((ks * 4 + g_) ^ kswz(row)) << 4)); asm volatile("" : "+v"(fa[i][ks])); }
#pragma unroll
                for (int j = 0; j < 2; ++j) { const int row = wc * 32 + j * 16 + lr_; fb[j][ks] = lds0 + (unsigned)(SLOT + row * 128 + (((ks * 4 + g_) ^ kswz(row)) << 4)); asm volatile("" : "+v"(fb[j][ks])); }
            }
        }
    };

#define G3_ISSUE_A(T, H) stage_half<A_KC, 64>(A, p.lda, m0, kbeg + (T) * BK, p.M, smem + G3_OFF_A((T) & 1, H), H, wave, lane)
#define G3_ISSUE_B(T, H) stage_half<B_KC, 32, GSB>(B, p.ldb, n0, kbeg + (T) * BK, p.N, smem + G3_OFF_B((T) & 1, H), H, wave, lane)
#define G3_ISSUE_A1(T, H, PC) stage_half<A_KC, 64, 128, PC, PC + 1>(A, p.lda, m0, kbeg + (T) * BK, p.M, smem + G3_OFF_A((T) & 1, H), H, wave, lane)
#define G3_ISSUE_B1(T, H, PC) stage_half<B_KC, 32, GSB, PC, PC + 1>(B, p.ldb, n0, kbeg + (T) * BK, p.N, smem + G3_OFF_B((T) & 1, H), H, wave, lane)
    // K-split (NT) path: LDS-DMA addressing = SGPR base of the item's operand panel at its first k + one 32-bit lane offset per piece (row clamped once per item)
    unsigned voA[2][2] = {{0u, 0u}, {0u, 0u}}, voB[2][2] = {{0u, 0u}, {0u, 0u}};
    const char *sA = nullptr, *sB = nullptr;
    auto dma_setup = [&]() {
        if constexpr (KS_TN) {                                         // image [64 k][128 columns]: a piece = 4 k-rows, a lane = 8 columns (clamped once per item)
            int lq = lane; asm volatile("" : "+v"(lq));
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int krow = (wave * 2 + n) * 4 + (lq >> 4), hc = (((lq & 15) ^ rswz(krow))) * 8;
                    const int ca = min(m0 + (hc / 64) * 128 + h * 64 + (hc % 64), p.M - 8) - m0;
                    const int cb = min(n0 + (hc / 32) * GSB + h * 32 + (hc % 32), p.N - 8) - n0;
                    voA[h][n] = (unsigned)((krow * (int)p.lda + ca) * 2);
                    voB[h][n] = (unsigned)((krow * (int)p.ldb + cb) * 2);
                }
            sA = sgpr_ptr(A + (long)kbeg * p.lda + m0);
            sB = sgpr_ptr(B + (long)kbeg * p.ldb + n0);
        }
        if constexpr (KS_NT) {
            int lq = lane; asm volatile("" : "+v"(lq));
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int hr = (wave * 2 + n) * 8 + (lq >> 3), chunk = (lq & 7) ^ kswz(hr);
                    const int ra = min(m0 + (hr / 64) * 128 + h * 64 + (hr % 64), p.M - 1) - m0;
                    const int rb = min(n0 + (hr / 32) * GSB + h * 32 + (hr % 32), p.N - 1) - n0;
                    voA[h][n] = (unsigned)((ra * (int)p.lda + chunk * 8) * 2);
                    voB[h][n] = (unsigned)((rb * (int)p.ldb + chunk * 8) * 2);
                }
            sA = sgpr_ptr(A + (long)m0 * p.lda + kbeg);
            sB = sgpr_ptr(B + (long)n0 * p.ldb + kbeg);
        }
    };
#define G3S_ISSUE_A(T, H) do { _Pragma("unroll") for (int n_ = 0; n_ < 2; ++n_) glds16_s(voA[H][n_], sA + (long)(T) * (A_KC ? (long)(BK * 2) : (long)(BK * 2) * p.lda), lds0 + (unsigned)(G3_OFF_A((T) & 1, H) + (wave * 2 + n_) * 1024)); } while (0)
#define G3S_ISSUE_B(T, H) do { _Pragma("unroll") for (int n_ = 0; n_ < 2; ++n_) glds16_s(voB[H][n_], sB + (long)(T) * (B_KC ? (long)(BK * 2) : (long)(BK * 2) * p.ldb), lds0 + (unsigned)(G3_OFF_B((T) & 1, H) + (wave * 2 + n_) * 1024)); } while (0)
#define G3_READ_A(SL, H)                                                                                          \
    do {                                                                                                          \
        if constexpr (A_KC) {                                                                                     \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)         \
                a[i][ks] = frag<true, 128>(smem + G3_OFF_A(SL, H), wr * 64 + i * 16, ks, lane);            \
        } else {                                                                                                  \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) ds_tr_frags(ta[i], aoff[i] + (unsigned)G3_OFF_A(SL, H)); \
        }                                                                                                         \
    } while (0)
#define G3_READ_B(SL, H)                                                                                          \
    do {                                                                                                          \
        if constexpr (B_KC) {                                                                                     \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)         \
                b[j][ks] = frag<true, 128>(smem + G3_OFF_B(SL, H), wc * 32 + j * 16, ks, lane);      \
        } else {                                                                                                  \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) ds_tr_frags(tb[j], boff[j] + (unsigned)G3_OFF_B(SL, H)); \
        }                                                                                                         \
    } while (0)
    // RA / RB: this phase read A / B fragments (the asm destinations among them are named in the wait)
#define G3_MMA(MH, NH, RA, RB, MID)                                                                               \
    do {                                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        __builtin_amdgcn_s_barrier();                                                                             \
        if constexpr ((RA) && !A_KC && (RB) && !B_KC) tr_wait_ab(ta, tb);                                         \
        else if constexpr ((RA) && !A_KC) tr_wait_a(ta);                                                          \
        else if constexpr ((RB) && !B_KC) tr_wait_b(tb);                                                          \
        else __builtin_amdgcn_s_waitcnt(0xc07f);                                                                  \
        if constexpr ((RA) && !A_KC) {                                                                            \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) a[i][ks] = tr_join(ta[i][ks][0], ta[i][ks][1]); \
        }                                                                                                         \
        if constexpr ((RB) && !B_KC) {                                                                            \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) b[j][ks] = tr_join(tb[j][ks][0], tb[j][ks][1]); \
        }                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        __builtin_amdgcn_s_setprio(1);                                                                            \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                         \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                          \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                      \
                    if ((NH) * 2 + j < TNW)                                                                           \
                        acc[(MH) * 4 + i][((NH) * 2 + j < TNW) ? (NH) * 2 + j : 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j][ks], a[i][ks], acc[(MH) * 4 + i][((NH) * 2 + j < TNW) ? (NH) * 2 + j : 0], 0, 0, 0); \
            if (ks == 0) { __builtin_amdgcn_sched_barrier(0); MID; __builtin_amdgcn_sched_barrier(0); }           \
        }                                                                                                         \
        __builtin_amdgcn_s_setprio(0);                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        __builtin_amdgcn_s_barrier();                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)

    // the tile's 256 bias values ride along as one more DMA piece (wave 0, issued FIRST so that the counted waits, which spare
    // only the youngest operations, never see it): an ordinary bias load in the epilogue would draw the compiler's vmcnt(0) and
    // drain the next item's prologue. Two alternating 1 KiB slots above the stage ring: item i's epilogue reads while item i+1's
    // piece lands.
#define G3_PROLOGUE()                                                                                             \
    do {                                                                                                          \
        bslot ^= 1;                                                                                               \
        if (p.bias && wave == 0) { int lq = lane; asm volatile("" : "+v"(lq)); glds16(reinterpret_cast<const bf16_t*>(p.bias + min(n0 + lq * 4, p.N - 4)), smem + 2 * SLOT + bslot * 1024); } \
        if constexpr (KSPLIT) {                                                                                   \
            dma_setup();                                                                                          \
            if (nk > 0) { G3S_ISSUE_A(0, 0); G3S_ISSUE_B(0, 0); G3S_ISSUE_B(0, 1); G3S_ISSUE_A(0, 1); }           \
            if (nk > 1) { G3S_ISSUE_A(1, 0); G3S_ISSUE_B(1, 0); }                                                 \
        } else {                                                                                                  \
            if (nk > 0) { G3_ISSUE_A(0, 0); G3_ISSUE_B(0, 0); G3_ISSUE_B(0, 1); G3_ISSUE_A(0, 1); }               \
            if (nk > 1) { G3_ISSUE_A(1, 0); G3_ISSUE_B(1, 1); }                                                   \
        }                                                                                                         \
    } while (0)
    int bslot = 0;
#ifdef PB_G3_STAMPS
    unsigned g3_acc[6] = {0, 0, 0, 0, 0, 0}, g3_last = (unsigned)__builtin_amdgcn_s_memtime();     // 32-bit cycle sums: a launch is < 2^32 cycles
    const unsigned g3_t0 = g3_last, g3_r0 = (unsigned)__builtin_amdgcn_s_memrealtime();          // s_memrealtime: constant 100 MHz -> the clock the wave really ran at
    unsigned g3_items = 0;
#endif
    G3_PROLOGUE();
    int pend = 0;                                                    // store instructions this wave left in flight behind the prologue
    G3_STAMP(0);                                                     // [0] first prologue: addressing + DMA issue
    while (true) {
        if constexpr (KSPLIT) { dma_setup(); frag_setup(); }        // (again: the copies the prologue used died with it, the epilogue in between ran without them)
        // K-tile 0 of this item must have landed. vmcnt counts loads, stores and DMA pieces in ONE in-order queue, and behind
        // K-tile 0's pieces sit the 4 pieces of K-tile 1 and the `pend` stores of the previous item's epilogue (exactly 16 / 32
        // per wave when that tile was interior and store-only; 0 = "unknown", which waits for the stores too): leave them flying.
        if (nk > 1) {
            if (pend == 40) { asm volatile("s_waitcnt vmcnt(44)" ::: "memory"); }
            else if (pend == 36) { asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); }
            else if (pend == 32) { asm volatile("s_waitcnt vmcnt(36)" ::: "memory"); }
            else if (pend == 16) { asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); }
            else { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        } else {
            if (pend == 40) { asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); }
            else if (pend == 36) { asm volatile("s_waitcnt vmcnt(36)" ::: "memory"); }
            else if (pend == 32) { asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); }
            else if (pend == 16) { asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }
            else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        }
        __builtin_amdgcn_s_barrier();
        if (wr == 1) __builtin_amdgcn_s_barrier();
        G3_STAMP(1);                                                 // [1] wait for K-tile 0 (+ the stagger barrier)
        if constexpr (KSPLIT) {
            // Round 5, both operands K-contiguous. Knock-out builds (profiles/r05_gemm_kloop_knockouts.txt) put the quadrant loop's excess over the MFMA floor
            // (17.6 against 13.1 us per 12 K-tiles) on the fragment reads: not their number (24 or 28 per K-tile measured equal) but their LATENCY, paid behind
            // the barrier in front of every MFMA cluster. So a K-tile is consumed BY K-STEP -- phases (rows half 0, k-step 0) (half 0, k-step 1) (half 1, k-step 0)
            // (half 1, k-step 1): 4 row tiles x ALL 4 column tiles on one k-step -- and the A fragments of the NEXT phase are requested INSIDE this phase's MFMA
            // cluster, each row tile's right behind the four MFMAs that were its last readers (same registers: no second set), the B fragments of a k-step in
            // the load half one phase ahead of their first use (they live for two phases). Behind a barrier the MFMAs start at once.
            //     phase 4t+0: load half: B.k1(t) -> phase 1; DMA B1(t+1); vmcnt(6): A1(t) landed      MFMAs A0.k0 x B.k0, between them A0.k1(t) -> phase 1
            //     phase 4t+1: load half: DMA A1(t+1)                                                   MFMAs A0.k1 x B.k1, between them A1.k0(t) -> phase 2
            //     phase 4t+2: load half: DMA A0(t+2); vmcnt(4): A0, B0, B1 of t+1 landed               MFMAs A1.k0 x B.k0, between them A1.k1(t) -> phase 3
            //     phase 4t+3: load half: B.k0(t+1) -> phase 0; DMA B0(t+2)                             MFMAs A1.k1 x B.k1, between them A0.k0(t+1) -> phase 0
            // Every counted wait sits one phase (two barriers, the staggered group's included) ahead of the first read of what it retires; a half-tile is
            // re-staged >= 2 barrier intervals after the last request that reads it (A0: phase 0's cluster -> phase 2; B: phase 0's load half -> phases 3 and,
            // for the other slot, 0; A1: phase 2's cluster -> phase 1 of the next K-tile).
            bf16x8 a1[4], bk[2][4];
            // fragment addresses: one register per (row tile, k-step) for A and per (column tile, k-step) for B, the B ones relative to the B region (its
            // offsets would not fit the ds_read's 16-bit immediate from the ring's start, and hipcc then keeps one register per (half, slot) as well)
            // The weight-gradient layout (TN: both operands row-contiguous) runs the same schedule. A fragment there = two ds_read_b64_tr_b16 (k rows +0..3 and
            // +4..7 of each 8-row group) that must land in the two halves of ONE 128-bit operand. Written as two 64-bit values joined afterwards (asm or the
            // compiler's builtin alike) the allocator falls apart once such a pair is in flight across phases (370 - 507 spilled registers); so each of the 12
            // fragments is ONE asm statement with a 128-bit result PINNED to a register quad, its halves named literally (v200-v247), and the cluster's first
            // statement is the counted wait that names what it is about to use: lgkmcnt(8) where the load half just requested the 8 B reads of the next k-step
            // (LDS returns in order), lgkmcnt(0) elsewhere. The disassembly must show no copy of v200-v247 inside the loop (tools/check_gemm_isa.py).
#define G3K_TRL(VAR, R0, R1, R2, R3, ADDR, OFF)                                                                   \
    asm volatile("ds_read_b64_tr_b16 v[" #R0 ":" #R1 "], %1 offset:%2\n\tds_read_b64_tr_b16 v[" #R2 ":" #R3 "], %1 offset:%3"         \
                 : "={v[" #R0 ":" #R3 "]}"(VAR) : "v"(ADDR), "i"(OFF), "i"((OFF) + 1024))
#define G3K_TRA(I, OFF) do { if ((I) == 0) G3K_TRL(a1[0], 200, 201, 202, 203, aoff[0], OFF); else if ((I) == 1) G3K_TRL(a1[1], 204, 205, 206, 207, aoff[1], OFF); \
                             else if ((I) == 2) G3K_TRL(a1[2], 208, 209, 210, 211, aoff[2], OFF); else G3K_TRL(a1[3], 212, 213, 214, 215, aoff[3], OFF); } while (0)
#define G3K_TRB(SL, KS) do { if constexpr ((KS) == 0) {                                                          \
            G3K_TRL(bk[0][0], 216, 217, 218, 219, boff[0], G3_OFF_B(SL, 0) - SLOT); G3K_TRL(bk[0][1], 220, 221, 222, 223, boff[1], G3_OFF_B(SL, 0) - SLOT); \
            G3K_TRL(bk[0][2], 224, 225, 226, 227, boff[0], G3_OFF_B(SL, 1) - SLOT); G3K_TRL(bk[0][3], 228, 229, 230, 231, boff[1], G3_OFF_B(SL, 1) - SLOT); \
        } else {                                                                                                  \
            G3K_TRL(bk[1][0], 232, 233, 234, 235, boff[0], G3_OFF_B(SL, 0) - SLOT + 8192); G3K_TRL(bk[1][1], 236, 237, 238, 239, boff[1], G3_OFF_B(SL, 0) - SLOT + 8192); \
            G3K_TRL(bk[1][2], 240, 241, 242, 243, boff[0], G3_OFF_B(SL, 1) - SLOT + 8192); G3K_TRL(bk[1][3], 244, 245, 246, 247, boff[1], G3_OFF_B(SL, 1) - SLOT + 8192); \
        } } while (0)
#define G3K_A4 "+v"(a1[0]), "+v"(a1[1]), "+v"(a1[2]), "+v"(a1[3])
#define G3K_B4(KS) "+v"(bk[KS][0]), "+v"(bk[KS][1]), "+v"(bk[KS][2]), "+v"(bk[KS][3])
    // NEWB: the phase is the first reader of B k-step KS; AHEAD: the load half in front of it requested 8 reads the phase does not need
#define G3K_TRWAIT(KS, NEWB, AHEAD) do { if constexpr (!A_KC) {                                                   \
            if constexpr ((NEWB) && (AHEAD)) asm volatile("s_waitcnt lgkmcnt(8)" : G3K_A4, G3K_B4(KS));           \
            else if constexpr (NEWB) asm volatile("s_waitcnt lgkmcnt(0)" : G3K_A4, G3K_B4(KS));                   \
            else if constexpr (AHEAD) asm volatile("s_waitcnt lgkmcnt(8)" : G3K_A4);                              \
            else asm volatile("s_waitcnt lgkmcnt(0)" : G3K_A4); } } while (0)
#define G3K_READ_A1(I, SL, H, KS) do { if constexpr (A_KC) a1[I] = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(fa[I][KS] + (unsigned)G3_OFF_A(SL, H)); \
                                       else G3K_TRA(I, G3_OFF_A(SL, H) + (KS) * 8192); } while (0)
#define G3K_READ_B(SL, KS) do { if constexpr (B_KC) { _Pragma("unroll") for (int jj = 0; jj < 4; ++jj)                                                 \
        bk[KS][jj] = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(fb[jj & 1][KS] + (unsigned)(G3_OFF_B(SL, jj >> 1) - SLOT)); }     \
        else G3K_TRB(SL, KS); } while (0)
    // the cluster of a phase: for every row tile its four MFMAs, then (COND) the request of that row tile's fragment for the next phase
#define G3K_MMA(MH, KS, COND, SL, H, NKS, NEWB, AHEAD)                                                            \
    do {                                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        __builtin_amdgcn_s_barrier();                                                                             \
        G3K_TRWAIT(KS, NEWB, AHEAD);                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        __builtin_amdgcn_s_setprio(1);                                                                            \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                            \
            _Pragma("unroll") for (int jj = 0; jj < 4; ++jj)                                                       \
                acc[(MH) * 4 + i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bk[KS][jj], a1[i], acc[(MH) * 4 + i][jj], 0, 0, 0); \
            __builtin_amdgcn_sched_barrier(0);                                                                    \
            if (COND) G3K_READ_A1(i, SL, H, NKS);                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                    \
        }                                                                                                         \
        __builtin_amdgcn_s_setprio(0);                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        __builtin_amdgcn_s_barrier();                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)
            if (nk > 0) {                                               // phase 0 of K-tile 0: the one exposed read latency of the item
                G3K_READ_B(0, 0);
                _Pragma("unroll") for (int i = 0; i < 4; ++i) G3K_READ_A1(i, 0, 0, 0);
            }
            // one K-tile out of LDS slot SL (a literal: the loop is written out for both slots)
#define G3K_TILE(SL, KT)                                                                                          \
    do {                                                                                                          \
        G3K_READ_B(SL, 1);                                                                                        \
        if ((KT) + 1 < nk) {                                                                                      \
            G3S_ISSUE_B((KT) + 1, 1);                                                                              \
            if ((KT) > 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      /* A1(t) is older than A0(t+1), B0(t+1), B1(t+1) */ \
        } else if ((KT) > 0) {                                                                                    \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                      \
        }                                                                                                         \
        G3K_MMA(0, 0, true, SL, 0, 1, true, true);                                                                            \
        if ((KT) + 1 < nk) G3S_ISSUE_A((KT) + 1, 1);                                                               \
        G3K_MMA(0, 1, true, SL, 1, 0, true, false);                                                                           \
        if ((KT) + 2 < nk) { G3S_ISSUE_A((KT) + 2, 0); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }          /* in flight: A1(t+1), A0(t+2) */ \
        else if ((KT) + 1 < nk) { asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }                              /* in flight: A1(t+1) */ \
        G3K_MMA(1, 0, true, SL, 1, 1, false, false);                                                                           \
        G3K_READ_B(1 - (SL), 0);                /* behind the last K-tile these read stale LDS that nobody uses: no branch inside the schedule */ \
        if ((KT) + 2 < nk) G3S_ISSUE_B((KT) + 2, 0);                                                               \
        G3K_MMA(1, 1, true, 1 - (SL), 0, 0, false, true);                                                                     \
    } while (0)
            for (int kt = 0; kt < nk; kt += 2) {
                G3K_TILE(0, kt);
                if (kt + 1 >= nk) break;
                G3K_TILE(1, kt + 1);
            }
            if constexpr (!A_KC) asm volatile("s_waitcnt lgkmcnt(0)" : G3K_A4, G3K_B4(0));    // the stale requests behind the last K-tile retire before their registers are anyone else's
#undef G3K_TILE
#undef G3K_TRL
#undef G3K_TRA
#undef G3K_TRB
#undef G3K_A4
#undef G3K_B4
#undef G3K_TRWAIT
#undef G3K_READ_A1
#undef G3K_READ_B
#undef G3K_MMA
        } else
        for (int kt = 0; kt < nk; ++kt) {
            const int sl = kt & 1;
            // phase 0: quadrant (0,0)
            G3_READ_B(sl, 0);
            __builtin_amdgcn_sched_barrier(0);
            G3_READ_A(sl, 0);
            // Where an operand needs transposed fragments (NN, TN) each phase's two DMA pieces are SPLIT: one in the load half, one
            // between the two k-steps of the MFMA half -- there the load half (24 reads + a DMA pair) sets the interval and a DMA
            // piece is its most expensive instruction (TN w1 999 -> 1037 TF, NN dfc1 1052 -> 1115). With two K-contiguous
            // operands the load half is short and the split only delays MFMAs (NT 8192^3 1431 -> 1298): both pieces stay there.
            constexpr bool SPLIT = !(A_KC && B_KC);
#define G3_ISSUE2(ISS, T, H) do { if constexpr (SPLIT) { ISS##1(T, H, 0); } else { ISS(T, H); } } while (0)
#define G3_MID(ISS, COND, T, H) do { if constexpr (SPLIT) { if (COND) ISS##1(T, H, 1); } } while (0)
            if (kt + 1 < nk) G3_ISSUE2(G3_ISSUE_A, kt + 1, 1);
            G3_MMA(0, 0, true, true, G3_MID(G3_ISSUE_A, kt + 1 < nk, kt + 1, 1));
            // phase 1: quadrant (0,1)
            G3_READ_B(sl, 1);
            if (kt + 1 < nk) G3_ISSUE2(G3_ISSUE_B, kt + 1, 0);
            G3_MMA(0, 1, false, true, G3_MID(G3_ISSUE_B, kt + 1 < nk, kt + 1, 0));
            // phase 2: quadrant (1,1)
            G3_READ_A(sl, 1);
            if (kt + 2 < nk) G3_ISSUE2(G3_ISSUE_A, kt + 2, 0);
            G3_MMA(1, 1, true, false, G3_MID(G3_ISSUE_A, kt + 2 < nk, kt + 2, 0));
            // phase 3: quadrant (1,0)
            G3_READ_B(sl, 0);
            if (kt + 2 < nk) {
                G3_ISSUE2(G3_ISSUE_B, kt + 2, 1);
                // in flight behind the wait: A0(t+2) (2 pieces) + B1(t+2) (both pieces, or the first one when split)
                if constexpr (SPLIT) { asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); } else { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            G3_MMA(1, 0, false, true, G3_MID(G3_ISSUE_B, kt + 2 < nk, kt + 2, 1));
#undef G3_ISSUE2
#undef G3_MID
        }
        if (wr == 0) __builtin_amdgcn_s_barrier();
        G3_STAMP(2);                                                 // [2] K loop
        // all LDS reads of this item are complete: both slots are free for the next item's first pieces
        const int em0 = m0, en0 = n0;
        const float* ebias = reinterpret_cast<const float*>(smem + 2 * SLOT + bslot * 1024) + wc * GSB;
        const long ecoff = z1 * p.sC1 + z2 * p.sC2 + zs * p.sCz;
        const int eL = L;
        L += gridDim.x;
        const bool more = L < total;
        if (more) {
            item(L);
            G3_PROLOGUE();
        }
        G3_STAMP(3);                                                 // [3] next item's addressing + DMA issue
        const bool etail = p.tail_split > 1 && eL >= p.n_full;       // the finished item was one K range of a tail tile
        bool epf = false, epf3 = false;                              // the item took a prefetching epilogue: 16 loads + 16 stores (+ 4 column-sum stores, or + 8 row-sum stores) per wave
        if (etail) {
            // a tail item dumps its accumulators in register order (1 KiB per wave instruction); tail_finish_kernel knows the layout
            int lq = lane; asm volatile("" : "+v"(lq));                     // (not hoisted across the K loop: see epilogue_pf)
            float* dst = p.tail_slabs + (long)(eL - p.n_full) * (256 * BNT) + wave * (8 * TNW * 256) + lq * 4;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < TNW; ++j) *reinterpret_cast<f32x4*>(dst + (i * TNW + j) * 256) = acc[i][j] * p.alpha;
        } else if (!(p.flags & 128)) {                                 // bit 7: profiling build without the epilogue
            float* cs_row = p.cs_ws ? p.cs_ws + (long)((em0 >> 8) * 2 + wr) * p.N : nullptr;
            if constexpr (A_KC && B_KC && TNW == 4) {
                const int rmw = p.flags & (PB_GEMM_ACCUM | PB_GEMM_C_F32 | PB_GEMM_GELU | PB_GEMM_MUL_GELU_GRAD | PB_GEMM_ROWDOT);
                const bool inner = em0 + 256 <= p.M && en0 + BNT <= p.N;
                if (inner && rmw == PB_GEMM_MUL_GELU_GRAD && cs_row) { epilogue_pf<1, true>(p, acc, em0 + wr * 128, en0 + wc * GSB, ecoff, lane, ebias, cs_row, lds0, wave); epf = true; }
                else if (inner && rmw == PB_GEMM_MUL_GELU_GRAD) { epilogue_pf<1, false>(p, acc, em0 + wr * 128, en0 + wc * GSB, ecoff, lane, ebias, nullptr, lds0, wave); epf = true; }
                else if (inner && rmw == PB_GEMM_ACCUM && !cs_row) { epilogue_pf<2, false>(p, acc, em0 + wr * 128, en0 + wc * GSB, ecoff, lane, ebias, nullptr, lds0, wave); epf = true; }
                else if (inner && rmw == 0 && !cs_row && !(p.flags & PB_GEMM_REG_EPILOGUE)) { epilogue_plain(p, acc, em0 + wr * 128, en0 + wc * GSB, ecoff, lane, ebias, lds0, wave); }   // 16 stores (PB_GEMM_REG_EPILOGUE: the register path, for A/B runs)
                else if (inner && rmw == PB_GEMM_GELU && p.aux_out) { epilogue_gelu(p, acc, em0 + wr * 128, en0 + wc * GSB, ecoff, lane, ebias, lds0, wave); }   // 32 stores: `pend` below counts them
                else if (rmw == PB_GEMM_ROWDOT) { epilogue_pf<3, false>(p, acc, em0 + wr * 128, en0 + wc * GSB, ecoff, lane, ebias, nullptr, lds0, wave); epf3 = true; }   // the host admits whole tiles only
                else epilogue_regs<8, TNW>(p, acc, em0 + wr * 128, en0 + wc * GSB, ecoff, lane, ebias, cs_row);
            } else {
                const bool inner = em0 + 256 <= p.M && en0 + BNT <= p.N;
                if (TNW == 4 && inner && (p.flags & (PB_GEMM_ACCUM | PB_GEMM_C_F32 | PB_GEMM_GELU | PB_GEMM_MUL_GELU_GRAD | PB_GEMM_ROWDOT | PB_GEMM_REG_EPILOGUE)) == PB_GEMM_C_F32 && !p.bias && !cs_row)
                    epilogue_f32_plain(p, acc, em0 + wr * 128, en0 + wc * GSB, ecoff, lane, lds0, wave);        // split-K slabs: 32 stores (`pend`)
                else
                    epilogue_regs<8, TNW>(p, acc, em0 + wr * 128, en0 + wc * GSB, ecoff, lane, ebias, cs_row);
            }
        }
        G3_STAMP(4);                                                 // [4] epilogue: arithmetic + store (and load) issue
#ifdef PB_G3_STAMPS
        ++g3_items;
#endif
        if (!more) break;
        if (etail) {
            pend = 8 * TNW == 32 ? 32 : 0;                            // 8 x TNW plain 16-byte stores per wave
        } else {   // an interior tile without read-modify-write issues exactly 8 x 2 (x 2 for f32 C or the GELU pair) stores per wave
            const bool interior = em0 + 256 <= p.M && en0 + BNT <= p.N;
            const bool plain = !(p.flags & (PB_GEMM_ACCUM | PB_GEMM_MUL_GELU_GRAD | 128));
            pend = (interior && plain) ? (((p.flags & PB_GEMM_C_F32) || (p.flags & PB_GEMM_GELU)) ? 32 : 16) : 0;
            if (((p.flags & PB_GEMM_C_F32) && (p.flags & PB_GEMM_GELU)) || p.cs_ws) pend = 0;
            if (epf) pend = p.cs_ws ? 36 : 32;
            if (epf3) pend = 40;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < TNW; ++j) { acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; if constexpr (KS_TN) asm volatile("" : "+v"(acc[i][j])); }   // TN: opaque (no second copy of the pinned-register loop); NT: the first K-tile's MFMAs take C = 0 and the 128 v_mov go away
    }
#ifdef PB_G3_STAMPS
    if (p.stamps) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        G3_STAMP(5);                                                 // [5] drain: the last item's stores
        if (lane == 0 && blockIdx.x < 1024) {                            // one 16-word slot per wave, plain stores (same-address atomics would serialize the tail)
            unsigned* slot = reinterpret_cast<unsigned*>(p.stamps) + ((size_t)blockIdx.x * 8 + wave) * 16;
            for (int i = 0; i < 6; ++i) slot[i] = g3_acc[i];
            slot[6] = g3_items; slot[7] = 1u; slot[8] = (unsigned)__builtin_amdgcn_s_memtime() - g3_t0;
            slot[9] = (unsigned)__builtin_amdgcn_s_memrealtime() - g3_r0;
        }
    }
#endif
#undef G3_PROLOGUE
#undef G3S_ISSUE_A
#undef G3S_ISSUE_B
#undef G3_OFF_A
#undef G3_OFF_B
#undef G3_ISSUE_A
#undef G3_ISSUE_B
#undef G3_ISSUE_A1
#undef G3_ISSUE_B1
#undef G3_READ_A
#undef G3_READ_B
#undef G3_MMA
}

__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ slabs, int nsplit, long n, float* __restrict__ out, int accum) {
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        f32x4 s = *reinterpret_cast<const f32x4*>(slabs + 4 * i);
        if (accum) s += *reinterpret_cast<const f32x4*>(out + 4 * i);
        for (int k = 1; k < nsplit; ++k) s += *reinterpret_cast<const f32x4*>(slabs + (long)k * n + 4 * i);
        *reinterpret_cast<f32x4*>(out + 4 * i) = s;
    }
}

// Sums the tail_split partial tiles of every tail tile and finishes them like epilogue_regs would have: alpha is already in the
// partials; + bias, (+= C), store bf16 / f32. One workgroup per tail tile. The partials lie in the register order of gemm3_kernel
// (wave, i, j, lane): float4 number q of a tile belongs to wave w = q / (8 TNW 64), row tile i, column tile j, lane l, i.e. to
// C[wr*128 + i*16 + (l & 15)][wc*16*TNW + j*16 + (l >> 4)*4 .. +3] with wr = w >> 2, wc = w & 3.
constexpr int TAIL_FIN_PARTS = 16;                                  // workgroups per tail tile (a tile is 64 K float4s)
template <int TNW>
__global__ __launch_bounds__(256) void tail_finish_kernel(const Gemm2Args p) {
    constexpr int BNT = 64 * TNW, PER = 8 * 8 * TNW * 64;             // float4s per tile
    const int t_ = blockIdx.x / TAIL_FIN_PARTS, part = blockIdx.x % TAIL_FIN_PARTS, tile = p.n_full + t_;
    int m0, n0, zs;
    block_tile<256, BNT>(p, tile, m0, n0, zs);
    const float* slab = p.tail_slabs + (long)t_ * p.tail_split * (256 * BNT);
    const bool accum = p.flags & PB_GEMM_ACCUM, c32 = p.flags & PB_GEMM_C_F32;
    for (int q = part * (PER / TAIL_FIN_PARTS) + threadIdx.x; q < (part + 1) * (PER / TAIL_FIN_PARTS); q += 256) {
        const int l = q & 63, ij = (q >> 6) % (8 * TNW), w = q / (64 * 8 * TNW);
        const int row = m0 + (w >> 2) * 128 + (ij / TNW) * 16 + (l & 15), col = n0 + (w & 3) * (16 * TNW) + (ij % TNW) * 16 + (l >> 4) * 4;
        if (row >= p.M || col >= p.N) continue;
        f32x4 v = *reinterpret_cast<const f32x4*>(slab + 4 * q);
        for (int k = 1; k < p.tail_split; ++k) v += *reinterpret_cast<const f32x4*>(slab + (long)k * (256 * BNT) + 4 * q);
        if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + col);
        const long ci = (long)row * p.ldc + col;
        if (c32) {
            float* C = reinterpret_cast<float*>(p.C) + ci;
            if (accum) v += *reinterpret_cast<const f32x4*>(C);
            *reinterpret_cast<f32x4*>(C) = v;
        } else {
            bf16_t* C = reinterpret_cast<bf16_t*>(p.C) + ci;
            if (accum) v += load4(C);
            store4(C, v);
        }
    }
}

}  // namespace

// CU count of the current device, rounded down to a multiple of 8 (one persistent workgroup per CU; a grid that is a multiple
// of 8 keeps every work item of a workgroup on the workgroup's own XCD chunk).
#include <atomic>
static std::atomic<int> g_reserved_cus{0};
static int pb_num_cus(bool leave_reserved = false) {
    static int cached = 0;
    if (!cached) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
        cached = n / 8 * 8;
    }
    const int r = leave_reserved ? g_reserved_cus.load(std::memory_order_relaxed) : 0;
    return std::max(8, (cached - r) / 8 * 8);
}
// CUs the persistent GEMM grids launched with PB_GEMM_LEAVE_CUS leave alone (the count is process-wide): a data-parallel job hands RCCL's resident kernels their share of the chip
// up front, so that the one-workgroup-per-CU grids keep their form (the next item's loads under the current epilogue) instead of
// falling back to ordinary grids. Rounded to whole XCD rows of 8; 0 = the whole chip.
extern "C" int pb_gemm_reserve_cus(int32_t n) {
    PB_REQUIRE(n >= 0 && n <= 128, "pb_gemm_reserve_cus: %d", n);
    g_reserved_cus.store((n + 7) / 8 * 8, std::memory_order_relaxed);
    return 0;
}

// Called by pb_gemm (pb_gemm.hip) when the problem qualifies. Returns 1 if it declined, 0 on success, <0 on error.
// Workspace of the tail split: one buffer per stream that ever ran one (the engine runs GEMMs on two streams at a time), sized for
// a full round of 256 x 256 f32 partial tiles. Never freed (process lifetime, like the zero page of the attention kernels).
#include <mutex>
#include <vector>
static float* tail_slabs_for(hipStream_t stream, size_t floats) {
    static std::mutex mu;
    static std::vector<std::pair<std::pair<int, hipStream_t>, std::pair<float*, size_t>>> pool;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    for (auto& e : pool)
        if (e.first.first == dev && e.first.second == stream && e.second.second >= floats) return e.second.first;
    if (pool.size() >= 16) return nullptr;
    float* ptr = nullptr;
    if (hipMalloc(&ptr, floats * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    pool.push_back({{dev, stream}, {ptr, floats}});
    return ptr;
}
// Profiling aid (PB_GEMM_LDS_TAG=1, tools/profile_round.sh): a kernel trace names the kernel, not its problem, and the persistent grid is 256
// workgroups whatever the shape. With the tag on, a launch asks for 16 x tag bytes of dynamic LDS it never touches (one workgroup per CU
// either way), and the trace's group_segment_size column then tells the step's GEMMs apart: tag = 64 nclass + 8 kclass + epilogue class,
// decoded by tools/rocpd_stats.py (classes: the step's N and K values; for the TN weight gradients, whose K is the row count, M takes K's slot).
static int gemm_lds_tag(const pb_gemm_desc* d) {
    static const int on = getenv("PB_GEMM_LDS_TAG") ? atoi(getenv("PB_GEMM_LDS_TAG")) : 0;
    if (!on) return 0;
    auto cls = [](long v) { return v == 768 ? 1 : v == 1280 ? 2 : v == 1536 ? 3 : v == 2304 ? 4 : v == 3072 ? 5 : v == 18432 ? 6 : v >= 4096 ? 7 : 0; };
    const int e = (d->flags & PB_GEMM_GELU) ? 1 : (d->flags & PB_GEMM_MUL_GELU_GRAD) ? 2 : (d->flags & PB_GEMM_ROWDOT) ? 4 : (d->flags & PB_GEMM_ACCUM) ? 3 : d->bias ? 5 : 0;
    const bool tn = !d->a_kcontig && !d->b_kcontig;                          // weight gradients: K is the row count; the slot carries M instead
    return 64 * cls(d->N) + 8 * cls(tn ? d->M : d->K) + e;
}
int pb_gemm2_try(const pb_gemm_desc* d, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (d->dtype != PB_BF16) return 1;
    const int nsplit = d->splitk > 1 ? d->splitk : 1;
    if (d->K <= 0 || d->K % BK != 0) return 1;
    if (d->N % 8 != 0) return 1;
    const bool a_kc = d->a_kcontig, b_kc = d->b_kcontig;
    auto al = [](const void* p, long ld, long s1, long s2) { return ((uintptr_t)p % 16 == 0) && ld % 8 == 0 && s1 % 8 == 0 && s2 % 8 == 0; };
    if (!al(d->A, d->lda, d->sA1, d->sA2) || !al(d->B, d->ldb, d->sB1, d->sB2)) return 1;
    if (!a_kc && d->M % 8 != 0) return 1;
    if (!a_kc && d->M < 8) return 1;
    if (!b_kc && d->N < 8) return 1;
    const bool c32 = d->flags & PB_GEMM_C_F32;
    const int cal = c32 ? 4 : 8;                                    // 16-byte output row segments
    if ((uintptr_t)d->C % 16 != 0 || d->ldc % cal != 0 || d->sC1 % cal != 0 || d->sC2 % cal != 0) return 1;
    if ((d->aux_in || d->aux_out) && (d->ldaux % 8 != 0 || (uintptr_t)d->aux_in % 16 != 0 || (uintptr_t)d->aux_out % 16 != 0)) return 1;
    if (d->bias && ((uintptr_t)d->bias % 16 != 0)) return 1;
    if (nsplit > 1 && (!c32 || !d->slabs || d->bias || (d->flags & ~(PB_GEMM_ACCUM | PB_GEMM_C_F32 | PB_GEMM_TILE128 | PB_GEMM_TILE256 | 128 | PB_GEMM_REG_EPILOGUE | 2048 | 4096 | 8192 | 16384 | 32768 | PB_GEMM_LEAVE_CUS)))) {
        pb_set_error("pb_gemm: split-K needs f32 C, a slab workspace and no epilogue other than accumulate");
        return -2;
    }
    Gemm2Args a;
    a.A = (const bf16_t*)d->A; a.B = (const bf16_t*)d->B;
    a.C = nsplit > 1 ? d->slabs : d->C;
    a.bias = d->bias; a.aux_in = (const bf16_t*)d->aux_in; a.aux_out = (bf16_t*)d->aux_out;
    a.M = d->M; a.N = d->N; a.K = d->K; a.Kc = ((d->K / BK + nsplit - 1) / nsplit) * BK;
    a.lda = d->lda; a.ldb = d->ldb; a.ldc = nsplit > 1 ? d->N : d->ldc; a.ldaux = d->ldaux;
    const int nb1 = d->nb1 > 0 ? d->nb1 : 1;
    a.nb2 = d->nb2 > 0 ? d->nb2 : 1;
    if (nsplit > 1 && nb1 * a.nb2 != 1) { pb_set_error("pb_gemm: split-K with batches is not supported"); return -2; }
    a.sA1 = d->sA1; a.sA2 = d->sA2; a.sB1 = d->sB1; a.sB2 = d->sB2; a.sC1 = d->sC1; a.sC2 = d->sC2;
    a.sCz = (long)d->M * d->N;
    a.alpha = d->alpha; a.cs_ws = nullptr;
    a.rowdot = nullptr; a.ld_rowdot = 0;
    if (d->flags & PB_GEMM_ROWDOT) {
        // only the form the attention backward asks for: whole 256 x 256 tiles of the NT ping-pong kernel, bf16 C, an operand tile to
        // multiply with, no other epilogue -- anything else is refused loudly (the caller then runs the row-sum pass itself)
        if (!(a_kc && b_kc) || nsplit != 1 || d->M % 256 || d->N % 256 || c32 || !d->aux_in || !d->rowdot_out || d->ld_rowdot < d->M || d->colsum_out ||
            (d->flags & (PB_GEMM_ACCUM | PB_GEMM_GELU | PB_GEMM_MUL_GELU_GRAD | PB_GEMM_TILE128 | 2048)) || (d->nb1 > 1) || (d->nb2 > 1) || d->M < 256) {
            pb_set_error("pb_gemm: PB_GEMM_ROWDOT needs the NT layout, M and N multiples of 256, bf16 C, aux_in, rowdot_out / ld_rowdot and no other epilogue");
            return -2;
        }
        a.rowdot = d->rowdot_out; a.ld_rowdot = d->ld_rowdot;
    }
#ifdef PB_G3_STAMPS
    a.stamps = nullptr;
    if (const char* e = getenv("PB_G3_STAMP_PTR")) a.stamps = (unsigned long long*)strtoull(e, nullptr, 0);
#endif
    a.flags = nsplit > 1 ? (d->flags & ~PB_GEMM_ACCUM) : d->flags;          // split-K: the slabs are overwritten, the accumulation into C happens in the reduce

    // Tile / kernel choice, from same-process A/B runs of every cfg-2 shape (tools/gemm_ab.py, T = 32768 tokens):
    // the 256x256 ping-pong kernel wherever the output is at least 512 wide -- NT fc1 910 vs 750 TF (128x128), fc2 1110 vs 1050,
    // NN dfc1 937 vs 899, TN w1 930 vs 820 (one-barrier 256x256) -- and 128x128 tiles (2 workgroups per CU) below that and for
    // the small split-K wgrads (768 x 768: 692 vs 620 TF). TN callers pass PB_GEMM_TILE256 together with their split-K factor.
    const bool big = !(d->flags & PB_GEMM_TILE128) && d->M >= 256 && d->N >= 256 &&
                     ((d->flags & (PB_GEMM_TILE256 | PB_GEMM_ROWDOT)) || (nsplit == 1 && d->M >= 2048 && d->N >= 512));
    // (A 256 x 192 instantiation -- 512 tiles = 2 full rounds at N = 768, T = 32768 -- measured +4-6 % back to back and -1.5 ms on the whole
    // step in round 2: every A row panel is then streamed by 4 column tiles instead of 3. It lost its A/B and was removed in round 3.)
    constexpr bool wide192 = false;
    // the ping-pong kernel is instantiated for the two layouts the step uses (NT: both operands K-contiguous; TN: neither). The mixed
    // layouts (NN dgrad without the transposed weight copies) take the one-barrier 256 x 256 kernel: their ping-pong
    // instantiations kept two VGPRs in scratch around the K loop (code-object metadata) and are not built any more.
    const bool pingpong = big && !(d->flags & 2048) && a_kc == b_kc;
    if ((d->flags & PB_GEMM_ROWDOT) && !pingpong) { pb_set_error("pb_gemm: PB_GEMM_ROWDOT is implemented by the 256 x 256 ping-pong kernel only"); return -2; }
    const int BMs = big ? 256 : 128, BNs = big ? (wide192 ? 192 : 256) : 128;
    a.tiles_m = (d->M + BMs - 1) / BMs; a.tiles_n = (d->N + BNs - 1) / BNs;
    a.nsplit = nsplit;
    dim3 grid(a.tiles_m * a.tiles_n * nsplit, nb1 * a.nb2, 1);
#define PB_G2_LAUNCH(AK, BK_, WM_, WN_, TM_, TN_)                                                                       \
    do {                                                                                                                 \
        auto kfn = gemm2_kernel<AK, BK_, WM_, WN_, TM_, TN_>;                                                              \
        const size_t lds = 2 * (size_t)(16 * WM_ * TM_ + 16 * WN_ * TN_) * 128 + 16 * (size_t)lds_tag;                     \
        if (lds > 65536) hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(kfn, grid, dim3(WM_ * WN_ * 64), lds, stream, a);                                                \
    } while (0)
    bool cs_fused = false;
    const int lds_tag = gemm_lds_tag(d);
    a.n_full = 0; a.tail_split = 1; a.tail_kc = 0; a.tail_slabs = nullptr;
    if (pingpong && (d->flags & PB_GEMM_TAIL_SPLIT) && !(d->flags & PB_GEMM_ROWDOT) && !wide192 && nsplit == 1 && nb1 * a.nb2 == 1 && !d->colsum_out &&
        !(d->flags & (PB_GEMM_GELU | PB_GEMM_MUL_GELU_GRAD | 128))) {
        // The persistent grid runs ceil(tiles / CUs) rounds; a last round that fills only part of the chip (N = 768: 312 tiles =
        // 1.22 rounds at 26 624 rows, 384 = 1.5 at 32 768) costs a whole one. Cut those tiles' K range so that they occupy the CUs
        // that would idle: the round then lasts 1 / split of a tile plus the trip of the f32 partials through the slabs.
        const int ncu = pb_num_cus(), ntile = a.tiles_m * a.tiles_n, rem = ntile % ncu, nkt = d->K / BK;
        if (rem > 0) {
            // Cost model in units of one K step (64) of a 256 x 256 tile, ~1.6 us in the step (same-process A/B of the N = 768 shapes,
            // tools/gemm_tail_ab.py): a split saves nkt - ceil(nkt / split) steps of the last round and costs the finishing launch
            // (~5 us) plus the trip of rem x split f32 tiles (256 KiB each) out to the slabs and back at ~5 TB/s. Measured at
            // 26 624 rows: K = 3072 154 -> 134 us, K = 2304 128 -> 107 us; K = 768 loses (46 -> 52 us) and so does a half-full round
            // (32 768 rows, 138 -> 136 us): the model turns those down. Only on request (PB_GEMM_TAIL_SPLIT): in the backward pass of the
            // training step the second stream's weight-gradient GEMMs already fill the CUs a short last round leaves idle (allowed
            // everywhere, the same-box A/B of the whole step showed no gain: 62.4 vs 62.6 ms); the forward projections, which run
            // alone, ask for it.
            const int smax = std::min(std::min(ncu / rem, nkt / 3), 8);
            int best = 1; float best_net = 0.f;
            for (int sp = 2; sp <= smax; ++sp) {
                const int per = (nkt + sp - 1) / sp;
                const float overhead = (5.f + 0.105f * rem * sp) / 1.6f;
                const float net = (float)(nkt - per) - 1.3f * overhead;
                if (net > best_net) { best = sp; best_net = net; }
            }
            if (best >= 2) {
                const int per = (nkt + best - 1) / best;
                float* slabs = tail_slabs_for(stream, (size_t)ncu * 256 * 256);
                if (slabs) { a.n_full = ntile - rem; a.tail_split = (nkt + per - 1) / per; a.tail_kc = per * BK; a.tail_slabs = slabs; }
            }
        }
    }
    // Row split (round 3; on request: PB_GEMM_ROW_SPLIT). A partly filled last round of the persistent grid costs a whole tile time: N = 768 at 26 624 rows is 312
    // tiles = 1.22 rounds of 256 CUs, paid as 2. When the K-range split above did not take the case (it moves f32 partials through
    // HBM and loses at small K), the M tiles of the FULL rounds stay with the 256 x 256 kernel and the remaining rows go to a second
    // launch of the 128 x 128 kernel (2 workgroups per CU, every tile resident at once): both write C directly, no partials, no
    // finishing pass. Model in us, from the in-step timings of the N = 768 shapes (profiles/r03_*): a 256 x 256 round costs
    // 1.53 nk + 6.5, a round of the 128 x 128 kernel 0.75 nk + 4 (one workgroup per CU) or 1.0 nk + 4 (two), a kernel boundary 2.
    Gemm2Args rest;
    bool row_split = false;
    if (pingpong && a.tail_split == 1 && (d->flags & PB_GEMM_ROW_SPLIT) && !(d->flags & (4096 | 128 | PB_GEMM_ROWDOT)) && !wide192 && nsplit == 1 && nb1 * a.nb2 == 1 &&
        !d->colsum_out && d->M % 8 == 0) {
        const int ncu = pb_num_cus(), ntile = a.tiles_m * a.tiles_n, rounds = ntile / ncu, rem = ntile % ncu, nkt = d->K / BK;
        if (rounds >= 1 && rem > 0) {
            const int m_main = (rounds * ncu / a.tiles_n) * 256;                      // rows of the M tiles that fill whole rounds
            const int m_rest = d->M - m_main;
            const long t2 = (long)((m_rest + 127) / 128) * ((d->N + 127) / 128);
            const float r256 = 1.53f * nkt + 6.5f;
            const float r128 = (t2 <= ncu ? 0.75f : 1.0f) * nkt + 4.f;
            const float now = (rounds + 1) * r256;
            const float then = rounds * r256 + (float)((t2 + 2 * ncu - 1) / (2 * ncu)) * r128 + 2.f;
            if (m_main >= 256 && m_rest > 0 && then < 0.92f * now) {
                row_split = true;
                rest = a;
                rest.M = m_rest;
                rest.A = a_kc ? a.A + (long)m_main * a.lda : a.A + m_main;
                rest.C = c32 ? (void*)((float*)a.C + (long)m_main * a.ldc) : (void*)((bf16_t*)a.C + (long)m_main * a.ldc);
                if (a.aux_in) rest.aux_in = a.aux_in + (long)m_main * a.ldaux;
                if (a.aux_out) rest.aux_out = a.aux_out + (long)m_main * a.ldaux;
                rest.tiles_m = (m_rest + 127) / 128; rest.tiles_n = (d->N + 127) / 128;
                a.M = m_main; a.tiles_m = m_main / 256;
                grid = dim3(a.tiles_m * a.tiles_n, 1, 1);
            }
        }
    }
    if (pingpong) {
        if (d->colsum_out && nsplit == 1 && nb1 * a.nb2 == 1) {
            float* slice = pb_defer_alloc((size_t)2 * a.tiles_m * d->N);          // deferred reduction: the partial rows must outlive this call
            a.cs_ws = slice ? slice : d->colsum_ws;
            cs_fused = true;
        }                          // bit 11: A/B against the one-barrier 256x256 kernel; bit 12: ordinary (non-persistent) grid
#define PB_G3_LAUNCH(AK, BK_)                                                                                              \
    do {                                                                                                                 \
        auto kfn = gemm3_kernel<AK, BK_, 4>;                                                                               \
        hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, G3_STAGE_OFF + G3_STAGE_BYTES + 8192); \
        const unsigned items = a.tail_split > 1 ? a.n_full + (grid.x - a.n_full) * a.tail_split : grid.x;                    \
        dim3 pgrid(std::min<unsigned>(items, (d->flags & 4096) ? items : (unsigned)pb_num_cus((d->flags & PB_GEMM_LEAVE_CUS) != 0)), grid.y, 1); \
        hipLaunchKernelGGL(kfn, pgrid, dim3(512), G3_STAGE_OFF + G3_STAGE_BYTES + 16 * (size_t)lds_tag, stream, a);                       \
        if (a.tail_split > 1) hipLaunchKernelGGL(tail_finish_kernel<4>, dim3((grid.x - a.n_full) * TAIL_FIN_PARTS), dim3(256), 0, stream, a); \
    } while (0)
        if (a_kc) PB_G3_LAUNCH(true, true);
        else PB_G3_LAUNCH(false, false);
#undef PB_G3_LAUNCH
    } else if (big) {
        if (a_kc && b_kc) PB_G2_LAUNCH(true, true, 2, 4, 8, 4);
        else if (a_kc && !b_kc) PB_G2_LAUNCH(true, false, 2, 4, 8, 4);
        else if (!a_kc && b_kc) PB_G2_LAUNCH(false, true, 2, 4, 8, 4);
        else PB_G2_LAUNCH(false, false, 2, 4, 8, 4);
    } else {
        if (a_kc && b_kc) PB_G2_LAUNCH(true, true, 2, 2, 4, 4);
        else if (a_kc && !b_kc) PB_G2_LAUNCH(true, false, 2, 2, 4, 4);
        else if (!a_kc && b_kc) PB_G2_LAUNCH(false, true, 2, 2, 4, 4);
        else PB_G2_LAUNCH(false, false, 2, 2, 4, 4);
    }
    if (row_split) {
        const Gemm2Args main_args = a;
        a = rest;
        grid = dim3(a.tiles_m * a.tiles_n, 1, 1);
        if (a_kc && b_kc) PB_G2_LAUNCH(true, true, 2, 2, 4, 4);
        else if (a_kc && !b_kc) PB_G2_LAUNCH(true, false, 2, 2, 4, 4);
        else if (!a_kc && b_kc) PB_G2_LAUNCH(false, true, 2, 2, 4, 4);
        else PB_G2_LAUNCH(false, false, 2, 2, 4, 4);
        a = main_args;
    }
#undef PB_G2_LAUNCH
    if (hipGetLastError() != hipSuccess) { pb_set_error("pb_gemm2 launch failed"); return -1; }
    if (cs_fused && pb_finalize_rows(a.cs_ws, 2 * a.tiles_m, d->N, d->colsum_out, stream)) return -1;
    if (nsplit > 1) {
        if (d->ldc != d->N) { pb_set_error("pb_gemm: split-K needs a dense C (ldc == N)"); return -2; }
        const long n = (long)d->M * d->N;
        const int g = (int)std::max(1L, std::min(2048L, (n / 4 + 255) / 256));
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3(g), dim3(256), 0, stream, (const float*)d->slabs, nsplit, n, (float*)d->C, (int)((d->flags & PB_GEMM_ACCUM) != 0));
        if (hipGetLastError() != hipSuccess) { pb_set_error("pb_reduce_slabs launch failed"); return -1; }
    }
    return (d->colsum_out && !cs_fused) ? 2 : 0;
}
