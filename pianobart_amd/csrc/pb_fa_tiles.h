// Tile, fragment and pipeline helpers shared by the head_dim 64 / 96 / 128 attention kernels (pb_flash64.hip, pb_flash1.hip):
// the [64 rows][128 B] LDS image with its one swizzle for row reads and transposed reads, the LDS-DMA staging, the asm
// ds_read_b64_tr_b16 fragments behind one wait statement, counted vmcnt waits, and the (batch, head) -> XCD block map.
#pragma once
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
    float* cs_q; float* cs_kv;        // bias-gradient partials (column sums of dQ | of dK, dV), or NULL
    const bf16_t* zeros;              // >= 16 bytes of zeros (source of the column chunks beyond head_dim in a partly filled image)
    // packed rows ("varlen", pb_flash_*_packed): batch b's query rows are rows vl_q_off[b] .. + vl_q_len[b] - 1 of q / o / dout / dq (row
    // stride *_ss; the batch strides are not used), its key rows vl_k_off[b] .. + vl_k_len[b] - 1 of k / v / dk / dv, of which the
    // first kmax[b] are visible; Sq / Sk are the maxima over the batch (grid, LDS, and the row length of lse / delta). NULL: dense.
    const int *vl_q_off, *vl_k_off, *vl_q_len, *vl_k_len;
    // dispatch order of the (batch, head) pairs (B * H entries, pair = b * H + h), or NULL = natural order: the grid hands its slots to
    // pairs in THIS order, so a caller that knows the pairs' costs (packed rows: 4x spread of q_len * visible keys) lists them
    // longest first, eight at a time across the XCDs. Only the order of the work changes, never a result.
    const int* bh_order;
};

// Packed rows: give the kernel body the view of ONE batch -- its own Sq / Sk and base pointers rebased so that the dense address
// arithmetic (ptr + b * batch_stride + row * row_stride) lands on the batch's first packed row.
__device__ __forceinline__ void varlen_localize(Fa64Args& p, int b) {
    if (!p.vl_q_off) return;
    const long qo = p.vl_q_off[b], ko = p.vl_k_off[b];
    p.Sq = p.vl_q_len[b]; p.Sk = p.vl_k_len[b];
    p.q += qo * p.q_ss - b * p.q_sb; p.k += ko * p.k_ss - b * p.k_sb; p.v += ko * p.v_ss - b * p.v_sb;
    if (p.o) p.o += qo * p.o_ss - b * p.o_sb;
    if (p.out) p.out += qo * p.o_ss - b * p.o_sb;
    if (p.dout) p.dout += qo * p.o_ss - b * p.o_sb;
    if (p.dq) p.dq += qo * p.dq_ss - b * p.dq_sb;
    if (p.dk) p.dk += ko * p.dk_ss - b * p.dk_sb;
    if (p.dv) p.dv += ko * p.dv_ss - b * p.dv_sb;
}

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

// image [64 rows][128 B]; f(row): 8 distinct values over (row>>1)&7, even values over an aligned group of 8 rows
__device__ __forceinline__ int fsw(int row) { return (((row >> 1) & 3) << 1) | ((row >> 3) & 1); }

__device__ __forceinline__ void glds16(const bf16_t* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// stage rows r0..r0+63 (clamped to nrows-1) of a [*, 64] bf16 matrix with row stride ld: 8 KiB = 8 DMA pieces, 2 per wave.
// The per-lane part of the source address (row within the tile, swizzled chunk) is computed once per kernel (StageOff); a
// whole tile then costs one 64-bit add per piece on top of a wave-uniform tile base -- the row * ld multiplies of the naive
// form were 16 % of the forward's vector cycles. Only a ragged last tile takes the clamped path.
struct StageOff { unsigned off[2]; unsigned chunk[2]; };
__device__ __forceinline__ StageOff stage_off(long ld, int wave, int lane) {
    StageOff o;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int row = (wave * 2 + n) * 8 + (lane >> 3);
        o.chunk[n] = (unsigned)((lane & 7) ^ fsw(row));               // the 8-column source chunk this lane fetches
        o.off[n] = (unsigned)(row * ld + (o.chunk[n] << 3));
    }
    return o;
}
// cmax = number of valid 8-column chunks of this image (8, or 4 for the second image of head_dim 96): the lanes of the other
// chunks fetch zeros, so that the padded columns contribute nothing to any product.
__device__ __forceinline__ void stage64(const bf16_t* __restrict__ base, long ld, int r0, int nrows, char* lds, int wave, int lane, const StageOff& so,
                                        int cmax = 8, const bf16_t* zeros = nullptr) {
    if (r0 + 64 <= nrows) {
        const bf16_t* tb = base + (long)r0 * ld;                          // wave-uniform
#pragma unroll
        for (int n = 0; n < 2; ++n) glds16((cmax == 8 || (int)so.chunk[n] < cmax) ? tb + so.off[n] : zeros, lds + (wave * 2 + n) * 1024);
    } else {
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int inst = wave * 2 + n;
            const int row = inst * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ fsw(row);
            const int gr = min(r0 + row, nrows - 1);
            glds16((cmax == 8 || chunk < cmax) ? base + (long)gr * ld + chunk * 8 : zeros, lds + inst * 1024);
        }
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
// reductions over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48) with v_permlane16/32_swap: after swap(v, v) one
// of the two results is the lane's own value and the other its partner's, for either parity -- VALU only, where
// __shfl_xor goes through ds_bpermute (an LDS round trip on the softmax's critical path).
__device__ __forceinline__ float grp_max(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float grp_sum(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
template <bool V> struct BoolTag { static constexpr bool value = V; };

constexpr int STG = 2 * 8192 + 512;        // one stage: two 8 KiB images + 128 floats

// 1-D grid -> (row block, head, batch). Workgroups are dealt round-robin over the 8 XCDs (id % 8): all row blocks of one
// (batch, head) are given to ONE XCD, back to back, so the K/V (or Q/dO) tiles they all stream stay in that XCD's 4 MiB
// L2 (measured before: FETCH_SIZE 705 MB per forward launch = K/V re-fetched from beyond L2 by each of the 8 q-blocks).
__device__ __forceinline__ void block_map(int nrb, int H, int B, int& rb, int& h, int& b, const int* order = nullptr) {
    const int L = blockIdx.x, BH = H * B;
    int bh;
    // The row block is ROTATED by the head's position (round 6). An XCD hands consecutive workgroups to its four shader engines in a fixed pattern of
    // period 4 whether or not their CUs are free (workgroup trace of the one-pass backward, tools/flash1_stamps.py --gaps: slot % 4 -> one engine for half of
    // the workgroups), so with rb = slot % nrb a causal call gave one engine all the long row blocks and another all the short ones: the CUs of the latter idled
    // 16 us between two workgroups (25 % of the kernel). With the rotation every engine sees every row block equally often. Results do not depend on the mapping.
    if ((BH & 7) == 0) { const int x = L & 7, slot = L >> 3, j = slot / nrb; bh = j * 8 + x; rb = (slot - j * nrb + j) % nrb; }
    else { bh = L / nrb; rb = (L - bh * nrb + bh) % nrb; }
    if (order) bh = order[bh];
    h = bh % H; b = bh / H;
}

// ---- pieces shared by the three kernels' pipelines ------------------------------------------------------------------------
// Transposed fragments through inline asm (form (ii) of cdna_hip_programming.md 5.7: "=v" loads, one wait statement naming all
// of them): in front of the ds_read_tr BUILTIN hipcc puts s_waitcnt vmcnt(0) whenever an LDS-DMA is in flight, i.e. in the
// middle of every tile, which is exactly the prefetch these kernels live on (found in the .s; rocprof: 22 % MFMA busy).
template <int OFF>
__device__ __forceinline__ void ds_tr(s16x4& d, unsigned addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "i"(OFF));
}
// lane part of frag_tr's address for column block c0 (the row part 32 s + 16 hi is an immediate: fsw() ignores row bits 4, 5)
__device__ __forceinline__ unsigned tr_lane_off64(int c0, int lane) {
    const int lr = lane & 15, g = lane >> 4, qq = lr >> 2, pp = lr & 3;
    const int r0 = 4 * g + qq, chunk = (c0 >> 3) + (pp >> 1);
    return (unsigned)(r0 * 128 + ((chunk ^ fsw(r0)) << 4) + ((pp & 1) << 3));
}
// the four 8-byte reads of one 16-column block: d[s][0/1] = rows 32 s + 4 g + qq (+16)
__device__ __forceinline__ void ds_tr_block(s16x4 (&d)[2][2], unsigned addr) {
    ds_tr<0>(d[0][0], addr); ds_tr<2048>(d[0][1], addr); ds_tr<4096>(d[1][0], addr); ds_tr<6144>(d[1][1], addr);
}
__device__ __forceinline__ bf16x8 tr_join(s16x4 lo, s16x4 hi) {
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}
#define TRW4(X) "+v"(X[0][0]), "+v"(X[0][1]), "+v"(X[1][0]), "+v"(X[1][1])
__device__ __forceinline__ void tr_wait4(s16x4 (&t)[4][2][2]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : TRW4(t[0]), TRW4(t[1]), TRW4(t[2]), TRW4(t[3]));
}
__device__ __forceinline__ void tr_wait8(s16x4 (&t)[4][2][2], s16x4 (&u)[4][2][2]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : TRW4(t[0]), TRW4(t[1]), TRW4(t[2]), TRW4(t[3]), TRW4(u[0]), TRW4(u[1]), TRW4(u[2]), TRW4(u[3]));
}
#undef TRW4
__device__ __forceinline__ unsigned lds_u32(const void* p) { return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p; }

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "i"(N) : "memory"); }

// The three kernels are templates on head_dim HD in {64, 96, 128}: a [64 rows][HD] operand tile is kept as NB = ceil(HD / 64)
// images of [64][64] in the layout above (one swizzle, one set of fragment routines; the second image of head_dim 96 is half
// zeros), the QK^T / dP chains run over the HD / 32 k-steps and the outputs carry HD / 16 column blocks. head_dim 64: 3-deep
// DMA ring, 2 row tiles per wave everywhere; head_dim 96 / 128 (128 = the reference's CLI default, 1024 / 8 heads; 96 = 768 / 8
// heads): 2-deep ring (64 KiB of tiles per workgroup) and one key tile per wave in dKV.
template <int NB> struct FaCfg {
    static constexpr int NS = NB == 1 ? 3 : 2;            // ring depth
    static constexpr int STB = 2 * NB * 8192;             // bytes of one ring stage: NB images of each of the two operands
    static constexpr int PCS = 4 * NB;                    // DMA pieces per wave per stage
    static constexpr int KT = NB == 1 ? 2 : 1;            // key tiles (16 keys) per wave in the dK/dV kernel
};

// scale a bf16x8 fragment by c (operand prescale: S = (c K) Q^T comes out of the MFMA in log2 units)
__device__ __forceinline__ bf16x8 scale_frag(bf16x8 v, float c) {
    bf16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (bf16_t)((float)v[e] * c);
    return r;
}

}  // namespace
