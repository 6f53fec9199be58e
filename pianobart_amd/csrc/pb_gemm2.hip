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
//   * the epilogue goes through LDS so that global stores are 8/16-byte row segments.
// Tile 128x128x64, 256 threads (2x2 waves of 64x64 = 4x4 MFMA 16x16x32), 2 LDS stages of 32 KiB;
// per K tile: issue next tile's 8 DMA pieces -> 32 MFMAs on the current tile -> vmcnt(0) + barrier.
// Requirements (else pb_gemm falls back to pb_gemm.hip): K % 64 == 0 per split, 16-byte aligned rows,
// contiguous dims multiples of 8.
#include "pb_common.h"
#include "pb_api_internal.h"
#include <algorithm>

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
};

__device__ __forceinline__ int kswz(int row) { return ((row >> 1) & 7) ^ ((row >> 4) & 7); }
__device__ __forceinline__ int rswz(int krow) { return ((krow & 3) | ((krow >> 1) & 4)) << 1; }

__device__ __forceinline__ void glds16(const bf16_t* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
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

// WM x WN waves, each TM x TN MFMA tiles of 16x16: block tile BM = 16*WM*TM by BN = 16*WN*TN.
// Measured (tools/gemm_probe.py, MI355X): the 128x128 main loop is bound by the L2 -> LDS load path (~60 GB/s per CU,
// 64 FLOP per loaded byte -> ~1 PF ceiling), the 256x256 one reaches 1.1-1.2 PF; the output write (HBM write rate,
// ~3.1 TB/s) is NOT overlapped with the main loop. Tried and measured (tools/gemm_probe.py history): persistent tiles with
// stores left in flight, register epilogue with swapped MFMA operands, non-temporal stores, and the store traffic spread
// over the K loop of the same launch (hides only ~1/3 of it) -- none pays; the 8-phase counted-vmcnt pipeline is next.
template <bool A_KC, bool B_KC, int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(WM * WN * 64) void gemm2_kernel(const Gemm2Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = WM * WN, BM = 16 * WM * TM, BN = 16 * WN * TN;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE_BYTES = A_BYTES + B_BYTES;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int ntiles = p.tiles_m * p.tiles_n;
    // 1-D grid over (split, tile). XCD x (= id % 8) takes a CONTIGUOUS chunk of the (split-major, tile-minor) list, so with
    // split-K the workgroups of one XCD work on (nearly) one K slice: its A/B rows are fetched into that L2 once and
    // shared by all its tiles (wgrad measured ~2x its algorithmic bytes from beyond L2 with the tile-only remap).
    int bid, zs;
    {
        const int total = ntiles * p.nsplit, L = blockIdx.x;
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
    const int m0 = tm * BM, n0 = tn * BN;
    const int z = blockIdx.y, z1 = z / p.nb2, z2 = z % p.nb2;
    const bf16_t* A = p.A + z1 * p.sA1 + z2 * p.sA2;
    const bf16_t* B = p.B + z1 * p.sB1 + z2 * p.sB2;
    const long coff = z1 * p.sC1 + z2 * p.sC2 + zs * p.sCz;
    const int kbeg = zs * p.Kc, kend = min(p.K, kbeg + p.Kc);
    const int nk = (p.flags & 256) ? 0 : max(0, kend - kbeg) / BK;   // bit 8: profiling build of the epilogue alone

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
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();          // hipcc adds s_waitcnt vmcnt(0) here: next tile has landed, current one is free
    }
    if (p.flags & 128) return;                                      // bit 7: profiling build without the epilogue

    // ---- epilogue through LDS: per wave a 32 x (16 TN) f32 panel, TM/2 passes ----
    constexpr int PW = 16 * TN + 4;                                 // panel row stride (floats)
    constexpr int LPRW = (16 * TN) / 4;                             // lanes per panel row in the read-back phase
    constexpr int RPI = 64 / LPRW;                                  // rows per read-back iteration
    const int lr = lane & 15, lg = lane >> 4;
    const bool accum = p.flags & PB_GEMM_ACCUM, c32 = p.flags & PB_GEMM_C_F32;
    const bool do_gelu = p.flags & PB_GEMM_GELU, mul_gg = p.flags & PB_GEMM_MUL_GELU_GRAD;
    float* panel = reinterpret_cast<float*>(smem) + wave * (32 * PW);
    float* C32 = reinterpret_cast<float*>(p.C) + coff;
    bf16_t* CT = reinterpret_cast<bf16_t*>(p.C) + coff;
    const int pcol = (lane % LPRW) * 4;
    const int colb = n0 + wn * (16 * TN) + pcol;                   // this lane's 4 output columns in the read-back phase
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (p.bias && colb < p.N) bv = *reinterpret_cast<const f32x4*>(p.bias + colb);   // N % 4 == 0 checked on the host
#pragma unroll
    for (int pass = 0; pass < TM / 2; ++pass) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) panel[(ii * 16 + lg * 4 + r) * PW + j * 16 + lr] = acc[pass * 2 + ii][j][r];
        __builtin_amdgcn_s_waitcnt(0xc07f);                         // lgkmcnt(0): the wave's own panel writes are done
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 32 / RPI; ++it) {
            const int prow = it * RPI + lane / LPRW;
            const int row = m0 + wm * (16 * TM) + pass * 32 + prow;
            if (row < p.M && colb < p.N) {
                f32x4 v = *reinterpret_cast<const f32x4*>(panel + prow * PW + pcol) * p.alpha + bv;
                if (do_gelu) {
                    if (p.flags & 1024) store4_nt(p.aux_out + (long)row * p.ldaux + colb, v); else store4(p.aux_out + (long)row * p.ldaux + colb, v);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gelu_fast(v[e]);
                }
                if (mul_gg) {
                    const f32x4 u = load4(p.aux_in + (long)row * p.ldaux + colb);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= gelu_grad_fast(u[e]);
                }
                const long ci = (long)row * p.ldc + colb;
                if (c32) {
                    if (accum) v += load4(C32 + ci);
                    store4(C32 + ci, v);
                } else {
                    if (accum) v += load4(CT + ci);
                    if (p.flags & 1024) store4_nt(CT + ci, v); else store4(CT + ci, v);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ slabs, int nsplit, long n, float* __restrict__ out) {
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        f32x4 s = *reinterpret_cast<const f32x4*>(slabs + 4 * i);
        for (int k = 1; k < nsplit; ++k) s += *reinterpret_cast<const f32x4*>(slabs + (long)k * n + 4 * i);
        *reinterpret_cast<f32x4*>(out + 4 * i) = s;
    }
}

}  // namespace

// Called by pb_gemm (pb_gemm.hip) when the problem qualifies. Returns 1 if it declined, 0 on success, <0 on error.
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
    if ((uintptr_t)d->C % 16 != 0 || d->ldc % 4 != 0 || d->sC1 % 4 != 0 || d->sC2 % 4 != 0) return 1;
    if ((d->aux_in || d->aux_out) && (d->ldaux % 4 != 0)) return 1;
    if (d->bias && ((uintptr_t)d->bias % 16 != 0)) return 1;
    if (nsplit > 1 && (!c32 || !d->slabs || d->bias || (d->flags & ~(PB_GEMM_C_F32 | PB_GEMM_TILE128 | PB_GEMM_TILE256 | 128 | 256 | 1024)))) {
        pb_set_error("pb_gemm: split-K needs f32 C, a slab workspace and no epilogue");
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
    a.alpha = d->alpha; a.flags = d->flags;
    // tile choice: 128x128 (4 waves, 2 blocks/CU) by default; 256x256 (8 waves, 128x64 per wave, half the L2->LDS
    // traffic per FLOP) when asked for (PB_GEMM_TILE256: the split-K wgrad GEMMs, whose output phase is negligible)
    // 256x256 when asked for, and by default for wide outputs (N >= 1536: fc1, du, qkv): measured 842 vs 731 TF at the fc1 shape
    const bool big = !(d->flags & PB_GEMM_TILE128) && d->M >= 256 && d->N >= 256 && ((d->flags & PB_GEMM_TILE256) || (d->N >= 1536 && d->M >= 2048 && nsplit == 1));
    const bool tall = !big && (d->flags & 512) && d->M >= 1024;   // measured: no gain over 128x128 (tools/gemm_bench.py), kept for experiments      // 256x128: 8 waves of 64x64, 25% less L2->LDS traffic per FLOP
    const int BMs = (big || tall) ? 256 : 128, BNs = big ? 256 : 128;
    a.tiles_m = (d->M + BMs - 1) / BMs; a.tiles_n = (d->N + BNs - 1) / BNs;
    a.nsplit = nsplit;
    dim3 grid(a.tiles_m * a.tiles_n * nsplit, nb1 * a.nb2, 1);
#define PB_G2_LAUNCH(AK, BK_, WM_, WN_, TM_, TN_)                                                                       \
    do {                                                                                                                 \
        auto kfn = gemm2_kernel<AK, BK_, WM_, WN_, TM_, TN_>;                                                              \
        const size_t lds = 2 * (size_t)(16 * WM_ * TM_ + 16 * WN_ * TN_) * 128;                                            \
        if (lds > 65536) hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(kfn, grid, dim3(WM_ * WN_ * 64), lds, stream, a);                                                \
    } while (0)
    if (tall) {
        if (a_kc && b_kc) PB_G2_LAUNCH(true, true, 4, 2, 4, 4);
        else if (a_kc && !b_kc) PB_G2_LAUNCH(true, false, 4, 2, 4, 4);
        else if (!a_kc && b_kc) PB_G2_LAUNCH(false, true, 4, 2, 4, 4);
        else PB_G2_LAUNCH(false, false, 4, 2, 4, 4);
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
#undef PB_G2_LAUNCH
    if (hipGetLastError() != hipSuccess) { pb_set_error("pb_gemm2 launch failed"); return -1; }
    if (nsplit > 1) {
        if (d->ldc != d->N) { pb_set_error("pb_gemm: split-K needs a dense C (ldc == N)"); return -2; }
        const long n = (long)d->M * d->N;
        const int g = (int)std::max(1L, std::min(2048L, (n / 4 + 255) / 256));
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3(g), dim3(256), 0, stream, (const float*)d->slabs, nsplit, n, (float*)d->C);
        if (hipGetLastError() != hipSuccess) { pb_set_error("pb_reduce_slabs launch failed"); return -1; }
    }
    return 0;
}
