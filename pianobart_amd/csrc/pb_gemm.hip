// GEMM family for the PianoBART step on gfx950 (CDNA4).
//
//   C[m][n] (+)= epilogue( alpha * sum_k A(m,k) * B(n,k) )
//
// One kernel template covers every contraction of the step:
//   forward  Y  = X W^T      A K-contiguous [M][K], B K-contiguous [N][K]      ("NT")
//   dgrad    dX = dY W       A K-contiguous,        B N-contiguous [K][N]      ("NN")
//   wgrad    dW = dY^T X     A M-contiguous [K][M], B N-contiguous [K][N]      ("TN")
// in two arithmetic types: bf16 storage -> v_mfma_f32_16x16x32_bf16 (throughput path) and
// f32 storage -> v_mfma_f32_16x16x4_f32 (exact-f32 parity path). Accumulation is f32.
//
// Tiling: 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave =
// 4x4 MFMA tiles of 16x16), BK = 128 bytes of K per row (64 bf16 / 32 f32). Both operands are
// staged in LDS as [row][K] images with 128-B rows whose 16-B chunks are XOR-swizzled so the
// ds_read_b128 fragment reads of 16 consecutive rows are bank-conflict free. Operands that are
// not K-contiguous in memory are transposed in registers (4(k) x 8(r) bf16 blocks, v_perm) on
// the way into LDS, so the MFMA side is identical for all layouts. Global loads for tile k+1 are
// issued before the MFMAs of tile k (register prefetch), one barrier per K tile, 2 LDS buffers.
#include "pb_common.h"
#include "pb_api_internal.h"
#include <algorithm>

namespace {

constexpr int BM = 128, BN = 128, NTHREADS = 256;
constexpr int ROW_BYTES = 128;                 // LDS bytes per tile row (= BK elements)
constexpr int TILE_BYTES = BM * ROW_BYTES;     // 16 KiB per operand per stage

template <typename T> struct Elem;
template <> struct Elem<bf16_t> { static constexpr int EPV = 8, BK = 64; };
template <> struct Elem<float>  { static constexpr int EPV = 4, BK = 32; };

struct Vec16 { uint32_t w[4]; };

__device__ __forceinline__ int swz(int row) { return ((row >> 1) & 7) ^ ((row >> 4) & 7); }

// ---- global -> registers -------------------------------------------------------------
// Loads one 16-byte vector of `EPV` elements that are contiguous in memory starting at element
// offset `off`; `nvalid` of them (0..EPV) are inside the matrix; the rest become zero.
template <typename T>
__device__ __forceinline__ Vec16 load_vec(const T* __restrict__ base, long off, int nvalid, bool aligned) {
    constexpr int EPV = Elem<T>::EPV;
    Vec16 v = {{0u, 0u, 0u, 0u}};
    if (nvalid <= 0) return v;
    if (aligned && nvalid == EPV) {
        const uint4 q = *reinterpret_cast<const uint4*>(base + off);
        v.w[0] = q.x; v.w[1] = q.y; v.w[2] = q.z; v.w[3] = q.w;
        return v;
    }
    T tmp[EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) tmp[j] = (j < nvalid) ? base[off + j] : from_f<T>(0.f);
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t lo = __builtin_bit_cast(unsigned short, tmp[2 * j]);
            const uint32_t hi = __builtin_bit_cast(unsigned short, tmp[2 * j + 1]);
            v.w[j] = lo | (hi << 16);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v.w[j] = __builtin_bit_cast(uint32_t, tmp[j]);
    }
    return v;
}

// Staging registers for one operand tile: 4 x 16 B per thread.
struct Stage { Vec16 v[4]; };

// K-contiguous operand: element (r,k) at base[r*ld + k]. Thread t, slot i -> row t/8 + 32 i, chunk t%8.
template <typename T>
__device__ __forceinline__ void gload_kc(Stage& s, const T* __restrict__ base, long ld, int r0, int k0, int R, int K, bool aligned, int t) {
    constexpr int EPV = Elem<T>::EPV;
    const int chunk = t & 7;
    const int gk = k0 + chunk * EPV;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int gr = r0 + (t >> 3) + 32 * i;
        const int nvalid = (gr < R) ? min(EPV, K - gk) : 0;
        s.v[i] = load_vec<T>(base, (long)gr * ld + gk, nvalid, aligned);
    }
}
template <typename T>
__device__ __forceinline__ void lstore_kc(const Stage& s, char* lds, int t) {
    const int chunk = t & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (t >> 3) + 32 * i;
        uint4 q = make_uint4(s.v[i].w[0], s.v[i].w[1], s.v[i].w[2], s.v[i].w[3]);
        *reinterpret_cast<uint4*>(lds + row * ROW_BYTES + ((chunk ^ swz(row)) << 4)) = q;
    }
}

// Row-contiguous operand ("[K][R]" in memory): element (r,k) at base[k*ld + r].
// Thread t owns a 4(k) x EPV(r) block: r-chunk rc = t % (128/EPV), k-group kg = t / (128/EPV).
template <typename T>
__device__ __forceinline__ void gload_rc(Stage& s, const T* __restrict__ base, long ld, int r0, int k0, int R, int K, bool aligned, int t) {
    constexpr int EPV = Elem<T>::EPV;
    constexpr int NRC = 128 / EPV;
    const int rc = t % NRC, kg = t / NRC;
    const int gr = r0 + rc * EPV;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int gk = k0 + kg * 4 + i;
        const int nvalid = (gk < K) ? min(EPV, R - gr) : 0;
        s.v[i] = load_vec<T>(base, (long)gk * ld + gr, nvalid, aligned);
    }
}
template <typename T>
__device__ __forceinline__ void lstore_rc(const Stage& s, char* lds, int t) {
    constexpr int EPV = Elem<T>::EPV;
    constexpr int NRC = 128 / EPV;
    const int rc = t % NRC, kg = t / NRC;
    if constexpr (sizeof(T) == 2) {
        // 4(k) x 8(r) bf16 block: word w of vector i holds r = 2w, 2w+1 at k = 4 kg + i.
        // Row r = 8 rc + j gets the 4 k-values as two words: (k0,k1), (k2,k3).
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int w = j >> 1;
            const uint32_t sel = (j & 1) ? 0x07060302u : 0x05040100u;   // pick hi/lo halves of two words
            const uint32_t lo = __builtin_amdgcn_perm(s.v[1].w[w], s.v[0].w[w], sel);
            const uint32_t hi = __builtin_amdgcn_perm(s.v[3].w[w], s.v[2].w[w], sel);
            const int row = rc * 8 + j;
            const int chunk = kg >> 1;
            *reinterpret_cast<uint2*>(lds + row * ROW_BYTES + ((chunk ^ swz(row)) << 4) + ((kg & 1) << 3)) = make_uint2(lo, hi);
        }
    } else {
        // 4(k) x 4(r) f32 block: row r = 4 rc + j gets one full 16-B chunk (k = 4 kg .. 4 kg + 3).
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = rc * 4 + j;
            uint4 q = make_uint4(s.v[0].w[j], s.v[1].w[j], s.v[2].w[j], s.v[3].w[j]);
            *reinterpret_cast<uint4*>(lds + row * ROW_BYTES + ((kg ^ swz(row)) << 4)) = q;
        }
    }
}

template <typename T, bool KC>
__device__ __forceinline__ void gload(Stage& s, const T* base, long ld, int r0, int k0, int R, int K, bool aligned, int t) {
    if constexpr (KC) gload_kc<T>(s, base, ld, r0, k0, R, K, aligned, t);
    else gload_rc<T>(s, base, ld, r0, k0, R, K, aligned, t);
}
template <typename T, bool KC>
__device__ __forceinline__ void lstore(const Stage& s, char* lds, int t) {
    if constexpr (KC) lstore_kc<T>(s, lds, t);
    else lstore_rc<T>(s, lds, t);
}

// ---- MFMA over one staged K tile ---------------------------------------------------------
template <typename T>
__device__ __forceinline__ void mma_tile(const char* ldsA, const char* ldsB, int wm, int wn, int lane, f32x4 (&acc)[4][4]) {
    const int lr = lane & 15, lg = lane >> 4;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ra = wm * 64 + i * 16 + lr;
                a[i] = *reinterpret_cast<const bf16x8*>(ldsA + ra * ROW_BYTES + (((ks * 4 + lg) ^ swz(ra)) << 4));
                const int rb = wn * 64 + i * 16 + lr;
                b[i] = *reinterpret_cast<const bf16x8*>(ldsB + rb * ROW_BYTES + (((ks * 4 + lg) ^ swz(rb)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    } else {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ra = wm * 64 + i * 16 + lr;
                a[i] = *reinterpret_cast<const float*>(ldsA + ra * ROW_BYTES + ((ks ^ swz(ra)) << 4) + (lg << 2));
                const int rb = wn * 64 + i * 16 + lr;
                b[i] = *reinterpret_cast<const float*>(ldsB + rb * ROW_BYTES + ((ks ^ swz(rb)) << 4) + (lg << 2));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
}

struct GemmArgs {
    const void* A; const void* B; void* C;
    const float* bias; const void* aux_in; void* aux_out;
    int M, N, K;
    long lda, ldb, ldc, ldaux;
    int nb2; long sA1, sA2, sB1, sB2, sC1, sC2;
    float alpha;
    int flags;
    int a_aligned, b_aligned;
    int tiles_m, tiles_n;
};

template <typename T, bool A_KC, bool B_KC>
__global__ __launch_bounds__(NTHREADS) void gemm_kernel(const GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BK = Elem<T>::BK;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware tile order: consecutive tiles along M on the same XCD share the B (weight) panel.
    int bid = blockIdx.x;
    const int ntiles = p.tiles_m * p.tiles_n;
    {
        const int q = ntiles >> 3, r = ntiles & 7, x = bid & 7, idx = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + idx;
    }
    const int tm = bid % p.tiles_m, tn = bid / p.tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const int z = blockIdx.y, z1 = z / p.nb2, z2 = z % p.nb2;
    const T* A = reinterpret_cast<const T*>(p.A) + z1 * p.sA1 + z2 * p.sA2;
    const T* B = reinterpret_cast<const T*>(p.B) + z1 * p.sB1 + z2 * p.sB2;
    const long coff = z1 * p.sC1 + z2 * p.sC2;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (p.K + BK - 1) / BK;
    Stage sa, sb;
    gload<T, A_KC>(sa, A, p.lda, m0, 0, p.M, p.K, p.a_aligned, t);
    gload<T, B_KC>(sb, B, p.ldb, n0, 0, p.N, p.K, p.b_aligned, t);
    lstore<T, A_KC>(sa, smem, t);
    lstore<T, B_KC>(sb, smem + TILE_BYTES, t);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        char* cur = smem + (kt & 1) * 2 * TILE_BYTES;
        char* nxt = smem + ((kt + 1) & 1) * 2 * TILE_BYTES;
        const bool more = kt + 1 < nk;
        if (more) {
            gload<T, A_KC>(sa, A, p.lda, m0, (kt + 1) * BK, p.M, p.K, p.a_aligned, t);
            gload<T, B_KC>(sb, B, p.ldb, n0, (kt + 1) * BK, p.N, p.K, p.b_aligned, t);
        }
        mma_tile<T>(cur, cur + TILE_BYTES, wm, wn, lane, acc);
        if (more) {
            lstore<T, A_KC>(sa, nxt, t);
            lstore<T, B_KC>(sb, nxt + TILE_BYTES, t);
        }
        __syncthreads();
    }

    // ---- epilogue: C/D map of the 16x16 MFMA: col = lane&15, row = 4*(lane>>4) + reg ----
    const int lr = lane & 15, lg = lane >> 4;
    const bool accum = p.flags & PB_GEMM_ACCUM, c32 = p.flags & PB_GEMM_C_F32;
    const bool do_gelu = p.flags & PB_GEMM_GELU, mul_gg = p.flags & PB_GEMM_MUL_GELU_GRAD;
    float* C32 = reinterpret_cast<float*>(p.C) + coff;
    T* CT = reinterpret_cast<T*>(p.C) + coff;
    const T* auxin = reinterpret_cast<const T*>(p.aux_in);
    T* auxout = reinterpret_cast<T*>(p.aux_out);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = n0 + wn * 64 + j * 16 + lr;
        if (col >= p.N) continue;
        const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * 64 + i * 16 + lg * 4 + r;
                if (row >= p.M) continue;
                float v = p.alpha * acc[i][j][r] + bv;
                if (do_gelu) {
                    auxout[(long)row * p.ldaux + col] = from_f<T>(gelu_grad_f(v));
                    v = gelu_f(v);
                }
                if (mul_gg) v *= to_f(auxin[(long)row * p.ldaux + col]);
                const long ci = (long)row * p.ldc + col;
                if (c32) {
                    if (accum) v += C32[ci];
                    C32[ci] = v;
                } else {
                    if (accum) v += to_f(CT[ci]);
                    CT[ci] = from_f<T>(v);
                }
            }
        }
    }
}

template <typename T>
int launch_gemm(const GemmArgs& a, int a_kc, int b_kc, int nbatch, hipStream_t stream) {
    dim3 grid(a.tiles_m * a.tiles_n, nbatch), block(NTHREADS);
    const size_t lds = 4 * TILE_BYTES;
    if (a_kc && b_kc) hipLaunchKernelGGL((gemm_kernel<T, true, true>), grid, block, lds, stream, a);
    else if (a_kc && !b_kc) hipLaunchKernelGGL((gemm_kernel<T, true, false>), grid, block, lds, stream, a);
    else if (!a_kc && b_kc) hipLaunchKernelGGL((gemm_kernel<T, false, true>), grid, block, lds, stream, a);
    else hipLaunchKernelGGL((gemm_kernel<T, false, false>), grid, block, lds, stream, a);
    PB_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// column sums of the stored C by a separate streaming pass (paths whose epilogue does not produce them)
static int colsum_of_c(const pb_gemm_desc* d, void* stream_) {
    const bool c32 = (d->flags & PB_GEMM_C_F32) || d->dtype == PB_F32;
    return pb_colsum(d->C, d->ldc, d->colsum_out, d->colsum_ws, d->M, d->N, c32 ? PB_F32 : PB_BF16, c32 ? 1 : 0, stream_);
}

extern "C" int pb_gemm(const pb_gemm_desc* d, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    PB_REQUIRE(d != nullptr, "pb_gemm: null descriptor");
    PB_REQUIRE(d->dtype == PB_F32 || d->dtype == PB_BF16 || d->dtype == PB_F32X3, "pb_gemm: bad dtype %d", d->dtype);
    PB_REQUIRE(d->M >= 0 && d->N >= 0 && d->K >= 0, "pb_gemm: negative size");
    if (d->M == 0 || d->N == 0) return 0;
    PB_REQUIRE(d->A && d->B && d->C, "pb_gemm: null operand");
    PB_REQUIRE(!(d->flags & PB_GEMM_GELU) || d->aux_out, "pb_gemm: GELU epilogue needs aux_out");
    PB_REQUIRE(!(d->flags & PB_GEMM_MUL_GELU_GRAD) || d->aux_in, "pb_gemm: gelu-grad epilogue needs aux_in");
    const int nbt = (d->nb1 > 0 ? d->nb1 : 1) * (d->nb2 > 0 ? d->nb2 : 1);
    PB_REQUIRE(!d->colsum_out || (d->colsum_ws && nbt == 1 && d->ldc == d->N), "pb_gemm: colsum_out needs colsum_ws, one batch and a dense C");
    if (d->dtype == PB_F32X3) return pb_gemm_x3(d, stream_);        // f32 operands as bf16 triples on the bf16 kernels (pb_gemm_x3.hip)
    if (!(d->flags & PB_GEMM_FORCE_V1)) {
        const int r2 = pb_gemm2_try(d, stream_);     // 0: done (column sums included), 2: done but the column sums are still owed, 1: declined
        if (r2 < 0) return r2;
        if (r2 == 0) return 0;
        if (r2 == 2) return colsum_of_c(d, stream_);
    }
    // The generic kernel below has no row-dot epilogue: a caller that asked for one must not get a C without it (its rowdot_out would
    // stay uninitialised and feed the attention backward). Refused here for EVERY way of reaching this point: PB_GEMM_FORCE_V1, f32
    // operands, K no multiple of 64, a misaligned operand (pb_gemm2_try declines those before it looks at the flag).
    if (d->flags & PB_GEMM_ROWDOT) {
        pb_set_error("pb_gemm: PB_GEMM_ROWDOT is only served by the bf16 NT 256x256 kernel (not with PB_GEMM_FORCE_V1, f32, K %% 64 != 0 or operands off 16 bytes)");
        return -2;
    }
    const int esz = d->dtype == PB_BF16 ? 2 : 4, epv = 16 / esz;
    GemmArgs a;
    a.A = d->A; a.B = d->B; a.C = d->C; a.bias = d->bias; a.aux_in = d->aux_in; a.aux_out = d->aux_out;
    a.M = d->M; a.N = d->N; a.K = d->K; a.lda = d->lda; a.ldb = d->ldb; a.ldc = d->ldc; a.ldaux = d->ldaux;
    const int nb1 = d->nb1 > 0 ? d->nb1 : 1;
    a.nb2 = d->nb2 > 0 ? d->nb2 : 1;
    a.sA1 = d->sA1; a.sA2 = d->sA2; a.sB1 = d->sB1; a.sB2 = d->sB2; a.sC1 = d->sC1; a.sC2 = d->sC2;
    a.alpha = d->alpha; a.flags = d->flags;
    auto aligned = [&](const void* p, long ld, long s1, long s2) {
        return ((uintptr_t)p % 16 == 0) && (ld % epv == 0) && (s1 % epv == 0) && (s2 % epv == 0);
    };
    a.a_aligned = aligned(d->A, d->lda, d->sA1, d->sA2);
    a.b_aligned = aligned(d->B, d->ldb, d->sB1, d->sB2);
    a.tiles_m = (d->M + BM - 1) / BM; a.tiles_n = (d->N + BN - 1) / BN;
    PB_REQUIRE((long)nb1 * a.nb2 <= 65535, "pb_gemm: too many batches");
    const int rc = d->dtype == PB_BF16 ? launch_gemm<bf16_t>(a, d->a_kcontig, d->b_kcontig, nb1 * a.nb2, stream)
                                       : launch_gemm<float>(a, d->a_kcontig, d->b_kcontig, nb1 * a.nb2, stream);
    if (rc) return rc;
    return d->colsum_out ? colsum_of_c(d, stream_) : 0;
}

extern "C" int64_t pb_gemm_colsum_ws_floats(int32_t M, int32_t N) {
    const int64_t fused = 2LL * ((M + 255) / 256) * N;                  // one partial row per (256-row tile, wave row) of the 256x256 kernel
    return std::max<int64_t>(fused, pb_colsum_partials_floats(N));
}
