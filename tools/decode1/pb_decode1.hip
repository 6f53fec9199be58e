// K13c: batch-1 KV-cached decode, third form (round 4, OPT-IN: PB_DECODE_GRAPH=2 / 3): one token = ONE persistent kernel, on ONE XCD
// (2) or on all eight (3).
// (PianoBartLM.forward(generate=True), /root/reference/model.py:28-66; the per-layer math is pb_decode.hip's.)
//
// OUTCOME: correct (logits equal the graph form's to bf16 rounding, bit-repeatable; tests/test_model_gpu.py) but SLOWER than the graph
// form it was meant to replace: 0.85 (one XCD) / 0.63 (all XCDs) vs 0.52 ms per token at cfg 2 (profiles/r04_decode_persistent.txt).
// It stays as the measured answer to "fewer seams" (VERDICT r3 item 4), not as the default.
//
// Why it was built: the graph form (pb_decode.hip, 6 launches per layer) is a chain of dependent kernels of ~5 us each plus ~1.9 us
// between two graph nodes. Every seam between two of those launches is an all-to-all (each output row needs the whole input vector),
// so fewer launches means synchronising INSIDE a kernel, and a device-wide barrier costs as much as a graph node
// (profiles/r03_grid_barrier_probe.txt: 3.8 us over 256 workgroups, linear in their number: one memory-side counter). But the
// workgroups of ONE XCD share one L2: an arrival is an L2 atomic, the poll an sc1 load, and a barrier over its 32 workgroups measures
// 0.8-1.1 us with a small exchange (profiles/r04_xcd_probe.txt); one XCD streams 1.3 TB/s (same probe) = 0.15 ms for the ~200 MB a
// token reads; placement is workgroup b -> XCD b % 8 (same probe, HW_REG_XCC_ID).
//
// What the probes and this kernel found (all in profiles/r04_xcd_probe.txt / r04_decode_persistent.txt):
//   * data another workgroup wrote: sc0 loads and plain loads can hit a stale line of the per-CU cache; `buffer_inv sc0` does not
//     invalidate it; `buffer_inv sc1` does and costs 7 us; sc1 loads are correct but travel over the fabric (2-9 us for the 26 KB of
//     attention records behind a weight stream). What works at L2 latency: write every exchanged row ONCE per launch to an address of
//     its own (struct Mail) -- a plain load of it can only miss the per-CU cache, which starts a launch empty.
//   * loads return in order per wave and the memory system is a queue: a load issued behind a prefetched weight set waits for all of it
//     (that includes spill reloads: every spilled register was a microsecond), and vmcnt counts loads and stores together, so a barrier
//     that waits for the phase's stores waits for the next phase's weights too -- unless ONE wave (the service wave) does all the
//     input loads, row stores and the barrier, and requests no weights.
//   * 8 waves on a CU issue slowly: the phases are bound by instruction count and dependent LDS / DPP chains as much as by bytes,
//     hence v_dot2c_f32_bf16 on a packed-bf16 input row, blocked units (one lane reduction per G chunks) and LayerNorm by one wave.
//   * what is left (tools/decode1_stamps.py): ~66 us per layer = 8 phases of 2.5 .. 11 us work + 0.8 .. 1.7 us barrier; the weight
//     stream is still exposed in the GEMV phases (one register set: a second one does not fit beside the attention rows), the
//     attention phases wait for their K/V rows, and 96 barriers are 0.16 ms by themselves. The phase that has everything it needs
//     (4: LayerNorm + a d x d GEMV whose weights landed a phase earlier) takes 3.5 us; eight of those would be 0.34 ms.
//
//   * on all eight XCDs (template ALLX, 128 workgroups) the work of a phase shrinks to 1.6 .. 5 us, but every exchange crosses the
//     fabric: sc1 loads and stores for every handed-over row, a two-level barrier of 2.3 us + 0.5 us of store acknowledgement; 8 of those
//     are half of a layer's 47 us.
//
// Shape of the kernel (one XCD): grid = 8 * 32 workgroups of 512 threads; workgroup b runs on XCD b % 8, those with b % 8 != xcd leave
// at once, the other 32 are the participants (b / 8). All XCDs: 128 workgroups, all participants. Per layer 8 phases with a barrier after each:
//   1 [LN2 of the layer below | embedding]  q|k|v rows of the token   -> q, cache row i
//   2 self-attention, (head, key split) items                         -> {max, sum, out} records
//   3 merge the records, out-projection                               -> a
//   4 LN1(h + a) -> y1, cross q rows                                  -> q
//   5 cross-attention items over the cached encoder K/V               -> records
//   6 merge, out-projection                                           -> a
//   7 LNc(y1 + a) -> yc, fc1 + GELU                                   -> g
//   8 fc2                                                             -> a
// then LN2 + the LM heads. A GEMV phase gives every participant N / 32 rows; a (row, 256-column chunk) unit is one 16-byte load
// per lane of a half-wave; a half-wave owns UPH consecutive units (consecutive chunks of a row, then the next row: the address just
// advances by 512 bytes), multiplies them against the input row kept as packed bf16 in the LDS, and sums over its lanes once per G
// chunks. The units of the NEXT GEMV phase are requested as soon as a phase has multiplied its own (same registers).
// Every spin is bounded: a lost arrival raises the error word and the workgroups leave (pb_decoder_step reports it).
// bf16, head_dim 64 or 128, d a multiple of 256 up to 1024, ffn a multiple of 256 -- the shapes of the graph form.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pb_common.h"
#include "pb_api_internal.h"

namespace {

// Two forms: ALLX = false, the 32 workgroups of ONE XCD (barriers and exchanged rows through its L2); ALLX = true, 128 workgroups on
// all 8 XCDs (16 each): a two-level barrier (an L2 counter per XCD, its last arrival goes to a device counter), every exchanged row
// written and read with sc1 accesses (the L2s are not coherent with each other), an eighth of the weights per XCD.
template <bool ALLX> struct Cfg {
    static constexpr int NP = ALLX ? 128 : 32;        // participants
    static constexpr int MAXU = ALLX ? 6 : 21;        // prefetched units per half-wave (16 bytes per lane each): 14 x 21 >= the 288 units of cfg 2's widest phases / 14 x 6 >= 72
    static constexpr int MAXI = ALLX ? 1 : 3;         // attention items per workgroup (their first two passes are requested together)
};
constexpr int D1_THREADS = 512;       // 8 waves: 256 registers per lane (at 1024 threads the 128 left spill thousands of values)
constexpr int D1_WAVES = D1_THREADS / 64;
constexpr int D1_SVC = D1_WAVES - 1;  // the service wave: it brings the input row in, writes the phase's rows out and arrives at the barrier; it
                                      // requests no weights, so its loads and stores never queue behind a weight set, and the barrier need not
                                      // wait for the other waves' requests (loads and stores share vmcnt) -- the stream runs through the barriers
constexpr int D1_HW = 2 * D1_SVC;     // half-waves that stream and multiply weights
constexpr unsigned D1_SPIN_MAX = 1u << 21;

// How a GEMV phase deals its (row, chunk) units: rows per participant, KC = K / 256 chunks per row, a half-wave owns UPH consecutive
// units and reduces over its lanes after every G of them (G divides KC and UPH): pl[] then holds KC / G partial sums per row.
struct Gv { int rows, KC, G, UPH; unsigned flush; };   // flush: bit k set = reduce after unit k of a round of MAXU

// Every row one phase hands to the next lives at an address that is written once and read once per launch, on 128-byte lines of its
// own: a plain load can then only miss the per-CU cache (which starts a launch empty) and is answered by the L2 the producers wrote
// through -- sc1 loads, the alternative, go out over the fabric and queue behind the weight stream (profiles/r04_xcd_probe.txt).
// Byte offsets inside a layer's region; h_in = the layer's input row (embedding / LN2 of the layer below).
struct Mail { int h_in, q_s, rec_s, a_s, y1, q_c, rec_c, a_c, yc, g, a_f, stride; };

struct D1Args {
    const pb_decode_plan* plan;       // DEVICE copy
    int* pos;                         // device: position of the previous token (-1 before the first)
    const int16_t* tok;               // device: the 8 ids of this token
    unsigned* sync;                   // [0] arrivals (L2 atomics), [16] error word
    int xcd, ns, lds_xb;              // key splits per head; bytes of the input-row area
    Gv qkv, dd, fc1, fc2, head;
    char* mail;                       // the rows the participants hand each other, one region per (layer, phase): see Mail
    Mail mo;
};

// Every pointer of the plan is a generic pointer to the compiler (it was loaded from memory); flat loads also count on lgkmcnt, so an
// LDS access behind a prefetched weight stream would wait for the whole stream. These casts make the accesses global_*.
#define D1_GLOBAL __attribute__((address_space(1)))
typedef unsigned __attribute__((ext_vector_type(4))) u4v;          // native vectors: usable as inline-asm register operands
typedef unsigned __attribute__((ext_vector_type(2))) u2v;
typedef float __attribute__((ext_vector_type(2))) f32x2;
__device__ __forceinline__ u4v ldg16(const void* p) { return *(const D1_GLOBAL u4v*)p; }
__device__ __forceinline__ u2v ldg8(const void* p) { return *(const D1_GLOBAL u2v*)p; }
__device__ __forceinline__ float ldgf(const float* p) { return *(const D1_GLOBAL float*)p; }
__device__ __forceinline__ f32x4 ldg4f(const float* p) { return *(const D1_GLOBAL f32x4*)p; }
__device__ __forceinline__ void stg32(void* p, unsigned v) { *(D1_GLOBAL unsigned*)p = v; }
__device__ __forceinline__ void stgf(float* p, float v) { *(D1_GLOBAL float*)p = v; }
__device__ __forceinline__ unsigned ld_sc1(const void* p) {
    return __hip_atomic_load((const D1_GLOBAL unsigned*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Rows another participant produced in this launch. One XCD: plain accesses (struct Mail explains why the loads are safe). All XCDs:
// sc1 dword loads / stores (agent scope: they bypass the per-XCD L2s), tracked by the compiler like any other access.
template <bool ALLX> __device__ __forceinline__ u4v xld16(const void* p) {
    if constexpr (!ALLX) return ldg16(p);
    else { const unsigned* q = (const unsigned*)p; return u4v{ld_sc1(q), ld_sc1(q + 1), ld_sc1(q + 2), ld_sc1(q + 3)}; }
}
template <bool ALLX> __device__ __forceinline__ u2v xld8(const void* p) {
    if constexpr (!ALLX) return ldg8(p);
    else { const unsigned* q = (const unsigned*)p; return u2v{ld_sc1(q), ld_sc1(q + 1)}; }
}
template <bool ALLX> __device__ __forceinline__ void xst32(void* p, unsigned v) {
    if constexpr (!ALLX) stg32(p, v);
    else __hip_atomic_store((D1_GLOBAL unsigned*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool ALLX> __device__ __forceinline__ void xstf(float* p, float v) { xst32<ALLX>(p, __float_as_uint(v)); }
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ unsigned pack_bf(float a, float b) {
    const bf16x2 r = {(bf16_t)a, (bf16_t)b};
    return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ float dot2(unsigned a, unsigned b, float c) {      // c + a.lo b.lo + a.hi b.hi (v_dot2c_f32_bf16)
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a), __builtin_bit_cast(bf16x2, b), c, false);
}
__device__ __forceinline__ float dot8(u4v a, u4v b, float c) { return dot2(a.w, b.w, dot2(a.z, b.z, dot2(a.y, b.y, dot2(a.x, b.x, c)))); }

__device__ __forceinline__ float half_sum32(float v) {           // sum over the 32 lanes of a half-wave, in every lane of it
    v += PB_DPP_F(v, 0xb1);
    v += PB_DPP_F(v, 0x4e);
    v += PB_DPP_F(v, 0x141);
    v += PB_DPP_F(v, 0x140);
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(a[0]) + __uint_as_float(a[1]);
}

#ifdef PB_D1_STAMPS
__device__ unsigned long long d1_stamps[2][512];
#define D1_STAMP(g) do { if ((g).t == 0 && ((g).wg == 0 || (g).wg == 17) && (g).nst < 512) d1_stamps[(g).wg ? 1 : 0][(g).nst++] = __builtin_amdgcn_s_memtime(); } while (0)
__device__ unsigned long long d1_sub[32];
// sub-phase accounting of participant 0: ticks since the previous mark are added to slot `id`
#define D1_SUB(g, id) do { if ((g).t == 0 && (g).wg == 0) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); d1_sub[id] += n_ - (g).last; (g).last = n_; } } while (0)
#else
#define D1_STAMP(g) do { } while (0)
#define D1_SUB(g, id) do { } while (0)
#endif

// The lane's offsets into every buffer are the same in every layer; the compiler would hoist all of them out of the layer loop and
// then spill them -- and a spill reload is a memory round trip behind the weight stream. Passing the lane ids through this fence at
// the top of a phase keeps the arithmetic (a few instructions) inside the phase.
__device__ __forceinline__ int fenced(int v) { asm volatile("" : "+v"(v)); return v; }

struct Wg {
    unsigned* xb;     // the phase's input row, packed bf16 pairs (the values the graph form stores as bf16)
    float* pl;        // [512] partial sums of the GEMV units
    float* red;       // [64] flags / small reductions
    float* rec;       // [MAXI][D1_WAVES][HD + 4] per-wave attention records
    float* stage;     // [H ns (HD + 4)] the split records of all heads, then [H ns] their merge weights
    int wg, t, lane, wave, l32, hw;
    bool svc;
    unsigned* sync;
    unsigned target;  // barriers passed so far + 1
    int xcd;          // the XCD this workgroup runs on (blockIdx.x % 8)
    bool dead;
    int nst;
    unsigned long long last;
};

// Barrier over the participants: every store of this workgroup (all of them are the service wave's) has left the CU and is
// acknowledged, then the service wave arrives and polls. One XCD: an L2 atomic on sync[0], sc1 polls of it. All XCDs: an L2 atomic on
// the XCD's own counter (sync[64 + 32 x]); the arrival that completes the XCD's 16 adds one to the device counter sync[0] (agent
// scope); everybody polls that one until all 8 XCDs are in. The other waves only meet at the workgroup barriers: their weight requests
// stay in flight.
template <bool ALLX>
__device__ __forceinline__ void xcd_barrier(Wg& g) {
    D1_STAMP(g);
    if (g.svc) __builtin_amdgcn_s_waitcnt(0);                     // every global store of a phase is the service wave's
    __syncthreads();
    D1_STAMP(g);
    if (g.t == D1_THREADS - 64 && !g.dead) {
        unsigned want;
        if constexpr (ALLX) {
            const unsigned old = __hip_atomic_fetch_add((D1_GLOBAL unsigned*)(g.sync + 64 + 32 * g.xcd), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (old + 1u == g.target * (Cfg<ALLX>::NP / 8))
                __hip_atomic_fetch_add((D1_GLOBAL unsigned*)g.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            want = g.target * 8u;
        } else {
            __hip_atomic_fetch_add((D1_GLOBAL unsigned*)g.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            want = g.target * Cfg<ALLX>::NP;
        }
        unsigned spins = 0;
        while (ld_sc1(g.sync) < want) {
            if (++spins > D1_SPIN_MAX || ld_sc1(g.sync + 16) != 0u) {
                __hip_atomic_store((D1_GLOBAL unsigned*)(g.sync + 16), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                g.red[63] = 1.f;
                break;
            }
        }
    }
    __syncthreads();
    if (g.red[63] != 0.f) g.dead = true;
    g.target += 1u;
    D1_STAMP(g);
    D1_SUB(g, 0);
}

// ---------------------------------------------------------------- GEMV phases
template <int MAXU> struct WSet { u4v w[MAXU]; };

// request units [kb, kb + MAXU) of this half-wave's UPH (kb = 0: the prefetched round)
template <int MAXU>
__device__ __forceinline__ void gemv_request(WSet<MAXU>& ws, const Wg& g, const bf16_t* __restrict__ W, int K, const Gv& gv, int kb) {
    // the lane's offsets are the same in every layer: without this fence the compiler hoists all of them out of the layer loop and
    // the kernel spills thousands of registers
    if (g.svc) return;
    int hw = g.hw, l32 = g.l32;
    asm volatile("" : "+v"(hw), "+v"(l32));
    const int nunits = gv.rows * gv.KC, u0 = hw * gv.UPH + kb;
    const bf16_t* base = W + (size_t)g.wg * gv.rows * K + l32 * 8;
#pragma unroll
    for (int k = 0; k < MAXU; ++k) {
        const int u = min(u0 + k, nunits - 1);                    // beyond the last unit: re-read it (its product is dropped)
        ws.w[k] = ldg16(base + (size_t)u * 256);
    }
}

// multiply units [kb, kb + MAXU) with the input row; acc / c / vr carry over from round to round
template <int MAXU>
__device__ __forceinline__ void gemv_units(const WSet<MAXU>& ws, Wg& g, const Gv& gv, int kb, float& acc, int& c, int& vr) {
    const int nvr = gv.rows * (gv.KC / gv.G);
#pragma unroll
    for (int k = 0; k < MAXU; ++k) {
        if (k % 7 == 0) __builtin_amdgcn_sched_barrier(0);       // 7 input reads in flight are enough: unbounded, the scheduler hoists all 21 (84 registers) and spills
        if (kb + k < gv.UPH) {                                    // uniform
            const u4v x = *reinterpret_cast<const u4v*>(g.xb + c * 128 + g.l32 * 4);
            acc = dot8(ws.w[k], x, acc);
            c = c + 1 == gv.KC ? 0 : c + 1;
            if ((gv.flush >> k) & 1u) {                           // uniform: G units done
                const float s = half_sum32(acc);
                if (g.l32 == 0 && vr < nvr) g.pl[vr] = s;
                acc = 0.f; ++vr;
            }
        }
    }
}

// The small operands of a phase (its rows' bias, the LayerNorm weights of its input): requested together with the weight set, one
// phase early -- a load issued on the spot would queue behind the weight stream for microseconds.
template <int NC> struct Small { f32x4 gm[NC], bt[NC]; float b0, b1; };
template <int NC>
__device__ __forceinline__ void small_request(Small<NC>& sm, const Wg& g, const float* bias, int rows, bool f32out, const float* gamma, const float* beta) {
    if (!g.svc) return;
    const int row0 = g.wg * rows, lane = fenced(g.lane), t = lane;
    sm.b0 = sm.b1 = 0.f;
    if (f32out) { if (t < rows) sm.b0 = ldgf(bias + row0 + t); }
    else if (t < (rows >> 1)) { sm.b0 = ldgf(bias + row0 + 2 * t); sm.b1 = ldgf(bias + row0 + 2 * t + 1); }
    if (gamma) {
#pragma unroll
        for (int j = 0; j < NC; ++j) { sm.gm[j] = ldg4f(gamma + 4 * lane + 256 * j); sm.bt[j] = ldg4f(beta + 4 * lane + 256 * j); }
    }
}

// act: 0 none, 1 exact GELU. Rows go out as bf16 pairs: global row n < n0 -> p0[n], n < n1 -> p1[n - n0], else p2[n - n1];
// F32: p0 is a float row (the logits).
struct RowsOut { void* p0; int n0; void* p1; int n1; void* p2; };

// `input` brings the phase's input row into g.xb (one wave may do it; a __syncthreads follows); `next` issues the requests of the NEXT
// GEMV phase (weights, small operands) once this phase's units are multiplied: into the SAME registers (two sets in flight spill, and
// a spill reload is a memory round trip that queues behind the weight stream like every other load), and behind this phase's own
// input loads. The set then has the row output, the barrier and the next phase's input to stream in.
template <bool F32, bool ALLX, typename FI, typename FN>
__device__ __forceinline__ void gemv_phase(WSet<Cfg<ALLX>::MAXU>& ws, Wg& g, const bf16_t* __restrict__ W, const float& sb0, const float& sb1, int K, int act,
                                           const Gv& gv, const RowsOut& out, FI&& input, FN&& next) {
    const int rows = gv.rows, row0 = g.wg * rows;
    input();
    const float bv0 = sb0, bv1 = sb1;                             // this phase's bias (input() may have requested it; next() overwrites the set)
    D1_SUB(g, 1);
    __syncthreads();                                              // the input row is complete
    D1_SUB(g, 2);
    if (!g.svc) {
        int hw = g.hw;
        asm volatile("" : "+v"(hw));
        const int u0 = hw * gv.UPH;
        int c = u0 % gv.KC, vr = u0 / gv.G;
        float acc = 0.f;
        gemv_units(ws, g, gv, 0, acc, c, vr);
        for (int kb = Cfg<ALLX>::MAXU; kb < gv.UPH; kb += Cfg<ALLX>::MAXU) {      // wider shapes: further rounds, loaded on the spot (flush pattern repeats: MAXU % G == 0 is required then)
            gemv_request(ws, g, W, K, gv, kb);
            gemv_units(ws, g, gv, kb, acc, c, vr);
        }
    }
    next();                                                       // the set is consumed: the next phase's requests reuse its registers
    D1_SUB(g, 3);
    __syncthreads();
    D1_SUB(g, 4);
    const int per = gv.KC / gv.G;
    const int t = fenced(g.lane);                                 // rows out: the service wave (rows / 2 <= 64 pairs, or <= 64 f32 rows)
    if (!g.svc) {
    } else if (F32) {
        if (t < rows) {
            float s = 0.f;
            for (int j = 0; j < per; ++j) s += g.pl[t * per + j];
            stgf(reinterpret_cast<float*>(out.p0) + row0 + t, s + bv0);
        }
    } else if (t < (rows >> 1)) {
        float v[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = 2 * t + j;
            float s = 0.f;
            for (int q = 0; q < per; ++q) s += g.pl[r * per + q];
            s += j ? bv1 : bv0;
            v[j] = act ? gelu_f(s) : s;
        }
        const int n = row0 + 2 * t;
        bf16_t* dst = n < out.n0 ? reinterpret_cast<bf16_t*>(out.p0) + n
                    : n < out.n1 ? reinterpret_cast<bf16_t*>(out.p1) + (n - out.n0) : reinterpret_cast<bf16_t*>(out.p2) + (n - out.n1);
        xst32<ALLX>(dst, pack_bf(v[0], v[1]));
    }
    D1_SUB(g, 5);
}

// xb[0 .. d) = LayerNorm(res + add) gamma + beta, rounded to bf16 like the stored row; the first participant also stores it. ONE wave
// does it (lane l holds elements 4 l + 256 j .. + 4, j < NC: wave reductions only, no workgroup barrier, and the other waves go on
// to request weights meanwhile).
template <int NC, bool ALLX>
__device__ __forceinline__ void ln_to_xb(Wg& g, const bf16_t* res, const bf16_t* add, const Small<NC>& sm, bf16_t* ln_out, float eps, bool store = true) {
    if (!g.svc) return;
    constexpr int d = 256 * NC;
    const int lane = fenced(g.lane);
    u2v r[NC], a[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) { r[j] = xld8<ALLX>(res + 4 * lane + 256 * j); a[j] = xld8<ALLX>(add + 4 * lane + 256 * j); }
    D1_SUB(g, 14);
    float v[NC][4], s = 0.f;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        v[j][0] = bf_lo(r[j].x) + bf_lo(a[j].x); v[j][1] = bf_hi(r[j].x) + bf_hi(a[j].x);
        v[j][2] = bf_lo(r[j].y) + bf_lo(a[j].y); v[j][3] = bf_hi(r[j].y) + bf_hi(a[j].y);
        s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
    }
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NC; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[j][e] -= mean; q = fmaf(v[j][e], v[j][e], q); }
    const float rstd = rsqrtf(wave_sum(q) / (float)d + eps);
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        u2v pk;
        pk.x = pack_bf(v[j][0] * rstd * sm.gm[j][0] + sm.bt[j][0], v[j][1] * rstd * sm.gm[j][1] + sm.bt[j][1]);
        pk.y = pack_bf(v[j][2] * rstd * sm.gm[j][2] + sm.bt[j][2], v[j][3] * rstd * sm.gm[j][3] + sm.bt[j][3]);
        *reinterpret_cast<u2v*>(g.xb + 2 * lane + 128 * j) = pk;
        if (g.wg == 0 && store) { xst32<ALLX>(ln_out + 4 * lane + 256 * j, pk.x); xst32<ALLX>(ln_out + 4 * lane + 256 * j + 2, pk.y); }
    }
    D1_SUB(g, 15);
}

// xb[0 .. n) = a bf16 row produced inside this launch (n <= 8192), by the service wave
template <bool ALLX>
__device__ __forceinline__ void row_to_xb(Wg& g, const bf16_t* row, int n) {
    if (!g.svc) return;
    const int lane = fenced(g.lane), n8 = n >> 3;
    for (int k0 = 0; k0 < n8; k0 += 64 * 8) {
        u4v r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = xld16<ALLX>(row + 8 * min(k0 + lane + 64 * k, n8 - 1));
#pragma unroll
        for (int k = 0; k < 8; ++k) if (k0 + lane + 64 * k < n8) *reinterpret_cast<u4v*>(g.xb + 4 * (k0 + lane + 64 * k)) = r[k];
    }
    D1_SUB(g, 16);
}

// xb[0 .. d) = the attention context: the splits' records {m, l, -, -, o[HD]} of every head merged, rounded to bf16 like the
// stored context of the graph form (gemv_kernel / MergeIn). The records come in with coalesced sc1 loads, all in flight at once;
// a thread per (head, split) turns {m, l} into the split's weight e^(m - M) / L, a thread per element pair sums the outputs.
template <int HD, bool ALLX, typename FM>
__device__ __forceinline__ void ctx_to_xb(Wg& g_, const float* part, int d, int ns, FM&& meanwhile) {
    constexpr int RS = HD + 4;
    const int H = d / HD, n = H * ns * RS;
    Wg g = g_;
    g.t = fenced(g_.t);
    float* st = g.stage;
    float* wt = st + n;
    if constexpr (ALLX) {                                         // all threads: the weight requests in flight are small in this form
        const int n4 = n >> 2;
        for (int k0 = 0; k0 < n4; k0 += D1_THREADS * 4) {
            u4v r[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) r[k] = xld16<true>(part + 4 * min(k0 + g.t + D1_THREADS * k, n4 - 1));
#pragma unroll
            for (int k = 0; k < 4; ++k) if (k0 + g.t + D1_THREADS * k < n4) *reinterpret_cast<u4v*>(st + 4 * (k0 + g.t + D1_THREADS * k)) = r[k];
        }
        D1_SUB(g, 12);
    } else if (g.svc) {                                           // the records come in through the service wave (nothing queued before its loads)
        const int n4 = n >> 2, lane = g.t & 63;                   // RS is a multiple of 4
        for (int k0 = 0; k0 < n4; k0 += 64 * 14) {
            u4v r[14];
#pragma unroll
            for (int k = 0; k < 14; ++k) r[k] = ldg16(part + 4 * min(k0 + lane + 64 * k, n4 - 1));
#pragma unroll
            for (int k = 0; k < 14; ++k) if (k0 + lane + 64 * k < n4) *reinterpret_cast<u4v*>(st + 4 * (k0 + lane + 64 * k)) = r[k];
        }
        D1_SUB(g, 12);
    }
    meanwhile();              // the other waves request this phase's weights while the service wave stages (in program order AFTER the staging registers die)
    __syncthreads();
    if (g.t < H * ns) {
        const int h = g.t / ns;
        const float* rec = st + h * ns * RS;
        float ms[PB_DECODE_MAX_SPLITS], ls[PB_DECODE_MAX_SPLITS];
#pragma unroll
        for (int s = 0; s < PB_DECODE_MAX_SPLITS; ++s) {
            const int sc = s < ns ? s : 0;
            ms[s] = rec[sc * RS]; ls[s] = rec[sc * RS + 1];
            if (s >= ns) ms[s] = -INFINITY;
        }
        float M = -INFINITY;
#pragma unroll
        for (int s = 0; s < PB_DECODE_MAX_SPLITS; ++s) M = fmaxf(M, ms[s]);
        float L = 0.f;
#pragma unroll
        for (int s = 0; s < PB_DECODE_MAX_SPLITS; ++s) L = fmaf(ls[s], (M == -INFINITY || ms[s] == -INFINITY) ? 0.f : __expf(ms[s] - M), L);
        const float mine = st[g.t * RS];
        wt[g.t] = (L > 0.f && mine != -INFINITY) ? __expf(mine - M) / L : 0.f;      // nothing visible -> zero row (oracle header)
    }
    __syncthreads();
    for (int c2 = g.t; c2 < (d >> 1); c2 += D1_THREADS) {
        const int h = (2 * c2) / HD, e = 2 * c2 - h * HD;
        const float* rec = st + h * ns * RS + 4 + e;
        float o0 = 0.f, o1 = 0.f;
#pragma unroll
        for (int s = 0; s < PB_DECODE_MAX_SPLITS; ++s) {
            if (s < ns) {
                const float w = wt[h * ns + s];
                const f32x2 ov = *reinterpret_cast<const f32x2*>(rec + s * RS);
                o0 = fmaf(ov[0], w, o0); o1 = fmaf(ov[1], w, o1);
            }
        }
        g.xb[c2] = pack_bf(o0, o1);
    }
    D1_SUB(g, 13);
    g_.last = g.last;
}

// ---------------------------------------------------------------- single-query attention over (head, key split) items
// Item it = wg + 32 m: head it / ns, keys [sp ck, min(Sk, (sp + 1) ck)). The waves of the workgroup take KPW keys per pass each
// (a key row = CPR lanes of 16 bytes), keep an online {max, sum, out} per wave, and the workgroup merges its waves' records.
template <int MAXI> struct Items { int h[MAXI], sp[MAXI]; };

template <int HD, bool ALLX, typename FN>
__device__ __forceinline__ void attn_phase(Wg& g_, const Items<Cfg<ALLX>::MAXI>& its, const bf16_t* qrow, const bf16_t* __restrict__ kc, const bf16_t* __restrict__ vc, long kv_ss,
                                           const float* __restrict__ key_mask, int Sk, int ck, int ns, int H, int d, float scale, float* part, FN&& next) {
    constexpr int CPR = HD / 8, KPW = 64 / CPR, STEP = D1_WAVES * KPW, RS = HD + 4, D1_MAXI = Cfg<ALLX>::MAXI, D1_WGS = Cfg<ALLX>::NP;
    Wg g = g_;
    g.lane = fenced(g_.lane); g.wave = fenced(g_.wave); g.t = fenced(g_.t);
    const int sub = g.lane % CPR, grp = g.lane / CPR;
    const int nit = H * ns;
    // the first two passes of every item: requested before q arrives (none of it depends on q)
    u4v kpre[D1_MAXI][2], vpre[D1_MAXI][2];
    float mpre[D1_MAXI][2];
#pragma unroll
    for (int m = 0; m < D1_MAXI; ++m) {
        const int it = g.wg + D1_WGS * m;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            kpre[m][ps] = u4v{0u, 0u, 0u, 0u}; vpre[m][ps] = u4v{0u, 0u, 0u, 0u}; mpre[m][ps] = 1.f;
            if (it < nit) {
                const int j = min(its.sp[m] * ck + ps * STEP + g.wave * KPW + grp, Sk - 1);     // beyond the split: any cached row (its weight is 0)
                kpre[m][ps] = ldg16(kc + (long)j * kv_ss + its.h[m] * HD + sub * 8);
                vpre[m][ps] = ldg16(vc + (long)j * kv_ss + its.h[m] * HD + sub * 8);
                if (key_mask) mpre[m][ps] = ldgf(key_mask + j);
            }
        }
    }
    D1_SUB(g, 6);
    row_to_xb<ALLX>(g, qrow, d);
    D1_SUB(g, 7);
    __syncthreads();
    D1_SUB(g, 8);
#pragma unroll
    for (int m = 0; m < D1_MAXI; ++m) {
        const int it = g.wg + D1_WGS * m;
        if (it >= nit) break;
        const int h = its.h[m], sp = its.sp[m];
        const int j0 = sp * ck, j1 = min(Sk, j0 + ck);
        const u4v qv = *reinterpret_cast<const u4v*>(g.xb + (h * HD + sub * 8) / 2);
        float mw = -INFINITY, l = 0.f, o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = 0.f;
        const int jw = j0 + g.wave * KPW;
        for (int jb = jw; jb < j1; jb += STEP) {
            const int j = jb + grp;
            u4v kr = jb == jw ? kpre[m][0] : kpre[m][1], vr = jb == jw ? vpre[m][0] : vpre[m][1];
            float mk = jb == jw ? mpre[m][0] : mpre[m][1];
            if (jb >= jw + 2 * STEP) {
                const int jc = min(j, Sk - 1);
                kr = ldg16(kc + (long)jc * kv_ss + h * HD + sub * 8);
                vr = ldg16(vc + (long)jc * kv_ss + h * HD + sub * 8);
                mk = key_mask ? ldgf(key_mask + jc) : 1.f;
            }
            float s = dot8(kr, qv, 0.f);
            s += PB_DPP_F(s, 0xb1); s += PB_DPP_F(s, 0x4e); s += PB_DPP_F(s, 0x141);
            if (CPR == 16) s += PB_DPP_F(s, 0x140);
            const bool vis = j < j1 && mk != 0.f;
            s = vis ? s * scale : -INFINITY;
            const float mn = fmaxf(mw, wave_max(s));
            if (mn != -INFINITY) {
                const float corr = mw == -INFINITY ? 0.f : __expf(mw - mn);
                const float p = vis ? __expf(s - mn) : 0.f;
                l = fmaf(l, corr, p);
                const float vv[8] = {bf_lo(vr.x), bf_hi(vr.x), bf_lo(vr.y), bf_hi(vr.y), bf_lo(vr.z), bf_hi(vr.z), bf_lo(vr.w), bf_hi(vr.w)};
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = fmaf(o[e], corr, p * vv[e]);
                mw = mn;
            }
        }
        // sums over the key groups of the wave (lanes of equal `sub`): strides CPR .. 32
        auto over_groups = [&](float v) {
            if (CPR == 8) v += PB_DPP_F(v, 0x128);                 // row_ror:8
            auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
            v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
            auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
            return __uint_as_float(b[0]) + __uint_as_float(b[1]);
        };
        l = over_groups(l);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = over_groups(o[e]);
        float* wr = g.rec + (size_t)(m * D1_WAVES + g.wave) * RS;
        if (g.lane == 0) { wr[0] = mw; wr[1] = l; }
        if (grp == 0) {
            *reinterpret_cast<f32x4*>(wr + 4 + sub * 8) = f32x4{o[0], o[1], o[2], o[3]};
            *reinterpret_cast<f32x4*>(wr + 4 + sub * 8 + 4) = f32x4{o[4], o[5], o[6], o[7]};
        }
    }
    next();                   // the cached rows are consumed: the next GEMV phase's requests go out here (no weight set is live while the items hold their rows)
    D1_SUB(g, 9);
    __syncthreads();
    D1_SUB(g, 10);
    // merge the waves of every item: lane (m, e) of the service wave
    if (g.svc)
    for (int idx = g.lane; idx < D1_MAXI * HD; idx += 64) {
        const int m = idx / HD, e = idx - m * HD, it = g.wg + D1_WGS * m;
        if (it >= nit) break;
        const float* wr = g.rec + (size_t)m * D1_WAVES * RS;
        float mv[D1_WAVES], lv[D1_WAVES], ov[D1_WAVES];
#pragma unroll
        for (int w = 0; w < D1_WAVES; ++w) { mv[w] = wr[w * RS]; lv[w] = wr[w * RS + 1]; ov[w] = wr[w * RS + 4 + e]; }
        float M = -INFINITY;
#pragma unroll
        for (int w = 0; w < D1_WAVES; ++w) M = fmaxf(M, mv[w]);
        float L = 0.f, O = 0.f;
#pragma unroll
        for (int w = 0; w < D1_WAVES; ++w) {
            const float wt = (M == -INFINITY || mv[w] == -INFINITY) ? 0.f : __expf(mv[w] - M);
            L = fmaf(lv[w], wt, L); O = fmaf(ov[w], wt, O);
        }
        float* out = part + (size_t)it * RS;
        xstf<ALLX>(out + 4 + e, O);
        if (e == 0) { xstf<ALLX>(out, M); xstf<ALLX>(out + 1, L); }
    }
    D1_SUB(g, 11);
    g_.last = g.last;
}

// ---------------------------------------------------------------- the token
template <int NC, int HD, bool ALLX>
__global__ __launch_bounds__(D1_THREADS) void decode1_kernel(const D1Args a) {
    if (!ALLX && (int)(blockIdx.x & 7) != a.xcd) return;
    using C = Cfg<ALLX>;
    constexpr int D1_MAXI = C::MAXI, D1_WGS = C::NP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Wg g;
    g.xb = reinterpret_cast<unsigned*>(smem);
    g.pl = reinterpret_cast<float*>(smem + a.lds_xb);
    g.red = g.pl + 512;
    g.rec = g.red + 64;
    g.stage = g.rec + D1_MAXI * D1_WAVES * (HD + 4);
    g.wg = ALLX ? blockIdx.x : blockIdx.x >> 3; g.xcd = blockIdx.x & 7; g.t = threadIdx.x; g.lane = g.t & 63; g.wave = g.t >> 6; g.l32 = g.t & 31; g.hw = g.t >> 5; g.svc = g.wave == D1_SVC;
    g.sync = a.sync; g.dead = false; g.nst = 0; g.last = __builtin_amdgcn_s_memtime();
    D1_STAMP(g);
    if (g.t == 0) g.red[63] = 0.f;
    const pb_decode_plan* __restrict__ p = a.plan;
    constexpr int d = 256 * NC;
    const int H = d / HD, f = p->ffn, nl = p->n_layers, ns = a.ns;
    const float scale = rsqrtf((float)HD), eps = 1e-5f;
    const int i = *(const D1_GLOBAL int*)a.pos + 1;                // position of this token
    g.target = (unsigned)i * (unsigned)(8 * nl) + 1u;
    const long kv_ss = 2 * d;
    WSet<C::MAXU> wa;
    Items<D1_MAXI> its;
#pragma unroll
    for (int m = 0; m < D1_MAXI; ++m) { const int it = g.wg + D1_WGS * m; its.h[m] = it / ns; its.sp[m] = it - its.h[m] * ns; }
    // token embedding + learned position + LayerNorm (dec_embed_kernel's sums), by every participant for itself (wave 0: 4 columns
    // per lane and 256-column block)
    if (g.svc) {
        const u4v raw = ldg16(a.tok);
        const int id[8] = {(int)(raw.x & 0xffff), (int)(raw.x >> 16), (int)(raw.y & 0xffff), (int)(raw.y >> 16),
                           (int)(raw.z & 0xffff), (int)(raw.z >> 16), (int)(raw.w & 0xffff), (int)(raw.w >> 16)};
        f32x4 v[NC];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const int col = 4 * g.lane + 256 * j;
            v[j] = ldg4f(p->lin_b + col) + ldg4f(p->pos + (size_t)(i + 2) * d + col);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[j] += ldg4f(p->ptab + (size_t)(p->tab_off[k] + id[k]) * d + col);
            s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
        }
        const float mean = wave_sum(s) / (float)d;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NC; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float c = v[j][e] - mean; q = fmaf(c, c, q); }
        const float rstd = rsqrtf(wave_sum(q) / (float)d + eps);
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const int col = 4 * g.lane + 256 * j;
            const f32x4 y = (v[j] - mean) * rstd * ldg4f(p->lne_w + col) + ldg4f(p->lne_b + col);
            const u2v pk = {pack_bf(y[0], y[1]), pack_bf(y[2], y[3])};
            *reinterpret_cast<u2v*>(g.xb + col / 2) = pk;
            if (g.wg == 0) { xst32<ALLX>((bf16_t*)(a.mail + a.mo.h_in) + col, pk.x); xst32<ALLX>((bf16_t*)(a.mail + a.mo.h_in) + col + 2, pk.y); }
        }
    }
    // layer 0's q|k|v rows (the working waves; in program order behind the embedding, whose table rows would otherwise be live together
    // with the weight set)
    gemv_request(wa, g, (const bf16_t*)p->layers[0].wqkv, d, a.qkv, 0);
    constexpr int KPW = 64 / (HD / 8);
    const int cross_ck = ((p->S_enc + ns - 1) / ns + KPW - 1) / KPW * KPW;
    const Mail& mo = a.mo;
    // One weight set and one set of small operands: the requests of a GEMV phase go out in the GEMV phase before it, as soon as that
    // phase has multiplied its units (gemv_phase).
    Small<NC> sa;
    small_request<NC>(sa, g, p->layers[0].bqkv, a.qkv.rows, false, nullptr, nullptr);
    for (int l = 0; l < nl; ++l) {
        const pb_decode_layer& L = p->layers[l];
        char* ml = a.mail + (size_t)l * mo.stride;
        char* mp = a.mail + (size_t)(l ? l - 1 : 0) * mo.stride;   // the layer below (l > 0)
        bf16_t* h = (bf16_t*)(ml + mo.h_in);
        // ---- 1: q | k | v rows of this token (wa); input: the embedding row, or LN2(yc + a) of the layer below
        {
            bf16_t* krow = (bf16_t*)L.kv_self + (size_t)i * kv_ss;
            gemv_phase<false, ALLX>(wa, g, (const bf16_t*)L.wqkv, sa.b0, sa.b1, d, 0, a.qkv, RowsOut{ml + mo.q_s, d, krow, 2 * d, krow + d},
                              [&] { if (l) ln_to_xb<NC, ALLX>(g, (const bf16_t*)(mp + mo.yc), (const bf16_t*)(mp + mo.a_f), sa, h, eps); }, [] {});
        }
        xcd_barrier<ALLX>(g);
        // ---- 2: self-attention over rows 0 .. i
        {
            const int Sk = i + 1;
            const int ck = ((Sk + ns - 1) / ns + KPW - 1) / KPW * KPW;
            attn_phase<HD, ALLX>(g, its, (const bf16_t*)(ml + mo.q_s), (const bf16_t*)L.kv_self, (const bf16_t*)L.kv_self + d, kv_ss, nullptr, Sk, ck, ns, H, d, scale,
                           (float*)(ml + mo.rec_s), [] {});
        }
        xcd_barrier<ALLX>(g);
        // ---- 3: out-projection of the merged context (wb)
        gemv_phase<false, ALLX>(wa, g, (const bf16_t*)L.wo, sa.b0, sa.b1, d, 0, a.dd, RowsOut{ml + mo.a_s, d, ml + mo.a_s, d, ml + mo.a_s},
                          [&] { ctx_to_xb<HD, ALLX>(g, (const float*)(ml + mo.rec_s), d, ns,
                                              [&] { gemv_request(wa, g, (const bf16_t*)L.wo, d, a.dd, 0); small_request<NC>(sa, g, L.bo, a.dd.rows, false, nullptr, nullptr); }); },
                          [&] { gemv_request(wa, g, (const bf16_t*)L.wq_c, d, a.dd, 0); small_request<NC>(sa, g, L.bq_c, a.dd.rows, false, L.ln1_w, L.ln1_b); });
        xcd_barrier<ALLX>(g);
        // ---- 4: y1 = LN1(h + a); cross q rows (wa)
        gemv_phase<false, ALLX>(wa, g, (const bf16_t*)L.wq_c, sa.b0, sa.b1, d, 0, a.dd, RowsOut{ml + mo.q_c, d, ml + mo.q_c, d, ml + mo.q_c},
                          [&] { ln_to_xb<NC, ALLX>(g, h, (const bf16_t*)(ml + mo.a_s), sa, (bf16_t*)(ml + mo.y1), eps); }, [] {});
        xcd_barrier<ALLX>(g);
        // ---- 5: cross-attention over the cached encoder keys
        attn_phase<HD, ALLX>(g, its, (const bf16_t*)(ml + mo.q_c), (const bf16_t*)L.kv_cross, (const bf16_t*)L.kv_cross + d, kv_ss, p->enc_mask, p->S_enc, cross_ck, ns, H, d, scale,
                       (float*)(ml + mo.rec_c), [] {});
        xcd_barrier<ALLX>(g);
        // ---- 6: cross out-projection (wb)
        gemv_phase<false, ALLX>(wa, g, (const bf16_t*)L.wo_c, sa.b0, sa.b1, d, 0, a.dd, RowsOut{ml + mo.a_c, d, ml + mo.a_c, d, ml + mo.a_c},
                          [&] { ctx_to_xb<HD, ALLX>(g, (const float*)(ml + mo.rec_c), d, ns,
                                              [&] { gemv_request(wa, g, (const bf16_t*)L.wo_c, d, a.dd, 0); small_request<NC>(sa, g, L.bo_c, a.dd.rows, false, nullptr, nullptr); }); },
                          [&] { gemv_request(wa, g, (const bf16_t*)L.w1, d, a.fc1, 0); small_request<NC>(sa, g, L.b1, a.fc1.rows, false, L.lnc_w, L.lnc_b); });
        xcd_barrier<ALLX>(g);
        // ---- 7: yc = LNc(y1 + a); fc1 + GELU (wa)
        gemv_phase<false, ALLX>(wa, g, (const bf16_t*)L.w1, sa.b0, sa.b1, d, 1, a.fc1, RowsOut{ml + mo.g, f, ml + mo.g, f, ml + mo.g},
                          [&] { ln_to_xb<NC, ALLX>(g, (const bf16_t*)(ml + mo.y1), (const bf16_t*)(ml + mo.a_c), sa, (bf16_t*)(ml + mo.yc), eps); },
                          [&] { gemv_request(wa, g, (const bf16_t*)L.w2, f, a.fc2, 0); small_request<NC>(sa, g, L.b2, a.fc2.rows, false, nullptr, nullptr); });
        xcd_barrier<ALLX>(g);
        // ---- 8: fc2 (wb)
        gemv_phase<false, ALLX>(wa, g, (const bf16_t*)L.w2, sa.b0, sa.b1, f, 0, a.fc2, RowsOut{ml + mo.a_f, d, ml + mo.a_f, d, ml + mo.a_f},
                          [&] { row_to_xb<ALLX>(g, (const bf16_t*)(ml + mo.g), f); },
                          [&] {
                              if (l + 1 < nl) { gemv_request(wa, g, (const bf16_t*)p->layers[l + 1].wqkv, d, a.qkv, 0); small_request<NC>(sa, g, p->layers[l + 1].bqkv, a.qkv.rows, false, L.ln2_w, L.ln2_b); }
                              else { gemv_request(wa, g, (const bf16_t*)p->head_w, d, a.head, 0); small_request<NC>(sa, g, p->head_b, a.head.rows, true, L.ln2_w, L.ln2_b); }
                          });
        xcd_barrier<ALLX>(g);
    }
    // ---- LM heads on LN2 of the last layer (wa)
    {
        char* mp = a.mail + (size_t)(nl - 1) * mo.stride;
        gemv_phase<true, ALLX>(wa, g, (const bf16_t*)p->head_w, sa.b0, sa.b1, d, 0, a.head, RowsOut{p->logits, p->vocab, p->logits, p->vocab, p->logits},
                         [&] { ln_to_xb<NC, ALLX>(g, (const bf16_t*)(mp + mo.yc), (const bf16_t*)(mp + mo.a_f), sa, (bf16_t*)(mp + mo.h_in), eps, false); }, [] {});
    }
    if (g.dead && g.t == 0) stg32(p->logits, PB_DECODE1_POISON);       // a barrier lost an arrival: the row is not a result
    D1_STAMP(g);
    if (g.wg == 0 && g.t == 0) *(D1_GLOBAL int*)a.pos = i;          // every participant read *pos before its first barrier
}

template <int NC, bool ALLX>
int launch_nc(const D1Args& a, int hd, size_t lds, hipStream_t st) {
    const int grid = ALLX ? Cfg<true>::NP : 8 * Cfg<false>::NP;
    if (hd == 64) hipLaunchKernelGGL((decode1_kernel<NC, 64, ALLX>), dim3(grid), dim3(D1_THREADS), lds, st, a);
    else hipLaunchKernelGGL((decode1_kernel<NC, 128, ALLX>), dim3(grid), dim3(D1_THREADS), lds, st, a);
    PB_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// key splits per head: as many (head, split) items as keep the participants evenly loaded, at most `maxi` per participant
static int d1_splits(int H, int np, int maxi) {
    int best = 0;
    double best_cost = 1e9;
    for (int ns = 2; ns <= PB_DECODE_MAX_SPLITS; ns *= 2) {
        const int items = (H * ns + np - 1) / np;
        if (items > maxi) break;
        const double cost = (double)items / ns;
        if (cost <= best_cost) { best_cost = cost; best = ns; }
    }
    return best;
}

// the dealing of a GEMV phase's units (Gv): the group size G | KC that minimises the instructions of a half-wave
static int d1_deal(int N, int K, int np, int maxu, Gv* out) {
    const int rows = N / np, KC = K / 256;
    double best = 1e18;
    Gv gv{rows, KC, 0, 0, 0u};
    for (int G = 1; G <= KC; ++G) {
        if (KC % G) continue;
        const int nvr = rows * KC / G, vph = (nvr + D1_HW - 1) / D1_HW, UPH = G * vph;
        if (UPH > maxu && maxu % G) continue;                     // further rounds repeat the flush pattern of the first
        const double cost = 5.0 * UPH + 10.0 * vph;
        if (cost < best) { best = cost; gv.G = G; gv.UPH = UPH; }
    }
    if (!gv.G) return -1;
    for (int k = 0; k < maxu; ++k) if ((k + 1) % gv.G == 0) gv.flush |= 1u << k;
    *out = gv;
    return 0;
}

#ifdef PB_D1_STAMPS
extern "C" int pb_decode1_stamps(unsigned long long* out) {                    // diagnostic build: 2 x 512 cycle stamps of the last token
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(d1_stamps), sizeof(unsigned long long) * 1024) == hipSuccess ? 0 : -1;
}
extern "C" int pb_decode1_sub(unsigned long long* out, int reset) {             // 32 sub-phase accumulators of participant 0 (all tokens since the reset)
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(d1_sub), sizeof(unsigned long long) * 32) != hipSuccess) return -1;
    if (reset) { unsigned long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(d1_sub), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif

static int d1_args(const pb_decode_plan* plan, int allx, D1Args* a) {
    const int d = plan->d, H = plan->H, hd = H > 0 ? d / H : 0;
    const int np = allx ? Cfg<true>::NP : Cfg<false>::NP, maxu = allx ? Cfg<true>::MAXU : Cfg<false>::MAXU, maxi = allx ? Cfg<true>::MAXI : Cfg<false>::MAXI;
    if (plan->dtype != PB_BF16 || H <= 0 || d % H != 0 || (hd != 64 && hd != 128) || d % 256 != 0 || d > 1024 || !plan->attn_part) return -1;
    if (plan->ffn % 256 != 0 || plan->ffn > 8192 || plan->vocab % (2 * np) != 0 || plan->n_layers <= 0 || plan->S_enc <= 0) return -1;
    if ((3 * d) % (2 * np) || d % (2 * np) || plan->ffn % (2 * np)) return -1;
    if (d1_deal(3 * d, d, np, maxu, &a->qkv) || d1_deal(d, d, np, maxu, &a->dd) || d1_deal(plan->ffn, d, np, maxu, &a->fc1) ||
        d1_deal(d, plan->ffn, np, maxu, &a->fc2) || d1_deal(plan->vocab, d, np, maxu, &a->head)) return -1;
    const Gv* all[5] = {&a->qkv, &a->dd, &a->fc1, &a->fc2, &a->head};
    for (const Gv* gv : all) if (gv->rows * (gv->KC / gv->G) > 512 || gv->rows > 128) return -1;    // pl[]; rows out: <= 64 pairs (or f32 rows) on the service wave
    a->ns = d1_splits(H, np, maxi);
    if (!a->ns) return -1;
    if (a->head.rows > 64) return -1;
    a->lds_xb = 2 * (plan->ffn > d ? plan->ffn : d);
    int off = 0;
    auto take = [&](int bytes) { const int o = off; off += (bytes + 127) / 128 * 128; return o; };
    const int row = 2 * d, recs = 4 * H * a->ns * (hd + 4);
    Mail& m = a->mo;
    m.h_in = take(row); m.q_s = take(row); m.rec_s = take(recs); m.a_s = take(row); m.y1 = take(row); m.q_c = take(row); m.rec_c = take(recs);
    m.a_c = take(row); m.yc = take(row); m.g = take(2 * plan->ffn); m.a_f = take(row); m.stride = off;
    return 0;
}

int64_t pb_decode1_mail_bytes(const pb_decode_plan* plan, int allx) {
    D1Args a;
    if (d1_args(plan, allx, &a)) return 0;
    return (int64_t)a.mo.stride * plan->n_layers;
}

int pb_decode1_supported(const pb_decode_plan* plan, int allx) {
    D1Args a;
    return d1_args(plan, allx, &a) == 0;
}

int pb_decode1_launch(const pb_decode_plan* plan_host, const pb_decode_plan* plan_dev, int* pos, const int16_t* tok, unsigned* sync, void* mail, int xcd, void* stream) {
    const int d = plan_host->d, H = plan_host->H, hd = d / H, allx = xcd < 0;
    D1Args a;
    PB_REQUIRE(d1_args(plan_host, allx, &a) == 0, "pb_decode1_launch: shape not covered");
    a.plan = plan_dev; a.pos = pos; a.tok = tok; a.sync = sync; a.xcd = xcd; a.mail = (char*)mail;
    const int maxi = allx ? Cfg<true>::MAXI : Cfg<false>::MAXI;
    const size_t lds = (size_t)a.lds_xb + sizeof(float) * (512 + 64 + (size_t)maxi * D1_WAVES * (hd + 4) + (size_t)H * a.ns * (hd + 5));
    hipStream_t st = (hipStream_t)stream;
    switch ((d / 256) * 2 + allx) {
        case 2: return launch_nc<1, false>(a, hd, lds, st);
        case 3: return launch_nc<1, true>(a, hd, lds, st);
        case 4: return launch_nc<2, false>(a, hd, lds, st);
        case 5: return launch_nc<2, true>(a, hd, lds, st);
        case 6: return launch_nc<3, false>(a, hd, lds, st);
        case 7: return launch_nc<3, true>(a, hd, lds, st);
        case 8: return launch_nc<4, false>(a, hd, lds, st);
        default: return launch_nc<4, true>(a, hd, lds, st);
    }
}
