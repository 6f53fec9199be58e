// Device-side BART corruptions of the pre-train step: the counterpart of Pretrainer.gen_mask
// (/root/reference/pretrain.py:211-546, live branches only): per sample one of
//   1 TokenDeletion (n=-1, :217-239)   2 TokenMask octuple-level 80/10/10 (n=0, :276-295)
//   3 SentencePermutation (:368-397)   4 TokenInfilling octuple-level, Poisson(3) (n=0, :399-436)
//   5 DocumentRotation (:508-517)
// The reference does this in serial Python on the host (20-80 ms per sample, D2H copy per sample); here
// one 256-thread workgroup corrupts one (S,8) int16 sequence entirely in LDS (S <= 2048). Byte/integer
// work, HBM traffic = one read + one write of the sequence + the f32 loss mask.
// Randomness is a counter-based Philox stream keyed by (seed, sample, purpose): same distributions as the
// reference (exact-size uniform subsets, uniform permutation of bars, Bernoulli(p/3) + Poisson(3) span
// process with <= 10 retries, uniform rotation), NOT the Mersenne-Twister bit stream of Python's `random`
// (bit parity with that stream is the oracle's job, SURVEY 7 hard part 2).
#include "pb_common.h"
#include "pb_api_internal.h"

namespace {

constexpr int CT = 256, SMAX = 2048;
struct Rows8 { int16_t v[8]; };
struct CorruptArgs {
    const int16_t* ids; int16_t* out; float* loss_mask; const int32_t* choice; int32_t* choice_out;
    int B, S; float mask_percent; uint64_t seed;
    Rows8 pad, mask; int ntok[8];
};

__device__ __forceinline__ uint32_t rnd(uint64_t seed, uint32_t sample, uint32_t purpose, uint32_t idx) {
    const uint4 r = philox4x32(idx, sample, purpose, 0xC0221u, (uint32_t)seed, (uint32_t)(seed >> 32));
    return r.x;
}
__device__ __forceinline__ uint4 rnd4(uint64_t seed, uint32_t sample, uint32_t purpose, uint32_t idx) {
    return philox4x32(idx, sample, purpose, 0xC0221u, (uint32_t)seed, (uint32_t)(seed >> 32));
}
__device__ __forceinline__ float u01(uint32_t x) { return (x >> 8) * (1.0f / 16777216.0f); }

__device__ __forceinline__ uint4 row_load(const int16_t* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ bool row_eq(const uint4& a, const uint4& b) { return a.x == b.x && a.y == b.y && a.z == b.z && a.w == b.w; }
__device__ __forceinline__ uint4 row_pack(const Rows8& r) {
    uint4 q;
    q.x = (uint16_t)r.v[0] | ((uint32_t)(uint16_t)r.v[1] << 16); q.y = (uint16_t)r.v[2] | ((uint32_t)(uint16_t)r.v[3] << 16);
    q.z = (uint16_t)r.v[4] | ((uint32_t)(uint16_t)r.v[5] << 16); q.w = (uint16_t)r.v[6] | ((uint32_t)(uint16_t)r.v[7] << 16);
    return q;
}

// python round() (banker's rounding) of a non-negative float product, as in round(max_seq_len * mask_percent)
__device__ __forceinline__ int py_round(double x) { return (int)rint(x); }

__global__ __launch_bounds__(CT) void corrupt_kernel(const CorruptArgs p) {
    __shared__ uint4 rows[SMAX];          // the input sequence, one 16-byte row per position
    __shared__ uint32_t key[SMAX];        // random keys / scratch
    __shared__ int src[SMAX];             // infilling: source index per output position (-1 MASK, -2 PAD)
    __shared__ int sh[8];
    const int b = blockIdx.x, t = threadIdx.x, S = p.S;
    const int16_t* in = p.ids + (size_t)b * S * 8;
    int16_t* out = p.out + (size_t)b * S * 8;
    float* lm = p.loss_mask + (size_t)b * S * 8;
    for (int i = t; i < S; i += CT) rows[i] = row_load(in + i * 8);
    int choice = p.choice ? p.choice[b] : 0;
    if (choice < 1 || choice > 5) choice = 1 + (int)(rnd(p.seed, b, 0, 0) % 5u);      // random.randint(1, 5)
    if (t == 0 && p.choice_out) p.choice_out[b] = choice;
    const uint4 PADR = row_pack(p.pad), MASKR = row_pack(p.mask);
    __syncthreads();
    auto emit = [&](int pos, const uint4& r, float m) {
        *reinterpret_cast<uint4*>(out + pos * 8) = r;
        f32x4 mv = {m, m, m, m};
        *reinterpret_cast<f32x4*>(lm + pos * 8) = mv; *reinterpret_cast<f32x4*>(lm + pos * 8 + 4) = mv;
    };

    if (choice == 1 || choice == 2) {
        // exact-size uniform random subset: position i is selected iff rank(key_i) < k
        for (int i = t; i < S; i += CT) key[i] = rnd(p.seed, b, 1, i);
        __syncthreads();
        const int k = choice == 1 ? (int)(S * (double)p.mask_percent) : py_round(S * (double)p.mask_percent);
        if (choice == 2) {
            const int k80 = py_round(k * 0.8), k10 = py_round(k * 0.1);
            for (int i = t; i < S; i += CT) {
                const uint32_t ki = key[i];
                int rank = 0;
                for (int j = 0; j < S; ++j) rank += (key[j] < ki) || (key[j] == ki && j < i);
                uint4 r = rows[i];
                float m = 0.f;
                if (rank < k) {
                    m = 1.f;
                    if (rank < k80) r = MASKR;
                    else if (rank < k80 + min(k10, k - k80)) {          // rand10 is sampled from the k - k80 left-overs
                        const uint4 a = rnd4(p.seed, b, 2, 2 * i), c = rnd4(p.seed, b, 2, 2 * i + 1);
                        Rows8 rr;
                        rr.v[0] = a.x % p.ntok[0]; rr.v[1] = a.y % p.ntok[1]; rr.v[2] = a.z % p.ntok[2]; rr.v[3] = a.w % p.ntok[3];
                        rr.v[4] = c.x % p.ntok[4]; rr.v[5] = c.y % p.ntok[5]; rr.v[6] = c.z % p.ntok[6]; rr.v[7] = c.w % p.ntok[7];
                        r = row_pack(rr);
                    }
                }
                emit(i, r, m);
            }
        } else {
            // deletion: kept rows are compacted in order, k PAD rows appended, loss mask = 1 from the first deleted index on
            if (t == 0) sh[0] = S;
            __syncthreads();
            for (int i = t; i < S; i += CT) {
                const uint32_t ki = key[i];
                int rank = 0;
                for (int j = 0; j < S; ++j) rank += (key[j] < ki) || (key[j] == ki && j < i);
                src[i] = rank < k ? 1 : 0;                                 // deleted flag
                if (rank < k) atomicMin(&sh[0], i);
            }
            __syncthreads();
            const int first = sh[0];
            for (int i = t; i < S; i += CT) {
                if (!src[i]) {
                    int before = 0;
                    for (int j = 0; j < i; ++j) before += src[j];
                    const int pos = i - before;
                    emit(pos, rows[i], pos >= first ? 1.f : 0.f);
                }
            }
            for (int i = S - k + t; i < S; i += CT) emit(i, PADR, i >= first ? 1.f : 0.f);
        }
    } else if (choice == 3) {
        // bars (column 0) are shuffled as units; rows keep their order inside a bar; mask = rows that changed
        for (int i = t; i < S; i += CT) key[i] = rnd(p.seed, b, 3, rows[i].x & 0xffffu);   // one key per bar value
        __syncthreads();
        for (int i = t; i < S; i += CT) {
            const uint32_t ki = key[i], bi = rows[i].x & 0xffffu;
            int pos = 0;
            for (int j = 0; j < S; ++j) {
                const uint32_t bj = rows[j].x & 0xffffu;
                pos += (bj == bi) ? (j < i) : ((key[j] < ki) || (key[j] == ki && bj < bi));
            }
            src[pos] = i;
        }
        __syncthreads();
        for (int i = t; i < S; i += CT) { const uint4 r = rows[src[i]]; emit(i, r, row_eq(r, rows[i]) ? 0.f : 1.f); }
    } else if (choice == 4) {
        // span infilling: sequential process on one lane (S steps), up to 10 attempts (pretrain.py:407-430)
        if (t == 0) {
            const float thr = p.mask_percent / 3.0f;
            int ok = 0;
            for (int att = 0; att < 10 && !ok; ++att) {
                int n = 0, i = 0; uint32_t ctr = 0;
                bool over = false;
                while (i < S) {
                    const uint4 r = rnd4(p.seed, b, 4 + att, ctr++);
                    if (u01(r.x) < thr) {
                        // Poisson(3) by inversion
                        const float u = u01(r.y);
                        float pk = 0.049787068f, cdf = pk; int pz = 0;
                        while (u > cdf && pz < 64) { ++pz; pk *= 3.0f / pz; cdf += pk; }
                        if (pz == 0) {
                            if (n + 2 > S) { over = true; break; }
                            src[n++] = i; src[n++] = -1; i += 1;
                        } else {
                            if (n + 1 > S) { over = true; break; }
                            src[n++] = -1; i += pz;
                        }
                    } else {
                        if (n + 1 > S) { over = true; break; }
                        src[n++] = i; i += 1;
                    }
                }
                if (!over) { for (int j = n; j < S; ++j) src[j] = -2; ok = 1; }
            }
            sh[1] = ok;
        }
        __syncthreads();
        if (sh[1]) {
            for (int i = t; i < S; i += CT) {
                const int s = src[i];
                const uint4 r = s >= 0 ? rows[s] : (s == -1 ? MASKR : PADR);
                emit(i, r, row_eq(r, rows[i]) ? 0.f : 1.f);
            }
        } else {
            for (int i = t; i < S; i += CT) emit(i, rows[i], 0.f);        // reference fallback: unchanged, all-zero mask
        }
    } else {
        const int ran = (int)(rnd(p.seed, b, 5, 0) % (uint32_t)S);        // random.randint(0, l-1)
        for (int i = t; i < S; i += CT) emit(i, rows[(i + ran) % S], ran != 0 ? 1.f : 0.f);
    }
}

}  // namespace

extern "C" int pb_corrupt(const int16_t* ids, int16_t* out, float* loss_mask, const int32_t* choice, int32_t* choice_out, int32_t B,
                          int32_t S, float mask_percent, uint64_t seed, const int16_t* pad_row, const int16_t* mask_row,
                          const int32_t* n_tokens, void* stream_) {
    PB_REQUIRE(S > 0 && S <= SMAX, "pb_corrupt: S=%d out of range (1..%d)", S, SMAX);
    PB_REQUIRE(pad_row && mask_row && n_tokens, "pb_corrupt: null special rows");
    if (B <= 0) return 0;
    CorruptArgs a;
    a.ids = ids; a.out = out; a.loss_mask = loss_mask; a.choice = choice; a.choice_out = choice_out; a.B = B; a.S = S;
    a.mask_percent = mask_percent; a.seed = seed;
    for (int i = 0; i < 8; ++i) { a.pad.v[i] = pad_row[i]; a.mask.v[i] = mask_row[i]; a.ntok[i] = n_tokens[i]; }
    hipLaunchKernelGGL(corrupt_kernel, dim3(B), dim3(CT), 0, (hipStream_t)stream_, a);
    PB_LAUNCH_CHECK();
    return 0;
}
