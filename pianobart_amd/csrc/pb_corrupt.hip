// Device-side BART corruptions of the pre-train step: the counterpart of Pretrainer.gen_mask
// (/root/reference/pretrain.py:211-546, live branches only): per sample one of
//   1 TokenDeletion (n=-1, :217-239)   2 TokenMask octuple-level 80/10/10 (n=0, :276-295)
//   3 SentencePermutation (:368-397)   4 TokenInfilling octuple-level, Poisson(3) (n=0, :399-436)
//   5 DocumentRotation (:508-517)
// The reference does this in serial Python on the host (20-80 ms per sample, D2H copy per sample); here
// one 256-thread workgroup corrupts one (S,8) int16 sequence entirely in LDS (S <= 2048). Byte/integer
// work, HBM traffic = one read + one write of the sequence + the f32 loss mask.
//
// The kernel has two stages. DECIDE fills LDS with the random decisions of the chosen corruption (which
// positions are deleted / masked / replaced, the order of the bars, the span process' per-step draws, the
// rotation offset); APPLY turns decisions + input into the output rows and the loss mask. DECIDE has two
// sources: a counter-based Philox stream keyed by (seed, sample, purpose) -- same distributions as the
// reference, not the Mersenne-Twister bit stream of Python's `random` -- or a caller-supplied decision
// buffer (pb_corrupt_replay). With the reference's own decisions replayed, APPLY must reproduce the
// reference's outputs bit for bit (tests/test_corrupt_gpu.py against tests/golden/g6_gen_mask.npz).
#include "pb_common.h"
#include "pb_api_internal.h"

namespace {

constexpr int CT = 256, SMAX = 2048;
struct Rows8 { int16_t v[8]; };
struct CorruptArgs {
    const int16_t* ids; int16_t* out; float* loss_mask; const int32_t* choice; int32_t* choice_out;
    int B, S; double mask_percent; uint64_t seed;
    Rows8 pad, mask; int ntok[8];
    const int32_t* dec; int64_t dec_stride; const int16_t* rand_rows;      // replay source (NULL: Philox)
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

// python round() (banker's rounding) of a non-negative double product, as in round(max_seq_len * mask_percent)
__device__ __forceinline__ int py_round(double x) { return (int)rint(x); }

__global__ __launch_bounds__(CT) void corrupt_kernel(const CorruptArgs p) {
    __shared__ uint4 rows[SMAX];          // the input sequence, one 16-byte row per position
    __shared__ uint32_t key[SMAX];        // DECIDE scratch: random keys / bar ranks / span-process draws of one attempt
    __shared__ int src[SMAX];             // decisions per position (deleted flag / mask kind) or source index per output position
    __shared__ int sh[8];
    const int b = blockIdx.x, t = threadIdx.x, S = p.S;
    const int16_t* in = p.ids + (size_t)b * S * 8;
    int16_t* out = p.out + (size_t)b * S * 8;
    float* lm = p.loss_mask + (size_t)b * S * 8;
    const int32_t* dec = p.dec ? p.dec + (size_t)b * p.dec_stride : nullptr;
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
    // rank of position i among the S random keys (ties by index): an exact-size uniform subset is {i : rank(i) < k}
    auto rank_of = [&](int i) {
        const uint32_t ki = key[i];
        int rank = 0;
        for (int j = 0; j < S; ++j) rank += (key[j] < ki) || (key[j] == ki && j < i);
        return rank;
    };

    if (choice == 1) {
        // ---- TokenDeletion: int(l * p) rows removed, kept rows compacted in order, as many PAD rows appended, loss mask = 1
        // from the first deleted index on (pretrain.py:217-239)
        const int k = (int)(S * p.mask_percent);
        if (dec) {
            for (int i = t; i < S; i += CT) src[i] = dec[i] != 0;
        } else {
            for (int i = t; i < S; i += CT) key[i] = rnd(p.seed, b, 1, i);
            __syncthreads();
            for (int i = t; i < S; i += CT) src[i] = rank_of(i) < k;
        }
        if (t == 0) sh[0] = S;
        __syncthreads();
        for (int i = t; i < S; i += CT) if (src[i]) atomicMin(&sh[0], i);
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
    } else if (choice == 2) {
        // ---- TokenMask, octuple level: round(S p) positions selected; round(0.8 k) of them -> MASK row, round(0.1 k) of the
        // rest -> a random row (get_rand_tok), what is left keeps its row; all selected positions enter the loss
        // (pretrain.py:276-295). Decision per position: 0 untouched, 1 MASK, 2 random row, 3 kept
        if (dec) {
            for (int i = t; i < S; i += CT) src[i] = dec[i];
        } else {
            const int k = py_round(S * p.mask_percent);
            const int k80 = py_round(k * 0.8), k10 = py_round(k * 0.1);
            for (int i = t; i < S; i += CT) key[i] = rnd(p.seed, b, 1, i);
            __syncthreads();
            for (int i = t; i < S; i += CT) {
                const int rank = rank_of(i);
                src[i] = rank >= k ? 0 : (rank < k80 ? 1 : (rank < k80 + min(k10, k - k80) ? 2 : 3));   // rand10 is drawn from the k - k80 left-overs
            }
        }
        __syncthreads();
        for (int i = t; i < S; i += CT) {
            const int kind = src[i];
            uint4 r = rows[i];
            if (kind == 1) r = MASKR;
            else if (kind == 2) {
                if (p.rand_rows) r = row_load(p.rand_rows + ((size_t)b * S + i) * 8);
                else {
                    const uint4 a = rnd4(p.seed, b, 2, 2 * i), c = rnd4(p.seed, b, 2, 2 * i + 1);
                    Rows8 rr;
                    rr.v[0] = a.x % p.ntok[0]; rr.v[1] = a.y % p.ntok[1]; rr.v[2] = a.z % p.ntok[2]; rr.v[3] = a.w % p.ntok[3];
                    rr.v[4] = c.x % p.ntok[4]; rr.v[5] = c.y % p.ntok[5]; rr.v[6] = c.z % p.ntok[6]; rr.v[7] = c.w % p.ntok[7];
                    r = row_pack(rr);
                }
            }
            emit(i, r, kind != 0 ? 1.f : 0.f);
        }
    } else if (choice == 3) {
        // ---- SentencePermutation: bars (column 0) are shuffled as units; rows keep their order inside a bar; loss mask = rows
        // that changed (pretrain.py:368-397). Decision: one sort key per bar value (replay: the bar's place in the shuffled order)
        for (int i = t; i < S; i += CT) {
            const uint32_t bar = rows[i].x & 0xffffu;
            key[i] = dec ? (uint32_t)dec[bar] : rnd(p.seed, b, 3, bar);
        }
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
        // ---- TokenInfilling, octuple level: a sequential span process with up to 10 attempts (pretrain.py:399-436). Decision of
        // step s of an attempt: -1 = copy the row, p >= 0 = a span starts here with Poisson(3) length p (p = 0 inserts a MASK row
        // behind the copied one). An attempt's <= S decisions are drawn in parallel, then one lane walks them.
        const float thr = (float)(p.mask_percent / 3.0);
        int ok = 0;
        for (int att = 0; att < 10 && !ok; ++att) {
            for (int s = t; s < S; s += CT) {
                int v;
                if (dec) v = dec[(size_t)att * S + s];
                else {
                    const uint4 r = rnd4(p.seed, b, 4 + att, s);
                    v = -1;
                    if (u01(r.x) < thr) {                                    // Poisson(3) by inversion
                        const float u = u01(r.y);
                        float pk = 0.049787068f, cdf = pk; int pz = 0;
                        while (u > cdf && pz < 64) { ++pz; pk *= 3.0f / pz; cdf += pk; }
                        v = pz;
                    }
                }
                key[s] = (uint32_t)v;
            }
            __syncthreads();
            if (t == 0) {
                int n = 0, i = 0, s = 0;
                bool over = false;
                while (i < S) {
                    const int v = (int)key[s++];
                    if (v == 0) {
                        if (n + 2 > S) { over = true; break; }
                        src[n++] = i; src[n++] = -1; i += 1;
                    } else if (v > 0) {
                        if (n + 1 > S) { over = true; break; }
                        src[n++] = -1; i += v;
                    } else {
                        if (n + 1 > S) { over = true; break; }
                        src[n++] = i; i += 1;
                    }
                }
                if (!over) for (int j = n; j < S; ++j) src[j] = -2;
                sh[1] = over ? 0 : 1;
            }
            __syncthreads();
            ok = sh[1];
            __syncthreads();
        }
        if (ok) {
            for (int i = t; i < S; i += CT) {
                const int s = src[i];
                const uint4 r = s >= 0 ? rows[s] : (s == -1 ? MASKR : PADR);
                emit(i, r, row_eq(r, rows[i]) ? 0.f : 1.f);
            }
        } else {
            for (int i = t; i < S; i += CT) emit(i, rows[i], 0.f);        // reference fallback: unchanged, all-zero mask
        }
    } else {
        // ---- DocumentRotation by random.randint(0, l-1) (pretrain.py:508-517)
        const int ran = dec ? dec[0] : (int)(rnd(p.seed, b, 5, 0) % (uint32_t)S);
        for (int i = t; i < S; i += CT) emit(i, rows[(i + ran) % S], ran != 0 ? 1.f : 0.f);
    }
}

int launch(CorruptArgs& a, const int16_t* pad_row, const int16_t* mask_row, const int32_t* n_tokens, void* stream_) {
    PB_REQUIRE(a.S > 0 && a.S <= SMAX, "pb_corrupt: S=%d out of range (1..%d)", a.S, SMAX);
    PB_REQUIRE(pad_row && mask_row, "pb_corrupt: null special rows");
    if (a.B <= 0) return 0;
    for (int i = 0; i < 8; ++i) { a.pad.v[i] = pad_row[i]; a.mask.v[i] = mask_row[i]; a.ntok[i] = n_tokens ? n_tokens[i] : 1; }
    hipLaunchKernelGGL(corrupt_kernel, dim3(a.B), dim3(CT), 0, (hipStream_t)stream_, a);
    PB_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" int pb_corrupt(const int16_t* ids, int16_t* out, float* loss_mask, const int32_t* choice, int32_t* choice_out, int32_t B,
                          int32_t S, double mask_percent, uint64_t seed, const int16_t* pad_row, const int16_t* mask_row,
                          const int32_t* n_tokens, void* stream_) {
    PB_REQUIRE(n_tokens, "pb_corrupt: null n_tokens");
    CorruptArgs a;
    a.ids = ids; a.out = out; a.loss_mask = loss_mask; a.choice = choice; a.choice_out = choice_out; a.B = B; a.S = S;
    a.mask_percent = mask_percent; a.seed = seed; a.dec = nullptr; a.dec_stride = 0; a.rand_rows = nullptr;
    return launch(a, pad_row, mask_row, n_tokens, stream_);
}

extern "C" int pb_corrupt_replay(const int16_t* ids, int16_t* out, float* loss_mask, const int32_t* choice, int32_t B, int32_t S,
                                 double mask_percent, const int32_t* decisions, int64_t dec_stride, const int16_t* rand_rows,
                                 const int16_t* pad_row, const int16_t* mask_row, void* stream_) {
    PB_REQUIRE(choice && decisions, "pb_corrupt_replay: needs a choice per sample and the decision buffer");
    PB_REQUIRE(dec_stride >= pb_corrupt_replay_stride(S), "pb_corrupt_replay: dec_stride %lld < %lld", (long long)dec_stride,
               (long long)pb_corrupt_replay_stride(S));
    CorruptArgs a;
    a.ids = ids; a.out = out; a.loss_mask = loss_mask; a.choice = choice; a.choice_out = nullptr; a.B = B; a.S = S;
    a.mask_percent = mask_percent; a.seed = 0; a.dec = decisions; a.dec_stride = dec_stride; a.rand_rows = rand_rows;
    return launch(a, pad_row, mask_row, nullptr, stream_);
}

extern "C" int64_t pb_corrupt_replay_stride(int32_t S) { return (int64_t)10 * S > 65536 ? (int64_t)10 * S : 65536; }
