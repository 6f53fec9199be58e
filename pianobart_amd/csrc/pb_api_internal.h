// Internal: the public C ABI plus shared launch helpers.
#pragma once
#include "../../include/pianobart_hip.h"

// pb_gemm2.hip: bf16 fast path; returns 1 when it declines (caller falls back), 0 ok, <0 error.
int pb_gemm2_try(const pb_gemm_desc* d, void* stream);
// o_k[c] += sum over nblk rows of partials (nblk, nacc, d), k < nacc <= 2 (pb_norm.hip)
int pb_finalize_rows(const float* partials, int nblk, int d, float* out, void* stream, int nacc = 1, float* out1 = nullptr);
// storage for partial sums whose reduction is deferred to pb_defer_flush (NULL when no deferral is open or it is full): pass it to
// the kernel instead of the caller's workspace, then hand it to pb_finalize_rows as usual
float* pb_defer_alloc(size_t nfloats);
// pb_gemm_x3.hip: dtype PB_F32X3 -- f32 operands cut into bf16 (hi, lo) pairs, one bf16 GEMM over 3 K, f32 C
int pb_gemm_x3(const pb_gemm_desc* d, void* stream);
