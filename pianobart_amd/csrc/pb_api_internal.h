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

// pb_decode1.hip: one decoder token as ONE persistent kernel, on the 32 workgroups of one XCD (xcd >= 0) or on 128 workgroups over all
// 8 XCDs (xcd < 0 / allx != 0). pb_decode1_supported: 1 when the plan's shape is covered. pb_decode1_launch: plan_dev = a DEVICE copy of
// the plan, pos / tok device words, sync: 512 unsigned, zero at the start of a prompt (word 16 is raised when a barrier lost an arrival,
// and the logits row then starts with PB_DECODE1_POISON), mail: pb_decode1_mail_bytes device bytes (the rows handed from phase to phase).
#define PB_DECODE1_POISON 0x7fc0deadu
int pb_decode1_supported(const pb_decode_plan* plan, int allx);
int64_t pb_decode1_mail_bytes(const pb_decode_plan* plan, int allx);
int pb_decode1_launch(const pb_decode_plan* plan_host, const pb_decode_plan* plan_dev, int* pos, const int16_t* tok, unsigned* sync, void* mail, int xcd, void* stream);
