// Persistent fused logits + cross-entropy forward (bf16, row-major operands): arguments and launcher.
#pragma once
#include "common.h"

struct HeadFwdArgs {
    const bf16_t* A;      // h [M, K]
    const bf16_t* B;      // W [N, K]
    const float* bias;    // [N] or NULL
    bf16_t* C;            // softmax numerators [M, ldc] or NULL
    long lda, ldb, ldc;
    int M, N, K;
    int store_cols;       // columns of C that exist (those from N on receive zeros)
    f32x4* part4;         // [np slots][M rows] of {reference value, sum exp(z - reference), -inf, argmax as int bits}
    int np;
    unsigned long long* stamps;   // diagnostic: 8 words per workgroup, NULL in normal use
};

// geometry 0: 256x128 tiles, 4 waves, two workgroups per CU; 1: 256x256 tiles, 8 waves, one workgroup per CU
bool head_fwd_supported(const HeadFwdArgs& a);
int head_fwd_launch(const HeadFwdArgs& a, int geometry, hipStream_t st);
