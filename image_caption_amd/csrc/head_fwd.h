// Persistent bf16 product C = epilogue(A B^T + bias) with row-major operands: arguments and launcher.
// mode 0: fused logits + cross-entropy forward; mode 1: bias / column scale / ReLU / dropout, bf16 output;
// mode 2: residual + dropout(z + bias), f32 output; mode 3: mode 0 against a caller-given reference value per row.
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
    // mode 1 (C required, part4 unused): y = dropout(relu?((z + bias) * (column < col_scale_n ? col_scale : 1)))
    int mode;
    float col_scale; int col_scale_n;     // col_scale_n a multiple of 64
    int relu;
    unsigned drop_thr, drop_k0, drop_k1; float drop_scale;      // thr 0: off; element index m * N + n
    // mode 2 (C is the f32 output [M, ldc], given through Cf; C itself unused): y = residual + dropout(z + bias)
    float* Cf; const float* residual; long ldr;
    // mode 3: mode 0 with C required and everything taken against a given reference per row: the partials are 8-byte
    // records [np slots][M rows] of {max_n (z - rowref) with 63 - (column in the slot) in its low 6 bits, sum_n exp(z - rowref)}
    // in the same buffer; C holds exp(z - rowref)
    const float* rowref;
    // k-panel images of the operands (NULL: read A / B as given): [K / 32 panels][rows][32] bf16, so that the 16 rows x 64
    // bytes one LDS-DMA instruction fetches are 1 KiB of consecutive bytes -- whole 128-byte lines instead of half lines
    const bf16_t* Ap; long a_rows;       // rows per panel of A's image (>= M)
    const bf16_t* Bp; long b_rows;       // rows per panel of B's image (>= N)
    int store_nt;                        // modes 0 / 3: numerator stores non-temporal (set by head_fwd_launch)
    long c_panel_rows;                   // modes 0 / 3: > 0: C is written as a column-panel image [ldc / 32][c_panel_rows][32] (ldc % 32 == 0)
    // filled by head_fwd_launch: bytes from one k-step's columns to the next, and by how much a view shrinks per k-step
    long kstepA, kstepB; int remdecA, remdecB;
    int half_tiles;                      // set by the launchers: this many tiles of the last round are walked as two 256x128 halves
    int b_tr;                            // ppipe only: B is reduction-major [K, ldb] (the weight of a data gradient, read along its output index)
};

// geometry 0: 256x128 tiles, 4 waves, two workgroups per CU; 1: 256x256 tiles, 8 waves, one workgroup per CU
bool head_fwd_supported(const HeadFwdArgs& a);
int head_fwd_launch(const HeadFwdArgs& a, int geometry, hipStream_t st);

// The persistent pipelined kernel (ppipe.hip): 256x256 tiles, 4 waves of 128x128, mode 1 only; B row-major (or its
// k-panel image) or, with b_tr, reduction-major.
bool ppipe_supported(const HeadFwdArgs& a);
int ppipe_launch(const HeadFwdArgs& a, hipStream_t st);
