// Argument block shared by the 3-D convolution forward kernels (conv3d.hip: generic LDS-tiled
// implicit GEMM; conv3d_wres.hip: the weight-resident persistent kernel for Cin = 32, Cout = 64).
#pragma once
#include "common.h"

struct Conv3dArgs {
    const bf16* x; const bf16* w;
    int B, D, H, W, Cin, Cout;
    const float* shift;           // [Cout] bias (nullptr = 0)
    int kc;                       // generic kernel: channels per LDS chunk (32 or 64), set by launch3d
    float* stats;                 // [MM_REPL][2][Cout] sum / sumsq of (acc + shift)   (nullptr)
    float* out_f32;               // [B][D][H][W][Cout]
    bf16* out_bf16;
    unsigned mtw, mth, mtd;       // conv3d_wres: floor(2^32 / d) + 1 for d = tiles along W, H, D (q = mulhi(x, m) = x / d for x < 2^24)
};

// conv3d_wres.hip
bool conv3d_wres_applies(const Conv3dArgs& a);
int launch3d_wres(const Conv3dArgs& a, hipStream_t st);

// conv3d_stream.hip: weight-streaming kernel for (Cin, Cout) = (64, 128), (128, 64), (64, 32)
bool conv3d_stream_applies(const Conv3dArgs& a);
int launch3d_stream(const Conv3dArgs& a, hipStream_t st);
