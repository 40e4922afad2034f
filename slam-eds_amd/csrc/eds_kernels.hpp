// Kernel argument block + launcher declarations (layout: eds_layout.hpp).
#pragma once
#include <hip/hip_runtime.h>

#include "eds_layout.hpp"

struct EdsArrays {
    // keyframe SoA  [B][Np]
    const double* X; const double* Y; const double* Z;
    const float* x; const float* y; const float* rho;
    const float* gx; const float* gy; const float* w;
    float* mhat;
    // frames [B][H*W]
    const float* frame;
    // per-slot constants
    double* pose;        // [B][EDS_POSE_STRIDE]
    double* G;           // [B][EDS_MAX_BLOCKS][36]
    // per-pass outputs
    float* r;            // [B][Np]
    float* J;            // [12][B][Np]
    double* part;        // [B][max_seg][EDS_RED_K]
    int B, Np, H, W, max_seg;
};

void eds_launch_gram(const EdsArrays& A, int slot, int nb, hipStream_t st);
void eds_launch_model(const EdsArrays& A, int first, int count, int nchunk, hipStream_t st);
void eds_launch_resjac(const EdsArrays& A, int sampling, int ncols, int first, int count, int nchunk, hipStream_t st);
void eds_launch_reduce(const EdsArrays& A, int ncols, int first, int count, int nseg, int nb_red, int cpb, hipStream_t st);
