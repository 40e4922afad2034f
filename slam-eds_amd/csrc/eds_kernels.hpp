// Kernel argument block + launcher declarations (layout: eds_layout.hpp).
#pragma once
#include <hip/hip_runtime.h>

#include "eds_layout.hpp"

struct EdsArrays {
    // keyframe SoA  [B][Np]  (see eds_device.hpp PointKf)
    const float* x; const float* y; const float* rho;      // normalised coords, raw inverse depth
    const float* gx; const float* gy; const float* w;      // log-image gradient, point weight
    const float* f0x; const float* f0y; const int* cell0;  // keyframe pixel u0 = fx x + cx split into cell + fraction
    // the nine planes above are consecutive slices of ONE allocation, in the order of EDS_KF_* (eds_layout.hpp): kernels that
    // read all of them address kf + k * kf_plane and keep one base pointer in scalar registers instead of nine
    const float* kf; size_t kf_plane;
    float* mhat;                                            // normalised model (pose-only solvers)
    // frames [B][Hp*Wp]
    const float* frame;
    const float* strips;   // [B][2 * strip_phases][strips copy] or null: the strip layout of the same frames (eds_layout.hpp), valid for the slots a solve asked for
    int strip_phases;      // row phases of the strip copies: 1, 2 or 4
    // per-slot constants
    double* pose;        // [B][EDS_POSE_STRIDE]
    double* G;           // [B][EDS_MAX_BLOCKS][36]
    // per-pass outputs
    float* r;            // [B][Np]
    float* rmap;         // pinned host mirror of the first EDS_RHOST_SLOTS rows of `r` (device view) or null: team launches store the kept
                         // residuals there themselves (no mirror launch behind a lone solve)
    // Small solves (round 6): EVERY workgroup of the launch, once everything it writes is out (system-scope fence), stores `done_tag` into
    // its word done[slot * EDS_DONE_WORDS + workgroup-of-the-alignment] in pinned host memory — the host sees the solve finish, records AND
    // residual mirror, without the runtime (eds_capi.hip: wait_stream).  null: a batch; the stream is waited for.
    unsigned* done; unsigned done_tag;
    float* J;            // [12][B][Np]
    double* part;        // [B][max_seg][EDS_RED_K]
    double* ncstat;      // [B][EDS_MAX_BLOCKS][8]  PhotometricErrorNC block statistics (eds_layout.hpp)
    int B, Np, H, W, max_seg;
    int Hp, Wp, tiled;   // frame allocation: padded to multiples of 4; 4x4-tiled or row-major (eds_device.hpp FrameView)
};

#define EDS_RHOST_SLOTS 8
#define EDS_DONE_WORDS 32          // workgroups per alignment at most: teams x candidate groups (4 x 8, 8 x 4, 16 x 1)

void eds_launch_gram(const EdsArrays& A, int slot, int nb, hipStream_t st, const float* new_rho = nullptr);
void eds_launch_model(const EdsArrays& A, int first, int count, int nchunk, hipStream_t st);
void eds_launch_resjac(const EdsArrays& A, int sampling, int ncols, int first, int count, int nchunk, hipStream_t st);
// PhotometricErrorNC (reference PhotometricErrorNC.hpp:124-192): block statistics of the sampled brightness, then the
// per-point correction of r and of the pose columns; between the residual/Jacobian pass and the reduction
void eds_launch_nc_normalise(const EdsArrays& A, int first, int count, int nb, int nchunk, hipStream_t st);
// one-pass residual/Jacobian launches fetch their patch rows non-temporally when the lines they touch (128 B per point on the strip copies) plus
// the planes they write (28 B per point) exceed the Infinity Cache: nothing of a frame would survive to the next pass, and the planes should
static inline bool eds_resjac_nt_rule(int count, int Np) { return (long long)count * Np * 156ll > 256ll * 1024 * 1024; }
// points a lane of the reduction folds: 4 (16-byte loads) for the 6-column pass over one residual block, else 1
static inline int eds_reduce_points_per_lane(int ncols, int nb_red, int knob = 4) { return (ncols == 6 && nb_red == 1) ? (knob == 8 ? 8 : 4) : 1; }
void eds_launch_reduce(const EdsArrays& A, int ncols, int first, int count, int nseg, int nb_red, int cpb, hipStream_t st, int ppl = 4);
