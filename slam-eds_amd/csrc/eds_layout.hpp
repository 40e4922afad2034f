// HBM data layout of libeds_hip (shared by host code and gfx950 kernels).
//
// One handle holds B alignment slots.  Everything per point is structure-of-arrays
// with a padded point stride Np (multiple of 256) so that a wavefront's 64 lanes
// read/write 64 consecutive elements of each plane:
//
//   X,Y,Z   f64 [B][Np]   back-projected point kp = (x/rho', y/rho', 1/rho'), rho' = idp + 1e-5
//                         (reference PhotometricError.hpp:95-106).  fp64 because 1e-5 px of
//                         sub-pixel phase at u ~ 640..1280 needs > 24 mantissa bits.
//   x,y,rho f32 [B][Np]   normalised coords and RAW inverse depth for the flow model
//                         (PhotometricError.hpp:114-122,136-137 use idp without eps)
//   gx,gy,w f32 [B][Np]   log-image gradient and point weight (KeyFrame.hpp:80,90)
//   mhat    f32 [B][Np]   normalised model a_i.v / n_block  (6-DoF solvers: velocity is fixed)
//   frame   f32 [B][Hp*Wp]  brightness-increment frame (EventFrame.hpp:59): H x W padded to multiples of 4 PLUS a margin of
//                         EDS_FRAME_MARGIN replicated pixels on every side, in 4x4-pixel tiles of one 64-byte sector each
//                         (or row-major); with the margin every clamped 4x4 neighbourhood is a plain interior read
//   r       f32 [B][Np]   residuals of the last pass
//   J       f32 [12][B][Np]  Jacobian planes (6 used by the pose-only solvers) — column-major
//                         per point so the residual/Jacobian kernel's stores are coalesced
//   pose    f64 [B][EDS_POSE_STRIDE]  per-slot constants of one pass, see PoseBlock below
//   G       f64 [B][EDS_MAX_BLOCKS][36]  A^T A of each residual block (constant per keyframe;
//                         gives ||m||^2 = v^T G v + 1e-3 and A^T m = G v in O(1))
//   part    f64 [B][nseg][EDS_RED_K]  per-workgroup partial sums of the reduction kernel
//   ncstat  f64 [B][EDS_MAX_BLOCKS][8]  PhotometricErrorNC only: per block 1/||E||, then sum_j E_j J'_j (6) / ||E||^3
#pragma once
#include <stddef.h>
#include <stdint.h>

#define EDS_MAX_BLOCKS 16          // upper bound on options.num_threads residual blocks
#define EDS_POSE_STRIDE 256        // doubles per slot
#define EDS_POINT_ALIGN 256        // Np is a multiple of this
#define EDS_TPB 256                // threads per workgroup of the streaming kernels (4 wavefronts)

// offsets (in doubles) inside a slot's pose block
#define EDS_PB_R 0                 // 9   rotation, row-major (Eigen toRotationMatrix of q, PhotometricError.hpp:163)
#define EDS_PB_T 9                 // 3   translation px
#define EDS_PB_K 12                // 4   fx fy cx cy
#define EDS_PB_V 16                // 6   velocity vx
#define EDS_PB_Q 22                // 4   quaternion xyzw
#define EDS_PB_HUBER 26            // 1   per-point Huber threshold (0 = off)
#define EDS_PB_NB 27               // 1   number of residual blocks
#define EDS_PB_NE 28               // 1   points per block (N / nb, Tracker.cpp:178)
#define EDS_PB_N 29                // 1   number of points
#define EDS_PB_NCMODE 30           // 1   1 = PhotometricErrorNC residual: sampled brightness L2-normalised per block too
#define EDS_PB_FRAME 31            // 1   slot whose frame storage this alignment samples (its own, or another slot's: eds_trk_share_event_frame)
#define EDS_PB_PV 32               // 36  d(unit-norm plus)/d delta = (I - v v^T/|v|^2)/|v|  (PhotometricError.hpp:32-54)
#define EDS_PB_BLK 68              // 8 per block: inv_n, gvec[6] = G v / n^3, S
#define EDS_PB_BLK_STRIDE 8
#define EDS_PB_D 200               // 9   R - I, formed WITHOUT cancellation from the quaternion: the kernels work on the
                                   //     small displacement (R - I) m + t rho' so that fp32 resolves 1e-6 px (eds_device.hpp)

// reduction widths: upper triangle of J^T J + J^T r + sum r^2 (+ count of Huber-active points)
#define EDS_RED_N6 28              // 21 + 6 + 1
#define EDS_RED_K6 32
#define EDS_RED_N12 91             // 78 + 12 + 1
#define EDS_RED_K12 128
#define EDS_RED_K 128              // stride of the partial-sum records

// ---- per-point planes of a keyframe: slices of one [EDS_KF_PLANES][B][Np] allocation ---------------------------------
#define EDS_KF_X 0
#define EDS_KF_Y 1
#define EDS_KF_RHO 2
#define EDS_KF_GX 3
#define EDS_KF_GY 4
#define EDS_KF_W 5
#define EDS_KF_F0X 6
#define EDS_KF_F0Y 7
#define EDS_KF_CELL0 8             // int32 bit patterns
#define EDS_KF_PLANES 9

// ---- frame allocation ------------------------------------------------------------------------------------------
#define EDS_FRAME_MARGIN 4         // replicated border pixels (= one tile) on every side of the padded frame
#if defined(__HIPCC__)
#define EDS_LAYOUT_HD __host__ __device__
#else
#define EDS_LAYOUT_HD
#endif
// padded extent of a frame dimension, and the element index of LOGICAL pixel (r, c), -MARGIN <= r < Hp - MARGIN
EDS_LAYOUT_HD static inline int eds_frame_extent(int n) { return ((n + 3) & ~3) + 2 * EDS_FRAME_MARGIN; }
EDS_LAYOUT_HD static inline size_t eds_frame_index(int r, int c, int Wp, int tiled) {
    const int rr = r + EDS_FRAME_MARGIN, cc = c + EDS_FRAME_MARGIN;
    return tiled ? ((size_t)((rr >> 2) * (Wp >> 2) + (cc >> 2)) * 16 + ((rr & 3) << 2) + (cc & 3)) : ((size_t)rr * Wp + cc);
}

// ---- strips: the second frame layout of the persistent pose-only kernel (round 3) ----------------------------------------
// A bicubic patch row (4 taps) read out of 4x4 tiles is TWO aligned 16-byte pieces plus a barrel shift, and a patch touches
// 3.06 sectors of 64 bytes.  Strips: the allocation (margins included) cut into 8-column-wide full-height strips — a strip row is
// 32 bytes, rows follow one another — and stored TWICE, the second copy cut 4 columns later.  Any 4 consecutive columns then lie
// inside one strip of one of the two copies: a patch row is ONE 16-byte read at a 4-byte-aligned address, the four rows of a patch
// are 128 contiguous bytes (2.5 sectors on average), no shift.  Measured (tools/ubench_gather_lds.hip, 256 frames in flight):
// 37.7 G patches/s against 32.0 for the tiles — the gather is what bounds the solve.  Costs 2 x the frame's bytes beside the
// tiles, filled by one conversion launch when a solve finds a slot's strips out of date (eds_strips.hip).
EDS_LAYOUT_HD static inline int eds_strips_count(int Wp) { return (Wp + 7) >> 3; }
EDS_LAYOUT_HD static inline size_t eds_strips_copy_elems(int Hp, int Wp) { return (size_t)eds_strips_count(Wp) * Hp * 8; }
// ROW PHASES.  The L2 fills whole 128-byte lines from the fabric, and the gather is bound by exactly those fills
// (profiles/r03_summary.md: 1.64 fabric requests per gathered patch on plain strips, where the 128 bytes of a patch start at
// any multiple of 32).  With `phases` = 2 or 4 every (column) copy exists `phases` times, copy p holding allocation row r at position
// r - p: a patch whose first row is ra reads copy p = ra mod phases, where its 128 bytes start at a multiple of 64 / 128 bytes —
// with 4 phases a patch is exactly ONE line.  Measured (tools/ubench_gather_lds.hip, 256 frames in flight): 36.9 / 40.5 / 58.0 G
// patches/s for 1 / 2 / 4 phases.  A frame then costs 2 x phases x its bytes (8 copies: 10 MB per 640x480 frame — memory a
// 288 GB part has; the conversion writes them once per frame, eds_strips.hip).
// byte offset, from the start of a frame's strips, of the 4 taps at allocation row ra + k (k = 0 .. 3 rows: add 32 k), allocation
// columns ca .. ca + 3; copy_bytes = bytes of one copy, phases in {1, 2, 4}
EDS_LAYOUT_HD static inline unsigned eds_strips_row_offset(int ra, int ca, int Hp, unsigned copy_bytes, int phases = 1) {
    const int copy = ((ca & 7) + 3) >> 3;            // columns 5, 6, 7 (mod 8) would cross a strip of column copy 0
    const int cc = ca - 4 * copy;
    const int p = ra & (phases - 1);
    return (unsigned)(2 * p + copy) * copy_bytes + (unsigned)((((cc >> 3) * Hp + (ra - p)) << 5) + ((cc & 7) << 2));
}

// ---- when the strip copies are made (eds_strips.hip) -------------------------------------------------------------------------
// `stale` sampled slots of a solve's range have no up-to-date copy, `fresh` of those hold a frame no solve has sampled yet.
// policy 0 (default, "reuse"): copies for frames that are solved AGAIN — a launch whose frames are (mostly) new samples the tiles;
// a few new frames among many converted ones (< 1 in 11: a tile launch for the whole range would cost more than their copies) are
// converted at once.  policy 1 ("eager"): convert whatever is stale.  policy 2 ("never").
// Returns 0: sample the tiles, 1: the copies are current, use them, 2: convert the stale ones, then use them.
EDS_LAYOUT_HD static inline int eds_strips_decide(int policy, int stale, int fresh, int count) {
    if (policy == 2) return 0;
    if (stale == 0) return 1;
    if (policy == 0 && fresh * 11 > count) return 0;
    return 2;
}
