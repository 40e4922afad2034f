// Device-side building blocks (gfx950 / CDNA4, wave64) shared by the streaming kernels
// (eds_kernels.hip) and the persistent per-alignment solver (eds_fused.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "eds_layout.hpp"

namespace edsd {

typedef float float4u __attribute__((ext_vector_type(4), aligned(4)));   // dword-aligned 16-byte load
typedef float float2u __attribute__((ext_vector_type(2), aligned(4)));

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ---------------------------------------------------------------------------------------
// Frame sampling.  Indices clamp to the border exactly like ceres::Grid2D::GetValue
// (reference use: PhotometricError.hpp:110-111).
// ---------------------------------------------------------------------------------------
// Catmull-Rom cubic Hermite through p1 (x=0), p2 (x=1): value and derivative
// (ceres CubicHermiteSpline; same spline restated at reference src/utils/globalFuncs.h:192-207).
__device__ __forceinline__ void hermite(float p0, float p1, float p2, float p3, float x, float& f, float& df) {
    const float a = 0.5f * (-p0 + 3.0f * p1 - 3.0f * p2 + p3);
    const float b = 0.5f * (2.0f * p0 - 5.0f * p1 + 4.0f * p2 - p3);
    const float c = 0.5f * (p2 - p0);
    f = p1 + x * (c + x * (b + x * a));
    df = c + x * (2.0f * b + 3.0f * a * x);
}

// A frame in HBM.  Two layouts:
//   row-major  pixel (r,c) at r*Wp + c
//   tiled      4x4-pixel tiles of 16 floats = ONE 64-byte HBM sector each; tile (r>>2, c>>2) at
//              ((r>>2)*TW + (c>>2))*16, pixel at + (r&3)*4 + (c&3).
// A bicubic 4x4 neighbourhood at a random offset spans 4 rows x 1.19 sectors = 4.75 sectors of a
// row-major frame but only (1 + 3/4)^2 = 3.06 tiles, so the tiled frame cuts the gather traffic
// (the bound of every pass) by a third and halves the touched cache footprint.
// The allocation is padded to multiples of 4 with border-replicated pixels, which is exactly what
// Grid2D's index clamp returns there.
struct FrameView {
    const float* __restrict__ base;   // LOGICAL pixel (0, 0): the allocation starts EDS_FRAME_MARGIN rows and columns earlier
    int H, W;          // logical size (clamp range)
    int Hp, Wp;        // allocated size (multiples of 4, margins included)
    int TW;            // tiles per tile-row = Wp / 4
    int tiled;
};
// view of slot `slot`'s frame inside the [B][Hp*Wp] allocation
__device__ __forceinline__ FrameView make_frame_view(const float* frames, int slot, int H, int W, int Hp, int Wp, int tiled) {
    FrameView f;
    f.H = H; f.W = W; f.Hp = Hp; f.Wp = Wp; f.TW = Wp >> 2; f.tiled = tiled;
    f.base = frames + (size_t)slot * Hp * Wp + eds_frame_index(0, 0, Wp, tiled);
    return f;
}
// element offset of logical pixel (r, c) from f.base; valid for -MARGIN <= r < Hp - MARGIN (arithmetic shifts)
__device__ __forceinline__ ptrdiff_t frame_index(const FrameView& f, int r, int c) {
    return f.tiled ? ((ptrdiff_t)((r >> 2) * f.TW + (c >> 2)) * 16 + ((r & 3) << 2) + (c & 3)) : ((ptrdiff_t)r * f.Wp + c);
}

// Loads the 4x4 neighbourhood rows r0-1..r0+2, cols c0-1..c0+2, indices clamped to the frame like Grid2D::GetValue.
// The replicated margin makes the clamp a clamp of the ORIGIN: for r0 <= -2 all four rows read row 0, for r0 >= H all
// read row H-1, and anything in between lies inside the allocation — so there is one code path, no border branch.
__device__ __forceinline__ void load_patch16(const FrameView& f, int r0, int c0, float (&p)[16]) {
    r0 = clampi(r0, -2, f.H); c0 = clampi(c0, -2, f.W);
    if (f.tiled) {                                                    // per row two aligned 16-byte tile rows
        const int ca = c0 - 1, s = ca & 3;
        const int txa = ca >> 2, txb = (c0 + 2) >> 2;
        const bool s2 = s & 2, s1 = s & 1;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = r0 - 1 + k;
            const float* trow = f.base + ((ptrdiff_t)(r >> 2) * f.TW) * 16 + ((r & 3) << 2);
            const float4 a = *reinterpret_cast<const float4*>(trow + (ptrdiff_t)txa * 16);
            const float4 b = *reinterpret_cast<const float4*>(trow + (ptrdiff_t)txb * 16);
            // 8 floats [a.x a.y a.z a.w b.x b.y b.z ..] shifted left by s in {0..3}: two-stage barrel shift
            const float t0 = s2 ? a.z : a.x, t1 = s2 ? a.w : a.y, t2 = s2 ? b.x : a.z, t3 = s2 ? b.y : a.w, t4 = s2 ? b.z : b.x;
            p[4 * k + 0] = s1 ? t1 : t0;
            p[4 * k + 1] = s1 ? t2 : t1;
            p[4 * k + 2] = s1 ? t3 : t2;
            p[4 * k + 3] = s1 ? t4 : t3;
        }
    } else {                                                          // row-major: four dword-aligned 16-byte row segments
        const float* base = f.base + (ptrdiff_t)(r0 - 1) * f.Wp + (c0 - 1);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4u v = *reinterpret_cast<const float4u*>(base + (ptrdiff_t)k * f.Wp);
            p[4 * k + 0] = v.x; p[4 * k + 1] = v.y; p[4 * k + 2] = v.z; p[4 * k + 3] = v.w;
        }
    }
}

// ---------------------------------------------------------------------------------------
// Quad-cooperative patch gather (round 2).  A 4x4 patch read by ONE lane is 8 scattered 16-byte loads whose 64 lanes touch 64
// different tiles per wave-instruction; the vector memory pipeline retires such an instruction at ~2.2 clocks per lane, and that
// address rate — not bytes — bounds the pass together with the fabric's miss rate (tools/ubench_gather_quad.hip: 1.4-1.5x).
// Here the four lanes of a quad share their four points: for each point q of the quad, lane j loads ROW j of q's patch (the
// same two aligned 16-byte pieces per row as load_patch16).  The four lanes of one instruction then touch the rows of ONE patch
// — one or two tiles instead of four unrelated ones — so the instruction covers ~1.75 lines per quad instead of 4.  The row's
// cubic Hermite runs on the lane that loaded it, and a 4x4 DPP transpose hands every lane the four row results of ITS OWN
// point: per-lane arithmetic is what it was (four row splines, then the column splines), plus ~40 cross-lane moves.
// ---------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true); }
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true)); }
// value of lane Q (0..3) of this lane's quad
template <int Q>
__device__ __forceinline__ int quad_bcast_i(int v) { return dpp_i<Q * 0x55>(v); }
template <int Q>
__device__ __forceinline__ float quad_bcast_f(float v) { return dpp_f<Q * 0x55>(v); }
// In-quad 4x4 transpose: on entry m[q] is what THIS lane (row `lane & 3`) computed for the quad's point q; on return m[k] is
// what lane k computed for this lane's own point.  Two exchange rounds (partner lane ^ 1, then lane ^ 2), 12 instructions.
__device__ __forceinline__ void quad_transpose(float (&m)[4], int lane) {
    const bool b0 = lane & 1, b1 = lane & 2;
#pragma unroll
    for (int a = 0; a < 4; a += 2) {
        const float send = b0 ? m[a] : m[a + 1];
        const float recv = dpp_f<0xB1>(send);                // quad_perm [1,0,3,2]
        if (b0) m[a] = recv; else m[a + 1] = recv;
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const float send = b1 ? m[c] : m[c + 2];
        const float recv = dpp_f<0x4E>(send);                // quad_perm [2,3,0,1]
        if (b1) m[c] = recv; else m[c + 2] = recv;
    }
}
// Origin of a patch packed for the in-quad broadcast: clamped like load_patch16 (rows -2..H, columns -2..W), biased by 2.
// (bits 0-15: column + 2; 16-28: row + 2; 29-30: (column - 1) & 3, the barrel-shift amount of the patch's rows, where every lane that
// loads a row of this patch can turn each bit into a select mask with ONE v_bfe_i32; bit 31 is the caller's "gather it" flag)
__device__ __forceinline__ int pack_origin(const FrameView& f, int r0, int c0) {
    const int c = clampi(c0, -2, f.W);
    return (((c - 1) & 3) << 29) | ((clampi(r0, -2, f.H) + 2) << 16) | (c + 2);
}
// Row `jr` (0..3) of the patch at a packed origin: the two aligned 16-byte tile pieces that hold columns c0-1 .. c0+2.
// `tiles` is the START of the frame's allocation (tile (0,0) of the margin), offsets are unsigned 32-bit element counts.
__device__ __forceinline__ void load_patch_row(const float* __restrict__ tiles, int TW, int origin, int jr, float4& a, float4& b) {
    const unsigned r = (unsigned)((origin >> 16) & 0x1fff) + (unsigned)(jr + EDS_FRAME_MARGIN - 3);   // allocation row, >= 1
    const unsigned cc = (unsigned)(origin & 0xffff) + (unsigned)(EDS_FRAME_MARGIN - 2);              // allocation column of c0
    // byte offsets from the start of the allocation, 32-bit: uniform base + per-lane offset (global_load ... saddr form)
    const unsigned row_b = (r >> 2) * ((unsigned)TW * 64u) + ((r & 3u) << 4);
    const char* base = reinterpret_cast<const char*>(tiles);
    a = *reinterpret_cast<const float4*>(base + (row_b + (((cc - 1u) >> 2) << 6)));
    b = *reinterpret_cast<const float4*>(base + (row_b + (((cc + 2u) >> 2) << 6)));
}
// three registers whose content does not matter and that cost NO instruction: outputs of an empty asm (freeze(undef), below, is
// materialised as v_mov 0 once it meets a phi)
__device__ __forceinline__ void dont_care3(float4& b) { asm("" : "=v"(b.x), "=v"(b.y), "=v"(b.z)); b.w = b.z; }
// a register whose content does not matter (no instruction emitted): the unselected arm of a bit_select
__device__ __forceinline__ float4 dont_care4() {
    float x = 0.f;
    x = __builtin_nondeterministic_value(x);             // freeze(undef): any value, but a defined one — no instruction
    return make_float4(x, x, x, x);
}
// the 4 taps starting at column c0-1 out of the two pieces (same two-stage barrel shift as load_patch16)
// (bit-mask selects, one v_bfi_b32 each: written as ?: on the vector components the compiler turns the shift into a
// dynamically indexed private array, i.e. scratch memory traffic inside the point loop)
__device__ __forceinline__ float bit_select(int mask, float yes, float no) {
    // v_bfi_b32 D = (S0 & S1) | (~S0 & S2), spelled out: from the C expression the compiler derives ~mask arithmetically (mask = -(bit)
    // gives ~mask = bit - 1) and then no longer recognises the pattern — three instructions (and, and, or) per select instead of one
    float r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(mask), "v"(yes), "v"(no));
    return r;
}
// the same selection for the LAST select in front of a 16-byte store: a compare + v_cndmask_b32 (one instruction too), whose results the
// register allocator can place in the four consecutive registers the store needs — the asm outputs above pin their registers early
// (13 fewer VGPR spills, -4.5 % kernel time)
__device__ __forceinline__ float flag_select(bool pick, float yes, float no) { return pick ? yes : no; }
__device__ __forceinline__ void shift_patch_row(const float4& a, const float4& b, int origin, float (&t)[4]) {
    const int m2 = __builtin_amdgcn_sbfe(origin, 30, 1), m1 = __builtin_amdgcn_sbfe(origin, 29, 1);    // all ones / zero: bit 1 / bit 0 of (c0 - 1) & 3
    const float t0 = bit_select(m2, a.z, a.x), t1 = bit_select(m2, a.w, a.y), t2 = bit_select(m2, b.x, a.z), t3 = bit_select(m2, b.y, a.w),
                t4 = bit_select(m2, b.z, b.x);
    t[0] = bit_select(m1, t1, t0); t[1] = bit_select(m1, t2, t1); t[2] = bit_select(m1, t3, t2); t[3] = bit_select(m1, t4, t3);
}
// The same barrel shift as ONE v_bitop3_b32 per select through the gfx950 builtin (round 3; truth table 0xCA = S0 ? S1 : S2 bit by
// bit): no inline asm in the point loop, so the scheduler may move the selects and the register allocator may put their results
// wherever the consumer wants them (the 16-byte cache store, the pairs of the packed row splines) — and, being a target intrinsic,
// nothing the SLP vectoriser can turn into <2 x i32> logic plus shuffles (which it did to the same selects written as and / xor).
__device__ __forceinline__ float mask_select(int m, float yes, float no) {
    return __int_as_float(__builtin_amdgcn_bitop3_b32(m, __float_as_int(yes), __float_as_int(no), 0xCA));
}
__device__ __forceinline__ void shift_patch_row3(const float4& a, const float4& b, int origin, float (&t)[4]) {
    const int m2 = __builtin_amdgcn_sbfe(origin, 30, 1), m1 = __builtin_amdgcn_sbfe(origin, 29, 1);
    const float t0 = mask_select(m2, a.z, a.x), t1 = mask_select(m2, a.w, a.y), t2 = mask_select(m2, b.x, a.z), t3 = mask_select(m2, b.y, a.w),
                t4 = mask_select(m2, b.z, b.x);
    t[0] = mask_select(m1, t1, t0); t[1] = mask_select(m1, t2, t1); t[2] = mask_select(m1, t3, t2); t[3] = mask_select(m1, t4, t3);
}

// LDS patch cache of the quad path: [point][row][4 taps], one 16-byte unit per (point, row), units XOR-swizzled by the
// quad index so that the 16 lanes one ds_read_b128 / ds_write_b128 services together hit 16 different bank groups.
__device__ __forceinline__ int patch_unit(int point, int row) { return ((point << 2) + row) ^ (((point >> 2) & 3) << 2); }

// Bicubic value and derivatives from a register-resident 4x4 patch (ay: row phase, ax: col phase).
__device__ __forceinline__ void bicubic_patch(const float (&p)[16], float ay, float ax, float& E, float& Erow, float& Ecol) {
    float f[4], d[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) hermite(p[4 * k], p[4 * k + 1], p[4 * k + 2], p[4 * k + 3], ax, f[k], d[k]);
    hermite(f[0], f[1], f[2], f[3], ay, E, Erow);
    float unused;
    hermite(d[0], d[1], d[2], d[3], ay, Ecol, unused);
}

// 2x2 neighbourhood rows r0, r0+1, cols c0, c0+1 (clamped the same way)
__device__ __forceinline__ void load_patch4(const FrameView& f, int r0, int c0, float (&p)[4]) {
    r0 = clampi(r0, -1, f.H - 1); c0 = clampi(c0, -1, f.W - 1);
    if (!f.tiled) {
        const float* base = f.base + (ptrdiff_t)r0 * f.Wp + c0;
        const float2u a = *reinterpret_cast<const float2u*>(base);
        const float2u b = *reinterpret_cast<const float2u*>(base + f.Wp);
        p[0] = a.x; p[1] = a.y; p[2] = b.x; p[3] = b.y;
    } else {
        p[0] = f.base[frame_index(f, r0, c0)]; p[1] = f.base[frame_index(f, r0, c0 + 1)];
        p[2] = f.base[frame_index(f, r0 + 1, c0)]; p[3] = f.base[frame_index(f, r0 + 1, c0 + 1)];
    }
}
__device__ __forceinline__ void bilinear_patch(const float (&p)[4], float ay, float ax, float& E, float& Erow, float& Ecol) {
    const float top = p[0] + ax * (p[1] - p[0]), bot = p[2] + ax * (p[3] - p[2]);
    E = top + ay * (bot - top);
    Erow = bot - top;
    Ecol = (1.0f - ay) * (p[1] - p[0]) + ay * (p[3] - p[2]);
}

// ---------------------------------------------------------------------------------------
// Per-point geometry in fp32 — WITHOUT the 1e-4 px sub-pixel noise a naive fp32 projection has at
// u ~ 640..1280.  The reference computes P = R kp + t, u = fx Px/Pz + cx in fp64
// (PhotometricError.hpp:157-168).  Here everything is expressed through the SMALL displacement
//     d = (R - I) m + t rho'          m = (x, y, 1),  rho' = idp + 1e-5,   P = (m + d) / rho'
// (R - I comes from the quaternion without cancellation, eds_math.hpp), so that
//     u - u0 = fx (d0 - x d2) / (1 + d2),   u0 = fx x + cx  (the point's own keyframe pixel)
// is a few pixels known to ~1e-7 relative, and u0 is carried as integer cell + fp32 fraction
// computed once in fp64 when the keyframe is uploaded.  Measured against the fp64 oracle this
// keeps residuals within 2e-7 of max|r| (tests/test_parity_gpu.py) at half the fp64 issue cost.
// ---------------------------------------------------------------------------------------
struct PoseF {                     // wave-uniform (SGPRs)
    float D[9], t[3], fx, fy;
};
// Forces a wave-uniform value into an SGPR (values read from LDS otherwise occupy a VGPR per lane).
__device__ __forceinline__ float uniformf(float x) {
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x)));
}
// pb: a slot's pose block (doubles) in global memory or LDS
__device__ __forceinline__ void load_pose(const double* pb, PoseF& ps) {
#pragma unroll
    for (int i = 0; i < 9; ++i) ps.D[i] = uniformf((float)pb[EDS_PB_D + i]);
#pragma unroll
    for (int i = 0; i < 3; ++i) ps.t[i] = uniformf((float)pb[EDS_PB_T + i]);
    ps.fx = uniformf((float)pb[EDS_PB_K]);
    ps.fy = uniformf((float)pb[EDS_PB_K + 1]);
}

// the same with the pass-dependent part (R - I, t) and the constant part (intrinsics) in different places
__device__ __forceinline__ void load_pose_rt(const double* D, const double* t, const double* pb, PoseF& ps) {
#pragma unroll
    for (int i = 0; i < 9; ++i) ps.D[i] = uniformf((float)D[i]);
#pragma unroll
    for (int i = 0; i < 3; ++i) ps.t[i] = uniformf((float)t[i]);
    ps.fx = uniformf((float)pb[EDS_PB_K]);
    ps.fy = uniformf((float)pb[EDS_PB_K + 1]);
}

// The pose of the packed point phase travels as fourteen scalar VALUES (macro below: plain locals, never a struct).  From a struct
// — array or named members alike — the vectoriser widened the loads of the scalars it had to splat for the packed multiplies into
// overlapping <2 x float> loads of the stack object, the object could then no longer be promoted to registers, and the whole pose
// went through scratch memory every pass.  rt: R - I then t (12 floats, narrowed once by the lane that made the pose), kf32 = {fx, fy}.
#define EDS_LOAD_POSE_SCALARS(rt, kf32)                                                                                          \
    const float ps_d0 = uniformf((rt)[0]), ps_d1 = uniformf((rt)[1]), ps_d2 = uniformf((rt)[2]), ps_d3 = uniformf((rt)[3]),    \
                ps_d4 = uniformf((rt)[4]), ps_d5 = uniformf((rt)[5]), ps_d6 = uniformf((rt)[6]), ps_d7 = uniformf((rt)[7]),    \
                ps_d8 = uniformf((rt)[8]), ps_t0 = uniformf((rt)[9]), ps_t1 = uniformf((rt)[10]), ps_t2 = uniformf((rt)[11]), \
                ps_fx = uniformf((kf32)[0]), ps_fy = uniformf((kf32)[1])
#define EDS_POSE_SCALARS ps_d0, ps_d1, ps_d2, ps_d3, ps_d4, ps_d5, ps_d6, ps_d7, ps_d8, ps_t0, ps_t1, ps_t2, ps_fx, ps_fy

// 1 / x: v_rcp_f32 (1 ulp) + one Newton step = 3 instructions, against the ~10 of the IEEE division sequence; the point loop
// divides twice per point and is instruction-bound when a launch holds few alignments
__device__ __forceinline__ float fast_recip(float x) {
    const float r = __builtin_amdgcn_rcpf(x);
    return r * (2.0f - x * r);
}

struct PointKf {                   // per-point keyframe constants (SoA in HBM, registers in the persistent kernel)
    float x, y, rhop;              // normalised coords, idp + 1e-5
    float f0x, f0y;                // fractional part of the keyframe pixel (u0, v0)
    int cell0;                     // integer part, packed (row << 16) | (col & 0xffff)
};
struct PointGeom {
    float Px, Py, Pz, iz, un, vn;
    int r0, c0;
    float ay, ax;
};
__device__ __forceinline__ void split_rel(float s, int base, int& cell, float& phase) {
    // s = keyframe fraction + displacement [px]; robust to NaN / inf / huge (the reference has no
    // in-bounds test, PhotometricError.hpp:157-172: Grid2D simply clamps)
    const float fl = floorf(s);
    const bool sane = fabsf(s) < 60000.0f;
    cell = base + (sane ? (int)fl : (s > 0.0f ? 60000 : -60000));
    phase = sane ? s - fl : 0.0f;
}
__device__ __forceinline__ void project_point(const PoseF& ps, const PointKf& k, PointGeom& g) {
    const float d0 = ps.D[0] * k.x + ps.D[1] * k.y + ps.D[2] + ps.t[0] * k.rhop;
    const float d1 = ps.D[3] * k.x + ps.D[4] * k.y + ps.D[5] + ps.t[1] * k.rhop;
    const float d2 = ps.D[6] * k.x + ps.D[7] * k.y + ps.D[8] + ps.t[2] * k.rhop;
    const float s = 1.0f + d2;                   // Pz rho'
    const float is = fast_recip(s);
    const float Z = fast_recip(k.rhop);
    g.un = (k.x + d0) * is;
    g.vn = (k.y + d1) * is;
    g.Px = (k.x + d0) * Z;
    g.Py = (k.y + d1) * Z;
    g.Pz = s * Z;
    g.iz = k.rhop * is;
    const float du = ps.fx * (d0 - k.x * d2) * is;      // u - u0   (column, PhotometricError.hpp:167)
    const float dv = ps.fy * (d1 - k.y * d2) * is;      // v - v0   (row,    PhotometricError.hpp:168)
    split_rel(k.f0x + du, (int)(short)(k.cell0 & 0xffff), g.c0, g.ax);
    split_rel(k.f0y + dv, k.cell0 >> 16, g.r0, g.ay);
}

struct PointProj {
    float Px, Py, Pz;              // point in the event-frame camera
    float g0, g1, g2;              // dE/dP  (gradE_P of SURVEY §8a)
    float E;
};
// dE/dP from the sampled frame derivatives.
__device__ __forceinline__ void finish_point(const PoseF& ps, const PointGeom& g, float E, float Er, float Ec, PointProj& o) {
    const float dx = ps.fx * Ec, dy = ps.fy * Er;
    o.g0 = dx * g.iz;
    o.g1 = dy * g.iz;
    o.g2 = -(dx * g.un + dy * g.vn) * g.iz;
    o.Px = g.Px; o.Py = g.Py; o.Pz = g.Pz;
    o.E = E;
}

// Projects, samples (straight from HBM / L2) and forms dE/dP for one point.
template <int SAMPLING>
__device__ __forceinline__ void project_sample(const FrameView& frame, const PoseF& ps, const PointKf& k, PointProj& o) {
    PointGeom g;
    project_point(ps, k, g);
    float E, Er, Ec;
    if (SAMPLING == 0) {
        float p[16];
        load_patch16(frame, g.r0, g.c0, p);
        bicubic_patch(p, g.ay, g.ax, E, Er, Ec);
    } else {
        float p[4];
        load_patch4(frame, g.r0, g.c0, p);
        bilinear_patch(p, g.ay, g.ax, E, Er, Ec);
    }
    finish_point(ps, g, E, Er, Ec, o);
}

// The same for four consecutive points held by the four lanes of a quad, bicubic sampler, tiled frame: the quad-cooperative gather
// of the persistent kernels without a patch cache — lane j loads ROW j of each of the quad's four patches, runs that row's spline,
// and a 4x4 DPP transpose returns the four row results of a point to its own lane.  EVERY lane of the quad must call it (a lane
// without a point passes valid = false: its patch travels as origin 0 and its result is not used).  Same arithmetic, same order of
// operations as project_sample<0>.
__device__ __forceinline__ void project_sample_quad(const FrameView& frame, const float* __restrict__ tiles, const PoseF& ps, const PointKf& k,
                                                    bool valid, int lane, PointProj& o) {
    PointGeom g;
    project_point(ps, k, g);
    const int org = valid ? pack_origin(frame, g.r0, g.c0) : 0;
    const int o0 = quad_bcast_i<0>(org), o1 = quad_bcast_i<1>(org), o2 = quad_bcast_i<2>(org), o3 = quad_bcast_i<3>(org);
    const int oq[4] = {o0, o1, o2, o3};
    const float x0 = quad_bcast_f<0>(g.ax), x1 = quad_bcast_f<1>(g.ax), x2 = quad_bcast_f<2>(g.ax), x3 = quad_bcast_f<3>(g.ax);
    const float xq[4] = {x0, x1, x2, x3};
    const int jr = lane & 3;
    float4 ra[4], rb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) load_patch_row(tiles, frame.TW, oq[q], jr, ra[q], rb[q]);
    float f[4], d[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float t[4];
        shift_patch_row(ra[q], rb[q], oq[q], t);
        hermite(t[0], t[1], t[2], t[3], xq[q], f[q], d[q]);
    }
    quad_transpose(f, lane);
    quad_transpose(d, lane);
    float E, Er, Ec, unused;
    hermite(f[0], f[1], f[2], f[3], g.ay, E, Er);
    hermite(d[0], d[1], d[2], d[3], g.ay, Ec, unused);
    finish_point(ps, g, E, Er, Ec, o);
}

// The same on the STRIP copies of the frame (eds_layout.hpp; round 3): the owner of a point computes the byte offset of its patch's
// first row once, it goes round the quad, and lane j reads row j of each of the quad's four patches with ONE 16-byte load at a
// 4-byte-aligned address (no second piece, no barrel shift).  `sbase`: start of the frame's strips; Hp: allocated rows;
// copy_bytes / phases: the strip geometry.  Same arithmetic after the loads as project_sample_quad.
template <bool NT = false>
__device__ __forceinline__ void project_sample_quad_strips(const FrameView& frame, const char* __restrict__ sbase, int Hp, unsigned copy_bytes, int phases,
                                                           const PoseF& ps, const PointKf& k, bool valid, int lane, PointProj& o) {
    PointGeom g;
    project_point(ps, k, g);
    const int ra = clampi(g.r0, -2, frame.H) + (EDS_FRAME_MARGIN - 1), ca = clampi(g.c0, -2, frame.W) + (EDS_FRAME_MARGIN - 1);
    const int off = valid ? (int)eds_strips_row_offset(ra, ca, Hp, copy_bytes, phases) : 0;
    const int o0 = quad_bcast_i<0>(off), o1 = quad_bcast_i<1>(off), o2 = quad_bcast_i<2>(off), o3 = quad_bcast_i<3>(off);
    const int oq[4] = {o0, o1, o2, o3};
    const float x0 = quad_bcast_f<0>(g.ax), x1 = quad_bcast_f<1>(g.ax), x2 = quad_bcast_f<2>(g.ax), x3 = quad_bcast_f<3>(g.ax);
    const float xq[4] = {x0, x1, x2, x3};
    const unsigned jr32 = 32u * (unsigned)(lane & 3);
    float4u t[4];
#pragma unroll
    // NT: the rows are fetched with the non-temporal bit — for ONE-PASS launches over frames that do not fit the Infinity Cache anyway
    // (the caller's rule): the lines are then not kept, and what the pass WRITES (r and the Jacobian planes, read by the reduction next)
    // stays in the cache instead (tools/bench_reduce.py: the reduction behind it 44 -> 36 us at 4 096 alignments).  Never for the solvers,
    // whose passes re-read their lines (profiles/r04_ubench_sector.txt).
    for (int q = 0; q < 4; ++q) {
        const float4u* src = reinterpret_cast<const float4u*>(sbase + ((unsigned)oq[q] + jr32));
        if (NT) t[q] = __builtin_nontemporal_load(src); else t[q] = *src;
    }
    float f[4], d[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) hermite(t[q].x, t[q].y, t[q].z, t[q].w, xq[q], f[q], d[q]);
    quad_transpose(f, lane);
    quad_transpose(d, lane);
    float E, Er, Ec, unused;
    hermite(f[0], f[1], f[2], f[3], g.ay, E, Er);
    hermite(d[0], d[1], d[2], d[3], g.ay, Ec, unused);
    finish_point(ps, g, E, Er, Ec, o);
}

// SE(3) left-perturbation row: J = -w [gradE_P, P x gradE_P]  (= DSO's row, CoarseTracker.cpp:311-321)
__device__ __forceinline__ void jacobian6(const PointProj& pp, float w, float (&J)[6]) {
    J[0] = -w * pp.g0;
    J[1] = -w * pp.g1;
    J[2] = -w * pp.g2;
    J[3] = -w * (pp.Py * pp.g2 - pp.Pz * pp.g1);
    J[4] = -w * (pp.Pz * pp.g0 - pp.Px * pp.g2);
    J[5] = -w * (pp.Px * pp.g1 - pp.Py * pp.g0);
}

// a_i = -(gx df0/dv + gy df1/dv): the row of the linear model m = A v
// (PhotometricError.hpp:114-122,136-143).
__device__ __forceinline__ void model_row(float x, float y, float rho, float gx, float gy, float (&a)[6]) {
    a[0] = gx * rho;
    a[1] = gy * rho;
    a[2] = -(gx * x + gy * y) * rho;
    a[3] = -(gx * x * y + gy * (1.0f + y * y));
    a[4] = gx * (1.0f + x * x) + gy * x * y;
    a[5] = -gx * y + gy * x;
}

// Residual block of point i for N points in nb blocks (Tracker.cpp:178-195).
__device__ __forceinline__ int block_of(int i, int ne, int nb) {
    if (ne <= 0) return nb - 1;
    const int k = i / ne;
    return k < nb ? k : nb - 1;
}

// ---------------------------------------------------------------------------------------
// Wavefront reduce-scatter: K (power of two, >= 32) partial sums per lane, 64 lanes.
// Each butterfly step halves the number of live values per lane, so the whole reduction
// costs K-1 (+1) cross-lane exchanges instead of 6*K.  No MFMA: this is a tall-skinny
// N x 7 -> 7 x 7 contraction.  On return lane l holds, in v[0 .. max(K/64,1)-1], the
// wavefront totals of value indices  wave_red_index<K>(l, j).
// ---------------------------------------------------------------------------------------
template <int K, int STEP>
struct WaveRedStep {
    static __device__ __forceinline__ void run(float* v, int lane) {
        constexpr int HALF = K >> (STEP + 1);
        if (HALF > 0) {
            const bool upper = (lane >> STEP) & 1;
#pragma unroll
            for (int j = 0; j < HALF; ++j) {
                const float send = upper ? v[j] : v[j + HALF];
                const float keep = upper ? v[j + HALF] : v[j];
                v[j] = keep + __shfl_xor(send, 1 << STEP, 64);
            }
        } else {
            v[0] += __shfl_xor(v[0], 1 << STEP, 64);
        }
        WaveRedStep<K, STEP + 1>::run(v, lane);
    }
};
template <int K>
struct WaveRedStep<K, 6> {
    static __device__ __forceinline__ void run(float*, int) {}
};
template <int K>
__device__ __forceinline__ void wave_reduce_scatter(float* v, int lane) {
    WaveRedStep<K, 0>::run(v, lane);
}
template <int K>
__device__ __forceinline__ int wave_red_index(int lane, int j) {
    int idx = j;
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const int half = K >> (s + 1);
        if (half > 0) idx += ((lane >> s) & 1) * half;
    }
    return idx;
}

// Accumulates one point's contribution: upper triangle of J J^T (row-major pairs a<=b),
// then J r, then r^2 — the layout of the EDS_RED_* records.
template <int NC>
__device__ __forceinline__ void accumulate_normal(float* acc, const float (&J)[NC], float r, float hw, float cost_term) {
    int o = 0;
#pragma unroll
    for (int a = 0; a < NC; ++a) {
        const float ja = hw * J[a];
#pragma unroll
        for (int b = a; b < NC; ++b) acc[o++] += ja * J[b];
    }
#pragma unroll
    for (int a = 0; a < NC; ++a) acc[o++] += hw * J[a] * r;
    acc[o] += cost_term;
}

// One point of a pose-only pass, shared by the persistent kernels (eds_fused.hip, eds_stream6.hip): sample from the
// register-resident taps, residual r = w (mhat - E), 1x6 SE(3) row, optional per-point Huber weight (extension, cf.
// reference CoarseTracker.cpp:445), and the contribution to the 28 running sums.  Returns r.
// ... from the sampled value and derivatives (shared by the lane-per-point and the quad-cooperative gathers)
__device__ __forceinline__ float point_row6_sampled(const PoseF& ps, const PointGeom& pg, float E, float Er, float Ec, float w, float mhat,
                                                    float tau, float* acc);
template <int SAMPLING, int NTAP>
__device__ __forceinline__ float point_row6(const PoseF& ps, const PointGeom& pg, float (&tap)[NTAP], float w, float mhat, float tau,
                                            float* acc) {
    float E, Er, Ec;
    if (SAMPLING == 0) bicubic_patch(reinterpret_cast<float(&)[16]>(tap), pg.ay, pg.ax, E, Er, Ec);
    else bilinear_patch(reinterpret_cast<float(&)[4]>(tap), pg.ay, pg.ax, E, Er, Ec);
    return point_row6_sampled(ps, pg, E, Er, Ec, w, mhat, tau, acc);
}
__device__ __forceinline__ float point_row6_sampled(const PoseF& ps, const PointGeom& pg, float E, float Er, float Ec, float w, float mhat,
                                                    float tau, float* acc) {
    PointProj pp;
    finish_point(ps, pg, E, Er, Ec, pp);
    const float r = w * (mhat - pp.E);
    float J[6];
    jacobian6(pp, w, J);
    float hw = 1.0f, ct = r * r;
    if (tau > 0.0f) {
        const float ar = fabsf(r);
        if (ar > tau) hw = tau / ar;
        ct = hw * r * r * (2.0f - hw);
    }
    accumulate_normal<6>(acc, J, r, hw, ct);
    return r;
}

// ---------------------------------------------------------------------------------------
// Round 3: the point phase of the pose-only persistent kernel on an instruction diet.  That phase is VALU-issue-bound (two
// wavefronts per SIMD, one wave64 instruction per 4 clocks), and gfx950 issues v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 — two
// fp32 operations per lane — at the same rate as their scalar forms.  So everything below works on PAIRS (ext_vector float2, which
// the backend keeps in aligned register pairs): two points of a lane through the projection, two patch rows through the row spline,
// {value, column-derivative} through the column spline, and the 28 running sums as 12 pairs + 4 scalars.  The arithmetic is the
// same closed form as above, re-associated where that removes instructions (noted at each place); parity tolerances are unchanged.
// ---------------------------------------------------------------------------------------
typedef float f2 __attribute__((ext_vector_type(2)));
typedef int i2 __attribute__((ext_vector_type(2)));

// Two independent Catmull-Rom splines at once (element k of every argument belongs to spline k).  With a2 = 2a, b2 = 2b, c2 = 2c of
// hermite():  f = p1 + (x/2)(c2 + x(b2 + x a2)),  f' = c2/2 + (x/2)(2 b2 + 3x a2);  xh = x/2 and x3 = 3x are passed in because the
// caller shares them between splines.  14 packed instructions for two values and two derivatives (hermite(): 17 scalar ones each).
__device__ __forceinline__ void hermite_pair(f2 p0, f2 p1, f2 p2, f2 p3, f2 x, f2 xh, f2 x3, f2& f, f2& df) {
    const f2 a2 = (p3 - p0) + 3.0f * (p1 - p2);
    const f2 b2 = 4.0f * p2 + (-5.0f * p1 + (2.0f * p0 - p3));
    const f2 c2 = p2 - p0;
    f = p1 + xh * (c2 + x * (b2 + x * a2));
    df = 0.5f * c2 + xh * (x3 * a2 + (b2 + b2));
}

// Geometry of two points of one lane that the sampler and the row need: the reference's P = R kp + t, u = fx Px/Pz + cx
// (PhotometricError.hpp:157-168) in the small-displacement form of project_point — without P itself: the SE(3) row only needs
// un = Px/Pz, vn = Py/Pz and iz = 1/Pz (row6_accumulate), so the second reciprocal and three products of project_point go.
struct PairGeom { f2 iz, un, vn, ax, ay; };
__device__ __forceinline__ void project_pair(float D0, float D1, float D2, float D3, float D4, float D5, float D6, float D7, float D8, float T0, float T1,
                                             float T2, float fx, float fy, f2 x, f2 y, f2 rhop, f2 f0x, f2 f0y, int cell_a, int cell_b, PairGeom& g,
                                             int (&r0)[2], int (&c0)[2]) {
    // (one wave-uniform operand per instruction: VOP3P reads a single SGPR pair, a second scalar would be copied to a VGPR first)
    const f2 d0 = (D0 * x + (D1 * y + T0 * rhop)) + D2;
    const f2 d1 = (D3 * x + (D4 * y + T1 * rhop)) + D5;
    const f2 d2 = (D6 * x + (D7 * y + T2 * rhop)) + D8;
    const f2 s = 1.0f + d2;                      // Pz rho'
    f2 is = {__builtin_amdgcn_rcpf(s.x), __builtin_amdgcn_rcpf(s.y)};
    is = is * (2.0f - s * is);                   // one Newton step (fast_recip)
    g.un = (x + d0) * is;
    g.vn = (y + d1) * is;
    g.iz = rhop * is;
    const f2 du = fx * (d0 - x * d2) * is;       // u - u0   (column, PhotometricError.hpp:167)
    const f2 dv = fy * (d1 - y * d2) * is;       // v - v0   (row,    PhotometricError.hpp:168)
    // split_rel for both: a non-finite or huge coordinate lands 60 000 cells out with phase 0 (Grid2D then clamps: the reference has
    // no in-bounds test) — the clamp of the COORDINATE gives exactly that (a NaN becomes -60 000: fmax / fmin return the number)
    const f2 su = __builtin_elementwise_min(__builtin_elementwise_max(f0x + du, (f2)(-60000.0f)), (f2)(60000.0f));
    const f2 sv = __builtin_elementwise_min(__builtin_elementwise_max(f0y + dv, (f2)(-60000.0f)), (f2)(60000.0f));
    const f2 flu = __builtin_elementwise_floor(su), flv = __builtin_elementwise_floor(sv);
    g.ax = su - flu;
    g.ay = sv - flv;
    const i2 cu = __builtin_convertvector(flu, i2), cv = __builtin_convertvector(flv, i2);
    c0[0] = (int)(short)(cell_a & 0xffff) + cu.x; c0[1] = (int)(short)(cell_b & 0xffff) + cu.y;
    r0[0] = (cell_a >> 16) + cv.x;                r0[1] = (cell_b >> 16) + cv.y;
}

// The 28 running sums of one lane (upper triangle of J^T J, J^T r, sum r^2) as pairs.  With the row in three pairs
// P0 = (J0, J1), P1 = (J2, J3), P2 = (J4, J5):  dg[i] = Pi * Pi (two diagonal entries), x01/x02/x12 = Pi * splat(J_k) (the four
// entries of an off-diagonal 2x2 block in two instructions), o = the entry inside each pair, b[i] = Pi * r: 12 packed + 4 scalar
// fused multiply-adds per point instead of 28 (+ 6 pre-multiplications).
struct Acc6 {
    f2 dg[3], x01[2], x02[2], x12[2], b[3];
    float o[3], cost;
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int i = 0; i < 3; ++i) { dg[i] = (f2)(0.0f); b[i] = (f2)(0.0f); o[i] = 0.0f; }
#pragma unroll
        for (int i = 0; i < 2; ++i) { x01[i] = (f2)(0.0f); x02[i] = (f2)(0.0f); x12[i] = (f2)(0.0f); }
        cost = 0.0f;
    }
    // the EDS_RED_* record order: row-major upper triangle, then J^T r, then the cost
    __device__ __forceinline__ void unpack(float (&acc)[EDS_RED_K6]) const {
        acc[0] = dg[0].x;  acc[1] = o[0];      acc[2] = x01[0].x;  acc[3] = x01[1].x;  acc[4] = x02[0].x;  acc[5] = x02[1].x;
        acc[6] = dg[0].y;  acc[7] = x01[0].y;  acc[8] = x01[1].y;  acc[9] = x02[0].y;  acc[10] = x02[1].y;
        acc[11] = dg[1].x; acc[12] = o[1];     acc[13] = x12[0].x; acc[14] = x12[1].x;
        acc[15] = dg[1].y; acc[16] = x12[0].y; acc[17] = x12[1].y;
        acc[18] = dg[2].x; acc[19] = o[2];     acc[20] = dg[2].y;
        acc[21] = b[0].x;  acc[22] = b[0].y;   acc[23] = b[1].x;   acc[24] = b[1].y;   acc[25] = b[2].x;   acc[26] = b[2].y;
        acc[27] = cost;
#pragma unroll
        for (int k = EDS_RED_N6; k < EDS_RED_K6; ++k) acc[k] = 0.0f;
    }
};

// One point of a pose-only pass from its sampled value and derivatives: residual r = w (mhat - E) and the SE(3) left-perturbation
// row J = -w [gradE_P, P x gradE_P] (= DSO's row, CoarseTracker.cpp:311-321), written WITHOUT P: with a = -w fx E_col,
// b = -w fy E_row, t = a un + b vn (un = Px/Pz, vn = Py/Pz, iz = 1/Pz, and Pz iz = 1):
//     J = [a iz, b iz, -t iz, -(vn t + b), a + un t, un b - vn a]
// 13 instructions against the 30 of finish_point + jacobian6 + the P of project_point.  HUBER: per-point Huber weight
// (extension, cf. reference CoarseTracker.cpp:445), a template parameter so that the plain path carries no branch per point.
template <bool HUBER>
__device__ __forceinline__ float row6_accumulate(float fx, float fy, float iz, float un, float vn, float E, float Er, float Ec, float w,
                                                 float mhat, float tau, Acc6& A) {
    const float nw = -w;
    const float a = nw * (fx * Ec), b = nw * (fy * Er);
    const float t = a * un + b * vn;
    f2 P0 = {a * iz, b * iz};
    f2 P1 = {-t * iz, -(vn * t) - b};
    f2 P2 = {a + un * t, un * b - vn * a};
    const float r = w * (mhat - E);
    f2 Q0 = P0, Q1 = P1, Q2 = P2;                // the weighted row (hw J); the plain row when HUBER is off
    float hr = r, ct = r * r;
    if (HUBER) {
        const float ar = fabsf(r);
        const float hw = ar > tau ? tau / ar : 1.0f;
        Q0 = hw * P0; Q1 = hw * P1; Q2 = hw * P2;
        hr = hw * r;
        ct = hw * r * r * (2.0f - hw);
    }
    A.dg[0] += Q0 * P0; A.dg[1] += Q1 * P1; A.dg[2] += Q2 * P2;
    A.o[0] += Q0.x * P0.y; A.o[1] += Q1.x * P1.y; A.o[2] += Q2.x * P2.y;
    A.x01[0] += Q0 * P1.x; A.x01[1] += Q0 * P1.y;
    A.x02[0] += Q0 * P2.x; A.x02[1] += Q0 * P2.y;
    A.x12[0] += Q1 * P2.x; A.x12[1] += Q1 * P2.y;
    A.b[0] += P0 * hr; A.b[1] += P1 * hr; A.b[2] += P2 * hr;
    A.cost += ct;
    return r;
}

}  // namespace edsd
