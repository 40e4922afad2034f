// Microbenchmark 6 (round 3): can the quad-cooperative row gather land in LDS without passing through VGPRs?
//
// gfx950 has global_load_lds_dwordx4: a wave-instruction whose 64 lanes fetch 16 bytes each from their own global addresses and
// deposit them at M0-base + lane * 16 in LDS.  With the STRIP layout of ubench_gather_quad.hip (8-pixel-wide full-height strips, two
// copies 4 pixels apart: every patch row is ONE 16-byte read at a 4-byte-aligned address) a patch row then needs no VGPRs while in
// flight, no barrel shift, and the landing zone doubles as the patch cache (a lane that hits simply does not issue its load).
//
// Part 1 (correctness): do 4-byte-aligned global addresses work with the LDS-direct 16-byte form, and where does lane i's data land?
// Part 2 (rate): tracker-shaped gather, 512-thread workgroups, 2 048 points per frame, 16 row loads per lane per pass:
//   W0  quad rows, 64-B tiles  -> VGPRs  (two aligned 16-B loads per row: what eds_fused6_kernel does today)
//   W1  quad rows, strips      -> VGPRs  (one unaligned 16-B load per row)
//   W2  quad rows, strips      -> LDS    (global_load_lds_dwordx4, rows read back with ds_read_b128)
//   W3  as W2 with every strip copy stored twice more, the second one row later: a patch whose first row is odd reads the shifted copy,
//       so its 128 contiguous bytes always start on a 64-byte boundary — exactly 2 sectors per patch instead of 2.5 (4 copies in all)
//   W4  four row phases (8 copies in all): the 128 bytes of a patch always start on a 128-byte boundary = ONE L2 line per patch
//       (the L2 fills whole 128-byte lines from the fabric: profiles/r03_summary.md — 1.64 fabric requests per gathered patch on 2 copies)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* glb_ptr;

__device__ __forceinline__ void point_pos(int frame_id, int pid, int sh, int sv, int& r0, int& c0) {
    unsigned s = (frame_id * 2048 + pid) * 2654435761u + 777u;
    s = s * 1664525u + 1013904223u;
    c0 = 16 + (int)(((unsigned long long)(s >> 4) * 608) >> 28) + sh;
    s = s * 1664525u + 1013904223u;
    r0 = 16 + (int)(((unsigned long long)(s >> 4) * 448) >> 28) + sv;
}
__device__ __forceinline__ const float* tile_seg(const float* frame, int TWt, int r, int cseg) {
    return frame + ((size_t)(r >> 2) * TWt + (cseg >> 2)) * 16 + ((r & 3) << 2);
}
__device__ __forceinline__ const float* strip_row(const float* frame, size_t copy_stride, int Hp, int r, int ca) {
    const int copy = (ca & 7) > 4;
    const int cc = ca - 4 * copy;
    return frame + copy * copy_stride + ((size_t)(cc >> 3) * Hp + r) * 8 + (cc & 7);
}
struct __attribute__((packed, aligned(4))) f4u { float x, y, z, w; };

// ---- part 1 -----------------------------------------------------------------------------------------------------------------
__global__ void check_lds_direct(const float* __restrict__ src, float* __restrict__ out, int shift) {
    __shared__ __attribute__((aligned(16))) float land[4][64 * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // lane i reads 16 bytes at element (37 * i + shift): 4-byte aligned, arbitrary otherwise
    const float* g = src + 37 * (lane + 64 * wave) + shift;
    __builtin_amdgcn_global_load_lds((glb_ptr)g, (lds_ptr)&land[wave][0], 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    const float4 v = *reinterpret_cast<const float4*>(&land[wave][4 * lane]);
    float* o = out + 4 * threadIdx.x;
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}

// ---- part 2 -----------------------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(512) void gather(const float* __restrict__ buf, float* out, int passes, size_t frame_stride, int amp) {
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Hp = 488, TWt = 162;
    const float* __restrict__ frame = buf + (size_t)b * 8 * frame_stride;
    __shared__ __attribute__((aligned(16))) float land[MODE >= 2 ? 8 : 1][MODE >= 2 ? 16 : 1][64 * 4];      // [wave][j * 4 + q][lane * 4]: 128 KB
    float acc = 0.f;
    const int row = tid & 3;
    for (int r = 0; r < passes; ++r) {
        const int sh = ((r * 7) % 11 - 5) * amp / 5, sv = ((r * 5) % 9 - 4) * amp / 4;
        if (MODE == 0) {
            float4 va[16], vb[16];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    int r0, c0;
                    point_pos(b, j * 512 + (tid & ~3) + q, sh, sv, r0, c0);
                    const int ca = (c0 - 1) & ~3;
                    va[4 * j + q] = *reinterpret_cast<const float4*>(tile_seg(frame, TWt, r0 - 1 + row, ca));
                    vb[4 * j + q] = *reinterpret_cast<const float4*>(tile_seg(frame, TWt, r0 - 1 + row, ca + 4));
                }
#pragma unroll
            for (int k = 0; k < 16; ++k) acc += va[k].x + va[k].y + va[k].z + va[k].w + vb[k].x + vb[k].y + vb[k].z;
        } else if (MODE == 1) {
            f4u v[16];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    int r0, c0;
                    point_pos(b, j * 512 + (tid & ~3) + q, sh, sv, r0, c0);
                    v[4 * j + q] = *reinterpret_cast<const f4u*>(strip_row(frame, frame_stride, Hp, r0 - 1 + row, c0 - 1));
                }
#pragma unroll
            for (int k = 0; k < 16; ++k) acc += v[k].x + v[k].y + v[k].z + v[k].w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    int r0, c0;
                    point_pos(b, j * 512 + (tid & ~3) + q, sh, sv, r0, c0);
                    const float* src = strip_row(frame, frame_stride, Hp, r0 - 1 + row, c0 - 1);
                    if (MODE >= 3) {                     // row-phase copy: rows stored rp positions earlier, behind the two plain copies
                        const int rp = (r0 - 1) & (MODE == 3 ? 1 : 3);
                        src = strip_row(frame + rp * 2 * frame_stride, frame_stride, Hp, r0 - 1 + row - rp, c0 - 1);
                    }
                    __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)&land[wave][4 * j + q][0], 16, 0, 0);
                }
            __builtin_amdgcn_s_waitcnt(0);                // vmcnt(0): the rows have landed (each lane reads back only what it wrote)
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const float4 v = *reinterpret_cast<const float4*>(&land[wave][k][4 * lane]);
                acc += v.x + v.y + v.z + v.w;
            }
        }
    }
    if (acc == 12345.678f) out[b * 512 + tid] = acc;
}

int main(int argc, char** argv) {
    // ---- part 1
    {
        const int n = 64 * 1024;
        std::vector<float> h(n);
        for (int i = 0; i < n; ++i) h[i] = (float)i;
        float *d, *o;
        (void)hipMalloc(&d, n * 4); (void)hipMalloc(&o, 256 * 4 * 4);
        (void)hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
        int bad = 0;
        for (int shift = 0; shift < 4; ++shift) {
            check_lds_direct<<<1, 256>>>(d, o, shift);
            std::vector<float> r(256 * 4);
            (void)hipMemcpy(r.data(), o, 256 * 4 * 4, hipMemcpyDeviceToHost);
            for (int t = 0; t < 256; ++t)
                for (int k = 0; k < 4; ++k)
                    if (r[4 * t + k] != (float)(37 * t + shift + k)) { if (bad < 8) printf("  shift %d thread %d elem %d: got %g want %d\n", shift, t, k, r[4 * t + k], 37 * t + shift + k); ++bad; }
        }
        printf("part 1: global_load_lds_dwordx4 with 4-byte-aligned addresses, lane i -> base + 16 i: %s (%d mismatches)\n", bad ? "WRONG" : "ok", bad);
        (void)hipFree(d); (void)hipFree(o);
    }
    // ---- part 2
    const size_t frame_stride = 162 * 122 * 16;
    float* buf; float* out;
    const int Gmax = 512;
    const int amp = argc > 1 ? atoi(argv[1]) : 5;
    printf("shift amplitude %d px, 512-thread workgroups (one per CU up to 256), 16 row loads per lane per pass\n", amp);
    (void)hipMalloc(&buf, 8 * Gmax * frame_stride * 4); (void)hipMemset(buf, 0, 8 * Gmax * frame_stride * 4);
    (void)hipMalloc(&out, (size_t)Gmax * 512 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int Gs[] = {64, 128, 256, 512};
    for (int G : Gs) {
        const int passes = 64;
        printf("G = %4d frames in flight, %d passes x 2 048 patches each\n", G, passes);
#define RUN(M, name) do { \
            gather<M><<<G, 512>>>(buf, out, passes, frame_stride, amp); (void)hipDeviceSynchronize(); \
            (void)hipEventRecord(e0); for (int r = 0; r < 3; ++r) gather<M><<<G, 512>>>(buf, out, passes, frame_stride, amp); \
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3; \
            const double patches = (double)G * 2048 * passes; \
            printf("  %s: %8.3f ms  %7.2f G patches/s  %6.2f us per pass\n", name, ms, patches / ms / 1e6, ms * 1e3 / passes * (G > 256 ? 256.0 / G : 1.0)); } while (0)
        RUN(0, "W0 quad rows, tiles  -> VGPR");
        RUN(1, "W1 quad rows, strips -> VGPR");
        RUN(2, "W2 quad rows, strips -> LDS ");
        RUN(3, "W3 W2 + 2 row phases (x4)   ");
        RUN(4, "W4 W2 + 4 row phases (x8)   ");
    }
    return 0;
}
