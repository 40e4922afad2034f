// Microbenchmark 5: how should a sparse 4x4-patch gather be ISSUED?  (tracker-shaped: 2 048 fixed random points per 640x480
// frame, positions shifting by a few pixels per pass, one frame per workgroup, G workgroups of 256 threads)
//   V0  lane per point, 4x4-pixel tiles of 64 B, two aligned 16-B loads per patch row       (round-1 kernels)
//   V1  quad per 4 points: lane j of a quad loads ROW j of each of the quad's four patches — the 4 lanes of one
//       instruction then touch 1-2 tiles instead of 4 different ones                          (same tiles, same loads per lane)
//   V2  lane per point, 8x4-pixel tiles of 128 B (a patch overlaps 2.41 tiles instead of 3.06)
//   V3  quad per 4 points on 128-B tiles
//   V4  lane per point, 8-pixel-wide full-height STRIPS (32 B per row), stored twice (second copy shifted by 4 pixels) so that every
//       patch row is ONE unaligned 16-B load and the four rows of a patch are 128 contiguous bytes (immediate offsets)
//   V5  quad per 4 points on strips: lane j loads row j of the quad's four patches (4 loads per lane per 4 patches)
// V0-V3 issue 8 loads per lane per patch, V4/V5 four; what changes is how many DISTINCT lines one wave-instruction touches and
// how many lines a patch costs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int TILE_W>   // 4: 64-B tiles (4x4 px), 8: 128-B tiles (8 wide x 4 tall)
__device__ __forceinline__ const float* row_seg(const float* frame, int TWt, int r, int cseg) {
    // address of the aligned 4-pixel segment starting at column cseg (multiple of 4) in row r
    if (TILE_W == 4) return frame + ((size_t)(r >> 2) * TWt + (cseg >> 2)) * 16 + ((r & 3) << 2);
    return frame + ((size_t)(r >> 2) * TWt + (cseg >> 3)) * 32 + ((r & 3) << 3) + (cseg & 4);
}

__device__ __forceinline__ void point_pos(int frame_id, int pid, int sh, int sv, int& r0, int& c0) {
    unsigned s = (frame_id * 2048 + pid) * 2654435761u + 777u;
    s = s * 1664525u + 1013904223u;
    c0 = 16 + (int)(((unsigned long long)(s >> 4) * 608) >> 28) + sh;
    s = s * 1664525u + 1013904223u;
    r0 = 16 + (int)(((unsigned long long)(s >> 4) * 448) >> 28) + sv;
}

template <int TILE_W, int QUAD>
__global__ __launch_bounds__(256, 2) void gather(const float* __restrict__ buf, float* out, int passes, size_t frame_stride, int amp) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const int TWt = TILE_W == 4 ? 162 : 81;
    const float* __restrict__ frame = buf + (size_t)b * frame_stride;
    float acc = 0.f;
    for (int r = 0; r < passes; ++r) {
        const int sh = ((r * 7) % 11 - 5) * amp / 5, sv = ((r * 5) % 9 - 4) * amp / 4;
        for (int j0 = 0; j0 < 8; j0 += 2) {                       // 8 points per lane per pass, two in flight
            float4 v[16];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                if (!QUAD) {
                    int r0, c0;
                    point_pos(b, (j0 + jj) * 256 + tid, sh, sv, r0, c0);
                    const int ca = (c0 - 1) & ~3;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        v[jj * 8 + 2 * q] = *reinterpret_cast<const float4*>(row_seg<TILE_W>(frame, TWt, r0 - 1 + q, ca));
                        v[jj * 8 + 2 * q + 1] = *reinterpret_cast<const float4*>(row_seg<TILE_W>(frame, TWt, r0 - 1 + q, ca + 4));
                    }
                } else {
                    const int row = tid & 3;                              // this lane's row of every patch of its quad
#pragma unroll
                    for (int q = 0; q < 4; ++q) {                         // the quad's four points
                        int r0, c0;
                        point_pos(b, (j0 + jj) * 256 + (tid & ~3) + q, sh, sv, r0, c0);
                        const int ca = (c0 - 1) & ~3;
                        v[jj * 8 + 2 * q] = *reinterpret_cast<const float4*>(row_seg<TILE_W>(frame, TWt, r0 - 1 + row, ca));
                        v[jj * 8 + 2 * q + 1] = *reinterpret_cast<const float4*>(row_seg<TILE_W>(frame, TWt, r0 - 1 + row, ca + 4));
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) acc += v[j].x + v[j].y + v[j].z + v[j].w;
        }
    }
    if (acc == 12345.678f) out[b * 256 + tid] = acc;
}

// strips: element offset of the patch row (r, columns ca .. ca + 3)
__device__ __forceinline__ const float* strip_row(const float* frame, size_t copy_stride, int Hp, int r, int ca) {
    const int copy = (ca & 7) > 4;
    const int cc = ca - 4 * copy;
    return frame + copy * copy_stride + ((size_t)(cc >> 3) * Hp + r) * 8 + (cc & 7);
}
struct __attribute__((packed, aligned(4))) f4u { float x, y, z, w; };
template <int QUAD>
__global__ __launch_bounds__(256, 2) void gather_strips(const float* __restrict__ buf, float* out, int passes, size_t frame_stride, int amp) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const int Hp = 488;
    const float* __restrict__ frame = buf + (size_t)b * 2 * frame_stride;
    float acc = 0.f;
    for (int r = 0; r < passes; ++r) {
        const int sh = ((r * 7) % 11 - 5) * amp / 5, sv = ((r * 5) % 9 - 4) * amp / 4;
        for (int j0 = 0; j0 < 8; j0 += 2) {
            f4u v[8];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                if (!QUAD) {
                    int r0, c0;
                    point_pos(b, (j0 + jj) * 256 + tid, sh, sv, r0, c0);
                    const float* p = strip_row(frame, frame_stride, Hp, r0 - 1, c0 - 1);
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[jj * 4 + q] = *reinterpret_cast<const f4u*>(p + 8 * q);
                } else {
                    const int row = tid & 3;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        int r0, c0;
                        point_pos(b, (j0 + jj) * 256 + (tid & ~3) + q, sh, sv, r0, c0);
                        v[jj * 4 + q] = *reinterpret_cast<const f4u*>(strip_row(frame, frame_stride, Hp, r0 - 1 + row, c0 - 1));
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += v[j].x + v[j].y + v[j].z + v[j].w;
        }
    }
    if (acc == 12345.678f) out[b * 256 + tid] = acc;
}

int main(int argc, char** argv) {
    const size_t frame_stride = 162 * 122 * 16;                        // floats per frame (both layouts)
    float* buf; float* out;
    const int Gmax = 1024;
    const int amp = argc > 1 ? atoi(argv[1]) : 5;     // largest shift between passes in pixels (the tracker: a few pixels early, sub-pixel later)
    printf("shift amplitude %d px\n", amp);
    (void)hipMalloc(&buf, 2 * Gmax * frame_stride * 4); (void)hipMemset(buf, 0, 2 * Gmax * frame_stride * 4);   // x 2: the strips' two copies
    (void)hipMalloc(&out, (size_t)Gmax * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int Gs[] = {64, 128, 256, 512, 1024};
    for (int G : Gs) {
        const int passes = 64;
        printf("G = %4d frames in flight (%5.1f per XCD), %d passes x 2 048 patches each\n", G, G / 8.0, passes);
#define RUN(TW, Q, name) do { \
            gather<TW, Q><<<G, 256>>>(buf, out, passes, frame_stride, amp); (void)hipDeviceSynchronize(); \
            (void)hipEventRecord(e0); for (int r = 0; r < 3; ++r) gather<TW, Q><<<G, 256>>>(buf, out, passes, frame_stride, amp); \
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3; \
            const double patches = (double)G * 2048 * passes; \
            printf("  %s: %8.3f ms  %7.2f G patches/s  %6.2f clocks per patch per CU\n", name, ms, patches / ms / 1e6, \
                   2.4e9 * (G < 256 ? G : 256) / (patches / ms * 1e3)); } while (0)
        RUN(4, 0, "V0 lane/point, 64-B tiles ");
        RUN(4, 1, "V1 quad rows,  64-B tiles ");
        RUN(8, 0, "V2 lane/point, 128-B tiles");
        RUN(8, 1, "V3 quad rows,  128-B tiles");
#define RUNS(Q, name) do { \
            gather_strips<Q><<<G, 256>>>(buf, out, passes, frame_stride, amp); (void)hipDeviceSynchronize(); \
            (void)hipEventRecord(e0); for (int r = 0; r < 3; ++r) gather_strips<Q><<<G, 256>>>(buf, out, passes, frame_stride, amp); \
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3; \
            const double patches = (double)G * 2048 * passes; \
            printf("  %s: %8.3f ms  %7.2f G patches/s  %6.2f clocks per patch per CU\n", name, ms, patches / ms / 1e6, \
                   2.4e9 * (G < 256 ? G : 256) / (patches / ms * 1e3)); } while (0)
        RUNS(0, "V4 lane/point, strips x 2 ");
        RUNS(1, "V5 quad rows,  strips x 2 ");
    }
    return 0;
}
