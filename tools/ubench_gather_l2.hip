// Microbenchmark 4: what does the scattered 64-B-line gather cost when the lines come from the XCD's L2 instead of the fabric?
// Persistent-style: G workgroups of 256 threads (2 per CU); every lane issues 8 independent 16-byte loads per repetition at
// random 64-B lines of its TEAM's window.  A team is k workgroups with equal blockIdx % 8 (one XCD under round-robin
// placement) sharing one window of W bytes, so the bytes live per XCD are (G / 8 / k) * W.
//   k = 1, W = 1.25 MB : the tracker of round 1 (every workgroup its own event frame; 64 frames live per XCD)
//   larger k           : fewer frames live per XCD -> the window becomes L2-resident
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(256, 2) void gather(const float4* __restrict__ buf, float* out, int k, unsigned nloc, int reps) {
    const int b = blockIdx.x;
    const int team = (b & 7) + 8 * ((b >> 3) / k);
    const float4* __restrict__ win = buf + (size_t)team * nloc * 4;        // nloc 64-B lines of 4 float4 each
    unsigned s = (b * 256 + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.f;
    for (int r = 0; r < reps; ++r) {
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            s = s * 1664525u + 1013904223u;
            const unsigned line = (unsigned)(((unsigned long long)(s >> 4) * nloc) >> 28);
            v[j] = win[(size_t)line * 4 + ((s >> 2) & 3)];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[j].x + v[j].y + v[j].z + v[j].w;
    }
    if (acc == 12345.678f) out[b * 256 + threadIdx.x] = acc;
}

// Tracker-shaped variant: a team's window is one 4x4-tiled 640x480 frame (Wp = 648 -> 162 tiles per tile-row); the team's 2 048
// points sit at fixed random pixels and move by a few pixels from pass to pass (like LM candidates); a pass reads each
// point's 4x4 neighbourhood as 8 aligned 16-byte loads (two per row), i.e. ~3.06 distinct 64-B tiles.  k workgroups split the
// points of one frame; the number of passes is scaled by k so every configuration does the same number of loads.
__global__ __launch_bounds__(256, 2) void gather_patch(const float* __restrict__ buf, float* out, int k, int passes) {
    const int b = blockIdx.x;
    const int team = (b & 7) + 8 * ((b >> 3) / k), member = (b >> 3) % k;
    const int TW = 162, THt = 122;
    const float* __restrict__ frame = buf + (size_t)team * TW * THt * 16;
    const int ppl = 8 / k;                                   // points per lane
    float acc = 0.f;
    for (int r = 0; r < passes; ++r) {
        const int sh = (r * 7) % 11 - 5, sv = (r * 5) % 9 - 4;
        for (int j0 = 0; j0 < ppl; j0 += 2) {
            float4 v[16];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int pid = ((member * ppl + j0 + jj) * 256 + threadIdx.x);
                unsigned s = (team * 2048 + pid) * 2654435761u + 777u;
                s = s * 1664525u + 1013904223u;
                const int c0 = 16 + (int)(((unsigned long long)(s >> 4) * 608) >> 28) + sh;
                s = s * 1664525u + 1013904223u;
                const int r0 = 16 + (int)(((unsigned long long)(s >> 4) * 448) >> 28) + sv;
                const int ca = c0 - 1, txa = ca >> 2, txb = (c0 + 2) >> 2;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int rr = r0 - 1 + q;
                    const float* trow = frame + ((size_t)(rr >> 2) * TW) * 16 + ((rr & 3) << 2);
                    v[jj * 8 + 2 * q] = *reinterpret_cast<const float4*>(trow + (size_t)txa * 16);
                    v[jj * 8 + 2 * q + 1] = *reinterpret_cast<const float4*>(trow + (size_t)txb * 16);
                }
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) acc += v[j].x + v[j].y + v[j].z + v[j].w;
        }
    }
    if (acc == 12345.678f) out[b * 256 + threadIdx.x] = acc;
}
int main(int argc, char** argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 512;
    const int reps = argc > 2 ? atoi(argv[2]) : 256;
    const size_t bytes = 2ull << 30;
    float4* buf; hipMalloc(&buf, bytes); hipMemset(buf, 0, bytes);
    float* out; hipMalloc(&out, (size_t)G * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("G = %d workgroups x 256 threads, %d repetitions x 8 loads per lane\n", G, reps);
    const double windows[] = {0.4e6, 1.25e6};
    for (double W : windows) {
        for (int k = 1; k <= 64; k *= 2) {
            const unsigned nloc = (unsigned)(W / 64);
            if ((size_t)(G / k + 8) * nloc * 64 > bytes) continue;
            gather<<<G, 256>>>(buf, out, k, nloc, reps); hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int r = 0; r < 3; ++r) gather<<<G, 256>>>(buf, out, k, nloc, reps);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
            const double loads = (double)G * 256 * reps * 8;
            printf("W = %5.2f MB  k = %2d  (%6.1f MB live per XCD): %8.3f ms  %7.2f G lines/s  (%5.2f clocks per line per CU at 2.4 GHz)\n", W / 1e6, k,
                   (double)G / 8 / k * W / 1e6, ms, loads / ms / 1e6, 2.4e9 * 256 / (loads / ms * 1e3));
        }
    }
    printf("\ntracker-shaped gather: 2 048 points per frame, 4x4 patches from a tiled 640x480 frame, %d passes per frame\n", 64);
    for (int k = 1; k <= 4; k *= 2) {
        const int passes = 64 * k;
        gather_patch<<<G, 256>>>((const float*)buf, out, k, passes); hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 3; ++r) gather_patch<<<G, 256>>>((const float*)buf, out, k, passes);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
        const double patches = (double)G * 256 * 8 * 64;
        printf("k = %d  (%3d frames live per XCD): %8.3f ms  %7.2f G patches/s  (~%6.1f G tile lines/s)\n", k, G / 8 / k, ms, patches / ms / 1e6,
               patches * 3.06 / ms / 1e6);
    }
    return 0;
}
