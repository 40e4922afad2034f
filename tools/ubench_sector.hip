// Microbenchmark 7 (round 4): does ANY flavour of vector load make the L2 fill less than a whole 128-byte line from the fabric?
//
// Everything that gathers bicubic patches here is bound by 128-byte line fills: a 4x4 patch is one 64-byte tile row set, but
// the L2 fetches the whole line it lies in.  If a cache-policy bit (sc0 / sc1 / nt) made the fill a 64-byte SECTOR, the 4x4 tiles — the
// single-copy layout a first solve samples — would cost 3.06 x 64 B per patch instead of 2.4 x 128 B, a third less through the fabric.
// Each lane reads 16 bytes at a random 64-byte-aligned address of a buffer far larger than the caches; 8 independent loads per lane and
// iteration; the flavour is an instruction modifier.  Reported: loads/s and, from rocprofv3 (--pmc TCC_EA0_RDREQ_sum, FETCH_SIZE in
// separate runs), fabric requests per load.   hipcc --offload-arch=gfx950 -O3 tools/ubench_sector.hip -o tools/ubench_sector.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define LOADX4(MOD)                                                                                   \
    asm volatile("global_load_dwordx4 %0, %8, off " MOD "\n\t"                                       \
                 "global_load_dwordx4 %1, %9, off " MOD "\n\t"                                       \
                 "global_load_dwordx4 %2, %10, off " MOD "\n\t"                                      \
                 "global_load_dwordx4 %3, %11, off " MOD "\n\t"                                      \
                 "global_load_dwordx4 %4, %12, off " MOD "\n\t"                                      \
                 "global_load_dwordx4 %5, %13, off " MOD "\n\t"                                      \
                 "global_load_dwordx4 %6, %14, off " MOD "\n\t"                                      \
                 "global_load_dwordx4 %7, %15, off " MOD "\n\t"                                      \
                 "s_waitcnt vmcnt(0)"                                                                 \
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])       \
                 : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7])                       \
                 : "memory")

template <int FLAVOUR>
__global__ __launch_bounds__(256) void k_gather(const float* __restrict__ buf, unsigned long long nsect, int iters, float* out) {
    unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        float4 v[8];
        const float* p[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            s = s * 1664525u + 1013904223u;
            unsigned long long sec = ((unsigned long long)s * nsect) >> 32;          // a random 64-byte sector
            p[k] = buf + sec * 16 + 4 * (s & 3);                                      // 16 bytes inside it
        }
        if (FLAVOUR == 0) LOADX4("");
        else if (FLAVOUR == 1) LOADX4("nt");
        else if (FLAVOUR == 2) LOADX4("sc0");
        else if (FLAVOUR == 3) LOADX4("sc1");
        else if (FLAVOUR == 4) LOADX4("sc0 sc1");
        else if (FLAVOUR == 5) LOADX4("sc0 sc1 nt");
        else if (FLAVOUR == 6) LOADX4("sc1 nt");
        else LOADX4("sc0 nt");
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += v[k].x + v[k].w;
    }
    if (acc == 123.456f) out[0] = acc;
}

int main(int argc, char** argv) {
    const size_t bytes = (size_t)(argc > 1 ? atof(argv[1]) : 8.0) * (1ull << 30);      // GiB of table: far beyond the 256 MiB Infinity Cache
    const int iters = argc > 2 ? atoi(argv[2]) : 64;
    float *buf, *out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 0, bytes);
    const unsigned long long nsect = bytes / 64;
    const int grid = 256 * 8;
    const char* names[8] = {"plain", "nt", "sc0", "sc1", "sc0 sc1", "sc0 sc1 nt", "sc1 nt", "sc0 nt"};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
#define RUN(F) do { hipLaunchKernelGGL((k_gather<F>), dim3(grid), dim3(256), 0, 0, buf, nsect, 4, out); hipDeviceSynchronize();            \
        hipEventRecord(e0); hipLaunchKernelGGL((k_gather<F>), dim3(grid), dim3(256), 0, 0, buf, nsect, iters, out); hipEventRecord(e1);     \
        hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);                                                                 \
        const double loads = (double)grid * 256 * iters * 8;                                                                               \
        printf("%-12s %8.3f ms  %7.2f G loads/s  = %6.2f TB/s at 64 B per load, %6.2f TB/s at 128 B per load\n", names[F], ms, loads / ms / 1e6, \
               loads * 64 / ms / 1e9, loads * 128 / ms / 1e9); } while (0)
    for (int rep = 0; rep < 2; ++rep) { RUN(0); RUN(1); RUN(2); RUN(3); RUN(4); RUN(5); RUN(6); RUN(7); }
    return 0;
}
