// Microbenchmark: streaming HBM bandwidth of the MI355X this repository is measured on (report the measured
// peak beside the nominal 8 TB/s).  Read-only sum, copy and triad over 4 GiB buffers with dwordx4 accesses.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_stream.hip -o /tmp/ubench_stream && /tmp/ubench_stream
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_read(const float4* __restrict__ a, float* out, size_t n) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const float4 v = a[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 12345.678f) out[0] = acc;
}
__global__ void k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void k_triad(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ c, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 x = a[i], y = b[i];
        c[i] = make_float4(x.x + 3.f * y.x, x.y + 3.f * y.y, x.z + 3.f * y.z, x.w + 3.f * y.w);
    }
}
int main() {
    const size_t bytes = 4ull << 30, n = bytes / 16;
    float4 *a, *b, *c; float* out;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&c, bytes); hipMalloc(&out, 4);
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes); hipMemset(c, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * 32, block = 256;
#define RUN(name, moved, launch) do { launch; hipDeviceSynchronize(); hipEventRecord(e0); for (int r = 0; r < 5; ++r) { launch; } hipEventRecord(e1); \
        hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5; \
        printf("%-6s %7.3f ms  %6.2f TB/s (%.1f GiB moved per launch)\n", name, ms, (double)(moved) / ms / 1e9, (double)(moved) / (1ull << 30)); } while (0)
    RUN("read", bytes, (k_read<<<grid, block>>>(a, out, n)));
    RUN("copy", 2 * bytes, (k_copy<<<grid, block>>>(a, b, n)));
    RUN("triad", 3 * bytes, (k_triad<<<grid, block>>>(a, b, c, n)));
    return 0;
}
