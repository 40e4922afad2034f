// Microbenchmark: random-gather throughput of MI355X HBM as a function of the contiguous bytes fetched per
// random location (what decides the frame tiling of the tracker).  Each lane picks a random G-byte-aligned
// block in a 4 GiB buffer and loads G bytes with dwordx4 loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
template <int G>   // bytes per random location: 16, 32, 64, 128, 256
__global__ void gather(const float4* __restrict__ buf, const unsigned* __restrict__ idx, float* out, size_t nblocks_mask) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t b = ((size_t)idx[t] * 2654435761u) & nblocks_mask;      // block index
    const float4* p = buf + b * (G / 16);
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < G / 16; ++k) { const float4 v = p[k]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 12345.678f) out[t] = acc;
}
int main() {
    const size_t bytes = 4ull << 30;
    float4* buf; hipMalloc(&buf, bytes); hipMemset(buf, 0, bytes);
    const size_t n = 64ull << 20;   // 64 M random locations
    std::vector<unsigned> h(n); unsigned s = 12345; for (auto& v : h) { s = s * 1664525u + 1013904223u; v = s >> 4; }
    unsigned* idx; hipMalloc(&idx, n * 4); hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice);
    float* out; hipMalloc(&out, n * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
#define RUN(G) do { size_t mask = bytes / G - 1; \
        gather<G><<<n / 256, 256>>>(buf, idx, out, mask); hipDeviceSynchronize(); \
        hipEventRecord(e0); for (int r = 0; r < 3; ++r) gather<G><<<n / 256, 256>>>(buf, idx, out, mask); hipEventRecord(e1); hipEventSynchronize(e1); \
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3; \
        printf("G=%4d B/location: %7.3f ms  %7.2f G locations/s  %7.2f TB/s useful (+%.2f TB/s index reads)\n", G, ms, n / ms / 1e6, (double)n * G / ms / 1e9, (double)n * 4 / ms / 1e9); } while (0)
    RUN(16); RUN(32); RUN(64); RUN(128); RUN(256);
    return 0;
}
