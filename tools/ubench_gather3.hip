// Microbenchmark 2: is the scattered-gather limit a per-lane request rate or a per-distinct-line rate?
//  mode A: every lane loads 16 B from its OWN random 64-B sector                (64 lines per wave-instruction)
//  mode Q: 4 adjacent lanes load the four 16-B pieces of ONE random 64-B sector (16 lines per wave-instruction)
//  mode O: 8 adjacent lanes cover one random 128-B line                         ( 8 lines per wave-instruction)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int LANES_PER_LOC>
__global__ void gather(const float4* __restrict__ buf, const unsigned* __restrict__ idx, float* out, size_t mask, int reps) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    for (int r = 0; r < reps; ++r) {
        // locality like the tracker: each workgroup gathers inside its own 1.25 MB window (one event frame)
        const size_t wmask = (size_t)(1310720 / (16 * LANES_PER_LOC)) - 1;   // not a power of two minus one -> use modulo
        const size_t win = ((size_t)blockIdx.x * 2654435761u) % (mask / (wmask + 1));
        const size_t loc = win * (wmask + 1) + (((size_t)(idx[(t / LANES_PER_LOC) + (size_t)r * 4096] + r * 977) * 2654435761u) % (wmask + 1));
        const float4 v = buf[loc * LANES_PER_LOC + (t % LANES_PER_LOC)];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) out[t] = acc;
}
int main() {
    const size_t bytes = 4ull << 30;
    float4* buf; hipMalloc(&buf, bytes); hipMemset(buf, 0, bytes);
    const size_t n = 16ull << 20;
    std::vector<unsigned> h(n + 65536); unsigned s = 12345; for (auto& v : h) { s = s * 1664525u + 1013904223u; v = s >> 4; }
    unsigned* idx; hipMalloc(&idx, h.size() * 4); hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    float* out; hipMalloc(&out, n * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 8;
#define RUN(L, name) do { size_t mask = bytes / (16 * L) - 1; \
        gather<L><<<n / 256, 256>>>(buf, idx, out, mask, reps); hipDeviceSynchronize(); \
        hipEventRecord(e0); for (int r = 0; r < 3; ++r) gather<L><<<n / 256, 256>>>(buf, idx, out, mask, reps); hipEventRecord(e1); hipEventSynchronize(e1); \
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3; double lanes = (double)n * reps; \
        printf("%s: %7.3f ms  %7.2f G lane-loads/s  %7.2f G distinct lines/s  %6.2f TB/s loaded\n", name, ms, lanes / ms / 1e6, lanes / L / ms / 1e6, lanes * 16 / ms / 1e9); } while (0)
    RUN(1, "A: 1 lane per 16-B location (64 lines/instr)   ");
    RUN(2, "P: 2 lanes per 32-B location (32 lines/instr)  ");
    RUN(4, "Q: 4 lanes per 64-B sector   (16 lines/instr)  ");
    RUN(8, "O: 8 lanes per 128-B line    ( 8 lines/instr)  ");
    RUN(16, "X: 16 lanes per 256-B block  ( 4 blocks/instr) ");
    return 0;
}
