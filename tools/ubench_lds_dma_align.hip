// Where does global_load_lds_dwordx4 land when the LDS base (M0) is NOT a multiple of 1 KB?  One wavefront, lane L loads 16 bytes of
// src[L] (four floats 4L..4L+3) to  lds + OFF floats;  the program prints, for a few OFFs, the LDS float index at which lane 0's and lane 1's
// first words were found.   hipcc --offload-arch=gfx950 -O2 tools/ubench_lds_dma_align.hip -o /tmp/lds_align && /tmp/lds_align
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* glb_ptr;
template <int OFF>
__global__ void k(const float* __restrict__ src, float* __restrict__ out) {
    __shared__ __attribute__((aligned(1024))) float lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = -1.0f;
    __syncthreads();
    __builtin_amdgcn_global_load_lds((glb_ptr)(src + 4 * threadIdx.x), (lds_ptr)(lds + OFF), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) out[i] = lds[i];
}
template <int OFF>
void run(const float* d_src, float* d_out, float* h) {
    hipLaunchKernelGGL(k<OFF>, dim3(1), dim3(64), 0, 0, d_src, d_out);
    hipMemcpy(h, d_out, 4096, hipMemcpyDeviceToHost);
    int at0 = -1, at1 = -1, n = 0;
    for (int i = 0; i < 1024; ++i) { if (h[i] == 0.0f && at0 < 0) at0 = i; if (h[i] == 4.0f && at1 < 0) at1 = i; if (h[i] >= 0.0f) ++n; }
    printf("LDS base + %3d floats (%4d bytes): lane 0's word 0 found at float %d, lane 1's at %d, %d floats written (expected at %d and %d, 256)\n", OFF, OFF * 4, at0, at1, n, OFF, OFF + 4);
}
int main() {
    float *d_src, *d_out, h[1024];
    for (int i = 0; i < 256; ++i) h[i] = (float)i;
    hipMalloc(&d_src, 1024); hipMalloc(&d_out, 4096);
    hipMemcpy(d_src, h, 1024, hipMemcpyHostToDevice);
    run<0>(d_src, d_out, h); run<4>(d_src, d_out, h); run<8>(d_src, d_out, h); run<16>(d_src, d_out, h); run<32>(d_src, d_out, h); run<64>(d_src, d_out, h); run<128>(d_src, d_out, h); run<256>(d_src, d_out, h);
    return 0;
}
