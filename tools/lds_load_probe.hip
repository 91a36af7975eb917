// tools/lds_load_probe.hip -- where does global_load_lds_ubyte put lane i's byte: at base + i or at base + 4 i (zero-extended)?
// hipcc --offload-arch=gfx950 -O3 tools/lds_load_probe.hip -o /tmp/lds_load_probe && /tmp/lds_load_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__global__ void k(const uint8_t *src, uint8_t *out) {
    __shared__ uint8_t s[1024];
    for (int t = threadIdx.x; t < 1024; t += blockDim.x) s[t] = 0xEE;
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + threadIdx.x),
                                     (__attribute__((address_space(3))) void *)(s + wave * 256), 1, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = threadIdx.x; t < 1024; t += blockDim.x) out[t] = s[t];
}
int main() {
    uint8_t h[256], o[1024], *d, *dout;
    for (int i = 0; i < 256; i++) h[i] = (uint8_t)(i + 1);
    (void)hipMalloc(&d, 256); (void)hipMalloc(&dout, 1024);
    (void)hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(128), 0, 0, d, dout);
    (void)hipMemcpy(o, dout, 1024, hipMemcpyDeviceToHost);
    printf("wave 0, first 16 LDS bytes:"); for (int i = 0; i < 16; i++) printf(" %02x", o[i]); printf("\n");
    printf("wave 1, first 16 LDS bytes:"); for (int i = 0; i < 16; i++) printf(" %02x", o[256 + i]); printf("\n");
    printf("wave 0, bytes 60..67:"); for (int i = 60; i < 68; i++) printf(" %02x", o[i]); printf("\n");
    return 0;
}
