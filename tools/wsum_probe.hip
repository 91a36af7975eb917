// tools/wsum_probe.hip -- wave_sum64 (permlane swaps + DPP, iqhip_internal.h) against the shuffle butterfly it replaced:
// the same bits in every lane, for random magnitudes, cancellations and sparse inputs.
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iiq-tree_amd/csrc tools/wsum_probe.hip -o /tmp/wsum_probe && /tmp/wsum_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include "iqhip_internal.h"

__global__ void k_probe(const double *in, double *fast, double *ref, int nwaves) {
    const int w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    if (w >= nwaves) return;
    const int i = w * 64 + (threadIdx.x & 63);
    double v = in[i];
    fast[i] = iqhip::wave_sum64(v);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    ref[i] = v;
}

int main() {
    const int nwaves = 1 << 14, n = nwaves * 64;
    std::vector<double> h(n);
    srand(7);
    for (int i = 0; i < n; i++) {
        const int kind = (i / 64) % 4;
        const double u = (double)rand() / RAND_MAX - 0.5;
        if (kind == 0) h[i] = u;
        else if (kind == 1) h[i] = ldexp(u, rand() % 600 - 300);
        else if (kind == 2) h[i] = (rand() % 5 == 0) ? ldexp(u, rand() % 40) : 0.0;
        else h[i] = (i & 1) ? 1e16 * u : -1e16 * u + 1e-3 * u;
    }
    double *d_in, *d_f, *d_r;
    hipMalloc(&d_in, n * 8); hipMalloc(&d_f, n * 8); hipMalloc(&d_r, n * 8);
    hipMemcpy(d_in, h.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_probe, dim3(nwaves / 4), dim3(256), 0, 0, d_in, d_f, d_r, nwaves);
    std::vector<double> f(n), r(n);
    hipMemcpy(f.data(), d_f, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(r.data(), d_r, n * 8, hipMemcpyDeviceToHost);
    long bad = 0;
    for (int i = 0; i < n; i++) bad += memcmp(&f[i], &r[i], 8) != 0;
    printf("wave_sum64 vs shuffle butterfly: %d lanes, %ld differ\n", n, bad);
    return bad != 0;
}
