// mfma444_probe.hip -- determines the lane layout of v_mfma_f64_4x4x4_4b_f64 empirically
// (the guides list no f64 4x4x4 map).  Block (la, lb): a = 1 on lane la, b = 1 on lane lb, else 0;
// prints for every (la, lb) the lanes whose D is 1.  Also times it against 16x16x4.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef double v4f64 __attribute__((ext_vector_type(4)));
__global__ void probe(double *out) {
    const int la = blockIdx.x, lb = blockIdx.y, l = threadIdx.x;
    const double a = (l == la) ? 1.0 : 0.0, b = (l == lb) ? 1.0 : 0.0;
    double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    out[((size_t)la * 64 + lb) * 64 + l] = d;
}
template <int WHICH>
__global__ void timing(double *out, int iters) {
    const int l = threadIdx.x & 63;
    double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
    if (WHICH == 0) {
        double d0 = 0, d1 = 0, d2 = 0, d3 = 0;
        for (int i = 0; i < iters; i++) {
            d0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d1, 0, 0, 0);
            d2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d2, 0, 0, 0);
            d3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d3, 0, 0, 0);
        }
        out[blockIdx.x * blockDim.x + threadIdx.x] = d0 + d1 + d2 + d3;
    } else {
        v4f64 d0 = {0, 0, 0, 0}, d1 = d0, d2 = d0, d3 = d0;
        for (int i = 0; i < iters; i++) {
            d0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d1, 0, 0, 0);
            d2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d2, 0, 0, 0);
            d3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d3, 0, 0, 0);
        }
        out[blockIdx.x * blockDim.x + threadIdx.x] = d0[0] + d1[1] + d2[2] + d3[3];
    }
}
int main() {
    double *d; hipMalloc(&d, 64 * 64 * 64 * sizeof(double));
    hipLaunchKernelGGL(probe, dim3(64, 64), dim3(64), 0, 0, d);
    std::vector<double> h(64 * 64 * 64);
    hipMemcpy(h.data(), d, h.size() * sizeof(double), hipMemcpyDeviceToHost);
    // for each (la, lb): which output lanes are nonzero
    printf("pairs (la,lb)->lanes with D=1 (first 40 nonempty, then a summary)\n");
    int shown = 0;
    for (int la = 0; la < 64; la++) for (int lb = 0; lb < 64; lb++) {
        std::vector<int> hit;
        for (int l = 0; l < 64; l++) if (h[((size_t)la * 64 + lb) * 64 + l] != 0.0) hit.push_back(l);
        if (!hit.empty() && shown < 40) { printf("la %2d lb %2d ->", la, lb); for (int x : hit) printf(" %d", x); printf("\n"); shown++; }
    }
    // infer: for output lane l, which la set and lb set contribute
    for (int l = 0; l < 64; l += 1) {
        if (!(l < 20 || l % 16 == 0)) continue;
        printf("D lane %2d <= sum over (la,lb):", l);
        for (int la = 0; la < 64; la++) for (int lb = 0; lb < 64; lb++)
            if (h[((size_t)la * 64 + lb) * 64 + l] != 0.0) printf(" (%d,%d)", la, lb);
        printf("\n");
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int which = 0; which < 2; which++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(timing<0>, dim3(256 * 4), dim3(64), 0, 0, d, iters);
            else hipLaunchKernelGGL(timing<1>, dim3(256 * 4), dim3(64), 0, 0, d, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double macs = (which == 0 ? 256.0 : 1024.0) * 4 * iters * 1024;  // per wave x 1024 waves (1 per SIMD)
        printf("%s: %.3f ms for %d x4 MFMAs per wave, 1 wave/SIMD -> %.1f cycles/MFMA at 2.4 GHz, %.1f TFLOP/s\n",
               which == 0 ? "4x4x4_4b" : "16x16x4", ms, iters, ms * 1e-3 * 2.4e9 / (iters * 4.0), 2 * macs / (ms * 1e-3) / 1e12);
    }
    return 0;
}
