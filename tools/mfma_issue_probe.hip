// mfma_issue_probe.hip -- microbenchmark (not part of the product): true shader-clock cost of the two fp64 matrix
// instructions the 20-state kernel uses, alone, alternating, grouped, with vector instructions in between, for 1 and 2
// waves per SIMD.  Cycles come from s_memtime inside the kernel, so clock throttling does not distort them.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef double v4f64 __attribute__((ext_vector_type(4)));
#define BIG(acc) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0)
#define SML(acc) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc, 0, 0, 0)
#define PIN() __builtin_amdgcn_sched_barrier(0)
template <int MODE>
__global__ __launch_bounds__(512) void k(unsigned long long *out, double *sink, int iters) {
    const int l = threadIdx.x & 63;
    double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
    v4f64 B0 = {0, 0, 0, 0}, B1 = B0, B2 = B0;
    double s0 = 0, s1 = 0, s2 = 0;
    double v0 = a, v1 = b, v2 = a + b, v3 = a - b, v4 = a * b, v5 = 1.0;
    unsigned i0 = l, i1 = l * 3, i2 = l * 5, i3 = l * 7;
    double g0 = 0, g1 = 0, g2 = 0;
    __shared__ double lds[512];
    lds[threadIdx.x & 511] = a;
    const double *gsrc = sink + 300000;
    double *gdst = sink + 400000 + (blockIdx.x * blockDim.x + threadIdx.x - l) * 4;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) { BIG(B0); PIN(); BIG(B1); PIN(); BIG(B2); PIN(); }                                  // 3 big
        if (MODE == 1) { SML(s0); PIN(); SML(s1); PIN(); SML(s2); PIN(); }                                  // 3 small
        if (MODE == 2) { BIG(B0); PIN(); SML(s0); PIN(); BIG(B1); PIN(); SML(s1); PIN(); BIG(B2); PIN(); SML(s2); PIN(); }   // alternating
        if (MODE == 3) { BIG(B0); PIN(); BIG(B1); PIN(); BIG(B2); PIN(); SML(s0); PIN(); SML(s1); PIN(); SML(s2); PIN(); }   // grouped
        if (MODE == 4) {   // alternating + 6 vector instructions per pair
            BIG(B0); PIN(); SML(s0); PIN(); v0 = fma(v0, v1, v2); v3 = fma(v3, v4, v5); PIN();
            BIG(B1); PIN(); SML(s1); PIN(); v1 = fma(v1, v2, v3); v4 = fma(v4, v5, v0); PIN();
            BIG(B2); PIN(); SML(s2); PIN(); v2 = fma(v2, v3, v4); v5 = fma(v5, v0, v1); PIN();
        }
        if (MODE == 5) {   // dependent chain: one big accumulator, one small
            BIG(B0); PIN(); SML(s0); PIN(); BIG(B0); PIN(); SML(s0); PIN(); BIG(B0); PIN(); SML(s0); PIN();
        }
        if (MODE == 6) { BIG(B0); PIN(); BIG(B0); PIN(); BIG(B0); PIN(); }                                  // dependent big only
        if (MODE == 7) {   // 3 pairs + 6 v_cndmask_b32 / v_max_u32 (32-bit vector work)
            BIG(B0); PIN(); SML(s0); PIN(); i0 = max(i0, i1); i2 = max(i2, i3); PIN();
            BIG(B1); PIN(); SML(s1); PIN(); i1 = max(i1, i2); i3 = max(i3, i0); PIN();
            BIG(B2); PIN(); SML(s2); PIN(); i0 = max(i0, i3); i2 = max(i2, i1); PIN();
        }
        if (MODE == 8) {   // 3 pairs + 12 32-bit vector instructions
            BIG(B0); PIN(); SML(s0); PIN(); i0 = max(i0, i1); i2 = max(i2, i3); i1 = max(i1, i2); i3 = max(i3, i0); PIN();
            BIG(B1); PIN(); SML(s1); PIN(); i0 = max(i0, i3); i2 = max(i2, i1); i1 = max(i1, i0); i3 = max(i3, i2); PIN();
            BIG(B2); PIN(); SML(s2); PIN(); i0 = max(i0, i1); i2 = max(i2, i3); i1 = max(i1, i2); i3 = max(i3, i0); PIN();
        }
        if (MODE == 9) {   // 3 pairs + 6 LDS reads
            BIG(B0); PIN(); SML(s0); PIN(); v0 += lds[l]; v1 += lds[l + 64]; PIN();
            BIG(B1); PIN(); SML(s1); PIN(); v2 += lds[l + 128]; v3 += lds[l + 192]; PIN();
            BIG(B2); PIN(); SML(s2); PIN(); v4 += lds[l + 256]; v5 += lds[l + 320]; PIN();
        }
        if (MODE == 10) {  // 3 pairs + 3 global loads + 3 global stores (L2-resident)
            BIG(B0); PIN(); SML(s0); PIN(); g0 = gsrc[(i & 1023) * 64 + l]; gdst[l] = v0; PIN();
            BIG(B1); PIN(); SML(s1); PIN(); g1 = gsrc[(i & 1023) * 64 + l + 4096]; gdst[l + 64] = v1; PIN();
            BIG(B2); PIN(); SML(s2); PIN(); g2 = gsrc[(i & 1023) * 64 + l + 8192]; gdst[l + 128] = v2; PIN();
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (l == 0) out[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = B0[0] + B1[1] + B2[2] + s0 + s1 + s2 + v0 + v1 + v2 + v3 + v4 + v5 + i0 + i1 + i2 + i3 + g0 + g1 + g2;
}
template <int MODE>
void run(const char *name, int nbig, int nsml) {
    unsigned long long *d; double *sink;
    hipMalloc(&d, 4096 * 8); hipMalloc(&sink, 4 * 512 * 512 * 8); hipMemset(sink, 0, 4 * 512 * 512 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int wps = 1; wps <= 2; wps++) {
        const int threads = 256 * wps;   // 4 or 8 waves per workgroup = 1 or 2 per SIMD (one workgroup per CU: 256 blocks)
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, sink, 2000);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, sink, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(256 * threads / 64);
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        double mean = 0; for (auto x : h) mean += (double)x; mean /= h.size();
        const double ns_per_iter_simd = ms * 1e6 / iters / wps;   // wall time the SIMD spends per iteration of ONE wave
        printf("%-36s %d wave/SIMD: %7.1f ns per iteration per SIMD (wall) | %7.1f memtime ticks per iteration per wave | %.2f ticks/ns\n",
               name, wps, ns_per_iter_simd, mean / iters, mean / iters / (ms * 1e6 / iters));
    }
    hipFree(d); hipFree(sink);
}
int main() {
    run<0>("3 big, independent", 3, 0);
    run<1>("3 small, independent", 0, 3);
    run<2>("big/small alternating", 3, 3);
    run<3>("3 big then 3 small", 3, 3);
    run<4>("alternating + 2 v_fma per pair", 3, 3);
    run<5>("dependent big/small chain", 3, 3);
    run<6>("dependent big chain", 3, 0);
    run<7>("alternating + 2 v_max_u32 per pair", 3, 3);
    run<8>("alternating + 4 v_max_u32 per pair", 3, 3);
    run<9>("alternating + 2 ds_read per pair", 3, 3);
    run<10>("alternating + load + store per pair", 3, 3);
    {   // wall-clock rate of independent 16x16x4 at 1, 2, 4 waves per SIMD
        unsigned long long *d; double *sink; hipMalloc(&d, 1 << 20); hipMalloc(&sink, 4 * 512 * 512 * 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int wps = 1; wps <= 2; wps++) {
            const int iters = 200000;
            hipLaunchKernelGGL(k<0>, dim3(256), dim3(256 * wps), 0, 0, d, sink, 1000);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k<0>, dim3(256), dim3(256 * wps), 0, 0, d, sink, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = 2048.0 * 3 * iters * 1024 * wps;
            printf("wall clock, %d wave/SIMD, 3 independent 16x16x4 per iteration: %.2f ms -> %.1f TFLOP/s\n", wps, ms, flops / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
