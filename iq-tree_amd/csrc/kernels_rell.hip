// kernels_rell.hip -- consumers of the device-resident per-pattern log-likelihoods:
//   k_pattern_lh_scaled : PhyloTree::computePatternLikelihood (phylotree.cpp:1200-1230) -- the scaling
//                         events of both ends of the evaluated branch put back into _pattern_lh
//   k_rell              : UFBoot's RELL scores (IQTree::saveCurrentTree, iqtree.cpp:2726-2736):
//                         one dot product <pattern_lh, boot_sample[s]> per bootstrap sample
// Both are one pass over HBM-resident data (the sample matrix is the only traffic that matters:
// nsamples * nptn * 4 bytes); nothing but the nsamples scores leaves the device.
// The reference accumulates in float over eight AVX lanes (dotProductSIMD<float,Vec8f,8>); here
// the products are formed from the double pattern lnL and the float weight and accumulated in
// double in a fixed order (deterministic, more accurate; tests bound the difference).
#include <hip/hip_runtime.h>

#include "iqhip_internal.h"

namespace iqhip {

__global__ __launch_bounds__(256) void k_pattern_lh_scaled(const double *__restrict__ pattern_lh,
                                                           const int16_t *__restrict__ sc_a,
                                                           const int16_t *__restrict__ sc_b, int64_t nobs,
                                                           int64_t nptn_pad, double shift,
                                                           double *__restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= nptn_pad) return;
    double v = 0.0;  // unobserved (+ASC) and padding patterns carry no site
    if (p < nobs) {
        int s = 0;
        if (sc_a) s += max((int)sc_a[p], 0);
        if (sc_b) s += max((int)sc_b[p], 0);
        v = (pattern_lh[p] - shift) + (double)s * kLogScalingThreshold;
    }
    out[p] = v;
}

// one workgroup per bootstrap sample; thread t owns patterns 4*(t + 256*j) .. +3
__global__ __launch_bounds__(256) void k_rell(const double *__restrict__ ptn_lh, const float *__restrict__ samples,
                                              int64_t nptn_pad, double *__restrict__ out) {
    __shared__ double red[256];
    const float *w = samples + (size_t)blockIdx.x * nptn_pad;
    double acc = 0.0;
    for (int64_t p = (int64_t)threadIdx.x * 4; p < nptn_pad; p += 1024) {  // nptn_pad is a multiple of 64
        const float4 wv = *reinterpret_cast<const float4 *>(w + p);
        const double2 l0 = *reinterpret_cast<const double2 *>(ptn_lh + p);
        const double2 l1 = *reinterpret_cast<const double2 *>(ptn_lh + p + 2);
        acc += l0.x * (double)wv.x;
        acc += l0.y * (double)wv.y;
        acc += l1.x * (double)wv.z;
        acc += l1.y * (double)wv.w;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}

// _pattern_lh_cat of the scalar kernels (phylotreesse.cpp:1190-1237): per pattern and category
// sum_i exp(eval_i r_c t) prop_c theta[ptn][c][i], from the theta buffer of the current branch, unscaled.
// Generic in the vector layout: 64-pattern tiles (4 states) or 16-pattern tiles (20 / 64 states).
__global__ __launch_bounds__(256) void k_pattern_lh_cat(const double *__restrict__ theta, const double *__restrict__ evalc,
                                                        const double *__restrict__ rates, const double *__restrict__ props,
                                                        double len, int n, int ncat, int tile, int64_t nptn,
                                                        double *__restrict__ out) {
    extern __shared__ double s_val[];  // [ncat][n]
    const int B = n * ncat;
    for (int t = threadIdx.x; t < B; t += 256) {
        const int c = t / n;
        s_val[t] = exp(evalc[t] * rates[c] * len) * props[c];
    }
    __syncthreads();
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= nptn * ncat) return;
    const int64_t p = idx / ncat;
    const int c = (int)(idx - p * ncat);
    const int64_t tl = p / tile;
    const int pl = (int)(p - tl * tile);
    const double *base = theta + (size_t)tl * tile * B;
    double acc = 0.0;
    for (int i = 0; i < n; i++) {
        const int e = c * n + i;
        const double th = (tile == 64) ? base[(size_t)(e >> 1) * 128 + pl * 2 + (e & 1)] : base[(size_t)e * 16 + pl];
        acc += s_val[e] * th;
    }
    out[idx] = acc;
}

hipError_t launch_pattern_lh_cat(iqhip_engine *e, double len, double *out) {
    const int64_t total = e->nptn * e->ncat;
    hipLaunchKernelGGL(k_pattern_lh_cat, dim3((unsigned)((total + 255) / 256)), dim3(256), sizeof(double) * e->block, e->stream,
                       e->d_theta, e->d_evalc, e->d_rates, e->d_props, len, e->n, e->ncat, e->tile, e->nptn, out);
    return hipGetLastError();
}

hipError_t launch_pattern_lh_scaled(iqhip_engine *e, const int16_t *sc_a, const int16_t *sc_b, double *out) {
    const int64_t P = e->nptn_pad;
    hipLaunchKernelGGL(k_pattern_lh_scaled, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, e->stream, e->d_pattern_lh,
                       sc_a, sc_b, e->nptn - e->n_unobs, P, e->asc_active ? e->pattern_lh_shift : 0.0, out);
    return hipGetLastError();
}

hipError_t launch_rell(iqhip_engine *e, double *out) {
    hipLaunchKernelGGL(k_rell, dim3((unsigned)e->nboot), dim3(256), 0, e->stream, e->d_ptn_scaled, e->d_boot,
                       e->nptn_pad, out);
    return hipGetLastError();
}

}  // namespace iqhip
