// kernels_valu4.hip -- gfx950 kernels for the 4-state (DNA) path: HBM-bound, fp64 VALU.
//
// Design (DESIGN.md "DNA kernels"): lane = alignment pattern.  Every pattern is independent
// through the whole pruning recursion (phylokernel.h:254,338,413 are plain loops over ptn),
// so ONE launch executes the whole post-ordered list of node updates for its patterns and,
// optionally, the root-branch log-likelihood:  a child vector written by an earlier op of
// the same launch is re-read by the SAME lane (L2 / Infinity-Cache hit, no inter-workgroup
// hand-off, no barrier), and a child that is the previous op's result never leaves the
// registers.  The wave-uniform per-branch matrices E = U*exp(lambda*r*t) (phylokernel.h:
// 159-181) are produced once per submission by k_echild and reach the FMAs as scalar
// (SGPR) operands through the constant address space.
//
// Reference semantics restated here (file:line are /root/reference paths):
//   node update     phylokernel.h:183-479   (TIP-TIP / TIP-INTERNAL / INTERNAL-INTERNAL)
//   scaling rule    phylokernel.h:379-392,461-474  (SIMD rule: *2^256, scale_num+1)
//   branch lnL      phylokernel.h:779-966
//   theta / derv    phylokernel.h:516-651
//   lnL from theta  phylokernel.h:1040-1099
#include "iqhip_internal.h"

namespace iqhip {

// constant-address-space view: loads through it are invariant, so a wave-uniform address
// becomes an s_load (SGPR operand) instead of 64 identical vector loads.
#define CONST_AS __attribute__((address_space(4)))
template <typename T>
__device__ __forceinline__ const CONST_AS T *as_const(const T *p) {
    return (const CONST_AS T *)(p);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---------------------------------------------------------------------------------------
// K1 (phylokernel.h:159-181): opmat[op][child][c][x][i] = U[x][i] * exp(eval[i]*rate_c*len)
// one block per (op, child)
// ---------------------------------------------------------------------------------------
__global__ void k_echild(const DevOp *__restrict__ ops, int n, int ncat,
                         const double *__restrict__ eval, const double *__restrict__ evec,
                         const double *__restrict__ rates, double *__restrict__ opmat) {
    const int op = blockIdx.x >> 1, child = blockIdx.x & 1;
    const double len = child ? ops[op].right_len : ops[op].left_len;
    const int nn = n * n, total = ncat * nn;
    double *out = opmat + (size_t)blockIdx.x * total;
    for (int t = threadIdx.x; t < total; t += blockDim.x) {
        const int c = t / nn, xi = t - c * nn, i = xi % n;
        out[t] = evec[xi] * exp(eval[i] * (rates[c] * len));
    }
}

hipError_t launch_echild(iqhip_engine *e, int nops) {
    if (nops <= 0) return hipSuccess;
    int threads = e->n * e->n * e->ncat;
    threads = threads < 64 ? 64 : (threads > 256 ? 256 : ((threads + 63) / 64) * 64);
    hipLaunchKernelGGL(k_echild, dim3(nops * 2), dim3(threads), 0, e->stream, e->d_ops, e->n,
                       e->ncat, e->d_eval, e->d_evec, e->d_rates, e->d_opmat);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// child access for the 4-state path
// ---------------------------------------------------------------------------------------
template <int C>
__device__ __forceinline__ void load_vec4(const double *__restrict__ base, int64_t tile, int lane,
                                          double (&v)[4 * C]) {
    const double2 *p = reinterpret_cast<const double2 *>(base + tile * (64 * 4 * C)) + lane;
#pragma unroll
    for (int j = 0; j < 2 * C; j++) {
        double2 t = p[j * 64];
        v[2 * j] = t.x;
        v[2 * j + 1] = t.y;
    }
}
template <int C>
__device__ __forceinline__ void store_vec4(double *__restrict__ base, int64_t tile, int lane,
                                           const double (&v)[4 * C]) {
    double2 *p = reinterpret_cast<double2 *>(base + tile * (64 * 4 * C)) + lane;
#pragma unroll
    for (int j = 0; j < 2 * C; j++) p[j * 64] = make_double2(v[2 * j], v[2 * j + 1]);
}

// ---------------------------------------------------------------------------------------
// the fused traversal kernel
// ---------------------------------------------------------------------------------------
struct Trav4Args {
    const DevOp *ops;
    const double *opmat;
    const double *inv_evec;
    const double *tip;
    const double *freq;
    const double *invar;
    const double *eval;
    const double *rates;
    const double *props;
    double *pattern_lh;
    double *slab;      // [nvals][nwaves] wave partials
    int64_t ntiles;
    int64_t nptn;
    int nops;
    int nwaves;
    int state_unknown;
    int has_root;
    DevBranch root;
};

template <int C>
__global__ __launch_bounds__(256) void k_traverse4(const Trav4Args A) {
    constexpr int B = 4 * C;
    __shared__ double s_tip[32 * 4];  // tip_partial_lh rows, state_unknown <= 31
    __shared__ double s_val[B];       // root-branch val[c][i]

    const int nst = A.state_unknown + 1;
    for (int t = threadIdx.x; t < nst * 4; t += 256) s_tip[t] = A.tip[t];
    if (A.has_root && threadIdx.x < B) {
        const int c = threadIdx.x >> 2, i = threadIdx.x & 3;
        s_val[threadIdx.x] = exp(A.eval[i] * (A.rates[c] * A.root.len)) * A.props[c];
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= A.ntiles) return;  // wave-uniform; no barrier below
    const int gw = (int)tile;      // global wave id == tile id
    const int64_t ptn = tile * 64 + lane;

    const double freq = A.freq[ptn];
    const double invar = A.invar[ptn];

    const CONST_AS double *uinv = as_const(A.inv_evec);
    double prev[B];
#pragma unroll
    for (int e = 0; e < B; e++) prev[e] = 0.0;

    for (int k = 0; k < A.nops; k++) {
        const CONST_AS DevOp *op = as_const(A.ops) + k;
        const int lk = op->left_kind, rk = op->right_kind;
        double L[B], R[B];
        bool unkL = false, unkR = false;
        int sc = 0;
        // ---- left child
        if (lk == CHILD_LEAF) {
            const int s = op->left_states[ptn];
            unkL = (s == A.state_unknown);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const double t = s_tip[s * 4 + i];
#pragma unroll
                for (int c = 0; c < C; c++) L[c * 4 + i] = t;
            }
        } else {
            if (lk == CHILD_PREV) {
#pragma unroll
                for (int e = 0; e < B; e++) L[e] = prev[e];
            } else {
                load_vec4<C>(op->left, tile, lane, L);
            }
            sc += op->left_sc[ptn];
        }
        // ---- right child
        if (rk == CHILD_LEAF) {
            const int s = op->right_states[ptn];
            unkR = (s == A.state_unknown);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const double t = s_tip[s * 4 + i];
#pragma unroll
                for (int c = 0; c < C; c++) R[c * 4 + i] = t;
            }
        } else {
            if (rk == CHILD_PREV) {
#pragma unroll
                for (int e = 0; e < B; e++) R[e] = prev[e];
            } else {
                load_vec4<C>(op->right, tile, lane, R);
            }
            sc += op->right_sc[ptn];
        }
        // ---- out[c] = U^-1 * ((E_L[c]*L[c]) .* (E_R[c]*R[c]))   phylokernel.h:420-459
        const CONST_AS double *EL = as_const(A.opmat) + (size_t)k * (2 * C * 16);
        const CONST_AS double *ER = EL + C * 16;
        double lh_max = 0.0;
#pragma unroll
        for (int c = 0; c < C; c++) {
            double tmp[4];
#pragma unroll
            for (int x = 0; x < 4; x++) {
                const CONST_AS double *el = EL + c * 16 + x * 4;
                const CONST_AS double *er = ER + c * 16 + x * 4;
                double a = el[0] * L[c * 4];
                a = fma(el[1], L[c * 4 + 1], a);
                a = fma(el[2], L[c * 4 + 2], a);
                a = fma(el[3], L[c * 4 + 3], a);
                double b = er[0] * R[c * 4];
                b = fma(er[1], R[c * 4 + 1], b);
                b = fma(er[2], R[c * 4 + 2], b);
                b = fma(er[3], R[c * 4 + 3], b);
                // the reference's lookup row for STATE_UNKNOWN is exactly 1.0 (:228-232)
                a = unkL ? 1.0 : a;
                b = unkR ? 1.0 : b;
                tmp[x] = a * b;
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                double r = uinv[i * 4] * tmp[0];
                r = fma(uinv[i * 4 + 1], tmp[1], r);
                r = fma(uinv[i * 4 + 2], tmp[2], r);
                r = fma(uinv[i * 4 + 3], tmp[3], r);
                prev[c * 4 + i] = r;
                lh_max = fmax(lh_max, fabs(r));
            }
        }
        // ---- scaling (SIMD rule, phylokernel.h:379-392,461-474); TIP-TIP never scales
        const bool both_leaf = (lk == CHILD_LEAF) && (rk == CHILD_LEAF);
        const bool do_scale = !both_leaf && (lh_max < kScalingThreshold) && (invar == 0.0);
        double my_scale = 0.0;
        if (do_scale) {
#pragma unroll
            for (int e = 0; e < B; e++) prev[e] *= kScalingThresholdInv;
            sc += 1;
            my_scale = (ptn < A.nptn) ? kLogScalingThreshold * freq : 0.0;
        }
        store_vec4<C>(op->dst, tile, lane, prev);
        op->dst_sc[ptn] = (int16_t)sc;
        // deterministic reduction: wave partial -> slab[2+k][gw]
        const unsigned long long any = __ballot(do_scale);
        double ws = 0.0;
        if (any) ws = wave_sum(my_scale);
        if (lane == 0) A.slab[(size_t)(2 + k) * A.nwaves + gw] = ws;
    }

    if (A.has_root) {
        // ---- branch lnL, phylokernel.h:806-838 (leaf form) / :930-956 (internal form)
        double Bv[B];
        if (A.root.b_kind == CHILD_PREV) {
#pragma unroll
            for (int e = 0; e < B; e++) Bv[e] = prev[e];
        } else {
            load_vec4<C>(A.root.b, tile, lane, Bv);
        }
        double lh = 0.0;
        if (A.root.a_kind == CHILD_LEAF) {
            const int s = A.root.a_states[ptn];
#pragma unroll
            for (int e = 0; e < B; e++) lh = fma(s_val[e] * s_tip[s * 4 + (e & 3)], Bv[e], lh);
        } else {
            double Av[B];
            if (A.root.a_kind == CHILD_PREV) {
#pragma unroll
                for (int e = 0; e < B; e++) Av[e] = prev[e];
            } else {
                load_vec4<C>(A.root.a, tile, lane, Av);
            }
#pragma unroll
            for (int e = 0; e < B; e++) lh = fma(s_val[e] * Av[e], Bv[e], lh);
        }
        lh += invar;
        const double plh = log(fabs(lh));
        A.pattern_lh[ptn] = plh;
        const double acc = (ptn < A.nptn) ? plh * freq : 0.0;
        const double ws = wave_sum(acc);
        if (lane == 0) {
            A.slab[gw] = ws;
            A.slab[(size_t)A.nwaves + gw] = 0.0;
        }
    }
}

template <int C>
static hipError_t launch_trav_c(iqhip_engine *e, const Trav4Args &A) {
    const int grid = (int)((e->ntiles + 3) / 4);
    hipLaunchKernelGGL(k_traverse4<C>, dim3(grid), dim3(256), 0, e->stream, A);
    return hipGetLastError();
}

hipError_t launch_traverse4(iqhip_engine *e, int nops, const DevBranch *root, int nwaves) {
    Trav4Args A;
    A.ops = e->d_ops;
    A.opmat = e->d_opmat;
    A.inv_evec = e->d_inv_evec;
    A.tip = e->d_tip;
    A.freq = e->d_freq;
    A.invar = e->d_invar;
    A.eval = e->d_eval;
    A.rates = e->d_rates;
    A.props = e->d_props;
    A.pattern_lh = e->d_pattern_lh;
    A.slab = e->d_slab;
    A.ntiles = e->ntiles;
    A.nptn = e->nptn;
    A.nops = nops;
    A.nwaves = nwaves;
    A.state_unknown = e->state_unknown;
    A.has_root = root ? 1 : 0;
    if (root) A.root = *root; else A.root = DevBranch{nullptr, nullptr, nullptr, 0, 0, 0.0};
    switch (e->ncat) {
        case 1: return launch_trav_c<1>(e, A);
        case 2: return launch_trav_c<2>(e, A);
        case 3: return launch_trav_c<3>(e, A);
        case 4: return launch_trav_c<4>(e, A);
        case 5: return launch_trav_c<5>(e, A);
        case 6: return launch_trav_c<6>(e, A);
        case 8: return launch_trav_c<8>(e, A);
        default: return hipErrorInvalidValue;
    }
}

// ---------------------------------------------------------------------------------------
// K7 theta = a .* b  (phylokernel.h:535-573); streaming, lane = pattern
// ---------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void k_theta4(DevBranch br, const double *__restrict__ tip,
                                                int state_unknown, double *__restrict__ theta,
                                                int64_t ntiles) {
    constexpr int B = 4 * C;
    __shared__ double s_tip[32 * 4];
    for (int t = threadIdx.x; t < (state_unknown + 1) * 4; t += 256) s_tip[t] = tip[t];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= ntiles) return;
    const int64_t ptn = tile * 64 + lane;
    double Bv[B], Th[B];
    load_vec4<C>(br.b, tile, lane, Bv);
    if (br.a_kind == CHILD_LEAF) {
        const int s = br.a_states[ptn];
#pragma unroll
        for (int e = 0; e < B; e++) Th[e] = s_tip[s * 4 + (e & 3)] * Bv[e];
    } else {
        double Av[B];
        load_vec4<C>(br.a, tile, lane, Av);
#pragma unroll
        for (int e = 0; e < B; e++) Th[e] = Av[e] * Bv[e];
    }
    store_vec4<C>(theta, tile, lane, Th);
}

hipError_t launch_theta4(iqhip_engine *e, const DevBranch &br) {
    const int grid = (int)((e->ntiles + 3) / 4);
#define IQ_THETA(Cv)                                                                        \
    case Cv:                                                                                \
        hipLaunchKernelGGL(k_theta4<Cv>, dim3(grid), dim3(256), 0, e->stream, br, e->d_tip, \
                           e->state_unknown, e->d_theta, e->ntiles);                       \
        break;
    switch (e->ncat) {
        IQ_THETA(1) IQ_THETA(2) IQ_THETA(3) IQ_THETA(4) IQ_THETA(5) IQ_THETA(6) IQ_THETA(8)
        default: return hipErrorInvalidValue;
    }
#undef IQ_THETA
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// K8 df/ddf from theta (phylokernel.h:516-532,583-651) and K9 lnL from theta (:1040-1099)
// MODE 0: derivatives -> slab[0]=df, slab[1]=ddf ; MODE 1: lnL -> slab[0], writes pattern_lh
// ---------------------------------------------------------------------------------------
template <int C, int MODE>
__global__ __launch_bounds__(256) void k_theta_reduce4(
    const double *__restrict__ theta, const double *__restrict__ eval,
    const double *__restrict__ rates, const double *__restrict__ props, double len,
    const double *__restrict__ freq, const double *__restrict__ invar,
    double *__restrict__ pattern_lh, double *__restrict__ slab, int64_t ntiles, int64_t nptn,
    int nwaves) {
    constexpr int B = 4 * C;
    __shared__ double s_v0[B], s_v1[B], s_v2[B];
    if (threadIdx.x < B) {
        const int c = threadIdx.x >> 2, i = threadIdx.x & 3;
        const double cof = eval[i] * rates[c];
        const double v = exp(cof * len) * props[c];
        s_v0[threadIdx.x] = v;
        s_v1[threadIdx.x] = cof * v;
        s_v2[threadIdx.x] = cof * (cof * v);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= ntiles) return;
    const int gw = (int)tile;
    const int64_t ptn = tile * 64 + lane;
    double Th[B];
    load_vec4<C>(theta, tile, lane, Th);
    const double f = (ptn < nptn) ? freq[ptn] : 0.0;
    double lh = 0.0, d1 = 0.0, d2 = 0.0;
#pragma unroll
    for (int e = 0; e < B; e++) {
        lh = fma(s_v0[e], Th[e], lh);
        if (MODE == 0) {
            d1 = fma(s_v1[e], Th[e], d1);
            d2 = fma(s_v2[e], Th[e], d2);
        }
    }
    lh += invar[ptn];
    if (MODE == 0) {
        const double inv = 1.0 / fabs(lh);
        const double dfp = d1 * inv;
        const double ddfp = fma(-dfp, dfp, d2 * inv);
        const double a = (ptn < nptn) ? dfp * f : 0.0;
        const double b = (ptn < nptn) ? ddfp * f : 0.0;
        const double wa = wave_sum(a), wb = wave_sum(b);
        if (lane == 0) {
            slab[gw] = wa;
            slab[(size_t)nwaves + gw] = wb;
        }
    } else {
        const double plh = log(fabs(lh));
        pattern_lh[ptn] = plh;
        const double a = (ptn < nptn) ? plh * f : 0.0;
        const double wa = wave_sum(a);
        if (lane == 0) slab[gw] = wa;
    }
}

template <int MODE>
static hipError_t launch_theta_reduce(iqhip_engine *e, double len, int nwaves) {
    const int grid = (int)((e->ntiles + 3) / 4);
#define IQ_TR(Cv)                                                                              \
    case Cv:                                                                                   \
        hipLaunchKernelGGL((k_theta_reduce4<Cv, MODE>), dim3(grid), dim3(256), 0, e->stream,   \
                           e->d_theta, e->d_eval, e->d_rates, e->d_props, len, e->d_freq,      \
                           e->d_invar, e->d_pattern_lh, e->d_slab, e->ntiles, e->nptn, nwaves); \
        break;
    switch (e->ncat) {
        IQ_TR(1) IQ_TR(2) IQ_TR(3) IQ_TR(4) IQ_TR(5) IQ_TR(6) IQ_TR(8)
        default: return hipErrorInvalidValue;
    }
#undef IQ_TR
    return hipGetLastError();
}

hipError_t launch_derv4(iqhip_engine *e, double len, int nwaves) {
    return launch_theta_reduce<0>(e, len, nwaves);
}
hipError_t launch_lnl_theta4(iqhip_engine *e, double len, int nwaves) {
    return launch_theta_reduce<1>(e, len, nwaves);
}

// ---------------------------------------------------------------------------------------
// fixed-order reduction of the wave partials: block v sums slab[v][0..nwaves) -> result[v]
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_reduce(const double *__restrict__ slab, int nwaves,
                                                int first_row, double *__restrict__ result) {
    __shared__ double s[256];
    const double *row = slab + (size_t)(first_row + blockIdx.x) * nwaves;
    double acc = 0.0;
    for (int i = threadIdx.x; i < nwaves; i += 256) acc += row[i];
    s[threadIdx.x] = acc;
    __syncthreads();
#pragma unroll
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) result[first_row + blockIdx.x] = s[0];
}

hipError_t launch_reduce(iqhip_engine *e, int first_row, int nrows, int nwaves) {
    if (nrows <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_reduce, dim3(nrows), dim3(256), 0, e->stream, e->d_slab, nwaves,
                       first_row, e->d_result);
    return hipGetLastError();
}

}  // namespace iqhip
