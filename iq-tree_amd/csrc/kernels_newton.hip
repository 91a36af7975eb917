// kernels_newton.hip -- SURVEY.md 8(f)-1: a whole Newton-Raphson branch-length solve in ONE launch.
//
// The reference optimises a branch with Optimization::minimizeNewton (optimization.cpp:388-465)
// calling PhyloTree::computeFuncDerv -> computeLikelihoodDerv (phylotree.cpp:2135-2146,
// phylokernel.h:583-651) once per step: on a GPU that is one launch + one host round trip per
// step (hot loop 2, latency-bound).  Here the loop itself runs on the device: every step evaluates
// df/ddf from the resident theta_all (L2-resident after the first step), reduces them across the
// grid in a fixed order, and every workgroup applies the reference's update rule redundantly.
// A single workgroup needs no grid barrier at all (the common case: real alignments have a few
// thousand patterns); larger grids use a monotonic-counter barrier (agent-scope release/acquire,
// bounded spin) with one workgroup per CU.
#include <string.h>

#include <mutex>

#include "iqhip_internal.h"

namespace iqhip {

// Kernels with a hand-rolled grid barrier (k_newton, k_newton_batch) need all their workgroups resident at once.
// Their grids are sized for that on an otherwise free chip, and ordinary kernels of other engines only delay them
// (those finish without waiting for anybody).  What must never happen is TWO barrier kernels of different engines
// (partition analyses drive one engine per host thread, phylosupertree.cpp:970) each holding part of the CUs while
// waiting for workgroups that cannot be scheduled.  So barrier kernels of one device are chained: each launch waits
// (hipStreamWaitEvent, device side, no host stall) for the previous barrier kernel of ANY engine on that device.
// The barriers use relaxed agent-scope atomics + s_waitcnt vmcnt(0) for the hand-off of the partial sums, which is
// outside the HIP memory model proper; it is the R2 "sc1 loads and stores both sides" form measured for gfx950 in
// MI355X_MICROARCH.md (Workgroup dispatch, valid forms).  A barrier that still times out reports status 4 and the
// engine finishes the solve with the barrier-free chain form.
namespace {
struct BarrierChain {
    std::mutex mu;
    hipEvent_t ev[64] = {};
    bool valid[64] = {};
};
BarrierChain &barrier_chain() {
    static BarrierChain c;
    return c;
}
struct BarrierLaunchGuard {
    int dev;
    hipStream_t stream;
    bool active;
    BarrierLaunchGuard(iqhip_engine *e, bool multi_wg) : dev(e->device), stream(e->stream), active(multi_wg && e->device < 64) {
        if (!active) return;
        BarrierChain &c = barrier_chain();
        c.mu.lock();
        if (c.valid[dev]) (void)hipStreamWaitEvent(stream, c.ev[dev], 0);
    }
    ~BarrierLaunchGuard() {
        if (!active) return;
        BarrierChain &c = barrier_chain();
        if (!c.ev[dev] && hipEventCreateWithFlags(&c.ev[dev], hipEventDisableTiming) != hipSuccess) c.ev[dev] = nullptr;
        if (c.ev[dev]) c.valid[dev] = hipEventRecord(c.ev[dev], stream) == hipSuccess;
        c.mu.unlock();
    }
};
}  // namespace

struct NewtonArgs {
    const double *theta;
    const double *eval;
    const double *rates;
    const double *props;
    const double *freq;
    const double *invar;
    double *partials;        // [2 parities][grid][2]
    double *posts;           // non-NULL: posted exchange, this launch's slots [kNewtonPostEpochs][grid][2] (sentinel = not yet)
    double *posts_other;     // the other launch parity's slots: reset here for the launch after this one
    // non-NULL: the result vector is mapped host memory; workgroup 0 publishes this sequence number when everything the
    // host will read has been written, and the host polls it instead of paying a stream synchronisation (read_result)
    volatile unsigned long long *done;
    unsigned long long seq;
    unsigned int *barrier;   // arrival counter of this launch (zero at its start), counts up over the epochs
    unsigned int *barrier_next;  // the next launch's counter
    double *out;             // {optx, d2l, nsteps, status}
    // fused front end (iqhip_optimize_branch): the first evaluation builds theta = a .* b (phylokernel.h:535-573)
    // while it accumulates, and the sum_scale rows of the preceding node updates are reduced here
    int build;               // 1: theta is written by the first evaluation from `br`
    DevBranch br;
    const double *tipc;      // [state][ncat][n]
    const double *slab;      // wave partials of the node updates
    double *result;          // result vector (rows 2.. receive the sums)
    int nrows, nwaves;
    int64_t ntiles;          // tiles of `tile` patterns
    int64_t nptn;
    int n, ncat, mfma;
    double xguess, x1, x2, xacc;
    int max_steps;
    // branch-length sweeps (iqhip_optimize_sweep): the accepted length also goes to device memory, where the node
    // updates of the later steps of the same submission read it (DevOp::left_len_p / right_len_p)
    double *len_out;
    // +ASC (phylokernel.h:655-725): patterns [nobs, nptn) are the unobserved constant patterns (nobs == nptn: no correction)
    int64_t nobs;
    double asc_nsites;
    // > 0: PhyloTree::optimizeOneBranch's "newton raphson diverged, reset" rule (phylotree.cpp:2167-2176) applied here:
    // a result above this length is kept only if the branch lnL there is not below the lnL at the starting length
    double diverge_x;
};

// sum over the four lane groups of a pattern (lanes p, p+16, p+32, p+48) with gfx950's permlane swaps instead of
// ds_bpermute: swap(x, x) hands every lane its partner's value; (x + x^16) + (that of the lanes ^32), the order of
// the shuffle form (fp addition commutes), so the bits are the same
__device__ __forceinline__ double group_sum(double v) {
    unsigned lo = __double2loint(v), hi = __double2hiint(v);
    auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    v = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
    lo = __double2loint(v);
    hi = __double2hiint(v);
    auto c = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto d = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double(d[0], c[0]) + __hiloint2double(d[1], c[1]);
}

__device__ __forceinline__ double wsum(double v) { return wave_sum64(v); }

// sum over this workgroup's patterns of f*df_ptn and f*ddf_ptn at the val arrays in LDS
// MODE 0: derivative sums (f*df_ptn, f*ddf_ptn); MODE 1: lnL sum (f*log|lh_ptn|) in odf.  The workgroup is
// number `wg` of `nwg` that share the patterns of one branch (the whole grid for k_newton).
// Register-resident theta (ThetaRegs): when every wave of the solve owns at most ONE tile and a lane's share of it is at
// most 20 doubles (4 states: block <= 20; 16-pattern tiles: block <= 80), the values the first evaluation loads (or
// builds) stay in registers and the later evaluations of the solve touch no memory at all: a derivative pass then
// costs its arithmetic plus the exchange between the workgroups instead of an L2 round trip on top (protein 50 x 20k:
// 10 -> 5.5 us per evaluation).  Same values, same order of operations: same bits.
struct ThetaRegs {
    double v[20];
    bool use;    // launch-constant, wave-uniform
    bool have;   // v holds this wave's tile
};

template <bool BUILD, int MODE = 0>
__device__ __forceinline__ void wg_partial(const NewtonArgs &A, const double *theta_c, const DevBranch &br, int wg, int nwg,
                                           const double *s_v0, const double *s_v1,
                                           const double *s_v2, double *s_red, double &odf, double &oddf, ThetaRegs *R = nullptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int B = A.n * A.ncat;
    double adf = 0.0, addf = 0.0;
    for (int64_t tile = (int64_t)wg * 4 + wave; tile < A.ntiles; tile += (int64_t)nwg * 4) {
        double lh = 0.0, d1 = 0.0, d2 = 0.0;
        int64_t ptn;
        bool mine;
        if (R && R->use && !A.mfma) {
            // (at most one tile per wave: this loop body runs once)
            ptn = tile * 64 + lane;
            mine = ptn < A.nobs;
            const double2 *p = reinterpret_cast<const double2 *>(theta_c + tile * (64 * B)) + lane;
            if (!R->have) {
                const double2 *pb = nullptr, *pa = nullptr;
                const double *tp = nullptr;
                if (BUILD) {
                    pb = reinterpret_cast<const double2 *>(br.b + tile * (64 * B)) + lane;
                    if (br.a_kind == CHILD_LEAF) tp = A.tipc + (size_t)br.a_states[ptn] * B;
                    else pa = reinterpret_cast<const double2 *>(br.a + tile * (64 * B)) + lane;
                }
#pragma unroll
                for (int j = 0; j < 10; j++) {
                    double2 t = make_double2(0.0, 0.0);
                    if (2 * j < B) {
                        if (BUILD) {
                            const double2 bv = pb[j * 64];
                            const double2 av = tp ? make_double2(tp[2 * j], tp[2 * j + 1]) : pa[j * 64];
                            t = make_double2(av.x * bv.x, av.y * bv.y);
                            const_cast<double2 *>(p)[j * 64] = t;
                        } else {
                            t = p[j * 64];
                        }
                    }
                    R->v[2 * j] = t.x;
                    R->v[2 * j + 1] = t.y;
                }
                R->have = true;
            }
#pragma unroll
            for (int j = 0; j < 10; j++)
                if (2 * j < B) {
                    lh = fma(s_v0[2 * j], R->v[2 * j], lh); lh = fma(s_v0[2 * j + 1], R->v[2 * j + 1], lh);
                    d1 = fma(s_v1[2 * j], R->v[2 * j], d1); d1 = fma(s_v1[2 * j + 1], R->v[2 * j + 1], d1);
                    d2 = fma(s_v2[2 * j], R->v[2 * j], d2); d2 = fma(s_v2[2 * j + 1], R->v[2 * j + 1], d2);
                }
        } else if (R && R->use) {
            const int p = lane & 15, g = lane >> 4;
            ptn = tile * 16 + p;
            mine = (g == 0) && ptn < A.nobs;
            if (!R->have) {
                const double *th = theta_c + (size_t)tile * 16 * B;
                const double *bv = nullptr, *av = nullptr, *tp = nullptr;
                if (BUILD) {
                    bv = br.b + (size_t)tile * 16 * B;
                    if (br.a_kind == CHILD_LEAF) tp = A.tipc + (size_t)br.a_states[ptn] * B;
                    else av = br.a + (size_t)tile * 16 * B;
                }
#pragma unroll
                for (int q = 0; q < 20; q++) {
                    const int e = g + 4 * q;
                    double t = 0.0;
                    if (e < B) {
                        if (BUILD) {
                            t = (tp ? tp[e] : av[(size_t)e * 16 + p]) * bv[(size_t)e * 16 + p];
                            const_cast<double *>(th)[(size_t)e * 16 + p] = t;
                        } else {
                            t = th[(size_t)e * 16 + p];
                        }
                    }
                    R->v[q] = t;
                }
                R->have = true;
            }
#pragma unroll
            for (int q = 0; q < 20; q++) {
                const int e = g + 4 * q;
                if (e < B) {
                    lh = fma(s_v0[e], R->v[q], lh);
                    d1 = fma(s_v1[e], R->v[q], d1);
                    d2 = fma(s_v2[e], R->v[q], d2);
                }
            }
            lh = group_sum(lh);
            d1 = group_sum(d1);
            d2 = group_sum(d2);
        } else if (!A.mfma) {
            ptn = tile * 64 + lane;
            mine = ptn < A.nobs;   // (the unobserved patterns of +ASC enter through asc_unobserved_sums only)
            const double2 *p = reinterpret_cast<const double2 *>(theta_c + tile * (64 * B)) + lane;
            const double2 *pb = nullptr, *pa = nullptr;
            const double *tp = nullptr;
            if (BUILD) {
                pb = reinterpret_cast<const double2 *>(br.b + tile * (64 * B)) + lane;
                if (br.a_kind == CHILD_LEAF) tp = A.tipc + (size_t)br.a_states[ptn] * B;
                else pa = reinterpret_cast<const double2 *>(br.a + tile * (64 * B)) + lane;
            }
            for (int j = 0; j < B / 2; j++) {
                double2 t;
                if (BUILD) {
                    const double2 bv = pb[j * 64];
                    const double2 av = tp ? make_double2(tp[2 * j], tp[2 * j + 1]) : pa[j * 64];
                    t = make_double2(av.x * bv.x, av.y * bv.y);
                    const_cast<double2 *>(p)[j * 64] = t;
                } else {
                    t = p[j * 64];
                }
                lh = fma(s_v0[2 * j], t.x, lh); lh = fma(s_v0[2 * j + 1], t.y, lh);
                d1 = fma(s_v1[2 * j], t.x, d1); d1 = fma(s_v1[2 * j + 1], t.y, d1);
                d2 = fma(s_v2[2 * j], t.x, d2); d2 = fma(s_v2[2 * j + 1], t.y, d2);
            }
        } else {
            const int p = lane & 15, g = lane >> 4;
            ptn = tile * 16 + p;
            mine = (g == 0) && ptn < A.nobs;
            const double *th = theta_c + (size_t)tile * 16 * B;
            const double *bv = nullptr, *av = nullptr, *tp = nullptr;
            if (BUILD) {
                bv = br.b + (size_t)tile * 16 * B;
                if (br.a_kind == CHILD_LEAF) tp = A.tipc + (size_t)br.a_states[ptn] * B;
                else av = br.a + (size_t)tile * 16 * B;
            }
            // (unrolled: the loads of several rows are in flight together; the accumulation order is unchanged)
            // a pattern's rows 20 at a time: all 20 loads of a lane are in flight together (one L2 round trip per
            // 80 block entries instead of one per unrolled handful); the accumulation order is unchanged
            for (int e0 = g; e0 < B; e0 += 80) {
                double tv[20];
#pragma unroll
                for (int q = 0; q < 20; q++) {
                    const int e = e0 + 4 * q;
                    if (e < B) {
                        if (BUILD) tv[q] = (tp ? tp[e] : av[(size_t)e * 16 + p]) * bv[(size_t)e * 16 + p];
                        else tv[q] = th[(size_t)e * 16 + p];
                    } else {
                        tv[q] = 0.0;
                    }
                }
#pragma unroll
                for (int q = 0; q < 20; q++) {
                    const int e = e0 + 4 * q;
                    if (e < B) {
                        if (BUILD) const_cast<double *>(th)[(size_t)e * 16 + p] = tv[q];
                        lh = fma(s_v0[e], tv[q], lh);
                        d1 = fma(s_v1[e], tv[q], d1);
                        d2 = fma(s_v2[e], tv[q], d2);
                    }
                }
            }
            lh = group_sum(lh);
            d1 = group_sum(d1);
            d2 = group_sum(d2);
        }
        if (mine) {
            lh += A.invar[ptn];
            const double f = A.freq[ptn];
            if (MODE == 1) {
                double l = log(fabs(lh));
                if (isnan(l) || isinf(l)) l = kLogScalingThreshold * 4;  // the reference's repair, phylokernel.h:1100-1122
                adf = fma(l, f, adf);
            } else {
                const double inv = 1.0 / fabs(lh);
                const double dfp = d1 * inv;
                const double ddfp = fma(-dfp, dfp, d2 * inv);
                adf = fma(dfp, f, adf);
                addf = fma(ddfp, f, addf);
            }
        }
    }
    adf = wsum(adf);
    addf = wsum(addf);
    if (lane == 0) { s_red[2 * wave] = adf; s_red[2 * wave + 1] = addf; }
    __syncthreads();
    odf = (s_red[0] + s_red[2]) + (s_red[4] + s_red[6]);
    oddf = (s_red[1] + s_red[3]) + (s_red[5] + s_red[7]);
    __syncthreads();
}

// +ASC, computeLikelihoodDervEigenSIMD's tail (phylokernel.h:655-725): sums over the unobserved constant patterns of
// {lh_ptn + ptn_invar, sum val1*theta, sum val2*theta} at the val arrays in LDS -- no rescaling, no weights.  There are
// nstates such patterns, in the last tile(s); wave 0 of EVERY workgroup computes them for itself (theta of those
// patterns from the resident theta_all, or on the fly from the branch ends while theta is being built by its owner), so
// the exchange between the workgroups stays two doubles and every workgroup applies the same correction.
template <bool BUILD>
__device__ __forceinline__ void asc_unobserved_sums(const NewtonArgs &A, const double *theta_c, const DevBranch &br,
                                                    const double *s_v0, const double *s_v1, const double *s_v2, double *s_asc) {
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const int B = A.n * A.ncat;
        double u0 = 0.0, u1 = 0.0, u2 = 0.0;
        const int T = A.mfma ? 16 : 64;
        for (int64_t tile = A.nobs / T; tile * T < A.nptn; tile++) {
            double lh = 0.0, d1 = 0.0, d2 = 0.0;
            int64_t ptn;
            bool mine;
            if (!A.mfma) {
                ptn = tile * 64 + lane;
                mine = true;
                const double2 *p = reinterpret_cast<const double2 *>(theta_c + tile * (64 * B)) + lane;
                const double2 *pb = reinterpret_cast<const double2 *>(br.b + tile * (64 * B)) + lane;
                const double2 *pa = nullptr;
                const double *tp = nullptr;
                if (BUILD) {
                    if (br.a_kind == CHILD_LEAF) tp = A.tipc + (size_t)br.a_states[ptn] * B;
                    else pa = reinterpret_cast<const double2 *>(br.a + tile * (64 * B)) + lane;
                }
                for (int j = 0; j < B / 2; j++) {
                    double2 t;
                    if (BUILD) {
                        const double2 bv = pb[j * 64];
                        const double2 av = tp ? make_double2(tp[2 * j], tp[2 * j + 1]) : pa[j * 64];
                        t = make_double2(av.x * bv.x, av.y * bv.y);
                    } else {
                        t = p[j * 64];
                    }
                    lh = fma(s_v0[2 * j], t.x, lh); lh = fma(s_v0[2 * j + 1], t.y, lh);
                    d1 = fma(s_v1[2 * j], t.x, d1); d1 = fma(s_v1[2 * j + 1], t.y, d1);
                    d2 = fma(s_v2[2 * j], t.x, d2); d2 = fma(s_v2[2 * j + 1], t.y, d2);
                }
            } else {
                const int p = lane & 15, g = lane >> 4;
                ptn = tile * 16 + p;
                mine = g == 0;
                const double *th = theta_c + (size_t)tile * 16 * B;
                const double *bv = br.b + (size_t)tile * 16 * B, *av = nullptr, *tp = nullptr;
                if (BUILD) {
                    if (br.a_kind == CHILD_LEAF) tp = A.tipc + (size_t)br.a_states[ptn] * B;
                    else av = br.a + (size_t)tile * 16 * B;
                }
                for (int e = g; e < B; e += 4) {
                    const double t = BUILD ? (tp ? tp[e] : av[(size_t)e * 16 + p]) * bv[(size_t)e * 16 + p] : th[(size_t)e * 16 + p];
                    lh = fma(s_v0[e], t, lh);
                    d1 = fma(s_v1[e], t, d1);
                    d2 = fma(s_v2[e], t, d2);
                }
                lh = group_sum(lh);
                d1 = group_sum(d1);
                d2 = group_sum(d2);
            }
            const bool unobs = mine && ptn >= A.nobs && ptn < A.nptn;
            const double w0 = wsum(unobs ? lh + A.invar[ptn] : 0.0), w1 = wsum(unobs ? d1 : 0.0), w2 = wsum(unobs ? d2 : 0.0);
            u0 += w0; u1 += w1; u2 += w2;
        }
        if (lane == 0) { s_asc[0] = u0; s_asc[1] = u1; s_asc[2] = u2; }
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_newton(const NewtonArgs A) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int B = A.n * A.ncat;
    double *s_v0 = smem, *s_v1 = smem + B, *s_v2 = smem + 2 * B, *s_red = smem + 3 * B;  // s_red[8]
    __shared__ double s_bcast[2];
    __shared__ double s_asc[3];
    __shared__ int s_fail;
    if (threadIdx.x == 0) s_fail = 0;
    // the arrival counters of consecutive launches alternate; this launch clears the next one's
    if (blockIdx.x == 0 && threadIdx.x == 0)
        __hip_atomic_store(A.barrier_next, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    // sum_scale rows of the node updates that ran just before (k_reduce's order: 256 threads, LDS tree)
    if (A.nrows > 0) {
        __shared__ double s_rr[256];
        for (int r = blockIdx.x; r < A.nrows; r += gridDim.x) {
            const double *row = A.slab + (size_t)(2 + r) * A.nwaves;
            double acc = 0.0;
            for (int i = threadIdx.x; i < A.nwaves; i += 256) acc += row[i];
            s_rr[threadIdx.x] = acc;
            __syncthreads();
#pragma unroll
            for (int o = 128; o > 0; o >>= 1) {
                if ((int)threadIdx.x < o) s_rr[threadIdx.x] += s_rr[threadIdx.x + o];
                __syncthreads();
            }
            if (threadIdx.x == 0) A.result[2 + r] = s_rr[0];
            __syncthreads();
        }
        // (the rows are host-visible before this workgroup's first post, hence before workgroup 0 publishes `done`)
        if (A.done && threadIdx.x == 0) __threadfence_system();
    }
    // posted exchange: the slots the NEXT launch will use (the other parity) go back to the sentinel; nobody reads
    // them during this launch
    if (A.posts) {
        for (int t = threadIdx.x; t < kNewtonPostEpochs * 2; t += 256) {
            const int ep = t >> 1;
            reinterpret_cast<unsigned long long *>(A.posts_other)[((size_t)ep * gridDim.x + blockIdx.x) * 2 + (t & 1)] = ~0ull;
        }
    }
    unsigned int epoch = 0;
    bool first = true;
    ThetaRegs treg;
    treg.have = false;
    treg.use = A.ntiles <= (int64_t)gridDim.x * 4 && (A.mfma ? B <= 80 : B <= 20);
    // f = -dlnL/dt, df = -d2lnL/dt2 at x (phylotree.cpp:2135-2146)
    auto eval_at = [&](double x, double &f, double &df) {
        for (int t = threadIdx.x; t < B; t += 256) {
            const int c = t / A.n, i = t - c * A.n;
            const double cof = A.eval[t] * A.rates[c];  // eval: per-category expansion [ncat][n]
            const double v = exp(cof * x) * A.props[c];
            s_v0[t] = v;
            s_v1[t] = cof * v;
            s_v2[t] = cof * (cof * v);
        }
        __syncthreads();
        double pdf, pddf;
        const bool asc = A.nobs < A.nptn;
        if (asc) {
            // a launch that builds theta never reads another workgroup's theta tiles (they are plain stores of this very
            // launch: not visible across workgroups without a release / acquire): from the branch ends in every evaluation
            if (A.build) asc_unobserved_sums<true>(A, A.theta, A.br, s_v0, s_v1, s_v2, s_asc);
            else asc_unobserved_sums<false>(A, A.theta, A.br, s_v0, s_v1, s_v2, s_asc);
        }
        if (A.build && first) wg_partial<true>(A, A.theta, A.br, blockIdx.x, gridDim.x, s_v0, s_v1, s_v2, s_red, pdf, pddf, &treg);
        else wg_partial<false>(A, A.theta, A.br, blockIdx.x, gridDim.x, s_v0, s_v1, s_v2, s_red, pdf, pddf, &treg);
        first = false;
        if (gridDim.x > 1 && A.posts) {
            // Posted exchange (round 2): every (evaluation, workgroup) owns a slot {df, ddf} that holds the all-ones
            // pattern until its owner stores into it; each double is an 8-byte agent-scope store, valid on its own, so
            // there is no arrival counter, no wait for a store acknowledgement before it and no ordering between the
            // two stores to rely on.  Wave 0 of every workgroup spins on the slots of the evaluation and sums them in
            // the fixed order of the counter form (same bits).  9 -> 5.5 us per evaluation at 256 workgroups.
            unsigned long long *slots = reinterpret_cast<unsigned long long *>(A.posts) + (size_t)epoch * gridDim.x * 2;
            if (threadIdx.x == 0) {
                unsigned long long ua = __double_as_longlong(pdf), ub = __double_as_longlong(pddf);
                if (ua == ~0ull) ua = 0x7ff8000000000000ull;   // (a NaN that happens to be the sentinel: any other NaN)
                if (ub == ~0ull) ub = 0x7ff8000000000000ull;
                __hip_atomic_store(&slots[2 * blockIdx.x], ua, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&slots[2 * blockIdx.x + 1], ub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (threadIdx.x < 64) {
                double a = 0.0, b = 0.0;
                long spins = 0;
                for (;;) {
                    bool ready = true;
                    a = 0.0; b = 0.0;
                    for (int w = threadIdx.x; w < (int)gridDim.x; w += 64) {
                        const unsigned long long ua = __hip_atomic_load(&slots[2 * w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const unsigned long long ub = __hip_atomic_load(&slots[2 * w + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ready = ready && ua != ~0ull && ub != ~0ull;
                        a += __longlong_as_double(ua);
                        b += __longlong_as_double(ub);
                    }
                    if (__all(ready)) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > 2000000L) { if (threadIdx.x == 0) s_fail = 1; break; }  // never hang the GPU
                }
                a = wsum(a);
                b = wsum(b);
                if (threadIdx.x == 0) { s_bcast[0] = a; s_bcast[1] = b; }
            }
            __syncthreads();
            pdf = s_bcast[0];
            pddf = s_bcast[1];
            __syncthreads();
            epoch++;
        } else if (gridDim.x > 1) {
            double *slot = A.partials + (size_t)(epoch & 1) * gridDim.x * 2;
            if (threadIdx.x == 0) {
                // The partials travel as agent-scope (write-through) stores and are read back with agent-scope
                // loads; no release / acquire fence -- a fence writes back and invalidates the XCD's L2, i.e.
                // throws away the theta slice this workgroup re-reads in every Newton step.
                __hip_atomic_store(&slot[2 * blockIdx.x], pdf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&slot[2 * blockIdx.x + 1], pddf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_fetch_add(A.barrier, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned int target = (epoch + 1) * gridDim.x;
                long spins = 0;
                while (__hip_atomic_load(A.barrier, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > 4000000L) { s_fail = 1; break; }  // never hang the GPU (~0.2 s)
                }
            }
            __syncthreads();
            // fixed-order sum of the workgroup partials by wave 0 (identical on every workgroup)
            if (threadIdx.x < 64) {
                double a = 0.0, b = 0.0;
                for (int w = threadIdx.x; w < (int)gridDim.x; w += 64) {
                    a += __hip_atomic_load(&slot[2 * w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    b += __hip_atomic_load(&slot[2 * w + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                a = wsum(a);
                b = wsum(b);
                if (threadIdx.x == 0) { s_bcast[0] = a; s_bcast[1] = b; }
            }
            __syncthreads();
            pdf = s_bcast[0];
            pddf = s_bcast[1];
            __syncthreads();
            epoch++;
        }
        if (isnan(pdf) || isinf(pdf)) { pdf = 0.0; pddf = 0.0; }  // phylokernel.h:647-651
        if (asc) {   // phylokernel.h:719-724
            const double prob_const = 1.0 - s_asc[0];
            const double df_frac = s_asc[1] / prob_const, ddf_frac = s_asc[2] / prob_const;
            pdf += A.asc_nsites * df_frac;
            pddf += A.asc_nsites * (ddf_frac + df_frac * df_frac);
        }
        f = -pdf;
        df = -pddf;
    };

    // sum over all patterns of f * log|lh_ptn| at branch length x (computeLikelihoodFromBuffer's sum, phylokernel.h:1040-1099),
    // exchanged like the derivative sums; only the diverged-solve rule below needs it
    auto lnl_at = [&](double x) -> double {
        for (int t = threadIdx.x; t < B; t += 256) {
            const int c = t / A.n;
            const double cof = A.eval[t] * A.rates[c];
            const double v = exp(cof * x) * A.props[c];
            s_v0[t] = v;
            s_v1[t] = cof * v;
            s_v2[t] = cof * (cof * v);
        }
        __syncthreads();
        double p0, p1;
        wg_partial<false, 1>(A, A.theta, A.br, blockIdx.x, gridDim.x, s_v0, s_v1, s_v2, s_red, p0, p1, &treg);
        if (gridDim.x > 1 && A.posts) {
            unsigned long long *slots = reinterpret_cast<unsigned long long *>(A.posts) + (size_t)epoch * gridDim.x * 2;
            if (threadIdx.x == 0) {
                unsigned long long ua = __double_as_longlong(p0);
                if (ua == ~0ull) ua = 0x7ff8000000000000ull;
                __hip_atomic_store(&slots[2 * blockIdx.x], ua, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (threadIdx.x < 64) {
                double a = 0.0;
                long spins = 0;
                for (;;) {
                    bool ready = true;
                    a = 0.0;
                    for (int w = threadIdx.x; w < (int)gridDim.x; w += 64) {
                        const unsigned long long ua = __hip_atomic_load(&slots[2 * w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ready = ready && ua != ~0ull;
                        a += __longlong_as_double(ua);
                    }
                    if (__all(ready)) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > 2000000L) { if (threadIdx.x == 0) s_fail = 1; break; }
                }
                a = wsum(a);
                if (threadIdx.x == 0) s_bcast[0] = a;
            }
            __syncthreads();
            p0 = s_bcast[0];
            __syncthreads();
            epoch++;
        } else if (gridDim.x > 1) {
            double *slot = A.partials + (size_t)(epoch & 1) * gridDim.x * 2;
            if (threadIdx.x == 0) {
                __hip_atomic_store(&slot[2 * blockIdx.x], p0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_fetch_add(A.barrier, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned int target = (epoch + 1) * gridDim.x;
                long spins = 0;
                while (__hip_atomic_load(A.barrier, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > 4000000L) { s_fail = 1; break; }
                }
            }
            __syncthreads();
            if (threadIdx.x < 64) {
                double a = 0.0;
                for (int w = threadIdx.x; w < (int)gridDim.x; w += 64)
                    a += __hip_atomic_load(&slot[2 * w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                a = wsum(a);
                if (threadIdx.x == 0) s_bcast[0] = a;
            }
            __syncthreads();
            p0 = s_bcast[0];
            __syncthreads();
            epoch++;
        }
        return p0;
    };

    // ---- Optimization::minimizeNewton (optimization.cpp:388-450), same control flow
    double df, dx, f, temp, xh, xl, rts, rts_old, d2l;
    int nsteps = 1, status = 0;
    rts = A.xguess;
    if (rts < A.x1) rts = A.x1;
    if (rts > A.x2) rts = A.x2;
    eval_at(rts, f, df);
    d2l = df;
    double result = rts;
    bool done = false;
    if (!isfinite(f) || !isfinite(df)) { status = 2; done = true; }
    if (!done && df >= 0.0 && fabs(f) < A.xacc) done = true;
    if (!done) {
        if (f < 0.0) { xl = rts; xh = A.x2; } else { xh = rts; xl = A.x1; }
        dx = fabs(xh - xl);
        int j;
        for (j = 1; j <= A.max_steps; j++) {
            rts_old = rts;
            if ((df <= 0.0) || (((rts - xh) * df - f) * ((rts - xl) * df - f) >= 0.0)) {
                dx = 0.5 * (xh - xl);
                rts = xl + dx;
                d2l = df;
                if (xl == rts) { result = rts; break; }
            } else {
                dx = f / df;
                temp = rts;
                rts -= dx;
                d2l = df;
                if (temp == rts) { result = rts; break; }
            }
            if (fabs(dx) < A.xacc || (j == A.max_steps)) { result = rts_old; break; }
            eval_at(rts, f, df);
            nsteps++;
            if (!isfinite(f) || !isfinite(df)) { status = 2; result = rts_old; break; }
            if (df > 0.0 && fabs(f) < A.xacc) { d2l = df; result = rts; break; }
            if (f < 0.0) xl = rts; else xh = rts;
        }
        if (j > A.max_steps) status = 3;  // "Maximum number of iterations exceeded"
    }
    // "newton raphson diverged, reset" (phylotree.cpp:2167-2176): opt_lh at the result against orig_lh at the starting
    // length, both computeLikelihoodFromBuffer sums of this branch (their lh_scale_factor terms are the same)
    double diverged = 0.0;
    if (A.diverge_x > 0.0 && status == 0 && result > A.diverge_x) {
        const double opt_lh = lnl_at(result);
        const double orig_lh = lnl_at(A.xguess);
        diverged = 1.0;
        if (orig_lh > opt_lh) result = A.xguess;
    }
    __syncthreads();
    if (s_fail) status = 4;  // grid barrier timed out
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        A.out[0] = result;
        A.out[1] = d2l;
        A.out[2] = (double)nsteps;
        A.out[3] = (double)status;
        if (A.diverge_x > 0.0) A.out[4] = diverged;
        if (A.len_out) *A.len_out = result;
        if (A.done) {
            __threadfence_system();
            *A.done = A.seq;
        }
    }
}

// ---------------------------------------------------------------------------------------
// Batched form: M independent branches (the NNI candidates of a tree, IQTree::evaluateNNIs ->
// getBestNNIForBran, phylotree.cpp:2873-3066) in ONE launch.  Task t owns workgroups [t*G, (t+1)*G), its own
// theta buffer, partial-sum slots and arrival counter; inside a task everything is k_newton: theta is built
// by the first evaluation, every workgroup of the task runs minimizeNewton's control flow redundantly, and a
// last pass gives the lnL at the optimum (computeLikelihoodFromBuffer).  All M*G workgroups must be resident
// (the host sizes G for that and splits larger batches).
// ---------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) NewtonTask {
    DevBranch br;
    double xguess, x1, x2, xacc;
    int32_t max_steps;
    int32_t _pad[3];
};

struct NewtonBatchArgs {
    NewtonArgs c;               // the model / alignment fields are used; theta, br, x*, out are per task
    const NewtonTask *tasks;
    double *theta_base;         // [ntasks][theta_stride]
    size_t theta_stride;
    double *partials;           // [ntasks][2 parities][G][2]
    unsigned int *barriers;     // [ntasks] arrival counters of this launch (zero at its start)
    unsigned int *barriers_next;  // the next launch's counters, cleared here
    double *out;                // [ntasks][6] = {optx, d2l, nsteps, status, lnl, 0}
    int G;
    // posted exchange (see k_newton): slots [task][evaluation][workgroup][2] of this launch, sentinel = not yet posted;
    // the first posts_other_used doubles of the other parity's buffer are reset for the launch after this one
    double *posts;
    double *posts_other;
    size_t posts_other_used;
    int post_epochs;
};

__global__ __launch_bounds__(256) void k_newton_batch(const NewtonBatchArgs P) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const NewtonArgs &A = P.c;
    const int B = A.n * A.ncat;
    double *s_v0 = smem, *s_v1 = smem + B, *s_v2 = smem + 2 * B, *s_red = smem + 3 * B;  // s_red[8]
    __shared__ double s_bcast[2];
    __shared__ int s_fail;
    const int task = (int)blockIdx.x / P.G, wg = (int)blockIdx.x - task * P.G, G = P.G;
    // (read through the constant address space: the branch-end pointers inside are then known to be global and the first
    // evaluation's vector reads are global_load, not flat_load with its catch-all waits)
    typedef const __attribute__((address_space(4))) NewtonTask CTask;
    CTask &Tc = ((CTask *)P.tasks)[task];
    NewtonTask T;
    T.br.a = Tc.br.a; T.br.a_states = Tc.br.a_states; T.br.b = Tc.br.b; T.br.a_sc = Tc.br.a_sc; T.br.b_sc = Tc.br.b_sc;
    T.br.a_kind = Tc.br.a_kind; T.br.b_kind = Tc.br.b_kind; T.br.len = Tc.br.len;
    T.xguess = Tc.xguess; T.x1 = Tc.x1; T.x2 = Tc.x2; T.xacc = Tc.xacc; T.max_steps = Tc.max_steps;
    double *theta = P.theta_base + (size_t)task * P.theta_stride;
    double *slots = P.partials + (size_t)task * 4 * G;
    unsigned int *bar = P.barriers + task;
    if (threadIdx.x == 0) {
        s_fail = 0;
        if (wg == 0) __hip_atomic_store(P.barriers_next + task, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (P.posts)
        for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < P.posts_other_used; t += (size_t)gridDim.x * 256)
            reinterpret_cast<unsigned long long *>(P.posts_other)[t] = ~0ull;
    unsigned int epoch = 0;
    bool first = true;
    // sums over the task's patterns of (f*df, f*ddf) -- or of f*log|lh| when lnl_pass -- at branch length x
    auto eval_at = [&](double x, bool lnl_pass, double &r0, double &r1) {
        for (int t = threadIdx.x; t < B; t += 256) {
            const int c = t / A.n;
            const double cof = A.eval[t] * A.rates[c];
            const double v = exp(cof * x) * A.props[c];
            s_v0[t] = v;
            s_v1[t] = cof * v;
            s_v2[t] = cof * (cof * v);
        }
        __syncthreads();
        double p0, p1;
        if (lnl_pass) wg_partial<false, 1>(A, theta, T.br, wg, G, s_v0, s_v1, s_v2, s_red, p0, p1);
        else if (first) wg_partial<true, 0>(A, theta, T.br, wg, G, s_v0, s_v1, s_v2, s_red, p0, p1);
        else wg_partial<false, 0>(A, theta, T.br, wg, G, s_v0, s_v1, s_v2, s_red, p0, p1);
        first = false;
        if (G > 1 && P.posts) {
            unsigned long long *ps = reinterpret_cast<unsigned long long *>(P.posts) +
                                     ((size_t)task * P.post_epochs + epoch) * G * 2;
            if (threadIdx.x == 0) {
                unsigned long long ua = __double_as_longlong(p0), ub = __double_as_longlong(p1);
                if (ua == ~0ull) ua = 0x7ff8000000000000ull;
                if (ub == ~0ull) ub = 0x7ff8000000000000ull;
                __hip_atomic_store(&ps[2 * wg], ua, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&ps[2 * wg + 1], ub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (threadIdx.x < 64) {
                double a = 0.0, b = 0.0;
                long spins = 0;
                for (;;) {
                    bool ready = true;
                    a = 0.0; b = 0.0;
                    for (int w = threadIdx.x; w < G; w += 64) {
                        const unsigned long long ua = __hip_atomic_load(&ps[2 * w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const unsigned long long ub = __hip_atomic_load(&ps[2 * w + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ready = ready && ua != ~0ull && ub != ~0ull;
                        a += __longlong_as_double(ua);
                        b += __longlong_as_double(ub);
                    }
                    if (__all(ready)) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > 2000000L) { if (threadIdx.x == 0) s_fail = 1; break; }  // never hang the GPU
                }
                a = wsum(a);
                b = wsum(b);
                if (threadIdx.x == 0) { s_bcast[0] = a; s_bcast[1] = b; }
            }
            __syncthreads();
            p0 = s_bcast[0];
            p1 = s_bcast[1];
            __syncthreads();
            epoch++;
        } else if (G > 1) {
            double *slot = slots + (size_t)(epoch & 1) * G * 2;
            if (threadIdx.x == 0) {
                __hip_atomic_store(&slot[2 * wg], p0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&slot[2 * wg + 1], p1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned int target = (epoch + 1) * (unsigned int)G;
                long spins = 0;
                while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > 4000000L) { s_fail = 1; break; }  // never hang the GPU (~0.2 s)
                }
            }
            __syncthreads();
            if (threadIdx.x < 64) {  // fixed-order sum of the task's workgroup partials
                double a = 0.0, b = 0.0;
                for (int w = threadIdx.x; w < G; w += 64) {
                    a += __hip_atomic_load(&slot[2 * w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    b += __hip_atomic_load(&slot[2 * w + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                a = wsum(a);
                b = wsum(b);
                if (threadIdx.x == 0) { s_bcast[0] = a; s_bcast[1] = b; }
            }
            __syncthreads();
            p0 = s_bcast[0];
            p1 = s_bcast[1];
            __syncthreads();
            epoch++;
        }
        r0 = p0;
        r1 = p1;
    };
    auto derv_at = [&](double x, double &f, double &df) {
        double pdf, pddf;
        eval_at(x, false, pdf, pddf);
        if (isnan(pdf) || isinf(pdf)) { pdf = 0.0; pddf = 0.0; }  // phylokernel.h:647-651
        f = -pdf;
        df = -pddf;
    };

    // ---- Optimization::minimizeNewton (optimization.cpp:388-450), same control flow as k_newton
    double df, dx, f, temp, xh, xl, rts, rts_old, d2l;
    int nsteps = 1, status = 0;
    rts = T.xguess;
    if (rts < T.x1) rts = T.x1;
    if (rts > T.x2) rts = T.x2;
    derv_at(rts, f, df);
    d2l = df;
    double result = rts;
    bool done = false;
    if (!isfinite(f) || !isfinite(df)) { status = 2; done = true; }
    if (!done && df >= 0.0 && fabs(f) < T.xacc) done = true;
    if (!done) {
        if (f < 0.0) { xl = rts; xh = T.x2; } else { xh = rts; xl = T.x1; }
        dx = fabs(xh - xl);
        int j;
        for (j = 1; j <= T.max_steps; j++) {
            rts_old = rts;
            if ((df <= 0.0) || (((rts - xh) * df - f) * ((rts - xl) * df - f) >= 0.0)) {
                dx = 0.5 * (xh - xl);
                rts = xl + dx;
                d2l = df;
                if (xl == rts) { result = rts; break; }
            } else {
                dx = f / df;
                temp = rts;
                rts -= dx;
                d2l = df;
                if (temp == rts) { result = rts; break; }
            }
            if (fabs(dx) < T.xacc || (j == T.max_steps)) { result = rts_old; break; }
            derv_at(rts, f, df);
            nsteps++;
            if (!isfinite(f) || !isfinite(df)) { status = 2; result = rts_old; break; }
            if (df > 0.0 && fabs(f) < T.xacc) { d2l = df; result = rts; break; }
            if (f < 0.0) xl = rts; else xh = rts;
        }
        if (j > T.max_steps) status = 3;
    }
    // lnL of the branch at the returned length (computeLikelihoodFromBuffer, phylokernel.h:1022-1192)
    double lnl, unused;
    eval_at(result, true, lnl, unused);
    __syncthreads();
    if (s_fail) status = 4;
    if (wg == 0 && threadIdx.x == 0) {
        double *o = P.out + (size_t)task * 6;
        o[0] = result;
        o[1] = d2l;
        o[2] = (double)nsteps;
        o[3] = (double)status;
        o[4] = lnl;
        o[5] = 0.0;
    }
}

size_t newton_task_bytes() { return sizeof(NewtonTask); }
void newton_task_fill(void *dst, const DevBranch &br, double xguess, double x1, double x2, double xacc, int max_steps) {
    NewtonTask t;
    memset(&t, 0, sizeof t);
    t.br = br;
    t.xguess = xguess;
    t.x1 = x1;
    t.x2 = x2;
    t.xacc = xacc;
    t.max_steps = max_steps;
    memcpy(dst, &t, sizeof t);
}

hipError_t launch_newton_batch(iqhip_engine *e, const void *d_tasks, int ntasks, int G, double *theta_base,
                               size_t theta_stride, double *partials, unsigned int *barriers, unsigned int *barriers_next,
                               double *out, double *posts, double *posts_other, size_t posts_other_used, int post_epochs) {
    NewtonBatchArgs P;
    NewtonArgs &A = P.c;
    A.theta = nullptr;
    A.eval = e->d_evalc;
    A.rates = e->d_rates;
    A.props = e->d_props;
    A.freq = e->d_freq;
    A.invar = e->d_invar;
    A.partials = nullptr;
    A.barrier = nullptr;
    A.barrier_next = nullptr;
    A.out = nullptr;
    A.build = 1;
    A.br = DevBranch{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0.0};
    A.tipc = e->d_tipc;
    A.slab = nullptr;
    A.result = nullptr;
    A.nrows = 0;
    A.nwaves = 0;
    A.ntiles = e->ntiles;
    A.nptn = e->nptn;
    A.n = e->n;
    A.ncat = e->ncat;
    A.mfma = e->mfma ? 1 : 0;
    A.xguess = A.x1 = A.x2 = A.xacc = 0.0;
    A.max_steps = 0;
    A.len_out = nullptr;
    A.diverge_x = 0.0;
    A.nobs = e->nptn;   // (+ASC is refused for batches)
    A.asc_nsites = 0.0;
    P.tasks = static_cast<const NewtonTask *>(d_tasks);
    P.theta_base = theta_base;
    P.theta_stride = theta_stride;
    P.partials = partials;
    P.barriers = barriers;
    P.barriers_next = barriers_next;
    P.out = out;
    P.G = G;
    P.posts = posts;
    P.posts_other = posts_other;
    P.posts_other_used = posts_other_used;
    P.post_epochs = post_epochs;
    const size_t lds = (size_t)(3 * e->block + 8) * sizeof(double);
    BarrierLaunchGuard guard(e, G > 1);
    hipLaunchKernelGGL(k_newton_batch, dim3((unsigned)(ntasks * G)), dim3(256), lds, e->stream, P);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// Newton as a chain of enqueued steps (sharded engines: every step needs an all-reduce of {df, ddf} across the
// ranks, so the loop cannot live inside one kernel).  Per step the engine enqueues
//     derivative kernel at state->rts -> k_reduce -> ncclAllReduce(result[0..1]) -> k_newton_state_update
// and reads the state back once per chunk of steps; steps enqueued after `done` are no-ops.
// ---------------------------------------------------------------------------------------
__global__ void k_newton_state_init(NewtonState *st, double xguess, double x1, double x2, double xacc, int max_steps) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        NewtonState s;
        newton_init(s, xguess, x1, x2, xacc, max_steps);
        *st = s;
    }
}

// asc_nsites > 0: +ASC -- result[2..4] are prob_const, df_const, ddf_const summed over all shards (phylokernel.h:719-724)
__global__ void k_newton_state_update(NewtonState *st, const double *result, double asc_nsites) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        NewtonState s = *st;
        if (!s.done) {
            double df = result[0], ddf = result[1];
            if (asc_nsites > 0.0) {
                if (isnan(df) || isinf(df)) { df = 0.0; ddf = 0.0; }   // (phylokernel.h:647-651 comes first)
                const double prob_const = 1.0 - result[2];
                const double df_frac = result[3] / prob_const, ddf_frac = result[4] / prob_const;
                df += asc_nsites * df_frac;
                ddf += asc_nsites * (ddf_frac + df_frac * df_frac);
            }
            newton_update(s, df, ddf);
            *st = s;
        }
    }
}

// batched chain: thread t advances task t's state machine from result[2t], result[2t + 1]
__global__ void k_newton_state_update_batch(NewtonState *st, const double *result, int ntasks) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntasks) return;
    NewtonState s = st[t];
    if (s.done) return;
    newton_update(s, result[2 * t], result[2 * t + 1]);
    st[t] = s;
}

hipError_t launch_newton_state_update_batch(iqhip_engine *e, NewtonState *states, int ntasks) {
    hipLaunchKernelGGL(k_newton_state_update_batch, dim3((unsigned)((ntasks + 63) / 64)), dim3(64), 0, e->stream, states,
                       e->d_result, ntasks);
    return hipGetLastError();
}

hipError_t launch_newton_state_init(iqhip_engine *e, double xguess, double x1, double x2, double xacc, int max_steps) {
    hipLaunchKernelGGL(k_newton_state_init, dim3(1), dim3(64), 0, e->stream, e->d_nstate, xguess, x1, x2, xacc, max_steps);
    return hipGetLastError();
}

hipError_t launch_derv_at_state(iqhip_engine *e, int nwaves) {
    if (e->mfma) return launch_stream_mfma(e, 2, nullptr, 0.0, nwaves, e->d_nstate);
    return launch_derv4(e, 0.0, nwaves, e->d_nstate);
}

hipError_t launch_newton_state_update(iqhip_engine *e) {
    hipLaunchKernelGGL(k_newton_state_update, dim3(1), dim3(64), 0, e->stream, e->d_nstate, e->d_result, e->asc_active ? e->asc_nsites : 0.0);
    return hipGetLastError();
}

hipError_t launch_newton(iqhip_engine *e, double xguess, double x1, double x2, double xacc, int max_steps,
                         double *out, const DevBranch *build_from, int reduce_rows, int reduce_nwaves,
                         const NewtonSweepStep *sweep) {
    NewtonArgs A;
    A.len_out = sweep ? sweep->len_out : nullptr;
    A.nobs = e->nptn - e->n_unobs;
    A.asc_nsites = e->asc_nsites;
    A.diverge_x = sweep ? sweep->diverge_x : 0.0;
    A.build = build_from ? 1 : 0;
    A.br = build_from ? *build_from : DevBranch{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0.0};
    A.tipc = e->d_tipc;
    A.slab = e->d_slab;
    A.result = sweep ? sweep->rows_base - 2 : e->d_result;   // (row r of the node updates goes to result[2 + r])
    A.nrows = reduce_rows;
    A.nwaves = reduce_nwaves;
    A.theta = e->d_theta;
    A.eval = e->d_evalc;
    A.rates = e->d_rates;
    A.props = e->d_props;
    A.freq = e->d_freq;
    A.invar = e->d_invar;
    A.partials = e->d_newton_partials;
    A.out = out;
    A.ntiles = e->ntiles;
    A.nptn = e->nptn;
    A.n = e->n;
    A.ncat = e->ncat;
    A.mfma = e->mfma ? 1 : 0;
    A.xguess = xguess;
    A.x1 = x1;
    A.x2 = x2;
    A.xacc = xacc;
    A.max_steps = max_steps;
    // every workgroup must be resident for the exchange: one per CU with the arrival counter (its slots are sized for that),
    // two per CU with the posted exchange (125 registers, little LDS: four would fit), so that a 100 k-pattern alignment
    // has one tile per wave and a derivative pass is one round trip
    const bool posts = e->newton_posts && max_steps + 5 <= kNewtonPostEpochs;   // (+2: the lnL passes of the diverged-solve rule)
    const int64_t wgs = (e->ntiles + 3) / 4;
    const int64_t max_grid = posts ? 2 * (int64_t)e->num_cus : e->num_cus;
    int grid = (int)(wgs < 1 ? 1 : (wgs > max_grid ? max_grid : wgs));
    A.barrier = e->d_newton_barrier + (e->newton_launches & 1);
    A.barrier_next = e->d_newton_barrier + ((e->newton_launches + 1) & 1);
    e->newton_launches++;
    A.done = nullptr;
    A.seq = 0;
    // (a single workgroup reduces all rows itself; several need the posted exchange in between -- the counter form
    // gives the same ordering, the rows are written before the workgroup's first arrival)
    if ((!sweep || sweep->publish) && e->poll_result && e->d_result == e->d_result_own && out >= e->d_result &&
        out < e->d_result + e->result_cap) {
        A.seq = ++e->result_seq;
        A.done = e->d_done;
        e->poll_pending = true;
    }
    A.posts = A.posts_other = nullptr;
    if (grid > 1 && posts) {
        const size_t per = (size_t)kNewtonPostEpochs * (2 * e->num_cus) * 2;   // (slots are indexed with the launch's grid <= 2 num_cus)
        A.posts = e->d_newton_posts + (e->newton_post_launches & 1) * per;
        A.posts_other = e->d_newton_posts + ((e->newton_post_launches + 1) & 1) * per;
        e->newton_post_launches++;
    }
    const size_t lds = (size_t)(3 * e->block + 8) * sizeof(double);
    BarrierLaunchGuard guard(e, grid > 1);
    hipLaunchKernelGGL(k_newton, dim3(grid), dim3(256), lds, e->stream, A);
    return hipGetLastError();
}

}  // namespace iqhip
