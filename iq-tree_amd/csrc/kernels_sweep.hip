// kernels_sweep.hip -- SURVEY.md 8(f)-1, hot loop 2 as ONE launch: a whole branch-length sweep
// (PhyloTree::optimizeAllBranches' loop over optimizeOneBranch, phylotree.cpp:2252-2332, 2148-2192) of a 4-state engine.
//
// Why one kernel.  A step of the sweep is small -- the one or two node updates that became pending when the previous
// branch changed (phylokernel.h:183-479), theta = a .* b of the branch (phylokernel.h:535-573) and a handful of derivative
// evaluations (phylokernel.h:583-651) driven by Optimization::minimizeNewton (optimization.cpp:388-465) -- and every
// step depends on the length the step before it accepted.  Launched step by step that is two dependent launches per
// branch: measured on MI355X 31 us per branch at 355 patterns, of which the kernels' own work is a fraction and the
// host (30 us of plan building and launches per step) is as slow as the device.  But nothing in a step needs another
// workgroup's vectors: every pattern is independent through the node updates and theta (lane = pattern, the wave that
// wrote a vector is the one that reads it back), and the accepted length is a pure function of the exchanged derivative
// sums, so every workgroup can compute it for itself.  The only traffic between workgroups is the exchange of
// {sum f*df, sum f*ddf} per evaluation that k_newton already has.  So the grid stays resident for the whole sweep, walks
// the steps in order, keeps the accepted lengths in LDS, and the host reads one result block at the end.
//
// Arithmetic: the node update, theta and the derivative sums are the expressions of k_traverse4 / k_theta4 / k_newton
// term for term (same association, same fixed-order reductions over the same tile -> wave -> workgroup assignment), so
// a sweep accepts the same lengths, to the last bit, as the one-submission-per-branch form; tests/test_sweep_gpu.py
// checks exactly that and re-evaluates the optimised tree with the oracle.
#include "iqhip_internal.h"

namespace iqhip {

#define CONST_AS __attribute__((address_space(4)))
template <typename T>
__device__ __forceinline__ const CONST_AS T *sw_const(const T *p) {
    return (const CONST_AS T *)(p);
}

__device__ __forceinline__ double sw_wsum(double v) { return wave_sum64(v); }

struct SweepArgs {
    const SweepOp *ops;
    const SweepStep *steps;
    int nsteps;
    const double *evec, *inv_evec, *tip;   // class-0 eigen-system, tip_partial_lh[state][4]
    const double *eval;                    // [4]
    const double *evalc;                   // [ncat][4] (the derivative passes' per-category expansion, as k_newton)
    const double *rates, *props, *freq, *invar;
    double *theta;
    double *slab;       // [total ops][nwaves] wave partials of sum_scale
    int nwaves;         // gridDim.x * 4
    int64_t ntiles, nptn;
    int state_unknown;
    double x1, x2, xacc, diverge_x;
    int max_steps;
    double *posts;      // [2 step parities][kNewtonPostEpochs][grid][2], all-ones = not posted yet
    double *out;        // [nsteps][6] = {optx, d2l, evaluations, status, diverged, -}
    unsigned long long *prof;   // IQHIP_DEBUG_SWEEP: [8] ticks of the 100 MHz clock spent in {node updates, theta, evaluations, step tail} + counts
};

// one node update of one 64-pattern tile, lane = pattern: the arithmetic of node_update4 / leaf_cat4 (kernels_valu4.hip)
template <int C>
__device__ __forceinline__ void sweep_node_update(const CONST_AS SweepOp &op, const double *s_ex /* [2][B] */,
                                                  const double *s_tab /* [2][5B] */, const double *s_tip,
                                                  const CONST_AS double *U, const CONST_AS double *uinv, int64_t tile, int lane,
                                                  int64_t nptn, int state_unknown, const double *freq_p, const double *invar_p,
                                                  double &scale_acc) {
    constexpr int B = 4 * C;
    const int64_t ptn = tile * 64 + lane;
    const bool leafL = op.lv == nullptr, leafR = op.rv == nullptr;
    double Lv[B], Rv[B];
    int sc = 0, sL = 0, sR = 0;
    if (leafL) {
        sL = op.ls[ptn];
    } else {
        const double2 *p = reinterpret_cast<const double2 *>(op.lv + tile * (64 * B)) + lane;
#pragma unroll
        for (int j = 0; j < 2 * C; j++) { const double2 t = p[j * 64]; Lv[2 * j] = t.x; Lv[2 * j + 1] = t.y; }
        sc += op.lsc[ptn];
    }
    if (leafR) {
        sR = op.rs[ptn];
    } else {
        const double2 *p = reinterpret_cast<const double2 *>(op.rv + tile * (64 * B)) + lane;
#pragma unroll
        for (int j = 0; j < 2 * C; j++) { const double2 t = p[j * 64]; Rv[2 * j] = t.x; Rv[2 * j + 1] = t.y; }
        sc += op.rsc[ptn];
    }
    const bool slowL = leafL && __any((sL >= 4) && (sL != state_unknown));
    const bool slowR = leafR && __any((sR >= 4) && (sR != state_unknown));
    const int rowL = sL < 4 ? sL : 4, rowR = sR < 4 ? sR : 4;
    auto child = [&](bool leaf, bool slow, int s, int row, const double *ex, const double *tab, const double (&V)[B], int c,
                     double (&a)[4]) {
        if (leaf && !slow) {   // K2 table row (phylokernel.h:187-232): A, C, G, T, unknown = exactly 1.0
            const double2 *t = reinterpret_cast<const double2 *>(tab + row * B + c * 4);
            const double2 v0 = t[0], v1 = t[1];
            a[0] = v0.x; a[1] = v0.y; a[2] = v1.x; a[3] = v1.y;
            return;
        }
        double l[4];
        if (leaf) {            // IUPAC ambiguity code somewhere in the wave: E * tip evaluated on the fly
#pragma unroll
            for (int i = 0; i < 4; i++) l[i] = ex[c * 4 + i] * s_tip[s * 4 + i];
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) l[i] = ex[c * 4 + i] * V[c * 4 + i];
        }
#pragma unroll
        for (int x = 0; x < 4; x++) {
            double v = U[x * 4] * l[0];
            v = fma(U[x * 4 + 1], l[1], v);
            v = fma(U[x * 4 + 2], l[2], v);
            v = fma(U[x * 4 + 3], l[3], v);
            a[x] = (leaf && s == state_unknown) ? 1.0 : v;
        }
    };
    double out[B];
    double lh_max = 0.0;
#pragma unroll
    for (int c = 0; c < C; c++) {
        double a[4], b[4], tmp[4];
        child(leafL, slowL, sL, rowL, s_ex, s_tab, Lv, c, a);
        child(leafR, slowR, sR, rowR, s_ex + B, s_tab + 5 * B, Rv, c, b);
#pragma unroll
        for (int x = 0; x < 4; x++) tmp[x] = a[x] * b[x];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            double o = uinv[i * 4] * tmp[0];
            o = fma(uinv[i * 4 + 1], tmp[1], o);
            o = fma(uinv[i * 4 + 2], tmp[2], o);
            o = fma(uinv[i * 4 + 3], tmp[3], o);
            out[c * 4 + i] = o;
            lh_max = fmax(lh_max, fabs(o));
        }
    }
    // scaling (SIMD rule, phylokernel.h:379-392, 461-474); TIP-TIP never scales
    // (the last update of a multifurcating node: the scalar kernel's rule, lh_max == 0 first, phylotreesse.cpp:774-788)
    const int rule = op.no_scale;
    const bool zero = rule == 2 && !(leafL && leafR) && lh_max == 0.0;   // (TIP-TIP never scales)
    const bool do_scale = zero || (!(leafL && leafR) && (lh_max < kScalingThreshold) && (invar_p[ptn] == 0.0) && rule != 1);
    if (do_scale) {
        if (zero) {
#pragma unroll
            for (int e = 0; e < B; e++) out[e] = s_tip[state_unknown * 4 + (e & 3)];
        } else {
#pragma unroll
            for (int e = 0; e < B; e++) out[e] *= kScalingThresholdInv;
        }
        sc += zero ? 4 : 1;
        if (ptn < nptn) scale_acc += (zero ? 4.0 : 1.0) * (kLogScalingThreshold * freq_p[ptn]);
    }
    double2 *d = reinterpret_cast<double2 *>(op.dst + tile * (64 * B)) + lane;
#pragma unroll
    for (int j = 0; j < 2 * C; j++) d[j * 64] = make_double2(out[2 * j], out[2 * j + 1]);
    op.dst_sc[ptn] = (int16_t)sc;
}

// REG: every wave owns at most one tile, whose theta then stays in registers for all evaluations of the step (and goes to
// memory only in the last step: PhyloTree::theta_all after the sweep is that of the last branch).
// WAVES = 8: an alignment of at most 8 tiles (512 patterns) is one workgroup of 512 threads -- no exchange between
// workgroups at all -- whose wave sums are added in the order the two 4-wave workgroups of k_newton add them (same bits).
template <int C, bool REG, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 2) void k_sweep4(const SweepArgs A) {
    constexpr int B = 4 * C;
    constexpr int NT = 64 * WAVES;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *s_tip = smem;               // [32][4]
    double *s_ex = s_tip + 128;         // [2][B]
    double *s_tab = s_ex + 2 * B;       // [2][5B]
    double *s_red = s_tab + 10 * B;     // [2 parities][2 * WAVES]
    double *s_bcast = s_red + 32;       // [2 parities][2]
    // launch constants, read from memory once (every step of the sweep used to start with a round trip for them)
    double *s_cof = s_bcast + 4;        // [B] evalc[c][i] * rates[c]
    double *s_prop = s_cof + B;         // [B] props[c]
    double *s_eval = s_prop + B;        // [4]
    double *s_rate = s_eval + 4;        // [C]
    double *s_evec = s_rate + C;        // [16]
    double *s_vw = s_evec + 16;         // [WAVES][3 B] exp / derivative factors of an evaluation, one copy per wave
    double *s_len = s_vw + WAVES * 3 * B;   // [nsteps] accepted lengths of the steps so far
    __shared__ int s_fail;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int G = (int)gridDim.x, wg = (int)blockIdx.x;
    const int gw = wg * WAVES + wave;
    const CONST_AS double *U = sw_const(A.evec);
    const CONST_AS double *uinv = sw_const(A.inv_evec);
    const CONST_AS SweepOp *ops = sw_const(A.ops);
    const CONST_AS SweepStep *steps = sw_const(A.steps);
    for (int t = threadIdx.x; t < (A.state_unknown + 1) * 4; t += NT) s_tip[t] = A.tip[t];
    for (int t = threadIdx.x; t < B; t += NT) {
        s_cof[t] = A.evalc[t] * A.rates[t >> 2];
        s_prop[t] = A.props[t >> 2];
        s_evec[t & 15] = A.evec[t & 15];
        if (t < 4) s_eval[t] = A.eval[t];
        if (t < C) s_rate[t] = A.rates[t];
    }
    if (B < 16) for (int t = threadIdx.x; t < 16; t += NT) s_evec[t] = A.evec[t];
    if (threadIdx.x == 0) s_fail = 0;
    // REG: the wave's one tile -- its pattern's frequency and invariant-site term stay in registers
    const int64_t my_ptn = (int64_t)gw * 64 + lane;
    const bool my_in = REG && gw < A.ntiles && my_ptn < A.nptn;
    const double my_freq = my_in ? A.freq[my_ptn] : 0.0, my_invar = my_in ? A.invar[my_ptn] : 0.0;
    unsigned int evals = 0;   // evaluations so far: parity of the reduction buffers
    __syncthreads();
    const size_t per_parity = (size_t)kNewtonPostEpochs * G * 2;

    const bool prof = A.prof != nullptr && wg == 0 && threadIdx.x == 0;
    unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = 0; j < A.nsteps; j++) {
        const CONST_AS SweepStep &st = steps[j];
        unsigned long long tk0 = prof ? wall_clock64() : 0;
        unsigned long long *posts = reinterpret_cast<unsigned long long *>(A.posts) + (size_t)(j & 1) * per_parity;
        unsigned long long *posts_other = reinterpret_cast<unsigned long long *>(A.posts) + (size_t)((j + 1) & 1) * per_parity;
        unsigned int epoch = 0;

        // ---- the node updates that are pending at both ends of the branch
        for (int k = st.op_begin; k < st.op_begin + st.nops; k++) {
            const CONST_AS SweepOp &op = ops[k];
            __syncthreads();   // (s_len of the previous step is written; the previous op's regions are no longer read)
            for (int t = threadIdx.x; t < 2 * B; t += NT) {
                const int childi = t / B, e = t - childi * B;
                const int from = childi ? op.rlen_step : op.llen_step;
                const double len = from >= 0 ? s_len[from] : (childi ? op.rlen : op.llen);
                s_ex[t] = exp(s_eval[e & 3] * (s_rate[e >> 2] * len));
            }
            __syncthreads();
            for (int t = threadIdx.x; t < 2 * 5 * B; t += NT) {   // K2 tables of leaf children, the reference's association
                const int childi = t / (5 * B), q = t - childi * (5 * B);
                if ((childi ? op.rv : op.lv) != nullptr) continue;
                const int row = q / B, e = q - row * B, c = e >> 2, x = e & 3;
                const double *ex = s_ex + childi * B;
                double v = 1.0;
                if (row < 4) {
                    const double e0 = __dmul_rn(s_evec[x * 4 + 0], ex[c * 4 + 0]);
                    const double e1 = __dmul_rn(s_evec[x * 4 + 1], ex[c * 4 + 1]);
                    const double e2 = __dmul_rn(s_evec[x * 4 + 2], ex[c * 4 + 2]);
                    const double e3 = __dmul_rn(s_evec[x * 4 + 3], ex[c * 4 + 3]);
                    const double *tp = s_tip + row * 4;
                    v = __dadd_rn(__dadd_rn(__dmul_rn(e0, tp[0]), __dmul_rn(e1, tp[1])),
                                  __dadd_rn(__dmul_rn(e2, tp[2]), __dmul_rn(e3, tp[3])));
                }
                s_tab[t] = v;
            }
            __syncthreads();
            double scale_acc = 0.0;
            for (int64_t tile = gw; tile < A.ntiles; tile += (int64_t)G * WAVES)
                sweep_node_update<C>(op, s_ex, s_tab, s_tip, U, uinv, tile, lane, A.nptn, A.state_unknown, A.freq, A.invar, scale_acc);
            const double ws = __any(scale_acc != 0.0) ? sw_wsum(scale_acc) : 0.0;
            if (lane == 0) A.slab[(size_t)op.row * A.nwaves + gw] = ws;
        }

        unsigned long long tk1 = prof ? wall_clock64() : 0;
        // ---- theta = a .* b of the branch (phylokernel.h:535-573), kept in registers when the wave has one tile
        double th[REG ? B : 1];
        const bool have_tile = gw < A.ntiles;
        for (int64_t tile = gw; tile < A.ntiles; tile += (int64_t)G * WAVES) {
            const int64_t ptn = tile * 64 + lane;
            const double2 *pb = reinterpret_cast<const double2 *>(st.br.b + tile * (64 * B)) + lane;
            const double2 *pa = st.br.a_kind == CHILD_LEAF ? nullptr : reinterpret_cast<const double2 *>(st.br.a + tile * (64 * B)) + lane;
            const int s = st.br.a_kind == CHILD_LEAF ? (int)st.br.a_states[ptn] : 0;
            double2 *pt = reinterpret_cast<double2 *>(A.theta + tile * (64 * B)) + lane;
#pragma unroll
            for (int jj = 0; jj < 2 * C; jj++) {
                const double2 bv = pb[jj * 64];
                double2 av;
                if (pa) av = pa[jj * 64];
                else av = make_double2(s_tip[s * 4 + ((2 * jj) & 3)], s_tip[s * 4 + ((2 * jj + 1) & 3)]);
                const double2 t = make_double2(av.x * bv.x, av.y * bv.y);
                if (!REG || j == A.nsteps - 1) pt[jj * 64] = t;
                if (REG) { th[REG ? 2 * jj : 0] = t.x; th[REG ? 2 * jj + 1 : 0] = t.y; }
            }
        }

        // sums over all patterns of {f*df_ptn, f*ddf_ptn} (MODE 0) or f*log|lh_ptn| (MODE 1) at branch length x,
        // exchanged between the workgroups in k_newton's posted form and fixed order
        auto eval_at = [&](double x, int mode, double &r0, double &r1) {
            // every wave makes its own copy of the B factors (lanes < B): no workgroup barrier before the sums
            double *s_v0 = s_vw + wave * 3 * B, *s_v1 = s_v0 + B, *s_v2 = s_v1 + B;
            __builtin_amdgcn_wave_barrier();
            if (lane < B) {
                const double cof = s_cof[lane];
                const double v = exp(cof * x) * s_prop[lane];
                s_v0[lane] = v;
                s_v1[lane] = cof * v;
                s_v2[lane] = cof * (cof * v);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            double *red = s_red + (evals & 1u) * 16, *bcast = s_bcast + (evals & 1u) * 2;
            evals++;
            double adf = 0.0, addf = 0.0;
            for (int64_t tile = gw; tile < A.ntiles; tile += (int64_t)G * WAVES) {
                const int64_t ptn = tile * 64 + lane;
                double lh = 0.0, d1 = 0.0, d2 = 0.0;
                const double2 *p = reinterpret_cast<const double2 *>(A.theta + tile * (64 * B)) + lane;
#pragma unroll
                for (int jj = 0; jj < B / 2; jj++) {
                    double2 t;
                    if (REG) t = make_double2(th[REG ? 2 * jj : 0], th[REG ? 2 * jj + 1 : 0]);
                    else t = p[jj * 64];
                    lh = fma(s_v0[2 * jj], t.x, lh); lh = fma(s_v0[2 * jj + 1], t.y, lh);
                    d1 = fma(s_v1[2 * jj], t.x, d1); d1 = fma(s_v1[2 * jj + 1], t.y, d1);
                    d2 = fma(s_v2[2 * jj], t.x, d2); d2 = fma(s_v2[2 * jj + 1], t.y, d2);
                }
                if (ptn < A.nptn) {
                    lh += REG ? my_invar : A.invar[ptn];
                    const double f = REG ? my_freq : A.freq[ptn];
                    if (mode == 1) {
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
            (void)have_tile;
            adf = sw_wsum(adf);
            addf = sw_wsum(addf);
            // (two buffers, by the evaluation's parity: the waves that are still reading this one cannot be overtaken by a
            // write of the evaluation after the next, which lies behind the next evaluation's barrier)
            if (lane == 0) { red[2 * wave] = adf; red[2 * wave + 1] = addf; }
            __syncthreads();
            double p0 = (red[0] + red[2]) + (red[4] + red[6]);
            double p1 = (red[1] + red[3]) + (red[5] + red[7]);
            if (WAVES == 8) {
                p0 += (red[8] + red[10]) + (red[12] + red[14]);
                p1 += (red[9] + red[11]) + (red[13] + red[15]);
            }
            if (G > 1) {
                unsigned long long *slots = posts + (size_t)epoch * G * 2;
                if (threadIdx.x == 0) {
                    unsigned long long ua = __double_as_longlong(p0), ub = __double_as_longlong(p1);
                    if (ua == ~0ull) ua = 0x7ff8000000000000ull;   // (a NaN that happens to be the sentinel: any other NaN)
                    if (ub == ~0ull) ub = 0x7ff8000000000000ull;
                    __hip_atomic_store(&slots[2 * wg], ua, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&slots[2 * wg + 1], ub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (threadIdx.x < 64) {
                    double a = 0.0, b = 0.0;
                    long spins = 0;
                    for (;;) {
                        bool ready = true;
                        a = 0.0; b = 0.0;
                        for (int w = threadIdx.x; w < G; w += 64) {
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
                    a = sw_wsum(a);
                    b = sw_wsum(b);
                    if (threadIdx.x == 0) { bcast[0] = a; bcast[1] = b; }
                }
                __syncthreads();
                p0 = bcast[0];
                p1 = bcast[1];
                if (epoch == 0) {
                    // every workgroup has posted evaluation 0 of this step, so every workgroup is done with the previous
                    // step's slots (the other parity): back to "not posted" for the step after this one, and acknowledged
                    // before this workgroup posts anything else
                    for (int t = threadIdx.x; t < kNewtonPostEpochs * 2; t += NT)
                        __hip_atomic_store(&posts_other[((size_t)(t >> 1) * G + wg) * 2 + (t & 1)], ~0ull, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                epoch++;
            }
            r0 = p0;
            r1 = p1;
        };

        // ---- Optimization::minimizeNewton (optimization.cpp:388-465) as the state machine of iqhip_internal.h
        unsigned long long tk2 = prof ? wall_clock64() : 0;
        NewtonState ns;
        newton_init(ns, st.xguess, A.x1, A.x2, A.xacc, A.max_steps);
        while (!ns.done && !s_fail) {
            double pdf, pddf;
            eval_at(ns.rts, 0, pdf, pddf);
            newton_update(ns, pdf, pddf);
        }
        unsigned long long tk3 = prof ? wall_clock64() : 0;
        double result = ns.result, diverged = 0.0;
        // "newton raphson diverged, reset" (phylotree.cpp:2167-2176)
        if (A.diverge_x > 0.0 && ns.status == 0 && !s_fail && result > A.diverge_x) {
            double opt_lh, orig_lh, unused;
            eval_at(result, 1, opt_lh, unused);
            eval_at(st.xguess, 1, orig_lh, unused);
            diverged = 1.0;
            if (orig_lh > opt_lh) result = st.xguess;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            s_len[j] = result;
            if (wg == 0) {
                double *o = A.out + (size_t)j * 6;
                o[0] = result;
                o[1] = ns.d2l;
                o[2] = (double)ns.neval;
                o[3] = (double)(s_fail ? 4 : ns.status);
                o[4] = diverged;
                o[5] = 0.0;
            }
        }
        if (prof) {
            const unsigned long long tk4 = wall_clock64();
            pt[0] += tk1 - tk0; pt[1] += tk2 - tk1; pt[2] += tk3 - tk2; pt[3] += tk4 - tk3;
            pt[4] += (unsigned long long)st.nops; pt[5] += (unsigned long long)ns.neval; pt[6] += 1;
        }
        if (s_fail) break;   // (the exchange gave up: every later length would be garbage; the host finishes step by step)
    }
    if (prof)
        for (int q = 0; q < 8; q++) A.prof[q] = pt[q];
    if (s_fail && blockIdx.x == 0 && threadIdx.x == 0)
        for (int j = 0; j < A.nsteps; j++)
            if (A.out[(size_t)j * 6 + 3] == 0.0 && A.out[(size_t)j * 6 + 2] == 0.0) A.out[(size_t)j * 6 + 3] = 4.0;
}

// IQHIP_DEBUG_SWEEP: workgroup 0's split of the persistent kernel's time (100 MHz clock); the previous launch's figures are
// printed when the next sweep is launched
static unsigned long long *sweep_prof_buffer() {
    static const bool dbg = getenv("IQHIP_DEBUG_SWEEP") != nullptr;
    static unsigned long long *d_prof = nullptr;
    if (!dbg) return nullptr;
    if (!d_prof) {
        if (hipMalloc((void **)&d_prof, 8 * sizeof(unsigned long long)) != hipSuccess) return nullptr;
        (void)hipMemset(d_prof, 0, 8 * sizeof(unsigned long long));
        return d_prof;
    }
    unsigned long long h[8];
    if (hipDeviceSynchronize() == hipSuccess && hipMemcpy(h, d_prof, sizeof h, hipMemcpyDeviceToHost) == hipSuccess && h[6] != 0)
        fprintf(stderr, "[iqhip] k_sweep4 workgroup 0: %llu steps, %llu node updates %.2f us each, theta %.2f us, %llu evaluations %.2f us each, tail %.2f us per step\n",
                h[6], h[4], h[4] ? h[0] * 0.01 / h[4] : 0.0, h[1] * 0.01 / h[6], h[5], h[5] ? h[2] * 0.01 / h[5] : 0.0, h[3] * 0.01 / h[6]);
    (void)hipMemset(d_prof, 0, 8 * sizeof(unsigned long long));
    return d_prof;
}

template <int C>
static hipError_t launch_sweep_c(iqhip_engine *e, SweepArgs &A, int grid, bool reg, int waves) {
    constexpr int B = 4 * C;
    const size_t lds = (size_t)(128 + 2 * B + 10 * B + 32 + 4 + 2 * B + 4 + C + 16 + (waves == 8 ? 8 : 4) * 3 * B + A.nsteps + 2) * sizeof(double);
    if (waves == 8) hipLaunchKernelGGL((k_sweep4<C, true, 8>), dim3(1), dim3(512), lds, e->stream, A);
    else if (reg) hipLaunchKernelGGL((k_sweep4<C, true, 4>), dim3(grid), dim3(256), lds, e->stream, A);
    else hipLaunchKernelGGL((k_sweep4<C, false, 4>), dim3(grid), dim3(256), lds, e->stream, A);
    return hipGetLastError();
}

// grid: every workgroup must be resident for the exchange -- two per CU at most, as k_newton's posted form (whose tile ->
// wave -> workgroup assignment this kernel shares, so that the derivative sums are the same bits); up to 8 tiles: one
// workgroup of 8 waves
int sweep4_waves(const iqhip_engine *e) { return (e->ntiles > 4 && e->ntiles <= 8) ? 8 : 4; }
int sweep4_grid(const iqhip_engine *e) {
    if (sweep4_waves(e) == 8) return 1;
    const int64_t wgs = (e->ntiles + 3) / 4;
    const int64_t max_grid = 2 * (int64_t)e->num_cus;
    return (int)(wgs < 1 ? 1 : (wgs > max_grid ? max_grid : wgs));
}

hipError_t launch_sweep4(iqhip_engine *e, const SweepOp *d_ops, const SweepStep *d_steps, int nsteps, double x1, double x2,
                         double xacc, int max_steps, double diverge_x, double *posts, double *out) {
    SweepArgs A;
    A.ops = d_ops;
    A.steps = d_steps;
    A.nsteps = nsteps;
    A.evec = e->d_evec;
    A.inv_evec = e->d_inv_evec;
    A.tip = e->d_tip;
    A.eval = e->d_eval;
    A.evalc = e->d_evalc;
    A.rates = e->d_rates;
    A.props = e->d_props;
    A.freq = e->d_freq;
    A.invar = e->d_invar;
    A.theta = e->d_theta;
    A.slab = e->d_slab;
    const int grid = sweep4_grid(e), waves = sweep4_waves(e);
    A.nwaves = grid * waves;
    A.ntiles = e->ntiles;
    A.nptn = e->nptn;
    A.state_unknown = e->state_unknown;
    A.x1 = x1;
    A.x2 = x2;
    A.xacc = xacc;
    A.diverge_x = diverge_x;
    A.max_steps = max_steps;
    A.posts = posts;
    A.out = out;
    A.prof = sweep_prof_buffer();
    const bool reg = e->ntiles <= (int64_t)grid * waves;
    switch (e->ncat) {
        case 1: return launch_sweep_c<1>(e, A, grid, reg, waves);
        case 2: return launch_sweep_c<2>(e, A, grid, reg, waves);
        case 3: return launch_sweep_c<3>(e, A, grid, reg, waves);
        case 4: return launch_sweep_c<4>(e, A, grid, reg, waves);
        case 5: return launch_sweep_c<5>(e, A, grid, reg, waves);
        case 6: return launch_sweep_c<6>(e, A, grid, reg, waves);
        case 7: return launch_sweep_c<7>(e, A, grid, reg, waves);
        case 8: return launch_sweep_c<8>(e, A, grid, reg, waves);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace iqhip
