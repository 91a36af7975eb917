/*
 * lh_oracle.c -- CPU restatement of IQ-TREE 1.4.3's eigen-space Felsenstein-pruning
 * likelihood kernels (the AVX <Vec4d,4,nstates> behaviour).
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the product path (libiqhip.so) never does.
 *
 * PARITY STATUS: "parity unpinned" by execution of the reference's likelihood functions.
 * The reference kernels are PhyloTree members whose translation units need the
 * cmake-generated iqtree_config.h (tools.h:24), so they cannot be built in this image
 * without the reference's own build system.  What IS pinned:
 *   - the elementary arithmetic the kernels are made of (Vec4d mul_add / horizontal_add /
 *     exp / log) against the reference's own vectorclass headers, compiled in place by
 *     oracle/Makefile into oracle/_ref/vcl_probe (fixtures: tests/golden/vcl_probe.json);
 *   - the mathematics against an independent probability-space pruning implementation
 *     (tests/textbook.py: expm(Qt) + classical Felsenstein recursion).
 * Every function below cites the reference lines it restates (paths relative to the
 * reference root).
 *
 * Data layout = the reference's host layout (phylonode.h:102-127):
 *   partial_lh[ptn*block + c*n + i]   (i = eigen index), block = n*ncat
 *   scale_num[ptn]                    (short, "UBYTE" phylonode.h:17)
 *   evec[x*n+i] = U[x][i],  inv_evec[i*n+x] = U^-1[i][x]
 *
 * Summation order follows the AVX kernels (no -mfma => mul_add is an unfused a*b+c,
 * vectorclass/vectorf256.h:1870-1877): an n-term dot product is 4 lane accumulators
 * (lane k takes terms k, k+4, ...) combined as (l0+l1)+(l2+l3) (phylokernel.h:31-43).
 * Build with -ffp-contract=off so gcc does not fuse.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define SCALING_THRESHOLD_INVER 0x1p256                  /* 2^256   phylotree.h:51 */
#define SCALING_THRESHOLD 0x1p-256                       /* 2^-256  phylotree.h:52 */
#define LOG_SCALING_THRESHOLD log(SCALING_THRESHOLD)     /* phylotree.h:53 */

enum { ORACLE_SEQ_DNA = 0, ORACLE_SEQ_PROTEIN = 1, ORACLE_SEQ_CODON = 2, ORACLE_SEQ_OTHER = 3 };

/* (l0+l1)+(l2+l3) with lane k = sum_i a[4i+k]*b[4i+k], unfused; phylokernel.h:427-441 */
static inline double dot4(const double *a, const double *b, int n) {
    if (n == 2) return a[0] * b[0] + a[1] * b[1]; /* binary data: <Vec2d, 2, 2> (phylotreesse.cpp:270-274), l0 + l1 */
    double l0 = a[0] * b[0], l1 = a[1] * b[1], l2 = a[2] * b[2], l3 = a[3] * b[3];
    for (int i = 4; i < n; i += 4) {
        l0 = a[i] * b[i] + l0;
        l1 = a[i + 1] * b[i + 1] + l1;
        l2 = a[i + 2] * b[i + 2] + l2;
        l3 = a[i + 3] * b[i + 3] + l3;
    }
    return (l0 + l1) + (l2 + l3);
}

/* exported for tests/test_vcl_probe.py */
double oracle_dot4(const double *a, const double *b, int n) { return dot4(a, b, n); }
double oracle_exp(double x) { return exp(x); }
double oracle_log(double x) { return log(x); }
/* threads used by the pattern loop of oracle_partial_update (1 without OpenMP) */
/* Mixture models (phylokernelmixture.h:20-460, phylokernelmixrate.h:22-450): the `ncat` categories of every
 * function below are (class, rate) components; component c uses the eigen-system of class CLS(c).  eval /
 * evec / inv_evec are then the per-class arrays concatenated and tip is [state][class][n]
 * (phylokernelmixture.h:151).  One class (the default) is the plain model. */
static int g_nclass = 1;
static int g_cat_class[512];
#define CLS(c) (g_nclass > 1 ? g_cat_class[(c)] : 0)
#define TIPIDX(e) ((size_t)CLS((e) / n) * n + (e) % n)
void oracle_set_mixture(int nclass, const int *cat_class, int ncat) {
    g_nclass = nclass > 1 ? nclass : 1;
    for (int c = 0; c < ncat && c < 512; c++) g_cat_class[c] = (nclass > 1) ? cat_class[c] : 0;
}

int oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}
double oracle_scaling_threshold(void) { return SCALING_THRESHOLD; }
double oracle_log_scaling_threshold(void) { return LOG_SCALING_THRESHOLD; }

/* phylotreesse.cpp:459-527 -- tip vectors = columns of U^-1; unknown = row sums;
 * DNA states 4..17 = bitmask (state-3); protein 20..22 = B,Z,U. Table has
 * (state_unknown+1) rows of n doubles; rows the reference leaves untouched are zeroed. */
void oracle_tip_partial_lh(int n, int seq_type, int state_unknown,
                           const double *inv_evec, double *tip) {
    memset(tip, 0, sizeof(double) * (size_t)(state_unknown + 1) * n);
    for (int state = 0; state < n; state++)
        for (int i = 0; i < n; i++) tip[state * n + i] = inv_evec[i * n + state];
    for (int i = 0; i < n; i++) {
        double s = 0.0;
        for (int x = 0; x < n; x++) s += inv_evec[i * n + x];
        tip[state_unknown * n + i] = s;
    }
    if (seq_type == ORACLE_SEQ_DNA && n == 4) {
        for (int state = 4; state < 18 && state < state_unknown; state++) {
            int cstate = state - n + 1;
            for (int i = 0; i < n; i++) {
                double s = 0.0;
                for (int x = 0; x < n; x++)
                    if (cstate & (1 << x)) s += inv_evec[i * n + x];
                tip[state * n + i] = s;
            }
        }
    } else if (seq_type == ORACLE_SEQ_PROTEIN && n == 20) {
        static const int ambi_aa[3] = {4 + 8, 32 + 64, 512 + 1024};
        for (int k = 0; k < 3 && 20 + k < state_unknown; k++)
            for (int i = 0; i < n; i++) {
                double s = 0.0;
                for (int x = 0; x < 11; x++)
                    if (ambi_aa[k] & (1 << x)) s += inv_evec[i * n + x];
                tip[(20 + k) * n + i] = s;
            }
    }
}

/* K1, phylokernel.h:159-181: E[c][x][i] = U[x][i]*exp(eval[i]*(rate_c*len)). */
void oracle_echild(int n, int ncat, const double *eval, const double *evec,
                   const double *rates, double len, double *E) {
    for (int c = 0; c < ncat; c++) {
        double l = rates[c] * len;
        const double *ev = evec + (size_t)CLS(c) * n * n, *el = eval + (size_t)CLS(c) * n;
        for (int x = 0; x < n; x++)
            for (int i = 0; i < n; i++)
                E[(size_t)c * n * n + x * n + i] = ev[x * n + i] * exp(el[i] * l);
    }
}

/* K2, phylokernel.h:187-232,293-317: table[state][c*n+x] = dot4(E[c][x][:], tip[state][:]);
 * the STATE_UNKNOWN row is exactly 1.0. */
void oracle_tip_table(int n, int ncat, int state_unknown, const double *E,
                      const double *tip, double *table) {
    size_t block = (size_t)n * ncat;
    for (int state = 0; state < state_unknown; state++)
        for (int c = 0; c < ncat; c++)
            for (int x = 0; x < n; x++)
                table[state * block + c * n + x] =
                    dot4(&E[(size_t)c * n * n + x * n], &tip[((size_t)state * g_nclass + CLS(c)) * n], n);
    for (size_t x = 0; x < block; x++) table[state_unknown * block + x] = 1.0;
}

#ifdef __AVX__
#include <immintrin.h>
/* [sum(x0), sum(x1), sum(x2), sum(x3)] with each sum associated as (l0+l1)+(l2+l3): the same
 * hadd/blend/permute combination the reference uses (phylokernel.h:31-43), so the 4-state fast path
 * below is bit-identical to dot4(). */
static inline __m256d hsum4(__m256d x0, __m256d x1, __m256d x2, __m256d x3) {
    __m256d s01 = _mm256_hadd_pd(x0, x1);
    __m256d s23 = _mm256_hadd_pd(x2, x3);
    __m256d blend = _mm256_blend_pd(s01, s23, 12);
    __m256d perm = _mm256_permute2f128_pd(s01, s23, 0x21);
    return _mm256_add_pd(perm, blend);
}
/* E (4x4, row-major) times v (4) for one category -> 4 dot products */
static inline __m256d mat4_vec(const double *E, __m256d v) {
    return hsum4(_mm256_mul_pd(_mm256_loadu_pd(E), v), _mm256_mul_pd(_mm256_loadu_pd(E + 4), v),
                 _mm256_mul_pd(_mm256_loadu_pd(E + 8), v), _mm256_mul_pd(_mm256_loadu_pd(E + 12), v));
}
#endif

/* one pattern of the node update; returns its contribution to sum_scale */
static inline double update_one_pattern(int n, int ncat, size_t block, size_t ptn, const double *EL,
                                        const double *ER, const double *tabL, const double *tabR,
                                        const uint8_t *left_states, const double *left_plh,
                                        const short *left_scale, const uint8_t *right_states,
                                        const double *right_plh, const short *right_scale,
                                        const double *inv_evec, const double *ptn_freq,
                                        const double *ptn_invar, double *tmp, double *out_plh,
                                        short *out_scale) {
    double sum_scale = 0.0;
    {
        double *out = out_plh + ptn * block;
        const double *tl = left_states ? tabL + (size_t)left_states[ptn] * block : NULL;
        const double *tr = right_states ? tabR + (size_t)right_states[ptn] * block : NULL;
        const double *pl = left_states ? NULL : left_plh + ptn * block;
        const double *pr = right_states ? NULL : right_plh + ptn * block;
        double lh_max = 0.0;
        /* scale_num init: TIP-TIP 0 (:247); TIP-INT copy of right (:290); INT-INT sum (:418) */
        short sc = 0;
        if (!left_states) sc = (short)(sc + left_scale[ptn]);
        if (!right_states) sc = (short)(sc + right_scale[ptn]);
#ifdef __AVX__
        if (n == 4) { /* same arithmetic, 4 dot products per instruction group */
            __m256d vmax = _mm256_setzero_pd();
            const __m256d absmask = _mm256_castsi256_pd(_mm256_set1_epi64x(0x7fffffffffffffffLL));
            for (int c = 0; c < ncat; c++) {
                __m256d a = tl ? _mm256_loadu_pd(tl + c * 4) : mat4_vec(EL + c * 16, _mm256_loadu_pd(pl + c * 4));
                __m256d b = tr ? _mm256_loadu_pd(tr + c * 4) : mat4_vec(ER + c * 16, _mm256_loadu_pd(pr + c * 4));
                __m256d r = mat4_vec(inv_evec + (size_t)CLS(c) * 16, _mm256_mul_pd(a, b));
                _mm256_storeu_pd(out + c * 4, r);
                vmax = _mm256_max_pd(vmax, _mm256_and_pd(r, absmask));
            }
            double m4[4];
            _mm256_storeu_pd(m4, vmax);
            lh_max = fmax(fmax(m4[0], m4[1]), fmax(m4[2], m4[3]));
        } else
#endif
        for (int c = 0; c < ncat; c++) {
            for (int x = 0; x < n; x++) {
                double a = tl ? tl[c * n + x] : dot4(&EL[(size_t)c * n * n + x * n], &pl[c * n], n);
                double b = tr ? tr[c * n + x] : dot4(&ER[(size_t)c * n * n + x * n], &pr[c * n], n);
                tmp[x] = a * b;
            }
            for (int i = 0; i < n; i++) {
                double r = dot4(tmp, &inv_evec[(size_t)CLS(c) * n * n + i * n], n);
                out[c * n + i] = r;
                double ar = fabs(r);
                if (ar > lh_max) lh_max = ar;
            }
        }
        /* the TIP-TIP case has no scaling check at all (phylokernel.h:246-281) */
        if (!(left_states && right_states) &&
            lh_max < SCALING_THRESHOLD && ptn_invar[ptn] == 0.0) {
            for (size_t i = 0; i < block; i++) out[i] *= SCALING_THRESHOLD_INVER;
            sum_scale += LOG_SCALING_THRESHOLD * ptn_freq[ptn];
            sc = (short)(sc + 1);
        }
        out_scale[ptn] = sc;
        }
    return sum_scale;
}

/*
 * a7 / K3-K5, phylokernel.h:183-479.  One internal-node update in the reference's layout.
 * left_states/right_states != NULL marks a leaf child (one state byte per pattern,
 * alignment.cpp:470-472 encoding); otherwise left_plh/left_scale are the child's vectors.
 * As in the reference the caller passes the leaf (if exactly one) as `left`
 * (phylokernel.h:116-121).  Returns sum_scale (the amount added to lh_scale_factor).
 */
double oracle_partial_update(int n, int ncat, size_t nptn,
                             const double *eval, const double *evec, const double *inv_evec,
                             const double *rates, const double *tip, int state_unknown,
                             const uint8_t *left_states, const double *left_plh,
                             const short *left_scale, double left_len,
                             const uint8_t *right_states, const double *right_plh,
                             const short *right_scale, double right_len,
                             const double *ptn_freq, const double *ptn_invar,
                             double *out_plh, short *out_scale) {
    size_t block = (size_t)n * ncat;
    double *EL = (double *)malloc(sizeof(double) * block * n);
    double *ER = (double *)malloc(sizeof(double) * block * n);
    double *tmp = (double *)malloc(sizeof(double) * n);
    double *tabL = NULL, *tabR = NULL;
    double sum_scale = 0.0;
    oracle_echild(n, ncat, eval, evec, rates, left_len, EL);
    oracle_echild(n, ncat, eval, evec, rates, right_len, ER);
    if (left_states) {
        tabL = (double *)malloc(sizeof(double) * (state_unknown + 1) * block);
        oracle_tip_table(n, ncat, state_unknown, EL, tip, tabL);
    }
    if (right_states) {
        tabR = (double *)malloc(sizeof(double) * (state_unknown + 1) * block);
        oracle_tip_table(n, ncat, state_unknown, ER, tip, tabR);
    }
    /* like the reference (phylokernel.h:251,335,410: "#pragma omp parallel for reduction(+: sum_scale)"
     * over ptn); every pattern is independent */
#ifdef _OPENMP
#pragma omp parallel reduction(+ : sum_scale) if (nptn >= 2048) /* small test cases: no thread team */
    {
        double *tmp_t = (double *)malloc(sizeof(double) * n);
#pragma omp for schedule(static)
        for (size_t ptn = 0; ptn < nptn; ptn++)
            sum_scale += update_one_pattern(n, ncat, block, ptn, EL, ER, tabL, tabR, left_states, left_plh,
                                            left_scale, right_states, right_plh, right_scale, inv_evec,
                                            ptn_freq, ptn_invar, tmp_t, out_plh, out_scale);
        free(tmp_t);
    }
#else
    for (size_t ptn = 0; ptn < nptn; ptn++)
        sum_scale += update_one_pattern(n, ncat, block, ptn, EL, ER, tabL, tabR, left_states, left_plh,
                                        left_scale, right_states, right_plh, right_scale, inv_evec, ptn_freq,
                                        ptn_invar, tmp, out_plh, out_scale);
#endif
    free(EL); free(ER); free(tmp); free(tabL); free(tabR);
    return sum_scale;
}

/* val[c*n+i] = exp(eval[i]*rate_c*len)*prop_c  (phylokernel.h:767-775) */
static void branch_val(int n, int ncat, const double *eval, const double *rates,
                       const double *props, double len, double *val) {
    for (int c = 0; c < ncat; c++) {
        double l = rates[c] * len;
        for (int i = 0; i < n; i++) val[c * n + i] = exp(eval[(size_t)CLS(c) * n + i] * l) * props[c];
    }
}

/* lane-strided running sum over patterns then (l0+l1)+(l2+l3):
 * phylokernel.h:836,954 + vectorclass/vectorf256.h:1652-1657 */
/* vector width of the reference instantiation: Vec4d (AVX) for 4 / 20 / 64 states, Vec2d (SSE) for binary data */
/* ... and plain running sums (width 1) for every other state count: those go to the reference's SCALAR kernels
 * (phylotreesse.cpp:281-309 -> :1013-1339: `lh_ptn += val * partial_lh_node * partial_lh_dad`, `tree_lh += lh_ptn * freq`) */
#define VCW(n) ((n) == 2 ? 2 : (((n) == 4 || (n) == 20 || (n) == 64) ? 4 : 1))
static inline double hsum_l(const double *l, int vc) { return vc == 1 ? l[0] : (vc == 2 ? l[0] + l[1] : (l[0] + l[1]) + (l[2] + l[3])); }
typedef struct { double l[4]; int vc; } lane4;
static inline void lane4_acc(lane4 *s, size_t ptn, double v, double f) {
    const size_t k = ptn % (size_t)s->vc;
    s->l[k] = v * f + s->l[k];
}
static inline double lane4_sum(const lane4 *s) { return hsum_l(s->l, s->vc); }

/*
 * a8 / K6, phylokernel.h:733-1020 (no ascertainment-bias patterns).
 * dad_states != NULL: `dad` is a leaf (tip-internal form :779-866) and node_plh is the
 * internal side; otherwise internal-internal (:912-966) on dad_plh/node_plh.
 * Writes pattern_lh[ptn] = log|lh_ptn| and returns sum_ptn f*log|lh_ptn| -- the caller
 * adds the two lh_scale_factors (:751).  Includes the NaN/Inf repair of :848-866.
 */
double oracle_branch_lnl(int n, int ncat, size_t nptn, const double *eval,
                         const double *rates, const double *props, double len,
                         const double *tip, const uint8_t *dad_states, const double *dad_plh,
                         const double *node_plh, const double *ptn_freq,
                         const double *ptn_invar, double *pattern_lh) {
    size_t block = (size_t)n * ncat;
    double *val = (double *)malloc(sizeof(double) * block);
    branch_val(n, ncat, eval, rates, props, len, val);
    lane4 fin = {{0, 0, 0, 0}, VCW(n)};
    const int vc = VCW(n);
    for (size_t ptn = 0; ptn < nptn; ptn++) {
        const double *b = node_plh + ptn * block;
        double l[4];
        if (dad_states) {
            const double *t = tip + (size_t)dad_states[ptn] * n * g_nclass;
            for (int k = 0; k < vc; k++) l[k] = (val[k] * t[TIPIDX(k)]) * b[k];
            for (size_t i = (size_t)vc; i < block; i += (size_t)vc)
                for (int k = 0; k < vc; k++)
                    l[k] = (val[i + k] * t[TIPIDX(i + k)]) * b[i + k] + l[k];
        } else {
            const double *a = dad_plh + ptn * block;
            for (int k = 0; k < vc; k++) l[k] = 0.0;
            for (size_t i = 0; i < block; i += (size_t)vc)
                for (int k = 0; k < vc; k++)
                    l[k] = (val[i + k] * b[i + k]) * a[i + k] + l[k];
        }
        double lh = hsum_l(l, vc) + ptn_invar[ptn];
        lh = log(fabs(lh));
        pattern_lh[ptn] = lh;
        lane4_acc(&fin, ptn, lh, ptn_freq[ptn]);
    }
    double tree_lh = lane4_sum(&fin);
    if (isnan(tree_lh) || isinf(tree_lh)) {
        tree_lh = 0.0;
        for (size_t ptn = 0; ptn < nptn; ptn++) {
            if (isnan(pattern_lh[ptn]) || isinf(pattern_lh[ptn]))
                pattern_lh[ptn] = LOG_SCALING_THRESHOLD * 4;
            tree_lh += pattern_lh[ptn] * ptn_freq[ptn];
        }
    }
    free(val);
    return tree_lh;
}

/* K7, phylokernel.h:535-573: theta = tip(state) .* dad_plh (leaf form) or node_plh .* dad_plh */
void oracle_theta(int n, int ncat, size_t nptn, const double *tip,
                  const uint8_t *dad_states, const double *dad_plh, const double *node_plh,
                  double *theta) {
    size_t block = (size_t)n * ncat;
    for (size_t ptn = 0; ptn < nptn; ptn++) {
        const double *b = node_plh + ptn * block;
        double *th = theta + ptn * block;
        if (dad_states) {
            const double *t = tip + (size_t)dad_states[ptn] * n * g_nclass;
            for (size_t i = 0; i < block; i++) th[i] = t[TIPIDX(i)] * b[i];
        } else {
            const double *a = dad_plh + ptn * block;
            for (size_t i = 0; i < block; i++) th[i] = a[i] * b[i];
        }
    }
}

/* a9 / K8, phylokernel.h:516-532,583-651: df, ddf from theta at branch length len. */
void oracle_derv(int n, int ncat, size_t nptn, const double *eval, const double *rates,
                 const double *props, double len, const double *theta,
                 const double *ptn_freq, const double *ptn_invar, double *df, double *ddf) {
    size_t block = (size_t)n * ncat;
    double *v0 = (double *)malloc(sizeof(double) * block * 3);
    double *v1 = v0 + block, *v2 = v1 + block;
    for (int c = 0; c < ncat; c++)
        for (int i = 0; i < n; i++) {
            double cof = eval[(size_t)CLS(c) * n + i] * rates[c];
            double val = exp(cof * len) * props[c];
            v0[c * n + i] = val;
            v1[c * n + i] = cof * val;
            v2[c * n + i] = cof * v1[c * n + i];
        }
    lane4 sdf = {{0, 0, 0, 0}, VCW(n)}, sddf = {{0, 0, 0, 0}, VCW(n)};
    const int vc = VCW(n);
    for (size_t ptn = 0; ptn < nptn; ptn++) {
        const double *th = theta + ptn * block;
        double p[4], d1[4], d2[4];
        for (int k = 0; k < vc; k++) {
            p[k] = v0[k] * th[k]; d1[k] = v1[k] * th[k]; d2[k] = v2[k] * th[k];
        }
        for (size_t i = (size_t)vc; i < block; i += (size_t)vc)
            for (int k = 0; k < vc; k++) {
                p[k] = th[i + k] * v0[i + k] + p[k];
                d1[k] = th[i + k] * v1[i + k] + d1[k];
                d2[k] = th[i + k] * v2[i + k] + d2[k];
            }
        double lh = hsum_l(p, vc) + ptn_invar[ptn];
        double inv = 1.0 / fabs(lh);
        double dfp = hsum_l(d1, vc) * inv;
        double ddfp = hsum_l(d2, vc) * inv;
        ddfp = ddfp - dfp * dfp;
        lane4_acc(&sdf, ptn, dfp, ptn_freq[ptn]);
        lane4_acc(&sddf, ptn, ddfp, ptn_freq[ptn]);
    }
    *df = lane4_sum(&sdf);
    *ddf = lane4_sum(&sddf);
    if (isnan(*df) || isinf(*df)) { *df = 0.0; *ddf = 0.0; }
    free(v0);
}

/* a10 / K9, phylokernel.h:1022-1122: lnL from theta (caller adds both lh_scale_factors). */
double oracle_lnl_from_theta(int n, int ncat, size_t nptn, const double *eval,
                             const double *rates, const double *props, double len,
                             const double *theta, const double *ptn_freq,
                             const double *ptn_invar, double *pattern_lh) {
    size_t block = (size_t)n * ncat;
    double *val = (double *)malloc(sizeof(double) * block);
    for (int c = 0; c < ncat; c++)
        for (int i = 0; i < n; i++) {
            double cof = eval[(size_t)CLS(c) * n + i] * rates[c];
            val[c * n + i] = exp(cof * len) * props[c];
        }
    lane4 fin = {{0, 0, 0, 0}, VCW(n)};
    const int vc = VCW(n);
    for (size_t ptn = 0; ptn < nptn; ptn++) {
        const double *th = theta + ptn * block;
        double p[4];
        for (int k = 0; k < vc; k++) p[k] = val[k] * th[k];
        for (size_t i = (size_t)vc; i < block; i += (size_t)vc)
            for (int k = 0; k < vc; k++) p[k] = th[i + k] * val[i + k] + p[k];
        double lh = hsum_l(p, vc) + ptn_invar[ptn];
        lh = log(fabs(lh));
        pattern_lh[ptn] = lh;
        lane4_acc(&fin, ptn, lh, ptn_freq[ptn]);
    }
    double tree_lh = lane4_sum(&fin);
    if (isnan(tree_lh) || isinf(tree_lh)) {
        tree_lh = 0.0;
        for (size_t ptn = 0; ptn < nptn; ptn++) {
            if (isnan(pattern_lh[ptn]) || isinf(pattern_lh[ptn]))
                pattern_lh[ptn] = LOG_SCALING_THRESHOLD * 4;
            tree_lh += pattern_lh[ptn] * ptn_freq[ptn];
        }
    }
    free(val);
    return tree_lh;
}

/* ---------------------------------------------------------------------------------------------
 * Ascertainment-bias correction (+ASC): the unobserved constant patterns are the trailing
 * `n_unobs` patterns of every array (phylokernel.h:87,100: nptn = aln->size() + unobserved_ptns).
 * The callers above are run on the observed prefix only; these two functions restate what the
 * reference does with the tail.
 * ------------------------------------------------------------------------------------------- */

/* prob_const of computeLikelihoodBranchEigenSIMD, phylokernel.h:868-909 (leaf form) and :968-1005
 * (internal form): sum over the unobserved patterns of lh_ptn, where the block sum is multiplied
 * by 2^-256 ONCE when the summed scale counters are >= 1 (the 2016-01-21 bugfix, :894-897,:989-992),
 * then ptn_invar is added.  All pointers address the first unobserved pattern. */
double oracle_asc_prob_const_branch(int n, int ncat, size_t n_unobs, const double *eval,
                                    const double *rates, const double *props, double len,
                                    const double *tip, const uint8_t *dad_states, const double *dad_plh,
                                    const short *dad_scale, const double *node_plh,
                                    const short *node_scale, const double *ptn_invar) {
    const int vc = VCW(n);
    size_t block = (size_t)n * ncat;
    double *val = (double *)malloc(sizeof(double) * block);
    branch_val(n, ncat, eval, rates, props, len, val);
    double prob_const = 0.0;
    for (size_t ptn = 0; ptn < n_unobs; ptn++) {
        const double *b = node_plh + ptn * block;
        double l[4];
        int sc = node_scale[ptn];
        if (dad_states) {
            const double *t = tip + (size_t)dad_states[ptn] * n * g_nclass;
            for (int k = 0; k < vc; k++) l[k] = (val[k] * t[TIPIDX(k)]) * b[k];
            for (size_t i = (size_t)vc; i < block; i += (size_t)vc)
                for (int k = 0; k < vc; k++) l[k] = (val[i + k] * t[TIPIDX(i + k)]) * b[i + k] + l[k];
        } else {
            const double *a = dad_plh + ptn * block;
            sc += dad_scale[ptn];
            for (int k = 0; k < vc; k++) l[k] = 0.0;
            for (size_t i = 0; i < block; i += (size_t)vc)
                for (int k = 0; k < vc; k++) l[k] = (val[i + k] * b[i + k]) * a[i + k] + l[k];
        }
        if (sc >= 1)
            for (int k = 0; k < vc; k++) l[k] *= SCALING_THRESHOLD;
        prob_const += hsum_l(l, vc) + ptn_invar[ptn];
    }
    free(val);
    return prob_const;
}

/* The tail of computeLikelihoodDervEigenSIMD (phylokernel.h:655-725, no rescale there) and of
 * computeLikelihoodFromBufferEigenSIMD (:1124-1187, rescale by sum_scale_num >= 1) on theta.
 * out[0] = sum lh_ptn, out[1] = sum df_ptn, out[2] = sum ddf_ptn over the unobserved patterns. */
void oracle_asc_theta_sums(int n, int ncat, size_t n_unobs, const double *eval, const double *rates,
                           const double *props, double len, const double *theta,
                           const short *sum_scale /* NULL: no rescale (Derv) */,
                           const double *ptn_invar, double *out) {
    const int vc = VCW(n);
    size_t block = (size_t)n * ncat;
    double *v0 = (double *)malloc(sizeof(double) * block * 3);
    double *v1 = v0 + block, *v2 = v1 + block;
    for (int c = 0; c < ncat; c++)
        for (int i = 0; i < n; i++) {
            double cof = eval[(size_t)CLS(c) * n + i] * rates[c];
            double v = exp(cof * len) * props[c];
            v0[c * n + i] = v;
            v1[c * n + i] = cof * v;
            v2[c * n + i] = cof * v1[c * n + i];
        }
    out[0] = out[1] = out[2] = 0.0;
    for (size_t ptn = 0; ptn < n_unobs; ptn++) {
        const double *th = theta + ptn * block;
        double p[4], d1[4], d2[4];
        for (int k = 0; k < vc; k++) { p[k] = v0[k] * th[k]; d1[k] = v1[k] * th[k]; d2[k] = v2[k] * th[k]; }
        for (size_t i = (size_t)vc; i < block; i += (size_t)vc)
            for (int k = 0; k < vc; k++) {
                p[k] = th[i + k] * v0[i + k] + p[k];
                d1[k] = th[i + k] * v1[i + k] + d1[k];
                d2[k] = th[i + k] * v2[i + k] + d2[k];
            }
        if (sum_scale && sum_scale[ptn] >= 1)
            for (int k = 0; k < vc; k++) p[k] *= SCALING_THRESHOLD;
        out[0] += hsum_l(p, vc) + ptn_invar[ptn];
        out[1] += hsum_l(d1, vc);
        out[2] += hsum_l(d2, vc);
    }
    free(v0);
}

/* ---- UFBoot / RELL ---------------------------------------------------------------
 * PhyloTree::computePatternLikelihood (phylotree.cpp:1200-1230): per-pattern lnL with the scaling
 * events of both ends of the evaluated branch put back: ptn_lh = _pattern_lh + (sc_a + sc_b) * log(2^-256). */
void oracle_pattern_lh_scaled(size_t nptn, const double *pattern_lh, const short *sc_a /* may be NULL */,
                              const short *sc_b /* may be NULL */, double *out) {
    for (size_t p = 0; p < nptn; p++) {
        int s = (sc_a && sc_a[p] > 0 ? sc_a[p] : 0) + (sc_b && sc_b[p] > 0 ? sc_b[p] : 0);
        out[p] = pattern_lh[p] + s * LOG_SCALING_THRESHOLD;
    }
}

/* dotProductSIMD<float, Vec8f, 8> (phylokernel.h:55-61; dispatch phylotreeavx.cpp:26): eight float
 * lanes, lane k accumulates terms k, k+8, ... with an unfused multiply-add, then
 * horizontal_add(Vec8f) = ((l0+l1)+(l2+l3)) + ((l4+l5)+(l6+l7)) (vectorclass/vectorf256.h:897-903).
 * The arrays are zero padded to a multiple of 8 by the caller (iqtree.cpp:2697-2699). */
float oracle_dot_float8(const float *x, const float *y, size_t n) {
    volatile float l[8];
    for (int k = 0; k < 8; k++) l[k] = (k < (int)n) ? x[k] * y[k] : 0.0f;
    for (size_t i = 8; i < n; i += 8)
        for (int k = 0; k < 8 && i + k < n; k++) {
            volatile float prod = x[i + k] * y[i + k];
            l[k] = prod + l[k];
        }
    volatile float a = l[0] + l[1], b = l[2] + l[3], c = l[4] + l[5], d = l[6] + l[7];
    volatile float ab = a + b, cd = c + d;
    return ab + cd;
}

/* Multifurcating node, the reference's SCALAR kernel (phylotreesse.cpp:609-806; the SIMD kernels hand nodes of
 * degree > 3 to it, phylokernel.h:73-77): per pattern the product over ALL children of (E_child * child) in
 * probability space (plain running sums, not the lane-strided ones), then U^-1, then ONE scaling test with the
 * scalar kernel's rule: lh_max == 0 -> copy the unknown tip vector, count += 4 (phylotreesse.cpp:776-788), else
 * if ptn_invar == 0 multiply by 2^256, count += 1.  scale_num starts at the sum over the internal children.
 * child k: states_k != NULL -> leaf (state bytes), else plh_k / sc_k; returns this node's own sum_scale. */
double oracle_partial_update_multi(int n, int ncat, size_t nptn, int nchild, const double *eval, const double *evec,
                                   const double *inv_evec, const double *rates, const double *tip, int state_unknown,
                                   const uint8_t *const *states, const double *const *plh, const short *const *sc,
                                   const double *lens, const double *ptn_freq, const double *ptn_invar,
                                   double *out, short *out_scale) {
    const int block = n * ncat;
    double *E = (double *)malloc(sizeof(double) * (size_t)nchild * block * n);
    for (int k = 0; k < nchild; k++) oracle_echild(n, ncat, eval, evec, rates, lens[k], E + (size_t)k * block * n);
    double sum_scale = 0.0;
#ifdef _OPENMP
#pragma omp parallel for reduction(+ : sum_scale) schedule(static)
#endif
    for (size_t ptn = 0; ptn < nptn; ptn++) {
        double all[block];
        for (int i = 0; i < block; i++) all[i] = 1.0;
        int cnt = 0;
        for (int k = 0; k < nchild; k++) {
            const double *Ek = E + (size_t)k * block * n;
            if (states[k]) {
                const int s = states[k][ptn];
                for (int c = 0; c < ncat; c++)
                    for (int x = 0; x < n; x++) {
                        double v = 1.0; /* STATE_UNKNOWN row (phylotreesse.cpp:677-681) */
                        if (s != state_unknown) {
                            v = 0.0;
                            for (int i = 0; i < n; i++) v += Ek[(size_t)c * n * n + x * n + i] * tip[(size_t)s * n * g_nclass + TIPIDX(c * n + i)];
                        }
                        all[c * n + x] *= v;
                    }
            } else {
                cnt += sc[k][ptn];
                const double *ch = plh[k] + ptn * block;
                for (int c = 0; c < ncat; c++)
                    for (int x = 0; x < n; x++) {
                        double v = 0.0;
                        for (int i = 0; i < n; i++) v += Ek[(size_t)c * n * n + x * n + i] * ch[c * n + i];
                        all[c * n + x] *= v;
                    }
            }
        }
        double *o = out + ptn * block;
        double lh_max = 0.0;
        for (int c = 0; c < ncat; c++) {
            const double *ie = inv_evec + (size_t)CLS(c) * n * n;
            for (int i = 0; i < n; i++) {
                double r = 0.0;
                for (int x = 0; x < n; x++) r += all[c * n + x] * ie[i * n + x];
                o[c * n + i] = r;
                if (fabs(r) > lh_max) lh_max = fabs(r);
            }
        }
        if (lh_max < SCALING_THRESHOLD) {
            if (lh_max == 0.0) {
                for (int c = 0; c < ncat; c++)
                    for (int i = 0; i < n; i++) o[c * n + i] = tip[(size_t)state_unknown * n * g_nclass + TIPIDX(c * n + i)];
                sum_scale += LOG_SCALING_THRESHOLD * 4 * ptn_freq[ptn];
                cnt += 4;
            } else if (ptn_invar[ptn] == 0.0) {
                for (int i = 0; i < block; i++) o[i] *= SCALING_THRESHOLD_INVER;
                sum_scale += LOG_SCALING_THRESHOLD * ptn_freq[ptn];
                cnt += 1;
            }
        }
        out_scale[ptn] = (short)cnt;
    }
    free(E);
    return sum_scale;
}
