// kernels_mfma.hip -- gfx950 kernels for the 20-state (protein) and 64-state (codon) paths.
//
// Here a node update is a genuine dense contraction per category (phylokernel.h:420-459):
//     out[c] = U^-1 * ( (U*diag(exL[c]) * left[c]) .* (U*diag(exR[c]) * right[c]) )
// with n x n matrices against n x (patterns) panels, so it runs on the fp64 matrix cores
// (v_mfma_f64_16x16x4_f64).  Mapping:
//   * a wave owns a tile of 16 patterns = the N dimension of the MFMA; the vectors live in HBM
//     as [tile16][c][i][16 patterns], which is at once the B-operand image (k-step s of a
//     child panel = 64 consecutive doubles, one coalesced 512-B load) and the D-result image
//     (register r of M-tile m = rows 16m+4r..+3 = 64 consecutive doubles);
//   * the A operands are only U and U^-1 (zero-padded to a multiple of 16 rows): the per-branch
//     part of E = U*diag(ex) is folded into the B operand (one v_mul_f64 per k-step), so the
//     matrices are staged into LDS once per launch, not once per branch;
//   * the f64 accumulator layout (row = (lane>>4)+4r, col = lane&15) is exactly the B layout of
//     the next product's k-step 4m+r, so the Hadamard product and the third contraction chain
//     in registers with no lane movement and no LDS round trip;
//   * leaves use tip_partial_lh columns as their B panel (read from the U^-1 image in LDS);
//     the reference's "unknown state row is exactly 1.0" (phylokernel.h:228-232) is restored
//     by a select on the accumulators.
// Scaling follows the SIMD rule (phylokernel.h:461-474): per pattern max |out| over the whole
// block (taken on the high words, see kScalingThresholdHi); the vector is stored unscaled per
// category and re-scaled in place in the rare case.
// Kernels: k_traverse_mfma (generic category count, mixtures of 4 / 64 states: both children from
// memory), k_traverse_mfma2 (20 and 64 states, 1 or 4 categories: previous result in registers,
// streamed child prefetched, leaf tables, parked results in LDS, staged plans),
// k_traverse_mfma_mix20 (20-state mixtures), k_traverse_mfma_rows64 / _top64 (64 states, a tile
// shared by four waves), k_leaf_tables (K2), k_stream_mfma (theta, derivatives, branch lnL).
// Cost model measured for gfx950: fp64 vector instructions add to the time of
// the fp64 matrix instructions (same unit), 32-bit vector and LDS / memory instructions overlap.
#include <algorithm>
#include <type_traits>

#include "iqhip_internal.h"

namespace iqhip {

// constant-address-space view of the plan: the op descriptors are wave-uniform, so their fields become
// s_load results (SGPRs), and pointers loaded through it are known to be global -- without it every
// child load / result store is a FLAT instruction, which also counts in lgkmcnt and so is drained by
// every wait on an LDS read.
#define CONST_AS __attribute__((address_space(4)))
// LDS reads spelled with their address space: a generic pointer lets the optimiser merge "tip value from LDS" and
// "previous result from a register array" into ONE flat load through a selected address, which parks the array in
// scratch and -- flat loads return out of order -- turns the wait into s_waitcnt vmcnt(0), i.e. a drain of every
// outstanding store and prefetch once per category step (measured r02: the 20-state kernel's whole stall)
#define LDS_AS __attribute__((address_space(3)))
template <typename T>
__device__ __forceinline__ const LDS_AS T *as_lds(const T *p) {
    return (const LDS_AS T *)p;
}

// ---------------------------------------------------------------------------------------
// IQHIP_WAVE_TRACE (timing-study build, tools/build_alt.sh trace -DIQHIP_WAVE_TRACE; never defined in the shipped
// library): every wave of a traversal kernel records its begin/end (constant 100 MHz clock + shader clock) and the SIMD
// it ran on; a few waves also stamp the shader clock at the phases of every (op, category) step.  The host appends the
// records of selected launches to $IQHIP_TRACE_FILE (tools/wave_trace.py reads it).
// ---------------------------------------------------------------------------------------
#ifdef IQHIP_WAVE_TRACE
struct WaveRec { unsigned long long rt0, rt1, ct0, ct1; unsigned hw, xcc, vblock, wave, nstamp, det, kind, pad; };
#define TRACE_MAXW (1 << 15)
#define TRACE_NDET 48
#define TRACE_MAXSTAMP 2048
__device__ WaveRec g_wrec[TRACE_MAXW];
__device__ unsigned g_wrec_n;
__device__ unsigned long long g_stamps[TRACE_NDET][TRACE_MAXSTAMP];
__device__ unsigned g_det_n;
struct WaveTracer {
    unsigned long long rt0, ct0;
    int det;
    unsigned n, vblock, wave, kind;
    __device__ __forceinline__ void begin(int vb, int w, int k) {
        vblock = vb; wave = w; kind = k; n = 0; det = -1;
        if ((vb % 37) == 0 && w == (vb / 37) % 4) {
            unsigned d = 0;
            if ((threadIdx.x & 63) == 0) d = atomicAdd(&g_det_n, 1u);
            d = __builtin_amdgcn_readfirstlane(d);
            if (d < TRACE_NDET) det = (int)d;
        }
        rt0 = __builtin_amdgcn_s_memrealtime();
        ct0 = __builtin_amdgcn_s_memtime();
    }
    __device__ __forceinline__ void stamp() {
        __builtin_amdgcn_sched_barrier(0);
        if (det >= 0) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if ((threadIdx.x & 63) == 0 && n < TRACE_MAXSTAMP) g_stamps[det][n] = t;
            n++;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    __device__ __forceinline__ void end() {
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime(), ct1 = __builtin_amdgcn_s_memtime();
        if ((threadIdx.x & 63) == 0) {
            const unsigned slot = atomicAdd(&g_wrec_n, 1u);
            if (slot < TRACE_MAXW) {
                WaveRec r;
                r.rt0 = rt0; r.rt1 = rt1; r.ct0 = ct0; r.ct1 = ct1;
                r.hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
                r.xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
                r.vblock = vblock; r.wave = wave; r.nstamp = n; r.det = (unsigned)det; r.kind = kind; r.pad = 0;
                g_wrec[slot] = r;
            }
        }
    }
};
#define TRACE_DECL WaveTracer wtr
#define TRACE_BEGIN(vb, w, k) wtr.begin(vb, w, k)
#define TRACE_STAMP() wtr.stamp()
#define TRACE_END() wtr.end()
#else
#define TRACE_DECL
#define TRACE_BEGIN(vb, w, k)
#define TRACE_STAMP()
#define TRACE_END()
#endif
template <typename T>
__device__ __forceinline__ const CONST_AS T *as_const(const T *p) {
    return (const CONST_AS T *)(p);
}

typedef double v4f64 __attribute__((ext_vector_type(4)));

// maximum over the four lane groups of a pattern (lanes p, p+16, p+32, p+48) on the VALU: gfx950's permlane swaps
// exchange half-waves / odd-even rows of two registers, so swap(x, x) hands every lane its partner's value without
// the LDS round trips of ds_bpermute (the per-op scaling test used to cost two of them, exposed, per op)
__device__ __forceinline__ double group_max(double v) {
    unsigned lo = __double2loint(v), hi = __double2hiint(v);
    auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    v = fmax(__hiloint2double(b[0], a[0]), __hiloint2double(b[1], a[1]));
    lo = __double2loint(v);
    hi = __double2hiint(v);
    auto c = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto d = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return fmax(__hiloint2double(d[0], c[0]), __hiloint2double(d[1], c[1]));
}

// The scaling test max|x| < 2^-256 on the 32-bit vector ALU: for finite doubles |x| < 2^-256 <=> the high word of |x|
// is below that of 2^-256 (whose low word is 0), and the high words order like the values.  fp64 vector instructions
// run on the unit that executes the fp64 matrix instructions and ADD to their time (3.1 ns each against 29.5 ns for a
// 16x16x4, tools/mfma_issue_probe.hip), 32-bit ones overlap with them: v_and + v_max_u32 replace two v_max_f64 per value.
constexpr unsigned kScalingThresholdHi = 0x2FF00000u;   // high word of 0x1p-256
__device__ __forceinline__ unsigned amax_hi(unsigned m, double v) {
    const unsigned h = (unsigned)__double2hiint(v) & 0x7fffffffu;
    return m > h ? m : h;
}
// exactly-zero test for the scalar kernel's `lh_max == 0.0` branch (IQHIP_OP_SCALAR_RULE ops only): any bit of |x|
__device__ __forceinline__ unsigned nonzero_bits(double v) {
    return ((unsigned)__double2hiint(v) & 0x7fffffffu) | (unsigned)__double2loint(v);
}
__device__ __forceinline__ unsigned group_max_u(unsigned v) {
    auto a = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    v = a[0] > a[1] ? a[0] : a[1];
    auto b = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return b[0] > b[1] ? b[0] : b[1];
}

__device__ __forceinline__ double wave_sum_m(double v) { return wave_sum64(v); }

struct TravMArgs {
    const DevOp *ops;
    const double *evec;
    const double *inv_evec;
    const double *tip;      // [(state_unknown+1)][n]
    const double *freq;
    const double *invar;
    const double *eval;
    const double *rates;
    const double *evalc;    // [ncat][n] eigenvalues of each category's class
    const double *tipc;     // [state][ncat][n]
    const int *cls;         // [ncat] class of each category (mixtures)
    const double *aimg;     // class 0: fragment image of U / U^-1 in the pipelined kernels' LDS order (iqhip_engine::d_aimg)
    const double *img;      // mixture A images, k_traverse_mfma_mix20 layout [class][U16|U4|Ui16|Ui4][KS][64]
    const double *img_generic;  // ... generic kernel layout [class][U|U^-1][MT][KS][64]
    double *slab;           // [nvals][nwaves]
    int64_t ntiles;         // tiles of 16 patterns
    int64_t nptn;
    const int *segs;        // {begin, nops} per segment; workgroup b works on segment b / ngroups
    int ngroups;            // workgroups per segment
    int nsegs_launch; // host side only: segments of this launch
    int nwaves;
    int ncat;
    int state_unknown;
    int *fold_flags;        // per result row: raised by a wave that rescaled patterns at that node (FoldArgs::flags)
    int hold_off;           // 20-state pipelined kernel: offset (doubles) of the waves' parking places in LDS, -1: none
    // a small plan inside the kernel arguments (iqhip_engine::plan_small; read by trav_mfma2_body only)
    int small_plan;
    int small_segs[2];
    DevOp small_ops[kSmallPlanOps];
};

// LDS image index of A[m][s][lane]
template <int KS>
__device__ __forceinline__ int aidx(int m, int s, int lane) { return (m * KS + s) * 64 + lane; }

// MIX: mixture model -- category c takes its A operands (U, U^-1 of class cls[c]) from the per-class images
// in global memory (L2-resident) and its tip vectors from tipc[state][c][:] instead of the LDS copies.
template <int N, int WG, bool MIX>
__global__ __launch_bounds__(WG) void k_traverse_mfma(const TravMArgs A) {
    constexpr int MT = (N + 15) / 16;  // M tiles (rows padded to 16)
    constexpr int KS = N / 4;          // k-steps of 4
    constexpr int WPB = WG / 64;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *sU = smem;                       // [MT][KS][64]   A image of U     (rows x, k = i)
    double *sUi = sU + MT * KS * 64;         // [MT][KS][64]   A image of U^-1  (rows i, k = x)
    double *sTipx = sUi + MT * KS * 64;      // [nx][N]        tip vectors of states >= N
    const int nx = A.state_unknown + 1 - N;
    double *sReg = sTipx + nx * N;           // per (op, child) exponentials [C*N] of the chunk
    const int C = A.ncat;
    const int B = C * N;

    for (int t = threadIdx.x; t < MT * KS * 64; t += WG) {
        const int l = t & 63, ms = t >> 6, s = ms % KS, m = ms / KS;
        const int row = 16 * m + (l & 15), k = 4 * s + (l >> 4);
        sU[t] = row < N ? A.evec[row * N + k] : 0.0;
        sUi[t] = row < N ? A.inv_evec[row * N + k] : 0.0;
    }
    for (int t = threadIdx.x; t < nx * N; t += WG) sTipx[t] = A.tip[N * N + t];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int seg = (int)blockIdx.x / A.ngroups;  // scalar
    const int k_begin = as_const(A.segs)[2 * seg], k_end = k_begin + as_const(A.segs)[2 * seg + 1];
    const int64_t tile = (int64_t)((int)blockIdx.x - seg * A.ngroups) * WPB + wave;
    const bool active = tile < A.ntiles;
    const int64_t tl = active ? tile : 0;
    const int p = lane & 15, g = lane >> 4;
    const int64_t ptn = tl * 16 + p;
    const size_t tbase = (size_t)tl * 16 * B;  // doubles
    const double freq = A.freq[ptn];
    const double invar = A.invar[ptn];

    int k = k_begin;
    while (k < k_end) {
        const int kn = A.ops[k].chunk_nops;
        __syncthreads();
        for (int t = threadIdx.x; t < kn * 2 * B; t += WG) {
            const int o = t / (2 * B), r = t - o * (2 * B), child = r / B, e = r - child * B;
            const CONST_AS DevOp &d = as_const(A.ops)[k + o];
            const double len = op_child_len(d, child);
            sReg[(child ? d.lds_right : d.lds_left) + e] = exp(A.evalc[e] * (A.rates[e / N] * len));
        }
        __syncthreads();
        if (!active) { k += kn; continue; }

        for (int kk = 0; kk < kn; kk++, k++) {
            const CONST_AS DevOp &op = as_const(A.ops)[k];
            const bool leafL = op.left_kind == CHILD_LEAF, leafR = op.right_kind == CHILD_LEAF;
            const double *exL = sReg + op.lds_left, *exR = sReg + op.lds_right;
            // the scale counter of a pattern is carried by its g == 0 lane only, so that every
            // lane reads back from memory nothing but what it stored itself
            int sc = 0, sL = 0, sR = 0;
            if (leafL) sL = op.sl[ptn]; else if (g == 0) sc += op.pf_sc[ptn];
            if (leafR) sR = op.sr[ptn]; else if (g == 0) sc += op.ld_sc[ptn];
            const bool unkL = leafL && sL == A.state_unknown, unkR = leafR && sR == A.state_unknown;
            const double *vL = op.pf + tbase, *vR = op.ld + tbase;
            double *dst = op.dst + tbase;
            unsigned lmax = 0, nz = 0;
            for (int c = 0; c < C; c++) {
                // A operands of this category's class
                const double *aU = sU, *aUi = sUi;
                if (MIX) {
                    aU = A.img_generic + (size_t)as_const(A.cls)[c] * 2 * MT * KS * 64;
                    aUi = aU + MT * KS * 64;
                }
                v4f64 YL[MT], YR[MT];
#pragma unroll
                for (int m = 0; m < MT; m++) { YL[m] = (v4f64){0, 0, 0, 0}; YR[m] = (v4f64){0, 0, 0, 0}; }
#pragma unroll
                for (int s = 0; s < KS; s++) {
                    const int i = 4 * s + g;  // this lane's k index
                    double bl, br;
                    if (leafL) {
                        // tip_partial_lh[state][i] = U^-1[i][state] for state < N (phylotreesse.cpp:464-471)
                        if (MIX) bl = A.tipc[((size_t)sL * C + c) * N + i];
                        else bl = sL < N ? as_lds(sUi)[aidx<KS>(i >> 4, sL >> 2, (sL & 3) * 16 + (i & 15))] : as_lds(sTipx)[(sL - N) * N + i];
                    } else {
                        bl = vL[(size_t)c * N * 16 + s * 64 + lane];
                    }
                    if (leafR) {
                        if (MIX) br = A.tipc[((size_t)sR * C + c) * N + i];
                        else br = sR < N ? as_lds(sUi)[aidx<KS>(i >> 4, sR >> 2, (sR & 3) * 16 + (i & 15))] : as_lds(sTipx)[(sR - N) * N + i];
                    } else {
                        br = vR[(size_t)c * N * 16 + s * 64 + lane];
                    }
                    bl *= exL[c * N + i];
                    br *= exR[c * N + i];
#pragma unroll
                    for (int m = 0; m < MT; m++) {
                        const double a = aU[aidx<KS>(m, s, lane)];
                        YL[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bl, YL[m], 0, 0, 0);
                        YR[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, br, YR[m], 0, 0, 0);
                    }
                }
                // T = YL .* YR; unknown-state leaf columns are exactly 1.0
                v4f64 T[MT];
#pragma unroll
                for (int m = 0; m < MT; m++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const double a = unkL ? 1.0 : YL[m][r];
                        const double b = unkR ? 1.0 : YR[m][r];
                        T[m][r] = a * b;
                    }
                v4f64 O[MT];
#pragma unroll
                for (int m = 0; m < MT; m++) O[m] = (v4f64){0, 0, 0, 0};
#pragma unroll
                for (int s = 0; s < KS; s++) {
                    const double bt = T[s >> 2][s & 3];  // accumulator layout == B layout of k-step s
#pragma unroll
                    for (int m = 0; m < MT; m++)
                        O[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(aUi[aidx<KS>(m, s, lane)], bt, O[m], 0, 0, 0);
                }
#pragma unroll
                for (int m = 0; m < MT; m++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int row = 16 * m + 4 * r + g;
                        if (16 * m + 4 * r < N) {  // whole 4-row group valid (N is a multiple of 4)
                            dst[(size_t)(c * N + row) * 16 + p] = O[m][r];
                            lmax = amax_hi(lmax, O[m][r]);
                            if (op.no_scale == 2) nz |= nonzero_bits(O[m][r]);
                        }
                    }
            }
            // column (pattern) max over the 4 lane groups (lmax orders the high words; a denormal below 2^-1042 still counts
            // as non-zero for the scalar rule's exact test)
            if (nz != 0 && lmax == 0) lmax = 1;
            lmax = group_max_u(lmax);
            const int rule = op.no_scale;
            const bool zero = rule == 2 && !(leafL && leafR) && lmax == 0;   // the scalar kernel's `lh_max == 0.0`, phylotreesse.cpp:777-788
            const bool do_scale = zero || (!(leafL && leafR) && (lmax < kScalingThresholdHi) && (invar == 0.0) && rule != 1);
            double my_scale = 0.0;
            if (__any(do_scale)) {
                if (do_scale) {
                    if (zero) for (int e = g; e < B; e += 4) dst[(size_t)e * 16 + p] = A.tipc[(size_t)A.state_unknown * B + e];
                    else for (int e = g; e < B; e += 4) dst[(size_t)e * 16 + p] *= kScalingThresholdInv;
                    sc += zero ? 4 : 1;
                    if (g == 0 && ptn < A.nptn) my_scale = (zero ? 4.0 : 1.0) * (kLogScalingThreshold * freq);
                }
            }
            if (g == 0) op.dst_sc[ptn] = (int16_t)sc;
            const double ws = __any(my_scale != 0.0) ? wave_sum_m(my_scale) : 0.0;  // (no rescaling in this op: nothing to add)
            if (lane == 0) {
                A.slab[(size_t)(2 + op.out_row) * A.nwaves + (int)tl] = ws;
                if (ws != 0.0) __hip_atomic_fetch_or(&A.fold_flags[2 + op.out_row], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// K2 for the matrix-core kernels (phylokernel.h:187-232, 293-317): a LEAF child contributes, per category, the column
// P_c(t)[:, state] = sum_i U[x][i] exp(eval_i r_c t) U^-1[i][state] of its branch's transition matrix -- a table
// look-up by the pattern's state in the reference, 0 flops per pattern.  k_leaf_tables builds, per submission, one
// table per (op, leaf side): tab[c][state][pos(x)], state < STATE_UNKNOWN (ambiguity states use their tip vectors,
// the unknown state is a select of exactly 1.0 in the kernel, phylokernel.h:228-232), with the reference's own
// association (E = U*ex rounded, then lane-strided unfused sums, (l0+l1)+(l2+l3)), so a leaf child's values are the
// oracle's bit for bit.  The tables (codon: 32 KB, protein+G4: 15 KB per leaf branch) stay L2-resident; the traversal
// kernels read a pattern's row straight into the accumulator image of the product that the matrix pipe no longer has
// to compute: 50 of the 97 child products of a 50-taxon traversal.
// pos(x): row x of an M-tile sits in accumulator register r = (x%16)/4 of lane group g = x%4, so a lane's four
// registers are made contiguous: pos = 16*(x/16) + 4*(x%4) + (x%16)/4; the 4 tail rows of N = 20 keep their place.
// ---------------------------------------------------------------------------------------
__host__ __device__ inline int leaf_tab_pos(int x, int n) {
    const int full = (n / 16) * 16;
    if (x >= full) return x;
    const int w = x & 15;
    return (x & ~15) + 4 * (w & 3) + (w >> 2);
}

// block = (job, category, state range): thread <-> row x of U (its E row lives in registers), the block's thread groups
// share out the states of the range; tip vectors are LDS broadcasts
template <int N>
__global__ __launch_bounds__(256) void k_leaf_tables(const TabJob *jobs, int ncat, int nstate_rows, int nsplit,
                                                     const double *__restrict__ eval, const double *__restrict__ evec,
                                                     const double *__restrict__ rates, const double *__restrict__ tip) {
    constexpr int GW = (N > 32) ? 64 : 32;   // threads per group (one thread per x, padded)
    constexpr int NG = 256 / GW;             // groups per block
    __shared__ double s_ex[N];
    extern __shared__ __attribute__((aligned(16))) double s_tip[];  // [rows of this block][N]
    const int job = blockIdx.x / (ncat * nsplit), r = blockIdx.x - job * ncat * nsplit, c = r / nsplit, sp = r - c * nsplit;
    const double len = jobs[job].len_p ? *jobs[job].len_p : jobs[job].len;
    double *tab = jobs[job].tab + (size_t)c * nstate_rows * N;
    const int per = (nstate_rows + nsplit - 1) / nsplit, s_lo = sp * per, s_hi = min(nstate_rows, s_lo + per);
    if (threadIdx.x < N) s_ex[threadIdx.x] = exp(eval[threadIdx.x] * (rates[c] * len));
    for (int t = threadIdx.x; t < (s_hi - s_lo) * N; t += 256) s_tip[t] = tip[(size_t)s_lo * N + t];
    __syncthreads();
    const int x = threadIdx.x % GW, grp = threadIdx.x / GW;
    if (x >= N) return;
    double E[N];  // E[x][i] = U[x][i] * exp(eval_i r_c t), rounded (K1, phylokernel.h:159-181)
#pragma unroll
    for (int i = 0; i < N; i++) E[i] = __dmul_rn(evec[x * N + i], s_ex[i]);
    const int pos = leaf_tab_pos(x, N);
    for (int s = s_lo + grp; s < s_hi; s += NG) {
        const double *b = s_tip + (size_t)(s - s_lo) * N;
        double l0 = __dmul_rn(E[0], b[0]), l1 = __dmul_rn(E[1], b[1]), l2 = __dmul_rn(E[2], b[2]), l3 = __dmul_rn(E[3], b[3]);
#pragma unroll
        for (int i = 4; i < N; i += 4) {
            l0 = __dadd_rn(__dmul_rn(E[i], b[i]), l0);
            l1 = __dadd_rn(__dmul_rn(E[i + 1], b[i + 1]), l1);
            l2 = __dadd_rn(__dmul_rn(E[i + 2], b[i + 2]), l2);
            l3 = __dadd_rn(__dmul_rn(E[i + 3], b[i + 3]), l3);
        }
        tab[(size_t)s * N + pos] = __dadd_rn(__dadd_rn(l0, l1), __dadd_rn(l2, l3));
    }
}

size_t leaf_table_doubles(const iqhip_engine *e) { return (size_t)e->ncat * e->state_unknown * e->n; }

hipError_t launch_leaf_tables(iqhip_engine *e, const TabJob *d_jobs, int njobs) {
    if (njobs <= 0) return hipSuccess;
    const int rows = e->state_unknown;
    const int nsplit = e->n >= 64 ? 4 : 1;
    const int per = (rows + nsplit - 1) / nsplit;
    const size_t lds = (size_t)per * e->n * sizeof(double);
    const dim3 grid((unsigned)(njobs * e->ncat * nsplit));
    if (e->n == 64)
        hipLaunchKernelGGL(k_leaf_tables<64>, grid, dim3(256), lds, e->stream, d_jobs, e->ncat, rows, nsplit, e->d_eval,
                           e->d_evec, e->d_rates, e->d_tip);
    else if (e->n == 20)
        hipLaunchKernelGGL(k_leaf_tables<20>, grid, dim3(256), lds, e->stream, d_jobs, e->ncat, rows, nsplit, e->d_eval,
                           e->d_evec, e->d_rates, e->d_tip);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

// Cherry tables (20 states): a node whose two children are leaves has one of (STATE_UNKNOWN + 1)^2 vectors per pattern.
// The engine computes them once per (pair of taxa, pendant lengths, model) by running the ordinary node update on a
// pseudo-alignment that lists every pair of states (iqhip_engine::pair: same kernels, same bits), and this kernel moves
// the result out of the tile layout [tile][c][row][16] into the order in which a lane of the traversal kernel holds
// an op's result: entry q = block / 2 double2, piece j * 4 + g = values (2j, 2j + 1) of lane group g, value s = VPL c + v
// (VPL = 4 n/16 + tail values per category and lane), v < 4 n/16: row 16 (v / 4) + 4 (v % 4) + g, else tail row 16 (n/16) + g.
struct CherryMoves {
    const double *src[32];
    double *dst[32];
    int n;
};
__global__ __launch_bounds__(256) void k_cherry_transpose(const CherryMoves M, int npairs, int block, int n) {
    const double *src = M.src[blockIdx.y];
    double *dst = M.dst[blockIdx.y];
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= npairs * block) return;
    const int q = t / block, idx = t - q * block;
    const int j = idx >> 3, g = (idx >> 1) & 3, hbit = idx & 1;
    const int mtf = n / 16, vpl = 4 * mtf + ((n % 16) == 4 ? 1 : 0);
    const int s = 2 * j + hbit, c = s / vpl, v = s - vpl * c;
    const int row = v < 4 * mtf ? 16 * (v >> 2) + 4 * (v & 3) + g : 16 * mtf + g;
    dst[t] = src[((size_t)(q >> 4) * block + c * n + row) * 16 + (q & 15)];
}

hipError_t launch_cherry_transpose(iqhip_engine *e, const double *const *src, double *const *dst, int n, int npairs) {
    for (int first = 0; first < n; first += 32) {
        CherryMoves M;
        M.n = std::min(32, n - first);
        for (int i = 0; i < 32; i++) {
            M.src[i] = src[first + (i < M.n ? i : 0)];
            M.dst[i] = dst[first + (i < M.n ? i : 0)];
        }
        hipLaunchKernelGGL(k_cherry_transpose, dim3((unsigned)((npairs * e->block + 255) / 256), (unsigned)M.n), dim3(256), 0,
                           e->stream, M, npairs, e->block, e->n);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// Pipelined variant (compile-time category count).  Differences from k_traverse_mfma:
//   * the previous op's result stays in registers: its accumulator image IS the B operand of
//     the next op (CHILD_PREV), consumed and overwritten in place category by category;
//   * the streamed child (CHILD_PF) of the next (op, category) step is requested one step ahead
//     (unconditional request folded onto a dummy window when not needed, as in k_traverse4);
//   * for N = 20 the A fragments of U and U^-1 (2 x 10 doubles) live in registers for the whole
//     launch; for N = 64 they are read from the LDS image next to the MFMA that uses them.
// Host canonical form as for DNA: left in {LEAF, PF}, right in {LEAF, PREV}; (PF, LOAD) reads the
// right child synchronously into the `prev` registers.
// ---------------------------------------------------------------------------------------
// CS > 1 (category split): the CS waves of a workgroup share ONE tile and own C = ncat/CS categories each, so a
// small alignment yields CS times the waves with 1/CS of the dependent MFMA chain per op; the only cross-wave step is
// the scaling maximum of a pattern (LDS + one workgroup barrier per op).
// TAB: LEAF children are table look-ups (k_leaf_tables) instead of U * (ex .* tip) products on the matrix pipe.
// tile0: first tile of this role (mixed-role top stages)
template <int N, int C, int WG, int CS, bool TAB>
__device__ __forceinline__ void trav_mfma2_body(const TravMArgs &A, const int vblock, const int64_t tile0 = 0) {
    // TABL (20 states): the K2 tables of the chunk's LEAF children are copied into LDS when the chunk is filled (the four
    // waves of a workgroup walk the same ops) and a lane reads its pattern's row with ds_read_b128: gathered from global
    // memory the rows cost more than the matrix products they replace once result stores are in flight (r02: 1.07 vs
    // 1.03 ms; even without any other traffic 0.78 vs 0.89 ms where the products removed are 34 % of the matrix work)
    constexpr bool TABL = TAB && (N < 64);
    // values of one category that a lane holds of an op's result (4 per 16-row tile + the tail row), see k_cherry_transpose
    constexpr int VPL = 4 * (N / 16) + ((N % 16) == 4 ? 1 : 0);
    // (20 states only.  Measured for 64 states, 50 x 20k codons: the extra block costs that kernel 39 spilled registers --
    // 0.400 ms without tables, 0.346 with, against 0.326 before the block existed; the two-waves-per-tile form of 20 states
    // has no registers to spare either)
    constexpr bool CHERRY = (N == 20) && ((C * VPL) % 2 == 0) && (CS == 1);
    // rows of U / U^-1: MTF full 16-row tiles on v_mfma_f64_16x16x4_f64 plus, for N = 20, the four
    // left-over rows on v_mfma_f64_4x4x4_4b_f64 (4 blocks = the tile's 4 groups of 4 patterns).
    // Lane layouts of the 4x4x4 form (measured, tools/mfma444_probe.hip): A[i][k] at lane
    // 16k+4blk+i, B[k][j] at 16k+4blk+j, D[i][j] at 16i+4blk+j -- i.e. its B operand and its D
    // result coincide with the 16x16x4 B fragment and with accumulator register 0 of a 16-row
    // tile, so the tail needs no lane movement and no padding (23 vs 71.5 cycles per instruction).
    constexpr int MTF = N / 16;
    constexpr bool TAIL4 = (N % 16) == 4;
    static_assert(N % 16 == 0 || TAIL4, "N must be 16m or 16m+4");
    constexpr int KS = N / 4;
    constexpr int WPB = WG / 64;
    constexpr int CT = C * CS;   // categories of the block
    constexpr int B = CT * N;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    // (20 states keep the fragments of U / U^-1 in registers, AREG below: no LDS image)
    constexpr int IMG1 = (N < 64) ? 0 : MTF * KS * 64, IMG4 = (N < 64) ? 0 : (TAIL4 ? KS * 64 : 0);
    double *sU = smem;                           // [MTF][KS][64]
    double *sUi = sU + IMG1;                     // [MTF][KS][64]
    double *sU4 = sUi + IMG1;                    // [KS][64] tail rows (TAIL4)
    double *sUi4 = sU4 + IMG4;
    // tip_partial_lh rows of states < N (= columns of U^-1): a plain [N][N] copy when it is small,
    // otherwise read out of the U^-1 fragment image
    constexpr bool TIP_COPY = (N * N * 8 <= 4096);
    double *sUiT = sUi4 + IMG4;
    double *sTipx = sUiT + (TIP_COPY ? N * N : 0);
    const int nx = A.state_unknown + 1 - N;
    double *sReg = sTipx + nx * N;

    // fragment images of U and U^-1: a straight 16-byte copy of the engine's pre-formatted image (only the 64-state
    // instantiations read them from LDS; 20 states keep the fragments in registers, read below)
    if constexpr (N >= 64) {
        constexpr int NIMG2 = (2 * MTF * KS * 64 + (TAIL4 ? 2 * KS * 64 : 0)) / 2;
        const double2 *src = reinterpret_cast<const double2 *>(A.aimg);
        double2 *dst2 = reinterpret_cast<double2 *>(smem);
#pragma unroll 8
        for (int t = threadIdx.x; t < NIMG2; t += WG) dst2[t] = src[t];
    }
    __shared__ int s_opi[64][4];      // chunk fill: per op of a pass the LDS offsets of its two regions, skip flags
    __shared__ double s_opl[64][2];   // ... child branch lengths
    __shared__ double s_evr[64 + 8];  // eigenvalues [N], rates [CT]
    if constexpr (N < 64)
        for (int t = threadIdx.x; t < N + CT; t += WG) s_evr[t < N ? t : 64 + (t - N)] = t < N ? A.eval[t] : A.rates[t - N];
    if (!TAB) {
        if (TIP_COPY)
            for (int t = threadIdx.x; t < N * N; t += WG) sUiT[t] = A.tip[t];  // tip[state][i] = U^-1[i][state]
        for (int t = threadIdx.x; t < nx * N; t += WG) sTipx[t] = A.tip[N * N + t];
    }

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int seg = vblock / A.ngroups;  // scalar
    // (a small plan is read out of the kernel-argument segment, constant memory like the plan buffer; A is the
    // kernel's only parameter -- k_traverse_mfma2 -- so its members sit at their struct offsets there)
    const CONST_AS char *kargs = (const CONST_AS char *)__builtin_amdgcn_kernarg_segment_ptr();
    const CONST_AS int *segs = A.small_plan ? (const CONST_AS int *)(kargs + offsetof(TravMArgs, small_segs)) : as_const(A.segs);
    const int k_begin = segs[2 * seg], k_end = k_begin + segs[2 * seg + 1];
    const int64_t tile = tile0 + (int64_t)(vblock - seg * A.ngroups) * (WPB / CS) + wave / CS;
    const int coff = (wave % CS) * C;          // first category of this wave
    const bool lead = (wave % CS) == 0;        // the wave that owns the tile's counters and sums
    const bool active = tile < A.ntiles;
    const int64_t tl = active ? tile : 0;
    const int p = lane & 15, g = lane >> 4;
    const int64_t ptn = tl * 16 + p;
    const size_t tbase = (size_t)tl * 16 * B;     // doubles into a vector slab
    const double freq = A.freq[ptn];
    const double invar = A.invar[ptn];
    const CONST_AS DevOp *ops = A.small_plan ? (const CONST_AS DevOp *)(kargs + offsetof(TravMArgs, small_ops)) : as_const(A.ops);
    const int S = A.state_unknown;                // rows of a leaf table
    TRACE_DECL;
    TRACE_BEGIN(vblock, wave, N);

    // AREG (20 states): the A fragments of U and U^-1 (16-row tile + 4-row tail: 4 x 5 doubles) stay in registers for
    // the whole launch; with them in LDS every k-step of every chain waited for an LDS round trip before its MFMAs
    constexpr bool AREG = (N < 64);
    static_assert(!AREG || MTF == 1, "register-resident fragments: one 16-row tile (+ tail)");
    double aU[AREG ? KS : 1], aU4[AREG ? KS : 1], aUi[AREG ? KS : 1], aUi4[AREG ? KS : 1];
    if constexpr (AREG) {
#pragma unroll
        for (int s = 0; s < KS; s++) {   // (coalesced reads of the engine's fragment image; MTF == 1 here)
            aU[s] = A.aimg[s * 64 + lane];
            aUi[s] = A.aimg[(MTF * KS + s) * 64 + lane];
            aU4[s] = TAIL4 ? A.aimg[(2 * MTF * KS + s) * 64 + lane] : 0.0;
            aUi4[s] = TAIL4 ? A.aimg[(2 * MTF * KS + KS + s) * 64 + lane] : 0.0;
        }
    }
    v4f64 prev[C][MTF];
    double prevT[C];  // tail rows 16*MTF+g
#pragma unroll
    for (int c = 0; c < C; c++) {
        prevT[c] = 0.0;
#pragma unroll
        for (int m = 0; m < MTF; m++) prev[c][m] = (v4f64){0, 0, 0, 0};
    }
    int prev_sc = 0;
    double PFn[KS];
    int pfn_sc = 0;
    int sLn = 0, sRn = 0;  // TAB: leaf states of the next op, requested one op ahead
    {   // prime: streamed child of (first op, category 0); the op after the last one is a sentinel
        const CONST_AS DevOp &f = ops[k_begin];
        const double *src = f.pf + ((f.real_mask & 1) ? tbase + (size_t)coff * N * 16 : 0);
#pragma unroll
        for (int s = 0; s < KS; s++) PFn[s] = src[s * 64 + lane];
        if (g == 0) pfn_sc = f.pf_sc[(f.real_mask & 1) ? ptn : (int64_t)p];
        sLn = f.sl[ptn];
        sRn = f.sr[ptn];
    }
    // CHILD_HOLD (20 states): a result parked in LDS by the op that produced it (push_hold), in the tile layout
    // [c][row][16 patterns], so that k-step s of category c is the 64 consecutive doubles at (c N + 4 s) 16
    // (two waves per tile could park their own categories -- measured: the extra state costs that kernel 33 spilled registers,
    // 1.62 vs 1.10 ms with the form in every stage, profiles/r03/experiments.txt)
    constexpr bool HOLDS = (N < 64) && (CS == 1);
    LDS_AS double *const hold = HOLDS && A.hold_off >= 0 ? (LDS_AS double *)(smem + A.hold_off) + (size_t)(wave / CS) * 16 * B + lane : nullptr;
    int hold_sc = 0;
    int k = k_begin;
    while (k < k_end) {
        const int kn = ops[k].chunk_nops;
        __syncthreads();
        // (20 states: op descriptors' fields staged per pass of 64 ops, eigenvalues and rates copied at kernel start -- every
        // exponential used to begin with dependent reads of both from memory, as in the 4-state kernel's fill: protein 0.963 ->
        // 0.943 ms.  The 64-state kernel keeps the plain form: with the staging it spills 30 registers instead of 18, 0.334 ->
        // 0.349 ms.)
        if constexpr (N >= 64) {
            for (int t = threadIdx.x; t < kn * 2 * B; t += WG) {
                const int o = t / (2 * B), r = t - o * (2 * B), child = r / B, e = r - child * B;
                const CONST_AS DevOp &d = ops[k + o];
                if (TAB && (child ? d.right_kind : d.left_kind) == CHILD_LEAF) continue;  // a table child needs no exponentials
                const double len = op_child_len(d, child);
                sReg[(child ? d.lds_right : d.lds_left) + e] = exp(A.eval[e % N] * (A.rates[e / N] * len));
            }
        } else
        for (int o0 = 0; o0 < kn; o0 += 64) {
            const int on = kn - o0 < 64 ? kn - o0 : 64;
            if (o0 > 0) __syncthreads();
            for (int t = threadIdx.x; t < on; t += WG) {
                const CONST_AS DevOp &d = ops[k + o0 + t];
                s_opi[t][0] = d.lds_left;
                s_opi[t][1] = d.lds_right;
                s_opi[t][2] = TAB && d.left_kind == CHILD_LEAF;    // a table child needs no exponentials
                s_opi[t][3] = TAB && d.right_kind == CHILD_LEAF;
                s_opl[t][0] = op_child_len(d, 0);
                s_opl[t][1] = op_child_len(d, 1);
            }
            __syncthreads();
            for (int t = threadIdx.x; t < on * 2 * B; t += WG) {
                const int o = t / (2 * B), r = t - o * (2 * B), child = r / B, e = r - child * B;
                if (s_opi[o][2 + child]) continue;
                sReg[s_opi[o][child] + e] = exp(s_evr[e % N] * (s_evr[64 + e / N] * s_opl[o][child]));
            }
        }
        if constexpr (TABL) {
            // the leaf children's tables [CT][S][N] (k_leaf_tables, L2-resident): straight 16-byte copies into their regions
            const int tab2 = CT * S * N / 2;   // double2 per table (N is even)
            for (int o = 0; o < kn; o++) {
                const CONST_AS DevOp &d = ops[k + o];
#pragma unroll
                for (int child = 0; child < 2; child++) {
                    if ((child ? d.right_kind : d.left_kind) != CHILD_LEAF) continue;
                    const double2 *src = reinterpret_cast<const double2 *>(child ? d.tabR : d.tabL);
                    double2 *dst2 = reinterpret_cast<double2 *>(sReg + (child ? d.lds_right : d.lds_left));
                    for (int t = threadIdx.x; t < tab2; t += WG) dst2[t] = src[t];
                }
            }
        }
        __syncthreads();
        if (!active) { k += kn; continue; }

        for (int kk = 0; kk < kn; kk++, k++) {
            TRACE_STAMP();   // op start
            const CONST_AS DevOp &op = ops[k];
            const CONST_AS DevOp &nxop = ops[k + 1];
            const bool leafL = op.left_kind == CHILD_LEAF, leafR = op.right_kind == CHILD_LEAF;
            const double *exL = sReg + op.lds_left, *exR = sReg + op.lds_right;
            // leaf states are requested one op ahead (sl / sr of non-leaf children and of the sentinel point at valid
            // dummy rows): fetched at the op itself, the first products of every leaf op waited out an L2 round trip
            int sc = 0;
            const int sL = sLn, sR = sRn;
            // (only for leaf children: requests to the dummy rows of the others cost 1.3 % of the protein traversal)
            if (nxop.left_kind == CHILD_LEAF) sLn = nxop.sl[ptn];
            if (nxop.right_kind == CHILD_LEAF) sRn = nxop.sr[ptn];
            const bool holdL = HOLDS && op.left_kind == CHILD_HOLD;
            const bool push = HOLDS && op.push_hold;
            // CHERRY: both children are leaves and the engine holds the node's vector for every pair of leaf states (L2 /
            // Infinity-Cache resident table, DevOp::cherry): this lane's 5 * C values arrive as 16-byte pieces and no
            // matrix instruction is issued for the op
            const double *const cherry = CHERRY ? op.cherry : nullptr;
            if (!leafL) sc += holdL ? hold_sc : pfn_sc;                 // pfn_sc / hold_sc: valid on g == 0 lanes
            // scalar fields of the op that are needed only in its tail: requested now
            int16_t *const dst_sc = op.dst_sc;
            const int out_row = op.out_row;
            const int no_scale = op.no_scale;
            if (op.right_kind == CHILD_LOAD) {
                // rare (PF, LOAD): read the right child now into the `prev` registers
                const double *src = op.ld + tbase;
#pragma unroll
                for (int c = 0; c < C; c++)
#pragma unroll
                    for (int s = 0; s < KS; s++) {
                        const double v = src[(size_t)(coff + c) * N * 16 + s * 64 + lane];
                        if (s < 4 * MTF) prev[c][s >> 2][s & 3] = v; else prevT[c] = v;
                    }
                if (g == 0) prev_sc = op.ld_sc[ptn];
            }
            if (!leafR) sc += prev_sc;
            const bool unkL = leafL && sL == A.state_unknown, unkR = leafR && sR == A.state_unknown;
            const bool anyUnk = __any(unkL || unkR);
            // tip_partial_lh row of this lane's pattern state (phylotreesse.cpp:464-527)
            auto tip_at = [&](int st, int i) -> double {
                if (st >= N) return as_lds(sTipx)[(st - N) * N + i];
                if (TIP_COPY) return as_lds(sUiT)[st * N + i];
                return as_lds(sUi)[aidx<KS>(i >> 4, st >> 2, (st & 3) * 16 + (i & 15))];  // U^-1[i][st] in the A image
            };
            // TAB: this lane's slice of the leaf children's table rows (4 contiguous doubles per M-tile)
            const double *rowL = (TABL ? sReg + op.lds_left : op.tabL) + (size_t)(sL < S ? sL : 0) * N + 4 * g;
            const double *rowR = (TABL ? sReg + op.lds_right : op.tabR) + (size_t)(sR < S ? sR : 0) * N + 4 * g;
            double *dst = op.dst + tbase;
            unsigned lmax = 0;
            bool from_table = false;
            if constexpr (CHERRY) {
                if (cherry) {
                    // a category's result: kept as the next op's `prev`, stored, parked if a later op of the launch wants it
                    auto finish_cat = [&](int c, const v4f64 (&O)[MTF], double o4) {
        #pragma unroll
                        for (int m = 0; m < MTF; m++) {
                            prev[c][m] = O[m];
        #pragma unroll
                            for (int r = 0; r < 4; r++) {
                                const int row = 16 * m + 4 * r + g;
        #ifndef IQHIP_MFMA_ABLATE_NOSTORE  // timing-only build switch; never defined in the shipped library
                                // (streaming hint: most results are not read again before they leave the L2; protein -1 %, codon -2 %)
                                __builtin_nontemporal_store(O[m][r], &dst[(size_t)((coff + c) * N + row) * 16 + p]);
        #endif
                                if (HOLDS && push) hold[((coff + c) * N + 16 * m + 4 * r) * 16] = O[m][r];
                                lmax = amax_hi(lmax, O[m][r]);
                            }
                        }
                        if (TAIL4) {
                            prevT[c] = o4;
        #ifndef IQHIP_MFMA_ABLATE_NOSTORE
                            __builtin_nontemporal_store(o4, &dst[(size_t)((coff + c) * N + 16 * MTF + g) * 16 + p]);
        #endif
                            if (HOLDS && push) hold[((coff + c) * N + 16 * MTF) * 16] = o4;
                            lmax = amax_hi(lmax, o4);
                        }
                    };
                    // (the streamed child of the next op's first category, as the last category step of any op requests it)
#ifdef IQHIP_MFMA_ABLATE_NOLOAD
                    const bool nreal = false;
#else
                    const bool nreal = nxop.real_mask & 1;
#endif
                    const double *nsrc = nxop.pf + (nreal ? tbase + (size_t)coff * N * 16 : 0);
                    if (nreal)   // (no request to the dummy window for a step that streams nothing: protein 0.952 -> 0.927 ms)
                    {
#pragma unroll
                    for (int s = 0; s < KS; s++) PFn[s] = nsrc[s * 64 + lane];
                    }
                    if (g == 0) pfn_sc = nxop.pf_sc[nreal ? ptn : (int64_t)p];
                    const double2 *src = reinterpret_cast<const double2 *>(cherry + (size_t)(sL * (A.state_unknown + 1) + sR) * B) +
                                         (coff * VPL / 2) * 4 + g;
                    {
                        double2 CTv[VPL * C / 2];   // values (2j, 2j + 1) of this lane; value VPL c + v: category c, register v
#pragma unroll
                        for (int j = 0; j < VPL * C / 2; j++) CTv[j] = src[j * 4];
                        auto val = [&](int s) { return (s & 1) ? CTv[s >> 1].y : CTv[s >> 1].x; };
#pragma unroll
                        for (int c = 0; c < C; c++) {
                            TRACE_STAMP();
                            v4f64 O[MTF];
#pragma unroll
                            for (int m = 0; m < MTF; m++)
                                O[m] = (v4f64){val(VPL * c + 4 * m), val(VPL * c + 4 * m + 1), val(VPL * c + 4 * m + 2), val(VPL * c + 4 * m + 3)};
                            finish_cat(c, O, TAIL4 ? val(VPL * c + (TAIL4 ? 4 * MTF : 0)) : 0.0);
                            TRACE_STAMP();
                        }
                    }
                    from_table = true;
                }
            }
            if (!from_table) {
#pragma unroll
            for (int c = 0; c < C; c++) {
                // streamed child of the next step: (k, c+1) or (k+1, 0); its k-step slices replace the
                // PFn registers one by one, right after this step has consumed them
                const CONST_AS DevOp &nd = (c + 1 < C) ? op : nxop;
#ifdef IQHIP_MFMA_ABLATE_NOLOAD  // timing-only build switch; never defined in the shipped library
                const bool nreal = false;
#else
                const bool nreal = nd.real_mask & 1;
#endif
                const double *nsrc = nd.pf + (nreal ? tbase + (size_t)(coff + ((c + 1 < C) ? c + 1 : 0)) * N * 16 : 0);
                // N = 20 (5 k-steps): cheaper to copy the operands out and issue the whole prefetch up
                // front; N = 64 (16 k-steps): stream it, the copies would not fit the register file.
                // Measured on one box: protein 1.51 ms (copy) vs 1.63 ms (stream); codon 0.708 vs 0.678.
                constexpr bool STREAM = (N >= 64);
                // one child product chain per non-table child (DOL / DOR compile-time): Y += U * (ex .* child)
                auto chain = [&](auto DOL, auto DOR, v4f64 (&YL)[MTF], v4f64 (&YR)[MTF], double &yl4, double &yr4) {
                    constexpr bool doL = decltype(DOL)::value, doR = decltype(DOR)::value;
                    double bl[STREAM ? 1 : KS], br[STREAM ? 1 : KS];
                    double exl[STREAM ? 1 : KS], exr[STREAM ? 1 : KS];   // the chain's exponentials, fetched together
                    if constexpr (!STREAM) {
#pragma unroll
                        for (int s = 0; s < KS; s++) {
                            exl[s] = doL ? exL[(coff + c) * N + 4 * s + g] : 0.0;
                            exr[s] = doR ? exR[(coff + c) * N + 4 * s + g] : 0.0;
                        }
#pragma unroll
                        for (int s = 0; s < KS; s++) bl[s] = PFn[s];
                        if (HOLDS && holdL) {
#pragma unroll
                            for (int s = 0; s < KS; s++) bl[s] = hold[((coff + c) * N + 4 * s) * 16];
                        }
                        if (nreal)
                        {
#pragma unroll
                        for (int s = 0; s < KS; s++) PFn[s] = nsrc[s * 64 + lane];
                        }
                        if (!TAB && leafL) {
#pragma unroll
                            for (int s = 0; s < KS; s++) bl[s] = tip_at(sL, 4 * s + g);
                        }
                        if (!TAB && leafR) {
#pragma unroll
                            for (int s = 0; s < KS; s++) br[s] = tip_at(sR, 4 * s + g);
                        } else {
#pragma unroll
                            for (int s = 0; s < KS; s++) br[s] = (s < 4 * MTF) ? prev[c][(s >> 2) < MTF ? (s >> 2) : 0][s & 3] : prevT[c];
                        }
                    }
                    // A fragments of U for the k-step after this one are fetched before this one's MFMAs are issued
                    // (STREAM keeps the k-steps in program order, so nothing else would cover the LDS latency)
                    double aC[MTF], exlC = 0.0, exrC = 0.0;
                    if constexpr (STREAM) {
#pragma unroll
                        for (int m = 0; m < MTF; m++) aC[m] = sU[aidx<KS>(m, 0, lane)];
                        if (doL) exlC = exL[(coff + c) * N + g];
                        if (doR) exrC = exR[(coff + c) * N + g];
                    }
#pragma unroll
                    for (int s = 0; s < KS; s++) {
                        const int i = 4 * s + g;
                        double vl = 0.0, vr = 0.0;
                        double aN[MTF], exlN = 0.0, exrN = 0.0;
                        if constexpr (STREAM) {
                            if (doL) vl = (!TAB && leafL) ? tip_at(sL, i) : PFn[s];
                            if (nreal)
                            PFn[s] = nsrc[s * 64 + lane];
                            if (doR) vr = (!TAB && leafR) ? tip_at(sR, i)
                                                          : ((s < 4 * MTF) ? prev[c][(s >> 2) < MTF ? (s >> 2) : 0][s & 3] : prevT[c]);
                            if (s + 1 < KS) {
#pragma unroll
                                for (int m = 0; m < MTF; m++) aN[m] = sU[aidx<KS>(m, s + 1, lane)];
                                if (doL) exlN = exL[(coff + c) * N + i + 4];
                                if (doR) exrN = exR[(coff + c) * N + i + 4];
                            }
                        } else {
                            vl = bl[s];
                            vr = br[s];
                        }
                        if constexpr (doL) {
                            const double xl = vl * (STREAM ? exlC : exl[STREAM ? 0 : s]);
#pragma unroll
                            for (int m = 0; m < MTF; m++)
                                YL[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(STREAM ? aC[m] : (AREG ? aU[AREG ? s : 0] : sU[aidx<KS>(m, s, lane)]), xl, YL[m], 0, 0, 0);
                            if (TAIL4) yl4 = __builtin_amdgcn_mfma_f64_4x4x4f64(AREG ? aU4[AREG ? s : 0] : sU4[s * 64 + lane], xl, yl4, 0, 0, 0);
                        }
                        if constexpr (doR) {
                            const double xr = vr * (STREAM ? exrC : exr[STREAM ? 0 : s]);
#pragma unroll
                            for (int m = 0; m < MTF; m++)
                                YR[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(STREAM ? aC[m] : (AREG ? aU[AREG ? s : 0] : sU[aidx<KS>(m, s, lane)]), xr, YR[m], 0, 0, 0);
                            if (TAIL4) yr4 = __builtin_amdgcn_mfma_f64_4x4x4f64(AREG ? aU4[AREG ? s : 0] : sU4[s * 64 + lane], xr, yr4, 0, 0, 0);
                        }
                        // 64 states: keep the k-steps in program order, or the scheduler hoists all 16 operand
                        // fetches above the first MFMA and the kernel spills
                        if constexpr (STREAM) {
                            if (s + 1 < KS) {
#pragma unroll
                                for (int m = 0; m < MTF; m++) aC[m] = aN[m];
                                exlC = exlN;
                                exrC = exrN;
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                };
                using T_ = std::true_type;
                using F_ = std::false_type;
                // T = (U ex_L left) .* (U ex_R right); an unknown state (gap) at a leaf child counts as exactly 1.0
                v4f64 T[MTF];
                double t4 = 0.0;
                auto hadamard = [&](v4f64 (&Y)[MTF], double y4, bool unkT, bool unkY) {  // T .*= Y with the selects
                    if (anyUnk) {
#pragma unroll
                        for (int m = 0; m < MTF; m++)
#pragma unroll
                            for (int r = 0; r < 4; r++) T[m][r] = (unkT ? 1.0 : T[m][r]) * (unkY ? 1.0 : Y[m][r]);
                        t4 = (unkT ? 1.0 : t4) * (unkY ? 1.0 : y4);
                    } else {
#pragma unroll
                        for (int m = 0; m < MTF; m++) T[m] = T[m] * Y[m];
                        t4 = t4 * y4;
                    }
                };
                // TAB: a leaf child's product is its pattern's row of the (L2-resident) table: requested before the
                // other child's MFMA chain, multiplied in after it
                auto table_row = [&](bool right, v4f64 (&Y)[MTF], double &y4) {
                    if constexpr (TABL) {
                        const LDS_AS double *r = as_lds((right ? rowR : rowL) + (size_t)(coff + c) * S * N);
#pragma unroll
                        for (int m = 0; m < MTF; m++) Y[m] = *reinterpret_cast<const LDS_AS v4f64 *>(r + 16 * m);
                        if (TAIL4) y4 = r[16 * MTF - 3 * g];   // (row carries +4g: tail row 16*MTF + g)
                    } else {
                        const double *r = (right ? rowR : rowL) + (size_t)(coff + c) * S * N;
#pragma unroll
                        for (int m = 0; m < MTF; m++) Y[m] = *reinterpret_cast<const v4f64 *>(r + 16 * m);
                        if (TAIL4) y4 = r[16 * MTF - 3 * g];   // (row carries +4g: tail row 16*MTF + g)
                    }
                };
                if (!TAB || (!leafL && !leafR)) {
                    v4f64 YR[MTF];
                    double yr4 = 0.0;
#pragma unroll
                    for (int m = 0; m < MTF; m++) { T[m] = (v4f64){0, 0, 0, 0}; YR[m] = (v4f64){0, 0, 0, 0}; }
                    chain(T_{}, T_{}, T, YR, t4, yr4);
                    hadamard(YR, yr4, unkL, unkR);
                } else if (leafL && leafR) {
                    v4f64 YR[MTF];
                    double yr4 = 0.0;
                    table_row(false, T, t4);
                    table_row(true, YR, yr4);
                    chain(F_{}, F_{}, T, YR, t4, yr4);   // (only refills the prefetch registers)
                    hadamard(YR, yr4, unkL, unkR);
                } else if (leafL) {
                    v4f64 YR[MTF];
                    double yr4 = 0.0;
                    table_row(false, T, t4);
#pragma unroll
                    for (int m = 0; m < MTF; m++) YR[m] = (v4f64){0, 0, 0, 0};
                    chain(F_{}, T_{}, YR, YR, yr4, yr4);
                    hadamard(YR, yr4, unkL, false);
                } else {
                    v4f64 YL[MTF];
                    double yl4 = 0.0;
                    table_row(true, T, t4);
#pragma unroll
                    for (int m = 0; m < MTF; m++) YL[m] = (v4f64){0, 0, 0, 0};
                    chain(T_{}, F_{}, YL, YL, yl4, yl4);
                    hadamard(YL, yl4, unkR, false);
                }
                if (c + 1 == C && g == 0) pfn_sc = nd.pf_sc[nreal ? ptn : (int64_t)p];
                TRACE_STAMP();   // products + Hadamard done
                v4f64 O[MTF];
                double o4 = 0.0;
#pragma unroll
                for (int m = 0; m < MTF; m++) O[m] = (v4f64){0, 0, 0, 0};
#pragma unroll
                for (int s = 0; s < KS; s++) {
                    // accumulator layout == B layout of k-step s (tail: the 4x4x4 result == k-step 4*MTF)
                    const double bt = (s < 4 * MTF) ? T[(s >> 2) < MTF ? (s >> 2) : 0][s & 3] : t4;
#pragma unroll
                    for (int m = 0; m < MTF; m++)
                        O[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(AREG ? aUi[AREG ? s : 0] : sUi[aidx<KS>(m, s, lane)], bt, O[m], 0, 0, 0);
                    if (TAIL4) o4 = __builtin_amdgcn_mfma_f64_4x4x4f64(AREG ? aUi4[AREG ? s : 0] : sUi4[s * 64 + lane], bt, o4, 0, 0, 0);
                }
#pragma unroll
                for (int m = 0; m < MTF; m++) {
                    prev[c][m] = O[m];
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int row = 16 * m + 4 * r + g;
#ifndef IQHIP_MFMA_ABLATE_NOSTORE  // timing-only build switch; never defined in the shipped library
                        // (streaming hint: most results are not read again before they leave the L2; protein -1 %, codon -2 %)
                        __builtin_nontemporal_store(O[m][r], &dst[(size_t)((coff + c) * N + row) * 16 + p]);
#endif
                        if (HOLDS && push) hold[((coff + c) * N + 16 * m + 4 * r) * 16] = O[m][r];
                        lmax = amax_hi(lmax, O[m][r]);
                    }
                }
                if (TAIL4) {
                    prevT[c] = o4;
#ifndef IQHIP_MFMA_ABLATE_NOSTORE
                    __builtin_nontemporal_store(o4, &dst[(size_t)((coff + c) * N + 16 * MTF + g) * 16 + p]);
#endif
                    if (HOLDS && push) hold[((coff + c) * N + 16 * MTF) * 16] = o4;
                    lmax = amax_hi(lmax, o4);
                }
                TRACE_STAMP();   // contraction done, stores issued
            }
            }   // (!from_table)
            if (no_scale == 2 && lmax == 0) {   // scalar rule: exactly zero, or only below 2^-1042?
                unsigned nz = 0;
#pragma unroll
                for (int c = 0; c < C; c++) {
#pragma unroll
                    for (int m = 0; m < MTF; m++)
#pragma unroll
                        for (int r = 0; r < 4; r++) nz |= nonzero_bits(prev[c][m][r]);
                    if (TAIL4) nz |= nonzero_bits(prevT[c]);
                }
                if (nz) lmax = 1;
            }
            lmax = group_max_u(lmax);
            if constexpr (CS > 1) {
                // maximum over the categories held by the other waves of this tile (LDS + one workgroup barrier per op; a
                // barrier-free exchange -- maxima posted with tags, a wave waits for its partners only when it has a candidate
                // pattern -- was measured equal, 199 vs 197 us for the top stage, and is not worth its moving parts)
                __shared__ unsigned s_lmax[2][WG / 64][16];
                const int par = k & 1;
                if (g == 0) s_lmax[par][wave][p] = lmax;
                __syncthreads();
                const int w0 = (wave / CS) * CS;
#pragma unroll
                for (int q = 0; q < CS; q++) lmax = max(lmax, s_lmax[par][w0 + q][p]);
            }
#if defined(IQHIP_MFMA_ABLATE_NOLOAD) || defined(IQHIP_MFMA_ABLATE_NOSTORE)
            const bool zero = false;
            const bool do_scale = lmax == 0xffffffffu;  // (garbage inputs must not take the rescaling path in a timing build)
#else
            const bool zero = no_scale == 2 && !(leafL && leafR) && lmax == 0;   // the scalar kernel's `lh_max == 0.0`, phylotreesse.cpp:777-788
            const bool do_scale = zero || (!(leafL && leafR) && (lmax < kScalingThresholdHi) && (invar == 0.0) && no_scale != 1);
#endif
            double my_scale = 0.0;
            if (__any(do_scale)) {
                if (do_scale) {
#pragma unroll
                    for (int c = 0; c < C; c++) {
                        const double *tu = A.tipc + ((size_t)A.state_unknown * CT + coff + c) * N;   // unknown tip, this category
#pragma unroll
                        for (int m = 0; m < MTF; m++)
#pragma unroll
                            for (int r = 0; r < 4; r++) {
                                if (zero) prev[c][m][r] = tu[16 * m + 4 * r + g]; else
                                prev[c][m][r] *= kScalingThresholdInv;
                                dst[(size_t)((coff + c) * N + 16 * m + 4 * r + g) * 16 + p] = prev[c][m][r];
                                if (HOLDS && push) hold[((coff + c) * N + 16 * m + 4 * r) * 16] = prev[c][m][r];
                            }
                        if (TAIL4) {
                            if (zero) prevT[c] = tu[16 * MTF + g]; else
                            prevT[c] *= kScalingThresholdInv;
                            dst[(size_t)((coff + c) * N + 16 * MTF + g) * 16 + p] = prevT[c];
                            if (HOLDS && push) hold[((coff + c) * N + 16 * MTF) * 16] = prevT[c];
                        }
                    }
                    sc += zero ? 4 : 1;
                    if (lead && g == 0 && ptn < A.nptn) my_scale = (zero ? 4.0 : 1.0) * (kLogScalingThreshold * freq);
                }
            }
            prev_sc = sc;
            if (HOLDS && push) hold_sc = sc;
            if (lead && g == 0) dst_sc[ptn] = (int16_t)sc;
            const double ws = __any(my_scale != 0.0) ? wave_sum_m(my_scale) : 0.0;  // (no rescaling in this op: nothing to add)
            if (lead && lane == 0) {
                A.slab[(size_t)(2 + out_row) * A.nwaves + (int)tl] = ws;
                if (ws != 0.0) __hip_atomic_fetch_or(&A.fold_flags[2 + out_row], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    TRACE_END();
}

// (the two-waves-per-tile form of 20 states fits three waves per SIMD: 173 -> 168 registers without spills)
template <int N, int C, int WG, int CS = 1, bool TAB = false>
__global__ __launch_bounds__(WG, (N == 20 && C == 2 && CS == 2) ? 3 : 2) void k_traverse_mfma2(const TravMArgs A) {
    trav_mfma2_body<N, C, WG, CS, TAB>(A, (int)blockIdx.x);
}

// Mixed-role top stage of 20 states x 4 categories (IQHIP_MIXED_TOP=0 switches it off): whole rounds of the chip as two waves
// per tile; the few tiles beyond the last whole round (53 of 3125 at 50 000 patterns: a third round on a nearly empty chip, 15 %
// of the top stage in the wave trace) as one wave per category, dispatched LAST, so that the tail is a quarter chain per wave:
// top stage 200 -> 187 us on one box (tools/top_stage_ab.sh).  Dispatched FIRST the same workgroups push 53 two-wave
// workgroups into a third round: 220 us.  Same per-pattern arithmetic in both roles: same bits.
__global__ __launch_bounds__(256, 3) void k_traverse_mfma_top20(const TravMArgs F, const int nfull, const TravMArgs R, const int64_t tile0R) {
    if ((int)blockIdx.x < nfull) trav_mfma2_body<20, 2, 256, 2, false>(F, (int)blockIdx.x);
    else trav_mfma2_body<20, 1, 256, 4, false>(R, (int)blockIdx.x - nfull, tile0R);
}

static hipError_t launch_trav_top20(iqhip_engine *e, TravMArgs &A, int nfull_wg) {
    const int nx = e->state_unknown + 1 - 20;
    const size_t lds = (size_t)(mfma2_fixed_lds_doubles(20) + nx * 20 + e->plan_lds_doubles) * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_traverse_mfma_top20), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        attr_set = true;
    }
    A.hold_off = -1;
    if (e->plan_small && A.nsegs_launch == 1) {
        A.small_plan = 1;
        A.small_segs[0] = 0;
        A.small_segs[1] = e->plan_small_nops;
        for (int q = 0; q < kSmallPlanOps; q++) A.small_ops[q] = e->h_ops[q];
    }
    TravMArgs F = A, R = A;
    F.ntiles = (int64_t)nfull_wg * 2;
    F.ngroups = nfull_wg;
    const int nrest = (int)(A.ntiles - F.ntiles);
    R.ngroups = nrest;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_traverse_mfma_top20, dim3((unsigned)(nrest + nfull_wg)), dim3(256), lds, e->stream, F, nfull_wg, R, F.ntiles);
    return hipGetLastError();
}

template <int N, bool MIX>
static hipError_t launch_trav_m(iqhip_engine *e, TravMArgs &A) {
    constexpr int MT = (N + 15) / 16, KS = N / 4, WG = 256;
    const int nx = e->state_unknown + 1 - N;
    const size_t lds = (size_t)(2 * MT * KS * 64 + nx * N + e->plan_lds_doubles) * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_traverse_mfma<N, WG, MIX>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    A.ngroups = (int)((A.ntiles + 3) / 4);
    const int grid = A.ngroups * A.nsegs_launch;
    hipLaunchKernelGGL((k_traverse_mfma<N, WG, MIX>), dim3(grid), dim3(WG), lds, e->stream, A);
    return hipGetLastError();
}


// ---------------------------------------------------------------------------------------
// Mixture models (phylokernelmixture.h:20-460, phylokernelmixrate.h:22-450), 20 states.  The block of a
// pattern is C = (class, rate) components x 20 doubles -- too large for the register-resident scheme of
// k_traverse_mfma2 -- so both children are streamed from memory, one component ahead of the one being
// multiplied.  The A fragments (U, U^-1 of the component's class: 16-row tile + 4-row tail, 20 doubles)
// are re-read from the per-class images only when the class changes (block order [class][rate]).
// ---------------------------------------------------------------------------------------
// CS > 1: the CS waves of a workgroup share one tile and split its components (C % CS == 0), as k_traverse_mfma2's
// category split does; the scaling maximum crosses the waves through LDS.
template <int WG, int CS>
__global__ __launch_bounds__(WG, 2) void k_traverse_mfma_mix20(const TravMArgs A) {
    constexpr int N = 20, KS = 5, WPB = WG / 64;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *sReg = smem;  // per (op, child) exponentials [C*N] of the chunk
    const int C = A.ncat;
    const int B = C * N;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int seg = (int)blockIdx.x / A.ngroups;  // scalar
    const int k_begin = as_const(A.segs)[2 * seg], k_end = k_begin + as_const(A.segs)[2 * seg + 1];
    const int64_t tile = (int64_t)((int)blockIdx.x - seg * A.ngroups) * (WPB / CS) + wave / CS;
    const int c_lo = (wave % CS) * (C / CS), c_hi = c_lo + C / CS;  // this wave's components
    const bool lead = (wave % CS) == 0;
    const bool active = tile < A.ntiles;
    const int64_t tl = active ? tile : 0;
    const int p = lane & 15, g = lane >> 4;
    const int64_t ptn = tl * 16 + p;
    const size_t tbase = (size_t)tl * 16 * B;  // doubles
    const double freq = A.freq[ptn];
    const double invar = A.invar[ptn];
    const CONST_AS DevOp *ops = as_const(A.ops);
    const CONST_AS int *cls = as_const(A.cls);

    double aU[KS], aU4[KS], aUi[KS], aUi4[KS];
    int cur_class = -1;

    int k = k_begin;
    while (k < k_end) {
        const int kn = ops[k].chunk_nops;
        __syncthreads();
        for (int t = threadIdx.x; t < kn * 2 * B; t += WG) {
            const int o = t / (2 * B), r = t - o * (2 * B), child = r / B, e = r - child * B;
            const CONST_AS DevOp &d = ops[k + o];
            const double len = op_child_len(d, child);
            sReg[(child ? d.lds_right : d.lds_left) + e] = exp(A.evalc[e] * (A.rates[e / N] * len));
        }
        __syncthreads();
        if (!active) { k += kn; continue; }

        for (int kk = 0; kk < kn; kk++, k++) {
            const CONST_AS DevOp &op = ops[k];
            const bool leafL = op.left_kind == CHILD_LEAF, leafR = op.right_kind == CHILD_LEAF;
            const double *exL = sReg + op.lds_left, *exR = sReg + op.lds_right;
            int sc = 0, sL = 0, sR = 0;
            if (leafL) sL = op.sl[ptn]; else if (g == 0) sc += op.pf_sc[ptn];
            if (leafR) sR = op.sr[ptn]; else if (g == 0) sc += op.ld_sc[ptn];
            const bool unkL = leafL && sL == A.state_unknown, unkR = leafR && sR == A.state_unknown;
            // component c of a child: 5 k-step slices -- a vector in memory, or this pattern's tip vector
            const double *srcL = leafL ? A.tipc + (size_t)sL * B + g : op.pf + tbase + lane;
            const double *srcR = leafR ? A.tipc + (size_t)sR * B + g : op.ld + tbase + lane;
            const int strideL = leafL ? N : N * 16, stepL = leafL ? 4 : 64;  // per component / per k-step
            const int strideR = leafR ? N : N * 16, stepR = leafR ? 4 : 64;
            double *dst = op.dst + tbase;
            unsigned lmax = 0, nz = 0;
            double nl[KS], nr[KS];
#pragma unroll
            for (int s = 0; s < KS; s++) { nl[s] = srcL[(size_t)c_lo * strideL + s * stepL]; nr[s] = srcR[(size_t)c_lo * strideR + s * stepR]; }
            for (int c = c_lo; c < c_hi; c++) {
                double bl[KS], br[KS];
#pragma unroll
                for (int s = 0; s < KS; s++) { bl[s] = nl[s]; br[s] = nr[s]; }
                {   // request component c+1 (the last request of an op re-reads component C-1: harmless)
                    const int cn = (c + 1 < c_hi) ? c + 1 : c;
#pragma unroll
                    for (int s = 0; s < KS; s++) {
                        nl[s] = srcL[(size_t)cn * strideL + s * stepL];
                        nr[s] = srcR[(size_t)cn * strideR + s * stepR];
                    }
                }
                const int m = cls[c];
                if (m != cur_class) {  // wave-uniform
                    const double *im = A.img + (size_t)m * 4 * KS * 64 + lane;
#pragma unroll
                    for (int s = 0; s < KS; s++) {
                        aU[s] = im[s * 64];
                        aU4[s] = im[(KS + s) * 64];
                        aUi[s] = im[(2 * KS + s) * 64];
                        aUi4[s] = im[(3 * KS + s) * 64];
                    }
                    cur_class = m;
                }
                v4f64 YL = {0, 0, 0, 0}, YR = {0, 0, 0, 0};
                double yl4 = 0.0, yr4 = 0.0;
#pragma unroll
                for (int s = 0; s < KS; s++) {
                    const int i = 4 * s + g;
                    const double xl = bl[s] * exL[c * N + i];
                    const double xr = br[s] * exR[c * N + i];
                    YL = __builtin_amdgcn_mfma_f64_16x16x4f64(aU[s], xl, YL, 0, 0, 0);
                    YR = __builtin_amdgcn_mfma_f64_16x16x4f64(aU[s], xr, YR, 0, 0, 0);
                    yl4 = __builtin_amdgcn_mfma_f64_4x4x4f64(aU4[s], xl, yl4, 0, 0, 0);
                    yr4 = __builtin_amdgcn_mfma_f64_4x4x4f64(aU4[s], xr, yr4, 0, 0, 0);
                }
                double T[KS];
#pragma unroll
                for (int r = 0; r < 4; r++) T[r] = (unkL ? 1.0 : YL[r]) * (unkR ? 1.0 : YR[r]);
                T[4] = (unkL ? 1.0 : yl4) * (unkR ? 1.0 : yr4);
                v4f64 O = {0, 0, 0, 0};
                double o4 = 0.0;
#pragma unroll
                for (int s = 0; s < KS; s++) {
                    O = __builtin_amdgcn_mfma_f64_16x16x4f64(aUi[s], T[s], O, 0, 0, 0);
                    o4 = __builtin_amdgcn_mfma_f64_4x4x4f64(aUi4[s], T[s], o4, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    dst[(size_t)(c * N + 4 * r + g) * 16 + p] = O[r];
                    lmax = amax_hi(lmax, O[r]);
                }
                dst[(size_t)(c * N + 16 + g) * 16 + p] = o4;
                lmax = amax_hi(lmax, o4);
                if (op.no_scale == 2) nz |= nonzero_bits(O[0]) | nonzero_bits(O[1]) | nonzero_bits(O[2]) | nonzero_bits(O[3]) | nonzero_bits(o4);
            }
            if (nz != 0 && lmax == 0) lmax = 1;   // (scalar rule: only an exact zero takes its lh_max == 0 branch)
            lmax = group_max_u(lmax);
            if constexpr (CS > 1) {
                __shared__ unsigned s_lmax[2][WG / 64][16];
                const int par = k & 1;
                if (g == 0) s_lmax[par][wave][p] = lmax;
                __syncthreads();
                const int w0 = (wave / CS) * CS;
#pragma unroll
                for (int q = 0; q < CS; q++) lmax = max(lmax, s_lmax[par][w0 + q][p]);
            }
            const int rule = op.no_scale;
            const bool zero = rule == 2 && !(leafL && leafR) && lmax == 0;   // the scalar kernel's `lh_max == 0.0`, phylotreesse.cpp:777-788
            const bool do_scale = zero || (!(leafL && leafR) && (lmax < kScalingThresholdHi) && (invar == 0.0) && rule != 1);
            double my_scale = 0.0;
            if (__any(do_scale)) {
                if (do_scale) {
                    if (zero) for (int e = c_lo * N + g; e < c_hi * N; e += 4) dst[(size_t)e * 16 + p] = A.tipc[(size_t)A.state_unknown * B + e];
                    else for (int e = c_lo * N + g; e < c_hi * N; e += 4) dst[(size_t)e * 16 + p] *= kScalingThresholdInv;
                    sc += zero ? 4 : 1;
                    if (lead && g == 0 && ptn < A.nptn) my_scale = (zero ? 4.0 : 1.0) * (kLogScalingThreshold * freq);
                }
            }
            if (lead && g == 0) op.dst_sc[ptn] = (int16_t)sc;
            const double ws = __any(my_scale != 0.0) ? wave_sum_m(my_scale) : 0.0;  // (no rescaling in this op: nothing to add)
            if (lead && lane == 0) {
                A.slab[(size_t)(2 + op.out_row) * A.nwaves + (int)tl] = ws;
                if (ws != 0.0) __hip_atomic_fetch_or(&A.fold_flags[2 + op.out_row], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

template <int CS>
static hipError_t launch_trav_mix20_cs(iqhip_engine *e, TravMArgs &A) {
    constexpr int WG = 256;
    const size_t lds = (size_t)e->plan_lds_doubles * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_traverse_mfma_mix20<WG, CS>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        attr_set = true;
    }
    A.ngroups = (int)((A.ntiles * CS + 3) / 4);
    (void)hipGetLastError();
    hipLaunchKernelGGL((k_traverse_mfma_mix20<WG, CS>), dim3((unsigned)(A.ngroups * A.nsegs_launch)), dim3(WG), lds, e->stream, A);
    return hipGetLastError();
}

static hipError_t launch_trav_mix20(iqhip_engine *e, TravMArgs &A) {
    // component split while the alignment is small (at most 3/4 tile per SIMD: 10k patterns x 40 components 1.96 ->
    // 1.77 ms, but 30k patterns x 8 components 1.78 -> 2.62 ms) and the components divide by 4
    bool split = (e->ncat % 4 == 0) && 4 * e->ntiles <= 3 * (int64_t)e->num_cus * 4;
    if (const char *cs = getenv("IQHIP_CAT_SPLIT")) split = (atoi(cs) != 0) && (e->ncat % 4 == 0);
    return split ? launch_trav_mix20_cs<4>(e, A) : launch_trav_mix20_cs<1>(e, A);
}

// ---------------------------------------------------------------------------------------
// 64 states, small alignments: ROW SPLIT.  The four waves of a workgroup share one tile; wave w owns the output
// rows [16w, 16w+16) of every product (one M-tile), i.e. 48 instead of 192 dependent MFMAs per op.  Its 2 x 16 A
// fragments (U and U^-1 rows) live in registers.  What the other waves need of a result -- the Hadamard product
// T before the U^-1 contraction, and the previous result when it is the next op's operand -- is exchanged through
// LDS as B-operand k-step slices (the accumulator image of M-tile w IS k-steps 4w..4w+3), one workgroup barrier
// each; the scaling maximum takes a third.  Same canonical plan form as k_traverse_mfma2 (C = 1).
// ---------------------------------------------------------------------------------------
// tile0: first tile of this role (the mixed top-stage kernel gives the row-split role the tiles behind the full ones)
template <int WG, bool TAB>
__device__ __forceinline__ void trav_rows64_body(const TravMArgs &A, const int vblock, const int64_t tile0) {
    constexpr int N = 64, KS = 16, B = 64;
    static_assert(WG == 256, "one tile per workgroup of four waves");
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int nx = A.state_unknown + 1 - N;
    double *sTip = smem;                       // [N + nx][N] tip_partial_lh rows
    double *sX = sTip + (N + nx) * N;          // [2 parities][KS][64] previous result as k-step slices
    double *sT = sX + 2 * KS * 64;             // [2 parities][KS][64] Hadamard product
    double *sReg = sT + 2 * KS * 64;           // per (op, child) exponentials [N] of the chunk
    __shared__ unsigned s_lmax[2][4][16];
    for (int t = threadIdx.x; t < (N + nx) * N; t += WG) sTip[t] = A.tip[t];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int seg = vblock / A.ngroups;  // scalar
    const int k_begin = as_const(A.segs)[2 * seg], k_end = k_begin + as_const(A.segs)[2 * seg + 1];
    const int64_t tile = tile0 + (int64_t)(vblock - seg * A.ngroups);
    const bool active = tile < A.ntiles;          // uniform over the workgroup
    const int64_t tl = active ? tile : 0;
    const int p = lane & 15, g = lane >> 4;
    const int64_t ptn = tl * 16 + p;
    const size_t tbase = (size_t)tl * 16 * B;
    const double freq = A.freq[ptn];
    const double invar = A.invar[ptn];
    const CONST_AS DevOp *ops = as_const(A.ops);
    const bool lead = wave == 0;
    TRACE_DECL;
    TRACE_BEGIN(vblock, wave, 1064);

    // A fragments of this wave's 16 rows: lane (row = 16*wave + (lane & 15), k = 4s + (lane >> 4))
    double aU[KS], aUi[KS];
#pragma unroll
    for (int s = 0; s < KS; s++) {   // = M-tile `wave` of the engine's fragment image (coalesced)
        aU[s] = A.aimg[(wave * KS + s) * 64 + lane];
        aUi[s] = A.aimg[((4 + wave) * KS + s) * 64 + lane];
    }
    v4f64 prev = {0, 0, 0, 0};   // rows 16*wave + 4r + g of the previous result
    int prev_sc = 0;
    double PFn[KS];
    int pfn_sc = 0;
    {
        const CONST_AS DevOp &f = ops[k_begin];
        const double *src = f.pf + ((f.real_mask & 1) ? tbase : 0);
#pragma unroll
        for (int s = 0; s < KS; s++) PFn[s] = src[s * 64 + lane];
        if (g == 0) pfn_sc = f.pf_sc[(f.real_mask & 1) ? ptn : (int64_t)p];
    }

    int k = k_begin;
    while (k < k_end) {
        const int kn = ops[k].chunk_nops;
        __syncthreads();
        for (int t = threadIdx.x; t < kn * 2 * B; t += WG) {
            const int o = t / (2 * B), r = t - o * (2 * B), child = r / B, e = r - child * B;
            const CONST_AS DevOp &d = ops[k + o];
            const double len = op_child_len(d, child);
            sReg[(child ? d.lds_right : d.lds_left) + e] = exp(A.eval[e] * (A.rates[0] * len));
        }
        __syncthreads();
        if (!active) { k += kn; continue; }

        for (int kk = 0; kk < kn; kk++, k++) {
            const CONST_AS DevOp &op = ops[k];
            const CONST_AS DevOp &nxop = ops[k + 1];
            const int par = k & 1;
            const bool leafL = op.left_kind == CHILD_LEAF, leafR = op.right_kind == CHILD_LEAF;
            const double *exL = sReg + op.lds_left, *exR = sReg + op.lds_right;
            int sc = 0, sL = 0, sR = 0;
            if (leafL) sL = op.sl[ptn]; else sc += pfn_sc;
            if (leafR) sR = op.sr[ptn];
            double xr[KS];
            if (op.right_kind == CHILD_LOAD) {
                const double *src = op.ld + tbase;
#pragma unroll
                for (int s = 0; s < KS; s++) xr[s] = src[s * 64 + lane];
                if (g == 0) prev_sc = op.ld_sc[ptn];
            } else if (!leafR) {  // CHILD_PREV: gather the four waves' row blocks
                double *xb = sX + par * KS * 64;
#pragma unroll
                for (int r = 0; r < 4; r++) xb[(4 * wave + r) * 64 + lane] = prev[r];
                __syncthreads();
#pragma unroll
                for (int s = 0; s < KS; s++) xr[s] = xb[s * 64 + lane];
            } else {
#pragma unroll
                for (int s = 0; s < KS; s++) xr[s] = sTip[sR * N + 4 * s + g];
            }
            if (!leafR) sc += prev_sc;
            const bool unkL = leafL && sL == A.state_unknown, unkR = leafR && sR == A.state_unknown;
            // streamed child of the next op, requested while this one computes
            const bool nreal = nxop.real_mask & 1;
            const double *nsrc = nxop.pf + (nreal ? tbase : 0);
            v4f64 YL = {0, 0, 0, 0}, YR = {0, 0, 0, 0};
            const bool tabL = TAB && leafL, tabR = TAB && leafR;
            if (TAB) {  // leaf children: this wave's 16 rows of the pattern's table row (k_leaf_tables)
                const int S = A.state_unknown;
                if (leafL) YL = *reinterpret_cast<const v4f64 *>(op.tabL + (size_t)(sL < S ? sL : 0) * N + 16 * wave + 4 * g);
                if (leafR) YR = *reinterpret_cast<const v4f64 *>(op.tabR + (size_t)(sR < S ? sR : 0) * N + 16 * wave + 4 * g);
            }
#pragma unroll
            for (int s = 0; s < KS; s++) {
                const int i = 4 * s + g;
                const double vl = leafL ? sTip[sL * N + i] : PFn[s];
                PFn[s] = nsrc[s * 64 + lane];
                if (!tabL) {
                    const double xl = vl * exL[i];
                    YL = __builtin_amdgcn_mfma_f64_16x16x4f64(aU[s], xl, YL, 0, 0, 0);
                }
                if (!tabR) {
                    const double xrs = xr[s] * exR[i];
                    YR = __builtin_amdgcn_mfma_f64_16x16x4f64(aU[s], xrs, YR, 0, 0, 0);
                }
            }
            if (g == 0) pfn_sc = nxop.pf_sc[nreal ? ptn : (int64_t)p];
            double *tb = sT + par * KS * 64;
#pragma unroll
            for (int r = 0; r < 4; r++)
                tb[(4 * wave + r) * 64 + lane] = (unkL ? 1.0 : YL[r]) * (unkR ? 1.0 : YR[r]);
            __syncthreads();
            v4f64 O = {0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < KS; s++) O = __builtin_amdgcn_mfma_f64_16x16x4f64(aUi[s], tb[s * 64 + lane], O, 0, 0, 0);
            double *dst = op.dst + tbase;
            unsigned lmax = 0;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                __builtin_nontemporal_store(O[r], &dst[(size_t)(16 * wave + 4 * r + g) * 16 + p]);
                lmax = amax_hi(lmax, O[r]);
            }
            prev = O;
            const int rule = op.no_scale;
            if (rule == 2 && lmax == 0 && (nonzero_bits(O[0]) | nonzero_bits(O[1]) | nonzero_bits(O[2]) | nonzero_bits(O[3])) != 0)
                lmax = 1;   // (scalar rule: only an exact zero takes its lh_max == 0 branch)
            lmax = group_max_u(lmax);
            if (g == 0) s_lmax[par][wave][p] = lmax;
            __syncthreads();
            lmax = max(max(s_lmax[par][0][p], s_lmax[par][1][p]), max(s_lmax[par][2][p], s_lmax[par][3][p]));
            const bool zero = rule == 2 && !(leafL && leafR) && lmax == 0;   // the scalar kernel's `lh_max == 0.0`, phylotreesse.cpp:777-788
            const bool do_scale = zero || (!(leafL && leafR) && (lmax < kScalingThresholdHi) && (invar == 0.0) && rule != 1);
            double my_scale = 0.0;
            if (__any(do_scale)) {
                if (do_scale) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        if (zero) prev[r] = A.tipc[(size_t)A.state_unknown * 64 + 16 * wave + 4 * r + g]; else
                        prev[r] *= kScalingThresholdInv;
                        dst[(size_t)(16 * wave + 4 * r + g) * 16 + p] = prev[r];
                    }
                    sc += zero ? 4 : 1;
                    if (lead && g == 0 && ptn < A.nptn) my_scale = (zero ? 4.0 : 1.0) * (kLogScalingThreshold * freq);
                }
            }
            prev_sc = sc;
            if (lead && g == 0) op.dst_sc[ptn] = (int16_t)sc;
            const double ws = __any(my_scale != 0.0) ? wave_sum_m(my_scale) : 0.0;  // (no rescaling in this op: nothing to add)
            if (lead && lane == 0) {
                A.slab[(size_t)(2 + op.out_row) * A.nwaves + (int)tl] = ws;
                if (ws != 0.0) __hip_atomic_fetch_or(&A.fold_flags[2 + op.out_row], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    TRACE_END();
}

template <int WG, bool TAB>
__global__ __launch_bounds__(WG, 2) void k_traverse_mfma_rows64(const TravMArgs A) {
    trav_rows64_body<WG, TAB>(A, (int)blockIdx.x, 0);
}

// ---------------------------------------------------------------------------------------
// 64 states, the sequential top stage of a staged plan, MIXED roles.  A tile's op list is one wave's dependent chain,
// so an alignment with slightly more tiles than the chip has SIMDs (20 000 codon patterns = 1250 tiles on 1024 SIMDs)
// leaves most SIMDs with one chain and a fifth of them with two -- the launch takes as long as two chains although
// the chip is 61 % loaded.  Here the first `nfull` workgroups walk four tiles each as k_traverse_mfma2 does (one chain
// per SIMD when nfull = number of CUs) and every tile beyond those gets a workgroup of its own whose four waves own 16
// of the 64 output rows each (k_traverse_mfma_rows64): a quarter chain per SIMD, placed beside the full chains.
// ---------------------------------------------------------------------------------------
template <bool TAB>
__global__ __launch_bounds__(256, 2) void k_traverse_mfma_top64(const TravMArgs A, const int nfull, const TravMArgs R) {
    if ((int)blockIdx.x < nfull) trav_mfma2_body<64, 1, 256, 1, TAB>(A, (int)blockIdx.x);
    else trav_rows64_body<256, TAB>(R, (int)blockIdx.x - nfull, (int64_t)nfull * 4);
}

template <bool TAB>
static hipError_t launch_trav_rows64(iqhip_engine *e, TravMArgs &A) {
    constexpr int WG = 256;
    const int nx = e->state_unknown + 1 - 64;
    const size_t lds = (size_t)((64 + nx) * 64 + 4 * 16 * 64 + e->plan_lds_doubles) * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_traverse_mfma_rows64<WG, TAB>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        attr_set = true;
    }
    A.ngroups = (int)A.ntiles;
    (void)hipGetLastError();
    hipLaunchKernelGGL((k_traverse_mfma_rows64<WG, TAB>), dim3((unsigned)(A.ngroups * A.nsegs_launch)), dim3(WG), lds, e->stream, A);
    return hipGetLastError();
}

// LDS doubles of k_traverse_mfma2 that do not depend on the plan (A images, tail images, U^-1 transposed)
int mfma2_fixed_lds_doubles(int n) {
    const int mtf = n / 16, ks = n / 4;
    const int images = n < 64 ? 0 : 2 * mtf * ks * 64 + ((n % 16) == 4 ? 2 * ks * 64 : 0);   // (20 states: fragments in registers)
    return images + (n * n * 8 <= 4096 ? n * n : 0);
}

template <int N, int C, int CS = 1, bool TAB = false>
static hipError_t launch_trav_m2(iqhip_engine *e, TravMArgs &A) {
    constexpr int KS = N / 4, WG = 256;
    const int nx = e->state_unknown + 1 - N;
    size_t lds = (size_t)(mfma2_fixed_lds_doubles(N) + nx * N + e->plan_lds_doubles) * sizeof(double);
    A.hold_off = -1;
    if (N < 64 && CS == 1 && e->plan_nhold > 0) {   // parking places: one tile vector (16 patterns x block) per tile
        A.hold_off = (int)(lds / sizeof(double));
        lds += (size_t)(WG / 64 / CS) * 16 * e->block * sizeof(double);
    }
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_traverse_mfma2<N, C, WG, CS, TAB>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        attr_set = true;
    }
    A.ngroups = (int)((A.ntiles * CS + 3) / 4);
    const int grid = A.ngroups * A.nsegs_launch;
    if (e->plan_small && A.nsegs_launch == 1) {   // (the plan was not copied to d_ops: it travels with the launch)
        A.small_plan = 1;
        A.small_segs[0] = 0;
        A.small_segs[1] = e->plan_small_nops;
        for (int q = 0; q < kSmallPlanOps; q++) A.small_ops[q] = e->h_ops[q];
    }
    (void)hipGetLastError();
    hipLaunchKernelGGL((k_traverse_mfma2<N, C, WG, CS, TAB>), dim3(grid), dim3(WG), lds, e->stream, A);
    return hipGetLastError();
}

// the mixed-role top stage (k_traverse_mfma_top64): full-chain workgroups for whole rounds of the chip, row-split
// workgroups for the tiles that are left over
template <bool TAB>
static hipError_t launch_trav_top64(iqhip_engine *e, TravMArgs &A, int nfull) {
    const int nx = e->state_unknown + 1 - 64;
    const size_t fixed = std::max<size_t>((size_t)mfma2_fixed_lds_doubles(64) + (size_t)nx * 64, (size_t)(64 + nx) * 64 + 4 * 16 * 64);
    const size_t lds = (fixed + e->plan_lds_doubles) * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_traverse_mfma_top64<TAB>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        attr_set = true;
    }
    TravMArgs F = A, R = A;
    F.ntiles = (int64_t)nfull * 4;
    F.ngroups = nfull;
    R.ngroups = (int)(A.ntiles - F.ntiles);
    (void)hipGetLastError();
    hipLaunchKernelGGL((k_traverse_mfma_top64<TAB>), dim3((unsigned)(nfull + R.ngroups)), dim3(256), lds, e->stream, F, nfull, R);
    return hipGetLastError();
}

#ifdef IQHIP_WAVE_TRACE
static hipError_t launch_traverse_mfma_impl(iqhip_engine *e, const int *seg_table, int nsegs, int nwaves, bool top_stage);
// timing-study build: drain the stream after every traversal launch and append the wave records of launches
// [IQHIP_TRACE_FIRST, IQHIP_TRACE_FIRST + IQHIP_TRACE_COUNT) to $IQHIP_TRACE_FILE
hipError_t launch_traverse_mfma(iqhip_engine *e, const int *seg_table, int nsegs, int nwaves, bool top_stage) {
    static int launch_no = 0;
    static const int first = getenv("IQHIP_TRACE_FIRST") ? atoi(getenv("IQHIP_TRACE_FIRST")) : 40;
    static const int count = getenv("IQHIP_TRACE_COUNT") ? atoi(getenv("IQHIP_TRACE_COUNT")) : 4;
    const unsigned zero = 0;
    (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_wrec_n), &zero, sizeof(zero), 0, hipMemcpyHostToDevice, e->stream);
    (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_det_n), &zero, sizeof(zero), 0, hipMemcpyHostToDevice, e->stream);
    hipError_t rc = launch_traverse_mfma_impl(e, seg_table, nsegs, nwaves, top_stage);
    if (rc != hipSuccess) return rc;
    rc = hipStreamSynchronize(e->stream);
    if (rc != hipSuccess) return rc;
    const int no = launch_no++;
    const char *path = getenv("IQHIP_TRACE_FILE");
    if (path && no >= first && no < first + count) {
        unsigned nw = 0, nd = 0;
        (void)hipMemcpyFromSymbol(&nw, HIP_SYMBOL(g_wrec_n), sizeof(nw));
        (void)hipMemcpyFromSymbol(&nd, HIP_SYMBOL(g_det_n), sizeof(nd));
        nw = nw < TRACE_MAXW ? nw : TRACE_MAXW;
        nd = nd < TRACE_NDET ? nd : TRACE_NDET;
        std::vector<WaveRec> recs(nw);
        if (nw) (void)hipMemcpyFromSymbol(recs.data(), HIP_SYMBOL(g_wrec), sizeof(WaveRec) * nw);
        std::vector<unsigned long long> st((size_t)TRACE_NDET * TRACE_MAXSTAMP);
        (void)hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * st.size());
        FILE *f = fopen(path, "a");
        if (f) {
            fprintf(f, "L %d top %d nsegs %d nwaves %d n %d ncat %d\n", no, (int)top_stage, nsegs, nwaves, e->n, e->ncat);
            for (const WaveRec &r : recs) {
                fprintf(f, "W %u %u %u %u %u %llu %llu %llu %llu %d %u\n", r.kind, r.vblock, r.wave, r.xcc, r.hw, r.rt0, r.rt1, r.ct0,
                        r.ct1, (int)r.det, r.nstamp);
                if ((int)r.det >= 0 && r.det < nd) {
                    fprintf(f, "S %u", r.det);
                    const unsigned ns = r.nstamp < TRACE_MAXSTAMP ? r.nstamp : TRACE_MAXSTAMP;
                    for (unsigned i = 0; i < ns; i++) fprintf(f, " %llu", st[(size_t)r.det * TRACE_MAXSTAMP + i] - r.ct0);
                    fprintf(f, "\n");
                }
            }
            fclose(f);
        }
    }
    return hipSuccess;
}
static hipError_t launch_traverse_mfma_impl(iqhip_engine *e, const int *seg_table, int nsegs, int nwaves, bool top_stage) {
#else
hipError_t launch_traverse_mfma(iqhip_engine *e, const int *seg_table, int nsegs, int nwaves, bool top_stage) {
#endif
    TravMArgs A;
    A.ops = e->d_ops;
    A.evec = e->d_evec;
    A.inv_evec = e->d_inv_evec;
    A.aimg = e->d_aimg;
    A.tip = e->d_tip;
    A.freq = e->d_freq;
    A.invar = e->d_invar;
    A.eval = e->d_eval;
    A.rates = e->d_rates;
    A.evalc = e->d_evalc;
    A.tipc = e->d_tipc;
    A.cls = e->d_cls;
    A.img = e->d_img;
    A.img_generic = e->d_img ? e->d_img + e->img_generic_off : nullptr;
    A.slab = e->d_slab;
    A.ntiles = e->ntiles;
    A.nptn = e->nptn;
    A.segs = seg_table;
    A.nsegs_launch = nsegs;
    A.nwaves = nwaves;
    A.ncat = e->ncat;
    A.state_unknown = e->state_unknown;
    A.fold_flags = e->d_fold_flags;
    A.hold_off = -1;
    A.small_plan = 0;
    A.small_segs[0] = A.small_segs[1] = 0;
    for (int q = 0; q < kSmallPlanOps; q++) A.small_ops[q] = DevOp{};
    if (nsegs <= 0) return hipSuccess;
    if (e->mfma_pipelined && top_stage && nsegs == 1 && e->n == 64 && e->ncat == 1 && !e->row_split && e->mixed_top) {
        // whole rounds of one chain per SIMD go to full-chain workgroups, a small remainder to row-split ones
        const int64_t per_round = (int64_t)e->num_cus * 4;
        const int64_t rounds = e->ntiles / per_round, rest = e->ntiles - rounds * per_round;
        if (rounds >= 1 && rest > 0 && rest <= per_round / 2) {
            const int nfull = (int)(rounds * e->num_cus);
            return (e->plan_nleaf_tabs > 0 || e->leaf_tables) ? launch_trav_top64<true>(e, A, nfull) : launch_trav_top64<false>(e, A, nfull);
        }
    }
    if (e->mfma_pipelined) {  // plan was built in canonical (PF, PREV) form
        if (e->plan_nleaf_tabs > 0 || e->leaf_tables) {  // leaf children from the K2 tables (k_leaf_tables)
            if (e->n == 20 && e->ncat == 4)
                return e->cat_split ? launch_trav_m2<20, 1, 4, true>(e, A) : launch_trav_m2<20, 4, 1, true>(e, A);
            if (e->n == 20 && e->ncat == 1) return launch_trav_m2<20, 1, 1, true>(e, A);
            if (e->n == 64 && e->ncat == 1) return e->row_split ? launch_trav_rows64<true>(e, A) : launch_trav_m2<64, 1, 1, true>(e, A);
            return hipErrorInvalidValue;
        }
        if (e->n == 20 && e->ncat == 4) {
            if (e->cat_split) return launch_trav_m2<20, 1, 4>(e, A);
            if (e->top_cs2 && top_stage) {
                // whole rounds of two-waves-per-tile workgroups; a small remainder as one wave per category (launch_trav_top20)
                const bool mixed20 = e->mixed_top;
                const int64_t per_round = (int64_t)e->num_cus * 3 * 2;
                const int64_t rounds = e->ntiles / per_round, rest = e->ntiles - rounds * per_round;
                if (mixed20 && nsegs == 1 && rounds >= 1 && rest > 0 && rest <= per_round / 4)
                    return launch_trav_top20(e, A, (int)(rounds * e->num_cus * 3));
                return launch_trav_m2<20, 2, 2>(e, A);
            }
            return launch_trav_m2<20, 4>(e, A);
        }
        if (e->n == 20 && e->ncat == 1) return launch_trav_m2<20, 1>(e, A);
        if (e->n == 64 && e->ncat == 1) return e->row_split ? launch_trav_rows64<false>(e, A) : launch_trav_m2<64, 1>(e, A);
        return hipErrorInvalidValue;
    }
    switch (e->n) {
        case 20:
            if (e->nclass > 1) return getenv("IQHIP_MIX_GENERIC") ? launch_trav_m<20, true>(e, A) : launch_trav_mix20(e, A);
            // plain model, category count without a pipelined instantiation (+G8, +R5, ...): the mixture kernel
            // with one class (16+4-row MFMA split, A fragments in registers) beats the padded generic kernel
            if (e->d_img && !getenv("IQHIP_MIX_GENERIC")) return launch_trav_mix20(e, A);
            return launch_trav_m<20, false>(e, A);
        case 64: return e->nclass > 1 ? launch_trav_m<64, true>(e, A) : launch_trav_m<64, false>(e, A);
        case 4: return launch_trav_m<4, true>(e, A);  // 4-state mixtures (a plain 4-state model never comes here)
        default: return hipErrorInvalidValue;
    }
}

// ---------------------------------------------------------------------------------------
// streaming kernels on the [tile16][e][16] layout (generic n, ncat): lane = (pattern p, group g),
// group g handles block entries e = g, g+4, ...; the 4 groups are combined with two shuffles.
// MODE 0: branch lnL (phylokernel.h:806-838, :930-956)   -> slab[0], pattern_lh
// MODE 1: theta = a .* b (phylokernel.h:535-573)
// MODE 2: df/ddf from theta (phylokernel.h:583-651)      -> slab[0], slab[1]
// MODE 3: lnL from theta (phylokernel.h:1067-1090)       -> slab[0], pattern_lh
// ---------------------------------------------------------------------------------------
struct StreamMArgs {
    DevBranch br;
    const double *tip;
    const double *eval;
    const double *rates;
    const double *props;
    const double *freq;
    const double *invar;
    double *theta;
    double *pattern_lh;
    double *slab;
    int64_t ntiles;
    int64_t nptn;
    int64_t nobs;             // [nobs, nptn): +ASC unobserved constant patterns
    const int16_t *a_sc;      // scale counters of the branch ends (lnL modes, +ASC rescale rule)
    const int16_t *b_sc;
    int nwaves;
    int n;
    int ncat;
    double len;
    const NewtonState *st;    // a step of the enqueued Newton chain: len = st->rts, nothing to do once st->done
    size_t theta_stride;      // > 0: batched chain, blockIdx.y = task (own theta, state, pair of slab rows)
    FoldArgs fold;            // the last workgroup sums the slab itself (no k_reduce launch)
};

template <int MODE>
__global__ __launch_bounds__(256) void k_stream_mfma(const StreamMArgs A) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int N = A.n, B = A.n * A.ncat;
    double *s_v0 = smem, *s_v1 = smem + B, *s_v2 = smem + 2 * B;
    double len = A.len;
    const NewtonState *st = A.st;
    const double *theta_in = A.theta;
    double *slab = A.slab;
    if (A.theta_stride) {
        theta_in += (size_t)blockIdx.y * A.theta_stride;
        slab += (size_t)2 * blockIdx.y * A.nwaves;
        st += blockIdx.y;
    }
    if (st) {
        if (MODE == 2) {
            if (st->done) return;
            len = st->rts;
        } else {
            len = st->result;   // (lnL at the accepted length, after the chain has finished)
        }
    }
    if (MODE != 1) {
        for (int t = threadIdx.x; t < B; t += 256) {
            const int c = t / N, i = t - c * N;
            const double cof = A.eval[t] * A.rates[c];  // eval: per-category expansion [ncat][n]
            const double v = exp(cof * len) * A.props[c];
            s_v0[t] = v;
            s_v1[t] = cof * v;
            s_v2[t] = cof * (cof * v);
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= A.ntiles) return;
    const int p = lane & 15, g = lane >> 4;
    const int64_t ptn = tile * 16 + p;
    const size_t tbase = (size_t)tile * 16 * B;
    const bool in = ptn < A.nptn;
    double lh = 0.0, d1 = 0.0, d2 = 0.0;
    if (MODE == 0 || MODE == 1) {
        const double *bv = A.br.b + tbase;
        const bool leaf = A.br.a_kind == CHILD_LEAF;
        const int s = leaf ? A.br.a_states[ptn] : 0;
        const double *av = leaf ? A.tip + (size_t)s * B : A.br.a + tbase;  // tip: [state][ncat][n]
        for (int e = g; e < B; e += 4) {
            const double b = bv[(size_t)e * 16 + p];
            const double a = leaf ? av[e] : av[(size_t)e * 16 + p];
            if (MODE == 1) A.theta[tbase + (size_t)e * 16 + p] = a * b;
            else lh = fma(s_v0[e] * a, b, lh);
        }
        if (MODE == 1) return;
    } else {
        const double *th = theta_in + tbase;
        for (int e = g; e < B; e += 4) {
            const double t = th[(size_t)e * 16 + p];
            lh = fma(s_v0[e], t, lh);
            if (MODE == 2) {
                d1 = fma(s_v1[e], t, d1);
                d2 = fma(s_v2[e], t, d2);
            }
        }
    }
    lh += __shfl_xor(lh, 16, 64);
    lh += __shfl_xor(lh, 32, 64);
    if (MODE == 2) {
        d1 += __shfl_xor(d1, 16, 64);
        d1 += __shfl_xor(d1, 32, 64);
        d2 += __shfl_xor(d2, 16, 64);
        d2 += __shfl_xor(d2, 32, 64);
    }
    const double iv = A.invar[ptn];
    const bool obs = ptn < A.nobs;
    const bool unobs = (g == 0) && ptn >= A.nobs && in;
    const double f = obs ? A.freq[ptn] : 0.0;
    const bool mine = (g == 0) && obs;
    const bool asc = A.nobs < A.nptn;
    if (MODE == 2) {
        const double lhi = lh + iv;
        const double inv = 1.0 / fabs(lhi);
        const double dfp = d1 * inv;
        const double ddfp = fma(-dfp, dfp, d2 * inv);
        const double wa = wave_sum_m(mine ? dfp * f : 0.0), wb = wave_sum_m(mine ? ddfp * f : 0.0);
        if (lane == 0) {
            fold_store(&slab[tile], wa);
            fold_store(&slab[(size_t)A.nwaves + tile], wb);
        }
        if (asc) {  // phylokernel.h:655-725
            const double w2 = wave_sum_m(unobs ? lhi : 0.0), w3 = wave_sum_m(unobs ? d1 : 0.0),
                         w4 = wave_sum_m(unobs ? d2 : 0.0);
            if (lane == 0) {
                slab[(size_t)2 * A.nwaves + tile] = w2;
                slab[(size_t)3 * A.nwaves + tile] = w3;
                slab[(size_t)4 * A.nwaves + tile] = w4;
            }
        }
    } else {
        double pc = 0.0;
        if (asc && unobs) {  // phylokernel.h:894-900, 989-995, 1157-1163
            int ssc = 0;
            if (A.a_sc) ssc += A.a_sc[ptn];
            if (A.b_sc) ssc += A.b_sc[ptn];
            pc = (ssc >= 1 ? lh * kScalingThreshold : lh) + iv;
        }
        const double plh = log(fabs(lh + iv));
        if (g == 0 && A.pattern_lh) A.pattern_lh[ptn] = plh;
        const double wa = wave_sum_m(mine ? plh * f : 0.0);
        const double wpc = asc ? wave_sum_m(pc) : 0.0;
        if (lane == 0) {
            fold_store(&slab[tile], wa);
            fold_store(&slab[(size_t)A.nwaves + tile], wpc);
        }
    }
    if (A.fold.enabled) fold_tail(A.fold);
}

hipError_t launch_stream_mfma(iqhip_engine *e, int mode, const DevBranch *br, double len, int nwaves,
                              const NewtonState *st, int fold_rows, double *theta_out, const BatchChain *bc) {
    StreamMArgs A;
    A.st = bc ? bc->states : st;
    A.theta_stride = bc ? bc->theta_stride : 0;
    if (bc) fold_rows = -1;
    A.fold.slab = e->d_slab;
    A.fold.result = e->d_result;
    A.fold.ticket = e->d_fold_ticket;
    A.fold.flags = e->d_fold_flags;
    A.fold.nwaves = nwaves;
    A.fold.nrows_scale = fold_rows > 0 ? fold_rows : 0;
    A.fold.root_rows = 2;
    A.fold.enabled = (fold_rows >= 0 && mode != 1 && e->n_unobs == 0) ? 1 : 0;
    A.fold.done = nullptr;
    A.fold.seq = 0;
    if (br) A.br = *br; else A.br = DevBranch{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0.0};
    A.tip = e->d_tipc;
    A.eval = e->d_evalc;
    A.rates = e->d_rates;
    A.props = e->d_props;
    A.freq = e->d_freq;
    A.invar = e->d_invar;
    A.theta = bc ? const_cast<double *>(bc->theta) : (theta_out ? theta_out : e->d_theta);
    A.pattern_lh = bc ? nullptr : e->d_pattern_lh;
    A.slab = e->d_slab;
    A.ntiles = e->ntiles;
    A.nptn = e->nptn;
    A.nobs = e->nptn - e->n_unobs;
    A.a_sc = (mode == 0) ? (br ? br->a_sc : nullptr) : e->theta_a_sc;
    A.b_sc = (mode == 0) ? (br ? br->b_sc : nullptr) : e->theta_b_sc;
    A.nwaves = nwaves;
    A.n = e->n;
    A.ncat = e->ncat;
    A.len = len;
    const dim3 grid((unsigned)((e->ntiles + 3) / 4), (unsigned)(bc ? bc->ntasks : 1));
    const size_t lds = (size_t)3 * e->block * sizeof(double);
    switch (mode) {
        case 0: hipLaunchKernelGGL(k_stream_mfma<0>, grid, dim3(256), lds, e->stream, A); break;
        case 1: hipLaunchKernelGGL(k_stream_mfma<1>, grid, dim3(256), lds, e->stream, A); break;
        case 2: hipLaunchKernelGGL(k_stream_mfma<2>, grid, dim3(256), lds, e->stream, A); break;
        default: hipLaunchKernelGGL(k_stream_mfma<3>, grid, dim3(256), lds, e->stream, A); break;
    }
    return hipGetLastError();
}

}  // namespace iqhip
