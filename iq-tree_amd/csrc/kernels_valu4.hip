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

__device__ __forceinline__ double wave_sum(double v) { return wave_sum64(v); }

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
//
// Each wave walks the whole op list for its 64 patterns, so the list is a chain of nops
// dependent steps.  Ablation on MI355X (profiles/r01/ablation_early.txt) showed the chain to be
// instruction-ISSUE bound, not memory bound (removing every load and store changed nothing),
// so the body is built to issue as little as possible per step:
//   * per-branch uniform data is reduced to the exponentials ex[c][i] = exp(eval_i r_c t)
//     (E = U*diag(ex), phylokernel.h:159-181, so E*v = U*(ex .* v)), computed by the workgroup
//     itself into LDS once per launch (no K1 launch, no matrix traffic) and read back as
//     broadcast ds_read; U and U^-1 are op-invariant scalar (SGPR) operands;
//   * a LEAF child is the reference's K2 lookup (phylokernel.h:187-232,293-317): a 5-row table
//     (A,C,G,T,unknown -- the unknown row exactly 1.0) built in LDS per (op, leaf child) with
//     the reference's own unfused association, read with the lane's state as row index; IUPAC
//     ambiguity codes take a wave-uniform slow path that evaluates E*tip on the fly;
//   * the body is specialised at compile time on the kind of each child (LEAF / PREV / PF):
//     the previous result is consumed in place in its registers, no copies, no selects;
//   * the one child of op k+1 that has to come from memory (CHILD_PF), its scale counter and
//     the leaf state bytes are requested while op k computes.
// ---------------------------------------------------------------------------------------
struct Trav4Args {
    const DevOp *ops;
    const double *evec;
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
    int64_t nobs;      // observed patterns; [nobs, nptn) are the +ASC unobserved constant patterns
    const int *segs;   // {begin, nops} per segment; workgroup b works on segment b / ngroups
    int ngroups;       // workgroups per segment
    int nwaves;
    int state_unknown;
    int has_root;
    int lds_reg_doubles;  // size of the per-branch region area (largest chunk)
    int nsegs_launch;     // host side only: segments of this launch, instantiation choice
    int has_load;
    FoldArgs fold;
    DevBranch root;
    // a small plan inside the kernel arguments (iqhip_engine::plan_small): ops + look-ahead sentinels, one segment
    int small_plan;
    int small_segs[2];
    DevOp small_ops[kSmallPlanOps];
};

// a[x] (x = 0..3) of a LEAF child for category c: table row (fast path) or E*tip evaluated on
// the fly when some lane of the wave holds an IUPAC ambiguity code (slow, wave-uniform)
template <int CF>
__device__ __forceinline__ void leaf_cat4(const double *reg /* [ex B][table 5B] */, const double *s_tip,
                                          const CONST_AS double *U, int s, int row, bool slow,
                                          int state_unknown, int c, double (&a)[4]) {
    constexpr int B = 4 * CF;
    if (__builtin_expect(slow, 0)) {
        double l[4];
#pragma unroll
        for (int i = 0; i < 4; i++) l[i] = reg[c * 4 + i] * s_tip[s * 4 + i];
#pragma unroll
        for (int x = 0; x < 4; x++) {
            double v = U[x * 4] * l[0];
            v = fma(U[x * 4 + 1], l[1], v);
            v = fma(U[x * 4 + 2], l[2], v);
            v = fma(U[x * 4 + 3], l[3], v);
            a[x] = (s == state_unknown) ? 1.0 : v;
        }
    } else {
        const double2 *tab = reinterpret_cast<const double2 *>(reg + B + row * B + c * 4);
        const double2 v0 = tab[0], v1 = tab[1];
        a[0] = v0.x; a[1] = v0.y; a[2] = v1.x; a[3] = v1.y;
    }
}

// One node update.  The host canonicalises every op (the Hadamard product commutes) so that
//   left  is a LEAF, the streamed child held in the PF registers (CHILD_PF) or a result parked
//         in the HOLD registers by an earlier op of this launch      (CHILD_HOLD), and
//   right is a LEAF or the previous result held in `prev`            (CHILD_PREV),
// which leaves two independent wave-uniform two-way choices instead of a kind x kind product.
// `prev` is read in place and receives the new vector.  The PF registers are consumed category
// by category; right after category c has been read the same registers are re-targeted at
// op k+1's streamed child (nx_pf), so the prefetch needs no second register set and no copy.
// Returns lh_max (0 for LEAF-LEAF, which the reference never rescales).
// With SP = 2 lanes per pattern a lane owns C = CF/2 of the block's CF categories, starting at `coff`.
template <int C, int CF, bool USE_HOLD>
__device__ __forceinline__ double node_update4(bool leafL, bool holdL, bool leafR, const double *regL,
                                               const double *regR, const double *s_tip,
                                               const CONST_AS double *U, const CONST_AS double *uinv,
                                               int sL, int sR, int state_unknown, const char *nx_pf,
                                               char *dst, int coff, double (&PF)[4 * C],
                                               const double (&HOLD)[USE_HOLD ? 4 * C : 1], double (&prev)[4 * C]) {
    bool slowL = false, slowR = false;
    int rowL = 0, rowR = 0;
    if (leafL) {
        slowL = __any((sL >= 4) && (sL != state_unknown));
        rowL = sL < 4 ? sL : 4;
    }
    if (leafR) {
        slowR = __any((sR >= 4) && (sR != state_unknown));
        rowR = sR < 4 ? sR : 4;
    }
    double lh_max = 0.0;
#pragma unroll
    for (int c = 0; c < C; c++) {
        double a[4], b[4], tmp[4];
        if (leafL) {
            leaf_cat4<CF>(regL, s_tip, U, sL, rowL, slowL, state_unknown, coff + c, a);
        } else {
            double l[4];
            if (USE_HOLD && holdL) {
#pragma unroll
                for (int i = 0; i < 4; i++) l[i] = regL[(coff + c) * 4 + i] * HOLD[USE_HOLD ? c * 4 + i : 0];
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++) l[i] = regL[(coff + c) * 4 + i] * PF[c * 4 + i];
            }
#pragma unroll
            for (int x = 0; x < 4; x++) {
                double v = U[x * 4] * l[0];
                v = fma(U[x * 4 + 1], l[1], v);
                v = fma(U[x * 4 + 2], l[2], v);
                a[x] = fma(U[x * 4 + 3], l[3], v);
            }
        }
        // category c of PF is dead now: stream in the same slice of op k+1's memory child
        // (only if it has one: requests to a cache-resident dummy window, issued so that every op has the same instruction
        // sequence and counted vmcnt waits, were measured against no request at all: 0.136 vs 0.132 ms, 4.96 vs 4.79 ms at 1 M patterns)
        if (nx_pf != nullptr) {
            const double2 t0 = *reinterpret_cast<const double2 *>(nx_pf + (2 * c) * 1024);
            const double2 t1 = *reinterpret_cast<const double2 *>(nx_pf + (2 * c + 1) * 1024);
            PF[c * 4] = t0.x; PF[c * 4 + 1] = t0.y; PF[c * 4 + 2] = t1.x; PF[c * 4 + 3] = t1.y;
        }
        if (leafR) {
            leaf_cat4<CF>(regR, s_tip, U, sR, rowR, slowR, state_unknown, coff + c, b);
        } else {
            double r[4];
#pragma unroll
            for (int i = 0; i < 4; i++) r[i] = regR[(coff + c) * 4 + i] * prev[c * 4 + i];
#pragma unroll
            for (int x = 0; x < 4; x++) {
                double v = U[x * 4] * r[0];
                v = fma(U[x * 4 + 1], r[1], v);
                v = fma(U[x * 4 + 2], r[2], v);
                b[x] = fma(U[x * 4 + 3], r[3], v);
            }
        }
#pragma unroll
        for (int x = 0; x < 4; x++) tmp[x] = a[x] * b[x];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            double o = uinv[i * 4] * tmp[0];
            o = fma(uinv[i * 4 + 1], tmp[1], o);
            o = fma(uinv[i * 4 + 2], tmp[2], o);
            o = fma(uinv[i * 4 + 3], tmp[3], o);
            prev[c * 4 + i] = o;
            lh_max = fmax(lh_max, fabs(o));
        }
        // store the slice now (unscaled; the rare rescale re-stores the vector): every wave runs
        // the same op at about the same time, so stores issued only at the end of the op reach the
        // memory system in chip-wide bursts that alternate with compute instead of overlapping it
#ifndef IQHIP_ABLATE_NOSTORE  // timing-only build switch; never defined in the shipped library
        // (nontemporal stores measured here: 0.178 vs 0.160 ms -- the 4-state vectors are re-read from the L2 too soon)
        *reinterpret_cast<double2 *>(dst + (2 * c) * 1024) = make_double2(prev[c * 4], prev[c * 4 + 1]);
        *reinterpret_cast<double2 *>(dst + (2 * c + 1) * 1024) = make_double2(prev[c * 4 + 2], prev[c * 4 + 3]);
#endif
    }
    return (leafL && leafR) ? 0.0 : lh_max;
}

// byte-offset addressing: a wave-uniform base (SGPR pair) plus one per-lane 32-bit offset that
// is computed once per kernel, so no per-op VALU goes into addresses
template <int C>
__device__ __forceinline__ void load_vec4_off(const double *base, uint32_t voff, double (&v)[4 * C]) {
    const char *p = reinterpret_cast<const char *>(base) + voff;
#pragma unroll
    for (int j = 0; j < 2 * C; j++) {
        const double2 t = *reinterpret_cast<const double2 *>(p + j * 1024);
        v[2 * j] = t.x;
        v[2 * j + 1] = t.y;
    }
}
template <int C>
__device__ __forceinline__ void store_vec4_off(double *base, uint32_t voff, const double (&v)[4 * C]) {
    char *p = reinterpret_cast<char *>(base) + voff;
#pragma unroll
    for (int j = 0; j < 2 * C; j++)
        *reinterpret_cast<double2 *>(p + j * 1024) = make_double2(v[2 * j], v[2 * j + 1]);
}

// HAS_LOAD: the plan contains an op with two memory children (the second one is CHILD_LOAD and
// is read synchronously); the fast instantiation has no such path, so that the VMEM sequence
// of a loop iteration is the same on every path and the compiler can use counted vmcnt waits
// (stores of op k-1 stay in flight while op k computes).
// SP = 2: two lanes per pattern -- a wave covers half a tile (32 patterns) and each lane half of the
// categories, so a small alignment yields twice as many waves with half the register state each (the
// 4-state kernel is latency / occupancy limited below ~4 waves per SIMD).  The memory layout is
// unchanged: the lane's categories are rows [2*coff, 2*coff + 2*CL) of the tile.
// USE_HOLD = false: no HOLD register set (the plan then contains no CHILD_HOLD / push_hold): 2*BL fewer registers
// -DIQHIP_T4_TRACE (timing study, tools/build_alt.sh t4trace -DIQHIP_T4_TRACE): a few waves split their op loop with the
// shader clock -- chunk fill / op head / node update / op tail -- and every 100th launch prints the averages
#ifdef IQHIP_T4_TRACE
__device__ unsigned long long g_t4[8];
#define T4_CLK() (t4_on ? (unsigned long long)clock64() : 0ull)
#else
#define T4_CLK() 0ull
#endif
template <int C, int WG, bool HAS_LOAD, int SP, bool USE_HOLD = true>
__global__ __launch_bounds__(WG, 2) void k_traverse4(const Trav4Args A) {
    static_assert(SP == 1 || (SP == 2 && C % 2 == 0), "SP = 2 needs an even category count");
    constexpr int B = 4 * C;
    constexpr int CL = C / SP, BL = 4 * CL;  // categories / doubles per lane
    constexpr int WPB = WG / 64;  // waves per block (64 patterns each, 32 with SP = 2)
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *s_tip = smem;             // [32][4]
    double *s_val = smem + 128;       // [B]
    double *s_reg = smem + 128 + B;   // per (op, child) regions of the current chunk
    uint8_t *s_states = reinterpret_cast<uint8_t *>(s_reg + A.lds_reg_doubles);  // [slot][WG] leaf states

    __shared__ double s_model[32];    // U [16], eigenvalues [4], rates [C <= 8]: the chunk fills read them many times
    __shared__ int s_opi[64][4];      // per op of a fill pass: LDS offsets of its two regions, is-leaf flags
    __shared__ double s_opl[64][2];   // ... child branch lengths
    const int nst = A.state_unknown + 1;
    for (int t = threadIdx.x; t < nst * 4; t += WG) s_tip[t] = A.tip[t];
    for (int t = threadIdx.x; t < 20 + C; t += WG) s_model[t] = t < 16 ? A.evec[t] : (t < 20 ? A.eval[t - 16] : A.rates[t - 20]);
    if (A.has_root && threadIdx.x < B) {
        const int c = threadIdx.x >> 2, i = threadIdx.x & 3;
        s_val[threadIdx.x] = exp(A.eval[i] * (A.rates[c] * A.root.len)) * A.props[c];
    }

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // make it provably uniform
    const int seg = (int)blockIdx.x / A.ngroups;  // scalar
    // (a small plan is read out of the kernel-argument segment, which is constant memory like the plan buffer)
    const CONST_AS char *kargs = (const CONST_AS char *)__builtin_amdgcn_kernarg_segment_ptr();
    const CONST_AS int *segs = A.small_plan ? (const CONST_AS int *)(kargs + offsetof(Trav4Args, small_segs)) : as_const(A.segs);
    const int k_begin = segs[2 * seg], k_end = k_begin + segs[2 * seg + 1];
    // wave -> (tile, 32-pattern half); lane -> (pattern, category half)
    const int64_t wtile = (int64_t)((int)blockIdx.x - seg * A.ngroups) * WPB + wave;
    const bool active = wtile < A.ntiles * SP;
    const int64_t wt = active ? wtile : 0;
    const int gw = (int)wt;               // global wave id
    const int64_t tl = wt / SP;           // 64-pattern tile of the vector layout
    const int pl = (SP == 1) ? lane : (int)(wt & 1) * 32 + (lane & 31);  // pattern within the tile
    const int hl = (SP == 1) ? 0 : (lane >> 5);                           // category half of this lane
    const int coff = hl * CL;
    const bool lead = (hl == 0);          // the lane that owns the pattern's scalars
    const int64_t ptn = tl * 64 + pl;
    const uint32_t voff = (uint32_t)(tl * (64 * B * 8) + pl * 16 + coff * 2048);  // bytes into a vector slab
    const uint32_t doff = (uint32_t)(lane * 16);                      // ... into the dummy window
    const uint32_t soff = (uint32_t)(ptn * 2);                        // bytes into a scale array
    const uint32_t poff = (uint32_t)ptn;                              // bytes into a state row

    const double freq = A.freq[ptn];
    const double invar = A.invar[ptn];

    const CONST_AS double *U = as_const(A.evec);
    const CONST_AS double *uinv = as_const(A.inv_evec);
    const CONST_AS DevOp *ops = A.small_plan ? (const CONST_AS DevOp *)(kargs + offsetof(Trav4Args, small_ops)) : as_const(A.ops);

    double prev[BL], PF[BL], HOLD[USE_HOLD ? BL : 1];
    int hold_sc = 0;
    int pf_sc = 0, prev_sc = 0;
#pragma unroll
    for (int e = 0; e < BL; e++) { prev[e] = 0.0; PF[e] = 0.0; }
#pragma unroll
    for (int e = 0; e < (USE_HOLD ? BL : 1); e++) HOLD[e] = 0.0;

    // everything op k needs from memory is requested while op k-1 computes.  Prime for op 0.
    // (ops[nops] is a sentinel whose pointers are valid dummies, so the requests are unconditional)
    if (active) {
        const CONST_AS DevOp *nx = ops + k_begin;
        const int nreal = nx->real_mask;
        load_vec4_off<CL>(nx->pf, (nreal & 1) ? voff : doff, PF);
        pf_sc = *reinterpret_cast<const int16_t *>(reinterpret_cast<const char *>(nx->pf_sc) +
                                                   ((nreal & 1) ? soff : (uint32_t)(lane * 2)));
    }

#ifdef IQHIP_T4_TRACE
    const bool t4_on = (blockIdx.x % 97) == 5 && threadIdx.x == 0;
    unsigned long long t4_fill = 0, t4_head = 0, t4_upd = 0, t4_tail = 0, t4_n = 0, t4_f1 = 0, t4_f2 = 0;
#endif
    int k = k_begin;
    while (k < k_end) {
        // ---- fill the LDS regions of the chunk that starts at op k (host-chosen boundaries)
        const int kn = ops[k].chunk_nops;
        [[maybe_unused]] const unsigned long long c0 = T4_CLK();
        __syncthreads();  // the previous chunk's regions are no longer read
        // The fill reads every op's descriptor and the model once per table entry: from LDS copies (the descriptors' few
        // fields staged per pass of 64 ops, eigenvalues / rates / U at kernel start) instead of dependent reads from memory --
        // those cost 34 k of a wave's 254 k cycles per traversal at 66 k patterns (shader clock; the exponentials themselves
        // are about 9 k of that).
        [[maybe_unused]] unsigned long long c0a = T4_CLK();
        for (int o0 = 0; o0 < kn; o0 += 64) {
            const int on = kn - o0 < 64 ? kn - o0 : 64;
            if (o0 > 0) __syncthreads();
            for (int t = threadIdx.x; t < on; t += WG) {
                const CONST_AS DevOp &d = ops[k + o0 + t];
                s_opi[t][0] = d.lds_left;
                s_opi[t][1] = d.lds_right;
                s_opi[t][2] = d.left_kind == CHILD_LEAF;
                s_opi[t][3] = d.right_kind == CHILD_LEAF;
                s_opl[t][0] = op_child_len(d, 0);
                s_opl[t][1] = op_child_len(d, 1);
            }
            __syncthreads();
            for (int t = threadIdx.x; t < on * 2 * B; t += WG) {  // phase 1: exponentials
                const int o = t / (2 * B), r = t - o * (2 * B), child = r / B, e = r - child * B;
                s_reg[s_opi[o][child] + e] = exp(s_model[16 + (e & 3)] * (s_model[20 + (e >> 2)] * s_opl[o][child]));
            }
            __syncthreads();
            c0a = T4_CLK();
            // phase 2: leaf tables (K2).  One item = one (op, child, category, x): the four rounded products E = U[x][.] * ex are
            // made once and serve the table's four state rows (A, C, G, T); the fifth row, STATE_UNKNOWN, is exactly 1.0
            // (phylokernel.h:228-232).  (Entry by entry -- every entry remaking its E -- this phase was 14.6 k of a wave's cycles.)
            for (int t = threadIdx.x; t < on * 2 * B; t += WG) {
                const int o = t / (2 * B), r = t - o * (2 * B), child = r / B, e = r - child * B, c = e >> 2, x = e & 3;
                if (!s_opi[o][2 + child]) continue;
                double *reg = s_reg + s_opi[o][child];
                // E = U*ex rounded first, then the reference's (t0+t1)+(t2+t3), all unfused
                const double e0 = __dmul_rn(s_model[x * 4 + 0], reg[c * 4 + 0]);
                const double e1 = __dmul_rn(s_model[x * 4 + 1], reg[c * 4 + 1]);
                const double e2 = __dmul_rn(s_model[x * 4 + 2], reg[c * 4 + 2]);
                const double e3 = __dmul_rn(s_model[x * 4 + 3], reg[c * 4 + 3]);
#pragma unroll
                for (int row = 0; row < 4; row++) {
                    const double *tp = s_tip + row * 4;
                    reg[B + row * B + e] = __dadd_rn(__dadd_rn(__dmul_rn(e0, tp[0]), __dmul_rn(e1, tp[1])),
                                                     __dadd_rn(__dmul_rn(e2, tp[2]), __dmul_rn(e3, tp[3])));
                }
                reg[B + 4 * B + e] = 1.0;
            }
        }
        [[maybe_unused]] const unsigned long long c0b = T4_CLK();
        // phase 3: this thread's leaf state bytes of the whole chunk -> LDS.  They are read once per
        // traversal (cold misses); issuing them as one burst pays the miss latency once per chunk
        // instead of once per op and keeps the op loop's VMEM sequence short.
        // (Unconditional requests, sixteen ops = thirty-two bytes in flight at a time: sl / sr of a non-leaf child point at a valid
        // row and its slot 0 is never read.  The conditional form -- load, wait, LDS store, next child -- paid one cold miss
        // per LEAF CHILD instead of per chunk: 87 k of a wave's 313 k cycles per traversal at 66 k patterns, measured with
        // the shader clock.)
        // One lane per pattern (SP = 1): the wave's 64 state bytes of a leaf child are 16 dwords, which `global_load_lds_dword`
        // moves straight into LDS (lane i's dword lands at base + 4 i: tools/lds_load_probe.hip) -- no registers, every
        // request of the chunk in flight at once, one wait.  (Through registers, 32 bytes in flight at a time, the staging
        // was 22 k of a wave's 230 k cycles per traversal.)
        if constexpr (SP == 1) {
            if (active && lane < 16) {
                for (int o = 0; o < kn; o++) {
                    const CONST_AS DevOp *d = ops + (k + o);
                    if (d->left_kind == CHILD_LEAF)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(d->sl + tl * 64 + 4 * lane),
                                                         (__attribute__((address_space(3))) void *)(s_states + d->sl_slot * WG + wave * 64), 4, 0, 0);
                    if (d->right_kind == CHILD_LEAF)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(d->sr + tl * 64 + 4 * lane),
                                                         (__attribute__((address_space(3))) void *)(s_states + d->sr_slot * WG + wave * 64), 4, 0, 0);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else
        if (active) {
            for (int o0 = 0; o0 < kn; o0 += 16) {
                uint8_t vl[16], vr[16];
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    const CONST_AS DevOp *d = ops + (k + (o0 + j < kn ? o0 + j : kn - 1));
                    vl[j] = *(d->sl + poff);
                    vr[j] = *(d->sr + poff);
                }
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    if (o0 + j < kn) {
                        const CONST_AS DevOp *d = ops + (k + o0 + j);
                        s_states[d->sl_slot * WG + threadIdx.x] = vl[j];
                        s_states[d->sr_slot * WG + threadIdx.x] = vr[j];
                    }
                }
            }
        }
        __syncthreads();
        if (!active) { k += kn; continue; }
#ifdef IQHIP_T4_TRACE
        { const unsigned long long cz = T4_CLK(); t4_fill += cz - c0; t4_f1 += c0a - c0; t4_f2 += c0b - c0a; }
#endif

        for (int kk = 0; kk < kn; kk++, k++) {
            [[maybe_unused]] const unsigned long long c1 = T4_CLK();
            const CONST_AS DevOp *op = ops + k;
            const bool leafL = op->left_kind == CHILD_LEAF;
            const bool holdL = USE_HOLD && op->left_kind == CHILD_HOLD;
            const bool leafR = op->right_kind == CHILD_LEAF;
            // (descriptor fields that only the op's tail needs: requested now, or the tail begins with their round trip)
            int16_t *const dst_sc_p = op->dst_sc;
            const int out_row = op->out_row, rule = op->no_scale, push_hold = op->push_hold;
            int sc = 0;
            // leaf states were staged into LDS when the chunk was filled
            const int sL = s_states[op->sl_slot * WG + threadIdx.x];
            const int sR = s_states[op->sr_slot * WG + threadIdx.x];
            if (!leafL) sc += holdL ? hold_sc : pf_sc;
            if (HAS_LOAD && op->right_kind == CHILD_LOAD) {
                // both children come from memory: the left one was streamed into PF; the right
                // one is read now into `prev`, which is dead here (it is not an input of this op)
                load_vec4_off<CL>(op->ld, voff, prev);
                prev_sc = *reinterpret_cast<const int16_t *>(reinterpret_cast<const char *>(op->ld_sc) + soff);
            }
            if (!leafR) sc += prev_sc;
            // ---- request op k+1's memory inputs while this op computes (sentinel at nops);
            //      the vector itself is streamed inside node_update4, slice by slice
            // (An op that streams nothing gets no request: see node_update4.)
            const CONST_AS DevOp *nx = ops + (k + 1);
            const int nreal = nx->real_mask;
            const char *nx_pf = (nreal & 1) ? reinterpret_cast<const char *>(nx->pf) + voff : nullptr;
            if (nreal & 1) pf_sc = *reinterpret_cast<const int16_t *>(reinterpret_cast<const char *>(nx->pf_sc) + soff);
            char *dstp = reinterpret_cast<char *>(op->dst) + voff;
            [[maybe_unused]] const unsigned long long c2 = T4_CLK();
            double lh_max = node_update4<CL, C, USE_HOLD>(leafL, holdL, leafR, s_reg + op->lds_left, s_reg + op->lds_right,
                                                          s_tip, U, uinv, sL, sR, A.state_unknown, nx_pf, dstp, coff, PF, HOLD, prev);
            if (SP == 2) lh_max = fmax(lh_max, __shfl_xor(lh_max, 32, 64));  // both category halves of the pattern
            [[maybe_unused]] const unsigned long long c3 = T4_CLK() + (lh_max == -1.0 ? 1 : 0);   // (after the update's results exist)
            // ---- scaling (SIMD rule, phylokernel.h:379-392,461-474); TIP-TIP never scales
            // (the last update of a multifurcating node carries the scalar kernel's rule: lh_max == 0 first, phylotreesse.cpp:774-788)
            const bool zero = rule == 2 && !(leafL && leafR) && lh_max == 0.0;   // (TIP-TIP never scales, scalar kernel :807-843)
            const bool do_scale = zero || (!(leafL && leafR) && (lh_max < kScalingThreshold) && (invar == 0.0) && rule != 1);
            double my_scale = 0.0;
            const unsigned long long any = __ballot(do_scale);
            if (__builtin_expect(any != 0, 0)) {  // rare, wave-uniform
                if (do_scale) {
                    if (zero) {   // "very shitty data": the unknown tip's vector in every category, four scaling events
#pragma unroll
                        for (int e = 0; e < BL; e++) prev[e] = s_tip[A.state_unknown * 4 + (e & 3)];
                    } else {
#pragma unroll
                        for (int e = 0; e < BL; e++) prev[e] *= kScalingThresholdInv;
                    }
                    sc += zero ? 4 : 1;
                    my_scale = (lead && ptn < A.nptn) ? (zero ? 4.0 : 1.0) * (kLogScalingThreshold * freq) : 0.0;
#ifndef IQHIP_ABLATE_NOSTORE
                    store_vec4_off<CL>(op->dst, voff, prev);
#endif
                }
            }
            prev_sc = sc;
            if (USE_HOLD && push_hold) {
                // this result is the left child of a join a few ops ahead whose other subtree is a
                // plain chain: park it in registers instead of re-reading 8 KiB per wave from memory
#pragma unroll
                for (int e = 0; e < BL; e++) HOLD[e] = prev[e];
                hold_sc = sc;
            }
#ifndef IQHIP_ABLATE_NOSTORE
            if (SP == 1 || lead) *reinterpret_cast<int16_t *>(reinterpret_cast<char *>(dst_sc_p) + soff) = (int16_t)sc;
#endif
            // deterministic reduction: wave partial -> slab[2+k][gw]
            double ws = 0.0;
            if (any) ws = wave_sum(my_scale);
            if (lane == 0) {
                fold_store(&A.slab[(size_t)(2 + out_row) * A.nwaves + gw], ws);
                if (any) fold_flag(A.fold, 2 + out_row);
            }
#ifdef IQHIP_T4_TRACE
            { const unsigned long long c4 = T4_CLK(); t4_head += c2 - c1; t4_upd += c3 - c2; t4_tail += c4 - c3; t4_n++; }
#endif
        }
    }
#ifdef IQHIP_T4_TRACE
    if (t4_on) {
        atomicAdd(&g_t4[0], t4_fill); atomicAdd(&g_t4[1], t4_head); atomicAdd(&g_t4[2], t4_upd); atomicAdd(&g_t4[3], t4_tail);
        atomicAdd(&g_t4[4], t4_n); atomicAdd(&g_t4[5], 1ull); atomicAdd(&g_t4[6], t4_f1); atomicAdd(&g_t4[7], t4_f2);
    }
#endif
    if (k_end == k_begin) __syncthreads();  // s_tip / s_val visibility for a root-only launch
    if (!active) return;   // (an exited wave no longer takes part in workgroup barriers)

    if (A.has_root) {
        // ---- branch lnL, phylokernel.h:806-838 (leaf form) / :930-956 (internal form)
        double Bv[BL];
        const double *sv = s_val + coff * 4;  // this lane's categories
        if (A.root.b_kind == CHILD_PREV) {
#pragma unroll
            for (int e = 0; e < BL; e++) Bv[e] = prev[e];
        } else {
            load_vec4_off<CL>(A.root.b, voff, Bv);
        }
        double lh = 0.0;
        if (A.root.a_kind == CHILD_LEAF) {
            const int s = A.root.a_states[ptn];
#pragma unroll
            for (int e = 0; e < BL; e++) lh = fma(sv[e] * s_tip[s * 4 + (e & 3)], Bv[e], lh);
        } else {
            double Av[BL];
            if (A.root.a_kind == CHILD_PREV) {
#pragma unroll
                for (int e = 0; e < BL; e++) Av[e] = prev[e];
            } else {
                load_vec4_off<CL>(A.root.a, voff, Av);
            }
#pragma unroll
            for (int e = 0; e < BL; e++) lh = fma(sv[e] * Av[e], Bv[e], lh);
        }
        if (SP == 2) lh += __shfl_xor(lh, 32, 64);  // sum over both category halves
        double pc = 0.0;
        if (A.nobs < A.nptn) {
            // +ASC: prob_const = sum over the unobserved constant patterns of lh_ptn, the block sum
            // multiplied by 2^-256 once when the summed scale counters are >= 1 (phylokernel.h:894-897,
            // :989-992), then + ptn_invar
            if (lead && ptn >= A.nobs && ptn < A.nptn) {
                int ssc = (A.root.b_kind == CHILD_PREV) ? prev_sc : (int)A.root.b_sc[ptn];
                if (A.root.a_kind != CHILD_LEAF) ssc += (A.root.a_kind == CHILD_PREV) ? prev_sc : (int)A.root.a_sc[ptn];
                pc = (ssc >= 1 ? lh * kScalingThreshold : lh) + invar;
            }
        }
        lh += invar;
        const double plh = log(fabs(lh));
        if (SP == 1 || lead) A.pattern_lh[ptn] = plh;
        const double acc = (lead && ptn < A.nobs) ? plh * freq : 0.0;
        const double ws = wave_sum(acc);
        const double wpc = (A.nobs < A.nptn) ? wave_sum(pc) : 0.0;
        if (lane == 0) {
            fold_store(&A.slab[gw], ws);
            fold_store(&A.slab[(size_t)A.nwaves + gw], wpc);
        }
    }
    if constexpr (WG == 256) {
        if (A.fold.enabled) fold_tail(A.fold);
    }
}

template <int C, int WG, bool HAS_LOAD, int SP, bool USE_HOLD>
static hipError_t launch_trav_h(iqhip_engine *e, Trav4Args &A) {
    constexpr int B = 4 * C;
    const size_t lds = (size_t)(128 + B + (size_t)e->plan_lds_doubles) * sizeof(double) +
                       (size_t)e->plan_state_slots * WG;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_traverse4<C, WG, HAS_LOAD, SP, USE_HOLD>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);  // (160 KB minus the static arrays: fold_tail, the fill's descriptor copies)
        attr_set = true;
    }
    constexpr int WPB = WG / 64;
    A.ngroups = (int)((e->ntiles * SP + WPB - 1) / WPB);
    hipLaunchKernelGGL((k_traverse4<C, WG, HAS_LOAD, SP, USE_HOLD>), dim3((unsigned)(A.ngroups * A.nsegs_launch)), dim3(WG), lds, e->stream, A);
    return hipGetLastError();
}

template <int C, int WG, bool HAS_LOAD, int SP>
static hipError_t launch_trav_l(iqhip_engine *e, Trav4Args &A) {
    // (only the default workgroup size is instantiated both ways)
    if constexpr (WG == 256 && SP == 1) {
        if (!e->use_hold) return launch_trav_h<C, WG, HAS_LOAD, SP, false>(e, A);
    }
    return launch_trav_h<C, WG, HAS_LOAD, SP, true>(e, A);
}

template <int C, int WG>
static hipError_t launch_trav_c(iqhip_engine *e, Trav4Args &A) {
    if constexpr (C % 2 == 0) {
        if (e->lane_split == 2)
            return A.has_load ? launch_trav_l<C, WG, true, 2>(e, A) : launch_trav_l<C, WG, false, 2>(e, A);
    }
    return A.has_load ? launch_trav_l<C, WG, true, 1>(e, A) : launch_trav_l<C, WG, false, 1>(e, A);
}

template <int C>
static hipError_t launch_trav_wg(iqhip_engine *e, Trav4Args &A) {
    switch (e->wg_size) {
        case 64: return launch_trav_c<C, 64>(e, A);
        case 128: return launch_trav_c<C, 128>(e, A);
        default: return launch_trav_c<C, 256>(e, A);
    }
}

hipError_t launch_traverse4(iqhip_engine *e, const int *seg_table, int nsegs, bool has_load, const DevBranch *root, int nwaves,
                            int fold_rows) {
#ifdef IQHIP_T4_TRACE
    {
        static int calls = 0;
        if (++calls % 100 == 0) {
            unsigned long long h[8];
            (void)hipDeviceSynchronize();
            (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_t4), sizeof h);
            if (h[4]) fprintf(stderr, "[t4] waves %llu ops %llu: per op head %.0f update %.0f tail %.0f clk; chunk fills %.0f clk per wave (exponentials %.0f, tables %.0f)\n",
                              h[5], h[4], (double)h[1] / h[4], (double)h[2] / h[4], (double)h[3] / h[4], (double)h[0] / h[5], (double)h[6] / h[5], (double)h[7] / h[5]);
            unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_t4), z, sizeof z);
        }
    }
#endif
    Trav4Args A;
    A.fold.slab = e->d_slab;
    A.fold.result = e->d_result;
    A.fold.ticket = e->d_fold_ticket;
    A.fold.flags = e->d_fold_flags;
    A.fold.nwaves = nwaves;
    A.fold.nrows_scale = fold_rows > 0 ? fold_rows : 0;
    A.fold.root_rows = root ? (e->n_unobs > 0 ? 2 : 1) : 0;
    A.fold.enabled = (fold_rows >= 0 && e->wg_size == 256) ? 1 : 0;
    A.fold.done = nullptr;
    A.fold.seq = 0;
    if (A.fold.enabled && e->poll_result && e->d_result == e->d_result_own) {   // mapped host memory: the host may poll
        A.fold.seq = ++e->result_seq;
        A.fold.done = e->d_done;
        e->poll_pending = true;
        e->folded_rows = A.fold.nrows_scale;
    }
    A.ops = e->d_ops;
    A.evec = e->d_evec;
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
    A.nobs = e->nptn - e->n_unobs;
    A.segs = seg_table;
    A.nsegs_launch = nsegs;
    A.has_load = has_load ? 1 : 0;
    A.nwaves = nwaves;
    A.state_unknown = e->state_unknown;
    A.has_root = root ? 1 : 0;
    A.lds_reg_doubles = e->plan_lds_doubles;
    if (root) A.root = *root; else A.root = DevBranch{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0.0};
    A.small_plan = 0;
    A.small_segs[0] = A.small_segs[1] = 0;
    if (e->plan_small && nsegs == 1) {   // (the plan was not copied to d_ops: it travels with the launch)
        A.small_plan = 1;
        A.small_segs[1] = e->plan_small_nops;
        for (int q = 0; q < kSmallPlanOps; q++) A.small_ops[q] = e->h_ops[q];
    } else {
        for (int q = 0; q < kSmallPlanOps; q++) A.small_ops[q] = DevOp{};
    }
    switch (e->ncat) {
        case 1: return launch_trav_wg<1>(e, A);
        case 2: return launch_trav_wg<2>(e, A);
        case 3: return launch_trav_wg<3>(e, A);
        case 4: return launch_trav_wg<4>(e, A);
        case 5: return launch_trav_wg<5>(e, A);
        case 6: return launch_trav_wg<6>(e, A);
        case 7: return launch_trav_wg<7>(e, A);
        case 8: return launch_trav_wg<8>(e, A);
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

hipError_t launch_theta4(iqhip_engine *e, const DevBranch &br, double *theta_out) {
    const int grid = (int)((e->ntiles + 3) / 4);
    double *const theta = theta_out ? theta_out : e->d_theta;
#define IQ_THETA(Cv)                                                                        \
    case Cv:                                                                                \
        hipLaunchKernelGGL(k_theta4<Cv>, dim3(grid), dim3(256), 0, e->stream, br, e->d_tip, \
                           e->state_unknown, theta, e->ntiles);                          \
        break;
    switch (e->ncat) {
        IQ_THETA(1) IQ_THETA(2) IQ_THETA(3) IQ_THETA(4) IQ_THETA(5) IQ_THETA(6) IQ_THETA(7) IQ_THETA(8)
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
    int nwaves, int64_t nobs, const int16_t *__restrict__ a_sc, const int16_t *__restrict__ b_sc,
    const NewtonState *st, size_t theta_stride) {
    constexpr int B = 4 * C;
    __shared__ double s_v0[B], s_v1[B], s_v2[B];
    if (theta_stride) {   // batched chain (blockIdx.y = task): own theta, own state, own pair of slab rows
        theta += (size_t)blockIdx.y * theta_stride;
        slab += (size_t)2 * blockIdx.y * nwaves;
        st += blockIdx.y;
    }
    if (st) {  // a step of the enqueued Newton chain: the branch length is the state's current iterate
        if (MODE == 0) {
            if (st->done) return;
            len = st->rts;
        } else {
            len = st->result;   // (lnL at the accepted length, after the chain has finished)
        }
    }
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
    const double f = (ptn < nobs) ? freq[ptn] : 0.0;
    const bool unobs = ptn >= nobs && ptn < nptn;  // +ASC tail
    double lh = 0.0, d1 = 0.0, d2 = 0.0;
#pragma unroll
    for (int e = 0; e < B; e++) {
        lh = fma(s_v0[e], Th[e], lh);
        if (MODE == 0) {
            d1 = fma(s_v1[e], Th[e], d1);
            d2 = fma(s_v2[e], Th[e], d2);
        }
    }
    const double iv = invar[ptn];
    if (MODE == 0) {
        const double lhi = lh + iv;
        const double inv = 1.0 / fabs(lhi);
        const double dfp = d1 * inv;
        const double ddfp = fma(-dfp, dfp, d2 * inv);
        const double a = (ptn < nobs) ? dfp * f : 0.0;
        const double b = (ptn < nobs) ? ddfp * f : 0.0;
        const double wa = wave_sum(a), wb = wave_sum(b);
        if (lane == 0) {
            slab[gw] = wa;
            slab[(size_t)nwaves + gw] = wb;
        }
        if (nobs < nptn) {  // phylokernel.h:655-725: plain sums over the unobserved patterns, no rescale
            const double w2 = wave_sum(unobs ? lhi : 0.0), w3 = wave_sum(unobs ? d1 : 0.0),
                         w4 = wave_sum(unobs ? d2 : 0.0);
            if (lane == 0) {
                slab[(size_t)2 * nwaves + gw] = w2;
                slab[(size_t)3 * nwaves + gw] = w3;
                slab[(size_t)4 * nwaves + gw] = w4;
            }
        }
    } else {
        double pc = 0.0;
        if (nobs < nptn && unobs) {  // phylokernel.h:1138-1163
            int ssc = 0;
            if (a_sc) ssc += a_sc[ptn];
            if (b_sc) ssc += b_sc[ptn];
            pc = (ssc >= 1 ? lh * kScalingThreshold : lh) + iv;
        }
        const double plh = log(fabs(lh + iv));
        if (pattern_lh) pattern_lh[ptn] = plh;
        const double a = (ptn < nobs) ? plh * f : 0.0;
        const double wa = wave_sum(a);
        const double wpc = (nobs < nptn) ? wave_sum(pc) : 0.0;
        if (lane == 0) {
            slab[gw] = wa;
            slab[(size_t)nwaves + gw] = wpc;
        }
    }
}

template <int MODE>
static hipError_t launch_theta_reduce(iqhip_engine *e, double len, int nwaves, const NewtonState *st = nullptr,
                                      const BatchChain *bc = nullptr) {
    const int grid = (int)((e->ntiles + 3) / 4);
    const double *theta = bc ? bc->theta : e->d_theta;
    double *plh = bc ? nullptr : e->d_pattern_lh;
    const size_t stride = bc ? bc->theta_stride : 0;
    const int ny = bc ? bc->ntasks : 1;
#define IQ_TR(Cv)                                                                              \
    case Cv:                                                                                   \
        hipLaunchKernelGGL((k_theta_reduce4<Cv, MODE>), dim3(grid, ny), dim3(256), 0, e->stream, \
                           theta, e->d_eval, e->d_rates, e->d_props, len, e->d_freq,           \
                           e->d_invar, plh, e->d_slab, e->ntiles, e->nptn, nwaves,             \
                           e->nptn - e->n_unobs, e->theta_a_sc, e->theta_b_sc, st, stride);    \
        break;
    switch (e->ncat) {
        IQ_TR(1) IQ_TR(2) IQ_TR(3) IQ_TR(4) IQ_TR(5) IQ_TR(6) IQ_TR(7) IQ_TR(8)
        default: return hipErrorInvalidValue;
    }
#undef IQ_TR
    return hipGetLastError();
}

hipError_t launch_derv4(iqhip_engine *e, double len, int nwaves, const NewtonState *st, const BatchChain *bc) {
    return launch_theta_reduce<0>(e, len, nwaves, bc ? bc->states : st, bc);
}
hipError_t launch_lnl_theta4(iqhip_engine *e, double len, int nwaves, const BatchChain *bc) {
    return launch_theta_reduce<1>(e, len, nwaves, bc ? bc->states : nullptr, bc);
}

// ---------------------------------------------------------------------------------------
// fixed-order reduction of the wave partials: block v sums slab[v][0..nwaves) -> result[v]
// ---------------------------------------------------------------------------------------
// done != nullptr: the result vector is mapped host memory and the host polls it instead of paying a stream
// synchronisation (engine.hip read_result): every block publishes its row with a system-scope fence and takes a ticket;
// the block that draws the last one stores the sequence number the host is waiting for.
__global__ __launch_bounds__(256) void k_reduce(const double *__restrict__ slab, int nwaves,
                                                int first_row, double *__restrict__ result, unsigned int *ticket,
                                                volatile unsigned long long *done, unsigned long long seq) {
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
    if (threadIdx.x == 0) {
        result[first_row + blockIdx.x] = s[0];
        if (done) {
            __threadfence_system();
            const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t == gridDim.x - 1) {
                __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __threadfence_system();
                *done = seq;
            }
        }
    }
}

hipError_t launch_reduce(iqhip_engine *e, int first_row, int nrows, int nwaves) {
    if (nrows <= 0) return hipSuccess;
    unsigned long long *done = nullptr;
    unsigned long long seq = 0;
    if (e->poll_result && e->d_result == e->d_result_own) {  // mapped host memory: the host may poll
        seq = ++e->result_seq;
        done = e->d_done;
        e->poll_pending = true;
    }
    hipLaunchKernelGGL(k_reduce, dim3(nrows), dim3(256), 0, e->stream, e->d_slab, nwaves,
                       first_row, e->d_result, e->d_fold_ticket + 1, done, seq);
    return hipGetLastError();
}

}  // namespace iqhip
