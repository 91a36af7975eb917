// iqhip_internal.h -- engine-private declarations shared by the engine and the kernel files.
// Device data layout (DESIGN.md "HBM layout"):
//   VALU path (nstates == 4): patterns are grouped in tiles of 64 (one wavefront); inside a
//   tile element e (= c*4+i, the reference's block index) of pattern p is stored at
//       tile*64*B + (e>>1)*128 + (p&63)*2 + (e&1)            [doubles]
//   so one global_load_dwordx4 per lane reads two consecutive elements and the wave reads
//   1 KiB contiguous.
//   MFMA path (nstates == 20 / 64): tiles of 16 patterns, state-major inside the tile:
//       tile*16*B + (c*n + i)*16 + (p&15)
//   which is exactly the B-operand / D-result image of v_mfma_f64_16x16x4_f64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <unordered_map>
#include <vector>
#include "../../include/iqhip.h"

namespace iqhip {

constexpr int kSmallPlanOps = 4;         // op descriptors (incl. the two look-ahead sentinels) that fit the kernel arguments
constexpr int kNewtonPostEpochs = 128;   // evaluations of one k_newton launch that have a post slot
constexpr double kScalingThreshold = 0x1p-256;       // phylotree.h:52
constexpr double kScalingThresholdInv = 0x1p256;     // phylotree.h:51
// log(2^-256) as libm returns it (phylotree.h:53)
constexpr double kLogScalingThreshold = -177.44567822334599;

// LEAF: state bytes; LOAD: read now; PREV: previous op's result (registers);
// PF: read one op ahead into the prefetch registers
// HOLD: parked in the HOLD registers by the producing op (push_hold) of the same launch
enum ChildKind : int32_t { CHILD_LEAF = 0, CHILD_LOAD = 1, CHILD_PREV = 2, CHILD_PF = 3, CHILD_HOLD = 4 };

// One node update as the device sees it. 16-byte aligned so that the wave-uniform reads
// of the descriptor become scalar loads.
struct __attribute__((aligned(16))) DevOp {
    double *dst;
    int16_t *dst_sc;
    const double *pf;        // the streamed child (kind CHILD_PF) or the engine's dummy slab
    const int16_t *pf_sc;    // its scale counters (or dummy)
    const uint8_t *sl;       // state row of a LEAF left child (or a dummy row)
    const uint8_t *sr;       // ... right child
    const double *ld;        // second memory child (kind CHILD_LOAD, right only), else dummy
    const int16_t *ld_sc;
    int32_t left_kind;
    int32_t right_kind;
    double left_len;
    double right_len;
    int32_t lds_left;   // offset (doubles) of the child's LDS region inside the chunk:
    int32_t lds_right;  //   internal child: [ex B]; leaf child: [ex B][table 5B]
    int32_t chunk_nops; // > 0 on the first op of an LDS chunk: number of ops in the chunk
    int32_t real_mask;  // bit0: pf/pf_sc are real (else dummies)
    int32_t sl_slot;    // LDS slot of the staged leaf states of the left / right child within
    int32_t sr_slot;    //   the chunk (slot 0 is shared by all non-leaf children)
    int32_t push_hold;  // 1: copy this op's result into the HOLD registers (consumed by a CHILD_HOLD)
    int32_t out_row;    // row of the caller's op list this op answers (sum_scale[out_row]); plans may be reordered
    int32_t no_scale;   // scaling rule: 0 SIMD kernel's; 1 never (IQHIP_OP_NO_SCALE: intermediate product of a multifurcating node);
                        // 2 scalar kernel's (IQHIP_OP_SCALAR_RULE: + the lh_max == 0 branch, phylotreesse.cpp:774-788)
    // matrix-core kernels: transition tables of the LEAF children (K2, phylokernel.h:187-232,293-317), built per
    // submission by k_leaf_tables into an L2-resident buffer: tab[c][state][n] (n permuted to the accumulator
    // image, see kernels_mfma.hip leaf_tab_pos); dummy-valid for non-leaf children
    const double *tabL;
    const double *tabR;
    // branch-length sweeps (iqhip_optimize_sweep): a child branch whose length is the result of an earlier step of the
    // same submission is read from device memory when the op runs (written there by that step's Newton solve);
    // nullptr: the host value left_len / right_len
    const double *left_len_p;
    const double *right_len_p;
    // 20 states x 4 categories, both children leaves ("cherry"): the node's whole vector is a function of the pair of leaf
    // states only -- cherry[(sL * (STATE_UNKNOWN + 1) + sR) * block + ...] holds it for every pair, in the order the lanes
    // keep it in registers (kernels_mfma.hip k_cherry_transpose); nullptr: compute the three matrix products
    const double *cherry;
    const double *_pad_cherry;
};

#if defined(__HIPCC__)
// Sum over the 64 lanes of a wave, every lane gets it: the xor butterfly v += v^32, ^16, ^8, ^4, ^2, ^1 of the shuffle
// form, on the vector ALU instead of six dependent ds_bpermute round trips through the LDS crossbar (a derivative
// evaluation of a branch-length sweep spent half its 2 us in two of these).  Same bits as the shuffle form: the permlane
// swaps and quad permutes deliver exactly lane^32, ^16, ^2, ^1; row_ror:8 is lane^8 within a 16-lane row; row_ror:4
// delivers lane (i + 4) mod 16 instead of i^4, which holds the same value, because after the ^8 step lanes that agree in
// their low three bits hold equal sums -- and fp addition commutes.
__device__ __forceinline__ double wave_sum64(double v) {
    unsigned lo = __double2loint(v), hi = __double2hiint(v);
    auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    v = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
    lo = __double2loint(v);
    hi = __double2hiint(v);
    auto c = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto d = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    v = __hiloint2double(d[0], c[0]) + __hiloint2double(d[1], c[1]);
#define IQHIP_DPP_STEP(ctrl)                                                                                  \
    v += __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(v), ctrl, 0xF, 0xF, true),                  \
                          __builtin_amdgcn_mov_dpp(__double2loint(v), ctrl, 0xF, 0xF, true));
    IQHIP_DPP_STEP(0x128)   // row_ror:8
    IQHIP_DPP_STEP(0x124)   // row_ror:4
    IQHIP_DPP_STEP(0x4E)    // quad_perm [2,3,0,1]
    IQHIP_DPP_STEP(0xB1)    // quad_perm [1,0,3,2]
#undef IQHIP_DPP_STEP
    return v;
}

// length of a child branch of a node update: the host value, or the result of an earlier step of a sweep
template <typename OpRef>
__device__ __forceinline__ double op_child_len(const OpRef &d, int child) {
    const double *p = child ? d.right_len_p : d.left_len_p;
    return p ? *p : (child ? d.right_len : d.left_len);
}
#endif

// Root branch descriptor for the lnL / theta kernels.
struct __attribute__((aligned(16))) DevBranch {
    const double *a;            // internal side A (nullptr when A is a leaf)
    const uint8_t *a_states;    // leaf side states (nullptr when A is internal)
    const double *b;            // always internal
    const int16_t *a_sc;        // scale counters of the internal ends (+ASC rescale rule), else nullptr
    const int16_t *b_sc;
    int32_t a_kind;             // CHILD_LEAF / CHILD_LOAD / CHILD_PREV
    int32_t b_kind;             // CHILD_LOAD / CHILD_PREV
    double len;
};

// one node update / one step of a persistent branch-length sweep (kernels_sweep.hip k_sweep4)
struct __attribute__((aligned(16))) SweepOp {
    double *dst;
    int16_t *dst_sc;
    const double *lv, *rv;      // child vectors; nullptr: the child is a leaf
    const int16_t *lsc, *rsc;
    const uint8_t *ls, *rs;     // state rows of leaf children
    double llen, rlen;
    int32_t llen_step, rlen_step;  // >= 0: the child branch's length is the one that step of this sweep accepted
    int32_t no_scale;           // scaling rule, as DevOp::no_scale
    int32_t row;                // row of the wave-partial slab / of the caller's concatenated sum_scale array
};
struct __attribute__((aligned(16))) SweepStep {
    DevBranch br;
    double xguess;
    int32_t op_begin, nops;
};

struct Slab {
    double *plh = nullptr;
    int16_t *sc = nullptr;
};

struct __attribute__((aligned(16))) TabJob {  // one K2 table to build: tab[c][state][pos(x)] for branch length len
    double len;
    double *tab;
    const double *len_p;   // non-null: the length is read from device memory (sweeps), `len` is ignored
    double _pad;
};

// Optimization::minimizeNewton (optimization.cpp:388-465) as a state machine that is advanced once per derivative
// evaluation: `newton_init`, then after every evaluation at `rts` one `newton_update(sum f*df_ptn, sum f*ddf_ptn)`
// until `done`.  The same function runs in a 1-thread kernel (sharded engines: derivative kernel -> all-reduce of
// {df, ddf} -> update kernel, all enqueued, no host round trip per step), on the host (single-process sharding with
// the pinned-host reduction) and in the CPU tests; k_newton / k_newton_batch keep the loop form.
struct __attribute__((aligned(16))) NewtonState {
    double x1, x2, xacc;
    double rts, rts_old, xl, xh, dx, f, df, d2l, result;
    int32_t max_steps, j, neval, status;  // status: 0 ok, 2 non-finite derivative, 3 step limit
    int32_t done, _pad[3];
};

__host__ __device__ inline void newton_init(NewtonState &s, double xguess, double x1, double x2, double xacc,
                                            int max_steps) {
    s.x1 = x1; s.x2 = x2; s.xacc = xacc;
    s.rts = xguess;
    if (s.rts < x1) s.rts = x1;
    if (s.rts > x2) s.rts = x2;
    s.rts_old = s.rts; s.xl = x1; s.xh = x2; s.dx = 0.0; s.f = s.df = s.d2l = 0.0;
    s.result = s.rts;
    s.max_steps = max_steps; s.j = 0; s.neval = 0; s.status = 0; s.done = 0;
    s._pad[0] = s._pad[1] = s._pad[2] = 0;
}

__host__ __device__ inline bool newton_finite(double v) { return v == v && v - v == 0.0; }

__host__ __device__ inline void newton_update(NewtonState &s, double pdf, double pddf) {
    if (s.done) return;
    if (!newton_finite(pdf)) { pdf = 0.0; pddf = 0.0; }  // computeLikelihoodDerv, phylokernel.h:647-651
    const double f = -pdf, df = -pddf;                     // computeFuncDerv, phylotree.cpp:2135-2146
    s.f = f; s.df = df;
    s.neval++;
    if (s.neval == 1) {
        s.d2l = df;
        if (!newton_finite(f) || !newton_finite(df)) { s.status = 2; s.result = s.rts; s.done = 1; return; }
        if (df >= 0.0 && fabs(f) < s.xacc) { s.result = s.rts; s.done = 1; return; }
        if (f < 0.0) { s.xl = s.rts; s.xh = s.x2; } else { s.xh = s.rts; s.xl = s.x1; }
        s.dx = fabs(s.xh - s.xl);
        s.j = 1;
    } else {
        if (!newton_finite(f) || !newton_finite(df)) { s.status = 2; s.result = s.rts_old; s.done = 1; return; }
        if (df > 0.0 && fabs(f) < s.xacc) { s.d2l = df; s.result = s.rts; s.done = 1; return; }
        if (f < 0.0) s.xl = s.rts; else s.xh = s.rts;
        s.j++;
        if (s.j > s.max_steps) { s.status = 3; s.result = s.rts; s.done = 1; return; }
    }
    s.rts_old = s.rts;
    if ((df <= 0.0) || (((s.rts - s.xh) * df - f) * ((s.rts - s.xl) * df - f) >= 0.0)) {
        s.dx = 0.5 * (s.xh - s.xl);
        s.rts = s.xl + s.dx;
        s.d2l = df;
        if (s.xl == s.rts) { s.result = s.rts; s.done = 1; return; }
    } else {
        s.dx = f / df;
        const double temp = s.rts;
        s.rts -= s.dx;
        s.d2l = df;
        if (temp == s.rts) { s.result = s.rts; s.done = 1; return; }
    }
    if (fabs(s.dx) < s.xacc || s.j == s.max_steps) { s.result = s.rts_old; s.done = 1; return; }
}

}  // namespace iqhip

struct iqhip_engine {
    int device = 0;
    // planning-only engine (iqhip_debug_create_planner): no HIP call is ever made for it; "device" allocations are
    // distinct fake addresses that are never dereferenced.  It exists so that build_plan and the plan check below run
    // in CPU tests (tests/test_plan_check.py): the descriptors a kernel would receive are validated without a GPU.
    bool planner = false;
    uint64_t fake_next = 0x100000000000ull;
    bool check_plans = false;  // IQHIP_CHECK_PLAN=1 (always on for a planner): validate every DevOp before upload
    int n = 0, ncat = 0, ntaxa = 0;
    // Binary data (2 states; phylotreesse.cpp:262-276 binds <Vec2d, 2, 2>): embedded EXACTLY into the 4-state kernels --
    // the eigen-system is padded block-diagonally (U = diag(U2, I2), eigenvalues (l0, l1, 0, 0)), so components 2, 3 of
    // every vector are identically zero, and a missing character becomes the ambiguity set {0, 1} (DNA code 6 = "A or
    // C", whose tip vector is the caller's unknown row) instead of the kernels' "unknown = exactly 1.0 in every
    // component" rule, which would leak into the padding.  n stays 4 internally; the ABI speaks n_user = 2.
    bool embed2 = false;
    // The same embedding carries every state count the reference hands to its scalar kernel (3, 5 .. 19, 21 .. 63: morphological
    // / multi-state data, phylotreesse.cpp:281-309) on the next kernel size up (4 / 20 / 64): U = diag(U_m, I), zero
    // eigenvalues and zero tip components in the padding; internal state n = "missing" (the caller's unknown row), the
    // kernels' own unknown state n + 1 never occurs.  scalar_rule_all: such engines apply the scalar kernel's scaling rule
    // (IQHIP_OP_SCALAR_RULE, incl. its lh_max == 0 branch) at every node, as the reference does for them.
    bool scalar_rule_all = false;
    int n_user = 0;
    int64_t nptn = 0;      // caller-visible patterns
    int64_t nptn_pad = 0;  // padded to the tile size
    int64_t ntiles = 0;    // tiles of `tile` patterns
    int tile = 64;         // 64 (VALU path) or 16 (MFMA path)
    bool mfma = false;     // nstates 20 / 64: matrix-core path (kernels_mfma.hip)
    bool mfma_pipelined = false;  // (n, ncat) has a k_traverse_mfma2 instantiation (IQHIP_MFMA_V1=1 disables)
    int wg_size = 256;     // threads per workgroup of the traversal kernel (IQHIP_WG env)
    bool row_split = false; // 64 states, 1 category: one wave per 16 output rows of a tile (small alignments)
    bool cat_split = false; // 20 states, 4 categories: one wave per category of a tile (small alignments)
    bool top_cs2 = false;   // 20 states, 4 categories: the sequential top stage with two waves per tile (two categories each), IQHIP_TOP_CS2
    int lane_split = 1;    // 4-state traversal: lanes per pattern (2: each lane owns half of the categories)
    int lane_split_valu = 1;  // ... remembered while a 4-state engine runs a mixture on the matrix-core kernels
    bool mixed_top = true; // 64 states: mixed-role top stage (kernels_mfma.hip k_traverse_mfma_top64; IQHIP_MIXED_TOP)
    bool use_hold = true;  // 4-state traversal: park join operands in a second register set (IQHIP_HOLD)
    // 20-state pipelined kernel: the same idea with the parking place in LDS (10 KB per wave) -- a result that is the
    // streamed child of a later op of the same unit is kept there instead of being read back from memory (IQHIP_HOLD_LDS)
    bool hold_lds = true;
    int plan_nhold = 0;      // ops of the current plan whose left child is parked (diagnostics)
    int ablate = 0;        // IQHIP_ABLATE: timing-only host-side switches (results wrong when set)
    int lds_budget_bytes = 64 * 1024;  // per-workgroup LDS for the per-branch regions (IQHIP_LDS_KB)
    int plan_lds_doubles = 0;
    int plan_state_slots = 1;    // leaf-state LDS slots of the largest chunk (4-state path)
    bool plan_has_load = false;  // some op has two memory children (slow kernel instantiation)
    // 4-state kernel: a plan of at most kSmallPlanOps - 2 ops with no units travels in the kernel arguments instead of
    // being copied to d_ops first (the copy kernel and the dependent-launch gap behind it cost ~9 us per changed plan,
    // i.e. per branch of a branch-length sweep); IQHIP_SMALL_PLANS=0 switches it off
    bool plan_small = false;
    int plan_small_nops = 0;
    bool small_plans = true;
    // Staged plans (engine.hip, build_plan): independent subtrees ("units") run as their own workgroups in a
    // first launch, the ops above them ("top") in a second one.  Segment table on the device, after the
    // sentinel descriptors: {top_begin, top_nops, unit1_begin, unit1_nops, ...}
    int plan_nunits = 0;
    std::vector<int> plan_stage_units;  // units per stage, in launch order
    int plan_top_nops = 0;
    std::vector<int> last_segs;   // explicit segment sizes the cached descriptors were built with
    int plan_table_off = 0;      // DevOp index where the table starts
    bool plan_units_have_load = false;
    int split_target = -1;       // IQHIP_SPLIT: -1 auto, 0 never, n > 0: unit size
    iqhip::Slab dummy;           // valid target of unconditional prefetches  // LDS region size (doubles) of the largest chunk of the current plan
    int block = 0;         // n*ncat
    int state_unknown = -1;
    bool model_set = false, aln_set = false, theta_valid = false;
    // +ASC (phylokernel.h:868-909,655-725,1124-1187): the last n_unobs patterns are the unobserved
    // constant patterns; asc_nsites = aln->getNSite()
    int64_t n_unobs = 0;
    // pattern-sharded runs: the correction is active on EVERY shard / rank (asc_active), while the unobserved patterns --
    // appended at the end of the alignment -- sit on the last one(s) only (n_unobs = this engine's share, possibly 0);
    // the sums prob_const / df_const / ddf_const travel with the result vector through the all-reduce
    bool asc_active = false;
    double asc_nsites = 0.0;
    double pattern_lh_shift = 0.0;        // log(1 - prob_const) of the last lnL evaluation
    const int16_t *theta_a_sc = nullptr;  // scale counters of the ends theta was built from
    const int16_t *theta_b_sc = nullptr;

    hipStream_t stream = nullptr;
    bool own_stream = false;

    // alignment side
    uint8_t *d_states = nullptr;  // [ntaxa][nptn_pad]
    double *d_freq = nullptr, *d_invar = nullptr;
    // model side
    double *d_eval = nullptr, *d_evec = nullptr, *d_inv_evec = nullptr;
    double *d_rates = nullptr, *d_props = nullptr, *d_tip = nullptr;
    std::vector<double> h_eval, h_rates, h_props;
    // per-call buffers
    iqhip::DevOp *d_ops = nullptr;
    int ops_cap = 0;
    double *d_slab = nullptr;   // wave partials [nvals][nwaves]
    int64_t slab_cap = 0;
    unsigned int *d_fold_ticket = nullptr;  // folded reduction (FoldArgs): ticket + per-row flags, zero between launches
    int *d_fold_flags = nullptr;
    bool fold_reduce = false;               // IQHIP_FOLD=1: the last kernel of a submission sums the slab itself (measured
                                            // slower than the k_reduce launch on MI355X; off by default)
    // host polling of the result (IQHIP_POLL): k_reduce's last block stores a sequence number to mapped host memory
    bool poll_result = true, poll_pending = false;
    unsigned long long result_seq = 0;
    volatile unsigned long long *h_done = nullptr;
    unsigned long long *d_done = nullptr;
    double *d_theta = nullptr, *d_pattern_lh = nullptr;
    // K2 tables of the leaf children (matrix-core pipelined kernels, kernels_mfma.hip k_leaf_tables).  A table depends
    // on (model, pendant branch length) only, so slot t < ntaxa belongs to taxon t and is rebuilt only when that
    // length or the model changed; a second length of one taxon inside one submission (batched NNI candidates)
    // takes an overflow slot >= ntaxa.
    double *d_leaf_tab = nullptr;
    size_t leaf_tab_slots = 0;            // capacity in tables
    std::vector<double> tab_len;          // per slot: branch length the table was built for (NaN: none)
    uint64_t model_version = 1, tab_model_version = 0;
    std::vector<iqhip::TabJob> plan_tab_jobs;  // the current plan's tables: [0, plan_tab_dirty) need (re)building
    // cherry tables (20 states x 4 categories, DevOp::cherry): slot = pair of taxa; `pair` is a small engine of our own on
    // the same stream whose pseudo-alignment lists every pair of states -- a cherry's table is that engine's ordinary
    // node update for the two pendant lengths, moved into register order by k_cherry_transpose
    struct CherrySlot {
        double len_l = -1.0, len_r = -1.0;
        uint64_t model_version = 0;   // model the table was built for (0: never built)
        uint64_t stamp = 0;           // last plan that used the slot
    };
    bool cherry_on = true;                 // IQHIP_CHERRY_TABLES=0 switches the tables off
    iqhip_engine *pair = nullptr;
    int cherry_s2 = 0, cherry_npairs = 0;  // STATE_UNKNOWN + 1; its square padded to whole 64-pattern groups
    double *d_cherry_tab = nullptr;
    size_t cherry_cap = 0;                 // slots allocated
    std::unordered_map<uint64_t, int> cherry_slot_of;   // (taxon_l << 32 | taxon_r) -> slot
    std::vector<CherrySlot> cherry_slots;
    std::vector<int> plan_cherry_jobs;     // slots the current plan needs (re)built before its traversal
    uint64_t cherry_stamp = 0, plan_cherry_model = 0;   // (model version the current plan's tables were scheduled for)
    bool plan_uses_cherry = false;
    int64_t cherry_built_total = 0, cherry_ops_total = 0;   // tables built / node updates answered from a table so far
    bool cherry_model_synced = false;      // the pair engine has the model of this engine's last set_model call
    int plan_tab_dirty = 0;
    int plan_jobs_off = 0;                // DevOp index where the device copy of the job list starts
    int plan_nleaf_tabs = 0;              // tables the current plan uses (0: kernel variant without tables)
    bool leaf_tables = false;       // IQHIP_LEAF_TABLES (default on for the pipelined matrix-core kernels)
    // Mixture models (phylokernelmixture.h, phylokernelmixrate.h): the ncat categories are (class, rate)
    // components; category c uses eigen-system cat_class[c].  Per-category expansions for the kernels
    // that are generic in (n, ncat): evalc[c][i], tipc[state][c][i]; per-class MFMA A images for the
    // mixture traversal kernel.  A plain model is the one-class case.
    int nclass = 1;
    double *d_evalc = nullptr;   // [ncat][n]
    // 20 / 64 states, class 0: the A-operand fragments of U and U^-1 in the exact order of the pipelined kernels' LDS
    // image ([U 16-row tiles][U^-1 tiles][U tail][U^-1 tail], element (m, s, lane) = row 16m + (lane & 15), column
    // 4s + (lane >> 4)), so that a workgroup stages it with 16-byte coalesced copies: gathered element by element
    // from evec / inv_evec the 64 KB of 64 states cost every workgroup 10-15 us before its first matrix instruction
    // (wave trace, r02)
    double *d_aimg = nullptr;
    int aimg_doubles = 0;
    double *d_tipc = nullptr;    // [state_unknown+1][ncat][n]
    int *d_cls = nullptr;        // [ncat]
    double *d_img = nullptr;     // mixture A images: mix20 layout, then the generic kernel's (engine.hip)
    size_t img_generic_off = 0, img_cap = 0;
    double *d_model = nullptr;   // the block all model arrays below point into (set_model_common)
    size_t model_cap = 0;
    bool mfma_pipelined_ok = false;  // (n, ncat) has a pipelined instantiation (used when nclass == 1)
    // UFBoot / RELL (kernels_rell.hip): scaled per-pattern lnL and the bootstrap sample matrix
    double *d_ptn_scaled = nullptr;
    float *d_boot = nullptr;  // [nboot][nptn_pad], zero padded
    int nboot = 0;
    double *d_result_own = nullptr, *d_result = nullptr;
    double *d_newton_partials = nullptr;   // [2][num_cus][2]
    unsigned int *d_newton_barrier = nullptr;  // [2], used alternately by consecutive k_newton launches
    unsigned int newton_launches = 0;
    // k_newton's exchange of the workgroup partial sums without an arrival counter: every (evaluation, workgroup) has a
    // slot of its own, {df, ddf}, that holds a sentinel until its owner posts; readers spin on the slots themselves.
    // Two buffers alternate between launches; a launch resets the other buffer's slots of its workgroups.
    double *d_newton_posts = nullptr;      // [2][kNewtonPostEpochs][num_cus][2]
    unsigned int newton_post_launches = 0;
    bool newton_posts = true;              // IQHIP_NEWTON_POSTS=0: the arrival-counter barrier of round 1
    // the same for k_newton_batch: [2 launch parities][batch_posts_cap]; a launch lays its slots out as
    // [task][evaluation][workgroup][2] and resets what the previous launch of the other parity used
    double *d_batch_posts = nullptr;
    size_t batch_posts_cap = 0;            // doubles per parity
    size_t batch_posts_used[2] = {0, 0};
    unsigned int batch_post_launches = 0;
    // batched branch optimisation (iqhip_optimize_branch_batch): per-task theta buffers, partial sums, arrival
    // counters (two sets, alternating per launch), results and the task descriptors
    double *d_theta_batch = nullptr, *d_batch_partials = nullptr, *d_batch_out = nullptr;
    unsigned int *d_batch_barriers = nullptr;
    void *d_batch_tasks = nullptr;
    size_t theta_batch_cap = 0;
    int batch_cap = 0;
    unsigned int batch_launches = 0;
    double *d_sweep_len = nullptr;   // iqhip_optimize_sweep: the accepted length of every step, read by later steps' node updates
    int sweep_len_cap = 0;
    // ... persistent form (4 states): descriptors of all steps (pinned staging + device copy) and the exchange slots
    char *h_sweep_desc = nullptr, *d_sweep_desc = nullptr;
    char *h_plan_arena = nullptr;     // pinned slices for the plan uploads of a per-step sweep (build_plan)
    size_t plan_arena_cap = 0, plan_arena_used = 0;
    bool plan_arena_on = false;
    size_t sweep_desc_cap = 0;
    double *d_sweep_posts = nullptr;
    size_t sweep_posts_cap = 0;
    int num_cus = 256;
    int result_cap = 0;
    // ---- collectives (comm.hip).  comm != nullptr: this engine is one rank of a pattern-sharded run; every
    // host-visible sum is all-reduced (ncclAllReduce, SUM, f64, in place in the device result vector, on the engine's
    // stream) before it is read back.  RCCL is loaded lazily (dlopen) the first time a communicator is made.
    void *comm = nullptr;            // ncclComm_t
    int comm_nranks = 1, comm_rank = 0;
    double *d_result_dev = nullptr;  // device-memory result vector of a comm engine (the default one is mapped host memory)
    iqhip::NewtonState *d_nstate = nullptr, *h_nstate = nullptr;  // Newton state machine: device copy / pinned host copy
    iqhip::NewtonState *d_bstates = nullptr;                      // batched chain: one state machine per task
    int bstates_cap = 0;
    // ---- single-process sharding (sharded.hip): this object owns no device memory, it fronts `shards`
    // (pattern ranges [shard_first[g], shard_first[g+1]) on the devices of iqhip_create_sharded)
    std::vector<iqhip_engine *> shards;
    std::vector<int64_t> shard_first;
    int reduce_mode = 0;             // IQHIP_REDUCE_RCCL / IQHIP_REDUCE_HOST
    // pinned host staging
    iqhip::DevOp *h_ops = nullptr;
    double *h_result = nullptr;
    std::vector<char> uploaded_plan;  // bytes of the descriptors currently in d_ops
    // the caller's op list the current descriptors were built from: an identical list (same keys,
    // leaves, lengths) with an unchanged key map needs no rebuilding at all
    std::vector<char> last_ops_in;
    uint64_t keymap_version = 1, last_plan_version = 0;
    int last_plan_dst = -1;
    hipEvent_t staging_free = nullptr;
    bool staging_busy = false;

    // key -> slab
    std::unordered_map<uint64_t, int> key2slab;
    std::vector<iqhip::Slab> slabs;
    std::vector<int> free_slabs;

    // timing of the dominant kernel
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> tev;
    size_t tev_used = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> cev;  // ... around the engine's own all-reduces (comm.hip)
    size_t cev_used = 0;
    int64_t tev_launches = 0;  // traversal-kernel launches inside the recorded brackets
    int last_nops = 0;
    bool last_has_root = false, last_root_loads_b = false;  // ... of the last submission (iqhip_timing_plan_bytes)
    // the last submission's reduction was folded into its kernel and the host polls: the kernel wrote only the rows it
    // summed plus, behind them, the number of flagged sum_scale rows; read_result zero-fills the rest when that is 0
    int folded_rows = -1;
};

namespace iqhip {

// ---------------------------------------------------------------------------------------------------------------
// Reduction folded into the producing kernel ("last workgroup sums"): every wave leaves its partial sums in the slab
// [row][wave]; rows: 0 = lnL (or df), 1 = prob_const (or ddf), 2+k = sum_scale of node update k.  sum_scale rows are
// zero unless some pattern was rescaled at that node, so a wave that did rescale also raises flags[row]; the last
// workgroup to finish sums row 0/1 and the flagged rows in k_reduce's order (same bits) and writes 0.0 for the rest.
// Hand-off between workgroups (MI355X_MICROARCH.md, valid forms): partials are agent-scope (sc1, write-through)
// stores, each wave drains them (s_waitcnt vmcnt(0)) before the workgroup barrier, one lane then takes a ticket with
// an agent-scope atomic; the workgroup that draws the last ticket reads with agent-scope (sc1) loads.
// ---------------------------------------------------------------------------------------------------------------
struct FoldArgs {
    double *slab;          // [rows][nwaves]
    double *result;        // result[row]
    unsigned int *ticket;  // zero between launches
    int *flags;            // [rows], zero between launches
    int nwaves;
    int nrows_scale;       // sum_scale rows (2 .. 2+nrows_scale)
    int root_rows;         // 0: none; 2: rows 0 and 1 hold the root-branch sums
    int enabled;
    // non-NULL: result is mapped host memory; the last workgroup publishes this sequence number for the polling host
    volatile unsigned long long *done;
    unsigned long long seq;
};

#if defined(__HIPCC__)
__device__ __forceinline__ void fold_store(double *p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void fold_flag(const FoldArgs &F, int row) {
    __hip_atomic_fetch_or(&F.flags[row], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// all live threads of a 256-thread workgroup call this once, after their last partial has been stored
__device__ inline void fold_tail(const FoldArgs &F) {
    __shared__ double s_f[256];
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int t = __hip_atomic_fetch_add(F.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == gridDim.x - 1) ? 1 : 0;
    }
    __syncthreads();
    if (!s_last) return;
    // Only wave 0 is certain to be alive (the workgroup's trailing waves have exited when the tile count is not a
    // multiple of four), so its 64 lanes do the work; the other live waves only keep the barriers company.
    __shared__ int s_rows[256];
    __shared__ int s_n;
    const int nrows = 2 + F.nrows_scale;
    // one row: k_reduce's order -- 256 strided running sums (lane = 4 of them), then the LDS tree
    auto sum_row = [&](int row) {
        const double *r = F.slab + (size_t)row * F.nwaves;
        if (threadIdx.x < 64) {
            // the four strided running sums of this lane, 8 terms of each at a time: all 32 loads are in flight
            // together (one round trip), the additions keep k_reduce's order
            double acc[4] = {0.0, 0.0, 0.0, 0.0};
            for (int base = 0; base < F.nwaves; base += 8 * 256) {
                double v[4][8];
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        const int i = base + q * 256 + (int)threadIdx.x + 64 * j;
                        v[j][q] = i < F.nwaves ? __hip_atomic_load(&r[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
                    }
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        const int i = base + q * 256 + (int)threadIdx.x + 64 * j;
                        if (i < F.nwaves) acc[j] += v[j][q];
                    }
            }
#pragma unroll
            for (int j = 0; j < 4; j++) s_f[(int)threadIdx.x + 64 * j] = acc[j];
        }
        __syncthreads();
#pragma unroll
        for (int o = 128; o > 0; o >>= 1) {
            if (threadIdx.x < 64)
                for (int v = (int)threadIdx.x; v < o; v += 64) s_f[v] += s_f[v + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) F.result[row] = s_f[0];
        __syncthreads();
    };
    if (F.root_rows) {
        sum_row(0);
        if (F.root_rows > 1) sum_row(1);
        else if (threadIdx.x == 0) F.result[1] = 0.0;   // (no +ASC: the prob_const row is zero)
    }
    // sum_scale rows, 256 at a time: the lanes look at the flags side by side.  With a polling host (F.done) the
    // unflagged rows are not written at all -- 48 stores over PCIe for a 50-taxon plan -- only their count is (the host
    // fills in the zeros, read_result); a device-resident result vector (collectives) gets every row.
    int nflagged = 0;
    for (int base = 2; base < nrows; base += 256) {
        if (threadIdx.x == 0) s_n = 0;
        __syncthreads();
        if (threadIdx.x < 64) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int row = base + (int)threadIdx.x + 64 * j;
                if (row < nrows) {
                    if (__hip_atomic_load(&F.flags[row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                        s_rows[atomicAdd(&s_n, 1)] = row;
                        __hip_atomic_store(&F.flags[row], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else if (!F.done) {
                        F.result[row] = 0.0;
                    }
                }
            }
        }
        __syncthreads();
        const int n = s_n;
        nflagged += n;
        if (n > 0 && F.done) {   // some rows are non-zero: the host will read all of them
            if (threadIdx.x < 64)
                for (int j = 0; j < 4; j++) {
                    const int row = base + (int)threadIdx.x + 64 * j;
                    if (row < nrows) {
                        bool flagged = false;
                        for (int q = 0; q < n; q++) flagged = flagged || s_rows[q] == row;
                        if (!flagged) F.result[row] = 0.0;
                    }
                }
        }
        for (int q = 0; q < n; q++) sum_row(s_rows[q]);
    }
    if (F.done && threadIdx.x == 0) F.result[nrows] = (double)nflagged;
    if (threadIdx.x == 0) {
        __hip_atomic_store(F.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (F.done) {
            __threadfence_system();
            *F.done = F.seq;
        }
    }
}
#endif

// engine.hip: records the calling thread's error text (iqhip_last_error) and returns `code`
int set_error(int code, const std::string &msg);

// comm.hip -- RCCL, loaded lazily.  comm_allreduce: in-place SUM/f64 all-reduce of the first n doubles of the
// engine's device result vector on its stream (no-op without a communicator).  comm_group_allreduce: the same for
// the shards of a single-process sharded engine, inside one ncclGroupStart/End.  comm_init_all: ncclCommInitAll
// over the shards' devices.  All return an IQHIP status.
int comm_allreduce(iqhip_engine *e, int n);
int comm_group_allreduce(const std::vector<iqhip_engine *> &shards, int n);
int comm_init_all(const std::vector<iqhip_engine *> &shards);
void comm_destroy(iqhip_engine *e);
int comm_use_device_result(iqhip_engine *e);  // switch the engine to a device-memory result vector

// engine.hip internals the sharded front drives its shards with
int eng_read_result(iqhip_engine *e, int ndoubles);   // D2H of the result vector (if it is device memory) + stream sync
// node updates, asynchronous; segs: op counts of independent groups (each runs on its own workgroups)
int eng_submit_updates(iqhip_engine *e, const iqhip_node_op *ops, int nops, const std::vector<int> *segs);
int eng_repair_lnl(iqhip_engine *e, double *lnl);     // phylokernel.h:848-866 on this engine's _pattern_lh -> its own sum
int newton_state_alloc(iqhip_engine *e);
int newton_state_read(iqhip_engine *e);               // -> e->h_nstate (syncs the stream)
int eng_newton_begin(iqhip_engine *e, double xguess, double x1, double x2, double xacc, int max_steps);
int eng_newton_eval_enqueue(iqhip_engine *e);         // derivative kernel at state->rts + k_reduce -> result[0..1]
int eng_newton_update_enqueue(iqhip_engine *e);       // state machine step from result[0..1]
// batched chain (iqhip_optimize_branch_batch on sharded engines): m tasks side by side, ONE reduction of 2m rows and
// hence one all-reduce of 2m doubles per Newton step.  prepare: theta of every task into its slot + the initial states
int eng_batch_prepare(iqhip_engine *e, const iqhip_branch_task *tasks, int m, const NewtonState *init);
int eng_batch_eval_enqueue(iqhip_engine *e, int m);   // derivative kernels at states[t].rts + k_reduce -> result[0..2m)
int eng_batch_update_enqueue(iqhip_engine *e, int m); // state machines step from result[0..2m)
int eng_batch_lnl_enqueue(iqhip_engine *e, int m);    // lnL at states[t].result -> result[2t]
int eng_batch_states_read(iqhip_engine *e, int m, NewtonState *out);        // (syncs the stream)
int eng_batch_states_write(iqhip_engine *e, int m, const NewtonState *in);

// sharded.hip -- the single-process multi-device front (iqhip_create_sharded); every public entry point of
// engine.hip forwards here when e->shards is non-empty
namespace sharded {
void destroy(iqhip_engine *p);
int reserve(iqhip_engine *p, int nvectors);
int release(iqhip_engine *p, uint64_t key);
int rekey(iqhip_engine *p, uint64_t old_key, uint64_t new_key);
int set_alignment(iqhip_engine *p, const uint8_t *states, const double *ptn_freq, const double *ptn_invar);
int set_ascertainment(iqhip_engine *p, int64_t n_unobserved, double nsites);
int set_ptn_array(iqhip_engine *p, const double *v, bool invar);
int set_model(iqhip_engine *p, int nclass, const int32_t *cat_class, const double *eval, const double *evec,
              const double *inv_evec, const double *rates, const double *props, int state_unknown, const double *tip);
int traverse(iqhip_engine *p, const iqhip_node_op *ops, int nops, bool has_root, iqhip_branch_end a, iqhip_branch_end b,
             double len, double *sum_scale, double *lnl);
int compute_theta(iqhip_engine *p, iqhip_branch_end a, iqhip_branch_end b);
int derv(iqhip_engine *p, double len, double *df, double *ddf);
int lnl_from_theta(iqhip_engine *p, double len, double *lnl);
int optimize_branch(iqhip_engine *p, const iqhip_node_op *ops, int nops, bool build_theta, iqhip_branch_end a,
                    iqhip_branch_end b, double xguess, double x1, double x2, double xacc, int max_steps,
                    double *sum_scale, double *optx, double *d2l, int *nsteps);
int optimize_branch_batch(iqhip_engine *p, const iqhip_branch_task *tasks, int ntasks, double *sum_scale,
                          iqhip_branch_result *results);
int fetch_scale_num(iqhip_engine *p, uint64_t key, int16_t *out);
int fetch_pattern_lh(iqhip_engine *p, double *out, int kind, iqhip_branch_end a, iqhip_branch_end b);  // kind 0 plain, 1 scaled
int fetch_vec(iqhip_engine *p, uint64_t key, bool theta, double *out);
int pattern_lh_cat(iqhip_engine *p, double len, double *out);
int upload_partial(iqhip_engine *p, uint64_t key, const double *partial_lh, const int16_t *scale_num);
int set_boot_samples(iqhip_engine *p, const float *samples, int nsamples);
int rell(iqhip_engine *p, iqhip_branch_end a, iqhip_branch_end b, double *out);
int synchronize(iqhip_engine *p);
}  // namespace sharded

// kernels_valu4.hip
// seg_table: device ints {begin, nops} x nsegs; every segment runs on its own set of workgroups
// fold_rows >= 0: this is the submission's last launch and it sums the slab itself (rows 2 .. 2+fold_rows, plus 0/1
// with a root branch) instead of a k_reduce launch
hipError_t launch_traverse4(iqhip_engine *e, const int *seg_table, int nsegs, bool has_load, const DevBranch *root, int nwaves,
                            int fold_rows = -1);
// BatchChain: the batched form of the enqueued Newton chain (iqhip_optimize_branch_batch on sharded engines) -- task t
// (blockIdx.y) has theta at theta + t * theta_stride, its state machine at states[t] and slab rows 2t, 2t + 1
struct BatchChain {
    const double *theta;
    size_t theta_stride;
    const NewtonState *states;
    int ntasks;
};
hipError_t launch_theta4(iqhip_engine *e, const DevBranch &br, double *theta_out = nullptr);
hipError_t launch_derv4(iqhip_engine *e, double len, int nwaves, const NewtonState *st = nullptr, const BatchChain *bc = nullptr);
hipError_t launch_lnl_theta4(iqhip_engine *e, double len, int nwaves, const BatchChain *bc = nullptr);
hipError_t launch_reduce(iqhip_engine *e, int first_row, int nrows, int nwaves);

// kernels_newton.hip
// build_from != nullptr: the first evaluation also builds theta from that branch; reduce_rows > 0: the
// sum_scale rows [2, 2 + reduce_rows) of the slab are summed into the result vector first
// sweep != nullptr (one step of iqhip_optimize_sweep): the accepted length also goes to *len_out (device memory read by
// later steps), the sum_scale rows go to rows_base[0 .. reduce_rows) instead of result[2 ..], the diverged-solve rule is
// applied on the device above diverge_x, and only the step with `publish` signals the polling host
struct NewtonSweepStep {
    double *len_out;
    double *rows_base;
    double diverge_x;
    bool publish;
};
hipError_t launch_newton(iqhip_engine *e, double xguess, double x1, double x2, double xacc, int max_steps,
                         double *out, const DevBranch *build_from = nullptr, int reduce_rows = 0,
                         int reduce_nwaves = 0, const NewtonSweepStep *sweep = nullptr);

// Newton as a chain of enqueued steps (kernels_newton.hip): state init, derivative evaluation at state->rts
// (skipped once state->done), state update from result[0..1]
hipError_t launch_newton_state_init(iqhip_engine *e, double xguess, double x1, double x2, double xacc, int max_steps);
hipError_t launch_derv_at_state(iqhip_engine *e, int nwaves);
hipError_t launch_newton_state_update(iqhip_engine *e);   // (+ASC: result[2..4] and asc_nsites enter the update)
hipError_t launch_newton_state_update_batch(iqhip_engine *e, NewtonState *states, int ntasks);   // from result[2t], result[2t+1]

// kernels_sweep.hip: a whole sweep of a 4-state engine in one launch; posts: [2][kNewtonPostEpochs][grid][2] all-ones
int sweep4_grid(const iqhip_engine *e);
int sweep4_waves(const iqhip_engine *e);   // waves per workgroup (8: one 512-thread workgroup for <= 8 tiles)
hipError_t launch_sweep4(iqhip_engine *e, const SweepOp *d_ops, const SweepStep *d_steps, int nsteps, double x1, double x2,
                         double xacc, int max_steps, double diverge_x, double *posts, double *out);

// kernels_rell.hip
hipError_t launch_pattern_lh_scaled(iqhip_engine *e, const int16_t *sc_a, const int16_t *sc_b, double *out);
hipError_t launch_rell(iqhip_engine *e, double *out);
hipError_t launch_pattern_lh_cat(iqhip_engine *e, double len, double *out);

// batched branch optimisation (k_newton_batch); d_tasks: device array of NewtonTask (kernels_newton.hip)
hipError_t launch_newton_batch(iqhip_engine *e, const void *d_tasks, int ntasks, int G, double *theta_base,
                               size_t theta_stride, double *partials, unsigned int *barriers, unsigned int *barriers_next,
                               double *out,
                               double *posts = nullptr, double *posts_other = nullptr, size_t posts_other_used = 0,
                               int post_epochs = 0);
size_t newton_task_bytes();
void newton_task_fill(void *dst, const DevBranch &br, double xguess, double x1, double x2, double xacc, int max_steps);

// kernels_mfma.hip (nstates 20 / 64)
hipError_t launch_traverse_mfma(iqhip_engine *e, const int *seg_table, int nsegs, int nwaves, bool top_stage = false);
hipError_t launch_leaf_tables(iqhip_engine *e, const TabJob *d_jobs, int njobs);
// cherry tables: src[i] (tile layout, npairs patterns) -> dst[i] (register order of the traversal kernel), host pointer arrays
hipError_t launch_cherry_transpose(iqhip_engine *e, const double *const *src, double *const *dst, int n, int npairs);
size_t leaf_table_doubles(const iqhip_engine *e);          // doubles per (leaf child) table: ncat * state_unknown * n
int mfma2_fixed_lds_doubles(int n);
// mode 0: branch lnL, 1: theta, 2: df/ddf from theta, 3: lnL from theta
// theta_out (mode 1): write theta there instead of the engine's buffer; bc (modes 2, 3): batched chain, see BatchChain
hipError_t launch_stream_mfma(iqhip_engine *e, int mode, const DevBranch *br, double len, int nwaves,
                              const NewtonState *st = nullptr, int fold_rows = -1, double *theta_out = nullptr,
                              const BatchChain *bc = nullptr);

}  // namespace iqhip
