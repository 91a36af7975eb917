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
};

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

struct Slab {
    double *plh = nullptr;
    int16_t *sc = nullptr;
};

}  // namespace iqhip

struct iqhip_engine {
    int device = 0;
    int n = 0, ncat = 0, ntaxa = 0;
    int64_t nptn = 0;      // caller-visible patterns
    int64_t nptn_pad = 0;  // padded to the tile size
    int64_t ntiles = 0;    // tiles of `tile` patterns
    int tile = 64;         // 64 (VALU path) or 16 (MFMA path)
    bool mfma = false;     // nstates 20 / 64: matrix-core path (kernels_mfma.hip)
    bool mfma_pipelined = false;  // (n, ncat) has a k_traverse_mfma2 instantiation (IQHIP_MFMA_V1=1 disables)
    int wg_size = 256;     // threads per workgroup of the traversal kernel (IQHIP_WG env)
    bool row_split = false; // 64 states, 1 category: one wave per 16 output rows of a tile (small alignments)
    bool cat_split = false; // 20 states, 4 categories: one wave per category of a tile (small alignments)
    int lane_split = 1;    // 4-state traversal: lanes per pattern (2: each lane owns half of the categories)
    int ablate = 0;        // IQHIP_ABLATE: timing-only host-side switches (results wrong when set)
    int lds_budget_bytes = 64 * 1024;  // per-workgroup LDS for the per-branch regions (IQHIP_LDS_KB)
    int plan_lds_doubles = 0;
    int plan_state_slots = 1;    // leaf-state LDS slots of the largest chunk (4-state path)
    bool plan_has_load = false;  // some op has two memory children (slow kernel instantiation)
    // Staged plans (engine.hip, build_plan): independent subtrees ("units") run as their own workgroups in a
    // first launch, the ops above them ("top") in a second one.  Segment table on the device, after the
    // sentinel descriptors: {top_begin, top_nops, unit1_begin, unit1_nops, ...}
    int plan_nunits = 0;
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
    double *d_theta = nullptr, *d_pattern_lh = nullptr;
    // Mixture models (phylokernelmixture.h, phylokernelmixrate.h): the ncat categories are (class, rate)
    // components; category c uses eigen-system cat_class[c].  Per-category expansions for the kernels
    // that are generic in (n, ncat): evalc[c][i], tipc[state][c][i]; per-class MFMA A images for the
    // mixture traversal kernel.  A plain model is the one-class case.
    int nclass = 1;
    double *d_evalc = nullptr;   // [ncat][n]
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
    // batched branch optimisation (iqhip_optimize_branch_batch): per-task theta buffers, partial sums, arrival
    // counters (two sets, alternating per launch), results and the task descriptors
    double *d_theta_batch = nullptr, *d_batch_partials = nullptr, *d_batch_out = nullptr;
    unsigned int *d_batch_barriers = nullptr;
    void *d_batch_tasks = nullptr;
    size_t theta_batch_cap = 0;
    int batch_cap = 0;
    unsigned int batch_launches = 0;
    int num_cus = 256;
    int result_cap = 0;
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
    int64_t tev_launches = 0;  // traversal-kernel launches inside the recorded brackets
    int last_nops = 0;
};

namespace iqhip {

// kernels_valu4.hip
// seg_table: device ints {begin, nops} x nsegs; every segment runs on its own set of workgroups
hipError_t launch_traverse4(iqhip_engine *e, const int *seg_table, int nsegs, bool has_load, const DevBranch *root, int nwaves);
hipError_t launch_theta4(iqhip_engine *e, const DevBranch &br);
hipError_t launch_derv4(iqhip_engine *e, double len, int nwaves);
hipError_t launch_lnl_theta4(iqhip_engine *e, double len, int nwaves);
hipError_t launch_reduce(iqhip_engine *e, int first_row, int nrows, int nwaves);

// kernels_newton.hip
// build_from != nullptr: the first evaluation also builds theta from that branch; reduce_rows > 0: the
// sum_scale rows [2, 2 + reduce_rows) of the slab are summed into the result vector first
hipError_t launch_newton(iqhip_engine *e, double xguess, double x1, double x2, double xacc, int max_steps,
                         double *out, const DevBranch *build_from = nullptr, int reduce_rows = 0,
                         int reduce_nwaves = 0);

// kernels_rell.hip
hipError_t launch_pattern_lh_scaled(iqhip_engine *e, const int16_t *sc_a, const int16_t *sc_b, double *out);
hipError_t launch_rell(iqhip_engine *e, double *out);
hipError_t launch_pattern_lh_cat(iqhip_engine *e, double len, double *out);

// batched branch optimisation (k_newton_batch); d_tasks: device array of NewtonTask (kernels_newton.hip)
hipError_t launch_newton_batch(iqhip_engine *e, const void *d_tasks, int ntasks, int G, double *theta_base,
                               size_t theta_stride, double *partials, unsigned int *barriers, unsigned int *barriers_next,
                               double *out);
size_t newton_task_bytes();
void newton_task_fill(void *dst, const DevBranch &br, double xguess, double x1, double x2, double xacc, int max_steps);

// kernels_mfma.hip (nstates 20 / 64)
hipError_t launch_traverse_mfma(iqhip_engine *e, const int *seg_table, int nsegs, int nwaves);
int mfma2_fixed_lds_doubles(int n);
// mode 0: branch lnL, 1: theta, 2: df/ddf from theta, 3: lnL from theta
hipError_t launch_stream_mfma(iqhip_engine *e, int mode, const DevBranch *br, double len, int nwaves);

}  // namespace iqhip
