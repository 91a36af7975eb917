/*
 * iqhip.h -- C ABI of the MI355X (gfx950) likelihood engine `libiqhip.so`.
 *
 * Drop-in boundary for ONE path of IQ-TREE 1.4.3: the Felsenstein-pruning likelihood
 * kernels that sit behind PhyloTree's four member-function pointers
 *   computePartialLikelihoodPointer   phylotree.h:658-659   (kernel: phylokernel.h:70-483)
 *   computeLikelihoodBranchPointer    phylotree.h:697-698   (kernel: phylokernel.h:733-1020)
 *   computeLikelihoodFromBufferPointer phylotree.h:742-743  (kernel: phylokernel.h:1022-1192)
 *   computeLikelihoodDervPointer      phylotree.h:979-980   (kernel: phylokernel.h:485-730)
 * selected by PhyloTree::setLikelihoodKernel (phylotreesse.cpp:60-311).  The reference-side
 * adapter (four PhyloTree member functions that do the recursion / flag handling and call
 * this ABI) is shown in INTEGRATION.md and integration/phylotree_hip.cpp.
 *
 * Conventions
 *   - plain C, POD arguments, every call returns an int status (IQHIP_OK == 0); the text of
 *     the last error of the calling thread is available from iqhip_last_error().  The
 *     reference has no error codes on this path (outError()/assert, tools.cpp:99-106); the
 *     adapter turns a non-zero status into outError().
 *   - one engine == one PhyloTree on one GPU.  No process-global state: engines of
 *     different trees may be driven from different host threads (phylosupertree.cpp:970).
 *   - all vectors stay resident in HBM.  A partial-likelihood vector is addressed by an
 *     opaque 64-bit KEY chosen by the caller -- the adapter uses the value of the host
 *     pointer PhyloNeighbor::partial_lh (phylonode.h:112), which the tree search re-points
 *     freely (phylotree.cpp:2921-2922, phylokernel.h:127-143); the engine maps key -> device
 *     slab lazily.  The companion scale_num array (phylonode.h:122, `short` per pattern)
 *     lives with the same key.
 *   - host-layout arrays use the reference's layout: partial_lh[ptn*block + c*nstates + i],
 *     evec[x*n+i] = U[x][i], inv_evec[i*n+x] = U^-1[i][x]; the device layout
 *     is private and converted by the fetch/upload calls.
 *   - patterns shard over GPUs (every pattern is independent; only scalar sums cross).  Two forms,
 *     both with ONE collective per evaluation -- ncclAllReduce (SUM, f64) over RCCL of the few result doubles:
 *       one process, several GPUs (what the single-process reference needs): iqhip_create_sharded returns an
 *         engine that fronts one shard per device; every call below works on it unchanged;
 *       one process per GPU (MPI / torch.distributed programs): each rank creates its own engine on its pattern
 *         range and joins a communicator with iqhip_comm_unique_id / iqhip_comm_init_rank; from then on every
 *         synchronous call all-reduces its result in place on the device before the host reads it, so all ranks
 *         return identical values.
 */
#ifndef IQHIP_H_
#define IQHIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IQHIP_ABI_VERSION 2

enum {
    IQHIP_OK = 0,
    IQHIP_ERR_NO_DEVICE = 1,   /* no HIP device / runtime failure at create */
    IQHIP_ERR_INVALID = 2,     /* bad argument (shape, key, leaf id, order of calls) */
    IQHIP_ERR_UNSUPPORTED = 3, /* nstates outside 2..64, too many categories, ... */
    IQHIP_ERR_HIP = 4,         /* a HIP runtime call failed; see iqhip_last_error() */
    IQHIP_ERR_NOMEM = 5
};

typedef struct iqhip_engine iqhip_engine;

/* One internal-node update = one call of computePartialLikelihoodEigenSIMD's pattern loop
 * (phylokernel.h:183-479).  A child is a leaf when *_leaf >= 0 (taxon id = row of the
 * alignment passed to iqhip_set_alignment); otherwise *_key names a computed vector.
 * Child order is free: the engine applies the reference's own "leaf goes left" swap
 * (phylokernel.h:116-121) where it matters. */
typedef struct iqhip_node_op {
    uint64_t dst_key;
    uint64_t left_key;
    uint64_t right_key;
    int32_t left_leaf;
    int32_t right_leaf;
    double left_len;  /* PhyloNeighbor::length of the left child branch  */
    double right_len; /* ... of the right child branch */
    uint32_t flags;   /* IQHIP_OP_* */
    uint32_t _pad;
} iqhip_node_op;
/* IQHIP_OP_NO_SCALE: never rescale this result.  A node of degree > 3 (the reference's scalar kernel multiplies ALL
 * children, applies U^-1 and tests for underflow once, phylotreesse.cpp:702-806) is submitted as a chain of binary
 * updates whose intermediate products travel over zero-length branches and carry this flag (iqhip_adapter.h). */
/* IQHIP_OP_SCALAR_RULE: the node's own (last) update of such a chain applies the SCALAR kernel's scaling rule
 * (phylotreesse.cpp:774-801): as the SIMD rule, plus -- before the ptn_invar test -- `lh_max == 0.0` ("very shitty data"):
 * the pattern's vector becomes tip_partial_lh[STATE_UNKNOWN] in every category, scale_num += 4 and
 * sum_scale += 4 * LOG_SCALING_THRESHOLD * ptn_freq.  Binary nodes go through the reference's SIMD kernel, which has no
 * such branch (phylokernel.h:461-474), and carry neither flag. */
enum { IQHIP_OP_NO_SCALE = 1, IQHIP_OP_SCALAR_RULE = 2 };

/* One end of a branch for the lnL / theta calls. leaf >= 0 -> taxon id, else key. */
typedef struct iqhip_branch_end {
    uint64_t key;
    int32_t leaf;
    int32_t _pad;
} iqhip_branch_end;

const char *iqhip_last_error(void);
int iqhip_abi_version(void);
int iqhip_device_count(void);

/* nstates in 2 .. 64.  4, 20, 64 are the reference's SIMD dispatch cases (phylotreeavx.cpp:34-134), 2 its <Vec2d, 2, 2> case
 * (phylotreesse.cpp:262-276); every other count (morphological / multi-state data) goes to the reference's scalar kernels
 * (phylotreesse.cpp:281-309) and here runs on the next kernel size up (4 / 20 / 64) through an exact embedding, with the
 * scalar kernel's scaling rule at every node.  All of them keep the reference's shapes at this boundary (n x n eigen-system,
 * tip table of STATE_UNKNOWN + 1 rows, vectors of nstates*ncat doubles per pattern); embedded counts (everything but 4, 20,
 * 64) have STATE_UNKNOWN = nstates and no mixtures.
 * nptn = aln->size() + unobserved patterns of this shard; ntaxa = leafNum.
 * ncat: 1..8 on the 4-state kernels; 1..16 on the 64-state kernels; 1..96 on the 20-state kernels (the (class, rate)
 * components of a mixture model count as categories, see iqhip_set_mixture_model). */
int iqhip_create(iqhip_engine **out, int device, int nstates, int ncat, int64_t nptn,
                 int ntaxa);
void iqhip_destroy(iqhip_engine *e);

/* ---- pattern sharding over GPUs -------------------------------------------------------------------------------
 * The reference sums over patterns inside one process (phylokernel.h:251,335,410,592-643,951-962); these calls
 * are what replaces those sums when the patterns live on several GPUs.
 *
 * iqhip_create_sharded: ONE engine handle over ndev GPUs.  Shard g holds the contiguous pattern range
 * [nptn*g/ndev, nptn*(g+1)/ndev) rounded down to multiples of 64 (iqhip_shard_range) of every vector; the model
 * tables are replicated.  All other arguments as iqhip_create; the handle is used with every entry point of this
 * header exactly like a single-device engine (the *_async / result-buffer calls excepted: it reduces itself).
 * reduce_mode: IQHIP_REDUCE_RCCL -- per evaluation one grouped ncclAllReduce on the shards' streams (devices must
 * be distinct); IQHIP_REDUCE_HOST -- every shard writes its result doubles to pinned host memory and the host adds
 * them in shard order (also allows shards that share a device). */
enum { IQHIP_REDUCE_RCCL = 0, IQHIP_REDUCE_HOST = 1 };
int iqhip_create_sharded(iqhip_engine **out, const int *device_ids, int ndev, int reduce_mode, int nstates, int ncat,
                         int64_t nptn, int ntaxa);
int iqhip_num_shards(iqhip_engine *e); /* 1 for a plain engine */
int iqhip_shard_range(iqhip_engine *e, int shard, int64_t *first, int64_t *count, int *device);
/* One process per GPU: rank 0 obtains the 128-byte id (ncclGetUniqueId), the caller distributes it (MPI_Bcast,
 * torch.distributed, a file), every rank attaches its engine.  nptn of each engine is that rank's pattern count.
 * Afterwards all ranks must make the same sequence of compute calls (each contains the collective). */
#define IQHIP_COMM_ID_BYTES 128
int iqhip_comm_unique_id(void *id_out /* IQHIP_COMM_ID_BYTES */);
int iqhip_comm_init_rank(iqhip_engine *e, int nranks, int rank, const void *id /* IQHIP_COMM_ID_BYTES */);
int iqhip_comm_size(iqhip_engine *e); /* ranks / shards that share the patterns; 1 = not sharded */

/* Optional: run on the caller's HIP stream (hipStream_t) instead of the engine's own. */
int iqhip_set_stream(iqhip_engine *e, void *hip_stream);
/* Pre-allocate device slabs for `nvectors` partial-likelihood vectors (the reference's
 * central_partial_lh arena, phylotree.cpp:867-873). Optional; slabs are created on demand. */
int iqhip_reserve(iqhip_engine *e, int nvectors);
/* Forget a key (PhyloTree::deleteAllPartialLh / aligned_free of an NNI scratch buffer). */
int iqhip_release(iqhip_engine *e, uint64_t key);
/* Move a vector to a new key without touching device data (LM_PER_NODE re-orientation is a
 * pointer move on the host, phylokernel.h:127-143, so the key usually does not change). */
int iqhip_rekey(iqhip_engine *e, uint64_t old_key, uint64_t new_key);

/* Alignment side inputs (phylotreesse.cpp:531-569, pattern.h:24, alignment.cpp:470-472).
 * states: ntaxa rows of nptn state bytes (row = taxon id);  ptn_freq, ptn_invar: nptn. */
int iqhip_set_alignment(iqhip_engine *e, const uint8_t *states, const double *ptn_freq,
                        const double *ptn_invar);
int iqhip_set_ptn_freq(iqhip_engine *e, const double *ptn_freq);   /* bootstrap re-weighting */
int iqhip_set_ptn_invar(iqhip_engine *e, const double *ptn_invar); /* +I changed */

/* +ASC (ascertainment-bias correction): the LAST n_unobserved patterns passed to
 * iqhip_set_alignment are the unobserved constant patterns (ModelFactory::unobserved_ptns,
 * phylokernel.h:87; frequency 0, ptn_invar = p_invar*pi) and nsites = aln->getNSite().  The lnL and
 * derivative calls then apply phylokernel.h:868-909,968-1016 (branch), :655-725 (derivatives) and
 * :1124-1187 (from buffer), also inside iqhip_newton_branch / iqhip_optimize_sweep.  0 switches it off.  Not available
 * with the *_async calls and the batched NNI evaluator.  Sharded engines (iqhip_create_sharded): the unobserved patterns
 * must fit the last shard; prob_const / df_const / ddf_const are reduced with the result.  Comm engines
 * (iqhip_comm_init_rank): the rank holding the unobserved patterns (the last one) passes their number, every other
 * rank passes (0, nsites). */
int iqhip_set_ascertainment(iqhip_engine *e, int64_t n_unobserved, double nsites);

/* Model side inputs (model/modelsubst.h:248-258, model/rateheterogeneity.h:95-141,
 * phylotreesse.cpp:359-529): eval[n], evec[n*n], inv_evec[n*n], rates[ncat], props[ncat],
 * tip_partial_lh[(state_unknown+1)*n]. Invalidates nothing by itself: like the reference,
 * the caller clears partial_lh_computed flags (clearAllPartialLH). */
int iqhip_set_model(iqhip_engine *e, const double *eval, const double *evec,
                    const double *inv_evec, const double *rates, const double *props,
                    int state_unknown, const double *tip_partial_lh);

/* Mixture models (computeMixturePartialLikelihoodEigenSIMD & co, phylokernelmixture.h:20-460; fused
 * mixture-rate models, phylokernelmixrate.h:22-450).  The engine's ncat categories are the (class, rate)
 * components in the reference's block order [class][rate] (block = nstates*ncat*nmixture there,
 * phylokernelmixture.h:55-56): component q uses eigen-system cat_class[q], rate rates[q] and weight
 * props[q] (= class weight x category proportion).  eval / evec / inv_evec are the nclass systems
 * concatenated (model->getEigenvalues() etc. of ModelMixture); tip_partial_lh is [state][class][n]
 * (phylotreesse.cpp:395-458, phylokernelmixture.h:151).  20 states only (IQHIP_ERR_UNSUPPORTED else). */
int iqhip_set_mixture_model(iqhip_engine *e, int nclass, const int32_t *cat_class /* [ncat] */,
                            const double *eval, const double *evec, const double *inv_evec,
                            const double *rates /* [ncat] */, const double *props /* [ncat] */,
                            int state_unknown, const double *tip_partial_lh);

/* Execute a post-ordered list of node updates (children before parents) in ONE submission.
 * sum_scale[k] receives op k's own sum_scale (phylokernel.h:389,471: LOG_SCALING_THRESHOLD *
 * sum of ptn_freq over the patterns rescaled at this node); the caller keeps
 * lh_scale_factor = left + right + sum_scale exactly as phylokernel.h:157,395,477. */
int iqhip_update_partials(iqhip_engine *e, const iqhip_node_op *ops, int nops,
                          double *sum_scale);

/* computeLikelihoodBranchEigenSIMD's pattern loop (phylokernel.h:779-966) on branch (a,b)
 * of length len.  At most one end may be a leaf.  *lnl = sum_ptn freq*log|lh_ptn| WITHOUT
 * the two lh_scale_factor terms (the caller adds them, phylokernel.h:751).  _pattern_lh is
 * kept on the device (iqhip_fetch_pattern_lh). */
int iqhip_branch_lnl(iqhip_engine *e, iqhip_branch_end a, iqhip_branch_end b, double len,
                     double *lnl);

/* Fused form of the reference's hot loop 1 (clearAllPartialLH(); computeLikelihood()):
 * the node updates and the branch lnL in one device pass. */
int iqhip_traverse_lnl(iqhip_engine *e, const iqhip_node_op *ops, int nops,
                       iqhip_branch_end a, iqhip_branch_end b, double len,
                       double *sum_scale, double *lnl);

/* theta_all = a .* b (phylokernel.h:535-579); kept on the device. */
int iqhip_compute_theta(iqhip_engine *e, iqhip_branch_end a, iqhip_branch_end b);
/* df, ddf at branch length len from theta (phylokernel.h:516-532,583-651). */
int iqhip_derv(iqhip_engine *e, double len, double *df, double *ddf);
/* lnL (without lh_scale_factors) from theta (phylokernel.h:1040-1122); writes _pattern_lh. */
int iqhip_lnl_from_theta(iqhip_engine *e, double len, double *lnl);

/* the whole Newton-Raphson solve of Optimization::minimizeNewton
 * (optimization.cpp:388-450) for the branch whose theta is resident, in ONE launch: the loop
 * "computeFuncDerv; update; test" of optimizeOneBranch (phylotree.cpp:2148-2192) runs on the device
 * with the reference's update rule, bracketing and stopping tests (x1/x2 = min/max branch length,
 * xacc = min branch length, max_steps = maxNRStep).  Returns the optimised length (*optx), the last
 * second derivative (*d2l, as minimizeNewton's d2l) and the number of derivative evaluations.
 * On a sharded engine every derivative evaluation needs the all-reduce of {df, ddf}, so the solve is a chain of
 * enqueued steps instead of one kernel (derivative kernel -> all-reduce -> 1-thread update kernel that applies the
 * same rule to device-resident state); the host reads the state once per few steps, not once per step. */
int iqhip_newton_branch(iqhip_engine *e, double xguess, double x1, double x2, double xacc,
                        int max_steps, double *optx, double *d2l, int *nsteps);
/* optimizeOneBranch in one submission and one host round trip: the pending node updates of both
 * ends of the branch (as iqhip_update_partials), theta (as iqhip_compute_theta) and the Newton
 * solve (as iqhip_newton_branch).  nops may be 0. */
int iqhip_optimize_branch(iqhip_engine *e, const iqhip_node_op *ops, int nops, iqhip_branch_end a,
                          iqhip_branch_end b, double xguess, double x1, double x2, double xacc,
                          int max_steps, double *sum_scale, double *optx, double *d2l, int *nsteps);

/* The same update rule as a host-side state machine (128 opaque bytes), for callers that own the collective: evaluate
 * {df, ddf} at *first_x / *next_x (iqhip_derv_async + their own all-reduce), feed the sums to _update until *done.
 * status as iqhip_branch_result.status.  No likelihood arithmetic happens in these three calls. */
#define IQHIP_NEWTON_STATE_BYTES 128
int iqhip_newton_host_init(void *state, double xguess, double x1, double x2, double xacc, int max_steps, double *first_x);
int iqhip_newton_host_update(void *state, double df_sum, double ddf_sum, double *next_x, int *done);
int iqhip_newton_host_result(const void *state, double *optx, double *d2l, int *nsteps, int *status);

/* Asynchronous use (a caller that owns the collective itself, e.g. a torch.distributed program that all-reduces a
 * tensor it bound as the result buffer; engines with a communicator and sharded engines do this internally and
 * refuse these calls).  The *_async forms enqueue the same work but leave the
 * result on the device: `iqhip_result_device_ptr` is a device array of
 * iqhip_result_capacity() doubles laid out as
 *    [0] = lnl or df, [1] = ddf, [2 .. 2+nops) = sum_scale per op
 * which the caller may all-reduce in place (ncclAllReduce / torch.distributed, SUM, f64)
 * on the engine's stream before iqhip_result_read copies it to the host. */
int iqhip_bind_result_buffer(iqhip_engine *e, void *device_ptr, int capacity_doubles);
void *iqhip_result_device_ptr(iqhip_engine *e);
int iqhip_result_capacity(iqhip_engine *e);
int iqhip_traverse_lnl_async(iqhip_engine *e, const iqhip_node_op *ops, int nops,
                             iqhip_branch_end a, iqhip_branch_end b, double len);
int iqhip_derv_async(iqhip_engine *e, double len);
int iqhip_update_partials_async(iqhip_engine *e, const iqhip_node_op *ops, int nops); /* result[2+k] = sum_scale */
int iqhip_lnl_from_theta_async(iqhip_engine *e, double len);                          /* result[0] = lnl */
int iqhip_result_read(iqhip_engine *e, double *out, int ndoubles); /* syncs the stream */
int iqhip_synchronize(iqhip_engine *e);

/* Lazy device->host views of what the reference keeps in host memory
 * (phylotree.cpp:1062,1218-1227 read scale_num and _pattern_lh on the host). */
int iqhip_fetch_scale_num(iqhip_engine *e, uint64_t key, int16_t *out /* nptn */);
int iqhip_fetch_pattern_lh(iqhip_engine *e, double *out /* nptn */);
int iqhip_fetch_partial(iqhip_engine *e, uint64_t key, double *out /* nptn*block, ref layout */);
int iqhip_fetch_theta(iqhip_engine *e, double *out /* nptn*block, ref layout */);
/* Batched branch optimisation: ntasks INDEPENDENT branches in one submission -- the NNI candidates of a tree
 * (IQTree::evaluateNNIs -> PhyloTree::getBestNNIForBran, phylotree.cpp:2873-3066, evaluates them one after the
 * other; on the device they run side by side).  Task t first runs its own node updates ops[0..nops) (they may
 * read any existing vector, must write vectors no other task touches -- the nni_partial_lh scratch buffers,
 * phylotree.cpp:2901-2924), then optimises the length of branch (a, b) as iqhip_optimize_branch does and
 * evaluates computeLikelihoodFromBuffer at the optimum.  results[t].lnl excludes the lh_scale_factor terms;
 * sum_scale receives the per-op values of all tasks, concatenated in task order.  +ASC is not supported.
 * On a sharded engine (iqhip_create_sharded, iqhip_comm_init_rank) the tasks advance side by side as well: per Newton
 * step one derivative launch for all tasks and ONE all-reduce of 2 * ntasks doubles (chunks of 64 tasks, the same on
 * every rank); +ASC engines run the tasks one after the other. */
typedef struct iqhip_branch_task {
    const iqhip_node_op *ops;
    int32_t nops;
    int32_t max_steps;
    iqhip_branch_end a, b;
    double xguess, x1, x2, xacc;
} iqhip_branch_task;
typedef struct iqhip_branch_result {
    double optx, d2l, lnl;
    int32_t nsteps;
    int32_t status; /* 0 ok, 2 non-finite derivative, 3 step limit reached (optx is still the last iterate), 5 (sweeps) diverged-solve rule applied */
} iqhip_branch_result;
int iqhip_optimize_branch_batch(iqhip_engine *e, const iqhip_branch_task *tasks, int ntasks,
                                double *sum_scale /* sum of nops, may be NULL */, iqhip_branch_result *results);

/* A whole branch-length sweep in one submission: PhyloTree::optimizeAllBranches' loop `for every branch in pre-order:
 * optimizeOneBranch(node1, node2, clearLH = true, maxNRStep)` (phylotree.cpp:2252-2332, 2148-2192).  Step j runs the node
 * updates that are pending at both ends of its branch, builds theta and solves for the branch length exactly as
 * iqhip_optimize_branch does; the length of a child branch that an EARLIER step of the same sweep optimised is not known
 * to the host when the sweep is submitted, so the op refers to it by step number: len_from[2k] / len_from[2k+1] >= 0
 * replaces ops[k].left_len / right_len by the accepted length of that step (the engine reads it from device memory when
 * the op runs); -1 keeps the value in the op.  The caller lists the steps as if every step changed its branch
 * (optimizeOneBranch's clearReversePartialLh on both sides of the branch).
 * diverge_frac > 0 applies optimizeOneBranch's "newton raphson diverged, reset" rule (phylotree.cpp:2167-2176, 0.95 there)
 * inside the sweep: a result above diverge_frac * x2 is kept only if the branch lnL there is not below the lnL at
 * xguess; results[j].status = 5 reports that the rule ran.  results[j].lnl is not filled.  sum_scale receives the per-op
 * values of all steps, concatenated.  Two launches per step, one host round trip per sweep; on sharded engines (every
 * Newton step contains an all-reduce) and with +ASC the steps run one after the other inside this call. */
typedef struct iqhip_sweep_step {
    const iqhip_node_op *ops;
    const int32_t *len_from; /* NULL or 2 * nops entries */
    int32_t nops;
    int32_t _pad;
    iqhip_branch_end a, b;
    double xguess; /* the branch's current length */
} iqhip_sweep_step;
int iqhip_optimize_sweep(iqhip_engine *e, const iqhip_sweep_step *steps, int nsteps, double x1, double x2, double xacc,
                         int max_steps, double diverge_frac, double *sum_scale /* sum of nops, may be NULL */,
                         iqhip_branch_result *results);

/* Consumers of the device-resident _pattern_lh (so -wsl / UFBoot need no full-vector round trip per tree).
 * iqhip_fetch_pattern_lh_scaled: PhyloTree::computePatternLikelihood (phylotree.cpp:1200-1230) for the
 *   branch (a,b) the last lnL evaluation ran on: _pattern_lh + (scale_num_a + scale_num_b)*log(2^-256).
 * iqhip_set_boot_samples: UFBoot's boot_samples (iqtree.h:670, BootValType = float), [nsamples][nptn],
 *   uploaded once.  iqhip_rell: the RELL scores of IQTree::saveCurrentTree (iqtree.cpp:2726-2736),
 *   rell[s] = dotProduct(pattern_lh, boot_samples[s]) (phylokernel.h:55-61), accumulated in double.
 *   The _async form leaves the scores in the result vector [0, nsamples) for a sharded caller to
 *   all-reduce before iqhip_result_read. */
/* _pattern_lh_cat of the reference's scalar kernels (phylotreesse.cpp:1190-1237; consumer
 * RateGamma::computePatternRates, model/rategamma.cpp:241-262): per pattern and category
 * sum_i exp(eval_i r_c len) prop_c theta[ptn][c][i] for the branch of the last iqhip_compute_theta,
 * unscaled, out[ptn*ncat + c]. */
int iqhip_pattern_lh_cat(iqhip_engine *e, double len, double *out /* nptn*ncat */);
int iqhip_fetch_pattern_lh_scaled(iqhip_engine *e, iqhip_branch_end a, iqhip_branch_end b, double *out /* nptn */);
int iqhip_set_boot_samples(iqhip_engine *e, const float *samples, int nsamples);
int iqhip_rell(iqhip_engine *e, iqhip_branch_end a, iqhip_branch_end b, double *rell /* nsamples */);
int iqhip_rell_async(iqhip_engine *e, iqhip_branch_end a, iqhip_branch_end b);
/* Host -> device (tests; SPR/NNI code that fills a buffer on the host). */
int iqhip_upload_partial(iqhip_engine *e, uint64_t key, const double *partial_lh,
                         const int16_t *scale_num);

/* Measurement hook for bench.py: average device time (ms) per launch of the traversal kernel over the
 * launches since the last reset (a staged plan is two launches per traversal), measured with HIP
 * events on the engine's stream. */
int iqhip_timing_enable(iqhip_engine *e, int on);
int iqhip_timing_read(iqhip_engine *e, double *avg_ms, int64_t *launches, int reset);
/* Bytes the traversal launches of the LAST submission store to / load from global memory, derived from its device
 * descriptors (result vectors + counters; children that are neither register-resident nor parked; leaf state rows; the
 * root-branch pass).  `stored` is exact; `loaded` is an upper bound of the fabric reads (same-launch re-reads may hit
 * the L2 / Infinity Cache).  bench.py prices a launch with it when no PMC pass of the shape is on file. */
int iqhip_timing_plan_bytes(iqhip_engine *e, double *stored, double *loaded);
/* Engines with a communicator (iqhip_comm_init_rank): average duration in microseconds of the engine's own all-reduces
 * since the last reset, HIP events on its stream around each ncclAllReduce while timing is enabled -- from the moment the
 * stream reaches the collective to its completion, i.e. including the wait for slower ranks. */
int iqhip_timing_collective_read(iqhip_engine *e, double *avg_us, int64_t *count, int reset);

/* Cherry tables (20 states x 4 categories, >= 8192 patterns; IQHIP_CHERRY_TABLES=0 switches them off): a node whose two
 * children are leaves (computePartialLikelihood's leaf-leaf case, phylokernel.h:187-260) takes one of (STATE_UNKNOWN+1)^2
 * values per pattern, so the engine computes that table once per (pair of taxa, pendant lengths, model) -- with the same
 * kernels on a pseudo-alignment of all state pairs, hence the same bits -- and the traversal copies rows instead of issuing
 * the node's three matrix products.  Counters since the engine was created: tables built, node updates answered. */
int iqhip_debug_cherry_tables(iqhip_engine *e, int64_t *tables_built, int64_t *ops_from_tables);

/* Debugging aid, no reference counterpart: a PLANNING-ONLY engine makes no HIP call and owns no device memory (its
 * vectors are distinct fake addresses).  iqhip_debug_plan turns an op list into the device descriptors exactly as
 * iqhip_update_partials would (key -> slab map, canonical child order, staging, LDS chunks, K2 table slots, look-ahead
 * sentinels) and validates the kernels' contract on them: every pointer of every descriptor, used or not, is a live
 * allocation of the right kind (the traversal kernels request op k+1's inputs unconditionally).  The same check runs on
 * a real engine before every plan upload when IQHIP_CHECK_PLAN=1.  Every other entry point fails on a planner. */
int iqhip_debug_create_planner(iqhip_engine **out, int nstates /* 4, 20, 64 */, int ncat, int64_t nptn, int ntaxa,
                               int num_cus, int state_unknown, int nclass);
int iqhip_debug_plan(iqhip_engine *e, const iqhip_node_op *ops, int nops);

#ifdef __cplusplus
}
#endif
#endif /* IQHIP_H_ */
