// phylo_host.h -- host-side mirror of the slice of IQ-TREE's PhyloTree that sits directly on
// top of the likelihood kernels.  It exists so that (a) the drop-in boundary can be exercised
// exactly the way the reference's callers exercise it (setLikelihoodKernel(); clearAllPartialLH();
// computeLikelihood(); computeLikelihoodDerv(); ...) on a box that has no reference sources,
// and (b) the adapter logic shipped in integration/phylotree_hip.cpp (recursion, lazy flags,
// LM_PER_NODE re-orientation, lh_scale_factor bookkeeping) is tested code.
//
// Names and argument meaning follow the reference (file:line = /root/reference):
//   PhyloNeighbor fields            phylonode.h:102-127
//   setLikelihoodKernel + 4 ptrs    phylotreesse.cpp:60-357, phylotree.h:658-659,697-698,742-743,979-980
//   initializeAllPartialLh          phylotree.cpp:667-716,834-987   (LM_PER_NODE / LM_ALL_BRANCH)
//   computeLikelihood               phylotree.cpp:1031-1072
//   clear*PartialLh                 phylonode.cpp:15-65, phylotree.cpp:495-502
//   computeTipPartialLikelihood     phylotreesse.cpp:359-529
//   optimizeOneBranch/AllBranches   phylotree.cpp:2120-2332, optimization.cpp:388-465
// The compute kernels themselves are NOT here: the four *HIP member functions only build a
// plan and call libiqhip.so (include/iqhip.h).  There is no CPU kernel in this library; asking
// for LK_EIGEN / LK_EIGEN_SSE throws.
#pragma once
#include <stdint.h>

#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/iqhip.h"

namespace iqhost {

typedef short int UBYTE;  // phylonode.h:17

enum LikelihoodKernel { LK_EIGEN = 0, LK_EIGEN_SSE = 1, LK_EIGEN_HIP = 2 };  // tools.h:397-399 (+HIP)
enum LhMemSave { LM_PER_NODE = 0, LM_ALL_BRANCH = 1 };                       // tools.cpp:907,2461
enum SeqType { SEQ_DNA = 0, SEQ_PROTEIN = 1, SEQ_CODON = 2, SEQ_OTHER = 3 };

struct PhyloNode;

struct PhyloNeighbor {
    PhyloNode *node = nullptr;  // the node this neighbor points TO
    double length = 0.0;
    int id = -1;                // branch id
    int partial_lh_computed = 0;  // bit0 = likelihood vector valid (phylonode.h:104)
    uint64_t partial_lh = 0;      // opaque device key; 0 == NULL (phylonode.h:112)
    double lh_scale_factor = 0.0; // phylonode.h:117
    inline void clearPartialLh() { partial_lh_computed = 0; }
    void clearForwardPartialLh(PhyloNode *dad);  // phylonode.cpp:15-20
};

struct PhyloNode {
    int id = -1;  // leaves: taxon id (alignment row); internal: >= leafNum
    std::string name;
    std::vector<PhyloNeighbor *> neighbors;
    double height = 0.0;
    bool isLeaf() const { return neighbors.size() <= 1; }
    int degree() const { return (int)neighbors.size(); }
    PhyloNeighbor *findNeighbor(const PhyloNode *n) const {
        for (PhyloNeighbor *x : neighbors)
            if (x->node == n) return x;
        return nullptr;
    }
    void clearReversePartialLh(PhyloNode *dad);              // phylonode.cpp:42-50
    void clearAllPartialLh(bool make_null, PhyloNode *dad);  // phylonode.cpp:52-65
};

// a recorded node update (also what the CPU-only tests inspect in dry-run mode); dst / left / right are NULL for
// the intermediate products of a multifurcating node
struct PlanOp {
    PhyloNeighbor *dst, *left, *right;
    iqhip_node_op op;
};

struct MirrorPolicy;  // phylo_host.cpp: this tree's answers to include/iqhip_adapter.h

class PhyloTree {
public:
    typedef void (PhyloTree::*ComputePartialLikelihoodType)(PhyloNeighbor *, PhyloNode *);
    typedef double (PhyloTree::*ComputeLikelihoodBranchType)(PhyloNeighbor *, PhyloNode *);
    typedef double (PhyloTree::*ComputeLikelihoodFromBufferType)();
    typedef void (PhyloTree::*ComputeLikelihoodDervType)(PhyloNeighbor *, PhyloNode *, double &, double &);

    PhyloTree();
    ~PhyloTree();
    PhyloTree(const PhyloTree &) = delete;
    PhyloTree &operator=(const PhyloTree &) = delete;

    // ---- tree ----------------------------------------------------------------------------
    // Newick with branch lengths; leaf labels are looked up in `names` (alignment order) or,
    // when names is empty, parsed as integer taxon ids.  A bifurcating top level is unrooted.
    void readTreeString(const std::string &newick, const std::vector<std::string> &names);
    std::string getTreeString() const;
    int leafNum = 0, nodeNum = 0, branchNum = 0;
    PhyloNode *root = nullptr;  // a leaf, as in the reference (phylotree.cpp:1034)
    std::vector<PhyloNode *> nodes;  // index == id
    PhyloNode *findFarthestLeaf(PhyloNode *node = nullptr, PhyloNode *dad = nullptr);  // mtree.cpp:2052

    // ---- data + model (the kernel's inputs, SURVEY 8a a5,a6,a13) -------------------------
    void setAlignment(int nstates, SeqType seq_type, int64_t nptn, const uint8_t *states /*[leaf][ptn]*/,
                      const double *ptn_freq, const double *ptn_invar);
    // +ASC (ModelFactory::unobserved_ptns, model/modelfactory.cpp:359-370): the last n_unobserved
    // patterns of setAlignment are the unobserved constant patterns; nsites = aln->getNSite()
    void setAscertainment(int64_t n_unobserved, double nsites);
    // new pattern weights / invariant-site terms on an attached engine (computePtnFreq / computePtnInvar,
    // phylotreesse.cpp:531-569: the reference recomputes them with every model change)
    void setPtnFreq(const double *ptn_freq);
    void setPtnInvar(const double *ptn_invar);
    int64_t n_unobserved = 0;
    double asc_nsites = 0.0;
    void setModel(int ncat, const double *eval, const double *evec, const double *inv_evec,
                  const double *rates, const double *props);
    // Mixture models (ModelMixture; phylokernelmixture.h / phylokernelmixrate.h): ncat_ components in the
    // reference's block order [class][rate], component q on eigen-system cat_class[q]; eval / evec /
    // inv_evec are the nclass systems concatenated, props[q] = class weight x category proportion
    void setMixtureModel(int nclass, int ncat, const int *cat_class, const double *eval, const double *evec,
                         const double *inv_evec, const double *rates, const double *props);
    int nmixture = 1;
    int num_states = 0, ncat = 0, STATE_UNKNOWN = 0;
    SeqType seq_type = SEQ_DNA;
    int64_t nptn = 0;
    std::vector<double> tip_partial_lh;
    void computeTipPartialLikelihood();  // phylotreesse.cpp:459-527

    // ---- kernel dispatch -----------------------------------------------------------------
    void setLikelihoodKernel(LikelihoodKernel lk);  // phylotreesse.cpp:60
    LikelihoodKernel sse = LK_EIGEN_HIP;
    LhMemSave lh_mem_save = LM_PER_NODE;
    // device: >= 0 creates an engine on that GPU; dry_run records plans without any device
    void attachEngine(int device);
    // pattern-sharded engines (include/iqhip.h "pattern sharding over GPUs"): one handle over several GPUs of this
    // process, or this process's engine joined to the other ranks' (one process per GPU); nothing else changes for
    // the callers below -- every host-visible sum is all-reduced inside the engine
    void attachEngineSharded(const int *device_ids, int ndev, int reduce_mode);
    void attachComm(int nranks, int rank, const void *unique_id);
    void setDryRun(bool on) { dry_run = on; }
    bool heavy_first = true;  // plan order of independent subtrees (see collectPlan)
    // optimizeOneBranch: run the whole Newton-Raphson solve on the device (iqhip_newton_branch)
    // instead of one computeLikelihoodDerv round trip per step; off -> the reference's host loop
    bool device_newton = true;
    bool device_sweep = true;   // optimizeAllBranches: a whole sweep as one engine submission (iqhip_optimize_sweep)
    long num_derv_calls = 0;  // derivative evaluations (host loop: calls; device loop: reported steps)
    iqhip_engine *engine = nullptr;
    // Pattern-sharded runs (one process per GPU): when set, every host-visible result vector
    // {lnL | df,ddf | sum_scale per op} is left on the device, handed to this hook (which
    // all-reduces it in place over RCCL on the engine's stream) and only then read back.
    typedef void (*AllReduceHook)(void *device_ptr, int ndoubles, void *ctx);
    void setAllReduceHook(AllReduceHook fn, void *ctx) { allreduce_hook = fn; allreduce_ctx = ctx; }

    void computePartialLikelihood(PhyloNeighbor *dad_branch, PhyloNode *dad);       // :335
    double computeLikelihoodBranch(PhyloNeighbor *dad_branch, PhyloNode *dad);      // :339
    void computeLikelihoodDerv(PhyloNeighbor *dad_branch, PhyloNode *dad, double &df, double &ddf);  // :344
    double computeLikelihoodFromBuffer();                                           // :349

    // ---- buffers / flags -----------------------------------------------------------------
    void initializeAllPartialLh();  // phylotree.cpp:667
    void deleteAllPartialLh();      // phylotree.cpp:718
    void clearAllPartialLH();       // phylotree.cpp:495
    bool central_partial_lh = false;
    bool theta_computed = false;
    PhyloNeighbor *current_it = nullptr, *current_it_back = nullptr;

    // ---- callers (hot loops 1 and 2 of SURVEY section 3) -----------------------------------
    double computeLikelihood(double *pattern_lh = nullptr);  // phylotree.cpp:1031
    double curScore = 0.0;
    double min_branch_length = 1e-6, max_branch_length = 100.0;  // tools.cpp defaults
    void optimizeOneBranch(PhyloNode *node1, PhyloNode *node2, bool clearLH = true, int maxNRStep = 100);
    double optimizeAllBranches(int my_iterations = 100, double tolerance = 0.001, int maxNRStep = 100);

    // ---- NNI evaluation (phylotree.cpp:2873-3066): both swaps around an internal branch, the central
    //      branch (and with nni5 the four adjacent ones) re-optimised on scratch buffers, tree restored
    struct NNIMove {
        int node1 = -1, node2 = -1;
        int node1_nei = -1, node2_nei = -1;  // the two subtrees (by their root node id) that were swapped
        double newloglh = 0.0;
        double newLen[5] = {0, 0, 0, 0, 0};
    };
    NNIMove getBestNNIForBran(PhyloNode *node1, PhyloNode *node2, bool nni5, NNIMove moves[2]);
    static const int NNI_MAX_NR_STEP = 10;  // phylotree.h
    // computeAllPartialLh (phylotree.cpp:504-514): every directed vector valid (needs LM_ALL_BRANCH)
    void computeAllPartialLh();
    // IQTree::evaluateNNIs with nni1 for EVERY internal branch, all 2(n-3) candidates side by side in one
    // iqhip_optimize_branch_batch submission (same swaps, same Newton solve, same lnL as getBestNNIForBran
    // gives one branch at a time).  moves: 2 per internal branch, in branch order.  Needs LM_ALL_BRANCH.
    void evaluateNNIsBatch(std::vector<NNIMove> &moves);
    // the same with nni5 (the reference's default, params.nni5): per candidate the two branches at node1, the
    // central branch and the two at node2 are optimised in that order (phylotree.cpp:2984-3024); five rounds of
    // batched tasks per swap, ten submissions per tree instead of ~10 per branch
    void evaluateNNIs5Batch(std::vector<NNIMove> &moves);

    // ---- consumers of the per-pattern lnL (phylotree.cpp:1200-1230, iqtree.cpp:2676-2750) ----------
    // computePatternLikelihood: lnL per pattern of the last computeLikelihood(), scaling events of
    // both ends of current_it put back -- computed on the device, one D2H of nptn doubles
    void computePatternLikelihood(double *ptn_lh);
    // _pattern_lh_cat[ptn*ncat + c] of the current branch (phylotree.cpp:1119-1124 -> scalar kernels), unscaled
    void computePatternLhCat(double *ptn_lh_cat);
    // UFBoot: boot_samples uploaded once; computeRELL = saveCurrentTree's dot products, on the device
    void setBootSamples(const float *samples /*[nsamples][nptn]*/, int nsamples);
    void computeRELL(std::vector<double> &rell);
    int num_boot_samples = 0;

    // ---- host views ----------------------------------------------------------------------
    void fetchScaleNum(PhyloNeighbor *nei, UBYTE *out);
    void fetchPartialLh(PhyloNeighbor *nei, double *out);
    void fetchPatternLh(double *out);

    std::vector<PlanOp> last_plan;  // the most recent submission (tests / tracing)
    long num_partial_lh_computations = 0;  // phylokernel.h:85 (counts leaf visits too)
    long num_submissions = 0;

private:
    friend struct MirrorPolicy;
    ComputePartialLikelihoodType computePartialLikelihoodPointer = nullptr;
    ComputeLikelihoodBranchType computeLikelihoodBranchPointer = nullptr;
    ComputeLikelihoodFromBufferType computeLikelihoodFromBufferPointer = nullptr;
    ComputeLikelihoodDervType computeLikelihoodDervPointer = nullptr;

    // the four kernels registered for LK_EIGEN_HIP
    void computePartialLikelihoodHIP(PhyloNeighbor *dad_branch, PhyloNode *dad);
    double computeLikelihoodBranchHIP(PhyloNeighbor *dad_branch, PhyloNode *dad);
    void computeLikelihoodDervHIP(PhyloNeighbor *dad_branch, PhyloNode *dad, double &df, double &ddf);
    double computeLikelihoodFromBufferHIP();

    // plan building, flag handling, re-orientation, scale-factor bookkeeping and the bodies of the four kernels are
    // include/iqhip_adapter.h instantiated with MirrorPolicy -- the same templates integration/phylotree_hip.cpp
    // instantiates with the reference's types
    iqhip_branch_end branchEnd(PhyloNeighbor *nei) const;
    void check(int rc, const char *what) const;
    void pushInputs();

    double computeFuncDerv(double value, double &df, double &ddf);
    double minimizeNewton(double x1, double xguess, double x2, double xacc, double &d2l, int maxNRStep);
    void getPreOrderBranches(std::vector<PhyloNode *> &n1, std::vector<PhyloNode *> &n2, PhyloNode *node,
                             PhyloNode *dad);

    AllReduceHook allreduce_hook = nullptr;
    void *allreduce_ctx = nullptr;
    bool dry_run = false;
    bool inputs_dirty = true;   // anything to push before the next submission
    bool model_dirty = true, aln_dirty = true, weights_dirty = false;
    uint64_t next_key = 1;
    uint64_t nni_keys[6] = {0, 0, 0, 0, 0, 0};
    std::vector<uint64_t> nni_batch_keys;  // scratch vectors of evaluateNNIsBatch (2 per candidate)  // nni_partial_lh scratch (phylotree.cpp:852-860)
    std::vector<uint8_t> aln_states;
    std::vector<double> ptn_freq, ptn_invar, m_eval, m_evec, m_inv_evec, m_rates, m_props;
    std::vector<int> m_cat_class;
    std::vector<PhyloNeighbor *> all_neighbors;
    void freeTree();
};

}  // namespace iqhost
