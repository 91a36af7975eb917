// phylotree_hip.cpp -- the reference-side binding of libiqhip.so.
//
// This file is what an IQ-TREE maintainer adds to the reference tree (next to phylotreesse.cpp);
// it is NOT built in this repository (it includes the reference's own headers, and the reference
// cannot be built in this image -- see DESIGN.md).  It holds only what is specific to the reference's
// classes: kernel selection, engine set-up from Alignment / ModelFactory / RateHeterogeneity (hipSync),
// a policy struct naming the reference's types, member wrappers and lazy host views.  ALL adapter logic
// is include/iqhip_adapter.h, which this repository compiles and tests through its own instantiation.
//
// Wiring (see INTEGRATION.md for the three small hunks in existing files):
//   phylotree.h      + four member declarations, + `struct iqhip_engine *hip_engine;`
//   phylotreesse.cpp   setLikelihoodKernel(): at the top of the LK_EIGEN_SSE branch (line 145),
//                      before the AVX hand-off (lines 155-157):
//                          if (params->lk_hip && hipKernelUsable()) { setLikelihoodKernelHIP(); return; }
//   tools.cpp          parseArg(): `-hip` sets params.lk_hip (keep params.SSE == LK_EIGEN_SSE so that
//                      every buffer-layout test `sse == LK_EIGEN || sse == LK_EIGEN_SSE` still holds,
//                      phylotree.cpp:706-713,780,813-828,865-874,887-904,944-966).
//
// Buffer contract kept: PhyloNeighbor::partial_lh / scale_num stay the host
// pointers the tree search juggles (NNI scratch phylotree.cpp:2921-2922, LM_PER_NODE stealing
// phylokernel.h:127-143); their VALUE is the key of the device-resident vector.  Host memory
// behind them is never written by this path; code that really reads it (upperbounds.cpp,
// computeBayesianBranchLength) calls hipFetchPartialLh() first.
#include "phylotree.h"
#include "model/modelfactory.h"
#include "model/modelmixture.h"  // ModelMixture::prop (as phylokernelmixture.h does)
#include "iqhip.h"

#define IQHIP_CHECK(call)                                                     \
    do {                                                                      \
        if ((call) != IQHIP_OK) outError("iqhip: ", iqhip_last_error());      \
    } while (0)

static inline uint64_t hipKey(PhyloNeighbor *nei) { return (uint64_t)(uintptr_t)nei->get_partial_lh(); }

// multi-GPU: `-hipdevs 0,1,..,7` (params->hip_devices) makes ONE engine over several GPUs of this process
// (iqhip_create_sharded: patterns sharded, one RCCL all-reduce per evaluation inside the engine); nothing else in
// this file changes
static int hipCreateEngine(iqhip_engine **e, Params *params, int nstates, int ncat, size_t nptn, int ntaxa) {
    if (params->hip_devices.size() > 1)
        return iqhip_create_sharded(e, &params->hip_devices[0], (int)params->hip_devices.size(), IQHIP_REDUCE_RCCL,
                                    nstates, ncat, (int64_t)nptn, ntaxa);
    return iqhip_create(e, params->hip_device, nstates, ncat, (int64_t)nptn, ntaxa);
}

bool PhyloTree::hipKernelUsable() {
    if (!aln || !model_factory || !model || !site_rate) return false;
    if (model->isSiteSpecificModel() || !model->isReversible()) return false;
    int n = aln->num_states;
    // mixtures (ModelMixture, incl. fused mixture-rate models): <= 96 (class, rate) components for 20 states, <= 16
    // for 64, <= 8 for 4; none for binary data
    if (model->isMixture()) {
        int ncomp = model_factory->fused_mix_rate ? site_rate->getNRate() : site_rate->getNRate() * model->getNMixtures();
        int limit = n == 20 ? 96 : n == 64 ? 16 : n == 4 ? 8 : 0;
        if (ncomp > limit) return false;
    }
    // +ASC on a pattern-sharded engine (-hipdevs with several GPUs): the unobserved constant patterns (at most nstates of them)
    // must all lie on the last shard, which any alignment worth sharding satisfies
    if (!model_factory->unobserved_ptns.empty() && params->hip_devices.size() > 1 &&
        aln->getNPattern() < params->hip_devices.size() * 64) return false;
    // the reference's SIMD dispatch cases (phylotreesse.cpp:262-357: binary, DNA, protein, codon) and, through the engine's
    // exact embedding, the state counts it hands to its scalar kernel (:281-309; STATE_UNKNOWN = nstates, no mixtures)
    if (n != 2 && n != 4 && n != 20 && n != 64 && (model->isMixture() || aln->STATE_UNKNOWN != n)) return false;
    return n >= 2 && n <= 64 && iqhip_device_count() > 0;
}

void PhyloTree::setLikelihoodKernelHIP() {
    computePartialLikelihoodPointer = &PhyloTree::computePartialLikelihoodHIP;
    computeLikelihoodBranchPointer = &PhyloTree::computeLikelihoodBranchHIP;
    computeLikelihoodDervPointer = &PhyloTree::computeLikelihoodDervHIP;
    computeLikelihoodFromBufferPointer = &PhyloTree::computeLikelihoodFromBufferHIP;
}

// (re)create the engine when the alignment / category count changes; push model + alignment when
// the reference recomputes its tip table (tip_partial_lh_computed == false after every
// clearAllPartialLH(), phylotree.cpp:495-502)
void PhyloTree::hipSync() {
    size_t norig = aln->size(), nun = model_factory->unobserved_ptns.size();
    size_t nptn = norig + nun;  // phylokernel.h:87: the +ASC constant patterns are appended
    // the engine's categories are the (class, rate) components of the reference's block:
    // block = nstates*ncat*nmixture, index (m*ncat + c) (phylokernelmixture.h:55-56); fused
    // mixture-rate models have one rate per class, block = nstates*ncat (phylokernelmixrate.h:52)
    const bool mix = model->isMixture(), fused = mix && model_factory->fused_mix_rate;
    const int nrate = site_rate->getNRate(), nmix = mix ? model->getNMixtures() : 1;
    int ncat = (mix && !fused) ? nrate * nmix : nrate;
    if (!hip_engine || hip_nptn != nptn || hip_ncat != ncat) {
        if (hip_engine) iqhip_destroy(hip_engine);
        IQHIP_CHECK(hipCreateEngine(&hip_engine, params, aln->num_states, ncat, nptn, leafNum));
        hip_nptn = nptn;
        hip_ncat = ncat;
        hip_aln_pushed = false;
    }
    if (!tip_partial_lh_computed) {
        computeTipPartialLikelihood();  // also fills ptn_freq, ptn_invar (phylotreesse.cpp:359-371)
        vector<double> rates(ncat), props(ncat);
        if (!mix) {
            for (int c = 0; c < ncat; c++) { rates[c] = site_rate->getRate(c); props[c] = site_rate->getProp(c); }
            IQHIP_CHECK(iqhip_set_model(hip_engine, model->getEigenvalues(), model->getEigenvectors(),
                                        model->getInverseEigenvectors(), &rates[0], &props[0],
                                        aln->STATE_UNKNOWN, tip_partial_lh));
        } else {
            // component weights as the reference folds them into `val` (phylokernelmixture.h:762-771:
            // getProp(c) * ModelMixture::prop[m]; fused: getProp(c), phylokernelmixrate.h:739);
            // eigen-systems and tip_partial_lh[state][class][i] are already concatenated per class
            vector<int32_t> cls(ncat);
            for (int q = 0; q < ncat; q++) {
                const int m = fused ? q : q / nrate, c = fused ? q : q % nrate;
                cls[q] = m;
                rates[q] = site_rate->getRate(c);
                props[q] = fused ? site_rate->getProp(c) : site_rate->getProp(c) * ((ModelMixture *)model)->prop[m];
            }
            IQHIP_CHECK(iqhip_set_mixture_model(hip_engine, nmix, &cls[0], model->getEigenvalues(),
                                                model->getEigenvectors(), model->getInverseEigenvectors(),
                                                &rates[0], &props[0], aln->STATE_UNKNOWN, tip_partial_lh));
        }
        if (!hip_aln_pushed) {
            vector<uint8_t> states((size_t)leafNum * nptn);
            for (size_t ptn = 0; ptn < nptn; ptn++)
                for (int t = 0; t < leafNum; t++)
                    states[(size_t)t * nptn + ptn] = ptn < norig ? (uint8_t)(*aln)[ptn][t]
                                                                 : (uint8_t)model_factory->unobserved_ptns[ptn - norig];
            IQHIP_CHECK(iqhip_set_alignment(hip_engine, &states[0], ptn_freq, ptn_invar));
            IQHIP_CHECK(iqhip_set_ascertainment(hip_engine, (int64_t)nun, (double)aln->getNSite()));
            hip_aln_pushed = true;
        } else {
            IQHIP_CHECK(iqhip_set_ptn_freq(hip_engine, ptn_freq));
            IQHIP_CHECK(iqhip_set_ptn_invar(hip_engine, ptn_invar));
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Everything below the engine set-up is include/iqhip_adapter.h -- recursion, lazy flags, LM_PER_NODE re-orientation,
// lh_scale_factor bookkeeping, leaf-side swap, the bodies of the four kernels, multifurcating nodes -- instantiated
// with the reference's own types.  The very same templates, instantiated with a stand-alone mirror of these types
// (iq-tree_amd/host/phylo_host.cpp, MirrorPolicy), are what this repository's test-suite runs on the GPU.
// phylonode.h / phylotree.h each gain one line: `friend struct HipPolicy;` (PhyloNeighbor's fields are private).
// ---------------------------------------------------------------------------------------------------------------
#include "iqhip_adapter.h"

struct HipPolicy : iqhip_adapter::EngineCalls<HipPolicy, PhyloTree> {
    typedef PhyloTree Tree;
    typedef PhyloNode Node;
    typedef PhyloNeighbor Neighbor;
    static Node *node(Neighbor *nb) { return (PhyloNode *)nb->node; }
    static double length(Neighbor *nb) { return nb->length; }
    static void setLength(Neighbor *nb, double len) { nb->length = len; }
    static bool isLeaf(Node *n) { return n->isLeaf(); }
    static int degree(Node *n) { return n->degree(); }
    static int leafId(Node *n) { return n->id; }
    static int numNeighbors(Node *n) { return (int)n->neighbors.size(); }
    static Neighbor *neighborAt(Node *n, int k) { return (PhyloNeighbor *)n->neighbors[k]; }
    static Neighbor *findNeighbor(Node *at, Node *to) { return (PhyloNeighbor *)at->findNeighbor(to); }
    static int &computed(Neighbor *nb) { return nb->partial_lh_computed; }
    static double &scaleFactor(Neighbor *nb) { return nb->lh_scale_factor; }
    static uint64_t key(Neighbor *nb) { return (uint64_t)(uintptr_t)nb->partial_lh; }
    static void stealBuffer(Neighbor *to, Neighbor *from) {  // phylokernel.h:131-137
        to->partial_lh = from->partial_lh;
        to->scale_num = from->scale_num;
        from->partial_lh = NULL;
        from->scale_num = NULL;
    }
    static bool perNodeMode(Tree *t) { return t->params->lh_mem_save == LM_PER_NODE; }
    static bool heavyFirst(Tree *) { return true; }
    static void ensureBuffers(Tree *t) { if (!t->central_partial_lh) t->initializeAllPartialLh(); }
    static void sync(Tree *t) { t->hipSync(); }
    static iqhip_engine *engine(Tree *t) { return t->hip_engine; }
    static void fail(Tree *, const char *what, const char *detail) { outError(what, detail); }  // tools.h:1786, exits
    static void countComputation(Tree *t) { t->num_partial_lh_computations++; }
    static bool &thetaComputed(Tree *t) { return t->theta_computed; }
    static Neighbor *currentIt(Tree *t) { return t->current_it; }
    static Neighbor *currentItBack(Tree *t) { return t->current_it_back; }
    static void clearReversePartialLh(Node *n, Node *dad) { n->clearReversePartialLh(dad); }
    static void setCurrent(Tree *t, Neighbor *it, Neighbor *back) { t->current_it = it; t->current_it_back = back; }
    static double minBranchLength(Tree *t) { return t->params->min_branch_length; }
    static double maxBranchLength(Tree *t) { return t->params->max_branch_length; }
};

void PhyloTree::computePartialLikelihoodHIP(PhyloNeighbor *dad_branch, PhyloNode *dad) {
    iqhip_adapter::computePartialLikelihood<HipPolicy>(this, dad_branch, dad);
}

double PhyloTree::computeLikelihoodBranchHIP(PhyloNeighbor *dad_branch, PhyloNode *dad) {
    hip_pattern_lh_stale = true;  // _pattern_lh lives on the device until somebody asks (hipFetchPatternLh)
    return iqhip_adapter::computeLikelihoodBranch<HipPolicy>(this, dad_branch, dad);
}

void PhyloTree::computeLikelihoodDervHIP(PhyloNeighbor *dad_branch, PhyloNode *dad, double &df, double &ddf) {
    iqhip_adapter::computeLikelihoodDerv<HipPolicy>(this, dad_branch, dad, df, ddf);
}

double PhyloTree::computeLikelihoodFromBufferHIP() {
    hip_pattern_lh_stale = true;
    return iqhip_adapter::computeLikelihoodFromBuffer<HipPolicy>(this);
}

// Optional fast path for hot loop 2: PhyloTree::optimizeOneBranch (phylotree.cpp:2148-2192) with its
// Newton-Raphson branch `optx = minimizeNewton(...)` replaced by ONE engine submission (pending
// updates of both ends + theta + the whole minimizeNewton loop on the device; on a sharded engine the loop is an
// enqueued chain with one in-stream all-reduce per step).  Hunk in optimizeOneBranch:
//   if (optimize_by_newton && hip_engine && computePartialLikelihoodPointer == &PhyloTree::computePartialLikelihoodHIP)
//       optx = hipMinimizeNewton(current_len, maxNRStep); else ...        (+ASC models included: phylokernel.h:655-725 runs
//                                                                         inside the device solve)
// The call leaves current_it / current_it_back at the length it returns -- the last evaluated point, as the reference's
// computeFuncDerv does -- so the diverged-Newton test that follows in optimizeOneBranch (phylotree.cpp:2167-2176) evaluates
// opt_lh = computeLikelihoodFromBuffer() at the right length, unchanged.
double PhyloTree::hipMinimizeNewton(double current_len, int maxNRStep) {
    return iqhip_adapter::minimizeNewtonOnBranch<HipPolicy>(this, current_len, maxNRStep);
}

// Optional fast path for a whole sweep: the loop `for (j...) optimizeOneBranch(nodes1[j], nodes2[j], true, maxNRStep)` of
// PhyloTree::optimizeAllBranches (phylotree.cpp:2285-2290) as ONE engine submission with one host round trip
// (iqhip_optimize_sweep: per branch the pending node updates, theta, the whole minimizeNewton loop and the diverged-Newton
// reset, later branches reading the lengths of earlier ones from device memory).  Hunk in optimizeAllBranches:
//   if (optimize_by_newton && hip_engine && computePartialLikelihoodPointer == &PhyloTree::computePartialLikelihoodHIP
//       && !isSuperTree()) hipOptimizeBranchSweep(nodes1, nodes2, maxNRStep); else for (j...) optimizeOneBranch(...);
void PhyloTree::hipOptimizeBranchSweep(PhyloNodeVector &nodes1, PhyloNodeVector &nodes2, int maxNRStep) {
    if (nodes1.empty()) return;
    theta_computed = false;
    iqhip_adapter::optimizeBranchSweep<HipPolicy>(this, &nodes1[0], &nodes2[0], (int)nodes1.size(), maxNRStep, 0.95);
}

// Lazy host views for the few callers that read kernel outputs on the host
// (computePatternLikelihood phylotree.cpp:1200-1273, computeLikelihood phylotree.cpp:1062).
void PhyloTree::hipFetchPatternLh() {
    if (hip_engine && hip_pattern_lh_stale) {
        IQHIP_CHECK(iqhip_fetch_pattern_lh(hip_engine, _pattern_lh));
        hip_pattern_lh_stale = false;
    }
}
void PhyloTree::hipFetchScaleNum(PhyloNeighbor *nei) {
    if (hip_engine && nei->scale_num) IQHIP_CHECK(iqhip_fetch_scale_num(hip_engine, hipKey(nei), nei->scale_num));
}
void PhyloTree::hipFetchPartialLh(PhyloNeighbor *nei) {
    if (hip_engine && nei->partial_lh) IQHIP_CHECK(iqhip_fetch_partial(hip_engine, hipKey(nei), nei->partial_lh));
}
