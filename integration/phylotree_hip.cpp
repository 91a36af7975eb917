// phylotree_hip.cpp -- the reference-side binding of libiqhip.so.
//
// This file is what an IQ-TREE maintainer adds to the reference tree (next to phylotreesse.cpp);
// it is NOT built in this repository (it includes the reference's own headers, and the reference
// cannot be built in this image -- see DESIGN.md).  The same logic, on a stand-alone copy of the
// PhyloTree slice, is what iq-tree_amd/host/phylo_host.cpp implements and what the tests run.
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
#include "iqhip.h"

#define IQHIP_CHECK(call)                                                     \
    do {                                                                      \
        if ((call) != IQHIP_OK) outError("iqhip: ", iqhip_last_error());      \
    } while (0)

static inline uint64_t hipKey(PhyloNeighbor *nei) { return (uint64_t)(uintptr_t)nei->get_partial_lh(); }

bool PhyloTree::hipKernelUsable() {
    if (!aln || !model_factory || !model || !site_rate) return false;
    if (model->isSiteSpecificModel() || !model->isReversible()) return false;
    int n = aln->num_states;
    // mixtures (ModelMixture, incl. fused mixture-rate models): 20 states, <= 96 (class, rate) components
    if (model->isMixture()) {
        int ncomp = model_factory->fused_mix_rate ? site_rate->getNRate() : site_rate->getNRate() * model->getNMixtures();
        if (n != 20 || ncomp > 96) return false;
    }
    return (n == 4 || n == 20 || n == 64) && iqhip_device_count() > 0;
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
        IQHIP_CHECK(iqhip_create(&hip_engine, params->hip_device, aln->num_states, ncat, nptn, leafNum));
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

struct HipPlanOp { PhyloNeighbor *dst, *left, *right; };

// phylokernel.h:70-157 with the three pattern loops replaced by "append one op"
void PhyloTree::hipCollectPlan(PhyloNeighbor *dad_branch, PhyloNode *dad, vector<iqhip_node_op> &ops,
                               vector<HipPlanOp> &plan) {
    if (dad_branch->partial_lh_computed & 1) return;
    dad_branch->partial_lh_computed |= 1;
    num_partial_lh_computations++;
    PhyloNode *node = (PhyloNode *)dad_branch->node;
    if (node->isLeaf()) { dad_branch->lh_scale_factor = 0.0; return; }
    if (node->degree() != 3) outError("HIP likelihood kernel: multifurcating node");  // scalar-only in the reference too
    PhyloNeighbor *left = NULL, *right = NULL;
    FOR_NEIGHBOR_IT(node, dad, it) { if (!left) left = (PhyloNeighbor *)(*it); else right = (PhyloNeighbor *)(*it); }
    if (!left->node->isLeaf() && right->node->isLeaf()) { PhyloNeighbor *t = left; left = right; right = t; }
    if ((left->partial_lh_computed & 1) == 0) hipCollectPlan(left, node, ops, plan);
    if ((right->partial_lh_computed & 1) == 0) hipCollectPlan(right, node, ops, plan);
    if (params->lh_mem_save == LM_PER_NODE && !dad_branch->partial_lh) dad_branch->reorientPartialLh(dad);
    iqhip_node_op op;
    memset(&op, 0, sizeof(op));
    op.dst_key = hipKey(dad_branch);
    op.left_leaf = left->node->isLeaf() ? left->node->id : -1;
    op.right_leaf = right->node->isLeaf() ? right->node->id : -1;
    op.left_key = left->node->isLeaf() ? 0 : hipKey(left);
    op.right_key = right->node->isLeaf() ? 0 : hipKey(right);
    op.left_len = left->length;
    op.right_len = right->length;
    ops.push_back(op);
    HipPlanOp p = {dad_branch, left, right};
    plan.push_back(p);
}

static void hipApplyScale(vector<HipPlanOp> &plan, vector<double> &sum_scale) {
    for (size_t k = 0; k < plan.size(); k++)  // phylokernel.h:157,395,477
        plan[k].dst->lh_scale_factor = plan[k].left->lh_scale_factor + plan[k].right->lh_scale_factor + sum_scale[k];
}

static iqhip_branch_end hipEnd(PhyloNeighbor *nei) {
    iqhip_branch_end e;
    e._pad = 0;
    e.leaf = nei->node->isLeaf() ? nei->node->id : -1;
    e.key = nei->node->isLeaf() ? 0 : hipKey(nei);
    return e;
}

void PhyloTree::computePartialLikelihoodHIP(PhyloNeighbor *dad_branch, PhyloNode *dad) {
    if (!central_partial_lh) initializeAllPartialLh();
    hipSync();
    vector<iqhip_node_op> ops;
    vector<HipPlanOp> plan;
    hipCollectPlan(dad_branch, dad, ops, plan);
    if (ops.empty()) return;
    vector<double> sum_scale(ops.size());
    IQHIP_CHECK(iqhip_update_partials(hip_engine, &ops[0], (int)ops.size(), &sum_scale[0]));
    hipApplyScale(plan, sum_scale);
}

double PhyloTree::computeLikelihoodBranchHIP(PhyloNeighbor *dad_branch, PhyloNode *dad) {
    PhyloNode *node = (PhyloNode *)dad_branch->node;
    PhyloNeighbor *node_branch = (PhyloNeighbor *)node->findNeighbor(dad);
    if (!central_partial_lh) initializeAllPartialLh();
    if (node->isLeaf()) {  // phylokernel.h:739-746
        PhyloNode *tn = dad; dad = node; node = tn;
        PhyloNeighbor *tb = dad_branch; dad_branch = node_branch; node_branch = tb;
    }
    hipSync();
    vector<iqhip_node_op> ops;
    vector<HipPlanOp> plan;
    if ((dad_branch->partial_lh_computed & 1) == 0) hipCollectPlan(dad_branch, dad, ops, plan);
    if ((node_branch->partial_lh_computed & 1) == 0) hipCollectPlan(node_branch, node, ops, plan);
    vector<double> sum_scale(ops.size() + 1);
    double lnl;
    IQHIP_CHECK(iqhip_traverse_lnl(hip_engine, ops.empty() ? NULL : &ops[0], (int)ops.size(),
                                   hipEnd(node_branch), hipEnd(dad_branch), dad_branch->length,
                                   &sum_scale[0], &lnl));
    hipApplyScale(plan, sum_scale);
    hip_pattern_lh_stale = true;  // _pattern_lh lives on the device until somebody asks (hipFetchPatternLh)
    return node_branch->lh_scale_factor + dad_branch->lh_scale_factor + lnl;  // phylokernel.h:751
}

void PhyloTree::computeLikelihoodDervHIP(PhyloNeighbor *dad_branch, PhyloNode *dad, double &df, double &ddf) {
    PhyloNode *node = (PhyloNode *)dad_branch->node;
    PhyloNeighbor *node_branch = (PhyloNeighbor *)node->findNeighbor(dad);
    if (!central_partial_lh) initializeAllPartialLh();
    if (node->isLeaf()) {  // phylokernel.h:491-498
        PhyloNode *tn = dad; dad = node; node = tn;
        PhyloNeighbor *tb = dad_branch; dad_branch = node_branch; node_branch = tb;
    }
    if ((dad_branch->partial_lh_computed & 1) == 0) computePartialLikelihoodHIP(dad_branch, dad);
    if ((node_branch->partial_lh_computed & 1) == 0) computePartialLikelihoodHIP(node_branch, node);
    hipSync();
    if (!theta_computed) {  // phylokernel.h:535-579
        theta_computed = true;
        IQHIP_CHECK(iqhip_compute_theta(hip_engine, hipEnd(node_branch), hipEnd(dad_branch)));
    }
    IQHIP_CHECK(iqhip_derv(hip_engine, dad_branch->length, &df, &ddf));
}

double PhyloTree::computeLikelihoodFromBufferHIP() {
    assert(theta_all && theta_computed);
    double lnl;
    IQHIP_CHECK(iqhip_lnl_from_theta(hip_engine, current_it->length, &lnl));
    hip_pattern_lh_stale = true;
    return current_it->lh_scale_factor + current_it_back->lh_scale_factor + lnl;  // phylokernel.h:1028
}

// Optional fast path for hot loop 2: PhyloTree::optimizeOneBranch (phylotree.cpp:2148-2192) with its
// Newton-Raphson branch `optx = minimizeNewton(...)` replaced by ONE engine submission (pending
// updates of both ends + theta + the whole minimizeNewton loop on the device).  Hunk in
// optimizeOneBranch: `if (optimize_by_newton && hip_engine && computePartialLikelihoodPointer ==
// &PhyloTree::computePartialLikelihoodHIP && model_factory->unobserved_ptns.empty())
//     optx = hipMinimizeNewton(current_len, maxNRStep); else ...`   (+ASC keeps the host loop)
double PhyloTree::hipMinimizeNewton(double current_len, int maxNRStep) {
    PhyloNeighbor *dad_branch = current_it, *node_branch = current_it_back;
    PhyloNode *dad = (PhyloNode *)current_it_back->node, *node = (PhyloNode *)current_it->node;
    if (node->isLeaf()) {
        PhyloNode *tn = dad; dad = node; node = tn;
        PhyloNeighbor *tb = dad_branch; dad_branch = node_branch; node_branch = tb;
    }
    if (!central_partial_lh) initializeAllPartialLh();
    hipSync();
    vector<iqhip_node_op> ops;
    vector<HipPlanOp> plan;
    if ((dad_branch->partial_lh_computed & 1) == 0) hipCollectPlan(dad_branch, dad, ops, plan);
    if ((node_branch->partial_lh_computed & 1) == 0) hipCollectPlan(node_branch, node, ops, plan);
    vector<double> sum_scale(ops.size() + 1);
    double optx, d2l;
    int nsteps;
    theta_computed = true;
    IQHIP_CHECK(iqhip_optimize_branch(hip_engine, ops.empty() ? NULL : &ops[0], (int)ops.size(), hipEnd(node_branch),
                                      hipEnd(dad_branch), current_len, params->min_branch_length,
                                      params->max_branch_length, params->min_branch_length, maxNRStep,
                                      &sum_scale[0], &optx, &d2l, &nsteps));
    hipApplyScale(plan, sum_scale);
    return optx;
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
