// iqhost_c.cpp -- flat C view of iqhost::PhyloTree for ctypes (tests, bench.py, smoke()).
// Not part of the drop-in boundary (that is include/iqhip.h); this is how the Python side of
// the repository drives the same call sequence the reference's C++ callers use.
#include <string.h>

#include <string>
#include <vector>

#include "phylo_host.h"

using namespace iqhost;

static thread_local std::string g_host_err;

#define IQHOST_TRY(body)                   \
    try {                                  \
        body;                              \
        return 0;                          \
    } catch (const std::exception &ex) {   \
        g_host_err = ex.what();            \
        return 1;                          \
    }

static PhyloNeighbor *nei(PhyloTree *t, int from, int to) {
    if (from < 0 || to < 0 || from >= t->nodeNum || to >= t->nodeNum) throw std::runtime_error("bad node id");
    PhyloNeighbor *n = t->nodes[from]->findNeighbor(t->nodes[to]);
    if (!n) throw std::runtime_error("nodes are not adjacent");
    return n;
}

extern "C" {

const char *iqhost_last_error(void) { return g_host_err.c_str(); }

int iqhost_create(void **out, const char *newick, const char **names, int nnames) {
    *out = nullptr;
    PhyloTree *t = nullptr;
    try {
        t = new PhyloTree();
        std::vector<std::string> nm;
        for (int i = 0; i < nnames; i++) nm.push_back(names[i]);
        t->readTreeString(newick, nm);
        *out = t;
        return 0;
    } catch (const std::exception &ex) {
        g_host_err = ex.what();
        delete t;
        return 1;
    }
}
void iqhost_destroy(void *h) { delete (PhyloTree *)h; }

int iqhost_set_alignment(void *h, int nstates, int seq_type, int64_t nptn, const uint8_t *states,
                         const double *freq, const double *invar) {
    IQHOST_TRY(((PhyloTree *)h)->setAlignment(nstates, (SeqType)seq_type, nptn, states, freq, invar));
}
int iqhost_set_ptn_freq(void *h, const double *f) { IQHOST_TRY(((PhyloTree *)h)->setPtnFreq(f)); }
int iqhost_set_ptn_invar(void *h, const double *v) { IQHOST_TRY(((PhyloTree *)h)->setPtnInvar(v)); }
int iqhost_set_ascertainment(void *h, int64_t n_unobserved, double nsites) {
    IQHOST_TRY(((PhyloTree *)h)->setAscertainment(n_unobserved, nsites));
}
int iqhost_set_model(void *h, int ncat, const double *eval, const double *evec, const double *inv_evec,
                     const double *rates, const double *props) {
    IQHOST_TRY(((PhyloTree *)h)->setModel(ncat, eval, evec, inv_evec, rates, props));
}
int iqhost_set_mixture_model(void *h, int nclass, int ncat, const int *cat_class, const double *eval, const double *evec,
                             const double *inv_evec, const double *rates, const double *props) {
    IQHOST_TRY(((PhyloTree *)h)->setMixtureModel(nclass, ncat, cat_class, eval, evec, inv_evec, rates, props));
}
int iqhost_set_mem_mode(void *h, int lm) {
    IQHOST_TRY(((PhyloTree *)h)->lh_mem_save = (LhMemSave)lm);
}
int iqhost_set_kernel(void *h, int lk) { IQHOST_TRY(((PhyloTree *)h)->setLikelihoodKernel((LikelihoodKernel)lk)); }
int iqhost_attach_engine(void *h, int device) { IQHOST_TRY(((PhyloTree *)h)->attachEngine(device)); }
int iqhost_attach_engine_sharded(void *h, const int *device_ids, int ndev, int reduce_mode) {
    IQHOST_TRY(((PhyloTree *)h)->attachEngineSharded(device_ids, ndev, reduce_mode));
}
int iqhost_attach_comm(void *h, int nranks, int rank, const void *unique_id) {
    IQHOST_TRY(((PhyloTree *)h)->attachComm(nranks, rank, unique_id));
}
int iqhost_set_device_newton(void *h, int on) { IQHOST_TRY(((PhyloTree *)h)->device_newton = on != 0); }
int iqhost_set_device_sweep(void *h, int on) { IQHOST_TRY(((PhyloTree *)h)->device_sweep = on != 0); }
long iqhost_num_derv_calls(void *h) { return ((PhyloTree *)h)->num_derv_calls; }
int iqhost_set_heavy_first(void *h, int on) { IQHOST_TRY(((PhyloTree *)h)->heavy_first = on != 0); }
int iqhost_set_dry_run(void *h, int on) { IQHOST_TRY(((PhyloTree *)h)->setDryRun(on != 0)); }
void *iqhost_engine(void *h) { return ((PhyloTree *)h)->engine; }
int iqhost_set_allreduce_hook(void *h, void (*fn)(void *, int, void *), void *ctx) {
    IQHOST_TRY(((PhyloTree *)h)->setAllReduceHook(fn, ctx));
}

int iqhost_num_nodes(void *h) { return ((PhyloTree *)h)->nodeNum; }
int iqhost_num_leaves(void *h) { return ((PhyloTree *)h)->leafNum; }
int iqhost_root(void *h) { return ((PhyloTree *)h)->root->id; }
int iqhost_state_unknown(void *h) { return ((PhyloTree *)h)->STATE_UNKNOWN; }
int iqhost_tip_partial_lh(void *h, double *out) {
    PhyloTree *t = (PhyloTree *)h;
    memcpy(out, t->tip_partial_lh.data(), sizeof(double) * t->tip_partial_lh.size());
    return 0;
}
// neighbours of `node` in stored order; returns the degree
int iqhost_neighbors(void *h, int node, int *out_ids, double *out_len, int cap) {
    PhyloTree *t = (PhyloTree *)h;
    if (node < 0 || node >= t->nodeNum) return -1;
    int d = t->nodes[node]->degree();
    for (int i = 0; i < d && i < cap; i++) {
        out_ids[i] = t->nodes[node]->neighbors[i]->node->id;
        if (out_len) out_len[i] = t->nodes[node]->neighbors[i]->length;
    }
    return d;
}
int iqhost_set_branch_length(void *h, int a, int b, double len, int clear_reverse) {
    IQHOST_TRY({
        PhyloTree *t = (PhyloTree *)h;
        nei(t, a, b)->length = len;
        nei(t, b, a)->length = len;
        if (clear_reverse) {
            t->nodes[a]->clearReversePartialLh(t->nodes[b]);
            t->nodes[b]->clearReversePartialLh(t->nodes[a]);
        }
        t->theta_computed = false;
    });
}
// state of the neighbour `from -> to`
int iqhost_neighbor_info(void *h, int from, int to, int *computed, uint64_t *key, double *lh_scale_factor,
                         double *length) {
    IQHOST_TRY({
        PhyloNeighbor *n = nei((PhyloTree *)h, from, to);
        if (computed) *computed = n->partial_lh_computed;
        if (key) *key = n->partial_lh;
        if (lh_scale_factor) *lh_scale_factor = n->lh_scale_factor;
        if (length) *length = n->length;
    });
}

int iqhost_initialize_all_partial_lh(void *h) { IQHOST_TRY(((PhyloTree *)h)->initializeAllPartialLh()); }
int iqhost_clear_all_partial_lh(void *h) { IQHOST_TRY(((PhyloTree *)h)->clearAllPartialLH()); }
// hot loop 1 in one call: clearAllPartialLH(); computeLikelihood()  (model/modelgtr.cpp:510-518)
int iqhost_clear_and_compute_likelihood(void *h, double *lnl) {
    IQHOST_TRY({
        PhyloTree *t = (PhyloTree *)h;
        t->clearAllPartialLH();
        *lnl = t->computeLikelihood(nullptr);
    });
}
int iqhost_compute_likelihood(void *h, double *lnl, double *pattern_lh) {
    IQHOST_TRY(*lnl = ((PhyloTree *)h)->computeLikelihood(pattern_lh));
}
// current_it = from->to as chosen by the last computeLikelihood / optimizeOneBranch
int iqhost_current_branch(void *h, int *from, int *to) {
    PhyloTree *t = (PhyloTree *)h;
    if (!t->current_it) return 1;
    *to = t->current_it->node->id;
    *from = t->current_it_back->node->id;
    return 0;
}
int iqhost_compute_partial(void *h, int dad, int node) {
    IQHOST_TRY({
        PhyloTree *t = (PhyloTree *)h;
        t->computePartialLikelihood(nei(t, dad, node), t->nodes[dad]);
    });
}
int iqhost_compute_branch(void *h, int dad, int node, double *lnl) {
    IQHOST_TRY({
        PhyloTree *t = (PhyloTree *)h;
        *lnl = t->computeLikelihoodBranch(nei(t, dad, node), t->nodes[dad]);
    });
}
int iqhost_compute_derv(void *h, int dad, int node, double *df, double *ddf) {
    IQHOST_TRY({
        PhyloTree *t = (PhyloTree *)h;
        t->current_it = nei(t, dad, node);
        t->current_it_back = nei(t, node, dad);
        t->computeLikelihoodDerv(t->current_it, t->nodes[dad], *df, *ddf);
    });
}
int iqhost_reset_theta(void *h) { IQHOST_TRY(((PhyloTree *)h)->theta_computed = false); }
int iqhost_compute_from_buffer(void *h, double *lnl) {
    IQHOST_TRY(*lnl = ((PhyloTree *)h)->computeLikelihoodFromBuffer());
}
int iqhost_optimize_one_branch(void *h, int a, int b, int clear_lh, int max_nr_step, double *new_len) {
    IQHOST_TRY({
        PhyloTree *t = (PhyloTree *)h;
        t->optimizeOneBranch(t->nodes[a], t->nodes[b], clear_lh != 0, max_nr_step);
        *new_len = nei(t, a, b)->length;
    });
}
int iqhost_optimize_all_branches(void *h, int iterations, double tolerance, int max_nr_step, double *lnl) {
    IQHOST_TRY(*lnl = ((PhyloTree *)h)->optimizeAllBranches(iterations, tolerance, max_nr_step));
}
// both NNI moves around the internal branch (a, b): out[cnt*8 + {0:newloglh, 1:node1_nei, 2:node2_nei, 3..7:newLen}]
int iqhost_nni_for_branch(void *h, int a, int b, int nni5, double *out) {
    IQHOST_TRY({
        PhyloTree *t = (PhyloTree *)h;
        PhyloTree::NNIMove mv[2];
        t->getBestNNIForBran(t->nodes[a], t->nodes[b], nni5 != 0, mv);
        for (int c = 0; c < 2; c++) {
            out[c * 8 + 0] = mv[c].newloglh;
            out[c * 8 + 1] = mv[c].node1_nei;
            out[c * 8 + 2] = mv[c].node2_nei;
            for (int k = 0; k < 5; k++) out[c * 8 + 3 + k] = mv[c].newLen[k];
        }
    });
}
int iqhost_set_branch_bounds(void *h, double minlen, double maxlen) {
    IQHOST_TRY({
        ((PhyloTree *)h)->min_branch_length = minlen;
        ((PhyloTree *)h)->max_branch_length = maxlen;
    });
}
int iqhost_tree_string(void *h, char *out, int cap) {
    std::string s = ((PhyloTree *)h)->getTreeString();
    if ((int)s.size() + 1 > cap) return (int)s.size() + 1;
    memcpy(out, s.c_str(), s.size() + 1);
    return 0;
}

int iqhost_fetch_scale_num(void *h, int from, int to, int16_t *out) {
    IQHOST_TRY({
        PhyloTree *t = (PhyloTree *)h;
        t->fetchScaleNum(nei(t, from, to), out);
    });
}
int iqhost_fetch_partial(void *h, int from, int to, double *out) {
    IQHOST_TRY({
        PhyloTree *t = (PhyloTree *)h;
        t->fetchPartialLh(nei(t, from, to), out);
    });
}
int iqhost_fetch_pattern_lh(void *h, double *out) { IQHOST_TRY(((PhyloTree *)h)->fetchPatternLh(out)); }
int iqhost_compute_pattern_likelihood(void *h, double *out) { IQHOST_TRY(((PhyloTree *)h)->computePatternLikelihood(out)); }
int iqhost_compute_pattern_lh_cat(void *h, double *out) { IQHOST_TRY(((PhyloTree *)h)->computePatternLhCat(out)); }
int iqhost_set_boot_samples(void *h, const float *samples, int nsamples) {
    IQHOST_TRY(((PhyloTree *)h)->setBootSamples(samples, nsamples));
}
int iqhost_compute_rell(void *h, double *out, int cap) {
    IQHOST_TRY({
        std::vector<double> r;
        ((PhyloTree *)h)->computeRELL(r);
        if ((int)r.size() > cap) throw std::runtime_error("output too small");
        memcpy(out, r.data(), r.size() * sizeof(double));
    });
}

// last submitted plan: 7 ints per op {dst_from, dst_to, left_node, right_node, left_leaf, right_leaf, 0}
// plus 2 doubles per op {left_len, right_len}; dst_from->dst_to is the neighbour that was filled
// all nni1 candidates of the tree in one submission: out[k*5 + {0..4}] = node1, node2, node1_nei, node2_nei ids,
// then lens/lnls: newLen[0], newloglh.  Returns the number of candidates in *n.
int iqhost_evaluate_nnis_batch(void *h, int *ids, double *vals, int cap, int *n) {
    IQHOST_TRY({
        std::vector<PhyloTree::NNIMove> mv;
        ((PhyloTree *)h)->evaluateNNIsBatch(mv);
        if ((int)mv.size() > cap) throw std::runtime_error("output too small");
        *n = (int)mv.size();
        for (size_t k = 0; k < mv.size(); k++) {
            ids[4 * k] = mv[k].node1; ids[4 * k + 1] = mv[k].node2;
            ids[4 * k + 2] = mv[k].node1_nei; ids[4 * k + 3] = mv[k].node2_nei;
            vals[2 * k] = mv[k].newLen[0]; vals[2 * k + 1] = mv[k].newloglh;
        }
    });
}
// nni5 batch: vals[6*k + {0..5}] = newLen[0..4], newloglh
int iqhost_evaluate_nnis5_batch(void *h, int *ids, double *vals, int cap, int *n) {
    IQHOST_TRY({
        std::vector<PhyloTree::NNIMove> mv;
        ((PhyloTree *)h)->evaluateNNIs5Batch(mv);
        if ((int)mv.size() > cap) throw std::runtime_error("output too small");
        *n = (int)mv.size();
        for (size_t k = 0; k < mv.size(); k++) {
            ids[4 * k] = mv[k].node1; ids[4 * k + 1] = mv[k].node2;
            ids[4 * k + 2] = mv[k].node1_nei; ids[4 * k + 3] = mv[k].node2_nei;
            for (int i = 0; i < 5; i++) vals[6 * k + i] = mv[k].newLen[i];
            vals[6 * k + 5] = mv[k].newloglh;
        }
    });
}
int iqhost_compute_all_partial_lh(void *h) { IQHOST_TRY(((PhyloTree *)h)->computeAllPartialLh()); }
int iqhost_last_plan(void *h, int *ints, double *lens, uint64_t *keys, int cap) {
    PhyloTree *t = (PhyloTree *)h;
    int n = (int)t->last_plan.size();
    for (int k = 0; k < n && k < cap; k++) {
        const PlanOp &p = t->last_plan[k];
        // find the owner node of dst: the node whose neighbour list contains it
        int from = -1;
        for (PhyloNode *nd : t->nodes)
            for (PhyloNeighbor *nb : nd->neighbors)
                if (nb == p.dst) from = nd->id;
        ints[k * 7 + 0] = from;
        ints[k * 7 + 1] = p.dst ? p.dst->node->id : -1;
        ints[k * 7 + 2] = p.left ? p.left->node->id : -1;
        ints[k * 7 + 3] = p.right ? p.right->node->id : -1;
        ints[k * 7 + 4] = p.op.left_leaf;
        ints[k * 7 + 5] = p.op.right_leaf;
        ints[k * 7 + 6] = (int)p.op.flags;
        if (lens) { lens[k * 2] = p.op.left_len; lens[k * 2 + 1] = p.op.right_len; }
        if (keys) { keys[k * 3] = p.op.dst_key; keys[k * 3 + 1] = p.op.left_key; keys[k * 3 + 2] = p.op.right_key; }
    }
    return n;
}
long iqhost_num_partial_lh_computations(void *h) { return ((PhyloTree *)h)->num_partial_lh_computations; }
long iqhost_num_submissions(void *h) { return ((PhyloTree *)h)->num_submissions; }

}  // extern "C"
