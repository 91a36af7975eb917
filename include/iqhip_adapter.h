/*
 * iqhip_adapter.h -- the ONE implementation of the adapter logic between IQ-TREE's PhyloTree and the C ABI of
 * include/iqhip.h: the lazy post-order recursion with its flag handling and LM_PER_NODE buffer re-orientation
 * (phylokernel.h:70-157), the lh_scale_factor bookkeeping (phylokernel.h:157,395,477), the leaf-side swap of the
 * branch functions (phylokernel.h:491-498,739-746) and the bodies of the four kernels that PhyloTree's member-function
 * pointers select (phylotreesse.cpp:335-357), plus optimizeOneBranch's Newton solve as one engine call.
 *
 * It is a header of templates over a policy class X that names the tree types and answers a handful of questions
 * about them, so that the SAME code is
 *   - instantiated with the reference's own PhyloTree / PhyloNode / PhyloNeighbor (phylonode.h:26-127) in
 *     integration/phylotree_hip.cpp -- the file an IQ-TREE maintainer adds next to phylotreesse.cpp -- and
 *   - instantiated with this repository's stand-alone mirror of that slice (iq-tree_amd/host/phylo_host.{h,cpp}),
 *     which is what libiqhost.so is built from and what every test drives.
 * Plain C++98 (the reference builds with -std=gnu++98): no auto, no lambdas, no nullptr.
 *
 * What X must provide (all static):
 *   typedef ... Tree, Node, Neighbor;
 *   Node *node(Neighbor *);  double length(Neighbor *);  void setLength(Neighbor *, double);
 *   bool isLeaf(Node *);  int degree(Node *);  int leafId(Node *);          // leafId = row of the alignment
 *   int numNeighbors(Node *);  Neighbor *neighborAt(Node *, int);  Neighbor *findNeighbor(Node *at, Node *to);
 *   int &computed(Neighbor *);            // partial_lh_computed (bit 0)
 *   double &scaleFactor(Neighbor *);      // lh_scale_factor
 *   uint64_t key(Neighbor *);             // VALUE of the partial_lh pointer, 0 = NULL: the device vector's key
 *   void stealBuffer(Neighbor *to, Neighbor *from);   // move partial_lh (+ scale_num) from `from` to `to`
 *   bool perNodeMode(Tree *);             // params->lh_mem_save == LM_PER_NODE
 *   bool heavyFirst(Tree *);              // plan the subtree with more pending updates first (any order is valid)
 *   void ensureBuffers(Tree *);           // if (!central_partial_lh) initializeAllPartialLh();
 *   void sync(Tree *);                    // push model / alignment changes to the engine (hipSync / pushInputs)
 *   iqhip_engine *engine(Tree *);
 *   void fail(Tree *, const char *what, const char *detail);   // outError() / throw; must not return
 *   void countComputation(Tree *);        // num_partial_lh_computations++ (phylokernel.h:85)
 *   bool &thetaComputed(Tree *);  Neighbor *currentIt(Tree *);  Neighbor *currentItBack(Tree *);
 *   double minBranchLength(Tree *), maxBranchLength(Tree *);
 *   // engine calls; iqhip_adapter::EngineCalls<X, Tree> has the defaults (straight C-ABI calls, status -> X::fail)
 *   void updatePartials(Tree *, Plan<X> &, double *sum_scale);
 *   double traverseLnl(Tree *, Plan<X> &, iqhip_branch_end a, iqhip_branch_end b, double len, double *sum_scale);
 *   void computeTheta(Tree *, iqhip_branch_end a, iqhip_branch_end b);
 *   void derv(Tree *, double len, double &df, double &ddf);
 *   double lnlFromTheta(Tree *, double len);
 *   double optimizeBranch(Tree *, Plan<X> &, iqhip_branch_end a, iqhip_branch_end b, double xguess, int max_steps,
 *                         double *sum_scale, int *nsteps);
 *   // only for optimizeBranchSweep:
 *   void clearReversePartialLh(Node *node, Node *dad);      // PhyloNode::clearReversePartialLh (phylonode.cpp:43-52)
 *   void setCurrent(Tree *, Neighbor *it, Neighbor *back);  // current_it / current_it_back
 *   void optimizeSweep(Tree *, const iqhip_sweep_step *, int nsteps, int max_steps, double diverge_frac, double *sum_scale,
 *                      iqhip_branch_result *);
 */
#ifndef IQHIP_ADAPTER_H_
#define IQHIP_ADAPTER_H_

#include <stdint.h>
#include <string.h>

#include <vector>

#include "iqhip.h"

namespace iqhip_adapter {

/* the node updates one engine submission will execute, with the neighbours they answer */
template <class X>
struct Plan {
    std::vector<iqhip_node_op> ops;
    std::vector<typename X::Neighbor *> dst; /* per op; NULL for an intermediate product (degree > 3) */
    /* per op: the child neighbours of dst's node (none for an intermediate product), flat: a traversal is planned
     * anew for every evaluation, and a vector per op (plus two per node in collectPlan) made the planning of a
     * 50-taxon traversal cost 250 heap calls, most of its 13 us */
    std::vector<typename X::Neighbor *> kid_flat;
    std::vector<uint32_t> kid_off; /* [size() + 1] */
    /* per op: the child neighbours whose branch lengths are ops[k].left_len / right_len ([2k], [2k+1]; NULL for the
     * zero-length hand-over of a multifurcating node's running product) -- what a sweep needs to refer to lengths that
     * are not known yet (optimizeBranchSweep) */
    std::vector<typename X::Neighbor *> len_nb;
    Plan() {
        ops.reserve(64);
        dst.reserve(64);
        kid_flat.reserve(128);
        kid_off.reserve(65);
        kid_off.push_back(0);
    }
    bool empty() const { return ops.empty(); }
    size_t size() const { return ops.size(); }
    size_t nkids(size_t k) const { return kid_off[k + 1] - kid_off[k]; }
    typename X::Neighbor *kid(size_t k, size_t i) const { return kid_flat[kid_off[k] + i]; }
};

/* Key of the intermediate product number i of the multifurcating node whose vector has key dst (see collectPlan):
 * bit 63 set -- never the value of a user-space pointer, never one of the mirror's small counters. */
inline uint64_t tempKey(uint64_t dst, int i) { return ((uint64_t)1 << 63) | (dst << 5) | (uint64_t)(i & 31); }

template <class X>
inline iqhip_branch_end branchEnd(typename X::Neighbor *nei) {
    iqhip_branch_end e;
    e._pad = 0;
    if (X::isLeaf(X::node(nei))) {
        e.key = 0;
        e.leaf = X::leafId(X::node(nei));
    } else {
        e.key = X::key(nei);
        e.leaf = -1;
    }
    return e;
}

template <class X>
inline int countPending(typename X::Neighbor *nei, typename X::Node *dad) {
    typename X::Node *node = X::node(nei);
    if ((X::computed(nei) & 1) || X::isLeaf(node)) return 0;
    int n = 1;
    for (int k = 0; k < X::numNeighbors(node); k++) {
        typename X::Neighbor *nb = X::neighborAt(node, k);
        if (X::node(nb) != dad) n += countPending<X>(nb, node);
    }
    return n;
}

template <class X>
inline void setChild(iqhip_node_op &op, bool left, typename X::Neighbor *child) {
    const bool leaf = X::isLeaf(X::node(child));
    if (left) {
        op.left_leaf = leaf ? X::leafId(X::node(child)) : -1;
        op.left_key = leaf ? 0 : X::key(child);
        op.left_len = X::length(child);
    } else {
        op.right_leaf = leaf ? X::leafId(X::node(child)) : -1;
        op.right_key = leaf ? 0 : X::key(child);
        op.right_len = X::length(child);
    }
}

/* phylokernel.h:70-157 with the three pattern loops replaced by "append one op".  A node of degree > 3 (the
 * reference's scalar kernel, phylotreesse.cpp:702-806: the product over ALL children, then U^-1, then ONE scaling
 * test) becomes a chain of binary ops: the running product is carried through intermediate vectors over branches of
 * length 0 (E(0) = U, so U * (U^-1 * T) = T) that are never rescaled; only the node's own op applies the test. */
template <class X>
inline void collectPlan(typename X::Tree *tree, typename X::Neighbor *dad_branch, typename X::Node *dad, Plan<X> &plan) {
    typedef typename X::Neighbor Neighbor;
    typedef typename X::Node Node;
    if (X::computed(dad_branch) & 1) return; /* don't recompute the likelihood (:81-82) */
    X::computed(dad_branch) |= 1;
    X::countComputation(tree);
    Node *node = X::node(dad_branch);
    if (X::isLeaf(node)) { /* :93-97 */
        X::scaleFactor(dad_branch) = 0.0;
        return;
    }
    struct KidList { /* at most 17 children (tempKey numbers the intermediate products with 5 bits) */
        Neighbor *v[18];
        size_t n;
        size_t size() const { return n; }
        Neighbor *&operator[](size_t i) { return v[i]; }
    };
    KidList kids;
    kids.n = 0;
    for (int k = 0; k < X::numNeighbors(node); k++) {
        Neighbor *nb = X::neighborAt(node, k);
        if (X::node(nb) != dad) {
            if (kids.n >= 17) X::fail(tree, "collectPlan", "node of degree > 18");
            kids.v[kids.n++] = nb;
        }
    }
    if (kids.size() < 2) X::fail(tree, "collectPlan", "internal node with fewer than two children");
    if (kids.size() == 2) {
        if (!X::isLeaf(X::node(kids[0])) && X::isLeaf(X::node(kids[1]))) { /* :116-121 */
            Neighbor *t = kids[0];
            kids[0] = kids[1];
            kids[1] = t;
        }
    }
    /* children first (:122-125).  Any order of independent subtrees gives the same numbers (every update is a pure
     * function of its children); with heavyFirst the subtree with more pending updates goes first, so that its
     * result waits only for the short one before it is consumed (register / cache residency in the engine). */
    KidList order = kids;
    if (X::heavyFirst(tree)) {
        for (size_t i = 1; i < order.size(); i++) /* stable insertion sort, descending pending count */
            for (size_t j = i; j > 0 && countPending<X>(order[j], node) > countPending<X>(order[j - 1], node); j--) {
                Neighbor *t = order[j];
                order[j] = order[j - 1];
                order[j - 1] = t;
            }
    }
    for (size_t i = 0; i < order.size(); i++)
        if ((X::computed(order[i]) & 1) == 0) collectPlan<X>(tree, order[i], node, plan);

    if (X::perNodeMode(tree) && X::key(dad_branch) == 0) {
        /* re-orient partial_lh (:127-143): steal the vector of a child-side back neighbour */
        bool done = false;
        for (size_t i = 0; i < kids.size() && !done; i++) {
            Neighbor *backnei = X::findNeighbor(X::node(kids[i]), node);
            if (X::key(backnei) != 0) {
                X::stealBuffer(dad_branch, backnei);
                X::computed(backnei) &= ~1;
                done = true;
            }
        }
        if (!done) X::fail(tree, "collectPlan", "partial_lh is not re-oriented");
    }
    if (X::key(dad_branch) == 0) X::fail(tree, "collectPlan", "neighbor has no partial_lh buffer");

    const uint64_t dst = X::key(dad_branch);
    for (size_t i = 1; i < kids.size(); i++) {
        const bool last = (i + 1 == kids.size());
        iqhip_node_op op;
        memset(&op, 0, sizeof(op));
        if (i == 1) {
            setChild<X>(op, true, kids[0]);
            plan.len_nb.push_back(kids[0]);
        } else { /* the running product of children 0..i-1 */
            op.left_leaf = -1;
            op.left_key = tempKey(dst, (int)i - 2);
            op.left_len = 0.0;
            plan.len_nb.push_back((Neighbor *)0);
        }
        setChild<X>(op, false, kids[i]);
        plan.len_nb.push_back(kids[i]);
        op.dst_key = last ? dst : tempKey(dst, (int)i - 1);
        /* (a node of degree > 3 is the scalar kernel's, with its lh_max == 0 branch, phylotreesse.cpp:774-788) */
        op.flags = last ? (kids.size() > 2 ? (uint32_t)IQHIP_OP_SCALAR_RULE : 0u) : (uint32_t)IQHIP_OP_NO_SCALE;
        plan.ops.push_back(op);
        plan.dst.push_back(last ? dad_branch : (Neighbor *)0);
        if (last)
            for (size_t q = 0; q < kids.size(); q++) plan.kid_flat.push_back(kids[q]);
        plan.kid_off.push_back((uint32_t)plan.kid_flat.size());
    }
}

/* dad_branch->lh_scale_factor = sum over the children (:157) + the node's own sum_scale (:395,477), in plan order
 * (children before parents).  The intermediate products of a multifurcating node carry no scaling events. */
template <class X>
inline void applyScaleFactors(Plan<X> &plan, const double *sum_scale) {
    for (size_t k = 0; k < plan.ops.size(); k++) {
        if (!plan.dst[k]) continue;
        double s = 0.0;
        for (size_t i = 0; i < plan.nkids(k); i++) s += X::scaleFactor(plan.kid(k, i));
        X::scaleFactor(plan.dst[k]) = s + sum_scale[k];
    }
}

/* the reference puts the leaf (if any) on the `dad` side of a branch (phylokernel.h:491-498, 739-746) */
template <class X>
inline void orientBranch(typename X::Neighbor *&dad_branch, typename X::Node *&dad, typename X::Neighbor *&node_branch,
                         typename X::Node *&node) {
    node = X::node(dad_branch);
    node_branch = X::findNeighbor(node, dad);
    if (X::isLeaf(node)) {
        typename X::Node *tn = dad;
        dad = node;
        node = tn;
        typename X::Neighbor *tb = dad_branch;
        dad_branch = node_branch;
        node_branch = tb;
    }
}

/* default engine calls: the C ABI, a non-zero status goes to X::fail with the engine's message.  A policy inherits them
 * as  struct MyPolicy : iqhip_adapter::EngineCalls<MyPolicy, MyTree> { ... }  (the tree type is a parameter of its own
 * because MyPolicy is still incomplete where its base class is instantiated). */
template <class X, class Tree>
struct EngineCalls {
    static void chk(Tree *t, int rc, const char *what) {
        if (rc != IQHIP_OK) X::fail(t, what, iqhip_last_error());
    }
    static void updatePartials(Tree *t, Plan<X> &plan, double *sum_scale) {
        chk(t, iqhip_update_partials(X::engine(t), &plan.ops[0], (int)plan.ops.size(), sum_scale), "iqhip_update_partials");
    }
    static double traverseLnl(Tree *t, Plan<X> &plan, iqhip_branch_end a, iqhip_branch_end b, double len, double *sum_scale) {
        double lnl = 0.0;
        chk(t, iqhip_traverse_lnl(X::engine(t), plan.empty() ? (const iqhip_node_op *)0 : &plan.ops[0], (int)plan.ops.size(),
                                  a, b, len, sum_scale, &lnl),
            "iqhip_traverse_lnl");
        return lnl;
    }
    static void computeTheta(Tree *t, iqhip_branch_end a, iqhip_branch_end b) {
        chk(t, iqhip_compute_theta(X::engine(t), a, b), "iqhip_compute_theta");
    }
    static void derv(Tree *t, double len, double &df, double &ddf) {
        chk(t, iqhip_derv(X::engine(t), len, &df, &ddf), "iqhip_derv");
    }
    static double lnlFromTheta(Tree *t, double len) {
        double lnl = 0.0;
        chk(t, iqhip_lnl_from_theta(X::engine(t), len, &lnl), "iqhip_lnl_from_theta");
        return lnl;
    }
    static double optimizeBranch(Tree *t, Plan<X> &plan, iqhip_branch_end a, iqhip_branch_end b, double xguess, int max_steps,
                                 double *sum_scale, int *nsteps) {
        double optx = xguess, d2l = 0.0;
        chk(t, iqhip_optimize_branch(X::engine(t), plan.empty() ? (const iqhip_node_op *)0 : &plan.ops[0], (int)plan.ops.size(),
                                     a, b, xguess, X::minBranchLength(t), X::maxBranchLength(t), X::minBranchLength(t),
                                     max_steps, sum_scale, &optx, &d2l, nsteps),
            "iqhip_optimize_branch");
        return optx;
    }
    static void optimizeSweep(Tree *t, const iqhip_sweep_step *steps, int nsteps, int max_steps, double diverge_frac,
                              double *sum_scale, iqhip_branch_result *results) {
        chk(t, iqhip_optimize_sweep(X::engine(t), steps, nsteps, X::minBranchLength(t), X::maxBranchLength(t),
                                    X::minBranchLength(t), max_steps, diverge_frac, sum_scale, results),
            "iqhip_optimize_sweep");
    }
};

/* ---- the four kernels behind PhyloTree's member-function pointers ------------------------------------------- */

/* computePartialLikelihoodEigenSIMD (phylokernel.h:70-483) */
template <class X>
inline void computePartialLikelihood(typename X::Tree *tree, typename X::Neighbor *dad_branch, typename X::Node *dad,
                                     Plan<X> *plan_out = 0) {
    X::ensureBuffers(tree);
    Plan<X> plan;
    collectPlan<X>(tree, dad_branch, dad, plan);
    if (!plan.empty()) {
        std::vector<double> sum_scale(plan.size(), 0.0);
        X::sync(tree);
        X::updatePartials(tree, plan, &sum_scale[0]);
        applyScaleFactors<X>(plan, &sum_scale[0]);
    }
    if (plan_out) *plan_out = plan;
}

/* computeLikelihoodBranchEigenSIMD (phylokernel.h:733-1020): pending node updates of both ends + the branch lnL in
 * one submission; returns tree_lh including both lh_scale_factors (:751) */
template <class X>
inline double computeLikelihoodBranch(typename X::Tree *tree, typename X::Neighbor *dad_branch, typename X::Node *dad,
                                      Plan<X> *plan_out = 0) {
    typename X::Node *node;
    typename X::Neighbor *node_branch;
    X::ensureBuffers(tree);
    orientBranch<X>(dad_branch, dad, node_branch, node);
    Plan<X> plan;
    if ((X::computed(dad_branch) & 1) == 0) collectPlan<X>(tree, dad_branch, dad, plan);
    if ((X::computed(node_branch) & 1) == 0) collectPlan<X>(tree, node_branch, node, plan);
    std::vector<double> sum_scale(plan.size() + 1, 0.0);
    X::sync(tree);
    /* dad_branch points at `node`'s subtree; node_branch at `dad`'s (a leaf after the swap) */
    const double lnl = X::traverseLnl(tree, plan, branchEnd<X>(node_branch), branchEnd<X>(dad_branch), X::length(dad_branch),
                                      &sum_scale[0]);
    applyScaleFactors<X>(plan, &sum_scale[0]);
    if (plan_out) *plan_out = plan;
    return X::scaleFactor(node_branch) + X::scaleFactor(dad_branch) + lnl;
}

/* computeLikelihoodDervEigenSIMD (phylokernel.h:485-730) */
template <class X>
inline void computeLikelihoodDerv(typename X::Tree *tree, typename X::Neighbor *dad_branch, typename X::Node *dad, double &df,
                                  double &ddf) {
    typename X::Node *node;
    typename X::Neighbor *node_branch;
    X::ensureBuffers(tree);
    orientBranch<X>(dad_branch, dad, node_branch, node);
    if ((X::computed(dad_branch) & 1) == 0) computePartialLikelihood<X>(tree, dad_branch, dad);
    if ((X::computed(node_branch) & 1) == 0) computePartialLikelihood<X>(tree, node_branch, node);
    df = ddf = 0.0;
    X::sync(tree);
    if (!X::thetaComputed(tree)) { /* :535-579 */
        X::thetaComputed(tree) = true;
        X::computeTheta(tree, branchEnd<X>(node_branch), branchEnd<X>(dad_branch));
    }
    X::derv(tree, X::length(dad_branch), df, ddf);
}

/* computeLikelihoodFromBufferEigenSIMD (phylokernel.h:1022-1192) */
template <class X>
inline double computeLikelihoodFromBuffer(typename X::Tree *tree) {
    typename X::Neighbor *it = X::currentIt(tree), *back = X::currentItBack(tree);
    if (!X::thetaComputed(tree)) X::fail(tree, "computeLikelihoodFromBuffer", "theta not computed");
    const double lnl = X::lnlFromTheta(tree, X::length(it));
    return X::scaleFactor(it) + X::scaleFactor(back) + lnl; /* :1028 */
}

/* optimizeOneBranch's `optx = minimizeNewton(...)` (phylotree.cpp:2148-2192, optimization.cpp:388-465) as ONE engine
 * call: pending node updates of both ends + theta + the whole solve.  current_it / current_it_back name the branch.
 * Like the reference's loop, the call leaves the branch at the LAST EVALUATED length: computeFuncDerv
 * (phylotree.cpp:2135-2137) stores every trial length into current_it / current_it_back, and on every return path of
 * minimizeNewton the value returned is the point evaluated last (`return rts_old` is taken before the new rts is
 * evaluated; the `xl == rts` / `temp == rts` exits return a point that equals it to the last bit or ulp).  The
 * diverged-Newton test that follows in optimizeOneBranch (phylotree.cpp:2167-2176) evaluates opt_lh at that length. */
template <class X>
inline double minimizeNewtonOnBranch(typename X::Tree *tree, double current_len, int max_steps, int *nsteps = 0,
                                     Plan<X> *plan_out = 0) {
    typename X::Neighbor *dad_branch = X::currentIt(tree), *node_branch;
    typename X::Node *dad = X::node(X::currentItBack(tree)), *node;
    X::ensureBuffers(tree);
    orientBranch<X>(dad_branch, dad, node_branch, node);
    Plan<X> plan;
    if ((X::computed(dad_branch) & 1) == 0) collectPlan<X>(tree, dad_branch, dad, plan);
    if ((X::computed(node_branch) & 1) == 0) collectPlan<X>(tree, node_branch, node, plan);
    std::vector<double> sum_scale(plan.size() + 1, 0.0);
    int steps = 0;
    X::sync(tree);
    X::thetaComputed(tree) = true;
    const double optx = X::optimizeBranch(tree, plan, branchEnd<X>(node_branch), branchEnd<X>(dad_branch), current_len,
                                          max_steps, &sum_scale[0], &steps);
    applyScaleFactors<X>(plan, &sum_scale[0]);
    X::setLength(X::currentIt(tree), optx);
    X::setLength(X::currentItBack(tree), optx);
    if (nsteps) *nsteps = steps;
    if (plan_out) *plan_out = plan;
    return optx;
}

/* PhyloTree::optimizeAllBranches' inner loop (phylotree.cpp:2285-2290)
 *     for (j = 0; j < nodes1.size(); j++) optimizeOneBranch(nodes1[j], nodes2[j], true, maxNRStep);
 * as ONE engine submission (iqhip_optimize_sweep).  The plans of all branches are collected up front with the reference's
 * own flag logic, pretending that every optimizeOneBranch changes its branch -- which is what makes it call
 * clearReversePartialLh on both sides (phylotree.cpp:2186-2189); a child branch optimised earlier in the sweep is
 * referred to by its step number (its length only exists on the device until the sweep returns).  Afterwards the
 * lengths, lh_scale_factors, current_it / current_it_back and theta_computed are what the loop would have left
 * (theta_all is that of the last branch).  Needs two more policy functions: X::clearReversePartialLh(node, dad) and
 * X::setCurrent(tree, it, back); X::optimizeSweep has a default in EngineCalls. */
template <class X>
inline void optimizeBranchSweep(typename X::Tree *tree, typename X::Node *const *nodes1, typename X::Node *const *nodes2,
                                int nbranches, int max_steps, double diverge_frac, int *nevals = 0) {
    typedef typename X::Neighbor Neighbor;
    typedef typename X::Node Node;
    if (nbranches <= 0) return;
    X::ensureBuffers(tree);
    X::sync(tree);
    std::vector<Plan<X> > plans((size_t)nbranches);
    std::vector<std::vector<int32_t> > len_from((size_t)nbranches);
    std::vector<iqhip_sweep_step> steps((size_t)nbranches);
    std::vector<Neighbor *> opt_nb; /* neighbours (both directions) of the branches optimised so far, with their step */
    std::vector<int32_t> opt_step;
    size_t total_ops = 0;
    for (int j = 0; j < nbranches; j++) {
        Neighbor *it = X::findNeighbor(nodes1[j], nodes2[j]), *back = X::findNeighbor(nodes2[j], nodes1[j]);
        Neighbor *dad_branch = it, *node_branch; /* as optimizeOneBranch: current_it = node1's neighbour towards node2 */
        Node *dad = nodes1[j], *node;
        orientBranch<X>(dad_branch, dad, node_branch, node);
        Plan<X> &plan = plans[(size_t)j];
        if ((X::computed(dad_branch) & 1) == 0) collectPlan<X>(tree, dad_branch, dad, plan);
        if ((X::computed(node_branch) & 1) == 0) collectPlan<X>(tree, node_branch, node, plan);
        std::vector<int32_t> &lf = len_from[(size_t)j];
        lf.assign(2 * plan.size(), -1);
        for (size_t q = 0; q < lf.size(); q++)
            for (size_t o = 0; o < opt_nb.size(); o++)
                if (plan.len_nb[q] == opt_nb[o]) lf[q] = opt_step[o];
        iqhip_sweep_step &st = steps[(size_t)j];
        st.ops = plan.empty() ? (const iqhip_node_op *)0 : &plan.ops[0];
        st.len_from = lf.empty() ? (const int32_t *)0 : &lf[0];
        st.nops = (int32_t)plan.size();
        st._pad = 0;
        st.a = branchEnd<X>(node_branch);
        st.b = branchEnd<X>(dad_branch);
        st.xguess = X::length(it);
        total_ops += plan.size();
        opt_nb.push_back(it);
        opt_step.push_back(j);
        opt_nb.push_back(back);
        opt_step.push_back(j);
        /* optimizeOneBranch with clearLH and a changed length (phylotree.cpp:2186-2189) */
        X::clearReversePartialLh(nodes1[j], nodes2[j]);
        X::clearReversePartialLh(nodes2[j], nodes1[j]);
    }
    std::vector<double> sum_scale(total_ops + 1, 0.0);
    std::vector<iqhip_branch_result> res((size_t)nbranches);
    X::optimizeSweep(tree, &steps[0], nbranches, max_steps, diverge_frac, &sum_scale[0], &res[0]);
    size_t off = 0;
    int evals = 0;
    for (int j = 0; j < nbranches; j++) {
        applyScaleFactors<X>(plans[(size_t)j], &sum_scale[off]);
        off += plans[(size_t)j].size();
        X::setLength(X::findNeighbor(nodes1[j], nodes2[j]), res[(size_t)j].optx);
        X::setLength(X::findNeighbor(nodes2[j], nodes1[j]), res[(size_t)j].optx);
        evals += res[(size_t)j].nsteps;
    }
    X::setCurrent(tree, X::findNeighbor(nodes1[nbranches - 1], nodes2[nbranches - 1]),
                  X::findNeighbor(nodes2[nbranches - 1], nodes1[nbranches - 1]));
    X::thetaComputed(tree) = true;
    if (nevals) *nevals = evals;
}

}  /* namespace iqhip_adapter */

#endif /* IQHIP_ADAPTER_H_ */
