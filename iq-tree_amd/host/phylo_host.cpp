// phylo_host.cpp -- see phylo_host.h.  Host logic only; all arithmetic on partial-likelihood
// vectors happens in libiqhip.so.
#include <time.h>
#include "phylo_host.h"

#include <assert.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <sstream>

namespace iqhost {

// log(2^-256), phylotree.h:53
static const double LOG_SCALING_THRESHOLD = log(0x1p-256);

// =========================================================================================
// flags (phylonode.cpp:15-65)
// =========================================================================================
void PhyloNeighbor::clearForwardPartialLh(PhyloNode *dad) {
    clearPartialLh();
    for (PhyloNeighbor *nb : node->neighbors)
        if (nb->node != dad) nb->clearForwardPartialLh(node);
}

void PhyloNode::clearReversePartialLh(PhyloNode *dad) {
    for (PhyloNeighbor *nb : neighbors)
        if (nb->node != dad) {
            nb->node->findNeighbor(this)->partial_lh_computed = 0;
            nb->node->clearReversePartialLh(this);
        }
}

void PhyloNode::clearAllPartialLh(bool make_null, PhyloNode *dad) {
    PhyloNeighbor *nei = findNeighbor(dad);
    nei->partial_lh_computed = 0;
    if (make_null) nei->partial_lh = 0;
    nei = dad->findNeighbor(this);
    nei->partial_lh_computed = 0;
    if (make_null) nei->partial_lh = 0;
    for (PhyloNeighbor *nb : neighbors)
        if (nb->node != dad) nb->node->clearAllPartialLh(make_null, this);
}

// =========================================================================================
// tree
// =========================================================================================
PhyloTree::PhyloTree() { setLikelihoodKernel(LK_EIGEN_HIP); }

PhyloTree::~PhyloTree() {
    if (engine) iqhip_destroy(engine);
    freeTree();
}

void PhyloTree::freeTree() {
    for (PhyloNeighbor *nb : all_neighbors) delete nb;
    for (PhyloNode *n : nodes) delete n;
    all_neighbors.clear();
    nodes.clear();
    root = nullptr;
    current_it = current_it_back = nullptr;
}

namespace {
struct NwkNode {
    std::string label;
    double len = 0.0;
    bool has_len = false;
    std::vector<NwkNode *> kids;
    ~NwkNode() { for (NwkNode *k : kids) delete k; }
};

struct NwkParser {
    const std::string &s;
    size_t pos = 0;
    explicit NwkParser(const std::string &str) : s(str) {}
    void skip() { while (pos < s.size() && isspace((unsigned char)s[pos])) pos++; }
    NwkNode *parse() {
        skip();
        NwkNode *n = new NwkNode();
        if (pos < s.size() && s[pos] == '(') {
            pos++;
            for (;;) {
                n->kids.push_back(parse());
                skip();
                if (pos >= s.size()) { delete n; throw std::runtime_error("newick: unexpected end"); }
                if (s[pos] == ',') { pos++; continue; }
                if (s[pos] == ')') { pos++; break; }
                delete n;
                throw std::runtime_error("newick: expected ',' or ')'");
            }
        }
        skip();
        size_t b = pos;
        while (pos < s.size() && !strchr(":,();", s[pos]) && !isspace((unsigned char)s[pos])) pos++;
        n->label = s.substr(b, pos - b);
        skip();
        if (pos < s.size() && s[pos] == ':') {
            pos++;
            skip();
            char *end = nullptr;
            n->len = strtod(s.c_str() + pos, &end);
            if (end == s.c_str() + pos) { delete n; throw std::runtime_error("newick: bad branch length"); }
            pos = end - s.c_str();
            n->has_len = true;
        }
        return n;
    }
};
}  // namespace

void PhyloTree::readTreeString(const std::string &newick, const std::vector<std::string> &names) {
    deleteAllPartialLh();
    freeTree();
    NwkParser P(newick);
    NwkNode *top = P.parse();
    struct Guard { NwkNode *p; ~Guard() { delete p; } } guard{top};

    // count leaves
    std::vector<NwkNode *> stack{top};
    int nleaf = 0;
    while (!stack.empty()) {
        NwkNode *n = stack.back();
        stack.pop_back();
        if (n->kids.empty()) nleaf++;
        for (NwkNode *k : n->kids) stack.push_back(k);
    }
    if (nleaf < 3) throw std::runtime_error("tree needs at least 3 taxa");
    if (!names.empty() && (int)names.size() != nleaf)
        throw std::runtime_error("newick leaf count differs from the number of sequences");
    leafNum = nleaf;
    nodes.assign(nleaf, nullptr);
    int branch_id = 0;

    auto connect = [&](PhyloNode *a, PhyloNode *b, double len) {
        PhyloNeighbor *ab = new PhyloNeighbor(), *ba = new PhyloNeighbor();
        ab->node = b; ab->length = len; ab->id = branch_id;
        ba->node = a; ba->length = len; ba->id = branch_id;
        branch_id++;
        a->neighbors.push_back(ab);
        b->neighbors.push_back(ba);
        all_neighbors.push_back(ab);
        all_neighbors.push_back(ba);
    };
    auto leaf_id = [&](const std::string &label) -> int {
        if (names.empty()) {
            char *end = nullptr;
            long v = strtol(label.c_str(), &end, 10);
            if (label.empty() || *end) throw std::runtime_error("leaf label '" + label + "' is not a taxon id");
            return (int)v;
        }
        for (size_t i = 0; i < names.size(); i++)
            if (names[i] == label) return (int)i;
        throw std::runtime_error("leaf label '" + label + "' not found in alignment");
    };
    // recursive build; returns the PhyloNode of a newick node
    struct Builder {
        PhyloTree *t; decltype(connect) &conn; decltype(leaf_id) &lid;
        PhyloNode *build(NwkNode *n) {
            PhyloNode *pn = new PhyloNode();
            if (n->kids.empty()) {
                int id = lid(n->label);
                if (id < 0 || id >= t->leafNum || t->nodes[id]) {
                    delete pn;
                    throw std::runtime_error("duplicate or out-of-range taxon '" + n->label + "'");
                }
                pn->id = id;
                pn->name = n->label;
                t->nodes[id] = pn;
            } else {
                pn->id = (int)t->nodes.size();
                pn->name = n->label;
                t->nodes.push_back(pn);
                for (NwkNode *k : n->kids) {
                    PhyloNode *c = build(k);
                    conn(pn, c, k->len);
                }
            }
            return pn;
        }
    } builder{this, connect, leaf_id};

    if (top->kids.size() == 2) {
        // rooted input: drop the degree-2 root, join its two children by one branch
        NwkNode *a = top->kids[0], *b = top->kids[1];
        if (a->kids.empty() && b->kids.empty()) throw std::runtime_error("tree needs at least 3 taxa");
        PhyloNode *na = builder.build(a), *nb = builder.build(b);
        connect(na, nb, a->len + b->len);
    } else {
        builder.build(top);
    }
    for (PhyloNode *n : nodes)
        if (!n) throw std::runtime_error("a taxon of the alignment is missing from the tree");
    nodeNum = (int)nodes.size();
    branchNum = branch_id;
    for (PhyloNode *n : nodes)
        if (!n->isLeaf() && n->degree() < 3) throw std::runtime_error("internal node of degree < 3");
    root = nodes[0];  // the reference roots at the first taxon by default (setRootNode)
    current_it = current_it_back = nullptr;
    theta_computed = false;
}

static void writeNewick(std::ostringstream &os, const PhyloNode *node, const PhyloNode *dad) {
    if (node->isLeaf() && dad) {
        if (node->name.empty()) os << node->id;  // leaf labels as read (MTree::printTree writes names)
        else os << node->name;
        return;
    }
    os << "(";
    bool first = true;
    for (PhyloNeighbor *nb : node->neighbors)
        if (nb->node != dad) {
            if (!first) os << ",";
            first = false;
            writeNewick(os, nb->node, node);
            os.precision(17);
            os << ":" << nb->length;
        }
    os << ")";
}

std::string PhyloTree::getTreeString() const {
    std::ostringstream os;
    if (!root) return ";";
    // print from the internal node next to the root leaf so the output is a trifurcation
    const PhyloNode *start = root->neighbors[0]->node;
    writeNewick(os, start, nullptr);
    os << ";";
    return os.str();
}

PhyloNode *PhyloTree::findFarthestLeaf(PhyloNode *node, PhyloNode *dad) {
    if (!node) node = root;
    if (dad && node->isLeaf()) {
        node->height = 0.0;
        return node;
    }
    PhyloNode *res = nullptr;
    node->height = 0.0;
    for (PhyloNeighbor *nb : node->neighbors)
        if (nb->node != dad) {
            PhyloNode *leaf = findFarthestLeaf(nb->node, node);
            if (node->height < nb->node->height + 1) {
                node->height = nb->node->height + 1;
                res = leaf;
            }
        }
    return res;
}

// =========================================================================================
// inputs
// =========================================================================================
void PhyloTree::setAlignment(int nstates, SeqType st, int64_t nptn_, const uint8_t *states,
                             const double *freq, const double *invar) {
    if (engine) throw std::runtime_error("setAlignment after attachEngine is not supported");
    if (leafNum <= 0) throw std::runtime_error("readTreeString first");
    num_states = nstates;
    seq_type = st;
    nptn = nptn_;
    // alignment.cpp:470-472
    STATE_UNKNOWN = (st == SEQ_DNA && nstates == 4) ? 18 : (st == SEQ_PROTEIN && nstates == 20) ? 23 : nstates;
    aln_states.assign(states, states + (size_t)leafNum * nptn);
    ptn_freq.assign(freq, freq + nptn);
    ptn_invar.assign(invar, invar + nptn);
    inputs_dirty = aln_dirty = true;
}

void PhyloTree::setPtnFreq(const double *f) {
    if (nptn <= 0) throw std::runtime_error("setAlignment first");
    ptn_freq.assign(f, f + nptn);
    inputs_dirty = weights_dirty = true;
}

void PhyloTree::setPtnInvar(const double *v) {
    if (nptn <= 0) throw std::runtime_error("setAlignment first");
    ptn_invar.assign(v, v + nptn);
    inputs_dirty = weights_dirty = true;
}

void PhyloTree::setAscertainment(int64_t n_unobs, double nsites) {
    if (n_unobs < 0 || n_unobs >= nptn) throw std::runtime_error("setAscertainment: bad pattern count");
    n_unobserved = n_unobs;
    asc_nsites = nsites;
    inputs_dirty = aln_dirty = true;
}

void PhyloTree::setModel(int ncat_, const double *eval, const double *evec, const double *inv_evec,
                         const double *rates, const double *props) {
    setMixtureModel(1, ncat_, nullptr, eval, evec, inv_evec, rates, props);
}

void PhyloTree::setMixtureModel(int nclass, int ncat_, const int *cat_class, const double *eval, const double *evec,
                                const double *inv_evec, const double *rates, const double *props) {
    if (num_states <= 0) throw std::runtime_error("setAlignment first");
    if (engine && ncat_ != ncat) throw std::runtime_error("ncat cannot change after attachEngine");
    if (nclass < 1 || (nclass > 1 && !cat_class)) throw std::runtime_error("bad mixture description");
    const int n = num_states;
    ncat = ncat_;
    nmixture = nclass;
    m_eval.assign(eval, eval + (size_t)nclass * n);
    m_evec.assign(evec, evec + (size_t)nclass * n * n);
    m_inv_evec.assign(inv_evec, inv_evec + (size_t)nclass * n * n);
    m_rates.assign(rates, rates + ncat);
    m_props.assign(props, props + ncat);
    m_cat_class.assign(ncat, 0);
    if (nclass > 1) m_cat_class.assign(cat_class, cat_class + ncat);
    computeTipPartialLikelihood();
    inputs_dirty = model_dirty = true;
    theta_computed = false;
}

// phylotreesse.cpp:359-529; for mixtures tip_partial_lh[state][class][i] (:395-458)
void PhyloTree::computeTipPartialLikelihood() {
    const int n = num_states, M = nmixture;
    tip_partial_lh.assign((size_t)(STATE_UNKNOWN + 1) * M * n, 0.0);
    for (int m = 0; m < M; m++) {
        const double *inv_evec = m_inv_evec.data() + (size_t)m * n * n;
        auto tip = [&](int state) { return &tip_partial_lh[((size_t)state * M + m) * n]; };
        for (int state = 0; state < n; state++)
            for (int i = 0; i < n; i++) tip(state)[i] = inv_evec[i * n + state];
        for (int i = 0; i < n; i++) {  // unknown character: sum over all states
            double lh_unknown = 0.0;
            for (int x = 0; x < n; x++) lh_unknown += inv_evec[i * n + x];
            tip(STATE_UNKNOWN)[i] = lh_unknown;
        }
        if (seq_type == SEQ_DNA && n == 4) {
            for (int state = 4; state < 18; state++) {  // IUPAC ambiguity = bitmask state-3
                const int cstate = state - n + 1;
                for (int i = 0; i < n; i++) {
                    double lh = 0.0;
                    for (int x = 0; x < n; x++)
                        if (cstate & (1 << x)) lh += inv_evec[i * n + x];
                    tip(state)[i] = lh;
                }
            }
        } else if (seq_type == SEQ_PROTEIN && n == 20) {
            const int ambi_aa[3] = {4 + 8, 32 + 64, 512 + 1024};  // B, Z, U
            for (int k = 0; k < 3; k++)
                for (int i = 0; i < n; i++) {
                    double lh = 0.0;
                    for (int x = 0; x < 11; x++)
                        if (ambi_aa[k] & (1 << x)) lh += inv_evec[i * n + x];
                    tip(20 + k)[i] = lh;
                }
        }
    }
}

void PhyloTree::check(int rc, const char *what) const {
    if (rc != IQHIP_OK) throw std::runtime_error(std::string(what) + ": " + iqhip_last_error());
}

void PhyloTree::attachEngine(int device) {
    if (engine) throw std::runtime_error("engine already attached");
    if (m_eval.empty() || aln_states.empty()) throw std::runtime_error("setAlignment and setModel first");
    check(iqhip_create(&engine, device, num_states, ncat, nptn, leafNum), "iqhip_create");
    inputs_dirty = model_dirty = aln_dirty = true;
    pushInputs();
    // the reference's arena: leafNum-2 vectors (LM_PER_NODE) or 3*leafNum-6 (phylotree.cpp:867-873)
    int nvec = (lh_mem_save == LM_PER_NODE) ? (leafNum - 2) : (3 * leafNum - 6);
    check(iqhip_reserve(engine, nvec), "iqhip_reserve");
}

void PhyloTree::attachEngineSharded(const int *device_ids, int ndev, int reduce_mode) {
    if (engine) throw std::runtime_error("engine already attached");
    if (m_eval.empty() || aln_states.empty()) throw std::runtime_error("setAlignment and setModel first");
    check(iqhip_create_sharded(&engine, device_ids, ndev, reduce_mode, num_states, ncat, nptn, leafNum),
          "iqhip_create_sharded");
    inputs_dirty = model_dirty = aln_dirty = true;
    pushInputs();
    int nvec = (lh_mem_save == LM_PER_NODE) ? (leafNum - 2) : (3 * leafNum - 6);
    check(iqhip_reserve(engine, nvec), "iqhip_reserve");
}

void PhyloTree::attachComm(int nranks, int rank, const void *unique_id) {
    if (!engine) throw std::runtime_error("attachEngine first");
    check(iqhip_comm_init_rank(engine, nranks, rank, unique_id), "iqhip_comm_init_rank");
}

void PhyloTree::pushInputs() {
    if (!engine || !inputs_dirty) return;
    // a model change re-sends the model only: the alignment (megabytes of state rows) stays where it is
    if (model_dirty) {
    if (nmixture > 1)
        check(iqhip_set_mixture_model(engine, nmixture, m_cat_class.data(), m_eval.data(), m_evec.data(),
                                      m_inv_evec.data(), m_rates.data(), m_props.data(), STATE_UNKNOWN,
                                      tip_partial_lh.data()),
              "iqhip_set_mixture_model");
    else
        check(iqhip_set_model(engine, m_eval.data(), m_evec.data(), m_inv_evec.data(), m_rates.data(),
                              m_props.data(), STATE_UNKNOWN, tip_partial_lh.data()),
              "iqhip_set_model");
    }
    if (aln_dirty) {
        check(iqhip_set_alignment(engine, aln_states.data(), ptn_freq.data(), ptn_invar.data()),
              "iqhip_set_alignment");
        check(iqhip_set_ascertainment(engine, n_unobserved, asc_nsites), "iqhip_set_ascertainment");
    } else if (weights_dirty) {
        check(iqhip_set_ptn_freq(engine, ptn_freq.data()), "iqhip_set_ptn_freq");
        check(iqhip_set_ptn_invar(engine, ptn_invar.data()), "iqhip_set_ptn_invar");
    }
    inputs_dirty = model_dirty = aln_dirty = weights_dirty = false;
}

// =========================================================================================
// dispatch (phylotreesse.cpp:60-357)
// =========================================================================================
void PhyloTree::setLikelihoodKernel(LikelihoodKernel lk) {
    sse = lk;
    if (lk != LK_EIGEN_HIP) {
        computePartialLikelihoodPointer = nullptr;
        computeLikelihoodBranchPointer = nullptr;
        computeLikelihoodDervPointer = nullptr;
        computeLikelihoodFromBufferPointer = nullptr;
        throw std::runtime_error(
            "setLikelihoodKernel: this build ships no CPU kernel; only LK_EIGEN_HIP is available");
    }
    computePartialLikelihoodPointer = &PhyloTree::computePartialLikelihoodHIP;
    computeLikelihoodBranchPointer = &PhyloTree::computeLikelihoodBranchHIP;
    computeLikelihoodDervPointer = &PhyloTree::computeLikelihoodDervHIP;
    computeLikelihoodFromBufferPointer = &PhyloTree::computeLikelihoodFromBufferHIP;
}

void PhyloTree::computePartialLikelihood(PhyloNeighbor *dad_branch, PhyloNode *dad) {
    (this->*computePartialLikelihoodPointer)(dad_branch, dad);
}
double PhyloTree::computeLikelihoodBranch(PhyloNeighbor *dad_branch, PhyloNode *dad) {
    return (this->*computeLikelihoodBranchPointer)(dad_branch, dad);
}
void PhyloTree::computeLikelihoodDerv(PhyloNeighbor *dad_branch, PhyloNode *dad, double &df, double &ddf) {
    (this->*computeLikelihoodDervPointer)(dad_branch, dad, df, ddf);
}
double PhyloTree::computeLikelihoodFromBuffer() {
    assert(current_it && current_it_back);
    if (computeLikelihoodFromBufferPointer) return (this->*computeLikelihoodFromBufferPointer)();
    return (this->*computeLikelihoodBranchPointer)(current_it, current_it_back->node);
}

// =========================================================================================
// buffers (phylotree.cpp:667-716, 834-987)
// =========================================================================================
void PhyloTree::initializeAllPartialLh() {
    if (!root) throw std::runtime_error("no tree");
    // depth-first from the root leaf, as initializeAllPartialLh(index, indexlh, node, dad)
    struct Rec {
        PhyloTree *t;
        void go(PhyloNode *node, PhyloNode *dad) {
            if (dad) {
                PhyloNeighbor *nei = node->findNeighbor(dad);   // node -> dad
                PhyloNeighbor *nei2 = dad->findNeighbor(node);  // dad -> node
                if (t->lh_mem_save == LM_PER_NODE) {
                    nei->partial_lh = 0;
                    nei2->partial_lh = node->isLeaf() ? 0 : t->next_key++;
                } else {
                    nei->partial_lh = nei->node->isLeaf() ? 0 : t->next_key++;
                    nei2->partial_lh = nei2->node->isLeaf() ? 0 : t->next_key++;
                }
            }
            for (PhyloNeighbor *nb : node->neighbors)
                if (nb->node != dad) go(nb->node, node);
        }
    } rec{this};
    rec.go(root, nullptr);
    central_partial_lh = true;
}

void PhyloTree::deleteAllPartialLh() {
    for (PhyloNeighbor *nb : all_neighbors) {
        if (nb->partial_lh && engine) iqhip_release(engine, nb->partial_lh);
        nb->partial_lh = 0;
        nb->partial_lh_computed = 0;
    }
    central_partial_lh = false;
}

void PhyloTree::clearAllPartialLH() {
    if (!root) return;
    root->neighbors[0]->node->clearAllPartialLh(false, root);
    current_it = current_it_back = nullptr;
}

// =========================================================================================
// The adapter (include/iqhip_adapter.h) instantiated for this mirror.  The policy answers the adapter's questions
// about the tree types and wraps the engine calls with what only the mirror has: dry-run mode (CPU tests record
// plans without a device), the legacy caller-owned collective (allreduce_hook) and the call counters.
// =========================================================================================
}  // namespace iqhost

#include "../../include/iqhip_adapter.h"

namespace iqhost {

struct MirrorPolicy : iqhip_adapter::EngineCalls<MirrorPolicy, PhyloTree> {
    typedef iqhip_adapter::EngineCalls<MirrorPolicy, PhyloTree> Base;
    typedef iqhip_adapter::Plan<MirrorPolicy> Plan;
    typedef PhyloTree Tree;
    typedef PhyloNode Node;
    typedef PhyloNeighbor Neighbor;
    static Node *node(Neighbor *nb) { return nb->node; }
    static double length(Neighbor *nb) { return nb->length; }
    static void setLength(Neighbor *nb, double len) { nb->length = len; }
    static bool isLeaf(Node *n) { return n->isLeaf(); }
    static int degree(Node *n) { return n->degree(); }
    static int leafId(Node *n) { return n->id; }
    static int numNeighbors(Node *n) { return (int)n->neighbors.size(); }
    static Neighbor *neighborAt(Node *n, int k) { return n->neighbors[k]; }
    static Neighbor *findNeighbor(Node *at, Node *to) { return at->findNeighbor(to); }
    static int &computed(Neighbor *nb) { return nb->partial_lh_computed; }
    static double &scaleFactor(Neighbor *nb) { return nb->lh_scale_factor; }
    static uint64_t key(Neighbor *nb) { return nb->partial_lh; }
    static void stealBuffer(Neighbor *to, Neighbor *from) { to->partial_lh = from->partial_lh; from->partial_lh = 0; }
    static bool perNodeMode(Tree *t) { return t->lh_mem_save == LM_PER_NODE; }
    static bool heavyFirst(Tree *t) { return t->heavy_first; }
    static void ensureBuffers(Tree *t) { if (!t->central_partial_lh) t->initializeAllPartialLh(); }
    static void sync(Tree *t) { t->pushInputs(); }
    static iqhip_engine *engine(Tree *t) { return t->engine; }
    static void fail(Tree *, const char *what, const char *detail) {
        throw std::runtime_error(std::string(what) + ": " + detail);
    }
    static void countComputation(Tree *t) { t->num_partial_lh_computations++; }
    static bool &thetaComputed(Tree *t) { return t->theta_computed; }
    static Neighbor *currentIt(Tree *t) { return t->current_it; }
    static Neighbor *currentItBack(Tree *t) { return t->current_it_back; }
    static void clearReversePartialLh(Node *n, Node *dad) { n->clearReversePartialLh(dad); }
    static void setCurrent(Tree *t, Neighbor *it, Neighbor *back) { t->current_it = it; t->current_it_back = back; }
    static void optimizeSweep(Tree *t, const iqhip_sweep_step *steps, int nsteps, int max_steps, double diverge_frac,
                              double *sum_scale, iqhip_branch_result *results) {
        needEngine(t);
        Base::optimizeSweep(t, steps, nsteps, max_steps, diverge_frac, sum_scale, results);
    }
    static double minBranchLength(Tree *t) { return t->min_branch_length; }
    static double maxBranchLength(Tree *t) { return t->max_branch_length; }

    static void record(Tree *t, const Plan &plan) {
        t->last_plan.clear();
        for (size_t k = 0; k < plan.ops.size(); k++) {
            PlanOp p;
            p.dst = plan.dst[k];
            const bool binary = plan.dst[k] && plan.nkids(k) == 2;
            p.left = binary ? plan.kid(k, 0) : nullptr;
            p.right = binary ? plan.kid(k, 1) : nullptr;
            p.op = plan.ops[k];
            t->last_plan.push_back(p);
        }
    }
    static void needEngine(Tree *t) {
        if (!t->engine) throw std::runtime_error("HIP likelihood kernel selected but no engine attached");
    }
    static void updatePartials(Tree *t, Plan &plan, double *sum_scale) {
        record(t, plan);
        if (t->dry_run) return;
        needEngine(t);
        Base::updatePartials(t, plan, sum_scale);
        t->num_submissions++;
    }
    static double traverseLnl(Tree *t, Plan &plan, iqhip_branch_end a, iqhip_branch_end b, double len, double *sum_scale) {
        record(t, plan);
        const int nops = (int)plan.ops.size();
        if (t->dry_run) {
            if (!t->allreduce_hook) return 0.0;
            // CPU rehearsal of the caller-owned collective (tests, gloo): the hook receives a HOST vector laid out
            // like the device result {lnl, -, sum_scale[k]...}, fills in this rank's share and all-reduces it; no
            // likelihood arithmetic happens in this library.
            std::vector<double> res(2 + plan.ops.size(), 0.0);
            t->allreduce_hook(res.data(), (int)res.size(), t->allreduce_ctx);
            for (int k = 0; k < nops; k++) sum_scale[k] = res[2 + k];
            return res[0];
        }
        needEngine(t);
        double lnl;
        if (t->allreduce_hook) {  // caller-owned collective (the engine's own: iqhip_comm_init_rank / iqhip_create_sharded)
            chk(t, iqhip_traverse_lnl_async(t->engine, plan.empty() ? nullptr : plan.ops.data(), nops, a, b, len),
                "iqhip_traverse_lnl_async");
            const int nres = 2 + nops;
            t->allreduce_hook(iqhip_result_device_ptr(t->engine), nres, t->allreduce_ctx);
            std::vector<double> res(nres);
            chk(t, iqhip_result_read(t->engine, res.data(), nres), "iqhip_result_read");
            lnl = res[0];
            for (int k = 0; k < nops; k++) sum_scale[k] = res[2 + k];
        } else {
            lnl = Base::traverseLnl(t, plan, a, b, len, sum_scale);
        }
        t->num_submissions++;
        return lnl;
    }
    static void computeTheta(Tree *t, iqhip_branch_end a, iqhip_branch_end b) {
        if (t->dry_run) return;
        needEngine(t);
        Base::computeTheta(t, a, b);
    }
    static void derv(Tree *t, double len, double &df, double &ddf) {
        t->num_derv_calls++;
        if (t->dry_run) return;
        needEngine(t);
        if (t->allreduce_hook) {
            chk(t, iqhip_derv_async(t->engine, len), "iqhip_derv_async");
            t->allreduce_hook(iqhip_result_device_ptr(t->engine), 2, t->allreduce_ctx);
            double res[2];
            chk(t, iqhip_result_read(t->engine, res, 2), "iqhip_result_read");
            df = res[0];
            ddf = res[1];
            if (isnan(df) || isinf(df)) df = ddf = 0.0;  // phylokernel.h:647-651
        } else {
            Base::derv(t, len, df, ddf);
        }
    }
    static double lnlFromTheta(Tree *t, double len) {
        if (t->dry_run) return 0.0;
        needEngine(t);
        return Base::lnlFromTheta(t, len);
    }
    static double optimizeBranch(Tree *t, Plan &plan, iqhip_branch_end a, iqhip_branch_end b, double xguess, int max_steps,
                                 double *sum_scale, int *nsteps) {
        record(t, plan);
        needEngine(t);
        const double optx = Base::optimizeBranch(t, plan, a, b, xguess, max_steps, sum_scale, nsteps);
        t->num_submissions++;
        t->num_derv_calls += *nsteps;
        return optx;
    }
};

iqhip_branch_end PhyloTree::branchEnd(PhyloNeighbor *nei) const { return iqhip_adapter::branchEnd<MirrorPolicy>(nei); }

void PhyloTree::computePartialLikelihoodHIP(PhyloNeighbor *dad_branch, PhyloNode *dad) {
    MirrorPolicy::Plan plan;
    iqhip_adapter::computePartialLikelihood<MirrorPolicy>(this, dad_branch, dad, &plan);
    if (plan.empty()) last_plan.clear();
}

double PhyloTree::computeLikelihoodBranchHIP(PhyloNeighbor *dad_branch, PhyloNode *dad) {
    return iqhip_adapter::computeLikelihoodBranch<MirrorPolicy>(this, dad_branch, dad);
}

void PhyloTree::computeLikelihoodDervHIP(PhyloNeighbor *dad_branch, PhyloNode *dad, double &df, double &ddf) {
    iqhip_adapter::computeLikelihoodDerv<MirrorPolicy>(this, dad_branch, dad, df, ddf);
}

double PhyloTree::computeLikelihoodFromBufferHIP() {
    assert(current_it && current_it_back);
    return iqhip_adapter::computeLikelihoodFromBuffer<MirrorPolicy>(this);
}

// =========================================================================================
// callers
// =========================================================================================
double PhyloTree::computeLikelihood(double *pattern_lh) {
    if (!root || !root->isLeaf()) throw std::runtime_error("computeLikelihood: no rooted-at-leaf tree");
    if (!current_it) {
        PhyloNode *leaf = findFarthestLeaf();
        current_it = leaf->neighbors[0];
        current_it_back = current_it->node->findNeighbor(leaf);
    }
    double score = computeLikelihoodBranch(current_it, current_it_back->node);
    if (pattern_lh) {
        fetchPatternLh(pattern_lh);
        if (current_it->lh_scale_factor < 0.0) {  // phylotree.cpp:1059-1069
            std::vector<UBYTE> sc((size_t)nptn);
            fetchScaleNum(current_it, sc.data());
            for (int64_t i = 0; i < nptn - n_unobserved; i++)
                pattern_lh[i] += std::max(sc[i], UBYTE(0)) * LOG_SCALING_THRESHOLD;
        }
    }
    curScore = score;
    return score;
}

double PhyloTree::computeFuncDerv(double value, double &df, double &ddf) {
    current_it->length = value;
    current_it_back->length = value;
    computeLikelihoodDerv(current_it, current_it_back->node, df, ddf);
    df = -df;
    ddf = -ddf;
    return 0.0;
}

// optimization.cpp:388-465 restated (Newton-Raphson with bisection safeguard on f = -dlnL/dt)
double PhyloTree::minimizeNewton(double x1, double xguess, double x2, double xacc, double &d2l, int maxNRStep) {
    double df, dx, f, temp, xh, xl, rts, rts_old;
    rts = xguess;
    if (rts < x1) rts = x1;
    if (rts > x2) rts = x2;
    computeFuncDerv(rts, f, df);
    d2l = df;
    if (!isfinite(f) || !isfinite(df)) throw std::runtime_error("Wrong computeFuncDerv");
    if (df >= 0.0 && fabs(f) < xacc) return rts;
    if (f < 0.0) { xl = rts; xh = x2; } else { xh = rts; xl = x1; }
    dx = fabs(xh - xl);
    for (int j = 1; j <= maxNRStep; j++) {
        rts_old = rts;
        if ((df <= 0.0) || (((rts - xh) * df - f) * ((rts - xl) * df - f) >= 0.0)) {
            dx = 0.5 * (xh - xl);
            rts = xl + dx;
            d2l = df;
            if (xl == rts) return rts;
        } else {
            dx = f / df;
            temp = rts;
            rts -= dx;
            d2l = df;
            if (temp == rts) return rts;
        }
        if (fabs(dx) < xacc || (j == maxNRStep)) return rts_old;
        computeFuncDerv(rts, f, df);
        if (!isfinite(f) || !isfinite(df)) throw std::runtime_error("Wrong computeFuncDerv");
        if (df > 0.0 && fabs(f) < xacc) { d2l = df; return rts; }
        if (f < 0.0) xl = rts; else xh = rts;
    }
    throw std::runtime_error("Maximum number of iterations exceeded in minimizeNewton");
}

void PhyloTree::optimizeOneBranch(PhyloNode *node1, PhyloNode *node2, bool clearLH, int maxNRStep) {
    current_it = node1->findNeighbor(node2);
    current_it_back = node2->findNeighbor(node1);
    assert(current_it && current_it_back);
    const double current_len = current_it->length;
    theta_computed = false;
    double d2l;
    double optx;
    if (device_newton && engine && !dry_run && !allreduce_hook) {
        // one engine call: pending node updates of both ends + theta + the whole minimizeNewton solve (on a sharded
        // engine as an enqueued chain of steps with an in-stream all-reduce each); allreduce_hook is the legacy
        // caller-owned collective, which has to come back to the host between steps
        int nsteps = 0;
        optx = iqhip_adapter::minimizeNewtonOnBranch<MirrorPolicy>(this, current_len, maxNRStep, &nsteps);
        (void)d2l;
    } else {
        optx = minimizeNewton(min_branch_length, current_len, max_branch_length, min_branch_length, d2l, maxNRStep);
    }
    if (optx > max_branch_length * 0.95) {  // newton raphson diverged, reset (phylotree.cpp:2167-2176)
        // (both solvers leave the branch at the last evaluated length, as the reference's computeFuncDerv does)
        double opt_lh = computeLikelihoodFromBuffer();
        current_it->length = current_it_back->length = current_len;
        double orig_lh = computeLikelihoodFromBuffer();
        if (orig_lh > opt_lh) optx = current_len;
    }
    current_it->length = optx;
    current_it_back->length = optx;
    if (clearLH && current_len != optx) {
        node1->clearReversePartialLh(node2);
        node2->clearReversePartialLh(node1);
    }
}

void PhyloTree::getPreOrderBranches(std::vector<PhyloNode *> &n1, std::vector<PhyloNode *> &n2,
                                    PhyloNode *node, PhyloNode *dad) {
    if (dad) { n1.push_back(node); n2.push_back(dad); }
    std::vector<PhyloNeighbor *> neivec = node->neighbors;  // mtree.cpp:2102-2113: ascending height
    for (size_t i = 0; i < neivec.size(); i++)
        for (size_t j = i + 1; j < neivec.size(); j++)
            if (neivec[i]->node->height > neivec[j]->node->height) std::swap(neivec[i], neivec[j]);
    for (PhyloNeighbor *nb : neivec)
        if (nb->node != dad) getPreOrderBranches(n1, n2, nb->node, node);
}

double PhyloTree::optimizeAllBranches(int my_iterations, double tolerance, int maxNRStep) {
    std::vector<PhyloNode *> nodes1, nodes2;
    PhyloNode *farleaf = findFarthestLeaf();
    findFarthestLeaf(farleaf);  // phylotree.cpp:2241-2250
    getPreOrderBranches(nodes1, nodes2, farleaf, nullptr);
    double tree_lh = computeLikelihoodBranch(nodes1[0]->findNeighbor(nodes2[0]), nodes1[0]);
    for (int i = 0; i < my_iterations; i++) {
        std::vector<double> lenvec;
        for (PhyloNeighbor *nb : all_neighbors) lenvec.push_back(nb->length);
        if (device_sweep && device_newton && engine && !dry_run && !allreduce_hook) {
            // the whole loop as one engine submission (iqhip_adapter::optimizeBranchSweep -> iqhip_optimize_sweep)
            theta_computed = false;
            int nevals = 0;
            timespec ts0, ts1;
            static const bool dbg = getenv("IQHIP_DEBUG_SWEEP") != nullptr;
            if (dbg) clock_gettime(CLOCK_MONOTONIC, &ts0);
            iqhip_adapter::optimizeBranchSweep<MirrorPolicy>(this, nodes1.data(), nodes2.data(), (int)nodes1.size(), maxNRStep,
                                                            0.95, &nevals);
            if (dbg) {
                clock_gettime(CLOCK_MONOTONIC, &ts1);
                fprintf(stderr, "[iqhost] optimizeBranchSweep: %.1f us in all\n", (ts1.tv_sec - ts0.tv_sec) * 1e6 + (ts1.tv_nsec - ts0.tv_nsec) * 1e-3);
            }
            num_derv_calls += nevals;
            num_submissions++;
        } else {
            for (size_t j = 0; j < nodes1.size(); j++) optimizeOneBranch(nodes1[j], nodes2[j], true, maxNRStep);
        }
        double new_tree_lh = computeLikelihoodFromBuffer();
        if (new_tree_lh < tree_lh) {  // rare: restore and stop (phylotree.cpp:2303-2319)
            clearAllPartialLH();
            for (size_t k = 0; k < all_neighbors.size(); k++) all_neighbors[k]->length = lenvec[k];
            new_tree_lh = computeLikelihood();
            curScore = new_tree_lh;
            return new_tree_lh;
        }
        if (tree_lh <= new_tree_lh && new_tree_lh <= tree_lh + tolerance) { curScore = new_tree_lh; return new_tree_lh; }
        tree_lh = new_tree_lh;
    }
    curScore = tree_lh;
    return tree_lh;
}

// =========================================================================================
// NNI evaluation, phylotree.cpp:2873-3066 (upper-bound shortcut and ptnlh output left out)
// =========================================================================================
static void updateNeighborNode(PhyloNode *at, PhyloNode *oldn, PhyloNode *newn) {
    for (PhyloNeighbor *nb : at->neighbors)
        if (nb->node == oldn) { nb->node = newn; return; }
    throw std::runtime_error("updateNeighbor: not adjacent");
}

PhyloTree::NNIMove PhyloTree::getBestNNIForBran(PhyloNode *node1, PhyloNode *node2, bool nni5, NNIMove moves[2]) {
    if (node1->isLeaf() || node2->isLeaf() || node1->degree() != 3 || node2->degree() != 3)
        throw std::runtime_error("getBestNNIForBran needs an internal branch of a binary tree");
    if (!central_partial_lh) initializeAllPartialLh();
    const int IT_NUM = nni5 ? 6 : 2;
    if (!nni_keys[0])
        for (int i = 0; i < 6; i++) nni_keys[i] = next_key++;
    // the neighbour slots that get a scratch copy (:2901-2924)
    std::vector<PhyloNeighbor **> slot;
    auto slot_of = [](PhyloNode *at, PhyloNode *to) -> PhyloNeighbor ** {
        for (auto &nb : at->neighbors)
            if (nb->node == to) return &nb;
        throw std::runtime_error("not adjacent");
    };
    slot.push_back(slot_of(node1, node2));
    slot.push_back(slot_of(node2, node1));
    if (nni5) {
        for (PhyloNeighbor *nb : node1->neighbors) if (nb->node != node2) slot.push_back(slot_of(nb->node, node1));
        for (PhyloNeighbor *nb : node2->neighbors) if (nb->node != node1) slot.push_back(slot_of(nb->node, node2));
    }
    std::vector<PhyloNeighbor *> saved(IT_NUM);
    for (int id = 0; id < IT_NUM; id++) {
        saved[id] = *slot[id];
        PhyloNeighbor *tmp = new PhyloNeighbor();
        tmp->node = saved[id]->node;
        tmp->length = saved[id]->length;
        tmp->id = saved[id]->id;
        tmp->partial_lh = nni_keys[id];
        *slot[id] = tmp;
    }
    PhyloNeighbor *node12_it = node1->findNeighbor(node2), *node21_it = node2->findNeighbor(node1);

    // the two moves: first non-node2 neighbour of node1 against each non-node1 neighbour of node2 (:2951-2960)
    size_t i1 = 0;
    while (node1->neighbors[i1]->node == node2) i1++;
    std::vector<size_t> i2s;
    for (size_t j = 0; j < node2->neighbors.size(); j++)
        if (node2->neighbors[j]->node != node1) i2s.push_back(j);
    const double backupScore = curScore;
    for (int cnt = 0; cnt < 2; cnt++) {
        const size_t i2 = i2s[cnt];
        PhyloNeighbor *node1_nei = node1->neighbors[i1], *node2_nei = node2->neighbors[i2];
        moves[cnt] = NNIMove();
        moves[cnt].node1 = node1->id;
        moves[cnt].node2 = node2->id;
        moves[cnt].node1_nei = node1_nei->node->id;
        moves[cnt].node2_nei = node2_nei->node->id;
        // do the NNI swap (:2976-2980)
        node1->neighbors[i1] = node2_nei;
        updateNeighborNode(node2_nei->node, node2, node1);
        node2->neighbors[i2] = node1_nei;
        updateNeighborNode(node1_nei->node, node1, node2);
        node12_it->clearPartialLh();
        node21_it->clearPartialLh();
        int i = 1;
        if (nni5) {
            for (PhyloNeighbor *nb : std::vector<PhyloNeighbor *>(node1->neighbors))
                if (nb->node != node2) {
                    nb->node->findNeighbor(node1)->clearPartialLh();
                    optimizeOneBranch(node1, nb->node, false, NNI_MAX_NR_STEP);
                    moves[cnt].newLen[i++] = node1->findNeighbor(nb->node)->length;
                }
            node21_it->clearPartialLh();
        }
        optimizeOneBranch(node1, node2, false, NNI_MAX_NR_STEP);
        moves[cnt].newLen[0] = node1->findNeighbor(node2)->length;
        if (nni5) {
            for (PhyloNeighbor *nb : std::vector<PhyloNeighbor *>(node2->neighbors))
                if (nb->node != node1) {
                    nb->node->findNeighbor(node2)->clearPartialLh();
                    optimizeOneBranch(node2, nb->node, false, NNI_MAX_NR_STEP);
                    moves[cnt].newLen[i++] = node2->findNeighbor(nb->node)->length;
                }
            node12_it->clearPartialLh();
        }
        moves[cnt].newloglh = computeLikelihoodFromBuffer();
        // swap back (:3027-3030)
        node1->neighbors[i1] = node1_nei;
        updateNeighborNode(node1_nei->node, node2, node1);
        node2->neighbors[i2] = node2_nei;
        updateNeighborNode(node2_nei->node, node1, node2);
    }
    // restore the Neighbor objects (:3036-3044)
    for (int id = IT_NUM - 1; id >= 0; id--) {
        if (*slot[id] == current_it) current_it = saved[id];
        if (*slot[id] == current_it_back) current_it_back = saved[id];
        delete *slot[id];
        *slot[id] = saved[id];
    }
    // restore the length of the 4 branches around node1, node2 (:3048-3051)
    for (PhyloNeighbor *nb : node1->neighbors) if (nb->node != node2) nb->length = nb->node->findNeighbor(node1)->length;
    for (PhyloNeighbor *nb : node2->neighbors) if (nb->node != node1) nb->length = nb->node->findNeighbor(node2)->length;
    theta_computed = false;
    curScore = backupScore;
    return moves[0].newloglh > moves[1].newloglh ? moves[0] : moves[1];
}

void PhyloTree::computeAllPartialLh() {
    if (lh_mem_save != LM_ALL_BRANCH) throw std::runtime_error("computeAllPartialLh needs LM_ALL_BRANCH");
    if (!central_partial_lh) initializeAllPartialLh();
    // one plan for every pending directed vector (each has its own buffer in this mode), one submission
    MirrorPolicy::Plan plan;
    for (PhyloNode *node : nodes)
        for (PhyloNeighbor *nb : node->neighbors)
            if (!nb->node->isLeaf() && (nb->partial_lh_computed & 1) == 0)
                iqhip_adapter::collectPlan<MirrorPolicy>(this, nb, node, plan);
    MirrorPolicy::record(this, plan);
    if (plan.empty()) return;
    if (!dry_run && allreduce_hook) throw std::runtime_error("computeAllPartialLh: not available with a caller-owned collective");
    std::vector<double> sum_scale(plan.size(), 0.0);
    pushInputs();
    MirrorPolicy::updatePartials(this, plan, sum_scale.data());
    iqhip_adapter::applyScaleFactors<MirrorPolicy>(plan, sum_scale.data());
}

void PhyloTree::evaluateNNIsBatch(std::vector<NNIMove> &moves) {
    if (!engine || dry_run) throw std::runtime_error("evaluateNNIsBatch needs an attached engine");
    if (allreduce_hook || n_unobserved > 0) throw std::runtime_error("evaluateNNIsBatch: sharded / +ASC runs use getBestNNIForBran");
    computeAllPartialLh();
    pushInputs();
    struct Cand {
        PhyloNode *node1, *node2;
        PhyloNeighbor *n1a, *n1b, *n2x, *n2o;  // n1a <-> n2x are swapped
        iqhip_node_op ops[2];
    };
    std::vector<Cand> cands;
    for (PhyloNode *node1 : nodes)
        for (PhyloNeighbor *nb12 : node1->neighbors) {
            PhyloNode *node2 = nb12->node;
            if (node1->isLeaf() || node2->isLeaf() || node1->id > node2->id) continue;
            if (node1->degree() != 3 || node2->degree() != 3)
                throw std::runtime_error("evaluateNNIsBatch needs a binary tree");
            std::vector<PhyloNeighbor *> s1, s2;
            for (PhyloNeighbor *nb : node1->neighbors) if (nb->node != node2) s1.push_back(nb);
            for (PhyloNeighbor *nb : node2->neighbors) if (nb->node != node1) s2.push_back(nb);
            for (int cnt = 0; cnt < 2; cnt++) {  // phylotree.cpp:2951-2960: first neighbour of node1 against each of node2's
                Cand c;
                c.node1 = node1; c.node2 = node2;
                c.n1a = s1[0]; c.n1b = s1[1]; c.n2x = s2[cnt]; c.n2o = s2[1 - cnt];
                cands.push_back(c);
            }
        }
    while (nni_batch_keys.size() < 2 * cands.size()) nni_batch_keys.push_back(next_key++);
    auto child = [&](PhyloNeighbor *nb, uint64_t &key, int32_t &leaf, double &len) {
        if (nb->node->isLeaf()) { key = 0; leaf = nb->node->id; }
        else { key = nb->partial_lh; leaf = -1; }
        len = nb->length;
    };
    std::vector<iqhip_branch_task> tasks(cands.size());
    for (size_t t = 0; t < cands.size(); t++) {
        Cand &c = cands[t];
        // after the swap node2's subtree (seen from node1) joins n1a and n2o, node1's joins n2x and n1b
        memset(c.ops, 0, sizeof c.ops);
        c.ops[0].dst_key = nni_batch_keys[2 * t];
        child(c.n1a, c.ops[0].left_key, c.ops[0].left_leaf, c.ops[0].left_len);
        child(c.n2o, c.ops[0].right_key, c.ops[0].right_leaf, c.ops[0].right_len);
        c.ops[1].dst_key = nni_batch_keys[2 * t + 1];
        child(c.n2x, c.ops[1].left_key, c.ops[1].left_leaf, c.ops[1].left_len);
        child(c.n1b, c.ops[1].right_key, c.ops[1].right_leaf, c.ops[1].right_len);
        iqhip_branch_task &k = tasks[t];
        k.ops = c.ops;
        k.nops = 2;
        k.max_steps = NNI_MAX_NR_STEP;
        k.a = iqhip_branch_end{nni_batch_keys[2 * t], -1, 0};
        k.b = iqhip_branch_end{nni_batch_keys[2 * t + 1], -1, 0};
        k.xguess = c.node1->findNeighbor(c.node2)->length;
        k.x1 = min_branch_length;
        k.x2 = max_branch_length;
        k.xacc = min_branch_length;
    }
    // the tasks hold pointers into cands: do not touch cands from here on.
    // Two rounds: the reference evaluates the second swap of a branch starting from the length the first swap's
    // optimisation left on that branch (the Neighbor copies are restored only after both, phylotree.cpp:3036-3051),
    // so the first swaps of all branches run side by side, then the second swaps with those lengths as guesses.
    std::vector<double> sum_scale(2 * cands.size(), 0.0);
    std::vector<iqhip_branch_result> res(cands.size());
    for (int round = 0; round < 2; round++) {
        std::vector<iqhip_branch_task> sub;
        for (size_t t = round; t < tasks.size(); t += 2) {
            if (round == 1) tasks[t].xguess = res[t - 1].optx;
            sub.push_back(tasks[t]);
        }
        std::vector<double> ss(2 * sub.size(), 0.0);
        std::vector<iqhip_branch_result> rr(sub.size());
        check(iqhip_optimize_branch_batch(engine, sub.data(), (int)sub.size(), ss.data(), rr.data()),
              "iqhip_optimize_branch_batch");
        num_submissions++;
        for (size_t q = 0; q < sub.size(); q++) {
            const size_t t = 2 * q + round;
            res[t] = rr[q];
            sum_scale[2 * t] = ss[2 * q];
            sum_scale[2 * t + 1] = ss[2 * q + 1];
        }
    }
    moves.assign(cands.size(), NNIMove());
    for (size_t t = 0; t < cands.size(); t++) {
        const Cand &c = cands[t];
        NNIMove &m = moves[t];
        m.node1 = c.node1->id;
        m.node2 = c.node2->id;
        m.node1_nei = c.n1a->node->id;
        m.node2_nei = c.n2x->node->id;
        m.newLen[0] = res[t].optx;
        num_derv_calls += res[t].nsteps;
        if (res[t].status == 2) throw std::runtime_error("Wrong computeFuncDerv (non-finite derivative)");
        const double sf = c.n1a->lh_scale_factor + c.n2o->lh_scale_factor + sum_scale[2 * t] +
                          c.n2x->lh_scale_factor + c.n1b->lh_scale_factor + sum_scale[2 * t + 1];
        m.newloglh = res[t].lnl + sf;
        const bool first_diverged = (t & 1) && res[t - 1].optx > max_branch_length * 0.95;
        if (res[t].optx > max_branch_length * 0.95 || first_diverged) {
            // diverged Newton (phylotree.cpp:2167-2176: optimizeOneBranch compares lnL at the optimum with lnL at the old
            // length and may restore the latter): rare; the one-branch path for this candidate -- and for the second swap
            // of a branch whose first swap diverged, because its starting length is whatever that comparison left
            NNIMove two[2];
            getBestNNIForBran(c.node1, c.node2, false, two);
            m = two[t & 1];
        }
    }
}

void PhyloTree::evaluateNNIs5Batch(std::vector<NNIMove> &moves) {
    if (!engine || dry_run) throw std::runtime_error("evaluateNNIs5Batch needs an attached engine");
    if (allreduce_hook || n_unobserved > 0) throw std::runtime_error("evaluateNNIs5Batch: sharded / +ASC runs use getBestNNIForBran");
    computeAllPartialLh();
    pushInputs();
    // an outward subtree vector around the branch: fixed while the candidate is evaluated
    struct Ext { uint64_t key; int32_t leaf; double sf; };
    auto ext_of = [](PhyloNeighbor *nb) {
        Ext x;
        if (nb->node->isLeaf()) { x.key = 0; x.leaf = nb->node->id; x.sf = 0.0; }
        else { x.key = nb->partial_lh; x.leaf = -1; x.sf = nb->lh_scale_factor; }
        return x;
    };
    struct Branch {  // one internal branch: the five Neighbor lengths carry over from the first swap to the second
        PhyloNode *node1, *node2;
        PhyloNeighbor *n1[2], *n2[2];
        double len1[2], len2[2], l0;
    };
    std::vector<Branch> branches;
    for (PhyloNode *node1 : nodes)
        for (PhyloNeighbor *nb12 : node1->neighbors) {
            PhyloNode *node2 = nb12->node;
            if (node1->isLeaf() || node2->isLeaf() || node1->id > node2->id) continue;
            if (node1->degree() != 3 || node2->degree() != 3) throw std::runtime_error("evaluateNNIs5Batch needs a binary tree");
            Branch b;
            b.node1 = node1; b.node2 = node2;
            int i = 0, j = 0;
            for (PhyloNeighbor *nb : node1->neighbors) if (nb->node != node2) { b.n1[i] = nb; b.len1[i++] = nb->length; }
            for (PhyloNeighbor *nb : node2->neighbors) if (nb->node != node1) { b.n2[j] = nb; b.len2[j++] = nb->length; }
            b.l0 = nb12->length;
            branches.push_back(b);
        }
    const size_t nb = branches.size();
    while (nni_batch_keys.size() < 4 * nb) nni_batch_keys.push_back(next_key++);
    moves.assign(2 * nb, NNIMove());
    struct Work {
        Ext X[2], Y[2];           // subtrees at node1 (X1 = moved in, X2) and at node2 in optimisation order
        double lX[2], lY[2], l0;
        double *outX[2], *outY[2];  // where the optimised lengths go back to (the Branch's Neighbor slots)
        uint64_t s0, s1, s2, s3;
        double sf0, sf1, sf2, sf3;
        iqhip_node_op ops[2];
        bool diverged;
    };
    auto mkop = [](uint64_t dst, const Ext *a, uint64_t akey, double alen, const Ext *b, uint64_t bkey, double blen) {
        iqhip_node_op o;
        memset(&o, 0, sizeof o);
        o.dst_key = dst;
        o.left_key = a ? a->key : akey;  o.left_leaf = a ? a->leaf : -1;  o.left_len = alen;
        o.right_key = b ? b->key : bkey; o.right_leaf = b ? b->leaf : -1; o.right_len = blen;
        return o;
    };
    for (int cnt = 0; cnt < 2; cnt++) {
        std::vector<Work> work(nb);
        for (size_t q = 0; q < nb; q++) {
            Branch &b = branches[q];
            Work &w = work[q];
            // swap n1[0] <-> n2[cnt]: node1 now holds {n2[cnt], n1[1]}, node2 holds n1[0] at n2[cnt]'s index
            w.X[0] = ext_of(b.n2[cnt]); w.lX[0] = b.len2[cnt];     w.outX[0] = &b.len2[cnt];
            w.X[1] = ext_of(b.n1[1]);   w.lX[1] = b.len1[1];       w.outX[1] = &b.len1[1];
            const int moved_pos = cnt, other = 1 - cnt;  // neighbour-index order at node2 (phylotree.cpp:3008-3016)
            Ext R = ext_of(b.n1[0]), S = ext_of(b.n2[other]);
            if (moved_pos < other) { w.Y[0] = R; w.lY[0] = b.len1[0]; w.outY[0] = &b.len1[0]; w.Y[1] = S; w.lY[1] = b.len2[other]; w.outY[1] = &b.len2[other]; }
            else { w.Y[0] = S; w.lY[0] = b.len2[other]; w.outY[0] = &b.len2[other]; w.Y[1] = R; w.lY[1] = b.len1[0]; w.outY[1] = &b.len1[0]; }
            w.l0 = b.l0;
            w.s0 = nni_batch_keys[4 * q]; w.s1 = nni_batch_keys[4 * q + 1];
            w.s2 = nni_batch_keys[4 * q + 2]; w.s3 = nni_batch_keys[4 * q + 3];
            w.diverged = false;
            NNIMove &m = moves[2 * q + cnt];
            m.node1 = b.node1->id; m.node2 = b.node2->id;
            m.node1_nei = b.n1[0]->node->id; m.node2_nei = b.n2[cnt]->node->id;
        }
        for (int round = 0; round < 5; round++) {
            std::vector<iqhip_branch_task> tasks(nb);
            for (size_t q = 0; q < nb; q++) {
                Work &w = work[q];
                iqhip_branch_task &k = tasks[q];
                memset(&k, 0, sizeof k);
                k.ops = w.ops;
                k.max_steps = NNI_MAX_NR_STEP;
                k.x1 = min_branch_length; k.x2 = max_branch_length; k.xacc = min_branch_length;
                const Ext &Ya = w.Y[0], &Yb = w.Y[1];
                switch (round) {
                    case 0:  // branch node1 - X1: u12 = f(Y...), w = f(u12, X2)
                        w.ops[0] = mkop(w.s0, &Ya, 0, w.lY[0], &Yb, 0, w.lY[1]);
                        w.ops[1] = mkop(w.s1, nullptr, w.s0, w.l0, &w.X[1], 0, w.lX[1]);
                        k.nops = 2; k.a = iqhip_branch_end{w.X[0].key, w.X[0].leaf, 0}; k.b = iqhip_branch_end{w.s1, -1, 0};
                        k.xguess = w.lX[0];
                        break;
                    case 1:  // branch node1 - X2
                        w.ops[0] = mkop(w.s1, nullptr, w.s0, w.l0, &w.X[0], 0, w.lX[0]);
                        k.nops = 1; k.a = iqhip_branch_end{w.X[1].key, w.X[1].leaf, 0}; k.b = iqhip_branch_end{w.s1, -1, 0};
                        k.xguess = w.lX[1];
                        break;
                    case 2:  // central branch
                        w.ops[0] = mkop(w.s2, &w.X[0], 0, w.lX[0], &w.X[1], 0, w.lX[1]);
                        k.nops = 1; k.a = iqhip_branch_end{w.s0, -1, 0}; k.b = iqhip_branch_end{w.s2, -1, 0};
                        k.xguess = w.l0;
                        break;
                    case 3:  // branch node2 - Y1
                        w.ops[0] = mkop(w.s3, nullptr, w.s2, w.l0, &Yb, 0, w.lY[1]);
                        k.nops = 1; k.a = iqhip_branch_end{Ya.key, Ya.leaf, 0}; k.b = iqhip_branch_end{w.s3, -1, 0};
                        k.xguess = w.lY[0];
                        break;
                    default:  // branch node2 - Y2
                        w.ops[0] = mkop(w.s3, nullptr, w.s2, w.l0, &Ya, 0, w.lY[0]);
                        k.nops = 1; k.a = iqhip_branch_end{Yb.key, Yb.leaf, 0}; k.b = iqhip_branch_end{w.s3, -1, 0};
                        k.xguess = w.lY[1];
                        break;
                }
            }
            std::vector<double> ss(2 * nb + 2, 0.0);
            std::vector<iqhip_branch_result> rr(nb);
            check(iqhip_optimize_branch_batch(engine, tasks.data(), (int)nb, ss.data(), rr.data()),
                  "iqhip_optimize_branch_batch");
            num_submissions++;
            size_t sp = 0;
            for (size_t q = 0; q < nb; q++) {
                Work &w = work[q];
                const iqhip_branch_result &r = rr[q];
                if (r.status == 2) throw std::runtime_error("Wrong computeFuncDerv (non-finite derivative)");
                if (r.optx > max_branch_length * 0.95) w.diverged = true;
                num_derv_calls += r.nsteps;
                NNIMove &m = moves[2 * q + cnt];
                switch (round) {
                    case 0:
                        w.sf0 = w.Y[0].sf + w.Y[1].sf + ss[sp];
                        w.sf1 = w.sf0 + w.X[1].sf + ss[sp + 1];
                        sp += 2;
                        w.lX[0] = r.optx; m.newLen[1] = r.optx;
                        break;
                    case 1:
                        w.sf1 = w.sf0 + w.X[0].sf + ss[sp++];
                        w.lX[1] = r.optx; m.newLen[2] = r.optx;
                        break;
                    case 2:
                        w.sf2 = w.X[0].sf + w.X[1].sf + ss[sp++];
                        w.l0 = r.optx; m.newLen[0] = r.optx;
                        break;
                    case 3:
                        w.sf3 = w.sf2 + w.Y[1].sf + ss[sp++];
                        w.lY[0] = r.optx; m.newLen[3] = r.optx;
                        break;
                    default:
                        w.sf3 = w.sf2 + w.Y[0].sf + ss[sp++];
                        w.lY[1] = r.optx; m.newLen[4] = r.optx;
                        m.newloglh = r.lnl + w.Y[1].sf + w.sf3;
                        break;
                }
            }
        }
        // the lengths the swap left on the five Neighbor objects are the second swap's starting point
        for (size_t q = 0; q < nb; q++) {
            Work &w = work[q];
            *w.outX[0] = w.lX[0]; *w.outX[1] = w.lX[1];
            *w.outY[0] = w.lY[0]; *w.outY[1] = w.lY[1];
            branches[q].l0 = w.l0;
            if (w.diverged) {  // diverged Newton (phylotree.cpp:2167-2176): rare; take the branch-by-branch path
                NNIMove two[2];
                getBestNNIForBran(branches[q].node1, branches[q].node2, true, two);
                moves[2 * q] = two[0];
                moves[2 * q + 1] = two[1];
            }
        }
    }
}

// =========================================================================================
// host views
// =========================================================================================
void PhyloTree::fetchScaleNum(PhyloNeighbor *nei, UBYTE *out) {
    if (!engine) throw std::runtime_error("no engine");
    if (!nei->partial_lh) throw std::runtime_error("neighbor has no buffer");
    check(iqhip_fetch_scale_num(engine, nei->partial_lh, out), "iqhip_fetch_scale_num");
}
void PhyloTree::fetchPartialLh(PhyloNeighbor *nei, double *out) {
    if (!engine) throw std::runtime_error("no engine");
    if (!nei->partial_lh) throw std::runtime_error("neighbor has no buffer");
    check(iqhip_fetch_partial(engine, nei->partial_lh, out), "iqhip_fetch_partial");
}
void PhyloTree::fetchPatternLh(double *out) {
    if (!engine) throw std::runtime_error("no engine");
    check(iqhip_fetch_pattern_lh(engine, out), "iqhip_fetch_pattern_lh");
}

void PhyloTree::computePatternLikelihood(double *ptn_lh) {
    if (!engine) throw std::runtime_error("no engine");
    if (!current_it) throw std::runtime_error("computePatternLikelihood before computeLikelihood");
    check(iqhip_fetch_pattern_lh_scaled(engine, branchEnd(current_it), branchEnd(current_it_back), ptn_lh),
          "iqhip_fetch_pattern_lh_scaled");
}

void PhyloTree::computePatternLhCat(double *ptn_lh_cat) {
    if (!engine) throw std::runtime_error("no engine");
    if (!current_it) throw std::runtime_error("computePatternLhCat before computeLikelihood");
    double df, ddf;
    theta_computed = false;
    computeLikelihoodDerv(current_it, current_it_back->node, df, ddf);  // (re)builds theta of current_it
    check(iqhip_pattern_lh_cat(engine, current_it->length, ptn_lh_cat), "iqhip_pattern_lh_cat");
}

void PhyloTree::setBootSamples(const float *samples, int nsamples) {
    if (!engine) throw std::runtime_error("no engine");
    pushInputs();
    check(iqhip_set_boot_samples(engine, samples, nsamples), "iqhip_set_boot_samples");
    num_boot_samples = nsamples;
}

void PhyloTree::computeRELL(std::vector<double> &rell) {
    if (!engine) throw std::runtime_error("no engine");
    if (!current_it) throw std::runtime_error("computeRELL before computeLikelihood");
    rell.assign((size_t)num_boot_samples, 0.0);
    if (allreduce_hook) {  // pattern shards: partial dot products are summed over the ranks
        check(iqhip_rell_async(engine, branchEnd(current_it), branchEnd(current_it_back)), "iqhip_rell_async");
        allreduce_hook(iqhip_result_device_ptr(engine), num_boot_samples, allreduce_ctx);
        check(iqhip_result_read(engine, rell.data(), num_boot_samples), "iqhip_result_read");
    } else
        check(iqhip_rell(engine, branchEnd(current_it), branchEnd(current_it_back), rell.data()), "iqhip_rell");
}

}  // namespace iqhost
