// sharded.hip -- ONE process, several GPUs: the form the reference can use as it is (IQ-TREE is a single process,
// pda.cpp:2137; its pattern sums are in-process reductions, phylokernel.h:251,335,410,592-643,951-962).
//
// iqhip_create_sharded returns an ordinary iqhip_engine* whose `shards` are one engine per device, each holding
// the contiguous pattern range [shard_first[g], shard_first[g+1]) (boundaries on multiples of 64 patterns) of
// every vector, scale_num array, theta, _pattern_lh, ptn_freq, ptn_invar, and a replica of the small model tables
// (SURVEY.md 8e).  Every entry point of include/iqhip.h works on it unchanged: the front fans a call out to the
// shards (one host thread, asynchronous enqueues on the shards' streams), reduces the handful of result doubles, and
// concatenates the per-pattern host views.  Reduction of the result vector:
//   IQHIP_REDUCE_RCCL  one ncclAllReduce (SUM, f64, in place) per evaluation on the shards' streams, grouped
//                      (ncclGroupStart/End) because one thread drives all ranks; shard 0's copy is read back;
//   IQHIP_REDUCE_HOST  every shard's k_reduce writes its result vector to pinned host memory; the host adds the
//                      ndev vectors in shard order (the alternative SURVEY.md 8e asks to be measured; also the
//                      only form for shards that share a device, which is how a one-GPU box tests this file).
// The Newton solve of a branch keeps its whole loop enqueued in the RCCL form: per step derivative kernel ->
// k_reduce -> grouped all-reduce of {df, ddf} -> 1-thread state-machine kernel on every shard (identical inputs,
// identical iterates), and the host looks at shard 0's state once per chunk of steps.
#include <math.h>
#include <string.h>

#include <algorithm>

#include "iqhip_internal.h"

using namespace iqhip;

namespace {
int bad(int code, const std::string &msg) { return set_error(code, msg); }

// sum of the shards' result vectors -> out[0..n)
int reduce_results(iqhip_engine *p, int n, std::vector<double> &out) {
    out.assign((size_t)n, 0.0);
    if (p->reduce_mode == IQHIP_REDUCE_RCCL) {
        int rc = comm_group_allreduce(p->shards, n);
        if (rc) return rc;
        rc = eng_read_result(p->shards[0], n);
        if (rc) return rc;
        memcpy(out.data(), p->shards[0]->h_result, sizeof(double) * (size_t)n);
        return IQHIP_OK;
    }
    for (iqhip_engine *c : p->shards) {
        int rc = eng_read_result(c, n);
        if (rc) return rc;
        for (int i = 0; i < n; i++) out[i] += c->h_result[i];
    }
    return IQHIP_OK;
}
}  // namespace

extern "C" int iqhip_create_sharded(iqhip_engine **out, const int *device_ids, int ndev, int reduce_mode, int nstates,
                                    int ncat, int64_t nptn, int ntaxa) {
    if (!out) return bad(IQHIP_ERR_INVALID, "iqhip_create_sharded: out == NULL");
    *out = nullptr;
    if (!device_ids || ndev < 1 || ndev > 64) return bad(IQHIP_ERR_INVALID, "iqhip_create_sharded: bad device list");
    if (reduce_mode != IQHIP_REDUCE_RCCL && reduce_mode != IQHIP_REDUCE_HOST)
        return bad(IQHIP_ERR_INVALID, "iqhip_create_sharded: bad reduce mode");
    if (nptn < (int64_t)64 * ndev)
        return bad(IQHIP_ERR_INVALID, "iqhip_create_sharded: fewer than 64 patterns per shard");
    iqhip_engine *p = new iqhip_engine();
    p->device = device_ids[0];
    p->n = nstates;
    p->ncat = ncat;
    p->ntaxa = ntaxa;
    p->nptn = nptn;
    p->block = nstates * ncat;
    p->reduce_mode = reduce_mode;
    p->stream = nullptr;
    // contiguous ranges, boundaries on tile multiples (64 covers both tile sizes), SURVEY.md 8e
    p->shard_first.assign((size_t)ndev + 1, 0);
    for (int g = 1; g < ndev; g++) p->shard_first[g] = (nptn * g / ndev) / 64 * 64;
    p->shard_first[ndev] = nptn;
    for (int g = 0; g < ndev; g++) {
        iqhip_engine *c = nullptr;
        int rc = iqhip_create(&c, device_ids[g], nstates, ncat, p->shard_first[g + 1] - p->shard_first[g], ntaxa);
        if (rc) {
            for (iqhip_engine *q : p->shards) iqhip_destroy(q);
            delete p;
            return rc;
        }
        p->shards.push_back(c);
    }
    if (reduce_mode == IQHIP_REDUCE_RCCL) {
        int rc = comm_init_all(p->shards);
        if (rc) {
            const std::string msg = iqhip_last_error();
            for (iqhip_engine *q : p->shards) iqhip_destroy(q);
            delete p;
            return bad(rc, msg);
        }
    }
    *out = p;
    return IQHIP_OK;
}

extern "C" int iqhip_num_shards(iqhip_engine *e) { return e ? std::max<int>(1, (int)e->shards.size()) : 0; }

extern "C" int iqhip_shard_range(iqhip_engine *e, int shard, int64_t *first, int64_t *count, int *device) {
    if (!e) return bad(IQHIP_ERR_INVALID, "null engine");
    if (e->shards.empty()) {
        if (shard != 0) return bad(IQHIP_ERR_INVALID, "shard index out of range");
        if (first) *first = 0;
        if (count) *count = e->nptn;
        if (device) *device = e->device;
        return IQHIP_OK;
    }
    if (shard < 0 || shard >= (int)e->shards.size()) return bad(IQHIP_ERR_INVALID, "shard index out of range");
    if (first) *first = e->shard_first[shard];
    if (count) *count = e->shard_first[shard + 1] - e->shard_first[shard];
    if (device) *device = e->shards[shard]->device;
    return IQHIP_OK;
}

namespace iqhip {
namespace sharded {

void destroy(iqhip_engine *p) {
    for (iqhip_engine *c : p->shards) iqhip_destroy(c);  // (destroys the shard's communicator as well)
    p->shards.clear();
}

int reserve(iqhip_engine *p, int nvectors) {
    for (iqhip_engine *c : p->shards) {
        int rc = iqhip_reserve(c, nvectors);
        if (rc) return rc;
    }
    return IQHIP_OK;
}

int release(iqhip_engine *p, uint64_t key) {
    for (iqhip_engine *c : p->shards) {
        int rc = iqhip_release(c, key);
        if (rc) return rc;
    }
    return IQHIP_OK;
}

int rekey(iqhip_engine *p, uint64_t old_key, uint64_t new_key) {
    for (iqhip_engine *c : p->shards) {
        int rc = iqhip_rekey(c, old_key, new_key);
        if (rc) return rc;
    }
    return IQHIP_OK;
}

int set_model(iqhip_engine *p, int nclass, const int32_t *cat_class, const double *eval, const double *evec,
              const double *inv_evec, const double *rates, const double *props, int state_unknown, const double *tip) {
    for (iqhip_engine *c : p->shards) {
        int rc = nclass > 1 ? iqhip_set_mixture_model(c, nclass, cat_class, eval, evec, inv_evec, rates, props,
                                                      state_unknown, tip)
                            : iqhip_set_model(c, eval, evec, inv_evec, rates, props, state_unknown, tip);
        if (rc) return rc;
    }
    p->state_unknown = state_unknown;
    p->model_set = true;
    p->nclass = nclass;
    return IQHIP_OK;
}

int set_alignment(iqhip_engine *p, const uint8_t *states, const double *ptn_freq, const double *ptn_invar) {
    if (!p->model_set) return bad(IQHIP_ERR_INVALID, "iqhip_set_alignment: call iqhip_set_model first (needs state_unknown)");
    // (binary data: p->n is the caller's 2; every shard embeds on its own)
    const size_t N = (size_t)p->nptn;
    std::vector<uint8_t> rows;
    for (size_t g = 0; g < p->shards.size(); g++) {
        const size_t f = (size_t)p->shard_first[g], cnt = (size_t)(p->shard_first[g + 1] - p->shard_first[g]);
        rows.resize((size_t)p->ntaxa * cnt);
        for (int t = 0; t < p->ntaxa; t++) memcpy(rows.data() + (size_t)t * cnt, states + (size_t)t * N + f, cnt);
        int rc = iqhip_set_alignment(p->shards[g], rows.data(), ptn_freq + f, ptn_invar + f);
        if (rc) return rc;
    }
    p->aln_set = true;
    return IQHIP_OK;
}

int set_ptn_array(iqhip_engine *p, const double *v, bool invar) {
    for (size_t g = 0; g < p->shards.size(); g++) {
        const double *src = v + p->shard_first[g];
        int rc = invar ? iqhip_set_ptn_invar(p->shards[g], src) : iqhip_set_ptn_freq(p->shards[g], src);
        if (rc) return rc;
    }
    return IQHIP_OK;
}

// +ASC: the unobserved constant patterns are the LAST n_unobserved patterns of the alignment, i.e. of the last shard;
// every shard knows that the correction is active and how many sites there are
int set_ascertainment(iqhip_engine *p, int64_t n_unobserved, double nsites) {
    const size_t last = p->shards.size() - 1;
    if (n_unobserved >= p->shard_first[last + 1] - p->shard_first[last])
        return bad(IQHIP_ERR_INVALID, "iqhip_set_ascertainment: the unobserved patterns must fit the last shard");
    for (size_t g = 0; g < p->shards.size(); g++) {
        iqhip_engine *c = p->shards[g];
        c->n_unobs = g == last ? n_unobserved : 0;
        c->asc_nsites = nsites;
        c->asc_active = n_unobserved > 0;
        c->pattern_lh_shift = 0.0;
    }
    p->n_unobs = n_unobserved;
    p->asc_nsites = nsites;
    p->asc_active = n_unobserved > 0;
    p->pattern_lh_shift = 0.0;
    return IQHIP_OK;
}

// phylokernel.h:1009-1016 / 1183-1186 on the summed prob_const; the shards learn the shift for their _pattern_lh views
static int asc_finish(iqhip_engine *p, double prob_const, double *lnl) {
    if (!p->asc_active) return IQHIP_OK;
    if (!(prob_const < 1.0 && prob_const >= 0.0))
        return bad(IQHIP_ERR_INVALID, "+ASC: prob_const outside [0,1) (the reference asserts here)");
    const double lp = log(1.0 - prob_const);
    *lnl -= p->asc_nsites * lp;
    p->pattern_lh_shift = lp;
    for (iqhip_engine *c : p->shards) c->pattern_lh_shift = lp;
    return IQHIP_OK;
}

// phylokernel.h:647-651, then :719-724 on the summed {prob_const, df_const, ddf_const} = v[2..4]
static void asc_derv(const iqhip_engine *p, std::vector<double> &v) {
    if (isnan(v[0]) || isinf(v[0])) v[0] = v[1] = 0.0;
    if (!p->asc_active) return;
    const double prob_const = 1.0 - v[2];
    const double df_frac = v[3] / prob_const, ddf_frac = v[4] / prob_const;
    v[0] += p->asc_nsites * df_frac;
    v[1] += p->asc_nsites * (ddf_frac + df_frac * df_frac);
}

int traverse(iqhip_engine *p, const iqhip_node_op *ops, int nops, bool has_root, iqhip_branch_end a, iqhip_branch_end b,
             double len, double *sum_scale, double *lnl) {
    if (nops < 0 || (nops > 0 && !ops)) return bad(IQHIP_ERR_INVALID, "bad ops array");
    for (iqhip_engine *c : p->shards) {
        int rc = has_root ? iqhip_traverse_lnl_async(c, ops, nops, a, b, len) : iqhip_update_partials_async(c, ops, nops);
        if (rc) return rc;
    }
    std::vector<double> v;
    int rc = reduce_results(p, 2 + nops, v);
    if (rc) return rc;
    if (sum_scale)
        for (int k = 0; k < nops; k++) sum_scale[k] = v[2 + k];
    if (has_root) {
        double r = v[0];
        if (isnan(r) || isinf(r)) {  // phylokernel.h:848-866, shard by shard; the repaired shares are added here
            r = 0.0;
            for (iqhip_engine *c : p->shards) {
                double s = 0.0;
                rc = eng_repair_lnl(c, &s);
                if (rc) return rc;
                r += s;
            }
        }
        rc = asc_finish(p, v[1], &r);
        if (rc) return rc;
        if (lnl) *lnl = r;
    }
    return IQHIP_OK;
}

int compute_theta(iqhip_engine *p, iqhip_branch_end a, iqhip_branch_end b) {
    for (iqhip_engine *c : p->shards) {
        int rc = iqhip_compute_theta(c, a, b);
        if (rc) return rc;
    }
    p->theta_valid = true;
    return IQHIP_OK;
}

int derv(iqhip_engine *p, double len, double *df, double *ddf) {
    for (iqhip_engine *c : p->shards) {
        int rc = iqhip_derv_async(c, len);
        if (rc) return rc;
    }
    std::vector<double> v;
    int rc = reduce_results(p, p->asc_active ? 5 : 2, v);
    if (rc) return rc;
    asc_derv(p, v);
    if (df) *df = v[0];
    if (ddf) *ddf = v[1];
    return IQHIP_OK;
}

int lnl_from_theta(iqhip_engine *p, double len, double *lnl) {
    for (iqhip_engine *c : p->shards) {
        int rc = iqhip_lnl_from_theta_async(c, len);
        if (rc) return rc;
    }
    std::vector<double> v;
    int rc = reduce_results(p, p->asc_active ? 2 : 1, v);
    if (rc) return rc;
    double r = v[0];
    if (isnan(r) || isinf(r)) {
        r = 0.0;
        for (iqhip_engine *c : p->shards) {
            double s = 0.0;
            rc = eng_repair_lnl(c, &s);
            if (rc) return rc;
            r += s;
        }
    }
    if (p->asc_active) {
        rc = asc_finish(p, v[1], &r);
        if (rc) return rc;
    }
    if (lnl) *lnl = r;
    return IQHIP_OK;
}

int optimize_branch(iqhip_engine *p, const iqhip_node_op *ops, int nops, bool build_theta, iqhip_branch_end a,
                    iqhip_branch_end b, double xguess, double x1, double x2, double xacc, int max_steps,
                    double *sum_scale, double *optx, double *d2l, int *nsteps) {
    int rc = IQHIP_OK;
    iqhip_branch_end none = {0, -1, 0};
    if (nops > 0) {
        rc = traverse(p, ops, nops, false, none, none, 0.0, sum_scale, nullptr);
        if (rc) return rc;
    }
    if (build_theta) {
        rc = compute_theta(p, a, b);
        if (rc) return rc;
    }
    NewtonState st;
    if (p->reduce_mode == IQHIP_REDUCE_RCCL) {
        for (iqhip_engine *c : p->shards) {
            rc = eng_newton_begin(c, xguess, x1, x2, xacc, max_steps);
            if (rc) return rc;
        }
        int enq = 0;
        for (;;) {
            const int chunk = enq == 0 ? std::min(4, max_steps + 1) : 2;
            for (int k = 0; k < chunk; k++) {
                for (iqhip_engine *c : p->shards) {
                    rc = eng_newton_eval_enqueue(c);
                    if (rc) return rc;
                }
                rc = comm_group_allreduce(p->shards, p->asc_active ? 5 : 2);
                if (rc) return rc;
                for (iqhip_engine *c : p->shards) {
                    rc = eng_newton_update_enqueue(c);
                    if (rc) return rc;
                }
            }
            enq += chunk;
            rc = newton_state_read(p->shards[0]);
            if (rc) return rc;
            if (p->shards[0]->h_nstate->done) break;
            if (enq > max_steps + 2) return bad(IQHIP_ERR_INVALID, "Newton chain did not terminate");
        }
        st = *p->shards[0]->h_nstate;
    } else {
        // pinned-host reduction: the host adds the shards' {df, ddf} and advances the same state machine itself
        newton_init(st, xguess, x1, x2, xacc, max_steps);
        for (int guard = 0; !st.done; guard++) {
            if (guard > max_steps + 2) return bad(IQHIP_ERR_INVALID, "Newton loop did not terminate");
            for (iqhip_engine *c : p->shards) {
                rc = iqhip_derv_async(c, st.rts);
                if (rc) return rc;
            }
            std::vector<double> v;
            rc = reduce_results(p, p->asc_active ? 5 : 2, v);
            if (rc) return rc;
            if (p->asc_active) asc_derv(p, v);
            newton_update(st, v[0], v[1]);
        }
    }
    if (st.status == 2) return bad(IQHIP_ERR_INVALID, "Wrong computeFuncDerv (non-finite derivative)");
    if (st.status == 3) return bad(IQHIP_ERR_INVALID, "Maximum number of iterations exceeded in minimizeNewton");
    if (optx) *optx = st.result;
    if (d2l) *d2l = st.d2l;
    if (nsteps) *nsteps = st.neval;
    return IQHIP_OK;
}

// iqhip_optimize_branch_batch on the front: the tasks' node updates in one submission per shard, then the batched chain
// (engine.hip eng_batch_*): per Newton step one derivative launch per shard for all tasks and ONE reduction of 2m values
// across the shards (grouped all-reduce, or the host's sum with the state machines advanced on the host)
int optimize_branch_batch(iqhip_engine *p, const iqhip_branch_task *tasks, int ntasks, double *sum_scale,
                          iqhip_branch_result *results) {
    std::vector<iqhip_node_op> all;
    std::vector<int> segs((size_t)ntasks);
    for (int t = 0; t < ntasks; t++) {
        const iqhip_branch_task &k = tasks[t];
        if (k.nops < 0 || (k.nops > 0 && !k.ops)) return bad(IQHIP_ERR_INVALID, "bad ops array in a task");
        if (!(k.x1 >= 0.0) || !(k.x2 > k.x1) || !(k.xacc > 0.0) || k.max_steps < 1 || !(k.xguess >= 0.0))
            return bad(IQHIP_ERR_INVALID, "iqhip_optimize_branch_batch: bad bounds / tolerance / step count");
        segs[t] = k.nops;
        all.insert(all.end(), k.ops, k.ops + k.nops);
    }
    const int total_ops = (int)all.size();
    int rc = IQHIP_OK;
    std::vector<double> v;
    if (total_ops > 0) {
        for (iqhip_engine *c : p->shards) {
            rc = eng_submit_updates(c, all.data(), total_ops, &segs);
            if (rc) return rc;
        }
        rc = reduce_results(p, 2 + total_ops, v);
        if (rc) return rc;
        if (sum_scale)
            for (int k = 0; k < total_ops; k++) sum_scale[k] = v[2 + k];
    }
    int chunk = std::min(ntasks, 64);
    if (const char *bc = getenv("IQHIP_BATCH_CHUNK")) chunk = std::max(1, std::min(chunk, atoi(bc)));
    const bool rccl = p->reduce_mode == IQHIP_REDUCE_RCCL;
    std::vector<NewtonState> st((size_t)chunk);
    for (int first = 0; first < ntasks; first += chunk) {
        const int m = std::min(chunk, ntasks - first);
        int max_steps = 1;
        for (int t = 0; t < m; t++) {
            const iqhip_branch_task &k = tasks[first + t];
            newton_init(st[t], k.xguess, k.x1, k.x2, k.xacc, k.max_steps);
            max_steps = std::max(max_steps, k.max_steps);
        }
        for (iqhip_engine *c : p->shards) {
            rc = eng_batch_prepare(c, tasks + first, m, st.data());
            if (rc) return rc;
        }
        auto all_done = [&] {
            for (int t = 0; t < m; t++)
                if (!st[t].done) return false;
            return true;
        };
        if (rccl) {
            int enq = 0;
            for (;;) {
                const int steps = enq == 0 ? std::min(4, max_steps + 1) : 2;
                for (int k = 0; k < steps; k++) {
                    for (iqhip_engine *c : p->shards) {
                        rc = eng_batch_eval_enqueue(c, m);
                        if (rc) return rc;
                    }
                    rc = comm_group_allreduce(p->shards, 2 * m);
                    if (rc) return rc;
                    for (iqhip_engine *c : p->shards) {
                        rc = eng_batch_update_enqueue(c, m);
                        if (rc) return rc;
                    }
                }
                enq += steps;
                rc = eng_batch_states_read(p->shards[0], m, st.data());
                if (rc) return rc;
                if (all_done()) break;
                if (enq > max_steps + 2) return bad(IQHIP_ERR_INVALID, "Newton chain did not terminate");
            }
        } else {
            for (int guard = 0; !all_done(); guard++) {
                if (guard > max_steps + 2) return bad(IQHIP_ERR_INVALID, "Newton loop did not terminate");
                for (iqhip_engine *c : p->shards) {
                    if (guard > 0) rc = eng_batch_states_write(c, m, st.data());
                    if (!rc) rc = eng_batch_eval_enqueue(c, m);
                    if (rc) return rc;
                }
                rc = reduce_results(p, 2 * m, v);
                if (rc) return rc;
                for (int t = 0; t < m; t++)
                    if (!st[t].done) newton_update(st[t], v[2 * t], v[2 * t + 1]);
            }
            for (iqhip_engine *c : p->shards) {   // (the lnL launch reads the accepted lengths from the states)
                rc = eng_batch_states_write(c, m, st.data());
                if (rc) return rc;
            }
        }
        for (int t = 0; t < m; t++) {
            if (st[t].status == 2) return bad(IQHIP_ERR_INVALID, "Wrong computeFuncDerv (non-finite derivative)");
            if (st[t].status == 3) return bad(IQHIP_ERR_INVALID, "Maximum number of iterations exceeded in minimizeNewton");
            iqhip_branch_result &r = results[first + t];
            r.optx = st[t].result;
            r.d2l = st[t].d2l;
            r.nsteps = st[t].neval;
            r.status = 0;
        }
        for (iqhip_engine *c : p->shards) {
            rc = eng_batch_lnl_enqueue(c, m);
            if (rc) return rc;
        }
        rc = reduce_results(p, 2 * m, v);
        if (rc) return rc;
        const std::vector<double> lnl(v);
        for (int t = 0; t < m; t++) {
            iqhip_branch_result &r = results[first + t];
            r.lnl = lnl[2 * t];
            if (isnan(r.lnl) || isinf(r.lnl)) {   // rare: redo this task alone, with the per-shard repair
                rc = compute_theta(p, tasks[first + t].a, tasks[first + t].b);
                if (!rc) rc = lnl_from_theta(p, r.optx, &r.lnl);
                if (rc) return rc;
            }
        }
    }
    return IQHIP_OK;
}

int fetch_scale_num(iqhip_engine *p, uint64_t key, int16_t *out) {
    for (size_t g = 0; g < p->shards.size(); g++) {
        int rc = iqhip_fetch_scale_num(p->shards[g], key, out + p->shard_first[g]);
        if (rc) return rc;
    }
    return IQHIP_OK;
}

int fetch_pattern_lh(iqhip_engine *p, double *out, int kind, iqhip_branch_end a, iqhip_branch_end b) {
    for (size_t g = 0; g < p->shards.size(); g++) {
        double *dst = out + p->shard_first[g];
        int rc = kind ? iqhip_fetch_pattern_lh_scaled(p->shards[g], a, b, dst) : iqhip_fetch_pattern_lh(p->shards[g], dst);
        if (rc) return rc;
    }
    return IQHIP_OK;
}

int fetch_vec(iqhip_engine *p, uint64_t key, bool theta, double *out) {
    for (size_t g = 0; g < p->shards.size(); g++) {
        double *dst = out + (size_t)p->shard_first[g] * p->block;  // reference layout: pattern-major
        int rc = theta ? iqhip_fetch_theta(p->shards[g], dst) : iqhip_fetch_partial(p->shards[g], key, dst);
        if (rc) return rc;
    }
    return IQHIP_OK;
}

int pattern_lh_cat(iqhip_engine *p, double len, double *out) {
    for (size_t g = 0; g < p->shards.size(); g++) {
        int rc = iqhip_pattern_lh_cat(p->shards[g], len, out + (size_t)p->shard_first[g] * p->ncat);
        if (rc) return rc;
    }
    return IQHIP_OK;
}

int upload_partial(iqhip_engine *p, uint64_t key, const double *partial_lh, const int16_t *scale_num) {
    for (size_t g = 0; g < p->shards.size(); g++) {
        int rc = iqhip_upload_partial(p->shards[g], key, partial_lh + (size_t)p->shard_first[g] * p->block,
                                      scale_num + p->shard_first[g]);
        if (rc) return rc;
    }
    return IQHIP_OK;
}

int set_boot_samples(iqhip_engine *p, const float *samples, int nsamples) {
    const size_t N = (size_t)p->nptn;
    std::vector<float> part;
    for (size_t g = 0; g < p->shards.size(); g++) {
        const size_t f = (size_t)p->shard_first[g], cnt = (size_t)(p->shard_first[g + 1] - p->shard_first[g]);
        part.resize((size_t)nsamples * cnt);
        for (int s = 0; s < nsamples; s++) memcpy(part.data() + (size_t)s * cnt, samples + (size_t)s * N + f, cnt * sizeof(float));
        int rc = iqhip_set_boot_samples(p->shards[g], nsamples ? part.data() : nullptr, nsamples);
        if (rc) return rc;
    }
    p->nboot = nsamples;
    return IQHIP_OK;
}

int rell(iqhip_engine *p, iqhip_branch_end a, iqhip_branch_end b, double *out) {
    for (iqhip_engine *c : p->shards) {
        int rc = iqhip_rell_async(c, a, b);
        if (rc) return rc;
    }
    std::vector<double> v;
    int rc = reduce_results(p, p->nboot, v);
    if (rc) return rc;
    memcpy(out, v.data(), sizeof(double) * (size_t)p->nboot);
    return IQHIP_OK;
}

int synchronize(iqhip_engine *p) {
    for (iqhip_engine *c : p->shards) {
        int rc = iqhip_synchronize(c);
        if (rc) return rc;
    }
    return IQHIP_OK;
}

}  // namespace sharded
}  // namespace iqhip
