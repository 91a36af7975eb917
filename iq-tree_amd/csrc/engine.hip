// engine.hip -- host side of libiqhip.so: device memory, key->slab map, plan building and the
// extern "C" entry points declared in include/iqhip.h.  There is NO CPU fallback in this
// library: every compute entry point launches HIP kernels or fails with a status.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <atomic>
#include <unordered_set>

#include "iqhip_internal.h"

using namespace iqhip;

static thread_local std::string g_err;
namespace iqhip {
int set_error(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
}  // namespace iqhip
static int fail(int code, const std::string &msg) { return set_error(code, msg); }
#define HIPCHK(call)                                                                  \
    do {                                                                              \
        hipError_t _s = (call);                                                       \
        if (_s != hipSuccess)                                                         \
            return fail(IQHIP_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(_s)); \
    } while (0)

extern "C" const char *iqhip_last_error(void) { return g_err.c_str(); }
extern "C" int iqhip_abi_version(void) { return IQHIP_ABI_VERSION; }
extern "C" int iqhip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

template <typename T>
static hipError_t dmalloc(T **p, size_t count) {
    return hipMalloc((void **)p, count * sizeof(T));
}
// a planning-only engine has no device: every entry point that would touch one fails here ("invalid device ordinal")
static hipError_t use_device(const iqhip_engine *e) { return e->planner ? hipErrorInvalidDevice : hipSetDevice(e->device); }
// planning-only engines: a distinct, 256-byte aligned address range that is never dereferenced
template <typename T>
static T *fake_alloc(iqhip_engine *e, size_t count) {
    const uint64_t a = e->fake_next;
    e->fake_next += (count * sizeof(T) + 255) / 256 * 256 + 256;
    return reinterpret_cast<T *>(a);
}

// everything about an engine that follows from its shape, the CU count (e->num_cus, set by the caller) and the
// environment switches -- no HIP call, so that the planning-only engine of iqhip_debug_create_planner shares it
static void configure_engine(iqhip_engine *e, int device, int nstates, int nstates_user, int ncat, int64_t nptn, int ntaxa) {
    e->device = device;
    e->n = nstates;
    e->n_user = nstates_user;
    e->embed2 = nstates_user != nstates;   // the caller's state count is embedded into the next kernel size
    e->scalar_rule_all = nstates_user != 2 && nstates_user != 4 && nstates_user != 20 && nstates_user != 64;
    e->ncat = ncat;
    e->ntaxa = ntaxa;
    e->nptn = nptn;
    e->mfma = nstates != 4;
    e->mfma_pipelined_ok = ((nstates == 20 && (ncat == 4 || ncat == 1)) || (nstates == 64 && ncat == 1)) &&
                           !getenv("IQHIP_MFMA_V1");
    e->mfma_pipelined = e->mfma_pipelined_ok;
    e->tile = e->mfma ? 16 : 64;
    e->block = nstates * ncat;
    e->nptn_pad = round_up(nptn, 64);
    e->ntiles = e->nptn_pad / e->tile;

    if (const char *ab = getenv("IQHIP_ABLATE")) e->ablate = atoi(ab);
    if (const char *cp = getenv("IQHIP_CHECK_PLAN")) e->check_plans = atoi(cp) != 0;
    if (const char *h = getenv("IQHIP_HOLD")) e->use_hold = atoi(h) != 0;
    if (const char *h = getenv("IQHIP_HOLD_LDS")) e->hold_lds = atoi(h) != 0;
    if (const char *h = getenv("IQHIP_NEWTON_POSTS")) e->newton_posts = atoi(h) != 0;
    if (const char *h = getenv("IQHIP_SMALL_PLANS")) e->small_plans = atoi(h) != 0;
    if (const char *h = getenv("IQHIP_MIXED_TOP")) e->mixed_top = atoi(h) != 0;
    if (const char *f = getenv("IQHIP_FOLD")) e->fold_reduce = atoi(f) != 0;
    if (const char *f = getenv("IQHIP_POLL")) e->poll_result = atoi(f) != 0;
    if (const char *f = getenv("IQHIP_CHERRY_TABLES")) e->cherry_on = atoi(f) != 0;
    if (const char *sp = getenv("IQHIP_SPLIT")) e->split_target = atoi(sp);
    if (const char *kb = getenv("IQHIP_LDS_KB")) {
        int v = atoi(kb);
        if (v >= 8 && v <= 150) e->lds_budget_bytes = v * 1024;
    }
    // K2 tables for leaf children: on for 64 states (matrix-pipe bound: 0.50 -> 0.40 ms per traversal at 50 x 20k);
    // off for 20 states, where the traversal is not bound by the MFMA count (1.08 vs 1.11 ms at 100 x 50k) and a model
    // change would cost a table rebuild per evaluation.  IQHIP_LEAF_TABLES=0|1 overrides (tests run both).
    e->leaf_tables = e->mfma_pipelined_ok && nstates == 64;
    if (const char *lt = getenv("IQHIP_LEAF_TABLES")) e->leaf_tables = e->mfma_pipelined_ok && atoi(lt) != 0;
    if (const char *wg = getenv("IQHIP_WG")) {
        int v = atoi(wg);
        if (v == 64 || v == 128 || v == 256) e->wg_size = v;
    }

    // 4-state kernel, two lanes per pattern: twice the waves with half the register state each, as long as
    // they all fit the chip at once (2 waves per SIMD).  Measured, GTR+G4 50 taxa: 10k..65k patterns 0.116..
    // 0.130 ms -> 0.086..0.114 ms; 80k patterns 0.145 -> 0.191 ms (second round).  IQHIP_LANE_SPLIT=1|2 overrides
    if (!e->mfma && ncat % 2 == 0) {
        e->lane_split = (2 * (e->nptn_pad / 64) <= 2 * (int64_t)e->num_cus * 4) ? 2 : 1;
        if (const char *ls = getenv("IQHIP_LANE_SPLIT")) e->lane_split = (atoi(ls) == 2) ? 2 : 1;
    }
    // 20 states x 4 categories on a small alignment: one wave per (tile, category) while that is at most one wave
    // per SIMD (100 taxa: 500 patterns 0.221 -> 0.111 ms, 2000 patterns 0.219 -> 0.145 ms, 8000 patterns 0.247 -> 0.267 ms)
    if (e->mfma_pipelined_ok && e->n == 20 && e->ncat == 4) {
        e->cat_split = 4 * e->ntiles <= (int64_t)e->num_cus * 4;
        if (const char *cs = getenv("IQHIP_CAT_SPLIT")) e->cat_split = atoi(cs) != 0;
        // the dependent top stage of a staged plan with two waves per tile (two categories each, three waves per SIMD) while the
        // alignment has only a few tiles per SIMD: 3125 tiles on 2048 two-wave slots took two rounds of full chains (-1.8 %)
        e->top_cs2 = !e->cat_split && e->ntiles < 6 * (int64_t)e->num_cus * 4;
        if (const char *cs = getenv("IQHIP_TOP_CS2")) e->top_cs2 = atoi(cs) != 0;
    }
    if (e->mfma_pipelined_ok && e->n == 64 && e->ncat == 1) {
        e->row_split = 4 * e->ntiles <= (int64_t)e->num_cus * 4;
        if (const char *rs = getenv("IQHIP_ROW_SPLIT")) e->row_split = atoi(rs) != 0;
    }
}

extern "C" int iqhip_create(iqhip_engine **out, int device, int nstates, int ncat, int64_t nptn,
                            int ntaxa) {
    if (!out) return fail(IQHIP_ERR_INVALID, "iqhip_create: out == NULL");
    *out = nullptr;
    if (nptn <= 0 || ntaxa < 2 || ncat < 1)
        return fail(IQHIP_ERR_INVALID, "iqhip_create: bad nptn/ntaxa/ncat");
    const int nstates_user = nstates;
    // binary data, and every state count the reference hands to its scalar kernel (morphological / multi-state data,
    // phylotreesse.cpp:281-309), run on the next kernel size up through an exact embedding (iqhip_internal.h: embed2)
    if (nstates < 2 || nstates > 64) return fail(IQHIP_ERR_UNSUPPORTED, "iqhip_create: nstates must be 2 .. 64");
    nstates = nstates <= 4 ? 4 : nstates <= 20 ? 20 : 64;
    if (nstates == 4 && !(ncat >= 1 && ncat <= 8))
        return fail(IQHIP_ERR_UNSUPPORTED, "iqhip_create: 4-state path supports ncat in {1..8}");
    if (nstates != 4 && ncat > (nstates == 20 ? 96 : 16))  // 20 states: (class, rate) components of mixtures
        return fail(IQHIP_ERR_UNSUPPORTED, "iqhip_create: ncat must be <= 16 (<= 96 components for 20 states)");
    // the 4-state kernels address a vector slab with a wave-uniform base + one 32-bit per-lane byte offset
    // (kernels_valu4.hip, voff): a slab of 4 GiB or more would wrap silently, so it is refused here
    if (nstates == 4 && (uint64_t)round_up(nptn, 64) * (uint64_t)(nstates * ncat) * 8u >= (1ull << 32))
        return fail(IQHIP_ERR_UNSUPPORTED,
                    "iqhip_create: nptn * nstates * ncat * 8 must stay below 4 GiB per vector on the 4-state path "
                    "(shard the patterns over more engines)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(IQHIP_ERR_NO_DEVICE, "iqhip_create: no HIP device available");
    if (device < 0 || device >= ndev) return fail(IQHIP_ERR_INVALID, "iqhip_create: bad device id");
    HIPCHK(hipSetDevice(device));

    iqhip_engine *e = new iqhip_engine();
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0)
            e->num_cus = cus;
    }
    configure_engine(e, device, nstates, nstates_user, ncat, nptn, ntaxa);

    hipError_t s = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
    if (s != hipSuccess) {
        delete e;
        return fail(IQHIP_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(s));
    }
    e->own_stream = true;
    const size_t P = (size_t)e->nptn_pad;
    bool ok = dmalloc(&e->d_states, (size_t)ntaxa * P) == hipSuccess &&
              dmalloc(&e->d_freq, P) == hipSuccess && dmalloc(&e->d_invar, P) == hipSuccess &&
              dmalloc(&e->d_theta, P * e->block) == hipSuccess &&
              dmalloc(&e->d_pattern_lh, P) == hipSuccess;
    e->result_cap = 8 + 16384;  // up to 16384 node updates per submission
    // default result buffer: pinned host memory mapped into the device address space -- the
    // reduction kernel writes the handful of result doubles straight to the host (no D2H copy
    // on the critical path); a caller-bound device buffer (RCCL) replaces it
    ok = ok && hipHostMalloc((void **)&e->h_result, e->result_cap * sizeof(double), hipHostMallocMapped) == hipSuccess &&
         hipHostGetDevicePointer((void **)&e->d_result_own, e->h_result, 0) == hipSuccess &&
         hipHostMalloc((void **)&e->h_done, 64, hipHostMallocMapped) == hipSuccess &&
         hipHostGetDevicePointer((void **)&e->d_done, (void *)e->h_done, 0) == hipSuccess &&
         hipEventCreateWithFlags(&e->staging_free, hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        iqhip_destroy(e);
        return fail(IQHIP_ERR_NOMEM, "iqhip_create: device allocation failed");
    }
    ok = dmalloc(&e->dummy.plh, P * e->block) == hipSuccess && dmalloc(&e->dummy.sc, P) == hipSuccess;
    if (!ok) {
        iqhip_destroy(e);
        return fail(IQHIP_ERR_NOMEM, "iqhip_create: device allocation failed");
    }
    hipMemsetAsync(e->dummy.plh, 0, P * e->block * sizeof(double), e->stream);
    hipMemsetAsync(e->dummy.sc, 0, P * sizeof(int16_t), e->stream);
    {
        ok = dmalloc(&e->d_newton_partials, (size_t)4 * e->num_cus) == hipSuccess &&
             dmalloc(&e->d_newton_barrier, 2) == hipSuccess && dmalloc(&e->d_fold_ticket, 4) == hipSuccess &&
             dmalloc(&e->d_newton_posts, (size_t)2 * kNewtonPostEpochs * (2 * e->num_cus) * 2) == hipSuccess &&
             dmalloc(&e->d_fold_flags, (size_t)e->result_cap) == hipSuccess;
        if (ok) hipMemsetAsync(e->d_newton_barrier, 0, 2 * sizeof(unsigned int), e->stream);
        if (ok) hipMemsetAsync(e->d_newton_posts, 0xFF, (size_t)2 * kNewtonPostEpochs * (2 * e->num_cus) * 2 * sizeof(double), e->stream);
        if (ok) hipMemsetAsync(e->d_fold_ticket, 0, 4 * sizeof(unsigned int), e->stream);
        if (ok) hipMemsetAsync(e->d_fold_flags, 0, (size_t)e->result_cap * sizeof(int), e->stream);
        if (!ok) {
            iqhip_destroy(e);
            return fail(IQHIP_ERR_NOMEM, "iqhip_create: device allocation failed");
        }
    }
    e->d_result = e->d_result_own;
    hipMemsetAsync(e->d_theta, 0, P * e->block * sizeof(double), e->stream);
    hipMemsetAsync(e->d_pattern_lh, 0, P * sizeof(double), e->stream);
    memset(e->h_result, 0, e->result_cap * sizeof(double));
    *e->h_done = 0;
    hipStreamSynchronize(e->stream);
    *out = e;
    return IQHIP_OK;
}

extern "C" void iqhip_destroy(iqhip_engine *e) {
    if (!e) return;
    if (!e->shards.empty()) {
        sharded::destroy(e);
        delete e;
        return;
    }
    if (e->planner) {
        free(e->h_ops);
        delete e;
        return;
    }
    use_device(e);
    if (e->stream) hipStreamSynchronize(e->stream);
    if (e->pair) iqhip_destroy(e->pair);   // (runs on this engine's stream, which it does not own)
    e->pair = nullptr;
    if (e->d_cherry_tab) hipFree(e->d_cherry_tab);
    comm_destroy(e);
    if (e->d_result_dev) hipFree(e->d_result_dev);
    if (e->d_nstate) hipFree(e->d_nstate);
    if (e->d_bstates) hipFree(e->d_bstates);
    if (e->h_nstate) hipHostFree(e->h_nstate);
    if (e->stream) hipStreamSynchronize(e->stream);
    for (auto &s : e->slabs) {
        if (s.plh) hipFree(s.plh);
        if (s.sc) hipFree(s.sc);
    }
    void *ptrs[] = {e->d_states, e->d_freq, e->d_invar, e->d_model, e->d_ops, e->d_slab,
                    e->d_theta, e->d_pattern_lh, e->d_leaf_tab, e->dummy.plh, e->dummy.sc, e->d_newton_partials,
                    e->d_newton_barrier, e->d_newton_posts, e->d_fold_ticket, e->d_fold_flags, e->d_ptn_scaled, e->d_boot, e->d_img, e->d_theta_batch, e->d_batch_partials,
                    e->d_batch_out, e->d_batch_barriers, e->d_batch_tasks, e->d_batch_posts, e->d_sweep_len};
    for (void *p : ptrs)
        if (p) hipFree(p);
    if (e->h_ops) hipHostFree(e->h_ops);
    if (e->h_sweep_desc) hipHostFree(e->h_sweep_desc);
    if (e->h_plan_arena) hipHostFree(e->h_plan_arena);
    if (e->d_sweep_desc) hipFree(e->d_sweep_desc);
    if (e->d_sweep_posts) hipFree(e->d_sweep_posts);
    if (e->h_result) hipHostFree(e->h_result);
    if (e->h_done) hipHostFree((void *)e->h_done);
    if (e->staging_free) hipEventDestroy(e->staging_free);
    for (auto &p : e->tev) {
        hipEventDestroy(p.first);
        hipEventDestroy(p.second);
    }
    for (auto &p : e->cev) {
        hipEventDestroy(p.first);
        hipEventDestroy(p.second);
    }
    if (e->own_stream && e->stream) hipStreamDestroy(e->stream);
    delete e;
}

extern "C" int iqhip_set_stream(iqhip_engine *e, void *hip_stream) {
    if (!e) return fail(IQHIP_ERR_INVALID, "null engine");
    if (!e->shards.empty()) return fail(IQHIP_ERR_UNSUPPORTED, "a sharded engine runs on its shards' own streams");
    HIPCHK(use_device(e));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (e->own_stream) hipStreamDestroy(e->stream);
    e->stream = (hipStream_t)hip_stream;
    e->own_stream = false;
    if (e->pair) return iqhip_set_stream(e->pair, hip_stream);
    return IQHIP_OK;
}

// ---------------------------------------------------------------------------------------
// slabs
// ---------------------------------------------------------------------------------------
static int new_slab(iqhip_engine *e, int *idx) {
    if (!e->free_slabs.empty()) {
        *idx = e->free_slabs.back();
        e->free_slabs.pop_back();
        return IQHIP_OK;
    }
    Slab s;
    const size_t P = (size_t)e->nptn_pad;
    if (e->planner) {
        s.plh = fake_alloc<double>(e, P * e->block);
        s.sc = fake_alloc<int16_t>(e, P);
        e->slabs.push_back(s);
        *idx = (int)e->slabs.size() - 1;
        return IQHIP_OK;
    }
    if (dmalloc(&s.plh, P * e->block) != hipSuccess) return fail(IQHIP_ERR_NOMEM, "slab alloc");
    if (dmalloc(&s.sc, P) != hipSuccess) {
        hipFree(s.plh);
        return fail(IQHIP_ERR_NOMEM, "slab alloc");
    }
    // padded lanes must hold finite values from the start
    hipMemsetAsync(s.plh, 0, P * e->block * sizeof(double), e->stream);
    hipMemsetAsync(s.sc, 0, P * sizeof(int16_t), e->stream);
    e->slabs.push_back(s);
    *idx = (int)e->slabs.size() - 1;
    return IQHIP_OK;
}

static int slab_for_key(iqhip_engine *e, uint64_t key, bool create, int *idx) {
    auto it = e->key2slab.find(key);
    if (it != e->key2slab.end()) {
        *idx = it->second;
        return IQHIP_OK;
    }
    if (!create) return fail(IQHIP_ERR_INVALID, "unknown partial_lh key (vector never computed)");
    int rc = new_slab(e, idx);
    if (rc) return rc;
    e->key2slab[key] = *idx;
    e->keymap_version++;
    return IQHIP_OK;
}

extern "C" int iqhip_reserve(iqhip_engine *e, int nvectors) {
    if (!e) return fail(IQHIP_ERR_INVALID, "null engine");
    if (!e->shards.empty()) return sharded::reserve(e, nvectors);
    HIPCHK(use_device(e));
    int have = (int)e->slabs.size();
    for (int i = have; i < nvectors; i++) {
        int idx;
        // temporarily empty the free list so that new_slab really allocates
        std::vector<int> keep;
        keep.swap(e->free_slabs);
        int rc = new_slab(e, &idx);
        keep.swap(e->free_slabs);
        if (rc) return rc;
        e->free_slabs.push_back(idx);
    }
    return IQHIP_OK;
}

extern "C" int iqhip_release(iqhip_engine *e, uint64_t key) {
    if (!e) return fail(IQHIP_ERR_INVALID, "null engine");
    if (!e->shards.empty()) return sharded::release(e, key);
    auto it = e->key2slab.find(key);
    if (it == e->key2slab.end()) return IQHIP_OK;
    e->free_slabs.push_back(it->second);
    e->key2slab.erase(it);
    e->keymap_version++;
    return IQHIP_OK;
}

extern "C" int iqhip_rekey(iqhip_engine *e, uint64_t old_key, uint64_t new_key) {
    if (!e) return fail(IQHIP_ERR_INVALID, "null engine");
    if (!e->shards.empty()) return sharded::rekey(e, old_key, new_key);
    auto it = e->key2slab.find(old_key);
    if (it == e->key2slab.end()) return fail(IQHIP_ERR_INVALID, "iqhip_rekey: unknown key");
    if (old_key == new_key) return IQHIP_OK;
    if (e->key2slab.count(new_key)) iqhip_release(e, new_key);
    int idx = it->second;
    e->key2slab.erase(it);
    e->key2slab[new_key] = idx;
    e->keymap_version++;
    return IQHIP_OK;
}

// ---------------------------------------------------------------------------------------
// inputs
// ---------------------------------------------------------------------------------------
extern "C" int iqhip_set_ptn_freq(iqhip_engine *e, const double *ptn_freq) {
    if (!e || !ptn_freq) return fail(IQHIP_ERR_INVALID, "null argument");
    if (!e->shards.empty()) return sharded::set_ptn_array(e, ptn_freq, false);
    HIPCHK(use_device(e));
    std::vector<double> tmp((size_t)e->nptn_pad, 0.0);
    memcpy(tmp.data(), ptn_freq, sizeof(double) * (size_t)e->nptn);
    HIPCHK(hipMemcpyAsync(e->d_freq, tmp.data(), tmp.size() * sizeof(double), hipMemcpyHostToDevice,
                          e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return IQHIP_OK;
}

extern "C" int iqhip_set_ptn_invar(iqhip_engine *e, const double *ptn_invar) {
    if (!e || !ptn_invar) return fail(IQHIP_ERR_INVALID, "null argument");
    if (!e->shards.empty()) return sharded::set_ptn_array(e, ptn_invar, true);
    HIPCHK(use_device(e));
    std::vector<double> tmp((size_t)e->nptn_pad, 0.0);
    memcpy(tmp.data(), ptn_invar, sizeof(double) * (size_t)e->nptn);
    HIPCHK(hipMemcpyAsync(e->d_invar, tmp.data(), tmp.size() * sizeof(double),
                          hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return IQHIP_OK;
}

extern "C" int iqhip_set_alignment(iqhip_engine *e, const uint8_t *states, const double *ptn_freq,
                                   const double *ptn_invar) {
    if (!e || !states || !ptn_freq || !ptn_invar) return fail(IQHIP_ERR_INVALID, "null argument");
    if (!e->shards.empty()) return sharded::set_alignment(e, states, ptn_freq, ptn_invar);
    if (!e->model_set)
        return fail(IQHIP_ERR_INVALID,
                    "iqhip_set_alignment: call iqhip_set_model first (needs state_unknown)");
    HIPCHK(use_device(e));
    const size_t P = (size_t)e->nptn_pad, N = (size_t)e->nptn;
    // (embedded data: padding patterns and missing characters are the ambiguity code whose tip row is the caller's
    // unknown row -- internal state n -- never the kernels' own unknown state, whose probability-space vector of exactly
    // 1.0 in every component would leak into the padding components)
    std::vector<uint8_t> tmp((size_t)e->ntaxa * P, (uint8_t)(e->embed2 ? e->n : e->state_unknown));
    for (int t = 0; t < e->ntaxa; t++) {
        const uint8_t *src = states + (size_t)t * N;
        if (e->embed2) {
            uint8_t *dst = tmp.data() + (size_t)t * P;
            for (size_t p = 0; p < N; p++) {
                if (src[p] > e->n_user) return fail(IQHIP_ERR_INVALID, "iqhip_set_alignment: state > STATE_UNKNOWN");
                dst[p] = src[p] == e->n_user ? (uint8_t)e->n : src[p];
            }
            continue;
        }
        for (size_t p = 0; p < N; p++)
            if (src[p] > e->state_unknown)
                return fail(IQHIP_ERR_INVALID, "iqhip_set_alignment: state > STATE_UNKNOWN");
        memcpy(tmp.data() + (size_t)t * P, src, N);
    }
    HIPCHK(hipMemcpyAsync(e->d_states, tmp.data(), tmp.size(), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    int rc = iqhip_set_ptn_freq(e, ptn_freq);
    if (rc) return rc;
    rc = iqhip_set_ptn_invar(e, ptn_invar);
    if (rc) return rc;
    e->aln_set = true;
    return IQHIP_OK;
}

extern "C" int iqhip_set_ascertainment(iqhip_engine *e, int64_t n_unobserved, double nsites) {
    if (!e) return fail(IQHIP_ERR_INVALID, "null engine");
    if (n_unobserved < 0 || n_unobserved >= e->nptn || (n_unobserved > 0 && !(nsites > 0.0)))
        return fail(IQHIP_ERR_INVALID, "iqhip_set_ascertainment: bad pattern count / site count");
    if (!e->shards.empty()) return sharded::set_ascertainment(e, n_unobserved, nsites);
    e->n_unobs = n_unobserved;
    e->asc_nsites = nsites;
    // one rank of a pattern-sharded run (iqhip_comm_init_rank): every rank passes the alignment's site count; the ranks
    // that hold none of the unobserved patterns pass n_unobserved = 0 with nsites > 0
    e->asc_active = n_unobserved > 0 || (e->comm && nsites > 0.0);
    e->pattern_lh_shift = 0.0;
    return IQHIP_OK;
}

static int set_model_common(iqhip_engine *e, int nclass, const int32_t *cat_class, const double *eval,
                            const double *evec, const double *inv_evec, const double *rates, const double *props,
                            int state_unknown, const double *tip /* [state][class][n] */) {
    if (!e || !eval || !evec || !inv_evec || !rates || !props || !tip)
        return fail(IQHIP_ERR_INVALID, "null argument");
    if (!e->shards.empty())
        return sharded::set_model(e, nclass, cat_class, eval, evec, inv_evec, rates, props, state_unknown, tip);
    if (state_unknown < e->n || state_unknown > (e->mfma ? 255 : 31))
        return fail(IQHIP_ERR_INVALID, "iqhip_set_model: state_unknown out of range");
    if (e->aln_set && state_unknown != e->state_unknown)
        return fail(IQHIP_ERR_INVALID, "iqhip_set_model: state_unknown changed after set_alignment");
    if (nclass < 1 || nclass > e->ncat) return fail(IQHIP_ERR_INVALID, "bad number of mixture classes");
    // Mixtures: 20 states have a kernel of their own (k_traverse_mfma_mix20); 64 and 4 states take the generic
    // matrix-core kernel with per-class A images.  A 4-state engine therefore changes its vector layout (64-pattern
    // tiles of the VALU kernels <-> 16-pattern tiles of the matrix-core kernels) when the model becomes / stops being
    // a mixture; as with every model change the caller invalidates all vectors (clearAllPartialLH).
    if (e->n == 4) {
        const bool want_mfma = nclass > 1;
        if (want_mfma != e->mfma) {
            HIPCHK(use_device(e));
            HIPCHK(hipStreamSynchronize(e->stream));
            e->mfma = want_mfma;
            if (want_mfma) { e->lane_split_valu = e->lane_split; e->lane_split = 1; }  // (a VALU-kernel notion)
            else e->lane_split = e->lane_split_valu;
            e->tile = want_mfma ? 16 : 64;
            e->ntiles = e->nptn_pad / e->tile;
            e->uploaded_plan.clear();
            e->last_plan_version = 0;
            e->theta_valid = false;
            // k_newton's posted exchange resets, per launch, the slots its OWN grid used in the other parity; the grid
            // follows ntiles, so after a layout change start both parities from the all-ones state again
            HIPCHK(hipMemsetAsync(e->d_newton_posts, 0xFF, (size_t)2 * kNewtonPostEpochs * (2 * e->num_cus) * 2 * sizeof(double), e->stream));
            e->newton_post_launches = 0;
        }
    }
    std::vector<int> cls(e->ncat, 0);
    if (nclass > 1) {
        if (!cat_class) return fail(IQHIP_ERR_INVALID, "null argument");
        for (int c = 0; c < e->ncat; c++) {
            if (cat_class[c] < 0 || cat_class[c] >= nclass) return fail(IQHIP_ERR_INVALID, "category class out of range");
            cls[c] = cat_class[c];
        }
    }
    HIPCHK(use_device(e));
    const int n = e->n, C = e->ncat;
    HIPCHK(hipStreamSynchronize(e->stream));  // previous work may still read the old model
    // One device block, one copy per model change (the model optimisers call this once per evaluation):
    // {eval, evec, inv_evec, tip} of class 0 for the 4-state and pipelined kernels, rates, props, the
    // per-category expansions evalc[c][i], tipc[state][c][i], and the category -> class map.
    const size_t nst = (size_t)state_unknown + 1;
    const size_t o_eval = 0, o_evec = o_eval + n, o_ievec = o_evec + (size_t)n * n, o_rates = o_ievec + (size_t)n * n,
                 o_props = o_rates + C, o_tip = o_props + C, o_evalc = o_tip + nst * n, o_tipc = o_evalc + (size_t)C * n,
                 o_cls = o_tipc + nst * C * n;
    const int a_mt = n / 16, a_ks = n / 4;
    const bool a_tail = (n % 16) == 4;
    const size_t aimg_doubles = (n == 20 || n == 64) ? (size_t)2 * a_mt * a_ks * 64 + (a_tail ? (size_t)2 * a_ks * 64 : 0) : 0;
    const size_t o_aimg = (o_cls + ((size_t)C + 1) / 2 + 1) / 2 * 2;   // 16-byte aligned
    const size_t total = o_aimg + aimg_doubles;
    std::vector<double> blk(total, 0.0);
    memcpy(&blk[o_eval], eval, sizeof(double) * n);
    memcpy(&blk[o_evec], evec, sizeof(double) * n * n);
    memcpy(&blk[o_ievec], inv_evec, sizeof(double) * n * n);
    memcpy(&blk[o_rates], rates, sizeof(double) * C);
    memcpy(&blk[o_props], props, sizeof(double) * C);
    for (size_t s = 0; s < nst; s++) memcpy(&blk[o_tip + s * n], &tip[(s * nclass) * n], sizeof(double) * n);
    for (int c = 0; c < C; c++) memcpy(&blk[o_evalc + (size_t)c * n], &eval[(size_t)cls[c] * n], sizeof(double) * n);
    for (size_t s = 0; s < nst; s++)
        for (int c = 0; c < C; c++)
            memcpy(&blk[o_tipc + (s * C + c) * n], &tip[(s * nclass + cls[c]) * n], sizeof(double) * n);
    memcpy(&blk[o_cls], cls.data(), sizeof(int) * C);
    if (aimg_doubles) {
        double *U = &blk[o_aimg], *Ui = U + (size_t)a_mt * a_ks * 64, *U4 = Ui + (size_t)a_mt * a_ks * 64, *Ui4 = U4 + (size_t)a_ks * 64;
        for (int m = 0; m < a_mt; m++)
            for (int ks = 0; ks < a_ks; ks++)
                for (int l = 0; l < 64; l++) {
                    const int row = 16 * m + (l & 15), k = 4 * ks + (l >> 4);
                    U[((size_t)m * a_ks + ks) * 64 + l] = evec[(size_t)row * n + k];
                    Ui[((size_t)m * a_ks + ks) * 64 + l] = inv_evec[(size_t)row * n + k];
                }
        if (a_tail)
            for (int ks = 0; ks < a_ks; ks++)
                for (int l = 0; l < 64; l++) {
                    const int row = 16 * a_mt + (l & 3), k = 4 * ks + (l >> 4);
                    U4[(size_t)ks * 64 + l] = evec[(size_t)row * n + k];
                    Ui4[(size_t)ks * 64 + l] = inv_evec[(size_t)row * n + k];
                }
    }
    if (e->d_model && e->model_cap < total) {
        HIPCHK(hipFree(e->d_model));
        e->d_model = nullptr;
    }
    if (!e->d_model) {
        HIPCHK(dmalloc(&e->d_model, total));
        e->model_cap = total;
    }
    HIPCHK(hipMemcpy(e->d_model, blk.data(), sizeof(double) * total, hipMemcpyHostToDevice));
    e->d_eval = e->d_model + o_eval;
    e->d_evec = e->d_model + o_evec;
    e->d_inv_evec = e->d_model + o_ievec;
    e->d_rates = e->d_model + o_rates;
    e->d_props = e->d_model + o_props;
    e->d_tip = e->d_model + o_tip;
    e->d_evalc = e->d_model + o_evalc;
    e->d_tipc = e->d_model + o_tipc;
    e->d_cls = reinterpret_cast<int *>(e->d_model + o_cls);
    e->d_aimg = aimg_doubles ? e->d_model + o_aimg : nullptr;
    e->aimg_doubles = (int)aimg_doubles;
    if (nclass > 1 || (e->n == 20 && !e->mfma_pipelined_ok)) {  // (20 states with a category count that has no
        // pipelined instantiation also run on the mixture kernel: one class)
        // MFMA A-operand images of every class for k_traverse_mfma_mix20: [class][U16 | U4 | Ui16 | Ui4][s][lane]
        // (16-row tile: row = lane & 15; 4-row tail: row = 16 + (lane & 3); k = 4s + (lane >> 4)), followed by
        // the padded two-tile images [class][U | U^-1][m][s][lane] of the generic kernel (IQHIP_MIX_GENERIC)
        const int MT = (n + 15) / 16, KS = n / 4;
        const size_t mix_doubles = (size_t)nclass * 4 * KS * 64;  // (only read by the 20-state kernel)
        std::vector<double> img(mix_doubles + (size_t)nclass * 2 * MT * KS * 64, 0.0);
        for (int m = 0; m < nclass; m++) {
            const double *U = evec + (size_t)m * n * n, *Ui = inv_evec + (size_t)m * n * n;
            for (int s = 0; s < KS; s++)
                for (int l = 0; l < 64; l++) {
                    const int k = 4 * s + (l >> 4), r16 = l & 15, r4 = 16 + (l & 3);
                    double *b = &img[(size_t)m * 4 * KS * 64];
                    if (r16 < n) {
                        b[(0 * KS + s) * 64 + l] = U[r16 * n + k];
                        b[(2 * KS + s) * 64 + l] = Ui[r16 * n + k];
                    }
                    if (r4 < n) {
                        b[(1 * KS + s) * 64 + l] = U[r4 * n + k];
                        b[(3 * KS + s) * 64 + l] = Ui[r4 * n + k];
                    }
                }
            for (int t = 0; t < MT * KS * 64; t++) {
                const int l = t & 63, ms = t >> 6, s = ms % KS, mt = ms / KS;
                const int row = 16 * mt + (l & 15), k = 4 * s + (l >> 4);
                if (row < n) {
                    img[mix_doubles + ((size_t)m * 2 + 0) * MT * KS * 64 + t] = U[row * n + k];
                    img[mix_doubles + ((size_t)m * 2 + 1) * MT * KS * 64 + t] = Ui[row * n + k];
                }
            }
        }
        e->img_generic_off = mix_doubles;
        if (e->d_img && e->img_cap < img.size()) {
            HIPCHK(hipFree(e->d_img));
            e->d_img = nullptr;
        }
        if (!e->d_img) {
            HIPCHK(dmalloc(&e->d_img, img.size()));
            e->img_cap = img.size();
        }
        HIPCHK(hipMemcpy(e->d_img, img.data(), sizeof(double) * img.size(), hipMemcpyHostToDevice));
    }
    // the pipelined kernels hold one eigen-system in registers / LDS: mixtures take the generic kernel,
    // whose plans have a different canonical form -> drop the cached descriptors
    const bool pipelined = e->mfma_pipelined_ok && nclass == 1;
    if (pipelined != e->mfma_pipelined || nclass != e->nclass) {
        e->mfma_pipelined = pipelined;
        e->uploaded_plan.clear();
        e->last_plan_version = 0;
    }
    e->nclass = nclass;
    e->state_unknown = state_unknown;
    e->model_set = true;
    e->theta_valid = false;
    e->model_version++;
    e->cherry_model_synced = false;
    return IQHIP_OK;
}

// embedded data: pad the caller's m-state system (m = n_user) to the n-state one the kernels run (see iqhip_engine::embed2):
// eigenvalues (l_0 .. l_m-1, 0 ...), U = diag(U_m, I), U^-1 = diag(U_m^-1, I); tip rows of the m states padded with zeros;
// internal state n = "missing" with the caller's unknown row, internal STATE_UNKNOWN = n + 1 (never present in the data)
// ---------------------------------------------------------------------------------------
// cherry tables (DevOp::cherry): which engines use them, and their pair engine
// ---------------------------------------------------------------------------------------
static bool cherry_candidate(const iqhip_engine *e) {
    if (!e || !e->cherry_on || !e->shards.empty() || e->ablate) return false;   // (a planning-only engine plans them too)
    if (!e->mfma_pipelined_ok || e->n_user != e->n) return false;
    if (e->n == 20) return e->ncat == 4 && !e->leaf_tables && !e->cat_split && e->nptn_pad >= 8 * 1024;
    return false;
}

static int cherry_sync_model(iqhip_engine *e, const double *eval, const double *evec, const double *inv_evec,
                             const double *rates, const double *props, int state_unknown, const double *tip) {
    const int s2 = state_unknown + 1;
    // a table costs one node update over s2^2 patterns per new pair of pendant lengths, and 8 * block * s2^2 bytes
    if (s2 * s2 > 4356 || (int64_t)4 * s2 * s2 > e->nptn_pad) {
        if (e->pair) iqhip_destroy(e->pair);
        e->pair = nullptr;
        return IQHIP_OK;
    }
    if (e->pair && e->cherry_s2 != s2) {
        HIPCHK(hipStreamSynchronize(e->stream));
        iqhip_destroy(e->pair);
        e->pair = nullptr;
    }
    int rc = IQHIP_OK;
    if (!e->pair) {
        const int npairs = s2 * s2;
        rc = iqhip_create(&e->pair, e->device, e->n, e->ncat, npairs, 2);
        if (rc) return rc;
        e->pair->cherry_on = false;
        e->pair->check_plans = e->check_plans;
        rc = iqhip_set_stream(e->pair, e->stream);
        if (rc) return rc;
        e->cherry_s2 = s2;
        e->cherry_npairs = (int)e->pair->nptn_pad;
        e->cherry_slot_of.clear();
        e->cherry_slots.clear();
    }
    rc = set_model_common(e->pair, 1, nullptr, eval, evec, inv_evec, rates, props, state_unknown, tip);
    if (rc) return rc;
    if (!e->pair->aln_set) {
        const int npairs = s2 * s2;
        std::vector<uint8_t> st((size_t)2 * npairs);
        for (int q = 0; q < npairs; q++) {
            st[q] = (uint8_t)(q / s2);
            st[(size_t)npairs + q] = (uint8_t)(q % s2);
        }
        const std::vector<double> ones((size_t)npairs, 1.0), zeros((size_t)npairs, 0.0);
        rc = iqhip_set_alignment(e->pair, st.data(), ones.data(), zeros.data());
    }
    return rc;
}

static int set_model_binary(iqhip_engine *e, const double *eval, const double *evec, const double *inv_evec,
                            const double *rates, const double *props, int state_unknown, const double *tip) {
    if (!eval || !evec || !inv_evec || !rates || !props || !tip) return fail(IQHIP_ERR_INVALID, "null argument");
    const int m = e->n_user, n = e->n;
    if (state_unknown != m)
        return fail(IQHIP_ERR_INVALID, "iqhip_set_model: data of this state count has STATE_UNKNOWN = nstates (no ambiguity codes)");
    std::vector<double> ev((size_t)n, 0.0), U((size_t)n * n, 0.0), Ui((size_t)n * n, 0.0), tp((size_t)(n + 2) * n, 0.0);
    for (int i = 0; i < m; i++) ev[i] = eval[i];
    for (int x = 0; x < m; x++)
        for (int i = 0; i < m; i++) { U[(size_t)x * n + i] = evec[x * m + i]; Ui[(size_t)x * n + i] = inv_evec[x * m + i]; }
    for (int x = m; x < n; x++) U[(size_t)x * n + x] = Ui[(size_t)x * n + x] = 1.0;
    for (int st = 0; st < m; st++)
        for (int i = 0; i < m; i++) tp[(size_t)st * n + i] = tip[st * m + i];
    for (int i = 0; i < m; i++) tp[(size_t)n * n + i] = tip[m * m + i];   // row n: the caller's unknown row
    return set_model_common(e, 1, nullptr, ev.data(), U.data(), Ui.data(), rates, props, n + 1, tp.data());
}

extern "C" int iqhip_set_model(iqhip_engine *e, const double *eval, const double *evec,
                               const double *inv_evec, const double *rates, const double *props,
                               int state_unknown, const double *tip_partial_lh) {
    if (e && e->embed2 && e->shards.empty())
        return set_model_binary(e, eval, evec, inv_evec, rates, props, state_unknown, tip_partial_lh);
    int rc = set_model_common(e, 1, nullptr, eval, evec, inv_evec, rates, props, state_unknown, tip_partial_lh);
    if (!rc && cherry_candidate(e)) {
        rc = cherry_sync_model(e, eval, evec, inv_evec, rates, props, state_unknown, tip_partial_lh);
        e->cherry_model_synced = !rc && e->pair != nullptr;
    }
    return rc;
}

extern "C" int iqhip_set_mixture_model(iqhip_engine *e, int nclass, const int32_t *cat_class, const double *eval,
                                       const double *evec, const double *inv_evec, const double *rates,
                                       const double *props, int state_unknown, const double *tip_partial_lh) {
    if (e && (e->embed2 || (e->n_user != 4 && e->n_user != 20 && e->n_user != 64)) && nclass > 1)
        return fail(IQHIP_ERR_UNSUPPORTED, "mixture models of embedded data (state counts other than 4, 20, 64) are not implemented");
    if (e && e->embed2 && e->shards.empty())
        return set_model_binary(e, eval, evec, inv_evec, rates, props, state_unknown, tip_partial_lh);
    return set_model_common(e, nclass, cat_class, eval, evec, inv_evec, rates, props, state_unknown, tip_partial_lh);
}

// ---------------------------------------------------------------------------------------
// plan building
// ---------------------------------------------------------------------------------------
static int ensure_plan_capacity(iqhip_engine *e, int nops) {
    if (nops <= e->ops_cap) return IQHIP_OK;
    int cap = std::max(64, nops * 2);
    if (e->planner) {
        free(e->h_ops);
        e->h_ops = static_cast<DevOp *>(calloc((size_t)cap, sizeof(DevOp)));
        if (!e->h_ops) return fail(IQHIP_ERR_NOMEM, "plan staging");
        e->d_ops = fake_alloc<DevOp>(e, (size_t)cap);
        e->ops_cap = cap;
        e->uploaded_plan.clear();
        return IQHIP_OK;
    }
    HIPCHK(hipStreamSynchronize(e->stream));
    if (e->d_ops) hipFree(e->d_ops);
    if (e->h_ops) hipHostFree(e->h_ops);
    e->d_ops = nullptr; e->h_ops = nullptr; e->ops_cap = 0;
    e->uploaded_plan.clear();
    HIPCHK(dmalloc(&e->d_ops, cap));
    HIPCHK(hipHostMalloc((void **)&e->h_ops, sizeof(DevOp) * cap));
    e->ops_cap = cap;
    return IQHIP_OK;
}

static int ensure_slab_rows(iqhip_engine *e, int nrows) {
    const int64_t need = (int64_t)nrows * e->ntiles * e->lane_split;
    if (need <= e->slab_cap) return IQHIP_OK;
    HIPCHK(hipStreamSynchronize(e->stream));
    if (e->d_slab) hipFree(e->d_slab);
    e->d_slab = nullptr;
    e->slab_cap = 0;
    HIPCHK(dmalloc(&e->d_slab, (size_t)need));
    e->slab_cap = need;
    return IQHIP_OK;
}

static int check_ready(iqhip_engine *e) {
    if (!e) return fail(IQHIP_ERR_INVALID, "null engine");
    if (!e->model_set || !e->aln_set)
        return fail(IQHIP_ERR_INVALID, "engine needs iqhip_set_model and iqhip_set_alignment first");
    hipError_t s = use_device(e);
    if (s != hipSuccess) return fail(IQHIP_ERR_HIP, hipGetErrorString(s));
    return IQHIP_OK;
}

// Resolve one child of a node op. prev_dst = slab index written by the previous op (-1: none).
static int resolve_child(iqhip_engine *e, uint64_t key, int32_t leaf, int prev_dst,
                         const double **plh, const int16_t **sc, const uint8_t **states,
                         int32_t *kind) {
    *plh = nullptr; *sc = nullptr; *states = nullptr;
    if (leaf >= 0) {
        if (leaf >= e->ntaxa) return fail(IQHIP_ERR_INVALID, "leaf id out of range");
        *states = e->d_states + (size_t)leaf * e->nptn_pad;
        *kind = CHILD_LEAF;
        return IQHIP_OK;
    }
    int idx;
    int rc = slab_for_key(e, key, false, &idx);
    if (rc) return rc;
    *plh = e->slabs[idx].plh;
    *sc = e->slabs[idx].sc;
    *kind = (idx == prev_dst) ? CHILD_PREV : CHILD_LOAD;
    return IQHIP_OK;
}

// The kernels' contract on a plan, checked on the host before the descriptors go to the device (IQHIP_CHECK_PLAN=1; always
// for a planning-only engine).  The traversal kernels issue the requests of op k+1 unconditionally while op k computes
// (streamed child, its counters, leaf state rows, K2 table rows), so EVERY pointer of EVERY descriptor -- the look-ahead
// sentinels behind the last op included -- must be a dereferenceable address of the right kind even when the op does not
// use it: a null tabL / tabR of a non-leaf child was a GPU memory fault in round 2 that this check finds without a GPU.
static int check_plan(iqhip_engine *e, int nops, int nsentinels) {
    std::unordered_set<const void *> vecs, scs;
    for (const Slab &sl : e->slabs) { vecs.insert(sl.plh); scs.insert(sl.sc); }
    vecs.insert(e->dummy.plh);
    scs.insert(e->dummy.sc);
    const size_t per = (e->mfma && e->mfma_pipelined) ? leaf_table_doubles(e) : 0;
    char msg[256];
    auto bad = [&](int k, const char *what) {
        snprintf(msg, sizeof msg, "plan check: op %d of %d (+%d sentinels): %s", k, nops, nsentinels, what);
        return fail(IQHIP_ERR_INVALID, msg);
    };
    auto state_row = [&](const uint8_t *p) {
        if (!p || !e->d_states || p < e->d_states) return false;
        const size_t off = (size_t)(p - e->d_states);
        return off % (size_t)e->nptn_pad == 0 && off / (size_t)e->nptn_pad < (size_t)e->ntaxa;
    };
    auto table = [&](const double *p) {
        if (e->plan_nleaf_tabs == 0 && !e->d_leaf_tab) return p == nullptr;  // kernel variant without tables
        if (!p || !e->d_leaf_tab || p < e->d_leaf_tab || per == 0) return false;
        const size_t off = (size_t)(p - e->d_leaf_tab);
        return off % per == 0 && off / per < e->leaf_tab_slots;
    };
    for (int k = 0; k < nops + nsentinels; k++) {
        const DevOp &d = e->h_ops[k];
        if (!vecs.count(d.dst) || d.dst == nullptr) return bad(k, "dst is not a vector slab");
        if (!scs.count(d.dst_sc)) return bad(k, "dst_sc is not a counter slab");
        if (!vecs.count(d.pf)) return bad(k, "pf (streamed child) is not a vector slab / the dummy slab");
        if (!scs.count(d.pf_sc)) return bad(k, "pf_sc is not a counter slab / the dummy");
        if (!vecs.count(d.ld)) return bad(k, "ld (second memory child) is not a vector slab / the dummy slab");
        if (!scs.count(d.ld_sc)) return bad(k, "ld_sc is not a counter slab / the dummy");
        if (!state_row(d.sl) || !state_row(d.sr)) return bad(k, "sl / sr is not a row of the state matrix");
        if (!table(d.tabL) || !table(d.tabR)) return bad(k, "tabL / tabR is not a K2 table slot");
        if (d.cherry) {
            const size_t cper = (size_t)e->cherry_npairs * e->block;
            if (!e->d_cherry_tab || cper == 0 || d.cherry < e->d_cherry_tab || (size_t)(d.cherry - e->d_cherry_tab) % cper != 0 ||
                (size_t)(d.cherry - e->d_cherry_tab) / cper >= e->cherry_cap || k >= nops || d.left_kind != CHILD_LEAF ||
                d.right_kind != CHILD_LEAF)
                return bad(k, "cherry is not a cherry-table slot of an op with two leaf children");
        }
        if (k >= nops) continue;  // sentinels: pointers only
        if (d.dst == e->dummy.plh || d.dst_sc == e->dummy.sc) return bad(k, "a real op writes the dummy slab");
        const bool lk = d.left_kind == CHILD_LEAF || d.left_kind == CHILD_PF || d.left_kind == CHILD_HOLD ||
                        (d.left_kind == CHILD_LOAD && e->mfma && !e->mfma_pipelined);
        const bool rk = d.right_kind == CHILD_LEAF || d.right_kind == CHILD_PREV || d.right_kind == CHILD_LOAD;
        if (!lk || !rk) return bad(k, "child kinds are not in canonical form");
        if (d.left_kind == CHILD_PF && (d.pf == e->dummy.plh || !(d.real_mask & 1))) return bad(k, "streamed child without a real vector");
        if (d.left_kind != CHILD_PF && !(e->mfma && !e->mfma_pipelined) && (d.real_mask & 1)) return bad(k, "real_mask set without a streamed child");
        if (d.right_kind == CHILD_LOAD && d.ld == e->dummy.plh) return bad(k, "second memory child without a real vector");
        if (d.dst == d.pf || d.dst == d.ld) return bad(k, "op writes one of its own children");
        if (!(d.left_len >= 0.0) || !(d.right_len >= 0.0)) return bad(k, "negative or NaN branch length");
        if (d.out_row < 0 || d.out_row >= nops) return bad(k, "out_row outside the caller's op list");
        if (d.lds_left < 0 || d.lds_right < 0 || d.lds_left > e->plan_lds_doubles || d.lds_right > e->plan_lds_doubles)
            return bad(k, "LDS region outside the launch's allocation");
        if (d.sl_slot < 0 || d.sr_slot < 0 || d.sl_slot >= e->plan_state_slots || d.sr_slot >= e->plan_state_slots)
            return bad(k, "leaf-state slot outside the launch's allocation");
        if (d.chunk_nops < 0 || k + d.chunk_nops > nops) return bad(k, "LDS chunk runs past the plan");
    }
    // chunks tile the plan; the segment table stays inside it
    for (int k = 0; k < nops;) {
        if (e->h_ops[k].chunk_nops <= 0) return bad(k, "op is not covered by an LDS chunk");
        k += e->h_ops[k].chunk_nops;
    }
    const int *tab = reinterpret_cast<const int *>(e->h_ops + e->plan_table_off);
    int covered = 0;
    for (int u = 0; u <= e->plan_nunits; u++) {
        const int b = tab[2 * u], n = tab[2 * u + 1];
        if (b < 0 || n < 0 || b + n > nops) return bad(b, "segment outside the plan");
        covered += n;
    }
    if (covered != nops) return bad(nops, "segments do not cover the plan exactly once");
    return IQHIP_OK;
}

// len_ptrs (sweeps): 2 * nops device pointers, [2k] / [2k+1] = where the length of op k's left / right child branch will
// be found when the op runs (nullptr: the host value in the op)
static int build_plan(iqhip_engine *e, const iqhip_node_op *ops, int nops, int *last_dst,
                      const std::vector<int> *explicit_segs = nullptr, const double *const *len_ptrs = nullptr) {
    constexpr int kSentinels = 2;  // >= the kernels' deepest look-ahead (streamed child: 1 op)
    if (nops + 2 > e->result_cap) return fail(IQHIP_ERR_INVALID, "too many node updates in one submission");
    // Same op list as last time and no key created / released / moved since: the descriptors on
    // the device are still the right ones (hot loop 1 re-evaluates one tree many times).
    const size_t in_bytes = sizeof(iqhip_node_op) * (size_t)nops;
    const std::vector<int> no_segs;
    const std::vector<int> &segs_in = explicit_segs ? *explicit_segs : no_segs;
    if (!len_ptrs && nops > 0 && e->last_plan_version == e->keymap_version && e->last_ops_in.size() == in_bytes &&
        memcmp(e->last_ops_in.data(), ops, in_bytes) == 0 && !e->uploaded_plan.empty() && e->last_segs == segs_in &&
        (!e->plan_uses_cherry || e->plan_cherry_model == e->model_version)) {
        *last_dst = e->last_plan_dst;
        return IQHIP_OK;
    }
    e->last_plan_version = 0;
    int rc = IQHIP_OK;
    if (e->staging_busy) {  // the previous submission may still be copying h_ops
        HIPCHK(hipEventSynchronize(e->staging_free));
        e->staging_busy = false;
    }
    int prev_dst = -1;
    const int B = e->block;
    e->plan_has_load = false;
    e->plan_units_have_load = false;
    // ---- staging.  A launch gives every 64/16-pattern tile one wave that walks the whole op list, so an
    // alignment with few tiles leaves SIMDs idle or unevenly loaded (a tile is an indivisible unit of
    // nops updates).  Independent subtrees of the plan are therefore cut out as "units" that run on
    // their own workgroups in a first launch (tiles x units waves), and only the ops above them
    // ("top") walk sequentially in a second launch.  order[p] = caller index of the op at position p.
    std::vector<int> order(nops), seg_of(nops, 0);
    std::vector<std::pair<int, int>> units;  // {begin, nops} in the new order, stage after stage
    std::vector<int> stage_units;            // units per stage (launch)
    int max_levels = 3;
    if (const char *ml = getenv("IQHIP_LEVELS")) max_levels = std::max(1, atoi(ml));
    int top_begin = 0;
    for (int k = 0; k < nops; k++) order[k] = k;
    if (explicit_segs) {
        // caller-defined independent segments (a batch of branch tasks): one set of workgroups each, no top stage
        std::unordered_set<uint64_t> dsts;
        for (int k = 0; k < nops; k++)
            if (!dsts.insert(ops[k].dst_key).second)
                return fail(IQHIP_ERR_INVALID, "batched node updates must write distinct vectors");
        int pos = 0;
        for (size_t s = 0; s < explicit_segs->size(); s++) {
            const int n = (*explicit_segs)[s];
            if (n <= 0) continue;
            units.push_back({pos, n});
            for (int q = 0; q < n; q++) seg_of[pos + q] = (int)units.size();
            pos += n;
        }
        if (pos != nops) return fail(IQHIP_ERR_INVALID, "segment sizes do not add up to the op count");
        stage_units.push_back((int)units.size());
        top_begin = nops;
    } else {
        int target = e->split_target;
        const int64_t simds = (int64_t)e->num_cus * 4;
        if (target < 0) {
            // auto: the matrix-core kernels (16-pattern tiles, long per-tile op lists) when there are fewer
            // than 6 tile-waves per SIMD; unit size so that the first launch has ~4 waves per SIMD.
            // Measured at the BASELINE shapes: protein 1.28 -> 1.10 ms, codon 0.64 -> 0.49 ms; the 4-state
            // kernel is store-bound and loses (0.161 -> 0.172..0.184 ms), so it stays unsplit.
            target = 0;
            if (e->mfma && e->ntiles < 6 * simds && nops >= 12)
                target = (int)std::min<int64_t>(nops / 2, std::max<int64_t>(3, ((int64_t)nops * e->ntiles + 4 * simds - 1) / (4 * simds)));
            // 4-state kernel: only while the whole alignment is at most one wave per SIMD, where a traversal
            // is a latency-bound chain (50 taxa GTR+G4: 5k patterns 0.082 -> 0.045 ms, 20k 0.086 -> 0.060 ms;
            // 60k patterns 0.109 -> 0.126 ms, so not there)
            if (!e->mfma && e->ntiles * e->lane_split <= simds && nops >= 12) target = std::max(6, nops / 4);
        }
        if (target > 0 && target < nops && nops >= 4) {
            std::unordered_map<uint64_t, int> prod;
            std::unordered_set<uint64_t> ext_in;
            std::vector<int> lc(nops, -1), rc(nops, -1);
            std::vector<char> consumed(nops, 0);
            bool safe = true;
            for (int k = 0; k < nops && safe; k++) {
                const iqhip_node_op &o = ops[k];
                auto child = [&](uint64_t key, int32_t leaf) -> int {
                    if (leaf >= 0) return -1;
                    auto it = prod.find(key);
                    if (it == prod.end()) { ext_in.insert(key); return -1; }
                    return it->second;
                };
                lc[k] = child(o.left_key, o.left_leaf);
                rc[k] = child(o.right_key, o.right_leaf);
                if (lc[k] >= 0) { if (consumed[lc[k]]) safe = false; consumed[lc[k]] = 1; }
                if (rc[k] >= 0) { if (consumed[rc[k]]) safe = false; consumed[rc[k]] = 1; }
                // re-ordering is only safe when no vector of the plan is both an outside input and a
                // destination (LM_PER_NODE buffer stealing) and nothing is written twice
                if (prod.count(o.dst_key)) safe = false;
                prod[o.dst_key] = k;
            }
            for (int k = 0; k < nops && safe; k++)
                if (ext_in.count(ops[k].dst_key)) safe = false;
            if (safe) {
                // Level by level: among the ops not yet placed, the maximal subtrees of 2..target ops (vectors of
                // earlier levels count as outside inputs) become the units of the next launch; what is left after
                // the last level walks sequentially.  Every level is a launch of tiles x units waves, so the
                // sequential tail -- where an alignment with slightly more tiles than SIMDs runs at half speed --
                // shrinks from "everything above the first cut" to a few ops.
                std::vector<char> placed(nops, 0);
                std::vector<int> new_order;
                new_order.reserve(nops);
                int nseg = 0, left = nops;
                const int min_top = std::max(2, std::min(target, 6));
                for (int level = 0; level < max_levels && left > min_top; level++) {
                    std::vector<int> rem;  // unplaced ops in post-order
                    for (int k = 0; k < nops; k++)
                        if (!placed[k]) rem.push_back(k);
                    const int R = (int)rem.size();
                    std::vector<int> pos_of(nops, -1), l2(R, -1), r2(R, -1), sz(R, 1);
                    std::vector<char> cont(R, 1), cons(R, 0);
                    for (int q = 0; q < R; q++) pos_of[rem[q]] = q;
                    for (int q = 0; q < R; q++) {
                        const int k = rem[q];
                        if (lc[k] >= 0 && !placed[lc[k]]) { l2[q] = pos_of[lc[k]]; cons[l2[q]] = 1; sz[q] += sz[l2[q]]; }
                        if (rc[k] >= 0 && !placed[rc[k]]) { r2[q] = pos_of[rc[k]]; cons[r2[q]] = 1; sz[q] += sz[r2[q]]; }
                        // post-order contiguity within the unplaced sequence: the subtree of q is exactly [q - sz + 1, q]
                        const int a = std::max(l2[q], r2[q]), b2 = std::min(l2[q], r2[q]);
                        bool c2 = true;
                        if (a >= 0) c2 = (a == q - 1) && cont[a];
                        if (b2 >= 0) c2 = c2 && (b2 == a - sz[a]) && cont[b2];
                        cont[q] = c2;
                    }
                    std::vector<int> stack;
                    for (int q = R - 1; q >= 0; q--)
                        if (!cons[q]) stack.push_back(q);
                    std::vector<std::pair<int, int>> found;  // {root position, size}
                    while (!stack.empty()) {
                        const int q = stack.back();
                        stack.pop_back();
                        if (sz[q] <= target && sz[q] >= 2 && cont[q]) {
                            found.push_back({q, sz[q]});
                        } else {
                            if (l2[q] >= 0) stack.push_back(l2[q]);
                            if (r2[q] >= 0) stack.push_back(r2[q]);
                        }
                    }
                    if (found.size() < 2) break;
                    std::stable_sort(found.begin(), found.end(),
                                     [](const std::pair<int, int> &x, const std::pair<int, int> &y2) { return x.second > y2.second; });
                    int in_stage = 0;
                    for (size_t u = 0; u < found.size(); u++) {
                        units.push_back({(int)new_order.size(), found[u].second});
                        nseg++;
                        in_stage++;
                        for (int q = found[u].first - found[u].second + 1; q <= found[u].first; q++) {
                            seg_of[new_order.size()] = nseg;
                            placed[rem[q]] = 1;
                            new_order.push_back(rem[q]);
                            left--;
                        }
                    }
                    stage_units.push_back(in_stage);
                }
                if (!units.empty()) {
                    top_begin = (int)new_order.size();
                    for (int k = 0; k < nops; k++)
                        if (!placed[k]) { seg_of[new_order.size()] = 0; new_order.push_back(k); }
                    order = new_order;
                }
            }
        }
    }
    if (getenv("IQHIP_DEBUG_PLAN")) {
        fprintf(stderr, "[iqhip] plan: %d ops, %zu stages of units (", nops, stage_units.size());
        size_t ui = 0;
        for (int n : stage_units) {
            for (int q = 0; q < n; q++) fprintf(stderr, "%d ", units[ui++].second);
            fprintf(stderr, "| ");
        }
        fprintf(stderr, ") top %d ops\n", nops - top_begin);
    }
    const int table_ints = 2 * (1 + (int)units.size());
    const int table_ops = (int)((table_ints * sizeof(int) + sizeof(DevOp) - 1) / sizeof(DevOp));
    // (room for the K2 table job list behind the segment table: at most two leaf children per op)
    const int jobs_ops_max = (int)((sizeof(TabJob) * (size_t)(2 * nops + 1) + sizeof(DevOp) - 1) / sizeof(DevOp));
    rc = ensure_plan_capacity(e, nops + kSentinels + table_ops + jobs_ops_max);
    if (rc) return rc;
    auto dummy_op = [&](DevOp &d) {
        memset(&d, 0, sizeof(d));
        d.dst = e->dummy.plh;
        d.dst_sc = e->dummy.sc;
        d.pf = d.ld = e->dummy.plh;
        d.pf_sc = d.ld_sc = e->dummy.sc;
        d.sl = d.sr = e->d_states;
        d.tabL = d.tabR = e->d_leaf_tab;
    };
    for (int k = 0; k < nops; k++) {
        const iqhip_node_op &o = ops[order[k]];
        DevOp &d = e->h_ops[k];
        dummy_op(d);
        d.out_row = order[k];
        d.no_scale = (o.flags & IQHIP_OP_NO_SCALE) ? 1 : (((o.flags & IQHIP_OP_SCALAR_RULE) || e->scalar_rule_all) ? 2 : 0);
        if (k > 0 && seg_of[k] != seg_of[k - 1]) prev_dst = -1;  // another workgroup: no register hand-over
        if (!(o.left_len >= 0.0) || !(o.right_len >= 0.0))
            return fail(IQHIP_ERR_INVALID, "negative or NaN branch length");
        const double *lp, *rp;
        const int16_t *lsc, *rsc;
        const uint8_t *lst, *rst;
        int32_t lkind, rkind;
        rc = resolve_child(e, o.left_key, o.left_leaf, prev_dst, &lp, &lsc, &lst, &lkind);
        if (rc) return rc;
        rc = resolve_child(e, o.right_key, o.right_leaf, prev_dst, &rp, &rsc, &rst, &rkind);
        if (rc) return rc;
        int didx;
        rc = slab_for_key(e, o.dst_key, true, &didx);
        if (rc) return rc;
        if ((lkind != CHILD_LEAF && lp == e->slabs[didx].plh) || (rkind != CHILD_LEAF && rp == e->slabs[didx].plh))
            return fail(IQHIP_ERR_INVALID, "node update writes onto one of its own children");
        if (lkind == CHILD_PREV && rkind == CHILD_PREV)
            return fail(IQHIP_ERR_INVALID, "node update uses the same vector for both children");
        double llen = o.left_len, rlen = o.right_len;
        const double *llen_p = len_ptrs ? len_ptrs[2 * order[k]] : nullptr, *rlen_p = len_ptrs ? len_ptrs[2 * order[k] + 1] : nullptr;
        d.dst = e->slabs[didx].plh;
        d.dst_sc = e->slabs[didx].sc;
        if (e->mfma && !e->mfma_pipelined) {
            // generic matrix-core kernel: both children are read from memory (pf = left, ld = right)
            if (lkind != CHILD_LEAF) { lkind = CHILD_LOAD; d.pf = lp; d.pf_sc = lsc; } else d.sl = lst;
            if (rkind != CHILD_LEAF) { rkind = CHILD_LOAD; d.ld = rp; d.ld_sc = rsc; } else d.sr = rst;
        } else {
            // canonical form (the Hadamard product commutes): left in {LEAF, PF}, right in
            // {LEAF, PREV}; the only other shape is (PF, LOAD): two memory children, neither of
            // them the previous result -- the kernel reads the second one synchronously.
            auto swap_children = [&]() {
                std::swap(lp, rp); std::swap(lsc, rsc); std::swap(lst, rst);
                std::swap(lkind, rkind); std::swap(llen, rlen); std::swap(llen_p, rlen_p);
            };
            if (lkind == CHILD_PREV) swap_children();                              // PREV goes right
            else if (lkind == CHILD_LEAF && rkind == CHILD_LOAD) swap_children();  // memory child goes left
            if (lkind == CHILD_LOAD) lkind = CHILD_PF;
            if (rkind == CHILD_LOAD) { if (seg_of[k]) e->plan_units_have_load = true; else e->plan_has_load = true; }  // (PF, LOAD)
            if (lkind == CHILD_PF) { d.pf = lp; d.pf_sc = lsc; d.real_mask |= 1; }
            if (rkind == CHILD_LOAD) { d.ld = rp; d.ld_sc = rsc; }
            if (lkind == CHILD_LEAF) d.sl = lst;
            if (rkind == CHILD_LEAF) d.sr = rst;
            if (e->ablate & 1) d.real_mask &= ~1;  // timing-only: never stream a child (results wrong)
        }
        d.left_kind = lkind;
        d.right_kind = rkind;
        d.left_len = llen;
        d.right_len = rlen;
        d.left_len_p = llen_p;
        d.right_len_p = rlen_p;
        prev_dst = didx;
    }
    // HOLD analysis (4-state kernel): a streamed left child produced by op j of this plan can stay
    // in registers until its join k if no op in (j, k) streams, loads or parks anything itself
    // (the usual case after heavier-first ordering: the other subtree is a short chain).
    const bool hold_regs = !e->mfma && !(e->ablate & 4) && (e->use_hold || e->lane_split != 1 || e->wg_size != 256);
    // (20 states, one wave per tile: the parking place is LDS; the launch reserves it when plan_nhold > 0)
    const bool hold_in_lds = e->mfma && e->mfma_pipelined && e->hold_lds && e->n == 20 && !e->cat_split;
    e->plan_nhold = 0;
    if (hold_regs || hold_in_lds) {
        std::unordered_map<const double *, int> producer;
        for (int k = 0; k < nops; k++) {
            DevOp &d = e->h_ops[k];
            if (d.left_kind == CHILD_PF) {
                auto it = producer.find(d.pf);
                if (it != producer.end()) {
                    const int j = it->second;
                    bool ok = !e->h_ops[j].push_hold && seg_of[j] == seg_of[k];
                    if (hold_in_lds && e->top_cs2 && seg_of[k] == 0) ok = false;   // (two waves per tile there: no parking place)
                    for (int q = j + 1; q < k && ok; q++) {
                        const DevOp &m = e->h_ops[q];
                        ok = m.left_kind != CHILD_PF && m.left_kind != CHILD_HOLD && m.right_kind != CHILD_LOAD &&
                             !m.push_hold;
                    }
                    if (ok) {
                        e->h_ops[j].push_hold = 1;
                        d.left_kind = CHILD_HOLD;
                        d.pf = e->dummy.plh;
                        d.pf_sc = e->dummy.sc;
                        d.real_mask &= ~1;
                        e->plan_nhold++;
                    }
                }
            }
            producer[d.dst] = k;
        }
    }
    if (getenv("IQHIP_DEBUG_PLAN")) {
        int npf = 0;
        for (int k = 0; k < nops; k++) npf += (e->h_ops[k].left_kind == CHILD_PF) + (e->h_ops[k].right_kind == CHILD_LOAD);
        fprintf(stderr, "[iqhip] plan: %d children read back from memory, %d parked\n", npf, e->plan_nhold);
    }
    // K2 tables of the leaf children (pipelined matrix-core kernels): slot = taxon, rebuilt by k_leaf_tables before
    // the traversal only where the pendant branch length (or the model) changed since the slot was last built
    e->plan_nleaf_tabs = 0;
    e->plan_tab_jobs.clear();
    e->plan_tab_dirty = 0;
    if (e->mfma && e->mfma_pipelined && e->leaf_tables) {
        const size_t per = leaf_table_doubles(e);
        struct Use { double len; const double *len_p; int slot; };
        std::unordered_map<int, std::vector<Use>> seen;  // taxon -> lengths used in this plan
        int noverflow = 0;
        std::vector<TabJob> dirty, clean;
        std::vector<std::pair<int, int>> uses;  // (op index, side) -> slot, resolved to pointers after (re)allocation
        std::vector<int> use_slot;
        const bool model_changed = e->tab_model_version != e->model_version;
        for (int k = 0; k < nops; k++) {
            DevOp &d = e->h_ops[k];
            for (int side = 0; side < 2; side++) {
                if ((side ? d.right_kind : d.left_kind) != CHILD_LEAF) continue;
                const uint8_t *row = side ? d.sr : d.sl;
                const int taxon = (int)((row - e->d_states) / e->nptn_pad);
                const double *len_p = side ? d.right_len_p : d.left_len_p;   // (sweeps: the length is on the device)
                const double len = len_p ? NAN : (side ? d.right_len : d.left_len);
                std::vector<Use> &u = seen[taxon];
                int slot = -1;
                for (const Use &x : u)
                    if (x.len_p == len_p && (len_p || x.len == len)) slot = x.slot;
                if (slot < 0) {
                    slot = u.empty() ? taxon : e->ntaxa + noverflow++;
                    u.push_back({len, len_p, slot});
                    TabJob j;
                    j.len = len;
                    j.len_p = len_p;
                    j._pad = 0.0;
                    j.tab = reinterpret_cast<double *>((size_t)slot);  // slot number for now
                    const bool cached = !len_p && slot < e->ntaxa && !model_changed && (size_t)slot < e->tab_len.size() &&
                                        e->tab_len[slot] == len;
                    (cached ? clean : dirty).push_back(j);
                }
                uses.push_back({k, side});
                use_slot.push_back(slot);
            }
        }
        const size_t need = (size_t)e->ntaxa + (size_t)noverflow;
        if (need > e->leaf_tab_slots) {
            const size_t slots = need + 16;
            if (e->planner) {
                e->d_leaf_tab = fake_alloc<double>(e, slots * per);
            } else {
                HIPCHK(hipStreamSynchronize(e->stream));
                if (e->d_leaf_tab) hipFree(e->d_leaf_tab);
                e->d_leaf_tab = nullptr;
                e->leaf_tab_slots = 0;
                HIPCHK(dmalloc(&e->d_leaf_tab, slots * per));
            }
            e->leaf_tab_slots = slots;
            e->uploaded_plan.clear();
            // a new buffer holds no tables: everything this plan uses is dirty
            dirty.insert(dirty.end(), clean.begin(), clean.end());
            clean.clear();
            e->tab_len.assign(slots, NAN);
        }
        if (e->tab_len.size() < e->leaf_tab_slots) e->tab_len.resize(e->leaf_tab_slots, NAN);
        if (model_changed) {  // tables of other plans are stale as well
            std::fill(e->tab_len.begin(), e->tab_len.end(), NAN);
            e->tab_model_version = e->model_version;
        }
        // non-leaf children point at slot 0: the kernels may request a row unconditionally (one step ahead)
        for (int k = 0; k < nops; k++) e->h_ops[k].tabL = e->h_ops[k].tabR = e->d_leaf_tab;
        for (size_t q = 0; q < uses.size(); q++) {
            DevOp &d = e->h_ops[uses[q].first];
            (uses[q].second ? d.tabR : d.tabL) = e->d_leaf_tab + (size_t)use_slot[q] * per;
        }
        for (std::vector<TabJob> *v : {&dirty, &clean})
            for (TabJob &j : *v) {
                const size_t slot = (size_t)j.tab;
                j.tab = e->d_leaf_tab + slot * per;
                e->tab_len[slot] = slot < (size_t)e->ntaxa ? j.len : NAN;  // overflow slots are never reused
                e->plan_tab_jobs.push_back(j);
            }
        e->plan_tab_dirty = (int)dirty.size();
        e->plan_nleaf_tabs = (int)e->plan_tab_jobs.size();
    }
    // cherry tables: an op whose two children are leaves reads its result out of the table of its pair of taxa
    e->plan_cherry_jobs.clear();
    e->plan_uses_cherry = false;
    e->plan_cherry_model = e->model_version;
    // (planning-only engine: the tables are never built, their slots are fake addresses like every other buffer)
    const bool have_pair = e->planner ? e->cherry_s2 > 0 : (e->pair && e->cherry_model_synced);
    if (cherry_candidate(e) && e->mfma_pipelined && have_pair && nops >= 8) {
        const size_t per = (size_t)e->cherry_npairs * B;
        const size_t want = (size_t)2 * e->ntaxa + 16;
        if (e->cherry_cap < want && e->planner) {
            e->d_cherry_tab = fake_alloc<double>(e, want * per);
            e->cherry_cap = want;
        } else if (e->cherry_cap < want) {
            HIPCHK(hipStreamSynchronize(e->stream));
            if (e->d_cherry_tab) hipFree(e->d_cherry_tab);
            e->d_cherry_tab = nullptr;
            e->cherry_cap = 0;
            HIPCHK(dmalloc(&e->d_cherry_tab, want * per));
            e->cherry_cap = want;
            e->cherry_slot_of.clear();
            e->cherry_slots.clear();
        }
        // room for every cherry a plan can hold; a search that has walked through more pairs than that starts over
        if (e->cherry_slots.size() + (size_t)e->ntaxa / 2 + 1 > e->cherry_cap) {
            e->cherry_slot_of.clear();
            e->cherry_slots.clear();
        }
        const uint64_t stamp = ++e->cherry_stamp;
        for (int k = 0; k < nops; k++) {
            DevOp &d = e->h_ops[k];
            if (d.left_kind != CHILD_LEAF || d.right_kind != CHILD_LEAF || d.left_len_p || d.right_len_p) continue;
            // (the top stage's two-waves-per-tile / row-split kernels compute their cherries)
            if ((e->top_cs2 || (e->n == 64 && e->mixed_top)) && seg_of[k] == 0) continue;
            const uint64_t tl = (uint64_t)((d.sl - e->d_states) / e->nptn_pad), tr = (uint64_t)((d.sr - e->d_states) / e->nptn_pad);
            const uint64_t key = (tl << 32) | tr;
            auto it = e->cherry_slot_of.find(key);
            int slot;
            if (it == e->cherry_slot_of.end()) {
                if (e->cherry_slots.size() >= e->cherry_cap) continue;
                slot = (int)e->cherry_slots.size();
                e->cherry_slots.emplace_back();
                e->cherry_slot_of[key] = slot;
            } else {
                slot = it->second;
            }
            iqhip_engine::CherrySlot &cs = e->cherry_slots[slot];
            const bool same = cs.len_l == d.left_len && cs.len_r == d.right_len;
            if (cs.stamp == stamp && !same) continue;   // (the same pair with other lengths in one plan: computed the ordinary way)
            if (!same || cs.model_version != e->model_version) {
                cs.len_l = d.left_len;
                cs.len_r = d.right_len;
                cs.model_version = 0;   // until built (submit_traverse)
                if (cs.stamp != stamp) e->plan_cherry_jobs.push_back(slot);
            }
            cs.stamp = stamp;
            d.cherry = e->d_cherry_tab + (size_t)slot * per;
            e->plan_uses_cherry = true;
        }
    }
    for (int q = 0; q < kSentinels; q++) dummy_op(e->h_ops[nops + q]);  // targets of the look-ahead requests
    *last_dst = prev_dst;
    // LDS layout of the per-(op, child) regions, cut into chunks that fit the budget
    {
        int budget;
        if (e->mfma) {
            const int MT = (e->n + 15) / 16, KS = e->n / 4;
            int fixed = (e->row_split && e->mfma_pipelined) ? (e->state_unknown + 1) * e->n + 4 * 16 * 64 + 128
                              : (e->mfma_pipelined ? mfma2_fixed_lds_doubles(e->n) : 2 * MT * KS * 64) +
                                    (e->state_unknown + 1 - e->n) * e->n;
            // (64 states: a launch may mix both roles, k_traverse_mfma_top64)
            if (e->mfma_pipelined && e->n == 64) fixed = std::max(fixed, (e->state_unknown + 1) * e->n + 4 * 16 * 64 + 128);
            // two workgroups per CU (160 KB LDS): <= 78 KB each, images included (a third workgroup
            // for the 20-state kernel was measured: no gain, more chunks); IQHIP_MFMA_LDS_KB overrides
            int total_kb = e->n == 20 ? 75 : 78;   // (20 states: + 2.6 KB of static arrays per workgroup, the fill's descriptor copies)
            if (const char *kb = getenv("IQHIP_MFMA_LDS_KB")) total_kb = atoi(kb);
            if (e->plan_nhold > 0) fixed += 4 * 16 * B;   // the waves' parking places (CHILD_HOLD in LDS)
            budget = (total_kb * 1024) / 8 - fixed;
        } else {
            budget = (e->lds_budget_bytes / 8) - 128 - B;
        }
        const bool tables_in_lds = e->mfma && e->mfma_pipelined && e->n == 20 && e->plan_nleaf_tabs > 0;
        int chunk_start = 0, used = e->mfma ? 0 : e->wg_size / 8, regs = 0, max_used = 0, slots = 1, max_slots = 1;
        for (int k = 0; k < nops; k++) {
            DevOp &d = e->h_ops[k];
            // a LEAF child's region: 4 states -- exponentials + the 5-row K2 table; 20 states with leaf tables -- the
            // child's whole K2 table [ncat][STATE_UNKNOWN][n], copied from the table buffer when the chunk is filled
            const int leaf_sz = !e->mfma ? 6 * B : (tables_in_lds ? (int)leaf_table_doubles(e) : B);
            const int szl = d.left_kind == CHILD_LEAF ? leaf_sz : B;
            const int szr = d.right_kind == CHILD_LEAF ? leaf_sz : B;
            // 4-state path: each leaf child also stages one state byte per thread in LDS
            const int nleaf = (d.left_kind == CHILD_LEAF) + (d.right_kind == CHILD_LEAF);
            const int need = szl + szr + (e->mfma ? 0 : nleaf * e->wg_size / 8);
            if (need > budget) return fail(IQHIP_ERR_UNSUPPORTED, "nstates*ncat too large for the LDS plan regions");
            if ((used + need > budget || seg_of[k] != seg_of[k - (k > 0)]) && k > chunk_start) {
                e->h_ops[chunk_start].chunk_nops = k - chunk_start;
                chunk_start = k;
                used = e->mfma ? 0 : e->wg_size / 8;  // slot 0
                regs = 0;
                slots = 1;
            }
            d.lds_left = regs;
            d.lds_right = regs + szl;
            regs += szl + szr;
            used += need;
            if (regs > max_used) max_used = regs;
            d.sl_slot = d.left_kind == CHILD_LEAF ? slots++ : 0;
            d.sr_slot = d.right_kind == CHILD_LEAF ? slots++ : 0;
            if (slots > max_slots) max_slots = slots;
        }
        e->plan_state_slots = max_slots;
        if (nops > 0) e->h_ops[chunk_start].chunk_nops = nops - chunk_start;
        e->plan_lds_doubles = max_used;
    }
    // the descriptors of a repeated plan (model-parameter optimisation re-evaluates the same
    // tree) are already on the device: skip the upload, never the computation
    {
        int *tab = reinterpret_cast<int *>(e->h_ops + nops + kSentinels);
        memset(tab, 0, sizeof(DevOp) * (size_t)table_ops);
        tab[0] = top_begin;
        tab[1] = nops - top_begin;
        for (size_t u = 0; u < units.size(); u++) { tab[2 + 2 * u] = units[u].first; tab[3 + 2 * u] = units[u].second; }
        e->plan_nunits = (int)units.size();
        e->plan_stage_units = stage_units;
        e->plan_top_nops = nops - top_begin;
        e->plan_table_off = nops + kSentinels;
    }
    e->plan_jobs_off = nops + kSentinels + table_ops;
    const int jobs_ops = (int)((sizeof(TabJob) * e->plan_tab_jobs.size() + sizeof(DevOp) - 1) / sizeof(DevOp));
    if (jobs_ops > 0) {
        memset(e->h_ops + e->plan_jobs_off, 0, sizeof(DevOp) * (size_t)jobs_ops);
        memcpy(e->h_ops + e->plan_jobs_off, e->plan_tab_jobs.data(), sizeof(TabJob) * e->plan_tab_jobs.size());
    }
    const size_t nbytes = sizeof(DevOp) * (size_t)(nops + kSentinels + table_ops + jobs_ops);
    e->last_ops_in.assign((const char *)ops, (const char *)ops + in_bytes);
    e->last_segs = segs_in;
    e->last_plan_version = len_ptrs ? 0 : e->keymap_version;  // (slabs created while building are included; a sweep step's plan is never re-used)
    e->last_plan_dst = prev_dst;
    // a small plan of the 4-state kernel rides in the kernel arguments (launch_traverse4 copies it out of h_ops)
    // (matrix-core path: the pipelined 20-state kernels without leaf tables -- tables come with a job list in the buffer)
    const bool small_kernel = !e->mfma || (e->mfma_pipelined && e->n == 20 && !e->leaf_tables && e->plan_nleaf_tabs == 0);
    e->plan_small = e->small_plans && small_kernel && !explicit_segs && units.empty() && nops > 0 && nops + kSentinels <= kSmallPlanOps;
    e->plan_small_nops = nops;
    if (e->planner) {   // negative tests: break one descriptor the way round 2's fault did
        const char *br = getenv("IQHIP_DEBUG_BREAK_PLAN");
        if (br && nops > 0) {
            if (!strcmp(br, "tab")) e->h_ops[nops - 1].tabL = nullptr;
            else if (!strcmp(br, "sentinel")) e->h_ops[nops + kSentinels - 1].pf = nullptr;
            else if (!strcmp(br, "states")) e->h_ops[0].sr = nullptr;
            else if (!strcmp(br, "cherry")) {
                for (int k = 0; k < nops; k++)
                    if (e->h_ops[k].cherry) { e->h_ops[k].cherry += 8; break; }   // (inside the buffer, not on a table)
            }
        }
    }
    if (e->check_plans && !e->ablate) {
        rc = check_plan(e, nops, kSentinels);
        if (rc) { e->last_plan_version = 0; e->uploaded_plan.clear(); return rc; }
    }
    if (e->planner) return IQHIP_OK;   // (nothing to upload to)
    if (e->plan_small) {
        e->uploaded_plan.assign((const char *)e->h_ops, (const char *)e->h_ops + nbytes);   // (what d_ops would hold)
        return IQHIP_OK;
    }
    if (e->uploaded_plan.size() == nbytes && memcmp(e->uploaded_plan.data(), e->h_ops, nbytes) == 0)
        return IQHIP_OK;
    if (e->plan_arena_on && e->plan_arena_used + nbytes <= e->plan_arena_cap) {
        // a sweep enqueues many plans before the device has run the first: each upload goes out of a slice of its own
        // of a pinned arena, so that re-using h_ops for the next plan never has to wait for the device (that wait -- the
        // staging event below -- made the host fall in step with the device eight times per 97-branch protein sweep)
        char *slice = e->h_plan_arena + e->plan_arena_used;
        memcpy(slice, e->h_ops, nbytes);
        e->plan_arena_used += (nbytes + 255) / 256 * 256;
        HIPCHK(hipMemcpyAsync(e->d_ops, slice, nbytes, hipMemcpyHostToDevice, e->stream));
        e->uploaded_plan.assign((const char *)e->h_ops, (const char *)e->h_ops + nbytes);
        return IQHIP_OK;
    }
    HIPCHK(hipMemcpyAsync(e->d_ops, e->h_ops, nbytes, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipEventRecord(e->staging_free, e->stream));
    e->staging_busy = true;
    e->uploaded_plan.assign((const char *)e->h_ops, (const char *)e->h_ops + nbytes);
    return IQHIP_OK;
}

static int build_branch(iqhip_engine *e, iqhip_branch_end a, iqhip_branch_end b, double len,
                        int prev_dst, DevBranch *br) {
    if (!(len >= 0.0)) return fail(IQHIP_ERR_INVALID, "negative or NaN branch length");
    if (a.leaf >= 0 && b.leaf >= 0)
        return fail(IQHIP_ERR_INVALID, "branch with two leaf ends (2-taxon tree) is not supported");
    if (b.leaf >= 0) std::swap(a, b);  // the reference puts the leaf on the `dad` side (:739-746)
    const uint8_t *st_unused;
    int rc = resolve_child(e, a.key, a.leaf, prev_dst, &br->a, &br->a_sc, &br->a_states, &br->a_kind);
    if (rc) return rc;
    rc = resolve_child(e, b.key, b.leaf, prev_dst, &br->b, &br->b_sc, &st_unused, &br->b_kind);
    if (rc) return rc;
    br->len = len;
    return IQHIP_OK;
}

static void timing_begin(iqhip_engine *e) {
    if (!e->timing) return;
    if (e->tev_used == e->tev.size()) {
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        e->tev.emplace_back(a, b);
    }
    hipEventRecord(e->tev[e->tev_used].first, e->stream);
}
static void timing_end(iqhip_engine *e) {
    if (!e->timing) return;
    hipEventRecord(e->tev[e->tev_used].second, e->stream);
    e->tev_used++;
}

static double g_dbg_build_us = 0.0;   // IQHIP_DEBUG_SWEEP: host time spent in build_plan
static int submit_traverse(iqhip_engine *e, const iqhip_node_op *ops, int nops, bool has_root, iqhip_branch_end a,
                           iqhip_branch_end b, double len, bool skip_reduce = false, const std::vector<int> *explicit_segs = nullptr,
                           const double *const *len_ptrs = nullptr);

// the tables the current plan needs: one node update per table on the pair engine (independent segments of one
// submission, same stream), then the move into register order
static int build_cherry_tables(iqhip_engine *e) {
    iqhip_engine *p = e->pair;
    const int n = (int)e->plan_cherry_jobs.size();
    std::vector<iqhip_node_op> ops((size_t)n);
    const std::vector<int> segs((size_t)n, 1);
    for (int i = 0; i < n; i++) {
        const iqhip_engine::CherrySlot &cs = e->cherry_slots[e->plan_cherry_jobs[i]];
        iqhip_node_op &o = ops[i];
        memset(&o, 0, sizeof o);
        o.dst_key = (uint64_t)e->plan_cherry_jobs[i] + 1;
        o.left_leaf = 0;
        o.right_leaf = 1;
        o.left_len = cs.len_l;
        o.right_len = cs.len_r;
    }
    iqhip_branch_end none = {0, -1, 0};
    int rc = submit_traverse(p, ops.data(), n, false, none, none, 0.0, /*skip_reduce=*/true, &segs, nullptr);
    if (rc) return rc;
    std::vector<const double *> src((size_t)n);
    std::vector<double *> dst((size_t)n);
    const size_t per = (size_t)e->cherry_npairs * e->block;
    for (int i = 0; i < n; i++) {
        int idx;
        rc = slab_for_key(p, ops[i].dst_key, false, &idx);
        if (rc) return rc;
        src[i] = p->slabs[idx].plh;
        dst[i] = e->d_cherry_tab + (size_t)e->plan_cherry_jobs[i] * per;
    }
    HIPCHK(launch_cherry_transpose(e, src.data(), dst.data(), n, e->cherry_npairs));
    for (int i = 0; i < n; i++) e->cherry_slots[e->plan_cherry_jobs[i]].model_version = e->model_version;
    e->cherry_built_total += n;
    e->plan_cherry_jobs.clear();
    return IQHIP_OK;
}

// enqueue: plan upload, K1, fused traversal (+ optional root lnL), fixed-order reduction
static int submit_traverse(iqhip_engine *e, const iqhip_node_op *ops, int nops, bool has_root,
                           iqhip_branch_end a, iqhip_branch_end b, double len, bool skip_reduce,
                           const std::vector<int> *explicit_segs, const double *const *len_ptrs) {
    int rc = check_ready(e);
    if (rc) return rc;
    if (nops < 0 || (nops > 0 && !ops)) return fail(IQHIP_ERR_INVALID, "bad ops array");
    int last_dst = -1;
    static const bool dbg_t = getenv("IQHIP_DEBUG_SWEEP") != nullptr;
    timespec q0, q1;
    if (dbg_t) clock_gettime(CLOCK_MONOTONIC, &q0);
    rc = build_plan(e, ops, nops, &last_dst, explicit_segs, len_ptrs);
    if (rc) return rc;
    if (dbg_t) { clock_gettime(CLOCK_MONOTONIC, &q1); g_dbg_build_us += (q1.tv_sec - q0.tv_sec) * 1e6 + (q1.tv_nsec - q0.tv_nsec) * 1e-3; }
    DevBranch br;
    if (has_root) {
        rc = build_branch(e, a, b, len, last_dst, &br);
        if (rc) return rc;
    }
    rc = ensure_slab_rows(e, 2 + nops);
    if (rc) return rc;
    const int nwaves = (int)e->ntiles * e->lane_split;  // columns of the wave-partial slab
    if (e->plan_nleaf_tabs > 0) {
        // tables whose branch length changed since they were built -- all of the plan's after a model change
        // (a cached plan skipped build_plan, which is where a model change is normally noticed)
        int njobs = e->plan_tab_dirty;
        if (e->tab_model_version != e->model_version) {
            std::fill(e->tab_len.begin(), e->tab_len.end(), NAN);
            for (const TabJob &j : e->plan_tab_jobs) {
                const size_t slot = (size_t)(j.tab - e->d_leaf_tab) / leaf_table_doubles(e);
                if (slot < (size_t)e->ntaxa) e->tab_len[slot] = j.len;
            }
            e->tab_model_version = e->model_version;
            njobs = e->plan_nleaf_tabs;
        }
        if (njobs > 0)
            HIPCHK(launch_leaf_tables(e, reinterpret_cast<const TabJob *>(e->d_ops + e->plan_jobs_off), njobs));
        e->plan_tab_dirty = 0;  // built; the same (cached) plan needs nothing until a length or the model changes
    }
    if (!e->plan_cherry_jobs.empty()) {
        rc = build_cherry_tables(e);
        if (rc) return rc;
    }
    if (e->plan_uses_cherry)
        for (int k = 0; k < nops; k++) e->cherry_ops_total += e->h_ops[k].cherry != nullptr;
    timing_begin(e);
    const int *table = reinterpret_cast<const int *>(e->d_ops + e->plan_table_off);
    {   // the stages of independent subtrees, level by level: one launch each, one set of workgroups per unit
        int off = 2;
        for (int n : e->plan_stage_units) {
            if (e->mfma) HIPCHK(launch_traverse_mfma(e, table + off, n, nwaves));
            else HIPCHK(launch_traverse4(e, table + off, n, e->plan_units_have_load, nullptr, nwaves));
            off += 2 * n;
        }
    }
    const bool empty_top = e->plan_nunits > 0 && !has_root && e->plan_top_nops == 0;  // explicit segments only
    // the submission's last kernel sums the wave partials itself (FoldArgs) where it can: the 4-state traversal
    // (its top-stage launch) and the matrix-core path's root-branch kernel; otherwise a k_reduce launch follows
    const bool fold4 = e->fold_reduce && !e->mfma && !empty_top && !skip_reduce && e->wg_size == 256;
    const bool foldm = e->fold_reduce && e->mfma && has_root && e->n_unobs == 0;
    if (empty_top) {
    } else if (e->mfma) HIPCHK(launch_traverse_mfma(e, table, nops > 0 ? 1 : 0, nwaves, /*top_stage=*/true));
    else HIPCHK(launch_traverse4(e, table, 1, e->plan_has_load, has_root ? &br : nullptr, nwaves, fold4 ? nops : -1));
    timing_end(e);
    if (e->timing) e->tev_launches += (int)e->plan_stage_units.size() + (empty_top ? 0 : 1);
    if (e->mfma && has_root) HIPCHK(launch_stream_mfma(e, 0, &br, br.len, nwaves, nullptr, foldm ? nops : -1));
    if (fold4 || foldm) {
    } else if (has_root) HIPCHK(launch_reduce(e, 0, 2 + nops, nwaves));
    else if (!skip_reduce) HIPCHK(launch_reduce(e, 2, nops, nwaves));
    e->last_nops = nops;
    e->last_has_root = has_root;
    e->last_root_loads_b = has_root && br.b_kind != CHILD_PREV;
    return IQHIP_OK;
}

static int read_result(iqhip_engine *e, int ndoubles) {
    if (e->d_result != e->d_result_own)  // caller-bound device buffer
        HIPCHK(hipMemcpyAsync(e->h_result, e->d_result, sizeof(double) * ndoubles, hipMemcpyDeviceToHost,
                              e->stream));
    else if (e->poll_pending) {
        // the last kernel of the submission is a k_reduce that publishes a sequence number in mapped host memory:
        // spinning on it sees the result a few microseconds before a stream synchronisation returns.  Bounded: long
        // kernels fall through to the ordinary wait.
        e->poll_pending = false;
        const unsigned long long want = e->result_seq;
        for (int spin = 0; spin < 200000; spin++) {
            if (*e->h_done == want) {
                std::atomic_thread_fence(std::memory_order_acquire);
                e->staging_busy = false;  // (in-order stream: the plan upload finished long before k_reduce)
                if (e->folded_rows >= 0) {   // folded reduction: unflagged sum_scale rows were not written (fold_tail)
                    if (e->h_result[2 + e->folded_rows] == 0.0)
                        for (int k = 0; k < e->folded_rows; k++) e->h_result[2 + k] = 0.0;
                    e->folded_rows = -1;
                }
                return IQHIP_OK;
            }
            __builtin_ia32_pause();
        }
    }
    e->poll_pending = false;
    HIPCHK(hipStreamSynchronize(e->stream));
    e->staging_busy = false;
    if (e->folded_rows >= 0) {
        if (e->h_result[2 + e->folded_rows] == 0.0)
            for (int k = 0; k < e->folded_rows; k++) e->h_result[2 + k] = 0.0;
        e->folded_rows = -1;
    }
    return IQHIP_OK;
}

// the NaN/Inf repair of phylokernel.h:848-866 / :1100-1122, done on the (rare) slow path
static int repair_lnl(iqhip_engine *e, double *lnl) {
    std::vector<double> plh((size_t)e->nptn_pad), freq((size_t)e->nptn_pad);
    HIPCHK(hipMemcpy(plh.data(), e->d_pattern_lh, sizeof(double) * plh.size(), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(freq.data(), e->d_freq, sizeof(double) * freq.size(), hipMemcpyDeviceToHost));
    double s = 0.0;
    for (int64_t p = 0; p < e->nptn; p++) {
        if (isnan(plh[p]) || isinf(plh[p])) plh[p] = kLogScalingThreshold * 4;
        s += plh[p] * freq[p];
    }
    HIPCHK(hipMemcpy(e->d_pattern_lh, plh.data(), sizeof(double) * plh.size(), hipMemcpyHostToDevice));
    *lnl = s;
    return IQHIP_OK;
}

// +ASC finalisation of a lnL evaluation (phylokernel.h:1009-1016, 1183-1186): result[1] holds
// prob_const; tree_lh -= nsites*log(1-prob_const), _pattern_lh[observed] -= log(1-prob_const)
static int asc_finish_lnl(iqhip_engine *e, double *lnl) {
    e->pattern_lh_shift = 0.0;
    if (!e->asc_active) return IQHIP_OK;
    const double pc = e->h_result[1];
    if (!(pc < 1.0 && pc >= 0.0))
        return fail(IQHIP_ERR_INVALID, "+ASC: prob_const outside [0,1) (the reference asserts here)");
    const double lp = log(1.0 - pc);
    e->pattern_lh_shift = lp;
    *lnl -= e->asc_nsites * lp;
    return IQHIP_OK;
}

namespace iqhip {
int eng_read_result(iqhip_engine *e, int ndoubles) { return read_result(e, ndoubles); }
int eng_submit_updates(iqhip_engine *e, const iqhip_node_op *ops, int nops, const std::vector<int> *segs) {
    iqhip_branch_end none = {0, -1, 0};
    return submit_traverse(e, ops, nops, false, none, none, 0.0, /*skip_reduce=*/false, segs);
}
int eng_repair_lnl(iqhip_engine *e, double *lnl) { return repair_lnl(e, lnl); }
}  // namespace iqhip

// Sharded engines (comm.hip): a non-finite lnL is repaired rank by rank (each rank fixes its own _pattern_lh and
// re-sums its share), then the shares are all-reduced again.  Every rank sees the same all-reduced value, so every
// rank takes this branch together.
static int repair_lnl_comm(iqhip_engine *e, double *lnl) {
    int rc = repair_lnl(e, lnl);
    if (rc || !e->comm) return rc;
    HIPCHK(hipMemcpyAsync(e->d_result, lnl, sizeof(double), hipMemcpyHostToDevice, e->stream));
    rc = comm_allreduce(e, 1);
    if (rc) return rc;
    rc = read_result(e, 1);
    if (rc) return rc;
    *lnl = e->h_result[0];
    return IQHIP_OK;
}

extern "C" int iqhip_update_partials(iqhip_engine *e, const iqhip_node_op *ops, int nops,
                                     double *sum_scale) {
    iqhip_branch_end none = {0, -1, 0};
    if (e && !e->shards.empty()) return sharded::traverse(e, ops, nops, false, none, none, 0.0, sum_scale, nullptr);
    int rc = submit_traverse(e, ops, nops, false, none, none, 0.0);
    if (rc) return rc;
    rc = comm_allreduce(e, 2 + nops);
    if (rc) return rc;
    rc = read_result(e, 2 + nops);
    if (rc) return rc;
    if (sum_scale)
        for (int k = 0; k < nops; k++) sum_scale[k] = e->h_result[2 + k];
    return IQHIP_OK;
}

extern "C" int iqhip_traverse_lnl(iqhip_engine *e, const iqhip_node_op *ops, int nops,
                                  iqhip_branch_end a, iqhip_branch_end b, double len,
                                  double *sum_scale, double *lnl) {
    if (e && !e->shards.empty()) return sharded::traverse(e, ops, nops, true, a, b, len, sum_scale, lnl);
    int rc = submit_traverse(e, ops, nops, true, a, b, len);
    if (rc) return rc;
    rc = comm_allreduce(e, 2 + nops);
    if (rc) return rc;
    rc = read_result(e, 2 + nops);
    if (rc) return rc;
    if (sum_scale)
        for (int k = 0; k < nops; k++) sum_scale[k] = e->h_result[2 + k];
    double v = e->h_result[0];
    if (isnan(v) || isinf(v)) {
        rc = repair_lnl_comm(e, &v);
        if (rc) return rc;
    }
    rc = asc_finish_lnl(e, &v);
    if (rc) return rc;
    if (lnl) *lnl = v;
    return IQHIP_OK;
}

extern "C" int iqhip_branch_lnl(iqhip_engine *e, iqhip_branch_end a, iqhip_branch_end b, double len,
                                double *lnl) {
    return iqhip_traverse_lnl(e, nullptr, 0, a, b, len, nullptr, lnl);
}

extern "C" int iqhip_traverse_lnl_async(iqhip_engine *e, const iqhip_node_op *ops, int nops,
                                        iqhip_branch_end a, iqhip_branch_end b, double len) {
    if (e && !e->shards.empty())
        return fail(IQHIP_ERR_UNSUPPORTED, "a sharded engine reduces its results itself: use the synchronous calls");
    // (+ASC: result[1] then holds this engine's share of prob_const; the caller owns the correction, phylokernel.h:1009-1016)
    return submit_traverse(e, ops, nops, true, a, b, len);
}

extern "C" int iqhip_update_partials_async(iqhip_engine *e, const iqhip_node_op *ops, int nops) {
    if (e && !e->shards.empty())
        return fail(IQHIP_ERR_UNSUPPORTED, "a sharded engine reduces its results itself: use the synchronous calls");
    iqhip_branch_end none = {0, -1, 0};
    return submit_traverse(e, ops, nops, false, none, none, 0.0);
}

extern "C" int iqhip_compute_theta(iqhip_engine *e, iqhip_branch_end a, iqhip_branch_end b) {
    if (e && !e->shards.empty()) return sharded::compute_theta(e, a, b);
    int rc = check_ready(e);
    if (rc) return rc;
    DevBranch br;
    rc = build_branch(e, a, b, 0.0, -1, &br);
    if (rc) return rc;
    if (e->mfma) HIPCHK(launch_stream_mfma(e, 1, &br, 0.0, (int)e->ntiles));
    else HIPCHK(launch_theta4(e, br));
    e->theta_valid = true;
    e->theta_a_sc = br.a_sc;
    e->theta_b_sc = br.b_sc;
    return IQHIP_OK;
}

extern "C" int iqhip_derv_async(iqhip_engine *e, double len) {
    if (e && !e->shards.empty())
        return fail(IQHIP_ERR_UNSUPPORTED, "a sharded engine reduces its results itself: use the synchronous calls");
    int rc = check_ready(e);
    if (rc) return rc;
    if (!e->theta_valid) return fail(IQHIP_ERR_INVALID, "iqhip_derv: theta not computed");
    if (!(len >= 0.0)) return fail(IQHIP_ERR_INVALID, "negative or NaN branch length");
    const int nrows = e->n_unobs > 0 ? 5 : 2;  // +ASC: prob_const, df_const, ddf_const as well
    if (e->asc_active && e->n_unobs == 0) {   // a shard without unobserved patterns contributes zeros to those three sums
        if (e->d_result == e->d_result_own) e->h_result[2] = e->h_result[3] = e->h_result[4] = 0.0;
        else HIPCHK(hipMemsetAsync(e->d_result + 2, 0, 3 * sizeof(double), e->stream));
    }
    rc = ensure_slab_rows(e, nrows);
    if (rc) return rc;
    const int nwaves = (int)e->ntiles;
    if (e->mfma) HIPCHK(launch_stream_mfma(e, 2, nullptr, len, nwaves));
    else HIPCHK(launch_derv4(e, len, nwaves));
    HIPCHK(launch_reduce(e, 0, nrows, nwaves));
    return IQHIP_OK;
}

extern "C" int iqhip_derv(iqhip_engine *e, double len, double *df, double *ddf) {
    if (e && !e->shards.empty()) return sharded::derv(e, len, df, ddf);
    int rc = iqhip_derv_async(e, len);
    if (rc) return rc;
    rc = comm_allreduce(e, e->asc_active ? 5 : 2);
    if (rc) return rc;
    rc = read_result(e, e->asc_active ? 5 : 2);
    if (rc) return rc;
    double a = e->h_result[0], b = e->h_result[1];
    if (isnan(a) || isinf(a)) { a = 0.0; b = 0.0; }  // phylokernel.h:647-651
    if (e->asc_active) {  // phylokernel.h:719-724
        const double prob_const = 1.0 - e->h_result[2];
        const double df_frac = e->h_result[3] / prob_const, ddf_frac = e->h_result[4] / prob_const;
        a += e->asc_nsites * df_frac;
        b += e->asc_nsites * (ddf_frac + df_frac * df_frac);
    }
    if (df) *df = a;
    if (ddf) *ddf = b;
    return IQHIP_OK;
}

// ---------------------------------------------------------------------------------------
// Newton as a chain of enqueued steps (kernels_newton.hip, "state machine" kernels): the form a sharded engine
// uses -- every derivative evaluation is followed by an in-stream all-reduce of {df, ddf}, so the loop cannot live
// in one kernel -- and the fallback when k_newton's grid barrier cannot be trusted.  Steps are enqueued in chunks;
// the host reads the 128-byte state once per chunk; steps enqueued after convergence do nothing.
// ---------------------------------------------------------------------------------------
namespace iqhip {
int newton_state_alloc(iqhip_engine *e) {
    if (e->d_nstate) return IQHIP_OK;
    if (hipMalloc((void **)&e->d_nstate, sizeof(NewtonState)) != hipSuccess ||
        hipHostMalloc((void **)&e->h_nstate, sizeof(NewtonState)) != hipSuccess)
        return set_error(IQHIP_ERR_NOMEM, "Newton state");
    return IQHIP_OK;
}

// enqueue `nsteps` evaluations + updates on the engine's stream (with the engine's own all-reduce in between)
int newton_chain_enqueue(iqhip_engine *e, int nsteps) {
    const int nwaves = (int)e->ntiles;
    for (int k = 0; k < nsteps; k++) {
        const int rows = e->asc_active ? 5 : 2;   // (+ASC: prob_const, df_const, ddf_const ride along)
        if (e->asc_active && e->n_unobs == 0 && hipMemsetAsync(e->d_result + 2, 0, 3 * sizeof(double), e->stream) != hipSuccess)
            return set_error(IQHIP_ERR_HIP, "Newton chain: memset failed");
        if (launch_derv_at_state(e, nwaves) != hipSuccess || launch_reduce(e, 0, e->n_unobs > 0 ? 5 : 2, nwaves) != hipSuccess)
            return set_error(IQHIP_ERR_HIP, "Newton chain: launch failed");
        int rc = comm_allreduce(e, rows);
        if (rc) return rc;
        if (launch_newton_state_update(e) != hipSuccess) return set_error(IQHIP_ERR_HIP, "Newton chain: launch failed");
    }
    return IQHIP_OK;
}

// the same pieces one at a time, for the single-process front, which interleaves its shards' steps with one grouped
// all-reduce per step
int eng_newton_begin(iqhip_engine *e, double xguess, double x1, double x2, double xacc, int max_steps) {
    int rc = check_ready(e);
    if (rc) return rc;
    if (!e->theta_valid) return set_error(IQHIP_ERR_INVALID, "Newton: theta not computed");
    rc = newton_state_alloc(e);
    if (rc) return rc;
    rc = ensure_slab_rows(e, 5);
    if (rc) return rc;
    if (launch_newton_state_init(e, xguess, x1, x2, xacc, max_steps) != hipSuccess)
        return set_error(IQHIP_ERR_HIP, "Newton chain: launch failed");
    return IQHIP_OK;
}
int eng_newton_eval_enqueue(iqhip_engine *e) {
    if (use_device(e) != hipSuccess) return set_error(IQHIP_ERR_HIP, "hipSetDevice");
    const int nwaves = (int)e->ntiles;
    if (e->asc_active && e->n_unobs == 0 && hipMemsetAsync(e->d_result + 2, 0, 3 * sizeof(double), e->stream) != hipSuccess)
        return set_error(IQHIP_ERR_HIP, "Newton chain: memset failed");
    if (launch_derv_at_state(e, nwaves) != hipSuccess || launch_reduce(e, 0, e->n_unobs > 0 ? 5 : 2, nwaves) != hipSuccess)
        return set_error(IQHIP_ERR_HIP, "Newton chain: launch failed");
    return IQHIP_OK;
}
int eng_newton_update_enqueue(iqhip_engine *e) {
    if (use_device(e) != hipSuccess) return set_error(IQHIP_ERR_HIP, "hipSetDevice");
    if (launch_newton_state_update(e) != hipSuccess) return set_error(IQHIP_ERR_HIP, "Newton chain: launch failed");
    return IQHIP_OK;
}

int newton_state_read(iqhip_engine *e) {
    if (use_device(e) != hipSuccess) return set_error(IQHIP_ERR_HIP, "hipSetDevice");
    if (hipMemcpyAsync(e->h_nstate, e->d_nstate, sizeof(NewtonState), hipMemcpyDeviceToHost, e->stream) != hipSuccess ||
        hipStreamSynchronize(e->stream) != hipSuccess)
        return set_error(IQHIP_ERR_HIP, "Newton chain: state read failed");
    e->staging_busy = false;
    return IQHIP_OK;
}
}  // namespace iqhip

// The state machine on the host, for callers that own the collective themselves (they evaluate {df, ddf} with
// iqhip_derv_async, all-reduce them their own way and need the reference's update rule between evaluations) and for
// the CPU tests, which check it step by step against the loop form of optimization.cpp:388-465.
extern "C" int iqhip_newton_host_init(void *state, double xguess, double x1, double x2, double xacc, int max_steps,
                                      double *first_x) {
    if (!state) return fail(IQHIP_ERR_INVALID, "null argument");
    if (!(x1 >= 0.0) || !(x2 > x1) || !(xacc > 0.0) || max_steps < 1 || !(xguess >= 0.0))
        return fail(IQHIP_ERR_INVALID, "iqhip_newton_host_init: bad bounds / tolerance / step count");
    static_assert(sizeof(NewtonState) == IQHIP_NEWTON_STATE_BYTES, "NewtonState size");
    NewtonState st;
    newton_init(st, xguess, x1, x2, xacc, max_steps);
    memcpy(state, &st, sizeof st);
    if (first_x) *first_x = st.rts;
    return IQHIP_OK;
}

extern "C" int iqhip_newton_host_update(void *state, double df_sum, double ddf_sum, double *next_x, int *done) {
    if (!state) return fail(IQHIP_ERR_INVALID, "null argument");
    NewtonState st;
    memcpy(&st, state, sizeof st);
    newton_update(st, df_sum, ddf_sum);
    memcpy(state, &st, sizeof st);
    if (next_x) *next_x = st.rts;
    if (done) *done = st.done;
    return IQHIP_OK;
}

extern "C" int iqhip_newton_host_result(const void *state, double *optx, double *d2l, int *nsteps, int *status) {
    if (!state) return fail(IQHIP_ERR_INVALID, "null argument");
    NewtonState st;
    memcpy(&st, state, sizeof st);
    if (!st.done) return fail(IQHIP_ERR_INVALID, "Newton state machine has not finished");
    if (optx) *optx = st.result;
    if (d2l) *d2l = st.d2l;
    if (nsteps) *nsteps = st.neval;
    if (status) *status = st.status;
    return IQHIP_OK;
}

// theta must be resident; result[2..] untouched
static int newton_chain(iqhip_engine *e, double xguess, double x1, double x2, double xacc, int max_steps,
                        double *optx, double *d2l, int *nsteps) {
    int rc = newton_state_alloc(e);
    if (rc) return rc;
    rc = ensure_slab_rows(e, 5);
    if (rc) return rc;
    HIPCHK(launch_newton_state_init(e, xguess, x1, x2, xacc, max_steps));
    // a typical solve converges in 3..5 evaluations: enqueue that many before looking
    int enq = 0;
    for (;;) {
        const int chunk = enq == 0 ? std::min(4, max_steps + 1) : 2;
        rc = newton_chain_enqueue(e, chunk);
        if (rc) return rc;
        enq += chunk;
        rc = newton_state_read(e);
        if (rc) return rc;
        if (e->h_nstate->done) break;
        if (enq > max_steps + 2) return fail(IQHIP_ERR_INVALID, "Newton chain did not terminate");
    }
    const NewtonState &st = *e->h_nstate;
    if (st.status == 2) return fail(IQHIP_ERR_INVALID, "Wrong computeFuncDerv (non-finite derivative)");
    if (st.status == 3) return fail(IQHIP_ERR_INVALID, "Maximum number of iterations exceeded in minimizeNewton");
    if (optx) *optx = st.result;
    if (d2l) *d2l = st.d2l;
    if (nsteps) *nsteps = st.neval;
    return IQHIP_OK;
}

// k_newton's grid barrier needs every workgroup resident; a single workgroup needs no barrier.  IQHIP_NEWTON=chain
// forces the chain form (tests).
static bool newton_use_chain(const iqhip_engine *e) {
    if (e->comm) return true;
    static const bool forced = [] { const char *v = getenv("IQHIP_NEWTON"); return v && !strcmp(v, "chain"); }();
    return forced;
}

extern "C" int iqhip_newton_branch(iqhip_engine *e, double xguess, double x1, double x2, double xacc,
                                   int max_steps, double *optx, double *d2l, int *nsteps) {
    if (!(x1 >= 0.0) || !(x2 > x1) || !(xacc > 0.0) || max_steps < 1 || !(xguess >= 0.0))
        return fail(IQHIP_ERR_INVALID, "iqhip_newton_branch: bad bounds / tolerance / step count");
    if (e && !e->shards.empty()) {
        iqhip_branch_end none = {0, -1, 0};
        return sharded::optimize_branch(e, nullptr, 0, false, none, none, xguess, x1, x2, xacc, max_steps, nullptr, optx,
                                        d2l, nsteps);
    }
    int rc = check_ready(e);
    if (rc) return rc;
    if (!e->theta_valid) return fail(IQHIP_ERR_INVALID, "iqhip_newton_branch: theta not computed");
    if (newton_use_chain(e)) {
        return newton_chain(e, xguess, x1, x2, xacc, max_steps, optx, d2l, nsteps);
    }
    HIPCHK(launch_newton(e, xguess, x1, x2, xacc, max_steps, e->d_result));
    rc = read_result(e, 4);
    if (rc) return rc;
    const int status = (int)e->h_result[3];
    if (status == 2) return fail(IQHIP_ERR_INVALID, "Wrong computeFuncDerv (non-finite derivative)");
    if (status == 3) return fail(IQHIP_ERR_INVALID, "Maximum number of iterations exceeded in minimizeNewton");
    if (status == 4) {  // the grid barrier gave up (another kernel held the CUs): the chain needs no barrier
        (void)hipStreamSynchronize(e->stream);
        return newton_chain(e, xguess, x1, x2, xacc, max_steps, optx, d2l, nsteps);
    }
    if (optx) *optx = e->h_result[0];
    if (d2l) *d2l = e->h_result[1];
    if (nsteps) *nsteps = (int)e->h_result[2];
    return IQHIP_OK;
}

extern "C" int iqhip_optimize_branch(iqhip_engine *e, const iqhip_node_op *ops, int nops, iqhip_branch_end a,
                                     iqhip_branch_end b, double xguess, double x1, double x2, double xacc,
                                     int max_steps, double *sum_scale, double *optx, double *d2l, int *nsteps) {
    if (!(x1 >= 0.0) || !(x2 > x1) || !(xacc > 0.0) || max_steps < 1 || !(xguess >= 0.0))
        return fail(IQHIP_ERR_INVALID, "iqhip_optimize_branch: bad bounds / tolerance / step count");
    if (e && !e->shards.empty())
        return sharded::optimize_branch(e, ops, nops, true, a, b, xguess, x1, x2, xacc, max_steps, sum_scale, optx, d2l,
                                        nsteps);
    iqhip_branch_end none = {0, -1, 0};
    int rc = IQHIP_OK;
    if (e && newton_use_chain(e)) {

        // sharded rank: node updates (their sum_scale rows all-reduced), theta, then the enqueued Newton chain
        if (nops > 0) {
            rc = iqhip_update_partials(e, ops, nops, sum_scale);
            if (rc) return rc;
        }
        rc = iqhip_compute_theta(e, a, b);
        if (rc) return rc;
        return newton_chain(e, xguess, x1, x2, xacc, max_steps, optx, d2l, nsteps);
    }
    // two launches per branch: the pending node updates, then one kernel that sums their sum_scale rows,
    // builds theta during its first derivative evaluation and runs the whole Newton-Raphson loop
    if (nops > 0) rc = submit_traverse(e, ops, nops, false, none, none, 0.0, /*skip_reduce=*/true);
    else rc = check_ready(e);
    if (rc) return rc;
    if (nops + 6 > e->result_cap) return fail(IQHIP_ERR_INVALID, "too many node updates in one submission");
    DevBranch br;
    rc = build_branch(e, a, b, 0.0, -1, &br);
    if (rc) return rc;
    e->theta_valid = true;
    e->theta_a_sc = br.a_sc;
    e->theta_b_sc = br.b_sc;
    double *out = e->d_result + 2 + nops;
    HIPCHK(launch_newton(e, xguess, x1, x2, xacc, max_steps, out, &br, nops, (int)e->ntiles * e->lane_split));
    rc = read_result(e, 2 + nops + 4);
    if (rc) return rc;
    if (sum_scale)
        for (int k = 0; k < nops; k++) sum_scale[k] = e->h_result[2 + k];
    const double *r = e->h_result + 2 + nops;
    const int status = (int)r[3];
    if (status == 2) return fail(IQHIP_ERR_INVALID, "Wrong computeFuncDerv (non-finite derivative)");
    if (status == 3) return fail(IQHIP_ERR_INVALID, "Maximum number of iterations exceeded in minimizeNewton");
    if (status == 4) {  // grid barrier gave up: theta was built by the first evaluation, finish with the chain
        (void)hipStreamSynchronize(e->stream);
        return newton_chain(e, xguess, x1, x2, xacc, max_steps, optx, d2l, nsteps);
    }
    if (optx) *optx = r[0];
    if (d2l) *d2l = r[1];
    if (nsteps) *nsteps = (int)r[2];
    return IQHIP_OK;
}

// ---------------------------------------------------------------------------------------
// A whole branch-length sweep -- PhyloTree::optimizeAllBranches' loop over optimizeOneBranch (phylotree.cpp:2252-2332,
// 2148-2192) -- in ONE submission.  Step j = {the node updates that are pending at both ends of branch j, theta, the
// Newton solve, the diverged-solve rule}; a child branch that an earlier step of the sweep optimised has its length read
// from device memory (sweep_len[step]), where that step's Newton kernel left it, so nothing comes back to the host
// between steps: 2 launches per step are enqueued back to back and the host reads one result block per sweep.
// The caller builds the steps as if every step changed its branch (optimizeOneBranch's clearReversePartialLh on both
// sides); a step that ends where it started only makes the later steps recompute vectors that were still valid.
// ---------------------------------------------------------------------------------------
static int sweep_resolve_ops(const iqhip_sweep_step &st, const iqhip_branch_result *results, std::vector<iqhip_node_op> &ops) {
    ops.assign(st.ops, st.ops + st.nops);
    if (st.len_from)
        for (int k = 0; k < st.nops; k++) {
            if (st.len_from[2 * k] >= 0) ops[k].left_len = results[st.len_from[2 * k]].optx;
            if (st.len_from[2 * k + 1] >= 0) ops[k].right_len = results[st.len_from[2 * k + 1]].optx;
        }
    return IQHIP_OK;
}

// the same sweep one step at a time (a host round trip per step): sharded engines and engines with a communicator, where
// every Newton step contains an all-reduce; +ASC; and the remainder of a sweep whose grid-wide exchange timed out
static int sweep_sequential(iqhip_engine *e, const iqhip_sweep_step *steps, int first, int nsteps, double x1, double x2,
                            double xacc, int max_steps, double diverge_frac, double *sum_scale, size_t ss_off,
                            iqhip_branch_result *results) {
    std::vector<iqhip_node_op> ops;
    for (int j = first; j < nsteps; j++) {
        const iqhip_sweep_step &st = steps[j];
        sweep_resolve_ops(st, results, ops);
        iqhip_branch_result &r = results[j];
        r.status = 0;
        r.lnl = 0.0;
        int rc = iqhip_optimize_branch(e, ops.empty() ? nullptr : ops.data(), st.nops, st.a, st.b, st.xguess, x1, x2, xacc, max_steps,
                                       sum_scale ? sum_scale + ss_off : nullptr, &r.optx, &r.d2l, &r.nsteps);
        if (rc) return rc;
        if (diverge_frac > 0.0 && r.optx > diverge_frac * x2) {   // phylotree.cpp:2167-2176
            double opt_lh = 0.0, orig_lh = 0.0;
            rc = iqhip_lnl_from_theta(e, r.optx, &opt_lh);
            if (!rc) rc = iqhip_lnl_from_theta(e, st.xguess, &orig_lh);
            if (rc) return rc;
            if (orig_lh > opt_lh) r.optx = st.xguess;
            r.status = 5;   // (informational: the rule was applied)
        }
        ss_off += (size_t)st.nops;
    }
    return IQHIP_OK;
}

// 4-state engines: the whole sweep as ONE launch of the persistent kernel k_sweep4 (kernels_sweep.hip) + one k_reduce for
// the sum_scale rows; the host only resolves keys into descriptors, copies them down once and reads one result block
static int sweep_persistent4(iqhip_engine *e, const iqhip_sweep_step *steps, int nsteps, size_t total_ops, double x1, double x2,
                             double xacc, int max_steps, double diverge_frac, double *sum_scale, iqhip_branch_result *results) {
    static const bool dbg = getenv("IQHIP_DEBUG_SWEEP") != nullptr;
    timespec t0, t1, t2;
    if (dbg) clock_gettime(CLOCK_MONOTONIC, &t0);
    const size_t bytes_ops = sizeof(SweepOp) * total_ops, bytes_steps = sizeof(SweepStep) * (size_t)nsteps;
    const size_t need = bytes_ops + bytes_steps;
    if (need > e->sweep_desc_cap) {
        HIPCHK(hipStreamSynchronize(e->stream));
        if (e->h_sweep_desc) hipHostFree(e->h_sweep_desc);
        if (e->d_sweep_desc) hipFree(e->d_sweep_desc);
        e->h_sweep_desc = e->d_sweep_desc = nullptr;
        e->sweep_desc_cap = 0;
        const size_t cap = need * 2 + 4096;
        HIPCHK(hipHostMalloc((void **)&e->h_sweep_desc, cap));
        HIPCHK(hipMalloc((void **)&e->d_sweep_desc, cap));
        e->sweep_desc_cap = cap;
    }
    SweepOp *hops = reinterpret_cast<SweepOp *>(e->h_sweep_desc);
    SweepStep *hsteps = reinterpret_cast<SweepStep *>(e->h_sweep_desc + bytes_ops);
    size_t row = 0;
    for (int j = 0; j < nsteps; j++) {
        const iqhip_sweep_step &st = steps[j];
        SweepStep &hs = hsteps[j];
        hs.op_begin = (int32_t)row;
        hs.nops = st.nops;
        hs.xguess = st.xguess;
        for (int k = 0; k < st.nops; k++, row++) {
            const iqhip_node_op &o = st.ops[k];
            SweepOp &d = hops[row];
            memset(&d, 0, sizeof d);
            if (!(o.left_len >= 0.0) || !(o.right_len >= 0.0)) return fail(IQHIP_ERR_INVALID, "negative or NaN branch length");
            const uint8_t *lst, *rst;
            int32_t lk, rk;
            int rc = resolve_child(e, o.left_key, o.left_leaf, -1, &d.lv, &d.lsc, &lst, &lk);
            if (rc) return rc;
            rc = resolve_child(e, o.right_key, o.right_leaf, -1, &d.rv, &d.rsc, &rst, &rk);
            if (rc) return rc;
            d.ls = lst ? lst : e->d_states;
            d.rs = rst ? rst : e->d_states;
            int didx;
            rc = slab_for_key(e, o.dst_key, true, &didx);
            if (rc) return rc;
            d.dst = e->slabs[didx].plh;
            d.dst_sc = e->slabs[didx].sc;
            if (d.lv == d.dst || d.rv == d.dst) return fail(IQHIP_ERR_INVALID, "node update writes onto one of its own children");
            d.llen = o.left_len;
            d.rlen = o.right_len;
            d.llen_step = st.len_from ? st.len_from[2 * k] : -1;
            d.rlen_step = st.len_from ? st.len_from[2 * k + 1] : -1;
            d.no_scale = (o.flags & IQHIP_OP_NO_SCALE) ? 1 : (((o.flags & IQHIP_OP_SCALAR_RULE) || e->scalar_rule_all) ? 2 : 0);
            d.row = (int32_t)row;
        }
        int rc = build_branch(e, st.a, st.b, 0.0, -1, &hs.br);
        if (rc) return rc;
    }
    const int grid = sweep4_grid(e), nwaves = grid * sweep4_waves(e);
    if ((int64_t)total_ops * nwaves > e->slab_cap) {
        HIPCHK(hipStreamSynchronize(e->stream));
        if (e->d_slab) hipFree(e->d_slab);
        e->d_slab = nullptr;
        e->slab_cap = 0;
        HIPCHK(dmalloc(&e->d_slab, total_ops * (size_t)nwaves + 1024));
        e->slab_cap = (int64_t)(total_ops * (size_t)nwaves + 1024);
    }
    const size_t posts_need = (size_t)2 * kNewtonPostEpochs * grid * 2;
    if (posts_need > e->sweep_posts_cap) {
        HIPCHK(hipStreamSynchronize(e->stream));
        if (e->d_sweep_posts) hipFree(e->d_sweep_posts);
        e->d_sweep_posts = nullptr;
        e->sweep_posts_cap = 0;
        HIPCHK(dmalloc(&e->d_sweep_posts, posts_need));
        e->sweep_posts_cap = posts_need;
    }
    HIPCHK(hipMemcpyAsync(e->d_sweep_desc, e->h_sweep_desc, need, hipMemcpyHostToDevice, e->stream));
    if (grid > 1) HIPCHK(hipMemsetAsync(e->d_sweep_posts, 0xFF, posts_need * sizeof(double), e->stream));
    double *out = e->d_result + total_ops;      // (rows [0, total_ops) receive the sum_scale sums from k_reduce)
    memset(e->h_result + total_ops, 0, sizeof(double) * 6 * (size_t)nsteps);
    HIPCHK(launch_sweep4(e, reinterpret_cast<const SweepOp *>(e->d_sweep_desc),
                         reinterpret_cast<const SweepStep *>(e->d_sweep_desc + bytes_ops), nsteps, x1, x2, xacc, max_steps,
                         diverge_frac * x2, e->d_sweep_posts, out));
    HIPCHK(launch_reduce(e, 0, (int)total_ops, nwaves));
    const DevBranch &last = hsteps[nsteps - 1].br;
    e->theta_valid = true;
    e->theta_a_sc = last.a_sc;
    e->theta_b_sc = last.b_sc;
    e->last_plan_version = 0;
    if (dbg) clock_gettime(CLOCK_MONOTONIC, &t1);
    int rc = read_result(e, (int)(total_ops + 6 * (size_t)nsteps));
    if (rc) return rc;
    if (dbg) {
        clock_gettime(CLOCK_MONOTONIC, &t2);
        fprintf(stderr, "[iqhip] persistent sweep of %d steps (%zu node updates): descriptors + enqueue %.1f us, wait %.1f us\n", nsteps,
                total_ops, (t1.tv_sec - t0.tv_sec) * 1e6 + (t1.tv_nsec - t0.tv_nsec) * 1e-3,
                (t2.tv_sec - t1.tv_sec) * 1e6 + (t2.tv_nsec - t1.tv_nsec) * 1e-3);
    }
    if (sum_scale)
        for (size_t k = 0; k < total_ops; k++) sum_scale[k] = e->h_result[k];
    size_t ss = 0;
    for (int j = 0; j < nsteps; j++) {
        const double *o = e->h_result + total_ops + 6 * (size_t)j;
        const int status = (int)o[3];
        if (status == 2) return fail(IQHIP_ERR_INVALID, "Wrong computeFuncDerv (non-finite derivative)");
        if (status == 3) return fail(IQHIP_ERR_INVALID, "Maximum number of iterations exceeded in minimizeNewton");
        if (status == 4) {   // the exchange between the workgroups gave up: finish from here one step at a time
            (void)hipStreamSynchronize(e->stream);
            return sweep_sequential(e, steps, j, nsteps, x1, x2, xacc, max_steps, diverge_frac, sum_scale, ss, results);
        }
        results[j].optx = o[0];
        results[j].d2l = o[1];
        results[j].nsteps = (int)o[2];
        results[j].status = o[4] != 0.0 ? 5 : 0;
        results[j].lnl = 0.0;
        ss += (size_t)steps[j].nops;
    }
    return IQHIP_OK;
}

extern "C" int iqhip_optimize_sweep(iqhip_engine *e, const iqhip_sweep_step *steps, int nsteps, double x1, double x2,
                                    double xacc, int max_steps, double diverge_frac, double *sum_scale,
                                    iqhip_branch_result *results) {
    if (!e) return fail(IQHIP_ERR_INVALID, "null engine");
    if (!steps || !results || nsteps < 1) return fail(IQHIP_ERR_INVALID, "iqhip_optimize_sweep: bad step array");
    if (!(x1 >= 0.0) || !(x2 > x1) || !(xacc > 0.0) || max_steps < 1 || !(diverge_frac >= 0.0) || diverge_frac >= 1.0)
        return fail(IQHIP_ERR_INVALID, "iqhip_optimize_sweep: bad bounds / tolerance / step count");
    size_t total_ops = 0;
    for (int j = 0; j < nsteps; j++) {
        const iqhip_sweep_step &st = steps[j];
        if (st.nops < 0 || (st.nops > 0 && !st.ops)) return fail(IQHIP_ERR_INVALID, "bad ops array in a sweep step");
        if (!(st.xguess >= 0.0)) return fail(IQHIP_ERR_INVALID, "iqhip_optimize_sweep: bad starting length");
        if (st.len_from)
            for (int q = 0; q < 2 * st.nops; q++)
                if (st.len_from[q] >= j) return fail(IQHIP_ERR_INVALID, "a sweep step may only use the lengths of earlier steps");
        total_ops += (size_t)st.nops;
    }
    static const bool one_submission = [] { const char *v = getenv("IQHIP_SWEEP"); return !v || atoi(v) != 0; }();
    if (!e->shards.empty() || e->comm || e->n_unobs > 0 || !one_submission || newton_use_chain(e) ||
        2 + total_ops + 6 * (size_t)nsteps > (size_t)e->result_cap || e->d_result != e->d_result_own)
        return sweep_sequential(e, steps, 0, nsteps, x1, x2, xacc, max_steps, diverge_frac, sum_scale, 0, results);
    int rc = check_ready(e);
    if (rc) return rc;
    static const bool persistent = [] { const char *v = getenv("IQHIP_SWEEP_KERNEL"); return !v || atoi(v) != 0; }();
    if (persistent && !e->mfma && e->nclass == 1 && nsteps <= 4096 && max_steps + 5 <= kNewtonPostEpochs && total_ops > 0)
        return sweep_persistent4(e, steps, nsteps, total_ops, x1, x2, xacc, max_steps, diverge_frac, sum_scale, results);
    if (!e->h_plan_arena) {
        e->plan_arena_cap = 1 << 20;
        if (hipHostMalloc((void **)&e->h_plan_arena, e->plan_arena_cap) != hipSuccess) { e->h_plan_arena = nullptr; e->plan_arena_cap = 0; }
    }
    struct ArenaScope {   // plan uploads of this sweep go through the arena; whatever happens, switched off on return
        iqhip_engine *e;
        explicit ArenaScope(iqhip_engine *e_) : e(e_) { e->plan_arena_on = e->h_plan_arena != nullptr; e->plan_arena_used = 0; }
        ~ArenaScope() { e->plan_arena_on = false; }
    } arena_scope(e);
    if (nsteps > e->sweep_len_cap) {
        HIPCHK(hipStreamSynchronize(e->stream));
        if (e->d_sweep_len) hipFree(e->d_sweep_len);
        e->d_sweep_len = nullptr;
        e->sweep_len_cap = 0;
        HIPCHK(dmalloc(&e->d_sweep_len, (size_t)nsteps + 64));
        e->sweep_len_cap = nsteps + 64;
    }
    static const bool dbg = getenv("IQHIP_DEBUG_SWEEP") != nullptr;
    double t_trav = 0.0, t_newt = 0.0;
    int n_uploaded = 0;
    timespec ts0;
    clock_gettime(CLOCK_MONOTONIC, &ts0);
    // result block in the (host-mapped) result vector: per step its sum_scale rows, then {optx, d2l, nsteps, status, diverged, -}
    std::vector<size_t> row_of(nsteps);
    size_t row = 2;
    std::vector<const double *> len_ptrs;
    iqhip_branch_end none = {0, -1, 0};
    for (int j = 0; j < nsteps; j++) {
        const iqhip_sweep_step &st = steps[j];
        row_of[j] = row;
        if (st.nops > 0) {
            const double *const *lp = nullptr;
            if (st.len_from) {
                len_ptrs.assign((size_t)2 * st.nops, nullptr);
                bool any = false;
                for (int q = 0; q < 2 * st.nops; q++)
                    if (st.len_from[q] >= 0) { len_ptrs[q] = e->d_sweep_len + st.len_from[q]; any = true; }
                if (any) lp = len_ptrs.data();
            }
            timespec ta, tb;
            if (dbg) clock_gettime(CLOCK_MONOTONIC, &ta);
            rc = submit_traverse(e, st.ops, st.nops, false, none, none, 0.0, /*skip_reduce=*/true, nullptr, lp);
            if (rc) return rc;
            if (dbg) { clock_gettime(CLOCK_MONOTONIC, &tb); t_trav += (tb.tv_sec - ta.tv_sec) * 1e6 + (tb.tv_nsec - ta.tv_nsec) * 1e-3; n_uploaded += e->plan_small ? 0 : 1; }
        }
        DevBranch br;
        rc = build_branch(e, st.a, st.b, 0.0, -1, &br);
        if (rc) return rc;
        e->theta_valid = true;
        e->theta_a_sc = br.a_sc;
        e->theta_b_sc = br.b_sc;
        NewtonSweepStep sw;
        sw.len_out = e->d_sweep_len + j;
        sw.rows_base = e->d_result + row;
        sw.diverge_x = diverge_frac * x2;
        sw.publish = (j == nsteps - 1);
        timespec tc, td;
        if (dbg) clock_gettime(CLOCK_MONOTONIC, &tc);
        HIPCHK(launch_newton(e, st.xguess, x1, x2, xacc, max_steps, e->d_result + row + st.nops, &br, st.nops,
                             (int)e->ntiles * e->lane_split, &sw));
        if (dbg) { clock_gettime(CLOCK_MONOTONIC, &td); t_newt += (td.tv_sec - tc.tv_sec) * 1e6 + (td.tv_nsec - tc.tv_nsec) * 1e-3; }
        row += (size_t)st.nops + 6;
    }
    timespec ts1;
    if (dbg) clock_gettime(CLOCK_MONOTONIC, &ts1);
    rc = read_result(e, (int)row);
    if (rc) return rc;
    if (dbg) {
        timespec ts2;
        clock_gettime(CLOCK_MONOTONIC, &ts2);
        fprintf(stderr, "[iqhip] sweep of %d steps (%d plans uploaded, the others in the kernel arguments): submit_traverse %.1f us (build_plan %.1f us since the last report), launch_newton %.1f us; enqueue %.1f us, wait %.1f us\n", nsteps, n_uploaded, t_trav, g_dbg_build_us, t_newt,
                (ts1.tv_sec - ts0.tv_sec) * 1e6 + (ts1.tv_nsec - ts0.tv_nsec) * 1e-3,
                (ts2.tv_sec - ts1.tv_sec) * 1e6 + (ts2.tv_nsec - ts1.tv_nsec) * 1e-3);
        g_dbg_build_us = 0.0;
    }
    size_t ss = 0;
    for (int j = 0; j < nsteps; j++) {
        const iqhip_sweep_step &st = steps[j];
        const double *r = e->h_result + row_of[j];
        if (sum_scale)
            for (int k = 0; k < st.nops; k++) sum_scale[ss + k] = r[k];
        const double *o = r + st.nops;
        const int status = (int)o[3];
        if (status == 2) return fail(IQHIP_ERR_INVALID, "Wrong computeFuncDerv (non-finite derivative)");
        if (status == 3) return fail(IQHIP_ERR_INVALID, "Maximum number of iterations exceeded in minimizeNewton");
        if (status == 4) {
            // the grid-wide exchange of this step's solve gave up (another kernel held the CUs): its length and everything
            // after it is void -- finish the sweep from here one step at a time (the chain form needs no co-residency)
            (void)hipStreamSynchronize(e->stream);
            return sweep_sequential(e, steps, j, nsteps, x1, x2, xacc, max_steps, diverge_frac, sum_scale, ss, results);
        }
        results[j].optx = o[0];
        results[j].d2l = o[1];
        results[j].nsteps = (int)o[2];
        results[j].status = o[4] != 0.0 ? 5 : 0;
        results[j].lnl = 0.0;
        ss += (size_t)st.nops;
    }
    return IQHIP_OK;
}

// ---------------------------------------------------------------------------------------
// Batched chain: iqhip_optimize_branch_batch on pattern shards.  The m tasks of a chunk advance side by side: per Newton
// step ONE derivative launch (grid.y = task, branch length read from the task's device-resident state machine), ONE
// k_reduce over 2m slab rows, ONE all-reduce of 2m doubles, ONE update kernel (thread = task); tasks that have converged
// do nothing.  Identical sums on every rank => identical iterates, so all ranks leave the loop together.
// ---------------------------------------------------------------------------------------
namespace iqhip {
static BatchChain batch_chain_of(const iqhip_engine *e, int m) {
    return BatchChain{e->d_theta_batch, (size_t)e->nptn_pad * e->block, e->d_bstates, m};
}

int eng_batch_prepare(iqhip_engine *e, const iqhip_branch_task *tasks, int m, const NewtonState *init) {
    int rc = check_ready(e);
    if (rc) return rc;
    if (2 * m > e->result_cap) return set_error(IQHIP_ERR_INVALID, "too many tasks in one chunk");
    rc = ensure_slab_rows(e, std::max(5, 2 * m));
    if (rc) return rc;
    const size_t theta_stride = (size_t)e->nptn_pad * e->block;
    if ((size_t)m * theta_stride > e->theta_batch_cap) {
        HIPCHK(hipStreamSynchronize(e->stream));
        if (e->d_theta_batch) hipFree(e->d_theta_batch);
        e->d_theta_batch = nullptr;
        e->theta_batch_cap = 0;
        HIPCHK(dmalloc(&e->d_theta_batch, (size_t)m * theta_stride));
        e->theta_batch_cap = (size_t)m * theta_stride;
    }
    if (m > e->bstates_cap) {
        HIPCHK(hipStreamSynchronize(e->stream));
        if (e->d_bstates) hipFree(e->d_bstates);
        e->d_bstates = nullptr;
        e->bstates_cap = 0;
        HIPCHK(hipMalloc((void **)&e->d_bstates, sizeof(NewtonState) * (size_t)m));
        e->bstates_cap = m;
    }
    for (int t = 0; t < m; t++) {
        DevBranch br;
        rc = build_branch(e, tasks[t].a, tasks[t].b, 0.0, -1, &br);
        if (rc) return rc;
        double *slot = e->d_theta_batch + (size_t)t * theta_stride;
        if (e->mfma) HIPCHK(launch_stream_mfma(e, 1, &br, 0.0, (int)e->ntiles, nullptr, -1, slot));
        else HIPCHK(launch_theta4(e, br, slot));
    }
    return eng_batch_states_write(e, m, init);
}

int eng_batch_states_write(iqhip_engine *e, int m, const NewtonState *in) {
    if (use_device(e) != hipSuccess) return set_error(IQHIP_ERR_HIP, "hipSetDevice");
    // (pageable source: the copy has left the host buffer when the call returns)
    if (hipMemcpyAsync(e->d_bstates, in, sizeof(NewtonState) * (size_t)m, hipMemcpyHostToDevice, e->stream) != hipSuccess)
        return set_error(IQHIP_ERR_HIP, "batched chain: state upload failed");
    return IQHIP_OK;
}

int eng_batch_states_read(iqhip_engine *e, int m, NewtonState *out) {
    if (use_device(e) != hipSuccess) return set_error(IQHIP_ERR_HIP, "hipSetDevice");
    if (hipMemcpyAsync(out, e->d_bstates, sizeof(NewtonState) * (size_t)m, hipMemcpyDeviceToHost, e->stream) != hipSuccess ||
        hipStreamSynchronize(e->stream) != hipSuccess)
        return set_error(IQHIP_ERR_HIP, "batched chain: state read failed");
    e->staging_busy = false;
    return IQHIP_OK;
}

int eng_batch_eval_enqueue(iqhip_engine *e, int m) {
    if (use_device(e) != hipSuccess) return set_error(IQHIP_ERR_HIP, "hipSetDevice");
    const BatchChain bc = batch_chain_of(e, m);
    const int nwaves = (int)e->ntiles;
    const hipError_t s = e->mfma ? launch_stream_mfma(e, 2, nullptr, 0.0, nwaves, nullptr, -1, nullptr, &bc)
                                 : launch_derv4(e, 0.0, nwaves, nullptr, &bc);
    if (s != hipSuccess || launch_reduce(e, 0, 2 * m, nwaves) != hipSuccess)
        return set_error(IQHIP_ERR_HIP, "batched chain: launch failed");
    return IQHIP_OK;
}

int eng_batch_update_enqueue(iqhip_engine *e, int m) {
    if (use_device(e) != hipSuccess) return set_error(IQHIP_ERR_HIP, "hipSetDevice");
    if (launch_newton_state_update_batch(e, e->d_bstates, m) != hipSuccess)
        return set_error(IQHIP_ERR_HIP, "batched chain: launch failed");
    return IQHIP_OK;
}

int eng_batch_lnl_enqueue(iqhip_engine *e, int m) {
    if (use_device(e) != hipSuccess) return set_error(IQHIP_ERR_HIP, "hipSetDevice");
    const BatchChain bc = batch_chain_of(e, m);
    const int nwaves = (int)e->ntiles;
    const hipError_t s = e->mfma ? launch_stream_mfma(e, 3, nullptr, 0.0, nwaves, nullptr, -1, nullptr, &bc)
                                 : launch_lnl_theta4(e, 0.0, nwaves, &bc);
    if (s != hipSuccess || launch_reduce(e, 0, 2 * m, nwaves) != hipSuccess)
        return set_error(IQHIP_ERR_HIP, "batched chain: launch failed");
    return IQHIP_OK;
}
}  // namespace iqhip

// tasks per chunk of the batched chain: the same on every rank (the ranks' collectives must pair up), so a fixed number
// and not what the free memory of this device suggests
static int batch_chain_chunk(int ntasks) {
    int chunk = std::min(ntasks, 64);
    if (const char *bc = getenv("IQHIP_BATCH_CHUNK")) chunk = std::max(1, std::min(chunk, atoi(bc)));
    return chunk;
}

static int batch_task_check(const iqhip_branch_task &k) {
    if (k.nops < 0 || (k.nops > 0 && !k.ops)) return fail(IQHIP_ERR_INVALID, "bad ops array in a task");
    if (!(k.x1 >= 0.0) || !(k.x2 > k.x1) || !(k.xacc > 0.0) || k.max_steps < 1 || !(k.xguess >= 0.0))
        return fail(IQHIP_ERR_INVALID, "iqhip_optimize_branch_batch: bad bounds / tolerance / step count");
    return IQHIP_OK;
}

static int batch_task_results(const std::vector<NewtonState> &st, int m, iqhip_branch_result *results) {
    for (int t = 0; t < m; t++) {
        if (st[t].status == 2) return fail(IQHIP_ERR_INVALID, "Wrong computeFuncDerv (non-finite derivative)");
        if (st[t].status == 3) return fail(IQHIP_ERR_INVALID, "Maximum number of iterations exceeded in minimizeNewton");
        results[t].optx = st[t].result;
        results[t].d2l = st[t].d2l;
        results[t].nsteps = st[t].neval;
        results[t].status = 0;
    }
    return IQHIP_OK;
}

// a rank with a communicator: node updates of all tasks in one submission, then the batched chain
static int optimize_branch_batch_comm(iqhip_engine *e, const iqhip_branch_task *tasks, int ntasks, double *sum_scale,
                                      iqhip_branch_result *results) {
    int rc = check_ready(e);
    if (rc) return rc;
    std::vector<iqhip_node_op> all;
    std::vector<int> segs(ntasks);
    for (int t = 0; t < ntasks; t++) {
        rc = batch_task_check(tasks[t]);
        if (rc) return rc;
        segs[t] = tasks[t].nops;
        all.insert(all.end(), tasks[t].ops, tasks[t].ops + tasks[t].nops);
    }
    const int total_ops = (int)all.size();
    if (total_ops + 2 > e->result_cap) return fail(IQHIP_ERR_INVALID, "too many node updates in one submission");
    iqhip_branch_end none = {0, -1, 0};
    if (total_ops > 0) {
        rc = submit_traverse(e, all.data(), total_ops, false, none, none, 0.0, /*skip_reduce=*/false, &segs);
        if (rc) return rc;
        rc = comm_allreduce(e, 2 + total_ops);
        if (rc) return rc;
        rc = read_result(e, 2 + total_ops);
        if (rc) return rc;
        if (sum_scale)
            for (int k = 0; k < total_ops; k++) sum_scale[k] = e->h_result[2 + k];
    }
    const int chunk = batch_chain_chunk(ntasks);
    std::vector<NewtonState> st((size_t)chunk);
    for (int first = 0; first < ntasks; first += chunk) {
        const int m = std::min(chunk, ntasks - first);
        int max_steps = 1;
        for (int t = 0; t < m; t++) {
            const iqhip_branch_task &k = tasks[first + t];
            newton_init(st[t], k.xguess, k.x1, k.x2, k.xacc, k.max_steps);
            max_steps = std::max(max_steps, k.max_steps);
        }
        rc = eng_batch_prepare(e, tasks + first, m, st.data());
        if (rc) return rc;
        int enq = 0;
        for (;;) {
            const int steps = enq == 0 ? std::min(4, max_steps + 1) : 2;
            for (int k = 0; k < steps; k++) {
                rc = eng_batch_eval_enqueue(e, m);
                if (!rc) rc = comm_allreduce(e, 2 * m);
                if (!rc) rc = eng_batch_update_enqueue(e, m);
                if (rc) return rc;
            }
            enq += steps;
            rc = eng_batch_states_read(e, m, st.data());
            if (rc) return rc;
            bool all_done = true;
            for (int t = 0; t < m; t++) all_done = all_done && st[t].done;
            if (all_done) break;
            if (enq > max_steps + 2) return fail(IQHIP_ERR_INVALID, "Newton chain did not terminate");
        }
        rc = batch_task_results(st, m, results + first);
        if (rc) return rc;
        // lnL of every task at its accepted length: one launch, one reduction, one all-reduce
        rc = eng_batch_lnl_enqueue(e, m);
        if (!rc) rc = comm_allreduce(e, 2 * m);
        if (!rc) rc = read_result(e, 2 * m);
        if (rc) return rc;
        std::vector<double> lnl((size_t)m);
        for (int t = 0; t < m; t++) lnl[t] = e->h_result[2 * t];
        for (int t = 0; t < m; t++) {
            results[first + t].lnl = lnl[t];
            if (isnan(lnl[t]) || isinf(lnl[t])) {   // phylokernel.h:1091-1109: redo this task alone (same decision on every rank)
                const iqhip_branch_task &k = tasks[first + t];
                rc = iqhip_compute_theta(e, k.a, k.b);
                if (!rc) rc = iqhip_lnl_from_theta(e, results[first + t].optx, &results[first + t].lnl);
                if (rc) return rc;
            }
        }
    }
    return IQHIP_OK;
}

// iqhip_optimize_branch_batch, one task after the other through the chain form (IQHIP_BATCH_SEQUENTIAL=1: the form the
// batched chain is tested against)
static int optimize_branch_batch_sequential(iqhip_engine *e, const iqhip_branch_task *tasks, int ntasks,
                                            double *sum_scale, iqhip_branch_result *results) {
    size_t off = 0;
    for (int t = 0; t < ntasks; t++) {
        const iqhip_branch_task &k = tasks[t];
        if (k.nops < 0 || (k.nops > 0 && !k.ops)) return fail(IQHIP_ERR_INVALID, "bad ops array in a task");
        iqhip_branch_result &r = results[t];
        r.status = 0;
        int rc = iqhip_optimize_branch(e, k.ops, k.nops, k.a, k.b, k.xguess, k.x1, k.x2, k.xacc, k.max_steps,
                                       sum_scale ? sum_scale + off : nullptr, &r.optx, &r.d2l, &r.nsteps);
        if (rc) return rc;
        rc = iqhip_lnl_from_theta(e, r.optx, &r.lnl);
        if (rc) return rc;
        off += (size_t)k.nops;
    }
    return IQHIP_OK;
}

extern "C" int iqhip_optimize_branch_batch(iqhip_engine *e, const iqhip_branch_task *tasks, int ntasks,
                                           double *sum_scale, iqhip_branch_result *results) {
    if (!e) return fail(IQHIP_ERR_INVALID, "null engine");
    if (!tasks || !results || ntasks < 1) return fail(IQHIP_ERR_INVALID, "bad task array");
    const char *seq_env = getenv("IQHIP_BATCH_SEQUENTIAL");   // (read per call: the tests compare the two forms)
    const bool sequential = (seq_env && atoi(seq_env) != 0) || e->asc_active;   // (+ASC: 5-row results, one task at a time)
    if (!e->shards.empty())
        return sequential ? optimize_branch_batch_sequential(e, tasks, ntasks, sum_scale, results)
                          : sharded::optimize_branch_batch(e, tasks, ntasks, sum_scale, results);
    if (e->comm)
        return sequential ? optimize_branch_batch_sequential(e, tasks, ntasks, sum_scale, results)
                          : optimize_branch_batch_comm(e, tasks, ntasks, sum_scale, results);
    int rc = check_ready(e);
    if (rc) return rc;
    if (e->n_unobs > 0) return fail(IQHIP_ERR_UNSUPPORTED, "iqhip_optimize_branch_batch: +ASC is not supported");
    std::vector<iqhip_node_op> all;
    std::vector<int> segs(ntasks);
    for (int t = 0; t < ntasks; t++) {
        const iqhip_branch_task &k = tasks[t];
        if (k.nops < 0 || (k.nops > 0 && !k.ops)) return fail(IQHIP_ERR_INVALID, "bad ops array in a task");
        if (!(k.x1 >= 0.0) || !(k.x2 > k.x1) || !(k.xacc > 0.0) || k.max_steps < 1 || !(k.xguess >= 0.0))
            return fail(IQHIP_ERR_INVALID, "iqhip_optimize_branch_batch: bad bounds / tolerance / step count");
        segs[t] = k.nops;
        all.insert(all.end(), k.ops, k.ops + k.nops);
    }
    const int total_ops = (int)all.size();
    if (total_ops + 2 > e->result_cap) return fail(IQHIP_ERR_INVALID, "too many node updates in one submission");
    iqhip_branch_end none = {0, -1, 0};
    if (total_ops > 0) {
        rc = submit_traverse(e, all.data(), total_ops, false, none, none, 0.0, /*skip_reduce=*/false, &segs);
        if (rc) return rc;
    }
    // workgroups per task: every workgroup of a launch must be resident (grid barrier inside each task)
    // ... which bounds the batch by what fits the chip at once: 3 workgroups per CU (k_newton_batch: 125 VGPRs, i.e.
    // four waves per SIMD would fit exactly -- keep a margin), fewer when its LDS (3 val arrays of a block) says so
    const size_t newton_lds = (size_t)(3 * e->block + 8) * sizeof(double) + 64;
    const int wg_per_cu = (int)std::max<size_t>(1, std::min<size_t>(3, (size_t)(150 * 1024) / newton_lds));
    const int capacity = e->num_cus * wg_per_cu;
    const int wgs_needed = (int)std::max<int64_t>(1, (e->ntiles + 3) / 4);
    int chunk = std::min(ntasks, capacity);             // tasks per launch
    {   // every task of a launch owns a theta buffer: keep them within a quarter of the free device memory and
        // run larger batches in several launches (protein+G4 at 50k patterns: 32 MB per task)
        size_t free_b = 0, total_b = 0;
        const size_t per_task = (size_t)e->nptn_pad * e->block * sizeof(double);
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const size_t have = e->theta_batch_cap * sizeof(double);
            const size_t room = (free_b + have) / 4;
            const size_t fit = std::max<size_t>(1, room / std::max<size_t>(1, per_task));
            if ((size_t)chunk > fit) chunk = (int)fit;
        }
    }
    if (const char *bc = getenv("IQHIP_BATCH_CHUNK")) chunk = std::max(1, std::min(chunk, atoi(bc)));
    const int G = std::max(1, std::min(wgs_needed, capacity / chunk));
    const size_t theta_stride = (size_t)e->nptn_pad * e->block;
    if ((size_t)chunk * theta_stride > e->theta_batch_cap) {
        HIPCHK(hipStreamSynchronize(e->stream));
        if (e->d_theta_batch) hipFree(e->d_theta_batch);
        e->d_theta_batch = nullptr;
        e->theta_batch_cap = 0;
        HIPCHK(dmalloc(&e->d_theta_batch, (size_t)chunk * theta_stride));
        e->theta_batch_cap = (size_t)chunk * theta_stride;
    }
    if (chunk > e->batch_cap) {
        HIPCHK(hipStreamSynchronize(e->stream));
        void *old[] = {e->d_batch_partials, e->d_batch_out, e->d_batch_barriers, e->d_batch_tasks};
        for (void *p : old)
            if (p) hipFree(p);
        e->d_batch_partials = e->d_batch_out = nullptr;
        e->d_batch_barriers = nullptr;
        e->d_batch_tasks = nullptr;
        e->batch_cap = 0;
        HIPCHK(dmalloc(&e->d_batch_partials, (size_t)chunk * 4 * (e->num_cus * 4)));
        HIPCHK(dmalloc(&e->d_batch_out, (size_t)chunk * 6));
        HIPCHK(dmalloc(&e->d_batch_barriers, (size_t)2 * chunk));
        HIPCHK(hipMalloc(&e->d_batch_tasks, newton_task_bytes() * (size_t)chunk));
        HIPCHK(hipMemsetAsync(e->d_batch_barriers, 0, sizeof(unsigned int) * 2 * chunk, e->stream));
        e->batch_cap = chunk;
    }
    std::vector<char> host_tasks(newton_task_bytes() * (size_t)chunk);
    std::vector<double> out((size_t)chunk * 6);
    for (int first = 0; first < ntasks; first += chunk) {
        const int m = std::min(chunk, ntasks - first);
        for (int t = 0; t < m; t++) {
            const iqhip_branch_task &k = tasks[first + t];
            DevBranch br;
            rc = build_branch(e, k.a, k.b, 0.0, -1, &br);
            if (rc) return rc;
            newton_task_fill(host_tasks.data() + newton_task_bytes() * (size_t)t, br, k.xguess, k.x1, k.x2, k.xacc,
                             k.max_steps);
        }
        HIPCHK(hipMemcpyAsync(e->d_batch_tasks, host_tasks.data(), newton_task_bytes() * (size_t)m,
                              hipMemcpyHostToDevice, e->stream));
        // posted exchange of the tasks' partial sums (k_newton_batch): slots of this launch, [task][evaluation][workgroup][2]
        double *posts = nullptr, *posts_other = nullptr;
        size_t posts_other_used = 0;
        int post_epochs = 0;
        if (G > 1 && e->newton_posts) {
            int max_steps = 1;
            for (int t = 0; t < m; t++) max_steps = std::max(max_steps, tasks[first + t].max_steps);
            post_epochs = max_steps + 4;   // derivative evaluations + the lnL pass(es)
            const size_t need = (size_t)m * post_epochs * G * 2;
            if (need > e->batch_posts_cap) {
                HIPCHK(hipStreamSynchronize(e->stream));
                if (e->d_batch_posts) hipFree(e->d_batch_posts);
                e->d_batch_posts = nullptr;
                e->batch_posts_cap = 0;
                HIPCHK(dmalloc(&e->d_batch_posts, 2 * need));
                HIPCHK(hipMemsetAsync(e->d_batch_posts, 0xFF, 2 * need * sizeof(double), e->stream));
                e->batch_posts_cap = need;
                e->batch_posts_used[0] = e->batch_posts_used[1] = 0;
            }
            const unsigned int pp = e->batch_post_launches & 1u;
            e->batch_post_launches++;
            posts = e->d_batch_posts + (size_t)pp * e->batch_posts_cap;
            posts_other = e->d_batch_posts + (size_t)(1u - pp) * e->batch_posts_cap;
            posts_other_used = e->batch_posts_used[1u - pp];
            e->batch_posts_used[pp] = need;
            e->batch_posts_used[1u - pp] = 0;   // (reset by this launch)
        }
        const unsigned int parity = e->batch_launches & 1u;
        e->batch_launches++;
        // this launch's arrival counters start at zero whatever the task counts of earlier launches were (a launch
        // only clears the first m counters of the other parity, so a smaller batch in between leaves the rest dirty)
        HIPCHK(hipMemsetAsync(e->d_batch_barriers + (size_t)parity * e->batch_cap, 0, sizeof(unsigned int) * (size_t)m,
                              e->stream));
        HIPCHK(launch_newton_batch(e, e->d_batch_tasks, m, G, e->d_theta_batch, theta_stride, e->d_batch_partials,
                                   e->d_batch_barriers + (size_t)parity * e->batch_cap,
                                   e->d_batch_barriers + (size_t)(1u - parity) * e->batch_cap, e->d_batch_out, posts, posts_other,
                                   posts_other_used, post_epochs));
        HIPCHK(hipMemcpyAsync(out.data(), e->d_batch_out, sizeof(double) * 6 * (size_t)m, hipMemcpyDeviceToHost,
                              e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));  // also: host_tasks / out are reused by the next chunk
        for (int t = 0; t < m; t++) {
            const double *o = &out[(size_t)t * 6];
            iqhip_branch_result &r = results[first + t];
            r.optx = o[0];
            r.d2l = o[1];
            r.nsteps = (int)o[2];
            r.status = (int)o[3];
            r.lnl = o[4];
            if (r.status == 4) {
                // the task's grid barrier gave up (its workgroups were not co-resident): redo this one task with the
                // barrier-free chain form -- its node updates have run, so only theta + the solve + lnL remain
                const iqhip_branch_task &k = tasks[first + t];
                (void)hipStreamSynchronize(e->stream);
                rc = iqhip_compute_theta(e, k.a, k.b);
                if (!rc) rc = newton_chain(e, k.xguess, k.x1, k.x2, k.xacc, k.max_steps, &r.optx, &r.d2l, &r.nsteps);
                if (!rc) rc = iqhip_lnl_from_theta(e, r.optx, &r.lnl);
                if (rc) return rc;
                r.status = 0;
            }
        }
    }
    if (total_ops > 0) {
        rc = read_result(e, 2 + total_ops);
        if (rc) return rc;
        if (sum_scale)
            for (int k = 0; k < total_ops; k++) sum_scale[k] = e->h_result[2 + k];
    }
    return IQHIP_OK;
}

extern "C" int iqhip_debug_cherry_tables(iqhip_engine *e, int64_t *tables_built, int64_t *ops_from_tables) {
    if (!e) return fail(IQHIP_ERR_INVALID, "null engine");
    if (tables_built) *tables_built = e->cherry_built_total;
    if (ops_from_tables) *ops_from_tables = e->cherry_ops_total;
    return IQHIP_OK;
}

extern "C" int iqhip_lnl_from_theta_async(iqhip_engine *e, double len) {
    if (e && !e->shards.empty())
        return fail(IQHIP_ERR_UNSUPPORTED, "a sharded engine reduces its results itself: use the synchronous calls");
    int rc = check_ready(e);
    if (rc) return rc;
    if (!e->theta_valid) return fail(IQHIP_ERR_INVALID, "iqhip_lnl_from_theta: theta not computed");
    if (!(len >= 0.0)) return fail(IQHIP_ERR_INVALID, "negative or NaN branch length");
    rc = ensure_slab_rows(e, 2);
    if (rc) return rc;
    const int nwaves = (int)e->ntiles;
    if (e->mfma) HIPCHK(launch_stream_mfma(e, 3, nullptr, len, nwaves));
    else HIPCHK(launch_lnl_theta4(e, len, nwaves));
    HIPCHK(launch_reduce(e, 0, 2, nwaves));
    return IQHIP_OK;
}

extern "C" int iqhip_lnl_from_theta(iqhip_engine *e, double len, double *lnl) {
    if (e && !e->shards.empty()) return sharded::lnl_from_theta(e, len, lnl);
    int rc = iqhip_lnl_from_theta_async(e, len);
    if (rc) return rc;
    rc = comm_allreduce(e, e->asc_active ? 2 : 1);
    if (rc) return rc;
    rc = read_result(e, 2);
    if (rc) return rc;
    double v = e->h_result[0];
    if (isnan(v) || isinf(v)) {
        rc = repair_lnl_comm(e, &v);
        if (rc) return rc;
    }
    rc = asc_finish_lnl(e, &v);
    if (rc) return rc;
    if (lnl) *lnl = v;
    return IQHIP_OK;
}

// ---------------------------------------------------------------------------------------
// result buffer / sync
// ---------------------------------------------------------------------------------------
extern "C" int iqhip_bind_result_buffer(iqhip_engine *e, void *device_ptr, int capacity_doubles) {
    if (!e) return fail(IQHIP_ERR_INVALID, "null engine");
    if (!e->shards.empty() || e->comm)
        return fail(IQHIP_ERR_UNSUPPORTED, "an engine with a communicator reduces in its own device result vector");
    HIPCHK(use_device(e));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (!device_ptr) {
        e->d_result = e->d_result_own;
        e->result_cap = 8 + 16384;
        return IQHIP_OK;
    }
    if (capacity_doubles < 2) return fail(IQHIP_ERR_INVALID, "result buffer too small");
    e->d_result = (double *)device_ptr;
    e->result_cap = std::min(capacity_doubles, 8 + 16384);
    return IQHIP_OK;
}
extern "C" void *iqhip_result_device_ptr(iqhip_engine *e) { return (e && e->shards.empty()) ? (void *)e->d_result : nullptr; }
extern "C" int iqhip_result_capacity(iqhip_engine *e) { return e ? e->result_cap : 0; }

extern "C" int iqhip_result_read(iqhip_engine *e, double *out, int ndoubles) {
    if (e && !e->shards.empty())
        return fail(IQHIP_ERR_UNSUPPORTED, "a sharded engine reduces its results itself: use the synchronous calls");
    if (!e || !out || ndoubles < 0 || ndoubles > e->result_cap)
        return fail(IQHIP_ERR_INVALID, "iqhip_result_read: bad arguments");
    HIPCHK(use_device(e));
    int rc = read_result(e, ndoubles);
    if (rc) return rc;
    memcpy(out, e->h_result, sizeof(double) * ndoubles);
    return IQHIP_OK;
}

extern "C" int iqhip_synchronize(iqhip_engine *e) {
    if (!e) return fail(IQHIP_ERR_INVALID, "null engine");
    if (!e->shards.empty()) return sharded::synchronize(e);
    HIPCHK(use_device(e));
    HIPCHK(hipStreamSynchronize(e->stream));
    e->staging_busy = false;
    return IQHIP_OK;
}

// ---------------------------------------------------------------------------------------
// host views (layout conversion on the host; these are off the hot path)
// ---------------------------------------------------------------------------------------
static inline size_t dev_index(const iqhip_engine *e, int64_t p, int k) {
    const int B = e->block;
    if (e->mfma) return (size_t)(p >> 4) * 16 * B + (size_t)k * 16 + (size_t)(p & 15);
    return (size_t)(p >> 6) * 64 * B + (size_t)(k >> 1) * 128 + (size_t)(p & 63) * 2 + (k & 1);
}

static int fetch_vec(iqhip_engine *e, const double *dptr, double *out) {
    std::vector<double> tmp((size_t)e->nptn_pad * e->block);
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(tmp.data(), dptr, tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
    const int B = e->block;
    if (e->embed2) {  // the caller's block is n_user doubles per category: the first components of the embedded vector
        const int m = e->n_user, n = e->n;
        for (int64_t p = 0; p < e->nptn; p++)
            for (int c = 0; c < e->ncat; c++)
                for (int i = 0; i < m; i++) out[((size_t)p * e->ncat + c) * m + i] = tmp[dev_index(e, p, c * n + i)];
        return IQHIP_OK;
    }
    for (int64_t p = 0; p < e->nptn; p++)
        for (int k = 0; k < B; k++) out[(size_t)p * B + k] = tmp[dev_index(e, p, k)];
    return IQHIP_OK;
}

extern "C" int iqhip_fetch_partial(iqhip_engine *e, uint64_t key, double *out) {
    if (!e || !out) return fail(IQHIP_ERR_INVALID, "null argument");
    if (!e->shards.empty()) return sharded::fetch_vec(e, key, false, out);
    HIPCHK(use_device(e));
    int idx;
    int rc = slab_for_key(e, key, false, &idx);
    if (rc) return rc;
    return fetch_vec(e, e->slabs[idx].plh, out);
}

extern "C" int iqhip_fetch_theta(iqhip_engine *e, double *out) {
    if (!e || !out) return fail(IQHIP_ERR_INVALID, "null argument");
    if (!e->shards.empty()) return sharded::fetch_vec(e, 0, true, out);
    HIPCHK(use_device(e));
    return fetch_vec(e, e->d_theta, out);
}

extern "C" int iqhip_fetch_scale_num(iqhip_engine *e, uint64_t key, int16_t *out) {
    if (!e || !out) return fail(IQHIP_ERR_INVALID, "null argument");
    if (!e->shards.empty()) return sharded::fetch_scale_num(e, key, out);
    HIPCHK(use_device(e));
    int idx;
    int rc = slab_for_key(e, key, false, &idx);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(out, e->slabs[idx].sc, sizeof(int16_t) * (size_t)e->nptn, hipMemcpyDeviceToHost));
    return IQHIP_OK;
}

extern "C" int iqhip_fetch_pattern_lh(iqhip_engine *e, double *out) {
    if (!e || !out) return fail(IQHIP_ERR_INVALID, "null argument");
    if (!e->shards.empty()) return sharded::fetch_pattern_lh(e, out, 0, iqhip_branch_end{0, -1, 0}, iqhip_branch_end{0, -1, 0});
    HIPCHK(use_device(e));
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(out, e->d_pattern_lh, sizeof(double) * (size_t)e->nptn, hipMemcpyDeviceToHost));
    if (e->asc_active) {  // phylokernel.h:1013-1014: observed patterns only
        const int64_t nobs = e->nptn - e->n_unobs;
        for (int64_t p = 0; p < nobs; p++) out[p] -= e->pattern_lh_shift;
        for (int64_t p = nobs; p < e->nptn; p++) out[p] = 0.0;
    }
    return IQHIP_OK;
}

// ---- consumers of the device-resident pattern lnL (kernels_rell.hip) ---------------------------
static int scaled_pattern_lh(iqhip_engine *e, iqhip_branch_end a, iqhip_branch_end b) {
    const int16_t *sc[2] = {nullptr, nullptr};
    const iqhip_branch_end ends[2] = {a, b};
    for (int k = 0; k < 2; k++) {
        if (ends[k].leaf >= 0) continue;  // leaves carry no scaling events
        int idx;
        int rc = slab_for_key(e, ends[k].key, false, &idx);
        if (rc) return rc;
        sc[k] = e->slabs[idx].sc;
    }
    if (!e->d_ptn_scaled) HIPCHK(hipMalloc((void **)&e->d_ptn_scaled, sizeof(double) * (size_t)e->nptn_pad));
    HIPCHK(launch_pattern_lh_scaled(e, sc[0], sc[1], e->d_ptn_scaled));
    return IQHIP_OK;
}

extern "C" int iqhip_fetch_pattern_lh_scaled(iqhip_engine *e, iqhip_branch_end a, iqhip_branch_end b, double *out) {
    if (!e || !out) return fail(IQHIP_ERR_INVALID, "null argument");
    if (!e->shards.empty()) return sharded::fetch_pattern_lh(e, out, 1, a, b);
    HIPCHK(use_device(e));
    int rc = scaled_pattern_lh(e, a, b);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(out, e->d_ptn_scaled, sizeof(double) * (size_t)e->nptn, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return IQHIP_OK;
}

extern "C" int iqhip_pattern_lh_cat(iqhip_engine *e, double len, double *out) {
    if (!e || !out) return fail(IQHIP_ERR_INVALID, "null argument");
    if (!e->shards.empty()) return sharded::pattern_lh_cat(e, len, out);
    if (!e->theta_valid) return fail(IQHIP_ERR_INVALID, "iqhip_pattern_lh_cat needs iqhip_compute_theta first");
    if (!(len >= 0.0)) return fail(IQHIP_ERR_INVALID, "negative or NaN branch length");
    HIPCHK(use_device(e));
    const size_t count = (size_t)e->nptn * e->ncat;
    double *d_out = nullptr;
    HIPCHK(hipMalloc((void **)&d_out, sizeof(double) * count));
    hipError_t s = launch_pattern_lh_cat(e, len, d_out);
    if (s == hipSuccess) s = hipMemcpyAsync(out, d_out, sizeof(double) * count, hipMemcpyDeviceToHost, e->stream);
    if (s == hipSuccess) s = hipStreamSynchronize(e->stream);
    hipFree(d_out);
    if (s != hipSuccess) return fail(IQHIP_ERR_HIP, hipGetErrorString(s));
    return IQHIP_OK;
}

extern "C" int iqhip_set_boot_samples(iqhip_engine *e, const float *samples, int nsamples) {
    if (!e || (nsamples > 0 && !samples) || nsamples < 0) return fail(IQHIP_ERR_INVALID, "bad bootstrap samples");
    if (nsamples > 16384) return fail(IQHIP_ERR_INVALID, "at most 16384 bootstrap samples");
    if (!e->shards.empty()) return sharded::set_boot_samples(e, samples, nsamples);
    HIPCHK(use_device(e));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (e->d_boot) HIPCHK(hipFree(e->d_boot));
    e->d_boot = nullptr;
    e->nboot = 0;
    if (nsamples == 0) return IQHIP_OK;
    const size_t pitch = (size_t)e->nptn_pad;
    HIPCHK(hipMalloc((void **)&e->d_boot, sizeof(float) * pitch * nsamples));
    HIPCHK(hipMemset(e->d_boot, 0, sizeof(float) * pitch * nsamples));
    HIPCHK(hipMemcpy2D(e->d_boot, pitch * sizeof(float), samples, (size_t)e->nptn * sizeof(float),
                       (size_t)e->nptn * sizeof(float), nsamples, hipMemcpyHostToDevice));
    e->nboot = nsamples;
    return IQHIP_OK;
}

extern "C" int iqhip_rell_async(iqhip_engine *e, iqhip_branch_end a, iqhip_branch_end b) {
    if (!e) return fail(IQHIP_ERR_INVALID, "null engine");
    if (!e->shards.empty())
        return fail(IQHIP_ERR_UNSUPPORTED, "a sharded engine reduces its results itself: use the synchronous calls");
    if (e->nboot == 0) return fail(IQHIP_ERR_INVALID, "no bootstrap samples (iqhip_set_boot_samples)");
    if (e->nboot > e->result_cap) return fail(IQHIP_ERR_INVALID, "result buffer too small for the sample count");
    HIPCHK(use_device(e));
    int rc = scaled_pattern_lh(e, a, b);
    if (rc) return rc;
    HIPCHK(launch_rell(e, e->d_result));
    return IQHIP_OK;
}

extern "C" int iqhip_rell(iqhip_engine *e, iqhip_branch_end a, iqhip_branch_end b, double *rell) {
    if (!rell) return fail(IQHIP_ERR_INVALID, "null argument");
    if (e && !e->shards.empty()) return sharded::rell(e, a, b, rell);
    int rc = iqhip_rell_async(e, a, b);
    if (rc) return rc;
    rc = comm_allreduce(e, e->nboot);
    if (rc) return rc;
    rc = read_result(e, e->nboot);
    if (rc) return rc;
    memcpy(rell, e->h_result, sizeof(double) * (size_t)e->nboot);
    return IQHIP_OK;
}

extern "C" int iqhip_upload_partial(iqhip_engine *e, uint64_t key, const double *partial_lh,
                                    const int16_t *scale_num) {
    if (!e || !partial_lh || !scale_num) return fail(IQHIP_ERR_INVALID, "null argument");
    if (!e->shards.empty()) return sharded::upload_partial(e, key, partial_lh, scale_num);
    HIPCHK(use_device(e));
    int idx;
    int rc = slab_for_key(e, key, true, &idx);
    if (rc) return rc;
    const int B = e->block;
    std::vector<double> tmp((size_t)e->nptn_pad * B, 0.0);
    if (e->embed2) {
        const int m = e->n_user, n = e->n;
        for (int64_t p = 0; p < e->nptn; p++)
            for (int c = 0; c < e->ncat; c++)
                for (int i = 0; i < m; i++) tmp[dev_index(e, p, c * n + i)] = partial_lh[((size_t)p * e->ncat + c) * m + i];
    } else
    for (int64_t p = 0; p < e->nptn; p++)
        for (int k = 0; k < B; k++) tmp[dev_index(e, p, k)] = partial_lh[(size_t)p * B + k];
    std::vector<int16_t> sc((size_t)e->nptn_pad, 0);
    memcpy(sc.data(), scale_num, sizeof(int16_t) * (size_t)e->nptn);
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(e->slabs[idx].plh, tmp.data(), tmp.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->slabs[idx].sc, sc.data(), sc.size() * sizeof(int16_t), hipMemcpyHostToDevice));
    return IQHIP_OK;
}

// ---------------------------------------------------------------------------------------
// timing of the dominant kernel (HIP events on the launch stream)
// ---------------------------------------------------------------------------------------
extern "C" int iqhip_timing_enable(iqhip_engine *e, int on) {
    if (!e) return fail(IQHIP_ERR_INVALID, "null engine");
    if (!e->shards.empty()) {
        for (iqhip_engine *c : e->shards) c->timing = on != 0;
        return IQHIP_OK;
    }
    e->timing = on != 0;
    return IQHIP_OK;
}

// average duration (us) of the engine's own all-reduces since the last reset, HIP events on its stream (comm.hip)
extern "C" int iqhip_timing_collective_read(iqhip_engine *e, double *avg_us, int64_t *count, int reset) {
    if (!e) return fail(IQHIP_ERR_INVALID, "null engine");
    if (!e->shards.empty()) e = e->shards[0];   // (grouped all-reduce of a single-process front: not bracketed; reads 0)
    HIPCHK(use_device(e));
    HIPCHK(hipStreamSynchronize(e->stream));
    double total = 0.0;
    for (size_t i = 0; i < e->cev_used; i++) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, e->cev[i].first, e->cev[i].second));
        total += ms;
    }
    if (avg_us) *avg_us = e->cev_used ? total * 1e3 / (double)e->cev_used : 0.0;
    if (count) *count = (int64_t)e->cev_used;
    if (reset) e->cev_used = 0;
    return IQHIP_OK;
}

// bytes the last submission's traversal launches ask the memory system for, from its descriptors: every result vector
// and its counters stored once; the children that are neither the previous result nor parked (streamed / second memory
// child) loaded once each; leaf state rows; the root-branch pass (its vector unless it is the previous result, ptn_freq,
// ptn_invar, _pattern_lh).  Loads of vectors written earlier in the same launch may be served by the L2 / Infinity Cache,
// so `loaded` bounds the fabric reads from above; `stored` is exact.
extern "C" int iqhip_timing_plan_bytes(iqhip_engine *e, double *stored, double *loaded) {
    if (!e) return fail(IQHIP_ERR_INVALID, "null engine");
    double st = 0.0, ld = 0.0;
    if (!e->shards.empty()) {
        for (iqhip_engine *c : e->shards) {
            double a = 0.0, b = 0.0;
            int rc = iqhip_timing_plan_bytes(c, &a, &b);
            if (rc) return rc;
            st += a; ld += b;
        }
    } else {
        const double P = (double)e->nptn_pad, V = (double)e->block * 8.0;
        for (int k = 0; k < e->last_nops && e->h_ops; k++) {
            const DevOp &d = e->h_ops[k];
            st += P * (V + 2.0);
            if (d.left_kind == CHILD_LEAF) ld += P; else if (d.left_kind == CHILD_PF || d.left_kind == CHILD_LOAD) ld += P * (V + 2.0);
            if (d.right_kind == CHILD_LEAF) ld += P; else if (d.right_kind == CHILD_LOAD) ld += P * (V + 2.0);
        }
        if (e->last_has_root) {
            st += P * 8.0;
            ld += P * 16.0 + (e->last_root_loads_b ? P * V : 0.0) + P;
        }
    }
    if (stored) *stored = st;
    if (loaded) *loaded = ld;
    return IQHIP_OK;
}

extern "C" int iqhip_timing_read(iqhip_engine *e, double *avg_ms, int64_t *launches, int reset) {
    if (!e) return fail(IQHIP_ERR_INVALID, "null engine");
    if (!e->shards.empty()) {  // the slowest shard's average; launches of shard 0
        double worst = 0.0;
        int64_t n0 = 0;
        for (size_t g = 0; g < e->shards.size(); g++) {
            double a = 0.0;
            int64_t n = 0;
            int rc = iqhip_timing_read(e->shards[g], &a, &n, reset);
            if (rc) return rc;
            if (a > worst) worst = a;
            if (g == 0) n0 = n;
        }
        if (avg_ms) *avg_ms = worst;
        if (launches) *launches = n0;
        return IQHIP_OK;
    }
    HIPCHK(use_device(e));
    HIPCHK(hipStreamSynchronize(e->stream));
    double total = 0.0;
    for (size_t i = 0; i < e->tev_used; i++) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, e->tev[i].first, e->tev[i].second));
        total += ms;
    }
    // a staged plan is two launches of the traversal kernel inside one bracket: report per launch, as a
    // profiler's per-kernel average does
    if (avg_ms) *avg_ms = e->tev_launches ? total / (double)e->tev_launches : 0.0;
    if (launches) *launches = e->tev_launches;
    if (reset) { e->tev_used = 0; e->tev_launches = 0; }
    return IQHIP_OK;
}

// ---------------------------------------------------------------------------------------
// planning-only engine (CPU tests of build_plan + check_plan; no HIP call)
// ---------------------------------------------------------------------------------------
extern "C" int iqhip_debug_create_planner(iqhip_engine **out, int nstates, int ncat, int64_t nptn, int ntaxa, int num_cus,
                                          int state_unknown, int nclass) {
    if (!out) return fail(IQHIP_ERR_INVALID, "iqhip_debug_create_planner: out == NULL");
    *out = nullptr;
    if (nptn <= 0 || ntaxa < 2 || ncat < 1 || num_cus < 1 || nclass < 1 || nclass > ncat)
        return fail(IQHIP_ERR_INVALID, "iqhip_debug_create_planner: bad shape");
    if (nstates != 4 && nstates != 20 && nstates != 64)
        return fail(IQHIP_ERR_UNSUPPORTED, "iqhip_debug_create_planner: nstates must be 4, 20 or 64");
    if (state_unknown < nstates || state_unknown > 255) return fail(IQHIP_ERR_INVALID, "state_unknown out of range");
    iqhip_engine *e = new iqhip_engine();
    e->planner = true;
    e->check_plans = true;
    e->num_cus = num_cus;
    configure_engine(e, -1, nstates, nstates, ncat, nptn, ntaxa);
    const size_t P = (size_t)e->nptn_pad;
    e->d_states = fake_alloc<uint8_t>(e, (size_t)ntaxa * P);
    e->dummy.plh = fake_alloc<double>(e, P * e->block);
    e->dummy.sc = fake_alloc<int16_t>(e, P);
    e->result_cap = 8 + 16384;
    e->state_unknown = state_unknown;
    e->nclass = nclass;
    if (nclass > 1 && nstates == 4) {   // as set_model_common: a 4-state mixture runs on the matrix-core kernels
        e->mfma = true;
        e->lane_split = 1;
        e->tile = 16;
        e->ntiles = e->nptn_pad / 16;
    }
    e->mfma_pipelined = e->mfma_pipelined_ok && nclass == 1;
    e->model_set = e->aln_set = true;
    {   // cherry tables as cherry_sync_model would set them up
        const int s2 = state_unknown + 1;
        if (cherry_candidate(e) && nclass == 1 && s2 * s2 <= 4356 && (int64_t)4 * s2 * s2 <= e->nptn_pad) {
            e->cherry_s2 = s2;
            e->cherry_npairs = (int)round_up(s2 * s2, 64);
        }
    }
    *out = e;
    return IQHIP_OK;
}

// build the descriptors of one submission exactly as iqhip_update_partials would and validate them (check_plan)
extern "C" int iqhip_debug_plan(iqhip_engine *e, const iqhip_node_op *ops, int nops) {
    if (!e || !e->planner) return fail(IQHIP_ERR_INVALID, "iqhip_debug_plan needs a planning-only engine");
    if (nops < 0 || (nops > 0 && !ops)) return fail(IQHIP_ERR_INVALID, "bad ops array");
    int last_dst = -1;
    const int rc = build_plan(e, ops, nops, &last_dst, nullptr);
    if (!rc) {   // (what submit_traverse would count: iqhip_debug_cherry_tables)
        e->cherry_built_total += (int64_t)e->plan_cherry_jobs.size();
        for (int slot : e->plan_cherry_jobs) e->cherry_slots[slot].model_version = e->model_version;   // ("built")
        e->plan_cherry_jobs.clear();
        for (int k = 0; k < nops; k++) e->cherry_ops_total += e->h_ops[k].cherry != nullptr;
    }
    return rc;
}
