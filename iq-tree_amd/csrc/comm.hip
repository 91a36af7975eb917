// comm.hip -- the one collective of the path: an in-place SUM / f64 all-reduce over RCCL of the
// handful of host-visible doubles an evaluation produces ({lnL | df, ddf | sum_scale per node update}).  Patterns
// shard contiguously over the GPUs, so nothing else ever crosses xGMI; the payload is <= a few KB, i.e. the
// collective is latency-bound and link bandwidth never matters.
//
// Two ways to get a communicator:
//   * one process per GPU (bench.py under torch.distributed.run, MPI programs): rank 0 calls
//     iqhip_comm_unique_id, the caller broadcasts the 128 bytes by whatever means it has, every rank calls
//     iqhip_comm_init_rank on its engine;
//   * one process, several GPUs (the reference is a single process, pda.cpp:2137): iqhip_create_sharded
//     (sharded.hip) makes one engine per device and calls comm_init_all (ncclCommInitAll).
// librccl is 0.5 GB: it is NOT a link-time dependency of libiqhip.so but is dlopen()ed the first time a
// communicator is made (a process that already holds a librccl.so.1, e.g. torch's, shares it).
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and enums only; the functions are resolved with dlsym
#include <string.h>

#include <mutex>

#include "iqhip_internal.h"

namespace iqhip {

namespace {
struct Rccl {
    void *handle = nullptr;
    std::string err;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl &rccl() {
    static Rccl R;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {getenv("IQHIP_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            if (!n || !*n) continue;
            // RTLD_LOCAL: librccl drags in librocm_smi64, whose global objects must not interpose with a second copy of that
            // library that the process may load later (torch ships its own): with RTLD_GLOBAL a test order that made this
            // dlopen come BEFORE `import torch` ended in a double free inside rocm_smi's static destructors at exit
            R.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (R.handle) break;
            R.err = dlerror();
        }
        if (!R.handle) return;
        bool all = true;
        auto sym = [&](const char *n) -> void * {
            void *p = dlsym(R.handle, n);
            if (!p) { all = false; R.err = std::string("librccl lacks ") + n; }
            return p;
        };
        R.GetUniqueId = reinterpret_cast<decltype(R.GetUniqueId)>(sym("ncclGetUniqueId"));
        R.CommInitRank = reinterpret_cast<decltype(R.CommInitRank)>(sym("ncclCommInitRank"));
        R.CommInitAll = reinterpret_cast<decltype(R.CommInitAll)>(sym("ncclCommInitAll"));
        R.CommDestroy = reinterpret_cast<decltype(R.CommDestroy)>(sym("ncclCommDestroy"));
        R.AllReduce = reinterpret_cast<decltype(R.AllReduce)>(sym("ncclAllReduce"));
        R.GroupStart = reinterpret_cast<decltype(R.GroupStart)>(sym("ncclGroupStart"));
        R.GroupEnd = reinterpret_cast<decltype(R.GroupEnd)>(sym("ncclGroupEnd"));
        R.GetErrorString = reinterpret_cast<decltype(R.GetErrorString)>(sym("ncclGetErrorString"));
        R.ok = all;
    });
    return R;
}

int need_rccl(Rccl **out) {
    Rccl &R = rccl();
    if (!R.ok) return set_error(IQHIP_ERR_UNSUPPORTED, "RCCL is not available: " + R.err);
    *out = &R;
    return IQHIP_OK;
}

int nccl_fail(Rccl &R, const char *what, ncclResult_t s) {
    return set_error(IQHIP_ERR_HIP, std::string(what) + ": " + (R.GetErrorString ? R.GetErrorString(s) : "RCCL error"));
}
}  // namespace

// RCCL reduces device memory; the engine's default result vector is mapped host memory (written directly by
// k_reduce so that the single-GPU path has no D2H copy).  A comm engine gets a device vector instead and copies
// the few doubles it needs back after the all-reduce (read_result's "caller-bound buffer" path).
int comm_use_device_result(iqhip_engine *e) {
    if (e->d_result_dev) return IQHIP_OK;
    if (e->d_result != e->d_result_own)
        return set_error(IQHIP_ERR_INVALID, "a caller-bound result buffer and an engine communicator exclude each other");
    if (hipSetDevice(e->device) != hipSuccess || hipStreamSynchronize(e->stream) != hipSuccess ||
        hipMalloc((void **)&e->d_result_dev, sizeof(double) * (size_t)e->result_cap) != hipSuccess)
        return set_error(IQHIP_ERR_NOMEM, "device result vector");
    hipMemsetAsync(e->d_result_dev, 0, sizeof(double) * (size_t)e->result_cap, e->stream);
    e->d_result = e->d_result_dev;
    return IQHIP_OK;
}

int comm_allreduce(iqhip_engine *e, int n) {
    if (!e->comm || n <= 0) return IQHIP_OK;
    Rccl *R;
    int rc = need_rccl(&R);
    if (rc) return rc;
    // measurement hook (iqhip_timing_enable): HIP events on the engine's stream around the collective -- from the moment
    // the stream reaches it (the kernels before it have finished) to its completion, i.e. incl. the wait for slower ranks
    const bool timed = e->timing;
    if (timed) {
        if (e->cev_used == e->cev.size()) {
            hipEvent_t a, b;
            hipEventCreate(&a);
            hipEventCreate(&b);
            e->cev.emplace_back(a, b);
        }
        hipEventRecord(e->cev[e->cev_used].first, e->stream);
    }
    ncclResult_t s = R->AllReduce(e->d_result, e->d_result, (size_t)n, ncclDouble, ncclSum, (ncclComm_t)e->comm, e->stream);
    if (timed) {
        hipEventRecord(e->cev[e->cev_used].second, e->stream);
        e->cev_used++;
    }
    if (s != ncclSuccess) return nccl_fail(*R, "ncclAllReduce", s);
    return IQHIP_OK;
}

int comm_group_allreduce(const std::vector<iqhip_engine *> &shards, int n) {
    if (n <= 0 || shards.empty() || !shards[0]->comm) return IQHIP_OK;
    Rccl *R;
    int rc = need_rccl(&R);
    if (rc) return rc;
    ncclResult_t s = R->GroupStart();
    if (s != ncclSuccess) return nccl_fail(*R, "ncclGroupStart", s);
    ncclResult_t bad = ncclSuccess;
    for (iqhip_engine *c : shards) {
        s = R->AllReduce(c->d_result, c->d_result, (size_t)n, ncclDouble, ncclSum, (ncclComm_t)c->comm, c->stream);
        if (s != ncclSuccess) bad = s;
    }
    s = R->GroupEnd();
    if (bad != ncclSuccess) return nccl_fail(*R, "ncclAllReduce", bad);
    if (s != ncclSuccess) return nccl_fail(*R, "ncclGroupEnd", s);
    return IQHIP_OK;
}

int comm_init_all(const std::vector<iqhip_engine *> &shards) {
    Rccl *R;
    int rc = need_rccl(&R);
    if (rc) return rc;
    const int ndev = (int)shards.size();
    std::vector<int> devs(ndev);
    for (int g = 0; g < ndev; g++) devs[g] = shards[g]->device;
    for (int g = 0; g < ndev; g++)
        for (int h = g + 1; h < ndev; h++)
            if (devs[g] == devs[h])
                return set_error(IQHIP_ERR_INVALID, "RCCL needs distinct devices (shards that share a GPU use IQHIP_REDUCE_HOST)");
    std::vector<ncclComm_t> comms(ndev, nullptr);
    ncclResult_t s = R->CommInitAll(comms.data(), ndev, devs.data());
    if (s != ncclSuccess) return nccl_fail(*R, "ncclCommInitAll", s);
    for (int g = 0; g < ndev; g++) {
        shards[g]->comm = comms[g];
        shards[g]->comm_nranks = ndev;
        shards[g]->comm_rank = g;
        rc = comm_use_device_result(shards[g]);
        if (rc) return rc;
    }
    return IQHIP_OK;
}

void comm_destroy(iqhip_engine *e) {
    if (!e->comm) return;
    Rccl &R = rccl();
    if (R.ok) R.CommDestroy((ncclComm_t)e->comm);
    e->comm = nullptr;
    e->comm_nranks = 1;
    e->comm_rank = 0;
}

}  // namespace iqhip

using namespace iqhip;

extern "C" int iqhip_comm_unique_id(void *id_out) {
    if (!id_out) return set_error(IQHIP_ERR_INVALID, "null argument");
    Rccl *R;
    int rc = need_rccl(&R);
    if (rc) return rc;
    static_assert(sizeof(ncclUniqueId) == IQHIP_COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    ncclResult_t s = R->GetUniqueId(&id);
    if (s != ncclSuccess) return nccl_fail(*R, "ncclGetUniqueId", s);
    memcpy(id_out, &id, sizeof id);
    return IQHIP_OK;
}

extern "C" int iqhip_comm_init_rank(iqhip_engine *e, int nranks, int rank, const void *id_in) {
    if (!e || !id_in) return set_error(IQHIP_ERR_INVALID, "null argument");
    if (!e->shards.empty()) return set_error(IQHIP_ERR_INVALID, "a sharded engine has its own communicators");
    if (nranks < 1 || rank < 0 || rank >= nranks) return set_error(IQHIP_ERR_INVALID, "bad rank / rank count");
    if (e->comm) return set_error(IQHIP_ERR_INVALID, "engine already has a communicator");
    Rccl *R;
    int rc = need_rccl(&R);
    if (rc) return rc;
    if (hipSetDevice(e->device) != hipSuccess) return set_error(IQHIP_ERR_HIP, "hipSetDevice");
    rc = comm_use_device_result(e);
    if (rc) return rc;
    ncclUniqueId id;
    memcpy(&id, id_in, sizeof id);
    ncclComm_t c = nullptr;
    ncclResult_t s = R->CommInitRank(&c, nranks, id, rank);
    if (s != ncclSuccess) return nccl_fail(*R, "ncclCommInitRank", s);
    e->comm = c;
    e->comm_nranks = nranks;
    e->comm_rank = rank;
    return IQHIP_OK;
}

extern "C" int iqhip_comm_size(iqhip_engine *e) {
    if (!e) return 0;
    if (!e->shards.empty()) return (int)e->shards.size();
    return e->comm_nranks;
}
