#!/usr/bin/env python3
"""bench.py -- million pattern-node partial-likelihood updates / s on MI355X.

Step  = the reference's hot loop 1: clearAllPartialLH(); computeLikelihood()  (SURVEY.md 3A/8d):
        invalidate everything, full post-order traversal (ntaxa-2 node updates), root-branch lnL.
        Driven through the host mirror -> C ABI (include/iqhip.h) -> HIP kernels.
Value = steps * (ntaxa-2) * patterns(all ranks) / wall / 1e6   (internal nodes only).
N=1   : BASELINE.json configs[1]: synthetic DNA 50 taxa x 100k patterns, GTR+G4.
N>1   : weak scaling -- every rank holds its own 100k-pattern shard of one (N*100k)-pattern
        alignment on the same tree; one RCCL all-reduce (SUM, f64) of the device result vector
        {lnL, sum_scale per node} per step.
Inputs are resident in HBM before the timed region.  One JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes_per_traversal(ntaxa, nptn, block):
    """P*[(2T-4)*V + (2T-4)*2 + T*1 + 8 + 16], V = block*8."""
    V = block * 8
    return nptn * ((2 * ntaxa - 4) * V + (2 * ntaxa - 4) * 2 + ntaxa + 8 + 16)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--ntaxa", type=int, default=50)
    ap.add_argument("--patterns", type=int, default=100000, help="patterns per GPU")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the likelihood path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    pkg = entry.load_package()
    import importlib
    synth = importlib.import_module("iqtree_amd.synth")
    lib = pkg.libiqhip()

    T, P = args.ntaxa, args.patterns
    model = synth.gtr_model(rates6=(1.5, 2.4, 1.8, 1.9, 2.8, 1.0), freqs=(0.25, 0.26, 0.25, 0.24),
                            alpha=0.9, ncat=4)
    # same tree on every rank (seed 1); each rank simulates its own shard of sites
    nwk = synth.random_tree_newick(T, 1)
    nsites = int(P * 1.02) + 64
    while True:
        st = synth.simulate_alignment(nwk, model, nsites, 1000 + rank)
        pat, freq = synth.compress_patterns(st)
        if pat.shape[1] >= P:
            break
        nsites = int(nsites * 1.3)
    pat = np.ascontiguousarray(pat[:, :P])
    freq = freq[:P].copy()

    tree = pkg.PhyloTree(nwk)
    tree.set_alignment(4, pkg.SEQ_DNA, pat, freq)
    tree.set_model(model)
    tree.set_likelihood_kernel(pkg.LK_EIGEN_HIP)
    tree.attach_engine(local_rank)
    eng = tree.engine
    stream = torch.cuda.current_stream()
    assert lib.iqhip_set_stream(eng, C.c_void_p(stream.cuda_stream)) == 0
    res = torch.zeros(2 + 4096, dtype=torch.float64, device="cuda")
    assert lib.iqhip_bind_result_buffer(eng, C.c_void_p(res.data_ptr()), res.numel()) == 0
    if world > 1:
        def hook(ptr, n):
            assert ptr == res.data_ptr()
            dist.all_reduce(res[:n], op=dist.ReduceOp.SUM)
        tree.set_allreduce_hook(hook)

    def step():
        tree.clear_all_partial_lh()
        return tree.compute_likelihood()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        lnl = step()
    lib.iqhip_timing_enable(eng, 1)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        lnl = step()
    barrier()
    dt = time.perf_counter() - t0
    avg_ms, launches = C.c_double(), C.c_int64()
    lib.iqhip_timing_read(eng, C.byref(avg_ms), C.byref(launches), 1)
    lib.iqhip_timing_enable(eng, 0)
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    updates = args.steps * (T - 2) * P * world
    value = updates / dt / 1e6
    block = 4 * model.ncat
    algo_bytes = algorithmic_bytes_per_traversal(T, P, block)
    kern_s = avg_ms.value * 1e-3
    achieved = algo_bytes / kern_s / 1e9 if kern_s > 0 else 0.0

    out = {
        "metric": "million pattern-node partial-likelihood updates/sec",
        "value": value,
        "unit": "M updates/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "DNA %d taxa x %d patterns/GPU, GTR+G4, fixed tree: clearAllPartialLH + "
                               "full traversal + root-branch lnL" % (T, P),
                   "ntaxa": T, "patterns_per_gpu": P, "nstates": 4, "ncat": 4,
                   "parallelism": "patterns sharded over %d GPU(s), 1 RCCL all-reduce/step" % world},
        "lnL": lnl,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "kernel": "k_traverse4<4>", "kernel_avg_ms": avg_ms.value,
                     "launches": launches.value, "algorithmic_bytes_per_launch": algo_bytes},
    }

    if rank == 0 and not args.no_cpu_baseline:
        od = entry.load_oracle()
        sample = min(P, 20000)
        ot = od.OracleTree(nwk, 4, od.SEQ_DNA, pat[:, :sample], freq[:sample], None, model)
        mups, reps, secs = ot.time_traversals(budget_s=args.cpu_seconds)
        # parity of the timed configuration itself, on the sample
        olnl, _ = ot.likelihood()
        out["cpu_baseline"] = {"value": mups, "unit": "M updates/s", "cores": 1, "kind": "port",
                               "sample": "oracle/lh_oracle.c (gcc -O3 -mavx, 1 thread): %d traversals of the "
                                         "same tree on the first %d patterns in %.1f s" % (reps, sample, secs)}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
