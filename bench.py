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
# fp64 matrix peak: the guide's MFMA table has no f64 row; MI355X datasheet value,
# equal to the fp64 vector peak on this chip
MFMA_F64_PEAK_TFLOPS = 78.6


def algorithmic_bytes_per_traversal(ntaxa, nptn, block):
    """P*[(2T-4)*V + (2T-4)*2 + T*1 + 8 + 16], V = block*8."""
    V = block * 8
    return nptn * ((2 * ntaxa - 4) * V + (2 * ntaxa - 4) * 2 + ntaxa + 8 + 16)


def algorithmic_flops_per_traversal(ntaxa, nptn, n, ncat):
    """SURVEY.md 8(d) per-update flops summed over a traversal rooted at a leaf branch:
    T-2 updates each pay the U^-1 product + Hadamard (2n^2+n), T-3 internal children each pay one
    E*v product (2n^2); leaf children are table look-ups in the reference (0 flops)."""
    return nptn * ncat * ((ntaxa - 2) * (2 * n * n + n) + (ntaxa - 3) * 2 * n * n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", choices=["dna", "protein", "codon", "mixture"], default="dna",
                    help="dna = BASELINE configs[1] (the headline); protein/codon = configs[2]/[4] shapes")
    ap.add_argument("--ntaxa", type=int, default=0)
    ap.add_argument("--patterns", type=int, default=0, help="patterns per GPU")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-collective", action="store_true",
                    help="exercise the N>1 path (RCCL all-reduce of the result vector) even with one rank")
    ap.add_argument("--reference-order", action="store_true",
                    help="plan subtrees in the reference's neighbour order instead of heavier-first")
    ap.add_argument("--ncat", type=int, default=0, help="rate categories (dna / protein workloads; default: the BASELINE shape)")
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
    collective = world > 1 or args.force_collective
    real_stdout = os.dup(1)
    if collective:
        # RCCL prints a version banner on stdout at communicator creation: keep stdout clean for the
        # single JSON line by pointing fd 1 at stderr until the result is printed
        sys.stdout.flush()
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    pkg = entry.load_package()
    import importlib
    synth = importlib.import_module("iqtree_amd.synth")
    lib = pkg.libiqhip()

    shapes = {"dna": (50, 100000, 4, 4, pkg.SEQ_DNA), "protein": (100, 50000, 20, 4, pkg.SEQ_PROTEIN),
              "codon": (50, 20000, 64, 1, pkg.SEQ_CODON),
              # protein profile mixture x Gamma (C10+G4 shape: 10 classes x 4 rates = 40 components)
              "mixture": (50, 10000, 20, 40, pkg.SEQ_PROTEIN)}
    T0, P0, nst, ncat, seq_type = shapes[args.workload]
    if args.ncat and args.workload in ("dna", "protein"):
        ncat = args.ncat
    T, P = args.ntaxa or T0, args.patterns or P0
    sim_model = None
    if args.workload == "mixture":
        model = synth.mixture_model(20, 10, 7, alpha=0.9, ncat=4)
        sim_model = model.classes[0]
    elif nst == 4:
        model = synth.gtr_model(rates6=(1.5, 2.4, 1.8, 1.9, 2.8, 1.0), freqs=(0.25, 0.26, 0.25, 0.24),
                                alpha=0.9, ncat=ncat)
    else:
        # random reversible 20-/64-state model of the LG+G4 / GY shape (the reference's empirical
        # matrices are constants of its source and are not copied); codon: ncat = 1 as GY+F1X4
        model = synth.random_reversible_model(nst, 7, alpha=0.9 if ncat > 1 else None, ncat=ncat,
                                              min_freq=1e-4)
    # same tree on every rank (seed 1); each rank simulates its own shard of sites
    nwk = synth.random_tree_newick(T, 1)
    nsites = int(P * 1.02) + 64
    while True:
        st = synth.simulate_alignment(nwk, sim_model or model, nsites, 1000 + rank)
        pat, freq = synth.compress_patterns(st)
        if pat.shape[1] >= P:
            break
        nsites = int(nsites * 1.3)
    pat = np.ascontiguousarray(pat[:, :P])
    freq = freq[:P].copy()

    tree = pkg.PhyloTree(nwk)
    tree.set_alignment(nst, seq_type, pat, freq)
    tree.set_model(model)
    tree.set_likelihood_kernel(pkg.LK_EIGEN_HIP)
    if args.reference_order:
        tree.set_heavy_first(False)
    tree.attach_engine(local_rank)
    eng = tree.engine
    res = None
    if collective:
        # sharded run: the engine works on torch's current stream and leaves each result vector in a
        # torch-owned device buffer, which is all-reduced over RCCL before the single host read
        stream = torch.cuda.current_stream()
        assert lib.iqhip_set_stream(eng, C.c_void_p(stream.cuda_stream)) == 0
        res = torch.zeros(2 + 4096, dtype=torch.float64, device="cuda")
        assert lib.iqhip_bind_result_buffer(eng, C.c_void_p(res.data_ptr()), res.numel()) == 0

        def hook(ptr, n):
            assert ptr == res.data_ptr()
            dist.all_reduce(res[:n], op=dist.ReduceOp.SUM)
        tree.set_allreduce_hook(hook)

    def step():  # clearAllPartialLH(); computeLikelihood() on the C++ side of the boundary
        return tree.clear_and_compute_likelihood()

    def barrier():
        torch.cuda.synchronize()
        if collective:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        lnl = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        lnl = step()
    barrier()
    dt = time.perf_counter() - t0
    # dominant-kernel duration: HIP events on the launch stream around that kernel, measured live in
    # a second pass of the same steps (kept out of the timed region: two event records per step)
    lib.iqhip_timing_enable(eng, 1)
    ntimed = min(args.steps, 100)
    for _ in range(ntimed):
        step()
    avg_ms, launches = C.c_double(), C.c_int64()
    lib.iqhip_timing_read(eng, C.byref(avg_ms), C.byref(launches), 1)
    lib.iqhip_timing_enable(eng, 0)
    if collective:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    updates = args.steps * (T - 2) * P * world
    value = updates / dt / 1e6
    block = nst * model.ncat
    # one traversal = one launch of the traversal kernel, or two for a staged plan (independent subtrees,
    # then the ops above them): algorithmic work per launch = work per traversal / launches per traversal
    lpt = max(1.0, launches.value / float(ntimed))
    algo_bytes = algorithmic_bytes_per_traversal(T, P, block) / lpt
    kern_s = avg_ms.value * 1e-3
    achieved = algo_bytes / kern_s / 1e9 if kern_s > 0 else 0.0

    algo_flops = algorithmic_flops_per_traversal(T, P, nst, model.ncat) / lpt
    if nst == 64:
        tf = algo_flops / kern_s / 1e12 if kern_s > 0 else 0.0
        roof = {"bound": "mfma", "achieved": tf, "peak": MFMA_F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": tf / MFMA_F64_PEAK_TFLOPS, "traffic": None,
                "algorithmic_flops_per_launch": algo_flops, "algorithmic_GBps": achieved}
    else:
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "algorithmic_bytes_per_launch": algo_bytes,
                "algorithmic_TFLOPs": algo_flops / kern_s / 1e12 if kern_s > 0 else 0.0}
    # HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    # (tools/profile_round.sh -> profiles/traffic.json); null when the shape was not profiled
    try:
        tr = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(args.workload)
        if tr and tr["ntaxa"] == T and tr["patterns_per_gpu"] == P and tr["ncat"] == model.ncat:
            roof["traffic"] = tr["hbm_traffic_bytes_per_launch"]
            roof["traffic_source"] = tr["source"]
    except Exception:
        pass
    roof.update({"kernel": "k_traverse4<4,256>" if nst == 4 else "k_traverse_mfma2<%d,%d,256>" % (nst, model.ncat),
                 "kernel_avg_ms": avg_ms.value, "launches": launches.value, "launches_per_traversal": lpt,
                 "kernel_ms_per_traversal": avg_ms.value * lpt})
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
        "config": {"workload": "%s %d taxa x %d patterns/GPU, %s, fixed tree: clearAllPartialLH + "
                               "full traversal + root-branch lnL" % (args.workload.upper(), T, P, model.name),
                   "ntaxa": T, "patterns_per_gpu": P, "nstates": nst, "ncat": model.ncat,
                   "parallelism": "patterns sharded over %d GPU(s), 1 RCCL all-reduce/step" % world},
        "lnL": lnl,
        "roofline": roof,
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:  # the CPU baseline is timed at N = 1 only
        od = entry.load_oracle()
        sample = min(P, 100000 if nst == 4 else (4000 if model.ncat <= 4 else 1000))
        ot = od.OracleTree(nwk, nst, seq_type, pat[:, :sample], freq[:sample], None, model)
        try:
            ncores = min(len(os.sched_getaffinity(0)), 16)
        except AttributeError:
            ncores = min(os.cpu_count() or 1, 16)
        od.lib().oracle_set_threads(1)
        mups1, reps1, secs1 = ot.time_traversals(budget_s=args.cpu_seconds * 0.4)
        ncores = od.lib().oracle_set_threads(ncores)
        mups, reps, secs = ot.time_traversals(budget_s=args.cpu_seconds * 0.6)
        olnl, _ = ot.likelihood()
        out["cpu_baseline"] = {"value": mups, "unit": "M updates/s", "cores": ncores, "kind": "port",
                               "sample": "oracle/lh_oracle.c (gcc -O3 -mavx -fopenmp, pattern loop threaded as the "
                                         "reference's '#pragma omp parallel for'): %d traversals of the same tree on the "
                                         "first %d patterns in %.1f s with %d threads; 1 thread: %.2f M updates/s "
                                         "(%d traversals in %.1f s)" % (reps, sample, secs, ncores, mups1, reps1, secs1)}
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if collective:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
