#!/usr/bin/env python3
"""bench.py -- million pattern-node partial-likelihood updates / s on MI355X.

Step  = the reference's hot loop 1: clearAllPartialLH(); computeLikelihood()  (SURVEY.md 3A/8d):
        invalidate everything, full post-order traversal (ntaxa-2 node updates), root-branch lnL.
        Driven through the host mirror -> C ABI (include/iqhip.h) -> HIP kernels.
Value = steps * (ntaxa-2) * patterns(all ranks) / wall / 1e6   (internal nodes only).

Workloads (iq-tree_amd/synth.py BASELINE_SHAPES):
  dna      BASELINE configs[1]: DNA 50 taxa x 100k patterns per GPU, GTR+G4 -- the headline; weak scaling
           (every rank holds its own 100k-pattern shard of one N*100k-pattern alignment, same tree)
  protein  configs[2]: 20-state 100 x 50k +G4 (per GPU, weak)
  dna4     configs[3]: DNA 200 taxa x 1M patterns, pattern-sharded: 1M/N per GPU (strong scaling)
  codon    configs[4]: 64-state 50 x 20k, 20k/N per GPU (strong scaling)
One RCCL all-reduce (SUM, f64) of the device result vector {lnL, sum_scale per node} per step when N > 1
(SURVEY.md 8e).  Inputs are resident in HBM before the timed region.  One JSON line on rank 0; with the default
workload the line also carries the configs[2] / configs[3] / configs[4] results of the same N under "also" (--no-also skips), so
that the driver's N = 1, 2, 4, 8 runs of the default command give the strong-scaling curves north_star names too.

`python bench.py --gpus N` without WORLD_SIZE in the environment starts its N ranks itself (fresh child
processes through torch.distributed.run, before this process touches a GPU).
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_ACHIEVABLE_GBS = 6300.0  # ... "6.29 TB/s measured (float4 copy)": what floor_ms is priced against
# fp64 matrix peak: the guide's MFMA table has no f64 row; MI355X datasheet value,
# equal to the fp64 vector peak on this chip
MFMA_F64_PEAK_TFLOPS = 78.6
STRONG = ("dna4", "codon")  # total pattern count fixed, split over the ranks


def algorithmic_bytes_per_traversal(ntaxa, nptn, block):
    """P*[(2T-4)*V + (2T-4)*2 + T*1 + 8 + 16], V = block*8."""
    V = block * 8
    return nptn * ((2 * ntaxa - 4) * V + (2 * ntaxa - 4) * 2 + ntaxa + 8 + 16)


def design_min_bytes_per_traversal(ntaxa, nptn, block):
    """What the fused whole-plan-per-launch design cannot avoid moving: every internal vector and its counters
    written once (T-2), the vectors that are neither the previous result nor register-parked re-read (not counted
    here: plan dependent), leaf states read once, _pattern_lh written, ptn_freq / ptn_invar read."""
    V = block * 8
    return nptn * ((ntaxa - 2) * (V + 2) + ntaxa + 8 + 16)


def algorithmic_flops_per_traversal(ntaxa, nptn, n, ncat):
    """SURVEY.md 8(d) per-update flops summed over a traversal rooted at a leaf branch:
    T-2 updates each pay the U^-1 product + Hadamard (2n^2+n), T-3 internal children each pay one
    E*v product (2n^2); leaf children are table look-ups in the reference (0 flops)."""
    return nptn * ncat * ((ntaxa - 2) * (2 * n * n + n) + (ntaxa - 3) * 2 * n * n)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(args):
    """--gpus N without a launcher: start the N ranks as fresh child processes (this process has not touched the
    GPU and never will) and pass rank 0's JSON line through."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


class Dist:
    """torch.distributed plumbing of one rank (backend nccl = RCCL)."""

    def __init__(self, args):
        import torch
        self.torch = torch
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != args.gpus:
            raise SystemExit("WORLD_SIZE %d != --gpus %d" % (self.world, args.gpus))
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the likelihood path has no CPU fallback")
        torch.cuda.set_device(self.local_rank)
        self.collective = self.world > 1 or args.force_collective
        self.dist = None
        if self.collective:
            import torch.distributed as dist
            self.dist = dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=self.rank, world_size=self.world,
                                    device_id=torch.device("cuda", self.local_rank))

    def barrier(self):
        self.torch.cuda.synchronize()
        if self.collective:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def max_over_ranks(self, v):
        if not self.collective:
            return v
        t = self.torch.tensor([v], dtype=self.torch.float64, device="cuda")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())


def enable_collective(tree, D, lib, pkg, mode):
    """Sharded run, one process per GPU.  mode "rccl" (default): the engine joins an RCCL communicator of its own
    (iqhip_comm_init_rank; the 128-byte id travels over torch.distributed) and all-reduces every result vector
    itself, in C++, on its stream.  mode "torch": the older caller-owned collective -- the engine works on torch's
    current stream and leaves each result vector in a torch-owned device buffer, which a Python hook all-reduces
    with torch.distributed before the single host read.  If the C++ communicator cannot be made on some rank, ALL
    ranks fall back to "torch" and the JSON line says so."""
    torch = D.torch
    if mode == "rccl":
        ok, err = 1, ""
        try:
            idt = torch.zeros(128, dtype=torch.uint8, device="cuda")
            if D.rank == 0:
                idt.copy_(torch.frombuffer(bytearray(pkg.comm_unique_id()), dtype=torch.uint8))
            D.dist.broadcast(idt, src=0)
            tree.attach_comm(D.world, D.rank, bytes(idt.cpu().numpy().tobytes()))
        except Exception as ex:  # noqa: BLE001
            ok, err = 0, str(ex)
        flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
        D.dist.all_reduce(flag, op=D.dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            return "rccl: ncclAllReduce inside libiqhip.so (iqhip_comm_init_rank)"
        if ok:
            raise SystemExit("C++ communicator made on this rank but not on all: cannot continue consistently")
        sys.stderr.write("[bench] C++ RCCL communicator failed (%s): falling back to torch.distributed\n" % err)
        mode = "torch (fallback: %s)" % err
    stream = torch.cuda.current_stream()
    assert lib.iqhip_set_stream(tree.engine, C.c_void_p(stream.cuda_stream)) == 0
    res = torch.zeros(2 + 4096, dtype=torch.float64, device="cuda")
    assert lib.iqhip_bind_result_buffer(tree.engine, C.c_void_p(res.data_ptr()), res.numel()) == 0

    def hook(ptr, n):
        assert ptr == res.data_ptr()
        D.dist.all_reduce(res[:n], op=D.dist.ReduceOp.SUM)
    tree._collective_buffer = res
    tree.set_allreduce_hook(hook)
    return "torch.distributed all_reduce from a Python hook%s" % (mode[5:] if mode.startswith("torch (") else "")


def kernel_name(pkg, nst, ncat, nclass):
    if nst == 4:
        return "k_traverse4<%d,256>" % ncat
    if nclass > 1 or (nst == 20 and ncat not in (1, 4)):
        return "k_traverse_mfma_mix20<256>"
    if (nst, ncat) in ((20, 4), (20, 1), (64, 1)):
        return "k_traverse_mfma2<%d,%d,256>" % (nst, ncat)
    return "k_traverse_mfma<%d,256>" % nst


def run_workload(args, D, pkg, synth, workload, steps, warmup, with_cpu_baseline):
    """-> the JSON object of one workload at this world size (rank 0 gets the full object)."""
    lib = pkg.libiqhip()
    torch = D.torch
    T0, P0, nst, ncat0, seq_type = synth.BASELINE_SHAPES[workload]
    T = args.ntaxa or T0
    P = args.patterns or P0
    strong = workload in STRONG and not args.patterns
    if strong:
        P = (P0 + D.world - 1) // D.world
    ncat = args.ncat if (args.ncat and workload in ("dna", "protein", "dna4")) else 0
    nwk, pat, freq, model = synth.baseline_workload(workload, ntaxa=T, patterns=P, shard=D.rank, ncat=ncat)

    tree = pkg.PhyloTree(nwk)
    tree.set_alignment(nst, seq_type, pat, freq)
    tree.set_model(model)
    tree.set_likelihood_kernel(pkg.LK_EIGEN_HIP)
    if args.reference_order:
        tree.set_heavy_first(False)
    tree.attach_engine(D.local_rank)
    eng = tree.engine
    collective = None
    if D.collective:
        collective = enable_collective(tree, D, lib, pkg, args.collective)

    def step():  # clearAllPartialLH(); computeLikelihood() on the C++ side of the boundary
        return tree.clear_and_compute_likelihood()

    lnl = None
    for _ in range(warmup):
        lnl = step()
    D.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        lnl = step()
    D.barrier()
    dt = D.max_over_ranks(time.perf_counter() - t0)

    # sustained leg (not the reported value): the same step back to back for --sustain-seconds, so that a
    # power / activity sampler sees the region; the K-step region above can be a few milliseconds long
    sustained = None
    if args.sustain_seconds > 0:
        nrep = max(steps, int(args.sustain_seconds / max(dt / steps, 1e-6)))
        nrep = int(D.max_over_ranks(float(nrep)))
        D.barrier()
        t1 = time.perf_counter()
        for _ in range(nrep):
            step()
        D.barrier()
        dts = D.max_over_ranks(time.perf_counter() - t1)
        sustained = {"steps": nrep, "seconds": dts, "value": nrep * (T - 2) * P * D.world / dts / 1e6,
                     "ms_per_step": dts / nrep * 1e3}

    # dominant-kernel duration: HIP events on the launch stream around that kernel, measured live in
    # a further pass of the same steps (kept out of the timed region: two event records per step)
    lib.iqhip_timing_enable(eng, 1)
    ntimed = min(max(steps, 20), 100)
    ch_built0, ch_ops0 = C.c_int64(), C.c_int64()
    lib.iqhip_debug_cherry_tables(eng, C.byref(ch_built0), C.byref(ch_ops0))
    for _ in range(ntimed):
        step()
    ch_built1, ch_ops1 = C.c_int64(), C.c_int64()
    lib.iqhip_debug_cherry_tables(eng, C.byref(ch_built1), C.byref(ch_ops1))
    avg_ms, launches = C.c_double(), C.c_int64()
    lib.iqhip_timing_read(eng, C.byref(avg_ms), C.byref(launches), 1)
    coll_us, coll_n = C.c_double(), C.c_int64()
    lib.iqhip_timing_collective_read(eng, C.byref(coll_us), C.byref(coll_n), 1)
    lib.iqhip_timing_enable(eng, 0)

    updates = steps * (T - 2) * P * D.world
    value = updates / dt / 1e6
    block = nst * model.ncat
    # one traversal = one launch of the traversal kernel, or several for a staged plan (independent subtrees, then the
    # ops above them).  Everything below is PER TRAVERSAL (= all its launches together): bytes / flops of the
    # traversal over the sum of its launches' durations; kernel_avg_ms stays the per-launch average a profiler reports.
    lpt = max(1.0, launches.value / float(ntimed))
    kern_s = avg_ms.value * lpt * 1e-3
    algo_bytes = algorithmic_bytes_per_traversal(T, P, block)
    algo_flops = algorithmic_flops_per_traversal(T, P, nst, model.ncat)
    floor_bytes = design_min_bytes_per_traversal(T, P, block)
    st_b, ld_b = C.c_double(), C.c_double()
    lib.iqhip_timing_plan_bytes(eng, C.byref(st_b), C.byref(ld_b))
    plan_bytes = st_b.value + ld_b.value
    # HBM bytes per traversal from the committed rocprofv3 PMC passes (tools/profile_round.sh -> profiles/traffic.json:
    # FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE, separate passes) when this exact shape was profiled; they are
    # measurements of the same binary on another box, not of this run -- said so in traffic_source
    pmc = None
    try:
        tr = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(workload)
        if tr and tr["ntaxa"] == T and tr["patterns_per_gpu"] == P and tr["ncat"] == model.ncat:
            pmc = tr
    except Exception:
        pass
    if pmc:
        traffic = pmc["hbm_traffic_bytes_per_launch"] * pmc["launches_per_traversal"]
        traffic_source = "PMC, NOT of this run: " + pmc["source"]
    else:
        traffic = None
        traffic_source = "no PMC pass of this shape on file: priced with the launch's own descriptors (iqhip_timing_plan_bytes)"
    phys_bytes = traffic if traffic is not None else plan_bytes
    gbs = lambda nbytes: nbytes / kern_s / 1e9 if kern_s > 0 else 0.0   # noqa: E731
    tfl = algo_flops / kern_s / 1e12 if kern_s > 0 else 0.0
    hbm = {"achieved": gbs(phys_bytes), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs(phys_bytes) / HBM_PEAK_GBS}
    # (algorithmic flops: what SURVEY 8(d)'s formula asks for, leaf children as look-ups.  The 20-state kernel runs leaf
    # children of leaf + internal nodes on the matrix pipe (more than the formula) and answers leaf-leaf nodes from cherry
    # tables (none of the formula's contraction for them): node_updates_from_cherry_tables says how many per traversal)
    mfma = {"achieved": tfl, "peak": MFMA_F64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tfl / MFMA_F64_PEAK_TFLOPS,
            "flops_per_traversal": algo_flops, "flops_are": "algorithmic (SURVEY 8d), not executed",
            "node_updates_from_cherry_tables": (ch_ops1.value - ch_ops0.value) / float(ntimed),
            "cherry_tables_rebuilt_in_timed_steps": ch_built1.value - ch_built0.value}
    if nst == 64:
        roof = dict(bound="mfma", **mfma)
        roof["hbm"] = hbm
    else:
        roof = dict(bound="hbm", **hbm)
        if nst == 20:
            roof["mfma"] = mfma
    assert roof["frac"] <= 1.0, "a roofline fraction above 1 means the byte / flop model is wrong, not that the chip is fast"
    roof.update({
        # bytes that crossed the memory fabric per traversal (PMC) or null; what `achieved` divides by the kernel time
        "traffic": traffic, "traffic_source": traffic_source,
        "bytes_priced": phys_bytes,
        # the launch's own requests, live from this run's descriptors: stores exact, loads an upper bound of the fabric reads
        "plan_bytes": {"stored": st_b.value, "loaded_upper_bound": ld_b.value},
        # SURVEY 8(d) algorithmic bytes (every child vector read, every result written) / kernel time / HBM peak: a MODEL
        # rate, not a roofline fraction -- the fused launch keeps most children in registers, so it can exceed 1
        "algorithmic_bytes_per_traversal": algo_bytes, "frac_algorithmic": gbs(algo_bytes) / HBM_PEAK_GBS,
        # the design's own floor: bytes it cannot avoid (results + counters written once, leaf states, root pass) at the
        # measured-achievable 6.3 TB/s, against the kernel time
        "floor_bytes_per_traversal": floor_bytes, "floor_ms": floor_bytes / (HBM_ACHIEVABLE_GBS * 1e9) * 1e3,
        "frac_of_floor": (floor_bytes / (HBM_ACHIEVABLE_GBS * 1e9)) / kern_s if kern_s > 0 else 0.0,
        "kernel": (pmc or {}).get("kernel") or kernel_name(pkg, nst, model.ncat, int(getattr(model, "nclass", 1))),
        "kernel_name_source": "rocprofv3 kernel trace (profiles/)" if pmc else "expected instantiation (not observed)",
        "kernel_avg_ms": avg_ms.value, "launches": launches.value, "launches_per_traversal": lpt,
        "kernel_ms_per_traversal": avg_ms.value * lpt})
    out = {
        "metric": "million pattern-node partial-likelihood updates/sec",
        "value": value,
        "unit": "M updates/s",
        "n_gpus": D.world,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": dt / steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "%s %d taxa x %d patterns/GPU, %s, fixed tree: clearAllPartialLH + "
                               "full traversal + root-branch lnL" % (workload.upper(), T, P, model.name),
                   "ntaxa": T, "patterns_per_gpu": P, "patterns_total": P * D.world, "nstates": nst,
                   "ncat": model.ncat,
                   "parallelism": "patterns sharded over %d GPU(s), 1 RCCL all-reduce/step" % D.world,
                   "collective": collective},
        "lnL": lnl,
        "host_overhead_ms_per_step": dt / steps * 1e3 - avg_ms.value * lpt,
        # N > 1 (or --force-collective): the engine's own all-reduce, HIP events around it (incl. waiting for slower ranks)
        "collective_us_per_step": coll_us.value * coll_n.value / float(ntimed) if coll_n.value else None,
        "roofline": roof,
    }
    if sustained:
        out["sustained"] = sustained

    if D.rank == 0 and D.world == 1 and with_cpu_baseline:
        # The oracle on the WHOLE workload: parity of this very run (lnL delta, root-side counters), then the CPU
        # baseline timed on the same input (bounded by --cpu-seconds; a traversal takes 0.06 .. 10 s)
        od = entry.load_oracle()
        ot = od.OracleTree(nwk, nst, seq_type, pat, freq, None, model)
        try:
            ncores = min(len(os.sched_getaffinity(0)), 16)
        except AttributeError:
            ncores = min(os.cpu_count() or 1, 16)
        ncores = od.lib().oracle_set_threads(ncores)
        olnl, (a, b) = ot.likelihood()
        frm, to = (a, b) if not ot.is_leaf(b) else (b, a)
        osc = ot.partial(frm, to)[1]
        gsc = tree.fetch_scale_num(frm, to)
        out["lnl_oracle"] = olnl
        out["lnl_rel_delta_vs_oracle"] = abs(lnl - olnl) / abs(olnl)
        out["scale_num_mismatches"] = int((gsc != osc).sum())
        out["scale_num_sum"] = int(osc.astype(np.int64).sum())
        mups, reps, secs = ot.time_traversals(budget_s=args.cpu_seconds * 0.6)
        od.lib().oracle_set_threads(1)
        mups1, reps1, secs1 = ot.time_traversals(budget_s=args.cpu_seconds * 0.4)
        out["cpu_baseline"] = {"value": mups, "unit": "M updates/s", "cores": ncores, "kind": "port",
                               "sample": "oracle/lh_oracle.c (gcc -O3 -mavx -fopenmp, pattern loop threaded as the "
                                         "reference's '#pragma omp parallel for'): %d traversals of the same tree on all "
                                         "%d patterns of the workload in %.1f s with %d threads; 1 thread: %.2f M updates/s "
                                         "(%d traversals in %.1f s)" % (reps, P, secs, ncores, mups1, reps1, secs1)}
    tree.close()
    return out


def run_branchopt(pkg, synth, shapes=((50, 100000), (44, 355))):
    """Hot loop 2: one optimizeAllBranches sweep (phylotree.cpp:2252-2332) over all 2T-3 branches of
    a DNA GTR+G4 tree, every branch starting at 0.1 -- as one engine submission per branch (iqhip_optimize_branch) and as
    ONE submission per sweep (iqhip_optimize_sweep: a persistent kernel for 4-state engines).  us per branch, and the
    fraction of the HBM peak that the bytes a branch cannot avoid (two node updates of three vectors each, theta written
    once, read once per derivative evaluation) amount to at that speed."""
    model = synth.gtr_model()
    out = []
    for (T, P) in shapes:
        nwk, pat, freq = synth.make_workload(T, P, model, seed=3)
        row = {"ntaxa": T, "patterns": P, "model": "GTR+G4", "branches": 2 * T - 3}
        for form in ("branch", "sweep"):
            t = pkg.PhyloTree(nwk)
            t.set_alignment(4, pkg.SEQ_DNA, pat, freq)
            t.set_model(model)
            t.attach_engine(0)
            t.set_device_newton(True)
            t.set_device_sweep(form == "sweep")
            best = None
            for rep in range(4):   # (the first repetition pays allocations)
                for a in range(t.num_nodes):
                    for b, _ in t.neighbors(a):
                        if a < b:
                            t.set_branch_length(a, b, 0.1, clear_reverse=False)
                t.clear_all_partial_lh()
                t.compute_likelihood()
                c0, t0 = t.num_derv_calls, time.perf_counter()
                lnl = t.optimize_all_branches(iterations=1, tolerance=1e-3)
                dt = time.perf_counter() - t0
                if best is None or dt < best[0]:
                    best = (dt, t.num_derv_calls - c0, lnl)
            t.close()
            dt, nev, lnl = best
            nb = 2 * T - 3
            V = P * 16 * 8.0
            stream_bytes = 6 * V + V + (nev / nb) * V
            us = dt * 1e6 / nb
            row[form] = {"us_per_branch": us, "sweep_ms": dt * 1e3, "derivative_evaluations": nev, "lnL": lnl,
                         "hbm_frac": stream_bytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS}
        row["unavoidable_bytes_per_branch"] = 6 * V + V + (row["sweep"]["derivative_evaluations"] / (2 * T - 3)) * V
        out.append(row)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", choices=["dna", "protein", "codon", "mixture", "dna4"], default=None,
                    help="dna = BASELINE configs[1] (the headline); protein / dna4 / codon = configs[2] / [3] / [4]")
    ap.add_argument("--ntaxa", type=int, default=0)
    ap.add_argument("--patterns", type=int, default=0, help="patterns per GPU (overrides the workload's count)")
    ap.add_argument("--cpu-seconds", type=float, default=14.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true",
                    help="default workload only: skip the configs[2..4] (protein, dna4, codon) results under 'also'")
    ap.add_argument("--also", action="store_true", help="add the 'also' results to a non-default workload")
    ap.add_argument("--sustain-seconds", type=float, default=2.0,
                    help="after the timed K steps, repeat the step for this long (activity evidence; 0 = off)")
    ap.add_argument("--collective", choices=["rccl", "torch"], default="rccl",
                    help="N > 1: rccl = the engine's own communicator (C++), torch = torch.distributed from a Python hook")
    ap.add_argument("--force-collective", action="store_true",
                    help="exercise the N>1 path (RCCL all-reduce of the result vector) even with one rank")
    ap.add_argument("--reference-order", action="store_true",
                    help="plan subtrees in the reference's neighbour order instead of heavier-first")
    ap.add_argument("--ncat", type=int, default=0, help="rate categories (dna / protein workloads; default: the BASELINE shape)")
    args = ap.parse_args()
    default_command = args.workload is None   # the driver's command: headline + the multi-GPU configs under "also"
    args.workload = args.workload or "dna"

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))

    real_stdout = os.dup(1)
    # RCCL prints a version banner on stdout at communicator creation: keep stdout clean for the single JSON line
    sys.stdout.flush()
    os.dup2(2, 1)
    D = Dist(args)
    pkg = entry.load_package()
    import importlib
    synth = importlib.import_module("iqtree_amd.synth")

    out = run_workload(args, D, pkg, synth, args.workload, args.steps, args.warmup, not args.no_cpu_baseline)
    # the multi-GPU configs north_star names, at this N, beside the headline (the driver runs one command per N)
    plain = not (args.ntaxa or args.patterns or args.ncat)
    if plain and ((default_command and not args.no_also) or args.also):
        also = []
        for w in ("protein", "dna4", "codon"):
            if w == args.workload:
                continue
            # (dna4 steps are 6 ms each; protein / codon steps are 1 / 0.36 ms and need the clocks ramped: a handful of
            # warm-up steps under-reports them by 10 %)
            ks, kw = (max(20, args.steps // 4), max(5, args.warmup // 4)) if w == "dna4" else (max(200, args.steps), max(60, args.warmup))
            r = run_workload(args, D, pkg, synth, w, ks, kw, False)
            r.pop("sustained", None)
            also.append(r)
        out["also"] = also
        if D.world == 1:
            out["branchopt"] = run_branchopt(pkg, synth)

    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    if D.rank == 0:
        print(json.dumps(out), flush=True)
    if D.collective:
        D.dist.destroy_process_group()


if __name__ == "__main__":
    main()
