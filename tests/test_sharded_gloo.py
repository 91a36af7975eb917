"""CPU, world_size 2, gloo: the pattern-sharded protocol of SURVEY.md 8e.  Each rank owns half of
the patterns; per evaluation ONE all-reduce (SUM, f64) of the result vector
{lnL part, sum_scale per node update}; every rank then applies the reference's lh_scale_factor
recursion on the reduced vector and obtains the full-alignment lnL.  On this GPU-less box the
"device" share of each rank is produced by the oracle inside the all-reduce hook (the host
mirror runs in dry-run mode); on MI355X the same hook all-reduces the engine's device buffer
over RCCL (bench.py)."""
import ctypes
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import __graft_entry__ as g
    import importlib
    pkg = g.load_package()
    synth = importlib.import_module("iqtree_amd.synth")
    od = g.load_oracle()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # deep tree with long branches so that scaling events (sum_scale != 0) are part of the reduction
    model = synth.random_reversible_model(20, 3, alpha=0.9, ncat=4)
    nwk = synth.random_tree_newick(120, 3, 0.3, 0.6, caterpillar=True)
    st = synth.simulate_alignment(nwk, model, 64, 4)
    pat, freq = synth.compress_patterns(st)
    nptn = pat.shape[1]
    lo, hi = rank * nptn // world, (rank + 1) * nptn // world
    shard = od.OracleTree(nwk, 20, 1, np.ascontiguousarray(pat[:, lo:hi]), freq[lo:hi].copy(), None, model)

    t = pkg.PhyloTree(nwk)
    t.set_alignment(20, 1, np.ascontiguousarray(pat[:, lo:hi]), freq[lo:hi])
    t.set_model(model)
    t.set_dry_run(True)
    state = {}

    def hook(ptr, n):
        buf = (ctypes.c_double * n).from_address(ptr)
        plan = state["tree"].last_plan_pending()
        arr = np.frombuffer(buf, dtype=np.float64)
        # this rank's share, computed by the oracle on its shard (stand-in for the device)
        a, b = state["branch"]
        total, _ = shard.branch_lnl(a, b)
        sums = []
        for p in plan:
            frm, to = p["dst"]
            kids = [x for x, _ in shard.adj[to] if x != frm]
            own = shard.partial(frm, to)[2]
            for c in kids:
                if not shard.is_leaf(c):
                    own -= shard.partial(to, c)[2]
            sums.append(own)
        arr[0] = total - (shard.partial(a, b)[2] if not shard.is_leaf(b) else 0.0) \
                       - (shard.partial(b, a)[2] if not shard.is_leaf(a) else 0.0)
        arr[2:2 + len(sums)] = sums
        tt = torch.from_numpy(arr)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)

    # the dry-run mirror exposes the plan it is about to submit through last_plan() only after
    # the call; compute it once up front (same tree, same flags) for the hook
    probe = pkg.PhyloTree(nwk)
    probe.set_alignment(20, 1, np.ascontiguousarray(pat[:, lo:hi]), freq[lo:hi])
    probe.set_model(model)
    probe.set_dry_run(True)
    probe.compute_likelihood()
    plan = probe.last_plan()
    state["branch"] = probe.current_branch()

    class T:
        def last_plan_pending(self):
            return plan
    state["tree"] = T()
    t.set_allreduce_hook(hook)
    lnl = t.compute_likelihood()
    a, b = t.current_branch()
    sf = t.neighbor_info(a, b)["lh_scale_factor"]
    # ---- hot loop 2 under sharding: df/ddf of a branch are all-reduced per evaluation and every rank advances the
    # engine's Newton state machine (the function the sharded engines run in a 1-thread kernel after the in-stream
    # all-reduce); all ranks take identical steps and end on the full-alignment optimum
    inner = [(x, y) for x in shard.adj for y, _ in shard.adj[x] if x < y and not shard.is_leaf(x) and not shard.is_leaf(y)]
    x, y = inner[len(inner) // 2]
    th, _ = shard.theta(x, y)
    m = pkg.NewtonStateMachine(shard.length(x, y), 1e-6, 100.0, 1e-6, 100)
    xs = []
    while not m.done:
        xs.append(m.x)
        d = torch.tensor(shard.derv(x, y, length=m.x, theta=th), dtype=torch.float64)
        dist.all_reduce(d, op=dist.ReduceOp.SUM)
        m.update(float(d[0]), float(d[1]))
    optx, d2l, nsteps, status = m.result()
    q.put((rank, lnl, sf, (x, y), xs, optx, status))
    dist.destroy_process_group()


def test_two_rank_pattern_sharding_reduces_to_full_lnl(oracle, synth):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    model = synth.random_reversible_model(20, 3, alpha=0.9, ncat=4)
    nwk = synth.random_tree_newick(120, 3, 0.3, 0.6, caterpillar=True)
    st = synth.simulate_alignment(nwk, model, 64, 4)
    pat, freq = synth.compress_patterns(st)
    full = oracle.OracleTree(nwk, 20, 1, pat, freq, None, model)
    ref, (a, b) = full.likelihood()
    sf_ref = full.partial(a, b)[2]
    assert sf_ref < 0  # scaling happened, so sum_scale entries took part in the reduction
    for rank, lnl, sf, *_ in res:
        assert abs(lnl - ref) <= 1e-11 * abs(ref), (rank, lnl, ref)
        assert abs(sf - sf_ref) <= 1e-11 * abs(sf_ref)
    assert res[0][1] == res[1][1]  # every rank holds the identical reduced value
    # Newton under sharding: both ranks evaluated the same points and agree with the one-rank solve on all patterns
    (x, y), xs0, optx0, st0 = res[0][3], res[0][4], res[0][5], res[0][6]
    assert res[1][3] == (x, y) and res[1][4] == xs0 and res[1][5] == optx0 and st0 == 0 and res[1][6] == 0
    from conftest import load_package
    pkg = load_package()
    th, _ = full.theta(x, y)
    m = pkg.NewtonStateMachine(full.length(x, y), 1e-6, 100.0, 1e-6, 100)
    xs = []
    while not m.done:
        xs.append(m.x)
        m.update(*full.derv(x, y, length=m.x, theta=th))
    assert len(xs) == len(xs0) and len(xs0) >= 2
    np.testing.assert_allclose(xs0, xs, rtol=1e-9)
    assert abs(m.result()[0] - optx0) <= 1e-9 * optx0
