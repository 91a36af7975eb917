"""GPU: pattern sharding inside the engine (include/iqhip.h "pattern sharding over GPUs", csrc/sharded.hip,
csrc/comm.hip).  A one-GPU box cannot give RCCL two ranks (it refuses two ranks on one device), so:
  * several shards ON THE SAME DEVICE with the pinned-host reduction exercise everything of the single-process front
    except the collective itself: splitting, fan-out, the Newton state machine, gathers of the host views;
  * the RCCL code path (communicator creation with ncclCommInitAll resp. ncclGetUniqueId + ncclCommInitRank,
    in-stream ncclAllReduce of the device result vector, the enqueued Newton chain with its update kernel) runs with
    ONE rank.
Everything is compared with the oracle on the unsharded alignment and with a plain engine."""
import ctypes as C

import numpy as np
import pytest

from test_parity_gpu import make_case, LNL_RTOL  # noqa: F401

pytestmark = pytest.mark.gpu


def sharded_tree(pkg, t_plain_args, devices, mode, mem_mode=0):
    nwk, n, seq_type, pat, freq, invar, model = t_plain_args
    t = pkg.PhyloTree(nwk)
    t.set_mem_mode(mem_mode)
    t.set_alignment(n, seq_type, pat, freq, invar)
    t.set_model(model)
    t.set_likelihood_kernel(pkg.LK_EIGEN_HIP)
    t.attach_engine_sharded(devices, mode)
    return t


def case(synth, oracle, pkg, ntaxa, nptn, n, ncat, seed, seq_type=0, mem_mode=0, missing=0.0, lo=0.02, hi=0.2,
         caterpillar=False):
    """-> plain-engine tree, oracle tree, the inputs (for a second, sharded tree over the same node numbering)"""
    if n == 4:
        model = synth.gtr_model(alpha=0.9, ncat=ncat)
    else:
        model = synth.random_reversible_model(n, seed, alpha=0.9, ncat=ncat)
    su = oracle.state_unknown_for(n, seq_type)
    nwk = synth.random_tree_newick(ntaxa, seed, lo, hi, caterpillar)
    st = synth.simulate_alignment(nwk, model, nptn, seed + 1, missing, su)
    pat, freq = synth.compress_patterns(st)
    invar = synth.ptn_invar_for(pat, model)
    ot = oracle.OracleTree(nwk, n, seq_type, pat, freq, invar, model)
    t = pkg.PhyloTree(nwk)
    t.set_mem_mode(mem_mode)
    t.set_alignment(n, seq_type, pat, freq, invar)
    t.set_model(model)
    t.set_likelihood_kernel(pkg.LK_EIGEN_HIP)
    t.attach_engine(0)
    return t, ot, (nwk, n, seq_type, pat, freq, invar, model)


def all_vectors_equal(ts, t, ot):
    n = 0
    for a in range(t.num_nodes):
        for b, _ in t.neighbors(a):
            i1, i2 = t.neighbor_info(a, b), ts.neighbor_info(a, b)
            assert (i1["computed"] & 1) == (i2["computed"] & 1)
            if ot.is_leaf(b) or not (i1["computed"] & 1) or i1["key"] == 0:
                continue
            # identical per-pattern arithmetic on every shard: the gathered vector is bit-identical
            assert np.array_equal(ts.fetch_partial(a, b), t.fetch_partial(a, b))
            assert np.array_equal(ts.fetch_scale_num(a, b), t.fetch_scale_num(a, b))
            assert abs(i1["lh_scale_factor"] - i2["lh_scale_factor"]) <= 1e-12 * max(1.0, abs(i1["lh_scale_factor"]))
            n += 1
    return n


@pytest.mark.parametrize("n,ncat,seq_type,ntaxa,nptn,kw", [
    (4, 4, 0, 14, 2600, dict(missing=0.04)),
    (4, 4, 0, 150, 900, dict(lo=0.4, hi=0.9, caterpillar=True)),      # every pattern rescaled: sum_scale rows reduced
    (20, 4, 1, 12, 1100, dict(missing=0.03)),
    (64, 1, 2, 9, 700, dict()),
])
@pytest.mark.parametrize("setup", ["host2", "host3", "rccl1"])
def test_sharded_engine_matches_plain_engine_and_oracle(pkg, synth, oracle, n, ncat, seq_type, ntaxa, nptn, kw, setup):
    t, ot, args = case(synth, oracle, pkg, ntaxa, nptn, n, ncat, 7100 + n + ntaxa, seq_type=seq_type, **kw)
    devices, mode = {"host2": ([0, 0], pkg.REDUCE_HOST), "host3": ([0, 0, 0], pkg.REDUCE_HOST),
                     "rccl1": ([0], pkg.REDUCE_RCCL)}[setup]
    ts = sharded_tree(pkg, args, devices, mode)
    lib = pkg.libiqhip()
    assert lib.iqhip_num_shards(ts.engine) == len(devices) and lib.iqhip_comm_size(ts.engine) == len(devices)
    ref, (a, b) = ot.likelihood()
    lnl, lnl_s = t.compute_likelihood(), ts.compute_likelihood()
    assert abs(lnl_s - ref) <= LNL_RTOL * abs(ref)
    assert abs(lnl_s - lnl) <= 1e-12 * abs(lnl)
    assert all_vectors_equal(ts, t, ot) == ntaxa - 2
    np.testing.assert_array_equal(ts.fetch_pattern_lh(), t.fetch_pattern_lh())
    # derivatives, lnL from theta
    df, ddf = t.compute_likelihood_derv(a, b)
    dfs, ddfs = ts.compute_likelihood_derv(a, b)
    assert abs(dfs - df) <= 1e-10 * max(abs(df), 1e-3 * abs(ddf)) and abs(ddfs - ddf) <= 1e-10 * abs(ddf)
    assert abs(ts.compute_likelihood_from_buffer() - t.compute_likelihood_from_buffer()) <= 1e-12 * abs(lnl)
    np.testing.assert_allclose(ts.compute_pattern_likelihood(), t.compute_pattern_likelihood(), rtol=1e-13)
    # Newton on a few branches: same number of derivative evaluations, same optimum as the plain engine's k_newton
    edges = [(x, y) for x in range(t.num_nodes) for y, _ in t.neighbors(x) if x < y]
    for (x, y) in edges[:5]:
        start = 0.31 if (x + y) % 2 else 0.004
        out = []
        for tree in (t, ts):
            tree.set_branch_length(x, y, start, clear_reverse=True)
            c0 = tree.num_derv_calls
            out.append((tree.optimize_one_branch(x, y), tree.num_derv_calls - c0))
        assert out[0][1] == out[1][1], (x, y, out)
        assert abs(out[0][0] - out[1][0]) <= 1e-9 * max(out[0][0], 1e-6)
    assert abs(ts.compute_likelihood() - t.compute_likelihood()) <= 1e-11 * abs(lnl)


def test_sharded_shard_ranges_and_refusals(pkg, synth, oracle):
    import ctypes as C
    lib = pkg.libiqhip()
    t, ot, args = case(synth, oracle, pkg, 8, 1000, 4, 4, 7300)
    ts = sharded_tree(pkg, args, [0, 0, 0], pkg.REDUCE_HOST)
    f, c, d = C.c_int64(), C.c_int64(), C.c_int()
    covered = 0
    for g in range(3):
        assert lib.iqhip_shard_range(ts.engine, g, C.byref(f), C.byref(c), C.byref(d)) == 0
        assert f.value == covered and f.value % 64 == 0 and c.value > 0 and d.value == 0
        covered += c.value
    assert covered == ts.nptn
    assert lib.iqhip_shard_range(ts.engine, 3, C.byref(f), C.byref(c), C.byref(d)) != 0
    # the front reduces itself: the caller-owned-collective calls are refused, not half-executed
    assert lib.iqhip_derv_async(ts.engine, 0.1) == 3 and b"sharded" in lib.iqhip_last_error()
    assert lib.iqhip_bind_result_buffer(ts.engine, None, 0) == 3
    assert lib.iqhip_set_stream(ts.engine, None) == 3
    # RCCL cannot put two ranks on one device; too few patterns for the shard count
    e = C.c_void_p()
    two = (C.c_int * 2)(0, 0)
    assert lib.iqhip_create_sharded(C.byref(e), two, 2, pkg.REDUCE_RCCL, 4, 4, 1000, 8) == 2
    assert b"distinct devices" in lib.iqhip_last_error()
    assert lib.iqhip_create_sharded(C.byref(e), two, 2, pkg.REDUCE_HOST, 4, 4, 100, 8) == 2
    with pytest.raises(pkg.HostError):   # +ASC: the unobserved patterns must fit the last shard (here: 320 patterns)
        ts.set_ascertainment(400, 100.0)
        ts.compute_likelihood()


@pytest.mark.parametrize("setup", ["host2", "rccl1", "comm1"])
def test_sharded_branch_optimisation_and_nni(pkg, synth, oracle, setup):
    """hot loop 2 on a sharded engine: optimizeAllBranches (device Newton as the enqueued chain / the host-advanced
    state machine) and the NNI evaluators (batch API -> sequential through the chain) against the plain engine."""
    t, ot, args = case(synth, oracle, pkg, 13, 1500, 4, 4, 7400, mem_mode=pkg.LM_ALL_BRANCH)
    if setup == "comm1":   # one process per GPU form, one rank
        nwk, n, seq_type, pat, freq, invar, model = args
        ts = pkg.PhyloTree(nwk)
        ts.set_mem_mode(pkg.LM_ALL_BRANCH)
        ts.set_alignment(n, seq_type, pat, freq, invar)
        ts.set_model(model)
        ts.attach_engine(0)
        ts.attach_comm(1, 0, pkg.comm_unique_id())
        assert pkg.libiqhip().iqhip_comm_size(ts.engine) == 1
    else:
        devices, mode = {"host2": ([0, 0], pkg.REDUCE_HOST), "rccl1": ([0], pkg.REDUCE_RCCL)}[setup]
        ts = sharded_tree(pkg, args, devices, mode, mem_mode=pkg.LM_ALL_BRANCH)
    lnl = t.compute_likelihood()
    assert abs(ts.compute_likelihood() - lnl) <= 1e-12 * abs(lnl)
    batch, batch_s = t.evaluate_nnis_batch(), ts.evaluate_nnis_batch()
    assert len(batch) == len(batch_s) == 2 * (13 - 3)
    for m, ms in zip(batch, batch_s):
        assert (m["node1"], m["node2"], m["node1_nei"], m["node2_nei"]) == (ms["node1"], ms["node2"], ms["node1_nei"], ms["node2_nei"])
        assert abs(m["new_len"] - ms["new_len"]) <= 1e-8 * max(m["new_len"], 1e-6)
        assert abs(m["newloglh"] - ms["newloglh"]) <= 1e-10 * abs(m["newloglh"])
    inner = [(m["node1"], m["node2"]) for m in batch[::2]][:3]
    for a, b in inner:
        s1, s2 = t.nni_for_branch(a, b, nni5=True), ts.nni_for_branch(a, b, nni5=True)
        for c in range(2):
            assert abs(s1[c][0] - s2[c][0]) <= 1e-10 * abs(s1[c][0])
            np.testing.assert_allclose(s1[c][3], s2[c][3], rtol=1e-7, atol=1e-12)
    v, vs = t.optimize_all_branches(iterations=3, tolerance=1e-4), ts.optimize_all_branches(iterations=3, tolerance=1e-4)
    assert abs(v - vs) <= 1e-10 * abs(v)
    ot2 = oracle.OracleTree(ts.tree_string(), 4, 0, ot.states, ot.freq, None, ot.model)
    ref, _ = ot2.likelihood()
    assert abs(vs - ref) <= 1e-8 * abs(ref)


@pytest.mark.parametrize("n,ncat,seq_type,nptn", [(4, 4, 0, 1500), (20, 4, 1, 900)])
@pytest.mark.parametrize("setup", ["host2", "host3", "rccl1", "comm1"])
def test_batched_nni_tasks_share_one_reduction_per_newton_step(pkg, synth, oracle, monkeypatch, setup, n, ncat, seq_type, nptn):
    """iqhip_optimize_branch_batch on pattern shards: the tasks advance side by side -- one derivative launch per shard
    with the task as grid.y, ONE reduction / all-reduce of 2m doubles per Newton step, one update kernel -- instead of one
    chain per task.  Same accepted lengths, evaluation counts and lnL as the one-task-at-a-time form
    (IQHIP_BATCH_SEQUENTIAL=1) and as the plain engine's k_newton_batch; chunked batches (IQHIP_BATCH_CHUNK) too."""
    t, ot, args = case(synth, oracle, pkg, 11, nptn, n, ncat, 7700 + n, seq_type=seq_type, mem_mode=pkg.LM_ALL_BRANCH)
    nwk, n_, st_, pat, freq, invar, model = args
    if setup == "comm1":
        ts = pkg.PhyloTree(nwk)
        ts.set_mem_mode(pkg.LM_ALL_BRANCH)
        ts.set_alignment(n, seq_type, pat, freq, invar)
        ts.set_model(model)
        ts.attach_engine(0)
        ts.attach_comm(1, 0, pkg.comm_unique_id())
    else:
        devices, mode = {"host2": ([0, 0], pkg.REDUCE_HOST), "host3": ([0, 0, 0], pkg.REDUCE_HOST), "rccl1": ([0], pkg.REDUCE_RCCL)}[setup]
        ts = sharded_tree(pkg, args, devices, mode, mem_mode=pkg.LM_ALL_BRANCH)
    t.compute_likelihood()
    ts.compute_likelihood()
    plain = t.evaluate_nnis_batch()
    lib = pkg.libiqhip()

    def collectives(run):   # all-reduces the rank issued while `run` ran (comm engines count them while timing is on)
        if setup != "comm1":
            return run(), None
        lib.iqhip_timing_enable(ts.engine, 1)
        avg, cnt = C.c_double(), C.c_int64()
        lib.iqhip_timing_collective_read(ts.engine, C.byref(avg), C.byref(cnt), 1)
        out = run()
        lib.iqhip_timing_collective_read(ts.engine, C.byref(avg), C.byref(cnt), 1)
        lib.iqhip_timing_enable(ts.engine, 0)
        return out, cnt.value
    monkeypatch.setenv("IQHIP_BATCH_SEQUENTIAL", "1")
    seq, nseq = collectives(ts.evaluate_nnis_batch)
    monkeypatch.delenv("IQHIP_BATCH_SEQUENTIAL")
    side, nside = collectives(ts.evaluate_nnis_batch)
    if setup == "comm1":   # 16 tasks: one all-reduce per Newton step of the slowest task (+ updates, lnL) instead of per task and step
        assert nside * 6 <= nseq, (nside, nseq)
    monkeypatch.setenv("IQHIP_BATCH_CHUNK", "3")
    chunked = ts.evaluate_nnis_batch()
    monkeypatch.delenv("IQHIP_BATCH_CHUNK")
    assert len(plain) == len(seq) == len(side) == len(chunked) == 2 * (11 - 3)
    for mp, ms, mb, mc in zip(plain, seq, side, chunked):
        assert ms["new_len"] == mb["new_len"] == mc["new_len"], (ms, mb, mc)        # same sums in the same order: same iterates
        assert abs(ms["newloglh"] - mb["newloglh"]) <= 1e-12 * abs(ms["newloglh"])
        assert mb["newloglh"] == mc["newloglh"]
        assert abs(mp["new_len"] - mb["new_len"]) <= 1e-8 * max(mp["new_len"], 1e-6)
        assert abs(mp["newloglh"] - mb["newloglh"]) <= 1e-10 * abs(mp["newloglh"])
    # the five-branch form goes through the same entry point
    five, five_s = t.evaluate_nnis5_batch(), ts.evaluate_nnis5_batch()
    for m, ms in zip(five, five_s):
        assert abs(m["newloglh"] - ms["newloglh"]) <= 1e-9 * abs(m["newloglh"])


def test_sharded_rell_and_uploads(pkg, synth, oracle):
    t, ot, args = case(synth, oracle, pkg, 10, 1300, 4, 4, 7500)
    ts = sharded_tree(pkg, args, [0, 0], pkg.REDUCE_HOST)
    rng = np.random.default_rng(5)
    boot = rng.poisson(1.0, size=(40, t.nptn)).astype(np.float32)
    for tree in (t, ts):
        tree.compute_likelihood()
        tree.set_boot_samples(boot)
    np.testing.assert_allclose(ts.compute_rell(), t.compute_rell(), rtol=1e-12)
    a, b = t.current_branch()
    t.compute_likelihood_derv(a, b)
    ts.compute_likelihood_derv(a, b)
    np.testing.assert_array_equal(ts.compute_pattern_lh_cat(), t.compute_pattern_lh_cat())


@pytest.mark.parametrize("n,ncat,seq_type,nsites", [(4, 4, 0, 6000), (20, 4, 1, 900)])
def test_newton_chain_on_a_plain_engine(pkg, synth, oracle, n, ncat, seq_type, nsites, monkeypatch):
    """IQHIP_NEWTON=chain: the enqueued-steps form (no grid barrier, what a rank of a sharded run uses) takes the
    host loop's path exactly: same evaluations, same optimum."""
    monkeypatch.setenv("IQHIP_NEWTON", "chain")
    t, ot, *_ = make_case(synth, oracle, pkg, 9, nsites, n, ncat, 7600 + n, seq_type=seq_type, missing=0.02)
    t.compute_likelihood()
    edges = [(a, b) for a in range(t.num_nodes) for b, _ in t.neighbors(a) if a < b]
    for (a, b) in edges[:6]:
        start = 0.31 if (a + b) % 2 else 0.004
        res = {}
        for mode in (False, True):
            t.set_device_newton(mode)
            t.set_branch_length(a, b, start, clear_reverse=True)
            c0 = t.num_derv_calls
            res[mode] = (t.optimize_one_branch(a, b), t.num_derv_calls - c0)
        (lh, ch), (ld, cd) = res[False], res[True]
        assert cd == ch and abs(ld - lh) <= 1e-12 * max(lh, 1e-6), (a, b, res)


@pytest.mark.parametrize("n,ncat,seq_type", [(4, 4, 0), (20, 4, 1)])
@pytest.mark.parametrize("setup", ["host2", "host3", "rccl1", "comm1"])
def test_sharded_ascertainment_bias_correction(pkg, synth, oracle, n, ncat, seq_type, setup):
    """+ASC on pattern shards (phylokernel.h:655-725, 868-909, 1124-1187): the unobserved constant patterns sit at the end of
    the alignment, i.e. on the last shard; prob_const / df_const / ddf_const travel with the result vector through the
    reduction and every shard / the front applies the correction to the summed values.  lnL, per-pattern lnL,
    derivatives, lnL from theta, the Newton chain (device state machine with the correction inside its update kernel,
    or the host-advanced one) and a whole optimizeAllBranches against the oracle and a plain engine."""
    model = synth.gtr_model(alpha=0.9, ncat=ncat) if n == 4 else synth.random_reversible_model(n, 51, alpha=0.9, ncat=ncat)
    nwk = synth.random_tree_newick(9, 52, 0.02, 0.15)
    st = synth.simulate_alignment(nwk, model, 2500, 53)
    pat, freq = synth.compress_patterns(st)
    const = np.all(pat == pat[0][None, :], axis=0)
    pat, freq = np.ascontiguousarray(pat[:, ~const]), freq[~const].copy()
    nun, nsites = n, float(freq.sum())
    pat = np.ascontiguousarray(np.concatenate([pat, np.tile(np.arange(n, dtype=np.uint8)[None, :], (9, 1))], axis=1))
    freq = np.concatenate([freq, np.zeros(n)])
    ot = oracle.OracleTree(nwk, n, seq_type, pat, freq, None, model, n_unobs=nun, nsites=nsites)

    def plain():
        t = pkg.PhyloTree(nwk)
        t.set_alignment(n, seq_type, pat, freq)
        t.set_ascertainment(nun, nsites)
        t.set_model(model)
        t.attach_engine(0)
        return t
    t = plain()
    if setup == "comm1":
        ts = plain()
        ts.attach_comm(1, 0, pkg.comm_unique_id())
    else:
        devices, mode = {"host2": ([0, 0], pkg.REDUCE_HOST), "host3": ([0, 0, 0], pkg.REDUCE_HOST), "rccl1": ([0], pkg.REDUCE_RCCL)}[setup]
        ts = pkg.PhyloTree(nwk)
        ts.set_alignment(n, seq_type, pat, freq)
        ts.set_ascertainment(nun, nsites)
        ts.set_model(model)
        ts.attach_engine_sharded(devices, mode)
    ref, (a, b) = ot.likelihood()
    lnl, plh = ts.compute_likelihood(want_pattern_lh=True)
    assert abs(lnl - ref) <= LNL_RTOL * abs(ref)
    _, oplh = ot.branch_lnl(a, b)
    np.testing.assert_allclose(plh[:-nun], oplh[:-nun], rtol=1e-10, atol=1e-10)
    for (x, y) in [(a, b), (t.num_leaves, t.neighbors(t.num_leaves)[0][0])]:
        assert abs(ts.compute_likelihood_branch(x, y) - ref) <= LNL_RTOL * abs(ref)
        ts.reset_theta()
        df, ddf = ts.compute_likelihood_derv(x, y)
        odf, oddf = ot.derv(x, y)
        assert abs(df - odf) <= 1e-9 * max(1.0, abs(odf)) + 1e-12 * abs(oddf)
        assert abs(ddf - oddf) <= 1e-9 * abs(oddf)
        o, _ = ot.lnl_from_theta(x, y)
        assert abs(ts.compute_likelihood_from_buffer() - o) <= LNL_RTOL * abs(o)
        # one branch: the chain's optimum and evaluation count are the oracle's minimizeNewton's
        ref_x, _, pts, status = ot.minimize_newton(x, y, 1e-6, ot.length(x, y), 100.0, 1e-6, 100)
        c0 = ts.num_derv_calls
        got = ts.optimize_one_branch(x, y)
        assert status == "ok" and ts.num_derv_calls - c0 == len(pts)
        assert abs(got - ref_x) <= 1e-9 * max(1.0, ref_x)
        ot.set_length(x, y, got)
        t.set_branch_length(x, y, got, clear_reverse=True)
        ref, _ = ot.likelihood()
    v, vs = t.optimize_all_branches(iterations=2, tolerance=1e-6), ts.optimize_all_branches(iterations=2, tolerance=1e-6)
    assert abs(v - vs) <= 1e-9 * abs(v)
