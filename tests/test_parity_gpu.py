"""GPU parity: the HIP path (through the C ABI of include/iqhip.h, driven by the host mirror the
way the reference's callers drive PhyloTree) against the CPU oracle on identical inputs.
Tolerances: lnL <= 1e-9 relative here (north_star allows 1e-6); partial vectors 1e-10 relative
to the vector's max; integer scaling counters bit-exact."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LNL_RTOL = 1e-9


def make_case(synth, oracle, pkg, ntaxa, nptn, n, ncat, seed, seq_type=0, missing=0.0, pinvar=0.0,
              lo=0.02, hi=0.2, caterpillar=False, mem_mode=0):
    if n == 4:
        model = synth.gtr_model(alpha=0.9, ncat=ncat, pinvar=pinvar)
    else:
        model = synth.random_reversible_model(n, seed, alpha=0.9, ncat=ncat, pinvar=pinvar)
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
    return t, ot, model, pat, freq


def check_all_vectors(t, ot):
    """every computed neighbour: vector, scale_num (bit-exact) and lh_scale_factor."""
    nchecked = 0
    for a in range(t.num_nodes):
        for b, _ in t.neighbors(a):
            info = t.neighbor_info(a, b)
            if ot.is_leaf(b) or not (info["computed"] & 1) or info["key"] == 0:
                continue
            plh, sc, sf = ot.partial(a, b)
            got = t.fetch_partial(a, b)
            scale = np.abs(plh).max(axis=1, keepdims=True)
            np.testing.assert_allclose(got / scale, plh / scale, rtol=0, atol=1e-10)
            assert np.array_equal(t.fetch_scale_num(a, b), sc)
            assert abs(info["lh_scale_factor"] - sf) <= 1e-12 * max(1.0, abs(sf))
            nchecked += 1
    return nchecked


@pytest.mark.parametrize("ncat", [1, 2, 3, 4, 5, 6, 7, 8])
def test_dna_full_traversal_all_ncat(pkg, synth, oracle, ncat):
    t, ot, *_ = make_case(synth, oracle, pkg, 14, 700, 4, ncat, 100 + ncat, missing=0.05)
    t.clear_all_partial_lh()
    lnl = t.compute_likelihood()
    ref, (a, b) = ot.likelihood()
    assert t.current_branch() == (a, b)
    assert abs(lnl - ref) <= LNL_RTOL * abs(ref)
    assert check_all_vectors(t, ot) == 14 - 2
    _, plh = ot.branch_lnl(a, b)
    np.testing.assert_allclose(t.fetch_pattern_lh(), plh, rtol=1e-11, atol=1e-11)


@pytest.mark.parametrize("nptn", [1, 63, 64, 65, 255, 257, 1000])
def test_ragged_pattern_counts(pkg, synth, oracle, nptn):
    """tile edges: nptn not a multiple of the 64-pattern tile / 256-pattern workgroup."""
    model = synth.gtr_model()
    nwk = synth.random_tree_newick(7, 5)
    st = synth.simulate_alignment(nwk, model, max(nptn, 4), 9)[:, :nptn]
    freq = np.arange(1, nptn + 1, dtype=np.float64)
    ot = oracle.OracleTree(nwk, 4, 0, st, freq, None, model)
    t = pkg.PhyloTree(nwk)
    t.set_alignment(4, 0, st, freq)
    t.set_model(model)
    t.attach_engine(0)
    lnl = t.compute_likelihood()
    ref, _ = ot.likelihood()
    assert abs(lnl - ref) <= LNL_RTOL * abs(ref)


def test_ambiguity_codes_and_gaps(pkg, synth, oracle):
    model = synth.gtr_model()
    nwk = synth.random_tree_newick(10, 3)
    st = synth.simulate_alignment(nwk, model, 500, 4)
    rng = np.random.default_rng(1)
    m = rng.random(st.shape) < 0.35
    st[m] = rng.integers(4, 19, m.sum())
    st[3, :] = 18  # an all-gap sequence
    pat, freq = synth.compress_patterns(st)
    ot = oracle.OracleTree(nwk, 4, 0, pat, freq, None, model)
    t = pkg.PhyloTree(nwk)
    t.set_alignment(4, 0, pat, freq)
    t.set_model(model)
    t.attach_engine(0)
    lnl = t.compute_likelihood()
    ref, _ = ot.likelihood()
    assert abs(lnl - ref) <= LNL_RTOL * abs(ref)
    check_all_vectors(t, ot)


def test_invariant_sites(pkg, synth, oracle):
    t, ot, *_ = make_case(synth, oracle, pkg, 9, 900, 4, 4, 77, pinvar=0.25, hi=0.06)
    assert (ot.invar > 0).any()
    lnl = t.compute_likelihood()
    ref, _ = ot.likelihood()
    assert abs(lnl - ref) <= LNL_RTOL * abs(ref)


def test_scaling_counters_bit_exact_deep_tree(pkg, synth, oracle):
    """400-taxon caterpillar with long branches: every pattern is rescaled, several times."""
    t, ot, *_ = make_case(synth, oracle, pkg, 400, 300, 4, 4, 5, lo=0.4, hi=0.9, caterpillar=True)
    lnl = t.compute_likelihood()
    ref, (a, b) = ot.likelihood()
    _, sc, sf = ot.partial(a, b)
    assert sc.max() >= 2 and sf < 0
    assert abs(lnl - ref) <= LNL_RTOL * abs(ref)
    assert check_all_vectors(t, ot) == 400 - 2
    # computePatternLikelihood-style host view (phylotree.cpp:1059-1069)
    lnl2, plh = t.compute_likelihood(want_pattern_lh=True)
    _, oplh = ot.branch_lnl(a, b)
    np.testing.assert_allclose(plh, oplh + sc * oracle.lib().oracle_log_scaling_threshold(), rtol=1e-11)


def test_lazy_recompute_and_reroot(pkg, synth, oracle):
    """partials are cached; evaluating on other branches re-orients LM_PER_NODE buffers and
    gives the same lnL (pulley principle), both leaf and internal branch forms."""
    t, ot, *_ = make_case(synth, oracle, pkg, 12, 400, 4, 4, 31, missing=0.02)
    ref, _ = ot.likelihood()
    lnl = t.compute_likelihood()
    n0 = t.num_partial_lh_computations
    assert abs(t.compute_likelihood() - lnl) == 0.0          # nothing recomputed
    assert t.last_plan() == []
    for a in range(t.num_nodes):
        for b, _ in t.neighbors(a):
            v = t.compute_likelihood_branch(a, b)
            assert abs(v - ref) <= LNL_RTOL * abs(ref), (a, b)
            o, _ = ot.branch_lnl(a, b)
            assert abs(v - o) <= LNL_RTOL * abs(ref)
    assert t.num_partial_lh_computations > n0


def test_all_branch_memory_mode(pkg, synth, oracle):
    t, ot, *_ = make_case(synth, oracle, pkg, 11, 300, 4, 4, 41, mem_mode=1)
    ref, _ = ot.likelihood()
    for a in range(t.num_nodes):
        for b, _ in t.neighbors(a):
            v = t.compute_likelihood_branch(a, b)
            assert abs(v - ref) <= LNL_RTOL * abs(ref)
    assert check_all_vectors(t, ot) == 2 * (11 - 3) + 11  # every directed edge into an internal node


def test_derivatives_and_from_buffer(pkg, synth, oracle):
    t, ot, *_ = make_case(synth, oracle, pkg, 10, 800, 4, 4, 51, missing=0.03)
    t.compute_likelihood()
    for (a, b) in [(0, t.neighbors(0)[0][0]), (t.num_leaves, t.neighbors(t.num_leaves)[0][0]),
                   (t.num_leaves + 2, t.neighbors(t.num_leaves + 2)[1][0])]:
        t.reset_theta()
        df, ddf = t.compute_likelihood_derv(a, b)
        odf, oddf = ot.derv(a, b)
        assert abs(df - odf) <= 1e-9 * max(1.0, abs(odf)) + 1e-12 * abs(oddf)
        assert abs(ddf - oddf) <= 1e-9 * abs(oddf)
        v = t.compute_likelihood_from_buffer()
        o, _ = ot.lnl_from_theta(a, b)
        assert abs(v - o) <= LNL_RTOL * abs(o)
        # a Newton-style sequence of lengths re-uses theta (phylotree.cpp:2135)
        for length in (0.01, 0.13, 1.7):
            t.set_branch_length(a, b, length, clear_reverse=False)
            t.lib.iqhost_reset_theta  # (theta stays valid: only the length changed)
            # theta_computed was reset by set_branch_length in the mirror; recompute is harmless
            df, ddf = t.compute_likelihood_derv(a, b)
            odf, oddf = ot.derv(a, b, length)
            assert abs(df - odf) <= 1e-9 * max(1.0, abs(odf)) + 1e-12 * abs(oddf)
            assert abs(ddf - oddf) <= 1e-9 * abs(oddf)
        ot.clear()
        t.set_branch_length(a, b, ot.length(a, b), clear_reverse=True)


def test_branch_length_change_invalidates_reverse_partials(pkg, synth, oracle):
    t, ot, *_ = make_case(synth, oracle, pkg, 9, 300, 4, 4, 61)
    t.compute_likelihood()
    a = t.num_leaves + 1
    b = t.neighbors(a)[0][0]
    t.set_branch_length(a, b, 0.5, clear_reverse=True)
    ot.set_length(a, b, 0.5)
    t.clear_all_partial_lh()  # new current_it, as model optimisers do
    lnl = t.compute_likelihood()
    ref, _ = ot.likelihood()
    assert abs(lnl - ref) <= LNL_RTOL * abs(ref)


def test_branch_length_optimisation_increases_lnl(pkg, synth, oracle):
    """hot loop 2: Newton-Raphson on every branch through computeLikelihoodDerv."""
    t, ot, *_ = make_case(synth, oracle, pkg, 8, 600, 4, 4, 71)
    start = t.compute_likelihood()
    # perturb all lengths, then optimise
    for a in range(t.num_nodes):
        for b, _ in t.neighbors(a):
            if a < b:
                t.set_branch_length(a, b, 0.3, clear_reverse=False)
    t.clear_all_partial_lh()
    bad = t.compute_likelihood()
    opt = t.optimize_all_branches(iterations=20, tolerance=1e-4)
    assert opt > bad and opt >= start - 1e-6 * abs(start)
    # the optimised tree's lnL agrees with the oracle evaluated on the optimised lengths
    ot2 = oracle.OracleTree(t.tree_string(), 4, 0, ot.states, ot.freq, None, ot.model)
    ref, _ = ot2.likelihood()
    assert abs(opt - ref) <= 1e-8 * abs(ref)


def test_c_abi_direct_ops_and_errors(pkg, synth, oracle):
    """the raw ABI: explicit op list, key re-use, upload/fetch round trip, error statuses."""
    lib = pkg.libiqhip()
    model = synth.gtr_model()
    nwk = "((0:0.1,1:0.2):0.05,2:0.3,(3:0.1,4:0.15):0.07);"
    st = synth.simulate_alignment(nwk, model, 200, 3)
    freq = np.ones(200)
    ot = oracle.OracleTree(nwk, 4, 0, st, freq, None, model)
    e = C.c_void_p()
    assert lib.iqhip_create(C.byref(e), 0, 4, 4, 200, 5) == 0
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    # compute before inputs -> INVALID
    ops = (pkg.NodeOp * 2)()
    ops[0] = pkg.NodeOp(101, 0, 0, 0, 1, 0.1, 0.2)
    ops[1] = pkg.NodeOp(102, 0, 0, 3, 4, 0.1, 0.15)
    ss = np.zeros(2)
    assert lib.iqhip_update_partials(e, ops, 2, dp(ss)) == 2
    assert b"set_model" in lib.iqhip_last_error()
    tip = ot.tip
    assert lib.iqhip_set_model(e, dp(ot.eval), dp(ot.evec), dp(ot.inv_evec), dp(ot.rates), dp(ot.props), 18, dp(tip)) == 0
    assert lib.iqhip_set_alignment(e, st.ctypes.data_as(C.POINTER(C.c_uint8)), dp(freq), dp(np.zeros(200))) == 0
    assert lib.iqhip_update_partials(e, ops, 2, dp(ss)) == 0
    lnl = C.c_double()
    # root branch: internal(101) -- internal node 5 .. evaluate at leaf 2: needs vector of node 5 seen from 2
    ops3 = (pkg.NodeOp * 1)()
    ops3[0] = pkg.NodeOp(103, 101, 102, -1, -1, 0.05, 0.07)
    assert lib.iqhip_traverse_lnl(e, ops3, 1, pkg.leaf_end(2), pkg.key_end(103), 0.3, dp(ss), C.byref(lnl)) == 0
    ref, _ = ot.branch_lnl(2, 5)
    assert abs(lnl.value - ref) <= LNL_RTOL * abs(ref)
    # fetch / upload round trip under a new key
    buf = np.zeros(200 * 16)
    sc = np.zeros(200, dtype=np.int16)
    assert lib.iqhip_fetch_partial(e, 103, dp(buf)) == 0
    assert lib.iqhip_fetch_scale_num(e, 103, sc.ctypes.data_as(C.POINTER(C.c_int16))) == 0
    np.testing.assert_allclose(buf.reshape(200, 16), ot.partial(2, 5)[0], rtol=1e-10, atol=1e-300)
    assert lib.iqhip_upload_partial(e, 777, dp(buf), sc.ctypes.data_as(C.POINTER(C.c_int16))) == 0
    lnl2 = C.c_double()
    assert lib.iqhip_branch_lnl(e, pkg.leaf_end(2), pkg.key_end(777), 0.3, C.byref(lnl2)) == 0
    assert lnl2.value == lnl.value
    # errors: unknown key, bad leaf, two-leaf branch, negative length
    assert lib.iqhip_branch_lnl(e, pkg.leaf_end(2), pkg.key_end(999), 0.3, C.byref(lnl2)) == 2
    assert lib.iqhip_branch_lnl(e, pkg.leaf_end(9), pkg.key_end(103), 0.3, C.byref(lnl2)) == 2
    assert lib.iqhip_branch_lnl(e, pkg.leaf_end(1), pkg.leaf_end(2), 0.3, C.byref(lnl2)) == 2
    assert lib.iqhip_branch_lnl(e, pkg.leaf_end(2), pkg.key_end(103), -1.0, C.byref(lnl2)) == 2
    assert lib.iqhip_derv(e, 0.1, C.byref(lnl2), C.byref(lnl2)) == 2  # theta not computed
    # rekey / release
    assert lib.iqhip_rekey(e, 777, 778) == 0
    assert lib.iqhip_branch_lnl(e, pkg.leaf_end(2), pkg.key_end(778), 0.3, C.byref(lnl2)) == 0
    assert lib.iqhip_release(e, 778) == 0
    assert lib.iqhip_branch_lnl(e, pkg.leaf_end(2), pkg.key_end(778), 0.3, C.byref(lnl2)) == 2
    lib.iqhip_destroy(e)


def test_async_result_buffer_path(pkg, synth, oracle):
    """what the multi-GPU bench uses: results left on the device in a caller-owned buffer."""
    import torch
    t, ot, *_ = make_case(synth, oracle, pkg, 9, 500, 4, 4, 81)
    lib = pkg.libiqhip()
    eng = t.engine
    res = torch.zeros(64, dtype=torch.float64, device="cuda:0")
    assert lib.iqhip_bind_result_buffer(eng, C.c_void_p(res.data_ptr()), 64) == 0
    t.set_dry_run(True)
    t.clear_all_partial_lh()
    t.compute_likelihood()            # dry run: only to obtain the plan
    plan = t.last_plan()
    a, b = t.current_branch()
    t.set_dry_run(False)
    ops = (pkg.NodeOp * len(plan))()
    for k, p in enumerate(plan):
        ops[k] = pkg.NodeOp(p["dst_key"], p["left_key"], p["right_key"], p["left_leaf"], p["right_leaf"],
                            p["left_len"], p["right_len"])
    end_b = pkg.key_end(plan[-1]["dst_key"])
    assert lib.iqhip_traverse_lnl_async(eng, ops, len(plan), pkg.leaf_end(a), end_b, ot.length(a, b)) == 0
    assert lib.iqhip_synchronize(eng) == 0
    out = res.cpu().numpy()
    ref, _ = ot.likelihood()
    total = out[0] + out[2:2 + len(plan)].sum()
    assert abs(total - ref) <= LNL_RTOL * abs(ref)
    assert lib.iqhip_bind_result_buffer(eng, None, 0) == 0


def test_lds_chunking_of_long_plans(pkg, synth, oracle, monkeypatch):
    """plans whose per-branch LDS regions exceed the budget are cut into chunks (barrier + refill)."""
    monkeypatch.setenv("IQHIP_LDS_KB", "8")
    t, ot, *_ = make_case(synth, oracle, pkg, 40, 500, 4, 4, 91, missing=0.05)
    monkeypatch.delenv("IQHIP_LDS_KB")
    lnl = t.compute_likelihood()
    ref, _ = ot.likelihood()
    assert abs(lnl - ref) <= LNL_RTOL * abs(ref)
    assert check_all_vectors(t, ot) == 40 - 2


@pytest.mark.parametrize("wg", ["64", "128"])
def test_other_workgroup_sizes(pkg, synth, oracle, monkeypatch, wg):
    monkeypatch.setenv("IQHIP_WG", wg)
    t, ot, *_ = make_case(synth, oracle, pkg, 13, 777, 4, 4, 93, missing=0.05)
    monkeypatch.delenv("IQHIP_WG")
    lnl = t.compute_likelihood()
    ref, _ = ot.likelihood()
    assert abs(lnl - ref) <= LNL_RTOL * abs(ref)


# ------------------------------------------------------------------------------------------
# 20-state (protein) and 64-state (codon) matrix-core path
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,ncat,seq_type", [(20, 4, 1), (20, 1, 1), (64, 1, 2), (64, 2, 2), (20, 5, 1)])
def test_mfma_full_traversal(pkg, synth, oracle, n, ncat, seq_type):
    t, ot, *_ = make_case(synth, oracle, pkg, 11, 300, n, ncat, 200 + n + ncat, seq_type=seq_type, missing=0.05)
    t.clear_all_partial_lh()
    lnl = t.compute_likelihood()
    ref, (a, b) = ot.likelihood()
    assert abs(lnl - ref) <= LNL_RTOL * abs(ref)
    assert check_all_vectors(t, ot) == 11 - 2
    _, plh = ot.branch_lnl(a, b)
    np.testing.assert_allclose(t.fetch_pattern_lh(), plh, rtol=1e-10, atol=1e-10)
    # every branch, leaf and internal forms
    for x in range(t.num_nodes):
        for y, _ in t.neighbors(x):
            if x < y:
                v = t.compute_likelihood_branch(x, y)
                assert abs(v - ref) <= LNL_RTOL * abs(ref), (x, y)


def test_mfma_protein_ambiguity_states(pkg, synth, oracle):
    model = synth.random_reversible_model(20, 5, alpha=0.7, ncat=4)
    nwk = synth.random_tree_newick(9, 6)
    st = synth.simulate_alignment(nwk, model, 400, 7)
    rng = np.random.default_rng(2)
    m = rng.random(st.shape) < 0.2
    st[m] = rng.integers(20, 24, m.sum())  # B, Z, U and unknown
    pat, freq = synth.compress_patterns(st)
    ot = oracle.OracleTree(nwk, 20, 1, pat, freq, None, model)
    t = pkg.PhyloTree(nwk)
    t.set_alignment(20, 1, pat, freq)
    t.set_model(model)
    t.attach_engine(0)
    lnl = t.compute_likelihood()
    ref, _ = ot.likelihood()
    assert abs(lnl - ref) <= LNL_RTOL * abs(ref)
    check_all_vectors(t, ot)


@pytest.mark.parametrize("n,seq_type", [(20, 1), (64, 2)])
def test_mfma_scaling_counters_bit_exact(pkg, synth, oracle, n, seq_type):
    ncat = 4 if n == 20 else 1
    t, ot, *_ = make_case(synth, oracle, pkg, 150, 80, n, ncat, 300 + n, seq_type=seq_type, lo=0.3, hi=0.7,
                          caterpillar=True)
    lnl = t.compute_likelihood()
    ref, (a, b) = ot.likelihood()
    _, sc, sf = ot.partial(a, b)
    assert sc.max() >= 1 and sf < 0
    assert abs(lnl - ref) <= LNL_RTOL * abs(ref)
    assert check_all_vectors(t, ot) == 150 - 2


@pytest.mark.parametrize("n,ncat,seq_type", [(20, 4, 1), (64, 1, 2)])
def test_mfma_derivatives_and_from_buffer(pkg, synth, oracle, n, ncat, seq_type):
    t, ot, *_ = make_case(synth, oracle, pkg, 9, 350, n, ncat, 400 + n, seq_type=seq_type, missing=0.03)
    t.compute_likelihood()
    for (a, b) in [(0, t.neighbors(0)[0][0]), (t.num_leaves, t.neighbors(t.num_leaves)[1][0])]:
        t.reset_theta()
        df, ddf = t.compute_likelihood_derv(a, b)
        odf, oddf = ot.derv(a, b)
        assert abs(df - odf) <= 1e-9 * max(1.0, abs(odf)) + 1e-12 * abs(oddf)
        assert abs(ddf - oddf) <= 1e-9 * abs(oddf)
        v = t.compute_likelihood_from_buffer()
        o, _ = ot.lnl_from_theta(a, b)
        assert abs(v - o) <= LNL_RTOL * abs(o)
    opt = t.optimize_all_branches(iterations=3, tolerance=1e-3)
    ot2 = oracle.OracleTree(t.tree_string(), n, seq_type, ot.states, ot.freq, None, ot.model)
    ref, _ = ot2.likelihood()
    assert abs(opt - ref) <= 1e-8 * abs(ref)


def test_rccl_allreduce_hook_single_rank(pkg, synth, oracle):
    """The N>1 plumbing of bench.py on one GPU: engine on torch's stream, result vector left in a
    torch-owned device buffer, all-reduced with the nccl(=RCCL) backend (world_size 1) inside the
    host mirror's hook, then read back.  Must give the same numbers as the synchronous path."""
    import os
    import torch
    import torch.distributed as dist
    t, ot, *_ = make_case(synth, oracle, pkg, 12, 900, 4, 4, 111, missing=0.03)
    ref_lnl = t.compute_likelihood()
    a, b = t.current_branch()
    ref_df, ref_ddf = t.compute_likelihood_derv(a, b)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        lib = pkg.libiqhip()
        res = torch.zeros(2 + 4096, dtype=torch.float64, device="cuda")
        stream = torch.cuda.current_stream()
        assert lib.iqhip_set_stream(t.engine, C.c_void_p(stream.cuda_stream)) == 0
        assert lib.iqhip_bind_result_buffer(t.engine, C.c_void_p(res.data_ptr()), res.numel()) == 0
        calls = []

        def hook(ptr, n):
            assert ptr == res.data_ptr()
            calls.append(n)
            dist.all_reduce(res[:n], op=dist.ReduceOp.SUM)
        t.set_allreduce_hook(hook)
        for _ in range(3):
            t.clear_all_partial_lh()
            lnl = t.compute_likelihood()
            assert lnl == ref_lnl
        t.reset_theta()
        df, ddf = t.compute_likelihood_derv(a, b)
        assert (df, ddf) == (ref_df, ref_ddf)
        assert calls[:3] == [2 + 10] * 3 and calls[-1] == 2
        t.set_allreduce_hook(None)
        assert lib.iqhip_bind_result_buffer(t.engine, None, 0) == 0
    finally:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------
# Newton-Raphson branch-length solve on the device
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,ncat,seq_type,nsites", [(4, 4, 0, 150), (4, 4, 0, 6000), (20, 4, 1, 900), (64, 1, 2, 500)])
def test_device_newton_matches_host_loop(pkg, synth, oracle, n, ncat, seq_type, nsites):
    """iqhip_newton_branch restates minimizeNewton: it must take the same path as the host loop over
    computeLikelihoodDerv (same number of derivative evaluations, same optimum) -- single-workgroup
    grids (no barrier) and multi-workgroup grids (grid barrier) alike."""
    t, ot, *_ = make_case(synth, oracle, pkg, 9, nsites, n, ncat, 500 + n + nsites, seq_type=seq_type, missing=0.02)
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
        assert abs(ld - lh) <= 1e-9 * max(lh, 1e-6), (a, b, lh, ld)
        assert cd == ch, (a, b, ch, cd)
        # the optimum is a stationary point of the oracle's lnL as well (or sits on a bound)
        ot.set_length(a, b, ld)
        odf, _ = ot.derv(a, b)
        assert abs(odf) < 1e-3 * max(1.0, abs(ot.branch_lnl(a, b)[0])) or ld <= 1.1e-6


def test_device_newton_optimize_all_branches(pkg, synth, oracle):
    t, ot, *_ = make_case(synth, oracle, pkg, 16, 2500, 4, 4, 601)
    vals = {}
    for mode in (False, True):
        t2, _, *_ = make_case(synth, oracle, pkg, 16, 2500, 4, 4, 601)
        t2.set_device_newton(mode)
        for a in range(t2.num_nodes):
            for b, _ in t2.neighbors(a):
                if a < b:
                    t2.set_branch_length(a, b, 0.25, clear_reverse=False)
        t2.clear_all_partial_lh()
        vals[mode] = (t2.optimize_all_branches(iterations=10, tolerance=1e-4), t2.tree_string())
    assert abs(vals[True][0] - vals[False][0]) <= 1e-9 * abs(vals[False][0])
    ot2 = oracle.OracleTree(vals[True][1], 4, 0, ot.states, ot.freq, None, ot.model)
    ref, _ = ot2.likelihood()
    assert abs(vals[True][0] - ref) <= 1e-8 * abs(ref)


def test_reference_example_alignment_with_branch_optimisation(pkg, synth, oracle):
    """Real data (the reference's example.phy: 44 taxa, 355 patterns, many gaps): full traversal,
    then optimizeAllBranches with the device-side Newton loop; the optimised tree's lnL is
    re-evaluated by the oracle."""
    import os
    import phylip
    names, st = phylip.read_phylip_dna(os.path.join(os.path.dirname(__file__), "golden", "example.phy"))
    pat, freq = synth.compress_patterns(st)
    model = synth.gtr_model(rates6=(1.513, 2.393, 1.769, 1.912, 2.838, 1.0), freqs=(0.249, 0.262, 0.251, 0.238),
                            alpha=0.934, ncat=4)
    nwk = synth.random_tree_newick(44, 12)
    ot = oracle.OracleTree(nwk, 4, 0, pat, freq, None, model)
    t = pkg.PhyloTree(nwk)
    t.set_alignment(4, 0, pat, freq)
    t.set_model(model)
    t.attach_engine(0)
    lnl = t.compute_likelihood()
    ref, _ = ot.likelihood()
    assert abs(lnl - ref) <= LNL_RTOL * abs(ref)
    opt = t.optimize_all_branches(iterations=5, tolerance=1e-3)
    assert opt > lnl
    ot2 = oracle.OracleTree(t.tree_string(), 4, 0, pat, freq, None, model)
    ref2, _ = ot2.likelihood()
    assert abs(opt - ref2) <= 1e-8 * abs(ref2)


@pytest.mark.parametrize("n,ncat,seq_type", [(4, 4, 0), (20, 4, 1)])
def test_model_parameter_changes_between_evaluations(pkg, synth, oracle, n, ncat, seq_type):
    """hot loop 1 as the model optimisers run it (model/modelgtr.cpp:510-518, rategamma.cpp:160-169):
    new eigen-system / rates, clearAllPartialLH(), computeLikelihood() -- same tree, same plan
    (whose descriptors are not re-uploaded), different numbers every time."""
    t, ot, model, pat, freq = make_case(synth, oracle, pkg, 10, 400, n, ncat, 700 + n, seq_type=seq_type)
    seen = set()
    for trial, alpha in enumerate((0.9, 0.3, 2.5, 0.9)):
        if n == 4:
            m2 = synth.gtr_model(rates6=(1.0 + trial, 2.0, 0.7, 1.3, 3.1, 1.0), alpha=alpha, ncat=ncat)
        else:
            m2 = synth.random_reversible_model(n, 40 + trial, alpha=alpha, ncat=ncat)
        t.set_model(m2)
        t.clear_all_partial_lh()
        lnl = t.compute_likelihood()
        ot.set_model(m2)
        ref, _ = ot.likelihood()
        assert abs(lnl - ref) <= LNL_RTOL * abs(ref)
        seen.add(round(lnl, 6))
    assert len(seen) == 4


def test_pattern_frequency_reweighting(pkg, synth, oracle):
    """bootstrap-style re-weighting (iqhip_set_ptn_freq): same vectors, new pattern frequencies."""
    t, ot, model, pat, freq = make_case(synth, oracle, pkg, 9, 600, 4, 4, 801)
    lnl = t.compute_likelihood()
    rng = np.random.default_rng(5)
    f2 = rng.multinomial(int(freq.sum()), freq / freq.sum()).astype(np.float64)
    lib = pkg.libiqhip()
    assert lib.iqhip_set_ptn_freq(t.engine, f2.ctypes.data_as(C.POINTER(C.c_double))) == 0
    a, b = t.current_branch()
    v = t.compute_likelihood_branch(a, b)   # partials are frequency-independent (no scaling here)
    ot.freq = f2
    ot.clear()
    ref, _ = ot.branch_lnl(a, b)
    assert abs(v - ref) <= LNL_RTOL * abs(ref)
    assert abs(v - lnl) > 1e-6 * abs(lnl)


def test_many_taxa_plan_chunks_and_key_reuse(pkg, synth, oracle):
    """600 taxa: the plan exceeds one LDS chunk several times over; afterwards every key is released
    and the tree is evaluated again with recycled slabs."""
    t, ot, *_ = make_case(synth, oracle, pkg, 600, 70, 4, 4, 901, lo=0.01, hi=0.05)
    lnl = t.compute_likelihood()
    ref, _ = ot.likelihood()
    assert abs(lnl - ref) <= LNL_RTOL * abs(ref)
    lib = pkg.libiqhip()
    keys = {t.neighbor_info(a, b)["key"] for a in range(t.num_nodes) for b, _ in t.neighbors(a)} - {0}
    assert len(keys) == 598
    for k in keys:
        assert lib.iqhip_release(t.engine, k) == 0
    t.clear_all_partial_lh()
    assert abs(t.compute_likelihood() - lnl) <= 1e-12 * abs(lnl)


# ------------------------------------------------------------------------------------------
# +ASC: ascertainment-bias correction (phylokernel.h:655-725, 868-909, 968-1016, 1124-1187)
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,ncat,seq_type,pinvar", [(4, 4, 0, 0.0), (4, 4, 0, 0.15), (20, 4, 1, 0.0), (64, 1, 2, 0.0)])
def test_ascertainment_bias_correction(pkg, synth, oracle, n, ncat, seq_type, pinvar):
    if n == 4:
        model = synth.gtr_model(alpha=0.9, ncat=ncat, pinvar=pinvar)
    else:
        model = synth.random_reversible_model(n, 51, alpha=0.9 if ncat > 1 else None, ncat=ncat, pinvar=pinvar)
    nwk = synth.random_tree_newick(9, 52, 0.02, 0.15)
    st = synth.simulate_alignment(nwk, model, 500, 53)
    pat, freq = synth.compress_patterns(st)
    const = np.all(pat == pat[0][None, :], axis=0)
    pat, freq = np.ascontiguousarray(pat[:, ~const]), freq[~const].copy()
    nun, nsites = n, float(freq.sum())
    pat = np.ascontiguousarray(np.concatenate([pat, np.tile(np.arange(n, dtype=np.uint8)[None, :], (9, 1))], axis=1))
    freq = np.concatenate([freq, np.zeros(n)])
    invar = synth.ptn_invar_for(pat, model)
    ot = oracle.OracleTree(nwk, n, seq_type, pat, freq, invar, model, n_unobs=nun, nsites=nsites)
    t = pkg.PhyloTree(nwk)
    t.set_alignment(n, seq_type, pat, freq, invar)
    t.set_ascertainment(nun, nsites)
    t.set_model(model)
    t.attach_engine(0)
    lnl, plh = t.compute_likelihood(want_pattern_lh=True)
    ref, (a, b) = ot.likelihood()
    assert abs(lnl - ref) <= LNL_RTOL * abs(ref)
    _, oplh = ot.branch_lnl(a, b)
    np.testing.assert_allclose(plh[:-nun], oplh[:-nun], rtol=1e-10, atol=1e-10)
    # uncorrected value differs, so the correction really took part
    t.set_ascertainment(0, 0.0)
    t.clear_all_partial_lh()
    assert abs(t.compute_likelihood() - ref) > 1e-3
    t.set_ascertainment(nun, nsites)
    t.clear_all_partial_lh()
    t.compute_likelihood()
    for (x, y) in [(a, b), (t.num_leaves, t.neighbors(t.num_leaves)[0][0]), (t.num_leaves + 1, t.neighbors(t.num_leaves + 1)[1][0])]:
        v = t.compute_likelihood_branch(x, y)
        assert abs(v - ref) <= LNL_RTOL * abs(ref), (x, y)
        t.reset_theta()
        df, ddf = t.compute_likelihood_derv(x, y)
        odf, oddf = ot.derv(x, y)
        assert abs(df - odf) <= 1e-9 * max(1.0, abs(odf)) + 1e-12 * abs(oddf)
        assert abs(ddf - oddf) <= 1e-9 * abs(oddf)
        w = t.compute_likelihood_from_buffer()
        o, _ = ot.lnl_from_theta(x, y)
        assert abs(w - o) <= LNL_RTOL * abs(o)
    # branch optimisation (device Newton solve with the +ASC correction, tests/test_newton_oracle_gpu.py) improves the lnL
    assert t.optimize_all_branches(iterations=2, tolerance=1e-3) >= lnl - 1e-9 * abs(lnl)


@pytest.mark.parametrize("nni5", [False, True])
def test_nni_evaluation_on_scratch_buffers(pkg, synth, oracle, nni5):
    """getBestNNIForBran (phylotree.cpp:2873-3066): the search re-points neighbours at scratch buffers,
    swaps subtrees, re-optimises branches and restores everything.  Each move's lnL must be the lnL of
    the swapped topology (with the lengths the move reports) as the oracle computes it from scratch,
    and the tree must be intact afterwards."""
    t, ot, model, pat, freq = make_case(synth, oracle, pkg, 10, 500, 4, 4, 1001)
    base = t.compute_likelihood()
    ref0, _ = ot.likelihood()
    internal = [(a, b) for a in range(t.num_leaves, t.num_nodes) for b, _ in t.neighbors(a)
                if b >= t.num_leaves and a < b]
    assert internal
    for (a, b) in internal[:4]:
        moves = t.nni_for_branch(a, b, nni5=nni5)
        for (lnl, sa, sb, lens) in moves:
            # oracle on the swapped topology: subtree sa (was at a) <-> subtree sb (was at b)
            adj = {k: [list(e) for e in v] for k, v in ot.adj.items()}
            la = [e for e in adj[a] if e[0] == sa][0]
            lb = [e for e in adj[b] if e[0] == sb][0]
            adj[a].remove(la); adj[b].remove(lb)
            adj[a].append([sb, lb[1]]); adj[b].append([sa, la[1]])
            for e in adj[sa]:
                if e[0] == a: e[0] = b
            for e in adj[sb]:
                if e[0] == b: e[0] = a
            o2 = oracle.OracleTree("(0:1,1:1,2:1);", 4, 0, pat, freq, None, model)
            o2.adj = adj
            o2.ntaxa = ot.ntaxa
            o2.set_length(a, b, lens[0])
            if nni5:  # newLen[1..2]: branches of a (other than b) in stored order, [3..4]: of b
                k = 1
                # the mirror reports lengths in its neighbour order after the swap; recover by name
                na = [sb if x == sa else x for x, _ in t.neighbors(a) if x != b]
                nb_ = [sa if x == sb else x for x, _ in t.neighbors(b) if x != a]
                for x in na:
                    o2.set_length(a, x, lens[k]); k += 1
                for x in nb_:
                    o2.set_length(b, x, lens[k]); k += 1
            ref, _ = o2.branch_lnl(a, b)
            assert abs(lnl - ref) <= 1e-8 * abs(ref), (a, b, sa, sb, lnl, ref)
        # tree restored: same lnL, same branch lengths
        t.clear_all_partial_lh()
        assert abs(t.compute_likelihood() - base) <= 1e-12 * abs(base)
    assert abs(base - ref0) <= LNL_RTOL * abs(ref0)


# ------------------------------------------------------------------------------------------
# staged plans (engine.hip build_plan): units of independent subtrees + top stage
# ------------------------------------------------------------------------------------------

@pytest.mark.parametrize("n,ncat,seq_type,ntaxa,cat", [(4, 4, 0, 40, False), (4, 4, 0, 60, True), (20, 4, 1, 30, False),
                                                        (64, 1, 2, 20, False), (20, 2, 1, 24, False)])
@pytest.mark.parametrize("mem_mode", [0, 1])
def test_staged_plans_give_identical_results(pkg, synth, oracle, n, ncat, seq_type, ntaxa, cat, mem_mode, monkeypatch):
    """IQHIP_SPLIT=0 (one launch) against forced unit sizes: every vector, scale counter and
    lh_scale_factor must come out the same, the lnL to rounding (the per-op partial sums are the same
    numbers; only the launch structure differs)."""
    kw = dict(lo=0.3, hi=0.8, caterpillar=True) if cat else dict(missing=0.03)
    results = []
    for split in ("0", "3", "7", "1000"):
        monkeypatch.setenv("IQHIP_SPLIT", split)
        t, ot, *_ = make_case(synth, oracle, pkg, ntaxa, 333, n, ncat, 4000 + n + ntaxa, seq_type=seq_type,
                              mem_mode=mem_mode, **kw)
        lnl = t.compute_likelihood()
        a, b = t.current_branch()
        info = t.neighbor_info(a, b)
        vec = t.fetch_partial(a, b)
        sc = t.fetch_scale_num(a, b)
        if split == "0":
            ref, _ = ot.likelihood()
            assert abs(lnl - ref) <= LNL_RTOL * abs(ref)
            assert check_all_vectors(t, ot) == ntaxa - 2
        # partial re-evaluation after a branch change deep in the tree (short plans, lazy flags)
        inner = [(x, y) for x in range(t.num_nodes) for y, _ in t.neighbors(x) if x < y]
        x, y = inner[len(inner) // 2]
        t.set_branch_length(x, y, 0.21)
        lnl2 = t.compute_likelihood()
        results.append((lnl, info["lh_scale_factor"], vec, sc, lnl2))
    base = results[0]
    for r in results[1:]:
        assert abs(r[0] - base[0]) <= 1e-13 * abs(base[0]) and r[1] == base[1]
        assert np.array_equal(r[2], base[2]) and np.array_equal(r[3], base[3])
        assert abs(r[4] - base[4]) <= 1e-13 * abs(base[4])


def test_staged_plan_is_refused_for_reused_buffers(pkg, synth, oracle, monkeypatch):
    """LM_PER_NODE re-orientation inside one submission (a vector that is both an outside input and a
    destination) must not be re-ordered: evaluate on a far branch, then on the opposite side."""
    monkeypatch.setenv("IQHIP_SPLIT", "3")
    t, ot, *_ = make_case(synth, oracle, pkg, 30, 200, 4, 4, 515)
    ref, _ = ot.likelihood()
    assert abs(t.compute_likelihood() - ref) <= LNL_RTOL * abs(ref)
    leaves = [v for v in range(t.num_nodes) if ot.is_leaf(v)]
    for leaf in leaves[::5]:
        nb = t.neighbors(leaf)[0][0]
        v = t.compute_likelihood_branch(leaf, nb)      # re-roots: steals buffers of the old orientation
        assert abs(v - ref) <= LNL_RTOL * abs(ref)


@pytest.mark.parametrize("ncat,ntaxa,kw", [(4, 30, dict(missing=0.05)), (2, 40, dict(lo=0.3, hi=0.8, caterpillar=True)),
                                           (6, 12, dict(pinvar=0.2)), (8, 9, {})])
def test_lane_split_gives_identical_vectors(pkg, synth, oracle, ncat, ntaxa, kw, monkeypatch):
    """4-state kernel with two lanes per pattern (each lane half of the categories) against one lane per
    pattern: same vectors and counters bit for bit; the root lnL sums the two halves in another order."""
    out = []
    for ls in ("1", "2"):
        monkeypatch.setenv("IQHIP_LANE_SPLIT", ls)
        t, ot, *_ = make_case(synth, oracle, pkg, ntaxa, 500, 4, ncat, 9100 + ncat + ntaxa, **kw)
        lnl = t.compute_likelihood()
        a, b = t.current_branch()
        out.append((lnl, t.fetch_partial(a, b), t.fetch_scale_num(a, b), t.neighbor_info(a, b)["lh_scale_factor"],
                    t.compute_pattern_likelihood(), t.compute_likelihood_derv(a, b)))
        ref, _ = ot.likelihood()
        assert abs(lnl - ref) <= LNL_RTOL * abs(ref)
        assert check_all_vectors(t, ot) == ntaxa - 2
    x, y = out
    assert abs(x[0] - y[0]) <= 1e-13 * abs(x[0])
    assert np.array_equal(x[1], y[1]) and np.array_equal(x[2], y[2]) and x[3] == y[3]
    np.testing.assert_allclose(x[4], y[4], rtol=1e-13)
    assert x[5] == y[5]                     # theta / derivative kernels are not affected by the traversal mapping


def test_invariant_site_proportion_changes_on_an_attached_engine(pkg, synth, oracle):
    """+I optimisation (model/rateinvar.cpp -> computePtnInvar, phylotreesse.cpp:543-569): a new p_invar means new
    rates, proportions AND ptn_invar; only the model block and the two per-pattern weight arrays travel, the state
    rows stay on the device."""
    t, ot, model, pat, freq = make_case(synth, oracle, pkg, 10, 500, 4, 4, 1234, pinvar=0.1)
    ref, _ = ot.likelihood()
    assert abs(t.compute_likelihood() - ref) <= LNL_RTOL * abs(ref)
    for pinv in (0.3, 0.05):
        m2 = synth.gtr_model(alpha=0.9, ncat=4, pinvar=pinv)
        inv2 = synth.ptn_invar_for(pat, m2)
        t.set_model(m2)
        t.set_ptn_invar(inv2)
        t.clear_all_partial_lh()
        ot.set_model(m2)
        ot.invar = np.ascontiguousarray(inv2, dtype=np.float64)
        ot.clear()
        ref2, _ = ot.likelihood()
        assert abs(t.compute_likelihood() - ref2) <= LNL_RTOL * abs(ref2)
    f2 = np.roll(freq, 7)
    t.set_ptn_freq(f2)
    a, b = t.current_branch()
    ot.freq = np.ascontiguousarray(f2, dtype=np.float64)
    ot.clear()
    assert abs(t.compute_likelihood_branch(a, b) - ot.branch_lnl(a, b)[0]) <= LNL_RTOL * abs(ref2)


@pytest.mark.parametrize("ntaxa,kw", [(20, dict(missing=0.04)), (120, dict(lo=0.3, hi=0.7, caterpillar=True))])
def test_category_split_gives_identical_vectors(pkg, synth, oracle, ntaxa, kw, monkeypatch):
    """20-state kernel, one wave per category of a tile (small alignments) against one wave per tile: identical
    vectors and counters (the per-category arithmetic is the same; the scaling maximum crosses waves through LDS)."""
    out = []
    for cs in ("0", "1"):
        monkeypatch.setenv("IQHIP_CAT_SPLIT", cs)
        t, ot, *_ = make_case(synth, oracle, pkg, ntaxa, 200, 20, 4, 6100 + ntaxa, seq_type=1, **kw)
        lnl = t.compute_likelihood()
        a, b = t.current_branch()
        out.append((lnl, t.fetch_partial(a, b), t.fetch_scale_num(a, b), t.neighbor_info(a, b)["lh_scale_factor"]))
        ref, _ = ot.likelihood()
        assert abs(lnl - ref) <= LNL_RTOL * abs(ref)
        assert check_all_vectors(t, ot) == ntaxa - 2
    x, y = out
    assert x[0] == y[0] and np.array_equal(x[1], y[1]) and np.array_equal(x[2], y[2]) and x[3] == y[3]
    if "caterpillar" in kw:
        assert x[2].max() >= 1


@pytest.mark.parametrize("ntaxa,kw", [(16, dict(missing=0.04)), (90, dict(lo=0.3, hi=0.7, caterpillar=True))])
def test_row_split_gives_identical_vectors(pkg, synth, oracle, ntaxa, kw, monkeypatch):
    """64-state kernel, one wave per 16 output rows of a tile (small alignments) against one wave per tile."""
    out = []
    for rs in ("0", "1"):
        monkeypatch.setenv("IQHIP_ROW_SPLIT", rs)
        t, ot, *_ = make_case(synth, oracle, pkg, ntaxa, 150, 64, 1, 7100 + ntaxa, seq_type=2, **kw)
        lnl = t.compute_likelihood()
        a, b = t.current_branch()
        out.append((lnl, t.fetch_partial(a, b), t.fetch_scale_num(a, b), t.neighbor_info(a, b)["lh_scale_factor"]))
        ref, _ = ot.likelihood()
        assert abs(lnl - ref) <= LNL_RTOL * abs(ref)
        assert check_all_vectors(t, ot) == ntaxa - 2
    x, y = out
    assert x[0] == y[0] and np.array_equal(x[1], y[1]) and np.array_equal(x[2], y[2]) and x[3] == y[3]
    if "caterpillar" in kw:
        assert x[2].max() >= 1


@pytest.mark.parametrize("n,ncat,seq_type,ntaxa,nptn,kw", [(20, 4, 1, 18, 900, dict(missing=0.05)), (20, 1, 1, 10, 400, dict()),
                                                           (64, 1, 2, 12, 500, dict(missing=0.05)),
                                                           (20, 4, 1, 110, 200, dict(lo=0.3, hi=0.7, caterpillar=True))])
@pytest.mark.parametrize("tables", ["0", "1"])
def test_leaf_table_variants_of_the_matrix_core_kernels(pkg, synth, oracle, n, ncat, seq_type, ntaxa, nptn, kw, tables, monkeypatch):
    """IQHIP_LEAF_TABLES: leaf children as K2 table look-ups (k_leaf_tables; default for 64 states) or as
    U * (ex .* tip) products on the matrix pipe (default for 20 states) -- both against the oracle, including a model
    change and a branch-length change on the same engine (cached tables must be rebuilt, and only then)."""
    monkeypatch.setenv("IQHIP_LEAF_TABLES", tables)
    t, ot, model, pat, freq = make_case(synth, oracle, pkg, ntaxa, nptn, n, ncat, 8800 + n + ntaxa, seq_type=seq_type, **kw)
    if n == 20:   # protein ambiguity states B, Z, U take their own table rows
        rng = np.random.default_rng(3)
        pat = pat.copy()
        m = rng.random(pat.shape) < 0.03
        pat[m] = rng.integers(20, 23, m.sum())
        ot = oracle.OracleTree(t.tree_string(), n, seq_type, pat, freq, None, model)
        t = pkg.PhyloTree(t.tree_string())
        t.set_alignment(n, seq_type, pat, freq)
        t.set_model(model)
        t.attach_engine(0)
    lnl = t.compute_likelihood()
    ref, (a, b) = ot.likelihood()
    assert abs(lnl - ref) <= LNL_RTOL * abs(ref)
    assert check_all_vectors(t, ot) == ntaxa - 2
    assert t.clear_and_compute_likelihood() == lnl              # cached plan, cached tables
    # a pendant branch changes: exactly that leaf's table is stale
    leaf = 3
    nb = t.neighbors(leaf)[0][0]
    t.set_branch_length(leaf, nb, 0.37, clear_reverse=True)
    ot.set_length(leaf, nb, 0.37)
    t.clear_all_partial_lh()
    ref2, _ = ot.likelihood()
    assert abs(t.compute_likelihood() - ref2) <= LNL_RTOL * abs(ref2)
    # the model changes: every table is stale
    m2 = synth.random_reversible_model(n, 4242, alpha=0.6 if ncat > 1 else None, ncat=ncat)
    t.set_model(m2)
    ot.set_model(m2)
    t.clear_all_partial_lh()
    ref3, _ = ot.likelihood()
    assert abs(t.compute_likelihood() - ref3) <= LNL_RTOL * abs(ref3)
    assert check_all_vectors(t, ot) == ntaxa - 2


@pytest.mark.parametrize("ntaxa,nsites,kw", [(40, 9000, dict(missing=0.03)), (33, 5001, dict(lo=0.25, hi=0.6)),
                                             (64, 6000, dict(lo=0.3, hi=0.8, caterpillar=True))])
@pytest.mark.parametrize("env", [dict(), dict(IQHIP_TOP_CS2="0"), dict(IQHIP_HOLD_LDS="0", IQHIP_TOP_CS2="1")])
def test_protein_staged_plan_variants(pkg, synth, oracle, ntaxa, nsites, kw, env, monkeypatch):
    """20 states x 4 categories between the small-alignment forms and the BASELINE size: staged plans (units + a dependent
    top stage), results parked in LDS (CHILD_HOLD), and the top stage with two waves per tile (IQHIP_TOP_CS2) -- every
    vector, counter and the derivatives against the oracle, with ragged tile counts and rescaling trees."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("IQHIP_CAT_SPLIT", "0")
    t, ot, *_ = make_case(synth, oracle, pkg, ntaxa, nsites, 20, 4, 9900 + ntaxa, seq_type=1, **kw)
    assert t.nptn > 4096
    lnl = t.compute_likelihood()
    ref, (a, b) = ot.likelihood()
    assert abs(lnl - ref) <= LNL_RTOL * abs(ref)
    assert check_all_vectors(t, ot) == ntaxa - 2
    if "caterpillar" in kw:
        assert ot.partial(a, b)[1].max() >= 1
    df, ddf = t.compute_likelihood_derv(a, b)
    odf, oddf = ot.derv(a, b)
    assert abs(df - odf) <= 1e-9 * max(1.0, abs(odf)) + 1e-12 * abs(oddf)
    assert abs(ddf - oddf) <= 1e-9 * abs(oddf)
    # a second traversal from another root re-uses and re-parks
    x = t.num_leaves + 3
    y = t.neighbors(x)[0][0]
    assert abs(t.compute_likelihood_branch(x, y) - ref) <= LNL_RTOL * abs(ref)
