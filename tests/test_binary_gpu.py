"""GPU parity for binary (2-state) data -- the reference's `case 2` of setLikelihoodKernel (phylotreesse.cpp:262-276,
<Vec2d, 2, 2>).  The engine runs it on the 4-state kernels through an exact embedding (iqhip_internal.h, embed2): the
tests check lnL, every vector, the scaling counters (bit-exact), derivatives and the Newton solve against the oracle's
Vec2d restatement, with missing characters (the embedded form of STATE_UNKNOWN) and invariant sites present."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LNL_RTOL = 1e-9
SEQ_BINARY = 3


def make_case(synth, oracle, pkg, ntaxa, nsites, ncat, seed, missing=0.0, pinvar=0.0, lo=0.02, hi=0.2,
              caterpillar=False, sharded=0):
    model = synth.random_reversible_model(2, seed, alpha=0.7, ncat=ncat, pinvar=pinvar)
    nwk = synth.random_tree_newick(ntaxa, seed, lo, hi, caterpillar)
    st = synth.simulate_alignment(nwk, model, nsites, seed + 1, missing, 2)
    pat, freq = synth.compress_patterns(st)
    invar = synth.ptn_invar_for(pat, model)
    ot = oracle.OracleTree(nwk, 2, SEQ_BINARY, pat, freq, invar, model)
    t = pkg.PhyloTree(nwk)
    t.set_alignment(2, SEQ_BINARY, pat, freq, invar)
    t.set_model(model)
    t.set_likelihood_kernel(pkg.LK_EIGEN_HIP)
    if sharded:
        t.attach_engine_sharded([0] * sharded, pkg.REDUCE_HOST)
    else:
        t.attach_engine(0)
    return t, ot, model, pat, freq


def check_all_vectors(t, ot):
    n = 0
    for a in range(t.num_nodes):
        for b, _ in t.neighbors(a):
            info = t.neighbor_info(a, b)
            if ot.is_leaf(b) or not (info["computed"] & 1) or info["key"] == 0:
                continue
            plh, sc, sf = ot.partial(a, b)
            got = t.fetch_partial(a, b)
            assert got.shape == plh.shape
            scale = np.abs(plh).max(axis=1, keepdims=True)
            np.testing.assert_allclose(got / scale, plh / scale, rtol=0, atol=1e-10)
            assert np.array_equal(t.fetch_scale_num(a, b), sc)
            assert abs(info["lh_scale_factor"] - sf) <= 1e-12 * max(1.0, abs(sf))
            n += 1
    return n


@pytest.mark.parametrize("ncat", [1, 2, 4, 8])
def test_binary_full_traversal(pkg, synth, oracle, ncat):
    # (few distinct binary patterns exist for few taxa: 14 taxa x 3000 sites gives several hundred)
    t, ot, _, pat, _ = make_case(synth, oracle, pkg, 14, 3000, ncat, 40 + ncat, missing=0.1)
    assert (pat == 2).any() and pat.shape[1] > 200
    t.clear_all_partial_lh()
    lnl = t.compute_likelihood()
    ref, (a, b) = ot.likelihood()
    assert abs(lnl - ref) <= LNL_RTOL * abs(ref)
    assert check_all_vectors(t, ot) == 14 - 2
    _, plh = ot.branch_lnl(a, b)
    np.testing.assert_allclose(t.fetch_pattern_lh(), plh, rtol=1e-11, atol=1e-11)


def test_binary_invariant_sites_and_every_branch(pkg, synth, oracle):
    t, ot, *_ = make_case(synth, oracle, pkg, 10, 2500, 4, 7, missing=0.05, pinvar=0.2, hi=0.08)
    assert (ot.invar > 0).any()
    ref, _ = ot.likelihood()
    assert abs(t.compute_likelihood() - ref) <= LNL_RTOL * abs(ref)
    for a in range(t.num_nodes):
        for b, _ in t.neighbors(a):
            if a < b:
                got = t.compute_likelihood_branch(a, b)
                assert abs(got - ref) <= LNL_RTOL * abs(ref)


def test_binary_scaling_counters_deep_tree(pkg, synth, oracle):
    t, ot, *_ = make_case(synth, oracle, pkg, 600, 4000, 4, 3, lo=0.3, hi=0.8, caterpillar=True)
    lnl = t.compute_likelihood()
    ref, (a, b) = ot.likelihood()
    _, sc, sf = ot.partial(a, b)
    assert sc.max() >= 1 and sf < 0
    assert abs(lnl - ref) <= LNL_RTOL * abs(ref)
    assert check_all_vectors(t, ot) == 600 - 2


def test_binary_derivatives_theta_and_newton(pkg, synth, oracle):
    t, ot, *_ = make_case(synth, oracle, pkg, 12, 3000, 4, 17, missing=0.05)
    t.compute_likelihood()
    for (a, b) in [(0, t.neighbors(0)[0][0]), (t.num_leaves, t.neighbors(t.num_leaves)[0][0])]:
        t.reset_theta()
        df, ddf = t.compute_likelihood_derv(a, b)
        odf, oddf = ot.derv(a, b)
        assert abs(df - odf) <= 1e-9 * max(1.0, abs(odf)) + 1e-12 * abs(oddf)
        assert abs(ddf - oddf) <= 1e-9 * abs(oddf)
        v = t.compute_likelihood_from_buffer()
        o, _ = ot.lnl_from_theta(a, b)
        assert abs(v - o) <= LNL_RTOL * abs(o)
        for length in (0.01, 0.13, 1.7):
            t.set_branch_length(a, b, length, clear_reverse=False)
            df, ddf = t.compute_likelihood_derv(a, b)
            odf, oddf = ot.derv(a, b, length)
            assert abs(df - odf) <= 1e-9 * max(1.0, abs(odf)) + 1e-12 * abs(oddf)
            assert abs(ddf - oddf) <= 1e-9 * abs(oddf)
        ot.clear()
        t.set_branch_length(a, b, ot.length(a, b), clear_reverse=True)
    # branch optimisation: lnL does not fall, and agrees with the oracle evaluated at the optimised lengths
    before = t.compute_likelihood()
    after = t.optimize_all_branches(iterations=3, tolerance=1e-3)
    assert after >= before - 1e-7 * abs(before)
    ot2 = oracle.OracleTree(t.tree_string(), 2, SEQ_BINARY, ot.states, ot.freq, ot.invar, ot.model)
    ref, _ = ot2.likelihood()
    assert abs(after - ref) <= 1e-8 * abs(ref)


def test_binary_ragged_pattern_counts(pkg, synth, oracle):
    model = synth.random_reversible_model(2, 3, alpha=0.8, ncat=4)
    nwk = synth.random_tree_newick(9, 5)
    st = synth.simulate_alignment(nwk, model, 400, 9, 0.1, 2)
    for nptn in (1, 63, 65, 257):
        sub = st[:, :nptn]
        freq = np.arange(1, nptn + 1, dtype=np.float64)
        ot = oracle.OracleTree(nwk, 2, SEQ_BINARY, sub, freq, None, model)
        t = pkg.PhyloTree(nwk)
        t.set_alignment(2, SEQ_BINARY, sub, freq)
        t.set_model(model)
        t.attach_engine(0)
        ref, _ = ot.likelihood()
        assert abs(t.compute_likelihood() - ref) <= LNL_RTOL * abs(ref)


def test_binary_sharded_host_sum(pkg, synth, oracle):
    t, ot, *_ = make_case(synth, oracle, pkg, 16, 6000, 4, 23, missing=0.05, sharded=3)
    ref, (a, b) = ot.likelihood()
    assert abs(t.compute_likelihood() - ref) <= LNL_RTOL * abs(ref)
    assert check_all_vectors(t, ot) == 16 - 2
    df, ddf = t.compute_likelihood_derv(a, b)
    odf, oddf = ot.derv(a, b)
    assert abs(df - odf) <= 1e-9 * max(1.0, abs(odf)) + 1e-12 * abs(oddf)
    assert abs(ddf - oddf) <= 1e-9 * abs(oddf)


def test_binary_mixture_is_refused(pkg, synth):
    mix = synth.mixture_model(2, 2, 5, ncat=2)
    t = pkg.PhyloTree(synth.random_tree_newick(6, 1))
    st = np.zeros((6, 70), dtype=np.uint8)
    t.set_alignment(2, SEQ_BINARY, st, np.ones(70))
    t.set_model(mix)
    with pytest.raises(Exception):
        t.attach_engine(0)
        t.compute_likelihood()
