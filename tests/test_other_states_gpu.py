"""GPU: state counts outside the reference's SIMD dispatch cases (morphological / multi-state data: 3, 5..19, 21..63 states).
The reference hands them to its scalar kernels (phylotreesse.cpp:281-309 -> :581-1339); the engine runs them on the next
kernel size up (4 / 20 / 64 states) through the exact embedding that carries binary data -- U = diag(U_m, I), zero
eigenvalues and zero tip components in the padding, missing characters as an ambiguity code with the caller's unknown row --
and applies the scalar kernel's scaling rule at every node (iqhip_engine::scalar_rule_all).  Checked against the oracle's
restatement of the scalar kernels (plain running sums; node updates through oracle_partial_update_multi), itself checked
against the probability-space recursion in tests/test_oracle.py."""
import numpy as np
import pytest

from test_parity_gpu import check_all_vectors, LNL_RTOL

pytestmark = pytest.mark.gpu

SEQ_OTHER = 3


def build(pkg, synth, oracle, n, ncat, ntaxa, nsites, seed, multif=False, **kw):
    model = synth.random_reversible_model(n, seed, alpha=0.9 if ncat > 1 else None, ncat=ncat)
    nwk = synth.random_multifurcating_newick(ntaxa, seed) if multif else synth.random_tree_newick(ntaxa, seed, **kw)
    st = synth.simulate_alignment(nwk, model, nsites, seed + 1, 0.04, n)   # 4 % missing characters (state n)
    pat, freq = synth.compress_patterns(st)
    ot = oracle.OracleTree(nwk, n, SEQ_OTHER, pat, freq, None, model)
    t = pkg.PhyloTree(nwk)
    t.set_alignment(n, SEQ_OTHER, pat, freq)
    t.set_model(model)
    t.set_likelihood_kernel(pkg.LK_EIGEN_HIP)
    t.attach_engine(0)
    return t, ot


@pytest.mark.parametrize("n,ncat,ntaxa,nsites", [(3, 4, 12, 400), (3, 1, 9, 70000), (5, 4, 14, 600), (7, 3, 10, 500), (12, 4, 10, 700),
                                                 (19, 1, 9, 300), (21, 1, 10, 400), (33, 1, 8, 300), (61, 1, 9, 2500)])
def test_other_state_counts_against_the_scalar_oracle(pkg, synth, oracle, n, ncat, ntaxa, nsites):
    t, ot = build(pkg, synth, oracle, n, ncat, ntaxa, nsites, 7700 + n)
    lnl = t.compute_likelihood()
    ref, (a, b) = ot.likelihood()
    assert t.current_branch() == (a, b)
    assert abs(lnl - ref) <= LNL_RTOL * abs(ref)
    assert check_all_vectors(t, ot) == ntaxa - 2          # caller-shaped vectors (n doubles per category), counters bit-exact
    for (x, y) in [(a, b), (t.num_leaves, t.neighbors(t.num_leaves)[0][0])]:
        assert abs(t.compute_likelihood_branch(x, y) - ref) <= LNL_RTOL * abs(ref)
        t.reset_theta()
        df, ddf = t.compute_likelihood_derv(x, y)
        odf, oddf = ot.derv(x, y)
        assert abs(df - odf) <= 1e-8 * max(1.0, abs(odf)) + 1e-11 * abs(oddf)
        assert abs(ddf - oddf) <= 1e-8 * abs(oddf)
        w = t.compute_likelihood_from_buffer()
        o, _ = ot.lnl_from_theta(x, y)
        assert abs(w - o) <= LNL_RTOL * abs(o)
    # branch optimisation (one submission per sweep for 4-state-sized engines, per-step launches for the others)
    opt = t.optimize_all_branches(iterations=2, tolerance=1e-6)
    assert opt >= lnl - 1e-9 * abs(lnl)
    ot2 = oracle.OracleTree(t.tree_string(), n, SEQ_OTHER, ot.states, ot.freq, None, ot.model)
    ref2, _ = ot2.likelihood()
    assert abs(opt - ref2) <= 1e-8 * abs(ref2)


@pytest.mark.parametrize("n,ncat", [(3, 4), (6, 2), (25, 1)])
def test_other_state_counts_deep_trees_rescale(pkg, synth, oracle, n, ncat):
    """long caterpillar: every pattern is rescaled along the way; the scalar kernel's rule at every node"""
    t, ot = build(pkg, synth, oracle, n, ncat, {3: 360, 6: 220, 25: 90}[n], 250, 8800 + n, lo=0.4, hi=0.9, caterpillar=True)
    lnl = t.compute_likelihood()
    ref, (a, b) = ot.likelihood()
    frm, to = (a, b) if not ot.is_leaf(b) else (b, a)
    plh, sc, sf = ot.partial(frm, to)
    assert sc.max() >= 1 and sf < 0
    assert abs(lnl - ref) <= LNL_RTOL * abs(ref)
    assert np.array_equal(t.fetch_scale_num(frm, to), sc)
    assert abs(t.neighbor_info(frm, to)["lh_scale_factor"] - sf) <= 1e-12 * abs(sf)


def test_other_state_counts_with_polytomies_and_sharding(pkg, synth, oracle):
    from test_sharded_gpu import sharded_tree
    n, ncat = 5, 4
    t, ot = build(pkg, synth, oracle, n, ncat, 14, 9000, 9300, multif=True)
    ref, _ = ot.likelihood()
    assert abs(t.compute_likelihood() - ref) <= LNL_RTOL * abs(ref)
    ts = sharded_tree(pkg, (t.tree_string(), n, SEQ_OTHER, ot.states, ot.freq, None, ot.model), [0, 0], pkg.REDUCE_HOST)
    assert abs(ts.compute_likelihood() - ref) <= LNL_RTOL * abs(ref)


def test_state_count_limits(pkg):
    import ctypes as C
    lib = pkg.libiqhip()
    e = C.c_void_p()
    assert lib.iqhip_create(C.byref(e), 0, 1, 4, 100, 5) == 3 and lib.iqhip_create(C.byref(e), 0, 65, 1, 100, 5) == 3
    assert lib.iqhip_create(C.byref(e), 0, 3, 9, 100, 5) == 3       # 3 states run on the 4-state kernels: <= 8 categories
    assert lib.iqhip_create(C.byref(e), 0, 5, 9, 100, 5) == 0       # 5 states on the 20-state kernels: <= 96
    lib.iqhip_destroy(e)
