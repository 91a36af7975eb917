"""Device-side consumers of the per-pattern lnL: computePatternLikelihood
(phylotree.cpp:1200-1230) and UFBoot's RELL dot products (iqtree.cpp:2726-2736, phylokernel.h:55-61).
The oracle restates the reference's float / eight-lane dot product bit-exactly (tests/test_vcl_probe.py
pins it to the reference's Vec8f); the device accumulates in double, so it must agree with an exact
float64 dot product to rounding and with the reference's float result to float accuracy."""
import ctypes as C

import numpy as np
import pytest

from test_parity_gpu import make_case

pytestmark = pytest.mark.gpu


def boot_samples(rng, freq, nsamples):
    """multinomial resampling of sites into patterns, as UFBoot's boot_samples (float counts)."""
    p = np.asarray(freq, dtype=np.float64)
    return rng.multinomial(int(p.sum()), p / p.sum(), size=nsamples).astype(np.float32)


@pytest.mark.parametrize("n,ncat,seq_type,ntaxa,deep", [(4, 4, 0, 12, False), (4, 4, 0, 200, True), (20, 4, 1, 10, False),
                                                         (20, 4, 1, 120, True), (64, 1, 2, 8, False)])
def test_pattern_likelihood_and_rell(pkg, synth, oracle, n, ncat, seq_type, ntaxa, deep):
    kw = dict(lo=0.4, hi=0.9, caterpillar=True) if deep else dict(missing=0.02)
    t, ot, model, pat, freq = make_case(synth, oracle, pkg, ntaxa, 400, n, ncat, 900 + n + ntaxa, seq_type=seq_type, **kw)
    lnl = t.compute_likelihood()
    ref, (a, b) = ot.likelihood()
    assert abs(lnl - ref) <= 1e-9 * abs(ref)
    _, oplh = ot.branch_lnl(a, b)
    _, sc_b, _ = ot.partial(a, b)                  # a is the leaf end: only b's subtree carries scaling events
    if deep:
        assert sc_b.max() >= 1
    expect = oracle.pattern_lh_scaled(oplh, None, sc_b)
    got = t.compute_pattern_likelihood()
    np.testing.assert_allclose(got, expect, rtol=1e-11, atol=0)
    assert abs(np.dot(got, freq) - lnl) <= 1e-9 * abs(lnl)          # the check commented out at phylotree.cpp:1257-1265
    # RELL
    rng = np.random.default_rng(3)
    w = boot_samples(rng, freq, 100)
    w[0] = np.asarray(freq, dtype=np.float32)                       # the original alignment as sample 0
    t.set_boot_samples(w)
    rell = t.compute_rell()
    exact = w.astype(np.float64) @ got
    np.testing.assert_allclose(rell, exact, rtol=1e-13)
    assert abs(rell[0] - lnl) <= 1e-9 * abs(lnl)
    pad = (-got.size) % 8
    x8 = np.concatenate([got.astype(np.float32), np.zeros(pad, np.float32)])
    for s in (0, 1, 57, 99):
        ref32 = oracle.dot_float8(x8, np.concatenate([w[s], np.zeros(pad, np.float32)]))
        assert abs(rell[s] - ref32) <= 2e-5 * abs(ref32)            # float accumulation of ~400 terms in the reference
    # a second tree evaluation reuses the uploaded samples
    t.set_branch_length(a, b, 0.33)
    ot.set_length(a, b, 0.33)
    lnl2 = t.compute_likelihood()
    assert abs(t.compute_rell()[0] - lnl2) <= 1e-9 * abs(lnl2)


def test_rell_internal_branch_and_errors(pkg, synth, oracle):
    t, ot, model, pat, freq = make_case(synth, oracle, pkg, 400, 200, 4, 4, 77, lo=0.5, hi=0.9, caterpillar=True)
    lib = pkg.libiqhip()
    lnl = t.compute_likelihood()
    # evaluate on an internal branch through the C ABI: both ends carry scale counters
    inner = [(a, b) for a in range(t.num_nodes) for b, _ in t.neighbors(a) if not ot.is_leaf(a) and not ot.is_leaf(b)]
    a, b = next((a, b) for a, b in inner if ot.partial(a, b)[1].max() >= 1 and ot.partial(b, a)[1].max() >= 1)
    v = t.compute_likelihood_branch(a, b)
    assert abs(v - lnl) <= 1e-9 * abs(lnl)
    ia, ib = t.neighbor_info(a, b), t.neighbor_info(b, a)
    out = np.zeros(t.nptn)
    dp = out.ctypes.data_as(C.POINTER(C.c_double))
    assert lib.iqhip_fetch_pattern_lh_scaled(t.engine, pkg.key_end(ia["key"]), pkg.key_end(ib["key"]), dp) == 0
    _, sc_ab, _ = ot.partial(a, b)
    _, sc_ba, _ = ot.partial(b, a)
    assert sc_ab.max() >= 1 and sc_ba.max() >= 1
    _, oplh = ot.branch_lnl(a, b)
    np.testing.assert_allclose(out, oracle.pattern_lh_scaled(oplh, sc_ab, sc_ba), rtol=1e-11)
    assert abs(np.dot(out, freq) - lnl) <= 1e-9 * abs(lnl)
    # errors
    r = np.zeros(4)
    assert lib.iqhip_rell(t.engine, pkg.key_end(ia["key"]), pkg.key_end(ib["key"]), r.ctypes.data_as(C.POINTER(C.c_double))) != 0
    assert b"no bootstrap samples" in lib.iqhip_last_error()
    assert lib.iqhip_fetch_pattern_lh_scaled(t.engine, pkg.key_end(0xdeadbeef), pkg.key_end(ib["key"]), dp) != 0
    assert lib.iqhip_set_boot_samples(t.engine, None, 3) != 0


def test_rell_with_ascertainment(pkg, synth, oracle):
    """+ASC: the unobserved constant patterns are not sites; their pattern lnL and weight are zero."""
    model = synth.gtr_model(alpha=0.7, ncat=4)
    nwk = synth.random_tree_newick(9, 5)
    st = synth.simulate_alignment(nwk, model, 500, 6)
    st = st[:, [s for s in range(st.shape[1]) if len(set(st[:, s].tolist())) > 1]]
    pat, freq = synth.compress_patterns(st)
    nsite = int(freq.sum())
    pat = np.concatenate([pat, np.tile(np.arange(4, dtype=np.uint8), (9, 1))], axis=1)
    freq = np.concatenate([freq, np.zeros(4)])
    t = pkg.PhyloTree(nwk)
    t.set_alignment(4, 0, pat, freq)
    t.set_ascertainment(4, nsite)
    t.set_model(model)
    t.attach_engine(0)
    lnl = t.compute_likelihood()
    ot = oracle.OracleTree(nwk, 4, 0, pat, freq, None, model, n_unobs=4, nsites=nsite)
    ref, (a, b) = ot.likelihood()
    assert abs(lnl - ref) <= 1e-9 * abs(ref)
    plh = t.compute_pattern_likelihood()
    assert np.all(plh[-4:] == 0.0) and abs(np.dot(plh, freq) - lnl) <= 1e-9 * abs(lnl)
    t.set_boot_samples(np.vstack([freq, freq * 2]).astype(np.float32))
    r = t.compute_rell()
    assert abs(r[0] - lnl) <= 1e-9 * abs(lnl) and abs(r[1] - 2 * lnl) <= 1e-9 * abs(lnl)


def test_rell_throughput_shape(pkg, synth, oracle):
    """1000 samples x 20k patterns (UFBoot's default sample count): one launch, results identical
    run to run (fixed-order reduction)."""
    t, ot, model, pat, freq = make_case(synth, oracle, pkg, 8, 20000, 4, 4, 4242)
    t.compute_likelihood()
    rng = np.random.default_rng(9)
    w = boot_samples(rng, freq, 1000)
    t.set_boot_samples(w)
    r1 = t.compute_rell()
    r2 = t.compute_rell()
    assert np.array_equal(r1, r2)
    np.testing.assert_allclose(r1, w.astype(np.float64) @ t.compute_pattern_likelihood(), rtol=1e-13)


@pytest.mark.parametrize("n,ncat,seq_type", [(4, 4, 0), (20, 4, 1), (64, 1, 2)])
def test_pattern_lh_cat(pkg, synth, oracle, n, ncat, seq_type):
    """_pattern_lh_cat of the scalar kernels (phylotreesse.cpp:1190-1237), the input of the empirical-Bayes
    site rates (model/rategamma.cpp:241-262): per category likelihoods of the current branch."""
    t, ot, model, pat, freq = make_case(synth, oracle, pkg, 10, 300, n, ncat, 5100 + n, seq_type=seq_type, missing=0.02)
    lnl = t.compute_likelihood()
    a, b = t.current_branch()
    cat = t.compute_pattern_lh_cat()
    th, _ = ot.theta(a, b)
    ln = ot.length(a, b)
    val = (np.exp(np.outer(model.rates * ln, model.eval)) * model.props[:, None]).reshape(-1)     # [c][i]
    expect = (th * val[None, :]).reshape(th.shape[0], ncat, n).sum(axis=2)
    np.testing.assert_allclose(cat, expect, rtol=1e-10, atol=1e-300)
    _, oplh = ot.branch_lnl(a, b)
    np.testing.assert_allclose(np.log(cat.sum(axis=1)), oplh, rtol=1e-9)      # no invariant sites in this model
