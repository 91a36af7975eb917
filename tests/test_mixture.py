"""Mixture models (phylokernelmixture.h:20-460, phylokernelmixrate.h:22-450): the partial vectors carry
nstates x (class, rate) components, every component with the eigen-system of its class.  CPU: the oracle's
mixture mode against an independent per-class evaluation (sum over classes of plain likelihoods).  GPU: the
matrix-core mixture kernel against the oracle."""
import numpy as np
import pytest

import textbook


def textbook_mixture(ot, model, seq_type):
    per = np.stack([textbook.site_log_likelihoods(ot.adj, ot.states, m, seq_type, ot.su) for m in model.classes])
    mx = per.max(axis=0)
    return mx + np.log(np.exp(per - mx[None, :]).sum(axis=0))


def make_mix(synth, oracle, n, nclass, ncat, fused, ntaxa, nptn, seed, seq_type, missing=0.02, **tree_kw):
    model = synth.mixture_model(n, nclass, seed, ncat=ncat, fused=fused)
    su = oracle.state_unknown_for(n, seq_type)
    nwk = synth.random_tree_newick(ntaxa, seed, **tree_kw)
    st = synth.simulate_alignment(nwk, model.classes[0], nptn, seed + 1, missing, su)
    pat, freq = synth.compress_patterns(st)
    ot = oracle.OracleTree(nwk, n, seq_type, pat, freq, None, model)
    return model, nwk, pat, freq, ot


@pytest.mark.parametrize("n,seq_type,nclass,ncat,fused", [(4, 0, 3, 4, False), (20, 1, 3, 2, False), (20, 1, 4, 1, True)])
def test_oracle_mixture_matches_per_class_evaluation(synth, oracle, n, seq_type, nclass, ncat, fused):
    model, nwk, pat, freq, ot = make_mix(synth, oracle, n, nclass, ncat, fused, 9, 150, 60 + n + nclass, seq_type)
    assert model.ncat == (nclass if fused else nclass * ncat) and abs(model.props.sum() - 1) < 1e-12
    lnl, (a, b) = ot.likelihood()
    _, plh = ot.branch_lnl(a, b)
    ref = textbook_mixture(ot, model, seq_type)
    np.testing.assert_allclose(plh, ref, rtol=1e-9)
    assert abs(lnl - np.dot(ref, freq)) <= 1e-9 * abs(lnl)
    # an interleaved plain-model tree must not see the mixture context
    plain = synth.gtr_model()
    st = synth.simulate_alignment(nwk, plain, 60, 5)
    p2, f2 = synth.compress_patterns(st)
    o2 = oracle.OracleTree(nwk, 4, 0, p2, f2, None, plain)
    v2, _ = o2.likelihood()
    assert abs(v2 - np.dot(textbook.site_log_likelihoods(o2.adj, o2.states, plain, 0, 18), f2)) <= 1e-9 * abs(v2)
    # derivatives of the mixture lnL by finite differences
    df, ddf = ot.derv(a, b)
    t0, h = ot.length(a, b), 1e-5
    fp, fm = ot.branch_lnl(a, b, t0 + h)[0], ot.branch_lnl(a, b, t0 - h)[0]
    assert abs(df - (fp - fm) / (2 * h)) <= 1e-5 * max(1.0, abs(df))


@pytest.mark.gpu
@pytest.mark.parametrize("nclass,ncat,fused,ntaxa,kw", [(3, 4, False, 12, {}), (4, 1, True, 15, {}), (10, 4, False, 8, {}),
                                                       (2, 3, False, 150, dict(lo=0.4, hi=0.9, caterpillar=True))])
def test_hip_mixture_matches_oracle(pkg, synth, oracle, nclass, ncat, fused, ntaxa, kw):
    from test_parity_gpu import check_all_vectors, LNL_RTOL
    model, nwk, pat, freq, ot = make_mix(synth, oracle, 20, nclass, ncat, fused, ntaxa, 300, 800 + nclass + ntaxa, 1, **kw)
    t = pkg.PhyloTree(nwk)
    t.set_alignment(20, 1, pat, freq)
    t.set_model(model)
    t.attach_engine(0)
    lnl = t.compute_likelihood()
    ref, (a, b) = ot.likelihood()
    assert abs(lnl - ref) <= LNL_RTOL * abs(ref)
    assert check_all_vectors(t, ot) == ntaxa - 2
    if kw:
        assert ot.partial(a, b)[1].max() >= 1            # scaling events over the whole (class, rate) block
    # pattern lnL, derivatives, lnL from the theta buffer, Newton on the device
    _, oplh = ot.branch_lnl(a, b)
    _, sc, _ = ot.partial(a, b)
    np.testing.assert_allclose(t.compute_pattern_likelihood(), oracle.pattern_lh_scaled(oplh, None, sc), rtol=1e-10)
    df, ddf = t.compute_likelihood_derv(a, b)
    odf, oddf = ot.derv(a, b)
    assert abs(df - odf) <= 1e-8 * max(1.0, abs(odf)) and abs(ddf - oddf) <= 1e-8 * abs(oddf)
    assert abs(t.compute_likelihood_from_buffer() - ref) <= LNL_RTOL * abs(ref)
    # an internal branch (both ends vectors) and a partial re-evaluation
    inner = [(x, y) for x in range(t.num_nodes) for y, _ in t.neighbors(x) if not ot.is_leaf(x) and not ot.is_leaf(y)]
    x, y = inner[len(inner) // 2]
    assert abs(t.compute_likelihood_branch(x, y) - ref) <= LNL_RTOL * abs(ref)
    t.set_branch_length(x, y, 0.31)
    ot.set_length(x, y, 0.31)
    ref2, _ = ot.likelihood()
    assert abs(t.compute_likelihood() - ref2) <= LNL_RTOL * abs(ref2)
    before = t.compute_likelihood()
    t.optimize_one_branch(a, b)
    after = t.compute_likelihood()
    assert after >= before - 1e-7
    ot.set_length(a, b, t.neighbor_info(a, b)["length"])
    ref3, _ = ot.likelihood()
    assert abs(after - ref3) <= 1e-8 * abs(ref3)


@pytest.mark.gpu
def test_mixture_model_switching_and_limits(pkg, synth, oracle):
    """plain -> mixture -> plain on one engine (the pipelined kernel holds one eigen-system; a mixture takes
    the generic matrix-core kernel and another plan form), and the documented limits."""
    from test_parity_gpu import LNL_RTOL
    mix = synth.mixture_model(20, 4, 31, fused=True)                     # 4 components
    plain = synth.random_reversible_model(20, 32, alpha=0.9, ncat=4)     # 4 categories
    nwk = synth.random_tree_newick(10, 33)
    st = synth.simulate_alignment(nwk, plain, 200, 34)
    pat, freq = synth.compress_patterns(st)
    t = pkg.PhyloTree(nwk)
    t.set_alignment(20, 1, pat, freq)
    t.set_model(plain)
    t.attach_engine(0)
    for model in (plain, mix, plain, mix):
        t.set_model(model)
        t.clear_all_partial_lh()
        ot = oracle.OracleTree(nwk, 20, 1, pat, freq, None, model)
        ref, _ = ot.likelihood()
        assert abs(t.compute_likelihood() - ref) <= LNL_RTOL * abs(ref)


@pytest.mark.gpu
@pytest.mark.parametrize("n,seq_type,nclass,ncat,fused,ntaxa,nsites,kw", [
    (4, 0, 2, 4, False, 14, 900, dict()),                                       # DNA mixture x Gamma: 8 components
    (4, 0, 3, 1, True, 10, 400, dict()),                                        # fused mixture-rate model
    (4, 0, 2, 2, False, 320, 200, dict(lo=0.4, hi=0.9, caterpillar=True)),      # with scaling events
    (64, 2, 2, 1, True, 9, 300, dict()),                                        # codon mixture (M-series style classes)
    (64, 2, 3, 2, False, 7, 150, dict()),
])
def test_hip_mixtures_of_4_and_64_states(pkg, synth, oracle, n, seq_type, nclass, ncat, fused, ntaxa, nsites, kw):
    """phylokernelmixture.h / phylokernelmixrate.h as bound for 4 and 64 states (phylotreeavx.cpp:62-73, 107-121): the
    generic matrix-core kernel with per-class A images; a 4-state engine switches to the 16-pattern tile layout while
    its model is a mixture and back when it is not."""
    from test_parity_gpu import check_all_vectors, LNL_RTOL
    model, nwk, pat, freq, ot = make_mix(synth, oracle, n, nclass, ncat, fused, ntaxa, nsites, 900 + n + nclass + ntaxa,
                                         seq_type, **kw)
    t = pkg.PhyloTree(nwk)
    t.set_alignment(n, seq_type, pat, freq)
    t.set_model(model)
    t.attach_engine(0)
    lnl = t.compute_likelihood()
    ref, (a, b) = ot.likelihood()
    assert abs(lnl - ref) <= LNL_RTOL * abs(ref)
    assert check_all_vectors(t, ot) == ntaxa - 2
    if kw:
        frm, to = (a, b) if not ot.is_leaf(b) else (b, a)
        assert ot.partial(frm, to)[1].max() >= 1
    df, ddf = t.compute_likelihood_derv(a, b)
    rdf, rddf = ot.derv(a, b)
    assert abs(ddf - rddf) <= 1e-8 * abs(rddf) and abs(df - rdf) <= 1e-8 * max(abs(rdf), 1e-3 * abs(rddf))
    assert abs(t.compute_likelihood_from_buffer() - ref) <= LNL_RTOL * abs(ref)
    before = t.compute_likelihood()
    x, y = [(x, y) for x in range(t.num_nodes) for y, _ in t.neighbors(x) if x < y][3]
    t.optimize_one_branch(x, y)
    assert t.compute_likelihood() >= before - 1e-9 * abs(before)
    if n == 4:
        # back to a plain model on the same engine (64-pattern tiles again), then to the mixture once more
        plain = synth.gtr_model(ncat=model.ncat)
        for m2 in (plain, model):
            t.set_model(m2)
            t.clear_all_partial_lh()
            o2 = oracle.OracleTree(t.tree_string(), 4, 0, pat, freq, None, m2)   # (one branch was optimised above)
            r2, _ = o2.likelihood()
            assert abs(t.compute_likelihood() - r2) <= LNL_RTOL * abs(r2)


@pytest.mark.gpu
def test_hip_mixture_with_ascertainment_rell_and_staged_plans(pkg, synth, oracle, monkeypatch):
    """combinations: mixture + ASC (variable sites only), mixture + RELL, mixture under forced plan staging."""
    from test_parity_gpu import LNL_RTOL
    model = synth.mixture_model(20, 3, 91, ncat=2)
    nwk = synth.random_tree_newick(14, 92)
    st = synth.simulate_alignment(nwk, model.classes[0], 400, 93)
    st = st[:, [s for s in range(st.shape[1]) if len(set(st[:, s].tolist())) > 1]]
    pat, freq = synth.compress_patterns(st)
    nsite = int(freq.sum())
    pat_a = np.concatenate([pat, np.tile(np.arange(20, dtype=np.uint8), (14, 1))], axis=1)
    freq_a = np.concatenate([freq, np.zeros(20)])
    for split in ("0", "4"):
        monkeypatch.setenv("IQHIP_SPLIT", split)
        t = pkg.PhyloTree(nwk)
        t.set_alignment(20, 1, pat_a, freq_a)
        t.set_ascertainment(20, nsite)
        t.set_model(model)
        t.attach_engine(0)
        ot = oracle.OracleTree(nwk, 20, 1, pat_a, freq_a, None, model, n_unobs=20, nsites=nsite)
        ref, (a, b) = ot.likelihood()
        lnl = t.compute_likelihood()
        assert abs(lnl - ref) <= LNL_RTOL * abs(ref)
        df, ddf = t.compute_likelihood_derv(a, b)
        odf, oddf = ot.derv(a, b)
        assert abs(df - odf) <= 1e-8 * max(1.0, abs(odf)) and abs(ddf - oddf) <= 1e-8 * abs(oddf)
        w = np.vstack([freq_a, np.roll(freq_a[:-20], 3).tolist() + [0.0] * 20]).astype(np.float32)
        t.set_boot_samples(w)
        r = t.compute_rell()
        plh = t.compute_pattern_likelihood()
        assert abs(r[0] - lnl) <= 1e-9 * abs(lnl) and abs(r[1] - np.dot(w[1].astype(np.float64), plh)) <= 1e-9 * abs(r[1])
