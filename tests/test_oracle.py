"""CPU tests of the oracle (oracle/lh_oracle.c): against an independent probability-space
implementation (tests/textbook.py) and against its own invariants.  The reference itself holds
no golden values for this path, and its likelihood translation units cannot be
built here without its cmake-generated header, so these tests plus tests/test_vcl_probe.py are
what pins the oracle ("parity unpinned" against reference execution)."""
import numpy as np
import pytest


def _setup(synth, oracle, ntaxa, nsites, n, ncat, seed, seq_type, missing=0.0, pinvar=0.0, lo=0.02, hi=0.2):
    if n == 4:
        model = synth.gtr_model(alpha=0.9, ncat=ncat, pinvar=pinvar)
    else:
        model = synth.random_reversible_model(n, seed, alpha=0.9, ncat=ncat, pinvar=pinvar)
    nwk = synth.random_tree_newick(ntaxa, seed, lo, hi)
    su = oracle.state_unknown_for(n, seq_type)
    st = synth.simulate_alignment(nwk, model, nsites, seed + 1, missing, su)
    pat, freq = synth.compress_patterns(st)
    invar = synth.ptn_invar_for(pat, model)
    tree = oracle.OracleTree(nwk, n, seq_type, pat, freq, invar, model)
    return model, nwk, pat, freq, invar, tree, su


@pytest.mark.parametrize("n,ncat,seq_type", [(4, 4, 0), (4, 1, 0), (20, 4, 1), (64, 1, 2), (2, 4, 3), (2, 1, 3)])
def test_oracle_matches_textbook(synth, oracle, n, ncat, seq_type):
    import textbook
    model, nwk, pat, freq, invar, tree, su = _setup(synth, oracle, 9, 60, n, ncat, 11 + n, seq_type, missing=0.1)
    lnl, (a, b) = tree.likelihood()
    site = textbook.site_log_likelihoods(tree.adj, pat, model, seq_type, su)
    ref = float((site * freq).sum())
    assert abs(lnl - ref) <= 1e-9 * abs(ref)
    _, plh = tree.branch_lnl(a, b)
    np.testing.assert_allclose(plh, site, rtol=1e-9, atol=1e-9)


def test_oracle_dna_ambiguity_codes(synth, oracle):
    import textbook
    model, nwk, pat, freq, invar, tree, su = _setup(synth, oracle, 8, 40, 4, 4, 5, 0)
    rng = np.random.default_rng(0)
    pat = pat.copy()
    mask = rng.random(pat.shape) < 0.3
    pat[mask] = rng.integers(4, 19, mask.sum())  # IUPAC codes 4..17 and unknown 18
    tree = oracle.OracleTree(nwk, 4, 0, pat, freq, None, model)
    lnl, _ = tree.likelihood()
    site = textbook.site_log_likelihoods(tree.adj, pat, model, 0, su)
    assert abs(lnl - float((site * freq).sum())) <= 1e-9 * abs(lnl)


def test_oracle_invariant_sites(synth, oracle):
    import textbook
    model, nwk, pat, freq, invar, tree, su = _setup(synth, oracle, 8, 200, 4, 4, 3, 0, pinvar=0.2, hi=0.05)
    assert (invar > 0).any()
    lnl, _ = tree.likelihood()
    site = textbook.site_log_likelihoods(tree.adj, pat, model, 0, su)
    # textbook gives the Gamma part with props summing to 1-pinvar; add the +I term
    site = np.log(np.exp(site) + invar)
    assert abs(lnl - float((site * freq).sum())) <= 1e-9 * abs(lnl)


def test_oracle_lnl_is_branch_independent(synth, oracle):
    """Pulley principle: the lnL is the same on every branch (leaf and internal forms)."""
    model, nwk, pat, freq, invar, tree, su = _setup(synth, oracle, 10, 80, 4, 4, 7, 0, missing=0.05)
    vals = []
    for a in tree.adj:
        for b, _ in tree.adj[a]:
            if a < b:
                vals.append(tree.branch_lnl(a, b)[0])
    assert max(vals) - min(vals) <= 1e-9 * abs(vals[0])


def test_oracle_scaling_counters(synth, oracle):
    """Deep tree with long branches: scaling events happen, counters are cumulative, and the lnL
    still agrees with the log-space textbook value."""
    import textbook
    model = synth.random_reversible_model(20, 3, alpha=0.9, ncat=4)
    nwk = synth.random_tree_newick(160, 3, 0.3, 0.6, caterpillar=True)
    st = synth.simulate_alignment(nwk, model, 24, 4)
    pat, freq = synth.compress_patterns(st)
    tree = oracle.OracleTree(nwk, 20, 1, pat, freq, None, model)
    lnl, (a, b) = tree.likelihood()
    _, sc, sf = tree.partial(a, b)
    assert sc.max() >= 1 and sf < 0
    np.testing.assert_allclose(sf, oracle.lib().oracle_log_scaling_threshold() * float((sc * freq).sum()),
                               rtol=1e-12)
    site = textbook.site_log_likelihoods(tree.adj, pat, model, 1, 23)
    assert abs(lnl - float((site * freq).sum())) <= 1e-9 * abs(lnl)


@pytest.mark.parametrize("n,seq_type", [(4, 0), (2, 3)])
def test_oracle_derivatives_match_finite_differences(synth, oracle, n, seq_type):
    model, nwk, pat, freq, invar, tree, su = _setup(synth, oracle, 9, 120, n, 4, 21, seq_type)
    for (a, b) in [(0, tree.adj[0][0][0]), (tree.ntaxa, tree.adj[tree.ntaxa][0][0])]:
        t = tree.length(a, b)
        th, sf = tree.theta(a, b)
        df, ddf = tree.derv(a, b, t, th)
        h = 1e-5
        f = lambda x: tree.lnl_from_theta(a, b, x, th, sf)[0]
        np.testing.assert_allclose(df, (f(t + h) - f(t - h)) / (2 * h), rtol=1e-6)
        np.testing.assert_allclose(ddf, (f(t + h) - 2 * f(t) + f(t - h)) / h ** 2, rtol=2e-4)
        # K9 == K6 at the current length
        assert abs(f(t) - tree.branch_lnl(a, b)[0]) <= 1e-10 * abs(f(t))


def test_oracle_on_reference_example_alignment(synth, oracle):
    """example/example.phy of the reference (44 taxa x 384 sites, many gaps): 355 distinct patterns as
    SURVEY.md section 4 reports, and oracle == textbook on a random tree."""
    import os
    import phylip
    import textbook
    names, st = phylip.read_phylip_dna(os.path.join(os.path.dirname(__file__), "golden", "example.phy"))
    assert st.shape == (44, 384)
    pat, freq = synth.compress_patterns(st)
    assert pat.shape[1] == 355 and freq.sum() == 384
    model = synth.gtr_model(rates6=(1.513, 2.393, 1.769, 1.912, 2.838, 1.0), freqs=(0.249, 0.262, 0.251, 0.238),
                            alpha=0.934, ncat=4)
    nwk = synth.random_tree_newick(44, 12)
    tree = oracle.OracleTree(nwk, 4, 0, pat, freq, None, model)
    lnl, _ = tree.likelihood()
    site = textbook.site_log_likelihoods(tree.adj, pat, model, 0, 18)
    assert abs(lnl - float((site * freq).sum())) <= 1e-9 * abs(lnl)


def _asc_case(synth, oracle, n, ncat, seq_type, seed, pinvar=0.0):
    """variable-sites-only alignment + the unobserved constant patterns appended (frequency 0)"""
    if n == 4:
        model = synth.gtr_model(alpha=0.9, ncat=ncat, pinvar=pinvar)
    else:
        model = synth.random_reversible_model(n, seed, alpha=0.9, ncat=ncat, pinvar=pinvar)
    nwk = synth.random_tree_newick(8, seed, 0.02, 0.15)
    st = synth.simulate_alignment(nwk, model, 400, seed + 1)
    pat, freq = synth.compress_patterns(st)
    const = np.all(pat == pat[0][None, :], axis=0)
    pat, freq = np.ascontiguousarray(pat[:, ~const]), freq[~const].copy()
    unobs = np.tile(np.arange(n, dtype=np.uint8)[None, :], (8, 1))   # one constant pattern per state
    pat2 = np.ascontiguousarray(np.concatenate([pat, unobs], axis=1))
    freq2 = np.concatenate([freq, np.zeros(n)])
    invar = synth.ptn_invar_for(pat2, model)
    return model, nwk, pat2, freq2, invar, n, float(freq.sum())


@pytest.mark.parametrize("n,ncat,seq_type,pinvar", [(4, 4, 0, 0.0), (4, 4, 0, 0.15), (20, 4, 1, 0.0)])
def test_oracle_ascertainment_bias_correction(synth, oracle, n, ncat, seq_type, pinvar):
    """+ASC: lnL = sum_i f_i*(log L_i - log(1 - P(constant pattern))) (Lewis 2001), which is what
    phylokernel.h:868-909,1009-1016 computes; checked against the probability-space textbook code."""
    import textbook
    model, nwk, pat, freq, invar, nun, nsites = _asc_case(synth, oracle, n, ncat, seq_type, 31 + n, pinvar)
    tree = oracle.OracleTree(nwk, n, seq_type, pat, freq, invar, model, n_unobs=nun, nsites=nsites)
    lnl, (a, b) = tree.likelihood()
    su = oracle.state_unknown_for(n, seq_type)
    site = np.exp(textbook.site_log_likelihoods(tree.adj, pat, model, seq_type, su)) + invar
    pconst = site[-nun:].sum()
    ref = float((freq[:-nun] * (np.log(site[:-nun]) - np.log(1.0 - pconst))).sum())
    assert abs(lnl - ref) <= 1e-9 * abs(ref)
    # every branch gives the same corrected lnL; K9 == K6; derivatives match finite differences
    for x in tree.adj:
        for y, _ in tree.adj[x]:
            if x < y:
                assert abs(tree.branch_lnl(x, y)[0] - lnl) <= 1e-9 * abs(lnl)
    for (x, y) in [(a, b), (tree.ntaxa, tree.adj[tree.ntaxa][0][0])]:
        t = tree.length(x, y)
        th, sf = tree.theta(x, y)
        f = lambda v: tree.lnl_from_theta(x, y, v, th, sf)[0]
        assert abs(f(t) - lnl) <= 1e-9 * abs(lnl)
        df, ddf = tree.derv(x, y, t, th)
        h = 1e-5
        np.testing.assert_allclose(df, (f(t + h) - f(t - h)) / (2 * h), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(ddf, (f(t + h) - 2 * f(t) + f(t - h)) / h ** 2, rtol=2e-3)


@pytest.mark.parametrize("n,ncat,seq_type", [(4, 4, 0), (20, 2, 1)])
def test_oracle_multifurcating_nodes_match_textbook(synth, oracle, n, ncat, seq_type):
    """degree > 3: the restatement of the reference's scalar kernel (phylotreesse.cpp:702-806) against the
    probability-space recursion, which handles any degree."""
    import textbook
    model = synth.gtr_model(ncat=ncat) if n == 4 else synth.random_reversible_model(n, 4, alpha=0.7, ncat=ncat)
    su = oracle.state_unknown_for(n, seq_type)
    nwk = synth.random_multifurcating_newick(15, 8, max_children=5, p_multi=0.6)
    assert max(nwk.count(","), 0) > 0
    st = synth.simulate_alignment(nwk, model, 200, 9, 0.05, su)
    pat, freq = synth.compress_patterns(st)
    tree = oracle.OracleTree(nwk, n, seq_type, pat, freq, None, model)
    assert max(len(v) for v in tree.adj.values()) > 3
    site = textbook.site_log_likelihoods(tree.adj, pat, model, seq_type, su)
    ref = float((site * freq).sum())
    lnl, (a, b) = tree.likelihood()
    assert abs(lnl - ref) <= 1e-9 * abs(ref)
    # pulley principle over all branches, including those at the polytomies
    for x in tree.adj:
        for y, _ in tree.adj[x]:
            if x < y:
                assert abs(tree.branch_lnl(x, y)[0] - ref) <= 1e-9 * abs(ref)


@pytest.mark.parametrize("n,ncat", [(3, 4), (5, 1), (7, 3), (13, 4), (21, 1), (33, 2), (61, 1)])
def test_scalar_kernel_state_counts_against_textbook(synth, oracle, n, ncat):
    """state counts outside 2 / 4 / 20 / 64 take the reference's scalar kernels (phylotreesse.cpp:281-309): the oracle
    restates those with plain running sums (VCW = 1) and the scalar node update; checked here against the
    probability-space recursion, derivatives against finite differences"""
    import textbook
    model = synth.random_reversible_model(n, 11, alpha=0.9 if ncat > 1 else None, ncat=ncat)
    nwk = synth.random_tree_newick(8, 5)
    st = synth.simulate_alignment(nwk, model, 300, 6, 0.03, n)
    pat, freq = synth.compress_patterns(st)
    ot = oracle.OracleTree(nwk, n, 3, pat, freq, None, model)
    lnl, (a, b) = ot.likelihood()
    sl = textbook.site_log_likelihoods(ot.adj, pat, model, 3, n)
    assert abs(lnl - (sl * freq).sum()) <= 1e-12 * abs(lnl)
    df, _ = ot.derv(a, b)
    h, L = 1e-5, ot.length(a, b)
    ot.set_length(a, b, L + h)
    lp, _ = ot.likelihood()
    ot.set_length(a, b, L - h)
    lm, _ = ot.likelihood()
    assert abs(df - (lp - lm) / (2 * h)) <= 1e-5 * max(1.0, abs(df))
    ot.set_length(a, b, L)
    assert ot.lnl_from_theta(a, b)[0] == pytest.approx(lnl, rel=1e-13)
