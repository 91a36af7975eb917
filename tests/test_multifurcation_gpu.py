"""GPU: trees with polytomies (the reference hands nodes of degree > 3 to its scalar kernel,
phylokernel.h:73-77 -> phylotreesse.cpp:702-806).  The adapter submits such a node as a chain of binary updates over
zero-length branches whose intermediate products are never rescaled (include/iqhip_adapter.h); the oracle restates
the scalar kernel (one product over all children, one scaling test)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LNL_RTOL = 1e-9


def build(pkg, synth, oracle, n, ncat, seq_type, ntaxa, nsites, seed, mem_mode=0, **kw):
    model = synth.gtr_model(ncat=ncat) if n == 4 else synth.random_reversible_model(n, seed, alpha=0.9, ncat=ncat)
    su = oracle.state_unknown_for(n, seq_type)
    nwk = synth.random_multifurcating_newick(ntaxa, seed, **kw)
    st = synth.simulate_alignment(nwk, model, nsites, seed + 1, 0.03, su)
    pat, freq = synth.compress_patterns(st)
    ot = oracle.OracleTree(nwk, n, seq_type, pat, freq, None, model)
    assert max(len(v) for v in ot.adj.values()) > 3
    t = pkg.PhyloTree(nwk)
    t.set_mem_mode(mem_mode)
    t.set_alignment(n, seq_type, pat, freq)
    t.set_model(model)
    t.attach_engine(0)
    return t, ot


@pytest.mark.parametrize("n,ncat,seq_type,ntaxa,nsites", [(4, 4, 0, 30, 1500), (4, 4, 0, 12, 70000), (20, 4, 1, 16, 600),
                                                          (64, 1, 2, 10, 400), (20, 5, 1, 10, 300)])
@pytest.mark.parametrize("mem_mode", [0, 1])
def test_polytomies_against_oracle(pkg, synth, oracle, n, ncat, seq_type, ntaxa, nsites, mem_mode):
    t, ot = build(pkg, synth, oracle, n, ncat, seq_type, ntaxa, nsites, 9000 + n + ntaxa, mem_mode=mem_mode,
                  max_children=5, p_multi=0.6)
    ref, (a, b) = ot.likelihood()
    lnl = t.compute_likelihood()
    assert t.current_branch() == (a, b)
    assert abs(lnl - ref) <= LNL_RTOL * abs(ref)
    nchecked = 0
    for x in range(t.num_nodes):
        for y, _ in t.neighbors(x):
            info = t.neighbor_info(x, y)
            if ot.is_leaf(y) or not (info["computed"] & 1) or info["key"] == 0:
                continue
            plh, sc, sf = ot.partial(x, y)
            got = t.fetch_partial(x, y)
            scale = np.abs(plh).max(axis=1, keepdims=True)
            # the intermediate products pass through U^-1 and U once more than the scalar kernel's single product
            np.testing.assert_allclose(got / scale, plh / scale, rtol=0, atol=1e-9)
            assert np.array_equal(t.fetch_scale_num(x, y), sc)
            assert abs(info["lh_scale_factor"] - sf) <= 1e-12 * max(1.0, abs(sf))
            nchecked += 1
    assert nchecked == sum(1 for v in ot.adj if len(ot.adj[v]) > 1)
    # every branch gives the same lnL; derivatives and the Newton solve work on branches at polytomies
    for x in range(t.num_nodes):
        for y, _ in t.neighbors(x):
            if x < y:
                assert abs(t.compute_likelihood_branch(x, y) - ref) <= LNL_RTOL * abs(ref)
    hub = max(ot.adj, key=lambda v: len(ot.adj[v]))
    y = ot.adj[hub][0][0]
    t.reset_theta()
    df, ddf = t.compute_likelihood_derv(hub, y)
    rdf, rddf = ot.derv(hub, y)
    assert abs(ddf - rddf) <= 1e-8 * abs(rddf) and abs(df - rdf) <= 1e-8 * max(abs(rdf), 1e-3 * abs(rddf))
    before = t.compute_likelihood()
    t.optimize_one_branch(hub, y)
    assert t.compute_likelihood() >= before - 1e-9 * abs(before)


def test_polytomy_with_scaling_events(pkg, synth, oracle):
    """long branches, many taxa: the node's own update rescales, the intermediate products never do; counters are
    the reference's (sum over the internal children + one per event at the node)."""
    t, ot = build(pkg, synth, oracle, 4, 4, 0, 260, 300, 9100, lo=0.4, hi=0.9, max_children=4, p_multi=0.5)
    ref, (a, b) = ot.likelihood()
    lnl = t.compute_likelihood()
    frm, to = (a, b) if not ot.is_leaf(b) else (b, a)
    plh, sc, sf = ot.partial(frm, to)
    assert sc.max() >= 1 and sf < 0
    assert abs(lnl - ref) <= LNL_RTOL * abs(ref)
    assert np.array_equal(t.fetch_scale_num(frm, to), sc)
    assert abs(t.neighbor_info(frm, to)["lh_scale_factor"] - sf) <= 1e-12 * abs(sf)
