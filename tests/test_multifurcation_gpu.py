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


@pytest.mark.parametrize("n,ncat,seq_type,nptn", [(4, 4, 0, 300), (4, 4, 0, 9000), (20, 4, 1, 200), (20, 4, 1, 6000), (20, 1, 1, 100),
                                                  (20, 6, 1, 150), (64, 1, 2, 120), (64, 1, 2, 3000), (64, 2, 2, 100)])
def test_scalar_kernel_zero_rule_at_a_multifurcating_node(pkg, synth, oracle, n, ncat, seq_type, nptn):
    """phylotreesse.cpp:774-788, the scalar kernel's `lh_max == 0.0` branch ("very shitty data"): the pattern's vector becomes
    the unknown tip's in every category, scale_num += 4, sum_scale += 4 log(2^-256) f -- whatever ptn_invar says.  Forced
    through the C ABI: three uploaded child vectors, some patterns of one child exactly zero (and one pattern only
    tiny: a denormal is not zero), the node submitted as the adapter submits a polytomy -- a NO_SCALE intermediate over a
    zero-length branch, then the node's own update with IQHIP_OP_SCALAR_RULE -- against the oracle's restatement of the
    scalar kernel.  Every traversal kernel family: 4 states (VALU), 20 / 64 states (pipelined, generic), small and
    larger pattern counts (cat-split / row-split variants)."""
    import ctypes as C
    lib = pkg.libiqhip()
    rng = np.random.default_rng(n * 100 + ncat)
    model = synth.gtr_model(ncat=ncat) if n == 4 else synth.random_reversible_model(n, 5, alpha=0.9 if ncat > 1 else None, ncat=ncat)
    su = oracle.state_unknown_for(n, seq_type)
    B = n * ncat
    kids = []
    for k in range(3):   # eigen-space vectors of plausible magnitude: U^-1 applied to positive probability-space vectors
        prob = rng.uniform(1e-3, 1.0, size=(nptn, ncat, n))
        v = np.einsum("ix,pcx->pci", model.inv_evec.reshape(n, n), prob).reshape(nptn, B)
        kids.append(np.ascontiguousarray(v))
    zero_ptn = [1, 7, nptn - 1, nptn // 2]
    kids[1][zero_ptn] = 0.0
    for k in range(3):                                  # pattern 3: every child ~1e-104 -> the product is a denormal (~1e-312):
        kids[k][3] *= 1e-104                            # lh_max is tiny but not 0 -> the ordinary rule (count + 1)
    sc = [rng.integers(0, 3, nptn).astype(np.int16) for _ in range(3)]
    freq = rng.integers(1, 5, nptn).astype(np.float64)
    invar = np.zeros(nptn)
    invar[zero_ptn[0]] = 0.01                           # the zero branch comes before the ptn_invar test
    lens = np.array([0.11, 0.07, 0.23])
    # oracle: the scalar kernel on the three children
    ot = oracle.OracleTree("(0:0.1,1:0.1,2:0.1);", n, seq_type, np.full((3, nptn), su, dtype=np.uint8), freq, invar, model)
    L, dp, sp, u8 = ot.L, C.POINTER(C.c_double), C.POINTER(C.c_short), C.POINTER(C.c_uint8)
    out, osc = np.empty((nptn, B)), np.empty(nptn, dtype=np.int16)
    ref_ss = L.oracle_partial_update_multi(
        n, ncat, nptn, 3, oracle._dp(ot.eval), oracle._dp(ot.evec), oracle._dp(ot.inv_evec), oracle._dp(ot.rates), oracle._dp(ot.tip),
        su, (u8 * 3)(), (dp * 3)(*[oracle._dp(k) for k in kids]), (sp * 3)(*[oracle._sp(s) for s in sc]), oracle._dp(lens),
        oracle._dp(freq), oracle._dp(invar), oracle._dp(out), oracle._sp(osc))
    assert set(np.nonzero(osc - (sc[0] + sc[1] + sc[2]) == 4)[0]) == set(zero_ptn)
    assert osc[3] - (sc[0][3] + sc[1][3] + sc[2][3]) == 1 and 0.0 < np.abs(out[3]).max() * 2.0 ** -256 < 2.3e-308
    # engine
    e = C.c_void_p()
    assert lib.iqhip_create(C.byref(e), 0, n, ncat, nptn, 3) == 0
    d = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))   # noqa: E731
    assert lib.iqhip_set_model(e, d(ot.eval), d(ot.evec), d(ot.inv_evec), d(ot.rates), d(ot.props), su, d(ot.tip)) == 0
    st = np.full((3, nptn), su, dtype=np.uint8)
    assert lib.iqhip_set_alignment(e, st.ctypes.data_as(C.POINTER(C.c_uint8)), d(freq), d(invar)) == 0
    for k in range(3):
        assert lib.iqhip_upload_partial(e, 11 + k, d(kids[k]), sc[k].ctypes.data_as(C.POINTER(C.c_int16))) == 0
    ops = (pkg.NodeOp * 2)()
    ops[0] = pkg.NodeOp(500, 11, 12, -1, -1, lens[0], lens[1], 1, 0)     # IQHIP_OP_NO_SCALE
    ops[1] = pkg.NodeOp(501, 500, 13, -1, -1, 0.0, lens[2], 2, 0)       # IQHIP_OP_SCALAR_RULE
    ss = np.zeros(2)
    assert lib.iqhip_update_partials(e, ops, 2, d(ss)) == 0, lib.iqhip_last_error()
    got, gsc = np.zeros(nptn * B), np.zeros(nptn, dtype=np.int16)
    assert lib.iqhip_fetch_partial(e, 501, d(got)) == 0
    assert lib.iqhip_fetch_scale_num(e, 501, gsc.ctypes.data_as(C.POINTER(C.c_int16))) == 0
    got = got.reshape(nptn, B)
    assert np.array_equal(gsc, osc)
    assert ss[0] == 0.0 and abs(ss[1] - ref_ss) <= 1e-12 * abs(ref_ss)
    unk = np.tile(ot.tip.reshape(-1, n)[su], ncat)
    for p in zero_ptn:
        assert np.array_equal(got[p], unk)              # the unknown tip's vector, bit for bit
    rest = np.setdiff1d(np.arange(nptn), zero_ptn)
    scale = np.abs(out[rest]).max(axis=1, keepdims=True)
    np.testing.assert_allclose(got[rest] / scale, out[rest] / scale, rtol=0, atol=1e-9)
    lib.iqhip_destroy(e)
