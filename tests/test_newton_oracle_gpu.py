"""SURVEY 8(f)-1 parity: the device Newton-Raphson solve (k_newton: the whole Optimization::minimizeNewton loop,
optimization.cpp:388-465, in one launch) against its ORACLE twin -- oracle_driver.OracleTree.minimize_newton, the same
loop restated in Python over the oracle's derivative kernel (oracle/lh_oracle.c oracle_derv, phylokernel.h:485-730).
Same optimum (<= 1e-9), same number of derivative evaluations, for DNA / protein / codon, from ordinary, bound-hitting
and far-off starting points."""
import ctypes as C

import pytest

from test_parity_gpu import make_case

pytestmark = pytest.mark.gpu


def device_newton(pkg, t, xguess, x1, x2, xacc, max_steps):
    lib = pkg.libiqhip()
    optx, d2l, ns = C.c_double(), C.c_double(), C.c_int()
    rc = lib.iqhip_newton_branch(t.engine, C.c_double(xguess), C.c_double(x1), C.c_double(x2), C.c_double(xacc),
                                 max_steps, C.byref(optx), C.byref(d2l), C.byref(ns))
    assert rc == 0, lib.iqhip_last_error()
    return optx.value, d2l.value, ns.value


CASES = [  # n, ncat, seq_type, ntaxa, nsites
    (4, 4, 0, 10, 300),      # one workgroup: no exchange between workgroups
    (4, 4, 0, 12, 40000),    # many workgroups: posted exchange
    (20, 4, 1, 9, 1200),
    (64, 1, 2, 8, 600),
]


@pytest.mark.parametrize("n,ncat,seq_type,ntaxa,nsites", CASES)
def test_device_newton_matches_oracle_newton(pkg, synth, oracle, n, ncat, seq_type, ntaxa, nsites):
    t, ot, *_ = make_case(synth, oracle, pkg, ntaxa, nsites, n, ncat, 900 + n + nsites % 97, seq_type=seq_type, missing=0.02)
    t.compute_likelihood()
    edges = [(a, b) for a in range(t.num_nodes) for b, _ in t.neighbors(a) if a < b]
    # (x1, xguess, x2, xacc, max_steps): the reference's call (min, current, max, min; phylotree.cpp:2160), a far-off
    # start, an upper bound below the optimum, a lower bound above it, a step limit that ends the loop early
    setups = [(1e-6, None, 100.0, 1e-6, 100), (1e-6, 60.0, 100.0, 1e-6, 100), (1e-6, 0.01, 0.03, 1e-6, 100),
              (0.6, 0.9, 100.0, 1e-6, 100), (1e-6, 3.0, 100.0, 1e-6, 3)]
    nchecked = 0
    for (a, b) in edges[:5]:
        t.reset_theta()
        t.compute_likelihood_derv(a, b)  # pending partials of both ends + theta of this branch on the device
        theta, _ = ot.theta(a, b)
        for (x1, xg, x2, xacc, ms) in setups:
            xg = ot.length(a, b) if xg is None else xg
            ref_x, ref_d2l, pts, status = ot.minimize_newton(a, b, x1, xg, x2, xacc, ms, theta=theta)
            assert status == "ok"
            optx, d2l, ns = device_newton(pkg, t, xg, x1, x2, xacc, ms)
            assert ns == len(pts), (a, b, x1, xg, x2, ns, pts)
            assert abs(optx - ref_x) <= 1e-9 * max(1.0, abs(ref_x)), (a, b, x1, xg, x2, optx, ref_x)
            # minimizeNewton's d2l output (second derivative of -lnL at the last step)
            assert abs(d2l - ref_d2l) <= 1e-6 * max(1.0, abs(ref_d2l)), (d2l, ref_d2l)
            nchecked += 1
    assert nchecked == 25


def test_optimize_one_branch_is_the_reference_sequence(pkg, synth, oracle):
    """optimizeOneBranch (phylotree.cpp:2148-2192) through the one-call device solve, checked against the oracle's
    loop incl. the diverged-Newton reset: a branch whose optimum lies beyond 0.95 * max_branch_length."""
    t, ot, *_ = make_case(synth, oracle, pkg, 8, 500, 4, 4, 4242)
    t.compute_likelihood()
    t.set_branch_bounds(1e-6, 0.05)   # every optimum above 0.0475 counts as diverged
    t.set_device_newton(True)
    edges = [(a, b) for a in range(t.num_nodes) for b, _ in t.neighbors(a) if a < b]
    ndiv = 0
    for (a, b) in edges:
        cur = ot.length(a, b)
        got = t.optimize_one_branch(a, b)
        optx, _, _, status = ot.minimize_newton(a, b, 1e-6, cur, 0.05, 1e-6, 100)
        assert status == "ok"
        want = optx
        if optx > 0.05 * 0.95:   # reset unless the solve's point is better (phylotree.cpp:2167-2176)
            opt_lh = ot.lnl_from_theta(a, b, length=optx)[0]
            orig_lh = ot.lnl_from_theta(a, b, length=cur)[0]
            if orig_lh > opt_lh:
                want = cur
            ndiv += 1
        assert abs(got - want) <= 1e-9 * max(1.0, want), (a, b, got, want, optx)
        ot.set_length(a, b, got)
    assert ndiv >= 3


@pytest.mark.parametrize("n,ncat,seq_type,nsites", [(4, 4, 0, 500), (4, 4, 0, 25000), (20, 4, 1, 700), (64, 1, 2, 400)])
def test_device_newton_with_ascertainment_correction(pkg, synth, oracle, n, ncat, seq_type, nsites):
    """+ASC inside the one-launch solve (phylokernel.h:655-725: the derivative correction from the unobserved constant
    patterns, computed by every workgroup for itself): same optimum and evaluation count as the oracle's minimizeNewton
    over its +ASC derivative, on small (one workgroup) and larger (posted exchange) alignments; and the whole
    optimizeAllBranches sweep -- device solve per branch, steps inside iqhip_optimize_sweep -- equals the host loop."""
    import numpy as np
    model = synth.gtr_model(alpha=0.9, ncat=ncat) if n == 4 else synth.random_reversible_model(n, 51, alpha=0.9 if ncat > 1 else None, ncat=ncat)
    nwk = synth.random_tree_newick(9, 52, 0.02, 0.15)
    st = synth.simulate_alignment(nwk, model, nsites, 53)
    pat, freq = synth.compress_patterns(st)
    const = np.all(pat == pat[0][None, :], axis=0)
    pat, freq = np.ascontiguousarray(pat[:, ~const]), freq[~const].copy()
    nun, ns = n, float(freq.sum())
    pat = np.ascontiguousarray(np.concatenate([pat, np.tile(np.arange(n, dtype=np.uint8)[None, :], (9, 1))], axis=1))
    freq = np.concatenate([freq, np.zeros(n)])
    ot = oracle.OracleTree(nwk, n, seq_type, pat, freq, None, model, n_unobs=nun, nsites=ns)

    def make():
        t = pkg.PhyloTree(nwk)
        t.set_alignment(n, seq_type, pat, freq)
        t.set_ascertainment(nun, ns)
        t.set_model(model)
        t.attach_engine(0)
        return t
    t = make()
    t.compute_likelihood()
    edges = [(a, b) for a in range(t.num_nodes) for b, _ in t.neighbors(a) if a < b]
    for (a, b) in edges[:4]:
        t.reset_theta()
        t.compute_likelihood_derv(a, b)
        theta, _ = ot.theta(a, b)
        for xg in (ot.length(a, b), 2.5):
            ref_x, _, pts, status = ot.minimize_newton(a, b, 1e-6, xg, 100.0, 1e-6, 100, theta=theta)
            optx, _, nsteps = device_newton(pkg, t, xg, 1e-6, 100.0, 1e-6, 100)
            assert status == "ok" and nsteps == len(pts), (a, b, nsteps, pts)
            assert abs(optx - ref_x) <= 1e-9 * max(1.0, abs(ref_x)), (a, b, optx, ref_x)
    vals = []
    for dev in (False, True):
        t2 = make()
        t2.set_device_newton(dev)
        vals.append((t2.optimize_all_branches(iterations=2, tolerance=1e-6), t2.tree_string()))
    assert abs(vals[0][0] - vals[1][0]) <= 1e-9 * abs(vals[0][0])
    ot2 = oracle.OracleTree(vals[1][1], n, seq_type, pat, freq, None, model, n_unobs=nun, nsites=ns)
    ref, _ = ot2.likelihood()
    assert abs(vals[1][0] - ref) <= 1e-8 * abs(ref)
