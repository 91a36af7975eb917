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
