"""CPU: the engine's Newton state machine (iqhip_newton_host_*; the same function the sharded engines run in a
1-thread kernel between all-reduces) against the loop form of Optimization::minimizeNewton
(optimization.cpp:388-465, restated below as test code) on derivative values the oracle computes."""
import math

import numpy as np
import pytest


def minimize_newton_loop(fd, x1, xguess, x2, xacc, max_steps):
    """optimization.cpp:388-450 with fd(x) -> (f, df) = (-dlnL/dt, -d2lnL/dt2); returns (result, d2l, evaluated xs)."""
    xs = []
    rts = min(max(xguess, x1), x2)
    xs.append(rts)
    f, df = fd(rts)
    d2l = df
    if not (math.isfinite(f) and math.isfinite(df)):
        raise RuntimeError("Wrong computeFuncDerv")
    if df >= 0.0 and abs(f) < xacc:
        return rts, d2l, xs
    if f < 0.0:
        xl, xh = rts, x2
    else:
        xh, xl = rts, x1
    dx = abs(xh - xl)
    for j in range(1, max_steps + 1):
        rts_old = rts
        if df <= 0.0 or ((rts - xh) * df - f) * ((rts - xl) * df - f) >= 0.0:
            dx = 0.5 * (xh - xl)
            rts = xl + dx
            d2l = df
            if xl == rts:
                return rts, d2l, xs
        else:
            dx = f / df
            temp = rts
            rts -= dx
            d2l = df
            if temp == rts:
                return rts, d2l, xs
        if abs(dx) < xacc or j == max_steps:
            return rts_old, d2l, xs
        xs.append(rts)
        f, df = fd(rts)
        if not (math.isfinite(f) and math.isfinite(df)):
            raise RuntimeError("Wrong computeFuncDerv")
        if df > 0.0 and abs(f) < xacc:
            d2l = df
            return rts, d2l, xs
        if f < 0.0:
            xl = rts
        else:
            xh = rts
    raise RuntimeError("Maximum number of iterations exceeded")


def run_machine(pkg, derv, xguess, x1, x2, xacc, max_steps):
    m = pkg.NewtonStateMachine(xguess, x1, x2, xacc, max_steps)
    xs = []
    while not m.done:
        xs.append(m.x)
        df, ddf = derv(m.x)
        m.update(df, ddf)
    optx, d2l, n, st = m.result()
    assert n == len(xs)
    return optx, d2l, xs, st


@pytest.mark.parametrize("xguess,max_steps", [(0.1, 100), (5.0, 100), (1e-6, 100), (90.0, 100), (0.1, 10), (0.1, 1),
                                              (0.1, 2), (30.0, 3)])
def test_state_machine_reproduces_the_loop(pkg, synth, oracle, xguess, max_steps):
    model = synth.gtr_model()
    nwk = synth.random_tree_newick(9, 21)
    st = synth.simulate_alignment(nwk, model, 300, 22)
    pat, freq = synth.compress_patterns(st)
    ot = oracle.OracleTree(nwk, 4, 0, pat, freq, None, model)
    inner = [(a, b) for a in ot.adj for b, _ in ot.adj[a] if a < b]
    for a, b in inner[:6]:
        th, _ = ot.theta(a, b)

        def derv(x):
            return ot.derv(a, b, length=x, theta=th)

        def fd(x):
            df, ddf = derv(x)
            return -df, -ddf

        ref, d2l_ref, xs_ref = minimize_newton_loop(fd, 1e-6, xguess, 100.0, 1e-6, max_steps)
        got, d2l, xs, status = run_machine(pkg, derv, xguess, 1e-6, 100.0, 1e-6, max_steps)
        assert status == 0
        assert xs == xs_ref            # the same points are evaluated, bit for bit
        assert got == ref and d2l == d2l_ref


def test_state_machine_special_values(pkg):
    # non-finite first derivative sum is zeroed (phylokernel.h:647-651): f = df = 0 -> "df >= 0 && |f| < xacc": done
    m = pkg.NewtonStateMachine(0.2, 1e-6, 100.0, 1e-6, 10)
    m.update(float("nan"), 1.0)
    assert m.done and m.result()[0] == 0.2 and m.result()[3] == 0
    # an infinite second-derivative sum is a "Wrong computeFuncDerv" (status 2), the iterate before it is returned
    m = pkg.NewtonStateMachine(0.2, 1e-6, 100.0, 1e-6, 10)
    x1 = m.update(-3.0, -50.0)
    assert not m.done and x1 != 0.2
    m.update(1.0, float("inf"))
    assert m.done and m.result()[3] == 2 and m.result()[0] == 0.2
    # clamping of the start value
    assert pkg.NewtonStateMachine(1e-9, 1e-6, 100.0, 1e-6, 10).x == 1e-6
    assert pkg.NewtonStateMachine(1e9, 1e-6, 100.0, 1e-6, 10).x == 100.0
    with pytest.raises(pkg.HostError):
        pkg.NewtonStateMachine(0.1, 1.0, 0.5, 1e-6, 10)
    with pytest.raises(pkg.HostError):
        pkg.NewtonStateMachine(0.1, 1e-6, 100.0, 1e-6, 10).result()   # not finished
