"""SURVEY 8(f)-1, hot loop 2: a whole optimizeAllBranches sweep (phylotree.cpp:2252-2332: optimizeOneBranch for every branch
in pre-order) as ONE engine submission -- iqhip_optimize_sweep through iqhip_adapter::optimizeBranchSweep -- against the
one-submission-per-branch form (iqhip_optimize_branch per optimizeOneBranch) and against the oracle.

Both forms evaluate the derivatives at the same branch lengths with the same kernels, so the optimised lengths must agree
to the last bit, with the same number of derivative evaluations; the lnL of the optimised tree is re-evaluated by the oracle."""
import numpy as np
import pytest

from test_parity_gpu import make_case
from test_sharded_gpu import case, sharded_tree

pytestmark = pytest.mark.gpu


def lengths(t):
    return {(a, b): ln for a in range(t.num_nodes) for b, ln in t.neighbors(a) if a < b}


def run_both(make, iterations=2, start=None, bounds=None, tolerance=1e-9):
    out = {}
    for sweep in (False, True):
        t = make()
        t.set_device_newton(True)
        t.set_device_sweep(sweep)
        if bounds:
            t.set_branch_bounds(*bounds)
        if start is not None:
            for (a, b) in lengths(t):
                t.set_branch_length(a, b, start, clear_reverse=False)
        t.clear_all_partial_lh()
        c0, s0 = t.num_derv_calls, t.num_submissions
        lnl = t.optimize_all_branches(iterations=iterations, tolerance=tolerance)
        out[sweep] = (lnl, lengths(t), t.num_derv_calls - c0, t.num_submissions - s0, t)
    return out


CASES = [  # n, ncat, seq_type, ntaxa, nsites, mem_mode
    (4, 4, 0, 14, 400, 0),      # one workgroup per solve
    (4, 4, 0, 20, 30000, 0),    # posted exchange between workgroups, one tile per wave (theta in registers)
    (4, 4, 0, 9, 140000, 0),    # more tiles than resident waves: theta re-read from memory
    (4, 4, 0, 11, 900, 1),      # LM_ALL_BRANCH
    (4, 1, 0, 9, 700, 0),
    (20, 4, 1, 12, 1500, 0),
    (20, 1, 1, 8, 300, 0),
    (64, 1, 2, 9, 500, 0),      # leaf tables: lengths that only exist on the device go through TabJob::len_p
]


@pytest.mark.parametrize("n,ncat,seq_type,ntaxa,nsites,mem_mode", CASES)
@pytest.mark.parametrize("kernel", ["persistent", "per-step"])
def test_sweep_equals_the_per_branch_form(pkg, synth, oracle, n, ncat, seq_type, ntaxa, nsites, mem_mode, kernel, monkeypatch):
    """kernel: 4-state engines run a sweep as ONE persistent launch (k_sweep4); IQHIP_SWEEP_KERNEL=0 selects the general form
    -- two launches per step enqueued back to back, later steps reading earlier lengths from device memory -- which is
    what 20- / 64-state engines always use"""
    if kernel == "per-step":
        if n != 4:
            pytest.skip("matrix-core engines have the per-step form only")
        monkeypatch.setenv("IQHIP_SWEEP_KERNEL", "0")
    made = []

    def make():
        t, ot, model, pat, freq = make_case(synth, oracle, pkg, ntaxa, nsites, n, ncat, 4300 + n + ntaxa, seq_type=seq_type,
                                            missing=0.02, mem_mode=mem_mode)
        made.append((ot, model, pat, freq))
        return t

    out = run_both(make, iterations=2, start=0.15)
    (l0, len0, c0, s0, _), (l1, len1, c1, s1, t1) = out[False], out[True]
    assert len0.keys() == len1.keys()
    for k in len0:
        assert len0[k] == len1[k], (k, len0[k], len1[k])     # same evaluated points, same kernels: same bits
    assert l0 == l1
    assert c0 == c1                                          # derivative evaluations
    nbranch = len(len0)
    assert s1 < s0 and s0 - s1 >= nbranch - 1                # one submission per sweep instead of one per branch
    ot, model, pat, freq = made[-1]
    ot2 = oracle.OracleTree(t1.tree_string(), n, seq_type, pat, freq, None, model)
    ref, _ = ot2.likelihood()
    assert abs(l1 - ref) <= 1e-8 * abs(ref)


def test_sweep_applies_the_diverged_newton_reset(pkg, synth, oracle):
    """phylotree.cpp:2167-2176 inside the sweep: with max_branch_length 0.05 most optima lie above 0.95 * max"""
    def make():
        return make_case(synth, oracle, pkg, 10, 600, 4, 4, 5151)[0]
    out = run_both(make, iterations=1, bounds=(1e-6, 0.05))
    (l0, len0, c0, _, _), (l1, len1, c1, _, _) = out[False], out[True]
    assert len0 == len1 and l0 == l1
    assert sum(1 for v in len1.values() if v > 0.0475) >= 3 and any(v <= 0.0475 for v in len1.values())


def test_sweep_on_a_multifurcating_tree(pkg, synth, oracle):
    nwk = synth.random_multifurcating_newick(13, 99)
    model = synth.gtr_model(alpha=0.9, ncat=4)
    st = synth.simulate_alignment(synth.random_tree_newick(13, 98), model, 800, 7)
    pat, freq = synth.compress_patterns(st)

    def make():
        t = pkg.PhyloTree(nwk)
        t.set_alignment(4, 0, pat, freq)
        t.set_model(model)
        t.set_likelihood_kernel(pkg.LK_EIGEN_HIP)
        t.attach_engine(0)
        return t
    out = run_both(make, iterations=2)
    assert out[False][1] == out[True][1] and out[False][0] == out[True][0]
    ot = oracle.OracleTree(out[True][4].tree_string(), 4, 0, pat, freq, None, model)
    ref, _ = ot.likelihood()
    assert abs(out[True][0] - ref) <= 1e-8 * abs(ref)


def test_sweep_on_a_sharded_engine_runs_step_by_step(pkg, synth, oracle):
    """every Newton step of a sharded engine contains an all-reduce: iqhip_optimize_sweep then walks its steps one after
    the other inside the call -- same interface, same results as a plain engine to summation order"""
    t, ot, args = case(synth, oracle, pkg, 10, 3000, 4, 4, 6161)
    ts = sharded_tree(pkg, args, [0, 0], pkg.REDUCE_HOST)
    vals = []
    for tree in (t, ts):
        tree.set_device_sweep(True)
        vals.append((tree.optimize_all_branches(iterations=2, tolerance=1e-9), lengths(tree)))
    assert abs(vals[0][0] - vals[1][0]) <= 1e-10 * abs(vals[0][0])
    for k in vals[0][1]:
        assert abs(vals[0][1][k] - vals[1][1][k]) <= 1e-7 * max(vals[0][1][k], 1e-3)


def test_c_abi_sweep_arguments_are_checked(pkg, synth, oracle):
    import ctypes as C
    t, *_ = make_case(synth, oracle, pkg, 8, 200, 4, 4, 77)
    t.compute_likelihood()
    lib = pkg.libiqhip()

    class Step(C.Structure):
        _fields_ = [("ops", C.c_void_p), ("len_from", C.POINTER(C.c_int32)), ("nops", C.c_int32), ("_pad", C.c_int32),
                    ("a", pkg.BranchEnd), ("b", pkg.BranchEnd), ("xguess", C.c_double)]

    class Res(C.Structure):
        _fields_ = [("optx", C.c_double), ("d2l", C.c_double), ("lnl", C.c_double), ("nsteps", C.c_int32), ("status", C.c_int32)]
    lib.iqhip_optimize_sweep.argtypes = [C.c_void_p, C.POINTER(Step), C.c_int, C.c_double, C.c_double, C.c_double, C.c_int,
                                         C.c_double, C.POINTER(C.c_double), C.POINTER(Res)]
    res = (Res * 2)()
    steps = (Step * 2)()
    assert lib.iqhip_optimize_sweep(t.engine, steps, 0, 1e-6, 100.0, 1e-6, 100, 0.95, None, res) == 2      # no steps
    assert lib.iqhip_optimize_sweep(t.engine, steps, 1, 1e-6, 1e-7, 1e-6, 100, 0.95, None, res) == 2       # x2 <= x1
    op = pkg.NodeOp(1, 0, 0, 0, 1, 0.1, 0.1, 0, 0)
    lf = (C.c_int32 * 2)(0, -1)   # step 0 refers to its own result
    steps[0].ops = C.cast(C.pointer(op), C.c_void_p)
    steps[0].len_from = lf
    steps[0].nops = 1
    steps[0].xguess = 0.1
    assert lib.iqhip_optimize_sweep(t.engine, steps, 1, 1e-6, 100.0, 1e-6, 100, 0.95, None, res) == 2
    assert b"earlier steps" in lib.iqhip_last_error()
