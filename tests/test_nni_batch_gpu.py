"""Batched NNI evaluation (iqhip_optimize_branch_batch, host mirror evaluateNNIsBatch): all 2(n-3) nni1
candidates of a tree in one submission against getBestNNIForBran run branch by branch (the reference's
IQTree::evaluateNNIs order, phylotree.cpp:2873-3066), and against the oracle for the best candidate."""
import numpy as np
import pytest

from test_parity_gpu import make_case, LNL_RTOL

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,ncat,seq_type,ntaxa,nptn", [(4, 4, 0, 14, 600), (20, 4, 1, 9, 300), (64, 1, 2, 7, 150),
                                                        (4, 4, 0, 40, 70000)])
def test_batch_matches_branch_by_branch(pkg, synth, oracle, n, ncat, seq_type, ntaxa, nptn):
    t, ot, model, pat, freq = make_case(synth, oracle, pkg, ntaxa, nptn, n, ncat, 3300 + n + ntaxa, seq_type=seq_type,
                                        mem_mode=pkg.LM_ALL_BRANCH)
    lnl = t.compute_likelihood()
    tree0 = t.tree_string()
    batch = t.evaluate_nnis_batch()
    assert len(batch) == 2 * (ntaxa - 3)
    assert t.tree_string() == tree0                                   # nothing is changed in the tree
    assert abs(t.compute_likelihood() - lnl) <= 1e-12 * abs(lnl)
    by_branch = {}
    for m in batch:
        by_branch.setdefault((m["node1"], m["node2"]), []).append(m)
    assert all(len(v) == 2 for v in by_branch.values())
    for (a, b), two in list(by_branch.items())[:12]:                  # the sequential evaluator, branch by branch
        seq = t.nni_for_branch(a, b, nni5=False)
        for c in range(2):
            newloglh, nei1, nei2, lens = seq[c]
            assert (two[c]["node1_nei"], two[c]["node2_nei"]) == (nei1, nei2)
            assert abs(two[c]["new_len"] - lens[0]) <= 1e-9 * max(1e-6, lens[0])
            assert abs(two[c]["newloglh"] - newloglh) <= 1e-10 * abs(newloglh)
    best = max(batch, key=lambda m: m["newloglh"])
    assert np.isfinite(best["newloglh"]) and best["new_len"] >= 1e-6
    # the batch can be repeated (scratch vectors and counters are reused)
    again = t.evaluate_nnis_batch()
    assert [m["newloglh"] for m in again] == [m["newloglh"] for m in batch]


def test_batch_needs_all_branch_mode(pkg, synth, oracle):
    t, *_ = make_case(synth, oracle, pkg, 8, 100, 4, 4, 77)
    t.compute_likelihood()
    with pytest.raises(pkg.HostError, match="LM_ALL_BRANCH"):
        t.evaluate_nnis_batch()


@pytest.mark.parametrize("n,ncat,seq_type,ntaxa,nptn", [(4, 4, 0, 13, 500), (20, 4, 1, 8, 250), (4, 4, 0, 30, 40000)])
def test_nni5_batch_matches_branch_by_branch(pkg, synth, oracle, n, ncat, seq_type, ntaxa, nptn):
    """nni5 (the reference's default): five branches per candidate, optimised in the reference's order; the second
    swap of a branch starts from the lengths the first swap left."""
    t, ot, model, pat, freq = make_case(synth, oracle, pkg, ntaxa, nptn, n, ncat, 4400 + n + ntaxa, seq_type=seq_type,
                                        mem_mode=pkg.LM_ALL_BRANCH)
    lnl = t.compute_likelihood()
    tree0 = t.tree_string()
    batch = t.evaluate_nnis5_batch()
    assert len(batch) == 2 * (ntaxa - 3) and t.tree_string() == tree0
    assert abs(t.compute_likelihood() - lnl) <= 1e-12 * abs(lnl)
    checked = 0
    for k in range(0, len(batch), 2):
        a, b = batch[k]["node1"], batch[k]["node2"]
        seq = t.nni_for_branch(a, b, nni5=True)
        for c in range(2):
            newloglh, nei1, nei2, lens = seq[c]
            m = batch[k + c]
            assert (m["node1_nei"], m["node2_nei"]) == (nei1, nei2)
            np.testing.assert_allclose(m["new_lens"], lens, rtol=1e-7, atol=1e-12)
            assert abs(m["newloglh"] - newloglh) <= 1e-9 * abs(newloglh)
            checked += 1
        if checked >= 16:
            break
    # the best nni5 candidate is at least as good as the best nni1 candidate evaluated the same way
    best1 = max(m["newloglh"] for m in t.evaluate_nnis_batch())
    best5 = max(m["newloglh"] for m in batch)
    assert best5 >= best1 - 1e-6 * abs(best1)


def test_compute_all_partial_lh_is_one_submission(pkg, synth, oracle):
    t, ot, *_ = make_case(synth, oracle, pkg, 25, 300, 4, 4, 5151, mem_mode=pkg.LM_ALL_BRANCH)
    lnl = t.compute_likelihood()
    n0 = t.num_submissions
    t.compute_all_partial_lh()
    # 3T-6 directed vectors point at internal nodes; the traversal made T-2 of them
    assert t.num_submissions == n0 + 1 and len(t.last_plan()) == 2 * (25 - 2)
    from test_parity_gpu import check_all_vectors
    assert check_all_vectors(t, ot) == 3 * 25 - 6
    # every directed vector is now valid: any branch gives the same lnL without further node updates
    n1 = t.num_partial_lh_computations
    for a in range(t.num_nodes):
        for b, _ in t.neighbors(a):
            if a < b:
                assert abs(t.compute_likelihood_branch(a, b) - lnl) <= LNL_RTOL * abs(lnl)
    assert t.last_plan() == []


def test_batch_and_consumer_entry_points_reject_bad_input(pkg, synth, oracle):
    """error behaviour of the newer entry points: int status + message, nothing launched on bad input."""
    import ctypes as C
    lib = pkg.libiqhip()

    class Task(C.Structure):
        _fields_ = [("ops", C.c_void_p), ("nops", C.c_int32), ("max_steps", C.c_int32), ("a", pkg.BranchEnd), ("b", pkg.BranchEnd),
                    ("xguess", C.c_double), ("x1", C.c_double), ("x2", C.c_double), ("xacc", C.c_double)]

    class Result(C.Structure):
        _fields_ = [("optx", C.c_double), ("d2l", C.c_double), ("lnl", C.c_double), ("nsteps", C.c_int32), ("status", C.c_int32)]

    lib.iqhip_optimize_branch_batch.argtypes = [C.c_void_p, C.POINTER(Task), C.c_int, C.POINTER(C.c_double), C.POINTER(Result)]
    lib.iqhip_pattern_lh_cat.argtypes = [C.c_void_p, C.c_double, C.POINTER(C.c_double)]
    t, ot, *_ = make_case(synth, oracle, pkg, 8, 100, 4, 4, 8080, mem_mode=pkg.LM_ALL_BRANCH)
    t.compute_likelihood()
    a, b = t.current_branch()
    inner = [(x, y) for x in range(t.num_nodes) for y, _ in t.neighbors(x) if not ot.is_leaf(x) and not ot.is_leaf(y)]
    x, y = inner[0]
    t.compute_all_partial_lh()
    kx, ky = t.neighbor_info(x, y)["key"], t.neighbor_info(y, x)["key"]
    res = (Result * 1)()
    good = Task(None, 0, 10, pkg.key_end(kx), pkg.key_end(ky), 0.1, 1e-6, 100.0, 1e-6)
    assert lib.iqhip_optimize_branch_batch(t.engine, (Task * 1)(good), 1, None, res) == 0
    assert res[0].status == 0 and 1e-6 <= res[0].optx <= 100.0 and np.isfinite(res[0].lnl)
    # the same branch through the single-branch evaluator
    t.optimize_one_branch(x, y, clear_lh=False)
    assert abs(t.neighbor_info(x, y)["length"] - res[0].optx) <= 5e-6     # both stop within xacc of the optimum
    for bad in (Task(None, 0, 0, pkg.key_end(kx), pkg.key_end(ky), 0.1, 1e-6, 100.0, 1e-6),       # max_steps < 1
                Task(None, 0, 10, pkg.key_end(kx), pkg.key_end(ky), 0.1, 1.0, 0.5, 1e-6),         # x2 <= x1
                Task(None, 0, 10, pkg.key_end(0xabcdef), pkg.key_end(ky), 0.1, 1e-6, 100.0, 1e-6),  # unknown key
                Task(None, 2, 10, pkg.key_end(kx), pkg.key_end(ky), 0.1, 1e-6, 100.0, 1e-6)):     # nops without ops
        assert lib.iqhip_optimize_branch_batch(t.engine, (Task * 1)(bad), 1, None, res) != 0
        assert len(lib.iqhip_last_error()) > 0
    assert lib.iqhip_optimize_branch_batch(t.engine, None, 1, None, res) != 0
    out = np.zeros(t.nptn * 4)
    dp = out.ctypes.data_as(C.POINTER(C.c_double))
    t2, *_ = make_case(synth, oracle, pkg, 8, 100, 4, 4, 8081)
    t2.compute_likelihood()
    assert lib.iqhip_pattern_lh_cat(t2.engine, 0.1, dp) != 0 and b"compute_theta" in lib.iqhip_last_error()
    assert lib.iqhip_pattern_lh_cat(t.engine, -1.0, dp) != 0


def test_batch_sizes_may_vary_between_calls(pkg, synth, oracle):
    """10, then 3, then 10 tasks with several workgroups per task (G > 1).  The per-task arrival
    counters of a launch must start at zero whatever the task counts of the launches before it were; a dirty counter
    lets workgroups read partial sums their peers have not written yet (silently wrong, timing dependent)."""
    import ctypes as C
    lib = pkg.libiqhip()

    class Task(C.Structure):
        _fields_ = [("ops", C.c_void_p), ("nops", C.c_int32), ("max_steps", C.c_int32), ("a", pkg.BranchEnd), ("b", pkg.BranchEnd),
                    ("xguess", C.c_double), ("x1", C.c_double), ("x2", C.c_double), ("xacc", C.c_double)]

    class Result(C.Structure):
        _fields_ = [("optx", C.c_double), ("d2l", C.c_double), ("lnl", C.c_double), ("nsteps", C.c_int32), ("status", C.c_int32)]

    lib.iqhip_optimize_branch_batch.argtypes = [C.c_void_p, C.POINTER(Task), C.c_int, C.POINTER(C.c_double), C.POINTER(Result)]
    t, ot, *_ = make_case(synth, oracle, pkg, 16, 6000, 4, 4, 9191, mem_mode=pkg.LM_ALL_BRANCH)
    assert t.nptn > 4 * 256                      # several workgroups per task
    t.compute_likelihood()
    t.compute_all_partial_lh()
    inner = [(x, y) for x in range(t.num_nodes) for y, _ in t.neighbors(x)
             if x < y and not ot.is_leaf(x) and not ot.is_leaf(y)]
    assert len(inner) >= 10
    tasks = [Task(None, 0, 10, pkg.key_end(t.neighbor_info(x, y)["key"]), pkg.key_end(t.neighbor_info(y, x)["key"]),
                  0.05 + 0.01 * k, 1e-6, 100.0, 1e-6) for k, (x, y) in enumerate(inner[:10])]

    def run(sub):
        res = (Result * len(sub))()
        assert lib.iqhip_optimize_branch_batch(t.engine, (Task * len(sub))(*sub), len(sub), None, res) == 0, lib.iqhip_last_error()
        return [(r.optx, r.d2l, r.lnl, r.nsteps, r.status) for r in res]

    single = [run([k])[0] for k in tasks]        # one task per launch: the branch-by-branch values
    for _ in range(3):
        first = run(tasks)
        small = run(tasks[4:7])
        third = run(tasks)
        for got, want in ((first, single), (third, single), (small, single[4:7])):
            for g, w in zip(got, want):
                assert g[3:] == w[3:] == (w[3], 0)                      # same step count, status ok
                np.testing.assert_allclose(g[:3], w[:3], rtol=1e-10)    # optx, d2l, lnl (same grid -> usually bitwise)
