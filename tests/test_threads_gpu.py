"""Partition-style use (PhyloSuperTree calls the kernels of different PhyloTree objects from an OpenMP team,
phylosupertree.cpp:970,1017): engines are per tree, each on its own stream, and may be driven from different host
threads at once."""
import threading
import time

import numpy as np
import pytest

from test_parity_gpu import make_case, LNL_RTOL

pytestmark = pytest.mark.gpu


def test_engines_on_concurrent_host_threads(pkg, synth, oracle):
    shapes = [(4, 4, 0, 12, 700), (20, 4, 1, 9, 300), (4, 2, 0, 20, 1500), (64, 1, 2, 7, 120),
              (4, 4, 0, 30, 4000), (20, 1, 1, 11, 500), (4, 8, 0, 10, 900), (20, 4, 1, 14, 2500)]
    trees, refs = [], []
    for k, (n, ncat, st, ntaxa, nptn) in enumerate(shapes):
        t, ot, *_ = make_case(synth, oracle, pkg, ntaxa, nptn, n, ncat, 9900 + k, seq_type=st)
        trees.append(t)
        refs.append(ot.likelihood()[0])
    reps = 30
    seq = [t.clear_and_compute_likelihood() for t in trees]
    for v, r in zip(seq, refs):
        assert abs(v - r) <= LNL_RTOL * abs(r)
    t0 = time.perf_counter()
    for _ in range(reps):
        for t in trees:
            t.clear_and_compute_likelihood()
    t_seq = time.perf_counter() - t0
    out = [None] * len(trees)
    errs = []

    def work(i):
        try:
            vals = [trees[i].clear_and_compute_likelihood() for _ in range(reps)]
            a, b = trees[i].current_branch()
            trees[i].optimize_one_branch(a, b)
            out[i] = vals
        except Exception as e:  # noqa
            errs.append((i, repr(e)))

    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(len(trees))]
    for x in th:
        x.start()
    for x in th:
        x.join()
    t_par = time.perf_counter() - t0
    assert not errs, errs
    for i, vals in enumerate(out):
        assert all(v == seq[i] for v in vals)          # same numbers as the single-threaded run, every time
    print("8 partitions x %d evaluations: one thread %.2f ms, eight threads %.2f ms" % (reps, t_seq * 1e3, t_par * 1e3))


def test_multi_workgroup_newton_solves_of_two_engines_overlap(pkg, synth, oracle):
    """Two >= 100 000-pattern trees (grid-barrier Newton kernels with one workgroup per CU each) optimised from two
    host threads at once, plus batched NNI evaluations: barrier kernels of different engines are chained per device
    (kernels_newton.hip, BarrierLaunchGuard), so neither can strand the other's workgroups; results equal the
    single-threaded ones."""
    trees = []
    for k in range(2):
        t, ot, *_ = make_case(synth, oracle, pkg, 40, 150000, 4, 4, 9950 + k, mem_mode=pkg.LM_ALL_BRANCH)
        assert t.nptn >= 100000
        t.compute_likelihood()
        trees.append(t)
    edges = [[(a, b) for a in range(t.num_nodes) for b, _ in t.neighbors(a) if a < b] for t in trees]
    lens0 = [[t.neighbor_info(a, b)["length"] for (a, b) in edges[i]] for i, t in enumerate(trees)]

    def sweep(i, out):
        t = trees[i]
        for (a, b), ln in zip(edges[i], lens0[i]):      # same starting tree for every sweep
            t.set_branch_length(a, b, ln, clear_reverse=False)
        t.clear_all_partial_lh()
        vals = []
        for rep in range(3):
            for (a, b) in edges[i]:
                t.set_branch_length(a, b, 0.05 + 0.01 * rep, clear_reverse=True)
                vals.append(t.optimize_one_branch(a, b))
            vals.append(max(m["newloglh"] for m in t.evaluate_nnis_batch()))
        out[i] = vals

    seq = [None, None]
    for i in range(2):
        sweep(i, seq)
    par = [None, None]
    errs = []

    def work(i):
        try:
            sweep(i, par)
        except Exception as e:  # noqa
            errs.append((i, repr(e)))

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errs, errs
    for i in range(2):
        np.testing.assert_allclose(par[i], seq[i], rtol=1e-12)
