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
