"""The two forms in which the workgroups of k_newton / k_newton_batch exchange their partial sums -- posted slots
(default) and the arrival counter (IQHIP_NEWTON_POSTS=0, also the form of solves with more than 125 steps) -- must give
the same iterates bit for bit, and so must plans that travel in the kernel arguments (IQHIP_SMALL_PLANS) and plans that
are copied to the plan buffer."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _tree(pkg, synth, seed=5, ntaxa=24, nsites=30000):
    model = synth.gtr_model()
    nwk, pat, freq = synth.make_workload(ntaxa, nsites, model, seed=seed)
    t = pkg.PhyloTree(nwk)
    t.set_mem_mode(1)   # LM_ALL_BRANCH: the batched NNI evaluator wants every vector
    t.set_alignment(4, 0, pat, freq)
    t.set_model(model)
    t.attach_engine(0)
    for a in range(t.num_nodes):
        for b, _ in t.neighbors(a):
            if a < b:
                t.set_branch_length(a, b, 0.12, clear_reverse=False)
    t.clear_all_partial_lh()
    return t


def _sweep(pkg, synth, monkeypatch, env, max_nr_step=100):
    for k in ("IQHIP_NEWTON_POSTS", "IQHIP_SMALL_PLANS"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    t = _tree(pkg, synth)
    assert t.nptn > 64 * 4 * 8           # several workgroups take part in a solve
    lnl0 = t.compute_likelihood()
    lnl = t.optimize_all_branches(iterations=1, tolerance=1e-3, max_nr_step=max_nr_step)
    lens = [t.neighbor_info(a, b)["length"] for a in range(t.num_nodes) for b, _ in t.neighbors(a) if a < b]
    nni = t.evaluate_nnis_batch()
    t.close()
    return lnl0, lnl, np.array(lens), [(m["new_len"], m["newloglh"]) for m in nni]


def test_posted_and_counted_exchange_give_the_same_iterates(pkg, synth, monkeypatch):
    ref = _sweep(pkg, synth, monkeypatch, {})
    assert ref[1] > ref[0]
    for env, steps in (({"IQHIP_NEWTON_POSTS": "0"}, 100), ({}, 200), ({"IQHIP_SMALL_PLANS": "0"}, 100),
                       ({"IQHIP_NEWTON_POSTS": "0", "IQHIP_SMALL_PLANS": "0"}, 100)):
        got = _sweep(pkg, synth, monkeypatch, env, max_nr_step=steps)
        assert got[0] == ref[0] and got[1] == ref[1], (env, steps)
        assert np.array_equal(got[2], ref[2]), (env, steps)
        assert got[3] == ref[3], (env, steps)
