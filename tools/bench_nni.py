"""NNI evaluation (hot loop 2 of the tree search): all 2(n-3) nni1 candidates, branch by branch
(getBestNNIForBran, the reference's order) vs one batched submission (evaluateNNIsBatch)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); synth = importlib.import_module("iqtree_amd.synth")
for (T, P) in ((44, 355), (50, 5000), (50, 100000)):
    model = synth.gtr_model()
    nwk, pat, freq = synth.make_workload(T, P, model, seed=3)
    t = pkg.PhyloTree(nwk); t.set_mem_mode(pkg.LM_ALL_BRANCH); t.set_alignment(4, 0, pat, freq); t.set_model(model)
    t.attach_engine(0)
    t.compute_likelihood()
    b = t.evaluate_nnis_batch()
    t0 = time.perf_counter(); reps = 5
    for _ in range(reps):
        b = t.evaluate_nnis_batch()
    tb = (time.perf_counter() - t0) / reps
    branches = sorted({(m["node1"], m["node2"]) for m in b})
    t0 = time.perf_counter()
    for (x, y) in branches:
        t.nni_for_branch(x, y, nni5=False)
    ts = time.perf_counter() - t0
    print("taxa %d patterns %d: %d candidates; nni1 branch by branch %.2f ms (%.1f us per candidate), batched %.2f ms "
          "(%.1f us per candidate)" % (T, P, len(b), ts * 1e3, ts * 1e6 / len(b), tb * 1e3, tb * 1e6 / len(b)))
    b5 = t.evaluate_nnis5_batch()
    t0 = time.perf_counter()
    for _ in range(reps):
        b5 = t.evaluate_nnis5_batch()
    tb5 = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for (x, y) in branches:
        t.nni_for_branch(x, y, nni5=True)
    ts5 = time.perf_counter() - t0
    print("    nni5 branch by branch %.2f ms (%.1f us per candidate), batched %.2f ms (%.1f us per candidate)" %
          (ts5 * 1e3, ts5 * 1e6 / len(b5), tb5 * 1e3, tb5 * 1e6 / len(b5)))
