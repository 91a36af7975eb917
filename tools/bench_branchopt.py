"""Hot loop 2: optimizeAllBranches, host Newton loop vs device Newton loop."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); synth = importlib.import_module("iqtree_amd.synth")
for (T, P, n) in ((44, 355, 4), (50, 5000, 4), (50, 100000, 4), (50, 20000, 20)):
    model = synth.gtr_model() if n == 4 else synth.random_reversible_model(20, 7, alpha=0.9, ncat=4)
    nwk, pat, freq = synth.make_workload(T, P, model, seed=3)
    for mode in (False, True):
        t = pkg.PhyloTree(nwk); t.set_alignment(n, 0 if n == 4 else 1, pat, freq); t.set_model(model); t.attach_engine(0)
        t.set_device_newton(mode)
        for a in range(t.num_nodes):
            for b, _ in t.neighbors(a):
                if a < b: t.set_branch_length(a, b, 0.1, clear_reverse=False)
        t.clear_all_partial_lh(); t.compute_likelihood()
        c0 = t.num_derv_calls; t0 = time.perf_counter()
        lnl = t.optimize_all_branches(iterations=1, tolerance=1e-3)
        dt = time.perf_counter() - t0
        print("taxa %d patterns %d states %d  %-6s newton: one sweep over %d branches %.2f ms, %d derivative evaluations, "
              "%.1f us per branch, lnL %.6f" % (T, P, n, "device" if mode else "host", 2 * T - 3, dt * 1e3,
                                                 t.num_derv_calls - c0, dt * 1e6 / (2 * T - 3), lnl))
