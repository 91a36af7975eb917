"""Hot loop 2: optimizeAllBranches, host Newton loop vs device Newton loop."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); synth = importlib.import_module("iqtree_amd.synth")
for (T, P) in ((44, 355), (50, 5000), (50, 100000)):
    model = synth.gtr_model()
    nwk, pat, freq = synth.make_workload(T, P, model, seed=3)
    for mode in (False, True):
        t = pkg.PhyloTree(nwk); t.set_alignment(4, 0, pat, freq); t.set_model(model); t.attach_engine(0)
        t.set_device_newton(mode)
        for a in range(t.num_nodes):
            for b, _ in t.neighbors(a):
                if a < b: t.set_branch_length(a, b, 0.1, clear_reverse=False)
        t.clear_all_partial_lh(); t.compute_likelihood()
        c0 = t.num_derv_calls; t0 = time.perf_counter()
        lnl = t.optimize_all_branches(iterations=1, tolerance=1e-3)
        dt = time.perf_counter() - t0
        print("taxa %d patterns %d  %-6s newton: one sweep over %d branches %.2f ms, %d derivative evaluations, "
              "%.1f us per branch, lnL %.6f" % (T, P, "device" if mode else "host", 2 * T - 3, dt * 1e3,
                                                 t.num_derv_calls - c0, dt * 1e6 / (2 * T - 3), lnl))
