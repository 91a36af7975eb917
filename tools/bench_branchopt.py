"""Hot loop 2: one optimizeAllBranches sweep -- host Newton loop (one submission per derivative evaluation),
device Newton loop (one submission per branch), whole sweep in one submission (iqhip_optimize_sweep).
usage: python tools/bench_branchopt.py [--json]   (one line per shape and form; --json: a list of dicts on stdout)"""
import importlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); synth = importlib.import_module("iqtree_amd.synth")


def one_shape(T, P, n, forms=("host", "branch", "sweep"), reps=3):
    model = synth.gtr_model() if n == 4 else synth.random_reversible_model(20, 7, alpha=0.9, ncat=4)
    nwk, pat, freq = synth.make_workload(T, P, model, seed=3)
    rows = []
    for form in forms:
        t = pkg.PhyloTree(nwk); t.set_alignment(n, 0 if n == 4 else 1, pat, freq); t.set_model(model); t.attach_engine(0)
        t.set_device_newton(form != "host")
        t.set_device_sweep(form == "sweep")
        best = None
        for rep in range(reps):   # (the first repetition pays allocations and clock ramp)
            for a in range(t.num_nodes):
                for b, _ in t.neighbors(a):
                    if a < b: t.set_branch_length(a, b, 0.1, clear_reverse=False)
            t.clear_all_partial_lh(); t.compute_likelihood()
            c0 = t.num_derv_calls; t0 = time.perf_counter()
            lnl = t.optimize_all_branches(iterations=1, tolerance=1e-3)
            dt = time.perf_counter() - t0
            if best is None or dt < best[0]: best = (dt, t.num_derv_calls - c0, lnl)
        dt, nev, lnl = best
        nb = 2 * T - 3
        # bytes a branch has to move at least: two node updates (3 vectors each), theta written once, read once per evaluation
        V = P * n * model.ncat * 8.0
        stream_us = (6 * V + V + (nev / nb) * V) / 6.3e12 * 1e6
        rows.append(dict(ntaxa=T, patterns=P, nstates=n, form=form, branches=nb, sweep_ms=dt * 1e3, us_per_branch=dt * 1e6 / nb,
                         derivative_evaluations=nev, lnL=lnl, stream_bound_us_per_branch=stream_us,
                         hbm_frac=stream_us / (dt * 1e6 / nb) * 6.3 / 8.0))
        t.close()
    return rows


if __name__ == "__main__":
    out = []
    for (T, P, n) in ((44, 355, 4), (50, 5000, 4), (50, 100000, 4), (50, 20000, 20)):
        out += one_shape(T, P, n)
    if "--json" in sys.argv:
        print(json.dumps(out))
    else:
        for r in out:
            print("taxa %(ntaxa)d patterns %(patterns)d states %(nstates)d  %(form)-6s: one sweep over %(branches)d branches %(sweep_ms).2f ms, "
                  "%(derivative_evaluations)d derivative evaluations, %(us_per_branch).1f us per branch (stream bound %(stream_bound_us_per_branch).1f), lnL %(lnL).6f" % r)
