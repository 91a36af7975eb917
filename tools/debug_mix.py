import sys, os, importlib
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import __graft_entry__ as g
import numpy as np
pkg = g.load_package(); synth = importlib.import_module("iqtree_amd.synth"); od = g.load_oracle()
def run(n, seq_type, nclass, ncat, fused, ntaxa, nsites, missing, seed=7):
    model = synth.mixture_model(n, nclass, seed, ncat=ncat, fused=fused)
    su = od.state_unknown_for(n, seq_type)
    nwk = synth.random_tree_newick(ntaxa, seed)
    st = synth.simulate_alignment(nwk, model.classes[0], nsites, seed + 1, missing, su)
    pat, freq = synth.compress_patterns(st)
    ot = od.OracleTree(nwk, n, seq_type, pat, freq, None, model)
    t = pkg.PhyloTree(nwk); t.set_alignment(n, seq_type, pat, freq); t.set_model(model); t.attach_engine(0)
    v = t.compute_likelihood(); r, (a, b) = ot.likelihood()
    _, oplh = ot.branch_lnl(a, b)
    plh = t.fetch_pattern_lh()
    bad = np.nonzero(np.abs(plh - oplh) > 1e-9 * np.abs(oplh))[0]
    print("n", n, "nclass", nclass, "ncat", ncat, "fused", fused, "missing", missing, "nptn", pat.shape[1], "rel", abs(v - r) / abs(r), "bad patterns", len(bad), bad[:10])
    if len(bad):
        p = bad[0]; print("   pattern", pat[:, p], "got", plh[p], "want", oplh[p])
for args in ((4, 0, 2, 4, False, 8, 300, 0.0), (4, 0, 2, 4, False, 8, 300, 0.05), (4, 0, 2, 1, True, 8, 300, 0.0), (4, 0, 1+1, 2, False, 5, 100, 0.0),
             (64, 2, 2, 1, True, 7, 150, 0.0), (20, 1, 2, 2, False, 7, 150, 0.02)):
    try:
        run(*args)
    except Exception as e:
        print(args, "ERR", e)
