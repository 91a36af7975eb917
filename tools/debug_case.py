import sys, os, importlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); synth = importlib.import_module("iqtree_amd.synth"); od = g.load_oracle()
ncat = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ntaxa = int(sys.argv[2]) if len(sys.argv) > 2 else 14
model = synth.gtr_model(alpha=0.9, ncat=ncat)
nwk = synth.random_tree_newick(ntaxa, 100 + ncat)
st = synth.simulate_alignment(nwk, model, 700, 101 + ncat, 0.05, 18)
pat, freq = synth.compress_patterns(st)
ot = od.OracleTree(nwk, 4, 0, pat, freq, None, model)
t = pkg.PhyloTree(nwk); t.set_alignment(4, 0, pat, freq); t.set_model(model); t.attach_engine(0)
lnl = t.compute_likelihood(); ref, _ = ot.likelihood()
print("lnl", lnl, "ref", ref)
for k, p in enumerate(t.last_plan()):
    a, b = p["dst"]
    got = t.fetch_partial(a, b); exp, sc, sf = ot.partial(a, b)
    err = np.abs(got - exp).max() / np.abs(exp).max()
    print(k, p["dst"], "L", p["left"], p["left_leaf"], "R", p["right"], p["right_leaf"], "err %.2e" % err,
          "sc_ok", np.array_equal(t.fetch_scale_num(a, b), sc))
