import sys, os, importlib
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import __graft_entry__ as g
import numpy as np
pkg = g.load_package(); synth = importlib.import_module("iqtree_amd.synth"); od = g.load_oracle()
def case(ntaxa, nsites, pinvar=0.0):
    model = synth.gtr_model(alpha=0.9, ncat=4, pinvar=pinvar)
    nwk = synth.random_tree_newick(ntaxa, 12)
    st = synth.simulate_alignment(nwk, model, nsites, 13)
    keep = [p for p in range(st.shape[1]) if len(set(st[:, p].tolist())) > 1]
    st = st[:, keep]
    pat, freq = synth.compress_patterns(st)
    const = np.tile(np.arange(4, dtype=np.uint8)[None, :], (ntaxa, 1))
    pat2 = np.concatenate([pat, const], axis=1); freq2 = np.concatenate([freq, np.zeros(4)])
    invar = np.zeros(pat2.shape[1])
    ot = od.OracleTree(nwk, 4, 0, pat2, freq2, invar, model, n_unobs=4, nsites=float(freq.sum()))
    t = pkg.PhyloTree(nwk); t.set_alignment(4, 0, pat2, freq2, invar); t.set_model(model)
    t.set_ascertainment(4, float(freq.sum())); t.attach_engine(0)
    return t, ot
for ntaxa, ns in ((8, 100), (44, 80), (44, 300), (20, 5000)):
    t, ot = case(ntaxa, ns)
    try:
        v = t.compute_likelihood(); r, _ = ot.likelihood(); print(ntaxa, ns, t.nptn, v, r, abs(v-r)/abs(r))
    except Exception as e:
        print(ntaxa, ns, "ERR", e)
