import importlib, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as g
pkg = g.load_package(); synth = importlib.import_module("iqtree_amd.synth")
model = synth.gtr_model()
T = 1100
nwk = synth.random_tree_newick(T, 5, 0.01, 0.05)
st = synth.simulate_alignment(nwk, model, 200, 6)
pat, freq = synth.compress_patterns(st)
t = pkg.PhyloTree(nwk); t.set_mem_mode(pkg.LM_ALL_BRANCH); t.set_alignment(4, 0, pat, freq); t.set_model(model); t.attach_engine(0)
lnl = t.compute_likelihood()
t0 = time.perf_counter(); b1 = t.evaluate_nnis_batch(); t1 = time.perf_counter(); b5 = t.evaluate_nnis5_batch(); t2 = time.perf_counter()
print("taxa", T, "patterns", pat.shape[1], "candidates", len(b1), "nni1 batch %.1f ms, nni5 batch %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
bad = 0
for k in (0, 400, 1000, 1600, 2190):
    a, b = b1[k]["node1"], b1[k]["node2"]
    k0 = k - (k % 2)
    s1 = t.nni_for_branch(a, b, nni5=False); s5 = t.nni_for_branch(a, b, nni5=True)
    for c in range(2):
        d1 = abs(b1[k0 + c]["newloglh"] - s1[c][0]) / abs(s1[c][0]); d5 = abs(b5[k0 + c]["newloglh"] - s5[c][0]) / abs(s5[c][0])
        if d1 > 1e-9 or d5 > 1e-9: bad += 1
        print(k0 + c, "rel diff nni1 %.2e nni5 %.2e" % (d1, d5))
print("BAD" if bad else "OK")
