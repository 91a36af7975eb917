"""Create / use / destroy engines repeatedly and watch the device's free memory (hipMemGetInfo)."""
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); synth = importlib.import_module("iqtree_amd.synth")
hip = C.CDLL("libamdhip64.so")
def free_mb():
    f, t = C.c_size_t(), C.c_size_t()
    hip.hipMemGetInfo(C.byref(f), C.byref(t))
    return f.value / 1e6
cases = [(4, 0, synth.gtr_model(), 30000), (20, 1, synth.random_reversible_model(20, 3, ncat=4), 3000),
         (20, 1, synth.random_reversible_model(20, 3, ncat=4), 9500),   # (>= 8192 patterns: cherry tables and their pair engine)
         (20, 1, synth.mixture_model(20, 3, 5, ncat=4), 3000), (64, 2, synth.random_reversible_model(64, 4, alpha=None, ncat=1), 2000)]
base = None
for rep in range(6):
    for n, st_type, model, P in cases:
        nwk = synth.random_tree_newick(20, 1)
        sim = model.classes[0] if hasattr(model, "classes") else model
        st = synth.simulate_alignment(nwk, sim, P, 2)
        pat, freq = synth.compress_patterns(st)
        t = pkg.PhyloTree(nwk); t.set_mem_mode(pkg.LM_ALL_BRANCH); t.set_alignment(n, st_type, pat, freq); t.set_model(model); t.attach_engine(0)
        t.compute_likelihood(); t.optimize_all_branches(iterations=1)
        if not hasattr(model, "classes"):
            t.evaluate_nnis_batch()
        t.set_boot_samples(freq[None, :].astype("float32")); t.compute_rell()
        t.close()
    f = free_mb()
    if base is None: base = f
    print("round %d free %.1f MB (delta vs first %.1f MB)" % (rep, f, f - base))
assert abs(f - base) < 64, "device memory is leaking"
print("OK")
