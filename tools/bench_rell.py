"""UFBoot RELL scores on the device (iqhip_rell): time per tree for 1000 bootstrap samples, against the
float / 8-lane CPU dot products of the reference restated in the oracle (phylokernel.h:55-61)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
import importlib  # noqa: E402

synth = importlib.import_module("iqtree_amd.synth")
od = g.load_oracle()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
NS = 1000
model = synth.gtr_model()
nwk = synth.random_tree_newick(20, 1)
st = synth.simulate_alignment(nwk, model, int(P * 1.05) + 64, 3)
pat, freq = synth.compress_patterns(st)
pat, freq = np.ascontiguousarray(pat[:, :P]), freq[:P].copy()
t = pkg.PhyloTree(nwk)
t.set_alignment(4, 0, pat, freq)
t.set_model(model)
t.attach_engine(0)
lnl = t.compute_likelihood()
rng = np.random.default_rng(1)
w = rng.multinomial(int(freq.sum()), freq / freq.sum(), size=NS).astype(np.float32)
t.set_boot_samples(w)
r = t.compute_rell()
reps = 50
t0 = time.perf_counter()
for _ in range(reps):
    r = t.compute_rell()
dt = (time.perf_counter() - t0) / reps
plh = t.compute_pattern_likelihood()
x = plh.astype(np.float32)
pad = (-x.size) % 8
x8 = np.concatenate([x, np.zeros(pad, np.float32)])
t0 = time.perf_counter()
ncpu = 100
for s in range(ncpu):
    od.dot_float8(x8, np.concatenate([w[s], np.zeros(pad, np.float32)]))
cpu = (time.perf_counter() - t0) / ncpu * NS
print("patterns %d samples %d: device %.3f ms per tree (%.1f GB/s of sample matrix), CPU float8 port %.1f ms per tree (1 thread), "
      "max rel diff vs float64 dot %.2e" % (P, NS, dt * 1e3, NS * P * 4 / dt / 1e9, cpu * 1e3,
                                              np.max(np.abs(r - w.astype(np.float64) @ plh) / np.abs(r))))
