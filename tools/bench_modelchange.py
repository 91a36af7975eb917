"""Hot loop 1 as a model optimiser runs it: new model parameters -> clearAllPartialLH -> computeLikelihood.
Reports ms per evaluation with and without a model change per evaluation (DNA 50 x 100k GTR+G4)."""
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
P = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
models = [synth.gtr_model(alpha=0.5 + 0.01 * k) for k in range(8)]
nwk = synth.random_tree_newick(50, 1)
st = synth.simulate_alignment(nwk, models[0], int(P * 1.05) + 64, 3)
pat, freq = synth.compress_patterns(st)
pat, freq = np.ascontiguousarray(pat[:, :P]), freq[:P].copy()
t = pkg.PhyloTree(nwk)
t.set_alignment(4, 0, pat, freq)
t.set_model(models[0])
t.attach_engine(0)
for _ in range(5):
    t.clear_and_compute_likelihood()
reps = 200
t0 = time.perf_counter()
for _ in range(reps):
    t.clear_and_compute_likelihood()
same = (time.perf_counter() - t0) / reps
t0 = time.perf_counter()
for k in range(reps):
    t.set_model(models[k % 8])
    t.clear_and_compute_likelihood()
chg = (time.perf_counter() - t0) / reps
print("patterns %d: %.4f ms per evaluation, %.4f ms with a model change per evaluation (+%.1f us)" %
      (P, same * 1e3, chg * 1e3, (chg - same) * 1e6))
