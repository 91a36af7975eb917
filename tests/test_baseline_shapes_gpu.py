"""GPU parity at the sizes BASELINE.json names: the workloads bench.py times, built from the
same generator (iq-tree_amd/synth.py baseline_workload), evaluated by the HIP path through the C ABI and by the
CPU oracle on identical inputs.  These sizes reach what the small parity cases cannot: several tiles per SIMD,
staged plans chosen by the engine itself (protein, codon), many LDS chunks, the lane-split / category-split
heuristics switched off by size.

Bars (north_star: lnL <= 1e-6 relative, integer counters bit-exact):
  lnL                                   <= 1e-9 relative
  root-side scale_num[nptn]             bit-exact (phylokernel.h:461-474), lh_scale_factor <= 1e-12 relative
  _pattern_lh                           <= 1e-11
  df, ddf on the root branch            <= 1e-9 of |ddf|-scale (phylokernel.h:583-651)
  every 8th computed vector             <= 1e-10 of the pattern's max, scale_num bit-exact
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LNL_RTOL = 1e-9

# name -> (workload, patterns (0 = the BASELINE count), tree kwargs, must_scale)
CASES = {
    "dna_50x100k_GTR+G4": ("dna", 0, None, False),
    "protein_100x50k_+G4": ("protein", 0, None, True),
    "codon64_50x20k": ("codon", 0, None, False),
    # the same codon shape on a tree whose vectors underflow (the survey's GY94 run scaled 109 patterns; the
    # random 64-state model on U(0.02,0.2) branches scales none)
    "codon64_50x20k_long_branches": ("codon", 0, dict(lo=0.3, hi=0.9), True),
    "dna_200x125k_shard_of_config4": ("dna4", 125000, None, True),
}


def build(pkg, synth, oracle, workload, patterns, tree_kw):
    T, P0, nst, ncat, seq_type = synth.BASELINE_SHAPES[workload]
    if tree_kw is None:
        nwk, pat, freq, model = synth.baseline_workload(workload, patterns=patterns)
    else:
        model, _ = synth.baseline_model(workload)
        nwk = synth.random_tree_newick(T, 1, **tree_kw)
        P = patterns or P0
        st = synth.simulate_alignment(nwk, model, int(P * 1.05) + 64, 1000)
        pat, freq = synth.compress_patterns(st)
        assert pat.shape[1] >= P
        pat, freq = np.ascontiguousarray(pat[:, :P]), freq[:P].copy()
    ot = oracle.OracleTree(nwk, nst, seq_type, pat, freq, None, model)
    oracle.lib().oracle_set_threads(16)
    t = pkg.PhyloTree(nwk)
    t.set_alignment(nst, seq_type, pat, freq)
    t.set_model(model)
    t.set_likelihood_kernel(pkg.LK_EIGEN_HIP)
    t.attach_engine(0)
    return t, ot


def sampled_vectors(t, ot, every=8):
    """vector + counters + lh_scale_factor of every `every`-th computed neighbour."""
    edges = [(a, b) for a in range(t.num_nodes) for b, _ in t.neighbors(a)
             if not ot.is_leaf(b) and (t.neighbor_info(a, b)["computed"] & 1) and t.neighbor_info(a, b)["key"]]
    picked = edges[::every]
    for a, b in picked:
        plh, sc, sf = ot.partial(a, b)
        got = t.fetch_partial(a, b)
        scale = np.abs(plh).max(axis=1, keepdims=True)
        np.testing.assert_allclose(got / scale, plh / scale, rtol=0, atol=1e-10)
        assert np.array_equal(t.fetch_scale_num(a, b), sc), (a, b)
        info = t.neighbor_info(a, b)
        assert abs(info["lh_scale_factor"] - sf) <= 1e-12 * max(1.0, abs(sf))
    return len(picked), len(edges)


@pytest.mark.parametrize("name", list(CASES))
def test_baseline_shape_against_oracle(pkg, synth, oracle, name):
    workload, patterns, tree_kw, must_scale = CASES[name]
    t, ot = build(pkg, synth, oracle, workload, patterns, tree_kw)
    T = ot.ntaxa
    # hot loop 1 exactly as bench.py times it
    lnl = t.clear_and_compute_likelihood()
    ref, (a, b) = ot.likelihood()
    assert t.current_branch() == (a, b)
    assert abs(lnl - ref) <= LNL_RTOL * abs(ref), (lnl, ref)
    # the root-side vector's cumulative counters and scale factor
    frm, to = (a, b) if not ot.is_leaf(b) else (b, a)
    plh, sc, sf = ot.partial(frm, to)
    got_sc = t.fetch_scale_num(frm, to)
    assert int((got_sc != sc).sum()) == 0
    if must_scale:
        assert int(sc.sum()) > 0 and sf < 0.0
    info = t.neighbor_info(frm, to)
    assert abs(info["lh_scale_factor"] - sf) <= 1e-12 * max(1.0, abs(sf))
    _, oplh = ot.branch_lnl(a, b)
    np.testing.assert_allclose(t.fetch_pattern_lh(), oplh, rtol=1e-11, atol=1e-11)
    # a second evaluation (cached descriptors, same buffers) is bit-identical
    assert t.clear_and_compute_likelihood() == lnl
    # derivatives on the root branch
    df, ddf = t.compute_likelihood_derv(a, b)
    rdf, rddf = ot.derv(a, b)
    assert abs(ddf - rddf) <= 1e-9 * abs(rddf), (ddf, rddf)
    assert abs(df - rdf) <= 1e-9 * max(abs(rdf), 1e-3 * abs(rddf)), (df, rdf)
    v = t.compute_likelihood_from_buffer()
    assert abs(v - ref) <= LNL_RTOL * abs(ref)
    npicked, nedges = sampled_vectors(t, ot)
    assert nedges == T - 2 and npicked >= (T - 2) // 8
    # the pulley principle on an internal branch far from the root (reverse vectors: re-oriented buffers)
    inner = [(x, y) for x in range(t.num_nodes) for y, _ in t.neighbors(x) if not ot.is_leaf(x) and not ot.is_leaf(y)]
    x, y = inner[len(inner) // 2]
    v = t.compute_likelihood_branch(x, y)
    assert abs(v - ref) <= LNL_RTOL * abs(ref)
