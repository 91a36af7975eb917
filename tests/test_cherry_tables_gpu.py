"""Cherry tables (include/iqhip.h iqhip_debug_cherry_tables; 20 states x 4 categories, >= 8192 patterns): a node whose two
children are leaves is answered from a table of its (STATE_UNKNOWN + 1)^2 possible vectors instead of three matrix products
per pattern.  The table is built by the same kernels on a pseudo-alignment of all state pairs, so every vector of the tree
equals the table-free engine's BIT FOR BIT (gaps and ambiguity codes included), the lnL equals the oracle's, and tables
follow pendant-length and model changes."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def counters(pkg, t):
    built, ops = C.c_int64(), C.c_int64()
    assert pkg.libiqhip().iqhip_debug_cherry_tables(t.engine, C.byref(built), C.byref(ops)) == 0
    return built.value, ops.value


def make(pkg, nwk, pat, freq, invar, model, mem_mode=0, n=20, seq_type=1):
    t = pkg.PhyloTree(nwk)
    t.set_mem_mode(mem_mode)
    t.set_alignment(n, seq_type, pat, freq, invar)
    t.set_model(model)
    t.attach_engine(0)
    return t


def vectors(t, ot):
    out = {}
    for a in range(t.num_nodes):
        for b, _ in t.neighbors(a):
            info = t.neighbor_info(a, b)
            if ot.is_leaf(b) or not (info["computed"] & 1) or info["key"] == 0:
                continue
            out[(a, b)] = (t.fetch_partial(a, b), t.fetch_scale_num(a, b))
    return out


@pytest.mark.parametrize("ntaxa,nsites,missing", [(24, 9500, 0.03), (60, 9800, 0.0)])
def test_cherry_tables_give_the_same_bits(pkg, synth, oracle, monkeypatch, ntaxa, nsites, missing):
    model = synth.random_reversible_model(20, 31, alpha=0.8, ncat=4)
    su = oracle.state_unknown_for(20, 1)
    nwk = synth.random_tree_newick(ntaxa, 32, 0.02, 0.25)
    st = synth.simulate_alignment(nwk, model, nsites, 33, missing, su)
    if missing:   # ambiguity codes (B, Z, J: states 20..22) as well as gaps
        rng = np.random.default_rng(5)
        mask = rng.random(st.shape) < 0.01
        st = np.where(mask, rng.integers(20, su, size=st.shape).astype(st.dtype), st)
    pat, freq = synth.compress_patterns(st)
    assert pat.shape[1] >= 8192
    invar = synth.ptn_invar_for(pat, model)
    ot = oracle.OracleTree(nwk, 20, 1, pat, freq, invar, model)
    ref, _ = ot.likelihood()

    monkeypatch.setenv("IQHIP_CHERRY_TABLES", "0")
    t0 = make(pkg, nwk, pat, freq, invar, model, pkg.LM_ALL_BRANCH)
    monkeypatch.delenv("IQHIP_CHERRY_TABLES")
    t1 = make(pkg, nwk, pat, freq, invar, model, pkg.LM_ALL_BRANCH)
    l0, l1 = t0.compute_likelihood(), t1.compute_likelihood()
    assert counters(pkg, t0) == (0, 0)
    built, ops = counters(pkg, t1)
    assert built >= 2 and ops >= built                  # a binary tree of >= 4 taxa has at least two cherries
    assert abs(l1 - ref) <= 1e-9 * abs(ref)
    assert l0 == l1
    v0, v1 = vectors(t0, ot), vectors(t1, ot)
    assert v0.keys() == v1.keys() and len(v0) >= ntaxa - 3
    for k in v0:
        assert np.array_equal(v0[k][0], v1[k][0]), k
        assert np.array_equal(v0[k][1], v1[k][1]), k

    # the same traversal again: nothing is rebuilt
    t1.clear_all_partial_lh()
    assert t1.compute_likelihood() == l1
    assert counters(pkg, t1)[0] == built

    # a pendant branch of a cherry changes: exactly its table is rebuilt
    # (leaf 0 is where the traversal starts: its neighbour is not a cherry of the rooted traversal)
    def sibling_leaves(a):
        return [c for c, _ in t1.neighbors(t1.neighbors(a)[0][0]) if c != a and ot.is_leaf(c) and c != 0]
    leaf = next(a for a in range(1, t1.num_leaves) if sibling_leaves(a))
    dad = t1.neighbors(leaf)[0][0]
    for t in (t0, t1):
        t.set_branch_length(leaf, dad, 0.37)
        t.clear_all_partial_lh()
    ot.set_length(leaf, dad, 0.37)
    ref2, _ = ot.likelihood()
    l0, l1 = t0.compute_likelihood(), t1.compute_likelihood()
    assert l0 == l1 and abs(l1 - ref2) <= 1e-9 * abs(ref2)
    assert counters(pkg, t1)[0] == built + 1

    # another model: every table the traversal uses is rebuilt
    model2 = synth.random_reversible_model(20, 77, alpha=1.3, ncat=4)
    for t in (t0, t1):
        t.set_model(model2)
        t.clear_all_partial_lh()
    ot2 = oracle.OracleTree(t1.tree_string(), 20, 1, pat, freq, synth.ptn_invar_for(pat, model2), model2)
    l0, l1 = t0.compute_likelihood(), t1.compute_likelihood()
    ref3, _ = ot2.likelihood()
    assert l0 == l1 and abs(l1 - ref3) <= 1e-9 * abs(ref3)
    assert counters(pkg, t1)[0] >= 2 * built

    # hot loop 2 on top of the tables: batched NNI candidates (independent segments, swapped subtrees make new cherries)
    # and a branch-length sweep (lengths that live on the device: those node updates are computed the ordinary way)
    n0, n1 = t0.evaluate_nnis_batch(), t1.evaluate_nnis_batch()
    assert [m["newloglh"] for m in n0] == [m["newloglh"] for m in n1]
    assert [m["new_len"] for m in n0] == [m["new_len"] for m in n1]
    for t in (t0, t1):
        t.set_device_newton(True)
        t.set_device_sweep(True)
    o0, o1 = t0.optimize_all_branches(iterations=1, tolerance=1e-4), t1.optimize_all_branches(iterations=1, tolerance=1e-4)
    assert o0 == o1 and t0.tree_string() == t1.tree_string()


def test_small_alignments_and_other_shapes_use_no_tables(pkg, synth, oracle):
    model = synth.random_reversible_model(20, 31, alpha=0.8, ncat=4)
    nwk = synth.random_tree_newick(12, 32, 0.02, 0.25)
    st = synth.simulate_alignment(nwk, model, 3000, 33)
    pat, freq = synth.compress_patterns(st)
    t = make(pkg, nwk, pat, freq, synth.ptn_invar_for(pat, model), model)
    t.compute_likelihood()
    assert counters(pkg, t) == (0, 0)
    # 64 states: measured slower with the extra block in that kernel (kernels_mfma.hip CHERRY), so no tables there
    model = synth.random_reversible_model(64, 41, alpha=None, ncat=1)
    st = synth.simulate_alignment(nwk, model, 18500, 43)
    pat, freq = synth.compress_patterns(st)
    t = make(pkg, nwk, pat, freq, synth.ptn_invar_for(pat, model), model, 0, 64, 2)
    t.compute_likelihood()
    assert counters(pkg, t) == (0, 0)


@pytest.mark.parametrize("ntaxa", [9, 18])
def test_mixed_role_top_stage_gives_the_same_bits(pkg, synth, oracle, monkeypatch, ntaxa):
    """20 states x 4 categories, top stage of a plan (kernels_mfma.hip k_traverse_mfma_top20): whole rounds of the chip as two
    waves per tile, the tiles beyond them as one wave per category -- 1563 tiles = one round of 1536 + 27.  Same vectors
    bit for bit as the single-role launch (IQHIP_MIXED_TOP=0), lnL equal to the oracle's; unstaged (9 taxa) and staged."""
    model = synth.random_reversible_model(20, 61, alpha=0.7, ncat=4)
    su = oracle.state_unknown_for(20, 1)
    nwk = synth.random_tree_newick(ntaxa, 62, 0.02, 0.3)
    pat = np.ascontiguousarray(synth.simulate_alignment(nwk, model, 25000, 63, 0.02, su))   # (sites as they are: 1563 tiles)
    freq = np.ones(25000)
    invar = synth.ptn_invar_for(pat, model)
    ot = oracle.OracleTree(nwk, 20, 1, pat, freq, invar, model)
    ref, _ = ot.likelihood()
    res = []
    for mixed in ("0", "1"):
        monkeypatch.setenv("IQHIP_MIXED_TOP", mixed)
        t = make(pkg, nwk, pat, freq, invar, model, pkg.LM_ALL_BRANCH)
        res.append((t.compute_likelihood(), vectors(t, ot)))
    monkeypatch.delenv("IQHIP_MIXED_TOP")
    (l0, v0), (l1, v1) = res
    assert l0 == l1 and abs(l1 - ref) <= 1e-9 * abs(ref)
    assert v0.keys() == v1.keys() and len(v0) >= ntaxa - 3
    for k in v0:
        assert np.array_equal(v0[k][0], v1[k][0]), k
        assert np.array_equal(v0[k][1], v1[k][1]), k
