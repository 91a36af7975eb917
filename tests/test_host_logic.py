"""CPU: the host mirror's control logic in dry-run mode (plans are recorded, nothing is
computed): lazy flags, post-order plans, LM_PER_NODE re-orientation, invalidation --
the semantics a replacement kernel must keep."""
import numpy as np
import pytest


def make(pkg, synth, ntaxa=12, seed=3, mem_mode=0):
    nwk = synth.random_tree_newick(ntaxa, seed)
    t = pkg.PhyloTree(nwk)
    t.set_mem_mode(mem_mode)
    t.set_alignment(4, 0, np.zeros((ntaxa, 8), dtype=np.uint8), np.ones(8))
    t.set_model(synth.gtr_model())
    t.set_dry_run(True)
    return t, nwk


def test_newick_round_trip_and_ids(pkg, synth, oracle):
    t, nwk = make(pkg, synth)
    assert t.num_leaves == 12 and t.num_nodes == 22 and t.root == 0
    ot = oracle.OracleTree(nwk, 4, 0, np.zeros((12, 8), dtype=np.uint8), np.ones(8), None, synth.gtr_model())
    for v in range(t.num_nodes):  # same node numbering and adjacency as the oracle's tree
        assert sorted(x for x, _ in t.neighbors(v)) == sorted(x for x, _ in ot.adj[v])
        for nb, ln in t.neighbors(v):
            assert ln == ot.length(v, nb)
    t2 = pkg.PhyloTree(t.tree_string())
    assert t2.num_nodes == t.num_nodes
    # rooted (bifurcating top) input is unrooted; named leaves map through the alignment order
    t3 = pkg.PhyloTree("((A:0.1,B:0.2):0.05,(C:0.1,D:0.1):0.07);", names=["A", "B", "C", "D"])
    assert t3.num_nodes == 6
    assert any(ln == pytest.approx(0.12) for v in range(6) for _, ln in t3.neighbors(v))  # 0.05+0.07 merged


@pytest.mark.parametrize("bad", ["((0:0.1,1:0.1):0.1,2:0.1", "(0:0.1,1:0.1);", "((0:0.1,0:0.1):0.1,1:0.1,2:0.2);",
                                 "((0:0.1,1:x):0.1,2:0.1,3:0.1);"])
def test_bad_newick_is_an_error(pkg, bad):
    with pytest.raises(pkg.HostError):
        pkg.PhyloTree(bad)


def test_full_traversal_plan_is_postorder_and_complete(pkg, synth):
    t, _ = make(pkg, synth, 20, 5)
    t.clear_all_partial_lh()
    t.compute_likelihood()
    plan = t.last_plan()
    assert len(plan) == 20 - 2                      # one update per internal node (LM_PER_NODE)
    assert t.num_partial_lh_computations == 2 * 20 - 2  # the reference's counter counts leaf visits too
    done = set()
    for p in plan:
        for side, leaf in (("left", "left_leaf"), ("right", "right_leaf")):
            if p[leaf] < 0:
                assert p[side + "_key"] in done     # children before parents
            else:
                assert p[leaf] == p[side]
        assert p["dst_key"] not in done
        done.add(p["dst_key"])
        assert not (p["left_leaf"] < 0 and p["right_leaf"] >= 0)  # the leaf, if any, is `left` (:116-121)
    a, b = t.current_branch()
    assert len(t.neighbors(a)) == 1                 # evaluated on a leaf branch (phylotree.cpp:1035)
    # lazy: a second evaluation submits nothing
    t.compute_likelihood()
    assert t.last_plan() == []


def test_per_node_reorientation_moves_the_buffer(pkg, synth):
    t, _ = make(pkg, synth, 10, 9)
    t.compute_likelihood()
    keys_before = {(a, b): t.neighbor_info(a, b)["key"] for a in range(t.num_nodes) for b, _ in t.neighbors(a)}
    assert sum(1 for k in keys_before.values() if k) == 10 - 2
    # evaluate on an internal branch far from the root: needs partials pointing the other way
    a = t.num_nodes - 1
    b = [x for x, _ in t.neighbors(a) if len(t.neighbors(x)) == 3][0]
    t.compute_likelihood_branch(a, b)
    keys_after = {(x, y): t.neighbor_info(x, y)["key"] for x in range(t.num_nodes) for y, _ in t.neighbors(x)}
    assert sum(1 for k in keys_after.values() if k) == 10 - 2       # still one buffer per internal node
    assert sorted(k for k in keys_after.values() if k) == sorted(k for k in keys_before.values() if k)
    moved = [e for e in keys_after if keys_after[e] != keys_before[e]]
    assert moved                                                     # some buffers changed direction
    # a neighbour that lost its buffer has its computed bit cleared (:135-137)
    for (x, y) in moved:
        if keys_after[(x, y)] == 0:
            assert t.neighbor_info(x, y)["computed"] & 1 == 0


def test_all_branch_mode_keeps_both_directions(pkg, synth):
    t, _ = make(pkg, synth, 10, 9, mem_mode=1)
    t.compute_likelihood()
    for a in range(t.num_nodes):
        for b, _ in t.neighbors(a):
            t.compute_likelihood_branch(a, b)
    nkeys = sum(1 for a in range(t.num_nodes) for b, _ in t.neighbors(a) if t.neighbor_info(a, b)["key"])
    assert nkeys == 3 * 10 - 6
    for a in range(t.num_nodes):
        for b, _ in t.neighbors(a):
            if len(t.neighbors(b)) == 3:
                assert t.neighbor_info(a, b)["computed"] & 1


def test_branch_change_invalidates_only_reverse_partials(pkg, synth):
    t, _ = make(pkg, synth, 14, 2, mem_mode=1)
    t.compute_likelihood()
    for a in range(t.num_nodes):
        for b, _ in t.neighbors(a):
            t.compute_likelihood_branch(a, b)
    a = t.num_leaves + 3
    b = [x for x, _ in t.neighbors(a) if len(t.neighbors(x)) == 3][0]
    t.set_branch_length(a, b, 0.77, clear_reverse=True)
    # vectors of the two subtrees hanging off the branch itself stay valid ...
    assert t.neighbor_info(a, b)["computed"] & 1 and t.neighbor_info(b, a)["computed"] & 1
    # ... everything that looks across the branch is stale
    t.compute_likelihood_branch(a, b)
    assert t.last_plan() == []
    stale = 0
    for x in range(t.num_nodes):
        for y, _ in t.neighbors(x):
            if len(t.neighbors(y)) == 3 and not (t.neighbor_info(x, y)["computed"] & 1):
                stale += 1
    assert stale > 0
    t.compute_likelihood_branch(0, t.neighbors(0)[0][0])
    assert 0 < len(t.last_plan()) <= 14 - 2


def test_tip_partial_lh_table(pkg, synth, oracle):
    t, _ = make(pkg, synth)
    tip = t.tip_partial_lh()
    m = synth.gtr_model()
    Ui = m.inv_evec.reshape(4, 4)
    np.testing.assert_array_equal(tip[2], Ui[:, 2])
    np.testing.assert_allclose(tip[18], Ui.sum(axis=1), rtol=1e-15)
    np.testing.assert_allclose(tip[4 + 1], Ui[:, 1], rtol=0, atol=0)     # state 5 = mask 2 = {C}
    np.testing.assert_allclose(tip[4 + 2], Ui[:, 0] + Ui[:, 1], rtol=1e-15)  # mask 3 = {A,C}
    ot = oracle.OracleTree("((0:0.1,1:0.1):0.1,2:0.1,3:0.1);", 4, 0, np.zeros((4, 2), dtype=np.uint8),
                           np.ones(2), None, m)
    np.testing.assert_array_equal(tip.reshape(-1), ot.tip)


def test_multifurcating_node_becomes_a_chain_of_binary_updates(pkg, synth):
    """degree > 3 (the reference's scalar kernel, phylotreesse.cpp:702-806): the product over all children is carried
    through intermediate vectors over zero-length branches that are never rescaled; only the node's own update tests
    for underflow."""
    t = pkg.PhyloTree("((0:0.1,1:0.2,2:0.3,5:0.35):0.4,3:0.1,4:0.1);")
    t.set_alignment(4, 0, np.zeros((6, 4), dtype=np.uint8), np.ones(4))
    t.set_model(synth.gtr_model())
    t.set_dry_run(True)
    t.compute_likelihood()
    plan = t.last_plan()
    multi = [p for p in plan if p["flags"] & 1]
    assert len(plan) == 4 and len(multi) == 2          # 2 internal nodes; the 4-child one takes 3 updates
    k = plan.index(multi[0])
    a, b, c = plan[k], plan[k + 1], plan[k + 2]
    assert (a["left_leaf"], a["right_leaf"]) == (0, 1) and (a["left_len"], a["right_len"]) == (0.1, 0.2)
    assert a["dst"][1] == -1 and b["dst"][1] == -1 and c["dst"][1] != -1      # only the last one answers a neighbour
    assert b["left_key"] == a["dst_key"] and b["left_len"] == 0.0 and b["right_leaf"] == 2 and b["flags"] == 1
    assert c["left_key"] == b["dst_key"] and c["left_len"] == 0.0 and c["right_leaf"] == 5 and c["flags"] == 2   # IQHIP_OP_SCALAR_RULE
    assert a["dst_key"] >> 63 == 1 and b["dst_key"] >> 63 == 1 and a["dst_key"] != b["dst_key"]
    assert c["dst_key"] >> 63 == 0
