"""CPU: the engine's plan builder + plan check on a planning-only engine (iqhip_debug_create_planner: no HIP call, fake
device addresses).  The traversal kernels request the inputs of op k+1 unconditionally while op k computes, so every
pointer of every descriptor -- look-ahead sentinels included -- must be a live allocation of the right kind whether the op
uses it or not.  Round 2 had a GPU memory fault from exactly that (null K2 table pointers of non-leaf children);
check_plan (engine.hip) finds that class of defect here, without a GPU, for every kernel family and staging variant."""
import ctypes as C
import os

import numpy as np
import pytest


def plan_of(pkg, synth, ntaxa, seed, nstates, multifurcating=False, mem_mode=0):
    """the op list the adapter submits for clearAllPartialLH(); computeLikelihood() (host mirror, dry run)"""
    nwk = synth.random_multifurcating_newick(ntaxa, seed) if multifurcating else synth.random_tree_newick(ntaxa, seed)
    t = pkg.PhyloTree(nwk)
    t.set_mem_mode(mem_mode)
    su = {4: 18, 20: 23, 64: 64}[nstates]
    t.set_alignment(nstates, {4: 0, 20: 1, 64: 2}[nstates], np.full((ntaxa, 8), su, dtype=np.uint8), np.ones(8))
    t.set_model(synth.gtr_model() if nstates == 4 else synth.random_reversible_model(nstates, 1, alpha=0.9, ncat=1))
    t.set_dry_run(True)
    t.clear_all_partial_lh()
    t.compute_likelihood()
    ops = (pkg.NodeOp * len(t.last_plan()))()
    for k, p in enumerate(t.last_plan()):
        ops[k] = pkg.NodeOp(p["dst_key"], p["left_key"], p["right_key"], p["left_leaf"], p["right_leaf"],
                            p["left_len"], p["right_len"], p["flags"], 0)
    assert not multifurcating or any(o.flags & 1 for o in ops)   # IQHIP_OP_NO_SCALE intermediates are in the plan
    return ops


def planner(pkg, nstates, ncat, nptn, ntaxa, cus=256, nclass=1):
    lib = pkg.libiqhip()
    e = C.c_void_p()
    su = {4: 18, 20: 23, 64: 64}[nstates]
    rc = lib.iqhip_debug_create_planner(C.byref(e), nstates, ncat, nptn, ntaxa, cus, su, nclass)
    assert rc == 0, lib.iqhip_last_error()
    return lib, e


SHAPES = [  # nstates, ncat, nptn, ntaxa, nclass: every kernel family and the size classes that switch variants
    (4, 4, 100000, 50, 1), (4, 4, 3000, 50, 1), (4, 1, 40000, 30, 1), (4, 3, 500, 9, 1), (4, 4, 20000, 24, 2),
    (20, 4, 50000, 100, 1), (20, 4, 400, 30, 1), (20, 1, 9000, 40, 1), (20, 6, 5000, 20, 1), (20, 8, 3000, 16, 2),
    (64, 1, 20000, 50, 1), (64, 1, 300, 20, 1), (64, 2, 4000, 12, 1), (64, 4, 2000, 12, 2),
]


@pytest.mark.parametrize("nstates,ncat,nptn,ntaxa,nclass", SHAPES)
@pytest.mark.parametrize("env", [{}, {"IQHIP_LEAF_TABLES": "1"}, {"IQHIP_LEAF_TABLES": "0", "IQHIP_HOLD": "0", "IQHIP_HOLD_LDS": "0"},
                                 {"IQHIP_SPLIT": "5", "IQHIP_LEVELS": "3"}, {"IQHIP_SPLIT": "0", "IQHIP_SMALL_PLANS": "0"}])
def test_plans_satisfy_the_kernel_contract(pkg, synth, monkeypatch, nstates, ncat, nptn, ntaxa, nclass, env):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    lib, e = planner(pkg, nstates, ncat, nptn, ntaxa, nclass=nclass)
    try:
        for seed, mf in ((1, False), (2, False), (3, True)):
            ops = plan_of(pkg, synth, ntaxa, 40 + seed, nstates, multifurcating=mf)
            rc = lib.iqhip_debug_plan(e, ops, len(ops))
            assert rc == 0, lib.iqhip_last_error()
            # a sweep-sized plan (one or two ops) of the same tree: the small-plan path and its sentinels
            rc = lib.iqhip_debug_plan(e, ops, min(2, len(ops)))
            assert rc == 0, lib.iqhip_last_error()
    finally:
        lib.iqhip_destroy(e)


def test_cherry_tables_are_planned_and_checked(pkg, synth, monkeypatch):
    """20 states x 4 categories from 8192 patterns on: ops with two leaf children get a slot of the cherry-table buffer
    (engine.hip build_plan, DevOp::cherry) -- in the unit stages only, one slot per pair of taxa, rebuilt only for new
    lengths -- and a pointer that is not a slot is refused."""
    lib = pkg.libiqhip()

    def counters(e):
        built, ops = C.c_int64(), C.c_int64()
        assert lib.iqhip_debug_cherry_tables(e, C.byref(built), C.byref(ops)) == 0
        return built.value, ops.value
    plan = plan_of(pkg, synth, 100, 41, 20)
    ncherry = sum(1 for o in plan if o.left_leaf >= 0 and o.right_leaf >= 0)
    assert ncherry >= 20
    lib_, e = planner(pkg, 20, 4, 50000, 100)
    try:
        assert lib.iqhip_debug_plan(e, plan, len(plan)) == 0, lib.iqhip_last_error()
        built, ops = counters(e)
        assert 0 < ops <= ncherry and built == ops          # (the top stage's kernel computes its cherries itself)
        assert lib.iqhip_debug_plan(e, plan, len(plan)) == 0
        assert counters(e)[0] == built                      # same plan again: the cached descriptors, nothing scheduled
        k = next(i for i, o in enumerate(plan) if o.left_leaf >= 0 and o.right_leaf >= 0)
        plan[k].left_len += 0.125                           # one pendant length changes: one table
        assert lib.iqhip_debug_plan(e, plan, len(plan)) == 0
        assert counters(e)[0] in (built, built + 1)         # (+0 if that cherry sits in the top stage)
    finally:
        lib.iqhip_destroy(e)
    for env in ({"IQHIP_CHERRY_TABLES": "0"}, {}):
        for k_, v in env.items():
            monkeypatch.setenv(k_, v)
        small = env == {}
        lib_, e = planner(pkg, 20, 4, 5000 if small else 50000, 100)   # switched off / below 8192 patterns: no tables
        try:
            assert lib.iqhip_debug_plan(e, plan, len(plan)) == 0
            assert counters(e) == (0, 0)
        finally:
            lib.iqhip_destroy(e)
        monkeypatch.delenv("IQHIP_CHERRY_TABLES", raising=False)
    monkeypatch.setenv("IQHIP_DEBUG_BREAK_PLAN", "cherry")
    lib_, e = planner(pkg, 20, 4, 50000, 100)
    try:
        rc = lib.iqhip_debug_plan(e, plan, len(plan))
        assert rc == 2 and "cherry" in lib.iqhip_last_error().decode(), lib.iqhip_last_error()
    finally:
        lib.iqhip_destroy(e)


@pytest.mark.parametrize("what,needle", [("tab", "K2 table"), ("sentinel", "pf"), ("states", "state matrix")])
def test_a_broken_descriptor_is_refused(pkg, synth, monkeypatch, what, needle):
    """the round-2 fault class: a descriptor pointer the op itself does not use is null"""
    monkeypatch.setenv("IQHIP_DEBUG_BREAK_PLAN", what)
    monkeypatch.setenv("IQHIP_LEAF_TABLES", "1")
    lib, e = planner(pkg, 64, 1, 20000, 20)
    try:
        ops = plan_of(pkg, synth, 20, 77, 64)
        rc = lib.iqhip_debug_plan(e, ops, len(ops))
        assert rc == 2 and needle in lib.iqhip_last_error().decode(), lib.iqhip_last_error()
    finally:
        lib.iqhip_destroy(e)


def test_a_planner_computes_nothing(pkg, synth):
    lib, e = planner(pkg, 4, 4, 1000, 10)
    try:
        ops = plan_of(pkg, synth, 10, 5, 4)
        ss = (C.c_double * len(ops))()
        assert lib.iqhip_update_partials(e, ops, len(ops), ss) != 0   # no device: compute entry points fail loudly
        assert lib.iqhip_synchronize(e) != 0
    finally:
        lib.iqhip_destroy(e)


def test_sweep_scripts_stop_on_failure():
    """tools/sweep_*.sh and the other GPU-box drivers end at the first failing step (set -e): a faulting kernel must
    not be followed by a dozen more runs on the same lease."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for name in sorted(os.listdir(os.path.join(root, "tools"))):
        if not name.endswith(".sh"):
            continue
        head = open(os.path.join(root, "tools", name)).read().split("\n")[:12]
        assert any(ln.strip().startswith("set -e") for ln in head), name
