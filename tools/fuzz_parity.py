"""Randomised differential run: random shapes / models / trees / memory modes through the HIP path against the
oracle (lnL, derivatives, every vector and scale counter).  usage: [FUZZ_BIG=1] python tools/fuzz_parity.py [ncases] [seed]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
pkg = g.load_package(); synth = importlib.import_module("iqtree_amd.synth"); od = g.load_oracle()
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
bad = 0
for case in range(ncases):
    n = int(rng.choice([2, 4, 4, 20, 20, 64]))
    seq_type = {2: 3, 4: 0, 20: 1, 64: 2}[n]
    mixture = n == 20 and rng.random() < 0.25
    ncat = int(rng.choice([1, 2, 3, 4, 4, 5, 6, 8])) if n <= 4 else (int(rng.choice([1, 2, 4, 4, 5])) if n == 20 else 1)
    ntaxa = int(rng.integers(4, 60))
    nptn = int(rng.choice([1, 17, 64, 200, 700, 3000]))
    if os.environ.get("FUZZ_BIG"):   # staged plans of the matrix-core kernels: units, parked results, split top stages
        n = int(rng.choice([20, 20, 64]))
        seq_type = {20: 1, 64: 2}[n]
        mixture = False
        ncat = int(rng.choice([4, 4, 1])) if n == 20 else 1
        ntaxa = int(rng.integers(14, 70))
        nptn = int(rng.choice([5000, 9000, 17000, 21000, 25500, 27000]))   # (>= 8192: cherry tables; >= 24576: mixed-role top stage)
    deep = rng.random() < 0.2
    pinv = float(rng.choice([0.0, 0.0, 0.15])) if not mixture else 0.0
    seed = int(rng.integers(1, 10 ** 6))
    for k in ("IQHIP_SPLIT", "IQHIP_LANE_SPLIT", "IQHIP_CAT_SPLIT", "IQHIP_ROW_SPLIT", "IQHIP_LEAF_TABLES", "IQHIP_CHERRY_TABLES",
              "IQHIP_MIXED_TOP", "IQHIP_TOP_CS2"):
        os.environ.pop(k, None)
    if rng.random() < 0.2: os.environ["IQHIP_CHERRY_TABLES"] = "0"
    if rng.random() < 0.2: os.environ["IQHIP_MIXED_TOP"] = "0"
    if rng.random() < 0.15: os.environ["IQHIP_TOP_CS2"] = str(int(rng.choice([0, 1])))
    if rng.random() < 0.4: os.environ["IQHIP_SPLIT"] = str(int(rng.choice([0, 2, 3, 5, 9])))
    if rng.random() < 0.3: os.environ["IQHIP_LANE_SPLIT"] = str(int(rng.choice([1, 2])))
    if rng.random() < 0.3: os.environ["IQHIP_CAT_SPLIT"] = str(int(rng.choice([0, 1])))
    if rng.random() < 0.3: os.environ["IQHIP_ROW_SPLIT"] = str(int(rng.choice([0, 1])))
    if rng.random() < 0.3: os.environ["IQHIP_LEAF_TABLES"] = str(int(rng.choice([0, 1])))
    if mixture:
        fused = rng.random() < 0.4
        model = synth.mixture_model(20, int(rng.integers(2, 5)), seed, ncat=1 if fused else int(rng.choice([1, 2, 4])), fused=fused)
        sim = model.classes[0]
    elif n == 4:
        model = sim = synth.gtr_model(alpha=0.7, ncat=ncat, pinvar=pinv)
    else:
        model = sim = synth.random_reversible_model(n, seed, alpha=0.9 if ncat > 1 else None, ncat=ncat, pinvar=pinv)
    su = od.state_unknown_for(n, seq_type)
    nwk = synth.random_tree_newick(ntaxa, seed, 0.35 if deep else 0.01, 0.9 if deep else 0.3, deep)
    st = synth.simulate_alignment(nwk, sim, nptn, seed + 1, float(rng.choice([0.0, 0.05])), su)
    pat, freq = synth.compress_patterns(st)
    invar = None if mixture else synth.ptn_invar_for(pat, model)
    mem = int(rng.choice([0, 1]))
    desc = "case %d: n=%d ncat=%d mix=%s taxa=%d ptn=%d deep=%s pinv=%.2f mem=%d env=%s" % (
        case, n, model.ncat, mixture, ntaxa, pat.shape[1], deep, pinv, mem,
        {k: os.environ[k] for k in os.environ if k.startswith("IQHIP_") and k != "IQHIP_LIB_DIR"})
    try:
        ot = od.OracleTree(nwk, n, seq_type, pat, freq, invar, model)
        t = pkg.PhyloTree(nwk); t.set_mem_mode(mem); t.set_alignment(n, seq_type, pat, freq, invar); t.set_model(model); t.attach_engine(0)
        lnl = t.compute_likelihood(); ref, (a, b) = ot.likelihood()
        assert abs(lnl - ref) <= 1e-9 * abs(ref), ("lnl", lnl, ref)
        df, ddf = t.compute_likelihood_derv(a, b); odf, oddf = ot.derv(a, b)
        assert abs(df - odf) <= 1e-7 * max(1.0, abs(odf)) and abs(ddf - oddf) <= 1e-7 * max(1.0, abs(oddf)), ("derv", df, odf, ddf, oddf)
        for x in range(t.num_nodes):
            for y, _ in t.neighbors(x):
                info = t.neighbor_info(x, y)
                if ot.is_leaf(y) or not (info["computed"] & 1) or info["key"] == 0: continue
                plh, sc, sf = ot.partial(x, y)
                got = t.fetch_partial(x, y); scale = np.abs(plh).max(axis=1, keepdims=True); scale[scale == 0] = 1.0
                assert np.max(np.abs(got / scale - plh / scale)) <= 1e-9, ("vector", x, y)
                assert np.array_equal(t.fetch_scale_num(x, y), sc), ("scale_num", x, y)
                assert abs(info["lh_scale_factor"] - sf) <= 1e-9 * max(1.0, abs(sf)), ("sf", x, y)
        inner = [(x, y) for x in range(t.num_nodes) for y, _ in t.neighbors(x) if not ot.is_leaf(x) and not ot.is_leaf(y)]
        if inner:
            x, y = inner[int(rng.integers(len(inner)))]
            v = t.compute_likelihood_branch(x, y)
            assert abs(v - ref) <= 1e-9 * abs(ref), ("pulley", v, ref)
            t.optimize_one_branch(x, y)
            ot.set_length(x, y, t.neighbor_info(x, y)["length"])
            r2, _ = ot.likelihood()
            assert abs(t.compute_likelihood() - r2) <= 1e-8 * abs(r2), ("after optimise", r2)
        if mem == 1 and ntaxa >= 5 and rng.random() < 0.7:   # batched NNI evaluation against the branch-by-branch evaluator
            nni5 = rng.random() < 0.5
            batch = t.evaluate_nnis5_batch() if nni5 else t.evaluate_nnis_batch()
            k = 2 * int(rng.integers(len(batch) // 2))
            seqm = t.nni_for_branch(batch[k]["node1"], batch[k]["node2"], nni5=nni5)
            for c in range(2):
                assert abs(batch[k + c]["newloglh"] - seqm[c][0]) <= 1e-9 * abs(seqm[c][0]), ("nni", nni5, k + c)
        t.close()
        print("ok  ", desc)
    except Exception as e:  # noqa
        bad += 1
        print("FAIL", desc, repr(e)[:300])
print("failures:", bad)
sys.exit(1 if bad else 0)
