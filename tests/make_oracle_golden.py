"""Regenerate tests/golden/oracle_cases.json: seeded inputs (iq-tree_amd/synth.py) -> values computed by
the CPU oracle (oracle/lh_oracle.c).  The reference itself cannot be run here, so these
are ORACLE goldens: they freeze the oracle's numbers so that a change of the oracle, of the input
generators or of the HIP path shows up as a diff against committed data."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g  # noqa: E402

CASES = [
    dict(name="dna_gtr_g4", n=4, ncat=4, seq_type=0, ntaxa=12, nsites=800, seed=11, missing=0.05, pinvar=0.0),
    dict(name="dna_gtr_g4_inv", n=4, ncat=4, seq_type=0, ntaxa=9, nsites=600, seed=12, missing=0.0, pinvar=0.2),
    dict(name="dna_deep_scaled", n=4, ncat=4, seq_type=0, ntaxa=300, nsites=120, seed=13, missing=0.0, pinvar=0.0,
         lo=0.4, hi=0.9, caterpillar=True),
    dict(name="protein_g4", n=20, ncat=4, seq_type=1, ntaxa=10, nsites=300, seed=14, missing=0.03, pinvar=0.0),
    dict(name="protein_deep_scaled", n=20, ncat=4, seq_type=1, ntaxa=140, nsites=60, seed=15, missing=0.0,
         pinvar=0.0, lo=0.3, hi=0.7, caterpillar=True),
    dict(name="codon64", n=64, ncat=1, seq_type=2, ntaxa=9, nsites=200, seed=16, missing=0.02, pinvar=0.0),
    # mixtures: ncat = number of (class, rate) components
    dict(name="protein_mix3_g4", n=20, ncat=12, seq_type=1, ntaxa=11, nsites=250, seed=17, missing=0.03, pinvar=0.0,
         mixture=dict(nclass=3, ncat=4, fused=False)),
    dict(name="protein_mix4_fused_deep", n=20, ncat=4, seq_type=1, ntaxa=120, nsites=60, seed=18, missing=0.0,
         pinvar=0.0, lo=0.3, hi=0.7, caterpillar=True, mixture=dict(nclass=4, ncat=1, fused=True)),
]


def build_case(c, synth, od):
    sim = None
    if c.get("mixture"):
        mx = c["mixture"]
        model = synth.mixture_model(c["n"], mx["nclass"], c["seed"], ncat=mx["ncat"], fused=mx["fused"])
        sim = model.classes[0]
    elif c["n"] == 4:
        model = synth.gtr_model(alpha=0.9, ncat=c["ncat"], pinvar=c["pinvar"])
    else:
        model = synth.random_reversible_model(c["n"], c["seed"], alpha=0.9 if c["ncat"] > 1 else None,
                                              ncat=c["ncat"], pinvar=c["pinvar"])
    su = od.state_unknown_for(c["n"], c["seq_type"])
    nwk = synth.random_tree_newick(c["ntaxa"], c["seed"], c.get("lo", 0.02), c.get("hi", 0.2),
                                   c.get("caterpillar", False))
    st = synth.simulate_alignment(nwk, sim or model, c["nsites"], c["seed"] + 1, c["missing"], su)
    pat, freq = synth.compress_patterns(st)
    invar = synth.ptn_invar_for(pat, model)
    return model, nwk, pat, freq, invar


def input_digest(model, nwk, pat, freq):
    h = hashlib.sha256()
    h.update(nwk.encode())
    h.update(pat.tobytes())
    h.update(freq.tobytes())
    h.update(np.round(model.eval, 9).tobytes())
    return h.hexdigest()[:16]


def evaluate(c, synth, od):
    model, nwk, pat, freq, invar = build_case(c, synth, od)
    ot = od.OracleTree(nwk, c["n"], c["seq_type"], pat, freq, invar, model)
    lnl, (a, b) = ot.likelihood()
    df, ddf = ot.derv(a, b)
    _, sc, sf = ot.partial(a, b)
    return dict(name=c["name"], inputs=input_digest(model, nwk, pat, freq), npatterns=int(pat.shape[1]),
                branch=[int(a), int(b)], lnl=lnl, df=df, ddf=ddf, lh_scale_factor=sf,
                sum_scale_num=int(sc.sum()), max_scale_num=int(sc.max()))


if __name__ == "__main__":
    import importlib
    g.load_package()
    synth = importlib.import_module("iqtree_amd.synth")
    od = g.load_oracle()
    out = [evaluate(c, synth, od) for c in CASES]
    with open(os.path.join(ROOT, "tests", "golden", "oracle_cases.json"), "w") as f:
        json.dump(out, f, indent=1)
    for o in out:
        print(o)
