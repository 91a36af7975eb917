"""tests/golden/oracle_cases.json (made by tests/make_oracle_golden.py): the CPU test pins the oracle and
the input generators to the committed numbers; the GPU test pins the HIP path to the same numbers."""
import json
import os

import numpy as np
import pytest

import make_oracle_golden as mg

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "oracle_cases.json")))
BY_NAME = {g["name"]: g for g in GOLD}


@pytest.mark.parametrize("case", mg.CASES, ids=[c["name"] for c in mg.CASES])
def test_oracle_reproduces_committed_values(case, synth, oracle):
    got = mg.evaluate(case, synth, oracle)
    exp = BY_NAME[case["name"]]
    assert got["inputs"] == exp["inputs"], "seeded inputs changed"
    assert got["npatterns"] == exp["npatterns"] and got["branch"] == exp["branch"]
    assert abs(got["lnl"] - exp["lnl"]) <= 1e-12 * abs(exp["lnl"])
    assert abs(got["df"] - exp["df"]) <= 1e-9 * max(1.0, abs(exp["df"]))
    assert abs(got["ddf"] - exp["ddf"]) <= 1e-9 * abs(exp["ddf"])
    assert got["sum_scale_num"] == exp["sum_scale_num"] and got["max_scale_num"] == exp["max_scale_num"]


@pytest.mark.gpu
@pytest.mark.parametrize("case", mg.CASES, ids=[c["name"] for c in mg.CASES])
def test_hip_path_matches_committed_values(case, pkg, synth, oracle):
    exp = BY_NAME[case["name"]]
    model, nwk, pat, freq, invar = mg.build_case(case, synth, oracle)
    assert mg.input_digest(model, nwk, pat, freq) == exp["inputs"]
    t = pkg.PhyloTree(nwk)
    t.set_alignment(case["n"], case["seq_type"], pat, freq, invar)
    t.set_model(model)
    t.attach_engine(0)
    lnl = t.compute_likelihood()
    a, b = t.current_branch()
    assert [a, b] == exp["branch"]
    assert abs(lnl - exp["lnl"]) <= 1e-9 * abs(exp["lnl"])          # north_star tolerance: 1e-6
    df, ddf = t.compute_likelihood_derv(a, b)
    assert abs(df - exp["df"]) <= 1e-9 * max(1.0, abs(exp["df"])) + 1e-12 * abs(exp["ddf"])
    assert abs(ddf - exp["ddf"]) <= 1e-9 * abs(exp["ddf"])
    sc = t.fetch_scale_num(a, b)
    assert int(sc.sum()) == exp["sum_scale_num"] and int(sc.max()) == exp["max_scale_num"]  # integer counters: exact
    info = t.neighbor_info(a, b)
    assert abs(info["lh_scale_factor"] - exp["lh_scale_factor"]) <= 1e-12 * max(1.0, abs(exp["lh_scale_factor"]))
