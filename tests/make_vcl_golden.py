"""Regenerate tests/golden/vcl_probe.json from oracle/_ref/vcl_probe (container only: the probe is
built by `make ref` from the reference's vectorclass headers where they lie)."""
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.check_output([os.path.join(ROOT, "oracle", "_ref", "vcl_probe")])
data = json.loads(out)
with open(os.path.join(ROOT, "tests", "golden", "vcl_probe.json"), "w") as f:
    json.dump(data, f)
print("dots", len(data["dots"]), "exp", len(data["exp"]), "log", len(data["log"]))
