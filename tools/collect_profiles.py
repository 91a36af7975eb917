"""Copy the judged summaries of a tools/profile_round.sh run (gpurun_out/prof_<tag>) into profiles/<round>/ and
refresh profiles/traffic.json.  usage: python tools/collect_profiles.py <gpurun_out/prof_tag> <profiles/rNN>"""
import csv
import glob
import json
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
summary = json.load(open(os.path.join(src, "summary.json")))
shapes = {"dna": (50, 100000, 4, 4), "protein": (100, 50000, 20, 4), "codon": (50, 20000, 64, 1)}
traffic = {}
for w in ("dna", "protein", "codon"):
    for f in glob.glob(os.path.join(src, "trace_" + w, "*", "*_kernel_stats.csv")):
        if any("k_traverse" in r["Name"] for r in csv.DictReader(open(f))):
            shutil.copy(f, os.path.join(dst, w + "_kernel_stats.csv"))
    for name, key in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        for f in glob.glob(os.path.join(src, name + "_" + w, "*", "*_counter_collection.csv")):
            rows = [r for r in csv.DictReader(open(f)) if "k_traverse" in r["Kernel_Name"]]
            if rows:
                with open(os.path.join(dst, "%s_pmc_%s.csv" % (w, key)), "w", newline="") as out:
                    wr = csv.DictWriter(out, fieldnames=list(rows[0].keys()))
                    wr.writeheader()
                    wr.writerows(rows)
    b = os.path.join(src, "bench_%s.json" % w)
    if os.path.exists(b):
        shutil.copy(b, os.path.join(dst, "bench_%s.json" % w))
    e = summary.get(w, {})
    if "hbm_traffic_bytes_per_launch" in e:
        T, P, n, c = shapes[w]
        traffic[w] = {"ntaxa": T, "patterns_per_gpu": P, "nstates": n, "ncat": c, "kernel": e.get("kernel"),
                      "launches_per_traversal": e.get("bench", {}).get("roofline", {}).get("launches_per_traversal"),
                      "hbm_traffic_bytes_per_launch": e["hbm_traffic_bytes_per_launch"],
                      "fetch_bytes_per_launch": e["fetch_bytes_per_launch"],
                      "write_bytes_per_launch": e["write_bytes_per_launch"],
                      "source": "%s/%s_pmc_FETCH_SIZE.csv (x2 gfx950 correction) + %s/%s_pmc_WRITE_SIZE.csv, rocprofv3 --pmc, "
                                "separate passes, averaged over the launches of the traversal kernel" % (dst, w, dst, w)}
shutil.copy(os.path.join(src, "summary.json"), os.path.join(dst, "summary.json"))
json.dump(traffic, open(os.path.join(os.path.dirname(dst.rstrip("/")), "traffic.json"), "w"), indent=1)
print("copied", sorted(os.listdir(dst)))
