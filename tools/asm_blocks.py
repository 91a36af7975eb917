#!/usr/bin/env python3
"""Basic-block summary of one kernel in a hipcc -S listing: MFMAs, scratch (spill) traffic, VMEM, LDS, waits per block.
usage: tools/asm_blocks.py file.s <substring of the mangled kernel name>"""
import re
import sys

path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and key in l.split(":")[0] and l.rstrip().endswith(tuple("E:")) or (l.startswith("_ZN") and key in l and ": ;" in l))
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
blocks = []
cur = {"name": "entry", "n": 0, "mfma": 0, "sld": 0, "sst": 0, "gld": 0, "gst": 0, "ds": 0, "wait": 0, "valu": 0}
for l in lines[start + 1:end]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append(cur)
        cur = {"name": m.group(1), "n": 0, "mfma": 0, "sld": 0, "sst": 0, "gld": 0, "gst": 0, "ds": 0, "wait": 0, "valu": 0}
        continue
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        continue
    cur["n"] += 1
    op = t.split()[0]
    if op.startswith("v_mfma"): cur["mfma"] += 1
    elif op.startswith("scratch_load"): cur["sld"] += 1
    elif op.startswith("scratch_store"): cur["sst"] += 1
    elif op.startswith("global_load") or op.startswith("buffer_load"): cur["gld"] += 1
    elif op.startswith("global_store") or op.startswith("buffer_store"): cur["gst"] += 1
    elif op.startswith("ds_"): cur["ds"] += 1
    elif op.startswith("s_waitcnt"): cur["wait"] += 1
    elif op.startswith("v_"): cur["valu"] += 1
blocks.append(cur)
print("%-14s %6s %5s %5s %5s %5s %5s %5s %5s %5s" % ("block", "insts", "mfma", "valu", "ds", "gld", "gst", "spLD", "spST", "wait"))
for b in blocks:
    if b["n"] >= 8 or b["mfma"] or b["sld"] or b["sst"]:
        print("%-14s %6d %5d %5d %5d %5d %5d %5d %5d %5d" % (b["name"], b["n"], b["mfma"], b["valu"], b["ds"], b["gld"], b["gst"], b["sld"], b["sst"], b["wait"]))
