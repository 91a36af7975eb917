#!/usr/bin/env python3
"""Per compute-unit slot: time between the end of a workgroup (its last wave) and the start of the next workgroup that
takes the slot, from a wave trace (tools/wave_trace.py format).  usage: tools/wave_gaps.py trace.txt [launch index] [slots per CU]"""
import collections
import sys
path = sys.argv[1]
want = int(sys.argv[2]) if len(sys.argv) > 2 else 0
slots = int(sys.argv[3]) if len(sys.argv) > 3 else 2
W = []
li = -1
for line in open(path):
    f = line.split()
    if f[0] == "L":
        li += 1
        continue
    if f[0] == "W" and li == want:
        kind, vb, wave, xcc, hw, rt0, rt1, ct0, ct1, det, ns = (int(x) for x in f[1:])
        W.append(dict(rt0=rt0, rt1=rt1, vb=vb, cu=(xcc & 15, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15)))
t0 = min(w["rt0"] for w in W)
wg = collections.defaultdict(list)
for w in W:
    wg[w["vb"]].append(w)
G = [dict(vb=vb, cu=ws[0]["cu"], s=min(w["rt0"] for w in ws) - t0, e=max(w["rt1"] for w in ws) - t0, e0=min(w["rt1"] for w in ws) - t0) for vb, ws in wg.items()]
bycu = collections.defaultdict(list)
for g in G:
    bycu[g["cu"]].append(g)
gaps = []
for cu, gs in bycu.items():
    gs.sort(key=lambda g: g["s"])
    running = []
    for g in gs:
        if len(running) < slots:
            running.append(g)
            continue
        ended = [r for r in running if r["e"] <= g["s"] + 5]
        if not ended:
            running.append(g)
            continue
        r = min(ended, key=lambda r: r["e"])
        gaps.append((g["s"] - r["e"]) / 100.0)
        running.remove(r)
        running.append(g)
gaps.sort()
spread = sorted((g["e"] - g["e0"]) / 100.0 for g in G)
print("workgroups %d on %d CUs; %d hand-overs" % (len(G), len(bycu), len(gaps)))
if gaps:
    print("gap between a workgroup's end and its successor's first stamp, us: p10 %.1f median %.1f p90 %.1f mean %.1f" % (
        gaps[len(gaps) // 10], gaps[len(gaps) // 2], gaps[len(gaps) * 9 // 10], sum(gaps) / len(gaps)))
print("spread between the first and the last wave of a workgroup ending, us: median %.1f p90 %.1f" % (spread[len(spread) // 2], spread[len(spread) * 9 // 10]))
