#!/usr/bin/env python3
"""Reads a $IQHIP_TRACE_FILE written by a -DIQHIP_WAVE_TRACE build (tools/build_alt.sh trace -DIQHIP_WAVE_TRACE) and
prints, per traced launch: launch length, SIMD occupancy over time (how long the chip is partly empty), waves per SIMD,
and for the waves that stamped their phases the cycles per phase of an (op, category) step.
usage: tools/wave_trace.py trace.txt [stamps_per_op]"""
import collections
import statistics as st
import sys

path = sys.argv[1]
launches = []
cur = None
for line in open(path):
    f = line.split()
    if f[0] == "L":
        cur = {"hdr": line.strip(), "waves": [], "stamps": {}}
        launches.append(cur)
    elif f[0] == "W":
        kind, vb, wave, xcc, hw, rt0, rt1, ct0, ct1, det, ns = (int(x) for x in f[1:])
        cur["waves"].append(dict(kind=kind, vb=vb, wave=wave, xcc=xcc & 15, simd=(hw >> 4) & 3, cu=(hw >> 8) & 15,
                                 sh=(hw >> 12) & 1, se=(hw >> 13) & 7, rt0=rt0, rt1=rt1, cyc=ct1 - ct0, det=det, ns=ns))
    elif f[0] == "S":
        cur["stamps"][int(f[1])] = [int(x) for x in f[2:]]

for L in launches:
    W = L["waves"]
    if not W:
        continue
    t0 = min(w["rt0"] for w in W)
    t1 = max(w["rt1"] for w in W)
    dur = t1 - t0
    print("=" * 100)
    print(L["hdr"], "| waves", len(W), "| length %.1f us" % (dur / 100.0))
    by_simd = collections.defaultdict(list)
    for w in W:
        by_simd[(w["xcc"], w["se"], w["sh"], w["cu"], w["simd"])].append(w)
    print("SIMDs used", len(by_simd), "| waves per SIMD: min %d median %d max %d" % (
        min(len(v) for v in by_simd.values()), st.median(len(v) for v in by_simd.values()), max(len(v) for v in by_simd.values())))
    # occupancy timeline: 20 slices
    nb = 20
    occ = [0.0] * nb       # wave-time per slice
    busy = [0.0] * nb      # SIMD-time with >= 1 wave
    for key, ws in by_simd.items():
        ev = []
        for w in ws:
            ev.append((w["rt0"] - t0, 1))
            ev.append((w["rt1"] - t0, -1))
        ev.sort()
        n = 0
        last = 0
        for t, d in ev:
            if n > 0 and t > last:
                a, b = last, t
                for s in range(int(a * nb / dur), min(nb - 1, int((b - 1) * nb / dur)) + 1):
                    lo, hi = max(a, s * dur / nb), min(b, (s + 1) * dur / nb)
                    if hi > lo:
                        occ[s] += n * (hi - lo)
                        busy[s] += hi - lo
            n += d
            last = t
    nsimd = 1024
    print("slice  waves/SIMD  SIMDs-with-work (of %d)" % nsimd)
    for s in range(nb):
        print("%4d%%   %6.2f      %6.1f%%" % (100 * (s + 1) // nb, occ[s] / (dur / nb) / nsimd, 100 * busy[s] / (dur / nb) / nsimd))
    tot_busy = sum(busy) / dur / nsimd
    print("average: %.2f waves/SIMD, %.1f%% of SIMD-time has work" % (sum(occ) / dur / nsimd, 100 * tot_busy))
    lens = sorted((w["rt1"] - w["rt0"]) / 100.0 for w in W)
    print("wave length us: min %.1f median %.1f p90 %.1f max %.1f" % (lens[0], lens[len(lens) // 2], lens[int(len(lens) * 0.9)], lens[-1]))
    kinds = collections.Counter(w["kind"] for w in W)
    print("kinds", dict(kinds))
    # phase stamps
    spo = int(sys.argv[2]) if len(sys.argv) > 2 else None
    dets = [w for w in W if w["det"] >= 0 and w["det"] in L["stamps"]]
    if dets and spo:
        agg = collections.defaultdict(list)
        for w in dets:
            s = L["stamps"][w["det"]]
            nops = len(s) // spo
            for o in range(nops):
                base = s[o * spo:(o + 1) * spo]
                nxt = s[(o + 1) * spo] if (o + 1) * spo < len(s) else None
                for i in range(spo - 1):
                    agg[i].append(base[i + 1] - base[i])
                if nxt is not None:
                    agg[spo - 1].append(nxt - base[-1])
                    agg["op"].append(nxt - base[0])
        print("phase cycles (shader clock) over %d stamped waves:" % len(dets))
        for k, v in agg.items():
            v.sort()
            print("  phase %-3s n %5d  median %7d  mean %8.0f  p10 %7d  p90 %7d" % (k, len(v), v[len(v) // 2], sum(v) / len(v), v[len(v) // 10], v[len(v) * 9 // 10]))
        w = dets[0]
        print("  (first stamped wave: %d stamps, %d cycles in all)" % (w["ns"], w["cyc"]))
