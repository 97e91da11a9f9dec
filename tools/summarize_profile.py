#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output for profiles/: per-kernel stats of ONE steady-state step of bench.py
(delimited by consecutive k_label_hist launches) and PMC FETCH/WRITE averages per kernel.

    python tools/summarize_profile.py trace <kernel_trace.csv> <scales> [step_index]
    python tools/summarize_profile.py pmc <counter_collection.csv>
"""
import collections
import csv
import sys


def trace(path, scales=3, step=3):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_label_hist")]
    s, e = idx[step * scales], idx[(step + 1) * scales]
    sel = rows[s:e]
    t0, t1 = int(sel[0]["Start_Timestamp"]), int(rows[e]["Start_Timestamp"])
    agg = collections.defaultdict(lambda: [0, 0])
    for r in sel:
        a = agg[r["Kernel_Name"]]
        a[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a[1] += 1
    tot = sum(a[0] for a in agg.values())
    print(f"# one steady-state step (#{step}): wall {1e-6 * (t1 - t0):.3f} ms, kernel time {1e-6 * tot:.3f} ms, "
          f"{len(sel)} launches")
    print("kernel,calls,total_ms,avg_us,percent")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        print(f"\"{k[:140]}\",{a[1]},{a[0] / 1e6:.3f},{a[0] / a[1] / 1e3:.2f},{100 * a[0] / tot:.2f}")


def pmc(path):
    rows = list(csv.DictReader(open(path)))
    agg = collections.defaultdict(list)
    for r in rows:
        agg[(r["Kernel_Name"][:100], r["Counter_Name"])].append(float(r["Counter_Value"]))
    print("kernel,counter,dispatches,avg,max")
    for (k, c), v in sorted(agg.items()):
        print(f"\"{k}\",{c},{len(v)},{sum(v) / len(v):.1f},{max(v):.1f}")


if __name__ == "__main__":
    if sys.argv[1] == "trace":
        trace(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 3, int(sys.argv[4]) if len(sys.argv) > 4 else 3)
    else:
        pmc(sys.argv[2])
