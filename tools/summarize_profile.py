#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output for profiles/: per-kernel stats of ONE steady-state step of bench.py
(delimited by the metrics kernel that closes a step on the main stream; runs without the metrics tail: by the
label-stage launches) and PMC FETCH/WRITE averages per kernel.

    python tools/summarize_profile.py trace <kernel_trace.csv> <label-stage launches per step | steps=N> [step_index]
    python tools/summarize_profile.py pmc <counter_collection.csv>
    python tools/summarize_profile.py bygrid <kernel_trace.csv>      per (kernel, grid, workgroup): calls, avg / min us --
                                                                     one row per SHAPE a kernel template was launched on
"""
import collections
import csv
import os
import sys


def trace(path, scales=3, step=3):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_label_hist")]
    if isinstance(scales, str) and scales.startswith("steps="):
        # label-stage launches per step = all of them / the number of steps the traced command ran (warm-up included)
        scales = len(idx) // int(scales[6:])
    cm = [i for i, r in enumerate(rows) if "k_confusion_pred" in r["Kernel_Name"]]
    if len(cm) > step + 1:
        # the metrics tail closes a step on the MAIN stream (the label stage runs ahead on its own stream, so windows
        # between label-stage launches follow the host's timeline, not the GPU's): step = (tail of step - 1, tail of step]
        s, e = cm[step - 1] + 1, cm[step] + 1
    else:
        s, e = idx[step * scales], idx[(step + 1) * scales]
    sel = rows[s:e]
    t0, t1 = int(sel[0]["Start_Timestamp"]), int(rows[e]["Start_Timestamp"])
    agg = collections.defaultdict(lambda: [0, 0])
    for r in sel:
        a = agg[r["Kernel_Name"]]
        a[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a[1] += 1
    tot = sum(a[0] for a in agg.values())
    # time with at least one kernel running (union of the kernel intervals) and concurrency histogram
    ev = sorted([(int(r["Start_Timestamp"]), 1) for r in sel] + [(int(r["End_Timestamp"]), -1) for r in sel])
    busy, depth, last, by_depth = 0, 0, t0, collections.defaultdict(int)
    for t, d in ev:
        if depth > 0:
            busy += t - last
        by_depth[min(depth, 4)] += t - last
        depth += d
        last = t
    conc = ", ".join(f"{k if k < 4 else '>=4'} running: {1e-6 * v:.1f} ms" for k, v in sorted(by_depth.items()))
    print(f"# one steady-state step (#{step}): wall {1e-6 * (t1 - t0):.3f} ms, kernel time {1e-6 * tot:.3f} ms, "
          f"{len(sel)} launches; some kernel running for {1e-6 * busy:.3f} ms ({conc})")
    # idle gaps (no kernel running): histogram and the longest ones with the kernels around them
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]) for r in sel)
    gaps, end, prev = [], iv[0][1], iv[0][2]
    for a, b, name in iv[1:]:
        if a > end:
            gaps.append((a - end, prev, name))
        if b > end:
            end, prev = b, name
    hist = collections.Counter(min(int(g[0] / 1e3) // 5 * 5, 50) for g in gaps)
    print("# idle gaps by length (us): " + ", ".join(f"{k}{'+' if k == 50 else f'-{k + 5}'}: {v} ({1e-6 * sum(g[0] for g in gaps if min(int(g[0] / 1e3) // 5 * 5, 50) == k):.2f} ms)"
                                                    for k, v in sorted(hist.items())))
    for g in sorted(gaps, reverse=True)[:8]:
        print(f"#   gap {g[0] / 1e3:.0f} us after {g[1]} before {g[2]}")
    if os.environ.get("DCL_TRACE_WINDOW"):
        # rows around the longest gap: start (us from the step start), duration, queue, kernel
        big = max(gaps)
        rowsel = sorted(sel, key=lambda r: int(r["Start_Timestamp"]))
        idx = next(i for i, r in enumerate(rowsel) if r["Kernel_Name"][:60] == big[2] and
                   any(int(r["Start_Timestamp"]) - e == big[0] for e in [int(x["End_Timestamp"]) for x in rowsel[max(0, i - 40):i]]))
        for r in rowsel[max(0, idx - 30):idx + 25]:
            print(f"#   t={1e-3 * (int(r['Start_Timestamp']) - t0):9.1f} us dur {1e-3 * (int(r['End_Timestamp']) - int(r['Start_Timestamp'])):8.1f} "
                  f"q={r.get('Queue_Id', '?')} {r['Kernel_Name'][:70]}")
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


def bygrid(path):
    """rocprofv3's own --stats table aggregates a kernel template over every shape it ran on; the dispatch rows of the
    kernel trace carry the grid, which identifies the shape (the head convolution 12 x 720 x 128 x 256 is the only
    k_conv3x3<3,4,1> launch with 6 144 workgroups)."""
    rows = list(csv.DictReader(open(path)))
    agg = collections.defaultdict(list)
    for r in rows:
        grid = "x".join(r.get(f"Grid_Size_{a}", r.get(f"Grid_Size{a}", "?")) for a in "XYZ")
        wg = "x".join(r.get(f"Workgroup_Size_{a}", r.get(f"Workgroup_Size{a}", "?")) for a in "XYZ")
        agg[(r["Kernel_Name"][:110], grid, wg)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print("kernel,grid_threads,workgroup,calls,avg_us,min_us,max_us")
    for (k, g, w), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print(f"\"{k}\",{g},{w},{len(v)},{sum(v) / len(v) / 1e3:.2f},{min(v) / 1e3:.2f},{max(v) / 1e3:.2f}")


if __name__ == "__main__":
    if sys.argv[1] == "bygrid":
        bygrid(sys.argv[2])
    elif sys.argv[1] == "trace":
        sc = sys.argv[3] if len(sys.argv) > 3 else "3"
        trace(sys.argv[2], sc if sc.startswith("steps=") else int(sc), int(sys.argv[4]) if len(sys.argv) > 4 else 3)
    else:
        pmc(sys.argv[2])
