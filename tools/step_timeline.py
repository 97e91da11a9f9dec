#!/usr/bin/env python3
"""Timeline of ONE steady-state step from a rocprofv3 kernel trace: the main queue's phases (time between named marker
kernels) and every gap of the main queue with what the other queues ran meanwhile.
   python tools/step_timeline.py <kernel_trace.csv> [step_index] [min_us]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
step = int(sys.argv[2]) if len(sys.argv) > 2 else 5
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 150.0
cm = [i for i, r in enumerate(rows) if "k_confusion_pred" in r["Kernel_Name"]]
sel = rows[cm[step - 1] + 1:cm[step] + 1]
t0 = int(sel[0]["Start_Timestamp"])


def short(r):
    return r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:40]


busy = {}
for r in sel:
    busy[r["Queue_Id"]] = busy.get(r["Queue_Id"], 0) + int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
main = max(busy, key=busy.get)
print(f"main queue {main}; step wall {(int(sel[-1]['End_Timestamp']) - t0) / 1e6:.2f} ms")
prev_end = None
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if r["Queue_Id"] != main:
        continue
    if prev_end is not None and s - prev_end > min_us * 1e3:
        others = {}
        for o in sel:
            if o["Queue_Id"] == main:
                continue
            a, b = max(int(o["Start_Timestamp"]), prev_end), min(int(o["End_Timestamp"]), s)
            if b > a:
                k = (o["Queue_Id"], short(o))
                others[k] = others.get(k, 0) + b - a
        top = sorted(others.items(), key=lambda kv: -kv[1])[:5]
        print(f"  t = {(prev_end - t0) / 1e6:7.2f} ms: main queue idle {(s - prev_end) / 1e3:7.0f} us before {short(r)}; meanwhile: "
              + ", ".join(f"q{q} {n} {v / 1e3:.0f}us" for (q, n), v in top))
    if (e - s) > 1.0e6:
        print(f"  t = {(s - t0) / 1e6:7.2f} ms: {short(r)} {(e - s) / 1e3:.0f} us")
    prev_end = e if prev_end is None else max(prev_end, e)
