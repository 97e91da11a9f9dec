#!/usr/bin/env python3
"""Per-HIP-queue busy time of ONE steady-state step from a rocprofv3 kernel trace: which stream (branch) is the
critical chain?   python tools/queue_busy.py <kernel_trace.csv> [step_index]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
step = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cm = [i for i, r in enumerate(rows) if "k_confusion_pred" in r["Kernel_Name"]]
sel = rows[cm[step - 1] + 1:cm[step] + 1]
t0, t1 = int(sel[0]["Start_Timestamp"]), int(sel[-1]["End_Timestamp"])
print(f"step wall {1e-6 * (t1 - t0):.2f} ms, {len(sel)} launches")
byq = collections.defaultdict(list)
for r in sel:
    byq[r.get("Queue_Id", "?")].append(r)
for q, rs in sorted(byq.items(), key=lambda kv: -sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in kv[1])):
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
    agg, cnt = collections.Counter(), collections.Counter()
    for r in rs:
        name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:44]
        agg[name] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        cnt[name] += 1
    print(f"queue {q}: {len(rs):5d} launches, busy {busy / 1e6:7.2f} ms ({100 * busy / (t1 - t0):4.1f} % of the step)")
    for k, v in agg.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 8):
        print(f"      {k:46s} {cnt[k]:4d} x  {v / 1e6:7.2f} ms")
