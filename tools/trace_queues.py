"""Summarises a rocprofv3 kernel trace: kernels per (queue, stream), mean durations, and a slice of the timeline."""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1])[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = rows[-1400:-200] if len(rows) > 1600 else rows
t0 = int(sel[0]["Start_Timestamp"])
name = lambda r: r["Kernel_Name"].split("(")[0][-30:]
q = collections.Counter((name(r), r.get("Queue_Id"), r.get("Stream_Id")) for r in sel)
for k, v in q.items():
    print(k, v)
d = collections.defaultdict(list)
for r in sel:
    d[name(r)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in d.items():
    print(k, len(v), round(sum(v) / len(v)))
for r in sel[600:648]:
    print(name(r)[-24:], r.get("Queue_Id"), int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0)
