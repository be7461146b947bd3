"""Per-step summary of a rocprofv3 kernel_stats csv + pack / colsum breakdown by grid size from the kernel trace.
usage: python tools/kstats.py <dir> <prefix> <steps-in-run>"""
import collections
import csv
import sys

d, pre, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
rows = list(csv.DictReader(open(f"{d}/{pre}_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"kernel time per step: {tot / 1e6 / n:.3f} ms")
for r in rows[:24]:
    print(f"{float(r['TotalDurationNs']) / 1e6 / n:8.3f} ms/step {int(r['Calls']) / n:6.1f} calls avg {float(r['AverageNs']) / 1e3:7.1f} us {float(r['Percentage']):5.1f}%  {r['Name'][:100]}")
tr = list(csv.DictReader(open(f"{d}/{pre}_kernel_trace.csv")))
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(tr) if "adamw" in r["Kernel_Name"]]
step = tr[idx[-3] + 1: idx[-2] + 1]
c = collections.OrderedDict()
for r in step:
    nm = r["Kernel_Name"]
    if "pack_bf16" in nm or "colsum_stage1" in nm:
        key = (nm.split("(")[0][-36:], r["Grid_Size_X"])
        dd = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        c.setdefault(key, [0, 0]); c[key][0] += 1; c[key][1] += dd
for k, v in sorted(c.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"{v[0]:4d} calls {v[1] / 1e3:8.1f} us total  avg {v[1] / v[0] / 1e3:6.1f}  {k}")
