"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel mean duration and the idle gap before each launch (probe)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2  # steady state: second half
rows = rows[skip:]
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].split("(")[0][:60]
    dur[n].append(e - s)
    if prev_end is not None: gap[n].append(s - prev_end)
    prev_end = e
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
busy = sum(sum(v) for v in dur.values())
print(f"kernels {len(rows)}  span {span/1e3:.1f} us  busy {busy/1e3:.1f} us ({100*busy/span:.1f}%)")
for n in sorted(dur, key=lambda k: -sum(dur[k])):
    d, g = dur[n], gap.get(n, [0])
    print(f"{n:60s} n={len(d):5d} dur {sum(d)/len(d)/1e3:7.2f} us  gap-before {sum(g)/max(1,len(g))/1e3:6.2f} us")
