"""Aggregate rocprofv3 --pmc counter_collection CSVs (one per pass) into profiles/rNN_pmc_summary.json.

usage: pmc_summarize.py OUT.json DOMINANT_KERNEL_SUBSTRING pass1_counter_collection.csv [pass2.csv ...]
Per kernel: mean counter value per dispatch.  FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports half the
bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section) -> hbm_read_bytes_corrected = 2 * FETCH_SIZE * 1024."""
import csv, json, sys, collections

out, dominant, files = sys.argv[1], sys.argv[2], sys.argv[3:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in files:
    per_dispatch = collections.defaultdict(float)  # (dispatch, kernel, counter) -> summed over XCD/instance rows
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        per_dispatch[(r["Dispatch_Id"], k, r["Counter_Name"])] += float(r["Counter_Value"])
    for (_, k, c), v in per_dispatch.items():
        acc[k][c].append(v)
kernels = {}
for k, cs in sorted(acc.items()):
    d = {"dispatches": max(len(v) for v in cs.values())}
    for c, v in sorted(cs.items()):
        d[c if c not in ("FETCH_SIZE", "WRITE_SIZE") else c + "_KiB"] = round(sum(v) / len(v), 1)
    if "FETCH_SIZE_KiB" in d:
        d["hbm_read_bytes_corrected"] = int(2 * d["FETCH_SIZE_KiB"] * 1024)
    if "WRITE_SIZE_KiB" in d:
        d["hbm_write_bytes"] = int(d["WRITE_SIZE_KiB"] * 1024)
    kernels[k] = d
dom = [k for k in kernels if dominant in k and "hbm_read_bytes_corrected" in kernels[k]]
traffic = None
if dom:
    # launches of the dominant kernel (all template instances), weighted by dispatch count
    n = sum(kernels[k]["dispatches"] for k in dom)
    traffic = int(sum((kernels[k]["hbm_read_bytes_corrected"] + kernels[k].get("hbm_write_bytes", 0)) * kernels[k]["dispatches"] for k in dom) / n)
json.dump({
    "command": "cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc <COUNTERS> --kernel-trace --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-split-extra --no-cells --no-live-pmc  (separate passes: FETCH_SIZE | WRITE_SIZE | SQ/GRBM set; see tools/profile_round.sh)",
    "notes": "FETCH_SIZE/WRITE_SIZE are KiB per dispatch; gfx950 correction: hbm_read_bytes = 2*FETCH_SIZE*1024 (MI355X_MICROARCH.md, HBM section). The read figure includes Infinity-Cache hits (each of the 8 XCD L2s fetches the whole 4.2 MB weight matrix). SQ counters are summed over the chip.",
    "dominant_kernel": dom,
    "dominant_kernel_traffic_bytes_per_launch": traffic,
    "kernels": kernels,
}, open(out, "w"), indent=1)
print("dominant", dom, "traffic/launch", traffic)
