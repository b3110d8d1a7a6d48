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
    if "SQ_WAVE_CYCLES" in d and d.get("GRBM_GUI_ACTIVE", 0) > 0:
        # SQ_WAVE_CYCLES counts quad-cycles, summed over the chip; GRBM_GUI_ACTIVE counts cycles per XCD, summed over the 8 XCDs
        # (checked on k_flow_gemm: 4 x 74.3 M / 2048 waves = 145 k cycles per wave = the kernel's length): time-averaged resident
        # waves per SIMD over the launch, dispatch ramp and drain included
        d["occupancy_waves_per_simd"] = round(4.0 * d["SQ_WAVE_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 3)
        d["occupancy_frac_of_8_waves"] = round(d["occupancy_waves_per_simd"] / 8.0, 4)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d:  # cycles, summed over the 1024 SIMDs
            d["mfma_busy_frac_of_launch"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (d["GRBM_GUI_ACTIVE"] / 8.0), 4)
    kernels[k] = d
dom = [k for k in kernels if dominant in k and "hbm_read_bytes_corrected" in kernels[k]]
traffic = None
if dom:
    # launches of the dominant kernel (all template instances), weighted by dispatch count
    n = sum(kernels[k]["dispatches"] for k in dom)
    traffic = int(sum((kernels[k]["hbm_read_bytes_corrected"] + kernels[k].get("hbm_write_bytes", 0)) * kernels[k]["dispatches"] for k in dom) / n)
json.dump({
    "command": "cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc <COUNTERS> --kernel-trace --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-split-extra --no-cells --no-live-pmc  (separate passes: FETCH_SIZE | WRITE_SIZE | SQ/GRBM set; see tools/profile_round.sh)",
    "notes": "FETCH_SIZE/WRITE_SIZE are KiB per dispatch; gfx950 correction: hbm_read_bytes = 2*FETCH_SIZE*1024 (MI355X_MICROARCH.md, HBM section). The read figure is the L2s' fabric-side traffic and includes Infinity-Cache hits (each of the 8 non-coherent XCD L2s fetches every weight it uses: the row-owner launch streams all 203 MB through every L2). SQ counters are summed over the chip; occupancy_waves_per_simd = 4 x SQ_WAVE_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs).",
    "dominant_kernel": dom,
    "dominant_kernel_traffic_bytes_per_launch": traffic,
    "kernels": kernels,
}, open(out, "w"), indent=1)
print("dominant", dom, "traffic/launch", traffic)
