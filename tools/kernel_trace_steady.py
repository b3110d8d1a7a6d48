"""Launch durations of one kernel out of a rocprofv3 --kernel-trace CSV: raw mean (what --stats prints), median, and the steady-state mean -
without the launches that follow an idle gap of the queue.  Behind every idle gap (> 200 us without a kernel) the chip needs ~10 launches /
~30 ms to come back to its clock (tools/launch_timing_check.py); a 200-step bench run has three such gaps (import, warm-up, the profiler's own
pauses), and their slow launches are in the --stats average but not in bench.py's timed region.

    python tools/kernel_trace_steady.py <kernel_trace.csv> <kernel-name-substring> [out.json]
"""
import csv
import json
import statistics
import sys

IDLE_GAP_NS = 200_000
RAMP_LAUNCHES = 10


def kernel_trace_stats(path, substr, exclude="skinny"):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    durs, steady, since_gap, prev_end = [], [], RAMP_LAUNCHES, None
    for start, end, name in rows:
        if prev_end is None or start - prev_end > IDLE_GAP_NS:
            since_gap = 0
        prev_end = max(prev_end or end, end)
        if substr in name and (not exclude or exclude not in name):
            d = (end - start) / 1e3
            durs.append(d)
            if since_gap >= RAMP_LAUNCHES:
                steady.append(d)
            since_gap += 1
    if not durs:
        return None
    return {"kernel": substr, "launches": len(durs), "mean_us": statistics.fmean(durs), "median_us": statistics.median(durs), "min_us": min(durs),
            "max_us": max(durs), "stdev_us": statistics.pstdev(durs), "steady_launches": len(steady),
            "steady_mean_us": statistics.fmean(steady) if steady else None, "steady_median_us": statistics.median(steady) if steady else None,
            "steady_rule": f"a launch counts as steady once {RAMP_LAUNCHES} launches of this kernel have run since the queue's last idle gap of > {IDLE_GAP_NS // 1000} us"}


if __name__ == "__main__":
    res = kernel_trace_stats(sys.argv[1], sys.argv[2])
    print(json.dumps(res, indent=1))
    if len(sys.argv) > 3 and res is not None:
        json.dump(res, open(sys.argv[3], "w"), indent=1)
