#!/usr/bin/env python
"""Is the f16x3 contraction power-bound?  (VERDICT r02 weak #7.)  Samples the shader clock and the socket power at >= 10 Hz
while the engine runs a sustained batch-4096 workload in the f32 mode and then in the f16x3 mode, with idle gaps between.

A child process samples (so the sampler never competes with the launch loop for the GIL):
  * amdsmi (python binding under /opt/rocm/share/amd_smi): gpu_metrics -> current_gfxclk, current_gfxclks[8] (one per XCD),
    current_socket_power / average_socket_power, ppt_residency_acc / prochot_residency_acc / socket_thm_residency_acc (the firmware's
    own accumulators of time spent limited by the package-power tracker / PROCHOT / thermals), throttle status, temperatures;
  * sysfs hwmon (freq1_input, power1_average / power1_input) when present;
  * `rocm-smi --showclocks --showpower --json` as the last resort (slower than 10 Hz: the achieved rate is reported).
The parent stamps the phase boundaries with time.time(); every sample is labelled with its phase afterwards.

  python tools/clock_power_probe.py [--seconds 2.5] [--hz 20] [--out gpurun_out/r03_clock_power.json]
"""
import argparse
import glob
import json
import multiprocessing as mp
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _sampler(stop, hz, q):
    """Runs in a child process (no torch, no HIP context)."""
    out = {"backend": None, "errors": [], "samples": []}
    period = 1.0 / hz
    smi = None
    try:
        sys.path.insert(0, "/opt/rocm/share/amd_smi")
        import amdsmi

        amdsmi.amdsmi_init()
        hs = amdsmi.amdsmi_get_processor_handles()
        smi = (amdsmi, hs[0])
        out["backend"] = "amdsmi gpu_metrics"
        out["n_processors"] = len(hs)
    except Exception as e:  # noqa: BLE001
        out["errors"].append(f"amdsmi: {type(e).__name__}: {e}")
    hw = {}
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        for key in ("freq1_input", "power1_average", "power1_input"):
            f = os.path.join(d, key)
            if os.path.exists(f) and key not in hw:
                hw[key] = f
    out["hwmon_files"] = hw
    if smi is None and hw:
        out["backend"] = "sysfs hwmon"
    keep = ("current_gfxclk", "current_gfxclks", "average_gfxclk_frequency", "current_socket_power", "average_socket_power", "current_uclk",
            "ppt_residency_acc", "prochot_residency_acc", "socket_thm_residency_acc", "vr_thm_residency_acc", "hbm_thm_residency_acc",
            "accumulation_counter", "throttle_status", "indep_throttle_status", "temperature_hotspot", "average_gfx_activity",
            "energy_accumulator", "gfxclk_lock_status")
    while not stop.is_set():
        t = time.time()
        s = {"t": t}
        if smi is not None:
            try:
                m = smi[0].amdsmi_get_gpu_metrics_info(smi[1])
                for k in keep:
                    if k in m:
                        s[k] = m[k]
            except Exception as e:  # noqa: BLE001
                if len(out["errors"]) < 5:
                    out["errors"].append(f"gpu_metrics: {type(e).__name__}: {e}")
        for key, f in hw.items():
            try:
                s["hwmon_" + key] = int(open(f).read().strip())
            except Exception:  # noqa: BLE001
                pass
        if smi is None and not hw:
            try:
                r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=10)
                s["rocm_smi"] = json.loads(r.stdout).get("card0", {})
                out["backend"] = "rocm-smi CLI"
            except Exception as e:  # noqa: BLE001
                if len(out["errors"]) < 5:
                    out["errors"].append(f"rocm-smi: {type(e).__name__}: {e}")
        out["samples"].append(s)
        dt = period - (time.time() - t)
        if dt > 0:
            time.sleep(dt)
    q.put(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=2.5)
    ap.add_argument("--hz", type=float, default=20.0)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r03_clock_power.json"))
    args = ap.parse_args()

    ctx = mp.get_context("spawn")
    stop, q = ctx.Event(), ctx.Queue()
    proc = ctx.Process(target=_sampler, args=(stop, args.hz, q))
    proc.start()

    import numpy as np
    import torch

    from ikflow_amd.ikflow_solver import IKFlowSolver
    from ikflow_amd.model import MODEL_DESCRIPTIONS, hparams_for, layout_from, random_state_dict
    from ikflow_amd.robots import get_robot

    name = "panda__full__lp191_5.25m"
    dev = torch.device("cuda:0")
    robot = get_robot(MODEL_DESCRIPTIONS[name]["robot_name"])
    hp = hparams_for(name)
    lay = layout_from(hp, robot)
    s = IKFlowSolver(hp, robot)
    s.load_state_dict_tensors(random_state_dict(lay, robot, seed=0))
    B = args.batch
    q_ = torch.tensor(robot.sample_joint_angles(B, 0.004363323129985824, np.random.default_rng(0)), device=dev)
    poses = robot.forward_kinematics(q_)
    lat = torch.randn(B, lay.dim, generator=torch.Generator().manual_seed(1)).to(dev)
    eng = s.engine(dev)
    eng.reserve(B)
    s.set_precision("f16x3")  # builds the split images before anything is timed
    eng.set_split_guard(False)  # no per-call synchronisation: the launch queue stays full
    for _ in range(3):
        s.generate_ik_solutions(poses, latent=lat)
    s.set_precision("f32")
    for _ in range(3):
        s.generate_ik_solutions(poses, latent=lat)
    torch.cuda.synchronize()

    phases = []

    def phase(label, fn=None):
        t0 = time.time()
        n = 0
        if fn is None:
            time.sleep(1.0)
        else:
            t_end = time.perf_counter() + args.seconds
            while time.perf_counter() < t_end:
                for _ in range(8):
                    fn()
                n += 8
                torch.cuda.synchronize()  # keeps the queue bounded; 8 calls = 26 ms (f32) / 13 ms (f16x3) of GPU work per sync
            torch.cuda.synchronize()
        t1 = time.time()
        phases.append({"label": label, "t0": t0, "t1": t1, "calls": n, "solutions_per_s": (n * B / (t1 - t0)) if n else None})

    call = lambda: s.generate_ik_solutions(poses, latent=lat)
    phase("idle_before")
    phase("f32", call)
    phase("idle_between")
    s.set_precision("f16x3")
    phase("f16x3", call)
    s.set_precision("f32")
    phase("idle_after")
    stop.set()
    sam = q.get(timeout=60)
    proc.join(timeout=30)

    def label(t):
        for p in phases:
            if p["t0"] <= t <= p["t1"]:
                return p["label"]
        return "transition"

    for x in sam["samples"]:
        x["phase"] = label(x["t"])

    def stats(vals):
        vals = [v for v in vals if isinstance(v, (int, float))]
        if not vals:
            return None
        vals = sorted(vals)
        return {"n": len(vals), "min": vals[0], "median": vals[len(vals) // 2], "mean": sum(vals) / len(vals), "max": vals[-1]}

    summary = {}
    for p in phases:
        xs = [x for x in sam["samples"] if x["phase"] == p["label"]]
        # skip the first 0.3 s of a loaded phase (clock ramp) for the steady-state figures
        steady = [x for x in xs if x["t"] >= p["t0"] + (0.3 if p["calls"] else 0.0)]
        d = {"seconds": p["t1"] - p["t0"], "samples": len(xs), "sample_hz": len(xs) / max(p["t1"] - p["t0"], 1e-9), "calls": p["calls"],
             "solutions_per_s": p["solutions_per_s"]}
        d["gfxclk_mhz"] = stats([x.get("current_gfxclk") for x in steady])
        per_xcd = [x.get("current_gfxclks") for x in steady if isinstance(x.get("current_gfxclks"), list)]
        if per_xcd:
            flat = [v for row in per_xcd for v in row[:8] if isinstance(v, (int, float))]
            d["gfxclk_mhz_over_xcds"] = stats(flat)
        d["socket_power_w"] = stats([x.get("current_socket_power") for x in steady]) or stats([x.get("average_socket_power") for x in steady])
        d["hwmon_freq_mhz"] = stats([x["hwmon_freq1_input"] / 1e6 for x in steady if "hwmon_freq1_input" in x])
        pw = [x.get("hwmon_power1_average", x.get("hwmon_power1_input")) for x in steady]
        d["hwmon_power_w"] = stats([v / 1e6 for v in pw if v is not None])
        for acc in ("ppt_residency_acc", "prochot_residency_acc", "socket_thm_residency_acc", "vr_thm_residency_acc", "hbm_thm_residency_acc", "accumulation_counter"):
            vs = [x.get(acc) for x in xs if isinstance(x.get(acc), (int, float))]
            if len(vs) >= 2:
                d[acc + "_delta"] = vs[-1] - vs[0]
        if d.get("ppt_residency_acc_delta") is not None and d.get("accumulation_counter_delta"):
            d["ppt_limited_share"] = d["ppt_residency_acc_delta"] / d["accumulation_counter_delta"]
        summary[p["label"]] = d
    doc = {"what": f"shader clock and socket power sampled at ~{args.hz:g} Hz by a child process while the engine runs sustained B={B} approximate-IK calls "
                   f"({args.seconds:g} s per mode, f16x3 range guard off so no call synchronises), idle gaps between",
           "backend": sam["backend"], "errors": sam["errors"], "hwmon_files": sam.get("hwmon_files"), "summary": summary,
           "phases": phases, "samples": sam["samples"]}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps({"backend": sam["backend"], "errors": sam["errors"], "summary": summary}, indent=1))
    print("wrote", args.out)


if __name__ == "__main__":
    main()
