#!/usr/bin/env python3
"""Throttle / DVFS evidence: samples the SMU's limiter state while a workload runs.

    python tools/throttle_probe.py OUT.json -- <command ...>

A sampler thread reads, through the amdsmi python binding (the library behind `amd-smi metric -v`), at >= 10 Hz:
  * the violation status: PPT (package power), socket / VR / HBM thermal and PROCHOT — `active_*` flags, `per_*` percentages and the
    `acc_*` accumulators (their deltas over the run divided by the delta of `acc_counter` = fraction of SMU ticks spent in each violation),
    and the per-XCC "gfx clock below host limit" accumulators split by cause (power / thermal / total);
  * gpu_metrics: current_gfxclks of the eight XCCs, current / average socket power, the residency accumulators, throttle_status words,
    hotspot / HBM temperature, gfx activity;
  * the power cap.
The summary says what limiter, if any, the firmware reports while the workload runs, and at which clock and power.  Everything is wrapped:
a field the firmware does not expose is recorded as missing, not fatal.  No GPU work is done by this script itself.
"""
import json
import subprocess
import sys
import threading
import time


def _num(x):
    return x if isinstance(x, (int, float)) and not isinstance(x, bool) else None


def main():
    out_path = sys.argv[1]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    res = {"command": " ".join(cmd), "errors": []}
    try:
        import amdsmi
        amdsmi.amdsmi_init()
        h = amdsmi.amdsmi_get_processor_handles()[0]
    except Exception as e:  # noqa: BLE001
        res["errors"].append(f"amdsmi init: {e!r}")
        amdsmi, h = None, None

    def snap():
        s = {"t": time.time()}
        if amdsmi is None:
            return s
        try:
            v = amdsmi.amdsmi_get_violation_status(h)
            s["viol"] = {k: (list(x) if isinstance(x, (list, tuple)) else x) for k, x in v.items()}
        except Exception as e:  # noqa: BLE001
            s["viol_err"] = repr(e)[:200]
        try:
            m = amdsmi.amdsmi_get_gpu_metrics_info(h)
            keep = ("current_gfxclks", "current_gfxclk", "average_gfxclk_frequency", "current_socket_power", "average_socket_power", "throttle_status",
                    "indep_throttle_status", "ppt_residency_acc", "prochot_residency_acc", "socket_thm_residency_acc", "vr_thm_residency_acc",
                    "hbm_thm_residency_acc", "accumulation_counter", "temperature_hotspot", "temperature_hbm", "temperature_mem", "average_gfx_activity",
                    "gfxclk_lock_status", "current_uclk", "energy_accumulator", "firmware_timestamp")
            s["gm"] = {k: (list(m[k]) if isinstance(m.get(k), (list, tuple)) else m.get(k)) for k in keep if k in m}
        except Exception as e:  # noqa: BLE001
            s["gm_err"] = repr(e)[:200]
        return s

    if amdsmi is not None:
        for name, fn in (("power_cap", "amdsmi_get_power_cap_info"), ("power_info", "amdsmi_get_power_info")):
            try:
                res[name] = {k: v for k, v in getattr(amdsmi, fn)(h).items()}
            except Exception as e:  # noqa: BLE001
                res["errors"].append(f"{fn}: {e!r}"[:200])
        try:
            res["clock_gfx"] = amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX)
        except Exception as e:  # noqa: BLE001
            res["errors"].append(f"clock_info: {e!r}"[:200])

    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            samples.append(snap())
            time.sleep(0.04)

    first = snap()
    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    t0 = time.time()
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    t1 = time.time()
    stop.set()
    th.join(timeout=2)
    last = snap()
    res["workload_rc"] = p.returncode
    res["workload_tail"] = p.stdout[-3000:]
    res["wall_s"] = t1 - t0
    res["n_samples"] = len(samples)
    res["sample_hz"] = len(samples) / max(t1 - t0, 1e-9)

    def stat(a):
        a = sorted(x for x in a if x is not None)
        return {"n": len(a), "min": a[0], "p10": a[len(a) // 10], "p50": a[len(a) // 2], "p90": a[(len(a) * 9) // 10], "max": a[-1]} if a else {"n": 0}

    # "busy" = samples whose gfx activity is >= 90 % (where reported) — the workload's own start-up (imports, model build) is excluded that way
    def busy(s):
        a = s.get("gm", {}).get("average_gfx_activity")
        return _num(a) is not None and a >= 90

    bs = [s for s in samples if busy(s)] or samples
    res["n_busy_samples"] = len(bs)
    clk_all, clk_min, clk_max = [], [], []
    for s in bs:
        c = [x for x in (s.get("gm", {}).get("current_gfxclks") or []) if _num(x) is not None and 0 < x < 60000]
        if c:
            clk_all += c
            clk_min.append(min(c))
            clk_max.append(max(c))
    res["busy_gfxclk_mhz_all_xcc"] = stat(clk_all)
    res["busy_gfxclk_mhz_slowest_xcc"] = stat(clk_min)
    res["busy_gfxclk_mhz_fastest_xcc"] = stat(clk_max)
    res["busy_socket_power_w"] = stat([_num(s.get("gm", {}).get("current_socket_power")) for s in bs])
    res["busy_avg_socket_power_w"] = stat([_num(s.get("gm", {}).get("average_socket_power")) for s in bs])
    res["busy_temperature_hotspot_c"] = stat([_num(s.get("gm", {}).get("temperature_hotspot")) for s in bs])
    res["busy_temperature_hbm_c"] = stat([max([x for x in (s.get("gm", {}).get("temperature_hbm") or []) if _num(x) is not None and x < 1000] or [None]) if isinstance(s.get("gm", {}).get("temperature_hbm"), list) else None for s in bs])
    # active flags: fraction of busy samples in which the firmware reports each limiter as ACTIVE
    act = {}
    for k in ("active_ppt_pwr", "active_prochot_thrm", "active_socket_thrm", "active_vr_thrm", "active_hbm_thrm", "active_gfx_clk_below_host_limit"):
        vals = [s.get("viol", {}).get(k) for s in bs if "viol" in s]
        vals = [v for v in vals if isinstance(v, (bool, int))]
        act[k] = {"n": len(vals), "frac_true": (sum(1 for v in vals if v) / len(vals)) if vals else None}
    res["busy_active_flags"] = act
    for k in ("active_gfx_clk_below_host_limit_pwr", "active_gfx_clk_below_host_limit_thm", "active_gfx_clk_below_host_limit_total"):
        rows = [s.get("viol", {}).get(k) for s in bs if isinstance(s.get("viol", {}).get(k), list)]
        if rows:
            flat = [x for r in rows for x in (r[0] if r and isinstance(r[0], (list, tuple)) else r) if isinstance(x, (bool, int)) and x in (0, 1, True, False)]
            res["busy_" + k] = {"n": len(flat), "frac_true": (sum(1 for x in flat if x) / len(flat)) if flat else None}
    res["busy_per_percent"] = {k: stat([_num(s.get("viol", {}).get(k)) for s in bs if "viol" in s]) for k in
                               ("per_ppt_pwr", "per_prochot_thrm", "per_socket_thrm", "per_vr_thrm", "per_hbm_thrm", "per_gfx_clk_below_host_limit")}
    # accumulator deltas over the whole run (first -> last snapshot)
    dv = {}
    fv, lv = first.get("viol", {}), last.get("viol", {})
    for k in ("acc_counter", "acc_ppt_pwr", "acc_prochot_thrm", "acc_socket_thrm", "acc_vr_thrm", "acc_hbm_thrm", "acc_gfx_clk_below_host_limit"):
        if _num(fv.get(k)) is not None and _num(lv.get(k)) is not None:
            dv[k] = lv[k] - fv[k]
    res["violation_accumulator_deltas"] = dv
    if dv.get("acc_counter"):
        res["violation_fraction_of_ticks"] = {k: dv[k] / dv["acc_counter"] for k in dv if k != "acc_counter"}
    dg = {}
    fg, lg = first.get("gm", {}), last.get("gm", {})
    for k in ("accumulation_counter", "ppt_residency_acc", "prochot_residency_acc", "socket_thm_residency_acc", "vr_thm_residency_acc", "hbm_thm_residency_acc", "energy_accumulator"):
        if _num(fg.get(k)) is not None and _num(lg.get(k)) is not None:
            dg[k] = lg[k] - fg[k]
    res["gpu_metrics_accumulator_deltas"] = dg
    if dg.get("accumulation_counter"):
        res["residency_fraction"] = {k: dg[k] / dg["accumulation_counter"] for k in dg if k.endswith("residency_acc")}
    thr = sorted({str(s.get("gm", {}).get("throttle_status")) + "/" + str(s.get("gm", {}).get("indep_throttle_status")) for s in bs})
    res["busy_throttle_status_words"] = thr[:16]
    res["first_snapshot"] = first
    res["last_snapshot"] = last
    res["example_busy_snapshot"] = bs[len(bs) // 2] if bs else None
    json.dump(res, open(out_path, "w"), indent=1, default=str)
    brief = {k: res.get(k) for k in ("wall_s", "sample_hz", "n_busy_samples", "busy_gfxclk_mhz_all_xcc", "busy_socket_power_w", "busy_active_flags", "violation_fraction_of_ticks",
                                     "residency_fraction", "busy_throttle_status_words", "power_cap", "errors")}
    print(json.dumps(brief, default=str))


if __name__ == "__main__":
    main()
