#!/usr/bin/env python3
"""Socket power / shader clock / throttle telemetry of GPU 0 while a command runs.

    python tools/power_trace.py --out gpurun_out/power_f32.json -- python bench.py --steps 5 --no-extra ...

Samples at ~20 Hz from a thread of THIS process (the command is a child), through the amdsmi python binding when it
imports and through the hwmon / sysfs files otherwise.  Every API call is individually guarded: a field the driver
does not expose is recorded as null, never fatal.  The summary block covers the "busy" samples only (socket power above
half of the run's maximum), which is where the GEMM family runs.
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import threading
import time


def _jsonable(x):
    if isinstance(x, dict):
        return {str(k): _jsonable(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_jsonable(v) for v in x]
    if isinstance(x, (int, float, str, bool)) or x is None:
        return x
    return str(x)


class SmiSampler:
    def __init__(self):
        import amdsmi

        self.a = amdsmi
        amdsmi.amdsmi_init()
        self.h = amdsmi.amdsmi_get_processor_handles()[0]

    def _try(self, fn, *args):
        try:
            return _jsonable(fn(self.h, *args))
        except Exception as e:  # noqa: BLE001 - telemetry must never kill the run
            return {"error": str(e)[:120]}

    def static(self):
        a = self.a
        out = {"source": "amdsmi"}
        out["power_cap"] = self._try(a.amdsmi_get_power_cap_info)
        for name in ("amdsmi_get_gpu_asic_info", "amdsmi_get_gpu_board_info"):
            if hasattr(a, name):
                out[name] = self._try(getattr(a, name))
        return out

    def sample(self):
        a = self.a
        s = {"power": self._try(a.amdsmi_get_power_info)}
        try:
            s["gfx_clock"] = _jsonable(a.amdsmi_get_clock_info(self.h, a.AmdSmiClkType.GFX))
        except Exception as e:  # noqa: BLE001
            s["gfx_clock"] = {"error": str(e)[:120]}
        m = self._try(a.amdsmi_get_gpu_metrics_info)
        if isinstance(m, dict):
            keep = ("average_socket_power", "current_socket_power", "current_gfxclk", "current_gfxclks", "average_gfxclk_frequency",
                    "throttle_status", "indep_throttle_status", "temperature_hotspot", "temperature_mem",
                    "prochot_residency_acc", "ppt_residency_acc", "socket_thm_residency_acc", "vr_thm_residency_acc",
                    "hbm_thm_residency_acc", "accumulation_counter", "average_gfx_activity", "average_umc_activity",
                    "gfxclk_lock_status", "energy_accumulator", "current_uclk")
            s["metrics"] = {k: m.get(k) for k in keep if k in m} or m
        else:
            s["metrics"] = m
        if hasattr(a, "amdsmi_get_violation_status"):
            s["violation"] = self._try(a.amdsmi_get_violation_status)
        return s


class SysfsSampler:
    def __init__(self):
        devs = sorted(glob.glob("/sys/class/drm/card*/device"))
        self.dev = next((d for d in devs if os.path.exists(os.path.join(d, "pp_dpm_sclk"))), devs[0] if devs else None)
        hw = glob.glob(os.path.join(self.dev or "", "hwmon", "hwmon*"))
        self.hw = hw[0] if hw else None

    @staticmethod
    def _read(path):
        try:
            return open(path).read().strip()
        except Exception:  # noqa: BLE001
            return None

    def static(self):
        return {"source": "sysfs", "device": self.dev,
                "power1_cap_uW": self._read(os.path.join(self.hw or "", "power1_cap")),
                "power1_cap_max_uW": self._read(os.path.join(self.hw or "", "power1_cap_max"))}

    def sample(self):
        hw, dev = self.hw or "", self.dev or ""
        return {"power1_average_uW": self._read(os.path.join(hw, "power1_average")),
                "power1_input_uW": self._read(os.path.join(hw, "power1_input")),
                "freq1_input_Hz": self._read(os.path.join(hw, "freq1_input")),
                "pp_dpm_sclk": self._read(os.path.join(dev, "pp_dpm_sclk")),
                "temp1_input": self._read(os.path.join(hw, "temp1_input"))}


def _num(x):
    try:
        v = float(x)
        return v if v == v and abs(v) < 1e12 else None
    except Exception:  # noqa: BLE001
        return None


def _watts(s):
    p = s.get("power") or {}
    for k in ("current_socket_power", "average_socket_power", "socket_power"):
        v = _num(p.get(k)) if isinstance(p, dict) else None
        if v is not None and v > 0:
            return v
    m = s.get("metrics") or {}
    for k in ("current_socket_power", "average_socket_power"):
        v = _num(m.get(k)) if isinstance(m, dict) else None
        if v is not None and 0 < v < 5000:
            return v
    for k in ("power1_average_uW", "power1_input_uW"):
        v = _num(s.get(k))
        if v:
            return v * 1e-6
    return None


def _mhz(s):
    c = s.get("gfx_clock") or {}
    v = _num(c.get("clk")) if isinstance(c, dict) else None
    if v:
        return v
    m = s.get("metrics") or {}
    if isinstance(m, dict):
        cl = m.get("current_gfxclks")
        if isinstance(cl, list):
            vals = [x for x in map(_num, cl) if x and x < 60000]
            if vals:
                return sum(vals) / len(vals)
        v = _num(m.get("current_gfxclk"))
        if v and v < 60000:
            return v
    v = _num(s.get("freq1_input_Hz"))
    return v * 1e-6 if v else None


def summarise(samples):
    w = [(_watts(s), _mhz(s), s) for s in samples]
    pw = [x[0] for x in w if x[0] is not None]
    if not pw:
        return {"note": "no power readings"}
    thr = 0.5 * max(pw)
    busy = [x for x in w if x[0] is not None and x[0] >= thr]

    def stats(v):
        v = sorted(x for x in v if x is not None)
        if not v:
            return None
        return {"mean": sum(v) / len(v), "p10": v[len(v) // 10], "p50": v[len(v) // 2], "p90": v[(len(v) * 9) // 10], "max": v[-1]}

    out = {"busy_samples": len(busy), "all_samples": len(samples), "busy_power_W": stats([x[0] for x in busy]),
           "busy_gfx_clock_MHz": stats([x[1] for x in busy])}
    # residency counters are cumulative: the difference over the busy window says which limiter was active
    first, last = busy[0][2].get("metrics") or {}, busy[-1][2].get("metrics") or {}
    if isinstance(first, dict) and isinstance(last, dict):
        res = {}
        for k in ("ppt_residency_acc", "prochot_residency_acc", "socket_thm_residency_acc", "vr_thm_residency_acc",
                  "hbm_thm_residency_acc", "accumulation_counter"):
            a, b = _num(first.get(k)), _num(last.get(k))
            if a is not None and b is not None:
                res[k] = b - a
        if res:
            acc = res.get("accumulation_counter") or 0
            out["throttler_residency_delta"] = res
            if acc > 0:
                out["throttler_residency_frac"] = {k: v / acc for k, v in res.items() if k != "accumulation_counter"}
        ts = sorted({str((x[2].get("metrics") or {}).get("throttle_status")) for x in busy})
        out["throttle_status_values_seen"] = ts
    viol = [x[2].get("violation") for x in busy if isinstance(x[2].get("violation"), dict)]
    if viol:
        out["violation_last_busy_sample"] = {k: v for k, v in viol[-1].items()
                                             if (k.startswith("active_") or k.startswith("per_")) and not isinstance(v, list)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--hz", type=float, default=20.0)
    ap.add_argument("--keep", type=int, default=400, help="raw samples kept in the file (evenly thinned)")
    ap.add_argument("cmd", nargs=argparse.REMAINDER)
    args = ap.parse_args()
    cmd = args.cmd[1:] if args.cmd and args.cmd[0] == "--" else args.cmd
    if not cmd:
        raise SystemExit("power_trace.py: no command given")
    try:
        smp = SmiSampler()
    except Exception as e:  # noqa: BLE001
        print(f"power_trace: amdsmi unavailable ({e}); using sysfs", file=sys.stderr)
        smp = SysfsSampler()
    static = smp.static()
    samples, stop = [], threading.Event()

    def loop():
        t0 = time.time()
        while not stop.is_set():
            s = smp.sample()
            s["t"] = round(time.time() - t0, 3)
            samples.append(s)
            stop.wait(1.0 / args.hz)

    th = threading.Thread(target=loop, daemon=True)
    th.start()
    t0 = time.time()
    rc = subprocess.call(cmd)
    wall = time.time() - t0
    stop.set()
    th.join(timeout=5)
    step = max(1, len(samples) // max(args.keep, 1))
    doc = {"command": " ".join(cmd), "rc": rc, "wall_s": wall, "static": static, "summary": summarise(samples),
           "samples_thinned": [{"t": s["t"], "W": _watts(s), "gfx_MHz": _mhz(s)} for s in samples[::step]],
           "raw_first_busy": next((s for s in samples if (_watts(s) or 0) >= 0.5 * max([_watts(x) or 0 for x in samples] or [0])), None)}
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps({"power_trace": args.out, "summary": doc["summary"]}))
    raise SystemExit(rc)


if __name__ == "__main__":
    main()
