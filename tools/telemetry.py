"""Socket power / shader clock sampler for the GPU box (measurement infrastructure, not product code).

``Sampler(device_index).start() ... .stop()`` polls the SMU's gpu_metrics table from a background thread (>= 20 Hz) and
returns a summary: mean / max socket power, mean shader clock over the XCDs, the power cap, throttle bits.  Sources, first
one that works: the ``amdsmi`` python bindings (one ``amdsmi_get_gpu_metrics_info`` call per sample), the amdgpu hwmon /
pp_dpm sysfs files, the ``rocm-smi`` CLI (slow: ~3 Hz).  Used by ``bench.py`` (``roofline.sclk_mhz`` / ``power_w``) and by
``tools/power_probe.py`` (the evidence behind DESIGN's "power-bound" statement).
"""
import glob
import os
import re
import subprocess
import threading
import time


def _num(v):
    try:
        f = float(v)
    except (TypeError, ValueError):
        return None
    return f if f == f and f < 6.0e4 else None          # 0xFFFF / N/A markers of the metrics table


class _AmdSmi:
    name = "amdsmi.amdsmi_get_gpu_metrics_info"

    def __init__(self, index):
        import amdsmi
        self.smi = amdsmi
        amdsmi.amdsmi_init()
        handles = amdsmi.amdsmi_get_processor_handles()
        self.h = handles[index]
        self.cap = None
        try:
            cap = float(amdsmi.amdsmi_get_power_cap_info(self.h).get("power_cap") or 0)
            self.cap = (cap / 1e6 if cap > 1e5 else cap) or None            # microwatts in this binding
        except Exception:
            pass
        self.read()                                      # raises if the table is not readable

    RESIDENCY = ("ppt_residency_acc", "prochot_residency_acc", "socket_thm_residency_acc", "vr_thm_residency_acc",
                 "hbm_thm_residency_acc", "gfx_activity_acc", "energy_accumulator")

    def counters(self):
        """Monotonic limiter-residency accumulators of the metrics table (PPT = package power tracking, thermal, VR,
        prochot): their deltas over a run say WHICH limiter held the clock down."""
        m = self.smi.amdsmi_get_gpu_metrics_info(self.h)
        out = {}
        for k in self.RESIDENCY:
            v = m.get(k)
            if isinstance(v, (int, float)) and v == v:
                out[k] = float(v)
        return out

    def read(self):
        m = self.smi.amdsmi_get_gpu_metrics_info(self.h)
        clks = [c for c in (_num(c) for c in (m.get("current_gfxclks") or [])) if c]
        if not clks and _num(m.get("current_gfxclk")):
            clks = [_num(m.get("current_gfxclk"))]
        power = _num(m.get("current_socket_power")) or _num(m.get("average_socket_power"))
        return {"power_w": power, "sclk_mhz": sum(clks) / len(clks) if clks else None,
                "sclk_min_mhz": min(clks) if clks else None,
                "throttle": m.get("indep_throttle_status") if m.get("indep_throttle_status") not in (None, "N/A")
                else m.get("throttle_status"),
                "activity": _num(m.get("average_gfx_activity"))}


class _Sysfs:
    name = "sysfs hwmon / pp_dpm_sclk"

    def __init__(self, index):
        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/hwmon/hwmon*"))
        if not cards:
            raise RuntimeError("no amdgpu hwmon")
        self.hw = cards[min(index, len(cards) - 1)]
        self.dev = os.path.dirname(os.path.dirname(self.hw))
        self.pfile = next((p for p in (os.path.join(self.hw, f) for f in ("power1_input", "power1_average"))
                           if os.path.exists(p)), None)
        self.cap = None
        try:
            self.cap = float(open(os.path.join(self.hw, "power1_cap")).read()) / 1e6
        except Exception:
            pass
        self.read()

    def read(self):
        power = sclk = None
        if self.pfile:
            power = float(open(self.pfile).read()) / 1e6
        f = os.path.join(self.hw, "freq1_input")
        if os.path.exists(f):
            sclk = float(open(f).read()) / 1e6
        else:
            m = re.search(r"(\d+)Mhz \*", open(os.path.join(self.dev, "pp_dpm_sclk")).read())
            sclk = float(m.group(1)) if m else None
        if power is None and sclk is None:
            raise RuntimeError("nothing readable")
        return {"power_w": power, "sclk_mhz": sclk, "sclk_min_mhz": sclk, "throttle": None, "activity": None}


class _RocmSmi:
    name = "rocm-smi --showpower --showclocks (CLI)"

    def __init__(self, index):
        self.index, self.cap = index, None
        self.read()

    def read(self):
        out = subprocess.run(["rocm-smi", "-d", str(self.index), "--showpower", "--showclocks"], capture_output=True,
                             text=True, timeout=10).stdout
        p = re.search(r"Power \(W\):\s*([\d.]+)", out)
        s = re.search(r"sclk clock level:.*?\((\d+)Mhz\)", out)
        if not p and not s:
            raise RuntimeError("rocm-smi gave nothing")
        return {"power_w": float(p.group(1)) if p else None, "sclk_mhz": float(s.group(1)) if s else None,
                "sclk_min_mhz": None, "throttle": None, "activity": None}


def open_source(index=0):
    errors = []
    for cls in (_AmdSmi, _Sysfs, _RocmSmi):
        try:
            return cls(index), errors
        except Exception as e:                      # try the next source; the reasons are reported
            errors.append(f"{cls.__name__}: {type(e).__name__}: {e}")
    return None, errors


class Sampler:
    def __init__(self, index=0, period_s=0.02):
        self.src, self.errors = open_source(index)
        self.period = period_s
        self.samples = []
        self._stop = threading.Event()
        self._thread = None

    @property
    def ok(self):
        return self.src is not None

    def _loop(self):
        while not self._stop.is_set():
            t = time.perf_counter()
            try:
                s = self.src.read()
                s["t"] = t
                self.samples.append(s)
            except Exception:
                pass
            left = self.period - (time.perf_counter() - t)
            if left > 0:
                self._stop.wait(left)

    def start(self):
        self.samples = []
        self._stop.clear()
        self._c0 = self.src.counters() if self.ok and hasattr(self.src, "counters") else {}
        if self.ok:
            self._thread = threading.Thread(target=self._loop, daemon=True)
            self._thread.start()
        return self

    def stop(self, skip_s=0.0):
        """-> summary over the samples taken since start() (the first ``skip_s`` seconds dropped: ramp-up)."""
        self._stop.set()
        if self._thread is not None:
            self._thread.join()
        if not self.ok:
            return {"source": None, "errors": self.errors}
        ss = self.samples
        if ss and skip_s:
            t0 = ss[0]["t"] + skip_s
            ss = [s for s in ss if s["t"] >= t0] or ss[-1:]

        def col(k):
            return [s[k] for s in ss if s.get(k) is not None]
        pw, ck, cm = col("power_w"), col("sclk_mhz"), col("sclk_min_mhz")
        dur = ss[-1]["t"] - ss[0]["t"] if len(ss) > 1 else 0.0
        thr = [s["throttle"] for s in ss if s.get("throttle") not in (None, "N/A")]
        c1 = self.src.counters() if hasattr(self.src, "counters") else {}
        deltas = {k: c1[k] - v for k, v in self._c0.items() if k in c1}
        return {"limiter_residency_deltas": deltas or None,"source": self.src.name, "samples": len(ss), "rate_hz": (len(ss) - 1) / dur if dur > 0 else None,
                "power_w": sum(pw) / len(pw) if pw else None, "power_w_max": max(pw) if pw else None,
                "power_cap_w": self.src.cap, "sclk_mhz": sum(ck) / len(ck) if ck else None,
                "sclk_mhz_min": min(cm) if cm else None, "sclk_mhz_max": max(ck) if ck else None,
                "throttle_seen": sorted({str(t) for t in thr})[:4] if thr else None}
