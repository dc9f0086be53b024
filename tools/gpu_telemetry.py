"""Clock / power / throttle telemetry of one GPU for bench.py (measurement tooling, not on the product path).

The round-4 verdict's question: the same kernels gave 205 ms per step on one lease and 238 ms on another, and nothing in
the benchmark's output could tell a DVFS / power-cap effect from a host-side launch stall.  `Telemetry` reads the SMU's
metrics table through `amdsmi` (the python package ships with ROCm): shader clock per XCD, memory clock, socket power,
power cap, hot-spot temperature and the firmware's throttle-residency accumulators (PPT = power, thermal, PROCHOT) --
before, during (a sampler thread) and after the timed loop.  Falls back to the amdgpu sysfs files.  Every failure is
reported as a field, never raised: telemetry must not be able to hide the benchmark number."""
import glob
import os
import threading
import time

_ACC = ("prochot_residency_acc", "ppt_residency_acc", "socket_thm_residency_acc", "vr_thm_residency_acc",
        "hbm_thm_residency_acc", "accumulation_counter", "energy_accumulator", "gfx_activity_acc", "mem_activity_acc")
_INST = ("current_gfxclk", "average_gfxclk_frequency", "current_uclk", "average_uclk_frequency", "current_socket_power",
         "average_socket_power", "temperature_hotspot", "temperature_mem", "throttle_status", "indep_throttle_status",
         "average_gfx_activity", "average_umc_activity", "gfxclk_lock_status")


def _num(v):
    return v if isinstance(v, (int, float)) and not isinstance(v, bool) else None


def _median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2] if xs else None


class Telemetry:
    def __init__(self, device_index: int = 0, pci_bus_id=None):
        self.h = None
        self.smi = None
        self.err = None
        self.sysfs = None
        self.samples = []
        self._stop = threading.Event()
        self._thr = None
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            hs = amdsmi.amdsmi_get_processor_handles()
            pick = None
            if pci_bus_id is not None:
                for h in hs:
                    try:
                        bdf = amdsmi.amdsmi_get_gpu_device_bdf(h)           # dddd:bb:dd.f
                        if int(bdf.split(":")[1], 16) == int(pci_bus_id):
                            pick = h
                            break
                    except Exception:
                        pass
            self.h = pick if pick is not None else hs[min(device_index, len(hs) - 1)]
            self.smi = amdsmi
        except Exception as e:      # no amdsmi / no permission: sysfs
            self.err = f"amdsmi: {e!r}"
            cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/pp_dpm_sclk"))
            if cards:
                self.sysfs = os.path.dirname(cards[min(device_index, len(cards) - 1)])

    # ---- one reading ---------------------------------------------------------------------------------------------
    def _metrics(self):
        return self.smi.amdsmi_get_gpu_metrics_info(self.h)

    def _sysfs_read(self, rel):
        try:
            hits = glob.glob(os.path.join(self.sysfs, rel))
            return open(hits[0]).read() if hits else None
        except OSError:
            return None

    def _sysfs_sample(self):
        out = {}
        for key, rel in (("sclk_mhz", "pp_dpm_sclk"), ("mclk_mhz", "pp_dpm_mclk")):
            txt = self._sysfs_read(rel) or ""
            for line in txt.splitlines():
                if "*" in line:
                    try:
                        out[key] = float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
                    except ValueError:
                        pass
        for key, rel, scale in (("power_w", "hwmon/hwmon*/power1_average", 1e-6), ("power_w", "hwmon/hwmon*/power1_input", 1e-6),
                                ("power_cap_w", "hwmon/hwmon*/power1_cap", 1e-6), ("temp_hotspot_c", "hwmon/hwmon*/temp2_input", 1e-3),
                                ("busy_pct", "gpu_busy_percent", 1.0)):
            txt = self._sysfs_read(rel)
            if txt and key not in out:
                try:
                    out[key] = float(txt.strip()) * scale
                except ValueError:
                    pass
        return out

    def sample(self):
        """One light reading: {t, sclk_mhz (mean over XCDs), sclk_min_mhz, mclk_mhz, power_w, temp_hotspot_c, throttle}."""
        t = time.perf_counter()
        if self.h is None:
            s = self._sysfs_sample() if self.sysfs else {}
            s["t"] = t
            return s
        try:
            m = self._metrics()
        except Exception as e:
            return {"t": t, "error": repr(e)}
        clks = [c for c in (m.get("current_gfxclks") or []) if _num(c)]
        s = {"t": t}
        if clks:
            s["sclk_mhz"] = sum(clks) / len(clks)
            s["sclk_min_mhz"] = min(clks)
        elif _num(m.get("current_gfxclk")):
            s["sclk_mhz"] = m["current_gfxclk"]
        for k_out, k_in in (("mclk_mhz", "current_uclk"), ("power_w", "current_socket_power"), ("temp_hotspot_c", "temperature_hotspot"),
                            ("gfx_activity_pct", "average_gfx_activity")):
            if _num(m.get(k_in)) is not None:
                s[k_out] = m[k_in]
        if s.get("power_w") is None and _num(m.get("average_socket_power")) is not None:
            s["power_w"] = m["average_socket_power"]
        for k_in in ("throttle_status", "indep_throttle_status"):
            v = m.get(k_in)
            if v not in (None, "N/A", False, 0):
                s[k_in] = v if isinstance(v, (int, bool)) else str(v)
        return s

    def snapshot(self):
        """A full reading (instantaneous fields + the firmware's accumulators + power cap + clock limits)."""
        out = {"source": "amdsmi" if self.h is not None else ("sysfs" if self.sysfs else "none")}
        if self.err:
            out["amdsmi_error"] = self.err
        if self.h is None:
            out.update(self._sysfs_sample() if self.sysfs else {})
            return out
        smi = self.smi
        try:
            m = self._metrics()
            for k in _INST + _ACC:
                v = m.get(k)
                if _num(v) is not None or isinstance(v, bool):
                    out[k] = v
            clks = [c for c in (m.get("current_gfxclks") or []) if _num(c)]
            if clks:
                out["current_gfxclks"] = clks
            hbm = [c for c in (m.get("temperature_hbm") or []) if _num(c)]
            if hbm:
                out["temperature_hbm_max"] = max(hbm)
            for k in ("xcp_stats.gfx_below_host_limit_ppt_acc", "xcp_stats.gfx_below_host_limit_thm_acc",
                      "xcp_stats.gfx_below_host_limit_total_acc", "xcp_stats.gfx_low_utilization_acc", "xcp_stats.gfx_busy_acc"):
                v = m.get(k)
                try:          # [partition][xcc]: keep partition 0's numeric entries
                    row = [x for x in v[0] if _num(x) is not None]
                    if row:
                        out[k] = row
                except Exception:
                    pass
        except Exception as e:
            out["metrics_error"] = repr(e)
        try:
            cap = smi.amdsmi_get_power_cap_info(self.h)
            for k in ("power_cap", "default_power_cap", "max_power_cap", "min_power_cap"):
                if _num(cap.get(k)) is not None:
                    out[k + "_w"] = cap[k] / 1e6 if cap[k] > 1e5 else cap[k]
        except Exception as e:
            out["power_cap_error"] = repr(e)
        for nm, ct in (("gfx", "GFX"), ("mem", "MEM")):
            try:
                ci = smi.amdsmi_get_clock_info(self.h, getattr(smi.AmdSmiClkType, ct))
                out[f"{nm}_clk_limits_mhz"] = [ci.get("min_clk"), ci.get("max_clk")]
                if ci.get("clk_locked") not in (None, "N/A"):
                    out[f"{nm}_clk_locked"] = ci.get("clk_locked")
            except Exception:
                pass
        # what distinguishes one box of a pool from another beyond the shader clock: fabric / SoC clocks, partition modes,
        # the ASIC itself (round 5: two leases ran every kernel of the same step 16.5 % apart at the SAME memory clock, the
        # slower one at a HIGHER shader clock and LOWER power)
        for nm, ct in (("soc", "SOC"), ("df", "DF")):
            try:
                ci = smi.amdsmi_get_clock_info(self.h, getattr(smi.AmdSmiClkType, ct))
                out[f"{nm}_clk_mhz"] = {k: ci.get(k) for k in ("clk", "min_clk", "max_clk") if _num(ci.get(k)) is not None}
            except Exception:
                pass
        try:
            m2 = self._metrics()
            for k in ("current_socclk", "average_socclk_frequency", "vram_max_bandwidth", "pcie_link_width", "pcie_link_speed",
                      "num_partition", "xgmi_link_width", "xgmi_link_speed"):
                if _num(m2.get(k)) is not None:
                    out[k] = m2[k]
            socs = [c for c in (m2.get("current_socclks") or []) if _num(c)]
            if socs:
                out["current_socclks"] = socs
        except Exception:
            pass
        for key, fn in (("compute_partition", "amdsmi_get_gpu_compute_partition"), ("memory_partition", "amdsmi_get_gpu_memory_partition"),
                        ("asic", "amdsmi_get_gpu_asic_info"), ("vram", "amdsmi_get_gpu_vram_info"), ("vbios", "amdsmi_get_gpu_vbios_info"),
                        ("driver", "amdsmi_get_gpu_driver_info"), ("perf_level", "amdsmi_get_gpu_perf_level")):
            try:
                v = getattr(smi, fn)(self.h)
                out[key] = {k: (x if isinstance(x, (int, float, str, bool)) else str(x)) for k, x in v.items()} if isinstance(v, dict) else str(v)
            except Exception:
                pass
        try:
            v = smi.amdsmi_get_violation_status(self.h)
            out["violation_status"] = {k: x for k, x in v.items() if ("active" in k or "per_" in k) and _num(x) is not None and x}
        except Exception:
            pass
        return out

    # ---- sampler thread ------------------------------------------------------------------------------------------
    def start(self, hz: float = 10.0):
        self.samples = []
        self._stop.clear()

        def loop():
            period = 1.0 / hz
            while not self._stop.is_set():
                self.samples.append(self.sample())
                self._stop.wait(period)
        self._thr = threading.Thread(target=loop, daemon=True)
        self._thr.start()

    def stop(self):
        if self._thr is not None:
            self._stop.set()
            self._thr.join(timeout=2.0)
            self._thr = None
        return self.samples

    @staticmethod
    def summarize(samples, t0=None, t1=None):
        """min / median / max per numeric field over the samples taken inside [t0, t1] (perf_counter times)."""
        ss = [s for s in samples if (t0 is None or s.get("t", 0) >= t0) and (t1 is None or s.get("t", 0) <= t1)]
        out = {"n": len(ss)}
        for k in ("sclk_mhz", "sclk_min_mhz", "mclk_mhz", "power_w", "temp_hotspot_c", "gfx_activity_pct"):
            xs = [s[k] for s in ss if _num(s.get(k)) is not None]
            if xs:
                out[k] = {"min": min(xs), "median": _median(xs), "max": max(xs), "first": xs[0], "last": xs[-1]}
        thr = sorted({str(s[k]) for s in ss for k in ("throttle_status", "indep_throttle_status") if k in s})
        if thr:
            out["throttle_flags_seen"] = thr
        errs = [s["error"] for s in ss if "error" in s]
        if errs:
            out["errors"] = len(errs)
            out["first_error"] = errs[0]
        return out

    @staticmethod
    def delta(before, after):
        """Differences of the firmware's accumulators between two snapshots (throttle residencies, energy)."""
        out = {}
        for k in _ACC:
            if _num(before.get(k)) is not None and _num(after.get(k)) is not None:
                out[k] = after[k] - before[k]
        for k in ("xcp_stats.gfx_below_host_limit_ppt_acc", "xcp_stats.gfx_below_host_limit_thm_acc",
                  "xcp_stats.gfx_below_host_limit_total_acc", "xcp_stats.gfx_low_utilization_acc", "xcp_stats.gfx_busy_acc"):
            a, b = before.get(k), after.get(k)
            if isinstance(a, list) and isinstance(b, list) and len(a) == len(b):
                out[k] = [y - x for x, y in zip(a, b)]
        return out


if __name__ == "__main__":
    import json
    t = Telemetry()
    print(json.dumps(t.snapshot(), default=str))
    t.start(10)
    time.sleep(0.5)
    print(json.dumps(Telemetry.summarize(t.stop()), default=str))
