"""Shader clock and socket power of the GPU while a timed region runs (bench.py: roofline.sclk_mhz_mean, power_w_mean,
frac_at_measured_clock).

The MI355X is power / clock limited under the contraction kernels (a launch fed with zeros runs 6 % faster than the same launch on
real operands): a roofline fraction priced at the 2.4 GHz specification clock mixes what the loop loses with what the clock loses.
This module samples the clock the part actually sustains.  It runs as a CHILD PROCESS (`python bcos_hip/telemetry.py`), so that the
sampling loop never competes with the benchmark's launch loop for the interpreter lock, and prints one line per sample:

    <time.time()> <mean gfx clock of the XCDs, MHz> <socket power, W> <max XCD clock> <min XCD clock>

Sources, first one that answers: the amdsmi Python binding (gpu metrics table: per-XCD gfx clocks, current socket power), librocm_smi64
through ctypes, then the sysfs files of the device (pp_dpm_sclk, hwmon power1_average / power1_input).  Development aid of the
measurement only: nothing of the product path imports it.
"""
import glob
import os
import subprocess
import sys
import time


def _amdsmi_source(index):
    import amdsmi
    amdsmi.amdsmi_init()
    handles = amdsmi.amdsmi_get_processor_handles()
    h = handles[min(index, len(handles) - 1)]

    def num(v):
        return float(v) if isinstance(v, (int, float)) else None

    def read():
        clk = pw = hi = lo = None
        try:
            m = amdsmi.amdsmi_get_gpu_metrics_info(h)
            cl = [float(c) for c in (m.get("current_gfxclks") or []) if isinstance(c, (int, float)) and 0 < c < 10000]
            if cl:
                clk, hi, lo = sum(cl) / len(cl), max(cl), min(cl)
            elif num(m.get("current_gfxclk")):
                clk = hi = lo = num(m.get("current_gfxclk"))
            pw = num(m.get("current_socket_power")) or num(m.get("average_socket_power"))
        except Exception:
            pass
        if clk is None:
            c = amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX)
            clk = hi = lo = num(c.get("clk"))
        if pw is None:
            p = amdsmi.amdsmi_get_power_info(h)
            pw = num(p.get("current_socket_power")) or num(p.get("average_socket_power"))
        return clk, pw, hi, lo

    read()
    return read, "amdsmi (gpu metrics: mean of the per-XCD gfx clocks, current socket power)"


def _rsmi_source(index):
    import ctypes as C
    lib = C.CDLL("librocm_smi64.so")
    if lib.rsmi_init(C.c_uint64(0)) != 0:
        raise RuntimeError("rsmi_init failed")

    class Freqs(C.Structure):
        _fields_ = [("has_deep_sleep", C.c_bool), ("num_supported", C.c_uint32), ("current", C.c_uint32), ("frequency", C.c_uint64 * 33)]

    def read():
        f = Freqs()
        clk = None
        if lib.rsmi_dev_gpu_clk_freq_get(C.c_uint32(index), C.c_int(0), C.byref(f)) == 0 and f.num_supported:
            clk = f.frequency[min(f.current, 32)] / 1e6
        pw = C.c_uint64(0)
        kind = C.c_int(0)
        p = None
        if lib.rsmi_dev_power_get(C.c_uint32(index), C.byref(pw), C.byref(kind)) == 0:
            p = pw.value / 1e6
        return clk, p, clk, clk

    if read()[0] is None:
        raise RuntimeError("rocm_smi: no clock")
    return read, "librocm_smi64 (current sclk level, socket power)"


def _sysfs_source(index):
    cards = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
    if not cards:
        raise RuntimeError("no pp_dpm_sclk")
    dev = os.path.dirname(cards[min(index, len(cards) - 1)])
    pfiles = glob.glob(os.path.join(dev, "hwmon/hwmon*/power1_average")) + glob.glob(os.path.join(dev, "hwmon/hwmon*/power1_input"))

    def read():
        clk = None
        for line in open(os.path.join(dev, "pp_dpm_sclk")):
            if "*" in line:
                clk = float(line.split(":")[1].strip().rstrip("*").strip().lower().replace("mhz", ""))
        pw = float(open(pfiles[0]).read()) / 1e6 if pfiles else None
        return clk, pw, clk, clk

    if read()[0] is None:
        raise RuntimeError("sysfs: no current level")
    return read, "sysfs (pp_dpm_sclk current level, hwmon power1)"


def physical_index(logical: int, environ=None) -> int:
    """Index of the PHYSICAL device behind logical HIP ordinal `logical` of this process: the SMI libraries and sysfs enumerate every
    GPU of the node, HIP only those that HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES (HIP runtime) and ROCR_VISIBLE_DEVICES (ROCr, applied
    first) leave visible, renumbered from 0 (sampling the logical ordinal reads another, possibly idle, GPU's clock).
    Entries that are not plain integers (UUIDs) cannot be resolved without the runtime: the logical ordinal is returned unchanged."""
    env = os.environ if environ is None else environ

    def parse(name):
        val = env.get(name)
        if val is None or val.strip() == "":
            return None
        ids = []
        for tok in val.split(","):
            tok = tok.strip()
            if not tok.lstrip("-").isdigit():
                return False
            if int(tok) < 0:
                break                      # (a negative entry ends the list, as in the runtimes)
            ids.append(int(tok))
        return ids
    idx = int(logical)
    hip = parse("HIP_VISIBLE_DEVICES")
    if hip is None:
        hip = parse("CUDA_VISIBLE_DEVICES")
    rocr = parse("ROCR_VISIBLE_DEVICES")
    for ids in (hip, rocr):                # HIP's list indexes what ROCr left visible
        if ids is False:
            return int(logical)
        if ids is not None:
            if idx >= len(ids):
                return int(logical)
            idx = ids[idx]
    return idx


def open_source(index=0):
    errs = []
    for mk in (_amdsmi_source, _rsmi_source, _sysfs_source):
        try:
            return mk(index)
        except Exception as exc:       # the next source
            errs.append(f"{mk.__name__}: {type(exc).__name__}: {exc}")
    raise RuntimeError("; ".join(errs))


def main():
    try:        # (the benchmark may have pinned itself -- OMP_PROC_BIND -- before starting this child: do not share its core)
        os.sched_setaffinity(0, set(range(os.cpu_count() or 1)))
    except (AttributeError, OSError):
        pass
    index = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    interval = float(sys.argv[2]) if len(sys.argv) > 2 else 0.02
    try:
        read, name = open_source(index)
    except Exception as exc:
        print("# unavailable: " + " ".join(str(exc).split()), flush=True)
        return 1
    print(f"# source: {name}", flush=True)
    out = sys.stdout
    while True:
        t = time.time()
        try:
            clk, pw, hi, lo = read()
        except Exception:
            clk = pw = hi = lo = None
        f = lambda v: "nan" if v is None else f"{v:.1f}"     # noqa: E731
        try:
            out.write(f"{t:.4f} {f(clk)} {f(pw)} {f(hi)} {f(lo)}\n")
            out.flush()
        except BrokenPipeError:
            return 0
        time.sleep(max(0.0, interval - (time.time() - t)))


class Sampler:
    """Context-free handle used by bench.py: start() before the warm-up, window(t0, t1) after the timed region."""

    def __init__(self, index=0, interval=0.02):
        self.proc = None
        self.index, self.interval = index, interval
        self.path = None

    def start(self):
        import tempfile
        fd, self.path = tempfile.mkstemp(prefix="bcos_telemetry_", suffix=".txt")
        try:       # run as a plain script: the child imports neither this package nor torch (seconds of start-up otherwise)
            self.proc = subprocess.Popen([sys.executable, os.path.abspath(__file__), str(self.index), str(self.interval)],
                                         stdout=fd, stderr=subprocess.DEVNULL)
        except Exception:
            self.proc = None
        os.close(fd)
        import atexit
        atexit.register(self._cleanup)          # a run that dies before window() must not leave the child sampling for ever
        return self

    def _cleanup(self):
        self.stop()
        try:
            if self.path and os.path.exists(self.path):
                os.unlink(self.path)
        except OSError:
            pass

    def wait_ready(self, timeout=10.0):
        """block until the child has printed its first line (source found, or none available)"""
        t_end = time.time() + timeout
        while self.proc is not None and time.time() < t_end:
            try:
                if os.path.getsize(self.path) > 0:
                    return True
            except OSError:
                return False
            time.sleep(0.05)
        return False

    def stop(self):
        if self.proc is not None:
            self.proc.terminate()
            try:
                self.proc.wait(timeout=5)
            except Exception:
                self.proc.kill()
            self.proc = None

    def window(self, t0, t1):
        """-> dict(sclk_mhz_mean, sclk_mhz_min, sclk_mhz_max, power_w_mean, power_w_max, samples, source) over the samples with
        t0 <= time.time() <= t1 (None values when the node offers no readable source)."""
        self.stop()
        res = dict(sclk_mhz_mean=None, sclk_mhz_min=None, sclk_mhz_max=None, power_w_mean=None, power_w_max=None, samples=0, source=None)
        if not self.path or not os.path.exists(self.path):
            return res
        clks, pws, los, his = [], [], [], []
        for line in open(self.path):
            if line.startswith("#"):
                res["source"] = line[1:].strip()
                continue
            p = line.split()
            if len(p) < 5:
                continue
            try:
                t = float(p[0])
                v = [float(q) for q in p[1:5]]
            except ValueError:
                continue
            if t < t0 or t > t1:
                continue
            if v[0] == v[0]:
                clks.append(v[0]); his.append(v[2]); los.append(v[3])
            if v[1] == v[1]:
                pws.append(v[1])
        try:
            os.unlink(self.path)
        except OSError:
            pass
        res["samples"] = len(clks)
        if clks:
            res.update(sclk_mhz_mean=round(sum(clks) / len(clks), 1), sclk_mhz_min=round(min(los), 1), sclk_mhz_max=round(max(his), 1))
        if pws:
            res.update(power_w_mean=round(sum(pws) / len(pws), 1), power_w_max=round(max(pws), 1))
        return res


if __name__ == "__main__":
    sys.exit(main())
