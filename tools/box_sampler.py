"""Power and shader-clock samples around a timed region, so that every kernel A/B carries the state of the box it ran on (the boxes of this pool
differ by 4-10 % and the matrix kernels are power-limited: DESIGN.md section 5).  Reads the amdgpu hwmon files directly (microseconds per sample; no
rocm-smi process in the timed region):  power1_average | power1_input [uW],  freq1_input [Hz] (the shader clock the SMU reports).
    with Sampler() as s:
        ... timed loop, ending in a synchronize ...
    print(s.summary())          # "box: 712 W, sclk 1.84 GHz (31 samples)"  or  "box: no hwmon"
"""
import glob
import os
import threading
import time


def _hwmon():
    for card in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        try:
            if open(os.path.join(card, "vendor")).read().strip() != "0x1002":
                continue
        except OSError:
            continue
        for h in sorted(glob.glob(os.path.join(card, "hwmon", "hwmon*"))):
            power = next((p for p in (os.path.join(h, "power1_average"), os.path.join(h, "power1_input")) if os.path.exists(p)), None)
            freq = os.path.join(h, "freq1_input")
            if power or os.path.exists(freq):
                return power, freq if os.path.exists(freq) else None
    return None, None


def _read(path):
    try:
        return float(open(path).read().strip())
    except (OSError, ValueError, TypeError):
        return None


class Sampler:
    def __init__(self, period=0.005):
        self.period, self.power, self.freq = period, [], []
        self.p_path, self.f_path = _hwmon()
        self._stop = threading.Event()
        self._thread = None

    def _run(self):
        while not self._stop.is_set():
            p, f = _read(self.p_path) if self.p_path else None, _read(self.f_path) if self.f_path else None
            if p is not None:
                self.power.append(p * 1e-6)
            if f is not None:
                self.freq.append(f * 1e-9)
            self._stop.wait(self.period)

    def __enter__(self):
        if self.p_path or self.f_path:
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._thread:
            self._thread.join()
        return False

    def summary(self):
        if not self.power and not self.freq:
            return "box: no hwmon"
        parts = []
        if self.power:
            parts.append(f"{sum(self.power) / len(self.power):.0f} W (max {max(self.power):.0f})")
        if self.freq:
            parts.append(f"sclk {sum(self.freq) / len(self.freq):.2f} GHz (min {min(self.freq):.2f})")
        return "box: " + ", ".join(parts) + f" ({max(len(self.power), len(self.freq))} samples)"
