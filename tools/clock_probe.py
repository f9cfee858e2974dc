"""Samples the device's sclk / mclk / power (sysfs) while the training step runs from a cold start:
is the step-time ramp of the first ~25 steps the device's power management?  python tools/clock_probe.py"""
import glob
import importlib
import pathlib
import sys
import threading
import time

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import bench  # noqa: E402

mf = importlib.import_module("matrix-factorization-torch_amd")
dev = torch.device("cuda:0")


def read(path):
    try:
        return pathlib.Path(path).read_text()
    except OSError:
        return ""


import os  # noqa: E402

cards = [c for c in glob.glob("/sys/class/drm/card*/device") if read(c + "/pp_dpm_sclk")]
props = torch.cuda.get_device_properties(0)
mine = f"{props.pci_domain_id:04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}"
cards = [c for c in cards if mine in os.path.realpath(c)] or cards
print("device 0 is", mine, "->", cards)
samples, stop = [], False


def current(txt):
    for line in txt.splitlines():
        if line.strip().endswith("*"):
            return line.split(":")[1].strip().rstrip("*").strip()
    return "?"


def sampler():
    hw = glob.glob(cards[0] + "/hwmon/hwmon*/power1_average") + glob.glob(cards[0] + "/hwmon/hwmon*/power1_input")
    while not stop:
        samples.append((time.perf_counter(), current(read(cards[0] + "/pp_dpm_sclk")), current(read(cards[0] + "/pp_dpm_mclk")),
                        read(hw[0]).strip() if hw else "?"))
        time.sleep(0.002)


batches, _ = bench.make_batches(8, 8192, seed=1000, device=dev)
tr = bench.Trainer(mf, dev, "adam", 0)
tr.step(batches[0])
torch.cuda.synchronize()
time.sleep(0.5)
if cards:
    th = threading.Thread(target=sampler)
    th.start()
t0 = time.perf_counter()
marks = []
for i in range(60):
    tr.step(batches[i % 8])
    if i % 10 == 9:
        torch.cuda.synchronize()
        marks.append(time.perf_counter() - t0)
stop = True
if cards:
    th.join()
print("time after 10, 20, ... steps (ms):", " ".join(f"{1e3 * m:.1f}" for m in marks))
last = None
for t, s, m, p in samples:
    cur = (s, m)
    if cur != last:
        print(f"t={1e3 * (t - t0):8.1f} ms  sclk {s}  mclk {m}  power {p}")
        last = cur
print("samples:", len(samples), " last:", samples[-1][1:] if samples else None)
