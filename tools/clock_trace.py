#!/usr/bin/env python3
"""Sample the GPU's shader clock, memory clock and socket power from sysfs (amdgpu hwmon: freq1_input = sclk,
freq2_input = mclk, power1_average / power1_input) while a command runs; one line per sample.

    python tools/clock_trace.py --out trace.txt --hz 20 -- python tools/gemm_bench.py --rows 8192 ...

The evidence behind "the fp32-MFMA GEMM runs at ~2.0-2.1 GHz, not the 2.4 GHz the 157.3 TFLOP/s peak assumes"
(DESIGN.md 4.2): the trace is taken DURING the timed launches, with the benchmark's own output beside it.
"""
import argparse
import glob
import os
import subprocess
import sys
import time


def read(p):
    try:
        with open(p) as f:
            return f.read().strip()
    except OSError:
        return ""


def sources():
    out = []
    for hw in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        out.append(hw)
    return out


def hip_device_pci():
    """PCI address of HIP device 0 (a shared host shows every GPU in sysfs: the trace marks which node is ours)."""
    code = ("import torch;p=torch.cuda.get_device_properties(0);"
            "print('%04x:%02x:%02x.0'%(p.pci_domain_id,p.pci_bus_id,p.pci_device_id))")
    try:
        return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=180).stdout.strip().splitlines()[-1]
    except Exception:
        return ""


def pci_of(hw):
    # /sys/class/drm/cardN/device -> ../../../0000:75:00.0
    return os.path.basename(os.path.realpath(os.path.join(hw, "..", "..")))


def sample(hw):
    f1, f2 = read(hw + "/freq1_input"), read(hw + "/freq2_input")
    pw = read(hw + "/power1_average") or read(hw + "/power1_input")
    t = read(hw + "/temp1_input")
    return (int(f1) / 1e6 if f1 else -1, int(f2) / 1e6 if f2 else -1, int(pw) / 1e6 if pw else -1, int(t) / 1e3 if t else -1)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--hz", type=float, default=20.0)
    ap.add_argument("cmd", nargs=argparse.REMAINDER)
    o = ap.parse_args()
    cmd = o.cmd[1:] if o.cmd and o.cmd[0] == "--" else o.cmd
    hws = sources()
    mine = hip_device_pci()
    with open(o.out, "w") as f:
        f.write("# HIP device 0 is PCI %s\n" % (mine or "?"))
        for k, hw in enumerate(hws):
            f.write("# node %d: %s  PCI %s%s\n" % (k, hw, pci_of(hw), "   <-- HIP device 0" if mine and pci_of(hw) == mine else ""))
        f.write("# t_s  then per node: sclk_MHz  mclk_MHz  power_W  temp_C\n")
        pr = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        t0 = time.time()
        while pr.poll() is None:
            row = ["%7.2f" % (time.time() - t0)]
            for hw in hws:
                row.append("%7.0f %7.0f %7.1f %5.1f" % sample(hw))
            f.write("  ".join(row) + "\n")
            time.sleep(1.0 / o.hz)
        out = pr.stdout.read()
        f.write("# ---- command output ----\n" + "".join("# " + l + "\n" for l in out.splitlines()))
    sys.stdout.write(out)
    sys.exit(pr.returncode)
