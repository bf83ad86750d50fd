#!/usr/bin/env python3
"""Socket power / shader clock (rocm-smi) while scripts/probes/gemm_power_split_probe.bin runs the persistent GEMM's K-step with its
activities added one at a time (MFMAs, fragment reads, operand DMA): where the energy of a K-step goes.  (build line in the .hip file)"""
import os, re, subprocess, sys, threading, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exe = os.path.join(root, "scripts", "probes", "gemm_power_split_probe.bin")
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 2):
    for kind in ((0, 1, 2, 3) if os.environ.get('PROBE_KINDS') is None else tuple(int(k) for k in os.environ['PROBE_KINDS'].split(','))):
        samples, stop = [], False
        def sampler():
            while not stop:
                try:
                    o = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
                    p = re.search(r"Power \(W\):\s*([\d.]+)", o); c = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", o)
                    if p and c:
                        samples.append((float(p.group(1)), int(c.group(1))))
                except Exception:
                    pass
                time.sleep(0.05)
        th = threading.Thread(target=sampler); th.start()
        r = subprocess.run([exe, str(kind), str(secs)], capture_output=True, text=True)
        stop = True; th.join()
        s = samples[len(samples) // 4:]          # drop the ramp
        w = sum(p for p, _ in s) / max(len(s), 1); mhz = sum(c for _, c in s) / max(len(s), 1)
        line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-200:]
        m = re.search(r"([\d.]+) TFLOP/s", line)
        tf = float(m.group(1)) if m else 0.0
        print(f"rep {rep} {line} | {w:6.0f} W, sclk {mhz:5.0f} MHz, {tf / max(w, 1):.3f} TFLOP/s per W", flush=True)
