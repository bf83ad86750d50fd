#!/usr/bin/env python3
"""Idle time between consecutive kernels of one pgd() call (is the step launch-bound anywhere?).

Run on the GPU box:  python scripts/launch_gaps.py   (wraps `rocprofv3 --kernel-trace` around a one-step bench.py run, reads
the per-dispatch start / end stamps, and prints for the LAST pgd() call: span, sum of kernel durations, sum of the gaps
between a kernel's end and the next kernel's start, the gap histogram and the ten largest gaps with the kernels either side).
Output: gpurun_out/launch_gaps.log
"""
import csv
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")


def main():
    os.makedirs(OUT, exist_ok=True)
    d = os.path.join(OUT, "gaps_prof")
    subprocess.run(["rm", "-rf", d])
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "trace", "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-roofline"]
    r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=900)
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        print(r.stdout[-2000:], r.stderr[-2000:])
        raise SystemExit("no kernel trace")
    rows = list(csv.DictReader(open(files[0])))
    ev = sorted(((int(x["Start_Timestamp"]), int(x["End_Timestamp"]), x["Kernel_Name"].split("(")[0][:60]) for x in rows))
    # one pgd() call = the longest run of kernels without a host-side pause (> 2 ms) inside it
    segs, cur = [], [ev[0]]
    for i in range(1, len(ev)):
        if ev[i][0] - ev[i - 1][1] > 2_000_000:
            segs.append(cur); cur = []
        cur.append(ev[i])
    segs.append(cur)
    ev = max(segs, key=len)
    span = ev[-1][1] - ev[0][0]
    dur = sum(e - s for s, e, _ in ev)
    gaps = [(ev[i][0] - ev[i - 1][1], ev[i - 1][2], ev[i][2]) for i in range(1, len(ev))]
    pos = sum(g for g, _, _ in gaps if g > 0)
    lines = [f"kernels {len(ev)}  span {span / 1e6:.2f} ms  sum of durations {dur / 1e6:.2f} ms  sum of positive gaps {pos / 1e6:.2f} ms "
             f"({100 * pos / span:.2f} % of the span)  overlaps {sum(-g for g, _, _ in gaps if g < 0) / 1e6:.2f} ms"]
    edges = [0, 1000, 2000, 3000, 5000, 10000, 20000, 50000, 10 ** 9]
    for lo, hi in zip(edges, edges[1:]):
        sel = [g for g, _, _ in gaps if lo <= g < hi]
        lines.append(f"gap {lo / 1000:>5.0f}-{hi / 1000:<8.0f} us: {len(sel):5d} gaps, {sum(sel) / 1e6:7.3f} ms")
    lines.append(f"negative (next kernel starts before the previous one ends): {sum(1 for g, _, _ in gaps if g < 0)}")
    for g, a, b in sorted(gaps, reverse=True)[:10]:
        lines.append(f"  {g / 1000:8.1f} us  {a}  ->  {b}")
    # by predecessor class
    by = {}
    for g, a, _ in gaps:
        k = by.setdefault(a, [0, 0]); k[0] += 1; k[1] += max(g, 0)
    for a, (n, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:12]:
        lines.append(f"  after {a:60s} {n:5d} gaps  {t / 1e6:7.3f} ms  {t / n / 1000:6.2f} us each")
    open(os.path.join(OUT, "launch_gaps.log"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
    subprocess.run(["rm", "-rf", d])


if __name__ == "__main__":
    main()
