#!/usr/bin/env python3
"""Measurement helpers of bench.py that are NOT part of the timed region (moved out of bench.py in round 6 so that the
headline harness reads top to bottom): host description, the HBM-class roofline objects, the in-run rocprofv3 PMC passes,
the rocm-smi clock / power sampler and the NUMA binding of a rank.  Nothing here touches the oracle or the timed loop."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def host_cpu_info():
    """(model string, physical cores, hardware threads) of this host from /proc/cpuinfo."""
    model, cores = "unknown", set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core = line.split(":", 1)[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    return model, (len(cores) or (os.cpu_count() or 1)), os.cpu_count() or 1


PEAK_HBM_TBPS = 8.0         # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md (6.3 TB/s achievable)


def hbm_classes(prof: dict, cfg, B: int, traffic: dict | None = None) -> dict:
    """Roofline objects of the HBM-bound kernel classes of one profiled pgd() call; `traffic` (bytes per launch, fabric side:
    2 x FETCH_SIZE + WRITE_SIZE of the in-run PMC passes) when it was measured, else null."""
    S = (cfg.image_size // cfg.patch) ** 2 + 1
    per_launch = {"attn_fwd": 4.0 * B * cfg.heads * S * 64 * 2, "attn_bwd": 8.0 * B * cfg.heads * S * 64 * 2}
    out = {}
    for k in ("attn_fwd", "attn_bwd", "layernorm_fwd", "layernorm_bwd"):
        v = prof.get(k)
        if not v or v["ms"] <= 0 or not v["launches"]:
            continue
        nbytes = per_launch[k] * v["launches"] if k in per_launch else v["bytes"]
        tbps = nbytes / (v["ms"] * 1e-3) / 1e12
        out[k] = {"bound": "hbm", "achieved": tbps, "peak": PEAK_HBM_TBPS, "unit": "TB/s", "frac": tbps / PEAK_HBM_TBPS,
                  "bytes_per_launch": nbytes / v["launches"], "avg_launch_us": 1e3 * v["ms"] / v["launches"],
                  "launches": v["launches"], "ms_per_step": round(v["ms"], 3),
                  "traffic": (traffic or {}).get(k)}
    return out


def pmc_in_run(argv_child, timeout_s=240):
    """`roofline.traffic` and the in-pipeline matrix-pipe figures measured IN THIS RUN, on this box and build: PMC counters cannot
    be read from inside the process, so rank 0 runs this same script under `rocprofv3 --pmc` as a child (one counter group per
    pass, --kernel-trace only - no sys/hip/hsa trace domains next to --pmc; the child times one pgd() call + the e0 forward,
    --no-roofline --no-cpu-baseline --no-pmc) while the parent's stream is idle, and sums the counters per kernel like
    scripts/pmc_traffic.sh / pmc_pipeline.sh.  FETCH_SIZE is DOUBLED (MI355X_MICROARCH.md, HBM section: gfx950 reports half the
    bytes of wide coalesced reads), both counters are KiB.  Returns (traffic_bytes_per_gemm_launch, info, pmc) or raises."""
    import collections
    import csv
    import glob
    import re
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        raise RuntimeError("rocprofv3 not found")
    passes = [("FETCH_SIZE",), ("WRITE_SIZE",),
              ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE")]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.Counter()
    t_begin = time.time()
    root = tempfile.mkdtemp(prefix="rvlm_pmc_", dir="/tmp")
    try:
        for i, ctrs in enumerate(passes):
            out = os.path.join(root, f"p{i}")
            os.makedirs(out)
            env = dict(os.environ, TMPDIR="/tmp", RVLM_BENCH_CHILD="1")
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_PORT", "RVLM_SELF_LAUNCHED"):
                env.pop(k, None)
            cmd = [exe, "--pmc", *ctrs, "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "pmc", "--",
                   sys.executable, os.path.join(ROOT, "bench.py"), *argv_child]
            left = timeout_s - (time.time() - t_begin)
            if left < 20:
                raise RuntimeError("time budget of the PMC passes exhausted")
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=left)
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                raise RuntimeError(f"rocprofv3 pass {ctrs[0]} failed (rc {r.returncode}): {(r.stderr or r.stdout)[-300:]}")
            for f in files:
                for row in csv.DictReader(open(f)):
                    k = re.sub(r"^rvlm::", "", re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "").strip())
                    agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
                    if row["Counter_Name"] == "FETCH_SIZE":
                        launches[k] += 1
    finally:
        shutil.rmtree(root, ignore_errors=True)
    gemm = {k: v for k, v in agg.items() if "gemm_bf16" in k or "splitk_reduce" in k}
    logical = sum(launches[k] for k in gemm if "256p" in k)
    if not logical:
        raise RuntimeError("no persistent-GEMM launches in the counter files")
    total = sum((2.0 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024.0 for v in gemm.values())
    info = {"measured_in_run": True, "seconds": round(time.time() - t_begin, 1), "gemm_logical_launches": logical,
            "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over a child run of this script (one pgd() call + the e0 forward) on "
                   "this box; bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB summed over every bf16 GEMM kernel / persistent-kernel "
                   "launches; fabric side: Infinity-Cache hits included (profiles/r04_pmc_l2.json separates them by read latency)"}
    tot_active = sum(v.get("GRBM_GUI_ACTIVE", 0.0) for v in agg.values()) or 1.0
    pmc = {"measured_in_run": True, "source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT "
                                              "SQ_LDS_IDX_ACTIVE pass over the same child run",
           "mfma_busy": {}, "lds_conflict_share": {}}
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0.0)):
        if v.get("GRBM_GUI_ACTIVE", 0.0) < 0.004 * tot_active:
            continue
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v:
            pmc["mfma_busy"][k] = round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (v["GRBM_GUI_ACTIVE"] / 8.0), 4)
        if v.get("SQ_LDS_IDX_ACTIVE"):
            pmc["lds_conflict_share"][k] = round(v.get("SQ_LDS_BANK_CONFLICT", 0.0) / v["SQ_LDS_IDX_ACTIVE"], 4)
    # fabric-side bytes per launch of the HBM-bound classes, from the same two passes (for roofline.hbm_classes[*].traffic)
    cls = {"attn_fwd": "attn_fwd_odd_kernel", "attn_bwd": "attn_bwd_fused_kernel", "layernorm_fwd": "layernorm_fwd8_kernel",
           "layernorm_bwd": "layernorm_bwd8_kernel"}
    pmc["hbm_class_traffic"] = {}
    for name, pat in cls.items():
        ks = [k for k in agg if pat in k and launches[k]]
        if ks:
            n = sum(launches[k] for k in ks)
            pmc["hbm_class_traffic"][name] = round(sum((2.0 * agg[k].get("FETCH_SIZE", 0.0) + agg[k].get("WRITE_SIZE", 0.0)) * 1024.0 for k in ks) / n)
    return round(total / logical), info, pmc


class ClockSampler:
    """Shader clock / socket power of one GPU while the timed region runs: `rocm-smi --showclocks --showpower` polled from
    a thread (~10 Hz; each call costs the HOST ~60 ms, nothing on the GPU) - run over extra, untimed calls of the workload.  The 2.5 PFLOP/s peak assumes 2.4 GHz; under the
    ~1.4 kW socket cap a dense MFMA loop holds 1.7-1.9 GHz, so `roofline.frac` is also reported against
    the peak at the clock the chip actually held."""

    def __init__(self, device_index: int):
        import threading
        self.dev, self.samples, self.stop, self.thread = device_index, [], False, None
        self._threading = threading

    def _run(self):
        import re
        import subprocess
        while not self.stop:
            try:
                o = subprocess.run(["rocm-smi", "-d", str(self.dev), "--showpower", "--showclocks"], capture_output=True,
                                   text=True, timeout=5).stdout
                p = re.search(r"Power \(W\):\s*([\d.]+)", o)
                c = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", o)
                if c:
                    self.samples.append((float(p.group(1)) if p else -1.0, int(c.group(1))))
            except Exception:
                pass
            time.sleep(0.03)

    def __enter__(self):
        self.thread = self._threading.Thread(target=self._run, daemon=True)
        self.thread.start()
        return self

    def __exit__(self, *a):
        self.stop = True
        self.thread.join(timeout=10)

    def summary(self):
        cs = [c for _, c in self.samples if c > 0]
        ps = [p for p, _ in self.samples if p > 0]
        if not cs:
            return None
        return {"sclk_mhz_mean": sum(cs) / len(cs), "sclk_mhz_min": min(cs), "sclk_mhz_max": max(cs), "samples": len(cs),
                "socket_power_w_mean": (sum(ps) / len(ps)) if ps else None,
                "source": "rocm-smi --showclocks --showpower polled during up to 3 extra pgd() calls AFTER the timed region "
                          "(the timed region itself runs without a poller)"}


def bind_rank_to_numa(local_rank: int, local_world: int):
    """One process per GPU on a 2-socket host: keep each rank's host threads (launch loop, RCCL proxy) on the cores of
    its GPU's NUMA node - sysfs numa_node of the GPU's PCI function when readable, else an even split of the cores.
    Returns a description for the JSON line."""
    if local_world <= 1:
        return None
    try:
        node = -1
        p = torch.cuda.get_device_properties(local_rank)
        bdf = f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        path = f"/sys/bus/pci/devices/{bdf}/numa_node"
        if os.path.exists(path):
            node = int(open(path).read().strip())
        cpus = None
        if node >= 0 and os.path.exists(f"/sys/devices/system/node/node{node}/cpulist"):
            cpus = set()
            for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
                a, _, b = part.partition("-")
                cpus.update(range(int(a), int(b or a) + 1))
            how = f"numa node {node} of {bdf}"
        if not cpus:
            n = os.cpu_count() or 1
            per = max(n // local_world, 1)
            cpus = set(range(local_rank * per, min(n, (local_rank + 1) * per)))
            how = f"even split ({per} cpus per rank)"
        os.sched_setaffinity(0, cpus)
        return f"{len(cpus)} cpus, {how}"
    except Exception as e:                      # affinity is a performance hint only
        return f"unbound ({type(e).__name__})"


