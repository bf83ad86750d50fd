#!/usr/bin/env python3
"""Per-kernel PMC summary of rocprofv3 --pmc passes over bench.py (scripts/pmc_pipeline.sh).
Counters are summed over a kernel's launches (one counter group per pass).  Derived values, the same readings as
profiles/r02_pmc_gemm256p_cube8k.json:
  mfma_busy          = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs): share of the kernel's cycles in
                       which a SIMD's matrix pipe is busy, averaged over the chip's 1024 SIMDs
  lds_conflict_share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (share of LDS-active cycles spent in bank conflicts)
  wait_any_share     = SQ_WAIT_ANY / SQ_WAVE_CYCLES
"""
import collections
import csv
import glob
import json
import re
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
launch = collections.Counter()
for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        first = None
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
            k = re.sub(r"^rvlm::", "", k)
            c = r["Counter_Name"]
            agg[k][c] += float(r["Counter_Value"])
            first = first or c
            if c == first and d == sys.argv[1]:
                launch[k] += 1
out = {"what": "rocprofv3 --pmc passes over `bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline` (ViT-L/14 bf16, B=128: the e0 "
               "forward + one 10-step pgd() call); counters summed over each kernel's launches, one counter group per pass",
       "derived": {"mfma_busy": "SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs)",
                   "lds_conflict_share": "SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE",
                   "wait_any_share": "SQ_WAIT_ANY / SQ_WAVE_CYCLES"},
       "kernels": {}}
tot = sum(v.get("GRBM_GUI_ACTIVE", 0.0) for v in agg.values()) or 1.0
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0.0)):
    if v.get("GRBM_GUI_ACTIVE", 0.0) < 0.002 * tot:
        continue
    e = {"launches": launch[k], "raw": {c: round(x, 1) for c, x in sorted(v.items())}}
    if v.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
        e["mfma_busy"] = round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (v["GRBM_GUI_ACTIVE"] / 8.0), 4)
    if v.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_conflict_share"] = round(v.get("SQ_LDS_BANK_CONFLICT", 0.0) / v["SQ_LDS_IDX_ACTIVE"], 4)
    if v.get("SQ_WAVE_CYCLES") and v.get("SQ_WAIT_ANY") is not None:
        e["wait_any_share"] = round(v.get("SQ_WAIT_ANY", 0.0) / v["SQ_WAVE_CYCLES"], 4)
    e["share_of_gpu_active"] = round(v.get("GRBM_GUI_ACTIVE", 0.0) / tot, 4)
    out["kernels"][k] = e
json.dump(out, sys.stdout, indent=1)
