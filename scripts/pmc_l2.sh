#!/bin/bash
# Where the GEMMs' fabric-side reads come from: rocprofv3 --pmc passes with the L2 (TCC) counters over
# the bench command (bench.py --steps 1 --warmup 0), --kernel-trace only, one counter group per pass.
#   pass 1: TCC_HIT / TCC_MISS / TCC_REQ / TCC_READ            -> L2 hit rate, reads per request
#   pass 2: TCC_EA0_RDREQ / _LEVEL / _DRAM_CREDIT_STALL        -> fabric read requests, their AVERAGE LATENCY at the L2's
#                                                                 memory-side port (LEVEL / RDREQ, in L2 clocks) and back-pressure
#   pass 3: TCC_EA0_RDREQ_32B / _64B / _128B                   -> request sizes (bytes that crossed the fabric)
# The Infinity Cache (MALL) is memory-side: no per-kernel hit counter is exposed; what separates a MALL hit from a DRAM
# access here is the read latency at the EA port, calibrated in the same run against a kernel that only streams
# (layernorm_fwd8: 135 MB read once, nothing to re-use).  Summary: gpurun_out/pmc_l2.json.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
i=0
for ctrs in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
            "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum GRBM_GUI_ACTIVE" \
            "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1)); out=$GRAFT_REPO_ROOT/gpurun_out/pmc_l2_$i
  rm -rf $out; mkdir -p $out
  ( cd /tmp && timeout 600 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -o pmc -- \
      python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline ${BENCH_ARGS:-} ) > $out/log.txt 2>&1
  echo "pass $i rc=$? $(grep -ci error $out/log.txt) error lines"
done
python - <<'PY' > gpurun_out/pmc_l2.json
import collections, csv, glob, json, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); launch = collections.Counter()
for i in (1, 2, 3):
    for f in glob.glob(f"gpurun_out/pmc_l2_{i}/**/*counter_collection.csv", recursive=True):
        first = None
        for r in csv.DictReader(open(f)):
            k = re.sub(r"^rvlm::", "", re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip())
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            first = first or r["Counter_Name"]
            if i == 1 and r["Counter_Name"] == first:
                launch[k] += 1
out = {"what": "rocprofv3 --pmc TCC passes over bench.py --steps 1 --warmup 0 (ViT-L/14 bf16, B = 128: e0 forward + one 10-step pgd())",
       "derived": {"l2_hit_rate": "TCC_HIT / (TCC_HIT + TCC_MISS)",
                   "ea_read_latency_l2clk": "TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ (average occupancy per request = latency at the fabric port)",
                   "ea_read_mb_per_launch": "(32 x RDREQ_32B + 64 x RDREQ_64B + 128 x RDREQ_128B) / launches / 1e6, when the size counters are "
                                            "populated (else 64 B x RDREQ, the guide's tally)",
                   "dram_credit_stall_per_req": "TCC_EA0_RDREQ_DRAM_CREDIT_STALL / TCC_EA0_RDREQ"},
       "kernels": {}}
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("TCC_REQ_sum", 0.0)):
    n = max(launch[k], 1)
    if v.get("TCC_REQ_sum", 0) < 1e6:
        continue
    e = {"launches": launch[k], "raw_per_launch": {c: round(x / n, 1) for c, x in sorted(v.items())}}
    if v.get("TCC_HIT_sum") is not None and (v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0)) > 0:
        e["l2_hit_rate"] = round(v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"]), 4)
    if v.get("TCC_EA0_RDREQ_sum"):
        e["ea_read_latency_l2clk"] = round(v.get("TCC_EA0_RDREQ_LEVEL_sum", 0.0) / v["TCC_EA0_RDREQ_sum"], 1)
        e["dram_credit_stall_per_req"] = round(v.get("TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum", 0.0) / v["TCC_EA0_RDREQ_sum"], 3)
        sized = 32 * v.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * v.get("TCC_EA0_RDREQ_64B_sum", 0) + 128 * v.get("TCC_EA0_RDREQ_128B_sum", 0)
        e["ea_read_mb_per_launch"] = round((sized if sized > 0 else 64 * v["TCC_EA0_RDREQ_sum"]) / n / 1e6, 1)
    out["kernels"][k] = e
json.dump(out, __import__("sys").stdout, indent=1)
PY
python - <<'PY'
import json
d = json.load(open("gpurun_out/pmc_l2.json"))
for k, e in list(d["kernels"].items())[:14]:
    print(k[:44].ljust(44), e["launches"], "hit", e.get("l2_hit_rate"), "lat", e.get("ea_read_latency_l2clk"), "stall/req", e.get("dram_credit_stall_per_req"), "MB", e.get("ea_read_mb_per_launch"))
PY
find gpurun_out/pmc_l2_* -name "*.csv" -size +8M -delete 2>/dev/null
