#!/bin/bash
# HBM-side traffic per kernel launch of the bench command: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE;
# --kernel-trace only), summarised into gpurun_out/pmc_hbm_traffic.json (copy to profiles/ after a GPU trip).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$c
  rm -rf $out; mkdir -p $out
  ( cd /tmp && timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out -o pmc -- \
      python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline ) > $out/log.txt 2>&1
done
python - <<'PY'
import csv, glob, collections, json, os, re
root = os.environ["GRAFT_REPO_ROOT"]
agg = collections.defaultdict(lambda: {"launches": 0, "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0})
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{root}/gpurun_out/pmc_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c:
                continue
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
            agg[k][c] += float(r["Counter_Value"])
            if c == "FETCH_SIZE":
                agg[k]["launches"] += 1
rows = {k: {"launches": v["launches"], "FETCH_SIZE": round(v["FETCH_SIZE"] / max(v["launches"], 1), 1),
            "WRITE_SIZE": round(v["WRITE_SIZE"] / max(v["launches"], 1), 1)} for k, v in agg.items() if v["launches"]}
gemm = {k: v for k, v in agg.items() if "gemm_bf16" in k or "splitk_reduce" in k}
logical = sum(v["launches"] for k, v in gemm.items() if "256p" in k or "256q" in k or "gemm_bf16_nt_256_kernel" in k)
total = sum((2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0 for v in gemm.values())
big = dict(sorted(rows.items(), key=lambda kv: -(2 * kv[1]["FETCH_SIZE"] + kv[1]["WRITE_SIZE"]) * kv[1]["launches"])[:16])
out = {
    "what": "HBM-side traffic per kernel launch from rocprofv3 PMC passes (separate --pmc FETCH_SIZE and --pmc WRITE_SIZE "
            "runs, --kernel-trace only), bench.py --steps 1 --warmup 0 (ViT-L/14 bf16 B=128, 10-step PGD + the e0 forward)",
    "corrections": "counter units are KiB; FETCH_SIZE is DOUBLED for these wide (16 B/lane) streaming reads as "
                   "MI355X_MICROARCH.md section HBM prescribes for gfx950; Infinity-Cache hits are included in the counter, "
                   "so this is fabric traffic, an upper bound on DRAM traffic",
    "per_launch_raw_kb": big,
    "gemm_logical_launches": logical,
    "gemm_bytes_per_logical_launch": round(total / max(logical, 1)),
    "note": "gemm_bytes_per_logical_launch = sum over every gemm_bf16 / splitk_reduce kernel of launches x (2*FETCH_SIZE + "
            "WRITE_SIZE) x 1024, divided by the number of persistent-kernel launches (a logical GEMM = its 256x256 launch; the few-row "
            "class-token-tail launches are included in the numerator)",
}
json.dump(out, open(f"{root}/gpurun_out/pmc_hbm_traffic.json", "w"), indent=1)
print(json.dumps({k: out[k] for k in ("gemm_logical_launches", "gemm_bytes_per_logical_launch")}))
PY
