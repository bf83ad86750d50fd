#!/bin/bash
# usage: pmc_run.sh <tag> "<counters...>" -- <command...>   (separate rocprofv3 --pmc pass, csv output)
tag=$1; ctrs=$2; shift 3
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
( cd /tmp && rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -o pmc -- "$@" ) > $out/log.txt 2>&1
python - <<PY
import csv, glob, collections
fs = glob.glob("$out/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in fs:
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "?")[:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "$ctrs".split()[0]: cnt[k] += 1
for k in sorted(agg, key=lambda k: -sum(agg[k].values()))[:12]:
    n = max(cnt[k], 1)
    print(k, "launches", n, {c: round(v / n, 1) for c, v in agg[k].items()})
PY
