#!/bin/bash
# HBM traffic of the render kernel on the tools/ab_bench.py workload (env SPP, NOTEX, BAND, TARGET as there): bash tools/pmc_traffic.sh lib.so
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_traffic
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
REP=1 timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f -- python3 $REPO/tools/ab_bench.py $REPO/$1 > $OUT/f.log 2>&1 || exit 2
REP=1 timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w -- python3 $REPO/tools/ab_bench.py $REPO/$1 > $OUT/w.log 2>&1 || exit 3
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "render_kernel" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in acc.items():
    m = sum(v) / len(v)
    print(k, f"{m:.4g} KB per launch ->", f"{m * 1024 * (2 if k == 'FETCH_SIZE' else 1) / 1e6:.1f} MB" + (" (x2: gfx950)" if k == "FETCH_SIZE" else ""))
PY
