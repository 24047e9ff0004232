#!/bin/bash
# kernel trace of the band-scaling check: how a 1/8-frame sliced launch splits into render / resolve time
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_band
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/band_scaling.py > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob, collections
rows = list(csv.DictReader(open(glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
by = collections.defaultdict(list)
for r in rows:
    by[(r["Kernel_Name"][:40], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size"), r.get("Grid_Size_Y"))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print(k, len(v), "avg us %.1f" % (sum(v) / len(v) / 1e3), "min %.1f" % (min(v) / 1e3))
PY
