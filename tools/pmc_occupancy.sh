#!/bin/bash
# How full are the wave slots? bash tools/pmc_occupancy.sh C4 [lib.so]  (one PMC pass; run on the GPU box through gpurun)
CFG=${1:-C4}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
[ -n "$2" ] && export RTHIP_LIB=$REPO/$2
OUT=$REPO/gpurun_out/occ_$CFG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1 -- python3 $REPO/tools/config_bench.py $CFG > $OUT/p1.log 2>&1 || exit 2
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "render_kernel" in row["Kernel_Name"]:
            k = row["Kernel_Name"][:48]
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
            acc[k]["_dur_ns"].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
for sig, cs in acc.items():
    m = lambda k: sum(cs[k]) / len(cs[k])
    print("kernel", sig, {k: f"{m(k):.4g}" for k in sorted(cs)})
    # SQ_WAVE_CYCLES, SQ_BUSY_CU_CYCLES in quad-cycles; GRBM_GUI_ACTIVE in cycles
    gui = m("GRBM_GUI_ACTIVE")
    print(f"   waves per SIMD while the GPU is active: {4 * m('SQ_WAVE_CYCLES') / gui / 1024:.2f}   CU busy fraction: {4 * m('SQ_BUSY_CU_CYCLES') / gui / 256:.3f}")
PY
