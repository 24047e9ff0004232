#!/bin/bash
# PMC counters of one config of tools/config_bench.py: bash tools/pmc_config.sh C4
CFG=${1:-C4}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$CFG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export SPP_SCALE=${SPP_SCALE:-0.25}
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/p1 -- python3 $REPO/tools/config_bench.py $CFG > $OUT/p1.log 2>&1 || exit 2
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR --output-format csv -d $OUT/p2 -- python3 $REPO/tools/config_bench.py $CFG > $OUT/p2.log 2>&1 || exit 3
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $OUT/p3 -- python3 $REPO/tools/config_bench.py $CFG > $OUT/p3.log 2>&1 || exit 4
timeout -s KILL 200 rocprofv3 --pmc TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $OUT/p4 -- python3 $REPO/tools/config_bench.py $CFG > $OUT/p4.log 2>&1 || echo "p4 failed"
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "render_kernel" in row["Kernel_Name"]:
            acc[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
            acc[row["Kernel_Name"][:60]]["_dur_ns"].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
            acc[row["Kernel_Name"][:60]]["_vgpr"] = [float(row["VGPR_Count"])]
for sig, cs in acc.items():
    print("kernel", sig)
    for k in sorted(cs):
        v = cs[k]
        print(f"   {k:28s} {sum(v)/len(v):.4g}")
    m = lambda k: sum(cs[k])/len(cs[k])
    if "SQ_THREAD_CYCLES_VALU" in cs: print(f"   lane util = {m('SQ_THREAD_CYCLES_VALU')/(m('SQ_ACTIVE_INST_VALU')*64):.3f}")
PY
