#!/bin/bash
# counters for the kernels of tools/ab_bench.py variants: bash tools/pmc_ab.sh libA.so libB.so
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_ab
rm -rf $OUT; mkdir -p $OUT
LIBS=""
for l in "$@"; do LIBS="$LIBS $REPO/$l"; done
cd /tmp && export TMPDIR=/tmp
SPP=${SPP:-64} timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/p1 -- python3 $REPO/tools/ab_bench.py $LIBS > $OUT/p1.log 2>&1 || exit 2
SPP=${SPP:-64} timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/p2 -- python3 $REPO/tools/ab_bench.py $LIBS > $OUT/p2.log 2>&1 || exit 3
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "render_kernel" in row["Kernel_Name"]:
            # dispatch order alternates libs: key by dispatch parity is fragile; key by VGPR/LDS signature + order
            acc[(row["VGPR_Count"], row["LDS_Block_Size"], row["SGPR_Count"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
for sig, cs in acc.items():
    print("kernel signature VGPR/LDS/SGPR =", sig)
    for k in sorted(cs):
        v = cs[k]
        print(f"   {k:24s} {sum(v)/len(v):.4g}")
    if "SQ_INSTS_VALU" in cs and "SQ_THREAD_CYCLES_VALU" in cs:
        m = lambda k: sum(cs[k])/len(cs[k])
        print(f"   lane util = {m('SQ_THREAD_CYCLES_VALU')/(m('SQ_ACTIVE_INST_VALU')*64):.3f}")
PY
