#!/bin/bash
# rocprofv3 passes over bench.py (run on the GPU box through gpurun): kernel trace + stats, then PMC passes
# (each in a run of its own; never combined with --sys-trace etc.). Outputs under gpurun_out/prof_<tag>/.
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-parity --no-configs"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/pmc_sq1 -- $BENCH > $OUT/pmc_sq1.log 2>&1 || exit 2
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU --output-format csv -d $OUT/pmc_sq2 -- $BENCH > $OUT/pmc_sq2.log 2>&1 || exit 3
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1 || exit 4
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1 || exit 5
echo done
