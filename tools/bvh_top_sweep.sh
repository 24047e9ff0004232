#!/bin/bash
# C4 / C5 kernel time against the number of top-of-tree nodes staged in LDS (RTHIP_BVH_TOP) and the LDS stack levels
# (RTHIP_STACK_LDS): bash tools/bvh_top_sweep.sh > gpurun_out/<dir>/top_sweep.log
for cfg in "0 8" "31 8" "63 8" "63 6" "127 8" "127 6" "255 8" "255 4"; do
  set -- $cfg
  echo "== RTHIP_BVH_TOP=$1 RTHIP_STACK_LDS=$2"
  RTHIP_BVH_TOP=$1 RTHIP_STACK_LDS=$2 timeout -k 10 200 python tools/config_bench.py C4 C5 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.rstrip()); continue
    print(d['config'], 'kernel_ms', d['kernel_ms'], 'mrays_s', d['mrays_s'], 'node/ray', d['node_tests_per_ray'])
" || exit 1
done
