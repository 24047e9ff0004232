#!/bin/bash
# A/B of library variants on configs of tools/config_bench.py: bash tools/ab_config.sh "C4 C5" build_ab/a.so build_ab/b.so ...
CFGS=$1; shift
for so in "$@"; do
  echo "== $so"
  RTHIP_LIB=$PWD/$so timeout -k 10 200 python tools/config_bench.py $CFGS 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.rstrip()); continue
    print(d['config'], 'kernel_ms', d['kernel_ms'], 'mrays_s', d['mrays_s'], 'node/ray', d['node_tests_per_ray'])
" || exit 1
done
