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
    rows = [r for r in csv.DictReader(open(f)) if "render_kernel" in r["Kernel_Name"]]
    # the LAST dispatch of each kernel: the steady state of the launch order (pass 1 of a layout measures, pass 2 sorts, rthip_api.cpp)
    last = {}
    for r in rows:
        last[r["Kernel_Name"]] = max(last.get(r["Kernel_Name"], 0), int(r["Dispatch_Id"]))
    for row in rows:
        if int(row["Dispatch_Id"]) == last[row["Kernel_Name"]]:
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
    if "render_kernel<false" in sig and "FETCH_SIZE" in cs and "SQ_WAVE_CYCLES" in cs:
        import json
        dur = m("_dur_ns") * 1e-9
        cyc = dur * 2.35e9  # shader cycles of the launch at the measured ~2.35 GHz
        d = {"sample": "tools/pmc_config.sh, SPP_SCALE=$SPP_SCALE, round 3 kernel, last dispatch (steady launch order)", "kernel_ms": round(dur * 1e3, 2),
             "hbm_fetch_gb_per_s": round(m("FETCH_SIZE") * 1024 * 2 / dur / 1e9, 1), "hbm_fetch_frac_of_peak": round(m("FETCH_SIZE") * 1024 * 2 / dur / 8e12, 4),
             "l2_hit_rate": round(m("TCC_HIT_sum") / max(1.0, m("TCC_REQ_sum")), 3) if "TCC_REQ_sum" in cs else None,
             "l1_accesses_per_vmem_read": round(m("TCP_TOTAL_CACHE_ACCESSES_sum") / m("SQ_INSTS_VMEM_RD"), 1) if "TCP_TOTAL_CACHE_ACCESSES_sum" in cs else None,
             "valu_lane_utilisation": round(m("SQ_THREAD_CYCLES_VALU") / (m("SQ_ACTIVE_INST_VALU") * 64), 3),
             "valu_issue_slots_used": round(m("SQ_INSTS_VALU") / 1024 / cyc * 2, 3),
             "wave_cycles_waiting": round(m("SQ_WAIT_ANY") / m("SQ_WAVE_CYCLES"), 3), "valu_insts_per_wave": round(m("SQ_INSTS_VALU") / m("SQ_WAVES")),
             "vmem_reads_per_wave": round(m("SQ_INSTS_VMEM_RD") / m("SQ_WAVES")), "waves": int(m("SQ_WAVES")),
             "resident_waves_per_simd": round(4 * m("SQ_WAVE_CYCLES") / cyc / 1024, 2)}
        json.dump(d, open("$OUT/derived.json", "w"), indent=1)
        print("derived:", json.dumps(d))
PY
