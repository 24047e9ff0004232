#!/usr/bin/env python3
"""Summarise the rocprofv3 CSVs written by tools/profile_bench.sh into profiles/<tag>_summary.md and
profiles/traffic.json (HBM bytes per launch of the render kernel; FETCH_SIZE doubled per the gfx950
correction of MI355X_MICROARCH.md §HBM — the render kernel's reads are 16-B-per-lane or scalar)."""
import csv, glob, json, os, sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
lines = [f"# rocprofv3 summary {tag} (python3 bench.py --steps 5 --warmup 1, 1 x MI355X)\n"]

def find(pattern):
    r = glob.glob(os.path.join(src, pattern), recursive=True)
    return r[0] if r else None

st = find("trace/**/*kernel_stats.csv")
if st:
    lines.append("## kernel stats (--kernel-trace --stats)\n")
    lines.append("| kernel | calls | total ns | avg ns | % |\n|---|---|---|---|---|")
    for row in csv.DictReader(open(st)):
        lines.append(f"| {row['Name'][:70]} | {row['Calls']} | {row['TotalDurationNs']} | {float(row['AverageNs']):.0f} | {row['Percentage']} |")
    lines.append("")
def code_object_resources(kernel_name):
    """VGPRs / SGPRs / static LDS / scratch of a kernel from the code object's own metadata (the amdhsa.kernels notes in the
    compiler's assembly, `make -C rustracer_amd/csrc asm`) — rocprofv3's VGPR_Count / LDS_Block_Size columns are allocation
    granules of the dispatch packet (they read 36 / 0 for a 72-VGPR kernel with 5 KB of dynamic LDS), not these."""
    import re, subprocess
    m = re.search(r"render_kernel<([^>]*)>", kernel_name)
    if not m:
        return None
    mangled = "_Z13render_kernelI" + "".join(("Lb1E" if a.strip() == "true" else "Lb0E" if a.strip() == "false" else f"Li{a.strip()}E") for a in m.group(1).split(",")) + "Ev12KernelParams"
    asm = os.path.join(root, "rustracer_amd", "csrc", "rt_kernels.s")
    if not os.path.exists(asm):
        subprocess.run(["make", "-C", os.path.join(root, "rustracer_amd", "csrc"), "asm"], capture_output=True)
    if not os.path.exists(asm):
        return None
    text = open(asm).read()
    at = text.find(f".name:           {mangled}\n")
    if at < 0:
        return None
    start = text.rfind("  - .agpr_count", 0, at)
    block = text[start:text.find("  - .agpr_count", at) if text.find("  - .agpr_count", at) > 0 else len(text)]
    g = lambda key: int(re.search(rf"\.{key}:\s+(\d+)", block).group(1))
    return {"vgpr": g("vgpr_count"), "sgpr": g("sgpr_count"), "static_lds": g("group_segment_fixed_size"), "scratch": g("private_segment_fixed_size"), "mangled": mangled}


kt = find("trace/**/*kernel_trace.csv")
render_ns = []
if kt:
    for row in csv.DictReader(open(kt)):
        if "render_kernel<false" in row["Kernel_Name"]:
            render_ns.append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
            meta = row
    if render_ns:
        res = code_object_resources(meta["Kernel_Name"])
        wg = int(meta.get("Workgroup_Size", meta.get("Workgroup_Size_X", 256)) or 256)
        # dynamic LDS of the bench launch (rt_launch_render: scan scene of 4 spheres, no texture queue): (4 + 4*4 + 64) * 16 bytes
        dyn_lds = int(os.environ.get("DYN_LDS", (4 + 16 + 64) * 16))
        lines.append(f"{meta['Kernel_Name'][:60]} (the timed kernel) dispatches: {len(render_ns)}, avg {sum(render_ns)/len(render_ns)/1e6:.3f} ms; "
                     f"grid {meta.get('Grid_Size', meta.get('Grid_Size_X'))}, workgroup {wg}")
        if res:
            alloc = (res["vgpr"] + 7) // 8 * 8
            by_vgpr = min(8, 512 // alloc)
            lds = res["static_lds"] + dyn_lds
            by_lds = (128 // -(-lds // 1280)) * (wg // 64) / 4.0 if lds else 8  # gfx950 allocates LDS in blocks of 1280 B, 128 per CU
            lines.append(f"resources from the code object ({res['mangled']}): {res['vgpr']} VGPRs (allocated {alloc}), {res['sgpr']} SGPRs, scratch {res['scratch']} B/lane, "
                         f"LDS {res['static_lds']} static + {dyn_lds} dynamic = {lds} B per workgroup => waves/SIMD: {by_vgpr} by VGPRs, {by_lds:.1f} by LDS => {min(by_vgpr, by_lds):.0f}\n")
        else:
            lines.append("(code-object metadata not found: run `make -C rustracer_amd/csrc asm`)\n")
counters = defaultdict(list)
for d in ("pmc_sq1", "pmc_sq2", "pmc_fetch", "pmc_write"):
    f = find(f"{d}/**/*counter_collection.csv")
    if not f:
        continue
    for row in csv.DictReader(open(f)):
        # the timed kernel only: bench.py also launches the instrumented variant render_kernel<true, ...> (untimed, for the ray
        # and flop counts), whose counters must not be averaged into the production kernel's
        if "render_kernel<false" in row["Kernel_Name"] or ("render_kernel" in row["Kernel_Name"] and "<" not in row["Kernel_Name"]):
            counters[row["Counter_Name"]].append(float(row["Counter_Value"]))
if counters:
    lines.append("## PMC, per dispatch of the timed kernel render_kernel<false, 0, false, true> (mean over its dispatches)\n")
    lines.append("| counter | value |\n|---|---|")
    mean = {k: sum(v) / len(v) for k, v in counters.items()}
    for k in sorted(mean):
        lines.append(f"| {k} | {mean[k]:.4g} |")
    lines.append("")
    if "SQ_INSTS_VALU" in mean and "SQ_WAVES" in mean:
        lines.append(f"VALU instructions per wave: {mean['SQ_INSTS_VALU']/mean['SQ_WAVES']:.0f}")
    if "SQ_INSTS_VALU" in mean and "SQ_BUSY_CYCLES" in mean and "GRBM_GUI_ACTIVE" in mean:
        lines.append(f"VALU issue: SQ_INSTS_VALU / (1024 SIMDs x kernel cycles) = {mean['SQ_INSTS_VALU']/1024/(mean['GRBM_GUI_ACTIVE']/8):.3f} wave-instructions per SIMD-cycle")
    if "SQ_ACTIVE_INST_VALU" in mean and "SQ_BUSY_CYCLES" in mean:
        lines.append(f"SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES = {mean['SQ_ACTIVE_INST_VALU']/mean['SQ_BUSY_CYCLES']:.3f}")
    if "SQ_THREAD_CYCLES_VALU" in mean and "SQ_ACTIVE_INST_VALU" in mean:
        lines.append(f"VALU lane utilisation (THREAD_CYCLES_VALU / (ACTIVE_INST_VALU*64)) = {mean['SQ_THREAD_CYCLES_VALU']/(mean['SQ_ACTIVE_INST_VALU']*64):.3f}")
    if "FETCH_SIZE" in mean or "WRITE_SIZE" in mean:
        fetch_b = mean.get("FETCH_SIZE", 0.0) * 1024 * 2  # KiB -> B, x2 gfx950 correction
        write_b = mean.get("WRITE_SIZE", 0.0) * 1024
        lines.append(f"HBM traffic per launch: fetch {fetch_b/1e6:.2f} MB (FETCH_SIZE x2), write {write_b/1e6:.2f} MB")
        with open(os.path.join(root, "profiles", "traffic.json"), "w") as f:
            json.dump({"tag": tag, "spp": 256, "hbm_bytes_per_launch": fetch_b + write_b, "fetch_bytes": fetch_b, "write_bytes": write_b,
                       "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH_SIZE doubled (gfx950), mean over render_kernel dispatches"}, f)
    if "GRBM_GUI_ACTIVE" in mean and render_ns:
        lines.append(f"effective clock ~ GRBM_GUI_ACTIVE/8/time = {mean['GRBM_GUI_ACTIVE']/8/(sum(render_ns)/len(render_ns))*1e3:.0f} MHz")
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
out = os.path.join(root, "profiles", f"{tag}_summary.md")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
