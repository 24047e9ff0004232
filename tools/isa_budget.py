"""Static VALU budget of a kernel from the compiler's assembly (make -C rustracer_amd/csrc asm): instructions per basic
block, by class. usage: python tools/isa_budget.py rustracer_amd/csrc/rt_kernels.s <mangled kernel name> [first_line last_line]"""
import re
import sys

TRANS = ("v_sqrt", "v_rsq", "v_rcp", "v_sin", "v_cos", "v_exp", "v_log")


def classify(op):
    if op.startswith(TRANS):
        return "trans"
    if op.startswith("v_cmp"):
        return "cmp"
    if op.startswith("v_cndmask"):
        return "select"
    if op.startswith(("v_mov", "v_accvgpr")):
        return "move"
    if op.startswith(("v_mul_lo", "v_mul_hi", "v_mad_u64", "v_mad_u32", "v_mul_u32")):
        return "imul"
    if op.startswith(("v_fma", "v_fmac", "v_mul_f32", "v_add_f32", "v_sub", "v_mac", "v_max", "v_min", "v_med", "v_fmaak", "v_fmamk")) and "u32" not in op and "i32" not in op:
        return "fp"
    if op.startswith("v_"):
        return "int/other"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("ds_", "global_", "flat_", "buffer_", "scratch_")):
        return "mem"
    return None


def main():
    path, kernel = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(kernel + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks, cur = [], None
    for i in range(start, end):
        l = lines[i]
        m = re.match(r"^(\.LBB\d+_\d+|; %bb\.\d+):?", l)
        if m or cur is None:
            cur = {"name": m.group(1) if m else "entry", "line": i - start + 1, "counts": {}, "depth": ""}
            d = re.search(r"Depth[= ](\d)", l)
            if d:
                cur["depth"] = d.group(1)
            blocks.append(cur)
            if m:
                continue
        t = l.strip().split()
        if not t or t[0].startswith((";", ".")):
            continue
        c = classify(t[0])
        if c:
            cur["counts"][c] = cur["counts"].get(c, 0) + 1
    cols = ["fp", "trans", "cmp", "select", "move", "imul", "int/other", "salu", "mem"]
    print("| block | line | VALU | " + " | ".join(cols) + " |")
    print("|---|---|---|" + "---|" * len(cols))
    tot = {c: 0 for c in cols}
    for b in blocks:
        valu = sum(v for k, v in b["counts"].items() if k not in ("salu", "mem"))
        if not b["counts"]:
            continue
        print(f"| {b['name']} | {b['line']} | {valu} | " + " | ".join(str(b["counts"].get(c, 0)) for c in cols) + " |")
        for c in cols:
            tot[c] += b["counts"].get(c, 0)
    print("| total | | " + str(sum(v for k, v in tot.items() if k not in ("salu", "mem"))) + " | " + " | ".join(str(tot[c]) for c in cols) + " |")


if __name__ == "__main__":
    main()
