#!/usr/bin/env python3
"""Instruction mix per basic block of one kernel in a hipcc -S listing (no GPU needed).
    hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 --cuda-device-only -S x.hip -o x.s
    python tools/isa_blocks.py x.s 'pair_fast_tight_kernelILj1537ELb1E' [--min 8] [--dump LABEL]"""
import re
import sys

path, key = sys.argv[1], sys.argv[2]
minn = int(sys.argv[sys.argv.index("--min") + 1]) if "--min" in sys.argv else 8
dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(key) + r"\w*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
blocks, cur = [], ["entry", []]
for l in lines[start + 1:end]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append(cur)
        cur = [m.group(1), []]
        continue
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        continue
    cur[1].append(t)
    if t.startswith(("s_cbranch", "s_branch")):            # a branch ends a block even without a label behind it
        blocks.append(cur)
        cur = [cur[0] + "+", []]
blocks.append(cur)
print("%-14s %5s %5s %5s %5s %5s  %s" % ("block", "valu", "salu", "vmem", "lds", "lane", "notes"))
for name, ins in blocks:
    op = [i.split()[0] for i in ins]
    valu = sum(o.startswith("v_") for o in op)
    salu = sum(o.startswith("s_") for o in op)
    vmem = sum(o.startswith(("buffer_", "global_", "flat_", "scratch_")) for o in op)
    lds = sum(o.startswith("ds_") for o in op)
    lane = sum(o.startswith(("v_readlane", "v_writelane", "v_readfirstlane")) for o in op)
    notes = []
    if any(o.startswith("v_rcp_f64") for o in op):
        notes.append("rcp x%d" % sum(o.startswith("v_rcp_f64") for o in op))
    if any("buffer_load_ushort" in o or "buffer_load_short" in o for o in op):
        notes.append("gather x%d" % sum(("buffer_load_ushort" in o or "buffer_load_short" in o) for o in op))
    br = [i for i in ins if i.startswith("s_cbranch") or i.startswith("s_branch")]
    if br:
        notes.append(" ".join(b.split()[0][2:] + ">" + b.split()[1] for b in br))
    if len(ins) >= minn:
        print("%-14s %5d %5d %5d %5d %5d  %s" % (name, valu, salu, vmem, lds, lane, "; ".join(notes)))
    if dump == name:
        print("\n".join("    " + i for i in ins))
