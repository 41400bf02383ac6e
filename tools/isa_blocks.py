#!/usr/bin/env python3
"""Static instruction mix per basic block of one kernel in a hipcc -S dump.

  hipcc --offload-arch=gfx950 ... --cuda-device-only -S -o kernels.s kernels.hip
  python tools/isa_blocks.py kernels.s 'match_table_kernelILi8E' [--min 8]

Prints, per label-delimited block: line range, VALU / SALU / LDS / VMEM / branch counts, and the source
lines (from `; kernels.hip:NNN` style comments when -g was used) if present.  Loops show up as a block
whose last branch targets an earlier label."""
import re
import sys


def main():
    path, pat = sys.argv[1], sys.argv[2]
    min_n = int(sys.argv[sys.argv.index("--min") + 1]) if "--min" in sys.argv else 1
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + pat + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks, cur = [], {"label": "entry", "first": start, "ins": []}
    for i in range(start + 1, end):
        l = lines[i].strip()
        if not l or l.startswith(";") or l.startswith("."):
            m = re.match(r"^(\.LBB\d+_\d+):", l)
            if m:
                blocks.append(cur)
                cur = {"label": m.group(1), "first": i, "ins": []}
            continue
        cur["ins"].append(l.split()[0] + " " + " ".join(l.split()[1:]))
    blocks.append(cur)
    order = {b["label"]: k for k, b in enumerate(blocks)}
    tot = dict(valu=0, salu=0, lds=0, vmem=0)
    print(f"{'block':12s} {'valu':>5s} {'salu':>5s} {'lds':>4s} {'vmem':>4s}  branches")
    for k, b in enumerate(blocks):
        ops = [x.split()[0] for x in b["ins"]]
        valu = sum(o.startswith("v_") for o in ops)
        salu = sum(o.startswith("s_") and not o.startswith(("s_waitcnt", "s_nop", "s_branch", "s_cbranch", "s_load", "s_barrier")) for o in ops)
        lds = sum(o.startswith("ds_") for o in ops)
        vmem = sum(o.startswith(("global_", "buffer_", "flat_", "scratch_")) for o in ops)
        for key, v in zip(tot, (valu, salu, lds, vmem)):
            tot[key] += v
        br = []
        for x in b["ins"]:
            m = re.search(r"(s_c?branch\w*)\s+(\.LBB\d+_\d+)", x)
            if m:
                back = order.get(m.group(2), 1 << 30) <= k
                br.append(("<-" if back else "->") + m.group(2).split("_")[-1])
        if len(ops) >= min_n:
            print(f"{b['label'].replace('.LBB', 'B'):12s} {valu:5d} {salu:5d} {lds:4d} {vmem:4d}  {' '.join(br)}")
    print("total", tot)


if __name__ == "__main__":
    main()
