#!/usr/bin/env python3
"""Static audit of a hipcc -S dump: for every kernel, the registers its instruction stream names against
what its kernel descriptor allocates (gfx90a+ unified file: arch VGPRs below .amdhsa_accum_offset, AGPRs above).

  hipcc --offload-arch=gfx950 ... --cuda-device-only -S -o kernels.s kernels.hip
  python tools/reg_audit.py kernels.s            (exit status 1 on any violation)

A wave that touches a register outside its allocation corrupts (or is corrupted by) whichever wave the SIMD
placed next to it -- silent, only at two or more waves per SIMD, timing dependent.  Also reported per kernel:
scratch bytes, v_accvgpr_* moves (VGPR spills into the accumulator half), SGPR-spill lane moves."""
import re
import sys


def regs(tok, cls):
    """Highest index of register class `cls` ('v', 'a', 's') named in operand text."""
    hi = -1
    for m in re.finditer(r"(?<![\w.])" + cls + r"(\d+)\b", tok):
        hi = max(hi, int(m.group(1)))
    for m in re.finditer(r"(?<![\w.])" + cls + r"\[(\d+):(\d+)\]", tok):
        hi = max(hi, int(m.group(2)))
    return hi


def main():
    lines = open(sys.argv[1]).read().split("\n")
    bad = 0
    i = 0
    bodies = {}
    while i < len(lines):
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", lines[i])
        if m:
            name, j, body = m.group(1), i + 1, []
            while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
                body.append(lines[j])
                j += 1
            bodies[name] = body
            i = j
        i += 1
    desc = {}
    for i, l in enumerate(lines):
        m = re.match(r"\s*\.amdhsa_kernel (\S+)", l)
        if m:
            d = {}
            j = i + 1
            while not lines[j].strip().startswith(".end_amdhsa_kernel"):
                mm = re.match(r"\s*\.amdhsa_(\w+)\s+(\S+)", lines[j])
                if mm:
                    d[mm.group(1)] = mm.group(2)
                j += 1
            desc[m.group(1)] = d
    print(f"{'kernel':64s} {'vmax':>4s} {'amax':>4s} {'nfv':>4s} {'aoff':>4s} {'smax':>4s} {'nfs':>4s} {'scr':>4s} {'accmov':>6s} {'lanemov':>7s}")
    for name, d in desc.items():
        body = bodies.get(name)
        if body is None:
            continue
        vmax = amax = smax = -1
        acc = lane = 0
        for l in body:
            t = l.split(";")[0].strip()
            if not t or t.startswith("."):
                continue
            ops = t.split(None, 1)
            if len(ops) < 2:
                continue
            vmax, amax, smax = max(vmax, regs(ops[1], "v")), max(amax, regs(ops[1], "a")), max(smax, regs(ops[1], "s"))
            acc += ops[0].startswith("v_accvgpr")
            lane += ops[0] in ("v_readlane_b32", "v_writelane_b32")
        nfv, aoff, nfs = int(d["next_free_vgpr"]), int(d["accum_offset"]), int(d["next_free_sgpr"])
        scr = int(d.get("private_segment_fixed_size", "0"))
        ok = vmax < aoff and (amax < 0 or aoff + amax < nfv) and (amax >= 0 or vmax < nfv) and smax < max(nfs, 1) + 0
        short = re.sub(r"^_ZN5genie12_GLOBAL__N_1\d+", "", name)[:64]
        print(f"{short:64s} {vmax:4d} {amax:4d} {nfv:4d} {aoff:4d} {smax:4d} {nfs:4d} {scr:4d} {acc:6d} {lane:7d}" + ("" if ok else "   <-- OUTSIDE ALLOCATION"))
        bad += not ok
    print("violations:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
