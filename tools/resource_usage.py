#!/usr/bin/env python3
"""One line per gfx950 kernel of csrc/kernels.hip: registers, spills, scratch, occupancy.

  python tools/resource_usage.py [substring ...]      (runs `make resource-usage`; CPU only)
"""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "genie-smem_amd", "csrc")


def main():
    out = subprocess.run(["make", "-s", "-C", CSRC, "resource-usage"], capture_output=True, text=True)
    rows, cur = [], None
    for line in (out.stdout + out.stderr).splitlines():
        m = re.search(r"remark:\s+(.*?)\s+\[-Rpass", line)
        if not m:
            continue
        key, _, val = m.group(1).partition(":")
        key, val = key.strip(), val.strip()
        if key == "Function Name":
            cur = {"name": val}
            rows.append(cur)
        elif cur is not None:
            cur[key] = val
    names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True,
                           text=True).stdout.splitlines()
    print(f"{'kernel':78s} {'sgpr':>4s} {'vgpr':>4s} {'sSpl':>4s} {'vSpl':>4s} {'scr':>4s} {'occ':>3s}")
    for r, n in zip(rows, names):
        n = re.sub(r"\(genie::DevIndex.*|\(int const\*.*|\(unsigned long long\*.*", "", n).replace("genie::(anonymous namespace)::", "")
        n = n.replace("void ", "")
        if sys.argv[1:] and not any(s in n for s in sys.argv[1:]):
            continue
        print(f"{n[:78]:78s} {r.get('TotalSGPRs', '?'):>4s} {r.get('VGPRs', '?'):>4s} {r.get('SGPRs Spill', '?'):>4s} "
              f"{r.get('VGPRs Spill', '?'):>4s} {r.get('ScratchSize [bytes/lane]', '?'):>4s} {r.get('Occupancy [waves/SIMD]', '?'):>3s}")


if __name__ == "__main__":
    main()
