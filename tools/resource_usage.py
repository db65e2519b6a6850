#!/usr/bin/env python3
"""Registers, scratch (spill) bytes, occupancy and LDS of every kernel in csrc/, from hipcc's own report
(-Rpass-analysis=kernel-resource-usage; cross-compiles without a GPU).

    python tools/resource_usage.py [file.hip ...] [--md profiles/r03_kernel_resource_usage.md]
"""
import glob
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mpnn_amd", "csrc")
PAT = re.compile(r"Function Name: (\S+).*?SGPRs: (\d+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?"
                 r"Occupancy \[waves/SIMD\]: (\d+).*?LDS Size \[bytes/block\]: (\d+)", re.S)


def one(src):
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-fast-math",
                        "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"], capture_output=True, text=True)
    rows = []
    for m in PAT.finditer(r.stderr):
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        rows.append((os.path.basename(src), re.sub(r"^void ", "", name).split("(")[0]) + tuple(int(x) for x in m.groups()[1:]))
    return rows


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    md = sys.argv[sys.argv.index("--md") + 1] if "--md" in sys.argv else None
    if md in args:
        args.remove(md)
    srcs = args or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    with ThreadPoolExecutor(6) as ex:
        rows = [r for rs in ex.map(one, srcs) for r in rs]
    rows.sort(key=lambda r: (-r[5], r[0], r[1]))
    lines = ["| file | kernel | SGPR | VGPR | AGPR | scratch B/lane | waves/SIMD | LDS B/block |", "|---|---|---|---|---|---|---|---|"]
    for r in rows:
        lines.append("| %s | `%s` | %d | %d | %d | %d | %d | %d |" % r)
    print("\n".join(lines))
    if md:
        commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
        with open(md, "w") as f:
            f.write("# Kernel resource usage (`hipcc -Rpass-analysis=kernel-resource-usage`, gfx950, commit %s)\n\n"
                    "Scratch > 0 means spilled registers.  Sorted by scratch, then file.\n\n%s\n" % (commit, "\n".join(lines)))


if __name__ == "__main__":
    main()
