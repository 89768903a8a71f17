#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table from hipcc -Rpass-analysis=kernel-resource-usage.
usage: tools/kernel_resources.py cvpr2020_manet_amd/csrc/global_match.hip [name filter]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
           "-fno-fast-math", "-fno-slp-vectorize", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "cvpr2020_manet_amd", "csrc"),
           "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = None
    rows = {}
    for line in err.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = subprocess.run(["c++filt", m.group(1)], capture_output=True,
                                 text=True).stdout.strip()
            cur = re.sub(r"\(anonymous namespace\)::|void ", "", cur)
            cur = re.sub(r"\(.*", "", cur)
            rows[cur] = {}
            continue
        m = re.search(r"remark: +(.*?): (\d+) \[-Rpass", line)
        if m and cur:
            rows[cur][m.group(1).strip()] = int(m.group(2))
    print("%-60s %5s %5s %6s %6s %8s %4s" % ("kernel", "VGPR", "AGPR", "SGPR", "spill", "scratch", "occ"))
    for k, v in rows.items():
        if flt and flt not in k:
            continue
        print("%-60s %5d %5d %6d %6d %8d %4d" % (k[:60], v.get("VGPRs", -1), v.get("AGPRs", -1), v.get("TotalSGPRs", -1),
                                               v.get("VGPRs Spill", -1), v.get("ScratchSize [bytes/lane]", -1),
                                               v.get("Occupancy [waves/SIMD]", -1)))


if __name__ == "__main__":
    main()
