#!/usr/bin/env python3
"""Registers, spills and scratch of every kernel, from the code-object metadata of a hipcc -S listing (gfx950).

  for f in aws-c-compression_amd/csrc/hip/*_kernels.hip aws-c-compression_amd/csrc/hip/decode_launch.hip; do
    hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S -Iinclude -Iinclude/compat $f -o /tmp/$(basename $f).s
  done; cat /tmp/*.hip.s > /tmp/k.s
  python profiles/tools/spill_census.py /tmp/k.s [--all]

Prints the kernels that have scratch memory or spilled registers (all kernels with --all) and a one-line summary.
A scalar register "spilled" with no scratch is parked in a lane of a vector register: no memory traffic."""
import re
import sys

txt = open(sys.argv[1]).read()
rows = re.findall(
    r"\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_spill_count:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)",
    txt, re.S)
worst = {"scratch": 0, "vgpr_spills": 0, "sgpr_spills": 0}
for name, scratch, sgpr_spills, vgprs, vgpr_spills in rows:
    short = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)[:60]
    if "--all" in sys.argv or int(scratch) or int(vgpr_spills) or int(sgpr_spills):
        print("%-60s vgprs %3s  scratch %4s B  vector spills %3s  scalar spills %3s" % (short, vgprs, scratch, vgpr_spills, sgpr_spills))
    worst["scratch"] = max(worst["scratch"], int(scratch))
    worst["vgpr_spills"] = max(worst["vgpr_spills"], int(vgpr_spills))
    worst["sgpr_spills"] = max(worst["sgpr_spills"], int(sgpr_spills))
print("%d kernels: most scratch %d B, most vector-register spills %d, most scalar-register spills %d" % (
    len(rows), worst["scratch"], worst["vgpr_spills"], worst["sgpr_spills"]))
