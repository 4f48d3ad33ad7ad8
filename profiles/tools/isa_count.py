#!/usr/bin/env python3
"""Static instruction census of one kernel in a hipcc -S listing (gfx950).

  for f in aws-c-compression_amd/csrc/hip/*_kernels.hip aws-c-compression_amd/csrc/hip/decode_launch.hip; do
    hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S -Iinclude -Iinclude/compat $f -o /tmp/$(basename $f).s
  done; cat /tmp/*.hip.s > /tmp/k.s
  python profiles/tools/isa_count.py /tmp/k.s enc_pack_wave_kernelILj4 [--blocks]

Prints the number of vector-ALU, scalar, LDS and vector-memory instructions of the kernel
(whole body, and per basic block with --blocks so that the hot loop can be read off), and the
opcode histogram of the vector-ALU ones.  A static count: loop bodies count once.
"""
import collections
import re
import sys


def kernel_body(path, needle):
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^_Z\S*:", l) and needle in l and start is None:
            start = i
        elif start is not None and l.startswith("\t.end_amdhsa_kernel") or (start is not None and l.startswith(".Lfunc_end")):
            return lines[start:i]
    raise SystemExit("kernel not found: " + needle)


def classify(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def main():
    path, needle = sys.argv[1], sys.argv[2]
    per_block = "--blocks" in sys.argv
    body = kernel_body(path, needle)
    total = collections.Counter()
    hist = collections.Counter()
    block, blocks = "entry", collections.OrderedDict()
    for l in body:
        m = re.match(r"^(\.LBB\S+):", l)
        if m:
            block = m.group(1)
            continue
        m = re.match(r"^\t([a-z_0-9]+)", l)
        if not m or l.startswith("\t."):
            continue
        op = m.group(1)
        c = classify(op)
        total[c] += 1
        blocks.setdefault(block, collections.Counter())[c] += 1
        if c == "valu":
            hist[op] += 1
    print("kernel", needle, dict(total))
    if per_block:
        for b, c in blocks.items():
            if sum(c.values()) >= 20:
                print("  %-14s %s" % (b, dict(c)))
    print("valu opcodes:", ", ".join("%s %d" % kv for kv in hist.most_common(40)))


if __name__ == "__main__":
    main()
