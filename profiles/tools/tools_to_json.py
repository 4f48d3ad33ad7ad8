#!/usr/bin/env python3
"""gpurun_out/<tag>/tools/*.txt (profiles/tools/run_tools.sh on the GPU box) -> profiles/<name>_tools.json: every
line a tool printed, with the numbers in it pulled out, so that the figures DESIGN.md quotes are tracked.

  usage: tools_to_json.py <tag> <name>     e.g.  r02b r02_b_round_end
"""
import glob
import json
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    tag, name = sys.argv[1], sys.argv[2]
    out = {"profile": name, "command": "bash profiles/tools/run_tools.sh " + tag, "tools": {}}
    for path in sorted(glob.glob(os.path.join(REPO, "gpurun_out", tag, "tools", "*.txt"))):
        tool = os.path.splitext(os.path.basename(path))[0]
        rows = []
        for line in open(path).read().splitlines():
            if line.strip():
                rows.append({"line": line.strip(),
                             "numbers": [float(x) for x in re.findall(r"(?<![\w.])-?\d+(?:\.\d+)?(?![\w.])", line.replace(",", ""))]})
        out["tools"][tool] = rows
    dst = os.path.join(REPO, "profiles", name + "_tools.json")
    json.dump(out, open(dst, "w"), indent=1)
    print(dst, {k: len(v) for k, v in out["tools"].items()})


if __name__ == "__main__":
    main()
