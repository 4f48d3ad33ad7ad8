#!/bin/bash
# GPU box: the side measurements DESIGN.md quotes (other input distributions, short and mid-sized items, the
# host-pointer rate, small-call latency, one long stream of a long-code coder), each tool's output kept as
# gpurun_out/<tag>/tools/<tool>.txt; profiles/tools/tools_to_json.py turns them into the JSON committed under profiles/.
#   usage: bash profiles/tools/run_tools.sh <tag>
set -u
TAG=${1:-tools}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG/tools
mkdir -p "$OUT"
cd "$ROOT"
for tool in other_distributions tiny_items mid_items plan_time host_path_rate small_call_latency long_code_stream coder_survey graph_capture contention; do
    timeout 900 python3 profiles/tools/$tool.py > "$OUT/$tool.txt" 2> "$OUT/$tool.err"
    if [ "$tool" = long_code_stream ]; then  # and a stream long enough to fill the chip a workgroup per 32 KiB block
        timeout 900 python3 profiles/tools/$tool.py hpack_lengths 134217728 >> "$OUT/$tool.txt" 2>> "$OUT/$tool.err"
        # ... and one whose walks never fall into step (printable symbols of len4to15: codes of 9, 12 and 15 bits only)
        timeout 900 python3 profiles/tools/$tool.py len4to15 134217728 >> "$OUT/$tool.txt" 2>> "$OUT/$tool.err"
    fi
    echo "== $tool"; tail -4 "$OUT/$tool.txt"
done
