#!/bin/bash
# Round 5: rocprofv3 kernel trace of the default bench command; prints the per-kernel table.
#   bash tools/r5_kernel_stats.sh <tag> [bench args...]     -> gpurun_out/<tag>_kernel_stats.csv, <tag>_bench.json
tag=${1:-r5}; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out
mkdir -p "$out"
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_prof" -o run -- python3 "$root/bench.py" "$@" --cpu-seconds 0 --busy-seconds 0 > "$out/${tag}_bench.json" 2> "$out/${tag}_bench.err" ) || { echo "profile run failed"; tail -5 "$out/${tag}_bench.err"; exit 1; }
f=$(find "$out/${tag}_prof" -name '*kernel_stats.csv' | head -1)
cp "$f" "$out/${tag}_kernel_stats.csv"
python3 - "$out/${tag}_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:30]:
    print("%-100s calls %6s  total %12s ns  avg %12s ns" % (r["Name"][:100], r["Calls"], r["TotalDurationNs"], r["AverageNs"]))
PY
