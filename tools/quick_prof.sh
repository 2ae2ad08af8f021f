#!/bin/bash
# quick kernel-trace of one bench configuration: tools/quick_prof.sh <tag> [bench args...]
R=${GRAFT_REPO_ROOT:-$PWD}
tag=$1; shift
O=$R/gpurun_out/prof_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=20
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o run -- python3 $R/bench.py --no-cpu-baseline --no-verify --steps 20 --warmup 3 "$@" > $O/bench.log 2>&1
find $O -name "*kernel_trace.csv" -size +4M -delete
f=$(find $O -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:40]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.1f} pct {r['Percentage']}")
PY
tail -1 $O/bench.log | cut -c1-600
