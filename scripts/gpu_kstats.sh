#!/bin/bash
# per-kernel time summary (rocprofv3 --kernel-trace --stats) of one bench step with extra bench arguments: gpu_kstats.sh <tag> [bench args]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/kstats_$TAG
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-variants --no-end-to-end "$@" > $O/stats.log 2>&1; echo "stats rc=$?"
f=$(ls $O/stats/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} total {float(r['TotalDurationNs'])/1e6:9.3f} ms avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Percentage']}%")
PY
find $O/stats -name "*agent_info.csv" -delete; find $O/stats -name "*kernel_trace.csv" -size +20M -delete
