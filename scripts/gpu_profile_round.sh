#!/bin/bash
# Round-end evidence: kernel-trace stats + separate FETCH_SIZE / WRITE_SIZE passes of the default bench command.
# usage (GPU box): bash scripts/gpu_profile_round.sh <tag>
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
python3 $R/bench.py --steps 3 --warmup 1 --profile-table > $O/bench.json 2> $O/bench.log; echo "bench rc=$?"; tail -1 $O/bench.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-variants --no-end-to-end > $O/stats.log 2>&1; echo "stats rc=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-variants --no-end-to-end > $O/fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-variants --no-end-to-end > $O/write.log 2>&1; echo "write rc=$?"
# keep only what is needed (the traces are large)
for d in stats fetch write; do find $O/$d -name "*agent_info.csv" -delete; done
ls -la $O/*/*/ | head -20
