#!/bin/bash
# Cycle shape re-tune after the 8-bit stencil format made the stored levels cheaper (warm and cold, headline stack)
run() { echo -n "$* : "; timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants --no-end-to-end "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['config']['iterations_mean'],3), d['config']['iterations_max'])" || exit 1; }
for ws in "" "--warm-start-stride 0"; do
run $ws
run $ws --w-cycle-visits 2
run $ws --w-cycle-visits 4
run $ws --nu-pre-coarse 2 --nu-post-coarse 1
run $ws --nu-pre-coarse 1 --nu-post-coarse 2
run $ws --nu-pre-coarse 2 --nu-post-coarse 2
run $ws --w-cycle-level 0 --w-cycle-visits 2
run $ws --w-cycle-level 2 --w-cycle-visits 2
done
