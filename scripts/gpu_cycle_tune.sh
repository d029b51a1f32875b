#!/bin/bash
# Re-tuning of the cycle shape after the level-0 passes got cheaper: W-cycle visits, coarse sweeps, wobble (time-varying flow) as a second workload
cd $GRAFT_REPO_ROOT
run() { timeout -k 10 300 python bench.py --no-cpu-baseline --no-variants --no-end-to-end "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('$*', '->', round(d['value'],1), 'pairs/s, iterations', round(c['iterations_mean'],3), 'max', c['iterations_max'], 'relres', '%.2e' % c['relres_max'], 'converged', c['converged'])"; }
for w in "" "--wobble 0.5"; do
run $w
run $w --w-cycle-visits 2
run $w --w-cycle-level -1
run $w --w-cycle-level 0 --w-cycle-visits 2
run $w --nu-pre-coarse 2 --nu-post-coarse 1 --w-cycle-visits 2
run $w --nu-pre-coarse 1 --nu-post-coarse 2 --w-cycle-visits 2
run $w --w-cycle-level 2 --w-cycle-visits 3
done
