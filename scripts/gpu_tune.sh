#!/bin/bash
# bench under several cycle shapes: prints pairs/s (warm / cold) and mean iterations per setting
run() { python bench.py --no-cpu-baseline --no-end-to-end --steps 2 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); v=d['variants']
print('%-60s warm %6.1f (%.2f it)  cold %6.1f (%.2f it)  wobble %6.1f' % (' '.join(sys.argv[1:]), d['value'], d['config']['iterations_mean'], v['constant_initial_fields_for_every_pair']['value'], v['constant_initial_fields_for_every_pair']['iterations_mean'], v['time_varying_flow_wobble_0.3']['value']))" "$@"; }
run
run --w-cycle-visits 4
run --w-cycle-visits 2
run --nu-pre-coarse 2 --nu-post-coarse 2
run --nu-pre-coarse 1 --nu-post-coarse 2
run --w-cycle-visits 4 --nu-pre-coarse 1 --nu-post-coarse 2
run --nu-pre 1 --nu-post 2
run --nu-pre 2 --nu-post 1
run --w-cycle-level 2
run --vcycle-precision auto
run --vcycle-precision float32
