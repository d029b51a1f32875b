#!/bin/bash
run() { python bench.py --no-cpu-baseline --no-end-to-end --steps 2 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); v=d['variants']
print('%-50s warm %6.1f (%.2f it)  cold %6.1f (%.2f it)  wobble %6.1f' % (' '.join(sys.argv[1:]), d['value'], d['config']['iterations_mean'], v['constant_initial_fields_for_every_pair']['value'], v['constant_initial_fields_for_every_pair']['iterations_mean'], v['time_varying_flow_wobble_0.3']['value']))" "$@"; }
run
run --nu-pre 4 --nu-post 4
run --nu-pre 2 --nu-post 4
run --nu-pre 4 --nu-post 2
run --nu-pre 3 --nu-post 3
run --nu-pre 4 --nu-post 4 --w-cycle-visits 2
run --nu-pre 2 --nu-post 4 --w-cycle-visits 2
run --nu-pre 4 --nu-post 4 --w-cycle-level -1
run --nu-pre 6 --nu-post 6
