#!/bin/bash
# Numbers quoted in DESIGN.md section 4 for the BASELINE configurations other than the headline one (single GPU, default settings)
for cfg in "128 8" "512 64" "1024 129" "2048 17" "2048 128"; do
  set -- $cfg
  timeout -k 10 300 python bench.py --size $1 --frames $2 --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end 2> gpurun_out/cfg_$1_$2.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); v=d.get('variants',{}).get('constant_initial_fields_for_every_pair',{})
print('$1 x $2:', round(d['value'],1), 'pairs/s warm,', round(v.get('value',0),1), 'cold; iterations', round(d['config']['iterations_mean'],2), 'converged', d['config']['converged'], 'levels', d['config']['levels'])"
done
