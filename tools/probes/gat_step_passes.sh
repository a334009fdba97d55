#!/bin/bash
# the six gather passes of the SpGAT step (config 4), three repetitions
for rep in 1 2 3; do
  timeout 600 python bench.py --full-line --workload gat --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
t=d.get('spmm_launch_table') or {}
print('step %.3f ms; ' % d['ms_per_step'] + ', '.join('%s %.3f' % (k.split(' bfloat16')[0].replace('gat ', '').replace(' heads x ', 'x'), v['avg_ms']) for k, v in t.items()))"
done
