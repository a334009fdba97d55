#!/bin/bash
# interleaved A/B of the GAT step (config 4): the rows pass of the backward gathering T (0) against forming t_j from the gathered rows (1)
for rep in 1 2 3; do
  for m in 0 1; do
    DGLL_GAT_ROW_SCORES_BWD=$m timeout 600 python bench.py --full-line --workload gat --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
t=d.get('spmm_launch_table') or {}
print('rows pass row-score $m: step %.3f ms; ' % d['ms_per_step'] + ', '.join('%s %.3f' % (k.split(' bfloat16')[0].replace('gat ', ''), v['avg_ms']) for k, v in t.items() if '8 heads' in k))"
  done
done
