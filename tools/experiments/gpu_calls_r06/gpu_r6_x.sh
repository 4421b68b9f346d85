#!/bin/bash
# round 6, GPU call X: what the step of the bench loop spends outside its phases (ms_per_step - phases total) by host wait policy
mkdir -p gpurun_out/r6x; export TMPDIR=/tmp
{
for round in 1 2 3; do
  for v in default spin blocking; do
    unset MNT753_SYNC_SPIN; [ $v = spin ] && export MNT753_SYNC_SPIN=1; [ $v = blocking ] && export MNT753_SYNC_SPIN=0
    python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-prove --no-extras --no-traffic 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['phases_ms']
print('$v', 'ms_per_step', round(d['ms_per_step'],3), {k:round(x,3) for k,x in p.items()}, 'outside the phases', round(d['ms_per_step']-p['total_ms'],3))"
  done
done
} > gpurun_out/r6x/sync_policy.txt 2>&1
cat gpurun_out/r6x/sync_policy.txt; nproc; cat /proc/cpuinfo | grep "model name" | head -1; uptime
