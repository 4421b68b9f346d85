#!/bin/bash
# round 5, GPU call M: where do the 7-8 s of `load params` of bench.py's first prover child go, when the same binary on the same
# files takes 3.84 s from a shell?  (a) main_hip from the shell, (b) the bench's prove legs with the load trace in the line, (c) from
# the shell again
mkdir -p gpurun_out/r5m
export TMPDIR=/tmp
O=$PWD/gpurun_out/r5m
R=$PWD
K=/tmp/prove_keep; mkdir -p $K
python3 tools/synth_files.py MNT4753 20 $K/p4 $K/i4 > /dev/null
M=$R/snark-challenge-prover-reference_amd/main_hip
{ echo "== (a) shell, first GPU process of the box"; MNT753_TRACE=1 $M MNT4753 compute $K/p4 $K/i4 $K/o4 2>&1 | grep -i "load params\|Total time"; } > $O/a.log 2>&1; tail -3 $O/a.log
sleep 25
{ echo "== (a2) python subprocess.run with pipes, same command"; python3 - <<PY
import subprocess, os
r = subprocess.run(["$M", "MNT4753", "compute", "$K/p4", "$K/i4", "$K/o4", "--repeat", "3"], capture_output=True, text=True, env=dict(os.environ, MNT753_TRACE="1"))
print("\n".join(l for l in (r.stderr + r.stdout).splitlines() if "load params" in l or "Total time" in l))
PY
} > $O/a2.log 2>&1; tail -5 $O/a2.log
sleep 25
BENCH_CPU_PROVE=0 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-traffic --no-exchange > $O/bench_b.json 2> $O/bench_b.err
python3 -c "
import json; j=json.loads([l for l in open('$O/bench_b.json') if l.startswith('{')][-1]); p=j['prove']; print('(b) bench child: load', p['load_params_s'], p['input_to_output_s_all']); print('\n'.join(p['load_params_phases'])); print('cold', p['cold_process'])"
