#!/bin/bash
# round 5, GPU call A: (1) the whole GPU suite on the new layout (test library split, peer access, modified-Jacobian window table,
# live reference-generator differential test, N-GPU prove legs of bench.py); (2) level 1 of the pairing pass with its table rows
# folded into spans of 2^20 / 2^22 / 2^24 rows against the full 40 x 2^20 (verdict item 1a); (3) what a fused A + C pass could gain
# (verdict item 5); (4) table-build time with the new doubling chain.
mkdir -p gpurun_out/r5a
export TMPDIR=/tmp
O=$PWD/gpurun_out/r5a
R=$PWD
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 $O/pytest.log | cut -c1-200
# (4) + headline on this box
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-prove --no-extras --no-traffic --no-exchange > $O/bench_quick.json 2> $O/bench_quick.err
python3 -c "
import json; j=json.load(open('$O/bench_quick.json')); print('bench', round(j['ms_per_step'],3), 'ms/step', round(j['value']/1e6,2), 'Mpts/s precompute_ms', round(j['precompute_ms'],1), j['phases_ms'], 'parity', j['parity_ok'])"
# (2) level-1 row span: per-kernel times, alternating
for round in 1 2; do for v in base span24 span22 span20; do
  if [ $v = base ]; then L=$R/snark-challenge-prover-reference_amd/libmnt753_hip.so; else L=$R/build_exp/$v/libmnt753_hip.so; fi
  (cd /tmp && MNT753_LIB=$L timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/span_${v}_$round -o x -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-prove --no-extras --no-traffic --no-exchange > /tmp/span_${v}_$round.json 2>/dev/null)
  python3 - /tmp/span_${v}_$round $v $round <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if "k_pair_level" in r["Name"] or "k_bucket_accumulate" in r["Name"]]
for r in rows[:6]:
    n = r["Name"].split("(")[0].replace("void mnt753::", "")[:70]
    print(f"span {sys.argv[2]:7s} round {sys.argv[3]}  {n:70s} calls {r['Calls']:>3s} avg_ms {float(r['AverageNs'])/1e6:8.3f} min_ms {float(r['MinNs'])/1e6:8.3f}")
PY
done; done > $O/level1_row_span.txt 2>&1
cat $O/level1_row_span.txt | grep "Lb1ELb0\|true, false" | head -20
cat $O/level1_row_span.txt | head -50
# (3)
timeout 600 python3 tools/experiments/ac_fusion_sizing.py > $O/ac_fusion_sizing.txt 2>&1; cat $O/ac_fusion_sizing.txt | cut -c1-400
# kernel stats of one full prove: where the parameter load goes now
K=/tmp/prove_keep; mkdir -p $K
python3 tools/synth_files.py MNT4753 20 $K/p4 $K/i4 > /dev/null
M=$R/snark-challenge-prover-reference_amd/main_hip
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_prove -o prove -- $M MNT4753 compute $K/p4 $K/i4 $K/o4 --repeat 2 > $O/prove_under_rocprof.log 2>&1)
cat $O/prove_under_rocprof.log | grep -i "load params\|Total time" ; sha256sum $K/o4; grep -A1 MNT4753_2p20 tests/golden/oracle_hashes.json | head -3
python3 - <<'PY' > $O/kt_prove_top.txt
import csv, glob
f = glob.glob("/tmp/kt_prove/**/*kernel_stats.csv", recursive=True)
for r in list(csv.DictReader(open(f[0])))[:14]:
    n = r["Name"].split("(")[0].replace("void mnt753::", "")[:80]
    print(f"{n:80s} calls {r['Calls']:>4s} total_ms {float(r['TotalDurationNs'])/1e6:10.2f} avg_ms {float(r['AverageNs'])/1e6:9.3f}")
PY
cat $O/kt_prove_top.txt
$M MNT4753 compute $K/p4 $K/i4 $K/o4b --repeat 3 | grep -i "load params\|Total time"
python3 tools/synth_files.py MNT6753 15 $K/p6 $K/i6 > /dev/null
$M MNT6753 compute $K/p6 $K/i6 $K/o6 --repeat 5 | grep -i "load params\|Total time"
rm -rf $K
