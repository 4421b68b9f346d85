# GPU box: the full MNT4753 2^20 prove at several CU budgets of the point kernels (main_hip --point-cus N, three proofs each) --
# does leaving CUs free let the latency-bound phases of one MSM run under the accumulation of another?   sh tools/experiments/prove_cus.sh 256 240 224
R=$PWD; K=${TMPDIR:-/tmp}/prove_cus; mkdir -p $K
[ -f $K/p4 ] || python3 tools/synth_files.py MNT4753 20 $K/p4 $K/i4 > /dev/null
M=$R/snark-challenge-prover-reference_amd/main_hip
for cu in "$@"; do
  echo "== --point-cus $cu"
  timeout 300 $M MNT4753 compute $K/p4 $K/i4 $K/o4 --point-cus $cu --repeat 4 | grep "Total time from input\|load params"
  sha256sum $K/o4 | cut -c1-16
done
grep -A8 MNT4753_2p20 tests/golden/oracle_hashes.json | grep output_sha256
