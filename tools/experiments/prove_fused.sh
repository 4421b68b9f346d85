# GPU box: the full prove with C = Ht + Lt + r Bt1 as one MSM over H | L | B1 (default) against the five separate MSMs (--unfused-c),
# alternating, four proofs per process:  sh tools/experiments/prove_fused.sh
R=$PWD; K=${TMPDIR:-/tmp}/prove_cus; mkdir -p $K
[ -f $K/p4 ] || python3 tools/synth_files.py MNT4753 20 $K/p4 $K/i4 > /dev/null
[ -f $K/p6 ] || python3 tools/synth_files.py MNT6753 15 $K/p6 $K/i6 > /dev/null
M=$R/snark-challenge-prover-reference_amd/main_hip
for round in 1 2; do for f in --fused-c --unfused-c --ref-order; do
  echo "== round $round MNT4753 2^20 $f: $(timeout 300 $M MNT4753 compute $K/p4 $K/i4 $K/o4$f $f --repeat 4 | grep "Total time from input\|load params" | sed 's/Total time from input to output: /prove /; s/load params: /params /' | tr '\n' ' ')"
  echo "== round $round MNT6753 2^15 $f: $(timeout 300 $M MNT6753 compute $K/p6 $K/i6 $K/o6$f $f --repeat 4 | grep "Total time from input\|load params" | sed 's/Total time from input to output: /prove /; s/load params: /params /' | tr '\n' ' ')"
done; done
sha256sum $K/o4--* $K/o6--* | cut -c1-16
