# GPU box: the full MNT4753 2^20 prove with 1 / 2 / 4 I/O lanes of the input loader, alternating, four proofs each:  sh tools/experiments/prove_io.sh
R=$PWD; K=${TMPDIR:-/tmp}/prove_cus; mkdir -p $K
[ -f $K/p4 ] || python3 tools/synth_files.py MNT4753 20 $K/p4 $K/i4 > /dev/null
M=$R/snark-challenge-prover-reference_amd/main_hip
for round in 1 2; do for l in 1 2 4; do
  echo "== round $round MNT753_IO_LANES=$l: $(MNT753_IO_LANES=$l timeout 300 $M MNT4753 compute $K/p4 $K/i4 $K/o4 --repeat 4 | grep "Total time from input\|input file on the device" | sed 's/Total time from input to output: /prove /; s/input file on the device after: /load /; s/ (background loader)//' | tr '\n' ' ')"
done; done
sha256sum $K/o4 | cut -c1-16
