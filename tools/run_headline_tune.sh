#!/bin/bash
# Launch variants of the headline kernel over FRESH processes (placement changes per process): REPS processes per variant, interleaved.
OUT=${1:-gpurun_out/headline_tune.txt}
REPS=${REPS:-8}
: > $OUT
for i in $(seq $REPS); do
  for t in "" "256,2,1,1" "512,2,1,1" "256,3,1,1" "512,3,1,1"; do
    python3 tools/headline_layout.py --label "tune_${t:-default}" --tune "$t" --steps 20 >> $OUT 2>> $OUT.err || echo "{\"failed\": \"$t\"}" >> $OUT
  done
done
