#!/bin/bash
# the probe builds of the matrix-pipe tile (tools/probes/fused_mfma_probes.patch + build_exp.sh) on one entry each of the row and the column tile
R=$PWD
for E in "+x:1,3" "+z:1,3"; do
  for n in 0 1 2 3 4; do
    if [ $n = 0 ]; then unset MUGIQ_HIP_LIB; else export MUGIQ_HIP_LIB=$R/tools/probes/build/libmugiq_hip_MUGIQ_MT_EXPERIMENT_$n.so; fi
    MUGIQ_HIP_CARRY_ULTRALOCAL=0 python tools/bench_displaced.py --entries "$E" --nev 200 --plans opt --reps 3 2> /dev/null | tail -n 1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$E', 'experiment $n', round(d['results']['opt']['seconds']*1e3,2), 'ms (incl. the ultra-local pass)')"
  done
done
