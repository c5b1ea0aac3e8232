#!/bin/bash
# the same job with two builds of the library, alternating, on one box: tools/probes/ab_lib.sh <other libmugiq_hip.so> [entries] [nev]
R=$PWD
L=$1; E=${2:-"+x:1,3;+y:1,3;+z:1,3;+t:1,3"}; N=${3:-200}
for rep in 1 2 3; do
  for lib in product $L; do
    if [ $lib = product ]; then unset MUGIQ_HIP_LIB; else export MUGIQ_HIP_LIB=$R/$lib; fi
    python tools/bench_displaced.py --entries "$E" --nev $N --plans opt --reps 3 2> /dev/null | tail -n 1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib', round(d['results']['opt']['seconds']*1e3,2), 'ms')"
  done
done
