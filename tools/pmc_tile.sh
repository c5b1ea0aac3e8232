#!/bin/bash
# SQ counters of the tiled displaced contraction (one entry, 48.48.24.24 fp64, N_ev 100), vector tile vs matrix-pipe tile.
# usage (on a GPU box, from the repo root): tools/pmc_tile.sh [entries]      -> gpurun_out/pmc_tile/*.txt
set -o pipefail
R=$PWD
O=$R/gpurun_out/pmc_tile
mkdir -p $O
E=${1:-+z:1,3}
cd /tmp && export TMPDIR=/tmp
for M in 0 1; do
  export MUGIQ_HIP_TILE_MFMA=$M
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE \
    --output-format csv -d $O -o a$M -- python3 $R/tools/bench_displaced.py --nev 100 --entries "$E" --plans opt --reps 2 > $O/a$M.json 2> $O/a$M.err || exit 1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM \
    --output-format csv -d $O -o b$M -- python3 $R/tools/bench_displaced.py --nev 100 --entries "$E" --plans opt --reps 2 > $O/b$M.json 2> $O/b$M.err || exit 1
done
cd $R
for M in 0 1; do
  echo "== MUGIQ_HIP_TILE_MFMA=$M  entries $E"
  for f in a b; do
    python3 tools/pmc_kernel.py $O/${f}${M}_counter_collection.csv displaced_contract_kernel
    python3 - $O/${f}${M}_kernel_trace.csv <<'PY'
import csv, sys
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(sys.argv[1])) if "displaced_contract_kernel" in r["Kernel_Name"]]
print("kernel ms (each launch):", " ".join("%.3f" % x for x in d))
PY
  done
done > $O/summary.txt
cat $O/summary.txt
