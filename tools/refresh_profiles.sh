#!/bin/bash
# Round-end refresh on a GPU box: full GPU suite, the default bench line, its rocprof kernel stats, and the PMC traffic passes
# (FETCH_SIZE / WRITE_SIZE, separate runs) for the headline kernel and the kernels of the extra legs.  Everything lands under
# gpurun_out/refresh/; copy what is to be kept into profiles/.
set -o pipefail
R=$PWD
O=$R/gpurun_out/refresh
mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -2 $O/pytest.log
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o bench -- python3 $R/bench.py --no-extra --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof.err || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O -o pmc_$c -- python3 $R/bench.py --steps 4 --warmup 1 --extra displaced,mg --no-cpu-baseline > $O/pmc_$c.json 2> $O/pmc_$c.err || exit 1
done
cd $R
python3 tools/pmc_traffic.py $O/pmc_FETCH_SIZE_counter_collection.csv $O/pmc_WRITE_SIZE_counter_collection.csv loop_contract_kernel "$(python3 -c "import json;print(json.load(open('$O/bench.json'))['config']['workload'])")" $O/traffic_latest.json 1048576
python3 tools/pmc_traffic_extra.py $O/pmc_FETCH_SIZE_counter_collection.csv $O/pmc_WRITE_SIZE_counter_collection.csv $O/traffic_extra_latest.json 400
for c in FETCH_SIZE WRITE_SIZE; do python3 tools/slim_pmc_csv.py $O/pmc_${c}_counter_collection.csv $O/pmc_${c}_mugiq.csv; done
ls $O
