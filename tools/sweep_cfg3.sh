mkdir -p gpurun_out/cfg3
for t in default 256,1,1 256,2,1 256,3,1 128,3,1 512,3,1 512,2,1 256,3,0; do
  if [ $t = default ]; then unset MUGIQ_HIP_CONTRACT_TUNE; else export MUGIQ_HIP_CONTRACT_TUNE=$t; fi
  python bench.py --steps 2 --warmup 1 --extra cfg3 --no-cpu-baseline > gpurun_out/cfg3/b_$t.json 2> gpurun_out/cfg3/b_$t.err || exit 1
  echo "$t done"
done
