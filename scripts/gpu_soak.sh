# longer randomised parity run (three seeds) + rocprofv3 kernel stats of the row-sharded workload
mkdir -p gpurun_out
for seed in 101 202 303; do
  (timeout 1200 python scripts/gpu_fuzz.py --cases 160 --seed $seed 2>&1 | tail -200) > gpurun_out/fuzz_$seed.log
  tail -1 gpurun_out/fuzz_$seed.log; grep -c "^FAIL\|FAIL " gpurun_out/fuzz_$seed.log
done
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_rowshard -- python3 $R/bench.py --workload rowshard --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/rocprof_rowshard.log 2>&1
cd $R
for f in $(find gpurun_out/prof_rowshard -name "*kernel_stats.csv"); do head -12 $f; cp $f gpurun_out/r02_rowshard_kernel_stats.csv; done
tail -1 gpurun_out/rocprof_rowshard.log | cut -c1-300
