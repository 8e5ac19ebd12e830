# round 3: GPU suite, kernel trace of the second pass with few / all queries flagged, the timed full build
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -25) > gpurun_out/pytest_gpu.log
cd /tmp
for sp in 0.35 0.10; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_p2_$sp -- python3 $R/scripts/probe/second_pass_trace.py $sp > $R/gpurun_out/p2_trace_$sp.log 2>&1
  for f in $(find $R/gpurun_out/prof_p2_$sp -name "*kernel_stats.csv"); do cp $f $R/gpurun_out/p2_kernel_stats_$sp.csv; done
done
cd $R
(timeout 900 python scripts/bench_build.py 2> gpurun_out/build_full.err) > gpurun_out/build_full.json
cat gpurun_out/pytest_gpu.log; for sp in 0.35 0.10; do tail -2 gpurun_out/p2_trace_$sp.log | cut -c1-300; head -12 gpurun_out/p2_kernel_stats_$sp.csv | cut -c1-200; done
tail -5 gpurun_out/build_full.err; head -60 gpurun_out/build_full.json
