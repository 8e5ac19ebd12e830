# full GPU validation: parity tests, smoke, bench, rocprofv3 kernel trace
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -15) > gpurun_out/pytest_gpu.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5) > gpurun_out/smoke.log
(timeout 600 python bench.py --steps 50 --warmup 5 2>&1 | tail -3) > gpurun_out/bench.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r01 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof_bench.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_r01 -name "*stats*" | head; 
for f in $(find gpurun_out/prof_r01 -name "*kernel_stats.csv"); do head -12 $f; done
cat gpurun_out/pytest_gpu.log gpurun_out/smoke.log gpurun_out/bench.log
