export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03_headline -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $R/gpurun_out/rocprof_headline.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03_exact -- python3 $R/bench.py --mode exact --steps 10 --warmup 3 --no-cpu-baseline --no-extras --repeats 0 > $R/gpurun_out/rocprof_exact.log 2>&1
cd $R
for f in $(find gpurun_out/prof_r03_headline -name "*kernel_stats.csv"); do cp $f gpurun_out/r03_bench_kernel_stats.csv; head -8 $f | cut -c1-160; done
for f in $(find gpurun_out/prof_r03_exact -name "*kernel_stats.csv"); do cp $f gpurun_out/r03_bench_exact_mode_kernel_stats.csv; head -4 $f | cut -c1-160; done
tail -1 gpurun_out/rocprof_exact.log | cut -c1-300
(ICD_BENCH_BACKEND=gloo ICD_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --rows-per-gpu 400000 --rowshard-queries 20000 2>&1 | tail -3) > gpurun_out/bench_2rank_one_device.log
cut -c1-900 gpurun_out/bench_2rank_one_device.log
(timeout 300 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-rowshard 2>&1 | tail -4) > gpurun_out/bench_gpus2_on_one_gpu_box.log; cat gpurun_out/bench_gpus2_on_one_gpu_box.log | cut -c1-300
