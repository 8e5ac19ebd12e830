# round-end evidence: PMC passes and the same-box GEMM reference FIRST (the bench lines that follow cite exactly these files, by
# digest), then tests, smoke, both bench workloads, e2e, the timed full build, the round's probes, fuzz
# usage: scripts/gpu_final.sh r06 [a|b|all]   (a: the reference GEMM, the PMC passes, tests, smoke and the bench lines - ONE box, ONE call, the
#        lines cite that call's files; b: end-to-end, build, probes, fuzz. A gpurun call ends after 20 minutes: the two parts are two calls)
TAG=${1:-r06}
PART=${2:-all}
mkdir -p gpurun_out
if [ "$PART" != "b" ]; then
# (1) the known-good GEMM on THIS box (bench.py: roofline.frac_of_reference_gemm) and the PMC passes (roofline.traffic)
bash scripts/gpu_gemm_reference.sh $TAG > gpurun_out/gemm_reference_$TAG.out 2>&1
cp gpurun_out/${TAG}_gemm_reference.log profiles/${TAG}_gemm_reference.log   # (on the box: the bench line cites THIS box's GEMM; committed afterwards under the same name)
bash scripts/gpu_pmc.sh $TAG > gpurun_out/pmc_$TAG.log 2>&1
cp gpurun_out/${TAG}_pmc_traffic.json gpurun_out/${TAG}_pmc_traffic_rowshard.json profiles/ 2>/dev/null
# (2) tests, smoke, bench
(timeout 1500 python -m pytest tests -q -m gpu -rx 2>&1 < /dev/null | tail -12) > gpurun_out/pytest_gpu.log   # (-rx: an XFAIL of the tight wall-clock limits is named in the log)
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3) > gpurun_out/smoke.log
(timeout 900 python bench.py --steps 50 --warmup 5 2>/dev/null | tail -1) > gpurun_out/bench.log
(timeout 600 python bench.py --workload rowshard --steps 3 2>/dev/null | tail -1) > gpurun_out/bench_rowshard.log
for f in bench_kernel_stats.csv bench_with_extras_kernel_stats.csv bench_exact_mode_kernel_stats.csv rowshard_kernel_stats.csv single_query_kernel_stats.csv pmc_counters.json pmc_counters_exact_mode.json pmc_traffic.json pmc_traffic_rowshard.json; do cp gpurun_out/${TAG}_$f gpurun_out/final_${TAG}_$f 2>/dev/null; done
cat gpurun_out/pytest_gpu.log gpurun_out/smoke.log; cut -c1-900 gpurun_out/bench.log; cut -c1-500 gpurun_out/bench_rowshard.log; tail -12 gpurun_out/pmc_$TAG.log | cut -c1-400
fi
if [ "$PART" = "a" ]; then exit 0; fi
(timeout 600 python scripts/bench_e2e.py 2>/dev/null < /dev/null | tail -80) > gpurun_out/e2e.json
(ICD_EMBEDDING_BATCH=fast timeout 600 python scripts/bench_e2e.py 2>/dev/null < /dev/null | tail -80) > gpurun_out/e2e_fast_batch.json
(timeout 600 python -m pytest tests/test_encoder_gpu.py -q -m gpu -s -k "split or both_encoder" 2>&1 < /dev/null | grep -E "split-bf16|canonical batch|passed|failed") > gpurun_out/encoder_paths.log
(timeout 600 python scripts/bench_build.py 2>/dev/null < /dev/null) > gpurun_out/build_full.json
(timeout 900 python scripts/bench_encoder_corpus.py 2>/dev/null) > gpurun_out/e2e_encoder_corpus.json
# two ranks on this box's ONE device: RCCL refuses duplicate devices; the gloo control flow with the HIP index on one device is what can run
(ICD_BENCH_BACKEND=gloo ICD_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --rows-per-gpu 400000 --rowshard-queries 20000 2>&1 | tail -2) > gpurun_out/bench_2rank_one_device.log
# (3) the round's probes, reproducible in the same call (copied to profiles/ by hand: see profiles/README.md)
(timeout 600 python3 scripts/probe/single_query_probe.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/single_query_probe.log
(timeout 900 python3 scripts/probe/exact_by_k.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/exact_by_k.log
(timeout 600 python3 scripts/probe/rowshard_pacing_probe.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/rowshard_pacing.log
bash scripts/gpu_encoder_small.sh $TAG > gpurun_out/encoder_small_$TAG.out 2>&1   # -> gpurun_out/${TAG}_encoder_small.log, ${TAG}_query_latency.log
# round 6: the batch form of the canonical encoder (kernel table, tile sweep on the diagnostic build), what it costs next to the fast path,
# coarse lists of 24 / 16 at k = 40 ... 100, the single-query fp16 certificate priced, half batches / two handles on two streams
bash scripts/gpu_encoder_big_profile.sh $TAG 4000 < /dev/null > gpurun_out/encoder_big_$TAG.out 2>&1
# (the tile sweeps of the batch form - scripts/gpu_encoder_big_sweep.sh (fp32 arithmetic: ICD_ENCODER_ARITH=fp32), gpu_encoder_big_bf_sweep.sh - ran mid-round on
#  the diagnostic build of their moment: profiles/r06_encoder_big_sweep.log, r06_encoder_big_bf_sweep.log)
(timeout 300 python3 scripts/probe/encode_many_probe.py 2>&1 < /dev/null | grep -v "amdgpu.ids\|SYNTHETIC") > gpurun_out/${TAG}_encode_many_probe.log
# ... what bounds its GEMMs: PMC counters per kernel, clock stamps inside a work-group (diagnostic build csrc/abe)
bash scripts/gpu_encoder_big_pmc.sh $TAG 2000 < /dev/null > gpurun_out/encoder_big_pmc_$TAG.out 2>&1   # -> gpurun_out/${TAG}_encoder_big_pmc.json
if [ -f rag_project_icd10_amd/csrc/abe/libicdsearch.so ]; then (timeout 300 python3 scripts/probe/encoder_big_stamps.py 440 2>&1 < /dev/null | grep -v "amdgpu.ids\|not resolvable\|SYNTHETIC") > gpurun_out/${TAG}_encoder_big_stamps.log; fi
(timeout 300 python3 scripts/probe/encoder_arith_probe.py 2>&1 < /dev/null | grep -v "amdgpu.ids\|SYNTHETIC") > gpurun_out/${TAG}_encoder_arith.log
(ICD_ENCODER_ARITH=fp32 timeout 600 python -m pytest tests/test_encoder_gpu.py tests/test_ner_gpu.py -q -m gpu 2>&1 < /dev/null | tail -2) > gpurun_out/${TAG}_pytest_encoder_fp32_arith.log
(timeout 400 python3 scripts/probe/k100_lists.py 2>&1 < /dev/null | grep -v amdgpu.ids) > gpurun_out/${TAG}_k100_lists.log
(timeout 300 python3 scripts/probe/single_query_fp16_sim.py 2>&1 < /dev/null | grep -v "amdgpu.ids\|SYNTHETIC") > gpurun_out/${TAG}_single_query_fp16_sim.log
(timeout 200 python3 scripts/probe/two_streams.py 2>&1 < /dev/null | grep -v amdgpu.ids) > gpurun_out/${TAG}_two_streams_overlap.log
: > gpurun_out/fuzz_final.log
for seed in 641 642; do (timeout 600 python scripts/gpu_fuzz.py --cases 60 --seed $seed 2>&1 | grep -E "FAIL|gpu_fuzz") >> gpurun_out/fuzz_final.log; done
(timeout 600 python scripts/gpu_fuzz.py --cases 40 --seed 643 --focus exact_k 2>&1 | grep -E "FAIL|gpu_fuzz") >> gpurun_out/fuzz_final.log
(timeout 600 python scripts/gpu_fuzz.py --cases 60 --seed 644 --focus one_query 2>&1 | grep -E "FAIL|gpu_fuzz") >> gpurun_out/fuzz_final.log
(timeout 600 python scripts/gpu_fuzz.py --cases 60 --seed 645 --focus encoder 2>&1 | grep -E "FAIL|gpu_fuzz") >> gpurun_out/fuzz_final.log
grep -A16 stages_ms gpurun_out/e2e.json; cut -c1-600 gpurun_out/bench_2rank_one_device.log; cat gpurun_out/fuzz_final.log
