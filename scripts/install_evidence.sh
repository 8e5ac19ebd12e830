# copy what scripts/gpu_final.sh TAG a|b left under gpurun_out/ (scratch) into profiles/ (tracked) under the names profiles/README.md lists
# usage: bash scripts/install_evidence.sh r06
TAG=${1:-r06}
G=gpurun_out; P=profiles
cpif() { [ -s "$1" ] && cp "$1" "$2"; }
cpif $G/bench.log $P/${TAG}_bench_n1.json
cpif $G/bench_rowshard.log $P/${TAG}_bench_rowshard_n1.json
for f in bench_kernel_stats.csv bench_with_extras_kernel_stats.csv bench_exact_mode_kernel_stats.csv rowshard_kernel_stats.csv single_query_kernel_stats.csv pmc_counters.json pmc_counters_exact_mode.json pmc_traffic.json pmc_traffic_rowshard.json; do cpif $G/final_${TAG}_$f $P/${TAG}_$f; done
cpif $G/${TAG}_gemm_reference.log $P/${TAG}_gemm_reference.log
{ echo "# pytest -m gpu -rx (scripts/gpu_final.sh $TAG a), then __graft_entry__.smoke()"; cat $G/pytest_gpu.log; grep -v amdgpu.ids $G/smoke.log; } > $P/${TAG}_pytest_gpu_and_smoke.log
cpif $G/e2e.json $P/${TAG}_e2e_config3.json
cpif $G/e2e_fast_batch.json $P/${TAG}_e2e_config3_fast_batch.json
cpif $G/e2e_encoder_corpus.json $P/${TAG}_e2e_encoder_corpus.json
cpif $G/build_full.json $P/${TAG}_build_full.json
cpif $G/encoder_paths.log $P/${TAG}_encoder_paths_final.log
cpif $G/bench_2rank_one_device.log $P/${TAG}_bench_2rank_one_device.log
cpif $G/single_query_probe.log $P/${TAG}_single_query.log
cpif $G/exact_by_k.log $P/${TAG}_exact_mode_by_k.log
cpif $G/rowshard_pacing.log $P/${TAG}_rowshard_pacing.log
cpif $G/fuzz_final.log $P/${TAG}_fuzz_final.log
for f in encoder_small.log query_latency.log encode_many_probe.log encoder_arith.log pytest_encoder_fp32_arith.log k100_lists.log single_query_fp16_sim.log two_streams_overlap.log encoder_big_pmc.json encoder_big_stamps.log; do cpif $G/${TAG}_$f $P/${TAG}_$f; done
git status --short $P | head -60
