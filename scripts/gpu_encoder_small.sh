#!/bin/bash
# The small-input encoder's evidence -> gpurun_out/${TAG}_encoder_small.log and ${TAG}_query_latency.log:
# latencies against the replayed framework graph, clock stamps inside the GEMM kernels (diagnostic build csrc/abe:
# make -C rag_project_icd10_amd/csrc ABLATE=1 OUT=abe EXTRA='-DICD_FV_LIST="ICD_FV_CASE(6326427)"'), rocprofv3 kernel
# durations of 200 forwards at 15 and 98 tokens, one /query-shaped request at a time.   usage: scripts/gpu_encoder_small.sh r05
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${TAG}_encoder_small.log
mkdir -p $R/gpurun_out
{
echo "# scripts/gpu_encoder_small.sh $TAG: one MI355X, synthetic BERT-base weights (the shapes of shibing624/text2vec-base-chinese)"
echo "# (1) scripts/probe/encoder_small_probe.py: small = csrc/encoder_small.hpp through icd_encoder_encode; graph = the framework's forward replayed from a HIP graph (rounds 1-4)"
timeout -k 10 300 python3 $R/scripts/probe/encoder_small_probe.py 2>&1 | grep -v "not resolvable\|amdgpu.ids"
echo "# (2) scripts/probe/encoder_small_stamps.py 15 (diagnostic build): wave 0 of work-group 1, last layer's four GEMMs"
if [ -f $R/rag_project_icd10_amd/csrc/abe/libicdsearch.so ]; then timeout -k 10 200 python3 $R/scripts/probe/encoder_small_stamps.py 15 2>&1 | grep -v "not resolvable\|amdgpu.ids"; else echo "(csrc/abe not built)"; fi
for nt in 15 98; do
  echo "# (3) rocprofv3 --kernel-trace --stats of scripts/probe/encoder_small_profile.py $nt (200 forwards of one $nt-token string): name, calls, total ns, average ns"
  cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/encp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/encp -o enc -- python3 $R/scripts/probe/encoder_small_profile.py $nt > /tmp/encp.log 2>&1
  f=$(find /tmp/encp -name "*kernel_stats.csv" 2>/dev/null | head -1)
  if [ -n "$f" ]; then python3 -c "
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'enc_' in r['Name'] or 'copyBuffer' in r['Name']:
        print(f\"{r['Name'][:110]:110s} calls {int(r['Calls']):5d}  average {float(r['AverageNs']) / 1e3:8.2f} us  min {float(r['MinNs']) / 1e3:7.2f}\")
" "$f"; else echo "(no kernel stats: $(tail -1 /tmp/encp.log))"; fi
  cd $R
done
} > $O 2>&1
(echo "# scripts/probe/query_latency.py: ONE /query-shaped request at a time (MultiDiagnosisService.match_multiple_diagnoses, top_k = 5), 40 474-row corpus, synthetic encoder / NER weights"; timeout -k 10 300 python3 $R/scripts/probe/query_latency.py 2>&1 | grep -v "not resolvable\|amdgpu.ids") > $R/gpurun_out/${TAG}_query_latency.log
tail -5 $O; cat $R/gpurun_out/${TAG}_query_latency.log
