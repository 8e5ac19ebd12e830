# a known-good GEMM, the bare MFMA stream and the shipped coarse kernel (with / without its select) on ONE box
# in ONE call, with clock and power; then the vendor kernels' names and tiles from a rocprofv3 kernel trace.
# needs: make -C rag_project_icd10_amd/csrc ABLATE=1 OUT=abc EXTRA='-DICD_FV_LIST="ICD_FV_CASE(6326427) ICD_FV_CASE(6334619) ICD_FV_CASE(39880859) ICD_FV_CASE(39889051)"'
#        (cd scripts/probe && hipcc --offload-arch=gfx950 -O3 -o bare_mfma bare_mfma.hip)
# usage: scripts/gpu_gemm_reference.sh r05
TAG=${1:-r05}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
(timeout 900 python3 scripts/probe/gemm_reference.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/${TAG}_gemm_reference.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_gemmref -- python3 $R/scripts/probe/gemm_reference.py --trace-only > $R/gpurun_out/rocprof_gemmref_$TAG.log 2>&1
cd $R
echo "## rocprofv3 --kernel-trace --stats of the three vendor GEMMs (12 launches each; name, calls, average ns)" >> gpurun_out/${TAG}_gemm_reference.log
for f in $(find gpurun_out/prof_${TAG}_gemmref -name "*kernel_stats.csv"); do head -8 $f | cut -c1-400 >> gpurun_out/${TAG}_gemm_reference.log; done
cat gpurun_out/${TAG}_gemm_reference.log
