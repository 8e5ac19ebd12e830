# rocprofv3 --kernel-trace --stats of the batch form of the canonical encoder (scripts/probe/encoder_big_profile.py N strings)
# usage: bash scripts/gpu_encoder_big_profile.sh r06 4000
TAG=${1:-r06}; N=${2:-4000}
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/encb
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/encb -o encb -- python3 $R/scripts/probe/encoder_big_profile.py $N > $R/gpurun_out/${TAG}_encoder_big_profile.log 2>&1 < /dev/null
cd $R
f=$(find /tmp/encb -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" gpurun_out/${TAG}_encoder_big_kernel_stats.csv; cut -d, -f1-4 "$f" | head -24 | cut -c1-170; else echo "no kernel_stats.csv"; fi
grep "ms per pass" gpurun_out/${TAG}_encoder_big_profile.log
