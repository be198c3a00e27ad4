# usage: bash tools/trace_step.sh <tag> [bench args]  -> gpurun_out/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats of bench.py)
TAG=$1; shift
mkdir -p gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/$TAG/stats -o r -- python /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing $@ > /root/repo/gpurun_out/$TAG/stats.log 2>&1
cd /root/repo
db=$(find gpurun_out/$TAG/stats -name '*.db' | head -1)
python tools/rocpd_summary.py stats $db gpurun_out/${TAG}_kernel_stats.csv
rm -rf gpurun_out/$TAG/stats
head -30 gpurun_out/${TAG}_kernel_stats.csv
