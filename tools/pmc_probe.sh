# usage: bash tools/pmc_probe.sh "<counters>" <python script + args>
mkdir -p gpurun_out/probe
C="$1"; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $C -d /root/repo/gpurun_out/probe/pmc -o p -- python /root/repo/$@ > /root/repo/gpurun_out/probe/pmc.log 2>&1
cd /root/repo
db=$(find gpurun_out/probe/pmc -name '*.db' | head -1)
python tools/rocpd_summary.py pmc $db gpurun_out/probe/pmc.csv && grep "fb::" gpurun_out/probe/pmc.csv
rm -rf gpurun_out/probe/pmc
