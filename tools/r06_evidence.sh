#!/bin/bash
# round-6 evidence: full GPU suite (achieved errors), smoke, bench lines (C2 driver command with the CPU
# baseline and the side legs, C3, C5), rocprofv3 kernel stats + PMC passes of the default C2 command, SQ
# counters of the one-pass sequence (DP issue floor), the atomics' memory-side requests, decomposed step
cd /root/repo; export TMPDIR=/tmp
TAG=${1:-r06_v1}
O=gpurun_out/$TAG
mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu > $O/t_all.log 2>&1; echo "gpu suite rc $?" > $O/summary.txt
cp gpurun_out/achieved_errors.json $O/achieved_errors.json 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/summary.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?" >> $O/summary.txt
timeout 400 python bench.py --config C3 --no-cpu-baseline > $O/bench_c3.json 2>/dev/null
timeout 600 python bench.py --config C5 --no-cpu-baseline --no-side-legs > $O/bench_c5.json 2>/dev/null
for f in bench bench_c3 bench_c5; do python -c "
import json,sys; d=json.loads(open('$O/$f.json').read().strip().split('\n')[-1]); print('$f', d['value'], d['ms_per_step'], d.get('extra',{}).get('repeat_ms_per_step'), d['roofline']['kernel'], round(d['roofline']['frac'],3), d['roofline'].get('hankel',{}).get('frac'), {k: v for k, v in d.get('extra', {}).items() if 'ms' in k and 'repeat' not in k}, d['roofline'].get('frac_of_dp_floor'))"; done
bash tools/profile_round.sh $TAG > $O/profile.log 2>&1
head -14 gpurun_out/${TAG}_kernel_stats.csv
bash tools/sq_probe.sh ${TAG}_onepass tools/onepass_only.py 24 > $O/sq.log 2>&1
grep -A40 "k_cycle_linear<2, false, false>" $O/sq.log | head -45
# what the deposition's atomics are on the memory side: requests of the L2s to the fabric, by kind
(cd /tmp && rocprofv3 -L 2>/dev/null | grep -o "TCC_EA0_[A-Z0-9_]*" | sort -u > /root/repo/$O/tcc_ea0_counters.txt)
for C in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_ATOMIC_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_ATOMIC_sum" "TCC_EA0_WR_UNCACHED_32B_sum TCC_WRITEBACK_sum TCC_REQ_sum"; do
  echo "== $C" >> $O/atomics_pmc.txt
  bash tools/pmc_probe.sh "$C" tools/onepass_only.py 12 2>&1 | grep -E "k_cycle_linear<2, false, false>|k_push_x|k_perm" >> $O/atomics_pmc.txt
done
cat $O/atomics_pmc.txt
python tools/loopback_multirank.py --single 2>&1 | grep -v amdgpu.ids | tee $O/loopback_times.txt
python tools/loopback_multirank.py 2>&1 | grep -v amdgpu.ids | tee -a $O/loopback_times.txt
tail -4 $O/t_all.log; cat $O/summary.txt
