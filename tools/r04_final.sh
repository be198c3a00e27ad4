#!/bin/bash
# final state of the round: whole GPU suite, smoke(), the default bench line
T=r04_final
mkdir -p gpurun_out/$T
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -v "^  File\|^Extension" | tail -250 > gpurun_out/$T/suite.txt
cp gpurun_out/achieved_errors.json gpurun_out/$T/ 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/$T/smoke.txt 2>&1
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err
grep -n "passed\|failed" gpurun_out/$T/suite.txt | tail -3; tail -2 gpurun_out/$T/smoke.txt; python -c "
import json; d=json.loads(open('gpurun_out/$T/bench.json').read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], d['extra']['repeat_ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['hankel']['frac'], d['extra']['clocks_before'], d['roofline']['traffic_source'][:40])"
