#!/bin/bash
# whole GPU suite + the default bench line
T=${1:-r04_suite}
mkdir -p gpurun_out/$T
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | grep -v "^  File\|^Extension" | tail -150 > gpurun_out/$T/suite.txt
timeout 300 python bench.py > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err
cp gpurun_out/achieved_errors.json gpurun_out/$T/ 2>/dev/null
grep -n "passed\|failed" gpurun_out/$T/suite.txt | tail -3; python -c "
import json; d=json.loads(open('gpurun_out/$T/bench.json').read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])"
