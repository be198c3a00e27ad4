#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run16; mkdir -p $O
for pol in homesort=1 homesort=0 homesort=1 homesort=0; do timeout 400 python bench.py --config C3 --no-cpu-baseline --no-kernel-timing --policy $pol > $O/c3.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/c3.json').read().strip().split('\n')[-1]); print('C3 $pol', d['ms_per_step'])"; done
