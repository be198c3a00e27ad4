#!/usr/bin/env python3
"""Static instruction census of one kernel in an ISA listing (tools/isa_report.sh writes /tmp/isa_<file>.s):
usage: isa_census.py /tmp/isa_cycle.s k_cycle_linearILi2"""
import re, sys, collections
s = open(sys.argv[1]).read()
pat = sys.argv[2]
blocks = re.split(r'\n(?=_Z\w+:)', s)
for b in blocks:
    nm = b.split(':')[0]
    if pat not in nm or not nm.startswith('_Z'):
        continue
    body = b.split('.Lfunc_end')[0]
    cnt = collections.Counter()
    for line in body.split('\n'):
        line = line.strip()
        if not line or line.startswith(('.', ';', '_Z')) or line.endswith(':'):
            continue
        op = line.split()[0]
        if op.startswith('v_mfma'): k = 'mfma'
        elif op.startswith('v_') and ('f64' in op): k = 'valu_f64'
        elif op.startswith(('v_readlane', 'v_writelane', 'v_readfirstlane')): k = 'valu_lane'
        elif op.startswith('v_') and ('dpp' in line or 'permlane' in op or 'bpermute' in op): k = 'valu_dpp'
        elif op.startswith(('v_mov', 'v_cndmask', 'v_accvgpr')): k = 'valu_mov'
        elif op.startswith('v_cmp'): k = 'valu_cmp'
        elif op.startswith('v_'): k = 'valu_int'
        elif op.startswith('ds_'): k = 'lds'
        elif op.startswith(('global_', 'flat_', 'buffer_', 'scratch_')): k = 'vmem'
        elif op.startswith('s_waitcnt'): k = 's_wait'
        elif op.startswith(('s_cbranch', 's_branch')): k = 's_branch'
        elif op.startswith('s_load'): k = 's_load'
        elif op.startswith('s_nop'): k = 's_nop'
        elif op.startswith('s_'): k = 'salu'
        else: k = 'other'
        cnt[k] += 1
    tot = sum(cnt.values())
    print(nm[:70], 'total', tot)
    for k, v in sorted(cnt.items(), key=lambda kv: -kv[1]):
        print('   %-10s %5d' % (k, v))
