#!/bin/bash
# per-section static instruction census of k_cycle_linear<Nm> (sections = the FB_MARK points of cycle.hip)
NM=${1:-2}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form=1 -mllvm -amdgpu-sched-strategy=max-memory-clause -DFB_ISA_MARKS \
  -I/root/repo/include -I/root/repo/fbpic_amd/csrc -S --cuda-device-only -o /tmp/isa_cycle_marks.s /root/repo/fbpic_amd/csrc/cycle.hip 2>/dev/null
python3 - $NM <<'PY'
import re, sys, collections
s = open('/tmp/isa_cycle_marks.s').read()
nm = '_ZN2fb14k_cycle_linearILi%sELb0ELb0EEEvNS_9CycleArgsE' % sys.argv[1]
b = [x for x in re.split(r'\n(?=_Z\w+:)', s) if x.startswith(nm + ':')][0].split('.Lfunc_end')[0]
sec = 'PROLOGUE'; cnt = collections.OrderedDict()
def kind(op, line):
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('v_') and 'f64' in op: return 'f64'
    if op.startswith(('v_readlane', 'v_writelane', 'v_readfirstlane')): return 'lane'
    if op.startswith(('v_mov', 'v_cndmask', 'v_accvgpr')): return 'mov'
    if op.startswith('v_cmp'): return 'cmp'
    if op.startswith('v_'): return 'int'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('global_', 'flat_', 'buffer_', 'scratch_')): return 'vmem'
    if op.startswith('s_nop'): return 'nop'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_'): return 'salu'
    return 'other'
for line in b.split('\n'):
    t = line.strip()
    m = re.match(r'; MARK (\w+)', t)
    if m: sec = m.group(1); continue
    if not t or t.startswith(('.', ';', '_Z')) or t.endswith(':'): continue
    cnt.setdefault(sec, collections.Counter())[kind(t.split()[0], t)] += 1
keys = ['f64', 'int', 'mov', 'cmp', 'lane', 'mfma', 'lds', 'vmem', 'salu', 'nop', 'wait']
print('%-12s' % 'section' + ''.join('%6s' % k for k in keys) + '   VALU')
for sec, c in cnt.items():
    valu = sum(c[k] for k in ('f64', 'int', 'mov', 'cmp', 'lane'))
    print('%-12s' % sec + ''.join('%6d' % c[k] for k in keys) + '  %5d' % valu)
PY
