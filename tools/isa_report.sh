# usage: bash tools/isa_report.sh <file.hip> [name filter]  -> VGPR / AGPR / scratch / occupancy of every kernel
SRC=/root/repo/fbpic_amd/csrc/$1
OUT=/tmp/isa_$(basename $1 .hip).s
mkdir -p /tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics \
  -I/root/repo/include -I/root/repo/fbpic_amd/csrc -S --cuda-device-only -o $OUT $SRC 2>&1 | grep -A8 "error" | head -40
python3 - "$OUT" "${2:-}" <<'PY'
import re, subprocess, sys
s = open(sys.argv[1]).read()
flt = sys.argv[2]
for b in re.split(r'\n(?=_Z\w+:)', s):
    nm = b.split(':')[0]
    if not nm.startswith('_Z'):
        continue
    v = re.search(r'; NumVgprs: (\d+)', b)
    if not v:
        continue
    d = subprocess.run(['c++filt', nm], capture_output=True, text=True).stdout.strip()
    d = re.sub(r'\(.*', '', d)
    if flt and flt not in d:
        continue
    a = re.search(r'; NumAgprs: (\d+)', b); sc = re.search(r'; ScratchSize: (\d+)', b)
    oc = re.search(r'; Occupancy: (\d+)', b); sg = re.search(r'; TotalNumSgprs: (\d+)', b)
    print('%-56s vgpr %3s agpr %3s sgpr %3s scratch %4s occ %s  lines %5d mfma %3d' % (
        d[:56], v.group(1), a.group(1), sg.group(1), sc.group(1), oc.group(1), b.count('\n'), b.count('v_mfma')))
PY
