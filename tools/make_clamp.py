#!/usr/bin/env python3
"""tests/golden/achieved_rNN.json from recorded full runs: for every check the WORST achieved deviation over
the given gpurun_out/achieved_errors.json files (conftest holds every check to 10x that figure).
usage: make_clamp.py out.json run1.json run2.json [...]   (entries whose name ends in '(count)' are skipped)"""
import json, sys
out, runs = sys.argv[1], sys.argv[2:]
worst = {}
seen = {}
for path in runs:
    d = json.load(open(path))
    for name, v in d.items():
        if name.endswith('(count)'):
            continue
        a = float(v['achieved'] if isinstance(v, dict) else v)
        worst[name] = max(worst.get(name, 0.), a)
        seen[name] = seen.get(name, 0) + 1
json.dump(dict(sorted(worst.items())), open(out, 'w'), indent=0)
print('%d checks; seen in all %d runs: %d' % (len(worst), len(runs), sum(1 for n in seen.values() if n == len(runs))))
