#!/usr/bin/env python3
"""tests/golden/achieved_rNN.json from recorded full runs: for every check the WORST achieved deviation over
the given gpurun_out/achieved_errors.json files (conftest holds every check to 10x that figure).

usage: make_clamp.py out.json [--previous prev.json] [--looser justification.json] run1.json run2.json [...]
  (entries whose name ends in '(count)' are skipped)

--previous: the clamp of the round before.  A check that exists there keeps min(previous, new worst): a round
  that loses accuracy FAILS the old figure instead of silently becoming the new bound.  A
  check may only get looser when --looser names it with a reason ({"check name": "why"}); the list of such
  entries is printed and stored under the key '__looser__' of the output (conftest ignores that key)."""
import json, sys
args = sys.argv[1:]
out = args.pop(0)
prev, looser = {}, {}
while args and args[0].startswith('--'):
    flag = args.pop(0)
    path = args.pop(0)
    if flag == '--previous':
        prev = {k: v for k, v in json.load(open(path)).items() if not k.startswith('__')}
    elif flag == '--looser':
        looser = json.load(open(path))
    else:
        sys.exit('unknown option ' + flag)
runs = args
worst = {}
seen = {}
for path in runs:
    d = json.load(open(path))
    for name, v in d.items():
        if name.endswith('(count)') or name.startswith('__'):
            continue
        a = float(v['achieved'] if isinstance(v, dict) else v)
        worst[name] = max(worst.get(name, 0.), a)
        seen[name] = seen.get(name, 0) + 1
kept, loosened, refused = 0, {}, []
for name, p in prev.items():
    p = float(p)
    if name not in worst:
        worst[name] = p                    # a check that did not run this time keeps its figure
        continue
    if worst[name] > p:
        if name in looser:
            loosened[name] = {'previous': p, 'new': worst[name], 'why': looser[name]}
        else:
            refused.append((name, p, worst[name]))
            worst[name] = p
            kept += 1
res = dict(sorted(worst.items()))
if loosened:
    res['__looser__'] = loosened
json.dump(res, open(out, 'w'), indent=0)
print('%d checks; seen in all %d runs: %d' % (len(worst), len(runs), sum(1 for n in seen.values() if n == len(runs))))
if prev:
    print('%d checks keep the tighter figure of the previous clamp; %d loosened with a stated reason' % (kept, len(loosened)))
    for name, p, w in refused:
        print('  kept %.2e (this round measured %.2e): %s' % (p, w, name))
