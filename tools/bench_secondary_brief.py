import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('head %.1f' % d['value'], d['stages_ms'])
for s in d.get('secondary', []): print('  sec %.1f' % s['value'], s['workload'][20:90], {k: round(v,3) for k,v in s['stages_ms'].items()})
