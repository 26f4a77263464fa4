#!/bin/bash
out=gpurun_out/r3m; mkdir -p $out
for b in 2 16; do
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-split-leg --batch $b --plan-json $out/plan_b$b.json > $out/b_b$b.json 2> $out/b_b$b.err
done
python - <<'PY'
import json
for b in (2,16):
    d=json.load(open('gpurun_out/r3m/plan_b%d.json'%b)); rows=d['rows']
    from collections import defaultdict
    agg=defaultdict(lambda:[0,0.0])
    for r in rows: agg[r['op']][0]+=1; agg[r['op']][1]+=r['ms']
    tot=sum(r['ms'] for r in rows)
    print("B=%d in-situ total %.3f ms"%(b,tot))
    for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:14]: print("   %-34s n=%3d %.3f ms  avg %.1f us"%(k,v[0],v[1],1e3*v[1]/v[0]))
    r=json.loads([l for l in open('gpurun_out/r3m/b_b%d.json'%b) if l.startswith('{')][-1]); print("   graph step %.3f ms -> %.1f img/s"%(r['ms_per_step'], r['value']))
PY
