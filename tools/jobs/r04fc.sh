set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04fc; mkdir -p $O
cd $R
: > $O/ab.jsonl
for w in "--workload masked_chain --fused" "--workload ndvi --fused" "--workload ndvi --fused --mixed" "--workload evi --fused" "--workload evi --fused --compiled"; do
  for k in 0 12 16 20 24 32 48; do
    python bench.py --no-cpu-baseline --no-reference-streams --no-resident-loop --steps 60 $w --tune fused_lds_kb=$k 2>>$O/err | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print(json.dumps({'w': '$w', 'lds_kb': $k, 'frac': round(d['roofline']['frac'],4), 'verified': d.get('verified')}))" >> $O/ab.jsonl
  done
  echo "$w done"
done
python - <<'PY'
import json, collections
d=collections.OrderedDict()
for l in open('gpurun_out/r04fc/ab.jsonl'):
    r=json.loads(l); d.setdefault(r['w'],[]).append((r['lds_kb'],r['frac']))
for w,x in d.items(): print(f"{w:45s}", ' '.join(f"{a}K:{c}" for a,c in x))
PY
