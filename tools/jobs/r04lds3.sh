set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04lds3; mkdir -p $O
cd $R
: > $O/ab.jsonl
for w in "--workload masked_chain" "--workload evi" "--workload ndvi"; do
  for cfg in "0:-1" "1:-1" "1:0" "1:16" "1:24" "1:28" "1:32" "1:36"; do
    v=${cfg%%:*}; k=${cfg##*:}
    python bench.py --no-cpu-baseline --no-reference-streams --no-resident-loop --steps 40 $w --tune binop_variant=$v --tune binop_lds_kb=$k 2>>$O/err | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print(json.dumps({'w': '$w', 'variant': $v, 'lds_kb': $k, 'frac': round(d['roofline']['frac'],4), 'value': round(d['value'],2), 'verified': d.get('verified')}))" >> $O/ab.jsonl
  done
  echo "$w done"
done
python - <<'PY'
import json, collections
d=collections.OrderedDict()
for l in open('gpurun_out/r04lds3/ab.jsonl'):
    r=json.loads(l); d.setdefault(r['w'],[]).append((r['variant'],r['lds_kb'],r['frac'],r['verified']))
for w,x in d.items(): print(w, ' '.join(f"v{a}/{b}K:{c}{'' if e else '!'}" for a,b,c,e in x))
PY
