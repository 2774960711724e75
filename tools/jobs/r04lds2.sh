set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04lds2; mkdir -p $O
cd $R
: > $O/ab.jsonl
for w in "--lt u32 --rt f32" "--lt f64 --rt f32" "--lt f32 --rt f64" "--lt f64 --rt u32" "--lt f64 --rt u16" "--lt u16 --rt f64" "--lt f64 --rt u8" "--lt f32 --rt u16" "--lt u16 --rt u32" "--lt f32 --rt u8"; do
  for cfg in "0:-1" "1:0" "1:12" "1:16" "1:20" "1:24" "1:28" "1:32" "1:40"; do
    v=${cfg%%:*}; k=${cfg##*:}
    python bench.py --no-cpu-baseline --no-reference-streams --no-resident-loop --steps 100 --workload binop --op add $w --tune binop_variant=$v --tune binop_lds_kb=$k 2>>$O/err | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print(json.dumps({'w': '$w', 'variant': $v, 'lds_kb': $k, 'frac': round(d['roofline']['frac'],4)}))" >> $O/ab.jsonl
  done
  echo "$w done"
done
python - <<'PY'
import json, collections
d=collections.OrderedDict()
for l in open('gpurun_out/r04lds2/ab.jsonl'):
    r=json.loads(l); d.setdefault(r['w'],[]).append((r['variant'],r['lds_kb'],r['frac']))
for w,x in d.items(): print(w, ' '.join(f"v{a}/{b}K:{c}" for a,b,c in x))
PY
