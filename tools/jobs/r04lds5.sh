set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04lds5; mkdir -p $O
cd $R
: > $O/ab.jsonl
for round in 1 2 3; do
for w in "--lt f32 --rt u16" "--lt u16 --rt f32" "--lt f32 --rt u8" "--lt u8 --rt f32" "--lt u32 --rt u16" "--lt u32 --rt u8" "--lt f32 --rt f32"; do
  for cfg in "0:-1" "1:0" "1:16"; do
    v=${cfg%%:*}; k=${cfg##*:}
    python bench.py --no-cpu-baseline --no-reference-streams --no-resident-loop --steps 100 --workload binop --op add $w --tune binop_variant=$v --tune binop_lds_kb=$k 2>>$O/err | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print(json.dumps({'w': '$w', 'variant': $v, 'lds_kb': $k, 'frac': round(d['roofline']['frac'],4)}))" >> $O/ab.jsonl
  done
done
echo round $round
done
python - <<'PY'
import json, collections, statistics
d=collections.OrderedDict()
for l in open('gpurun_out/r04lds5/ab.jsonl'):
    r=json.loads(l); d.setdefault(r['w'],collections.OrderedDict()).setdefault((r['variant'],r['lds_kb']),[]).append(r['frac'])
for w,x in d.items(): print(f"{w:22s}", '  '.join(f"v{a}/{b}K: med {statistics.median(c):.4f} {c}" for (a,b),c in x.items()))
PY
