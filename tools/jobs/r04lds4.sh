set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04lds4; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_parity.py tests/test_gpu_instantiations.py tests/test_gpu_fullsize.py -q -m gpu -x -s > $O/pytest.log 2>&1 || { grep -E "^(FAILED|ERROR)|passed|failed|Abort|fault|Error|assert" $O/pytest.log | tail -30; exit 1; }
grep -E "passed|failed" $O/pytest.log | tail -1
: > $O/ab.jsonl
for round in 1 2; do
for w in "--workload binop --lt f64 --rt u16" "--workload binop --lt u16 --rt f64" "--workload binop --lt f64 --rt f32" "--workload binop --lt f32 --rt f64" "--workload binop --lt f64 --rt u32" "--workload binop --lt f64 --rt u8" "--workload binop --lt f64 --rt u16 --op div" "--workload evi" "--workload masked_chain" "--workload ndvi --mixed"; do
  for v in -1 0; do
    python bench.py --no-cpu-baseline --no-reference-streams --no-resident-loop --steps 60 $w --tune binop_variant=$v 2>>$O/err | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print(json.dumps({'w': '$w', 'variant': $v, 'frac': round(d['roofline']['frac'],4), 'value': round(d['value'],2), 'verified': d.get('verified')}))" >> $O/ab.jsonl
  done
done
done
python - <<'PY'
import json, collections
d=collections.OrderedDict()
for l in open('gpurun_out/r04lds4/ab.jsonl'):
    r=json.loads(l); d.setdefault(r['w'],{}).setdefault(r['variant'],[]).append(r['frac'])
for w,x in d.items(): print(f"{w:60s} rule {x[-1]}  direct {x[0]}")
PY
