set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04x2; mkdir -p $O
cd $R
: > $O/ab.jsonl
for round in 1 2 3; do
  for v in A W; do
    for w in "" "--workload binop --lt u8 --rt u16 --op add" "--workload binop --lt f32 --rt f32 --op add" "--workload binop --lt f64 --rt f64 --op add" "--workload binop --lt u16 --rt u16 --op div" "--rows 2048"; do
      EC_HIP_LIB=$R/erased-cells_amd/ab/lib$v.so python bench.py --no-cpu-baseline --no-reference-streams --no-resident-loop --steps 200 $w 2>>$O/err | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print(json.dumps({'v': '$v', 'w': '$w', 'round': $round, 'frac': d['roofline']['frac'], 'launch_ms': d['roofline'].get('launch_ms'), 'value': d['value'], 'verified': d.get('verified')}))" >> $O/ab.jsonl
    done
  done
  echo round $round done
done
python - <<'PY'
import json, collections, statistics
d=collections.defaultdict(list)
for l in open('gpurun_out/r04x2/ab.jsonl'):
    r=json.loads(l); d[(r['w'],r['v'])].append(r['frac']); assert r['verified'] in (True,None), r
for (w,v),x in sorted(d.items()): print(f"{w or 'headline':50s} {v} med {statistics.median(x):.4f}  {['%.4f'%y for y in x]}")
PY
EC_HIP_LIB=$R/erased-cells_amd/ab/libW.so python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "binop or scalar" 2>&1 | tail -3
