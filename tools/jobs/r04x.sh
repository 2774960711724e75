set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04x; mkdir -p $O
cd $R
: > $O/ab.jsonl
for round in 1 2 3 4; do
  for v in A B C; do
    for w in "" "--workload binop --lt u8 --rt u16 --op add" "--workload binop --lt f64 --rt f64 --op add"; do
      EC_HIP_LIB=$R/erased-cells_amd/ab/lib$v.so python bench.py --no-cpu-baseline --no-reference-streams --no-resident-loop --steps 200 $w 2>>$O/err | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print(json.dumps({'v': '$v', 'w': '$w', 'round': $round, 'frac': d['roofline']['frac'], 'launch_ms': d['roofline'].get('launch_ms'), 'value': d['value']}))" >> $O/ab.jsonl
    done
  done
  echo round $round done
done
cat $O/ab.jsonl
