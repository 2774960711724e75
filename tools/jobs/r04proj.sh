set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04proj; mkdir -p $O
cd $R
: > $O/proj.jsonl
for round in 1 2; do
for rows in 16384 8192 4096 2048; do
  python bench.py --no-cpu-baseline --no-reference-streams --rows $rows 2>>$O/err | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print(json.dumps({'rows': $rows, 'value': round(d['value'],1), 'frac': round(r['frac'],4), 'launch_ms': round(r['launch_ms'],5), 'ms_per_step': round(d['ms_per_step'],5), 'resident_frac': round(r['cache_resident_loop']['frac'],4), 'sets': d['config']['operand_sets']}))" >> $O/proj.jsonl
done
done
cat $O/proj.jsonl
