set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04shprof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof --output-format csv -- python3 /root/repo/bench.py --no-cpu-baseline --no-reference-streams --no-resident-loop --rows 2048 > $O/bench.json 2> $O/err || tail -5 $O/err
cd $R
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); cp "$f" $O/kernel_stats.csv
t=$(find $O/prof -name '*kernel_trace.csv' | head -1)
python - "$t" <<'PY'
import csv, sys, statistics
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'k_binop_direct' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
rows=rows[-200:]
dur=[int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in rows]
gap=[int(rows[i+1]['Start_Timestamp'])-int(rows[i]['End_Timestamp']) for i in range(len(rows)-1)]
per=[int(rows[i+1]['Start_Timestamp'])-int(rows[i]['Start_Timestamp']) for i in range(len(rows)-1)]
print("last 200 launches: kernel duration mean %.2f us (median %.2f), gap between kernels mean %.2f us (median %.2f), start-to-start %.2f us" % (statistics.mean(dur)/1e3, statistics.median(dur)/1e3, statistics.mean(gap)/1e3, statistics.median(gap)/1e3, statistics.mean(per)/1e3))
PY
find $O/prof -name '*kernel_trace.csv' -delete
head -2 $O/kernel_stats.csv | cut -c1-200
python -c "
import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['frac'])"
