# The store sweep's mix variants at SMALL rasters: 4096² and the cell count of a 1/8 row-block (5792² ≈ 33.5 M cells) — does another tile shape
# win where the launch's ramp and tail are a tenth of the kernel?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04p; mkdir -p $O
cd $R
for side in 4096 5792 8192; do
  ./tools/tune_store $side 15 20 mix > $O/tune_store_mix_$side.log 2> $O/err_$side
  echo "== side $side"; grep -E "^mix" $O/tune_store_mix_$side.log | sort -k 1 | python -c "
import sys,re
rows=[l.rstrip() for l in sys.stdin]
rows.sort(key=lambda l:-float(re.search(r'(\d\.\d+)\s+\S+$',l).group(1)))
print('\n'.join(rows[:7])); print('  shipped:', next((l for l in rows if 'U2 x4w wg-interleave nt+sc1 2fronts lds0K' in l and 'plain' not in l and 'wait' not in l),''))"
done
