cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_lat -- python $R/tools/latent_nodes.py 20 > /dev/null 2>&1
f=$(ls /tmp/prof_lat/*/*kernel_stats.csv | head -1)
cp $f $R/gpurun_out/r04_latent_kernel_stats.csv
python - <<PY
import csv
rows = list(csv.DictReader(open('$f')))
tot = sum(int(r['Calls']) for r in rows); t = sum(float(r['TotalDurationNs']) for r in rows)
print('kernels launched: %d, kernel time %.1f ms' % (tot, t / 1e6))
for r in sorted(rows, key=lambda r: -int(r['Calls']))[:40]:
    print('%6d calls  %8.1f us avg  %7.2f ms  %s' % (int(r['Calls']), float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6, r['Name'][:110]))
PY
