cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python $R/bench.py --streams 1 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-lines --no-host-stream --no-strong-line > $R/gpurun_out/r05_bench_b1024_under_rocprof.json 2>/dev/null
cp $(ls /tmp/prof_bench/*/*kernel_stats.csv | head -1) $R/gpurun_out/r05_bench_b1024_kernel_stats.csv
# the co-scheduled two-stream pipeline (the headline configuration): what every kernel takes THERE
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cos -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-lines > $R/gpurun_out/r05_bench_coscheduled_under_rocprof.json 2>/dev/null < /dev/null
f=$(ls /tmp/prof_cos/*/*kernel_stats.csv 2>/dev/null | head -1); if [ -n "$f" ]; then cp "$f" $R/gpurun_out/r05_bench_coscheduled_kernel_stats.csv; fi
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_smpl -- python $R/tools/smpl_profile.py > /dev/null 2>&1
cp $(ls /tmp/prof_smpl/*/*kernel_stats.csv | head -1) $R/gpurun_out/r05_smpl_kernel_stats.csv
rocprofv3 --pmc MfmaUtil SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d /tmp/pmc_smpl -- python $R/tools/smpl_profile.py > /dev/null 2>&1
python - <<PY
import csv, glob
agg = {}
for f in glob.glob('/tmp/pmc_smpl/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('glamr::', '')
        agg.setdefault((n, r['Counter_Name']), []).append(float(r['Counter_Value']))
with open('$R/gpurun_out/r05_pmc_smpl.csv', 'w') as out:
    out.write('kernel,counter,dispatches,mean,max\n')
    for (n, c), v in sorted(agg.items()):
        out.write('"%s",%s,%d,%.6g,%.6g\n' % (n, c, len(v), sum(v) / len(v), max(v)))
PY
cd $R && GLAMR_ROUND_TAG=r05 python tools/collect_pmc.py > gpurun_out/r05_collect_pmc.log 2>&1
GLAMR_ROUND_TAG=r05 python tools/collect_pmc.py priors > gpurun_out/r05_collect_pmc_priors.log 2>&1
GLAMR_NETS_FREE=1 GLAMR_ROUND_TAG=r05cos python tools/collect_pmc.py priors > gpurun_out/r05cos_collect_pmc_priors.log 2>&1
head -12 gpurun_out/r05_bench_b1024_kernel_stats.csv | cut -c1-150
