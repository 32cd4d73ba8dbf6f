cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cos -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/r04j_bench_coscheduled_under_rocprof.json 2>/dev/null < /dev/null
f=$(ls /tmp/prof_cos/*/*kernel_stats.csv 2>/dev/null | head -1); if [ -n "$f" ]; then cp "$f" $R/gpurun_out/r04j_bench_coscheduled_kernel_stats.csv; fi
head -30 $R/gpurun_out/r04j_bench_coscheduled_kernel_stats.csv | cut -c1-140
