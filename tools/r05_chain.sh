cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_chain -- python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-kernel-lines --no-host-stream --no-strong-line > $R/gpurun_out/r05_chain_bench.json 2>/dev/null < /dev/null
f=$(ls /tmp/prof_chain/*/*kernel_trace.csv | head -1)
python $R/tools/chain_trace.py $f > $R/gpurun_out/r05_chain_trace.log 2>&1
python $R/tools/gap_trace.py $f > $R/gpurun_out/r05_gap_trace.log 2>&1
cat $R/gpurun_out/r05_chain_trace.log
