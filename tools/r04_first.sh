# round 4, first GPU call: the GPU suite with its prints, the skinning alone under rocprofv3, one default bench run
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${GLAMR_TAG:-r04a}
(cd $R && timeout 900 python -m pytest tests -m gpu -q -s -p no:cacheprovider 2>&1 | grep -v "Warning\|warn(\|^  " | tail -250 > gpurun_out/${T}_gputest.log)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_smpl -- python $R/tools/smpl_profile.py > /dev/null 2>&1
cp $(ls /tmp/prof_smpl/*/*kernel_stats.csv | head -1) $R/gpurun_out/${T}_smpl_kernel_stats.csv
(cd $R && timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err)
tail -5 $R/gpurun_out/${T}_gputest.log; head -8 $R/gpurun_out/${T}_smpl_kernel_stats.csv | cut -c1-160; cut -c1-400 $R/gpurun_out/${T}_bench.json
