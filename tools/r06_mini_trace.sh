#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/mini_trace
export GLAMR_LIB_PATH=$R/tools/_lib_spin.so
cd $R
GLAMR_MINI_NO_LENS=1 GLAMR_MINI_SKIP=3 GLAMR_NETS_NO_GRAPH=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/mini_trace/nets -- python tools/race_mini.py 5 3 nets > gpurun_out/mini_trace/nets.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/mini_trace/lin -- python tools/race_mini.py 0 3 lin > gpurun_out/mini_trace/lin.log 2>&1
for v in nets lin; do f=$(find gpurun_out/mini_trace/$v -name "*kernel_trace.csv" | head -1); echo "== $v $f"; grep SUMMARY gpurun_out/mini_trace/$v.log; python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last 40 kernels
t0 = int(rows[-40]['Start_Timestamp'])
for r in rows[-40:]:
    print('%10.1f us %8.1f us  q%s  %s  grid %s wg %s lds %s' % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r.get('Queue_Id', '?'), r['Kernel_Name'][:70], r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?')), r.get('LDS_Block_Size', '?')))
PY
done
