# rocprofv3 PMC passes on the skinning kernels (tools/smpl_profile.py): one counter group per pass, aggregated per kernel over the dispatches
# of the B = 19 200 calls (the dispatches with the largest grids).  GLAMR_TAG names the output: gpurun_out/<tag>_pmc_smpl.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${GLAMR_TAG:-r04}
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "FETCH_SIZE" "WRITE_SIZE" "MfmaUtil" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmc_smpl_$i -- python $R/tools/smpl_profile.py > /tmp/pmc_smpl_$i.log 2>&1 || echo "pass $i ($grp) failed" >> $R/gpurun_out/${T}_pmc_smpl.err
done
python - <<PY
import csv, glob
agg = {}
for f in glob.glob('/tmp/pmc_smpl_*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('glamr::', '')
        if not n.startswith('smpl_'):
            continue
        agg.setdefault((n, r['Counter_Name'], int(r['Grid_Size'])), []).append(float(r['Counter_Value']))
with open('$R/gpurun_out/${T}_pmc_smpl.csv', 'w') as out:
    out.write('kernel,grid_size,counter,dispatches,mean,max\n')
    for (n, c, gs), v in sorted(agg.items()):
        out.write('"%s",%d,%s,%d,%.6g,%.6g\n' % (n, gs, c, len(v), sum(v) / len(v), max(v)))
PY
grep "lbs_kernel<0" $R/gpurun_out/${T}_pmc_smpl.csv | sort -t, -k2 -n | tail -40
