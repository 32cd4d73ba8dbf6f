# LDS counters of the optimiser stage kernel (1024 scenes x 200 iterations, tools/microbench.py): bank conflicts against all LDS cycles
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_BUSY_CU_CYCLES" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES"; do
  d=/tmp/pmc_lds_$(echo $set | cut -c1-12 | tr ' ' _)
  GLAMR_MB_FRAMES=300 GLAMR_MB_SCENES=1024 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -- python $R/tools/microbench.py > /dev/null 2>&1
  python - <<PY
import csv, glob
agg = {}
for f in glob.glob('$d/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'grecon_stage_kernel<1, true, 1, 304>' in r['Kernel_Name']:
            agg.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
for k, v in sorted(agg.items()):
    print('%-28s launches %d  mean %.6g' % (k, len(v), sum(v) / len(v)))
PY
done
