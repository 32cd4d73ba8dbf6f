"""Development aid: from a rocprofv3 --kernel-trace CSV of tools/overlap_probe.py, the durations of the other queue's kernels that
ran entirely inside an optimiser-stage launch."""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
t0 = min(r['s'] for r in rows)
stage = [r for r in rows if 'grecon_stage_kernel<1, true, 1' in r['Kernel_Name']]
def key(n):
    for k, v in (('gemm_split_kernel<2>', 'gemm2'), ('gemm_split_kernel<1>', 'gemm1'), ('attention_kernel<50>', 'attn50'), ('grecon_stage', 'stage'), ('lstm', 'lstm')):
        if k in n:
            return v
    return 'other'
for st in stage:
    by = collections.defaultdict(list)
    for r in rows:
        if r['Queue_Id'] != st['Queue_Id'] and r['s'] >= st['s'] and r['e'] <= st['e']:
            by[key(r['Kernel_Name'])].append((r['e'] - r['s']) / 1e3)
    print('stage q%s at %.1f ms, %.1f ms | ' % (st['Queue_Id'], (st['s'] - t0) / 1e6, (st['e'] - st['s']) / 1e6) +
          ', '.join('%s n=%d avg %.0f us' % (k, len(v), sum(v) / len(v)) for k, v in sorted(by.items())))
