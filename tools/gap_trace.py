"""Development aid: from a rocprofv3 --kernel-trace CSV of the default bench.py run (two co-scheduled streams), what runs BETWEEN consecutive
optimiser-stage launches (the serial gap of the pipeline): per gap its length and the kernels inside it, then the average over the gaps.
usage: python tools/gap_trace.py kernel_trace.csv"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
short = lambda n: n.replace('void ', '').replace('glamr::', '').replace('(anonymous namespace)::', '').split('(')[0][:48]
stage = [r for r in rows if 'grecon_stage_kernel<1, true, 1, 304>' in r['Kernel_Name'] and (r['e'] - r['s']) > 5e6]
gaps = []
for a, b in zip(stage[:-1], stage[1:]):
    if b['s'] - a['e'] > 30e6 or b['s'] < a['e']:      # (not consecutive launches of the pipeline / overlapping ones)
        continue
    inside = [r for r in rows if r['e'] > a['e'] and r['s'] < b['s'] and r is not a and r is not b]
    gaps.append((a, b, inside))
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
print('%d stage launches, %d gaps' % (len(stage), len(gaps)))
for a, b, inside in gaps:
    for r in inside:
        s, e = max(r['s'], a['e']), min(r['e'], b['s'])
        k = short(r['Kernel_Name'])
        agg[k][0] += 1; agg[k][1] += (e - s) / 1e3; agg[k][2] += (r['e'] - r['s']) / 1e3
glen = [(b['s'] - a['e']) / 1e6 for a, b, _ in gaps]
slen = [(r['e'] - r['s']) / 1e6 for r in stage]
print('stage launch: mean %.2f ms (min %.2f max %.2f); gap between consecutive launches: mean %.2f ms (min %.2f max %.2f); sum %.2f ms'
      % (sum(slen) / len(slen), min(slen), max(slen), sum(glen) / len(glen), min(glen), max(glen), sum(slen) / len(slen) + sum(glen) / len(glen)))
print('%-50s %8s %14s %14s' % ('kernel (inside the gaps)', 'per gap', 'us in the gap', 'us full length'))
n = len(gaps)
for k, (c, t, full) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print('%-50s %8.1f %14.1f %14.1f' % (k, c / n, t / n, full / n))
# the timeline of ONE gap in the middle of the run
a, b, inside = gaps[len(gaps) // 2]
print('one gap (%.2f ms), kernels by start time relative to the end of the stage launch:' % ((b['s'] - a['e']) / 1e6))
for r in inside:
    if r['e'] - r['s'] > 20e3:
        name = short(r['Kernel_Name'])
        if name.startswith('at::'):      # torch's own kernels: which functor (fill / copy / ...) and how many threads
            name = r['Kernel_Name'].replace('void ', '')[:150] + '  grid ' + str(r.get('Grid_Size', r.get('Grid_Size_X', '?')))
        print('   %+9.1f us  %8.1f us  q%s  %s' % ((r['s'] - a['e']) / 1e3, (r['e'] - r['s']) / 1e3, r['Queue_Id'], name))
