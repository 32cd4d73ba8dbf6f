"""Development aid: from a rocprofv3 --kernel-trace CSV of the default bench.py run, ONE batch's chain on its stream -- from its per-person
preparation to the end of its optimiser stage -- as runs of consecutive launches of the same kernel: start (ms after the chain's start), number
of launches, busy time, wall time of the run.  Shows what the step's critical chain is made of (memsets and copies included).
usage: python tools/chain_trace.py kernel_trace.csv [which chain, default: the middle one]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
short = lambda n: n.replace('void ', '').replace('glamr::', '').replace('(anonymous namespace)::', '').split('(')[0][:60]
preps = [r for r in rows if 'prep_person_kernel' in r['Kernel_Name']]
pick = preps[int(sys.argv[2]) if len(sys.argv) > 2 else len(preps) // 2]
q = pick['Queue_Id']
mine = [r for r in rows if r['Queue_Id'] == q and r['s'] >= pick['s']]
chain = []
for r in mine:
    chain.append(r)
    if 'grecon_stage_kernel<1, true, 1, 304>' in r['Kernel_Name'] and r['e'] - r['s'] > 5e6:
        break
t0 = chain[0]['s']
print('chain on queue %s: %d launches, %.2f ms from the first to the end of the stage launch' % (q, len(chain), (chain[-1]['e'] - t0) / 1e6))
print('%10s %6s %10s %10s  %s' % ('start ms', 'n', 'busy ms', 'wall ms', 'kernel'))
i = 0
while i < len(chain):
    j = i
    while j + 1 < len(chain) and short(chain[j + 1]['Kernel_Name']) == short(chain[i]['Kernel_Name']):
        j += 1
    busy = sum(r['e'] - r['s'] for r in chain[i:j + 1]) / 1e6
    wall = (chain[j]['e'] - chain[i]['s']) / 1e6
    if busy > 0.03 or wall > 0.1:
        print('%10.3f %6d %10.3f %10.3f  %s' % ((chain[i]['s'] - t0) / 1e6, j - i + 1, busy, wall, short(chain[i]['Kernel_Name'])))
    i = j + 1
