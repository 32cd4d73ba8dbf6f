"""Development aid: compile grecon.hip to gfx950 assembly and print, for one kernel instance, registers / spills and the static
instruction mix of every barrier-delimited segment (the update-only iteration is the run of big segments in the middle).
usage: python tools/isa_stats.py [extra hipcc flags...]"""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = os.environ.get('GLAMR_ISA_KERNEL', '_ZN5glamr6grecon19grecon_stage_kernelILi1ELb1ELi1ELi304EEEvNS0_10KernelArgsE')
out = os.path.join(tempfile.gettempdir(), 'grecon_isa.s')
cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-hip-fp32-correctly-rounded-divide-sqrt', '-fno-slp-vectorize', '-Xclang', '-target-feature', '-Xclang', '-packed-fp32-ops',
       '--cuda-device-only', '-S', '-I' + os.path.join(ROOT, 'include'), os.path.join(ROOT, 'glamr_amd/csrc/grecon.hip'), '-o', out] + sys.argv[1:]
subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
s = open(out).read()
i = s.index(KERNEL + ':'); j = s.index('.Lfunc_end', i)
segs = [collections.Counter()]
for ln in s[i:j].split('\n'):
    t = ln.strip()
    if not t or t[0] in ';.':
        continue
    op = t.split()[0]
    if op == 's_barrier':
        segs.append(collections.Counter())
    else:
        segs[-1][op] += 1
m = re.search(re.escape(KERNEL) + r'\n(?:.*\n){0,60}?', s[j:])
meta = s[s.index('.amdhsa_kernel ' + KERNEL):]
md = s[s.index('.name:           ' + KERNEL) - 1500:s.index('.name:           ' + KERNEL) + 1500] if ('.name:           ' + KERNEL) in s else ''
sp = re.findall(r'\.vgpr_spill_count:\s+(\d+)', md)
print('vgpr_spill_count (metadata, nearest entries):', sp)
print('vgpr', re.search(r'\.amdhsa_next_free_vgpr (\d+)', meta).group(1), 'spills', re.search(r'; ScratchSize: (\d+)', s[j:j + 4000]).group(1) if re.search(r'; ScratchSize: (\d+)', s[j:j + 4000]) else '?')
def cls(c):
    v = sum(n for o, n in c.items() if o.startswith('v_'))
    div = c['v_ldexp_f32'] + c['v_frexp_exp_i32_f32_e32'] + c['v_frexp_mant_f32_e32'] + c['v_div_scale_f32'] + c['v_div_fmas_f32'] + c['v_div_fixup_f32']
    mov = sum(n for o, n in c.items() if o.startswith('v_mov') or o.startswith('v_pk_mov') or o.startswith('v_accvgpr'))
    pk = sum(n for o, n in c.items() if o.startswith('v_pk_') and not o.startswith('v_pk_mov'))
    lane = c['v_readlane_b32'] + c['v_readfirstlane_b32'] + c['v_writelane_b32']
    mem = sum(n for o, n in c.items() if o.startswith(('global_', 'flat_', 'scratch_', 'buffer_')))
    ds = sum(n for o, n in c.items() if o.startswith('ds_'))
    return sum(c.values()), v, div, mov, pk, lane, ds, mem
tot = [0] * 8
print('seg   all  valu  divscaf  mov  packed  lane   ds  mem')
for k, c in enumerate(segs):
    r = cls(c)
    if r[0] >= 150:
        print('%3d %5d %5d %7d %5d %6d %5d %4d %4d' % ((k,) + r))
    tot = [a + b for a, b in zip(tot, r)]
print('all %5d %5d %7d %5d %6d %5d %4d %4d' % tuple(tot))

# the iteration loop = the longest backward branch spanning exactly the update-only iteration's barriers (7 for one person)
lines = [t.strip() for t in s[i:j].split('\n') if t.strip() and t.strip()[0] != ';']
label_pos = {t.split(':')[0]: k for k, t in enumerate(lines) if re.match(r'^\.LBB\d+_\d+:', t)}
bar = [k for k, t in enumerate(lines) if t.split()[0] == 's_barrier']
NB = int(os.environ.get('GLAMR_ISA_LOOP_BARRIERS', '7'))
best = None
for k, t in enumerate(lines):
    op = t.split()[0]
    if op.startswith('s_cbranch') or op == 's_branch':
        tgt = t.split()[-1]
        if tgt in label_pos and label_pos[tgt] < k and sum(1 for b in bar if label_pos[tgt] < b < k) == NB:
            if best is None or k - label_pos[tgt] > best[1] - best[0]:
                best = (label_pos[tgt], k)
if best:
    c = collections.Counter(t.split()[0] for t in lines[best[0]:best[1]] if not t.startswith('.'))
    print('iteration loop: %d instructions, %d VALU (static; the joint loop inside runs ~14 times)' % (sum(c.values()), sum(n for o, n in c.items() if o.startswith('v_'))))
    # opcode histogram of the loop (GLAMR_ISA_HIST=1): the kernel is issue-bound on the SIMD with two waves, so its time follows this count
    if os.environ.get('GLAMR_ISA_HIST'):
        for o, n in c.most_common(45):
            print('   %-28s %5d' % (o, n))
    if os.environ.get('GLAMR_ISA_DUMP'):
        open(os.environ['GLAMR_ISA_DUMP'], 'w').write('\n'.join(lines[best[0]:best[1]]))
