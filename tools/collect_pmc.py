"""GPU box: memory-side traffic and wave-cycle counters of the shipped optimiser-stage instance, from rocprofv3 PMC passes (one counter
group per pass, kernel trace only -- the combination gpurun allows), plus a calibration of FETCH_SIZE / WRITE_SIZE on a kernel whose
bytes are known and whose accesses are 4 bytes per lane like the stage kernel's (glamr_adam_step: reads 4 arrays, writes 3).

    python tools/collect_pmc.py            # writes gpurun_out/<tag>_pmc_*.csv and gpurun_out/<tag>_pmc_stage_kernel.json (tag: GLAMR_ROUND_TAG, default r03)
Copy the json / csv into profiles/ (bench.py reads the newest profiles/rNN_pmc_stage_kernel.json)."""
import csv, glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'gpurun_out')
KERNEL = 'grecon_stage_kernel<1, true, 1, 304>'
TAG = os.environ.get('GLAMR_ROUND_TAG', 'r03')          # file prefix: profiles are named per round
B, ITERS = 1024, 500


def run_pass(tag, counters, cmd):
    d = '/tmp/pmc_' + tag
    subprocess.run('rm -rf %s' % d, shell=True)
    full = 'cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc %s --kernel-trace --output-format csv -d %s -- %s > /dev/null 2>&1' % (' '.join(counters), d, cmd)
    subprocess.run(full, shell=True, check=False)
    files = glob.glob(d + '/*/*counter_collection.csv')
    rows = []
    for f in files:
        rows += list(csv.DictReader(open(f)))
    return rows


def per_kernel(rows, name_part):
    """{counter: [value per dispatch]} for dispatches whose kernel name contains name_part"""
    by = {}
    for r in rows:
        if name_part in r.get('Kernel_Name', ''):
            by.setdefault(r['Counter_Name'], {}).setdefault(r['Dispatch_Id'], 0.0)
            by[r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
    return {k: list(v.values()) for k, v in by.items()}


def main():
    os.makedirs(OUT, exist_ok=True)
    bench = 'python %s/bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-kernel-lines --no-host-stream' % ROOT
    calib = 'python %s/tools/collect_pmc.py calib' % ROOT
    res = {}
    raw = []
    for tag, ctrs, cmd in (('fetch', ['FETCH_SIZE'], bench), ('write', ['WRITE_SIZE'], bench),
                           ('sq', ['SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY'], bench),
                           ('inst', ['SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR', 'SQ_INSTS_SMEM'], bench),
                           ('cal_fetch', ['FETCH_SIZE'], calib), ('cal_write', ['WRITE_SIZE'], calib)):
        rows = run_pass(tag, ctrs, cmd)
        k = per_kernel(rows, 'adam_step_kernel' if tag.startswith('cal') else KERNEL)
        for c, vals in k.items():
            # the bench run launches this kernel on other batch sizes too (one sequence alone, the 64-sequence line): only the launches of the
            # FULL batch count -- those within 40 % of the largest value (rounds 3 - 4 averaged over all of them, which read ~25 % low)
            full = [v for v in vals if v > 0.6 * max(vals)] if not tag.startswith('cal') else vals
            res.setdefault(tag, {})[c] = {'launches': len(full), 'launches_of_any_size': len(vals), 'mean': sum(full) / max(1, len(full)), 'min': min(full), 'max': max(full)}
            raw.append((tag, c, len(full), sum(full) / max(1, len(full)), min(full), max(full)))
    with open(os.path.join(OUT, TAG + '_pmc_stage_kernel_counters.csv'), 'w') as f:
        f.write('pass,counter,launches,mean_per_launch,min,max\n')
        for r in raw:
            f.write('%s,%s,%d,%.6g,%.6g,%.6g\n' % r)
    # calibration: glamr_adam_step on N elements reads 16 N bytes and writes 12 N bytes (4-byte-per-lane accesses)
    N = 1 << 26
    cal = {}
    if 'cal_fetch' in res and 'FETCH_SIZE' in res['cal_fetch']:
        cal['fetch_kb_counted_per_kb_read'] = res['cal_fetch']['FETCH_SIZE']['mean'] / (16.0 * N / 1024)
    if 'cal_write' in res and 'WRITE_SIZE' in res['cal_write']:
        cal['write_kb_counted_per_kb_written'] = res['cal_write']['WRITE_SIZE']['mean'] / (12.0 * N / 1024)
    out = {'kernel': KERNEL, 'scenes': B, 'iterations': ITERS, 'counters': res, 'calibration': cal,
           'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on `bench.py --streams 1`, kernel %s, %d scenes x %d iterations per launch; '
                     'counter units calibrated on glamr_adam_step (4-byte-per-lane accesses, known byte count); tools/collect_pmc.py' % (KERNEL, B, ITERS)}
    try:
        fk = res['fetch']['FETCH_SIZE']['mean'] / cal.get('fetch_kb_counted_per_kb_read', 1.0)
        wk = res['write']['WRITE_SIZE']['mean'] / cal.get('write_kb_counted_per_kb_written', 1.0)
        out['fetch_bytes_per_launch'], out['write_bytes_per_launch'] = fk * 1024, wk * 1024
        out['bytes_per_scene_iteration'] = (fk + wk) * 1024 / (B * ITERS)
    except KeyError:
        pass
    json.dump(out, open(os.path.join(OUT, TAG + '_pmc_stage_kernel.json'), 'w'), indent=1)
    print(json.dumps(out, indent=1))


def calib():
    sys.path.insert(0, ROOT)
    import torch
    from glamr_amd import _lib
    L = _lib.lib()
    N = 1 << 26
    dev = torch.device('cuda:0')
    p, m, v = (torch.zeros(N, device=dev) for _ in range(3))
    g = torch.randn(N, device=dev)
    for step in (1, 2, 3):
        _lib.check(L.glamr_adam_step(N, _lib.ptr(p), _lib.ptr(m), _lib.ptr(v), _lib.ptr(g), 1e-3, step, _lib.current_stream()))
    torch.cuda.synchronize()


def priors():
    """Matrix-pipe utilisation of the priors' kernels (one call of both networks on 1024 x 300 frames, tools/priors_ab.py)."""
    os.makedirs(OUT, exist_ok=True)
    cmd = 'python %s/tools/priors_ab.py' % ROOT
    agg = {}
    for tag, ctrs in (('mfma', ['MfmaUtil']), ('mops', ['SQ_INSTS_VALU_MFMA_MOPS_F16', 'SQ_INSTS_VALU_MFMA_MOPS_F32']), ('busy', ['SQ_BUSY_CU_CYCLES', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY'])):
        for r in run_pass('pri_' + tag, ctrs, cmd):
            name = r.get('Kernel_Name', '')
            short = name.split('(')[0].replace('void ', '').replace('glamr::nn::', '').replace('(anonymous namespace)::', '')
            a = agg.setdefault(short, {})
            a.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
    with open(os.path.join(OUT, TAG + '_pmc_priors.csv'), 'w') as f:
        f.write('kernel,counter,dispatches,mean,sum\n')
        for k in sorted(agg):
            for cname, vals in sorted(agg[k].items()):
                f.write('"%s",%s,%d,%.6g,%.6g\n' % (k, cname, len(vals), sum(vals) / len(vals), sum(vals)))
    print(open(os.path.join(OUT, TAG + '_pmc_priors.csv')).read())


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'calib':
        calib()
    elif len(sys.argv) > 1 and sys.argv[1] == 'priors':
        priors()
    else:
        main()
