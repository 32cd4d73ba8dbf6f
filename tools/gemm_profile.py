"""Development aid: joins the GEMM shapes logged by GLAMR_GEMM_LOG=1 with rocprofv3's kernel trace (same launch order).

    cd /tmp && GLAMR_GEMM_LOG=1 rocprofv3 --kernel-trace --output-format csv -d OUT -- python tools/gemm_profile.py run 2> shapes.log
    python tools/gemm_profile.py join shapes.log OUT/*/*_kernel_trace.csv
"""
import csv, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == 'run':
    import torch
    from bench import ensure_assets, build_model
    from glamr_amd.utils import synth
    root = ensure_assets(); dev = torch.device('cuda:0')
    model = build_model(root, dev)
    md = synth.make_smpl_model()
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    in_dicts = [synth.make_in_dict(seed=s, num_frames=300, num_persons=1, smpl_model=md) for s in range(B)]
    rin = model.stage_inputs(in_dicts)
    model.init_resident(rin); torch.cuda.synchronize()
    sys.stderr.write('MARK\n'); sys.stderr.flush()
    model.init_resident(rin); torch.cuda.synchronize()
else:
    lines = open(sys.argv[2]).read().split('\n')
    k = lines.index('MARK')
    shapes = [tuple(int(x) for x in l.split()[1:]) for l in lines[k + 1:] if l.startswith('GEMM')]
    rows = [r for r in csv.DictReader(open(sys.argv[3])) if 'gemm_' in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    rows = rows[len(rows) - len(shapes):]
    agg = {}
    for sh, r in zip(shapes, rows):
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9
        a = agg.setdefault(sh, [0, 0.0]); a[0] += 1; a[1] += d
    tot_t = sum(a[1] for a in agg.values()); tot_f = sum(2.0 * m * n * k * a[0] for (m, n, k), a in agg.items())
    print('%d GEMM launches, %.2f ms, %.2f GFLOP, %.1f TFLOP/s overall' % (len(shapes), tot_t * 1e3, tot_f * 1e-9, tot_f / tot_t * 1e-12))
    for (m, n, k), a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('  M=%6d N=%5d K=%5d  x%4d  %8.1f us each  %6.1f TFLOP/s  %5.1f%% of GEMM time' % (m, n, k, a[0], a[1] / a[0] * 1e6, 2.0 * m * n * k * a[0] / a[1] * 1e-12, 100 * a[1] / tot_t))
