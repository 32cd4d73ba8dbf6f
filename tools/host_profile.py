"""DEVELOPMENT AID (GPU box): where the host side of optimize_batch spends its time (wall clock of three calls in a row, then cProfile of
stage_inputs, init_resident, run_schedule and collect)."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import ensure_assets, build_model
from glamr_amd.utils import synth
root = ensure_assets(); dev = torch.device('cuda:0')
m = build_model(root, dev)
md = synth.make_smpl_model()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
in_dicts = [synth.make_in_dict(seed=s, num_frames=300, num_persons=1, smpl_model=md) for s in range(B)]
for i in range(3):
    t0 = time.time(); m.optimize_batch(in_dicts); dt = time.time() - t0
    print('optimize_batch call %d: %.1f ms = %.0f seq/s  ' % (i, dt * 1e3, B / dt), {k: round(v * 1e3, 1) for k, v in m.timings.items()})


def prof(name, fn, n=14):
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable(); t0 = time.time(); r = fn(); t_host = time.time() - t0; torch.cuda.synchronize(); dt = time.time() - t0; pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(n)
    print('%s: host %.1f ms, with the device %.1f ms' % (name, t_host * 1e3, dt * 1e3)); print(s.getvalue()[:3000])
    return r


rin = prof('stage_inputs', lambda: m.stage_inputs(in_dicts))
datas, packed = prof('init_resident', lambda: m.init_resident(rin, init_forward=False), 24)
prof('run_schedule', lambda: m.run_schedule(packed))
prof('collect', lambda: m.collect(datas, packed), 18)
