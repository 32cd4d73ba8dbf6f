"""DEVELOPMENT AID (GPU box): where the host side of optimize_batch spends its time (cProfile of stage_inputs and collect)."""
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
m.optimize_batch(in_dicts)
for name, fn in (('stage_inputs', lambda: m.stage_inputs(in_dicts)),):
    pr = cProfile.Profile(); pr.enable(); t0 = time.time(); rin = fn(); torch.cuda.synchronize(); dt = time.time() - t0; pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(14); print(name, '%.1f ms' % (dt * 1e3)); print(s.getvalue()[:2500])
datas, packed = m.init_resident(rin); m.run_schedule(packed); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable(); t0 = time.time(); m.collect(datas, packed); dt = time.time() - t0; pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(18); print('collect %.1f ms' % (dt * 1e3)); print(s.getvalue()[:3500])
