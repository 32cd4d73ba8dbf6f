"""DEVELOPMENT AID (GPU): BASELINE configs[3] -- 64 scenes of 4 persons x 300 frames, cfg glamr_static_multi -- stage launch times and us per scene-iteration
(as bench.py's `configs3_four_persons_shared_camera`).  usage: python tools/configs3_time.py [label]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from glamr_amd.utils import synth
dev = torch.device('cuda:0')
root = bench.ensure_assets()
m4 = bench.build_model(root, dev, 'glamr_static_multi')
md = synth.make_smpl_model()
B4 = 64
in_dicts = [synth.make_in_dict(seed=1000 + s, num_frames=bench.NUM_FRAMES, num_persons=4, smpl_model=md) for s in range(B4)]
rin = m4.stage_inputs(in_dicts)
best = None
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    _, packed = m4.optimize_resident(rin)
    torch.cuda.synchronize(); dt = time.time() - t0
    best = dt if best is None else min(best, dt)
stage_ms = [m4.launch_ms(ws) for ws in packed.stage_ws]
iters = [s['opt_niters'] for s in m4.opt_stage_specs.values()]
import hashlib
fp = hashlib.sha1(packed.t['kp_2d_pred'].cpu().numpy().tobytes()).hexdigest()[:12]
print('%-8s %.1f scenes/s  %.2f ms | stage launches %s ms | us per scene-iteration %s | fp %s' % (sys.argv[1] if len(sys.argv) > 1 else '', B4 / best, best * 1e3,
      [round(x, 2) for x in stage_ms], [round(x * 1e3 / n, 1) for x, n in zip(stage_ms, iters)], fp))
