"""DEVELOPMENT AID (GPU): host time between the batches optimize_stream yields, for a long stream (is the pipeline steady?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import ensure_assets, build_model
from glamr_amd.utils import synth
root = ensure_assets(); dev = torch.device('cuda:0')
m = build_model(root, dev)
md = synth.make_smpl_model()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 30
base = [synth.make_in_dict(seed=s, num_frames=300, num_persons=1, smpl_model=md) for s in range(B)]
import gc
MODE = os.environ.get('PROBE_GC', '')
if MODE == 'off':
    gc.disable()
for rep in range(2):
    t0 = time.time(); ts = []
    for r in m.optimize_stream([base] * NB):
        ts.append((time.time() - t0) * 1e3)
        if MODE == 'each':
            del r; gc.collect()
    d = [b - a for a, b in zip(ts[:-1], ts[1:])]
    print('pass %d: %d batches in %.1f ms = %.0f seq/s; between yields (ms): %s' % (rep, NB, ts[-1], B * NB / ts[-1] * 1e3, ' '.join('%.0f' % x for x in d)))
    s = sorted(d)
    print('   median %.1f ms = %.0f seq/s' % (s[len(s) // 2], B / s[len(s) // 2] * 1e3))
