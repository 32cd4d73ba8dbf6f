"""Latency of ONE 300-frame 1-person sequence through GlobalReconOptimizer.optimize (host dictionary in, host dictionary out): the
reference's call, with nothing to batch.  Prints the median of a few runs and the optimiser-stage kernel time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from glamr_amd.utils import synth

dev = torch.device('cuda:0')
model = bench.build_model(bench.ensure_assets(), dev)
md = synth.make_smpl_model()
for P in (1, 4):
    d = synth.make_in_dict(seed=5, num_frames=bench.NUM_FRAMES, num_persons=P, smpl_model=md)
    model.optimize(d)
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.time()
        model.optimize(d)
        torch.cuda.synchronize()
        ts.append(time.time() - t0)
    print('optimize(in_dict): %d person(s), %d frames, cfg %s: median %.1f ms (min %.1f) -> %.1f sequences/s with one sequence in flight'
          % (P, bench.NUM_FRAMES, bench.CFG_ID, np.median(ts) * 1e3, min(ts) * 1e3, 1.0 / np.median(ts)), flush=True)
