import os, sys, time, resource
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import torch
from bench import ensure_assets, build_model
from glamr_amd.utils import synth
root = ensure_assets(); dev = torch.device('cuda:0')
m = build_model(root, dev)
md = synth.make_smpl_model()
base = [synth.make_in_dict(seed=s, num_frames=300, num_persons=1, smpl_model=md) for s in range(512)]
def rss(): return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024
for rep in range(3):
    t0 = time.time(); n = 0
    for r in m.optimize_stream([base] * 60):
        n += len(r)
    dt = time.time() - t0
    print('pass %d: %d sequences in %.2f s = %.0f seq/s; device allocated %.0f MB reserved %.0f MB; host max RSS %.0f MB' % (rep, n, dt, n / dt, torch.cuda.memory_allocated() / 2**20, torch.cuda.memory_reserved() / 2**20, rss()))
