"""DEVELOPMENT AID (GPU box): cProfile of the pipelined host path (optimize_stream over 6 batches of 1024 sequences)."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import ensure_assets, build_model
from glamr_amd.utils import synth
root = ensure_assets(); dev = torch.device('cuda:0')
m = build_model(root, dev)
md = synth.make_smpl_model()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
base = [synth.make_in_dict(seed=s, num_frames=300, num_persons=1, smpl_model=md) for s in range(B)]
for _ in m.optimize_stream([base] * 4):          # warm-up: allocations (three pinned output sets of 400 MB), graph capture
    pass
torch.cuda.synchronize()
pr = cProfile.Profile(); t0 = time.time(); pr.enable()
n = sum(len(r) for r in m.optimize_stream([base] * 6))
pr.disable(); dt = time.time() - t0
print('%d sequences in %.3f s = %.0f seq/s (%.1f ms per batch)' % (n, dt, n / dt, dt / 6 * 1e3))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22); print(s.getvalue()[:4500])
