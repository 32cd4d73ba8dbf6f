"""DEVELOPMENT AID (GPU): time of the two priors on 1024 x 300 frames in THIS process (HIP events, best of 5); run it twice in one gpurun
call with and without GLAMR_NETS_NO_FUSE=1 to compare kernel variants on the same box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from glamr_amd.models.priors import num_windows
dev = torch.device('cuda:0')
m = bench.build_model(bench.ensure_assets(), dev)
g = torch.Generator().manual_seed(0)
B, T = 1024, 300
pose = (torch.randn(B, T, 69, generator=g) * 0.2).to(dev); vis = torch.ones(B, T, device=dev); vis[:, 100:160] = 0
meps, teps = torch.randn(B, num_windows(T), 128, generator=g).to(dev), torch.randn(B, 128, generator=g).to(dev)
dt = bench._timed(lambda: m.mt_model.infer_padded(pose, vis, [T] * B, meps, teps), reps=5)
print('priors %s: %.2f ms' % ('UNFUSED' if os.environ.get('GLAMR_NETS_NO_FUSE') else 'fused', dt * 1e3))
# fingerprints of the outputs (first 16 hex digits of sha1 over the bytes): equal between two builds = bit-identical priors
import hashlib
out = m.mt_model.infer_padded(pose, vis, [T] * B, meps, teps)
torch.cuda.synchronize()
items = out.items() if isinstance(out, dict) else enumerate(out if isinstance(out, (tuple, list)) else [out])
for k, v in items:
    if torch.is_tensor(v): print('bits %s %s' % (k, hashlib.sha1(v.detach().cpu().numpy().tobytes()).hexdigest()[:16]))
