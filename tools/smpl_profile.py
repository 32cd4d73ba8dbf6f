"""DEVELOPMENT AID (GPU): SMPL skinning alone for a profiler -- B = 19 200 and 300 frames, with vertices and joints-only (the optimiser
path's call), plus the general backward; run under `rocprofv3 --kernel-trace --stats` or `--pmc ...` (tools/README.md)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device('cuda:0')
model = bench.build_model(bench.ensure_assets(), dev)
smpl = model.smpl
g = torch.Generator().manual_seed(0)
for B in (300, 19200):
    pose = (torch.randn(B, 72, generator=g) * 0.3).to(dev)
    betas, trans = torch.randn(B, 10, generator=g).to(dev), torch.randn(B, 3, generator=g).to(dev)
    for verts in (True, False):
        for _ in range(5):
            smpl(global_orient=pose[:, :3], body_pose=pose[:, 3:], betas=betas, root_trans=trans, return_verts=verts)
    bp = pose[:, 3:].clone().requires_grad_(True)
    for _ in range(3):
        out = smpl(global_orient=pose[:, :3], body_pose=bp, betas=betas, root_trans=trans, return_verts=False)
        out.joints.sum().backward()
torch.cuda.synchronize()
print('done')
