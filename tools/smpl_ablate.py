"""DEVELOPMENT AID (GPU): ablations of the skinning kernel on a build with -DGLAMR_SMPL_EXPERIMENT (GLAMR_LIB_PATH points at it):
GLAMR_SMPL_EXP bits: 1 = no vertex stores, 2 = no LDS transpose either, 4 = joint-transform fragments always from the same 48 KB (L1 / L2
resident), 8 = feature fragments likewise.  B = 19 200 with vertices; times per call from HIP events.
    GLAMR_EXTRA_FLAGS="smpl.hip=-DGLAMR_SMPL_EXPERIMENT" python -c "from glamr_amd import build; build.build_library(force=True)"   (build container)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device('cuda:0')
model = bench.build_model(bench.ensure_assets(), dev)
smpl = model.smpl
g = torch.Generator().manual_seed(0)
for B in (19200, 300):
    pose = (torch.randn(B, 72, generator=g) * 0.3).to(dev)
    betas, trans = torch.randn(B, 10, generator=g).to(dev), torch.randn(B, 3, generator=g).to(dev)
    for exp in [int(x) for x in os.environ.get('GLAMR_ABLATE', '0,1,3,4,8,12,15').split(',')]:
        os.environ['GLAMR_SMPL_EXP'] = str(exp)
        dt = bench._timed(lambda: smpl(global_orient=pose[:, :3], body_pose=pose[:, 3:], betas=betas, root_trans=trans, return_verts=True), reps=5)
        print('B = %5d  exp %2d: %.4f ms' % (B, exp, dt * 1e3), flush=True)
os.environ['GLAMR_SMPL_EXP'] = '0'
