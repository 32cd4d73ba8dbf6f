"""DEVELOPMENT AID (GPU): skinning WITH vertices timed alone -- B = 19 200 and 300 frames -- plus a fingerprint of the outputs, for A/B runs of
smpl.hip variants (GLAMR_LIB_PATH=tools/_lib_<name>.so from tools/build_variant_files.sh).  usage: python tools/smpl_ab.py [label]"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device('cuda:0')
model = bench.build_model(bench.ensure_assets(), dev)
smpl = model.smpl
g = torch.Generator().manual_seed(0)
label = sys.argv[1] if len(sys.argv) > 1 else os.environ.get('GLAMR_LIB_PATH', 'base')
res = []
for B in (19200, 300):
    pose = (torch.randn(B, 72, generator=g) * 0.3).to(dev)
    betas, trans = torch.randn(B, 10, generator=g).to(dev), torch.randn(B, 3, generator=g).to(dev)
    call = lambda: smpl(global_orient=pose[:, :3], body_pose=pose[:, 3:], betas=betas, root_trans=trans, return_verts=True)
    for _ in range(3):
        out = call()
    torch.cuda.synchronize()
    fp = hashlib.sha1(out.vertices.cpu().numpy().tobytes() + out.joints.cpu().numpy().tobytes()).hexdigest()[:12]
    ts = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20 if B > 1000 else 200
        e0.record()
        for _ in range(n):
            call()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n)
    res.append('B=%d %s ms fp %s' % (B, ' '.join('%.4f' % t for t in ts), fp))
print('%-28s %s' % (label, ' | '.join(res)))
