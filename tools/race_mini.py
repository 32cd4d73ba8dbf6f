"""DEVELOPMENT AID (GPU): the smallest form, inside this package, of the corruption root-caused in round 6 (profiles/r06_pipeline_corruption.log).
Stream A: the skinning alone (SMPL.root_relative_joints: smpl_prep_kernel, smpl_lbs_kernel, smpl_finish_kernel) on STATIC inputs, plain launches.
Stream B: the priors (plain launches), a rocBLAS GEMM, or nothing, started `delay` microseconds into A's call.  Every A result against A alone, bit for
bit.  With the shipped library (no packed-fp32 instructions) 0 runs differ; with GLAMR_VARIANT_PACKED=" " tools/build_variant_files.sh packed
"smpl.hip init.hip nets.hip eval.hip" and GLAMR_LIB_PATH=tools/_lib_packed.so every run does (B = nets).  tools/race_repro.hip is the torch-free form.
usage: python tools/race_mini.py [runs] [B kind: nets | mm | none]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from glamr_amd.utils import synth
from glamr_amd.models.prior_models import num_windows, NZ

dev = torch.device('cuda:0')
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 24
kind = sys.argv[2] if len(sys.argv) > 2 else 'nets'
S, T = 1024, 300
model = bench.build_model(bench.ensure_assets(), dev)
g = torch.Generator(device=dev).manual_seed(3)
pose = 0.3 * torch.randn((S * T, 69), device=dev, generator=g)
betas = 0.5 * torch.randn((S * T, 10), device=dev, generator=g)
sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
torch.cuda.synchronize()
with torch.cuda.stream(sa):
    ref = model.smpl.root_relative_joints(pose, betas).clone()
    ref2 = model.smpl.root_relative_joints(pose, betas).clone()
torch.cuda.synchronize()
assert torch.equal(ref, ref2)
nw = num_windows(T)
h = model.mt_model.handle
with torch.cuda.stream(sb):
    rs = h.resident_set(S, T, nw)
    rs['nets_pose'].copy_(0.2 * torch.randn((S, T, 69), device=dev, generator=g))
    rs['nets_vis'].fill_(1.0)
    rs['nets_vis'][:, 100:160] = 0.0
    rs['meps'].copy_(torch.randn((S, nw, NZ), device=dev, generator=g))
    rs['teps'].copy_(torch.randn((S, NZ), device=dev, generator=g))
    rs['persistent'] = True
lens = np.full(S, T, np.int32)
xa = torch.randn(8192, 8192, device=dev); xb = torch.randn(8192, 8192, device=dev); xc = torch.empty(8192, 8192, device=dev)
torch.cuda.synchronize()


def b_work():
    if kind == 'nets':
        model.mt_model.infer_padded(rs['nets_pose'], rs['nets_vis'], lens, rs['meps'], rs['teps'], buffers=rs, coschedule=True)
    elif kind == 'mm':
        torch.mm(xa, xb, out=xc)


with torch.cuda.stream(sb):
    b_work(); b_work()
torch.cuda.synchronize()
n_bad = 0
for it in range(runs):
    delay = 20 + 40 * (it % 12)
    with torch.cuda.stream(sb):
        torch.cuda._sleep(int(delay * 2100))
        b_work()
    with torch.cuda.stream(sa):
        got = model.smpl.root_relative_joints(pose, betas)
    torch.cuda.synchronize()
    d = (got != ref).flatten(1).any(dim=1)
    nb = int(d.sum())
    n_bad += nb > 0
    print('run %d (B %d us after A): %d of %d frames differ%s' % (it, delay, nb, S * T, (', first ' + str(torch.nonzero(d).flatten()[:8].tolist())) if nb else ''))
print('SUMMARY: %d of %d runs differ from the skinning alone (B = %s)' % (n_bad, runs, kind))
