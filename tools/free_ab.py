"""DEVELOPMENT AID (GPU): the priors with the LDS-free one-wave kernels (GLAMR_NETS_FREE=1, nn_free.hpp) against the fused LDS kernels
(=0) on 1024 x 300 frames: results of one against the other, time alone, and time BESIDE a resident optimiser stage (two streams) --
the number that decides the step."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from glamr_amd import _lib
from glamr_amd.global_recon import packing
from glamr_amd.global_recon.configs import get_config
from glamr_amd.models.priors import num_windows
from glamr_amd.utils import synth
from oracle.port import build
from oracle import make_golden as mg
from tests.grecon_common import j_local_from_oracle

dev = torch.device('cuda:0')
root = bench.ensure_assets()
m = bench.build_model(root, dev)
g = torch.Generator().manual_seed(0)
B, T = int(os.environ.get('GLAMR_AB_BATCH', '1024')), 300
pose = (torch.randn(B, T, 69, generator=g) * 0.2).to(dev)
vis = torch.ones(B, T, device=dev)
vis[:, 100:160] = 0
meps, teps = torch.randn(B, num_windows(T), 128, generator=g).to(dev), torch.randn(B, 128, generator=g).to(dev)


BUF = {}


def priors():
    """on the CURRENT stream; with resident buffers (fixed addresses: the library replays its captured graph from the second call on)"""
    h = m.mt_model.handle
    sid = torch.cuda.current_stream().cuda_stream
    if sid not in BUF:
        b = h.resident_set(B, T, num_windows(T))
        b['nets_pose'].copy_(pose); b['nets_vis'].copy_(vis); b['meps'].copy_(meps); b['teps'].copy_(teps)
        BUF[sid] = b
    b = BUF[sid]
    b['persistent'] = True
    return h.infer(b['nets_pose'], b['nets_vis'], [T] * B, motion_eps=b['meps'], traj_eps=b['teps'], buffers=b)


def outputs(free):
    os.environ['GLAMR_NETS_FREE'] = '1' if free else '0'
    out = m.mt_model.infer_padded(pose, vis, [T] * B, meps, teps)
    torch.cuda.synchronize()
    return [out[k].clone() for k in sorted(out)]


ref, new = outputs(False), outputs(True)
for i, (a, b) in enumerate(zip(ref, new)):
    print('output %d %s: max |free - fused| = %.3e (max |fused| %.3e)' % (i, tuple(a.shape), (a - b).abs().max().item(), a.abs().max().item()), flush=True)

# the optimiser stage of 1024 scenes x 500 iterations (as tools/coresidency_probe.py)
cfg = get_config('glamr_dynamic')
md = synth.make_smpl_model()
in_dict = synth.make_in_dict(seed=0, num_frames=300, num_persons=1, smpl_model=md)
ora = build.load_optimizer(root, cfg)
data = ora.init_data(in_dict, latents=mg.latents_for(in_dict, 0))
jl = j_local_from_oracle(ora.smpl, data)
L = _lib.lib()
S = 1024
packed = packing.PackedScenes([data] * S, [jl] * S, dev)
sd = packing.stage_desc(cfg['opt_stage_specs']['init_opt'], cfg['grecon_model_specs'], False, niters=500)
sb = packed.struct()
ws = torch.empty(L.glamr_grecon_workspace_bytes(packed.S, packed.P, packed.T), dtype=torch.uint8, device=dev)
s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)


def stage():
    _lib.check(L.glamr_grecon_run_stage(ctypes.byref(sb), ctypes.byref(sd), None, _lib.ptr(ws), ctypes.c_void_p(s1.cuda_stream)))


def timed(do_stage, do_priors):
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record(torch.cuda.current_stream())
    s1.wait_event(e0)
    s2.wait_event(e0)
    if do_stage:
        with torch.cuda.stream(s1):
            stage()
    if do_priors:
        with torch.cuda.stream(s2):
            priors()
    e1, e2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e1.record(s1)
    e2.record(s2)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1), e0.elapsed_time(e2)


stage()
torch.cuda.synchronize()
t_stage = min(timed(True, False)[0] for _ in range(2))
print('stage alone: %.2f ms' % t_stage, flush=True)
for free in (0, 1):
    os.environ['GLAMR_NETS_FREE'] = str(free)
    with torch.cuda.stream(s2):
        for _ in range(3):
            priors()                                   # graph of this variant captured and instantiated
    torch.cuda.synchronize()
    alone = min(timed(False, True)[1] for _ in range(3))
    both = [timed(True, True) for _ in range(3)]
    print('GLAMR_NETS_FREE=%d: priors alone %.2f ms | beside the stage: %s | sum %.2f' % (
        free, alone, '  '.join('stage %.2f priors %.2f' % b for b in both), t_stage + alone), flush=True)
