"""DEVELOPMENT AID (GPU): the two-stream step pipeline in miniature -- per batch `priors` (the library's captured graph) then one
optimiser-stage launch, batches alternating between two streams -- with and without explicit staggering (a stream's priors start when
the OTHER stream's priors have finished, i.e. beside the other stream's stage launch).  Prints ms per batch for GLAMR_NETS_FREE=0/1."""
import ctypes
import os
import sys
import time

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
B, T = 1024, 300
pose = (torch.randn(B, T, 69, generator=g) * 0.2).to(dev)
vis = torch.ones(B, T, device=dev)
vis[:, 100:160] = 0
meps, teps = torch.randn(B, num_windows(T), 128, generator=g).to(dev), torch.randn(B, 128, generator=g).to(dev)
cfg = get_config('glamr_dynamic')
md = synth.make_smpl_model()
in_dict = synth.make_in_dict(seed=0, num_frames=300, num_persons=1, smpl_model=md)
ora = build.load_optimizer(root, cfg)
data = ora.init_data(in_dict, latents=mg.latents_for(in_dict, 0))
jl = j_local_from_oracle(ora.smpl, data)
L = _lib.lib()
packed = packing.PackedScenes([data] * B, [jl] * B, dev)
sd = packing.stage_desc(cfg['opt_stage_specs']['init_opt'], cfg['grecon_model_specs'], False, niters=500)
sb = packed.struct()
streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
wss = [torch.empty(L.glamr_grecon_workspace_bytes(packed.S, packed.P, packed.T), dtype=torch.uint8, device=dev) for _ in streams]
BUF = {}


def priors(st):
    h = m.mt_model.handle
    key = (st.cuda_stream, os.environ.get('GLAMR_NETS_FREE'))
    if key not in BUF:
        b = h.resident_set(B, T, num_windows(T))
        b['nets_pose'].copy_(pose); b['nets_vis'].copy_(vis); b['meps'].copy_(meps); b['teps'].copy_(teps)
        BUF[key] = dict(b)
        BUF[key]['ws'] = torch.empty_like(b['ws'])          # a workspace of its own per (stream, mode): separate graphs
    b = BUF[key]
    b['persistent'] = True
    return h.infer(b['nets_pose'], b['nets_vis'], [T] * B, motion_eps=b['meps'], traj_eps=b['teps'], buffers=b)


def priors_split(st, part):
    """part 0: the infiller only; part 1: the trajectory predictor on the infilled motion"""
    h = m.mt_model.handle
    key = (st.cuda_stream, os.environ.get('GLAMR_NETS_FREE'), 'split')
    if key not in BUF:
        b = h.resident_set(B, T, num_windows(T))
        b['nets_pose'].copy_(pose); b['nets_vis'].copy_(vis); b['meps'].copy_(meps); b['teps'].copy_(teps)
        BUF[key] = dict(b)
        BUF[key]['ws'] = torch.empty_like(b['ws'])
    b = BUF[key]
    b['persistent'] = True
    if part == 0:
        return h.infer(b['nets_pose'], b['nets_vis'], [T] * B, motion_eps=b['meps'], traj_eps=None, infill=True, traj=False, buffers=b)
    return h.infer(b['pose'], None, [T] * B, motion_eps=None, traj_eps=b['teps'], infill=False, traj=True, buffers=b)


def run(n, stagger):
    ev_prev = None
    torch.cuda.synchronize()
    t0 = time.time()
    for i in range(n):
        st = streams[i % 2]
        with torch.cuda.stream(st):
            if stagger and ev_prev is not None:
                st.wait_event(ev_prev)
            if stagger == 2:
                priors_split(st, 0)
                ev_prev = torch.cuda.Event()
                ev_prev.record(st)
                priors_split(st, 1)
            else:
                priors(st)
                ev_prev = torch.cuda.Event()
                ev_prev.record(st)
            _lib.check(L.glamr_grecon_run_stage(ctypes.byref(sb), ctypes.byref(sd), None, _lib.ptr(wss[i % 2]), ctypes.c_void_p(st.cuda_stream)))
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3


for free in os.environ.get('GLAMR_PP_FREE', '0,1').split(','):
    os.environ['GLAMR_NETS_FREE'] = free
    for st in streams:
        with torch.cuda.stream(st):
            for _ in range(3):
                priors(st)
    torch.cuda.synchronize()
    for st in streams:
        with torch.cuda.stream(st):
            for _ in range(3):
                priors_split(st, 0); priors_split(st, 1)
    torch.cuda.synchronize()
    for stagger in [int(v) for v in os.environ.get('GLAMR_PP_STAGGER', '0,1,2').split(',')]:
        run(4, stagger)
        print('GLAMR_NETS_FREE=%s %s: %.2f ms per batch (12 batches)  %.2f (second run)' % (
            free, ('free-running', 'staggered   ', 'staggered on the infiller'    )[stagger], run(12, stagger), run(12, stagger)), flush=True)
