"""Kernel micro-benchmarks on the GPU box (development aid): fused optimiser stage and SMPL LBS."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glamr_amd import _lib
from glamr_amd.global_recon import packing
from glamr_amd.global_recon.configs import get_config
from glamr_amd.utils import synth
from oracle.port import build
from oracle import make_golden as mg
from tests.grecon_common import j_local_from_oracle


def main():
    dev = torch.device('cuda:0')
    root = build.ensure_synthetic_assets('/tmp/glamr_bench_assets')
    cfg = get_config(os.environ.get('GLAMR_MB_CFG', 'glamr_dynamic'))
    md = synth.make_smpl_model()
    T = int(os.environ.get('GLAMR_MB_FRAMES', '300'))
    NP = int(os.environ.get('GLAMR_MB_PERSONS', '1'))
    in_dict = synth.make_in_dict(seed=0, num_frames=T, num_persons=NP, smpl_model=md)
    ora = build.load_optimizer(root, cfg)
    data = ora.init_data(in_dict, latents=mg.latents_for(in_dict, 0))
    jl = j_local_from_oracle(ora.smpl, data)
    L = _lib.lib()
    spec = cfg['opt_stage_specs'][os.environ.get('GLAMR_MB_STAGE', 'init_opt')]
    has_wd = os.environ.get('GLAMR_MB_STAGE', 'init_opt') != 'init_opt' and any('world_dheading' in s_['opt_variables'] for s_ in cfg['opt_stage_specs'].values())
    for S in [int(x) for x in os.environ['GLAMR_MB_SCENES'].split(',')] if 'GLAMR_MB_SCENES' in os.environ else ((1, 256) if 'GLAMR_MB_FRAMES' in os.environ else (1, 64, 256, 1024)):
        packed = packing.PackedScenes([data] * S, [jl] * S, dev)
        sd = packing.stage_desc(spec, cfg['grecon_model_specs'], False, niters=min(200, spec['opt_niters']))
        sb = packed.struct()
        ws = torch.empty(L.glamr_grecon_workspace_bytes(packed.S, packed.P, packed.T), dtype=torch.uint8, device=dev)
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.time()
            _lib.check(L.glamr_grecon_run_stage(ctypes.byref(sb), ctypes.byref(sd), None, _lib.ptr(ws), _lib.current_stream()))
            torch.cuda.synchronize()
            dt = time.time() - t0
        print('grecon stage: scenes=%4d  T=%d  P=%d  iters=%d  %.2f ms  (%.2f us/iter, %.1f scenes/s)' % (S, T, NP, sd.niters, dt * 1e3, dt * 1e6 / sd.niters, S / dt))
    if 'GLAMR_MB_FRAMES' in os.environ:
        return
    # SMPL LBS
    from glamr_amd.lib.models.smpl import SMPL
    smpl = SMPL(os.path.join(root, 'data', 'body_models', 'smpl'), pose_type='body26fk', extra_regressor_path=os.path.join(root, 'data', 'J_regressor_extra.npy')).to(dev)
    for B in (300, 19200):
        pose = torch.randn(B, 72, device=dev) * 0.3
        betas = torch.randn(B, 10, device=dev)
        trans = torch.randn(B, 3, device=dev)
        for verts in (True, False):
            for rep in range(3):
                torch.cuda.synchronize()
                t0 = time.time()
                out = smpl(global_orient=pose[:, :3], body_pose=pose[:, 3:], betas=betas, root_trans=trans, return_verts=verts)
                torch.cuda.synchronize()
                dt = time.time() - t0
            print('smpl forward: B=%5d verts=%d  %.3f ms  (%.1f TFLOP/s algorithmic at 15.8 MFLOP/frame)' % (B, verts, dt * 1e3, 15.8e6 * B / dt / 1e12))


if __name__ == '__main__':
    main()
