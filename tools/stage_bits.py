"""Development aid: bit-level fingerprint + time of the optimiser stage kernel, for A/B runs of two builds of the library on one box
(GLAMR_LIB_PATH selects the build).  Prints one sha1 per case over every output array of the stage (parameters, cameras, poses,
projections, losses) -- a change that only moves loads or barriers must keep every fingerprint -- and the launch time of 1 / 1024 scenes.
usage: python tools/stage_bits.py [--time] [--cases a,b,..]"""
import ctypes, hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from glamr_amd import _lib
from glamr_amd.global_recon import packing
from glamr_amd.global_recon.configs import get_config
from glamr_amd.utils import synth
from oracle.port import build
from oracle import make_golden as mg
from tests.grecon_common import j_local_from_oracle

CASES = {   # name: (cfg, frames, persons, seeds, gap)
    'dyn300': ('glamr_dynamic', 300, 1, (0, 1, 4, 6), True),
    'dyn300_nogap': ('glamr_dynamic', 300, 1, (0, 2), False),
    'dyn120': ('glamr_dynamic', 120, 1, (0, 3), True),
    'static300': ('glamr_static', 300, 1, (0,), True),
    '3dpw120': ('glamr_3dpw', 120, 1, (0,), False),
    'static_multi300': ('glamr_static_multi', 300, 4, (0,), True),
    'dyn_multi90': ('glamr_dynamic_multi', 90, 2, (0,), True),
}


def run_case(L, dev, root, md, name, iters=None):
    cfg_id, T, P, seeds, gap = CASES[name]
    cfg = get_config(cfg_id)
    ora = build.load_optimizer(root, cfg)
    datas, jls = [], []
    for s in seeds:
        in_dict = synth.make_in_dict(seed=s, num_frames=T, num_persons=P, smpl_model=md, gap=None if gap else (0, 0))
        d = ora.init_data(in_dict, latents=mg.latents_for(in_dict, s))
        datas.append(d); jls.append(j_local_from_oracle(ora.smpl, d))
    packed = packing.PackedScenes(datas, jls, dev)
    sb = packed.struct()
    ws = torch.empty(L.glamr_grecon_workspace_bytes(packed.S, packed.P, packed.T), dtype=torch.uint8, device=dev)
    h = hashlib.sha1()
    seen_wd = False
    for stage, spec in cfg['opt_stage_specs'].items():
        sd = packing.stage_desc(spec, cfg['grecon_model_specs'], seen_wd, niters=iters)
        seen_wd = seen_wd or ('world_dheading' in spec['opt_variables'])
        _lib.check(L.glamr_grecon_run_stage(ctypes.byref(sb), ctypes.byref(sd), None, _lib.ptr(ws), _lib.current_stream()))
        torch.cuda.synchronize()
        for k, v in sorted(packed.fetch().items()):
            h.update(np.ascontiguousarray(np.asarray(v)).tobytes())
    return h.hexdigest()[:16]


def main():
    dev = torch.device('cuda:0')
    root = build.ensure_synthetic_assets('/tmp/glamr_bench_assets')
    md = synth.make_smpl_model()
    L = _lib.lib()
    names = list(CASES)
    for a in sys.argv[1:]:
        if a.startswith('--cases='):
            names = a.split('=', 1)[1].split(',')
    for n in names:
        print('bits %-16s %s' % (n, run_case(L, dev, root, md, n)), flush=True)
    if '--time' in sys.argv:
        cfg = get_config('glamr_dynamic')
        ora = build.load_optimizer(root, cfg)
        in_dict = synth.make_in_dict(seed=0, num_frames=300, num_persons=1, smpl_model=md)
        d = ora.init_data(in_dict, latents=mg.latents_for(in_dict, 0))
        jl = j_local_from_oracle(ora.smpl, d)
        spec = cfg['opt_stage_specs']['init_opt']
        for S in (1, 1024):
            packed = packing.PackedScenes([d] * S, [jl] * S, dev)
            sd = packing.stage_desc(spec, cfg['grecon_model_specs'], False, niters=200)
            sb = packed.struct()
            ws = torch.empty(L.glamr_grecon_workspace_bytes(packed.S, packed.P, packed.T), dtype=torch.uint8, device=dev)
            best = 1e9
            for rep in range(4):
                torch.cuda.synchronize(); t0 = time.time()
                _lib.check(L.glamr_grecon_run_stage(ctypes.byref(sb), ctypes.byref(sd), None, _lib.ptr(ws), _lib.current_stream()))
                torch.cuda.synchronize(); best = min(best, time.time() - t0)
            print('time scenes=%4d  %.3f ms  %.2f us/iter' % (S, best * 1e3, best * 1e6 / 200), flush=True)


if __name__ == '__main__':
    main()
