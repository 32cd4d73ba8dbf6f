"""Development aid: where does an optimiser iteration spend its time?  Runs the stage kernel with parts of the loss switched off."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glamr_amd import _lib
from glamr_amd.global_recon import packing
from glamr_amd.global_recon.configs import get_config
from glamr_amd.utils import synth
from oracle.port import build
from oracle import make_golden as mg
from tests.grecon_common import j_local_from_oracle

dev = torch.device('cuda:0')
root = build.ensure_synthetic_assets('/tmp/glamr_bench_assets')
cfg = get_config('glamr_dynamic')
in_dict = synth.make_in_dict(seed=0, num_frames=300, num_persons=1, smpl_model=synth.make_smpl_model())
ora = build.load_optimizer(root, cfg)
data = ora.init_data(in_dict, latents=mg.latents_for(in_dict, 0))
jl = j_local_from_oracle(ora.smpl, data)
L = _lib.lib()
spec = cfg['opt_stage_specs']['init_opt']
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
def run(name, mod):
    packed = packing.PackedScenes([data] * S, [jl] * S, dev)
    sd = packing.stage_desc(spec, cfg['grecon_model_specs'], False)
    mod(sd)
    sb = packed.struct()
    ws = torch.empty(L.glamr_grecon_workspace_bytes(packed.S, packed.P, packed.T), dtype=torch.uint8, device=dev)
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        _lib.check(L.glamr_grecon_run_stage(ctypes.byref(sb), ctypes.byref(sd), None, _lib.ptr(ws), _lib.current_stream()))
        torch.cuda.synchronize(); best = min(best, time.time() - t0)
    print('%-28s %7.2f ms  %6.1f us/iter' % (name, best * 1e3, best * 1e6 / max(1, sd.niters)))
def off(*names):
    def f(sd):
        for n in names: sd.loss_mask &= ~(1 << packing.LOSS_IDS[n])
    return f
run('full', lambda sd: None)
run('no kp_2d(+dist)', off('kp_2d', 'kp_2d_dist'))
run('no cam_traj_rot', off('cam_traj_rot'))
run('no kp, no cam_traj_rot', off('kp_2d', 'kp_2d_dist', 'cam_traj_rot'))
run('no losses at all', lambda sd: setattr(sd, 'loss_mask', 0))
def novars(sd): sd.loss_mask = 0; sd.var_mask = 1
run('no losses, cam var only', novars)
