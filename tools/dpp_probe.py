"""DEVELOPMENT AID (GPU): the K-step state of one fixture case (tests/grecon_common.check_case's flow) saved per library build, to compare two
builds variable by variable.  usage: GLAMR_LIB_PATH=... python tools/dpp_probe.py out.npz [cfg_id T P K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import ensure_assets
from glamr_amd.global_recon import packing
from glamr_amd.global_recon.configs import get_config
from glamr_amd.utils import synth
from oracle import make_golden as mg
from oracle.port import build
from tests import grecon_common as gc
out = sys.argv[1]
cfg_id, T, P, K = (sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else ('glamr_3dpw', 120, 1, 15)
root = ensure_assets()
run, dev = gc.device_runner()
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'grecon_%s_T%d_P%d.npz' % (cfg_id, T, P)))
cfg = get_config(cfg_id); specs = cfg['grecon_model_specs']
in_dict = synth.make_in_dict(seed=3, num_frames=T, num_persons=P, smpl_model=synth.make_smpl_model())
ora = build.load_optimizer(root, cfg)
data = ora.init_data(in_dict, latents=mg.latents_for(in_dict, 3))
jl = gc.j_local_from_oracle(ora.smpl, data)
has_wd = False
save = {}
for stage, spec in cfg['opt_stage_specs'].items():
    for k in (1, 2, 3, 5, 8, min(K, spec['opt_niters'])):
        packed = packing.PackedScenes([data], [jl], dev)
        sd = packing.stage_desc(spec, specs, has_wd, niters=k)
        run(packed, sd, False)
        save['%s_it%d_params' % (stage, k)] = packed.t['params'].cpu().numpy()
        save['%s_it%d_kp' % (stage, k)] = packed.t['kp_2d_pred'].cpu().numpy()
    packed.unpack_into([data], spec, specs)
    has_wd = has_wd or 'world_dheading' in spec['opt_variables']
    break      # first stage only
vis = g['init_p0_vis_frames']
save['ref_kp'] = g['opt_p0_kp_2d_pred']; save['vis'] = vis
np.savez(out, **save)
print('saved', out)
