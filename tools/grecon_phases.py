"""Development aid: per-phase time of one optimiser iteration (workgroup 0), from a build with -DGLAMR_PHASE_TIMING:

    GLAMR_EXTRA_FLAGS="grecon.hip=-DGLAMR_PHASE_TIMING" python -c "from glamr_amd import build; build.build_library(force=True)"
    python tools/grecon_phases.py [scenes]
"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glamr_amd import _lib
from glamr_amd.global_recon import packing
from glamr_amd.global_recon.configs import get_config
from glamr_amd.utils import synth
from oracle.port import build
from oracle import make_golden as mg
from tests.grecon_common import j_local_from_oracle

NAMES = ['A heading + scan + barrier', 'B displacement + scan', 'C/D world pose, camera + barrier', 'camera from persons', 'E5 own-camera backward + Adam',
         'G shared-camera gradients', 'H orientation reverse + scan + barrier', 'I displacement reverse + scan', 'J heading reverse (+ loss reduce)',
         'E1 loads, camera-relative orientation', 'E2 keypoints', 'E3 smoothness, relative transforms, fold', 'E4 camera smoothness terms', 'H1 loads, orientation reverse (up to the updates)', 'H2 eight Adam updates + stores', '-']
dev = torch.device('cuda:0')
root = build.ensure_synthetic_assets('/tmp/glamr_bench_assets')
cfg_id = sys.argv[2] if len(sys.argv) > 2 else 'glamr_dynamic'
cfg = get_config(cfg_id)
NP = int(os.environ.get('GLAMR_PH_PERSONS', '1'))
in_dict = synth.make_in_dict(seed=0, num_frames=int(os.environ.get('GLAMR_PH_FRAMES', '300')), num_persons=NP, smpl_model=synth.make_smpl_model())
ora = build.load_optimizer(root, cfg)
data = ora.init_data(in_dict, latents=mg.latents_for(in_dict, 0))
jl = j_local_from_oracle(ora.smpl, data)
L = _lib.lib()
fn = L.glamr_debug_phase_ticks
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
spec = cfg['opt_stage_specs'][os.environ['GLAMR_PH_STAGE']] if 'GLAMR_PH_STAGE' in os.environ else next(iter(cfg['opt_stage_specs'].values()))
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
packed = packing.PackedScenes([data] * S, [jl] * S, dev)
sd = packing.stage_desc(spec, cfg['grecon_model_specs'], False)
sb = packed.struct()
ws = torch.empty(L.glamr_grecon_workspace_bytes(packed.S, packed.P, packed.T), dtype=torch.uint8, device=dev)
for rep in range(2):
    _lib.check(L.glamr_grecon_run_stage(ctypes.byref(sb), ctypes.byref(sd), None, _lib.ptr(ws), _lib.current_stream()))
    torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 16)()
assert fn(out) == 0
tot = sum(out)
print('%s, %d scenes, %d iterations: %.1f us/iteration in the loop' % (cfg_id, S, sd.niters, tot * 0.01 / sd.niters))
for n, t in zip(NAMES, out):
    print('  %-42s %6.2f us  %4.1f%%' % (n, t * 0.01 / sd.niters, 100.0 * t / max(1, tot)))
