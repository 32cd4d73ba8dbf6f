"""Development aid: one launch of the optimiser stage kernel (for rocprofv3 --pmc runs)."""
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
dev = torch.device('cuda:0')
root = build.ensure_synthetic_assets('/tmp/glamr_bench_assets')
cfg = get_config('glamr_dynamic')
in_dict = synth.make_in_dict(seed=0, num_frames=300, num_persons=1, smpl_model=synth.make_smpl_model())
ora = build.load_optimizer(root, cfg)
data = ora.init_data(in_dict, latents=mg.latents_for(in_dict, 0))
jl = j_local_from_oracle(ora.smpl, data)
L = _lib.lib()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
packed = packing.PackedScenes([data] * S, [jl] * S, dev)
sd = packing.stage_desc(cfg['opt_stage_specs']['init_opt'], cfg['grecon_model_specs'], False)
sb = packed.struct()
ws = torch.empty(L.glamr_grecon_workspace_bytes(packed.S, packed.P, packed.T), dtype=torch.uint8, device=dev)
_lib.check(L.glamr_grecon_run_stage(ctypes.byref(sb), ctypes.byref(sd), None, _lib.ptr(ws), _lib.current_stream()))
torch.cuda.synchronize()
print('done')
