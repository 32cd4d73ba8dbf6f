"""DEVELOPMENT AID (build container only).  Where does the fused optimiser's Adam trajectory leave the reference's?

Runs the kernel algorithm (glamr_amd/csrc/grecon_algo.hpp) through the CPU test runtime (tests/hostsim) from the REFERENCE's own
initial state and compares parameters / gradients with the per-iteration record of tools/ref_trace.py.

    python tools/ref_trace.py gap /tmp/ref_trace_gap.npz
    python tools/diverge_probe.py /tmp/ref_trace_gap.npz [gap|nogap] [--flags "-DX ..."]
"""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

DATA_SEED = int(os.environ.get('GLAMR_TRACE_DATA_SEED', '0'))      # seed of the synthetic sequence (and of its latent draws)


def reference_init(which):
    from oracle import ref_harness as rh
    from oracle import make_golden as mg
    from glamr_amd.utils import synth
    rh.setup()
    model, cfg = rh.reference_optimizer('glamr_dynamic', log=rh.QuietLog())
    in_dict = synth.make_in_dict(seed=DATA_SEED, num_frames=300, num_persons=1, smpl_model=synth.make_smpl_model(), gap=None if which == 'gap' else (0, 0))
    specs = {}          # no stage: init_data only
    data, _ = mg.run_reference(model, specs, in_dict, mg.latents_for(in_dict, DATA_SEED))
    return model, data, in_dict


def kernel_trace(data, smpl, extra_flags=(), niters=None):
    from tests import hostsim, grecon_common as gc
    from glamr_amd import _lib
    from glamr_amd.global_recon import packing
    from glamr_amd.global_recon.configs import get_config
    cfg = get_config('glamr_dynamic')
    spec = cfg['opt_stage_specs']['init_opt']
    lib = hostsim.build('grecon_host', extra_flags=extra_flags)
    fn = lib.hostsim_grecon_trace_stage
    fn.argtypes = [ctypes.POINTER(_lib.SceneBatch), ctypes.POINTER(_lib.StageDesc), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    jl = gc.j_local_from_oracle(smpl, data)
    packed = packing.PackedScenes([data], [jl], torch.device('cpu'))
    n = niters or spec['opt_niters']
    sd = packing.stage_desc(spec, cfg['grecon_model_specs'], False, niters=n)
    stride = packed.layout['scene_stride']
    P = torch.zeros(n, stride)
    G = torch.zeros(n, stride)
    scratch = torch.zeros_like(packed.t['params'])
    sb = packed.struct()
    assert fn(ctypes.byref(sb), ctypes.byref(sd), scratch.data_ptr(), P.data_ptr(), G.data_ptr()) == 0
    return packed, P.numpy(), G.numpy()


def split(packed, vec, T=300):
    l = packed.layout
    o = {}
    o['cam_rot_6d'] = vec[..., l['cam_rot6d']:l['cam_rot6d'] + 6 * T]
    o['cam_trans'] = vec[..., l['cam_trans']:l['cam_trans'] + 3 * T]
    b = l['person0']
    o['p0_traj_local_xy'] = vec[..., b + l['local_xy']:b + l['local_xy'] + 2]
    o['p0_traj_local_heading'] = vec[..., b + l['local_heading']:b + l['local_heading'] + 1]
    o['p0_traj_local_rot'] = vec[..., b + l['local_rot']:b + l['local_rot'] + 6 * T]
    o['p0_world_dheading'] = vec[..., b + l['world_dheading']:b + l['world_dheading'] + T]
    return o


def main(argv):
    tr = np.load(argv[0])
    which = argv[1] if len(argv) > 1 and not argv[1].startswith('--') else 'gap'
    flags = argv[argv.index('--flags') + 1].split() if '--flags' in argv else []
    model, data, in_dict = reference_init(which)
    from oracle.port import build as ob
    from bench import ensure_assets
    smpl = ob.load_smpl(ensure_assets())
    packed, P, G = kernel_trace(data, smpl, flags)
    names, sizes = list(tr['names']), list(tr['sizes'])
    off = np.cumsum([0] + sizes)
    mine_p, mine_g = split(packed, P), split(packed, G)
    vis = np.asarray(data['person_data'][0]['vis_frames'])
    print('iteration: per-parameter max |dp| (kernel - reference), max |dg| / max|g|')
    for it in list(range(0, 20)) + list(range(20, 500, 20)) + [499]:
        row = []
        for i, nm in enumerate(names):
            rp, rg = tr['p'][it, off[i]:off[i + 1]], tr['g'][it, off[i]:off[i + 1]]
            dp = np.abs(mine_p[nm][it] - rp).max()
            dg = np.abs(mine_g[nm][it] - rg).max() / max(np.abs(rg).max(), 1e-30)
            row.append('%s %.1e/%.1e' % (nm.replace('p0_traj_', '').replace('p0_', ''), dp, dg))
        print('%3d  ' % it + '  '.join(row))
    np.savez('/tmp/kernel_trace.npz', P=P, G=G)
    kp = packed.t['kp_2d_pred'][0, :300].numpy()
    d = np.abs(kp - tr['fin_p0_kp_2d_pred'])[vis].max(axis=(1, 2))
    print('final projected keypoints vs the reference: max %.4f px, median frame %.4f px, frames > 1 px: %d of %d' % (d.max(), np.median(d), int((d > 1).sum()), len(d)))
    return packed, P, G, tr


if __name__ == '__main__':
    main(sys.argv[1:])
