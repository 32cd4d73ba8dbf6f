"""Shared driver of the fused-optimiser parity tests: runs a reference-fixture case through a `runner` (host runtime of
tests/hostsim on the CPU, or the real libglamr_hip.so on an MI355X) and checks losses, gradients and the K-step state."""
import ctypes
import numpy as np
import torch

from oracle import make_golden as mg
from oracle.port import build
from glamr_amd import _lib
from glamr_amd.global_recon import packing
from glamr_amd.global_recon.configs import get_config
from glamr_amd.utils import synth


def _rot_err(aa_a, aa_b):
    """Axis-angle vectors are compared as ROTATIONS: near an angle of pi the same rotation has two far-apart axis-angle forms."""
    from oracle.port import transforms as tf
    Ra = tf.aa_to_rotmat(torch.as_tensor(np.asarray(aa_a), dtype=torch.float32))
    Rb = tf.aa_to_rotmat(torch.as_tensor(np.asarray(aa_b), dtype=torch.float32))
    return float((Ra - Rb).abs().max())


def kp_err(got, ref, frames=None):
    """Largest projection difference in pixels over the well-conditioned points.  A joint that comes within millimetres of the camera
    plane projects to 1e4..1e6 px (it happens to persons 0 and 3 of the synthetic 4-person scene); there a rounding difference in
    depth is hundreds of pixels and says nothing, so those points are left to the gradient / loss checks."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    if frames is not None:
        got, ref = got[frames], ref[frames]
    ok = np.abs(ref).max(axis=-1) < 2.5e3          # 1920 x 1080 images: beyond that the point is outside the picture anyway
    return float(np.abs(got - ref).max(axis=-1)[ok].max()) if ok.any() else 0.0


def j_local_from_oracle(smpl, data):
    out = {}
    for idx, pd in data['person_data'].items():
        T = pd['smpl_pose'].shape[0]
        z = torch.zeros(T, 3)
        with torch.no_grad():
            out[idx] = smpl(global_orient=z, body_pose=pd['smpl_pose'], betas=pd['smpl_beta'], root_trans=z).joints
    return out


def hostsim_runner():
    from tests import hostsim
    lib = hostsim.build('grecon_host')
    fn = lib.hostsim_grecon_run_stage
    fn.argtypes = [ctypes.POINTER(_lib.SceneBatch), ctypes.POINTER(_lib.StageDesc), ctypes.c_void_p]

    def run(packed, sd, want_grads):
        sb = packed.struct()
        grads = torch.zeros_like(packed.t['params']) if want_grads else None
        assert fn(ctypes.byref(sb), ctypes.byref(sd), ctypes.c_void_p(grads.data_ptr()) if want_grads else None) == 0
        return grads
    return run, torch.device('cpu')


def device_runner():
    L = _lib.lib()
    dev = torch.device('cuda:0')

    def run(packed, sd, want_grads):
        sb = packed.struct()
        grads = torch.zeros_like(packed.t['params']) if want_grads else None
        ws = torch.empty(L.glamr_grecon_workspace_bytes(packed.S, packed.P, packed.T), dtype=torch.uint8, device=dev)
        _lib.check(L.glamr_grecon_run_stage(ctypes.byref(sb), ctypes.byref(sd), _lib.ptr(grads), _lib.ptr(ws), _lib.current_stream()))
        torch.cuda.synchronize()
        return grads.cpu() if want_grads else None
    return run, dev


def grads_by_name(packed, grads, data, model_specs, opt_variables, si=0):
    l, T = packed.layout, packed.T
    g = grads[si]
    out = {}
    Ts = int(data['seq_len'])
    if 'cam' in opt_variables:
        if model_specs.get('flag_fixed_cam', False):
            out['cam_rot_6d_fix'] = g[l['cam_rot6d']:l['cam_rot6d'] + 6].view(1, 6)
            out['cam_trans_fix'] = g[l['cam_trans']:l['cam_trans'] + 3].view(1, 3)
        else:
            out['cam_rot_6d'] = g[l['cam_rot6d']:l['cam_rot6d'] + 6 * T].view(T, 6)[:Ts]
            out['cam_trans'] = g[l['cam_trans']:l['cam_trans'] + 3 * T].view(T, 3)[:Ts]
    else:
        empty = torch.where(data['fr_num_persons'] == 0)[0]
        out['cam_inv_rot_residual'] = g[l['cam_inv_rot_res']:l['cam_inv_rot_res'] + 6 * T].view(T, 6)[empty]
        out['cam_inv_trans_residual'] = g[l['cam_inv_trans_res']:l['cam_inv_trans_res'] + 3 * T].view(T, 3)[:Ts]
    for pi, idx in enumerate(packed.person_ids[si]):
        pd = data['person_data'][idx]
        n = int(pd['fr_end']) - int(pd['fr_start'])
        pp = g[l['person0'] + pi * l['person_stride']:l['person0'] + (pi + 1) * l['person_stride']]
        names = {'local_xy': pp[l['local_xy']:l['local_xy'] + 2], 'local_heading': pp[l['local_heading']:l['local_heading'] + 1],
                 'local_dxy': pp[l['local_dxy']:l['local_dxy'] + 2 * T].view(T, 2)[1:n], 'local_dheading': pp[l['local_dheading']:l['local_dheading'] + T][1:n],
                 'local_z': pp[l['local_z']:l['local_z'] + T][:n], 'local_rot': pp[l['local_rot']:l['local_rot'] + 6 * T].view(T, 6)[:n]}
        for key in opt_variables:
            if 'local' in key:
                out['p%d_traj_%s' % (idx, key)] = names[key]
        if 'world_dheading' in opt_variables:
            out['p%d_world_dheading' % idx] = pp[l['world_dheading']:l['world_dheading'] + T][:Ts].unsqueeze(-1)
    return out


def check_case(runner, asset_root, golden, cfg_id, T, P, K):
    run, dev = runner
    g = golden('grecon_%s_T%d_P%d' % (cfg_id, T, P))
    cfg = get_config(cfg_id)
    specs = cfg['grecon_model_specs']
    md = synth.make_smpl_model()
    in_dict = synth.make_in_dict(seed=3, num_frames=T, num_persons=P, smpl_model=md)
    ora = build.load_optimizer(asset_root, cfg)
    data = ora.init_data(in_dict, latents=mg.latents_for(in_dict, 3))
    jl = j_local_from_oracle(ora.smpl, data)
    first = True
    has_wd = False
    for stage, spec in cfg['opt_stage_specs'].items():
        packed = packing.PackedScenes([data], [jl], dev)
        if first:
            # forward-only evaluation reproduces forward(data, [], {'stage': 'init'}) (global_recon_model.py:246)
            sd0 = packing.stage_desc(spec, specs, has_wd, niters=0)
            sd0.var_mask = 0
            sd0.flags &= ~packing.FLAG_CAM_FROM_PERSON      # stage 'init' keeps the initial camera (:473)
            run(packed, sd0, False)
            for pi in range(P):
                ref = g['init_p%d_kp_2d_pred' % pi]
                err = kp_err(packed.t['kp_2d_pred'][pi, :T].cpu().numpy(), ref)
                assert err < 2e-2, 'initial projection %g px' % err
            # first-iteration losses and gradients vs autograd of the reference
            packed1 = packing.PackedScenes([data], [jl], dev)
            sd1 = packing.stage_desc(spec, specs, has_wd, niters=1)
            grads = run(packed1, sd1, True)
            named = grads_by_name(packed1, grads, data, specs, spec['opt_variables'])
            gmax = max([float(np.abs(g['%s_grad_%s' % (stage, n)]).max()) for n in named if g['%s_grad_%s' % (stage, n)].size] + [0.0])
            for name, val in named.items():
                ref = g['%s_grad_%s' % (stage, name)]
                assert tuple(val.shape) == ref.shape, name
                if ref.size == 0:
                    continue
                # cancellation noise scales with the largest gradient of the stage (gauge directions such as the 3dpw person
                # heading have true gradient 0 and a reference value that is itself rounding noise)
                tol = 3e-4 * max(1.0, float(np.abs(ref).max())) + 2e-5 * gmax
                err = np.abs(val.numpy() - ref).max()
                assert err < tol, 'gradient %s: abs err %.2e > %.2e' % (name, err, tol)
            for lname, lid in packing.LOSS_IDS.items():
                key = '%s_loss_%s' % (stage, lname)
                if key in g:
                    ref = float(g[key])
                    got = float(packed1.t['losses'][0, lid].cpu())
                    assert abs(got - ref) <= 2e-4 * max(1.0, abs(ref)), 'loss %s: %g vs %g' % (lname, got, ref)
            first = False
        sd = packing.stage_desc(spec, specs, has_wd, niters=min(K, spec['opt_niters']))
        run(packed, sd, False)
        packed.unpack_into([data], spec, specs)
        has_wd = has_wd or 'world_dheading' in spec['opt_variables']
    # free-running state after K steps per stage.  With the Adam update torch's to the bit (tests/test_adam_exact.py) the trajectories
    # stay together; what is left are the sign flips of rounding-noise gradients along directions the loss does not see (a structurally
    # zero gradient is +-1e-7 in both implementations and Adam turns its SIGN into a full +-lr step).  Bounds per case = 3 x what the
    # kernel achieves, measured on the MI355X and the CPU runtime.
    tol_kp, tol_tr, tol_rot = KSTEP_TOL[(cfg_id, T, P)]
    worst = [0.0, 0.0, 0.0]
    for pi in range(P):
        pd = data['person_data'][pi]
        vis = g['init_p%d_vis_frames' % pi] & g['init_p0_vis_frames']
        worst[0] = max(worst[0], kp_err(pd['kp_2d_pred'].numpy(), g['opt_p%d_kp_2d_pred' % pi], vis))
        if cfg_id != 'glamr_3dpw':
            worst[1] = max(worst[1], float(np.abs(pd['root_trans_world'].numpy() - g['opt_p%d_root_trans_world' % pi]).max()))
            worst[2] = max(worst[2], _rot_err(pd['smpl_orient_world'].numpy(), g['opt_p%d_smpl_orient_world' % pi]))
    print('K-step state %s T=%d P=%d K=%d: kp %.4f px, root_trans_world %.2e, smpl_orient_world %.2e' % (cfg_id, T, P, K, *worst))
    assert worst[0] < tol_kp, 'kp_2d_pred after optimisation: %g px' % worst[0]
    assert worst[1] < tol_tr, 'root_trans_world: %g' % worst[1]
    assert worst[2] < tol_rot, 'smpl_orient_world (as rotation): %g' % worst[2]


# (projected keypoints px, root_trans_world m, smpl_orient_world as rotation) after K free-running Adam steps per stage
KSTEP_TOL = {   # achieved (MI355X / CPU runtime):           kp px            root m            orientation
    ('glamr_dynamic', 120, 1): (0.15, 1.5e-5, 3.5e-3),        # 0.009 / 0.046    3.9e-6 / 2.9e-6   1.0e-3 / 7.4e-4
    ('glamr_static', 90, 1): (0.2, 3e-5, 6e-3),               # 0.064 / 0.046    9.7e-6 / 6.6e-7   2.0e-3 / 8.6e-4
    ('glamr_static_multi', 120, 2): (0.01, 2e-5, 6e-4),       # 0.001 / 0.001    5.7e-6 / 2.2e-6   1.9e-4 / 1.8e-4
    ('glamr_dynamic_multi', 100, 2): (0.15, 6e-4, 5e-3),      # 0.043 / 0.046    1.6e-4 / 1.8e-4   1.6e-3 / 1.6e-3
    # glamr_3dpw: the camera rides on the person, so the world trajectory is a GAUGE: the fixture's own gradient of `traj_local_xy` is
    # (3.0e-4, 5.9e-5) beside terms of 1.6e3 -- below fp32 rounding of what it is summed from -- and Adam's first steps are lr x its SIGN.
    # With the suffix sum of the planar-position gradient in the Hillis-Steele order (ds_bpermute shuffles, rounds 1-3) or sequential (CPU
    # runtime) that sign comes out as the reference's: 0.002 px; in the DPP row-shift order (round 4: another order of the same
    # exact-on-integers sums, tests/test_scan_gpu.py) it did not: 0.20 px, and round 4 widened this bound to 0.6.  Round 5: the instances that
    # derive the camera from the person keep the Hillis-Steele order for ALL FOUR scans of the iteration (grecon_algo.hpp `scan_shuffle =
    # cam_from_person`; with that order on the planar-gradient scan alone it was still 0.20 px -- the forward sums feed the same sign); the
    # other instances stay on DPP, and the bound is back where it was.
    ('glamr_3dpw', 120, 1): (0.01, 1e-6, 1e-6),               # 0.002 / 0.002    (the person's world pose is not compared)
    ('glamr_h36m', 100, 2): (0.55, 2e-5, 9e-4),               # 0.074 / 0.177    2.4e-6 / 5.0e-6   2.8e-4 / 2.6e-4
    ('glamr_static_multi', 300, 4): (0.05, 6e-5, 1.1e-3),     # 0.012 / 0.015    1.2e-5 / 1.9e-5   3.5e-4 / 3.3e-4
    # more than 8 persons (csrc/grecon_wide.hip; MI355X only).  Through optimize(): 0.009 px / 1.5e-5 / 1.9e-6 and 0.003 px / 5.5e-6 / 2.6e-6
    ('glamr_static_multi', 60, 10): (0.03, 6e-5, 1e-4),
    ('glamr_dynamic_multi', 48, 9): (0.03, 6e-5, 1e-4),
}


def full_schedule_errors(out_persons, cam_pose, g, P, prefix=''):
    """Differences between a finished optimisation and a full-schedule reference fixture (oracle/make_golden.py gen_full_cfg), over the
    frames each person is seen in: projected keypoints px, root-in-camera m, world root m, world orientation as a rotation, and the camera --
    `cam_rot` / `cam_trans` value by value (world -> camera matrices of the frames the first person is seen in) and, free of the world frame's
    gauge, `orient_cam`: the person's orientation SEEN FROM the camera, R_cam R(smpl_orient_world), as a rotation."""
    from oracle.port import transforms as tf
    def seen_from_cam(cam, aa):
        R = tf.aa_to_rotmat(torch.as_tensor(np.asarray(aa), dtype=torch.float32)).numpy()
        return np.einsum('tij,tjk->tik', np.asarray(cam)[:, :3, :3], R)
    def root_cam(cam, trans):
        return np.einsum('tij,tj->ti', cam[:, :3, :3], trans) + cam[:, :3, 3]
    cam_ref = g[prefix + 'cam_pose']
    worst = dict(kp=0.0, root_cam=0.0, root_world=0.0, orient=0.0, frames_over_1px=0, orient_cam=0.0, cam_rot=0.0, cam_trans=0.0)
    seen0 = g['%sp0_vis_frames' % prefix]
    worst['cam_rot'] = float(np.abs(np.asarray(cam_pose)[seen0][:, :3, :3] - cam_ref[seen0][:, :3, :3]).max())
    worst['cam_trans'] = float(np.abs(np.asarray(cam_pose)[seen0][:, :3, 3] - cam_ref[seen0][:, :3, 3]).max())
    for pi in range(P):
        pd = out_persons[pi]
        vis = g['%sp%d_vis_frames' % (prefix, pi)]
        kp, ref = np.asarray(pd['kp_2d_pred'], np.float64)[vis], np.asarray(g['%sp%d_kp_2d_pred' % (prefix, pi)], np.float64)[vis]
        ok = np.abs(ref).max(axis=-1) < 2.5e3                      # see kp_err: points near the camera plane say nothing
        d = np.where(ok, np.abs(kp - ref).max(axis=-1), 0.0).max(axis=-1)      # per visible frame
        worst['kp'] = max(worst['kp'], float(d.max()))
        worst['frames_over_1px'] = max(worst['frames_over_1px'], int((d > 1).sum()))
        tr, tr_ref = np.asarray(pd['root_trans_world']), g['%sp%d_root_trans_world' % (prefix, pi)]
        worst['root_world'] = max(worst['root_world'], float(np.abs(tr - tr_ref)[vis].max()))
        worst['root_cam'] = max(worst['root_cam'], float(np.abs(root_cam(np.asarray(cam_pose), tr) - root_cam(cam_ref, tr_ref))[vis].max()))
        worst['orient'] = max(worst['orient'], _rot_err(np.asarray(pd['smpl_orient_world'])[vis], g['%sp%d_smpl_orient_world' % (prefix, pi)][vis]))
        both = vis & seen0 if cam_ref.shape[0] == vis.shape[0] else vis
        oc = seen_from_cam(cam_pose, pd['smpl_orient_world']) - seen_from_cam(cam_ref, g['%sp%d_smpl_orient_world' % (prefix, pi)])
        worst['orient_cam'] = max(worst['orient_cam'], float(np.abs(oc[both]).max()))
    return worst


# Full schedules vs the unmodified reference (tests/golden/full_<cfg>_T300_P<P>[_nogap].npz), starting from the reference's own initial
# state (the oracle's / the numpy init_data): bounds = (keypoints px, root in camera m) ~3 x achieved, MI355X / CPU runtime in the comment
# CPU runtime (tests/test_grecon_hostsim.py), worst stage:        kp px   root m       achieved
FULL_TOL_CPU = {('glamr_3dpw', True): (0.1, 1e-3),                            # 0.026   3.3e-4
                ('glamr_dynamic_multi', False): (0.01, 5e-5)}                 # 0.0012  8.6e-6


def check_full_schedule(runner, asset_root, golden, cfg_id, T, P, gap):
    """All stages to their last iteration on the runner (CPU runtime or device kernel) from the oracle's init_data state."""
    run, dev = runner
    g = golden(mg.full_name(cfg_id, T, P, gap))
    cfg = get_config(cfg_id)
    specs = cfg['grecon_model_specs']
    seed = mg.FULL_SEED[(cfg_id, T, P)]
    assert int(g['seed']) == seed
    in_dict = synth.make_in_dict(seed=seed, num_frames=T, num_persons=P, smpl_model=synth.make_smpl_model(), gap=None if gap else (0, 0))
    ora = build.load_optimizer(asset_root, cfg)
    data = ora.init_data(in_dict, latents=mg.latents_for(in_dict, seed))
    jl = j_local_from_oracle(ora.smpl, data)
    has_wd = False
    report = []
    for si, (stage, spec) in enumerate(cfg['opt_stage_specs'].items()):
        packed = packing.PackedScenes([data], [jl], dev)
        run(packed, packing.stage_desc(spec, specs, has_wd), False)
        packed.unpack_into([data], spec, specs)
        has_wd = has_wd or 'world_dheading' in spec['opt_variables']
        last = si == len(cfg['opt_stage_specs']) - 1
        if last or 's1_cam_pose' in g:
            persons = {pi: {k: v.numpy() if hasattr(v, 'numpy') else v for k, v in data['person_data'][pi].items() if k in ('kp_2d_pred', 'root_trans_world', 'smpl_orient_world')}
                       for pi in range(P)}
            cam = data['cam_pose'].numpy() if hasattr(data['cam_pose'], 'numpy') else data['cam_pose']
            report.append((stage, full_schedule_errors(persons, cam, g, P, '' if last else 's1_')))
    return report


def check_poses_only(runner, asset_root):
    """GLAMR_FLAG_POSES_ONLY against the full forward-only launch: same orient_world / trans_world / cam_pose to the bit, projections untouched."""
    from oracle.port import build
    from oracle import make_golden as mg
    from glamr_amd.global_recon.configs import get_config
    from glamr_amd.utils import synth
    run, dev = runner
    for cfg_id, T, P in (('glamr_dynamic', 300, 1), ('glamr_dynamic_multi', 90, 2)):
        cfg = get_config(cfg_id)
        in_dict = synth.make_in_dict(seed=3, num_frames=T, num_persons=P, smpl_model=synth.make_smpl_model())
        ora = build.load_optimizer(asset_root, cfg)
        data = ora.init_data(in_dict, latents=mg.latents_for(in_dict, 3))
        jl = j_local_from_oracle(ora.smpl, data)
        spec = next(iter(cfg['opt_stage_specs'].values()))
        out = []
        for poses_only in (False, True):
            packed = packing.PackedScenes([data], [jl], dev)
            packed.t['kp_2d_pred'].fill_(-7.0)
            sd = packing.stage_desc(spec, cfg['grecon_model_specs'], False, niters=0)
            sd.var_mask = 0
            sd.flags &= ~packing.FLAG_CAM_FROM_PERSON
            if poses_only:
                sd.flags |= packing.FLAG_POSES_ONLY
            run(packed, sd, False)
            out.append({k: packed.t[k].cpu().numpy().copy() for k in ('orient_world', 'trans_world', 'cam_pose', 'kp_2d_pred')})
        for k in ('orient_world', 'trans_world', 'cam_pose'):
            assert np.array_equal(out[0][k], out[1][k]), (cfg_id, k)
        assert np.isfinite(out[0]['orient_world']).all() and np.abs(out[0]['trans_world']).max() > 0
        assert (out[1]['kp_2d_pred'] == -7.0).all() and not (out[0]['kp_2d_pred'] == -7.0).all()
