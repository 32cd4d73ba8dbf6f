"""End-to-end MI355X tests of the drop-in entry point GlobalReconOptimizer.optimize (host preprocessing + prior kernels + SMPL
kernel + fused optimiser) against fixtures produced by the UNMODIFIED reference on the same synthetic inputs and latents."""
import os
import numpy as np
import pytest
import torch

from oracle import make_golden as mg
from glamr_amd.utils import synth
from tests.grecon_common import kp_err, KSTEP_TOL

pytestmark = pytest.mark.gpu


def _rot_err(aa_a, aa_b):
    """Axis-angle vectors are compared as ROTATIONS: near an angle of pi the same rotation has two far-apart axis-angle forms."""
    from oracle.port import transforms as tf
    Ra = tf.aa_to_rotmat(torch.as_tensor(np.asarray(aa_a), dtype=torch.float32))
    Rb = tf.aa_to_rotmat(torch.as_tensor(np.asarray(aa_b), dtype=torch.float32))
    return float((Ra - Rb).abs().max())
INDEX_KEYS = ('visible', 'visible_orig', 'exist_frames', 'vis_frames', 'invis_frames', 'kp_2d_score')


@pytest.fixture(scope='module')
def make_model(asset_root):
    from glamr_amd.global_recon.models import model_dict
    from glamr_amd.global_recon.configs import get_config
    from glamr_amd.lib.models.smpl import SMPL
    from glamr_amd.models.prior_models import MotionTrajJointModel
    dev = torch.device('cuda:0')
    smpl = SMPL(os.path.join(asset_root, 'data', 'body_models', 'smpl'), pose_type='body26fk',
                extra_regressor_path=os.path.join(asset_root, 'data', 'J_regressor_extra.npy')).to(dev)
    mt = MotionTrajJointModel(None, dev, None, smpl=smpl, results_root=os.path.join(asset_root, 'results'))

    def make(cfg_id):
        return model_dict['global_recon_model'](get_config(cfg_id), dev, None, smpl=smpl, mt_model=mt)
    return make


@pytest.mark.parametrize('cfg_id,T,P,K', mg.GRECON_CASES + mg.GRECON_CASES_WIDE)
def test_optimize_matches_reference_fixture(make_model, golden, cfg_id, T, P, K):
    g = golden('grecon_%s_T%d_P%d' % (cfg_id, T, P))
    md = synth.make_smpl_model()
    in_dict = synth.make_in_dict(seed=3, num_frames=T, num_persons=P, smpl_model=md)
    model = make_model(cfg_id)
    # state right after init_data: indices bit-exact, continuous quantities to fp32 round-off
    data = model.init_data(in_dict, latents=mg.latents_for(in_dict, 3))
    for pi in range(P):
        pd = data['person_data'][pi]
        for key in INDEX_KEYS:
            assert np.array_equal(np.asarray(pd[key]), g['init_p%d_%s' % (pi, key)]), 'frame/visibility indexing must be bit-exact: ' + key
        assert int(pd['fr_start']) == int(g['init_p%d_fr_start' % pi]) and int(pd['fr_end']) == int(g['init_p%d_fr_end' % pi])
        for key, tol in (('smpl_pose', 1e-4), ('traj_local_pred', 1e-4), ('root_trans_world', 2e-4), ('kp_2d_pred', 5e-2)):
            if key == 'kp_2d_pred':
                err = kp_err(pd[key], g['init_p%d_%s' % (pi, key)])
            else:
                err = np.abs(np.asarray(pd[key], dtype=np.float64) - g['init_p%d_%s' % (pi, key)]).max()
            assert err < tol, 'init %s: %g' % (key, err)
        assert _rot_err(pd['smpl_orient_world'], g['init_p%d_smpl_orient_world' % pi]) < 2e-4
    seen = g['init_p0_vis_frames']
    assert np.abs(np.asarray(data['cam_pose'])[seen] - g['init_cam_pose'][seen]).max() < 2e-4
    # K iterations per stage
    out = model.optimize(in_dict, latents=mg.latents_for(in_dict, 3), max_iters=K)
    worst = [0.0, 0.0, 0.0]
    for pi in range(P):
        pd = out['person_data'][pi]
        vis = g['init_p%d_vis_frames' % pi] & g['init_p0_vis_frames']
        # through the whole entry point the start is the DEVICE init_data (1e-4 from the reference's priors): 3 x the kernel-only bounds
        tol_kp, tol_tr, tol_rot = KSTEP_TOL[(cfg_id, T, P)]
        err = kp_err(pd['kp_2d_pred'], g['opt_p%d_kp_2d_pred' % pi], vis)
        worst[0] = max(worst[0], err)
        assert err < (tol_kp if cfg_id == 'glamr_3dpw' else max(3 * tol_kp, 0.1)), 'kp_2d_pred after optimisation: %g px' % err      # (3dpw: a gauge bound already, grecon_common.KSTEP_TOL)
        if cfg_id != 'glamr_3dpw':
            err = np.abs(pd['root_trans_world'] - g['opt_p%d_root_trans_world' % pi]).max()
            worst[1] = max(worst[1], err)
            assert err < max(3 * tol_tr, 5e-4), 'root_trans_world: %g' % err
            err = _rot_err(pd['smpl_orient_world'], g['opt_p%d_smpl_orient_world' % pi])
            worst[2] = max(worst[2], err)
            assert err < max(3 * tol_rot, 5e-3), 'smpl_orient_world (as rotation): %g' % err
    print('optimize() %s T=%d P=%d K=%d: kp %.4f px, root_trans_world %.2e, smpl_orient_world %.2e' % (cfg_id, T, P, K, *worst))
    assert out['cam_pose'].shape == (T, 4, 4) and out['seq_len'] == T


def test_flag_opt_vis_local_rot_matches_the_reference(make_model, golden, asset_root):
    """A model flag no shipped config sets (global_recon_model.py:45,416-419): the rotation residual `traj_local_rot` is applied at the frames a
    person is SEEN in only.  Fixture: the unmodified reference with the flag on, 120 frames with detections missing in [40, 70), 12 iterations per
    stage (oracle/make_golden.py gen_grecon_flags).  Here the schedule then runs launch by launch with the residual's gradient masked."""
    from glamr_amd.global_recon.models import model_dict
    from glamr_amd.global_recon.configs import get_config
    tag, cfg_id, T, P, K, flags, gap, _ = mg.FLAG_CASES[0]
    g = golden('grecon_%s_T%d_P%d_%s' % (cfg_id, T, P, tag))
    base = make_model(cfg_id)
    cfg = get_config(cfg_id)
    cfg['grecon_model_specs'].update(flags)
    model = model_dict['global_recon_model'](cfg, base.device, None, smpl=base.smpl, mt_model=base.mt_model)
    in_dict = synth.make_in_dict(seed=3, num_frames=T, num_persons=P, smpl_model=synth.make_smpl_model(), gap=gap)
    lat = mg.latents_for(in_dict, 3)
    out = model.optimize(in_dict, latents=lat, max_iters=K)
    pd = out['person_data'][0]
    vis = g['init_p0_vis_frames']
    assert (~vis).sum() == gap[1] - gap[0]
    rot = np.asarray(pd['traj_local_rot'])
    assert np.abs(rot[~vis]).max() == 0.0 and np.abs(g['opt_p0_traj_local_rot'][~vis]).max() == 0.0      # never moved, here and in the reference
    e_rot = np.abs(rot - g['opt_p0_traj_local_rot']).max()
    e_kp = kp_err(pd['kp_2d_pred'], g['opt_p0_kp_2d_pred'], vis)
    e_tr = np.abs(pd['root_trans_world'] - g['opt_p0_root_trans_world']).max()
    e_or = _rot_err(pd['smpl_orient_world'], g['opt_p0_smpl_orient_world'])
    print('flag_opt_vis_local_rot, %d iterations per stage: kp %.4f px, root_trans_world %.2e m, smpl_orient_world %.2e, traj_local_rot %.2e (largest %.3g)'
          % (K, e_kp, e_tr, e_or, e_rot, np.abs(g['opt_p0_traj_local_rot']).max()))
    assert e_kp < 0.01 and e_tr < 1e-5 and e_or < 1.2e-2 and e_rot < 3e-5       # achieved 0.0023 px / 1.3e-6 m / 4.0e-3 / 6.6e-6
    # and the flag matters on this input: without it the residual of the unseen frames moves
    plain = base.optimize(in_dict, latents=lat, max_iters=K)
    assert np.abs(np.asarray(plain['person_data'][0]['traj_local_rot'])[~vis]).max() > 1e-4      # (7e-4 after 12 iterations per stage)


def test_flag_traj_from_cam_matches_the_reference(make_model, golden):
    """flag_traj_from_cam (global_recon_model.py:55,237,325-351; no shipped config): the world trajectory is first read off the initial camera.  With
    a trajectory predictor every existing frame is overwritten right after (:283-289), so the flag shows in the frames OUTSIDE a person's existence
    range: person 1 exists in [17, 83) of 100 frames.  Fixture: the unmodified reference with the flag on (oracle/make_golden.py gen_grecon_flags)."""
    from glamr_amd.global_recon.models import model_dict
    from glamr_amd.global_recon.configs import get_config
    tag, cfg_id, T, P, K, flags, gap, trim = mg.FLAG_CASES[1]
    g = golden('grecon_%s_T%d_P%d_%s' % (cfg_id, T, P, tag))
    base = make_model(cfg_id)
    cfg = get_config(cfg_id)
    cfg['grecon_model_specs'].update(flags)
    model = model_dict['global_recon_model'](cfg, base.device, None, smpl=base.smpl, mt_model=base.mt_model)
    seed = mg.FLAG_SEED[tag]
    in_dict = synth.trim_person(synth.make_in_dict(seed=seed, num_frames=T, num_persons=P, smpl_model=synth.make_smpl_model()), *trim)
    lat = mg.latents_for(in_dict, seed)
    data = model.init_data(in_dict, latents=lat)                      # round 5: init_data on the DEVICE (glamr_init_scenes_ex)
    host = model_dict['global_recon_model'](cfg, base.device, None, smpl=base.smpl, mt_model=base.mt_model)
    host.init_data_batch = host.init_data_batch_host                  # ... and its numpy twin
    data_host = host.init_data(in_dict, latents=lat)
    plain = base.init_data(in_dict, latents=lat)
    outside = ~g['init_p1_exist_frames']
    assert outside.sum() > 20
    for d in (data, data_host):
        for pi in range(P):
            pd = d['person_data'][pi]
            assert _rot_err(pd['smpl_orient_world'], g['init_p%d_smpl_orient_world' % pi]) < 3e-4
            assert np.abs(np.asarray(pd['root_trans_world'], np.float64) - g['init_p%d_root_trans_world' % pi]).max() < 3e-4
    # the flag matters on this input: the default initialisation leaves other orientations outside the existence range
    assert _rot_err(np.asarray(plain['person_data'][1]['smpl_orient_world'])[outside], g['init_p1_smpl_orient_world'][outside]) > 1e-2
    out = model.optimize(in_dict, latents=lat, max_iters=K)
    worst = [0.0, 0.0, 0.0]
    for pi in range(P):
        pd = out['person_data'][pi]
        vis = g['init_p%d_vis_frames' % pi] & g['init_p0_vis_frames']
        worst[0] = max(worst[0], kp_err(pd['kp_2d_pred'], g['opt_p%d_kp_2d_pred' % pi], vis))
        worst[1] = max(worst[1], float(np.abs(pd['root_trans_world'] - g['opt_p%d_root_trans_world' % pi]).max()))
        worst[2] = max(worst[2], _rot_err(pd['smpl_orient_world'], g['opt_p%d_smpl_orient_world' % pi]))
    print('flag_traj_from_cam, %d iterations per stage: kp %.4f px, root_trans_world %.2e m, smpl_orient_world %.2e (all frames, those outside the existence range included)' % (K, *worst))
    assert worst[0] < 0.05 and worst[1] < 2e-4 and worst[2] < 1.5e-3          # achieved 0.013 px / 5.0e-5 m / 3.8e-4
    # ... and HBM in -> HBM out (optimize_resident raised NotImplementedError for this flag until round 5)
    datas, packed = model.optimize_resident(model.stage_inputs([in_dict], [lat]), max_iters=K)
    res = model.collect(datas, packed)[0]
    for pi in range(P):
        assert float(np.abs(np.asarray(res['person_data'][pi]['kp_2d_pred']) - np.asarray(out['person_data'][pi]['kp_2d_pred'])).max()) == 0.0


@pytest.mark.parametrize('case', [c for c in mg.FLAG_CASES if c[0] == 'absolute_heading'], ids=lambda c: '%s-%d-%d' % (c[1], c[2], c[3]))
def test_absolute_heading_matches_the_reference(make_model, golden, case):
    """absolute_heading (global_recon_model.py:59,283,421; no shipped config): the heading entries of the local trajectory are absolute angles, so
    traj_local2global_heading sums nothing up.  Fixtures: the unmodified reference with the flag on (the shipped trajectory predictor emits
    INCREMENTS, so this is another motion than the default's -- the same code path).  Every launch of the stage kernel, the 'init' forward passes
    included, then runs on the instances of csrc/grecon_wide.hip (GLAMR_FLAG_ABSOLUTE_HEADING)."""
    from glamr_amd.global_recon.models import model_dict
    from glamr_amd.global_recon.configs import get_config
    tag, cfg_id, T, P, K, flags, gap, _ = case
    g = golden('grecon_%s_T%d_P%d_%s' % (cfg_id, T, P, tag))
    base = make_model(cfg_id)
    cfg = get_config(cfg_id)
    cfg['grecon_model_specs'].update(flags)
    model = model_dict['global_recon_model'](cfg, base.device, None, smpl=base.smpl, mt_model=base.mt_model)
    seed = mg.FLAG_SEED[tag]
    in_dict = synth.make_in_dict(seed=seed, num_frames=T, num_persons=P, smpl_model=synth.make_smpl_model(), gap=gap)
    lat = mg.latents_for(in_dict, seed)
    data = model.init_data(in_dict, latents=lat)
    plain = base.init_data(in_dict, latents=lat)
    for pi in range(P):
        pd = data['person_data'][pi]
        assert _rot_err(pd['smpl_orient_world'], g['init_p%d_smpl_orient_world' % pi]) < 5e-4
        assert np.abs(np.asarray(pd['root_trans_world'], np.float64) - g['init_p%d_root_trans_world' % pi]).max() < 5e-4
    # the flag matters: the default reading of the same predictor output is another trajectory
    assert np.abs(np.asarray(plain['person_data'][0]['root_trans_world'], np.float64) - g['init_p0_root_trans_world']).max() > 1e-2
    out = model.optimize(in_dict, latents=lat, max_iters=K)
    worst = [0.0, 0.0, 0.0]
    for pi in range(P):
        pd = out['person_data'][pi]
        vis = g['init_p%d_vis_frames' % pi] & g['init_p0_vis_frames']
        worst[0] = max(worst[0], kp_err(pd['kp_2d_pred'], g['opt_p%d_kp_2d_pred' % pi], vis))
        worst[1] = max(worst[1], float(np.abs(pd['root_trans_world'] - g['opt_p%d_root_trans_world' % pi]).max()))
        worst[2] = max(worst[2], _rot_err(pd['smpl_orient_world'], g['opt_p%d_smpl_orient_world' % pi]))
    print('absolute_heading %s T=%d P=%d, %d iterations per stage: kp %.4f px, root_trans_world %.2e m, smpl_orient_world %.2e' % (cfg_id, T, P, K, *worst))
    # achieved 0.014 / 0.0004 px, 5e-7 / 7e-7 m, 2.2e-3 / 2.9e-4 up to round 5; 0.054 px for the one-person case since round 6 built the library
    # without packed-fp32 instructions (glamr_amd/build.py: the wide instance's products fuse in another operand order, and ten sign-driven Adam
    # steps per stage amplify a last-bit difference)
    assert worst[0] < 0.08 and worst[1] < 1e-5 and worst[2] < 7e-3


def _full_schedule(make_model, golden, tag, gap, host_init=False):
    g = golden('full_glamr_dynamic_T300' + tag)
    md = synth.make_smpl_model()
    in_dict = synth.make_in_dict(seed=0, num_frames=300, num_persons=1, smpl_model=md, gap=gap)
    model = make_model('glamr_dynamic')
    if host_init:
        model.init_data_batch = model.init_data_batch_host
    out = model.optimize(in_dict, latents=mg.latents_for(in_dict, 0))
    pd = out['person_data'][0]
    vis = g['p0_vis_frames']
    d_kp = np.abs(pd['kp_2d_pred'] - g['p0_kp_2d_pred'])[vis].max(axis=(1, 2))           # per visible frame
    # joints in the camera frame are what the loss sees: compare the root in camera coordinates
    def root_cam(cam, trans):
        return np.einsum('tij,tj->ti', cam[:, :3, :3], trans) + cam[:, :3, 3]
    e_root = np.abs(root_cam(out['cam_pose'], pd['root_trans_world']) - root_cam(g['cam_pose'], g['p0_root_trans_world']))[vis].max()
    obs = in_dict['est'][0]['kp_2d'][:, :24, :2]
    def reproj(kp):
        return float(np.linalg.norm(kp[vis][:, :24] - obs, axis=-1).mean())
    from tests.grecon_common import full_schedule_errors
    w = full_schedule_errors(out['person_data'], out['cam_pose'], g, 1)
    print('full schedule%s: kp max %.3f px (median over frames %.3f), root-in-camera %.2e m, world root %.2e m, orientation %.2e, seen from the camera %.2e, '
          'cam_pose rotation %.2e translation %.2e m, reprojection %.3f vs reference %.3f px'
          % (tag, d_kp.max(), np.median(d_kp), e_root, w['root_world'], w['orient'], w['orient_cam'], w['cam_rot'], w['cam_trans'], reproj(pd['kp_2d_pred']), reproj(g['p0_kp_2d_pred'])))
    _full_schedule.last = w
    return d_kp, e_root, reproj(pd['kp_2d_pred']), reproj(g['p0_kp_2d_pred'])


def test_full_schedule_300_frames_all_detected(make_model, golden):
    """BASELINE.json configs[1] (300 frames, 1 person, dynamic camera, the full 500-iteration schedule) with the person detected in
    every frame: the problem is well conditioned and the result is compared with the reference VALUE BY VALUE."""
    d_kp, e_root, _, _ = _full_schedule(make_model, golden, '_nogap', (0, 0))
    assert d_kp.max() < 0.1 and e_root < 2e-3          # achieved on the MI355X: 0.015 px, 2.3e-4 m
    w = _full_schedule.last                            # smpl_orient_world, root_trans_world, cam_pose (<= 1e-3 after the full schedule)
    assert w['root_world'] < 1e-3 and w['orient'] < 1e-3 and w['orient_cam'] < 1e-3 and w['cam_rot'] < 1e-3 and w['cam_trans'] < 1e-3


def _family_envelope(golden):
    """Spread of the UNMODIFIED reference on this input (oracle/make_golden.py gen_full_family): re-runs with another thread count and
    with its initial cam_pose perturbed by 1e-7 stay within 0.05 px of the committed golden; 1e-6 perturbations push it into a
    neighbouring solution (up to 9.4 px in 18 of 240 frames)."""
    g = golden('full_glamr_dynamic_T300')
    fam = golden('full_glamr_dynamic_T300_family')
    vis = g['p0_vis_frames']
    stats = {}
    for key in fam:
        if key.endswith('_kp_2d_pred'):
            d = np.abs(fam[key] - g['p0_kp_2d_pred'])[vis].max(axis=(1, 2))
            stats[key[:-len('_kp_2d_pred')]] = (float(d.max()), int((d > 1).sum()))
    return stats


def test_full_schedule_300_frames_detection_gap_host_init(make_model, golden):
    """BASELINE.json configs[1] with person 0 undetected in frames [100,160), starting from the numpy variant of init_data_batch (rounds
    like the reference's CPU operators: the reference's own initial state).  The reference holds 0.03-0.04 px under 1e-7 perturbations and
    thread-count changes, so a start that is equal to the last bit must stay in its solution: VALUE BY VALUE, 0.1 px."""
    d_kp, e_root, ours, ref = _full_schedule(make_model, golden, '', None, host_init=True)
    assert d_kp.max() < 0.1 and e_root < 5e-3
    w = _full_schedule.last
    assert w['root_world'] < 5e-3 and w['orient'] < 5e-3 and w['orient_cam'] < 5e-3 and w['cam_rot'] < 5e-3 and w['cam_trans'] < 5e-3


def test_full_schedule_300_frames_detection_gap(make_model, golden):
    """Same input through the default path (init_data on the device).  Its initial camera poses differ from the reference's by a few
    1e-6 (library sine / cosine, the priors' 1e-4), and at that size of perturbation the UNMODIFIED reference itself ends in one of two
    neighbouring solutions (2 of 3 seeds at 1e-6 move by 9.4 px in 18 frames, the family fixture).  The device path must end in one of
    them: inside the envelope of the reference's own family, and within 0.25 px of ONE member, value by value."""
    g = golden('full_glamr_dynamic_T300')
    fam = golden('full_glamr_dynamic_T300_family')
    stats = _family_envelope(golden)
    loose_px = max(v[0] for v in stats.values())
    loose_n = max(v[1] for v in stats.values())
    d_kp, e_root, ours, ref = _full_schedule(make_model, golden, '', None)
    print('reference family: %s' % {k: ('%.3f px' % v[0], v[1]) for k, v in stats.items()})
    assert d_kp.max() <= loose_px + 0.1 and int((d_kp > 1).sum()) <= loose_n          # the reference's own envelope
    # ... and not merely inside it: on top of one of the reference's solutions
    model = make_model('glamr_dynamic')
    in_dict = synth.make_in_dict(seed=0, num_frames=300, num_persons=1, smpl_model=synth.make_smpl_model())
    kp = model.optimize(in_dict, latents=mg.latents_for(in_dict, 0))['person_data'][0]['kp_2d_pred']
    vis = g['p0_vis_frames']
    members = {'golden': g['p0_kp_2d_pred']}
    members.update({k[:-len('_kp_2d_pred')]: v for k, v in fam.items() if k.endswith('_kp_2d_pred')})
    best = min((float(np.abs(kp - v)[vis].max()), k) for k, v in members.items())
    print('closest member of the reference family: %s at %.3f px' % (best[1], best[0]))
    assert best[0] < 0.25


MULTI_SEEDS = (0,) + tuple(mg.FULL_SEEDS)
_MULTI = {}


def _seed_fixture(golden, seed):
    """(reference result, {member: projections}) of BASELINE configs[1] WITH the detection gap for one seed: seed 0 is the round-1 fixture and
    its family file, seeds 1.. come from oracle/make_golden.py gen_full_seeds (the 1e-6 family inside the same file for FULL_SEEDS_FAMILY)."""
    if seed == 0:
        g, fam = golden('full_glamr_dynamic_T300'), golden('full_glamr_dynamic_T300_family')
        members = {k[:-len('_kp_2d_pred')]: v for k, v in fam.items() if k.endswith('_kp_2d_pred')}
    else:
        g = golden(mg.seed_name(seed))
        members = {k[4:-len('_kp_2d_pred')]: v for k, v in g.items() if k.startswith('fam_') and k.endswith('_kp_2d_pred')}
    return g, members


@pytest.mark.parametrize('seed', MULTI_SEEDS)
def test_full_schedule_detection_gap_multi_seed(make_model, golden, seed):
    """bench.py's workload is synth.make_in_dict(seed) for seeds 0 .. B-1 with person 0 undetected in [100, 160): the zero-camera regime of
    DESIGN.md 4, where the unmodified reference itself has neighbouring solutions within 1e-6 of its own start.  NINE seeds, each through
    (a) the numpy init_data (the reference's own initial state): the 500-iteration result VALUE BY VALUE against the reference's, and
    (b) the default path (init_data on the device, a few 1e-6 from the reference's start).  Both must end ON the reference's result (0.25 px),
    or -- where the unmodified reference itself moves by pixels when it is re-run with its initial cameras perturbed in the 6th / 7th digit,
    another thread count, or its predicted trajectory perturbed in the 7th digit (the fam_* members of the fixture: seeds 0, 1, 4, 6) -- on
    one of those members or inside their envelope; always with the reference's reprojection quality.  Seeds without a family (5, 7, 8) and
    seeds whose family stays within 0.25 px (2, 3) are held to the reference's result itself."""
    g, members = _seed_fixture(golden, seed)
    md = synth.make_smpl_model()
    in_dict = synth.make_in_dict(seed=seed, num_frames=300, num_persons=1, smpl_model=md)
    lat = mg.latents_for(in_dict, seed)
    vis = g['p0_vis_frames']
    ref = g['p0_kp_2d_pred']
    obs = in_dict['est'][0]['kp_2d'][:, :24, :2]
    reproj = lambda kp: float(np.linalg.norm(kp[vis][:, :24] - obs, axis=-1).mean())
    per_frame = lambda kp, other: np.abs(kp - other)[vis].max(axis=(1, 2))
    res = {}
    for leg in ('host', 'device'):
        model = make_model('glamr_dynamic')
        if leg == 'host':
            model.init_data_batch = model.init_data_batch_host
        out = model.optimize(in_dict, latents=lat)
        kp = out['person_data'][0]['kp_2d_pred']
        d = per_frame(kp, ref)
        near = min([(float(per_frame(kp, v).max()), k) for k, v in members.items()] + [(float(d.max()), 'golden')])
        from tests.grecon_common import full_schedule_errors
        w = full_schedule_errors(out['person_data'], out['cam_pose'], g, 1)
        res[leg] = dict(kp=float(d.max()), over1=int((d > 1).sum()), nearest=near, reproj=reproj(kp), w=w)
        print('seed %d, %s init: %.3f px from the reference (%d frames > 1 px), nearest member %s at %.3f px; world root %.1e m, orientation %.1e, cam_pose %.1e / %.1e m; '
              'reprojection %.3f vs reference %.3f px' % (seed, leg, d.max(), int((d > 1).sum()), near[1], near[0], w['root_world'], w['orient'], w['cam_rot'], w['cam_trans'],
                                                          reproj(kp), reproj(ref)))
    spread = [(float(per_frame(v, ref).max()), int((per_frame(v, ref) > 1).sum())) for v in members.values()]
    res['family'] = spread
    ref_q = reproj(ref)
    env_px, env_n = (max(s[0] for s in spread), max(s[1] for s in spread)) if spread else (0.0, 0)
    for leg in ('host', 'device'):
        r = res[leg]
        assert abs(r['reproj'] - ref_q) < 0.02 * max(ref_q, 1.0), leg          # the reference's quality, whatever the basin
        # where it landed: on the reference's result, on a member of the reference's own family of re-runs, inside the family's envelope
        r['where'] = ('reference' if r['kp'] < 0.25 else 'member' if r['nearest'][0] < 0.25 else
                      'envelope' if (spread and r['kp'] <= 1.5 * env_px + 0.5 and r['over1'] <= env_n + 10) else 'outside')
    _MULTI[seed] = res
    # a run that is only INSIDE THE ENVELOPE of the reference's family (not on a member) must at least not have left the
    # reference's trajectory before the reference's own perturbed re-runs do.  The fixture holds the unmodified reference's camera parameters
    # after each of its first 64 Adam steps and the step at which each of five re-runs (initial cameras x (1 + 1e-7 / 1e-6 U)) is more than
    # 1e-4 away from them (oracle/make_golden.py gen_seed_divergence); the kernel's own state after k steps comes from a k-iteration launch.
    if 'div_ref_cam' in g and any(res[leg]['where'] == 'envelope' for leg in ('host', 'device')):
        a, b = [int(x) for x in g['div_frames']]
        ref_traj = g['div_ref_cam']
        member_steps = sorted(int(g[k]) for k in g if k.startswith('div_iter_'))
        for leg in ('host', 'device'):
            if res[leg]['where'] != 'envelope':
                continue
            model = make_model('glamr_dynamic')
            if leg == 'host':
                model.init_data_batch = model.init_data_batch_host
            first = len(ref_traj) + 1
            for k in range(1, len(ref_traj) + 1):
                o = model.optimize(in_dict, latents=lat, max_iters=k)
                cam = np.concatenate([np.asarray(o['cam_rot_6d']), np.asarray(o['cam_trans'])], axis=1)[a:b]
                if float(np.abs(cam.astype(np.float64) - ref_traj[k - 1]).max()) > 1e-4:
                    first = k
                    break
            res[leg]['first_divergence'] = first
            print('seed %d, %s init: leaves the reference trajectory (camera parameters of frames [%d, %d), 1e-4) after step %d; the reference\'s own re-runs after steps %s'
                  % (seed, leg, a, b, first, member_steps))
            assert first >= member_steps[0], 'left the reference trajectory at step %d, before any of its own re-runs (%s)' % (first, member_steps)
    for leg in ('host', 'device'):
        r = res[leg]
        assert r['where'] != 'outside', ('%s init: %.3f px from the reference, %d frames > 1 px, nearest member %s at %.3f px; family envelope %.3f px, %d frames'
                                         % (leg, r['kp'], r['over1'], r['nearest'][1], r['nearest'][0], env_px, env_n))


def test_multi_seed_summary():
    """How many of the nine seeds land where: printed for DESIGN.md 4 (needs the parametrised test above to have run in this process)."""
    if len(_MULTI) < len(MULTI_SEEDS):
        pytest.skip('the per-seed cases did not all run')
    for leg in ('host', 'device'):
        by = {}
        for s, r in sorted(_MULTI.items()):
            by.setdefault(r[leg]['where'], []).append('%d (%.2f px)' % (s, r[leg]['kp']) if r[leg]['where'] != 'reference' else str(s))
        print('multi-seed full schedules (configs[1] with the detection gap), %d seeds, %s init_data: %s' % (len(_MULTI), leg, by))
    print('reference families (max px / frames > 1 px per member of the reference\'s own re-runs): %s'
          % {s: ['%.2f/%d' % m for m in r['family']] for s, r in sorted(_MULTI.items()) if r['family']})
    # seeds whose reference family stays within 0.25 px of the reference are well-conditioned: there both starts must be ON the reference
    calm = [s for s, r in _MULTI.items() if r['family'] and max(m[0] for m in r['family']) < 0.25]
    assert all(_MULTI[s]['host']['where'] == 'reference' and _MULTI[s]['device']['where'] == 'reference' for s in calm), calm


def test_run_demo_entry_point(asset_root, tmp_path, monkeypatch):
    """pose.pkl in -> grecon/<seq>_seed<k>.pkl out, with the reference's working-directory conventions (run_demo.py:44-82)."""
    import pickle
    from glamr_amd.global_recon import run_demo
    est = synth.make_in_dict(seed=2, num_frames=100, num_persons=1, smpl_model=synth.make_smpl_model())['est']
    pose_dir = tmp_path / 'out' / 'walk' / 'pose_est'
    os.makedirs(pose_dir)
    with open(pose_dir / 'pose.pkl', 'wb') as f:
        pickle.dump(est, f)
    monkeypatch.chdir(asset_root)            # data/body_models/smpl, data/J_regressor_extra.npy, results/... are found relative to cwd
    out_file = run_demo.main(['--cfg', 'glamr_static', '--pose_est_dir', str(pose_dir), '--out_dir', str(tmp_path / 'out' / 'walk'), '--seed', '3'])
    assert os.path.basename(out_file) == 'walk_seed3.pkl'
    out = pickle.load(open(out_file, 'rb'))
    assert out['seq_name'] == 'walk' and out['meta']['num_fr'] == 100 and out['cam_pose'].shape == (100, 4, 4)
    pd = out['person_data'][0]
    assert pd['smpl_orient_world'].shape == (100, 3) and pd['kp_2d_pred'].shape == (100, 26, 2) and np.isfinite(pd['root_trans_world']).all()
    # cached=1: a second call returns the stored file without touching the device
    assert run_demo.main(['--cfg', 'glamr_static', '--pose_est_dir', str(pose_dir), '--out_dir', str(tmp_path / 'out' / 'walk'), '--seed', '3']) == out_file


_trim_person = synth.trim_person


def test_person_entering_late_and_leaving_early_matches_the_oracle(make_model, asset_root):
    """Edge case of init_data's frame bookkeeping (global_recon_model.py:92-95,146-147): a person whose existence range is strictly
    inside the video.  Device path vs the CPU restatement (oracle/port, pinned to the reference on the fixture cases)."""
    from oracle.port import build
    from glamr_amd.global_recon.configs import get_config
    md = synth.make_smpl_model()
    in_dict = _trim_person(synth.make_in_dict(seed=11, num_frames=100, num_persons=2, smpl_model=md), 1, 17, 83)
    lat = mg.latents_for(in_dict, 11)
    ora = build.load_optimizer(asset_root, get_config('glamr_dynamic_multi'))
    ref = ora.init_data(in_dict, latents=lat)
    data = make_model('glamr_dynamic_multi').init_data(in_dict, latents=lat)
    for idx in (0, 1):
        a, b = data['person_data'][idx], ref['person_data'][idx]
        assert int(a['fr_start']) == int(b['fr_start']) and int(a['fr_end']) == int(b['fr_end'])
        for key in ('visible', 'visible_orig', 'exist_frames', 'vis_frames', 'invis_frames'):
            assert np.array_equal(np.asarray(a[key]), np.asarray(b[key]).astype(np.asarray(a[key]).dtype)), key
        for key, tol in (('smpl_pose', 1e-4), ('traj_local_pred', 1e-4), ('root_trans_world', 3e-4), ('kp_2d_pred', 5e-2)):
            err = np.abs(np.asarray(a[key], np.float64) - b[key].detach().numpy().astype(np.float64)).max()
            assert err < tol, '%d %s: %g' % (idx, key, err)
    seen = np.flatnonzero(in_dict['est'][1]['bboxes_dict']['exist'])
    assert int(data['person_data'][1]['fr_start']) == seen[0] == 17 and int(data['person_data'][1]['fr_end']) == seen[-1] + 1 < 100


def test_nine_person_scene_with_ragged_existence_matches_the_oracle(make_model, asset_root):
    """A scene that needs the wide instances of the stage kernel (csrc/grecon_wide.hip: more than 8 persons) AND the frame bookkeeping of persons
    that enter late, leave early or go undetected for a while, in one batch with a short single-person sequence (person slots and frames padded).
    Device path against the CPU restatement (oracle/port, pinned to the reference on the fixture cases): init_data, then 4 iterations per stage."""
    import copy
    from oracle.port import build
    from glamr_amd.global_recon.configs import get_config
    md = synth.make_smpl_model()
    in_dict = synth.make_in_dict(seed=21, num_frames=64, num_persons=9, smpl_model=md, gap=(20, 30))
    synth.trim_person(in_dict, 2, 9, 50)
    synth.trim_person(in_dict, 5, 0, 41)
    synth.trim_person(in_dict, 8, 23, 64)
    other = synth.make_in_dict(seed=22, num_frames=40, num_persons=1, smpl_model=md)
    lat, lat_o = mg.latents_for(in_dict, 21), mg.latents_for(other, 22)
    K = 4
    cfg = get_config('glamr_dynamic_multi')
    for spec in cfg['opt_stage_specs'].values():
        spec['opt_niters'] = K
    ora = build.load_optimizer(asset_root, cfg)
    ref = ora.optimize(copy.deepcopy(in_dict), latents=lat)
    model = make_model('glamr_dynamic_multi')
    out = model.optimize_batch([in_dict, other], latents=[lat, lat_o], max_iters=K)[0]
    alone = model.optimize(in_dict, latents=lat, max_iters=K)
    worst = [0.0, 0.0, 0.0]
    for pi in range(9):
        a, b = out['person_data'][pi], ref['person_data'][pi]
        assert int(a['fr_start']) == int(b['fr_start']) and int(a['fr_end']) == int(b['fr_end'])
        for key in ('visible', 'exist_frames', 'vis_frames'):
            assert np.array_equal(np.asarray(a[key]), np.asarray(b[key]).astype(np.asarray(a[key]).dtype)), key
        vis = np.asarray(b['vis_frames']).astype(bool) & np.asarray(ref['person_data'][0]['vis_frames']).astype(bool)
        worst[0] = max(worst[0], kp_err(a['kp_2d_pred'], np.asarray(b['kp_2d_pred']), vis))
        worst[1] = max(worst[1], float(np.abs(np.asarray(a['root_trans_world'], np.float64) - np.asarray(b['root_trans_world'])).max()))
        worst[2] = max(worst[2], _rot_err(a['smpl_orient_world'], np.asarray(b['smpl_orient_world'])))
        # padding the batch (a second, shorter sequence with one person) changes nothing
        assert np.array_equal(np.asarray(a['kp_2d_pred']), np.asarray(alone['person_data'][pi]['kp_2d_pred']))
    print('9 persons, ragged existence, %d iterations per stage, against the CPU restatement: kp %.4f px, root_trans_world %.2e m, smpl_orient_world %.2e' % (K, *worst))
    assert worst[0] < 0.1 and worst[1] < 1e-3 and worst[2] < 5e-3
    assert [int(out['person_data'][i]['fr_start']) for i in (2, 5, 8)] == [9, 0, 23]
    # and the whole step as a replayed HIP graph (capture_resident: the entry bench.py uses) gives the plain launches' numbers for such a scene too
    rin = model.stage_inputs([in_dict, other], [lat, lat_o])
    graph = model.capture_resident(rin, max_iters=K, check=True)
    graph.replay()
    import torch
    torch.cuda.synchronize()
    res = model.collect(graph.datas, graph.packed)[0]
    for pi in range(9):
        assert np.array_equal(np.asarray(res['person_data'][pi]['kp_2d_pred']), np.asarray(out['person_data'][pi]['kp_2d_pred']))


def test_ragged_batch_equals_single_runs(make_model):
    """optimize_batch pads sequences to the longest one and person slots to the largest count: every sequence of a mixed batch must
    come out as when it is run alone (up to the rounding of the different kernel instances: single-person scenes run a
    specialised instance when they are alone)."""
    md = synth.make_smpl_model()
    specs = [(21, 100, 1), (22, 80, 2), (23, 120, 1)]
    in_dicts = [synth.make_in_dict(seed=s, num_frames=T, num_persons=P, smpl_model=md) for s, T, P in specs]
    lats = [mg.latents_for(d, s) for d, (s, T, P) in zip(in_dicts, specs)]
    model = make_model('glamr_dynamic_multi')
    K = 8
    batch = model.optimize_batch(in_dicts, lats, max_iters=K)
    for d, lat, out_b, (s, T, P) in zip(in_dicts, lats, batch, specs):
        out_s = model.optimize(d, latents=lat, max_iters=K)
        assert out_b['seq_len'] == T and out_b['cam_pose'].shape == (T, 4, 4) and len(out_b['person_data']) == P
        for idx in out_s['person_data']:
            a, b = out_b['person_data'][idx], out_s['person_data'][idx]
            vis = np.asarray(a['vis_frames'])
            assert np.array_equal(vis, np.asarray(b['vis_frames']))
            assert np.abs(a['kp_2d_pred'] - b['kp_2d_pred'])[vis].max() < 0.5, 'seed %d person %d' % (s, idx)
            assert np.abs(a['root_trans_world'] - b['root_trans_world']).max() < 5e-3
            assert a['smpl_pose'].shape == (T, 69)


def test_stream_of_batches_equals_batch_calls(make_model):
    """optimize_stream (the host-side software pipeline: pinned staging + uploads of batch i + 1 and output dictionaries of batch i - 1
    under the device work of batch i) returns, batch by batch and key by key, what optimize_batch returns for the same batches."""
    import pickle
    md = synth.make_smpl_model()
    batches, lats = [], []
    for b in range(3):
        specs = [(60 + 3 * b + i, 90 + 10 * i, 1 + (i % 2)) for i in range(3)]
        batches.append([synth.make_in_dict(seed=s, num_frames=T, num_persons=P, smpl_model=md) for s, T, P in specs])
        lats.append([mg.latents_for(d, s) for d, (s, T, P) in zip(batches[-1], specs)])
    model = make_model('glamr_dynamic_multi')
    K = 6
    ref = [model.optimize_batch(b, l, max_iters=K) for b, l in zip(batches, lats)]
    got = list(model.optimize_stream(batches, lats, max_iters=K))
    assert len(got) == 3
    for rb, gb in zip(ref, got):
        assert len(rb) == len(gb)
        for r, g in zip(rb, gb):
            assert set(r.keys()) == set(g.keys()) and r['seq_name'] == g['seq_name']
            for key in ('cam_pose', 'cam_pose_inv', 'cam_rot_6d', 'cam_trans', 'cam_inv_trans_residual', 'fr_num_persons'):
                assert np.array_equal(np.asarray(r[key]), np.asarray(g[key])), key
            for idx in r['person_data']:
                a, b = r['person_data'][idx], g['person_data'][idx]
                assert set(a.keys()) == set(b.keys())
                for key in a.keys():
                    if isinstance(a[key], np.ndarray):
                        assert a[key].dtype == b[key].dtype and np.array_equal(a[key], b[key], equal_nan=True), key
    # a second pass over the same batches REPLAYS the priors' launch sequences from the HIP graphs captured during the first one (same
    # stream, same resident buffers): bit-identical results
    again = list(model.optimize_stream(batches, lats, max_iters=K))
    for gb, ab in zip(got, again):
        for g, a in zip(gb, ab):
            for idx in g['person_data']:
                for key in ('kp_2d_pred', 'smpl_pose', 'traj_local_pred', 'root_trans_world'):
                    assert np.array_equal(g['person_data'][idx][key], a['person_data'][idx][key]), key
    # the lazily built dictionaries are ordinary dictionaries to every consumer: pickling (run_demo.py writes them) gives plain dicts
    back = pickle.loads(pickle.dumps(got[0][0]))
    assert type(back) is dict and type(back['person_data'][0]) is dict and back['person_data'][0]['visible'].dtype == np.float64
    assert np.array_equal(back['person_data'][0]['kp_2d_pred'], got[0][0]['person_data'][0]['kp_2d_pred'])


def test_value_checks_of_the_wire_format_run_on_the_device(make_model):
    """stage_inputs checks keys / shapes on the host and the VALUES (finite numbers, orthonormal rotation matrices) on the uploaded
    arrays; the verdict surfaces as a WireFormatError naming the sequence and the person when the results are collected."""
    from glamr_amd.utils import wire
    md = synth.make_smpl_model()
    good = synth.make_in_dict(seed=5, num_frames=60, num_persons=1, smpl_model=md)
    bad = synth.make_in_dict(seed=6, num_frames=60, num_persons=2, smpl_model=md)
    bad['est'][1]['smpl_pose_quat_wroot'] = bad['est'][1]['smpl_pose_quat_wroot'] * 1.2          # no longer rotation matrices
    model = make_model('glamr_dynamic_multi')
    with pytest.raises(wire.WireFormatError, match=r'sequence 1 .*person 1.*rotation matrices'):
        model.optimize_batch([good, bad], max_iters=2)
    nan = synth.make_in_dict(seed=7, num_frames=60, num_persons=1, smpl_model=md)
    nan['est'][0]['root_trans'][3, 1] = np.nan
    with pytest.raises(wire.WireFormatError, match='non-finite'):
        model.optimize_batch([good, nan], max_iters=2)
    assert len(model.optimize_batch([good], max_iters=2)) == 1


def test_shared_cu_arena_gives_the_same_values(make_model, monkeypatch):
    """More scenes than CUs and sequences of <= 256 frames: the optimiser caps its LDS arena so that several workgroups share a CU
    (part of the keypoint table then lives in the workspace).  Where an array lives must not change a single bit: the same batch is
    run with the occupancy policy, with the full 150 KB arena forced and with the arena squeezed below the keypoint table."""
    md = synth.make_smpl_model()
    T, n, K = 120, 288, 6
    in_dicts = [synth.make_in_dict(seed=40 + (i % 6), num_frames=T, num_persons=1, smpl_model=md) for i in range(n)]
    lats = [mg.latents_for(in_dicts[i], 40 + (i % 6)) for i in range(n)]
    model = make_model('glamr_dynamic')
    policy = model.optimize_batch(in_dicts, lats, max_iters=K)
    monkeypatch.setenv('GLAMR_GRECON_LDS_KB_RT', '150')
    full = model.optimize_batch(in_dicts, lats, max_iters=K)
    monkeypatch.setenv('GLAMR_GRECON_LDS_KB_RT', '26')
    tiny = model.optimize_batch(in_dicts, lats, max_iters=K)
    for i in (0, 5, 6, 100, 287):
        for other in (full, tiny):
            a, b = policy[i]['person_data'][0], other[i]['person_data'][0]
            for key in ('kp_2d_pred', 'root_trans_world', 'smpl_orient_world'):
                assert np.array_equal(a[key], b[key]), (i, key)
            assert np.array_equal(policy[i]['cam_pose'], other[i]['cam_pose'])
        # identical inputs in different slots of the batch give identical results
        assert np.array_equal(policy[i]['person_data'][0]['kp_2d_pred'], policy[i % 6]['person_data'][0]['kp_2d_pred'])


def test_mid_arena_gives_the_same_values(make_model, monkeypatch):
    """Scenes of several persons whose full arena does not fit a CU's LDS (BASELINE configs[3]: 4 persons x 300 frames; 3 persons likewise) run on
    the MID arena since round 6 -- the lite arena's arrays plus world translation and the two adjoint hand-over arrays on chip (instances
    grecon_stage_kernel<4, false, CAM>).  Where an array lives must not change a single bit: against the lite arena (GLAMR_GRECON_NO_MID_ARENA, read
    per launch), both stages of cfg glamr_static_multi (constant camera, then the shared optimised one), 12 iterations each, ragged existence ranges."""
    md = synth.make_smpl_model()
    in_dicts = [_trim_person(synth.make_in_dict(seed=70, num_frames=300, num_persons=4, smpl_model=md), 2, 11, 280),
                synth.make_in_dict(seed=71, num_frames=300, num_persons=3, smpl_model=md, gap=(120, 150))]
    model = make_model('glamr_static_multi')
    out = {}
    for name, env in (('mid', None), ('lite', '1')):
        if env is None:
            monkeypatch.delenv('GLAMR_GRECON_NO_MID_ARENA', raising=False)
        else:
            monkeypatch.setenv('GLAMR_GRECON_NO_MID_ARENA', env)
        out[name] = [model.optimize(d, latents=mg.latents_for(d, 70 + i), max_iters=12) for i, d in enumerate(in_dicts)]
    for a, b in zip(out['mid'], out['lite']):
        assert np.array_equal(a['cam_pose'], b['cam_pose'])
        for pid in a['person_data']:
            for key in ('kp_2d_pred', 'root_trans_world', 'smpl_orient_world', 'traj_local_rot', 'traj_local_dxy'):
                assert np.array_equal(a['person_data'][pid][key], b['person_data'][pid][key]), (pid, key)


def test_constant_layout_instance_gives_the_same_values(make_model, monkeypatch):
    """BASELINE configs[1]'s length selects the constant-layout instances of the stage kernel (arena, workspace and on-chip parameter
    blocks laid out for 304 frames, every address of the loop a compile-time constant, stage-constant inputs read from workspace
    copies).  Same arithmetic at other addresses: against the run-time-layout instances (GLAMR_GRECON_NO_CONST_LAYOUT, read per launch)
    not a bit may change -- with and without a detection gap, a sequence shorter than the padded length in the batch, 40 iterations of
    every stage."""
    md = synth.make_smpl_model()
    in_dicts = [synth.make_in_dict(seed=0, num_frames=300, num_persons=1, smpl_model=md, gap=(100, 160)),
                synth.make_in_dict(seed=1, num_frames=300, num_persons=1, smpl_model=md),
                synth.make_in_dict(seed=2, num_frames=270, num_persons=1, smpl_model=md, gap=(30, 50))]
    lats = [mg.latents_for(d, i) for i, d in enumerate(in_dicts)]
    model = make_model('glamr_dynamic')
    const = model.optimize_batch(in_dicts, lats, max_iters=40)
    monkeypatch.setenv('GLAMR_GRECON_NO_CONST_LAYOUT', '1')
    plain = model.optimize_batch(in_dicts, lats, max_iters=40)
    for a, b in zip(const, plain):
        for key in ('kp_2d_pred', 'root_trans_world', 'smpl_orient_world', 'smpl_orient_cam', 'root_trans_cam'):
            assert np.array_equal(a['person_data'][0][key], b['person_data'][0][key]), key
        assert np.array_equal(a['cam_pose'], b['cam_pose'])


def test_sequence_longer_than_a_workgroup(make_model, asset_root):
    """700 frames: the optimiser workgroup has 512 threads, so every frame loop makes two passes and the prefix sums run in two chunks;
    24 infiller windows.  Device path vs the CPU restatement on init_data and after 3 iterations of the stage."""
    from oracle.port import build
    from glamr_amd.global_recon.configs import get_config
    T, K = 700, 3
    cfg = get_config('glamr_dynamic')
    in_dict = synth.make_in_dict(seed=9, num_frames=T, num_persons=1, smpl_model=synth.make_smpl_model(), gap=(300, 380))
    lat = mg.latents_for(in_dict, 9)
    ora = build.load_optimizer(asset_root, cfg)
    data = ora.init_data(in_dict, latents=lat)
    ref_init = data['person_data'][0]['kp_2d_pred'].detach().numpy().copy()
    for stage, spec in cfg['opt_stage_specs'].items():
        ora.optimize_main(data, spec['opt_variables'], spec['opt_lr'], min(K, spec['opt_niters']), spec['loss_cfg'], {'stage': stage})
    model = make_model('glamr_dynamic')
    init = model.init_data(in_dict, latents=lat)
    out = model.optimize(in_dict, latents=lat, max_iters=K)
    vis = np.asarray(out['person_data'][0]['vis_frames'])
    assert vis.sum() == T - 80 and out['cam_pose'].shape == (T, 4, 4)
    assert kp_err(init['person_data'][0]['kp_2d_pred'], ref_init, vis) < 5e-2
    assert kp_err(out['person_data'][0]['kp_2d_pred'], data['person_data'][0]['kp_2d_pred'].detach().numpy(), vis) < 0.5
    assert np.abs(out['person_data'][0]['root_trans_world'] - data['person_data'][0]['root_trans_world'].detach().numpy()).max() < 1e-2


def test_run_dataset_reconstructs_and_evaluates(asset_root, tmp_path, monkeypatch):
    """pose.pkl + ground-truth pickle per sequence -> result pickles + the metric line (run_dataset.py:60-120), two seeds, one batch."""
    import pickle
    from glamr_amd.global_recon import run_dataset
    md = synth.make_smpl_model()
    if not os.path.exists(os.path.join(asset_root, 'data', 'J_regressor_h36m.npy')):
        np.save(os.path.join(asset_root, 'data', 'J_regressor_h36m.npy'), synth.make_h36m_regressor(md))
    out_dir, gt_dir = tmp_path / 'out', tmp_path / 'gt'
    os.makedirs(gt_dir)
    for seq, seed, T in (('a', 31, 90), ('b', 32, 110)):
        d = synth.make_in_dict(seed=seed, num_frames=T, num_persons=1, smpl_model=md, with_gt=True)
        os.makedirs(out_dir / seq / 'pose_est')
        pickle.dump(d['est'], open(out_dir / seq / 'pose_est' / 'pose.pkl', 'wb'))
        pickle.dump({'person_data': d['gt'], 'meta': {}}, open(gt_dir / (seq + '.pkl'), 'wb'))
    monkeypatch.chdir(asset_root)
    line = run_dataset.main(['--cfg', 'glamr_static', '--dataset', '', '--seqs', 'a', 'b', '--out_dir', str(out_dir), '--gt_dir', str(gt_dir), '--seeds', '1', '2'])
    assert 'G-MPJPE' in line and 'PA-MPJPE-invis' in line and 'sample_PA-MPJPE-invis' in line
    out = pickle.load(open(out_dir / 'b' / 'grecon' / 'b_seed2.pkl', 'rb'))
    assert out['seq_len'] == 110 and out['gt'][0]['pose'].shape == (110, 72)


def test_continue_opt_restarts_from_a_previous_result(make_model, asset_root):
    """optimize(out, continue_opt=True) (global_recon_model.py:572-573): the schedule runs again from the variables of a previous
    result.  Device path vs the CPU restatement continuing from the SAME dictionary."""
    import copy
    from oracle.port import build
    from oracle.port.grecon import to_torch
    from glamr_amd.global_recon.configs import get_config
    cfg = get_config('glamr_dynamic')
    in_dict = synth.make_in_dict(seed=13, num_frames=80, num_persons=1, smpl_model=synth.make_smpl_model())
    model = make_model('glamr_dynamic')
    first = model.optimize(in_dict, latents=mg.latents_for(in_dict, 13), max_iters=4)
    K = 3
    again = model.optimize(copy.deepcopy(first), continue_opt=True, max_iters=K)
    ora = build.load_optimizer(asset_root, cfg)
    data = to_torch(copy.deepcopy(first), torch.device('cpu'))
    for stage, spec in cfg['opt_stage_specs'].items():
        ora.optimize_main(data, spec['opt_variables'], spec['opt_lr'], min(K, spec['opt_niters']), spec['loss_cfg'], {'stage': stage})
    vis = np.asarray(again['person_data'][0]['vis_frames'])
    ref_kp = data['person_data'][0]['kp_2d_pred'].detach().numpy()
    assert kp_err(again['person_data'][0]['kp_2d_pred'], ref_kp, vis) < 0.5
    assert np.abs(again['person_data'][0]['root_trans_world'] - data['person_data'][0]['root_trans_world'].detach().numpy()).max() < 1e-2
    # it did move away from where it started, and the input dictionary was left alone
    assert np.abs(again['person_data'][0]['traj_local_rot'] - first['person_data'][0]['traj_local_rot']).max() > 1e-4
    assert 'world_dheading' in again['person_data'][0]


def test_bench_step_graph_reproduces_the_plain_step():
    """bench.py captures a whole step as one HIP graph per stream and replays it in the timed region; its self-check compares a replay with a
    plain step bit for bit (same seed) and falls back to plain launches otherwise.  The short run here must have used the graph."""
    import bench
    out = bench.run(['--steps', '2', '--warmup', '1', '--batch', '64', '--no-cpu-baseline', '--no-kernel-lines'])
    assert out['config']['step_graph'] is True
    assert out['value'] > 0 and out['steps'] == 2
    # the check that runs after the clock has stopped: both streams' graphs, alternated, two seeds, against plain steps of the same seeds
    rc = out['replay_check']
    assert rc['bit_identical_to_plain_steps'] is True and rc['max_projection_difference_px'] == 0.0 and rc['seeds'] >= 2, rc
    plain = bench.run(['--steps', '2', '--warmup', '1', '--batch', '64', '--no-cpu-baseline', '--no-kernel-lines', '--no-graph-step'])
    assert plain['config']['step_graph'] is False


def test_bench_runs_under_an_rccl_process_group_of_one_rank(monkeypatch):
    """The multi-GPU launch of bench.py (`torch.distributed.run --nproc-per-node N`) on the one GPU this box has: a world of ONE rank over
    the `nccl` backend (= RCCL) goes through everything the 8-GPU run will -- process-group init with the device id, the asset barrier,
    the capture of the step graph while the group's watchdog thread is alive, the barriers around the timed region, the max / sum
    reductions on device tensors and the small-collective latency probe -- so that run is not the first execution of that code."""
    import socket
    import bench
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    for k, v in (('MASTER_ADDR', '127.0.0.1'), ('MASTER_PORT', str(port)), ('RANK', '0'), ('LOCAL_RANK', '0'), ('WORLD_SIZE', '1')):
        monkeypatch.setenv(k, v)
    out = bench.run(['--gpus', '1', '--steps', '2', '--warmup', '1', '--batch', '64', '--no-cpu-baseline', '--no-kernel-lines', '--force-dist'])
    assert out['n_gpus'] == 1 and out['value'] > 0 and out['config']['step_graph'] is True
    coll = out['collective_alternative']
    assert coll['ranks'] == 1 and 0 < coll['us_per_iteration_allreduce9_plus_allgather_4x300x12'] < 5e3
    import torch.distributed as dist
    assert not dist.is_initialized()          # the group is torn down again


FULL_CFG_CASES = [(c, T, P, gap) for (c, T, P, _seed) in mg.FULL_CASES for gap in (False, True)]
# Bounds of the value-by-value comparison, ~3 x achieved on the MI355X (keypoints px, root in the camera frame m).  The per-frame-camera
# multi-person cases WITH detection gaps are the zero-camera situation of DESIGN.md 4 (cameras of frames the first person is not seen in
# start as zero matrices: 1e9 gradients, the result hangs on the last bit of a sum over joints of TWO persons that no other summation order
# reproduces): they are held to the reference's own family of solutions instead (full_<cfg>_family.npz) and to its reprojection quality.
# cases held to the reference's OWN family of solutions: (cfg, gap) -> gap flag of the family fixture.  glamr_dynamic_multi with gaps: the
# unmodified reference re-run with 3 threads moves by 34.5 px, with cam_pose x (1 + 1e-7 U) by 28.2 px (zero cameras); glamr_h36m (per-frame
# camera optimised at lr 1e-2 under 1e4-weighted smoothness terms): 3.7 px / 8.0 px -- there is no single reference answer to match.
FULL_FAMILY = {('glamr_dynamic_multi', True): True, ('glamr_h36m', False): False, ('glamr_h36m', True): False}
# (keypoints px, root in the camera frame m, world root m, world orientation as a rotation, orientation seen from the camera, camera rotation,
#  camera translation m): ~3 x the values achieved on the MI355X (in the comments).  None = not comparable (gauge, see below the table).
FULL_TOL_GPU = {          # achieved on the MI355X (round 4):  kp px   root_cam m  root_world m  orient   orient_cam  cam_rot  cam_trans m
    ('glamr_3dpw', 1, False): (0.1, 1e-3, None, None, 1e-4, None, None),                  # 0.030  2.5e-4  (1.6e-1)  (1.0e-2)  1.0e-6  (7.2e-3)  (1.4e-1)
    ('glamr_3dpw', 1, True): (0.1, 1e-3, None, None, 1e-4, None, None),                   # 0.026  2.3e-4  (2.3e-1)  (4.5e-5)  1.0e-6  (4.5e-5)  (2.4e-1)
    ('glamr_dynamic_multi', 2, False): (0.01, 5e-5, 3e-5, 5e-5, 5e-5, 1e-5, 5e-5),        # 0.0011 7.3e-6  6.7e-6  1.2e-5  1.2e-5  1.5e-6  1.4e-5
    ('glamr_static_multi', 4, True): (0.06, 2e-4, 2e-4, 5e-4, 5e-4, 5e-6, 5e-6),          # 0.0195 5.4e-5  5.0e-5  1.3e-4  1.1e-4  1.8e-7  7.2e-7   BASELINE configs[3]
    ('glamr_static_multi', 4, False): (0.03, 2e-4, 2e-4, 5e-5, 5e-5, 5e-6, 5e-6),         # 0.0072 5.3e-5  4.7e-5  7.1e-6  7.0e-6  1.8e-7  6.0e-7
    ('glamr_static', 1, False): (0.01, 2e-5, 2e-5, 1e-4, 1e-4, 5e-6, 5e-6),               # 0.0008 3.8e-6  3.3e-6  1.9e-5  1.9e-5  2.4e-7  4.8e-7
    ('glamr_static', 1, True): (0.1, 6e-5, 2e-4, 6e-4, 6e-4, 5e-5, 5e-6),                 # 0.0140 1.6e-5  4.8e-5  1.4e-4  1.3e-4  8.5e-6  2.1e-7
}
# glamr_3dpw DERIVES the camera from the person's world pose (flag_opt_cam_from_person_pose): person and camera share the world frame's gauge
# (a rigid motion of both changes no residual but the weak regularisers), so between two runs the WORLD root / orientation / camera wander
# by decimetres / 1e-2 (values in brackets) while everything seen FROM the camera -- root_cam, orient_cam, the projections -- agrees to 1e-6.


@pytest.mark.parametrize('cfg_id,T,P,gap', FULL_CFG_CASES)
def test_full_schedule_of_every_config_matches_the_reference(make_model, golden, cfg_id, T, P, gap):
    """BASELINE configs[3] (glamr_static_multi: 4 persons x 300 frames, shared fixed camera, 200 + 500 iterations) and the full schedules of
    glamr_3dpw (camera from the person's pose), glamr_dynamic_multi, glamr_static and glamr_h36m through the drop-in entry point, starting
    from the numpy init_data (the reference's own initial state), against the unmodified reference's result VALUE BY VALUE."""
    from tests.grecon_common import full_schedule_errors
    g = golden(mg.full_name(cfg_id, T, P, gap))
    seed = mg.FULL_SEED[(cfg_id, T, P)]
    assert int(g['seed']) == seed
    in_dict = synth.make_in_dict(seed=seed, num_frames=T, num_persons=P, smpl_model=synth.make_smpl_model(), gap=None if gap else (0, 0))
    model = make_model(cfg_id)
    model.init_data_batch = model.init_data_batch_host
    out = model.optimize(in_dict, latents=mg.latents_for(in_dict, seed))
    w = full_schedule_errors(out['person_data'], out['cam_pose'], g, P)
    # reprojection quality: mean distance of the projected SMPL joints from the detections, ours vs the reference's
    def reproj(person_kp):
        tot, n = 0.0, 0
        for pi in range(P):
            vis = g['p%d_vis_frames' % pi]
            obs = in_dict['est'][pi]['kp_2d'][:, :24, :2]
            d = np.linalg.norm(np.asarray(person_kp(pi))[vis][:, :24] - obs, axis=-1)
            tot, n = tot + float(d[d < 1e3].sum()), n + int((d < 1e3).sum())
        return tot / max(n, 1)
    ours, ref = reproj(lambda pi: out['person_data'][pi]['kp_2d_pred']), reproj(lambda pi: g['p%d_kp_2d_pred' % pi])
    print('full schedule %s T=%d P=%d gap=%s: kp %.4f px (%d frames > 1 px), root in camera %.2e m, world root %.2e m, orientation %.2e, orientation seen from the camera %.2e, '
          'cam_pose rotation %.2e translation %.2e m; reprojection %.3f vs reference %.3f px'
          % (cfg_id, T, P, gap, w['kp'], w['frames_over_1px'], w['root_cam'], w['root_world'], w['orient'], w['orient_cam'], w['cam_rot'], w['cam_trans'], ours, ref))
    assert abs(ours - ref) < 0.02 * max(ref, 1.0)
    if (cfg_id, gap) in FULL_FAMILY:
        # the reference's own spread on this case (oracle/make_golden.py gen_full_family_cfg: another thread count, cam_pose x (1 + 1e-7 .. 1e-6 U))
        fam = golden(mg.full_name(cfg_id, T, P, FULL_FAMILY[(cfg_id, gap)]) + '_family')
        base = golden(mg.full_name(cfg_id, T, P, FULL_FAMILY[(cfg_id, gap)]))
        spread = {}
        for key, v in fam.items():
            if key.endswith('_kp_2d_pred'):
                name, pi = key[:-len('_kp_2d_pred')].rsplit('_p', 1)
                d = np.abs(v - base['p%s_kp_2d_pred' % pi])[base['p%s_vis_frames' % pi]].max()
                spread[name] = max(spread.get(name, 0.0), float(d))
        print('  reference family: %s' % {k: '%.2f px' % v for k, v in spread.items()})
        assert w['kp'] < 1.5 * max(spread.values()) + 0.5
    if (cfg_id, P, gap) in FULL_TOL_GPU:
        tol_kp, tol_root, tol_world, tol_orient, tol_orient_cam, tol_cam_rot, tol_cam_trans = FULL_TOL_GPU[(cfg_id, P, gap)]
        assert w['kp'] < tol_kp and w['root_cam'] < tol_root and w['frames_over_1px'] == 0
        # what north_star names: smpl_orient_world, root_trans_world, cam_pose after the full schedule (SURVEY 8c asks for <= 1e-3)
        for name, tol in (('root_world', tol_world), ('orient', tol_orient), ('orient_cam', tol_orient_cam), ('cam_rot', tol_cam_rot), ('cam_trans', tol_cam_trans)):
            assert tol is None or w[name] < tol, '%s: %.3e >= %.1e' % (name, w[name], tol)


def test_captured_resident_step_follows_new_inputs(make_model):
    """GlobalReconOptimizer.capture_resident: the whole optimize_resident step as one replayable HIP graph over RESIDENT inputs.  A replay
    after the content of those inputs was overwritten with another batch of the same geometry gives what a plain call on that batch gives."""
    md = synth.make_smpl_model()
    model = make_model('glamr_dynamic')
    batch_a = [synth.make_in_dict(seed=80 + i, num_frames=96, num_persons=1, smpl_model=md) for i in range(5)]
    batch_b = [synth.make_in_dict(seed=90 + i, num_frames=96, num_persons=1, smpl_model=md) for i in range(5)]
    lat_a, lat_b = [mg.latents_for(d, 80 + i) for i, d in enumerate(batch_a)], [mg.latents_for(d, 90 + i) for i, d in enumerate(batch_b)]
    st = torch.cuda.Stream()
    rin = model.stage_inputs(batch_a, lat_a)
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        model.optimize_resident(rin, max_iters=6)                    # first run on this stream: allocations, attribute calls
    torch.cuda.synchronize()
    rg = model.capture_resident(rin, max_iters=6, stream=st, check=True)
    # new content into the SAME resident buffers
    rin_b = model.stage_inputs(batch_b, lat_b)
    torch.cuda.synchronize()
    for k in rin.g:
        rin.g[k].copy_(rin_b.g[k])
    rin.meps.copy_(rin_b.meps)
    rin.teps.copy_(rin_b.teps)
    torch.cuda.synchronize()
    rg.replay()                                                     # enqueued on the side stream
    torch.cuda.synchronize()
    got = rg.packed.t['kp_2d_pred'].clone()
    want = model.optimize_resident(rin_b, max_iters=6)[1].t['kp_2d_pred']
    torch.cuda.synchronize()
    assert torch.equal(got, want) and bool(torch.isfinite(got).all())


@pytest.mark.gpu
@pytest.mark.parametrize('cut', ['early', 'late'])
def test_gated_two_stream_step_graphs_use_the_coschedulable_kernels(make_model, monkeypatch, cut):
    """The staggered two-stream pipeline (PipelineGate): every batch starts its priors when the previous batch's are done, its infiller runs on
    the kernels that fit beside the other stream's optimiser stage (GLAMR_NETS_COSCHEDULE: 48 sequences x 2 windows = 4800 window rows, above
    the 2048-row threshold), and capture_resident cuts the step around the gate: three graphs (preparation | priors + skinning | rest; the
    default) or two (GLAMR_GATE_PREP=late).  The two streams work on DIFFERENT batches, their graphs are replayed alternately twelve times, and
    every array of every replay must equal a plain gated step on the same kernels BIT FOR BIT -- a value read from the other stream, left over
    from the previous replay or computed wrong beside the other stream's kernels is a difference (round 5 accepted 0.3 px here, which the
    corruption of that round passed: profiles/r06_pipeline_corruption.log).  With the LDS kernels forced (GLAMR_NETS_FREE=0) the gated, split,
    replayed step is also the UNGATED plain step bit for bit: gate and cut change nothing but the order of launches across streams."""
    from glamr_amd.global_recon.models.global_recon_model import PipelineGate
    md = synth.make_smpl_model()
    model = make_model('glamr_dynamic')
    monkeypatch.setenv('GLAMR_GATE_PREP', cut)
    rins = []
    for base in (300, 400):
        batch = [synth.make_in_dict(seed=base + i, num_frames=96, num_persons=1, smpl_model=md) for i in range(48)]
        rins.append(model.stage_inputs(batch, [mg.latents_for(d, base + i) for i, d in enumerate(batch)]))
    torch.cuda.synchronize()
    keys = ('kp_2d_pred', 'params', 'j_local', 'cam_pose', 'orient_world', 'trans_world', 'losses')
    ungated = [{k: v.clone() for k, v in model.optimize_resident(r, max_iters=6)[1].t.items() if k in keys} for r in rins]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for forced_lds in (True, False):
        if forced_lds:
            monkeypatch.setenv('GLAMR_NETS_FREE', '0')
        else:
            monkeypatch.delenv('GLAMR_NETS_FREE')
        model.pipeline_gate = PipelineGate()
        try:
            want = []
            for st, r in zip(streams, rins):
                with torch.cuda.stream(st):
                    want.append({k: v.clone() for k, v in model.optimize_resident(r, max_iters=6)[1].t.items() if k in keys})      # a plain GATED step
                torch.cuda.synchronize()
            model.pipeline_gate.last = None
            graphs = [model.capture_resident(r, max_iters=6, stream=st, check=True) for st, r in zip(streams, rins)]
            assert all(g.tail is not None and (g.head is not None) == (cut == 'early') and g.gate is model.pipeline_gate for g in graphs)
            for rep in range(6):
                torch.cuda.synchronize()
                for gi in ((0, 1) if rep % 2 == 0 else (1, 0)):
                    graphs[gi].replay()
                torch.cuda.synchronize()
                for gi, g in enumerate(graphs):
                    for k in keys:
                        got = g.packed.t[k]
                        assert bool(torch.isfinite(got).all()), (cut, forced_lds, rep, gi, k)
                        assert torch.equal(got, want[gi][k]), 'replay %d of stream %d, %s, %s kernels: %s differs from the plain gated step by %.3g' % (
                            rep, gi, cut, 'LDS' if forced_lds else 'co-schedulable', k, float((got.float() - want[gi][k].float()).abs().max()))
                        if forced_lds:
                            assert torch.equal(got, ungated[gi][k]), (cut, rep, gi, k)
                    g.packed.t['kp_2d_pred'].fill_(float('nan'))      # (a replay that did not rewrite its outputs would be seen)
            if not forced_lds:
                d = max(float((want[gi]['kp_2d_pred'] - ungated[gi]['kp_2d_pred']).abs().max()) for gi in range(2))
                print('co-schedulable kernels against the LDS kernels after six iterations: %.2e px' % d)
                assert 0.0 < d < 0.3      # (other kernels for the priors: ~1e-7 there, tests/test_nets_gpu.py; six sign-driven Adam steps later)
        finally:
            model.pipeline_gate = None
