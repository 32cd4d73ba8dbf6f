"""LATENT-OPTIMISATION mode on the MI355X (flag_opt_motion_latent / flag_opt_traj_latent, global_recon_model.py:43-44,155-158,434-437,
619-622): the priors run inside the Adam loop and the latent draws are parameters -- SURVEY.md 8f row 4.  Against fixtures the UNMODIFIED
reference produced with both flags switched on (oracle/make_golden.py gen_grecon_latent): the first iteration's gradient w.r.t.
`motion_latent` (reprojection loss -> stage kernel's dL/d j_local -> glamr_smpl_backward -> glamr_nets_infill_backward through every window)
and the state after K iterations of every stage."""
import os
import numpy as np
import pytest
import torch

from oracle import make_golden as mg
from glamr_amd.utils import synth
from tests.grecon_common import kp_err

pytestmark = pytest.mark.gpu


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

    def make(cfg_id, **flags):
        cfg = get_config(cfg_id)
        cfg['grecon_model_specs'].update(flags)
        return model_dict['global_recon_model'](cfg, dev, None, smpl=smpl, mt_model=mt)
    return make


@pytest.mark.parametrize('cfg_id,T,P,K', mg.LATENT_CASES)
def test_latent_optimisation_matches_the_reference(make_model, golden, cfg_id, T, P, K):
    g = golden('grecon_latent_%s_T%d_P%d' % (cfg_id, T, P))
    in_dict = synth.make_in_dict(seed=3, num_frames=T, num_persons=P, smpl_model=synth.make_smpl_model(), gap=mg.LATENT_GAP.get((cfg_id, T, P)))
    lat = mg.latents_for(in_dict, 3)
    model = make_model(cfg_id, flag_opt_motion_latent=True, flag_opt_traj_latent=True)
    model.latent_trace = {}
    out = model.optimize(in_dict, latents=lat, max_iters=K)
    tr = model.latent_trace
    stage = next(iter(model.opt_stage_specs))
    tol = LATENT_TOL[(cfg_id, T)]
    for pi in range(P):
        # first iteration: what the re-run priors produce, and the gradient that reaches the motion latent (in glamr_dynamic_multi's first
        # stage it is EXACTLY zero in the reference: the only active term there is the first visible frame's reprojection, frame 0 is one of
        # the ten context frames the infiller copies through -- and zero it must be here)
        n = int(g['init_p%d_exist_len' % pi])
        fs = int(g['init_p%d_fr_start' % pi])
        assert np.abs(tr['smpl_pose'][pi, :T] - g['%s_fwd_p%d_smpl_pose' % (stage, pi)]).max() < 1e-4
        assert np.abs(tr['traj_local_pred'][pi, :n] - g['%s_fwd_p%d_traj_local_pred' % (stage, pi)]).max() < 1e-4
        ref = g['%s_grad_p%d_motion_latent' % (stage, pi)]
        got = tr['g_motion_latent'][pi, :ref.shape[0]]
        err = np.abs(got - ref).max() / max(1.0, np.abs(ref).max())
        print('%s T=%d person %d (frames %d..%d): d loss / d motion_latent, first iteration: relative error %.2e (largest %.3g); traj_latent: gradient None in the reference = %s'
              % (cfg_id, T, pi, fs, fs + n, err, np.abs(ref).max(), bool(g['%s_gradnone_p%d_traj_latent' % (stage, pi)])))
        assert err < 1e-3
        assert bool(g['%s_gradnone_p%d_traj_latent' % (stage, pi)])
        # after K iterations of every stage
        pd = out['person_data'][pi]
        vis = g['init_p%d_vis_frames' % pi]
        e_lat = np.abs(pd['motion_latent'] - g['opt_p%d_motion_latent' % pi]).max()
        e_pose = np.abs(pd['smpl_pose'] - g['opt_p%d_smpl_pose' % pi]).max()
        e_kp = kp_err(pd['kp_2d_pred'], g['opt_p%d_kp_2d_pred' % pi], vis)
        e_tr = np.abs(pd['root_trans_world'] - g['opt_p%d_root_trans_world' % pi]).max()
        moved = np.abs(g['opt_p%d_motion_latent' % pi] - lat[pi]['motion']).max()
        print('after %d iterations per stage: motion_latent %.2e (moved %.2e), smpl_pose %.2e, kp_2d_pred %.3f px, root_trans_world %.2e'
              % (K, e_lat, moved, e_pose, e_kp, e_tr))
        assert np.array_equal(pd['traj_latent'], lat[pi]['traj'])                      # never updated: its gradient is None, Adam skips it
        assert np.abs(g['opt_p%d_traj_latent' % pi] - lat[pi]['traj']).max() == 0.0     # ... in the reference too
        assert moved > tol[4]                                                          # the motion latent did move
        assert e_lat < tol[0] and e_pose < tol[1] and e_kp < tol[2] and e_tr < tol[3]


# (motion_latent, smpl_pose rad, projected keypoints px, root_trans_world m) after K iterations per stage
# + how far the latent must have moved in the reference (a case whose latent stays put would test nothing)
LATENT_TOL = {('glamr_dynamic', 100): (1e-4, 1e-5, 0.1, 1e-4, 1e-3),       # achieved 3.4e-6 (moved 1.2e-2), 6.2e-8, 0.013 px, 2.4e-7
              ('glamr_static', 130): (1e-4, 1e-5, 0.1, 1e-4, 1e-3),        #          1.4e-6 (moved 8.1e-3), 5.7e-8, 0.012 px, 4.2e-6
              ('glamr_dynamic_multi', 90): (1e-4, 1e-5, 0.1, 1e-4, 2e-4)}  # two persons, 5 iterations per stage (main stage lr 1e-4)


def test_iteration_graph_equals_plain_launches(make_model, monkeypatch):
    """From the second iteration of a stage on, the mode replays ONE captured HIP graph per iteration (taped infiller, trajectory predictor,
    skinning, gradient launch, SMPL backward, infiller backward, two Adam steps with their step numbers on the device).  Same launches, same
    buffers: the result must equal the plain launch-by-launch schedule (GLAMR_LATENT_GRAPH=0) to the bit -- with two persons whose existence
    ranges differ (ragged fr_start: the gather / scatter between the priors' rows and the video-frame arrays), and the graph must have been used."""
    from tests.test_e2e_gpu import _trim_person
    md = synth.make_smpl_model()
    in_dict = _trim_person(synth.make_in_dict(seed=12, num_frames=100, num_persons=2, smpl_model=md), 1, 13, 91)
    lat = mg.latents_for(in_dict, 12)
    K = 7
    model = make_model('glamr_dynamic_multi', flag_opt_motion_latent=True, flag_opt_traj_latent=True)
    out_g = model.optimize(in_dict, latents=lat, max_iters=K)
    n_stages = len(model.opt_stage_specs)
    assert model.latent_graph_replays == n_stages * (K - 2), model.latent_graph_replays      # iterations 0 and 1 of every stage are plain launches
    monkeypatch.setenv('GLAMR_LATENT_GRAPH', '0')
    model_p = make_model('glamr_dynamic_multi', flag_opt_motion_latent=True, flag_opt_traj_latent=True)
    out_p = model_p.optimize(in_dict, latents=lat, max_iters=K)
    assert model_p.latent_graph_replays == 0
    for idx in (0, 1):
        a, b = out_g['person_data'][idx], out_p['person_data'][idx]
        for key in ('motion_latent', 'smpl_pose', 'kp_2d_pred', 'root_trans_world', 'smpl_orient_world', 'traj_local_pred'):
            assert np.array_equal(a[key], b[key]), (idx, key)
        assert np.abs(a['motion_latent'] - lat[idx]['motion']).max() > 1e-4          # the latents did move
    assert np.array_equal(out_g['cam_pose'], out_p['cam_pose'])


def test_shipped_configs_are_unaffected_and_flags_are_read(make_model):
    m = make_model('glamr_dynamic')
    assert not m.latent_mode
    assert make_model('glamr_dynamic', flag_opt_motion_latent=True).latent_mode and make_model('glamr_dynamic', flag_opt_traj_latent=True).latent_mode


def test_latent_mode_with_detection_gaps_stays_inside_the_reference_family(make_model, golden):
    """VERDICT r4 item 3c.  Two persons, per-frame cameras, latent mode, WITH the detection gaps: person 0 is undetected in [40, 60), and the
    initial camera of a frame comes from person 0 alone (init_cam_pose :294-317), so those twenty frames start as ZERO cameras whose first Adam
    steps are +-lr by the sign of 1e10-sized gradients and wake up one frame per iteration from both ends.  Person 1 IS seen in
    them.  The unmodified reference re-run with 3 threads or with its initial cameras x (1 + 1e-7 / 1e-6 U) ends 57 - 124 px from its own
    result there after five iterations per stage -- in 4 to 10 of the twenty frames -- and within 1 px everywhere else (oracle/make_golden.py
    gen_grecon_latent_gapfamily).  Round 4 replaced this input by one without the gap; this case keeps it and separates the two regimes:
    OUTSIDE the zero-camera frames the device is held to the usual bounds of the mode (0.1 px), INSIDE them to the scale of the reference's
    own re-runs (twice the farthest one); the motion latent, which feels those frames through person 1's reprojection, to five times the
    re-runs' spread (they differ in up to 10 of the 20 frames, the device -- whose rounding differs everywhere, not in the seventh digit of
    the non-zero cameras only -- in all 20)."""
    cfg_id, T, P, K = 'glamr_dynamic_multi', 90, 2, 5
    g = golden('grecon_latent_%s_T%d_P%d_gapfamily' % (cfg_id, T, P))
    members = sorted({k[4:].split('_p0_')[0] for k in g if k.startswith('fam_') and '_p0_kp_2d_pred' in k})
    assert len(members) >= 4
    in_dict = synth.make_in_dict(seed=3, num_frames=T, num_persons=P, smpl_model=synth.make_smpl_model())
    lat = mg.latents_for(in_dict, 3)
    model = make_model(cfg_id, flag_opt_motion_latent=True, flag_opt_traj_latent=True)
    out = model.optimize(in_dict, latents=lat, max_iters=K)
    zero_cam = ~g['p0_vis_frames']                                    # frames whose initial camera is a zero matrix
    assert zero_cam.sum() == 20
    rng = lambda m: (lambda idx: '%d..%d (%d)' % (idx[0], idx[-1], len(idx)) if len(idx) else '-')(np.nonzero(m)[0])
    for pi in range(P):
        vis = g['p%d_vis_frames' % pi]
        ref_kp, ref_lat = g['p%d_kp_2d_pred' % pi], g['p%d_motion_latent' % pi]
        per_frame = lambda kp: np.abs(np.asarray(kp, np.float64) - ref_kp).max(axis=(1, 2))
        fam = np.stack([per_frame(g['fam_%s_p%d_kp_2d_pred' % (m, pi)]) for m in members])          # (members, frames)
        fam_lat = max(float(np.abs(g['fam_%s_p%d_motion_latent' % (m, pi)] - ref_lat).max()) for m in members)
        pd = out['person_data'][pi]
        d = per_frame(pd['kp_2d_pred'])
        e_lat = float(np.abs(pd['motion_latent'] - ref_lat).max())
        calm, wild = vis & ~zero_cam, vis & zero_cam
        print('latent mode with gaps, person %d: outside the zero-camera frames %.4f px from the reference (its re-runs: %s px); inside them %.1f px in %s '
              '(re-runs: %s); motion latent %.2e (re-runs up to %.2e)'
              % (pi, d[calm].max(), ', '.join('%.3f' % f[calm].max() for f in fam), d[wild].max() if wild.any() else 0.0, rng(wild & (d > 1)),
                 ' | '.join('%.0f px in %s' % (f[wild].max() if wild.any() else 0.0, rng(wild & (f > 1))) for f in fam), e_lat, fam_lat))
        assert fam[:, calm].max() < 1.0                               # the fixture itself: the reference agrees with its re-runs outside those frames
        assert d[calm].max() < 0.1
        if wild.any():
            assert d[wild].max() <= 2.0 * fam[:, wild].max() + 0.5
        assert e_lat <= 5.0 * fam_lat + 1e-5


def test_latent_mode_takes_a_batch_of_sequences(make_model):
    """The reference runs the latent-optimisation mode one sequence at a time (run_dataset.py:67-105); the schedule here takes a batch of scenes
    (VERDICT r4 "latent-optimisation mode for a batch"): three sequences of different lengths and person counts in ONE optimize_batch call give what
    three optimize() calls give, per sequence (same kernels at these sizes: to rounding; the latents must have moved)."""
    from tests.test_e2e_gpu import _trim_person
    md = synth.make_smpl_model()
    in_dicts = [synth.make_in_dict(seed=12, num_frames=100, num_persons=1, smpl_model=md),
                _trim_person(synth.make_in_dict(seed=13, num_frames=80, num_persons=2, smpl_model=md), 1, 9, 71),
                synth.make_in_dict(seed=14, num_frames=60, num_persons=1, smpl_model=md, gap=(0, 0))]
    lats = [mg.latents_for(d, s) for d, s in zip(in_dicts, (12, 13, 14))]
    K = 5
    model = make_model('glamr_dynamic_multi', flag_opt_motion_latent=True, flag_opt_traj_latent=True)
    outs = model.optimize_batch(in_dicts, lats, K)
    assert model.latent_graph_replays > 0
    for d, lat, ob in zip(in_dicts, lats, outs):
        single = make_model('glamr_dynamic_multi', flag_opt_motion_latent=True, flag_opt_traj_latent=True)
        o1 = single.optimize(d, latents=lat, max_iters=K)
        for idx in o1['person_data']:
            a, b = o1['person_data'][idx], ob['person_data'][idx]
            assert np.abs(a['motion_latent'] - b['motion_latent']).max() < 2e-5
            assert np.abs(a['smpl_pose'] - b['smpl_pose']).max() < 1e-5
            vis = np.asarray(a['vis_frames'])
            assert np.abs(np.asarray(a['kp_2d_pred']) - np.asarray(b['kp_2d_pred']))[vis].max() < 0.05
            assert np.abs(a['motion_latent'] - lat[idx]['motion']).max() > 1e-5
