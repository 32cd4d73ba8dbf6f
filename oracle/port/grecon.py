"""TEST INFRASTRUCTURE ONLY -- CPU (torch autograd + torch.optim.Adam) restatement of GLAMR's global optimiser:
/root/reference/global_recon/models/global_recon_model.py (driver) and global_recon/models/loss_func.py (residuals), for
the branches reachable from the shipped configs (global_recon/cfg/*.yml; est_type 'hybrik', scalar heading, no latent
optimisation, no penetration loss).  Like the reference it evaluates the FULL SMPL skinning inside every iteration --
that is what makes it the `cpu_baseline` ("port") in bench.py.

Pinned against the unmodified reference executed under oracle/ref_harness.py (tests/test_oracle_vs_reference.py and the
fixtures in tests/golden/ produced by oracle/make_golden.py).
"""
import numpy as np
import torch
from scipy.interpolate import interp1d
from scipy.spatial.transform import Rotation
from . import transforms as tf

# (body26fk index, smpl index) pairs whose joint names coincide (lib/utils/joints.py:48-73,619-641 via
# global_recon_model.py:82-85): the only 2-D keypoints that receive confidence 1.
SMPL_TO_BODY26FK = np.array([[8, 8], [5, 5], [2, 2], [21, 17], [23, 19], [25, 21], [7, 7], [4, 4], [1, 1],
                             [20, 16], [22, 18], [24, 20], [6, 12], [0, 0]])


def to_torch(x, device):
    """lib/utils/torch_utils.py:101-116 tensor_to"""
    if isinstance(x, torch.Tensor):
        return x.to(device)
    if isinstance(x, np.ndarray):
        return torch.tensor(x).to(device)
    if isinstance(x, list):
        return [to_torch(v, device) for v in x]
    if isinstance(x, dict):
        return {k: to_torch(v, device) for k, v in x.items()}
    return x


def to_numpy(x):
    """lib/utils/torch_utils.py:119-127"""
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    if isinstance(x, list):
        return [to_numpy(v) for v in x]
    if isinstance(x, dict):
        return {k: to_numpy(v) for k, v in x.items()}
    return x


# =====================================================================================================================
# residual terms (global_recon/models/loss_func.py)
# =====================================================================================================================

def _gmof(x, sigma):
    """:6-12"""
    x2, s2 = x ** 2, sigma ** 2
    return (s2 * x2) / (s2 + x2)


def loss_kp_2d(data, specs):
    """:15-36.  NB: kp_2d_aligned / kp_2d_score are float64 (np.zeros default at global_recon_model.py:119), so this term
    is accumulated in double precision; with first_frame_only the (1,26,2) residual broadcasts against all visible scores."""
    total, n = 0, 0
    min_conf = specs.get('min_conf', 0.05)
    for pd in data['person_data'].values():
        vis = pd['vis_frames']
        diff = pd['kp_2d_pred'][vis] - pd['kp_2d_aligned'][vis]
        score = pd['kp_2d_score'][vis].clone()
        score[score < min_conf] = 0
        loss = _gmof(diff, sigma=100)
        if specs.get('first_frame_only', False):
            loss = loss[[0]]
        n += vis.sum()
        loss[:10] *= specs.get('first_frame_weight', 1.0)
        total += (loss.sum(-1) * (score ** 2)).sum()
    return total / n


def loss_kp_2d_dist(data, specs):
    """:39-57 (monitor only)"""
    out = []
    min_conf = specs.get('min_conf', 0.05)
    for pd in data['person_data'].values():
        s, p, a = pd['kp_2d_score'], pd['kp_2d_pred'], pd['kp_2d_aligned']
        if specs.get('first_frame_only', False):
            s, p, a = s[[0]], p[[0]], a[[0]]
        out.append((p - a).pow(2).sum(-1).sqrt().view(-1)[(s > min_conf).view(-1)])
    return torch.cat(out).mean()


def loss_cam_inv_rot_smoothness(data, specs):
    """:76-81"""
    vel = (data['cam_pose_inv'][:-1, :3, :2] - data['cam_pose_inv'][1:, :3, :2]) * 30
    return vel.pow(2).sum(-1).sum(-1).mean()


def loss_cam_origin_smoothness(data, specs):
    """:84-91"""
    vel = (data['cam_pose_inv'][1:, :3, 3] - data['cam_pose_inv'][:-1, :3, 3]) * 30
    return vel.pow(2).sum(-1).mean()


def loss_cam_up_reg(data, specs):
    """:106-114 -- LINEAR in the (2,1) entry of the camera-to-world rotation."""
    v = data['cam_pose_inv'][:, 2, 1].clone()
    v[:10] *= specs.get('first_frame_weight', 1.0)
    if specs.get('first_frame_only', False):
        v = v[[0]]
    return v.mean()


def loss_traj_rot_smoothness(data, specs):
    """:117-132 (rot_type '6d')"""
    total, n = 0, 0
    for pd in data['person_data'].values():
        n += pd['smpl_orient_world'].shape[0] - 1
        d6 = tf.aa_to_6d(pd['smpl_orient_world'])
        total += ((d6[1:] - d6[:-1]) * 30).pow(2).sum()
    return total / n


def loss_cam_traj_rot(data, specs):
    """:147-172 (rot_type '6d')"""
    total, n = 0, 0
    for pd in data['person_data'].values():
        vis = pd['vis_frames']
        diff = tf.aa_to_6d(pd['smpl_orient_cam'][vis]) - tf.aa_to_6d(pd['smpl_orient_cam_in_world'][vis])
        if specs.get('first_frame_only', False):
            diff = diff[[0]]
            n += 1
        else:
            diff[0] *= specs.get('first_frame_weight', 1.0)
            n += vis.sum()
        total += diff.pow(2).sum()
    return total / n


def _reg(data, key):
    """:189-196"""
    total, n = 0, 0
    for pd in data['person_data'].values():
        n += pd[key].shape[0]
        total += (pd[key] * 30).pow(2).sum()
    return total / n


def loss_dheading_reg_new(data, specs):
    """:220-230"""
    total, n = 0, 0
    for pd in data['person_data'].values():
        v = pd['traj_local_dheading']
        n += v.shape[0]
        diff = tf.heading_to_vec(v) - torch.tensor([1.0, 0.0]).type_as(v)
        total += (diff * 30).pow(2).sum()
    return total / n


def loss_rel_transform(data, specs):
    """:248-271"""
    total, n = 0, 0
    pdata = data['person_data']
    tw = specs.get('trans_weight', 1.0)
    ffw = specs.get('first_frame_weight', 10)
    for (i, j), rel_cam in data['rel_transform_cam'].items():
        n += rel_cam.shape[0]
        vis = pdata[i]['vis_frames'] & pdata[j]['vis_frames']
        if sum(vis) == 0:
            continue
        rel_w = torch.matmul(tf.invert_transform(pdata[i]['person_transform_world'][vis]), pdata[j]['person_transform_world'][vis])
        rel_cam = rel_cam[vis]
        d_rot = rel_cam[..., :3, :2] - rel_w[..., :3, :2]
        d_trans = rel_cam[..., :3, 3] - rel_w[..., :3, 3]
        d_rot[0] *= ffw
        d_trans[0] *= ffw
        if specs.get('first_frame_trans_only', False):
            d_trans[1:] = 0.0
        total += d_rot.pow(2).sum() + d_trans.pow(2).sum() * tw
    return total / n if n > 0 else total


LOSSES = {
    'kp_2d': loss_kp_2d, 'kp_2d_dist': loss_kp_2d_dist,
    'cam_inv_rot_smoothness': loss_cam_inv_rot_smoothness, 'cam_origin_smoothness': loss_cam_origin_smoothness,
    'cam_up_reg': loss_cam_up_reg, 'traj_rot_smoothness': loss_traj_rot_smoothness, 'cam_traj_rot': loss_cam_traj_rot,
    'local_traj_dxy_reg': lambda d, s: _reg(d, 'traj_local_dxy'), 'local_traj_rot_reg': lambda d, s: _reg(d, 'traj_local_rot'),
    'local_traj_z_reg': lambda d, s: _reg(d, 'traj_local_z'), 'local_traj_dheading_reg_new': loss_dheading_reg_new,
    'cam_inv_trans_residual_reg': lambda d, s: (d['cam_inv_trans_residual'] * 30).pow(2).sum() / d['cam_inv_trans_residual'].shape[0],
    'rel_transform': loss_rel_transform,
}


# =====================================================================================================================
# driver (global_recon/models/global_recon_model.py)
# =====================================================================================================================

class GlobalReconOptimizer:

    def __init__(self, cfg_dict, smpl, mt_model, device=torch.device('cpu'), log_fn=None):
        """cfg_dict: the parsed YAML of global_recon/cfg/<id>.yml.  :25-67"""
        self.specs = specs = cfg_dict['grecon_model_specs']
        self.opt_stage_specs = cfg_dict['opt_stage_specs']
        self.device, self.smpl, self.mt_model, self.log_fn = device, smpl, mt_model, log_fn
        g = specs.get
        self.flag_fixed_cam = g('flag_fixed_cam', False)
        self.flag_opt_cam = g('flag_opt_cam', True)
        self.flag_opt_traj = g('flag_opt_traj', True)
        self.flag_filter_pose = g('flag_filter_pose', True)
        self.flag_cam_from_person = g('flag_opt_cam_from_person_pose', False)
        self.flag_init_cam_all_frames = g('flag_init_cam_all_frames', False)
        self.flag_cam_inv_trans_res_all = g('flag_cam_inv_trans_res_all', True)
        self.cam_fix_frames = g('cam_fix_frames', [[0, None]])
        assert g('flag_infer_motion_traj', False) and g('flag_pred_traj', True) and g('est_type', 'hybrik') == 'hybrik'
        self.cur_iter = 0
        self.last_loss_dict = None

    # ---- init_data :76-248 -------------------------------------------------------------------------------------------
    def init_data(self, in_dict, latents=None):
        """`latents`: optional {person idx: {'motion': (n_windows,128), 'traj': (1,128)}} replacing the torch.randn draws of the
        two priors (the reference takes them through in_motion_latent / in_traj_latent, global_recon_model.py:364-367)."""
        dev = self.device
        num_fr = len(in_dict['est'][0]['bboxes_dict']['exist'])
        cam_pose = torch.eye(4).repeat((num_fr, 1, 1)).float().to(dev)
        cam_pose_inv = tf.invert_transform(cam_pose)
        person_data = {}
        for idx, src in in_dict['est'].items():
            d = {}
            d['visible'] = visible = src['bboxes_dict']['exist'].copy()
            d['visible_orig'] = visible.copy()
            d['fr_start'] = start = np.where(visible)[0][0]
            d['fr_end'] = end = np.where(visible)[0][-1] + 1
            d['exist_frames'] = visible == 1
            d['exist_frames'][start:end] = True
            d['exist_len'] = end - start
            d['max_len'] = max_len = visible.shape[0]
            d['frames'] = np.arange(max_len)
            d['vis_frames'] = vis_frames = visible == 1
            d['invis_frames'] = visible == 0
            d['frame2ind'] = {f: i for i, f in enumerate(d['frames'])}
            d['scale'] = None
            rm = src['smpl_pose_quat_wroot']
            nvis = rm.shape[0]
            aa = Rotation.from_matrix(rm.reshape((-1, 3, 3))).as_rotvec().reshape((nvis, -1, 3)).astype(np.float32)   # :105-108
            d['smpl_pose'] = aa[:, 1:].reshape(-1, 69)
            d['smpl_beta'] = src['smpl_beta']
            d['smpl_orient_cam'] = aa[:, 0]
            d['root_trans_cam'] = src['root_trans']
            kp = np.concatenate((src['kp_2d'][:, :24], np.ones_like(src['kp_2d'][:, :24, [0]])), axis=-1)
            kps = np.zeros((sum(vis_frames), 26, 3))                                                                  # float64
            kps[:, SMPL_TO_BODY26FK[:, 0]] = kp[:, SMPL_TO_BODY26FK[:, 1]]
            d['kp_2d'], d['kp_2d_score'] = kps[:, :, :2], kps[:, :, 2]
            d['kp_2d_aligned'] = d['kp_2d'].copy()
            d['cam_K'] = src['cam_K'].astype(np.float32)
            if not np.all(visible):                                                                                   # :127-136
                for key in ('kp_2d', 'kp_2d_score', 'kp_2d_aligned', 'cam_K'):
                    full = np.zeros((max_len,) + d[key].shape[1:], dtype=d[key].dtype)
                    full[vis_frames] = d[key]
                    d[key] = full
                vis_ind = np.where(visible)[0].astype(np.float32)
                for key in ('smpl_pose', 'smpl_beta', 'root_trans_cam', 'smpl_orient_cam'):
                    f = interp1d(vis_ind, d[key], axis=0, assume_sorted=True, fill_value='extrapolate')
                    d[key] = f(np.arange(max_len, dtype=np.float32))
            d = to_torch(d, dev)
            if self.flag_filter_pose:
                self.filter_pose(d)
            d['root_trans_world'] = tf.apply_transform(cam_pose_inv, d['root_trans_cam'])
            d['smpl_orient_world'] = tf.rotate_aa(cam_pose_inv, d['smpl_orient_cam'])
            d['root_trans_world_base'] = d['root_trans_world'].clone()
            d['smpl_orient_world_base'] = d['smpl_orient_world'].clone()
            d['smpl_pose_nofill'] = d['smpl_pose'].clone()
            d['smpl_pose_nofill'][~d['exist_frames']] = 0.0
            if latents is not None:
                d['in_motion_latent'] = torch.as_tensor(latents[idx]['motion'], device=dev)
                d['in_traj_latent'] = torch.as_tensor(latents[idx]['traj'], device=dev)
            person_data[idx] = d

        for d in person_data.values():
            self.infer_motion_traj(d)
        for d in person_data.values():                                                                               # :166-169
            d['person_transform_world'] = tf.make_transform(d['smpl_orient_world'], d['root_trans_world'], 'axis_angle')
            d['person_transform_cam'] = tf.make_transform(d['smpl_orient_cam'], d['root_trans_cam'], 'axis_angle')
            d['person2cam'] = tf.invert_transform(d['person_transform_cam'])

        rel = None
        if self.flag_opt_traj:
            last = d
            for d in person_data.values():
                d['smpl_orient_world_res'] = torch.zeros_like(last['smpl_orient_world'])
                d['root_trans_world_res'] = torch.zeros_like(last['root_trans_world'])
            rel = {}
            ids = list(person_data.keys())
            for i in range(len(ids)):
                for j in range(len(ids)):
                    if i != j:
                        rel[(i, j)] = torch.matmul(tf.invert_transform(person_data[ids[i]]['person_transform_cam']),
                                                   person_data[ids[j]]['person_transform_cam'])
            for d in person_data.values():                                                                           # :186-199
                n = int(d['exist_len'])
                d['traj_local_xy'] = torch.zeros((2,), device=dev)
                d['traj_local_dxy'] = torch.zeros((n - 1, 2), device=dev)
                d['traj_local_heading'] = torch.zeros((1,), device=dev)
                d['traj_local_dheading'] = torch.zeros((n - 1,), device=dev)
                d['traj_local_z'] = torch.zeros((n,), device=dev)
                d['traj_local_rot'] = torch.zeros((n, 6), device=dev)

        fr_num_persons = sum([d['vis_frames'] for d in person_data.values()])
        n_empty = (fr_num_persons == 0).sum()
        data = {
            'seq_name': in_dict['seq_name'], 'person_data': person_data, 'seq_len': cam_pose.shape[0],
            'fr_num_persons': fr_num_persons, 'cam_pose': cam_pose, 'cam_pose_inv': cam_pose_inv,
            'cam_inv_rot_residual': torch.zeros((n_empty, 6)).type_as(cam_pose),
            'cam_inv_trans_residual': torch.zeros((cam_pose.shape[0] if self.flag_cam_inv_trans_res_all else n_empty, 3)).type_as(cam_pose),
            'rel_transform_cam': rel, 'gt': in_dict['gt'], 'gt_meta': in_dict['gt_meta'],
            'meta': {'algo': 'global_recon', 'num_fr': num_fr},
        }
        self.init_cam_pose(data)
        self.init_traj_heading_from_cam(person_data, data)
        if self.flag_init_cam_all_frames:
            self.init_cam_pose(data, all_frames=True)
        self.forward(data, [], {'stage': 'init'})
        return data

    def filter_pose(self, d):
        """:250-271 -- sequential, data-dependent: drop one of two frames whose root orientation jumps by > 60 deg."""
        visible = d['visible']
        quat = tf.aa_to_quat(d['smpl_orient_cam'])
        jump = tf.quat_angle_between(quat[1:], quat[:-1])
        ind = torch.where((jump > np.pi / 3) & visible[1:].bool())[0] + 1
        for i in ind:
            if visible[i - 1]:
                if i + 1 < quat.shape[0] and visible[i + 1] and i + 1 not in ind:
                    visible[i - 1] = 0
                else:
                    visible[i] = 0
        d['vis_frames'] = visible == 1
        d['invis_frames'] = visible == 0

    def infer_motion_traj(self, d):
        """:353-392"""
        ex = d['exist_frames']
        batch = {'in_body_pose': d['smpl_pose_nofill'][ex].unsqueeze(0).clone(), 'frame_mask': d['visible'][ex].unsqueeze(0).clone()}
        for key in ('in_motion_latent', 'in_traj_latent'):          # latents supplied by the caller for RNG-free parity
            if key in d:
                batch[key] = d[key]
        out = self.mt_model.inference(batch, sample_num=1)
        d['infilled'] = True
        d['smpl_pose'] = d['smpl_pose'].detach().clone()
        d['smpl_pose'][ex] = out['infer_out_body_pose'][0, 0]
        d['traj_predicted'] = True
        d['traj_local_pred'] = out['infer_out_local_traj_tp'][:, 0, 0, :].clone()
        d['smpl_orient_world_base'] = d['smpl_orient_world_base'].detach().clone()
        d['root_trans_world_base'] = d['root_trans_world_base'].detach().clone()
        if 'infer_out_pose' in out:
            d['smpl_orient_world_base'][ex] = out['infer_out_pose'][0, 0, :, :3]
        d['smpl_orient_world_base'][ex] = out['infer_out_orient'][0, 0]
        d['root_trans_world_base'][ex] = out['infer_out_trans'][0, 0]
        d['smpl_orient_world'] = d['smpl_orient_world_base']
        d['root_trans_world'] = d['root_trans_world_base']

    def init_cam_pose(self, data, all_frames=False):
        """:294-317 -- camera-to-world from the FIRST person's world pose composed with its person->camera transform."""
        first = next(iter(data['person_data'].values()))
        cand = torch.matmul(first['person_transform_world'], first['person2cam']) * first['vis_frames'][:, None, None]
        ind = data['fr_num_persons'] > 0
        start = torch.where(ind)[0][0]
        inf = torch.zeros_like(data['cam_pose'])
        inf[ind] = cand[ind]
        if all_frames:
            if not torch.all(ind):
                last = inf[start]
                for i in range(len(ind)):
                    if not ind[i]:
                        data['cam_pose_inv'][i] = last
                    else:
                        last = data['cam_pose_inv'][i]
        else:
            inf[...] = inf[[start]].clone()
        inf[:, :3, :3] = tf.sixd_to_rotmat(tf.rotmat_to_6d(inf[:, :3, :3]))
        data['pose_infer_cam_pose_inv'] = inf
        data['cam_pose_inv'] = inf
        data['cam_pose'] = tf.invert_transform(inf)

    def init_traj_heading_from_cam(self, person_data, data):
        """:273-292"""
        for d in person_data.values():
            w = torch.matmul(data['cam_pose_inv'], d['person_transform_cam'])
            q = tf.rotmat_to_quat(w[:, :3, :3].contiguous())
            qi = tf.interp_orient_sep_heading(q[d['vis_frames']], d['vis_frames'])
            local = tf.global_to_local_traj(w[:, :3, 3], qi)
            for (s, e) in self.cam_fix_frames:
                d['traj_local_pred'][s:e, -2:] = local[d['exist_frames']][s:e, -2:]
            trans, q = tf.local_to_global_traj(d['traj_local_pred'])
            ex = d['exist_frames']
            d['smpl_orient_world_base'] = d['smpl_orient_world_base'].detach().clone()
            d['root_trans_world_base'] = d['root_trans_world_base'].detach().clone()
            d['smpl_orient_world_base'][ex] = tf.quat_to_aa(q)
            d['root_trans_world_base'][ex] = trans
            d['smpl_orient_world'] = d['smpl_orient_world_base'].clone()
            d['root_trans_world'] = d['root_trans_world_base'].clone()
            d['person_transform_world'] = tf.make_transform(d['smpl_orient_world'], d['root_trans_world'], 'axis_angle')

    # ---- forward :394-531 --------------------------------------------------------------------------------------------
    def pred_trajectory_base(self, d):
        """:394-426"""
        L = d['traj_local_pred'].detach().clone()
        L[0, :2] += d['traj_local_xy']
        L[1:, :2] += d['traj_local_dxy']
        mask = torch.ones_like(L[1:, 0])
        for (s, e) in self.cam_fix_frames:
            mask[s:e] = 0.0
        h0 = tf.vec_to_heading(L[[0], -2:].clone()) + d['traj_local_heading']
        L[0, -2:] = tf.heading_to_vec(h0).squeeze(0)
        h = tf.vec_to_heading(L[1:, -2:].clone()) + d['traj_local_dheading'] * mask
        L[1:, -2:] = tf.heading_to_vec(h)
        L[:, 2] += d['traj_local_z']
        L[:, 3:-2] += d['traj_local_rot']
        d['traj_local'] = L
        trans, q = tf.local_to_global_traj(L)
        ex = d['exist_frames']
        d['smpl_orient_world_base'] = d['smpl_orient_world_base'].detach().clone()
        d['root_trans_world_base'] = d['root_trans_world_base'].detach().clone()
        d['smpl_orient_world_base'][ex] = tf.quat_to_aa(q)
        d['root_trans_world_base'][ex] = trans

    def forward(self, data, opt_variables, opt_meta):
        for d in data['person_data'].values():
            self.pred_trajectory_base(d)
            if self.flag_opt_traj:
                d['smpl_orient_world'] = d['smpl_orient_world_base']
                d['root_trans_world'] = d['root_trans_world_base']
                if 'world_dheading' in d:                                                                            # :459-465
                    w = d['world_dheading']
                    wq = tf.aa_to_quat(torch.cat((torch.zeros([w.shape[0], 2], device=self.device), w), dim=-1))
                    d['smpl_orient_world'] = tf.quat_to_aa(tf.quat_mul(wq, tf.aa_to_quat(d['smpl_orient_world_base'])))
                    d['root_trans_world'] = d['root_trans_world_base']
            d['person_transform_world'] = tf.make_transform(d['smpl_orient_world'], d['root_trans_world'], 'axis_angle')

        if self.flag_opt_cam and opt_meta['stage'] != 'init':                                                      # :473-508
            if 'cam' in opt_variables:
                if self.flag_fixed_cam:
                    data['cam_rot_6d'] = data['cam_rot_6d_fix'].expand(data['cam_pose'].shape[0], -1)
                    data['cam_trans'] = data['cam_trans_fix'].expand(data['cam_pose'].shape[0], -1)
                if 'cam_rot_6d' in data:
                    data['cam_pose'] = tf.make_transform(data['cam_rot_6d'], data['cam_trans'], '6d')
                    data['cam_pose_inv'] = tf.invert_transform(data['cam_pose'])
            elif self.flag_cam_from_person:
                per = [torch.matmul(d['person_transform_world'], d['person2cam']) * d['vis_frames'][:, None, None]
                       for d in data['person_data'].values()]
                npers = data['fr_num_persons']
                ind = npers > 0
                inv = torch.zeros_like(data['cam_pose'])
                inv[ind] = sum(per)[ind] / npers[ind, None, None]
                last = inv[torch.where(ind)[0][0]]
                for i in range(len(npers)):
                    if npers[i] == 0:
                        inv[i] = last
                    else:
                        last = inv[i]
                r6 = tf.rotmat_to_6d(inv[:, :3, :3])
                r6[npers == 0] += data['cam_inv_rot_residual']
                inv[:, :3, :3] = tf.sixd_to_rotmat(r6)
                if self.flag_cam_inv_trans_res_all:
                    inv[:, :3, 3] += data['cam_inv_trans_residual']
                else:
                    inv[npers == 0, :3, 3] += data['cam_inv_trans_residual']
                data['cam_pose_inv'] = inv
                data['cam_pose'] = tf.invert_transform(inv)

        for d in data['person_data'].values():                                                                      # :511-528
            d['smpl_orient_cam_in_world'] = tf.rotate_aa(data['cam_pose'], d['smpl_orient_world'])
            d['root_trans_cam_in_world'] = tf.apply_transform(data['cam_pose'], d['root_trans_world'])
            out = self.smpl(global_orient=d['smpl_orient_world'], body_pose=d['smpl_pose'], betas=d['smpl_beta'],
                            root_trans=d['root_trans_world'], root_scale=None, return_full_pose=True)
            d['kp_2d_pred'] = tf.project(tf.apply_transform(data['cam_pose'], out.joints), d['cam_K'])

    # ---- optimisation :533-644 ---------------------------------------------------------------------------------------
    def compute_loss(self, data, loss_cfg):
        total, ld, lud = 0, {}, {}
        for name, spec in loss_cfg.items():
            lud[name] = LOSSES[name](data, spec)
            ld[name] = lud[name] * spec['weight']
            if not spec.get('monitor_only', False):
                total = total + ld[name]
        return total, ld, lud

    def get_parameter(self, data, opt_variables):
        """:591-633"""
        params = []
        if 'cam' not in opt_variables:
            params += [data['cam_inv_rot_residual'], data['cam_inv_trans_residual']]
        else:
            if self.flag_fixed_cam:
                data['cam_rot_6d_fix'] = tf.rotmat_to_6d(data['cam_pose'][[0], :3, :3]).detach()
                data['cam_trans_fix'] = data['cam_pose'][[0], :3, 3].clone().detach()
                params += [data['cam_rot_6d_fix'], data['cam_trans_fix']]
            else:
                data['cam_rot_6d'] = tf.rotmat_to_6d(data['cam_pose'][:, :3, :3]).detach()
                data['cam_trans'] = data['cam_pose'][:, :3, 3].clone().detach()
                params += [data['cam_rot_6d'], data['cam_trans']]
        for d in data['person_data'].values():
            if self.flag_opt_traj:
                for key in opt_variables:
                    if 'local' in key:
                        params.append(d['traj_' + key])
            if 'world_dheading' in opt_variables:
                if 'world_dheading' not in d:
                    d['world_dheading'] = torch.zeros_like(d['smpl_orient_world'][..., [0]])
                params.append(d['world_dheading'])
        return params

    def optimize_main(self, data, opt_variables, opt_lr, opt_niters, loss_cfg, opt_meta):
        """:547-570 -- fresh Adam(betas 0.9/0.999, eps 1e-8) per stage, one closure evaluation per step."""
        params = self.get_parameter(data, opt_variables)
        for p in params:
            p.requires_grad_(True)
        opt = torch.optim.Adam(params, lr=opt_lr, betas=(0.9, 0.999)) if params else None
        holder = {}

        def closure():
            opt.zero_grad()
            self.forward(data, opt_variables, opt_meta)
            loss, _, holder['uw'] = self.compute_loss(data, loss_cfg)
            loss.backward()
            return loss

        for it in range(opt_niters):
            self.cur_iter = it
            if opt is not None:
                opt.step(closure)
            if self.log_fn is not None:
                self.log_fn(opt_meta['stage'], it, {k: float(v) for k, v in holder['uw'].items()})
        self.last_loss_dict = holder.get('uw')
        for p in params:
            p.requires_grad_(False)
        data['cam_pose'] = data['cam_pose'].detach()
        data['cam_pose_inv'] = data['cam_pose_inv'].detach()
        return data

    def optimize(self, in_dict, continue_opt=False, latents=None):
        """:572-589"""
        data = to_torch(in_dict, self.device) if continue_opt else self.init_data(in_dict, latents)
        for stage, spec in self.opt_stage_specs.items():
            self.optimize_main(data, spec['opt_variables'], spec['opt_lr'], spec['opt_niters'], spec['loss_cfg'], {'stage': stage})
            if spec.get('reinitialize_cam', False):
                data['cam_pose'][:] = data['cam_pose'][[0]]
                data['cam_pose_inv'] = tf.invert_transform(data['cam_pose'])
        return to_numpy(data)
