"""Packs GLAMR's per-sequence `data` dictionaries (the structure GlobalReconOptimizer.init_data builds,
global_recon/models/global_recon_model.py:217-230) into the flat, padded arrays of `glamr_scene_batch` (include/glamr_hip.h)
and back.  Pure layout code: no reference arithmetic happens here."""
import ctypes

import numpy as np
import torch

from .. import _lib

NJ = 26
LOSS_IDS = {'kp_2d': 0, 'kp_2d_dist': 1, 'rel_transform': 2, 'cam_traj_rot': 3, 'traj_rot_smoothness': 4, 'local_traj_dxy_reg': 5,
            'local_traj_dheading_reg_new': 6, 'local_traj_rot_reg': 7, 'local_traj_z_reg': 8, 'cam_inv_trans_residual_reg': 9,
            'cam_inv_rot_smoothness': 10, 'cam_origin_smoothness': 11, 'cam_up_reg': 12}
VAR_BITS = {'cam': 1, 'local_xy': 2, 'local_heading': 4, 'world_dheading': 8, 'local_dxy': 16, 'local_rot': 32, 'local_z': 64,
            'local_dheading': 128}
FLAG_FIXED_CAM, FLAG_CAM_FROM_PERSON, FLAG_HAS_WORLD_DHEADING = 1, 2, 4
FLAG_KEEP_CAM_PARAMS, FLAG_NO_CAMERA_TERMS = 8, 16          # launch-by-launch stages (glamr_amd/parallel.py PersonShardedSchedule)
FLAG_POSES_ONLY = 32                                       # a forward-only launch that stops after the world poses
FLAG_KEEP_TABLES = 256                                     # launch-by-launch schedules: same stage, same workspace as the previous launch -- the set-up's tables are kept
FLAG_NO_REPORT = 128                                       # launch-by-launch schedules: the launch's last evaluation writes no outputs / loss values
FLAG_ABSOLUTE_HEADING = 64                                 # specs absolute_heading: per-frame headings are absolute (csrc/grecon_wide.hip instances)


def param_layout_py(max_persons, max_len):
    """Mirror of glamr::grecon::param_layout (glamr_amd/csrc/grecon_algo.hpp); checked against the library in the tests."""
    T = max_len
    l = dict(cam_rot6d=0, cam_trans=6 * T, cam_inv_rot_res=9 * T, cam_inv_trans_res=15 * T, person0=18 * T,
             local_xy=0, local_heading=2, local_dxy=4)
    l['local_dheading'] = l['local_dxy'] + 2 * T
    l['local_z'] = l['local_dheading'] + T
    l['local_rot'] = l['local_z'] + T
    l['world_dheading'] = l['local_rot'] + 6 * T
    l['person_stride'] = l['world_dheading'] + T
    l['scene_stride'] = l['person0'] + max_persons * l['person_stride']
    return l


def stage_desc(stage_specs, model_specs, has_world_dheading=False, niters=None):
    """opt_stage_specs[stage] (+ grecon_model_specs flags) -> glamr_stage_desc.  Unknown loss names raise: nothing is dropped."""
    sd = _lib.StageDesc()
    sd.var_mask = 0
    for v in stage_specs['opt_variables']:
        if v not in VAR_BITS:
            raise NotImplementedError('optimisation variable %r is not supported by the fused optimiser' % v)
        sd.var_mask |= VAR_BITS[v]
    flags = 0
    if model_specs.get('flag_fixed_cam', False):
        flags |= FLAG_FIXED_CAM
    if model_specs.get('flag_opt_cam_from_person_pose', False):
        flags |= FLAG_CAM_FROM_PERSON
    if has_world_dheading:
        flags |= FLAG_HAS_WORLD_DHEADING
    if model_specs.get('absolute_heading', False):
        flags |= FLAG_ABSOLUTE_HEADING
    sd.flags = flags
    sd.niters = stage_specs['opt_niters'] if niters is None else niters
    sd.lr = stage_specs['opt_lr']
    sd.kp_min_conf = 0.05
    sd.rel_trans_weight = 1.0
    for i in range(16):
        sd.first_frame_weight[i] = 1.0
        sd.loss_weight[i] = 0.0
    sd.first_frame_weight[LOSS_IDS['rel_transform']] = 10.0
    sd.loss_mask = sd.monitor_mask = sd.first_frame_only_mask = 0
    for name, spec in stage_specs['loss_cfg'].items():
        if name not in LOSS_IDS:
            raise NotImplementedError('loss %r is not supported by the fused optimiser' % name)
        i = LOSS_IDS[name]
        sd.loss_mask |= 1 << i
        sd.loss_weight[i] = spec['weight']
        if spec.get('monitor_only', False):
            sd.monitor_mask |= 1 << i
        if spec.get('first_frame_only', False) and name != 'rel_transform':     # rel_transform_loss never reads it
            sd.first_frame_only_mask |= 1 << i
        if 'first_frame_weight' in spec:
            sd.first_frame_weight[i] = spec['first_frame_weight']
        if name in ('kp_2d', 'kp_2d_dist') and 'min_conf' in spec:
            sd.kp_min_conf = spec['min_conf']
        if name == 'rel_transform':
            sd.rel_trans_weight = spec.get('trans_weight', 1.0)
        if spec.get('rot_type', '6d') != '6d' or spec.get('first_frame_trans_only', False):
            raise NotImplementedError('loss option not supported: %s %s' % (name, spec))
    return sd


def carve_zeros(spec, device):
    """{name: zero tensor} for [(name, dtype, shape)] of 4-byte dtypes, all views of ONE zero-filled device allocation (64-float aligned)."""
    import math
    offs, total = [], 0
    for _, _, shape in spec:
        offs.append(total)
        total += (math.prod(shape) + 63) // 64 * 64
    slab = torch.zeros(total, dtype=torch.float32, device=device)
    out = {}
    for (name, dtype, shape), o in zip(spec, offs):
        v = slab[o:o + math.prod(shape)]
        out[name] = (v if dtype == torch.float32 else v.view(dtype)).view(shape)
    return out


class PackedScenes:
    """Flat tensors of a batch of scenes on `device` + the ctypes struct that points at them."""

    INT_FIELDS = ('n_persons', 'seq_len', 'fr_start', 'fr_end')

    @classmethod
    def empty(cls, n_scenes, max_persons, max_len, device, with_rel=None):
        """All arrays allocated (zero) directly on `device`; the init kernels fill them (glamr_init_prepare / glamr_init_scenes)."""
        self = cls.__new__(cls)
        S, P, T = n_scenes, max_persons, max_len
        self.device, self.S, self.P, self.T = device, S, P, T
        self.layout = param_layout_py(P, T)
        # ONE zero allocation carved into the arrays (22 allocations + 22 fill launches otherwise: host time on every batch, and on hosts
        # with slow driver calls the launches of a step are paced by it)
        spec = [('n_persons', torch.int32, (S,)), ('seq_len', torch.int32, (S,)), ('fr_start', torch.int32, (S * P,)), ('fr_end', torch.int32, (S * P,)),
                ('vis', torch.float32, (S * P, T)), ('kp_2d', torch.float32, (S * P, T, NJ, 2)), ('kp_score', torch.float32, (S * P, T, NJ)),
                ('cam_K', torch.float32, (S * P, T, 9)), ('traj_local_pred', torch.float32, (S * P, T, 11)), ('orient_cam', torch.float32, (S * P, T, 3)),
                ('base_orient', torch.float32, (S * P, T, 3)), ('base_trans', torch.float32, (S * P, T, 3)), ('person2cam', torch.float32, (S * P, T, 12)),
                ('cam_pose', torch.float32, (S, T, 12)), ('params', torch.float32, (S, self.layout['scene_stride'])), ('losses', torch.float32, (S, _lib.NUM_LOSSES)),
                ('orient_world', torch.float32, (S * P, T, 3)), ('trans_world', torch.float32, (S * P, T, 3)), ('kp_2d_pred', torch.float32, (S * P, T, NJ, 2)),
                ('orient_cam_in_world', torch.float32, (S * P, T, 3))]
        if P > 1 if with_rel is None else with_rel:
            spec.append(('rel_transform_cam', torch.float32, (S, P, P, T, 12)))
        self.t = carve_zeros(spec, device)
        self.t['j_local'] = None
        self.person_ids = None
        self.has_world_dheading = False
        return self

    def __init__(self, datas, j_locals, device, cam_fix_frames=((0, None),)):
        """datas: list of `data` dicts (one per sequence); j_locals: list (per scene) of dict idx -> (T,26,3) joints computed
        with zero root orientation and zero root translation."""
        S = len(datas)
        self.device = device
        self.person_ids = [list(d['person_data'].keys()) for d in datas]
        P = max(len(ids) for ids in self.person_ids)
        T = max(int(d['seq_len']) for d in datas)
        if P > 32:
            raise NotImplementedError('at most 32 persons per scene (csrc/grecon_wide.hip)')
        self.S, self.P, self.T = S, P, T
        self.layout = param_layout_py(P, T)
        f32 = lambda *shape: torch.zeros(shape, dtype=torch.float32)
        i32 = lambda *shape: torch.zeros(shape, dtype=torch.int32)
        t = dict(n_persons=i32(S), seq_len=i32(S), fr_start=i32(S * P), fr_end=i32(S * P), vis=f32(S * P, T), j_local=f32(S * P, T, NJ, 3),
                 kp_2d=f32(S * P, T, NJ, 2), kp_score=f32(S * P, T, NJ), cam_K=f32(S * P, T, 9), traj_local_pred=f32(S * P, T, 11),
                 orient_cam=f32(S * P, T, 3), base_orient=f32(S * P, T, 3), base_trans=f32(S * P, T, 3), person2cam=f32(S * P, T, 12),
                 cam_pose=f32(S, T, 12), params=f32(S, self.layout['scene_stride']), losses=f32(S, _lib.NUM_LOSSES),
                 orient_world=f32(S * P, T, 3), trans_world=f32(S * P, T, 3), kp_2d_pred=f32(S * P, T, NJ, 2),
                 orient_cam_in_world=f32(S * P, T, 3))
        t['fr_end'] += 1
        need_mask = any(not (s == 0 and e is None) for (s, e) in cam_fix_frames)
        if need_mask:
            t['dheading_mask'] = f32(S * P, T)
        if P > 1:
            t['rel_transform_cam'] = f32(S, P, P, T, 12)
        cpu = lambda x: torch.as_tensor(x).detach().to('cpu')
        l = self.layout
        for si, d in enumerate(datas):
            ids = self.person_ids[si]
            Ts = int(d['seq_len'])
            t['n_persons'][si], t['seq_len'][si] = len(ids), Ts
            t['cam_pose'][si, :Ts] = cpu(d['cam_pose'])[:, :3, :].reshape(Ts, 12).float()
            prm = t['params'][si]
            npers = cpu(d['fr_num_persons'])
            empty = torch.where(npers == 0)[0]
            if len(empty) and 'cam_inv_rot_residual' in d:
                prm[l['cam_inv_rot_res']:l['cam_inv_rot_res'] + 6 * T].view(T, 6)[empty] = cpu(d['cam_inv_rot_residual']).float()
            res = cpu(d['cam_inv_trans_residual']).float()
            tr = prm[l['cam_inv_trans_res']:l['cam_inv_trans_res'] + 3 * T].view(T, 3)
            if res.shape[0] == Ts:
                tr[:Ts] = res
            elif len(empty):
                raise NotImplementedError('flag_cam_inv_trans_res_all=False is not supported')
            for pi, idx in enumerate(ids):
                pd = d['person_data'][idx]
                slot = si * P + pi
                fs, fe = int(pd['fr_start']), int(pd['fr_end'])
                n = fe - fs
                t['fr_start'][slot], t['fr_end'][slot] = fs, fe
                t['vis'][slot, :Ts] = cpu(pd['vis_frames']).float()
                t['kp_2d'][slot, :Ts] = cpu(pd['kp_2d_aligned']).float()
                t['kp_score'][slot, :Ts] = cpu(pd['kp_2d_score']).float()
                t['cam_K'][slot, :Ts] = cpu(pd['cam_K']).reshape(Ts, 9).float()
                t['traj_local_pred'][slot, :n] = cpu(pd['traj_local_pred']).float()
                t['orient_cam'][slot, :Ts] = cpu(pd['smpl_orient_cam']).float()
                t['base_orient'][slot, :Ts] = cpu(pd['smpl_orient_world_base']).float()
                t['base_trans'][slot, :Ts] = cpu(pd['root_trans_world_base']).float()
                t['person2cam'][slot, :Ts] = cpu(pd['person2cam'])[:, :3, :].reshape(Ts, 12).float()
                if need_mask:
                    m = torch.ones(n - 1)
                    for (s, e) in cam_fix_frames:
                        m[s:e] = 0.0
                    t['dheading_mask'][slot, 1:n] = m
                pp = prm[l['person0'] + pi * l['person_stride']:l['person0'] + (pi + 1) * l['person_stride']]
                pp[l['local_xy']:l['local_xy'] + 2] = cpu(pd['traj_local_xy'])
                pp[l['local_heading']] = cpu(pd['traj_local_heading'])[0]
                pp[l['local_dxy']:l['local_dxy'] + 2 * T].view(T, 2)[1:n] = cpu(pd['traj_local_dxy'])
                pp[l['local_dheading']:l['local_dheading'] + T][1:n] = cpu(pd['traj_local_dheading'])
                pp[l['local_z']:l['local_z'] + T][:n] = cpu(pd['traj_local_z'])
                pp[l['local_rot']:l['local_rot'] + 6 * T].view(T, 6)[:n] = cpu(pd['traj_local_rot'])
                if 'world_dheading' in pd:
                    pp[l['world_dheading']:l['world_dheading'] + T][:Ts] = cpu(pd['world_dheading'])[:, 0]
            if P > 1 and d.get('rel_transform_cam'):
                for (i, j), M in d['rel_transform_cam'].items():
                    t['rel_transform_cam'][si, i, j, :Ts] = cpu(M)[:, :3, :].reshape(Ts, 12).float()
        jl_host = t.pop('j_local')
        self.t = {k: v.contiguous().to(device) for k, v in t.items()}
        first_jl = j_locals[0][self.person_ids[0][0]]
        if torch.is_tensor(first_jl) and first_jl.device == torch.device(device) and first_jl.device.type != 'cpu':
            jl_dev = torch.zeros(jl_host.shape, dtype=torch.float32, device=device)      # device-to-device copies only
            for si, d in enumerate(datas):
                for pi, idx in enumerate(self.person_ids[si]):
                    jl_dev[si * P + pi, :int(d['seq_len'])] = j_locals[si][idx]
            self.t['j_local'] = jl_dev
        else:
            for si, d in enumerate(datas):
                for pi, idx in enumerate(self.person_ids[si]):
                    jl_host[si * P + pi, :int(d['seq_len'])] = cpu(j_locals[si][idx]).float()
            self.t['j_local'] = jl_host.contiguous().to(device)
        self.has_world_dheading = any('world_dheading' in pd for d in datas for pd in d['person_data'].values())
        self.seq_names = [str(d.get('seq_name', 'seq%d' % si)) for si, d in enumerate(datas)]      # (the per-iteration log names its sequence)

    def struct(self):
        sb = _lib.SceneBatch()
        sb.n_scenes, sb.max_persons, sb.max_len, sb.n_joints = self.S, self.P, self.T, NJ
        for name, _ in _lib.SceneBatch._fields_[4:]:
            ten = self.t.get(name)
            setattr(sb, name, ctypes.c_void_p(ten.data_ptr()) if ten is not None else None)
        return sb

    def person_params(self, si, pi):
        l = self.layout
        return self.t['params'][si, l['person0'] + pi * l['person_stride']:l['person0'] + (pi + 1) * l['person_stride']]

    def set_cam_pose(self, cam_poses):
        """cam_poses: list (per scene) of (T,4,4) world->camera arrays."""
        host = torch.zeros((self.S, self.T, 12), dtype=torch.float32)
        for si, c in enumerate(cam_poses):
            c = torch.as_tensor(c).float()
            host[si, :c.shape[0]] = c[:, :3, :].reshape(-1, 12)
        self.t['cam_pose'] = host.to(self.device)

    def fetch(self, names=('params', 'cam_pose', 'orient_world', 'trans_world', 'kp_2d_pred', 'orient_cam_in_world', 'losses')):
        """One device->host copy per tensor."""
        return {k: self.t[k].detach().cpu().numpy() for k in names}

    def unpack_into(self, datas, stage_specs, model_specs, as_torch=True):
        """Writes optimised variables and the outputs of the last forward pass back into the `data` dictionaries, with the
        tensor names the reference uses (global_recon_model.py:396-426,459-480,512-528,598-606)."""
        l, T, P = self.layout, self.T, self.P
        h = self.fetch()
        conv = (lambda a: torch.from_numpy(np.ascontiguousarray(a))) if as_torch else (lambda a: np.ascontiguousarray(a))
        var = set(stage_specs['opt_variables']) if stage_specs is not None else set()
        for si, d in enumerate(datas):
            Ts = int(d['seq_len'])
            cam = np.zeros((Ts, 4, 4), np.float32)
            cam[:, :3, :] = h['cam_pose'][si, :Ts].reshape(Ts, 3, 4)
            cam[:, 3, 3] = 1.0
            inv = np.zeros_like(cam)
            inv[:, :3, :3] = cam[:, :3, :3].transpose(0, 2, 1)
            inv[:, :3, 3] = -np.einsum('tji,tj->ti', cam[:, :3, :3], cam[:, :3, 3])
            inv[:, 3, 3] = 1.0
            d['cam_pose'], d['cam_pose_inv'] = conv(cam), conv(inv)
            prm = h['params'][si]
            if 'cam' in var:
                r6 = prm[l['cam_rot6d']:l['cam_rot6d'] + 6 * T].reshape(T, 6)
                tr = prm[l['cam_trans']:l['cam_trans'] + 3 * T].reshape(T, 3)
                if model_specs.get('flag_fixed_cam', False):
                    d['cam_rot_6d_fix'], d['cam_trans_fix'] = conv(r6[:1]), conv(tr[:1])
                else:
                    d['cam_rot_6d'], d['cam_trans'] = conv(r6[:Ts]), conv(tr[:Ts])
            # the residual tensors live in `data` from init_data on (:171-177) whatever the stages optimise: always written back, so a
            # later optimize(continue_opt=True) restarts from them (the device path's _materialise does the same)
            empty = np.where(np.asarray(d['fr_num_persons']) == 0)[0]
            d['cam_inv_rot_residual'] = conv(prm[l['cam_inv_rot_res']:l['cam_inv_rot_res'] + 6 * T].reshape(T, 6)[empty])
            d['cam_inv_trans_residual'] = conv(prm[l['cam_inv_trans_res']:l['cam_inv_trans_res'] + 3 * T].reshape(T, 3)[:Ts])
            for pi, idx in enumerate(self.person_ids[si]):
                pd = d['person_data'][idx]
                slot = si * P + pi
                n = int(pd['fr_end']) - int(pd['fr_start'])
                pp = prm[l['person0'] + pi * l['person_stride']:l['person0'] + (pi + 1) * l['person_stride']]
                pd['traj_local_xy'] = conv(pp[l['local_xy']:l['local_xy'] + 2])
                pd['traj_local_heading'] = conv(pp[l['local_heading']:l['local_heading'] + 1])
                pd['traj_local_dxy'] = conv(pp[l['local_dxy']:l['local_dxy'] + 2 * T].reshape(T, 2)[1:n])
                pd['traj_local_dheading'] = conv(pp[l['local_dheading']:l['local_dheading'] + T][1:n])
                pd['traj_local_z'] = conv(pp[l['local_z']:l['local_z'] + T][:n])
                pd['traj_local_rot'] = conv(pp[l['local_rot']:l['local_rot'] + 6 * T].reshape(T, 6)[:n])
                if 'world_dheading' in var or 'world_dheading' in pd:
                    pd['world_dheading'] = conv(pp[l['world_dheading']:l['world_dheading'] + T][:Ts, None])
                pd['smpl_orient_world'] = conv(h['orient_world'][slot, :Ts])
                pd['root_trans_world'] = conv(h['trans_world'][slot, :Ts])
                pd['kp_2d_pred'] = conv(h['kp_2d_pred'][slot, :Ts])
                pd['smpl_orient_cam_in_world'] = conv(h['orient_cam_in_world'][slot, :Ts])
