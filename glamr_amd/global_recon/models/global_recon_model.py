"""GlobalReconOptimizer on MI355X -- drop-in for global_recon/models/global_recon_model.py:23-659.

    model = GlobalReconOptimizer(cfg, device, log)        # cfg: reference-style Config object or the dict of configs.get_config()
    out = model.optimize(in_dict)                         # same input/output dictionaries as the reference (:572-589)
    outs = model.optimize_batch([in_dict, ...])           # extension: many independent sequences in one pass

Division of labour:
  * host (this file, numpy): the HybrIK wire format -> per-person arrays with the reference's frame/visibility bookkeeping
    (:88-148, bit-exact indices), scipy interpolation over gaps, `filter_pose` (a sequential data-dependent loop, :250-271), and
    the one-off camera / heading initialisation (:166-183, :273-317).
  * device (HIP kernels behind libglamr_hip.so): motion infilling + trajectory prediction for ALL persons of ALL sequences in one
    batched call, SMPL skinning for the cached root-relative joints, and every optimisation stage as ONE kernel launch that runs all
    Adam iterations of all scenes (glamr_grecon_run_stage).
There is no CPU path for the device part: a missing library or a non-HIP device raises.
"""
import contextlib
import os
import time

import numpy as np
import torch

from ... import _lib
from ...lib.models.smpl import SMPL, SMPL_MODEL_DIR
from ...lib.utils import np_transform as nt
from ...models.prior_models import MotionTrajJointModel
from ...models.priors import num_windows, NZ
from .. import packing
from ..configs import get_config

# (body26fk index, smpl index) pairs with identical joint names (lib/utils/joints.py:48-73,619-641 through :82-85): the 14
# keypoints of HybrIK's 24 SMPL joints that GLAMR trusts; the other 12 body26fk joints keep score 0.
SMPL_TO_BODY26FK = np.array([[8, 8], [5, 5], [2, 2], [21, 17], [23, 19], [25, 21], [7, 7], [4, 4], [1, 1],
                             [20, 16], [22, 18], [24, 20], [6, 12], [0, 0]])


def _cfg_parts(cfg):
    if isinstance(cfg, str):
        cfg = get_config(cfg)
    if isinstance(cfg, dict):
        return cfg['grecon_model_specs'], cfg['opt_stage_specs'], cfg.get('id', 'glamr')
    return cfg.grecon_model_specs, cfg.opt_stage_specs, getattr(cfg, 'id', 'glamr')


_PERSON_KEYS = ('smpl_pose', 'smpl_beta', 'smpl_orient_cam', 'root_trans_cam', 'cam_K', 'traj_local_pred', 'smpl_orient_world_base',
                'root_trans_world_base', 'smpl_orient_world', 'root_trans_world', 'kp_2d_pred', 'smpl_orient_cam_in_world', 'traj_local_xy',
                'traj_local_heading', 'traj_local_dxy', 'traj_local_dheading', 'traj_local_z', 'traj_local_rot', 'visible', 'visible_orig',
                'exist_frames', 'frames', 'vis_frames', 'invis_frames', 'frame2ind', 'kp_2d', 'kp_2d_aligned', 'kp_2d_score', 'person2cam')


class ResidentInputs:
    """A batch of sequences in HBM as stage_inputs() leaves it: the HybrIK arrays on their frame rows (`g`), lengths, optional latent
    draws, plus the host-side bookkeeping (person ids, sequence names) that never needs the device."""
    pass


class LazyDict(dict):
    """A dictionary some of whose values are produced on first access (and then kept).  optimize_batch returns the reference's per-person
    dictionaries (~40 arrays each); most consumers read a handful of them, so the slices of the batch arrays -- and
    the ones that need a conversion (float64 copies of the masks and keypoints, 4 x 4 expansions, frame tables) -- are cut when somebody
    asks.  Pending values come from per-key thunks (`lazy`) or from one `factory(key)` shared by the keys in `factory_keys`.  Behaves as
    a plain dict otherwise: `in`, iteration, len, items(), pickling and equality see every key."""

    def __init__(self, eager, lazy=(), factory=None, factory_keys=()):
        super().__init__(eager)
        self._lazy = dict(lazy)
        self._factory, self._fkeys, self._gone = factory, factory_keys, None

    def _pending(self, key):
        return key in self._lazy or (key in self._fkeys and not dict.__contains__(self, key) and not (self._gone and key in self._gone))

    def __missing__(self, key):
        if key in self._lazy:
            value = self._lazy.pop(key)()
        elif self._pending(key):
            value = self._factory(key)
        else:
            raise KeyError(key)
        dict.__setitem__(self, key, value)
        return value

    def _force(self):
        for k in list(self._lazy):
            self[k]
        for k in self._fkeys:
            if self._pending(k):
                self[k]
        return self

    def get(self, key, default=None):
        return self[key] if key in self else default

    def __contains__(self, key):
        return dict.__contains__(self, key) or self._pending(key)

    def __iter__(self):
        return iter(self._force().keys())

    def __len__(self):
        return dict.__len__(self) + len(self._lazy) + sum(1 for k in self._fkeys if self._pending(k) and k not in self._lazy)

    def keys(self):
        return dict.keys(self._force())

    def values(self):
        return dict.values(self._force())

    def items(self):
        return dict.items(self._force())

    def __eq__(self, other):
        return dict.__eq__(self._force(), other)

    def __ne__(self, other):
        return not self.__eq__(other)

    def __setitem__(self, key, value):
        self._lazy.pop(key, None)
        dict.__setitem__(self, key, value)

    def __delitem__(self, key):
        self.pop(key)

    def pop(self, key, *default):
        if self._pending(key):
            self[key]
        if key in self._fkeys:
            self._gone = (self._gone or set()) | {key}
        return dict.pop(self, key, *default)

    def copy(self):
        return dict(self._force())

    def __reduce__(self):
        return (dict, (dict(self._force()),))


def coschedule_enabled():
    """GLAMR_COSCHEDULE=0 switches the staggered two-stream pipeline (PipelineGate + GLAMR_NETS_COSCHEDULE kernels) off."""
    import os
    return os.environ.get('GLAMR_COSCHEDULE', '1') != '0'


class PipelineGate:
    """Staggers batches that alternate between TWO streams so that a batch's motion infiller runs while the previous batch's optimiser stage
    is resident (GlobalReconOptimizer.pipeline_gate).  A stage workgroup leaves three of a CU's four SIMDs half empty and its matrix pipes
    idle, and the infiller's co-schedulable kernels (GLAMR_NETS_COSCHEDULE, csrc/nn_free.hpp) fit into exactly that space -- but two streams
    left to themselves fall into step (both in their priors, then both stages one after the other) and nothing overlaps.  The rule: a batch
    starts its device work when the PREVIOUS batch's priors have finished, a few milliseconds before that batch's stage starts.  (Measured and
    dropped: opening the gate only when the stage is launched -- the infiller then loses its head start on an almost empty GPU, does not
    finish within the stage's 35 ms, and the step goes from 43.4 to 49.0 ms.)  Measured on
    1024 x 300 frames (tools/pipeline_probe.py): 42.6 ms per batch against 47.1 for the LDS kernels free-running."""

    def __init__(self):
        self.last = None

    def before(self, stream):
        if self.last is not None:
            stream.wait_event(self.last)

    def after(self, stream):
        # (an event created with hipEventReleaseToDevice instead of torch's system-scope one: measured, no difference -- profiles/r05_pipeline_experiments.log)
        ev = torch.cuda.Event()
        ev.record(stream)
        self.last = ev


class ResidentGraph:
    """A captured optimize_resident step (GlobalReconOptimizer.capture_resident): one HIP graph, or -- under a PipelineGate -- two, split
    where the gate's event is recorded (an event shared with another stream's graph cannot live inside a captured graph)."""

    def __init__(self, graph, datas, packed, stream, tail=None, gate=None, head=None):
        self.graph, self.datas, self.packed, self.stream, self.tail, self.gate, self.head = graph, datas, packed, stream, tail, gate, head

    def replay(self):
        with torch.cuda.stream(self.stream):
            if self.head is not None:
                self.head.replay()                                    # what precedes the priors does not wait for the gate (GLAMR_GATE_PREP=early)
            if self.gate is not None:
                self.gate.before(self.stream)
            self.graph.replay()
            if self.tail is not None:
                if self.gate is not None:
                    self.gate.after(self.stream)
                self.tail.replay()
        return self.packed


class GlobalReconOptimizer:

    def __init__(self, cfg, device=torch.device('cuda'), log=None, smpl=None, mt_model=None, results_root='results'):
        self.cfg = cfg
        self.specs, self.opt_stage_specs, self.cfg_id = _cfg_parts(cfg)
        self.device = torch.device(device)
        self.log = log
        if self.device.type != 'cuda':
            raise RuntimeError('glamr_amd.GlobalReconOptimizer runs on an MI355X (device %r given); there is no CPU fallback' % (device,))
        _lib.lib()                                                     # fail early and loudly if the HIP library is missing
        self.pipeline_gate = None                                      # a PipelineGate when the caller alternates batches between two streams
        self._capture_split = None
        self._capture_head_split = None
        # Per-iteration loss log (:564,646-659).  The reference calls write_logs after EVERY optimizer.step -- with `log` None it prints.  Here
        # the iterations of a stage are one kernel launch: with a `log` (or keep_loss_history = True) the launch records the unweighted value of
        # every term at every iteration (glamr_scene_batch.loss_history; the plain instance with the reporting evaluation, about 2x the time),
        # run_schedule hands them to `log.info` in the reference's line format once the stage is done, and `loss_history[stage]` keeps the
        # (scenes, iterations, GLAMR_NUM_LOSSES) array.  Without a log nothing is recorded and nothing is printed: a batch of 1024 sequences
        # would be half a million lines.
        self.keep_loss_history = False
        self.loss_history = {}
        g = self.specs.get
        if g('est_type', 'hybrik') != 'hybrik' or not g('flag_infer_motion_traj', False) or not g('flag_pred_traj', True) \
                or not g('flag_opt_traj', True) or not g('flag_infill_motion', True):
            raise NotImplementedError('only est_type=hybrik with motion infilling + trajectory prediction + trajectory optimisation '
                                      '(every shipped config) is supported')
        # flag_opt_vis_local_rot (:45,416-419): the per-frame rotation residual `traj_local_rot` is only applied at the frames a person is SEEN in.
        # A residual that is never applied gets no gradient (its regulariser's is 2 w r = 0 at r = 0) and Adam leaves it at its initial zero: the
        # flag amounts to "no update of traj_local_rot at invisible frames" -- run_schedule then goes launch by launch with that gradient mask
        self.flag_opt_vis_local_rot = bool(g('flag_opt_vis_local_rot', False))
        if self.flag_opt_vis_local_rot and (g('flag_opt_motion_latent', False) or g('flag_opt_traj_latent', False)):
            # (run_latent_schedule makes its own iterations and does not apply the gradient mask: refused rather than silently ignored)
            raise NotImplementedError('flag_opt_vis_local_rot together with the latent-optimisation mode')
        # flag_traj_from_cam (:55,237,325-351): the world trajectory is first read off the initial camera (orientation interpolated between the
        # frames a person is seen in, heading separately).  With a trajectory predictor -- the only mode this path runs -- init_traj_heading_from_cam
        # then overwrites every EXISTING frame (:283-289), so what the flag changes is the base pose of the frames outside a person's existence
        # range.  Round 5: on the device as well (glamr_init_scenes_ex, GLAMR_INIT_TRAJ_FROM_CAM); init_data_batch_host keeps its numpy twin.
        self.flag_traj_from_cam = bool(g('flag_traj_from_cam', False))
        if self.flag_traj_from_cam and g('traj_interp_method', 'linear_interp') != 'linear_interp':
            raise NotImplementedError("flag_traj_from_cam with traj_interp_method other than 'linear_interp'")
        # absolute_heading (:59,283,421): the heading entries of the local trajectory are absolute -- packing.stage_desc passes
        # GLAMR_FLAG_ABSOLUTE_HEADING, and every launch of the stage kernel (the 'init' forward passes included) then runs on the instances of
        # csrc/grecon_wide.hip.  (In latent mode the reference also converts the PREDICTED headings before opt_latent_start_iter, :441-444: refused.)
        if g('absolute_heading', False) and (g('flag_opt_motion_latent', False) or g('flag_opt_traj_latent', False)):
            raise NotImplementedError('absolute_heading together with the latent-optimisation mode')
        for flag in ('flag_opt_person2cam_rot',
                     'flag_opt_person2cam_trans', 'flag_use_pen_loss'):
            if g(flag, False):
                raise NotImplementedError('%s is not supported by the MI355X path' % flag)
        if g('heading_type', 'scalar') != 'scalar' or not g('flag_cam_inv_trans_res_all', True) or not g('flag_opt_cam', True):
            raise NotImplementedError('unsupported grecon_model_specs')
        # latent-optimisation mode (:43-44,155-158,434-437,619-622): the priors run INSIDE the Adam loop and the latent draws are parameters
        self.flag_opt_motion_latent = bool(g('flag_opt_motion_latent', False))
        self.flag_opt_traj_latent = bool(g('flag_opt_traj_latent', False))
        self.flag_filter_pose = g('flag_filter_pose', True)
        # keypoint-count filter inside filter_pose (:50-52,264-268); with HybrIK's binary scores (14 scored joints per detected frame) it is all or
        # nothing: the reference's default minimum of 15 removes every frame, 14 or less none
        self.flag_make_invis_with_keypoint = bool(g('flag_make_invis_with_keypoint', False))
        self.make_invis_keypoint_min_score = float(g('make_invis_keypoint_min_score', 0.6))
        self.make_invis_keypoint_min_num = int(g('make_invis_keypoint_min_num', 15))
        self.flag_init_cam_all_frames = g('flag_init_cam_all_frames', False)
        self.cam_fix_frames = [tuple(x) for x in g('cam_fix_frames', [[0, None]])]
        self.smpl = smpl if smpl is not None else SMPL(SMPL_MODEL_DIR, pose_type='body26fk', create_transl=False).to(self.device)
        self.mt_model = mt_model if mt_model is not None else MotionTrajJointModel(None, self.device, log, smpl=self.smpl, results_root=results_root)
        self.timings = {}

    # ------------------------------------------------------------------------------------------------------------------------
    # host preprocessing of one person (global_recon_model.py:88-148)
    # ------------------------------------------------------------------------------------------------------------------------
    def _person_arrays(self, src):
        d = {}
        d['visible'] = visible = src['bboxes_dict']['exist'].copy()                      # float64, compared with == 1 / == 0
        d['visible_orig'] = visible.copy()
        nz = np.where(visible)[0]
        d['fr_start'], d['fr_end'] = start, end = nz[0], nz[-1] + 1
        d['exist_frames'] = visible == 1
        d['exist_frames'][start:end] = True
        d['exist_len'] = end - start
        d['max_len'] = max_len = visible.shape[0]
        d['frames'] = np.arange(max_len)
        vis_frames = visible == 1
        d['frame2ind'] = {f: i for i, f in enumerate(d['frames'])}
        d['scale'] = None
        rm = src['smpl_pose_quat_wroot']
        nvis = rm.shape[0]
        aa = nt.rotmat_to_rotvec_nearest(rm.reshape((-1, 3, 3))).reshape((nvis, -1, 3)).astype(np.float32)     # = scipy Rotation.from_matrix(..).as_rotvec() (:105-108)
        d['smpl_pose'] = aa[:, 1:].reshape(-1, 69)
        d['smpl_beta'] = src['smpl_beta']
        d['smpl_orient_cam'] = aa[:, 0]
        d['root_trans_cam'] = src['root_trans']
        kp = np.concatenate((src['kp_2d'][:, :24], np.ones_like(src['kp_2d'][:, :24, [0]])), axis=-1)
        kps = np.zeros((int(vis_frames.sum()), 26, 3))
        kps[:, SMPL_TO_BODY26FK[:, 0]] = kp[:, SMPL_TO_BODY26FK[:, 1]]
        d['kp_2d'], d['kp_2d_score'] = kps[:, :, :2], kps[:, :, 2]
        d['kp_2d_aligned'] = d['kp_2d'].copy()
        d['cam_K'] = src['cam_K'].astype(np.float32)
        if not np.all(visible):
            for key in ('kp_2d', 'kp_2d_score', 'kp_2d_aligned', 'cam_K'):
                full = np.zeros((max_len,) + d[key].shape[1:], dtype=d[key].dtype)
                full[vis_frames] = d[key]
                d[key] = full
            vis_ind = np.where(visible)[0]
            for key in ('smpl_pose', 'smpl_beta', 'root_trans_cam', 'smpl_orient_cam'):
                d[key] = nt.lerp_extrapolate(vis_ind, d[key], max_len)
        for key in ('smpl_pose', 'smpl_beta', 'root_trans_cam', 'smpl_orient_cam'):
            d[key] = np.ascontiguousarray(d[key], dtype=np.float32)
        if self.flag_filter_pose:
            self._filter_pose(d)
            if getattr(self, 'flag_make_invis_with_keypoint', False):                    # :264-268
                vis_ind = np.where(d['visible'] == 1.0)[0]
                num_valid = (d['kp_2d_score'][vis_ind] > self.make_invis_keypoint_min_score).sum(axis=1)
                d['visible'][vis_ind[num_valid < self.make_invis_keypoint_min_num]] = 0.0
        d['vis_frames'] = d['visible'] == 1
        d['invis_frames'] = d['visible'] == 0
        # identity initial camera: world := camera frame (:141-144)
        d['root_trans_world_base'] = d['root_trans_cam'].copy()
        d['smpl_orient_world_base'] = nt.quat_to_aa(nt.rotmat_to_quat(nt.aa_to_rotmat(d['smpl_orient_cam'])))
        d['smpl_pose_nofill'] = d['smpl_pose'].copy()
        d['smpl_pose_nofill'][~d['exist_frames']] = 0.0
        return d

    @staticmethod
    def _filter_pose(d):
        """:250-262, the numpy twin of the loop in csrc/init.hip prep_person_kernel (sequential and data dependent: one thread there)."""
        visible = d['visible']
        quat = nt.aa_to_quat(d['smpl_orient_cam'])
        jump = nt.quat_angle_between(quat[1:], quat[:-1])
        ind = np.where((jump > np.pi / 3) & (visible[1:] != 0))[0] + 1
        ind_set = set(int(i) for i in ind)
        for i in ind:
            if visible[i - 1]:
                if i + 1 < quat.shape[0] and visible[i + 1] and (i + 1) not in ind_set:
                    visible[i - 1] = 0
                else:
                    visible[i] = 0

    # ------------------------------------------------------------------------------------------------------------------------
    # scene initialisation after the priors have run (:166-246)
    # ------------------------------------------------------------------------------------------------------------------------
    def _init_scene(self, in_dict, persons):
        first = next(iter(persons.values()))
        num_fr = first['max_len']
        for d in persons.values():
            d['person_transform_cam'] = nt.make_transform(d['smpl_orient_cam'], d['root_trans_cam'])
            d['person2cam'] = nt.invert_transform(d['person_transform_cam'])
            d['person_transform_world'] = nt.make_transform(d['smpl_orient_world'], d['root_trans_world'])
            n = int(d['exist_len'])
            d['traj_local_xy'] = np.zeros(2, np.float32)
            d['traj_local_dxy'] = np.zeros((n - 1, 2), np.float32)
            d['traj_local_heading'] = np.zeros(1, np.float32)
            d['traj_local_dheading'] = np.zeros(n - 1, np.float32)
            d['traj_local_z'] = np.zeros(n, np.float32)
            d['traj_local_rot'] = np.zeros((n, 6), np.float32)
        ids = list(persons.keys())
        rel = {}
        for i in range(len(ids)):
            for j in range(len(ids)):
                if i != j:
                    rel[(i, j)] = np.matmul(nt.invert_transform(persons[ids[i]]['person_transform_cam']), persons[ids[j]]['person_transform_cam'])
        fr_num_persons = sum(d['vis_frames'].astype(np.int64) for d in persons.values())
        n_empty = int((fr_num_persons == 0).sum())
        data = {
            'seq_name': in_dict['seq_name'], 'person_data': persons, 'seq_len': num_fr, 'fr_num_persons': fr_num_persons,
            'cam_pose': np.tile(np.eye(4, dtype=np.float32), (num_fr, 1, 1)), 'cam_pose_inv': np.tile(np.eye(4, dtype=np.float32), (num_fr, 1, 1)),
            'cam_inv_rot_residual': np.zeros((n_empty, 6), np.float32), 'cam_inv_trans_residual': np.zeros((num_fr, 3), np.float32),
            'rel_transform_cam': rel, 'gt': in_dict.get('gt', {}), 'gt_meta': in_dict.get('gt_meta', {}),
            'meta': {'algo': 'global_recon', 'num_fr': num_fr},
        }
        self._init_cam_pose(data, all_frames=False)
        if self.flag_traj_from_cam:                                  # get_traj_from_cam (:325-351), traj_interp_method 'linear_interp'
            for d in persons.values():
                w = np.matmul(data['cam_pose_inv'], d['person_transform_cam'])
                d['person_transform_world'] = w
                q = nt.rotmat_to_quat(np.ascontiguousarray(w[:, :3, :3]))
                qi = nt.interp_orient_sep_heading(q[d['vis_frames']], d['vis_frames'])
                d['root_trans_world'] = d['root_trans_world_base'] = np.ascontiguousarray(w[:, :3, 3]).astype(np.float32)
                d['smpl_orient_world'] = d['smpl_orient_world_base'] = nt.quat_to_aa(qi).astype(np.float32)
        # heading initialisation from the camera (:273-292); the resulting world trajectory is produced by the device forward pass
        for d in persons.values():
            w = np.matmul(data['cam_pose_inv'], d['person_transform_cam'])
            q = nt.rotmat_to_quat(np.ascontiguousarray(w[:, :3, :3]))
            qi = nt.interp_orient_sep_heading(q[d['vis_frames']], d['vis_frames'])
            local = nt.global_to_local_traj(w[:, :3, 3], qi)[d['exist_frames']]
            for (s, e) in self.cam_fix_frames:
                d['traj_local_pred'][s:e, -2:] = local[s:e, -2:]
        return data

    def _init_cam_pose(self, data, all_frames):
        """:294-317 -- camera-to-world from the FIRST person only; with all_frames the frames the first person is not seen in are
        left as zero matrices (the reference's forward fill writes into a discarded tensor)."""
        first = next(iter(data['person_data'].values()))
        cand = np.matmul(first['person_transform_world'], first['person2cam']) * first['vis_frames'][:, None, None].astype(np.float32)
        ind = data['fr_num_persons'] > 0
        start = np.where(ind)[0][0]
        inf = np.zeros_like(data['cam_pose'])
        inf[ind] = cand[ind]
        if not all_frames:
            inf[...] = inf[[start]].copy()
        inf[:, :3, :3] = nt.sixd_to_rotmat(nt.rotmat_to_6d(inf[:, :3, :3]))
        data['pose_infer_cam_pose_inv'] = inf
        data['cam_pose_inv'] = inf
        data['cam_pose'] = nt.invert_transform(inf)

    # ------------------------------------------------------------------------------------------------------------------------
    # batched pipeline
    # ------------------------------------------------------------------------------------------------------------------------
    def init_data_batch_host(self, in_dicts, latents=None, init_forward=True):
        """Host-side variant of init_data_batch (numpy; kept for cam_fix_frames other than the default and as a cross-check):
        host preprocessing + ONE batched prior inference + scene initialisation.  latents: optional list (per sequence) of
        {person idx: {'motion': (n_windows,128), 'traj': (1,128)}} replacing the Gaussian draws."""
        t0 = time.time()
        dev = self.device
        scenes = [{idx: self._person_arrays(src) for idx, src in in_dict['est'].items()} for in_dict in in_dicts]
        flat = [(si, idx, d) for si, persons in enumerate(scenes) for idx, d in persons.items()]
        lens = [int(d['exist_len']) for _, _, d in flat]
        Tm = max(lens)
        nw = num_windows(Tm)
        pose = np.zeros((len(flat), Tm, 69), np.float32)
        vis = np.zeros((len(flat), Tm), np.float32)
        for k, (si, idx, d) in enumerate(flat):
            ex = d['exist_frames']
            pose[k, :lens[k]] = d['smpl_pose_nofill'][ex]
            vis[k, :lens[k]] = d['visible'][ex]
        if latents is not None:
            meps = np.zeros((len(flat), nw, NZ), np.float32)
            teps = np.zeros((len(flat), NZ), np.float32)
            for k, (si, idx, d) in enumerate(flat):
                m = np.asarray(latents[si][idx]['motion'], np.float32)
                meps[k, :m.shape[0]] = m
                teps[k] = np.asarray(latents[si][idx]['traj'], np.float32).reshape(-1)
            meps, teps = torch.from_numpy(meps).to(dev), torch.from_numpy(teps).to(dev)
        else:
            meps, teps = torch.randn((len(flat), nw, NZ), device=dev), torch.randn((len(flat), NZ), device=dev)
        t1 = time.time()
        out = self.mt_model.infer_padded(torch.from_numpy(pose).to(dev), torch.from_numpy(vis).to(dev), lens, meps, teps)
        out = {k: v.cpu().numpy() for k, v in out.items()}
        t2 = time.time()
        for k, (si, idx, d) in enumerate(flat):                                              # infer_motion_traj :370-392
            ex, n = d['exist_frames'], lens[k]
            d['infilled'] = d['traj_predicted'] = True
            d['smpl_pose'] = d['smpl_pose'].copy()
            d['smpl_pose'][ex] = out['pose'][k, :n]
            d['traj_local_pred'] = out['local_traj'][k, :n].copy()
            d['smpl_orient_world_base'][ex] = out['orient'][k, :n]
            d['root_trans_world_base'][ex] = out['trans'][k, :n]
            d['smpl_orient_world'], d['root_trans_world'] = d['smpl_orient_world_base'], d['root_trans_world_base']
        datas = [self._init_scene(in_dict, persons) for in_dict, persons in zip(in_dicts, scenes)]
        t3 = time.time()
        # root-relative joints, cached for the whole optimisation: one batched skinning call
        poses = np.concatenate([d['smpl_pose'] for _, _, d in flat])
        betas = np.concatenate([d['smpl_beta'] for _, _, d in flat])
        zeros = torch.zeros((poses.shape[0], 3), device=dev)
        with torch.no_grad():
            jl = self.smpl(global_orient=zeros, body_pose=torch.from_numpy(poses).to(dev), betas=torch.from_numpy(betas).to(dev),
                           root_trans=zeros, return_verts=False).joints
        j_locals, off = [dict() for _ in scenes], 0
        for si, idx, d in flat:
            j_locals[si][idx] = jl[off:off + d['max_len']]
            off += d['max_len']
        # first world trajectory + (optionally) the all-frames camera, then the 'init' forward pass (:241-246)
        packed = packing.PackedScenes(datas, j_locals, dev, self.cam_fix_frames)
        self._run(packed, self._forward_only_desc())
        if self.flag_init_cam_all_frames:
            h = packed.fetch(('orient_world', 'trans_world'))
            for si, dn in enumerate(datas):
                for pi, idx in enumerate(packed.person_ids[si]):
                    T = dn['seq_len']
                    dn['person_data'][idx]['person_transform_world'] = nt.make_transform(h['orient_world'][si * packed.P + pi, :T],
                                                                                           h['trans_world'][si * packed.P + pi, :T])
                self._init_cam_pose(dn, all_frames=True)
            packed.set_cam_pose([dn['cam_pose'] for dn in datas])
            self._run(packed, self._forward_only_desc())
        packed.unpack_into(datas, None, self.specs, as_torch=False)      # world trajectory / projections of the 'init' forward pass
        self.timings.update(host_pre=t1 - t0, priors=t2 - t1, host_init=t3 - t2, lbs_pack_init=time.time() - t3)
        return datas, packed


    # ------------------------------------------------------------------------------------------------------------------------
    # device pipeline: the host only scatters the HybrIK arrays to their frame rows (pose_est/hybrik_demo/demo.py:317-354 layout)
    # ------------------------------------------------------------------------------------------------------------------------
    def _staging(self, n_slots, T):
        """Pinned host staging buffers for the HybrIK arrays, two sets used alternately (a set is rewritten only after the upload that
        last used it has finished).  Rows of frames without a detection keep whatever an earlier batch left there: the device
        preparation reads detection rows only (init.hip prep_person_kernel); `exist` and `K` are cleared because their other rows count."""
        pool = self.__dict__.setdefault('_pinned', {'sets': [None, None], 'turn': 0})
        i = pool['turn']
        pool['turn'] = 1 - i
        cur = pool['sets'][i]
        widths = dict(exist=0, rot=216, betas=10, trans=3, kp=48, K=9)
        if cur is None or cur['shape'] != (n_slots, T):
            mk = lambda w: torch.empty((n_slots, T, w) if w else (n_slots, T), dtype=torch.float32, pin_memory=True).zero_()
            cur = {'shape': (n_slots, T), 't': {k: mk(w) for k, w in widths.items()}, 'event': None}
            cur['np'] = {k: v.numpy() for k, v in cur['t'].items()}
            pool['sets'][i] = cur
        else:
            if cur['event'] is not None:
                cur['event'].synchronize()
            cur['np']['exist'].fill(0.0)
            cur['np']['K'].fill(0.0)
        return cur

    def _pool(self):
        import concurrent.futures
        p = self.__dict__.get('_thread_pool')
        if p is None:
            p = self.__dict__['_thread_pool'] = concurrent.futures.ThreadPoolExecutor(max_workers=8)
        return p

    def _scatter_inputs(self, in_dicts, ids, P, h):
        """Per-detection HybrIK arrays -> their frame rows in the staging arrays `h` (numpy views): glamr_host_scatter, a few host threads
        doing block copies without the GIL (numpy on one thread took 60-130 ms per 1024 sequences depending on the box, a Python thread
        pool was slower still).  This loop only collects the ten numbers per person the library needs.  Returns (seq_len per slot,
        length of the existing range per slot, {(sequence, person id): exist array})."""
        import ctypes
        n_slots = len(in_dicts) * P
        T = h['exist'].shape[1]
        seq_len_slot = np.zeros(n_slots, np.int32)
        lens = np.full(n_slots, 11, np.int32)
        exists = {}
        table = np.zeros((n_slots, 10), np.int64)
        keep = []                                                  # converted copies must outlive the call
        f32, f64 = np.dtype(np.float32), np.dtype(np.float64)

        def addr(a, width):
            if type(a) is not np.ndarray or a.dtype != f32 or not a.flags.c_contiguous:
                a = np.ascontiguousarray(a, dtype=np.float32)
                keep.append(a)
            if a.size != nv * width:
                raise ValueError('a per-detection array has %d values, expected %d x %d' % (a.size, nv, width))
            return a.__array_interface__['data'][0]
        for si, d in enumerate(in_dicts):
            est = d['est']
            for pi, idx in enumerate(ids[si]):
                src = est[idx]
                ex = np.asarray(src['bboxes_dict']['exist'])
                exists[(si, idx)] = ex
                exa = ex
                if exa.dtype != f64 and exa.dtype != f32 or not exa.flags.c_contiguous:
                    exa = np.ascontiguousarray(exa, dtype=np.float64)
                    keep.append(exa)
                nv = len(src['smpl_beta'])
                kpw = 2 * int(np.shape(src['kp_2d'])[1])               # floats per detection row: HybrIK writes 29 keypoints, the wire format asks for >= 24
                table[si * P + pi] = (exa.__array_interface__['data'][0], exa.dtype == f64, exa.shape[0], nv, addr(src['smpl_pose_quat_wroot'], 216),
                                      addr(src['smpl_beta'], 10), addr(src['root_trans'], 3), addr(src['cam_K'], 9), addr(src['kp_2d'], kpw), kpw)
        # (rows left at zero = the empty person slots of scenes with fewer persons than the batch maximum: seq_len 0, nothing copied)
        stg = _lib.HostStaging(*[ctypes.c_void_p(h[k].ctypes.data) for k in ('exist', 'rot', 'betas', 'trans', 'K', 'kp')])
        _lib.check(_lib.lib().glamr_host_scatter(n_slots, ctypes.c_void_p(table.ctypes.data), T, ctypes.byref(stg), ctypes.c_void_p(seq_len_slot.ctypes.data),
                                                 ctypes.c_void_p(lens.ctypes.data), self._copy_threads()))
        return seq_len_slot, lens, exists

    def stage_inputs(self, in_dicts, latents=None, validate=True):
        """Host dictionaries -> HBM: checks them against the wire format (glamr_amd/utils/wire.py: keys and shapes on the host, values --
        finite numbers, orthonormal rotation matrices -- on the uploaded arrays, reported by check_inputs() / collect()), scatters the
        per-detection HybrIK arrays to their frame rows in pinned staging buffers and uploads them asynchronously (the ONLY host->device
        traffic of a batch).  Returns a ResidentInputs; everything after this runs on device arrays."""
        t0 = time.time()
        dev = self.device
        if validate:
            from glamr_amd.utils import wire
            for d in in_dicts:
                wire.check_layout(d['est'])
        S = len(in_dicts)
        ids = [list(d['est'].keys()) for d in in_dicts]
        P = max(len(x) for x in ids)
        Ts = [len(d['est'][i[0]]['bboxes_dict']['exist']) for d, i in zip(in_dicts, ids)]
        T = max(Ts)
        if P > 32:
            raise NotImplementedError('at most 32 persons per scene (csrc/grecon_wide.hip)')
        n_slots = S * P
        stg = self._staging(n_slots, T)
        h = stg['np']
        seq_len_slot, lens, exists = self._scatter_inputs(in_dicts, ids, P, h)
        rin = ResidentInputs()
        rin.S, rin.P, rin.T, rin.Ts, rin.ids, rin.lens, rin.exists = S, P, T, Ts, ids, lens, exists
        rin.g = {k: v.to(dev, non_blocking=True) for k, v in stg['t'].items()}
        rin.n_persons = torch.tensor([len(x) for x in ids], dtype=torch.int32).to(dev, non_blocking=True)
        rin.seq_len = torch.tensor(Ts, dtype=torch.int32).to(dev, non_blocking=True)
        rin.seq_len_slot = torch.from_numpy(seq_len_slot).to(dev, non_blocking=True)
        rin.meps = rin.teps = None
        if latents is not None:
            nw = num_windows(int(lens.max()))
            meps = np.zeros((n_slots, nw, NZ), np.float32)
            teps = np.zeros((n_slots, NZ), np.float32)
            for si in range(S):
                for pi, idx in enumerate(ids[si]):
                    m = np.asarray(latents[si][idx]['motion'], np.float32)
                    meps[si * P + pi, :m.shape[0]] = m
                    teps[si * P + pi] = np.asarray(latents[si][idx]['traj'], np.float32).reshape(-1)
            rin.meps, rin.teps = torch.from_numpy(meps).to(dev), torch.from_numpy(teps).to(dev)
        rin.meta = [{'seq_name': d['seq_name'], 'seq_len': Ts[si], 'gt': d.get('gt', {}), 'gt_meta': d.get('gt_meta', {})} for si, d in enumerate(in_dicts)]
        # one event for "uploaded", on the stream this call ran on: the staging set is rewritten only after it and a pipelined caller's compute
        # stream waits for it.  The VALUE checks of the wire format run on the device later, on the stream that consumes the batch
        # (value_checks(), called by init_resident): as kernels of the upload stream they gated `upload_done` while competing for CUs with
        # the previous batch's optimiser stage -- a pipelined caller's next batch started 15 ms late.
        rin.verdict = None
        rin.validate = bool(validate)
        stg['event'] = torch.cuda.Event()
        stg['event'].record()
        rin.upload_done = stg['event']
        self.timings['host_pre'] = time.time() - t0
        return rin

    @staticmethod
    def value_checks(rin):
        """Every number of a detection row finite, the 24 matrices of `smpl_pose_quat_wroot` orthonormal to 1e-2 (the field is named after
        quaternions, demo.py:320): glamr_check_inputs, ONE pass over the uploaded arrays on the current stream (the torch expressions it
        replaces were 5 ms of reductions per 1024 sequences), once per ResidentInputs; the verdict (2, n_slots) stays on the device until
        check_inputs() / collect() reads it."""
        if not getattr(rin, 'validate', False) or rin.verdict is not None:
            return
        import ctypes
        n_slots, T, g = rin.S * rin.P, rin.T, rin.g
        raw = _lib.RawBatch()
        raw.n_slots, raw.max_len = n_slots, T
        for name, ten in (('seq_len', rin.seq_len_slot), ('exist', g['exist']), ('rotmats', g['rot']), ('betas', g['betas']), ('root_trans', g['trans']), ('kp_2d', g['kp'])):
            setattr(raw, name, ctypes.c_void_p(ten.data_ptr()))
        verdict = torch.empty((2, n_slots), dtype=torch.int32, device=g['exist'].device)
        _lib.check(_lib.lib().glamr_check_inputs(ctypes.byref(raw), _lib.ptr(g['K']), _lib.ptr(verdict), _lib.current_stream()))
        rin.verdict = verdict                                                                    # (2, n_slots) on the device
        rin.verdict_ready = torch.cuda.Event()
        rin.verdict_ready.record()
        rin.validate = False

    @staticmethod
    def _copy_threads():
        return max(1, min(8, (os.cpu_count() or 4) // 2))

    def check_inputs(self, rin):
        """Raises WireFormatError for the first person whose uploaded arrays failed the value checks (one small device->host copy)."""
        if getattr(rin, 'verdict', None) is None:
            if getattr(rin, 'validate', False) and not torch.cuda.is_current_stream_capturing():
                self.value_checks(rin)                                  # a batch nobody has checked yet (its only uses so far were graph replays)
            if getattr(rin, 'verdict', None) is None:
                return
        from glamr_amd.utils import wire
        hv = getattr(rin, 'verdict_host', None)
        if hv is not None:                                              # (copied with the batch's results: _fetch_async)
            hv[1].synchronize()
            v = hv[0].numpy()
            rin.verdict_host = None
        else:
            rin.verdict_ready.synchronize()                             # computed on the stream that consumed the batch
            v = rin.verdict.cpu().numpy()
        rin.verdict = None
        for kind, msg in ((0, 'smpl_pose_quat_wroot does not hold rotation matrices (|R R^T - I| > 1e-2); the field is named after quaternions but '
                              'carries 24 x 3 x 3 matrices regrouped by 4 (demo.py:320)'), (1, 'an input array contains non-finite values')):
            bad = np.flatnonzero(v[kind])
            if bad.size:
                si, pi = divmod(int(bad[0]), rin.P)
                raise wire.WireFormatError('sequence %d (%s), person %r: %s' % (si, rin.meta[si]['seq_name'], rin.ids[si][pi] if pi < len(rin.ids[si]) else pi, msg))

    def _init_prepare(self, rin, packed, pa_t):
        """glamr_init_prepare on the staged batch: visibility bookkeeping, rotation matrices -> axis-angle, interpolation over detection gaps,
        filter_pose (:88-148,250-271).  Returns the structures the following launches share."""
        import ctypes
        L, g, n_slots, T = _lib.lib(), rin.g, rin.S * rin.P, rin.T
        raw = _lib.RawBatch()
        raw.n_slots, raw.max_len = n_slots, T
        for name, ten in (('seq_len', rin.seq_len_slot), ('exist', g['exist']), ('rotmats', g['rot']), ('betas', g['betas']), ('root_trans', g['trans']), ('kp_2d', g['kp'])):
            setattr(raw, name, ctypes.c_void_p(ten.data_ptr()))
        pa = _lib.PersonArrays()
        for name, ten in pa_t.items():
            setattr(pa, name, ctypes.c_void_p(ten.data_ptr()))
        sb = packed.struct()
        ws = torch.empty(L.glamr_init_workspace_bytes(n_slots, T), dtype=torch.uint8, device=self.device)
        st = _lib.current_stream()
        fo = _lib.FilterOpts(int(bool(self.flag_filter_pose)), int(self.flag_make_invis_with_keypoint), self.make_invis_keypoint_min_score, self.make_invis_keypoint_min_num)
        _lib.check(L.glamr_init_prepare(ctypes.byref(raw), ctypes.byref(sb), ctypes.byref(pa), ctypes.byref(fo), _lib.ptr(ws), st))
        return pa, sb, ws, st

    def prepare_only(self, in_dicts):
        """The per-person preparation alone (no priors, no scene): `visible` after filter_pose, (person slots, frames) float32 on the host.
        For inspecting what the filters of :250-271 do to a batch (a batch they leave without a visible frame cannot be reconstructed)."""
        rin = self.stage_inputs(in_dicts)
        n_slots, T, dev = rin.S * rin.P, rin.T, self.device
        packed = packing.PackedScenes.empty(rin.S, rin.P, T, dev)
        packed.t['cam_K'], packed.t['n_persons'], packed.t['seq_len'] = rin.g['K'], rin.n_persons, rin.seq_len
        packed.t['j_local'] = torch.zeros(1, dtype=torch.float32, device=dev)
        pa_t = packing.carve_zeros([('visible_orig', torch.float32, (n_slots, T)), ('smpl_pose', torch.float32, (n_slots, T, 69)), ('smpl_beta', torch.float32, (n_slots, T, 10)),
                                    ('trans_cam', torch.float32, (n_slots, T, 3)), ('nets_pose', torch.float32, (n_slots, T, 69)), ('nets_vis', torch.float32, (n_slots, T))], dev)
        self._init_prepare(rin, packed, pa_t)
        torch.cuda.synchronize(dev)
        return packed.t['vis'].view(n_slots, T).cpu().numpy()

    def init_resident(self, rin, init_forward=True):
        """init_data (:76-248) on device-resident inputs: per-person preparation, motion priors, scene assembly, cached joints and
        the 'init' forward pass -- kernel launches only, nothing crosses PCIe.  Returns (datas, packed): `datas` are light
        per-sequence dictionaries that collect() completes from the device arrays."""
        import ctypes
        t1 = time.time()
        dev, L = self.device, _lib.lib()
        S, P, T, g = rin.S, rin.P, rin.T, rin.g
        n_slots = S * P
        gate = self.pipeline_gate
        # The gate is waited for where the PRIORS start, not where the batch starts: the arrays' zero fills and the per-person preparation (1.6 ms on
        # the chain preparation -> infiller -> predictor) read the inputs only and run beside the previous batch's priors; capture_resident cuts the
        # step into three graphs for it (preparation | priors + skinning | rest): 34.7 against 38.5 ms per 1024-sequence step.  Round 5 kept this
        # behind a knob because its replays sometimes differed from the plain step; round 6 found the cause (packed-fp32 VALU instructions beside the
        # other stream's MFMA kernels: glamr_amd/build.py) and removed it.  GLAMR_GATE_PREP=late selects the two-graph cut.
        prep_early = gate is not None and os.environ.get('GLAMR_GATE_PREP', 'early') != 'late'
        if not torch.cuda.is_current_stream_capturing():
            self.value_checks(rin)                                       # (first use of this batch only)
            if gate is not None and not prep_early:
                gate.before(torch.cuda.current_stream(dev))
        packed = packing.PackedScenes.empty(S, P, T, dev)
        packed.person_ids = rin.ids
        packed.seq_names = [str(m['seq_name']) for m in rin.meta]
        packed.t['cam_K'] = g['K']
        packed.t['n_persons'] = rin.n_persons
        packed.t['seq_len'] = rin.seq_len
        f32 = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)
        # inputs / outputs / workspace of the priors live at fixed addresses (two sets per stream): the library replays the call as a HIP graph
        nw = num_windows(int(rin.lens.max()))
        rs = self.mt_model.handle.resident_set(n_slots, T, nw) if hasattr(self.mt_model, 'handle') else None
        pa_t = packing.carve_zeros([('visible_orig', torch.float32, (n_slots, T)), ('smpl_pose', torch.float32, (n_slots, T, 69)),
                                    ('smpl_beta', torch.float32, (n_slots, T, 10)), ('trans_cam', torch.float32, (n_slots, T, 3))], dev)
        pa_t['nets_pose'] = rs['nets_pose'] if rs else f32(n_slots, T, 69)
        pa_t['nets_vis'] = rs['nets_vis'] if rs else f32(n_slots, T)
        # the cached joints are produced in place by the skinning kernel: keep a placeholder until then
        packed.t['j_local'] = f32(1)
        pa, sb, ws, st = self._init_prepare(rin, packed, pa_t)
        if prep_early:
            if self._capture_head_split is not None:
                self._capture_head_split()                             # capture_resident under a gate: the first cut (replay() waits for the gate there)
            elif not torch.cuda.is_current_stream_capturing():
                gate.before(torch.cuda.current_stream(dev))
        # the latent draws of the motion priors (given latents: copied into the priors' fixed arrays)
        if rs:
            meps, teps = rs['meps'], rs['teps']
            if rin.meps is not None:
                meps.copy_(rin.meps)
                teps.copy_(rin.teps)
            else:
                torch.randn(meps.shape, out=meps)
                torch.randn(teps.shape, out=teps)
        else:
            meps = rin.meps if rin.meps is not None else torch.randn((n_slots, nw, NZ), device=dev)
            teps = rin.teps if rin.teps is not None else torch.randn((n_slots, NZ), device=dev)
        # motion priors on every person of every sequence in one call
        def open_gate():
            if self._capture_split is not None:
                self._capture_split()                                   # capture_resident under a gate: the graph is cut here
            elif gate is not None and not torch.cuda.is_current_stream_capturing():
                gate.after(torch.cuda.current_stream(dev))
        # the next batch may start when this batch's priors are done -- or (GLAMR_GATE_AFTER=infiller) already when its infiller is
        # (development aid GLAMR_GATE_AFTER: 'infiller' / 'priors' (default) / 'scene' / 'skin' / 'forward' -- later = the rest of this batch's
        # preparation runs without the next batch's first kernels beside it, but that batch starts later: profiles/r05_pipeline_experiments.log)
        gate_at = os.environ.get('GLAMR_GATE_AFTER', 'priors') if gate is not None else 'priors'
        # Under a gate the SKINNING runs between the infiller and the trajectory predictor, i.e. before the gate opens: everything it needs (the
        # infilled poses in video-frame rows, glamr_init_scatter_pose) is known by then, and it runs alone (1.5 ms) instead of beside the next batch's
        # first kernels (2.4 ms): 34.7 against 35.4 ms per step (profiles/r06_pipeline_corruption.log).  (Round 5 moved it there to dodge the
        # corruption of its results beside the other stream's attention kernels; with that fixed at its root the order is kept for its speed.)
        # GLAMR_SKIN_AFTER_PRIORS=1 restores the old order (development aid).
        skin_early = gate is not None and os.environ.get('GLAMR_SKIN_AFTER_PRIORS') != '1' and gate_at in ('priors', 'infiller', 'skin') \
            and hasattr(self.mt_model, 'handle')

        def skin():
            # root-relative joints of every frame, cached for the whole optimisation
            packed.t['j_local'] = self.smpl.root_relative_joints(pa_t['smpl_pose'].view(-1, 69), pa_t['smpl_beta'].view(-1, 10)).view(n_slots, T, 26, 3)

        def after_infiller(out_inf):
            if gate_at == 'infiller':
                open_gate()
            if skin_early:
                _lib.check(L.glamr_init_scatter_pose(ctypes.byref(sb), ctypes.byref(pa), _lib.ptr(out_inf['pose']), st))
                skin()
                if gate_at == 'skin':
                    open_gate()                                         # (the next batch's infiller beside this batch's trajectory predictor)
        out = self.mt_model.infer_padded(pa_t['nets_pose'], pa_t['nets_vis'], rin.lens, meps, teps, buffers=rs, coschedule=gate is not None,
                                         between=after_infiller if (skin_early or gate_at == 'infiller') else None)
        if gate_at == 'priors':
            open_gate()
        packed.latents = (meps, teps)                                  # the draws this batch was initialised with (parameters in latent-optimisation mode)
        # (flag_traj_from_cam :237,325-351: the base pose of the frames outside a person's existence range read off the initial camera)
        _lib.check(L.glamr_init_scenes_ex(ctypes.byref(sb), ctypes.byref(pa), _lib.ptr(out['pose']), _lib.ptr(out['local_traj']), _lib.ptr(out['trans']),
                                          _lib.ptr(out['orient']), (1 if self.flag_traj_from_cam else 0) | (2 if skin_early else 0), _lib.ptr(ws), st))
        if gate_at == 'scene':
            open_gate()
        if not skin_early:
            skin()
        if gate_at == 'skin' and not skin_early:
            open_gate()
        # with flag_init_cam_all_frames this pass is only there for the world poses the cameras are initialised from; whoever needs the 'init'
        # outputs gets them from the second pass (init_forward), or from the first stage's last evaluation
        self._run(packed, self._forward_only_desc(poses_only=self.flag_init_cam_all_frames or not init_forward))
        if self.flag_init_cam_all_frames:
            sb = packed.struct()
            _lib.check(L.glamr_init_cam_all_frames(ctypes.byref(sb), st))
            # the 'init' forward pass with the new cameras (:246) only produces outputs (projections, camera-relative orientation, loss
            # values); a caller that runs the schedule right away overwrites every one of them with the first stage's last evaluation
            if init_forward:
                self._run(packed, self._forward_only_desc())
        if gate_at == 'forward':
            open_gate()
        packed.person_arrays = pa_t
        packed.exists = rin.exists
        packed.keepalive = (rin, ws, out)
        datas = [dict(m, meta={'algo': 'global_recon', 'num_fr': m['seq_len']}, _pending=True) for m in rin.meta]
        self.timings.update(priors=0.0, host_init=0.0, lbs_pack_init=time.time() - t1)
        return datas, packed

    def init_data_batch(self, in_dicts, latents=None, init_forward=True):
        """init_data (:76-248) for a batch of host dictionaries: stage_inputs + init_resident."""
        if self.cam_fix_frames != [(0, None)]:
            return self.init_data_batch_host(in_dicts, latents)
        return self.init_resident(self.stage_inputs(in_dicts, latents), init_forward=init_forward)

    _FETCH = ('fr_start', 'fr_end', 'vis', 'kp_2d', 'kp_score', 'cam_K', 'traj_local_pred', 'orient_cam', 'base_orient', 'base_trans', 'person2cam',
              'cam_pose', 'params', 'orient_world', 'trans_world', 'kp_2d_pred', 'orient_cam_in_world', 'losses')

    def _fetch_async(self, packed, stream=None):
        """Device -> pinned host copies of everything the output dictionaries are cut from, enqueued on `stream` (default: the current
        one).  Returns (host arrays, event); the arrays are valid once the event has completed."""
        src = {k: packed.t[k] for k in self._FETCH}
        src.update({'pa_' + k: v for k, v in packed.person_arrays.items() if k not in ('nets_pose', 'nets_vis')})
        if 'rel_transform_cam' in packed.t:
            src['rel'] = packed.t['rel_transform_cam']
        ctx = torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()
        host = {}
        rin = packed.keepalive[0] if getattr(packed, 'keepalive', None) else None
        with ctx:
            for k, v in src.items():
                hbuf = torch.empty(v.shape, dtype=v.dtype, pin_memory=True)      # (one pinned slab per batch was measured slower: 3 ms per allocation)
                hbuf.copy_(v, non_blocking=True)
                host[k] = hbuf
            hv = None
            if rin is not None and getattr(rin, 'verdict', None) is not None:
                # the wire-format verdict rides along (a blocking .cpu() of its own was 7 ms of host stall per batch: tools/host_profile.py)
                torch.cuda.current_stream(self.device).wait_event(rin.verdict_ready)
                hv = torch.empty(rin.verdict.shape, dtype=rin.verdict.dtype, pin_memory=True)
                hv.copy_(rin.verdict, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            if hv is not None:
                rin.verdict_host = (hv, ev)
        return host, ev

    def _materialise(self, datas, packed, stage_vars, fetched=None):
        """Builds the reference-style output dictionaries (numpy) from the device arrays: ONE device->host copy per array; per person, plain
        slices of the batch arrays right away and the converted ones (float64 masks / keypoints, 4 x 4 matrices, frame tables) on access."""
        P, T, l = packed.P, packed.T, packed.layout
        if fetched is None:
            fetched = self._fetch_async(packed)
        host, ev = fetched
        ev.synchronize()
        h = {k: v.numpy() for k, v in host.items()}
        self.last_losses = h['losses']
        rel = h.get('rel')
        eye_row = np.array([0, 0, 0, 1], np.float32)

        def to44(m12):
            out = np.zeros(m12.shape[:-1] + (4, 4), np.float32)
            out[..., :3, :] = m12.reshape(m12.shape[:-1] + (3, 4))
            out[..., 3, :] = eye_row
            return out
        fr_start, fr_end = h['fr_start'].tolist(), h['fr_end'].tolist()
        wd = 'world_dheading' in stage_vars
        person_keys = _PERSON_KEYS + (('world_dheading',) if wd else ())

        def person_factory(k, pi, si, Ts):
            fs, fe = fr_start[k], fr_end[k]
            n = fe - fs
            pp = h['params'][si, l['person0'] + pi * l['person_stride']:l['person0'] + (pi + 1) * l['person_stride']]
            frame = lambda name: (lambda: h[name][k, :Ts])
            cut = {
                'smpl_pose': frame('pa_smpl_pose'), 'smpl_beta': frame('pa_smpl_beta'), 'smpl_orient_cam': frame('orient_cam'), 'root_trans_cam': frame('pa_trans_cam'),
                'cam_K': lambda: h['cam_K'][k, :Ts].reshape(Ts, 3, 3), 'traj_local_pred': lambda: h['traj_local_pred'][k, :n],
                'smpl_orient_world_base': frame('base_orient'), 'root_trans_world_base': frame('base_trans'), 'smpl_orient_world': frame('orient_world'),
                'root_trans_world': frame('trans_world'), 'kp_2d_pred': frame('kp_2d_pred'), 'smpl_orient_cam_in_world': frame('orient_cam_in_world'),
                'traj_local_xy': lambda: pp[l['local_xy']:l['local_xy'] + 2], 'traj_local_heading': lambda: pp[l['local_heading']:l['local_heading'] + 1],
                'traj_local_dxy': lambda: pp[l['local_dxy']:l['local_dxy'] + 2 * T].reshape(T, 2)[1:n],
                'traj_local_dheading': lambda: pp[l['local_dheading']:l['local_dheading'] + T][1:n],
                'traj_local_z': lambda: pp[l['local_z']:l['local_z'] + T][:n], 'traj_local_rot': lambda: pp[l['local_rot']:l['local_rot'] + 6 * T].reshape(T, 6)[:n],
                'world_dheading': lambda: pp[l['world_dheading']:l['world_dheading'] + T][:Ts, None],
                'visible': lambda: h['vis'][k, :Ts].astype(np.float64), 'visible_orig': lambda: h['pa_visible_orig'][k, :Ts].astype(np.float64),
                'frames': lambda: np.arange(Ts), 'vis_frames': lambda: h['vis'][k, :Ts] == 1, 'invis_frames': lambda: h['vis'][k, :Ts] == 0,
                'frame2ind': lambda: {f: f for f in range(Ts)},
                'kp_2d': lambda: h['kp_2d'][k, :Ts].astype(np.float64), 'kp_2d_aligned': lambda: h['kp_2d'][k, :Ts].astype(np.float64),
                'kp_2d_score': lambda: h['kp_score'][k, :Ts].astype(np.float64), 'person2cam': lambda: to44(h['person2cam'][k, :Ts]),
            }

            def exist_frames():
                e = np.zeros(Ts, bool)
                e[fs:fe] = True
                return e
            cut['exist_frames'] = exist_frames
            return lambda key: cut[key]()
        fixed_cam = bool(self.specs.get('flag_fixed_cam', False))
        has_cam = 'cam' in stage_vars
        scene_keys = ('cam_pose', 'cam_pose_inv', 'fr_num_persons', 'cam_inv_rot_residual', 'rel_transform_cam', 'cam_inv_trans_residual') + \
            ((('cam_rot_6d_fix', 'cam_trans_fix') if fixed_cam else ('cam_rot_6d', 'cam_trans')) if has_cam else ())
        o_r6, o_tr, o_rres, o_tres = l['cam_rot6d'], l['cam_trans'], l['cam_inv_rot_res'], l['cam_inv_trans_res']
        h_vis, h_params, h_cam = h['vis'], h['params'], h['cam_pose']
        person_ids = packed.person_ids

        def scene_factory(si, Ts):
            """Everything of a scene that is cut from the batch arrays, built when the first such key is read."""
            prm = h_params[si]
            cam12 = h_cam[si, :Ts]
            n_p = len(person_ids[si])

            def num_persons():
                return sum((h_vis[si * P + pi, :Ts] == 1).astype(np.int64) for pi in range(n_p))

            def rot_residual():
                empty = np.where(num_persons() == 0)[0]
                return prm[o_rres:o_rres + 6 * T].reshape(T, 6)[empty]

            def rel_dict():
                if rel is None:
                    return {}
                return {(i, j): to44(rel[si, i, j, :Ts]) for i in range(n_p) for j in range(n_p) if i != j}
            r6 = lambda: prm[o_r6:o_r6 + 6 * T].reshape(T, 6)
            tr = lambda: prm[o_tr:o_tr + 3 * T].reshape(T, 3)
            cut = {'cam_pose': lambda: to44(cam12), 'cam_pose_inv': lambda: nt.invert_transform(to44(cam12)), 'fr_num_persons': num_persons,
                   'cam_inv_rot_residual': rot_residual, 'rel_transform_cam': rel_dict,
                   'cam_inv_trans_residual': lambda: prm[o_tres:o_tres + 3 * T].reshape(T, 3)[:Ts],
                   'cam_rot_6d_fix': lambda: r6()[:1], 'cam_trans_fix': lambda: tr()[:1], 'cam_rot_6d': lambda: r6()[:Ts], 'cam_trans': lambda: tr()[:Ts]}
            return lambda key: cut[key]()
        for si, d in enumerate(datas):
            Ts = d['seq_len']
            persons = {}
            for pi, idx in enumerate(person_ids[si]):
                k = si * P + pi
                fs, fe = fr_start[k], fr_end[k]
                eager = {'fr_start': fs, 'fr_end': fe, 'exist_len': fe - fs, 'max_len': Ts, 'scale': None, 'infilled': True, 'traj_predicted': True}

                def factory(key, args=(k, pi, si, Ts), cache=[None]):
                    if cache[0] is None:
                        cache[0] = person_factory(*args)
                    return cache[0](key)
                persons[idx] = LazyDict(eager, factory=factory, factory_keys=person_keys)

            def sfactory(key, args=(si, Ts), cache=[None]):
                if cache[0] is None:
                    cache[0] = scene_factory(*args)
                return cache[0](key)
            d.pop('_pending', None)
            d['person_data'] = persons
            datas[si] = LazyDict(d, factory=sfactory, factory_keys=scene_keys)
        return datas

    def _schedule_overwrites_init(self):
        """True when the configured schedule has at least one stage: its last evaluation rewrites all outputs of the 'init' forward pass."""
        return len(self.opt_stage_specs) > 0

    def _forward_only_desc(self, poses_only=False):
        first = next(iter(self.opt_stage_specs.values()))
        sd = packing.stage_desc(first, self.specs, has_world_dheading=False, niters=0)
        sd.var_mask = 0
        sd.flags &= ~packing.FLAG_CAM_FROM_PERSON           # stage 'init' keeps the initial camera (:473)
        if poses_only:
            sd.flags |= packing.FLAG_POSES_ONLY             # only orient_world / trans_world are wanted (the pass before init_cam_pose(all_frames))
        return sd

    def _run(self, packed, sd):
        import ctypes
        L = _lib.lib()
        sb = packed.struct()
        ws = torch.empty(L.glamr_grecon_workspace_bytes(packed.S, packed.P, packed.T), dtype=torch.uint8, device=self.device)
        _lib.check(L.glamr_grecon_run_stage(ctypes.byref(sb), ctypes.byref(sd), None, _lib.ptr(ws), _lib.current_stream()))
        return ws

    @staticmethod
    def launch_ms(ws):
        """Duration of the stage launch that used workspace `ws`, from the kernel's own clock (blocks until it has finished)."""
        import ctypes
        ns = ctypes.c_double()
        _lib.check(_lib.lib().glamr_grecon_last_launch_ns(_lib.ptr(ws), ctypes.byref(ns)))
        return ns.value * 1e-6

    def run_schedule(self, packed, max_iters=None, has_wd=False):
        """The staged optimisation (:250-262) of an initialised batch: one kernel launch per stage, asynchronous on the current
        stream.  `max_iters` caps the iterations of every stage (tests); None = the configured schedule.  `has_wd`: the scenes already
        carry a world heading offset (a continued optimisation, :459-465 applies it whenever the variable exists)."""
        if self.flag_opt_vis_local_rot:
            return self._run_schedule_masked(packed, max_iters, has_wd)
        events = []
        want_hist = (self.log is not None or self.keep_loss_history) and not torch.cuda.is_current_stream_capturing()
        for stage, spec in self.opt_stage_specs.items():
            sd = packing.stage_desc(spec, self.specs, has_world_dheading=has_wd,
                                    niters=None if max_iters is None else min(max_iters, spec['opt_niters']))
            if want_hist and sd.niters > 0:
                packed.t['loss_history'] = torch.zeros((packed.S, int(sd.niters), len(packing.LOSS_IDS)), dtype=torch.float32, device=self.device)
            events.append(self._run(packed, sd))               # the launch's workspace: its header carries the kernel's own clock stamps
            if 'loss_history' in packed.t:
                self._report_stage(packed, stage, spec, packed.t.pop('loss_history'), events[-1])
            has_wd = has_wd or 'world_dheading' in spec['opt_variables']
            if spec.get('reinitialize_cam', False):
                packed.t['cam_pose'][:] = packed.t['cam_pose'][:, :1]
        packed.has_world_dheading = has_wd
        packed.stage_ws = events
        return packed

    def _report_stage(self, packed, stage, spec, hist, ws):
        """write_logs (:646-659) for every iteration of a finished stage launch: `cfg id - sequence - stage | it/niters | TE: .. ETA: .. | LR: .. |
        term: value | ...` with the UNWEIGHTED values (loss_uw_dict, :564) of the stage's terms in the order of its loss_cfg.  The time per
        iteration is the launch's duration divided by its iterations."""
        import datetime
        h = hist.cpu().numpy()                                          # (waits for the launch)
        self.loss_history[stage] = h
        if self.log is None:
            return
        n_it = h.shape[1]
        it_secs = self.launch_ms(ws) * 1e-3 / max(1, n_it)
        hms = lambda secs: str(datetime.timedelta(seconds=round(secs)))
        names = [n for n in spec['loss_cfg'] if n in packing.LOSS_IDS]
        seqs = getattr(packed, 'seq_names', None) or ['seq%d' % si for si in range(packed.S)]
        for si in range(packed.S):
            head = '%s - %s - %s' % (self.cfg_id, seqs[si], stage)
            for it in range(n_it):
                loss_str = ' | '.join('%s: %7.3f' % (n, h[si, it, packing.LOSS_IDS[n]]) for n in names)
                self.log.info('%s | %4d/%d | TE: %s ETA: %s | LR: %.0e | %s' % (head, it, n_it, hms(it_secs), hms(it_secs * (n_it - it - 1)), spec['opt_lr'], loss_str))

    def _run_schedule_masked(self, packed, max_iters, has_wd):
        """flag_opt_vis_local_rot: the schedule launch by launch (parallel.PersonShardedSchedule on ONE rank without collectives: a gradient launch
        and glamr_adam_step per iteration, ~0.4 ms each) with the gradient of `traj_local_rot` zeroed at the frames a person is not seen in.  No
        shipped config sets the flag; the one-launch-per-stage kernel is untouched by it."""
        from glamr_amd import parallel
        if torch.cuda.is_current_stream_capturing():
            raise NotImplementedError('flag_opt_vis_local_rot runs launch by launch and cannot be captured into a step graph')
        if any(self.specs.get('flag_opt_cam_from_person_pose', False) and 'cam' not in spec['opt_variables'] for spec in self.opt_stage_specs.values()):
            raise NotImplementedError('flag_opt_vis_local_rot with a camera derived from the persons')
        S, P, T, l = packed.S, packed.P, packed.T, packed.layout
        vis = packed.t['vis'].view(S, P, T) > 0
        fr_start = packed.t['fr_start'].view(S, P).long()
        keep = torch.ones_like(packed.t['params'])
        t_idx = torch.arange(T, device=packed.device)
        for pi in range(P):
            # the residual of video frame t sits at row t - fr_start of the person's block (rows of frames before fr_start do not exist)
            e = t_idx[None, :] - fr_start[:, pi:pi + 1]                                        # (S, T)
            col = l['person0'] + pi * l['person_stride'] + l['local_rot'] + e * 6
            ok = (~vis[:, pi]) & (e >= 0)
            s_idx = torch.arange(S, device=packed.device)[:, None].expand(S, T)
            for k in range(6):
                keep[s_idx[ok], (col + k)[ok]] = 0.0

        def hook(packed_, stage, spec, grads):
            grads.mul_(keep)
        sched = parallel.PersonShardedSchedule(rank=0, world=1, grad_hook=hook, use_dist=False)
        sched.run(packed, self.opt_stage_specs, self.specs, max_iters=max_iters, has_wd=has_wd)
        packed.stage_ws = []
        return packed

    def optimize_resident(self, rin, max_iters=None):
        """HBM in, HBM out: init_data + the full schedule on a ResidentInputs batch.  Returns (datas, packed) with every result
        (optimised variables, world trajectories, projections, camera) in packed.t on the device; collect() brings them to the host."""
        if self.latent_mode:
            datas, packed = self.init_resident(rin, init_forward=True)
            self.run_latent_schedule(rin, packed, max_iters)
            return datas, packed
        datas, packed = self.init_resident(rin, init_forward=not self._schedule_overwrites_init())
        self.run_schedule(packed, max_iters)
        return datas, packed

    def capture_resident(self, rin, max_iters=None, stream=None, check=True):
        """optimize_resident(rin) as ONE replayable HIP graph: returns a ResidentGraph whose replay() enqueues the whole step (per-person
        preparation, priors, skinning, scene assembly, every optimisation stage: ~25 launches of this library, the priors' ~450 included) with a
        single launch on the host.  For callers that run the same batch geometry again and again on resident inputs whose CONTENT changes
        (a service loop; bench.py): a host whose driver calls are slow then no longer paces the GPU.  The step must have run once on this stream
        before (allocations, one-time attribute calls).  check=True replays once against a plain step with the same seed and requires the
        projections to agree bit for bit (RuntimeError otherwise).  Results live in graph.packed (collect() them after a replay)."""
        if self.latent_mode:
            raise NotImplementedError('the latent-optimisation schedule captures its own graph per iteration (run_latent_schedule); the whole step is not capturable')
        # the value checks of the wire format run once per batch, outside a capture (init_resident skips them while capturing): a batch whose
        # FIRST use is this capture is checked here -- the verdict surfaces in collect() / check_inputs() as usual
        self.value_checks(rin)
        st = stream if stream is not None else torch.cuda.current_stream(self.device)
        if st == torch.cuda.default_stream(self.device):
            st = torch.cuda.Stream(device=self.device)               # the legacy default stream cannot be captured
            st.wait_stream(torch.cuda.default_stream(self.device))
        graph = torch.cuda.CUDAGraph()
        # (thread-local capture mode: a watchdog thread of a process group may query events while this thread captures)
        if self.pipeline_gate is None:
            with torch.cuda.graph(graph, stream=st, capture_error_mode='thread_local'):
                datas, packed = self.optimize_resident(rin, max_iters)
            rg = ResidentGraph(graph, datas, packed, st)
        else:
            # two graphs sharing one memory pool: everything up to and including the priors, and the rest; replay() records the gate's event between them
            tail = torch.cuda.CUDAGraph()
            prep_early = os.environ.get('GLAMR_GATE_PREP', 'early') != 'late'
            head = torch.cuda.CUDAGraph() if prep_early else None     # (three graphs then: preparation | priors | the rest)
            torch.cuda.synchronize(self.device)
            with torch.cuda.stream(st):
                first = head if head is not None else graph
                first.capture_begin(capture_error_mode='thread_local')

                def head_split():
                    head.capture_end()
                    graph.capture_begin(pool=head.pool(), capture_error_mode='thread_local')

                def split():
                    graph.capture_end()
                    tail.capture_begin(pool=first.pool(), capture_error_mode='thread_local')
                self._capture_split = split
                self._capture_head_split = head_split if head is not None else None
                try:
                    datas, packed = self.optimize_resident(rin, max_iters)
                except BaseException:
                    # leave the stream out of capture mode whatever happened: whichever of the graphs is capturing is ended and dropped
                    # (a stream stuck in a broken capture fails every later launch -- bench.py's fall-back to plain launches included)
                    for gph in (tail, graph, head):
                        try:
                            if gph is not None:
                                gph.capture_end()
                        except Exception:      # noqa: BLE001 -- not capturing / already invalidated
                            pass
                    raise
                finally:
                    self._capture_split = None
                    self._capture_head_split = None
                tail.capture_end()
            rg = ResidentGraph(graph, datas, packed, st, tail=tail, gate=self.pipeline_gate, head=head)
        if check:
            torch.cuda.synchronize(self.device)
            with torch.random.fork_rng(devices=[self.device]):          # the caller's generators are left as they were
                seed = 20260926
                torch.manual_seed(seed)
                with torch.cuda.stream(st):
                    _, ref = self.optimize_resident(rin, max_iters)
                torch.cuda.synchronize(self.device)
                want = ref.t['kp_2d_pred'].clone()
                torch.manual_seed(seed)
                rg.replay()
                torch.cuda.synchronize(self.device)
            got = packed.t['kp_2d_pred']
            if not (bool(torch.isfinite(got).all()) and torch.equal(got, want)):
                raise RuntimeError('the replayed step does not reproduce the plain one (max |diff| %.3g)' % float((got - want).abs().max()))
        return rg

    @property
    def latent_mode(self):
        return self.flag_opt_motion_latent or self.flag_opt_traj_latent

    def run_latent_schedule(self, rin, packed, max_iters=None):
        """The staged optimisation in LATENT-OPTIMISATION mode (flag_opt_motion_latent / flag_opt_traj_latent; :155-158,434-437,619-622).
        Every iteration from `opt_latent_start_iter` on re-runs infer_motion_traj with the current latents (:352-392): the infiller's output
        becomes `smpl_pose`, the trajectory predictor's local trajectory the new `traj_local_pred`, and SMPL gives new joints; the loss
        reaches `motion_latent` through the reprojection term -> joints -> SMPL (body pose) -> infiller (all windows, autoregressively).
        `traj_latent` is in the parameter list but never receives a gradient: get_pred_trajectory_base detaches traj_local_pred (:396), and
        torch.optim.Adam skips a parameter whose grad is None -- its value stays, exactly as in the reference.
        Per iteration over the C ABI: taped infiller (glamr_nets_infill_taped), trajectory predictor (glamr_nets_infer), joints-only
        skinning, one gradient launch of the stage kernel (niters 1, lr 0, grads_out, g_j_local), glamr_smpl_backward, glamr_nets_infill_backward,
        glamr_adam_step_indexed on the scene parameters and on the latents (torch.optim.Adam's arithmetic; a parameter's step count advances only
        when it has a gradient: two indices on the device).  The first two iterations of a stage are plain launches; the third is CAPTURED as a HIP
        graph and the rest of the stage replays it (GLAMR_LATENT_GRAPH=0: plain launches throughout; self.latent_graph_replays counts)."""
        import ctypes
        from ... import parallel
        dev, L = self.device, _lib.lib()
        S, P, T = packed.S, packed.P, packed.T
        n_slots = S * P
        meps, teps = packed.latents
        meps, teps = meps.clone(), teps.clone()
        pa = packed.person_arrays
        h = self.mt_model.handle
        tape_gb = L.glamr_nets_tape_bytes(h.h, n_slots, T) / 2.0 ** 30
        if tape_gb > 96:
            raise ValueError('latent-optimisation mode keeps every activation of the infiller for its backward: %.0f GB for %d person slots of %d frames; '
                             'run it on smaller batches (the reference runs it on one sequence at a time)' % (tape_gb, n_slots, T))
        lens = np.ascontiguousarray(rin.lens, dtype=np.int32)
        fr_start = packed.t['fr_start'].cpu().numpy()
        occupied = rin.seq_len_slot.cpu().numpy() > 0                    # (person slots a scene with fewer persons leaves empty are skipped)
        # frame rows of the priors' outputs (row e of slot k = video frame fr_start[k] + e) <-> the per-slot video-frame arrays: ONE gather /
        # scatter index for the whole batch instead of a python loop over the slots, twice per iteration
        src, dst = [], []
        for k in range(n_slots):
            if occupied[k]:
                nk, fs = int(lens[k]), int(fr_start[k])
                src.append(k * T + np.arange(nk))
                dst.append(k * T + fs + np.arange(nk))
        src = torch.as_tensor(np.concatenate(src) if src else np.zeros(0, np.int64), device=dev)
        dst = torch.as_tensor(np.concatenate(dst) if dst else np.zeros(0, np.int64), device=dev)
        smpl_h = self.smpl._handle(dev)
        zeros3 = torch.zeros((n_slots * T, 3), device=dev)
        packed.t['g_j_local'] = torch.zeros((n_slots, T, packing.NJ, 3), device=dev)
        params = packed.t['params']
        m_lat, v_lat = torch.zeros_like(meps), torch.zeros_like(meps)
        m, v = torch.zeros_like(params), torch.zeros_like(params)
        step_idx = torch.zeros(2, dtype=torch.int32, device=dev)         # [0] scene parameters, [1] latents: 0-based row of the coefficient table
        has_wd = False
        use_graph = os.environ.get('GLAMR_LATENT_GRAPH', '1') != '0'
        self.latent_graph_replays = 0

        def iteration(spec, with_priors, first, coef):
            """One Adam iteration (:547-570 in latent mode), launches only -- nothing here reads a value back or depends on the iteration
            number except through `step_idx` on the device, so the same launch sequence is captured ONCE per stage and replayed."""
            tape = None
            if with_priors:
                # infer_motion_traj with the current latents (:352-392)
                pose_out, tape = h.infill_taped(pa['nets_pose'], pa['nets_vis'], lens, meps)
                tr = h.infer(pose_out, None, lens, traj_eps=teps, infill=False, traj=True)
                pa['smpl_pose'].view(-1, 69).index_copy_(0, dst, pose_out.view(-1, 69).index_select(0, src))
                packed.t['traj_local_pred'].view(-1, 11).index_copy_(0, src, tr['local_traj'].view(-1, 11).index_select(0, src))
                with torch.no_grad():
                    jl = self.smpl(global_orient=zeros3, body_pose=pa['smpl_pose'].view(-1, 69), betas=pa['smpl_beta'].view(-1, 10), root_trans=zeros3,
                                   return_verts=False).joints
                if packed.t['j_local'].shape == (n_slots, T, packing.NJ, 3):
                    packed.t['j_local'].copy_(jl.view(n_slots, T, packing.NJ, 3))          # (a fixed address: the gradient launch below is captured with it)
                else:
                    packed.t['j_local'] = jl.view(n_slots, T, packing.NJ, 3).clone()
            sd = packing.stage_desc(spec, self.specs, has_wd, niters=1)
            sd.lr = 0.0
            if not first:
                sd.flags |= packing.FLAG_KEEP_CAM_PARAMS
            grads = parallel._device_run_stage(packed, sd, True)
            g_lat = None
            if tape is not None and self.flag_opt_motion_latent:
                # dL/d j_local -> body pose (skinning, blend shapes, chain, re-anchoring in reverse) -> latents (all windows)
                pose72 = torch.cat([zeros3, pa['smpl_pose'].view(-1, 69)], dim=1).contiguous()
                g_pose = torch.empty((n_slots * T, 72), device=dev)
                ws = torch.empty(L.glamr_smpl_backward_workspace_bytes(smpl_h, n_slots * T, 0), dtype=torch.uint8, device=dev)
                _lib.check(L.glamr_smpl_backward(smpl_h, n_slots * T, _lib.ptr(pose72), _lib.ptr(pa['smpl_beta'].view(-1, 10)), _lib.ptr(zeros3), None, None, None,
                                                 None, _lib.ptr(packed.t['g_j_local']), _lib.ptr(g_pose), None, None, None, 0, _lib.ptr(ws), _lib.current_stream()))
                g_out = torch.zeros((n_slots * T, 69), device=dev)
                g_out.index_copy_(0, src, g_pose[:, 3:].index_select(0, dst))
                g_lat = h.infill_backward(tape, g_out.view(n_slots, T, 69))
                # (a parameter's step count advances only when it has a gradient: the latents have their own index)
                _lib.check(L.glamr_adam_step_indexed(meps.numel(), _lib.ptr(meps), _lib.ptr(m_lat), _lib.ptr(v_lat), _lib.ptr(g_lat), _lib.ptr(coef), _lib.ptr(step_idx[1:]),
                                                     _lib.current_stream()))
                _lib.check(L.glamr_counter_add(_lib.ptr(step_idx[1:]), 1, _lib.current_stream()))
            _lib.check(L.glamr_adam_step_indexed(params.numel(), _lib.ptr(params), _lib.ptr(m), _lib.ptr(v), _lib.ptr(grads), _lib.ptr(coef), _lib.ptr(step_idx),
                                                 _lib.current_stream()))
            _lib.check(L.glamr_counter_add(_lib.ptr(step_idx), 1, _lib.current_stream()))
            return g_lat

        for stage, spec in self.opt_stage_specs.items():
            n = spec['opt_niters'] if max_iters is None else min(max_iters, spec['opt_niters'])
            start = spec.get('opt_latent_start_iter', 0)                 # optimize() :581
            # init_opt creates a fresh optimiser per stage (:635-644): zero moments, step counts back to the first row of the stage's table
            m.zero_(); v.zero_(); m_lat.zero_(); v_lat.zero_(); step_idx.zero_()
            tab = np.empty(2 * max(n, 1), np.float32)
            _lib.check(L.glamr_adam_coef_table(float(spec['opt_lr']), max(n, 1), tab.ctypes.data_as(ctypes.c_void_p)))
            coef = torch.as_tensor(tab, device=dev)
            graph = None
            for it in range(n):
                with_priors = it >= start
                if graph is not None:
                    graph.replay()
                    self.latent_graph_replays += 1
                    continue
                g_lat = iteration(spec, with_priors, it == 0, coef)
                if g_lat is not None and getattr(self, 'latent_trace', None) is not None and not self.latent_trace:      # first gradient of the run, for the parity tests
                    self.latent_trace.update(g_motion_latent=g_lat.detach().cpu().numpy(), losses=packed.t['losses'].detach().cpu().numpy(),
                                             smpl_pose=pa['smpl_pose'].detach().cpu().numpy(), traj_local_pred=packed.t['traj_local_pred'].detach().cpu().numpy())
                # from here on every iteration of the stage is the same launch sequence: capture it once, replay it n - it - 2 times
                if use_graph and with_priors and it >= 1 and n - it - 1 >= 2 and not torch.cuda.is_current_stream_capturing():
                    try:
                        g = torch.cuda.CUDAGraph()
                        cur = torch.cuda.current_stream(dev)
                        side = self.__dict__.setdefault('_latent_capture_stream', torch.cuda.Stream(device=dev))
                        side.wait_stream(cur)
                        with torch.cuda.graph(g, stream=side):
                            iteration(spec, True, False, coef)
                        cur.wait_stream(side)
                        graph = g
                    except Exception as e:      # noqa: BLE001 -- the plain launches are always available
                        import sys
                        sys.stderr.write('latent-optimisation mode: iteration graph not used (%s); plain launches\n' % e)
                        torch.cuda.synchronize(dev)
                        graph, use_graph = None, False
            has_wd = has_wd or 'world_dheading' in spec['opt_variables']
            if spec.get('reinitialize_cam', False):
                packed.t['cam_pose'][:] = packed.t['cam_pose'][:, :1]
            del graph
        packed.has_world_dheading = has_wd
        packed.stage_ws = []
        packed.latents = (meps, teps)
        packed.t['g_j_local'] = None
        return packed

    def collect(self, datas, packed, fetched=None):
        """Device arrays -> the reference's output dictionaries (numpy): one device->host copy per array.  Waits for this batch only."""
        rin = packed.keepalive[0] if getattr(packed, 'keepalive', None) else None
        if datas and datas[0].get('_pending'):
            if fetched is None:
                fetched = self._fetch_async(packed)
            if rin is not None:
                self.check_inputs(rin)
            t0 = time.time()
            self._materialise(datas, packed, self._all_vars(), fetched)
        else:
            torch.cuda.synchronize(self.device)
            t0 = time.time()
            all_vars = self._all_vars()
            packed.unpack_into(datas, {'opt_variables': all_vars} if self.opt_stage_specs else None, self.specs, as_torch=False)
            self.last_losses = packed.t['losses'].cpu().numpy()
        if self.latent_mode and getattr(packed, 'latents', None) is not None and rin is not None:
            # the optimised draws, per person as the reference keeps them in pose_dict (:155-158)
            meps, teps = (x.detach().cpu().numpy() for x in packed.latents)
            for si, d in enumerate(datas):
                for pi, idx in enumerate(packed.person_ids[si]):
                    k = si * packed.P + pi
                    d['person_data'][idx]['motion_latent'] = meps[k, :num_windows(int(rin.lens[k]))].copy()
                    d['person_data'][idx]['traj_latent'] = teps[k][None].copy()
        if getattr(self, 'kernel_ms', None) is not None and getattr(packed, 'stage_ws', None):
            self.kernel_ms.extend(self.launch_ms(ws) for ws in packed.stage_ws)
        self.timings['unpack'] = time.time() - t0
        return datas

    def _all_vars(self):
        return sorted(set(v for s in self.opt_stage_specs.values() for v in s['opt_variables']))

    def optimize_stream(self, batches, latents=None, max_iters=None):
        """Host dictionaries in, host dictionaries out for a STREAM of batches (an iterable of lists of in_dicts): yields one list of
        result dictionaries per batch, in order.  Software pipeline on the host thread: while the device runs batch i (asynchronous
        launches on a compute stream), the host scatters and uploads batch i + 1 (copy stream, pinned staging) and cuts the output
        dictionaries of batch i - 1 from its device->host copies (copy stream, after that batch's own event -- never a device-wide wait)."""
        dev = self.device
        # consecutive batches alternate over two compute streams (the launch seams and tails of one batch are covered by the next, as in bench.py)
        computes = self.__dict__.setdefault('_compute_streams', [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)])
        # Uploads and downloads share ONE third stream.  Not more: the runtime multiplexes HIP streams onto 4 hardware queues, and with a
        # separate upload stream (5 streams with the default one) two of them shared a queue -- a batch then started only when the previous one
        # had finished (tools/stream_timeline.py; raising GPU_MAX_HW_QUEUES to 8, three compute streams and the default stream as the uploader
        # were all measured slower, round 4).  Rounds 2-3 put a batch's upload on ITS OWN compute stream: there it queues behind the batch
        # before last, so every batch began with 7 ms of PCIe and nothing else (50.5 ms from batch to batch against 43.4 resident).  On the
        # copy stream the order of the enqueues is what matters: a download waits for its batch's event, and an upload enqueued behind it
        # waits too -- _stream_loop enqueues a batch's download only when the upload of the batch two later is already in the queue.
        down = self.__dict__.setdefault('_download_stream', torch.cuda.Stream(device=dev))
        it = iter(batches)
        lat = iter(latents) if latents is not None else None
        turn = [0]
        # batches alternate between two streams: stagger them so that a batch's infiller runs beside the previous batch's stage (PipelineGate)
        own_gate = self.pipeline_gate is None and not self.latent_mode and coschedule_enabled()
        if own_gate:
            self.pipeline_gate = PipelineGate()
        # The interpreter's cyclic collector: a full collection walks every container object alive -- with torch imported, 30-45 ms -- and the
        # ~10 000 containers a batch of output dictionaries is made of trigger one every 5-6 batches (measured: every fifth yield 85-110 ms late,
        # tools/stream_yield_probe.py).  While the stream runs, everything alive at its start sits in the permanent generation (gc.freeze):
        # collections still run, over the objects created since.  GLAMR_STREAM_GC_FREEZE=0 leaves the collector alone.
        import gc
        # (a caller that keeps a frozen set of its own -- gc.get_freeze_count() > 0 -- manages the collector itself: gc.unfreeze() would release ITS set too)
        frozen = gc.isenabled() and gc.get_freeze_count() == 0 and os.environ.get('GLAMR_STREAM_GC_FREEZE', '1') != '0'
        if frozen:
            # (no gc.collect() first: a full collection at this point walks every object alive -- 67 ms with torch imported, tools/stream_profile.py,
            # at the head of every stream, before its first batch is even staged; whatever garbage is frozen along is collected after gc.unfreeze())
            gc.freeze()
        try:
            yield from self._stream_loop(it, lat, turn, computes, down, max_iters)
        finally:
            if frozen:
                gc.unfreeze()
            if own_gate:
                self.pipeline_gate = None

    def _stream_loop(self, it, lat, turn, computes, down, max_iters):
        """Host iteration k: enqueue batch k's device work -> scatter + upload batch k + 1 -> enqueue the download of batch k - 2 -> cut batch
        k - 2's dictionaries (the only wait: until that download has landed).  Three batches are in flight, so a batch's launches are queued
        BEFORE its compute stream comes free (the batch before last still runs there) and its inputs were uploaded a batch time earlier."""
        def stage(batch):
            with torch.cuda.stream(down):
                return self.stage_inputs(batch, next(lat) if lat is not None else None)

        def fetch(entry):
            datas, packed, done = entry
            down.wait_event(done)
            return datas, packed, self._fetch_async(packed, down)
        nxt = next(it, None)
        if nxt is None:
            return
        rin = stage(nxt)
        flight = []                                                     # batches whose download has not been enqueued yet, oldest first
        while rin is not None:
            compute = computes[turn[0] % len(computes)]
            turn[0] += 1
            compute.wait_event(rin.upload_done)
            with torch.cuda.stream(compute):
                datas, packed = self.optimize_resident(rin, max_iters) if self.latent_mode else self._resident_for_stream(rin, max_iters)
                done = torch.cuda.Event()
                done.record()
            flight.append((datas, packed, done))
            nxt = next(it, None)
            rin = stage(nxt) if nxt is not None else None              # host work under the device's; the copy stream is idle now
            if len(flight) > 2:
                yield self.collect(*fetch(flight.pop(0)))              # (enqueued only now: a download waiting for its batch must not sit in front of that upload)
        while flight:
            yield self.collect(*fetch(flight.pop(0)))

    def _resident_for_stream(self, rin, max_iters):
        datas, packed = self.init_resident(rin, init_forward=not self._schedule_overwrites_init())
        self.run_schedule(packed, max_iters)
        return datas, packed

    def optimize_batch(self, in_dicts, latents=None, max_iters=None):
        """Host dictionaries in, host dictionaries out (optimize() of the reference for a batch of independent sequences)."""
        if self.latent_mode:
            if self.cam_fix_frames != [(0, None)]:
                raise NotImplementedError('latent-optimisation mode with non-default cam_fix_frames')
            rin = self.stage_inputs(in_dicts, latents)
            datas, packed = self.optimize_resident(rin, max_iters)
            return self.collect(datas, packed)
        datas, packed = self.init_data_batch(in_dicts, latents, init_forward=not self._schedule_overwrites_init())
        t0 = time.time()
        self.run_schedule(packed, max_iters)
        torch.cuda.synchronize(self.device)
        self.timings['optimise'] = time.time() - t0
        return self.collect(datas, packed)

    @staticmethod
    def _to_numpy(x):
        if isinstance(x, torch.Tensor):
            return x.detach().cpu().numpy()
        if isinstance(x, dict):
            return {k: GlobalReconOptimizer._to_numpy(v) for k, v in x.items()}
        if isinstance(x, list):
            return [GlobalReconOptimizer._to_numpy(v) for v in x]
        return x

    # -- reference entry points ---------------------------------------------------------------------------------------------------
    def init_data(self, in_dict, latents=None):
        datas, packed = self.init_data_batch([in_dict], None if latents is None else [latents])
        if datas[0].get('_pending'):
            self._materialise(datas, packed, [])
        return datas[0]

    def optimize(self, in_dict, continue_opt=False, latents=None, max_iters=None):
        if continue_opt:
            return self.continue_batch([in_dict], max_iters)[0]
        return self.optimize_batch([in_dict], None if latents is None else [latents], max_iters)[0]

    def continue_batch(self, datas, max_iters=None):
        """optimize(in_dict, continue_opt=True) (:572-573): `in_dict` is the dictionary a previous optimize() returned; the schedule
        runs again from its variables.  The inputs are not modified."""
        import copy
        datas = [dict(d, person_data={i: dict(pd) for i, pd in d['person_data'].items()}) for d in datas]
        j_locals = []
        for d in datas:
            jl = {}
            for idx, pd in d['person_data'].items():
                pose = torch.as_tensor(np.asarray(pd['smpl_pose']), dtype=torch.float32, device=self.device)
                beta = torch.as_tensor(np.asarray(pd['smpl_beta']), dtype=torch.float32, device=self.device)
                z = torch.zeros(pose.shape[0], 3, device=self.device)
                with torch.no_grad():
                    jl[idx] = self.smpl(global_orient=z, body_pose=pose, betas=beta, root_trans=z, return_verts=False).joints.cpu()
            j_locals.append(jl)
        packed = packing.PackedScenes(datas, j_locals, self.device, self.cam_fix_frames)
        has_wd = any('world_dheading' in pd for d in datas for pd in d['person_data'].values())
        self.run_schedule(packed, max_iters, has_wd=has_wd)
        return self.collect(datas, packed)
