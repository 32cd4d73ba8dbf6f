"""Optimisation schedules of the global-reconstruction stage.

The reference keeps them as YAML files (global_recon/cfg/glamr_{dynamic,static,dynamic_multi,static_multi,3dpw,h36m}.yml,
parsed by global_recon/utils/config.py:12-46).  The same schedules are generated here from a compact table so the package is
self-contained; `load_yaml()` accepts a reference YAML unchanged for users who bring their own.  The resulting dictionary has
the YAML's structure: {'grecon_model_specs': {...flags...}, 'opt_stage_specs': {stage: {opt_lr, opt_niters, opt_variables,
loss_cfg: {name: {weight, ...}}}}}.
"""
import copy


def _losses(rot_reg, cam_rot_sm, cam_orig_sm, cam_up, first_frame_only=False, full=True):
    ffo = {'first_frame_only': True} if first_frame_only else {}
    cfg = {
        'rel_transform': dict(trans_weight=0.0, weight=200, **ffo),
        'kp_2d': dict(weight=1.0, min_conf=0.3, **ffo),
        'kp_2d_dist': dict(weight=1.0, min_conf=0.3, monitor_only=True, **ffo),
        'cam_traj_rot': dict(rot_type='6d', weight=1.e+5, **ffo),
    }
    if full:
        cfg.update({
            'traj_rot_smoothness': dict(weight=1.e+3),
            'local_traj_dxy_reg': dict(weight=3.e+2),
            'local_traj_dheading_reg_new': dict(weight=3.e+3),
            'local_traj_rot_reg': dict(weight=rot_reg),
            'local_traj_z_reg': dict(weight=1.e+2),
            'cam_inv_trans_residual_reg': dict(weight=1.e+2),
            'cam_inv_rot_smoothness': dict(weight=cam_rot_sm),
            'cam_origin_smoothness': dict(weight=cam_orig_sm),
            'cam_up_reg': dict(weight=cam_up),
        })
    return cfg


def _specs(dataset, **flags):
    s = dict(motion_traj_cfg='joint_motion_traj_demo', est_type='hybrik', flag_infer_motion_traj=True,
             flag_pred_traj=True, flag_opt_traj=True, flag_opt_cam=True)
    s.update(flags)
    return {'dataset': dataset, 'grecon_model_name': 'global_recon_model', 'grecon_model_specs': s}


_DYN_VARS = ['cam', 'local_xy', 'local_heading', 'world_dheading', 'local_rot']
_STA_VARS = ['cam', 'local_xy', 'local_heading', 'world_dheading', 'local_dxy', 'local_rot', 'local_z']


def _stage(lr, niters, variables, losses):
    return dict(opt_lr=lr, opt_niters=niters, opt_variables=list(variables), loss_cfg=losses)


def _build():
    C = {}
    c = _specs('demo', flag_fixed_cam=False, flag_init_cam_all_frames=True)
    c['opt_stage_specs'] = {'init_opt': _stage(1.e-3, 500, _DYN_VARS, _losses(5.e+3, 1.e+1, 1.e+3, 1.e+6))}
    C['glamr_dynamic'] = c
    c = _specs('demo', flag_fixed_cam=True)
    c['opt_stage_specs'] = {'init_opt': _stage(1.e-3, 500, _STA_VARS, _losses(5.e+3, 1.e+3, 1.e+3, 1.e+2))}
    C['glamr_static'] = c
    first = _stage(1.e-1, 200, ['local_xy', 'local_heading'], _losses(0, 0, 0, 0, first_frame_only=True, full=False))
    c = _specs('demo', flag_fixed_cam=True)
    c['opt_stage_specs'] = {'init_opt': copy.deepcopy(first),
                            'main_opt': _stage(1.e-4, 500, _STA_VARS, _losses(5.e+3, 1.e+3, 1.e+3, 1.e+2))}
    C['glamr_static_multi'] = c
    c = _specs('demo', flag_fixed_cam=False, flag_init_cam_all_frames=True)
    c['opt_stage_specs'] = {'init_opt': copy.deepcopy(first),
                            'main_opt': _stage(1.e-4, 500, _DYN_VARS, _losses(5.e+3, 1.e+1, 1.e+3, 1.e+6))}
    C['glamr_dynamic_multi'] = c
    c = _specs('3dpw', flag_fixed_cam=False, flag_init_cam_all_frames=False, flag_opt_cam_from_person_pose=True)
    c['opt_stage_specs'] = {
        'init_opt': _stage(1.e-2, 200, ['local_xy', 'local_heading'], _losses(5.e+2, 1.e+1, 1.e+2, 1.e+5)),
        'main_opt': _stage(1.e-4, 500, ['local_xy', 'local_heading', 'local_dheading', 'local_dxy', 'local_rot'],
                           _losses(5.e+2, 1.e+1, 1.e+2, 1.e+5))}
    C['glamr_3dpw'] = c
    c = _specs('h36m', flag_fixed_cam=False, flag_init_cam_all_frames=False)
    c['opt_stage_specs'] = {
        'init_opt': _stage(1.e-2, 200, ['cam', 'local_xy', 'local_heading'], _losses(5.e+2, 1.e+4, 1.e+4, 1.e+5)),
        'main_opt': _stage(1.e-4, 500, ['cam', 'local_xy', 'local_heading', 'world_dheading', 'local_dxy', 'local_rot'],
                           _losses(5.e+2, 1.e+4, 1.e+4, 1.e+5))}
    C['glamr_h36m'] = c
    return C


CONFIGS = _build()


def get_config(cfg_id):
    cfg = copy.deepcopy(CONFIGS[cfg_id])
    cfg.setdefault('id', cfg_id)          # Config.id of the reference (global_recon/utils/config.py): the name the per-iteration log line starts with
    return cfg


def load_yaml(path):
    import yaml
    with open(path, 'r') as f:
        return yaml.safe_load(f)
