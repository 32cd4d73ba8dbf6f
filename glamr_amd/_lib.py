"""ctypes binding of libglamr_hip.so (include/glamr_hip.h).  There is NO fallback: if the library is missing or a call
fails, a RuntimeError is raised -- the product path never computes on the CPU."""
import ctypes
import os
from ctypes import c_int, c_int32, c_int64, c_size_t, c_uint32, c_float, c_double, c_void_p, c_char_p, POINTER, Structure

_LIB = None
LIB_PATH = os.environ.get('GLAMR_LIB_PATH') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libglamr_hip.so')      # (override: A/B runs of kernel variants, tools/README.md)
NUM_LOSSES = 13
LOSS_CAMERA_ONLY = (9, 13)    # [first, last) of the camera-only terms: CAM_INV_TRANS_RES_REG, CAM_INV_ROT_SMOOTHNESS, CAM_ORIGIN_SMOOTHNESS, CAM_UP_REG
LOSS_KP_2D_DIST = 1          # index of the monitor-only keypoint distance in the loss record (GLAMR_LOSS_KP_2D_DIST, include/glamr_hip.h)


class TensorDesc(Structure):
    _fields_ = [('offset', c_int64), ('rows', c_int32), ('cols', c_int32)]


class StageDesc(Structure):
    _fields_ = [('var_mask', c_uint32), ('flags', c_uint32), ('loss_mask', c_uint32), ('monitor_mask', c_uint32),
                ('first_frame_only_mask', c_uint32), ('niters', c_int32), ('lr', c_double), ('loss_weight', c_float * 16),
                ('kp_min_conf', c_float), ('first_frame_weight', c_float * 16), ('rel_trans_weight', c_float)]


class SceneBatch(Structure):
    _fields_ = [('n_scenes', c_int32), ('max_persons', c_int32), ('max_len', c_int32), ('n_joints', c_int32)] + \
               [(n, c_void_p) for n in ('n_persons', 'seq_len', 'fr_start', 'fr_end', 'vis', 'j_local', 'kp_2d', 'kp_score', 'cam_K',
                                        'traj_local_pred', 'orient_cam', 'base_orient', 'base_trans', 'person2cam', 'dheading_mask',
                                        'rel_transform_cam', 'cam_pose', 'params', 'losses', 'orient_world', 'trans_world',
                                        'kp_2d_pred', 'orient_cam_in_world', 'frozen', 'g_j_local', 'loss_history')]


class FilterOpts(Structure):
    _fields_ = [('filter_pose', c_int32), ('make_invis_with_keypoint', c_int32), ('keypoint_min_score', c_float), ('keypoint_min_num', c_int32)]


class RawBatch(Structure):
    _fields_ = [('n_slots', c_int32), ('max_len', c_int32)] + [(n, c_void_p) for n in ('seq_len', 'exist', 'rotmats', 'betas', 'root_trans', 'kp_2d')]


class HostStaging(Structure):
    _fields_ = [(n, c_void_p) for n in ('exist', 'rot', 'betas', 'trans', 'K', 'kp')]


class PersonArrays(Structure):
    _fields_ = [(n, c_void_p) for n in ('visible_orig', 'smpl_pose', 'smpl_beta', 'trans_cam', 'nets_pose', 'nets_vis')]


class InfillerIO(Structure):
    _fields_ = [(n, c_void_p) for n in ('in_body_pose', 'body_pose', 'frame_mask', 'eps', 'context', 'q_z', 'p_z', 'z', 'out_body_pose')]


class TrajIO(Structure):
    _fields_ = [(n, c_void_p) for n in ('in_body_pose', 'in_joint_pos', 'trans', 'orient', 'eps', 'init_row')] + [('valid_len', c_int32)] + \
               [(n, c_void_p) for n in ('local_traj', 'q_z', 'p_z', 'z', 'out_orig_local_traj', 'out_local_traj', 'out_trans', 'out_orient', 'out_orient_q')]


class ParamLayout(Structure):
    _fields_ = [(n, c_int32) for n in ('scene_stride', 'cam_rot6d', 'cam_trans', 'cam_inv_rot_res', 'cam_inv_trans_res',
                                       'person_stride', 'person0', 'local_xy', 'local_dxy', 'local_heading', 'local_dheading',
                                       'local_z', 'local_rot', 'world_dheading')]


_SIGNATURES = {
    'glamr_version': (c_int, []),
    'glamr_last_error': (c_char_p, []),
    'glamr_device_info': (c_int, [POINTER(c_int), POINTER(c_int), POINTER(c_size_t)]),
    'glamr_smpl_create': (c_int, [POINTER(c_void_p), c_int, c_int] + [c_void_p] * 6 + [c_int, c_void_p, c_void_p, c_int, c_void_p, c_int]),
    'glamr_smpl_destroy': (c_int, [c_void_p]),
    'glamr_smpl_workspace_bytes': (c_size_t, [c_void_p, c_int]),
    'glamr_smpl_forward': (c_int, [c_void_p, c_int] + [c_void_p] * 6 + [c_int, c_void_p, c_void_p]),
    'glamr_smpl_fk': (c_int, [c_void_p, c_int] + [c_void_p] * 5),
    'glamr_smpl_backward_root': (c_int, [c_void_p, c_int] + [c_void_p] * 10 + [c_int, c_void_p]),
    'glamr_smpl_backward_workspace_bytes': (c_size_t, [c_void_p, c_int, c_int]),
    'glamr_smpl_backward': (c_int, [c_void_p, c_int] + [c_void_p] * 12 + [c_int, c_void_p, c_void_p]),
    'glamr_nets_create': (c_int, [POINTER(c_void_p), c_void_p, POINTER(TensorDesc), c_int, c_void_p, POINTER(TensorDesc), c_int, c_void_p, c_void_p]),
    'glamr_nets_destroy': (c_int, [c_void_p]),
    'glamr_nets_workspace_bytes': (c_size_t, [c_void_p, c_int, c_int]),
    'glamr_nets_precision': (c_int, [c_void_p, c_void_p]),
    'glamr_nets_tape_bytes': (c_size_t, [c_void_p, c_int, c_int]),
    'glamr_nets_infill_taped': (c_int, [c_void_p, c_int, c_int] + [c_void_p] * 4 + [c_int] + [c_void_p] * 3),
    'glamr_nets_infill_backward': (c_int, [c_void_p, c_int, c_int] + [c_void_p] * 2 + [c_int] + [c_void_p] * 4),
    'glamr_nets_infer': (c_int, [c_void_p, c_int, c_int] + [c_void_p] * 4 + [c_int] + [c_void_p] * 5 + [c_int, c_void_p, c_void_p]),
    'glamr_nets_infiller_window': (c_int, [c_void_p, c_int, c_int, POINTER(InfillerIO), c_void_p, c_void_p]),
    'glamr_nets_traj_clip': (c_int, [c_void_p, c_int, c_int, c_int, POINTER(TrajIO), c_void_p, c_void_p]),
    'glamr_traj_local_to_global_workspace_bytes': (c_size_t, [c_int, c_int]),
    'glamr_traj_local_to_global': (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'glamr_host_scatter': (c_int, [c_int, c_void_p, c_int, POINTER(HostStaging), c_void_p, c_void_p, c_int]),
    'glamr_init_workspace_bytes': (c_size_t, [c_int, c_int]),
    'glamr_init_prepare': (c_int, [POINTER(RawBatch), POINTER(SceneBatch), POINTER(PersonArrays), POINTER(FilterOpts), c_void_p, c_void_p]),
    'glamr_check_inputs': (c_int, [POINTER(RawBatch), c_void_p, c_void_p, c_void_p]),
    'glamr_init_scenes': (c_int, [POINTER(SceneBatch), POINTER(PersonArrays)] + [c_void_p] * 6),
    'glamr_init_scenes_ex': (c_int, [POINTER(SceneBatch), POINTER(PersonArrays)] + [c_void_p] * 4 + [c_int, c_void_p, c_void_p]),
    'glamr_init_scatter_pose': (c_int, [POINTER(SceneBatch), POINTER(PersonArrays), c_void_p, c_void_p]),
    'glamr_init_cam_all_frames': (c_int, [POINTER(SceneBatch), c_void_p]),
    'glamr_grecon_param_layout': (c_int, [c_int, c_int, POINTER(ParamLayout)]),
    'glamr_grecon_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'glamr_grecon_run_stage': (c_int, [POINTER(SceneBatch), POINTER(StageDesc), c_void_p, c_void_p, c_void_p]),
    'glamr_grecon_last_launch_ns': (c_int, [c_void_p, POINTER(ctypes.c_double)]),
    'glamr_adam_step': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_int, c_void_p]),
    'glamr_adam_coef_table': (c_int, [c_double, c_int, c_void_p]),
    'glamr_adam_step_indexed': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'glamr_counter_add': (c_int, [c_void_p, c_int, c_void_p]),
    'glamr_eval_regress_joints': (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'glamr_eval_procrustes': (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'glamr_eval_heading_align': (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
}


def exported_symbols():
    """Names every entry point include/glamr_hip.h declares (checked against the header in tests/test_abi.py)."""
    return sorted(_SIGNATURES)


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError('libglamr_hip.so is not built (%s missing): run `python -m glamr_amd.build` or '
                               '`python -c "import __graft_entry__ as g; g.build()"`.  There is no CPU fallback.' % LIB_PATH)
        # torch ships its own libamdhip64; it must be mapped first so this library binds to the SAME HIP runtime instance
        # (two runtimes in one process cannot both own the device)
        import torch  # noqa: F401
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        _LIB = handle
    return _LIB


def check(rc):
    if rc != 0:
        raise RuntimeError('libglamr_hip: %s (code %d)' % (lib().glamr_last_error().decode('utf-8', 'replace'), rc))


def ptr(t):
    """Device (or host) address of a contiguous torch tensor / numpy array, or None."""
    if t is None:
        return None
    if hasattr(t, 'data_ptr'):
        assert t.is_contiguous(), 'libglamr_hip needs contiguous tensors'
        return c_void_p(t.data_ptr())
    assert t.flags['C_CONTIGUOUS']
    return c_void_p(t.ctypes.data)


def current_stream():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)
