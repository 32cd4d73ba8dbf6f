/* libglamr_hip.so -- C ABI of the MI355X-native GLAMR global-reconstruction hot path.
 *
 * The reference (NVlabs/GLAMR) has no FFI layer: its hot path is Python calling PyTorch.  The entry points
 * below are what a binding for that path binds instead; each names the reference interface it replaces (paths relative to
 * the reference tree).  Conventions:
 *   - every function returns 0 on success or a negative GLAMR_E_* code; glamr_last_error() gives the thread-local message;
 *     nothing throws across the ABI and nothing calls exit();
 *   - pointers marked `dev` are device (HBM) addresses owned by the caller (the PyTorch caching allocator in the Python
 *     host); pointers marked `host` are host addresses read during the call only;
 *   - all work is enqueued on the caller's stream (`hipStream_t` passed as void*), no hidden synchronisation unless the
 *     function says so; handles are re-entrant per handle, one process per GPU;
 *   - fp32 everywhere; quaternions (w,x,y,z); matrices row-major.
 */
#ifndef GLAMR_HIP_H
#define GLAMR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GLAMR_OK 0
#define GLAMR_E_INVALID (-1)   /* bad argument */
#define GLAMR_E_HIP (-2)       /* HIP runtime error (message has the hipError string) */
#define GLAMR_E_NOMEM (-3)
#define GLAMR_E_UNSUPPORTED (-4)

int glamr_version(void);                 /* 100*major + minor */
const char* glamr_last_error(void);      /* thread-local, valid until the next failing call on this thread */
int glamr_device_info(int* cu_count, int* gfx_major_minor, size_t* hbm_bytes);

/* ---------------------------------------------------------------------------------------------------------------------
 * SMPL body model -- replaces lib/models/smpl.py:274-343 (`SMPL.__init__/forward/get_joints`) and the third-party
 * smplx.lbs.lbs / batch_rodrigues / batch_rigid_transform / vertices2joints it calls (lib/models/smpl.py:7-8,295,299).
 * ------------------------------------------------------------------------------------------------------------------- */
typedef struct glamr_smpl glamr_smpl;

/* Uploads and re-tiles the model constants.  All arrays `host`, fp32 / int32:
 *   v_template (V,3)  shapedirs (V,3,num_betas)  posedirs (207, V*3) [smplx layout: pose-feature major]
 *   J_regressor (24,V)  lbs_weights (V,24)  J_regressor_extra (n_extra,V)  parents (24; parents[0] = -1)
 *   extra_vertex_ids (n_picked)  -- vertices smplx appends as joints 24..24+n_picked-1
 *   joint_map (n_out) -- indices into [24 chain | n_picked vertices | n_extra regressed]  (smpl.py:284-287,300-301) */
int glamr_smpl_create(glamr_smpl** out, int V, int num_betas, const float* v_template, const float* shapedirs,
                      const float* posedirs, const float* J_regressor, const float* lbs_weights,
                      const float* J_regressor_extra, int n_extra, const int32_t* parents,
                      const int32_t* extra_vertex_ids, int n_picked, const int32_t* joint_map, int n_out);
int glamr_smpl_destroy(glamr_smpl* h);
/* Bytes of device scratch glamr_smpl_forward needs for a batch of B frames (caller allocates, any 256-B aligned buffer). */
size_t glamr_smpl_workspace_bytes(const glamr_smpl* h, int B);

#define GLAMR_SMPL_ORIG_JOINTS 1   /* joints = the 24 chain joints (smpl.py:296-297), n_out is ignored */
#define GLAMR_SMPL_BODY_POSE_ONLY 2 /* glamr_smpl_forward: `pose` is (B,69), the body pose alone; the global orientation is zero (the cached
                                    * root-relative joints of the optimiser, global_recon_model.py:517-524 with SURVEY.md 8 row a9) */

/* SMPL.forward (smpl.py:289-316).  dev in: pose (B,72) = [global_orient | body_pose] axis-angle, betas (B,num_betas),
 * root_trans (B,3) or NULL (no re-anchoring), root_scale (B) or NULL (=1).  dev out: verts (B,V,3) or NULL (skips the
 * vertex write-out entirely), joints (B,n_out,3).  workspace: glamr_smpl_workspace_bytes(h,B) bytes. */
int glamr_smpl_forward(glamr_smpl* h, int B, const float* pose, const float* betas, const float* root_trans,
                       const float* root_scale, float* verts, float* joints, int flags, void* workspace, void* stream);

/* SMPL.get_joints (smpl.py:318-343): forward kinematics of the 24 chain joints from the UNSHAPED template.
 * dev in: pose (B,72); root_trans/root_scale as above.  dev out: joints (B,24,3). */
int glamr_smpl_fk(glamr_smpl* h, int B, const float* pose, const float* root_trans, const float* root_scale,
                  float* joints, void* stream);

/* Backward of glamr_smpl_forward w.r.t. global_orient, root_trans and root_scale only (the quantities the optimiser
 * differentiates, global_recon_model.py:517-524 with get_parameter :591-633).  Uses the rigid identity
 * out = s * R(global_orient) * (x_local - pivot_local) + t  (SURVEY.md App. B step 8), so only the forward OUTPUTS are needed.
 * dev in: pose (B,72), root_trans, root_scale (may be NULL), verts/joints = forward outputs, g_verts (B,V,3) or NULL,
 * g_joints (B,n_out,3) or NULL.  dev out: g_orient (B,3), g_trans (B,3) or NULL, g_scale (B) or NULL. */
int glamr_smpl_backward_root(glamr_smpl* h, int B, const float* pose, const float* root_trans, const float* root_scale,
                             const float* verts, const float* joints, const float* g_verts, const float* g_joints,
                             float* g_orient, float* g_trans, float* g_scale, int flags, void* stream);

/* GENERAL backward of glamr_smpl_forward: gradients w.r.t. the whole pose (global orientation + body pose), the shape coefficients, the
 * root translation and scale, from gradients of the joints and / or the vertices -- what torch autograd gives through
 * lib/models/smpl.py:289-316 and smplx.lbs (blend shapes, kinematic chain, skinning, joint regression, re-anchoring).  Needed when the
 * body pose is itself a function of optimisation variables: the latent-optimisation mode (global_recon_model.py:434-437), where
 * smpl_pose comes out of the motion infiller every iteration.  Works with and without root_trans (re-anchored / plain call).
 * dev in : pose (B,72), betas (B,num_betas), root_trans (B,3) / root_scale (B) or NULL as in the forward call, g_verts (B,V,3) or NULL,
 *          g_joints (B,n_out,3) or NULL (at least one); verts / joints = the forward outputs, only read when g_scale is requested.
 * dev out: g_pose (B,72) or NULL, g_betas (B,num_betas) or NULL, g_trans (B,3) or NULL, g_scale (B) or NULL.
 * Without g_verts the joints-only tiling (picked + virtual vertices) is walked, as in the forward.  Sums over vertex tiles are made in a
 * fixed order (reproducible).  workspace: glamr_smpl_backward_workspace_bytes(h, B, g_verts != NULL). */
size_t glamr_smpl_backward_workspace_bytes(const glamr_smpl* h, int B, int with_vertex_gradient);
int glamr_smpl_backward(glamr_smpl* h, int B, const float* pose, const float* betas, const float* root_trans, const float* root_scale,
                        const float* verts, const float* joints, const float* g_verts, const float* g_joints, float* g_pose,
                        float* g_betas, float* g_trans, float* g_scale, int flags, void* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------------------------------
 * Motion priors -- replace MotionInfillerVAE.inference (motion_infiller/models/motion_infiller_vae.py:618-667),
 * TrajPredVAE.inference (traj_pred/models/traj_pred_vae.py:524-548) and MotionTrajJointModel.inference
 * (motion_infiller/models/motion_traj_joint_model.py:141-145).
 * ------------------------------------------------------------------------------------------------------------------- */
typedef struct glamr_nets glamr_nets;

/* Weights are handed over as ONE packed fp32 host blob plus a table of (offset, rows, cols) per tensor, in the order of
 * the checkpoint layout (glamr_amd/models/layouts.py == state_dict order of the reference modules). */
typedef struct glamr_tensor_desc { int64_t offset; int32_t rows; int32_t cols; } glamr_tensor_desc;

int glamr_nets_create(glamr_nets** out, const float* infiller_blob, const glamr_tensor_desc* infiller_desc, int n_infiller,
                      const float* trajpred_blob, const glamr_tensor_desc* trajpred_desc, int n_trajpred,
                      const float* fk_rest_joints /* host (24,3): J_regressor @ v_template */, const int32_t* parents);
int glamr_nets_destroy(glamr_nets* h);
size_t glamr_nets_workspace_bytes(const glamr_nets* h, int n_seq, int max_len);
/* Arithmetic of this handle: 0 = fp32-grade products on the fp16 matrix cores (every fp32 operand as two fp16 planes: 2^-22 of a product,
 * range 6e-8 .. 65504), 1 = plain fp32 kernels only.  glamr_nets_create decides from the WEIGHTS: a worst-case bound of every value the
 * split kernels convert (LayerNorm outputs are bounded whatever comes in, attention outputs by the value rows, LSTM states by 1; inputs
 * taken within |pose| <= 10, root-relative joints <= 4 m, translations <= 200 m, |z| <= 100) must stay below 3e4, and so must the weights -- otherwise (or with
 * GLAMR_NETS_FORCE_FP32 set) the handle never uses the fp16 planes.  worst_case[2] (may be NULL): that bound, and the largest weight. */
int glamr_nets_precision(const glamr_nets* h, double* worst_case);

/* Batched infiller + trajectory predictor over n_seq independent sequences (one "person" each).
 * dev in : body_pose (n_seq, max_len, 69) zero outside each sequence's [0,len) and on invisible frames
 *          (global_recon_model.py:146-147,357); visible (n_seq, max_len) 1/0; lens (n_seq) int32 host;
 *          motion_eps (n_seq, n_win_max, 128), traj_eps (n_seq, 128) -- the Gaussian draws of lib/utils/dist.py:21-23.
 * dev out: out_pose (n_seq, max_len, 69)   = infer_out_body_pose
 *          out_local_traj (n_seq, max_len, 11) = infer_out_local_traj_tp (time-major in the reference)
 *          out_trans (n_seq, max_len, 3), out_orient (n_seq, max_len, 3) = infer_out_trans / infer_out_orient. */
#define GLAMR_NETS_INFILL 1   /* run the motion infiller (MotionInfillerVAE.inference, multi_step) */
#define GLAMR_NETS_TRAJ 2     /* run the trajectory predictor on the (infilled) body pose (TrajPredVAE.inference) */
#define GLAMR_NETS_PERSISTENT 4   /* the caller promises that every buffer of this call (inputs, outputs, workspace) keeps its address and is
                                     only reused on this stream: the ~450 launches are captured as a HIP graph at the first call and replayed
                                     afterwards (without the flag: captured when the identical call is seen a second time).  The handle
                                     keeps at most 24 such graphs; the least recently used one is destroyed when a 25th geometry arrives */
#define GLAMR_NETS_COSCHEDULE 8   /* the caller pipelines batches over TWO streams, so that this call runs while an optimiser stage
                                     (glamr_grecon_run_stage) of the other stream is resident: the infiller then runs on kernels that fit BESIDE a
                                     stage workgroup (no LDS, one wave per workgroup, <= 128 registers, fragment-major activations:
                                     csrc/nn_free.hpp) -- slower on an empty GPU (26 against 20 ms per 1024 x 300 frames), faster in
                                     the pipeline.  Same results to ~1e-7 (different summation order in LayerNorm).  Ignored for batches below
                                     2048 window rows and for fp32-only handles.  GLAMR_NETS_FREE=0 / 1 (environment) overrides the flag */
/* Under a CALLER's stream capture (the whole step as one graph) the call records its plain launch sequence into that graph, the upload of
 * `lens` included: the lengths are copied into a pinned table owned by the handle (alive until glamr_nets_destroy; room for 262 144
 * lengths over the handle's lifetime, GLAMR_E_INVALID beyond), so the caller's graph carries its own lengths whatever workspace it uses. */
int glamr_nets_infer(glamr_nets* h, int n_seq, int max_len, const int32_t* lens, const float* body_pose,
                     const float* visible, const float* motion_eps, int n_win_max, const float* traj_eps,
                     float* out_pose, float* out_local_traj, float* out_trans, float* out_orient,
                     int flags, void* workspace, void* stream);

/* The infiller INSIDE an optimisation loop (latent-optimisation mode, global_recon_model.py:43-44,155-158,434-437: `motion_latent` is a
 * parameter and infer_motion_traj runs every iteration).  glamr_nets_infill_taped is the infiller half of glamr_nets_infer with every
 * activation of every window kept in `tape` (glamr_nets_tape_bytes: ~4 MB per window and sequence, values + gradients);
 * glamr_nets_infill_backward then gives dL/d motion_eps (n_seq, n_win_max, 128) for a gradient g_out_pose (n_seq, max_len, 69) of the
 * generated body pose -- what torch autograd returns for `in_motion_latent` through MotionInfillerVAE.inference_multi_step (:618-632):
 * decoder, reparameterisation, prior and context encoder of every window and the autoregression between windows (:604-607).  Weights are
 * constants.  (The trajectory predictor needs no backward: get_pred_trajectory_base :396 detaches its output.)  lens / motion_eps must
 * be the ones of the taped call. */
size_t glamr_nets_tape_bytes(const glamr_nets* h, int n_seq, int max_len);
int glamr_nets_infill_taped(glamr_nets* h, int n_seq, int max_len, const int32_t* lens, const float* body_pose, const float* visible,
                            const float* motion_eps, int n_win_max, float* out_pose, void* tape, void* stream);
int glamr_nets_infill_backward(glamr_nets* h, int n_seq, int max_len, const int32_t* lens, const float* motion_eps, int n_win_max,
                               const float* g_out_pose, float* g_motion_eps, void* tape, void* stream);

/* Training-mode / reconstruction passes of the two VAEs -- what `forward(data)` and `inference(recon=True)` run:
 *   MotionInfillerVAE.forward  motion_infiller/models/motion_infiller_vae.py:478-482 = ContextEncoder :92-123, DataEncoder (posterior)
 *                              :126-249, DataDecoder in mode 'train' / 'recon' / 'infer' :345-433; one-shot inference :659-666
 *   TrajPredVAE.forward        traj_pred/models/traj_pred_vae.py:378-382 = ContextEncoder :72-92, DataEncoder :95-199, DataDecoder :269-334;
 *                              init_batch_data :396-457 (global -> local trajectory); one-shot and chunked inference :478-548
 * The decoder mode selects the latent: INFER z = prior mu + eps sigma, TRAIN z = posterior mu + eps sigma, RECON z = posterior mu. */
enum { GLAMR_VAE_INFER = 0, GLAMR_VAE_TRAIN = 1, GLAMR_VAE_RECON = 2 };

typedef struct glamr_infiller_io {       /* one 50-frame window per sequence; all dev fp32 */
  const float* in_body_pose;             /* (n_seq, 50, 69) in_body_pose_tp: the masked window the context encoder sees */
  const float* body_pose;                /* (n_seq, 50, 69) body_pose_tp: the full window, posterior input (TRAIN / RECON), else NULL */
  const float* frame_mask;               /* (n_seq, 50) 1 = frame visible; the key-padding mask is its complement (:497-499) */
  const float* eps;                      /* (n_seq, 128) Gaussian draw (dist.py:21-23) for INFER / TRAIN, NULL for RECON */
  float* context;                        /* out (n_seq, 50, 256) data['context'], or NULL */
  float* q_z;                            /* out (n_seq, 2, 128) posterior mu, logvar (TRAIN / RECON), or NULL */
  float* p_z;                            /* out (n_seq, 2, 128) prior mu, logvar, or NULL */
  float* z;                              /* out (n_seq, 128) the latent the decoder used, or NULL */
  float* out_body_pose;                  /* out (n_seq, 30, 69) the generated current frames [10, 40) */
} glamr_infiller_io;
/* workspace: glamr_nets_workspace_bytes(h, n_seq, 50) */
int glamr_nets_infiller_window(glamr_nets* h, int n_seq, int mode, const glamr_infiller_io* io, void* workspace, void* stream);

typedef struct glamr_traj_io {           /* one clip of T frames per sequence; all dev fp32 */
  const float* in_body_pose;             /* (n_seq, T, 69): joints by forward kinematics (get_joint_pos :384-394), or NULL with in_joint_pos */
  const float* in_joint_pos;             /* (n_seq, T, 69) joint positions handed over as they are, or NULL */
  const float* trans;                    /* (n_seq, T, 3) root translation: posterior input + first row of the output (TRAIN / RECON), else NULL */
  const float* orient;                   /* (n_seq, T, 3) root orientation, axis-angle, with trans */
  const float* eps;                      /* (n_seq, 128) for INFER / TRAIN */
  const float* init_row;                 /* (n_seq, 11) columns 0,1 (xy) and 9,10 (heading vector) pin the FIRST output row (DataDecoder :319-321 init_xy /
                                            init_heading); NULL: the first row of local_traj when trans is given (:322-324), else zeros and (0, 1) (:325-327) */
  int32_t valid_len;                     /* > 0: frames >= valid_len are zero-padded joint rows (get_seg_data :487-496), the networks still run over T */
  float* local_traj;                     /* out (n_seq, T, 11) local_traj_tp = traj_global2local_heading(trans, orient), or NULL */
  float* q_z;                            /* out (n_seq, 256) posterior mu | logvar, or NULL */
  float* p_z;                            /* out (n_seq, 256) prior mu | logvar, or NULL */
  float* z;                              /* out (n_seq, 128), or NULL */
  float* out_orig_local_traj;            /* out (n_seq, T, 11) <mode>_orig_out_local_traj_tp (decoder output before the first row is pinned), or NULL */
  float* out_local_traj;                 /* out (n_seq, T, 11) <mode>_out_local_traj_tp */
  float* out_trans;                      /* out (n_seq, T, 3), or NULL */
  float* out_orient;                     /* out (n_seq, T, 3) axis-angle, or NULL */
  float* out_orient_q;                   /* out (n_seq, T, 4) quaternion (w, x, y, z), or NULL */
} glamr_traj_io;
/* workspace: glamr_nets_workspace_bytes(h, n_seq, T) */
int glamr_nets_traj_clip(glamr_nets* h, int n_seq, int T, int mode, const glamr_traj_io* io, void* workspace, void* stream);
/* traj_local2global_heading (traj_pred/utils/traj_utils.py:65-88) on its own: the chunked inference concatenates local rows first (:498-517) */
size_t glamr_traj_local_to_global_workspace_bytes(int n_seq, int T);
int glamr_traj_local_to_global(int n_seq, int T, const float* local_traj, float* out_trans, float* out_orient, float* out_orient_q,
                               void* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------------------------------
 * Fused global optimiser -- replaces GlobalReconOptimizer.forward/compute_loss/optimize_main
 * (global_recon/models/global_recon_model.py:394-570) and the residuals of global_recon/models/loss_func.py.
 * One launch runs ALL iterations of one stage for a batch of scenes; each scene (a sequence with P persons sharing a
 * camera) is owned by one workgroup, so iterations need no grid-wide synchronisation.
 * ------------------------------------------------------------------------------------------------------------------- */

/* optimisation variables (opt_variables in global_recon/cfg/ *.yml; get_parameter :591-633) */
#define GLAMR_VAR_CAM (1u << 0)
#define GLAMR_VAR_LOCAL_XY (1u << 1)
#define GLAMR_VAR_LOCAL_HEADING (1u << 2)
#define GLAMR_VAR_WORLD_DHEADING (1u << 3)
#define GLAMR_VAR_LOCAL_DXY (1u << 4)
#define GLAMR_VAR_LOCAL_ROT (1u << 5)
#define GLAMR_VAR_LOCAL_Z (1u << 6)
#define GLAMR_VAR_LOCAL_DHEADING (1u << 7)

/* model flags (grecon_model_specs) */
#define GLAMR_FLAG_FIXED_CAM (1u << 0)             /* flag_fixed_cam */
#define GLAMR_FLAG_CAM_FROM_PERSON (1u << 1)       /* flag_opt_cam_from_person_pose */
#define GLAMR_FLAG_HAS_WORLD_DHEADING (1u << 2)    /* a previous stage created pose_dict['world_dheading'] (:624-627) */
/* flags of a stage that is driven launch by launch from outside (the person-sharded variant, glamr_amd/parallel.py) */
#define GLAMR_FLAG_KEEP_CAM_PARAMS (1u << 3)       /* the camera parameters in `params` are current: do NOT re-derive them from cam_pose when the
                                                      launch starts (get_parameter :596-606 runs once per stage, not once per launch) */
#define GLAMR_FLAG_NO_CAMERA_TERMS (1u << 4)       /* leave the camera-only residuals (cam_inv_rot_smoothness, cam_origin_smoothness, cam_up_reg,
                                                      cam_inv_trans_residual_reg) out of the gradient: another rank owns them */

#define GLAMR_FLAG_POSES_ONLY (1u << 5)            /* niters == 0 only: stop after orient_world / trans_world (and cam_pose) are written -- no
                                                      projections, no loss values.  init_data's forward pass before init_cam_pose(all_frames=True)
                                                      (global_recon_model.py:243-246) is followed by a full one whenever its outputs are used */

#define GLAMR_FLAG_ABSOLUTE_HEADING (1u << 6)      /* absolute_heading (global_recon_model.py:59,283,421; no shipped config): the heading entries of the local
                                                      trajectory are absolute, not increments.  Runs on the instances of csrc/grecon_wide.hip whatever the
                                                      number of persons (the instances with the on-chip arena are compiled without it) */
#define GLAMR_FLAG_KEEP_TABLES (1u << 8)           /* niters > 0, launch-by-launch schedules: the caller runs this stage again on the SAME workspace (contents
                                                      untouched since the previous launch of this stage on this batch): the stage-constant tables the set-up
                                                      leaves there -- visibility ranks, pair tables, per-joint score sums, orientation targets, the workspace rows
                                                      of the keypoint table -- are not rebuilt.  Scenes of several persons only; ignored otherwise (round 6) */
#define GLAMR_FLAG_NO_REPORT (1u << 7)             /* niters > 0, launch-by-launch schedules (glamr_amd/parallel.py): the launch's LAST evaluation updates and
                                                      records gradients like the others but writes no outputs and no loss values -- the gradient launch of
                                                      every iteration except a stage's last one, whose outputs are the stage's (round 6).  Ignored (the launch reports as usual) by
                                                      launches that record the per-iteration loss history and by scenes of one person */

/* loss ids (loss_func_dict, loss_func.py:314-340) -- order of glamr_stage_desc.loss_weight[] */
enum {
  GLAMR_LOSS_KP_2D = 0, GLAMR_LOSS_KP_2D_DIST, GLAMR_LOSS_REL_TRANSFORM, GLAMR_LOSS_CAM_TRAJ_ROT,
  GLAMR_LOSS_TRAJ_ROT_SMOOTHNESS, GLAMR_LOSS_LOCAL_DXY_REG, GLAMR_LOSS_LOCAL_DHEADING_REG_NEW, GLAMR_LOSS_LOCAL_ROT_REG,
  GLAMR_LOSS_LOCAL_Z_REG, GLAMR_LOSS_CAM_INV_TRANS_RES_REG, GLAMR_LOSS_CAM_INV_ROT_SMOOTHNESS,
  GLAMR_LOSS_CAM_ORIGIN_SMOOTHNESS, GLAMR_LOSS_CAM_UP_REG, GLAMR_NUM_LOSSES
};

typedef struct glamr_stage_desc {
  uint32_t var_mask;                     /* GLAMR_VAR_* */
  uint32_t flags;                        /* GLAMR_FLAG_* */
  uint32_t loss_mask;                    /* bit i set: loss i is in the stage's loss_cfg */
  uint32_t monitor_mask;                 /* bit i set: monitor_only (reported, not optimised; compute_loss :540-542) */
  uint32_t first_frame_only_mask;        /* per-loss `first_frame_only` */
  int32_t niters;                        /* opt_niters */
  double lr;                             /* opt_lr as the DOUBLE the YAML holds: torch forms lr / (1 - beta1^t) in double before rounding to fp32;
                                            Adam betas (0.9, 0.999), eps 1e-8 (:642) */
  float loss_weight[16];
  float kp_min_conf;                     /* kp_2d / kp_2d_dist `min_conf` */
  float first_frame_weight[16];          /* per-loss `first_frame_weight` (rel_transform default 10, others 1) */
  float rel_trans_weight;                /* rel_transform `trans_weight` */
} glamr_stage_desc;

/* Geometry of the packed scene batch.  All per-frame arrays are padded to max_len frames; person p of scene s lives at
 * slot s*max_persons + p. */
typedef struct glamr_scene_batch {
  int32_t n_scenes, max_persons, max_len, n_joints;   /* n_joints = 26 */
  /* dev, int32 */
  const int32_t* n_persons;              /* (n_scenes) persons actually present in each scene (<= max_persons <= 32; more than 8: the workspace / lite-arena instances of csrc/grecon_wide.hip) */
  const int32_t* seq_len;                /* (n_scenes) frames actually present in each scene (<= max_len) */
  const int32_t* fr_start;               /* (slots) first existing frame  (exist_frames = [fr_start, fr_end), :92-95) */
  const int32_t* fr_end;                 /* (slots) */
  /* dev, fp32 constants (never written) */
  const float* vis;                      /* (slots, max_len) 1 = vis_frames (post filter_pose), 0 otherwise */
  const float* j_local;                  /* (slots, max_len, n_joints, 3) joints for zero root orient / trans (pivot-relative) */
  const float* kp_2d;                    /* (slots, max_len, n_joints, 2) kp_2d_aligned */
  const float* kp_score;                 /* (slots, max_len, n_joints)   kp_2d_score */
  const float* cam_K;                    /* (slots, max_len, 9) */
  const float* traj_local_pred;          /* (slots, max_len, 11) rows [0, exist_len) */
  const float* orient_cam;               /* (slots, max_len, 3) smpl_orient_cam (HybrIK, interpolated) */
  const float* base_orient;              /* (slots, max_len, 3) smpl_orient_world_base (used outside exist_frames) */
  const float* base_trans;               /* (slots, max_len, 3) root_trans_world_base   (used outside exist_frames) */
  const float* person2cam;               /* (slots, max_len, 12) 3x4 */
  const float* dheading_mask;            /* (slots, max_len) row e >= 1 multiplies traj_local_dheading[e-1] (:400-402); NULL = 0 */
  const float* rel_transform_cam;        /* (n_scenes, max_persons, max_persons, max_len, 12) or NULL */
  /* dev, fp32 state (read and written) */
  float* cam_pose;                       /* (n_scenes, max_len, 12) world->camera 3x4; in: current, out: last evaluated */
  float* params;                         /* (n_scenes, scene_stride) optimisation variables, see glamr_grecon_param_layout() */
  float* losses;                         /* (n_scenes, GLAMR_NUM_LOSSES) unweighted values of the LAST evaluation */
  /* dev outputs of the last forward pass */
  float* orient_world;                   /* (slots, max_len, 3) */
  float* trans_world;                    /* (slots, max_len, 3) */
  float* kp_2d_pred;                     /* (slots, max_len, n_joints, 2) */
  float* orient_cam_in_world;            /* (slots, max_len, 3) */
  /* person-sharded scenes (several persons only): a person slot with frozen[slot] != 0 belongs to ANOTHER rank.  Its world pose on every
   * frame is GIVEN in base_orient / base_trans (what that rank published: smpl_orient_world, root_trans_world); it takes part in the
   * relative-transform pairs, the camera average and every normaliser (vis, fr_start, fr_end must be its own), has no residuals of its
   * own, receives no gradient and its outputs are not written.  NULL = nobody is frozen. */
  const int32_t* frozen;                 /* (slots) or NULL */
  /* out, optional: gradient of the (weighted, normalised) loss of the LAST evaluation w.r.t. the cached joints j_local -- the hand-over to
   * glamr_smpl_backward when the body pose itself depends on optimisation variables (latent-optimisation mode, global_recon_model.py:434-437).
   * Only the reprojection term reaches the joints; rows of joints without weight and of invisible frames are zero. */
  float* g_j_local;                      /* (slots, max_len, n_joints, 3) or NULL */
  /* out, optional: the unweighted value of every loss term at EVERY iteration of the launch -- what the reference hands to write_logs after
   * each optimizer.step (global_recon_model.py:564, 646-659: one log line per iteration).  Row it of a scene = the terms evaluated at the
   * parameters iteration it started from (the closure's forward pass), in the order of the GLAMR_LOSS_* ids, like `losses`.  A launch that
   * records the history runs every iteration on the plain (workspace) instance with the reporting evaluation: about 2x the time of a launch
   * without it, same update arithmetic.  NULL = not recorded. */
  float* loss_history;                   /* (n_scenes, stage->niters, GLAMR_NUM_LOSSES) or NULL */
} glamr_scene_batch;

/* Offsets (in floats) of each variable block inside one scene's parameter vector; see glamr_grecon_param_layout().
 * Per-frame person variables are stored by existing-frame row e (row 0 of local_dxy / local_dheading is unused: the reference's
 * (n-1)-row tensors start at e = 1); world_dheading and the camera blocks are stored by video frame t. */
typedef struct glamr_param_layout {
  int32_t scene_stride;                  /* floats per scene */
  int32_t cam_rot6d, cam_trans;          /* (max_len,6) (max_len,3); fixed camera uses row 0 only */
  int32_t cam_inv_rot_res, cam_inv_trans_res;   /* (max_len,6) (max_len,3) */
  int32_t person_stride, person0;        /* person p block at person0 + p*person_stride */
  int32_t local_xy, local_dxy, local_heading, local_dheading, local_z, local_rot, world_dheading;  /* within a person block */
} glamr_param_layout;

int glamr_grecon_param_layout(int max_persons, int max_len, glamr_param_layout* out);
size_t glamr_grecon_workspace_bytes(int n_scenes, int max_persons, int max_len);

/* Runs stage->niters Adam iterations (fresh zero moments, :635-644) on every scene of the batch.  With niters == 0 it
 * performs one forward pass + loss evaluation without touching the parameters (forward(data, [], {'stage':'init'}) :246).
 * If `grads_out` (dev, n_scenes*scene_stride) is non-NULL the gradient of the LAST evaluated iteration is stored there. */
int glamr_grecon_run_stage(const glamr_scene_batch* batch, const glamr_stage_desc* stage, float* grads_out,
                           void* workspace, void* stream);
/* Duration of the last glamr_grecon_run_stage that used `workspace`, from the kernel's own clock (earliest workgroup start to latest
 * workgroup end, 10 ns resolution): what a profiler reports for the dispatch, also when other streams share the GPU.  Blocks until
 * the work enqueued so far on THAT launch's stream has finished -- other streams keep running (no device-wide wait). */
int glamr_grecon_last_launch_ns(const void* workspace, double* ns);

/* One torch.optim.Adam step (betas 0.9 / 0.999, eps 1e-8, no weight decay; torch/optim/adam.py _single_tensor_adam, the path
 * GlobalReconOptimizer.init_opt selects, global_recon_model.py:642) on a flat fp32 vector, with the update function the fused
 * optimiser uses: operation order of torch's CPU kernels, IEEE quotient / square root, `lr` and the bias corrections formed in double
 * as Python does.  `step` is 1-based.  All arrays dev, n floats; params / exp_avg / exp_avg_sq are updated in place.  The parity
 * tests compare it bit for bit with torch.optim.Adam. */
int glamr_adam_step(int n, float* params, float* exp_avg, float* exp_avg_sq, const float* grad, double lr, int step, void* stream);
/* The same step with its number ON THE DEVICE, for loops replayed as HIP graphs (a captured launch cannot carry a per-iteration scalar):
 * glamr_adam_coef_table fills, on the HOST, two floats per step 1 .. n_steps (-lr / (1 - 0.9^step), sqrt(1 - 0.999^step), formed in double
 * with libm pow as Python does -- what glamr_adam_step computes per call); the caller uploads it.  glamr_adam_step_indexed takes the step
 * `*step_index` (dev int32, 0-based row of the table: 0 = the first step) and leaves the index alone; glamr_counter_add advances it on the
 * stream.  Replaces the per-iteration `optimizer.step()` of global_recon_model.py:561-565 inside a captured iteration. */
int glamr_adam_coef_table(double lr, int n_steps, float* out_host);
int glamr_adam_step_indexed(int n, float* params, float* exp_avg, float* exp_avg_sq, const float* grad, const float* coef_table,
                            const int32_t* step_index, void* stream);
int glamr_counter_add(int32_t* counter, int value, void* stream);


/* ---------------------------------------------------------------------------------------------------------------------
 * Device-side init_data -- replaces GlobalReconOptimizer.init_data (global_recon_model.py:76-248) between the HybrIK wire format
 * (pose_est/hybrik_demo/demo.py:317-354) and the first optimisation stage.  The host only scatters the per-detection arrays to
 * their video-frame rows; everything else (rotation matrices -> axis-angle, gap interpolation, keypoint remap, filter_pose, masks,
 * person / relative transforms, initial camera, heading initialisation) runs in three kernels.
 * ------------------------------------------------------------------------------------------------------------------- */
typedef struct glamr_raw_batch {
  int32_t n_slots, max_len;              /* n_slots = n_scenes * max_persons */
  const int32_t* seq_len;                /* dev (n_slots) video frames of the slot's scene */
  const float* exist;                    /* dev (n_slots, max_len) bboxes_dict['exist'] */
  const float* rotmats;                  /* dev (n_slots, max_len, 24, 9) smpl_pose_quat_wroot rows at their frame positions */
  const float* betas;                    /* dev (n_slots, max_len, 10) */
  const float* root_trans;               /* dev (n_slots, max_len, 3) */
  const float* kp_2d;                    /* dev (n_slots, max_len, 24, 2) first 24 HybrIK keypoints */
} glamr_raw_batch;

typedef struct glamr_person_arrays {     /* per-person state that is not an input of the optimiser kernel; all dev, written */
  float* visible_orig;                   /* (n_slots, max_len) */
  float* smpl_pose;                      /* (n_slots, max_len, 69) interpolated, then infilled on existing frames */
  float* smpl_beta;                      /* (n_slots, max_len, 10) */
  float* trans_cam;                      /* (n_slots, max_len, 3) root_trans_cam */
  float* nets_pose;                      /* (n_slots, max_len, 69) rows [0, exist_len): input of glamr_nets_infer */
  float* nets_vis;                       /* (n_slots, max_len) */
} glamr_person_arrays;

/* Host-side scatter of the HybrIK wire format (one array row per DETECTED frame, demo.py:317-354) into the frame-indexed staging arrays
 * whose device copies glamr_raw_batch points at -- what the reference does person by person with fancy indexing inside init_data
 * (global_recon_model.py:98-136).  All pointers are HOST pointers.  `table` has one row of TEN int64 per person slot:
 *   exist (address of bboxes_dict['exist']), exist_is_f64 (1: float64 as the reference stores it, 0: float32), n_frames, n_det (rows of
 *   the per-detection arrays), then the addresses of smpl_pose_quat_wroot (n_det,216), smpl_beta (n_det,10), root_trans (n_det,3),
 *   cam_K (n_det,9), kp_2d (n_det,n_kp,2) -- float32, C-contiguous -- and kp_stride = 2 n_kp, the floats per row of kp_2d (HybrIK: 58;
 *   at least 48: the first 24 keypoints are taken).
 * A row whose `exist` address is 0 is an empty person slot (seq_len 0, nothing copied).
 * Rows of undetected frames are left untouched (the device code never reads them); seq_len[k] = n_frames, exist_len[k] = last - first
 * + 1 detected frame.  Persons are split over `threads` host threads.  Errors: a person without detections, or whose exist array marks a
 * different number of frames than n_det. */
typedef struct glamr_host_staging {
  float* exist;                          /* (n_slots, max_len) */
  float* rot;                            /* (n_slots, max_len, 216) */
  float* betas;                          /* (n_slots, max_len, 10) */
  float* trans;                          /* (n_slots, max_len, 3) */
  float* K;                              /* (n_slots, max_len, 9) */
  float* kp;                             /* (n_slots, max_len, 48): the first 24 of the 29 keypoints */
} glamr_host_staging;
int glamr_host_scatter(int n_persons, const int64_t* table, int max_len, const glamr_host_staging* staging, int32_t* seq_len,
                       int32_t* exist_len, int threads);

size_t glamr_init_workspace_bytes(int n_slots, int max_len);
/* Fills vis, kp_2d, kp_score, orient_cam, base_orient, base_trans, fr_start, fr_end of `batch` (declared const there because the
 * optimiser only reads them) and every array of `pa`.  cam_K is written by the host directly. */
/* Value checks of the wire format on the uploaded raw batch (keys and shapes are checked on the host, glamr_amd/utils/wire.py): per detected
 * frame every number finite and the 24 matrices of `smpl_pose_quat_wroot` orthonormal to 1e-2 (the field is named after quaternions but holds
 * 3 x 3 matrices, demo.py:320).  One pass over the arrays.  dev out: verdict (2, n_slots) int32: [0][slot] != 0 = a matrix that is no
 * rotation, [1][slot] != 0 = a non-finite value.  cam_K: dev (n_slots, max_len, 9). */
int glamr_check_inputs(const glamr_raw_batch* raw, const float* cam_K, int32_t* verdict, void* stream);
/* `filter`: filter_pose (global_recon_model.py:250-271).  filter_pose != 0: a visible frame whose root orientation jumps by more than pi/3
 * from its predecessor makes itself or the predecessor invisible (the look-ahead rule :256-262, sequential); make_invis_with_keypoint != 0
 * (flag_make_invis_with_keypoint :264-268, applied inside filter_pose like the reference): a still-visible frame with fewer than
 * keypoint_min_num scores above keypoint_min_score becomes invisible (defaults of the reference: 0.6 / 15).  NULL = no filtering. */
typedef struct glamr_filter_opts {
  int32_t filter_pose;
  int32_t make_invis_with_keypoint;
  float keypoint_min_score;
  int32_t keypoint_min_num;
} glamr_filter_opts;
int glamr_init_prepare(const glamr_raw_batch* raw, const glamr_scene_batch* batch, const glamr_person_arrays* pa, const glamr_filter_opts* filter,
                       void* workspace, void* stream);
/* After glamr_nets_infer: scatters its outputs, fills traj_local_pred, person2cam, rel_transform_cam, cam_pose of `batch`. */
int glamr_init_scenes(const glamr_scene_batch* batch, const glamr_person_arrays* pa, const float* nets_pose_out,
                      const float* nets_local_traj, const float* nets_trans, const float* nets_orient, void* workspace, void* stream);
/* The same with model flags that change the initialisation.  GLAMR_INIT_TRAJ_FROM_CAM = flag_traj_from_cam (global_recon_model.py:55,237,
 * get_traj_from_cam :325-351 with traj_interp_method 'linear_interp'): the base pose of a person (smpl_orient_world_base / root_trans_world_base,
 * what the frames outside its existence range keep) is read off the initial camera -- translation of cam_pose_inv . person_transform_cam,
 * orientation interpolated between the frames the person is seen in, heading separately (interp_orient_q_sep_heading). */
enum { GLAMR_INIT_TRAJ_FROM_CAM = 1, GLAMR_INIT_POSE_SCATTERED = 2 };
/* The first step of glamr_init_scenes alone -- the motion infiller's poses (rows [0, n) of a person) into the video-frame rows of
 * person_arrays.smpl_pose (global_recon_model.py:152-159: `smpl_pose[exist range] = infer_out_pose`) -- so that the joints-only skinning (:517-528,
 * cached for the whole schedule) can run as soon as the infiller is done, before the trajectory predictor.  A later glamr_init_scenes_ex is then
 * given GLAMR_INIT_POSE_SCATTERED and leaves smpl_pose alone. */
int glamr_init_scatter_pose(const glamr_scene_batch* batch, const glamr_person_arrays* pa, const float* nets_pose_out, void* stream);
int glamr_init_scenes_ex(const glamr_scene_batch* batch, const glamr_person_arrays* pa, const float* nets_pose_out,
                         const float* nets_local_traj, const float* nets_trans, const float* nets_orient, int flags, void* workspace, void* stream);
/* init_cam_pose(all_frames=True) (:243-244) from the orient_world / trans_world of the last forward pass. */
int glamr_init_cam_all_frames(const glamr_scene_batch* batch, void* stream);

/* ---- evaluator pieces (global_recon/utils/evaluator.py:202-327).  All arrays dev, fp32. ----------------------------
 * glamr_eval_regress_joints: joints (B, n_joints, 3) = regressor (n_joints, V) . verts (B, V, 3) -- `torch.matmul(self.J_regressor, vertices)` of
 *   evaluator.py:262-263 with JOINT_REGRESSOR_H36M (lib/models/smpl.py:29); n_joints <= 32.
 * glamr_eval_procrustes: every frame of S1 (n, n_joints, 3) aligned onto S2 by the least-squares similarity transform (scale, rotation,
 *   translation; 3 x 3 SVD per frame, in double) -- batch_compute_similarity_transform_torch, lib/utils/torch_transform.py:282-345.
 * glamr_eval_heading_align: get_aligned_orient_trans (evaluator.py:202-216): the trajectory (axis-angle orientation, translation; n frames) cut
 *   in chunks of `align_freq` frames that overlap by one frame, each expressed in the heading frame of its first frame with that frame's xy as
 *   origin (convert_traj_world2heading with apply_base_orient_after, traj_pred/utils/traj_utils.py:97-107).  aligned_orient_q (n, 4) may be NULL. */
int glamr_eval_regress_joints(int B, int V, int n_joints, const float* verts, const float* regressor, float* joints, void* stream);
int glamr_eval_procrustes(int n, int n_joints, const float* S1, const float* S2, float* S1_aligned, void* stream);
int glamr_eval_heading_align(int n, int align_freq, const float* orient_aa, const float* trans, float* aligned_orient_aa, float* aligned_trans,
                             float* aligned_orient_q, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GLAMR_HIP_H */
