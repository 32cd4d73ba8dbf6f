"""Global reconstruction of one video: the reference's `global_recon/run_demo.py` on the MI355X path, with the reference's command
line (run_demo.py:21-31; README "Demo"):

    python -m glamr_amd.global_recon.run_demo --cfg glamr_dynamic --video_path assets/dynamic/running.mp4 \\
                                              --out_dir out/glamr_dynamic/running [--save_video] [--multi] [--vis [--vis_cam]]

What this package replaces is the step between the pose estimator's `pose.pkl` and the visualiser: `<pose_est_dir>/pose.pkl` is read
(`--pose_est_dir` defaults to `<out_dir>/pose_est` like the reference, :47-52), `GlobalReconOptimizer.optimize` runs on the device, and
`<out_dir>/grecon/<seq_name>_seed<seed>.pkl` is written (:74-82, same dictionary).  The two ends stay the reference's own code and are
CALLED when a flag asks for them: HybrIK (`pose_est.run_pose_est_demo.run_pose_est_on_video`) when there is no `pose.pkl` yet, the
pyvista visualiser (`global_recon.vis.vis_grecon.GReconVisualizer`) for `--vis / --vis_cam / --save_video`.  They are imported from the
reference checkout the process runs in (its root on `sys.path`, as `run_demo.py:3` arranges); when they cannot be imported the run stops
with an error that says which flag needed which module -- after the reconstruction has been saved, so nothing is lost.
Working-directory conventions are the reference's: `data/body_models/smpl/`, `data/J_regressor_extra.npy`, `results/...` checkpoints."""
import argparse
import os
import pickle

import numpy as np
import torch


class ReferenceComponentMissing(RuntimeError):
    """A flag asked for a part of the reference that is outside this package (pose estimator, renderer) and it is not importable."""


def build_parser():
    ap = argparse.ArgumentParser(description=__doc__.split('\n\n')[0])
    ap.add_argument('--cfg', default='glamr_static')
    ap.add_argument('--video_path', default=None, help="the reference's default is assets/static/basketball.mp4; only needed for pose estimation and rendering")
    ap.add_argument('--out_dir', default='out/glamr_static/basketball')
    ap.add_argument('--pose_est_dir', default=None, help='default: <out_dir>/pose_est')
    ap.add_argument('--seed', type=int, default=1)
    ap.add_argument('--gpu', type=int, default=0)
    ap.add_argument('--cached', type=int, default=1)
    ap.add_argument('--multi', action='store_true', default=False)
    ap.add_argument('--vis', action='store_true', default=False)
    ap.add_argument('--vis_cam', action='store_true', default=False)
    ap.add_argument('--save_video', action='store_true', default=False)
    ap.add_argument('--seq_name', default=None, help='extension: name of the result file (default: the video file name, else the last component of --out_dir)')
    return ap


def _need(module, attr, flag):
    import importlib
    import sys
    if os.getcwd() not in sys.path:
        sys.path.append(os.getcwd())                                 # run_demo.py:2 -- the reference imports its packages relative to the working directory
    try:
        return getattr(importlib.import_module(module), attr)
    except Exception as e:      # noqa: BLE001 -- ImportError of the module or of one of its own dependencies (pyvista, HybrIK, ...)
        raise ReferenceComponentMissing('%s needs %s.%s of the reference checkout, which could not be imported here (%s: %s). glamr_amd replaces the global '
                                        'reconstruction step only; run from the reference root with its pose-estimation / rendering dependencies installed, or '
                                        'drop the flag.' % (flag, module, attr, type(e).__name__, e))


def names(args):
    """(seq_name, pose_est_dir, result file) of a parsed command line."""
    video = args.video_path
    seq_name = args.seq_name or (os.path.splitext(os.path.basename(video))[0] if video else os.path.basename(os.path.normpath(args.out_dir)))
    pose_est_dir = args.pose_est_dir or os.path.join(args.out_dir, 'pose_est')
    return seq_name, pose_est_dir, os.path.join(args.out_dir, 'grecon', '%s_seed%d.pkl' % (seq_name, args.seed))


def main(argv=None):
    args = build_parser().parse_args(argv)
    from glamr_amd.global_recon.configs import get_config
    from glamr_amd.global_recon.models import model_dict
    from glamr_amd.utils import wire

    cfg = get_config(args.cfg)
    seq_name, pose_est_dir, out_file = names(args)
    video = args.video_path or 'assets/static/basketball.mp4'
    pose_file = os.path.join(pose_est_dir, 'pose.pkl')
    if not (args.cached and os.path.exists(out_file)) and not os.path.exists(pose_file):
        # :47-50 -- the reference runs its pose estimator on the video first
        run_pose_est = _need('pose_est.run_pose_est_demo', 'run_pose_est_on_video', 'a run without %s' % pose_file)
        run_pose_est(video, pose_est_dir, cfg['grecon_model_specs']['est_type'], cached_pose=int(args.cached), gpu_index=args.gpu, multi=args.multi)

    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    if args.cached and os.path.exists(out_file):
        print('cached result:', out_file)
        out_dict = None
    else:
        device = torch.device('cuda', args.gpu)
        torch.cuda.set_device(args.gpu)
        in_dict = wire.load_pose_pkl(pose_file, seq_name=seq_name)
        model = model_dict[cfg.get('grecon_model_name', 'global_recon_model')](cfg, device, None)
        out_dict = model.optimize(in_dict)
        os.makedirs(os.path.dirname(out_file), exist_ok=True)
        with open(out_file, 'wb') as f:
            pickle.dump(out_dict, f)
        print('saved', out_file, '| losses of the last evaluation:', np.round(model.last_losses[0], 4).tolist())
    if args.vis or args.save_video:
        if out_dict is None:
            with open(out_file, 'rb') as f:
                out_dict = pickle.load(f)
        render(args, out_dict, seq_name, video, pose_est_dir)
    return out_file


def render(args, out_dict, seq_name, video, pose_est_dir):
    """:84-131 -- hands the result dictionary to the reference's visualiser; nothing is drawn by this package."""
    flag = '--vis' if args.vis else '--save_video'
    Visualizer = _need('global_recon.vis.vis_grecon', 'GReconVisualizer', flag)
    specs = _need('global_recon.vis.vis_cfg', 'demo_seq_render_specs', flag)
    vt = _need('lib.utils', 'vis', flag)
    spec = specs.get(seq_name, specs['default'])
    frame_dir = os.path.join(pose_est_dir, 'frames')
    if (args.vis and args.vis_cam) or args.save_video:
        import glob
        if len(glob.glob(os.path.join(frame_dir, '*.jpg'))) != out_dict['meta']['num_fr']:
            vt.video_to_images(video, frame_dir, fps=30, verbose=False)
    img_w, img_h = vt.get_video_width_height(video)
    if args.vis:
        if args.vis_cam:
            Visualizer(out_dict, coord='cam_in_world', verbose=False, background_img_dir=frame_dir).show_animation(window_size=(img_w, img_h), show_axes=False)
        else:
            Visualizer(out_dict, coord='world', verbose=False, show_camera=True, render_cam_pos=spec.get('cam_pos'),
                       render_cam_focus=spec.get('cam_focus')).show_animation(window_size=(1920, 1080))
    if args.save_video:
        stem = os.path.join(args.out_dir, 'grecon_videos', '%s_seed%d' % (seq_name, args.seed))
        os.makedirs(os.path.dirname(stem), exist_ok=True)
        world, cam, sbs = stem + '_world.mp4', stem + '_cam.mp4', stem + '_sbs_all.mp4'
        Visualizer(out_dict, coord='world', verbose=False, show_camera=False, render_cam_pos=spec.get('cam_pos'),
                   render_cam_focus=spec.get('cam_focus')).save_animation_as_video(world, window_size=spec.get('wsize', (int(1.5 * img_h), img_h)), cleanup=True, crf=5)
        Visualizer(out_dict, coord='cam_in_world', verbose=False, background_img_dir=frame_dir).save_animation_as_video(cam, window_size=(img_w, img_h), cleanup=True)
        pose_video = os.path.join(pose_est_dir, 'render.mp4')
        vt.hstack_video_arr([pose_video, cam, world], sbs, verbose=False)
        print('saved videos:', world, sbs)


if __name__ == '__main__':
    main()
