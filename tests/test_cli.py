"""The command lines the reference documents (README.md "Demo" / "Evaluation", global_recon/run_demo.py:21-31, run_dataset.py:42-47) must be
accepted by the drop-in entry points, with the reference's defaults and file naming; flags that need the pose estimator or the renderer
(outside this package) fail with an error that names them.  Host logic only -- no GPU."""
import os
import pickle

import pytest

from glamr_amd.global_recon import run_demo, run_dataset

README_DEMO = [
    '--cfg glamr_dynamic --video_path assets/dynamic/running.mp4 --out_dir out/glamr_dynamic/running --save_video',
    '--cfg glamr_static --video_path assets/static/basketball.mp4 --out_dir out/glamr_static/basketball --save_video',
    '--cfg glamr_static_multi --video_path assets/static/basketball.mp4 --out_dir out/glamr_static_multi/basketball --save_video --multi',
]


@pytest.mark.parametrize('line', README_DEMO)
def test_run_demo_accepts_the_documented_command_lines(line):
    args = run_demo.build_parser().parse_args(line.split())
    seq_name, pose_est_dir, out_file = run_demo.names(args)
    video = os.path.splitext(os.path.basename(args.video_path))[0]
    assert seq_name == video and pose_est_dir == args.out_dir + '/pose_est'                 # run_demo.py:47-48,64
    assert out_file == '%s/grecon/%s_seed1.pkl' % (args.out_dir, video)                    # :74
    assert args.save_video and args.gpu == 0 and args.cached == 1 and args.seed == 1


def test_run_demo_defaults_and_every_flag():
    a = run_demo.build_parser().parse_args([])
    assert (a.cfg, a.out_dir, a.pose_est_dir, a.multi, a.vis, a.vis_cam, a.save_video) == ('glamr_static', 'out/glamr_static/basketball', None, False, False, False, False)
    assert run_demo.names(a)[0] == 'basketball'
    a = run_demo.build_parser().parse_args('--vis --vis_cam --pose_est_dir some/where --seed 7 --gpu 1 --cached 0'.split())
    assert a.vis and a.vis_cam and a.seed == 7 and a.gpu == 1 and a.cached == 0 and run_demo.names(a)[1] == 'some/where'


def test_run_demo_names_what_a_flag_needs(tmp_path, monkeypatch):
    """No pose.pkl and no HybrIK here: the error names the module; a cached result + --save_video reaches the renderer and names that."""
    monkeypatch.chdir(tmp_path)
    with pytest.raises(run_demo.ReferenceComponentMissing, match='pose_est.run_pose_est_demo'):
        run_demo.main(['--cfg', 'glamr_static', '--video_path', 'clip.mp4', '--out_dir', str(tmp_path / 'o')])
    os.makedirs(tmp_path / 'o' / 'grecon')
    with open(tmp_path / 'o' / 'grecon' / 'clip_seed1.pkl', 'wb') as f:
        pickle.dump({'meta': {'num_fr': 3}}, f)
    assert run_demo.main(['--video_path', 'clip.mp4', '--out_dir', str(tmp_path / 'o')]).endswith('clip_seed1.pkl')      # cached: nothing else needed
    with pytest.raises(run_demo.ReferenceComponentMissing, match=r'--save_video needs global_recon\.vis\.vis_grecon'):
        run_demo.main(['--video_path', 'clip.mp4', '--out_dir', str(tmp_path / 'o'), '--save_video'])


def test_run_dataset_accepts_the_documented_command_line():
    a = run_dataset.build_parser().parse_args('--dataset 3dpw --cfg glamr_3dpw --out_dir out/3dpw'.split())
    assert run_dataset.parse_seeds(a.seeds) == [1] and a.cached == 1
    seqs = run_dataset.test_sequences(a.dataset)
    assert len(seqs) == 24 and seqs[0] == 'downtown_arguing_00' and seqs[-1] == 'outdoors_fencing_01' and 'downtown_runForBus_01' in seqs
    a = run_dataset.build_parser().parse_args(['--seeds', '1,2,3', '--cached', '0'])
    assert run_dataset.parse_seeds(a.seeds) == [1, 2, 3] and a.cached == 0 and (a.dataset, a.cfg, a.out_dir) == ('3dpw', 'glamr_3dpw', 'out/3dpw')
    assert run_dataset.parse_seeds(['1', '2']) == [1, 2]
