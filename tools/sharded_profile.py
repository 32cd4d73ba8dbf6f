"""DEVELOPMENT AID (GPU, under rocprofv3 --kernel-trace --stats): bench.py's person-sharded line alone on a process group of ONE rank -- the kernels of a
sharded iteration (forward-only launch, gradient launch, Adam, the collectives' kernels) by name."""
import os, socket, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import bench
dev = torch.device('cuda:0')
sk = socket.socket(); sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]; sk.close()
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1, device_id=dev)
line = bench.person_sharded_line(bench.ensure_assets(), dev, 0, 1, iters=int(sys.argv[1]) if len(sys.argv) > 1 else 20)
print({k: v for k, v in line.items() if k != 'note'})
dist.destroy_process_group()
