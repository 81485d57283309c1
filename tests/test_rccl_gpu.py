"""RCCL on the GPU box: the collectives `bench.py --gpus N` and earl_benchmark_amd.sharding issue (barrier, all_reduce MAX of the job time, all_gather_into_tensor of the
evaluation summaries and of a packed trajectory buffer) run on DEVICE tensors over the `nccl` backend.  The test boxes have one GPU, and RCCL refuses two ranks on one device,
so this is world size 1 in a child process: it proves the backend initialises on this image (HSA_ENABLE_IPC_MODE_LEGACY=0, device_id=...) and accepts the dtypes, shapes and
in-place forms the job uses -- not that eight GPUs scale (the driver's SCALE run measures that; the W = 2 logic is covered over gloo in test_bench_sequence.py)."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
sys.path.insert(0, %(root)r)
import torch
import torch.distributed as dist
from earl_benchmark_amd import sharding
torch.cuda.set_device(0)
dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%(port)d', rank=0, world_size=1, device_id=torch.device('cuda:0'))
dist.barrier()
t = torch.tensor([1.25], dtype=torch.float64, device='cuda:0')
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == 1.25
n, T, D = 96, 7, 12
g = torch.Generator(device='cuda').manual_seed(3)
rew, suc = torch.rand(T, n, generator=g, device='cuda'), torch.rand(T, n, generator=g, device='cuda') > 0.5
summ = sharding.rollout_summary(rew, suc)                       # [n, 2]
out = torch.empty(n, 2, dtype=summ.dtype, device='cuda')
dist.all_gather_into_tensor(out, summ.contiguous())             # the job's one collective, as gather_summary issues it
assert torch.equal(out, summ)
obs = torch.rand(T, n, D, generator=g, device='cuda')
buf = sharding.pack_rollout(obs, rew, ~suc, suc)
big = torch.empty(1 * T, n, D + 2, dtype=buf.dtype, device='cuda')
dist.all_gather_into_tensor(big, buf.contiguous())              # ... and the trajectory variant (gather_rollout)
assert torch.equal(big.view(T, n, D + 2), buf)
sz = [torch.zeros(1, dtype=torch.int64, device='cuda')]
dist.all_gather(sz, torch.tensor([n], dtype=torch.int64, device='cuda'))     # the size exchange of ragged shards
assert int(sz[0].item()) == n
assert torch.equal(sharding.gather_summary(summ, sizes=[n]), summ) and torch.equal(sharding.gather_rollout(buf, sizes=[n]), buf)
dist.barrier()
dist.destroy_process_group()
print('RCCL_OK')
'''


@pytest.mark.gpu
def test_the_jobs_collectives_run_over_rccl_on_device_tensors():
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
  env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
  r = subprocess.run([sys.executable, '-c', CHILD % {'root': ROOT, 'port': port}], capture_output=True, text=True, timeout=300, env=env)
  assert r.returncode == 0 and 'RCCL_OK' in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
