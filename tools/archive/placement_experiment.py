"""GPU box: does the speed of the 28-episode evaluation launch (4096 envs x 200 steps, own actions, four action tensors) depend on WHERE its buffers
lie?  The output tensors are carved out of one pool at varying byte offsets / with varying gaps between them; same kernel, same work."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import earl_benchmark_amd as eb
n, T, E, R = 4096, 200, 28, 4
L = eb.EARLEnvs('tabletop_manipulation', reward_type='sparse', num_envs=n, eval_horizon=T, scalar_api=False)
_, env = L.get_envs()
acts = [(torch.rand(E, T, n, 3, device='cuda') * 2 - 1).contiguous() for _ in range(R)]
rows = E * T * n
pool = torch.zeros(3 * 1024 ** 3, dtype=torch.uint8, device='cuda')
print('pool base %x' % pool.data_ptr(), 'acts bases', ['%x' % a.data_ptr() for a in acts])


def carve(off, gap):
  o = off
  def take(nbytes, dtype, shape):
    nonlocal o
    t = pool[o:o + nbytes].view(dtype).view(shape)
    o += nbytes + gap
    o = (o + 255) // 256 * 256
    return t
  return (take(rows * 48, torch.float32, (E, T, n, 12)), take(rows * 4, torch.float32, (E, T, n)), take(rows, torch.bool, (E, T, n)), take(rows, torch.bool, (E, T, n)))


def timeit(out):
  for r in range(3):
    env.rollout_episodes(acts[r % R], out=out)
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  torch.cuda.synchronize()
  e0.record()
  for r in range(12):
    env.rollout_episodes(acts[r % R], out=out)
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / 12 * 1e3


for off, gap in ((0, 0), (4096, 0), (65536, 0), (1 << 20, 0), (3 << 20, 0), (0, 4096), (0, 1 << 20), (0, 33 << 20), (768, 768), (0, 0)):
  print(f'offset {off:9d} gap {gap:9d}: {timeit(carve(off, gap)):7.1f} us', flush=True)
own = tuple(torch.zeros(s, dtype=d, device='cuda') for s, d in (((E, T, n, 12), torch.float32), ((E, T, n), torch.float32), ((E, T, n), torch.bool), ((E, T, n), torch.bool)))
print('separate torch allocations:', f'{timeit(own):7.1f} us', ['%x' % t.data_ptr() for t in own])
