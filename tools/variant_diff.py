"""bitwise comparison of the door rollout through alternative builds (tools/ubench/libearl_<tag>.so) against the shipped library, forced to the
single-wave-workgroup path: which build flag changes the arithmetic?  Each build runs in its own process (one library per process)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 and sys.argv[1] == 'run':
  sys.path.insert(0, ROOT)
  import torch
  from earl_benchmark_amd import _abi
  tag = sys.argv[2]
  if tag != 'ship':
    _abi.LIB_PATH = os.path.join(ROOT, 'tools', 'ubench', f'libearl_{tag}.so')
  lib = _abi.load()
  lib.earl_debug_set_door_variant(1)
  from earl_benchmark_amd.envs.sawyer_door import SawyerDoor
  n, T = 203, 60
  env = SawyerDoor(num_envs=n, seed=2)
  env.reset()
  g = torch.Generator(device='cuda').manual_seed(8)
  acts = (torch.rand(T, n, 4, generator=g, device='cuda') * 2 - 1).float(); acts[:, :, 1] = acts[:, :, 1].abs()
  out = env.rollout(acts)
  torch.save({'obs': out['obs'].cpu(), 'qpos': env.qpos.cpu()}, f'/tmp/variant_{tag}.pt')
else:
  import torch
  tags = ['ship'] + sys.argv[1:]
  for t in tags:
    subprocess.run([sys.executable, __file__, 'run', t], check=True, stderr=subprocess.DEVNULL)
  ref = torch.load('/tmp/variant_ship.pt')
  for t in tags[1:]:
    x = torch.load(f'/tmp/variant_{t}.pt')
    d = (x['obs'] - ref['obs']).abs()
    first = int((d.amax((1, 2)) > 0).nonzero()[0]) if bool((d > 0).any()) else -1
    print(t, 'identical' if first < 0 else f'differs: max {float(d.max()):.3e}, first step {first}')
