"""Sustained HBM throughput of the fused tabletop rollout against a compute-free kernel with the same traffic (tools/ubench/traffic_twin.hip), same process,
same buffers, >= 60 launches each, interleaved in blocks so that clock / power state is shared (is the large-N regime at this machine's
ceiling for this read:write mix, or is the kernel leaving bandwidth on the table?).  Run on the GPU box:  python tools/traffic_twin.py [N ...]"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402


def build():
  so = os.path.join(ROOT, 'tools', 'ubench', 'libtwin.so')
  src = os.path.join(ROOT, 'tools', 'ubench', 'traffic_twin.hip')
  if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-shared', '-fPIC', '-o', so, src])
  return C.CDLL(so)


def timed(fn, reps):
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / reps * 1e-3


def main():
  import bench
  lib = build()
  T = 200
  sizes = [int(x) for x in sys.argv[1:]] or [16384, 65536, 262144, 1048576]
  for N in sizes:
    from earl_benchmark_amd.envs.tabletop import TabletopManipulation
    env = TabletopManipulation(num_envs=N, reward_type='sparse', seed=1)
    sets = 4 if N * T * 12 * 4 < 4e9 else 2
    acts = [bench.synth_actions(torch, T, N, r, 'cuda') for r in range(sets)]
    out = env.rollout(acts[0], reset_first=True)
    obs, rew, done, suc = out
    stream = torch.cuda.current_stream().cuda_stream
    lin = torch.empty(N * T * 54, dtype=torch.uint8, device='cuda')          # the linear control's own slab
    bytes_ = N * T * 66
    k = [0]

    def product():
      env.rollout(acts[k[0] % sets], out=out, reset_first=True); k[0] += 1

    def twin(tile, nt, remap):
      def f():
        rc = lib.twin_launch(C.c_void_p(acts[k[0] % sets].data_ptr()), C.c_void_p((lin if tile < 0 else obs).data_ptr()), C.c_void_p(rew.data_ptr()), C.c_void_p(done.data_ptr()),
                             C.c_void_p(suc.data_ptr()), N, T, tile, nt, remap, C.c_void_p(stream))
        assert rc == 0, rc
        k[0] += 1
      return f
    variants = [('product rollout', product), ('twin 64 NT remap', twin(64, 1, 1)), ('twin 64 NT', twin(64, 1, 0)), ('twin 256 NT', twin(256, 1, 0)), ('twin 1024 NT', twin(1024, 1, 0)), ('twin 256 NT remap', twin(256, 1, 1)), ('linear control', twin(-1, 1, 0))]
    reps = max(10, int(0.05 / (bytes_ / 5e12)))             # ~50 ms of launches per block
    for name, fn in variants:
      timed(fn, 3)
    res = {name: [] for name, _ in variants}
    for rnd in range(4):
      for name, fn in variants:
        res[name].append(timed(fn, reps))
    print(f'N = {N}, T = {T}, {bytes_ / 1e6:.0f} MB per launch, {reps} launches per block, 4 blocks each (TB/s per block):')
    for name, _ in variants:
      print(f'  {name:18s} ' + '  '.join(f'{bytes_ / t / 1e12:5.2f}' for t in res[name]) + f'   | us per launch {1e6 * min(res[name]):8.1f} .. {1e6 * max(res[name]):8.1f}')
    del env, acts, out, obs, rew, done, suc, lin
    torch.cuda.empty_cache()


if __name__ == '__main__':
  main()
