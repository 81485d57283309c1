"""What bounds the tabletop rollout at BASELINE configs[1] taken literally -- 4096 resident env instances, ONE episode in flight per env?
Three measurements on one MI355X, all at n = 4096, T = 200:

  1. shipped library: one episode per launch, and 28 episodes per launch one after the other (debug switch 38) -- the `strict` figures of the bench line;
  2. the same library with the storers' HBM stores compiled out (tools/build_ws_variant.sh nostores -DEARL_WS_NO_STORES: LDS reads and reward arithmetic stay,
     one row in 64 is written): if the launch does not get shorter, store throughput is not what holds it;
  3. the cycle-stamped build of the same kernel (experiments library, impl 29): cycles per step of the COMPUTE wave in its own instructions, in LDS reads and at the
     chunk barrier, next to the storers' and loaders' -- who waits for whom.

usage (GPU box): python tools/strict_bound.py   [needs tools/ubench/libearl_ws_nostores.so, built here by build_ws_variant.sh]"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N, T, E = 4096, 200, 28
B_STEP, B_STATE = 66, 2 * (32 + 1 + 4) + 4


def child(lib_tag):
  import numpy as np
  import torch
  from earl_benchmark_amd import _abi
  if lib_tag != 'ship':
    _abi.LIB_PATH = os.path.join(ROOT, 'tools', 'ubench', f'libearl_ws_{lib_tag}.so')
  import earl_benchmark_amd as eb
  lib = _abi.load()
  _, env = eb.EARLEnvs('tabletop_manipulation', reward_type='sparse', num_envs=N, eval_horizon=T, scalar_api=False, seed=0).get_envs()
  g = torch.Generator(device='cuda').manual_seed(1)
  sets = [(torch.rand(E, T, N, 3, generator=g, device='cuda') * 2 - 1).contiguous() for _ in range(4)]
  outE = env.unwrapped._new_out((E, T, N))[0] if hasattr(env.unwrapped, '_new_out') else None
  out1 = env.unwrapped._new_out((T, N))[0]
  res = {'lib': lib_tag}

  def timed(fn, reps, warm):
    for j in range(warm):
      fn(j)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for j in range(reps):
      fn(warm + j)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

  ms1 = timed(lambda j: env.rollout(sets[j % 4][j % E], out=out1, reset_first=True), 200, 60)
  res['one_episode_per_launch'] = {'us': ms1 * 1e3, 'ns_per_step': ms1 * 1e6 / T, 'env_steps_per_s': N * T / (ms1 * 1e-3),
                                   'frac_of_8TBs': N * (T * B_STEP + B_STATE) / (ms1 * 1e-3) / 8e12}
  lib.earl_debug_set_rollout_impl(38)
  msq = timed(lambda j: env.rollout_episodes(sets[j % 4], out=outE), 10, 4)
  lib.earl_debug_set_rollout_impl(0)
  res['one_episode_in_flight_28_per_launch'] = {'us': msq * 1e3, 'ns_per_step': msq * 1e6 / (E * T), 'env_steps_per_s': E * N * T / (msq * 1e-3),
                                                'frac_of_8TBs': N * (E * T * B_STEP + B_STATE) / (msq * 1e-3) / 8e12}
  msg = timed(lambda j: env.rollout_episodes(sets[j % 4], out=outE), 20, 45)
  res['four_episode_groups_in_flight'] = {'us': msg * 1e3, 'env_steps_per_s': E * N * T / (msg * 1e-3), 'frac_of_8TBs': N * (E * T * B_STEP + B_STATE) / (msg * 1e-3) / 8e12}
  if lib_tag != 'ship':            # the experiments library also holds the stamped instantiation (impl 29)
    lib.earl_debug_set_rollout_impl(29)
    for _ in range(5):
      env.rollout(sets[0][0], out=out1, reset_first=True)
    torch.cuda.synchronize()
    buf = np.zeros(64 * 16, np.uint64)
    lib.earl_debug_read_ws_profile(buf.ctypes.data, buf.size)
    lib.earl_debug_set_rollout_impl(0)
    m = np.median(buf.reshape(64, 16).astype(np.float64), axis=0)
    names = ['C.first_barrier', 'C.lds_read', 'C.compute', 'C.barrier', 'C.total', 'L.process+issue', 'L.barrier', 'L.total', 'S0.store', 'S0.barrier', 'S0.total',
             'Slast.store', 'Slast.barrier', 'C1.barrier', 'Llast.process', 'Llast.barrier']
    res['stamps_ticks_per_step'] = {nm: m[k] / T for k, nm in enumerate(names)}
  print('RESULT ' + json.dumps(res), flush=True)


def main():
  if len(sys.argv) > 1:
    return child(sys.argv[1])
  out = {}
  for tag in ('ship', 'nostores'):
    r = subprocess.run([sys.executable, os.path.abspath(__file__), tag], capture_output=True, text=True, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('RESULT ')]
    out[tag] = json.loads(lines[-1][7:]) if lines else {'error': r.stderr[-800:]}
  print(json.dumps(out, indent=1))


if __name__ == '__main__':
  main()
