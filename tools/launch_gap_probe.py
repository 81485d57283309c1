"""Why does rocprofv3's kernel trace show the bench's tabletop launch ~250 us when HIP events around the same K back-to-back launches give ~297 us per launch?
On one MI355X, 4096 envs x 200 steps x 28 episodes (the default line's launch):
  A  events around K back-to-back launches (what bench.py times)        -> period per launch
  B  an event pair around EVERY launch                                   -> kernel-side duration and the gap to the next launch
  C  host time of the enqueue loop alone (no sync inside)                -> is the host the bound?
  D  launches with an idle stretch in between (torch.cuda._sleep)        -> does a launch run faster after a pause?
  E  the K launches captured in one HIP graph                            -> period without any host cost
usage (GPU box): python tools/launch_gap_probe.py"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N, T, E, K = 4096, 200, 28, 20


def main():
  import torch
  import earl_benchmark_amd as eb
  _, env = eb.EARLEnvs('tabletop_manipulation', reward_type='sparse', num_envs=N, eval_horizon=T, scalar_api=False, seed=0).get_envs()
  u = env.unwrapped
  g = torch.Generator(device='cuda').manual_seed(1)
  sets = [(torch.rand(E, T, N, 3, generator=g, device='cuda') * 2 - 1).contiguous() for _ in range(4)]
  out = u._new_out((E, T, N))[0]
  ev = lambda: torch.cuda.Event(enable_timing=True)
  res = {}
  for j in range(10):
    env.rollout_episodes(sets[j % 4], out=out)
  torch.cuda.synchronize()

  def period(reps=5):
    ps = []
    for _ in range(reps):
      e0, e1 = ev(), ev()
      e0.record()
      for j in range(K):
        env.rollout_episodes(sets[j % 4], out=out)
      e1.record(); torch.cuda.synchronize()
      ps.append(e0.elapsed_time(e1) * 1e3 / K)
    return sorted(ps)[len(ps) // 2]
  res['A_period_us'] = period()

  pairs = [(ev(), ev()) for _ in range(K)]
  t0 = time.perf_counter()
  for j, (a, b) in enumerate(pairs):
    a.record(); env.rollout_episodes(sets[j % 4], out=out); b.record()
  t1 = time.perf_counter()
  torch.cuda.synchronize()
  d = [a.elapsed_time(b) * 1e3 for a, b in pairs]
  gaps = [pairs[i][1].elapsed_time(pairs[i + 1][0]) * 1e3 for i in range(K - 1)]
  res['B_duration_us'] = {'median': sorted(d)[K // 2], 'min': min(d), 'max': max(d)}
  res['B_gap_us'] = {'median': sorted(gaps)[len(gaps) // 2], 'min': min(gaps), 'max': max(gaps)}
  res['B_host_enqueue_us_per_launch_with_events'] = (t1 - t0) * 1e6 / K

  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for j in range(K):
    env.rollout_episodes(sets[j % 4], out=out)
  t1 = time.perf_counter()
  torch.cuda.synchronize()
  res['C_host_enqueue_us_per_launch'] = (t1 - t0) * 1e6 / K

  for cyc in (20000, 100000, 400000):           # torch.cuda._sleep spins that many device clock ticks
    pairs = [(ev(), ev()) for _ in range(K)]
    for j, (a, b) in enumerate(pairs):
      torch.cuda._sleep(cyc)
      a.record(); env.rollout_episodes(sets[j % 4], out=out); b.record()
    torch.cuda.synchronize()
    d = [a.elapsed_time(b) * 1e3 for a, b in pairs]
    res[f'D_duration_after_sleep_{cyc}_us'] = {'median': sorted(d)[K // 2], 'min': min(d), 'max': max(d)}

  s = torch.cuda.Stream()
  with torch.cuda.stream(s):
    for j in range(3):
      env.rollout_episodes(sets[j % 4], out=out)
    s.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=s):
      for j in range(K):
        env.rollout_episodes(sets[j % 4], out=out)
    gr.replay(); s.synchronize()
    ps = []
    for _ in range(5):
      e0, e1 = ev(), ev()
      e0.record(s); gr.replay(); e1.record(s); s.synchronize()
      ps.append(e0.elapsed_time(e1) * 1e3 / K)
    res['E_graph_period_us'] = sorted(ps)[2]
  res['A_period_us_again'] = period()
  print(json.dumps(res, indent=1))


if __name__ == '__main__':
  main()
