#!/usr/bin/env python3
"""VERDICT r04 item 4 "no scratch inside any timestep loop": where do the rollout kernels' scratch (spill) instructions sit?  Compiles the five stepper units (csrc/physics.hip, physics_kitchen.hip, physics_mt.hip, physics_w8.hip, physics_l64.hip) to
gfx950 assembly (device only), finds every loop of each rollout kernel (a backward branch to a label), nests them by extent and reports the scratch_load / scratch_store count of the
kernel, of its env-step loop (the largest loop) and of its TIMESTEP loop (the largest loop strictly inside the env-step loop that holds more than a third of it: the stepper's body).
  python tools/scratch_in_loops.py > profiles/r05_scratch_report.txt          (no GPU needed; ~2 min)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'earl_benchmark_amd', 'csrc')
FLAGS = '--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fPIC --cuda-device-only -S'.split()
KERNELS = ('sawyer_rollout_kernel', 'kitchen_rollout_kernel', 'minitaur_kernel', 'minitaur_duo_kernel', 'physics_kernel')


def loops_of(body):
  labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
  out = []
  for i, l in enumerate(body):
    m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
      out.append((labels[m.group(1)], i))
  return sorted(out, key=lambda ab: ab[0] - ab[1])


def main():
  for unit in ('physics.hip', 'physics_kitchen.hip', 'physics_mt.hip', 'physics_w8.hip', 'physics_l64.hip'):
    with tempfile.NamedTemporaryFile(suffix='.s') as f:
      subprocess.run(['/opt/rocm/bin/hipcc'] + FLAGS + ['-o', f.name, os.path.join(CSRC, unit)], check=True, stderr=subprocess.DEVNULL)
      lines = open(f.name).read().split('\n')
    demangle = lambda s: subprocess.run(['c++filt', s], capture_output=True, text=True).stdout.strip()
    i = 0
    while i < len(lines):
      m = re.match(r'^(_Z\w+):\s', lines[i])
      if m and any(k in m.group(1) for k in KERNELS):
        end = next(j for j in range(i, len(lines)) if lines[j].startswith('.Lfunc_end'))
        body = lines[i:end]
        sc = [j for j, l in enumerate(body) if 'scratch_load' in l or 'scratch_store' in l]
        name = demangle(m.group(1)).replace('(anonymous namespace)::', '')
        lp = loops_of(body)
        if not sc:
          print(f'{unit}: {name}: no scratch at all ({len(body)} lines)')
        elif not lp:
          print(f'{unit}: {name}: {len(sc)} scratch instructions, no loop')
        elif 'minitaur_duo_kernel' in name:
          # the two-waves-per-SIMD kernel: ONE loop over slots, each iteration one half-timestep per role (no inner loop of stepper size); its dynamics half sits at the 256-register
          # cap and keeps a few spills: reported as they are (round 6), bounded by tests/test_no_scratch_in_timestep_loops.py
          a, b = lp[0]
          n_ld = sum(a <= j <= b and 'scratch_load' in body[j] for j in sc); n_st = sum(a <= j <= b and 'scratch_store' in body[j] for j in sc)
          print(f'{unit}: {name}: {len(sc)} scratch instructions in {len(body)} lines; slot loop (a half-timestep per role and iteration) lines {a}-{b}: {n_ld} loads, {n_st} stores')
        else:
          a, b = lp[0]
          inner = [(x, y) for x, y in lp if x > a and y < b and (y - x) > (b - a) / 3]
          ts = min(inner, key=lambda xy: xy[1] - xy[0]) if inner else None
          n_outer = sum(a <= j <= b for j in sc)
          n_ts = sum(ts[0] <= j <= ts[1] for j in sc) if ts else None
          print(f'{unit}: {name}: {len(sc)} scratch instructions in {len(body)} lines; outermost loop (env steps) lines {a}-{b}: {n_outer}; '
                + (f'timestep loop lines {ts[0]}-{ts[1]} ({ts[1] - ts[0]} lines): {n_ts}' if ts else 'no inner loop of stepper size (the timestep is inlined into the outer loop)'))
        i = end
      i += 1


if __name__ == '__main__':
  main()
