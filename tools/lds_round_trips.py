"""Where a stepper kernel pays LDS latency one load at a time.  In the one-wave-per-SIMD kernels every `s_waitcnt lgkmcnt(0)` that follows only one or two
ds_read instructions is a full LDS round trip (64+ cycles) that nothing overlaps; a loop of conditional loads / stores unrolled into "read, wait, use" groups is a chain
of them.  Compiles a translation unit with -DEARL_PHYS_MARK (phase markers in the assembly) and prints, per phase of a kernel: instructions, LDS reads, waits, and
the number of POOR waits (a wait for everything outstanding with <= 2 reads in flight).

  python tools/lds_round_trips.py physics_kitchen.hip kitchen_rollout_kernelILb0
  python tools/lds_round_trips.py physics_mt.hip minitaur_kernelILb1ELb1
  python tools/lds_round_trips.py physics.hip sawyer_rollout_kernelILi15ELi16ELb0
"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'earl_benchmark_amd', 'csrc')
FLAGS = '--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fPIC --cuda-device-only -S -DEARL_PHYS_MARK'.split()


def main():
  unit, key = sys.argv[1], sys.argv[2]
  with tempfile.NamedTemporaryFile(suffix='.s') as f:
    subprocess.run(['/opt/rocm/bin/hipcc'] + FLAGS + ['-o', f.name, os.path.join(CSRC, unit)], check=True, stderr=subprocess.DEVNULL)
    txt = open(f.name).read()
  m = re.search(r'^(_Z\S*' + re.escape(key) + r'\S*):', txt, re.M)
  if not m:
    sys.exit(f'no kernel matching {key} in {unit}')
  body = txt[m.start():txt.index('.Lfunc_end', m.start())].splitlines()
  print(m.group(1))
  prev, stats = 'start', None
  rows = []

  def flush(name_to):
    if stats:
      rows.append((prev, name_to, dict(stats)))

  stats = dict(instr=0, reads=0, waits=0, poor=0, guarded=0)
  inflight = 0
  for k, ln in enumerate(body):
    mk = re.search(r'; EARL_PHASE_(END K?\d+|START)', ln)
    if mk:
      flush(mk.group(1))
      prev = mk.group(1)
      stats = dict(instr=0, reads=0, waits=0, poor=0, guarded=0)
      continue
    t = ln.strip()
    if not ln.startswith('\t') or t.startswith(('.', ';')):
      continue
    stats['instr'] += 1
    op = t.split()[0]
    if op.startswith('ds_read'):
      stats['reads'] += 1
      inflight += 1
    elif op == 's_waitcnt' and 'lgkmcnt' in t:
      stats['waits'] += 1
      n = int(re.search(r'lgkmcnt\((\d+)\)', t).group(1))
      if n == 0:
        if 0 < inflight <= 2:
          stats['poor'] += 1
        inflight = 0
      else:
        inflight = min(inflight, n)
    elif op.startswith('s_cbranch_exec'):
      for q in range(k + 1, min(k + 40, len(body))):          # a branch around a load that waits for itself
        l2 = body[q].strip()
        if l2.startswith('.LBB'):
          break
        if l2.startswith('ds_read'):
          stats['guarded'] += 1
          break
  flush('end')
  print(f'{"phase (from -> to)":>24} {"instr":>7} {"ds_read":>8} {"waits":>6} {"poor waits":>11} {"branch-guarded loads":>21}')
  for a, b, s in rows:
    print(f'{a + " -> " + b:>24} {s["instr"]:7d} {s["reads"]:8d} {s["waits"]:6d} {s["poor"]:11d} {s["guarded"]:21d}')
  print(f'{"total":>24} {sum(s["instr"] for *_, s in rows):7d} {sum(s["reads"] for *_, s in rows):8d} {sum(s["waits"] for *_, s in rows):6d} '
        f'{sum(s["poor"] for *_, s in rows):11d} {sum(s["guarded"] for *_, s in rows):21d}')


if __name__ == '__main__':
  main()
