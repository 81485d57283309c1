"""What this MI355X sustains for the traffic mixes of the env kernels: pure write, copy (1:1), and 18 % read / 82 % write
(the fused rollout writes 54 B and reads 12 B per env step).  torch kernels only: a yardstick, not the product."""
import torch


def timeit(fn, iters=20):
  for _ in range(3):
    fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters):
    fn()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / iters * 1e-3


def main():
  n = 1 << 30                                    # 1 GiB buffers
  a = torch.empty(n, dtype=torch.uint8, device='cuda').view(torch.float32)
  b = torch.empty_like(a)
  small = torch.empty(n * 12 // 54 // 4, dtype=torch.float32, device='cuda')   # read side of the 12:54 mix
  t = timeit(lambda: a.fill_(1.0))
  print(f'pure write (fill_)            {n / t / 1e12:.2f} TB/s')
  t = timeit(lambda: b.copy_(a))
  print(f'copy 1:1 (copy_)              {2 * n / t / 1e12:.2f} TB/s')
  t = timeit(lambda: torch.add(a, 1.0, out=b))
  print(f'read + write (add, out=)      {2 * n / t / 1e12:.2f} TB/s')
  t = timeit(lambda: a.sum())
  print(f'pure read (sum)               {n / t / 1e12:.2f} TB/s')
  # 12 B read : 54 B written -- a fill of the big buffer plus a read of a buffer 12/54 its size, back to back
  t = timeit(lambda: (a.fill_(2.0), small.sum()))
  print(f'12:54 read:write mix (two kernels, back to back) {(n + small.numel() * 4) / t / 1e12:.2f} TB/s')


if __name__ == '__main__':
  main()
