"""How fast can this GPU stream the benchmark's arrays? (reference points for the roofline)"""
import torch, time
dev = torch.device('cuda')
n = 512 ** 3
pos = torch.rand((n, 3), dtype=torch.float64, device=dev)
out = torch.empty(n, dtype=torch.float64, device=dev)
idx = torch.empty(n, dtype=torch.int32, device=dev)
def timeit(f, reps=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps
t = timeit(lambda: pos.sum()); print('sum(pos)         read 3.2 GB : %.3f ms  %.2f TB/s' % (t * 1e3, 3.22e9 / t / 1e12))
p2 = torch.empty_like(pos)
t = timeit(lambda: p2.copy_(pos)); print('copy(pos)        r+w 6.4 GB  : %.3f ms  %.2f TB/s' % (t * 1e3, 6.44e9 / t / 1e12))
t = timeit(lambda: torch.mul(pos[:, 0], 2.0, out=out)); print('strided col read 1 GB(+2 GB skipped) + write 1 GB: %.3f ms' % (t * 1e3))
t = timeit(lambda: out.zero_()); print('zero 1 GB        : %.3f ms  %.2f TB/s' % (t * 1e3, 1.07e9 / t / 1e12))
x = pos.reshape(-1)
t = timeit(lambda: torch.add(x[:n], 1.0, out=out)); print('add 1 GB r + 1 GB w: %.3f ms %.2f TB/s' % (t * 1e3, 2.15e9 / t / 1e12))
