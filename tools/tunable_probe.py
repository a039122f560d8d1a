import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import microbench as mb
m = 90107
x = torch.randn(m, 128, device='cuda'); h = torch.randn(m, 256, device='cuda')
shapes = []
for out in (128, 256):
    w = torch.randn(out, 128, device='cuda'); b = torch.randn(out, device='cuda')
    shapes.append((f'addmm {m}x128 @ 128x{out}', lambda w=w, b=b: torch.addmm(b, x, w.t())))
    dy = torch.randn(m, out, device='cuda')
    shapes.append((f'dgrad {m}x{out} @ {out}x128', lambda dy=dy, w=w: dy @ w))
w2 = torch.randn(128, 256, device='cuda'); b2 = torch.randn(128, device='cuda')
shapes.append((f'addmm {m}x256 @ 256x128', lambda: torch.addmm(b2, h, w2.t())))
dy2 = torch.randn(m, 128, device='cuda')
shapes.append((f'dgrad {m}x128 @ 128x256', lambda: dy2 @ w2))
t0 = time.time()
for name, fn in shapes:
    fn(); torch.cuda.synchronize()
print('first-call (tuning) time %.1f s' % (time.time() - t0))
for name, fn in shapes:
    med, mn = mb.timeit(fn, iters=20, warmup=3)
    print(f'{name:34s} {med*1e3:7.1f} us (min {mn*1e3:.1f})')
