"""Weight-gradient kernel timing per shape and mode.  Arguments are NAME=VALUE settings of the SST_WGRAD_* switches
(e.g. `TILED=0 TILED=1`); each runs in a fresh process (the switches are read once per process).  Times are per call
(split-K kernel + its reduce kernel), averaged over back-to-back launches between two events."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def one():
    import torch
    from sst_amd.dense import weight_bias_grad
    m = int(os.environ.get('WG_M', 90107))
    res = []
    for out, inn in ((256, 128), (128, 128), (128, 256), (384, 128)):
        dy = torch.randn(m, out, device='cuda')
        x = torch.randn(m, inn, device='cuda')
        for _ in range(5):
            weight_bias_grad(dy, x, True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(100):
            weight_bias_grad(dy, x, True)
        e1.record()
        torch.cuda.synchronize()
        res.append('%dx%d %.1f us' % (out, inn, e0.elapsed_time(e1) * 10))
    print('%s: %s' % (os.environ.get('WG_TAG', 'default'), ', '.join(res)), flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'one':
        one()
    else:
        for setting in (sys.argv[1:] or ['TILED=1']):
            name, value = setting.split('=')
            env = dict(os.environ, WG_TAG=setting)
            env['SST_WGRAD_' + name] = value
            subprocess.run([sys.executable, __file__, 'one'], env=env, check=True)
