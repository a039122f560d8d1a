"""CPU: the part of bench.py's contract that does not need a GPU - rank 0 prints ONE JSON line on stdout and nothing else,
whoever else writes to stdout meanwhile (Python prints of modules built from a shipped config, native libraries such as gloo
announcing its peers on fd 1, child processes)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_only_the_line_reaches_stdout():
    code = '\n'.join([
        'import os, subprocess, sys',
        'sys.path.insert(0, %r)' % ROOT,
        'import bench',
        'out = bench.claim_stdout()',
        'os.write(1, b"native library noise on fd 1\\n")',
        'print("python print")',
        'subprocess.call([sys.executable, "-c", "print(\'child process noise\')"])',
        'print(\'{"metric": "x", "value": 1}\', file=out)',
        'out.flush()',
    ])
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout == '{"metric": "x", "value": 1}\n', r.stdout
    for noise in ('native library noise', 'python print', 'child process noise'):
        assert noise in r.stderr


def test_help_goes_to_stdout():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--help'], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and '--gpus' in r.stdout and '--steps' in r.stdout and '--warmup' in r.stdout
