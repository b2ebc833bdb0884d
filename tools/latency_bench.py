#!/usr/bin/env python
"""Inference latency / throughput by batch size, graph replay vs plain launches (ICVL S=2 F=128).

    python tools/latency_bench.py > gpurun_out/latency.md
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(B, graphs):
    env = dict(os.environ, DR_GRAPHS='1' if graphs else '0')       # the library default is off
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--mode', 'infer', '--batch', str(B), '--steps', '50',
                          '--warmup', '10', '--no-cpu-baseline', '--no-profile', '--replicas', '1', '--merge', '1'], env=env, capture_output=True, text=True).stdout
    return json.loads(out.strip().splitlines()[-1])


def main():
    print('| batch | plain launches ms/step | crops/s | graph replay ms/step | crops/s |')
    print('|---:|---:|---:|---:|---:|')
    for B in (1, 2, 4, 8, 16, 40):
        a, b = run(B, False), run(B, True)
        print('| %d | %.3f | %.0f | %.3f | %.0f |' % (B, a['ms_per_step'], a['value'], b['ms_per_step'], b['value']))
        sys.stdout.flush()


if __name__ == '__main__':
    main()
