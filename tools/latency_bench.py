#!/usr/bin/env python
"""Inference latency / throughput by batch size (ICVL S=2 F=128): every op its own launch, the hourglass bottoms fused
(hg_fused.h, the default), and the fused path replayed from a captured graph.

    python tools/latency_bench.py > gpurun_out/latency.md
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(B, graphs, fused=True):
    env = dict(os.environ, DR_GRAPHS='1' if graphs else '0', DR_FUSE_TAIL='1' if fused else '0')       # library defaults: graphs off, fusion on
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--mode', 'infer', '--batch', str(B), '--steps', '50',
                          '--warmup', '10', '--no-cpu-baseline', '--no-profile', '--replicas', '1', '--merge', '1'], env=env, capture_output=True, text=True).stdout
    return json.loads(out.strip().splitlines()[-1])


def main():
    print('| batch | unfused ms/step | crops/s | fused hourglass bottoms ms/step | crops/s | + graph replay ms/step | crops/s |')
    print('|---:|---:|---:|---:|---:|---:|---:|')
    for B in (int(v) for v in os.environ.get('LAT_BATCHES', '1,2,4,8,16,40').split(',')):
        u, a, b = run(B, False, False), run(B, False), run(B, True)
        print('| %d | %.3f | %.0f | %.3f | %.0f | %.3f | %.0f |' % (B, u['ms_per_step'], u['value'], a['ms_per_step'], a['value'], b['ms_per_step'], b['value']))
        sys.stdout.flush()


if __name__ == '__main__':
    main()
