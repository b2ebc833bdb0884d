#!/usr/bin/env python
"""Throughput of the dataset decode pipeline in front of the network (densereg_amd/data): per stage, one host thread.

    python tools/dataio_bench.py > gpurun_out/dataio_bench.md
"""
import io
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from densereg_amd.data import png, tfrecord  # noqa: E402


def frame(rng, h, w, base):
    yy, xx = np.mgrid[0:h, 0:w]
    d = base + 80 * np.sin(xx / 9.0) * np.cos(yy / 11.0) + rng.normal(0, 3, (h, w))
    d[(yy - h / 2) ** 2 + (xx - w / 2) ** 2 > (min(h, w) * 0.35) ** 2] = 0
    return np.clip(d, 0, 65535).astype(np.uint16)


def main():
    from PIL import Image
    rng = np.random.default_rng(0)
    dev = torch.device('cuda', 0)
    print('| frames | PNG bytes/frame | inflate + row filters (host, 1 thread) | upload + unpack (device) | unpack kernel alone |')
    print('|---|---:|---:|---:|---:|')
    for name, (h, w), rgb in (('ICVL / MSRA 320x240 grey16', (240, 320), False), ('NYU 640x480 RGB8 (G<<8|B)', (480, 640), True)):
        blobs = []
        for i in range(16):
            d = frame(rng, h, w, 500 + 10 * i)
            a = d
            if rgb:
                a = np.zeros((h, w, 3), np.uint8); a[..., 1] = d >> 8; a[..., 2] = d & 0xFF
            buf = io.BytesIO(); Image.fromarray(a).save(buf, format='PNG'); blobs.append(buf.getvalue())
        t0 = time.perf_counter()
        dec = [png.decode_png(b) for b in blobs * 4]
        t_host = (time.perf_counter() - t0) / (len(blobs) * 4)
        info = dec[0][0]
        stack = np.stack([s for _, s in dec[:40]])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            frames = png.depth_from_samples(torch.from_numpy(stack).to(dev), info)
        torch.cuda.synchronize()
        t_dev = (time.perf_counter() - t0) / 20 / stack.shape[0]
        d_s = torch.from_numpy(stack).to(dev)
        out = torch.empty(stack.shape[0] * h * w, dtype=torch.float32, device=dev)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50):
            png.depth_from_samples(d_s, info, out=out)
        b.record(); torch.cuda.synchronize()
        t_k = a.elapsed_time(b) * 1e-3 / 50
        nbytes = stack.size + 4 * stack.shape[0] * h * w
        print('| %s | %d | %.2f ms/frame (%.0f frames/s) | %.3f ms/frame | %.1f us per %d frames = %.0f GB/s |' % (
            name, int(np.mean([len(b) for b in blobs])), t_host * 1e3, 1 / t_host, t_dev * 1e3, t_k * 1e6, stack.shape[0], nbytes / t_k / 1e9))


if __name__ == '__main__':
    main()
