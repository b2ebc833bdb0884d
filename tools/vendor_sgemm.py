#!/usr/bin/env python
"""Context for the conv roofline fraction: what the vendor fp32 GEMM (torch.matmul -> rocBLAS / hipBLASLt, fp32 MFMA) sustains on the
GEMM shapes of the big layers (M = B*H*W = 40960; a 3x3 conv as its im2col GEMM, which the library gets for free -- no gather,
no halo, no epilogue).  python tools/vendor_sgemm.py"""
import time

import torch


def main():
    dev = torch.device('cuda', 0)
    torch.backends.cuda.matmul.allow_tf32 = False
    print('| M | K | N | what | us | TFLOP/s |')
    print('|---:|---:|---:|---|---:|---:|')
    for M, K, N, what in ((40960, 512, 512, '1x1 512->512'), (40960, 2304, 256, '3x3 256->256 as im2col GEMM'), (40960, 256, 512, '1x1 256->512'),
                          (40960, 512, 256, '1x1 512->256'), (40960, 1152, 128, '3x3 128->128 as im2col GEMM'), (40960, 128, 128, '1x1 128->128'),
                          (40960, 256, 128, '1x1 256->128'), (163840, 512, 512, '1x1 512->512 at 64x64 (config 5)')):
        a = torch.randn(M, K, device=dev)
        b = torch.randn(K, N, device=dev)
        for _ in range(5):
            c = a @ b
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        it = 30
        for _ in range(it):
            c = a @ b
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / it
        print('| %d | %d | %d | %s | %.1f | %.1f |' % (M, K, N, what, us, 2.0 * M * K * N / us / 1e6))


if __name__ == '__main__':
    main()
