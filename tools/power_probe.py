#!/usr/bin/env python
"""Board power and shader clock while ONE kernel runs back to back: is the x3 family held by a power / clock budget?

A child process launches the kernel in a loop for a few seconds (dr_dbg_conv_bench on the debug library); this process samples
`rocm-smi --showpower --showclocks --json` ten times a second and reports the median of the samples taken while the child was running.

    python tools/power_probe.py [seconds per kernel, default 4]
"""
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [  # (label, x3 mode, B, hw, cin, cout, k[, abl])
    ('idle', None, 0, 0, 0, 0, 0),
    ('conv_x3h 3x3 256->256 (halo kernel)', 2, 200, 32, 256, 256, 3),
    ('conv_x3 3x3 256->256 (one tap at a time)', 7, 200, 32, 256, 256, 3),
    ('conv_x3 1x1 512->512', 2, 200, 32, 512, 512, 1),
    ('conv_x3 1x1 128->256', 2, 200, 32, 128, 256, 1),
    ('conv_x3h 3x3 256->256, ALL-ZERO operands', 2, 200, 32, 256, 256, 3, 6),
    ('conv_x3 1x1 512->512, ALL-ZERO operands', 2, 200, 32, 512, 512, 1, 6),
    ('fp32 MFMA 3x3 256->256', 0, 200, 32, 256, 256, 3),
    ('fp32 MFMA 1x1 512->512', 0, 200, 32, 512, 512, 1),
]
CHILD = r'''
import ctypes as C, sys, time, torch
sys.path.insert(0, %r)
from densereg_amd import _lib
dbg = _lib.load_debug(); ms = C.c_float()
mode, B, hw, cin, cout, k, secs, abl = [int(v) for v in sys.argv[1:9]]
dbg.dr_dbg_force_x3(mode)
t0 = time.time(); n = 0; tot = 0.0
print('READY', flush=True)
while time.time() - t0 < secs:
    assert dbg.dr_dbg_conv_bench(B, hw, hw, cin, cout, k, -1, abl, 200, C.byref(ms)) == 0
    n += 1; tot += ms.value
print('US', tot / n * 1e3, flush=True)
'''


def sample():
    try:
        out = subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--json'], capture_output=True, text=True, timeout=5).stdout
        d = json.loads(out)
        card = d[sorted(d)[0]]
        power = sclk = mclk = None
        for key, val in card.items():
            kl = key.lower()
            if 'power' in kl and power is None:
                try: power = float(str(val).split()[0])
                except ValueError: pass
            if 'sclk' in kl and 'speed' in kl:
                sclk = str(val)
            if 'mclk' in kl and 'speed' in kl:
                mclk = str(val)
        return power, sclk, mclk, card
    except Exception as e:  # noqa: BLE001
        return None, None, None, {'error': str(e)}


def mhz(s):
    import re
    m = re.search(r'\((\d+)\s*Mhz\)', s or '', re.I)
    return float(m.group(1)) if m else None


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
    p0, s0, m0, card = sample()
    print('# board power and clocks while one kernel runs back to back (rocm-smi, 10 samples / s, median)\n')
    print('first sample, raw keys: %s\n' % ', '.join('%s=%s' % kv for kv in list(card.items())[:12]))
    print('| kernel (200 crops at 32x32) | us / launch | TFLOP/s | board power W | sclk MHz | mclk MHz | samples |')
    print('|---|---:|---:|---:|---:|---:|---:|')
    for case in CASES:
        label, mode, B, hw, cin, cout, k = case[:7]
        abl = case[7] if len(case) > 7 else 0
        pw, sc, mc = [], [], []
        us = None
        if mode is None:
            t0 = time.time()
            while time.time() - t0 < 2.0:
                p, s, m, _ = sample()
                if p is not None: pw.append(p)
                if mhz(s): sc.append(mhz(s))
                if mhz(m): mc.append(mhz(m))
                time.sleep(0.1)
        else:
            ch = subprocess.Popen([sys.executable, '-c', CHILD % ROOT] + [str(v) for v in (mode, B, hw, cin, cout, k, int(secs), abl)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
            ch.stdout.readline()                      # READY (the operands are on the device, the loop starts)
            time.sleep(1.0)                           # let the clocks settle
            while ch.poll() is None:
                p, s, m, _ = sample()
                if ch.poll() is not None: break
                if p is not None: pw.append(p)
                if mhz(s): sc.append(mhz(s))
                if mhz(m): mc.append(mhz(m))
                time.sleep(0.1)
            rest = ch.stdout.read()
            for line in rest.splitlines():
                if line.startswith('US'): us = float(line.split()[1])
        fl = 2.0 * B * hw * hw * k * k * cin * cout
        med = lambda v: ('%.0f' % statistics.median(v)) if v else '-'
        print('| %s | %s | %s | %s | %s | %s | %d |' % (label, '%.1f' % us if us else '-', '%.1f' % (fl / us / 1e6) if us else '-', med(pw), med(sc), med(mc), len(pw)))
        sys.stdout.flush()


if __name__ == '__main__':
    main()
