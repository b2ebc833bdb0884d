"""PNG depth frames: container parsing and inflate on the host (zlib), row filters in ``dr_png_unfilter`` (C), samples to
an fp32 depth frame on the device (``dr_depth_from_samples``) -- the work of ``tf.image.decode_png`` in
``data/nyu.py:148-156`` (RGB8, depth = G<<8 | B), ``data/icvl.py`` / ``data/msra.py:190-191`` (16-bit grey).

Only what those files contain is accepted: non-interlaced, colour type 0 (grey) or 2 (RGB), bit depth 8 or 16.
``encode_png`` writes the same subset (``MsraDataset.cvtBin2Png``: ``cv2.imwrite(path, dm.astype('uint16'))``,
``data/msra.py:120-149``); its optional ``filter_type`` exists so that tests can exercise every filter.
"""
from __future__ import annotations

import ctypes as C
import struct
import zlib
from typing import NamedTuple

import numpy as np

_SIG = b'\x89PNG\r\n\x1a\n'


class PngError(ValueError):
    pass


class PngInfo(NamedTuple):
    width: int
    height: int
    bit_depth: int
    channels: int

    @property
    def bpp(self) -> int:
        return self.channels * self.bit_depth // 8

    @property
    def row_bytes(self) -> int:
        return self.width * self.bpp


def _chunks(data: bytes):
    if data[:8] != _SIG:
        raise PngError('not a PNG stream')
    pos = 8
    while pos + 8 <= len(data):
        n, = struct.unpack('>I', data[pos:pos + 4])
        kind = data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + n]
        if len(body) != n or pos + 12 + n > len(data):
            raise PngError('truncated %r chunk' % kind)
        crc, = struct.unpack('>I', data[pos + 8 + n:pos + 12 + n])
        if zlib.crc32(kind + body) & 0xFFFFFFFF != crc:
            raise PngError('bad CRC in %r chunk' % kind)
        yield kind, body
        pos += 12 + n
        if kind == b'IEND':
            return
    raise PngError('no IEND chunk')


def inflate_png(data: bytes):
    """-> (PngInfo, filtered scanlines as bytes: height x (1 + row_bytes))."""
    info, idat = None, []
    for kind, body in _chunks(data):
        if kind == b'IHDR':
            w, h, depth, ctype, comp, flt, lace = struct.unpack('>IIBBBBB', body)
            if comp or flt or lace:
                raise PngError('interlaced / non-standard PNG')
            if ctype not in (0, 2) or depth not in (8, 16):
                raise PngError('colour type %d / bit depth %d: not a depth-frame PNG' % (ctype, depth))
            info = PngInfo(w, h, depth, 1 if ctype == 0 else 3)
        elif kind == b'IDAT':
            idat.append(body)
    if info is None or not idat:
        raise PngError('missing IHDR / IDAT')
    raw = zlib.decompress(b''.join(idat))
    if len(raw) != info.height * (1 + info.row_bytes):
        raise PngError('IDAT inflates to %d bytes, expected %d' % (len(raw), info.height * (1 + info.row_bytes)))
    return info, raw


def decode_png(data: bytes):
    """-> (PngInfo, samples uint8 [height][row_bytes]) -- big-endian byte order for 16-bit samples, as stored."""
    from .. import _lib
    info, raw = inflate_png(data)
    out = np.empty((info.height, info.row_bytes), np.uint8)
    rc = _lib.load().dr_png_unfilter(raw, info.height, info.row_bytes, info.bpp, out.ctypes.data)   # bytes: passed by pointer, no copy
    if rc != 0:
        raise PngError('dr_png_unfilter failed (%d): unknown filter type' % rc)
    return info, out


def encode_png(img: np.ndarray, filter_type: int = 0, level: int = 6) -> bytes:
    """uint8 (H,W) / (H,W,3) or uint16 (H,W) -> PNG bytes; one filter type for every row (0..4)."""
    a = np.asarray(img)
    if a.dtype == np.uint16 and a.ndim == 2:
        samples, depth, ctype, bpp = a.astype('>u2').view(np.uint8).reshape(a.shape[0], -1), 16, 0, 2
    elif a.dtype == np.uint8 and a.ndim == 2:
        samples, depth, ctype, bpp = a, 8, 0, 1
    elif a.dtype == np.uint8 and a.ndim == 3 and a.shape[2] == 3:
        samples, depth, ctype, bpp = a.reshape(a.shape[0], -1), 8, 2, 3
    else:
        raise PngError('encode_png: uint8 (H,W[,3]) or uint16 (H,W) only')
    h, rb = samples.shape
    s = samples.astype(np.int32)
    left = np.zeros_like(s); left[:, bpp:] = s[:, :-bpp]
    up = np.zeros_like(s); up[1:] = s[:-1]
    ul = np.zeros_like(s); ul[1:, bpp:] = s[:-1, :-bpp]
    if filter_type == 0:
        f = s
    elif filter_type == 1:
        f = s - left
    elif filter_type == 2:
        f = s - up
    elif filter_type == 3:
        f = s - ((left + up) >> 1)
    elif filter_type == 4:
        p = left + up - ul
        pa, pb, pc = np.abs(p - left), np.abs(p - up), np.abs(p - ul)
        pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, up, ul))
        f = s - pred
    else:
        raise PngError('filter type %d' % filter_type)
    rows = np.empty((h, rb + 1), np.uint8)
    rows[:, 0] = filter_type
    rows[:, 1:] = (f & 0xFF).astype(np.uint8)

    def chunk(kind, body):
        return struct.pack('>I', len(body)) + kind + body + struct.pack('>I', zlib.crc32(kind + body) & 0xFFFFFFFF)
    w = a.shape[1]
    return _SIG + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, depth, ctype, 0, 0, 0)) + \
        chunk(b'IDAT', zlib.compress(rows.tobytes(), level)) + chunk(b'IEND', b'')


def depth_from_samples(samples, info: PngInfo, out=None):
    """Device step: uint8 sample tensor(s) of one or more frames (uploaded as they are) -> fp32 depth (..., H, W) in mm."""
    import torch
    from .. import _lib
    if info.channels == 3 and info.bit_depth == 8:
        mode, per = 0, 3
    elif info.channels == 1 and info.bit_depth == 16:
        mode, per = 1, 2
    else:
        raise PngError('depth frames are RGB8 (NYU) or 16-bit grey (ICVL, MSRA)')
    assert samples.is_cuda and samples.dtype == torch.uint8 and samples.is_contiguous()
    npix = samples.numel() // per
    if out is None:
        out = torch.empty(npix, dtype=torch.float32, device=samples.device)
    rc = _lib.load().dr_depth_from_samples(samples.data_ptr(), npix, mode, out.data_ptr(),
                                           C.c_void_p(torch.cuda.current_stream(samples.device).cuda_stream))
    if rc != 0:
        raise PngError('dr_depth_from_samples failed (%d)' % rc)
    return out.view(-1, info.height, info.width)
