"""CPU oracle of the dataset file formats in front of the crop front-end.

TEST INFRASTRUCTURE ONLY.  Plain Python / numpy restatement of what the reference obtains from TensorFlow ops:

* ``tf.image.decode_png``            data/nyu.py:148-149, data/icvl.py parse_example, data/msra.py:190
  -> ``png_decode`` (chunk walk, zlib inflate, the five row filters of PNG specification section 9.2, written from the
  specification, one byte at a time)
* depth from the decoded samples      data/nyu.py:151-156 (``(G << 8) | B``), ``tf.to_float`` of uint16 for ICVL / MSRA
* ``tf.TFRecordReader`` framing and ``tf.parse_single_example``       data/dataset_base.py:166-178, data/nyu.py:180-205
  -> ``records`` / ``example_features`` (bitwise CRC-32C, schema-free proto walk)

Pinning: PNG decoding is pinned against an independent codec (Pillow writes the files the tests decode, and decodes
the files ``densereg_amd.data.png.encode_png`` writes); CRC-32C against the RFC 3720 check value; the Example wire format
against a hand-assembled message.  PARITY UNPINNED for what only TensorFlow could confirm: that a TF-1.3-written
TFRecord file parses (none exists in this environment).
"""
import struct
import zlib

import numpy as np


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    if pa <= pb and pa <= pc:
        return a
    return b if pb <= pc else c


def png_unfilter(raw, height, row_bytes, bpp):
    out = np.zeros((height, row_bytes), np.uint8)
    prev = [0] * row_bytes
    pos = 0
    for y in range(height):
        ft = raw[pos]
        line = raw[pos + 1:pos + 1 + row_bytes]
        pos += 1 + row_bytes
        cur = [0] * row_bytes
        for i in range(row_bytes):
            a = cur[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            if ft == 0:
                pred = 0
            elif ft == 1:
                pred = a
            elif ft == 2:
                pred = b
            elif ft == 3:
                pred = (a + b) // 2
            elif ft == 4:
                pred = _paeth(a, b, c)
            else:
                raise ValueError('filter type %d' % ft)
            cur[i] = (line[i] + pred) & 0xFF
        out[y] = cur
        prev = cur
    return out


def png_decode(data):
    """-> (width, height, bit_depth, channels, samples uint8 [height][row_bytes])."""
    assert data[:8] == b'\x89PNG\r\n\x1a\n'
    pos, idat, hdr = 8, b'', None
    while pos < len(data):
        n, = struct.unpack('>I', data[pos:pos + 4])
        kind, body = data[pos + 4:pos + 8], data[pos + 8:pos + 8 + n]
        if kind == b'IHDR':
            hdr = struct.unpack('>IIBBBBB', body)
        elif kind == b'IDAT':
            idat += body
        pos += 12 + n
    w, h, depth, ctype = hdr[:4]
    ch = {0: 1, 2: 3}[ctype]
    bpp = ch * depth // 8
    return w, h, depth, ch, png_unfilter(zlib.decompress(idat), h, w * bpp, bpp)


def depth_from_samples(samples, channels, bit_depth):
    s = np.asarray(samples, np.uint8)
    if channels == 3 and bit_depth == 8:                       # nyu.py:151-156
        px = s.reshape(s.shape[0], -1, 3).astype(np.uint16)
        return ((px[..., 1] * 256) | px[..., 2]).astype(np.float32)
    if channels == 1 and bit_depth == 16:
        px = s.reshape(s.shape[0], -1, 2).astype(np.uint16)
        return ((px[..., 0] << 8) | px[..., 1]).astype(np.float32)
    raise ValueError('not a depth frame')


def crc32c(data, crc=0):
    c = crc ^ 0xFFFFFFFF
    for byte in data:
        c ^= byte
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
    return c ^ 0xFFFFFFFF


def masked_crc(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def records(blob):
    pos, out = 0, []
    while pos < len(blob):
        n, = struct.unpack('<Q', blob[pos:pos + 8])
        assert struct.unpack('<I', blob[pos + 8:pos + 12])[0] == masked_crc(blob[pos:pos + 8])
        body = blob[pos + 12:pos + 12 + n]
        assert struct.unpack('<I', blob[pos + 12 + n:pos + 16 + n])[0] == masked_crc(body)
        out.append(body)
        pos += 16 + n
    return out


def _varint(buf, pos):
    val = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, pos
        shift += 7


def _walk(buf):
    pos = 0
    while pos < len(buf):
        key, pos = _varint(buf, pos)
        wt = key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 2:
            n, pos = _varint(buf, pos)
            v = buf[pos:pos + n]
            pos += n
        elif wt == 5:
            v = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError('wire type %d' % wt)
        yield key >> 3, wt, v


def example_features(buf):
    out = {}
    for _, _, features in _walk(buf):
        for _, _, entry in _walk(features):
            kv = dict((f, v) for f, _, v in _walk(entry))
            for kind, _, lst in _walk(kv[2]):
                vals = [v for _, _, v in _walk(lst)]
                if kind == 1:
                    out[kv[1].decode()] = vals
                elif kind == 2:
                    out[kv[1].decode()] = np.frombuffer(b''.join(vals), '<f4')
                else:
                    out[kv[1].decode()] = vals
    return out
