// dataio.h -- the data formats in front of the crop front-end: the depth frames of the three
// datasets arrive as PNG streams inside TFRecord files.
//   NYU   8-bit RGB, depth = (G << 8) | B                      data/nyu.py:148-156 (_decode_png)
//   ICVL / MSRA   16-bit grey (big-endian samples in a PNG)     data/icvl.py parse_example, data/msra.py:183-196
// The inflate step is zlib on the host (the Python host calls it); what remains of "decode_png" is (1) undoing the
// per-row prediction filters -- a serial byte recurrence, host code below -- and (2) turning samples into the fp32
// depth frame the crop kernel reads, which is pure HBM-bound byte work and runs on the device so the upload is the
// 1.5-3 bytes per pixel of the samples instead of 4.
#pragma once
#include <stdint.h>

#include "dr_platform.h"

namespace dr {

// PNG specification 9.2: filter types 0 None, 1 Sub, 2 Up, 3 Average, 4 Paeth; bpp = bytes per complete pixel
// (at least 1).  `filtered` holds height rows of (1 filter byte + row_bytes); `out` receives height * row_bytes.
static inline int png_unfilter_host(const uint8_t* filtered, int height, int row_bytes, int bpp, uint8_t* out) {
    for (int y = 0; y < height; ++y) {
        const uint8_t* src = filtered + (size_t)y * (row_bytes + 1);
        const int ft = src[0];
        ++src;
        uint8_t* cur = out + (size_t)y * row_bytes;
        const uint8_t* up = y ? cur - row_bytes : nullptr;
        switch (ft) {
            case 0:
                for (int i = 0; i < row_bytes; ++i) cur[i] = src[i];
                break;
            case 1:
                for (int i = 0; i < row_bytes; ++i) cur[i] = (uint8_t)(src[i] + (i >= bpp ? cur[i - bpp] : 0));
                break;
            case 2:
                for (int i = 0; i < row_bytes; ++i) cur[i] = (uint8_t)(src[i] + (up ? up[i] : 0));
                break;
            case 3:
                for (int i = 0; i < row_bytes; ++i) {
                    const int a = i >= bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0;
                    cur[i] = (uint8_t)(src[i] + ((a + b) >> 1));
                }
                break;
            case 4:
                for (int i = 0; i < row_bytes; ++i) {
                    const int a = i >= bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
                    const int p = a + b - c;
                    const int pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
                    const int pr = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                    cur[i] = (uint8_t)(src[i] + pr);
                }
                break;
            default:
                return -1;
        }
    }
    return 0;
}

// samples -> fp32 depth in mm.  mode 0: 8-bit RGB triples, depth = (G << 8) | B (R is ignored, as the reference does);
// mode 1: big-endian 16-bit grey.  Four pixels per thread: 12 (or 8) bytes in, one 16-byte store out.
__global__ __launch_bounds__(256) void depth_unpack_kernel(const uint8_t* __restrict__ s, long npix, int mode, float* __restrict__ out) {
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;           // group of 4 pixels
    const long p0 = q * 4;
    if (p0 >= npix) return;
    float v[4];
    if (p0 + 4 <= npix && mode == 0) {
        const uint32_t* w = reinterpret_cast<const uint32_t*>(s + p0 * 3);   // 12 bytes, 4-byte aligned (p0 % 4 == 0)
        const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];                       // R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3
        v[0] = (float)((((w0 >> 8) & 0xFF) << 8) | ((w0 >> 16) & 0xFF));
        v[1] = (float)(((w1 & 0xFF) << 8) | ((w1 >> 8) & 0xFF));
        v[2] = (float)((((w1 >> 24) & 0xFF) << 8) | (w2 & 0xFF));
        v[3] = (float)((((w2 >> 16) & 0xFF) << 8) | ((w2 >> 24) & 0xFF));
    } else if (p0 + 4 <= npix) {
        const uint32_t* w = reinterpret_cast<const uint32_t*>(s + p0 * 2);   // 8 bytes
        const uint32_t w0 = w[0], w1 = w[1];                                  // H0 L0 H1 L1 | H2 L2 H3 L3
        v[0] = (float)(((w0 & 0xFF) << 8) | ((w0 >> 8) & 0xFF));
        v[1] = (float)((((w0 >> 16) & 0xFF) << 8) | (w0 >> 24));
        v[2] = (float)(((w1 & 0xFF) << 8) | ((w1 >> 8) & 0xFF));
        v[3] = (float)((((w1 >> 16) & 0xFF) << 8) | (w1 >> 24));
    } else {                                                                  // ragged tail of the buffer
        for (int i = 0; i < 4; ++i) {
            const long p = p0 + i;
            if (p >= npix) break;
            out[p] = mode == 0 ? (float)((s[p * 3 + 1] << 8) | s[p * 3 + 2]) : (float)((s[p * 2] << 8) | s[p * 2 + 1]);
        }
        return;
    }
    *reinterpret_cast<float4*>(out + p0) = make_float4(v[0], v[1], v[2], v[3]);
}

}  // namespace dr
