// x3_dma.h -- one 16-byte-per-lane LDS-DMA through a buffer descriptor, hidden from hipcc's wait-count insertion (inline asm): the copies
// of conv_p3.h, conv_x3h.h and conv_x3.h's weight tiles.  A lane that offers kP3Oob as its offset is out of range of the descriptor: the
// hardware writes ZEROS to its 16 bytes of LDS (TF 'SAME' padding / rows beyond the tensor with no select on data).  Completion is
// counted by hand: P3_WAIT_VM(n) = "at most n of this wave's copies (and ordinary loads) still in flight, every LDS access returned".
#pragma once
#include "dr_platform.h"

namespace dr {

typedef int dr_i32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kP3Oob = 0x80000000u;                    // per-lane offset of a lane that must read zeros (beyond num_records)
#if defined(DR_EMU)
struct P3Src { const unsigned char* base; };
static inline P3Src p3_src(const void* base, long bias_bytes, size_t /*bytes*/) { return P3Src{reinterpret_cast<const unsigned char*>(base) - bias_bytes}; }
static inline void p3_dma16(const P3Src& s, unsigned voff, unsigned soff, unsigned char* lds, unsigned lds_off) {
    unsigned char* d = lds + lds_off + (threadIdx.x & 63) * 16;
    if (voff & kP3Oob) memset(d, 0, 16); else memcpy(d, s.base + (size_t)voff + (size_t)soff, 16);
}
#define P3_WAIT_VM(n) ((void)0)
#else
struct P3Src { dr_i32x4 rsrc; };
// raw buffer descriptor (stride 0) over [base - bias, base + bytes): the scalar offset of a K-tile may shift a pixel back by up to
// one image row + one pixel (3x3 taps), so the base is biased down and the scalar offsets up -- both unsigned
__device__ __forceinline__ P3Src p3_src(const void* base, long bias_bytes, size_t bytes) {
    const unsigned long long a = (unsigned long long)reinterpret_cast<const unsigned char*>(base) - (unsigned long long)bias_bytes;
    P3Src s;
    s.rsrc[0] = (int)(unsigned)a;
    s.rsrc[1] = (int)((unsigned)(a >> 32) & 0xFFFFu);
    s.rsrc[2] = (int)(unsigned)(bytes + 2 * (size_t)bias_bytes);
    s.rsrc[3] = 0x00020000;
    return s;
}
// lds + lds_off must be wave-uniform (it travels in M0): lane L writes 16 bytes at lds + lds_off + 16 L
__device__ __forceinline__ void p3_dma16(const P3Src& s, unsigned voff, unsigned soff, unsigned char* lds, unsigned lds_off) {
    const unsigned dst = (unsigned)(unsigned long long)lds + lds_off;       // low half of the flat address of a __shared__ object = its LDS offset
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(voff), "s"(s.rsrc), "s"(soff) : "memory");
}
// (lgkmcnt(0): every fragment read of the tile has RETURNED before the barrier behind which its stage is overwritten)
#define P3_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ") lgkmcnt(0)" ::: "memory")
#endif

}  // namespace dr
