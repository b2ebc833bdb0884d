// densereg.cpp -- graph builder, parameter registry, executors and the C ABI (include/densereg.h).
// Compiled by hipcc for gfx950 into libdensereg_hip.so (and, for CPU-side unit tests only, by a
// host clang++ with -DDR_EMU against tests/hipemu).
#include <algorithm>
#include <cmath>
#include <map>
#include <mutex>

#include "conv_igemm.h"
#include "conv_splitk.h"
#include "conv_x3.h"
#include "conv_p3.h"
#include "conv_x3h.h"
#include "conv_wgrad.h"
#include "conv_wgrad_bf16.h"
#include "conv_wgrad_tr.h"
#include "conv_wgrad_x3.h"
#include "frontend.h"
#include "dataio.h"
#include "kernels_misc.h"
#include "net.h"
#include "train_kernels.h"
#include "vote.h"

using namespace dr;

static std::string g_create_err;

#define DR_FAIL(h, code, ...)                                  \
    do {                                                       \
        char _b[512];                                          \
        snprintf(_b, sizeof(_b), __VA_ARGS__);                 \
        (h)->err = _b;                                         \
        return (code);                                         \
    } while (0)

#define DR_CHECK_LAUNCH(h)                                                     \
    do {                                                                       \
        std::string _m;                                                        \
        if (rt::last_error(&_m)) DR_FAIL(h, DR_E_DEVICE, "HIP error: %s (%s:%d)", _m.c_str(), __FILE__, __LINE__); \
    } while (0)

// every entry point re-selects the handle's device: the HIP current device is per host thread and the caller
// (PyTorch) may have moved it
#define DR_ENTER(h) (void)rt::set_device((h)->cfg.device)

static inline int grid_for(long total, int block = 256, int cap = 0) {
    static const int dflt = [] { const char* e = getenv("DR_ELT_GRID"); const int v = e ? atoi(e) : 0; return v >= 64 ? v : 256 * 8; }();
    if (cap <= 0) cap = dflt;                                  // grid-stride elementwise kernels (experiment switch DR_ELT_GRID)
    long g = (total + block - 1) / block;
    return (int)std::max<long>(1, std::min<long>(g, cap));
}

// ==============================================================================================
// conv launcher
// ==============================================================================================
namespace dr {

template <int BM, int BN, int WM, int WN, int BK = 16, int GL = 0, int BF = 0, int WK = 1, int MF = 32>
static void launch_cfg(const ConvParams& p_in, hipStream_t s) {
    // N blocks of a row block back to back in dispatch order (conv_igemm.h nfast); DR_CONV_NFAST=0: plain 2-D grid order
    static const int nfast = [] { const char* e = getenv("DR_CONV_NFAST"); return (e && e[0] == '0') ? 0 : 1; }();
    ConvParams p = p_in;
    p.nfast = nfast;
    const int M = p.B * p.H * p.W;
    dim3 grid(dr_ceil_div(M, BM), dr_ceil_div(p.Ng > 0 ? p.Ng : p.Np, BN));
    if constexpr (MF == 16) {                                  // one column block spans every output channel (conv_tile_id): the 32-row
        grid.y = 1;                                            // padding of the packed weights beyond it is never computed
        p.gx = (int)grid.x; p.gy = 1;
        DR_LAUNCH((conv_igemm_kernel<BM, BN, WM, WN, 0, BK, GL, BF, WK, 0, 16>), grid, dim3(256), 0, s, p);
    } else {
        p.gx = (int)grid.x; p.gy = (int)grid.y;
        if constexpr (BF == 1) {
            if (p.y_bf16 || p.bst_raw_bf16) {                  // the instantiations with the bf16-storage copies of the epilogue (XB bit 1)
                if (p.x_bf16) DR_LAUNCH((conv_igemm_kernel<BM, BN, WM, WN, 0, BK, GL, BF, WK, 3>), grid, dim3(256), 0, s, p);
                else DR_LAUNCH((conv_igemm_kernel<BM, BN, WM, WN, 0, BK, GL, BF, WK, 2>), grid, dim3(256), 0, s, p);
                return;
            }
            if (p.x_bf16) { DR_LAUNCH((conv_igemm_kernel<BM, BN, WM, WN, 0, BK, GL, BF, WK, 1>), grid, dim3(256), 0, s, p); return; }
        }
        DR_LAUNCH((conv_igemm_kernel<BM, BN, WM, WN, 0, BK, GL, BF, WK, 0>), grid, dim3(256), 0, s, p);
    }
}

// LDS-DMA refill variant (conv_igemm.h GL) for inputs made of whole 16-byte channel chunks; DR_CONV_GLDS=0 selects the
// register-staged refill everywhere.  Default since round 2: with the two pipeline stages as separate LDS objects the copy
// really runs under the MFMAs (3x3 256->256 473 -> 457 us, 3x3 128->128 139 -> 129, 3x3 64->64 45 -> 42; training step
// +0.5 %, inference +1.1 %; profiles/r02_conv_glds_ab.md).
static bool conv_use_glds(const ConvParams& p) {
    static const bool on = [] { const char* e = getenv("DR_CONV_GLDS"); return !(e && e[0] == '0'); }();
    return on && p.Cin % 4 == 0;
}

static long g_x3h_launches = 0;      // conv_x3h.h launches of this process (dr_dbg_x3h_launches)
static long g_p3_launches = 0;       // conv_p3.h launches of this process (test hook dr_dbg_p3_launches: "the kernel under test is the one that ran")
static int g_force_tile = -1;        // test/bench hook (dr_dbg_conv_bench); -1 = heuristic
static int g_dbg_bf16 = 0;           // test/bench hook (dr_dbg_force_bf16): dr_dbg_conv2d / dr_dbg_conv_bench run the bf16 kernels
static int g_dbg_bf16_storage = 0;   // test hook (dr_dbg_force_bf16_storage): bf16-stored x / g / draw in the debug entries

static int conv_tile_heuristic(const ConvParams& p);
bool conv_use_x3(const ConvParams& p);
bool conv_use_p3(const ConvParams& p);
static int tile_rows_of(int t) {
    if (t == KID_CONV_SPLITK) return 32;
    if (t == KID_CONV_X3 || t == KID_CONV_P3) return 128;
    return (t == KID_CONV_64x128 || t == KID_CONV_64x64 || t == KID_CONV_64x64_K64 || t == KID_CONV_64x96 || t == KID_CONV_64x160 ||
            (t >= KID_CONV16_64x80 && t <= KID_CONV16_64x160)) ? 64 : 128;
}
// tile shape for a problem (shared by the launcher and the profiler labels)
int conv_tile_id(const ConvParams& p) {
    if (g_force_tile >= 0) return g_force_tile;
    if (conv_use_p3(p)) return KID_CONV_P3;
    if (conv_use_x3(p)) return KID_CONV_X3;
    const int t = conv_tile_heuristic(p);
    // micro-batch groups: a workgroup's rows must lie in one group (per-group statistics rows, per-group coefficients) -- where
    // the preferred tile straddles the boundaries (2x2 layers: 4 rows per crop) the 32-row split-K kernel takes the launch
    if (p.grp_rows > 0 && p.grp_rows % tile_rows_of(t) != 0 && p.grp_rows % 32 == 0 && !p.x_bf16) return KID_CONV_SPLITK;
    return t;
}
static int conv_tile_heuristic(const ConvParams& p) {
    const int M = p.B * p.H * p.W;
    const int ncols = p.Ng > 0 ? p.Ng : p.Np;                 // output columns this launch computes
    // Measured on MI355X (profiles/r01_conv_microbench.md): what decides between the tiles of one N width is how
    // evenly the workgroups fall on the 256 CUs.  M = 40960 rows: Np = 256 / 512 give 1280 / 2560 64x128 workgroups
    // = 5 / 10 per CU (3x3 256->256: 99 TFLOP/s, 128-row tiles 95); Np = 128 gives 640 = 2.5 per CU, and the 64x64
    // tile (1280 workgroups) wins by 5-12 % despite re-reading A twice (3x3 128->128: 141 vs 150 us).
    auto balance = [](long blocks) { return (double)blocks / (double)(dr_ceil_div((int)blocks, 256) * 256); };
    const long rows64 = dr_ceil_div(M, 64), rows128 = dr_ceil_div(M, 128);
    // Grids that leave CUs idle (everything below 32x32 at B=40) are chains of ~0.6 us K-tiles with nothing to
    // overlap: the fat K-tile variant moves 64 channels per round trip (3x3 64->64 at 8x8: 22.8 -> see
    // profiles/r01_conv_microbench.md).  Needs >= 2 fat tiles to pay for its larger prologue.
    // Below ~128 such workgroups (8x8 pixels and smaller at B = 40) even that leaves most of the chip idle while each wave
    // issues the whole K axis: the split-K kernel (conv_splitk.h) quarters the chain and quadruples the workgroups
    // (profiles/r01_conv_small_layers.md: 3x3 64->64 at 8x8 18.7 -> 10.9 us, 1x1 128->64 7.5 -> 5.5 us; at 16x16 it loses).
    // (an input STORED as bf16 stays on the tiled kernels: the split-K kernel stages fp32 only -- a mispredicted storage decision
    // upstream then costs speed, not a failed launch)
    if (!p.x_bf16 && rows64 * dr_ceil_div(ncols, 64) <= 128 && (long)p.ksize * p.ksize * dr_ceil_div(p.Kp, p.bf16 ? 32 : 16) >= 4) return KID_CONV_SPLITK;
    if (!p.bf16 && ncols % 64 == 0 && rows64 * (ncols / 64) <= 256 && (long)p.ksize * p.ksize * p.Kp >= 128) return KID_CONV_64x64_K64;
    if (ncols % 128 == 0) {
        const long b128 = rows128 * (ncols / 128), b64x128 = rows64 * (ncols / 128), b64x64 = rows64 * (ncols / 64);
        if (b128 >= 4096) return KID_CONV_128x128;
        if (b64x128 < 256 || 0.92 * balance(b64x64) > balance(b64x128)) return KID_CONV_64x64;
        return KID_CONV_64x128;
    }
    if (ncols % 64 == 0) return (rows128 * (ncols / 64) >= 4096) ? KID_CONV_128x64 : KID_CONV_64x64;
    // fp32, 65..80 and 129..160 output channels (the hm3 / um-head residuals and their input gradients: 65, 78, 131, 156 with
    // J = 14): 16-column MFMA tiles, one 64-row workgroup spans all columns (conv_igemm.h, MF = 16) -- 80 / 144 computed columns
    // where 32-column tiles pad to 96 / 160.  Measured at B = 40 against the 128x32 tile (profiles/r03_experiments.md): 3x3
    // 78->78 77.9 -> 67.5 us, 3x3 65->65 77.7 -> 67.3, 1x1 156->78 24.0 -> 21.4, 1x1 256->78 33.1 -> 29.6, 1x1 128->131 32.0 ->
    // 29.3, 1x1 256->156 52.9 -> 50.6; where 16 columns save nothing (MSRA's 85 -> 96, 105 -> 112, 170 -> 176) the tiles measured
    // equal and are not built.  DR_CONV_MF16=0 restores the 128x32 tile.
    if (!p.bf16 && p.Cout > 64) {
        static const bool mf16 = [] { const char* e = getenv("DR_CONV_MF16"); return !(e && e[0] == '0'); }();
        const int c16 = dr_round_up(p.Cout, 16);
        if (mf16 && (c16 == 80 || c16 == 144 || c16 == 160) && ncols >= c16)
            return c16 == 80 ? KID_CONV16_64x80 : c16 == 144 ? KID_CONV16_64x144 : KID_CONV16_64x160;
    }
    // narrow outputs (N = 65..96, 129..160: the hm3 / um-head residuals and their input gradients) on grids that fill the
    // chip: one 64-row workgroup spans all columns, its four waves split rows and K (conv_igemm.h, WK); DR_CONV_NARROW=0 off
    static const bool narrow = [] { const char* e = getenv("DR_CONV_NARROW"); return !(e && e[0] == '0'); }();
    // Measured (profiles/r02_conv_narrow_tiles.md): the MFMA count equals the 128x32 tile's, what the tile saves is the three-
    // to five-fold re-staging of A -- so it wins where traffic binds: bf16 (3x3 from M = 40960 rows: 37 -> 30 us; at
    // M = 163840 2x: 427 -> 204 us) and fp32 on very deep grids (M = 163840: +6..16 %); in fp32 at M = 40960 it loses 3-20 %.
    const bool pays = p.bf16 ? (rows64 >= 512 && (p.ksize == 3 || rows64 >= 2048)) : rows64 >= 2048;
    if (narrow && pays && (ncols == 96 || ncols == 160)) return ncols == 96 ? KID_CONV_64x96 : KID_CONV_64x160;
    return KID_CONV_128x32;
}

// conv_x3.h (fp32-accurate products on the bf16 matrix cores, 128x128 tile): where it is used.  DR_CONV_X3: 0 = never,
// 1 = the measured rule (default), 2 = wherever the kernel can run (tests)
static int g_dbg_x3 = -1;            // test/bench hook (dr_dbg_force_x3): overrides DR_CONV_X3
// conv_x3h.h: 3x3 layers whose 128-row tiles are whole rows of a 16- or 32-pixel-wide image keep the haloed input tile in LDS across
// the nine taps (DR_X3_HALO=0 / dr_dbg_force_x3(7): conv_x3_kernel everywhere)
static bool conv_x3h_shape(const ConvParams& p) {
    static const bool halo_on = [] { const char* e = getenv("DR_X3_HALO"); return !(e && e[0] == '0'); }();
    return halo_on && g_dbg_x3 != 7 && p.ksize == 3 && !p.rowmask && (p.W == 32 || p.W == 16) && (p.H * p.W) % 128 == 0;
}
// conv_x3_kernel's weight tiles by a hidden LDS-DMA (conv_x3.h, BD); DR_X3_BD=0: through registers
static bool x3_bd() {
    static const bool on = [] { const char* e = getenv("DR_X3_BD"); return !(e && e[0] == '0'); }();
    return on;
}
// ... and the pixels of two K-tiles in flight (conv_x3.h, PF = 1) or a scheduling barrier before the split (PF = 2).  DR_X3_PF = mode
// (0 off), DR_X3_PF_MAXT = n: only layers of at most n K-tiles (1x1: Cin <= 16 n)
[[maybe_unused]] static int x3_abl() {                                  // DR_X3_ABL: conv_x3.h, ABL_ (debug library)
    static const int v = [] { const char* e = getenv("DR_X3_ABL"); return e ? atoi(e) : 0; }();
    return v;
}
[[maybe_unused]] static int x3_pf(const ConvParams& p) {
    static const int mode = [] { const char* e = getenv("DR_X3_PF"); return e ? atoi(e) : 0; }();
    static const int lim = [] { const char* e = getenv("DR_X3_PF_MAXT"); return e ? atoi(e) : 1 << 30; }();
    const int T = p.ksize * p.ksize * (p.Kp / 16);
    return T <= lim && T >= 2 ? mode : 0;                     // (the two-tile prefetch wants two K-tiles: a one-tile layer measured wrong results)
}
bool conv_use_x3(const ConvParams& p) {
    static const int env = [] { const char* e = getenv("DR_CONV_X3"); return e ? atoi(e) : 1; }();
    const int mode = g_dbg_x3 >= 0 ? g_dbg_x3 : env;
    if (mode <= 0 || !p.w3 || p.bf16 || p.x_bf16 || p.y_bf16 || p.bst_raw_bf16 || g_force_tile >= 0) return false;
    if (!p.x && !p.xp3) return false;
    if (p.grp_rows > 0 && p.grp_rows % 128 != 0) return false;
    if (mode >= 2) return true;
    const long M = (long)p.B * p.H * p.W;
    const int ncols = p.Ng > 0 ? p.Ng : p.Np;
    // whole 128-column blocks (a 160-column layer would compute 256), and a grid of at least two workgroups per CU
    static const long min_wgs = [] { const char* e = getenv("DR_X3_MIN_WGS"); const long v = e ? atol(e) : 0; return v > 0 ? v : 512l; }();
    // 64-column blocks (the hourglass bottlenecks' 64 channels) on deep grids: 3x3 64->64 at 204 800 rows 165 -> 126 us, 1x1 128->64
    // 48.5 -> 43.6 (profiles/r05_x3_microbench.md); DR_X3_BN64=0 off
    static const bool bn64 = [] { const char* e = getenv("DR_X3_BN64"); return !(e && e[0] == '0'); }();
    // one 96-column block for the 65..96-channel layers (the hm3 / um-tower residuals and their input gradients) on deep grids;
    // DR_X3_BN96=0 leaves them on the fp32 16-column tiles
    static const bool bn96 = [] { const char* e = getenv("DR_X3_BN96"); return !(e && e[0] == '0'); }();
    // the halo kernel wins from one workgroup per CU on (profiles/r06_x3h_rule.md: at 320 row blocks 1.38-1.85x the fp32 tiles, at 400 row
    // blocks of 16x16 pixels 1.7-1.9x, at 80 equal): inference at B = 40 and the 16x16 level of a 200-crop window included
    // one 160-column block for the 129..160-channel layers (the input gradients of the 131 / 156-channel 1x1 layers; config 5's 129 / 142-
    // channel 3x3 layers) on deep grids, in place of the fp32 16-column tiles; DR_X3_BN160=0 off
    static const bool bn160 = [] { const char* e = getenv("DR_X3_BN160"); return !(e && e[0] == '0'); }();
    if (ncols == 160) return bn160 && dr_ceil_div((int)M, 128) >= 2 * min_wgs && (long)p.ksize * p.ksize * p.Kp >= 128;
    if (conv_x3h_shape(p) && p.Kp >= 64) {
        const long rb = dr_ceil_div((int)M, 128);
        if (ncols % 128 == 0) return rb * (ncols / 128) >= 256;
        if (ncols == 96) return bn96 && rb >= 256;
        if (ncols % 64 == 0) return bn64 && rb * (ncols / 64) >= 256;
        return false;
    }
    // (the halo rule first: round 6 shipped it BEHIND the 128-column rule below, which turned away the 3x3 128 -> 128 layers of a 40-crop
    // batch -- 320 workgroups -- that the halo rule was measured for: 116 us on the fp32 tile)
    if (ncols % 128 == 0) return dr_ceil_div((int)M, 128) * (long)(ncols / 128) >= min_wgs && (long)p.ksize * p.ksize * p.Kp >= 128;
    if (ncols == 96) return bn96 && dr_ceil_div((int)M, 128) >= 2 * min_wgs && (long)p.ksize * p.ksize * p.Kp >= 128;
    return bn64 && ncols % 64 == 0 && dr_ceil_div((int)M, 128) * (long)(ncols / 64) >= 2 * min_wgs && (long)p.ksize * p.ksize * p.Kp >= 128;
}

// conv_p3.h (the x3 products on an input stored as its three bf16 planes): what the kernel can run -- whole 128-column blocks, at
// least two K-tiles, the whole P3 tensor as input.  conv_p3_applies: the SHAPE rule alone, for the producer that must decide how
// to store the tensor before the consumer's launch exists (train_exec.inc); conv_use_p3: + "the input is there".
bool conv_p3_applies(const ConvParams& p) {
    if (!conv_use_x3(p)) return false;
    const int ncols = p.Ng > 0 ? p.Ng : p.Np;
    static const bool on = [] { const char* e = getenv("DR_CONV_P3"); return !(e && e[0] == '0'); }();
    return on && ncols % 128 == 0 && p.ksize * p.ksize * (p.Kp / 16) >= 2 && p.x_coff == 0;
}
bool conv_use_p3(const ConvParams& p) { return p.xp3 && p.xp3_cp == p.Kp && conv_p3_applies(p); }

// output rows per workgroup of the tile a problem gets
int conv_tile_rows(const ConvParams& p) { return tile_rows_of(conv_tile_id(p)); }
// rows of ConvParams::stat_part the launch writes = workgroups along M of the chosen tile
int conv_stat_rows(const ConvParams& p) { return dr_ceil_div(p.B * p.H * p.W, conv_tile_rows(p)); }

int launch_conv_igemm(const ConvParams& p, hipStream_t s) {
    if (p.x_cs % 4 || p.x_coff % 4 || p.Kp % 16 || p.Np % 32 || !p.zeros || p.Ng % 32 || p.Ng > p.Np) return -1;
    // the kernel addresses every tensor with 32-bit element offsets from its base pointer
    const long long M = (long long)p.B * p.H * p.W;
    const long long widest = std::max(std::max((long long)p.x_cs, (long long)p.y_cs), std::max((long long)p.res_cs, (long long)p.Cout));
    if (M * widest >= (1ll << 32)) return -1;
    // bf16-stored raw output / producer raw (conv_epilogue.inc, EP_IO16): the bf16 kernels only, and a plain output only
    if ((p.y_bf16 || p.bst_raw_bf16) && !p.bf16) return -1;
    if (p.y_bf16 && !p.bst_raw_bf16 && (p.scale || p.shift || p.relu || p.res || p.drop || p.drop_rng || p.bst_raw || p.out_rowmask)) return -1;
    if (p.y_bf16 && p.bst_raw_bf16 && p.res) return -1;                  // a bf16-stored dOut has ONE writer
    if (p.bst_raw_bf16 && (!p.bst_raw || p.bst_act || p.scale || p.shift || p.relu || p.drop || p.drop_rng)) return -1;
    if (conv_tile_id(p) == KID_CONV_P3) {
        static const int nfast = [] { const char* e = getenv("DR_CONV_NFAST"); return (e && e[0] == '0') ? 0 : 1; }();
        if (M * (long long)p.xp3_cp * 6 + 2ll * (p.W + 1) * p.xp3_cp * 6 >= (1ll << 31)) return -1;   // 31-bit byte offsets into the P3 tensor
        ConvParams q = p;
        q.nfast = nfast;
        const int ncols = p.Ng > 0 ? p.Ng : p.Np;
        dim3 grid(dr_ceil_div((int)M, 128), ncols / 128);
        q.gx = (int)grid.x; q.gy = (int)grid.y;
#if defined(DR_DEBUG_HOOKS)
        static const int var = [] { const char* e = getenv("DR_P3_VARIANT"); return e ? atoi(e) : 0; }();
        switch (var) {
            case 1: DR_LAUNCH((conv_p3_kernel<128, 1>), grid, dim3(512), 0, s, q); break;
            case 2: DR_LAUNCH((conv_p3_kernel<128, 2>), grid, dim3(512), 0, s, q); break;
            case 4: DR_LAUNCH((conv_p3_kernel<128, 4>), grid, dim3(512), 0, s, q); break;
            case 6: DR_LAUNCH((conv_p3_kernel<128, 6>), grid, dim3(512), 0, s, q); break;
            default: DR_LAUNCH((conv_p3_kernel<128>), grid, dim3(512), 0, s, q); break;
        }
#else
        DR_LAUNCH((conv_p3_kernel<128>), grid, dim3(512), 0, s, q);
#endif
        ++g_p3_launches;
        return 0;
    }
    if (conv_tile_id(p) == KID_CONV_X3) {
        static const int nfast = [] { const char* e = getenv("DR_CONV_NFAST"); return (e && e[0] == '0') ? 0 : 1; }();
        static const int variant = [] { const char* e = getenv("DR_X3_VARIANT"); return e ? atoi(e) : 0; }();
        ConvParams q = p;
        q.nfast = nfast;
        const int ncols = p.Ng > 0 ? p.Ng : p.Np;
        const bool bn96 = ncols == 96;                                       // the 65..96-channel layers: one 96-column block
        const bool bn160 = ncols == 160;                                     // the 129..160-channel layers: one 160-column block
        const bool bn64 = !bn96 && !bn160 && ncols % 128 != 0 && ncols % 64 == 0;      // 64-column blocks where 128 would compute padding
        {   // DR_X3_STAGGER = k: second workgroups of the first round start (K-tiles * k / 16) x s_sleep(127) late (conv_x3.h)
            static const int stag = [] { const char* e = getenv("DR_X3_STAGGER"); return e ? atoi(e) : 0; }();
            const int T_total = p.ksize * p.ksize * (p.Kp / 16);
            static const int stag_mode = [] { const char* e = getenv("DR_X3_STAGGER_MODE"); return e ? atoi(e) : 0; }();
            q.stagger = stag > 0 ? (std::max(1, (T_total * stag + 8) / 16) | (stag_mode << 16)) : 0;
        }
        // larger blocks for the wide layers the halo kernel does not take (DR_X3_BIG: 0 off; 256-column blocks: 1 eight waves of 64x64,
        // 2 sixteen of 64x32; 256-ROW blocks (the weight tile fetched once per 256 pixels): 3 eight waves of 64x64, 4 sixteen of 64x32)
#if defined(DR_DEBUG_HOOKS)
        static const int big_mode = [] { const char* e = getenv("DR_X3_BIG"); return e ? atoi(e) : 0; }();
#else
        constexpr int big_mode = 0;                                          // (measured equal or slower, profiles/r06_experiments.md: test / bench library only)
#endif
        const bool big_ok = big_mode > 0 && variant == 0 && (g_dbg_x3 < 3 || g_dbg_x3 == 7) && !conv_x3h_shape(p);
        const bool bn256 = big_ok && big_mode <= 2 && ncols % 256 == 0;
        const bool bm256 = big_ok && big_mode >= 3 && ncols % 128 == 0 && (p.grp_rows <= 0 || p.grp_rows % 256 == 0);
        dim3 grid(dr_ceil_div((int)M, bm256 ? 256 : 128), (bn96 || bn160) ? 1 : dr_ceil_div(ncols, bn64 ? 64 : bn256 ? 256 : 128));
        q.gx = (int)grid.x; q.gy = (int)grid.y;
        // variants (DR_X3_VARIANT bit 0: one accumulator, bit 1: three-stage LDS ring, bit 2: four waves of 64x64 instead of eight of 64x32;
        // dr_dbg_force_x3 3 / 4 / 5 select the same)
        const bool one_acc = (variant & 1) || g_dbg_x3 == 3;
#if defined(DR_DEBUG_HOOKS)
        const bool ring = ((variant & 2) || g_dbg_x3 == 4) && !one_acc;          // (measured slower than two stages: profiles/r05_experiments.md;
#else                                                                            //  instantiated in the test / bench library only)
        const bool ring = false;
#endif
        const bool w4 = (variant & 4) || g_dbg_x3 == 5 || one_acc || ring;
        if (bn160) {                                                         // (no halo form, no variants)
            DR_LAUNCH((conv_x3_kernel<128, 160, 1, 0, 4, 4, 1>), grid, dim3(256), 0, s, q);
            return 0;
        }
        if (conv_x3h_shape(p) && !one_acc && !ring && !w4) {
            if (bn96) {
                if (p.W == 32) DR_LAUNCH((conv_x3h_kernel<96, 5, 4, 4>), grid, dim3(256), 0, s, q);
                else DR_LAUNCH((conv_x3h_kernel<96, 4, 4, 4>), grid, dim3(256), 0, s, q);
            } else if (bn64) {
                if (p.W == 32) DR_LAUNCH((conv_x3h_kernel<64, 5, 4, 2>), grid, dim3(256), 0, s, q);
                else DR_LAUNCH((conv_x3h_kernel<64, 4, 4, 2>), grid, dim3(256), 0, s, q);
            } else {
                if (p.W == 32) DR_LAUNCH((conv_x3h_kernel<128, 5>), grid, dim3(512), 0, s, q);
                else DR_LAUNCH((conv_x3h_kernel<128, 4>), grid, dim3(512), 0, s, q);
            }
            ++g_x3h_launches;
            return 0;
        }
#if defined(DR_DEBUG_HOOKS)
        if (bn256) {
            if (big_mode == 2) DR_LAUNCH((conv_x3_kernel<128, 256, 1, 0, 16, 2, 1>), grid, dim3(1024), 0, s, q);
            else DR_LAUNCH((conv_x3_kernel<128, 256, 1, 0, 8, 2, 1>), grid, dim3(512), 0, s, q);
        } else if (bm256) {
            if (big_mode == 4) DR_LAUNCH((conv_x3_kernel<256, 128, 1, 0, 16, 4, 1>), grid, dim3(1024), 0, s, q);
            else DR_LAUNCH((conv_x3_kernel<256, 128, 1, 0, 8, 4, 1>), grid, dim3(512), 0, s, q);
        } else
#endif
        if (bn96) {
            DR_LAUNCH((conv_x3_kernel<128, 96, 1, 0, 4, 4>), grid, dim3(256), 0, s, q);
        } else if (bn64) {
            if (one_acc) DR_LAUNCH((conv_x3_kernel<128, 64, 0>), grid, dim3(256), 0, s, q);
#if defined(DR_DEBUG_HOOKS)
            else if (ring) DR_LAUNCH((conv_x3_kernel<128, 64, 1, 1>), grid, dim3(256), 0, s, q);
#endif
            else DR_LAUNCH((conv_x3_kernel<128, 64, 1>), grid, dim3(256), 0, s, q);
        } else {
            if (one_acc) DR_LAUNCH((conv_x3_kernel<128, 128, 0>), grid, dim3(256), 0, s, q);
#if defined(DR_DEBUG_HOOKS)
            else if (ring) DR_LAUNCH((conv_x3_kernel<128, 128, 1, 1>), grid, dim3(256), 0, s, q);
#endif
            else if (w4) DR_LAUNCH((conv_x3_kernel<128, 128, 1>), grid, dim3(256), 0, s, q);
#if defined(DR_DEBUG_HOOKS)
            else if (x3_bd() && x3_abl() == 3) DR_LAUNCH((conv_x3_kernel<128, 128, 1, 0, 8, 2, 1, 0, 3>), grid, dim3(512), 0, s, q);
            else if (x3_bd() && x3_abl() == 16) DR_LAUNCH((conv_x3_kernel<128, 128, 1, 0, 8, 2, 1, 0, 16>), grid, dim3(512), 0, s, q);
            else if (x3_bd() && x3_abl() == 32) DR_LAUNCH((conv_x3_kernel<128, 128, 1, 0, 8, 2, 1, 0, 32>), grid, dim3(512), 0, s, q);
            else if (x3_bd() && x3_abl() == 48) DR_LAUNCH((conv_x3_kernel<128, 128, 1, 0, 8, 2, 1, 0, 48>), grid, dim3(512), 0, s, q);
            else if (x3_bd() && x3_abl() == 51) DR_LAUNCH((conv_x3_kernel<128, 128, 1, 0, 8, 2, 1, 0, 51>), grid, dim3(512), 0, s, q);
            else if (x3_bd() && x3_abl() == 112) DR_LAUNCH((conv_x3_kernel<128, 128, 1, 0, 8, 2, 1, 0, 112>), grid, dim3(512), 0, s, q);
            else if (x3_bd() && x3_abl() == 115) DR_LAUNCH((conv_x3_kernel<128, 128, 1, 0, 8, 2, 1, 0, 115>), grid, dim3(512), 0, s, q);
            else if (x3_bd() && x3_pf(p) == 1) DR_LAUNCH((conv_x3_kernel<128, 128, 1, 0, 8, 2, 1, 1>), grid, dim3(512), 0, s, q);
            else if (x3_bd() && x3_pf(p) == 2) DR_LAUNCH((conv_x3_kernel<128, 128, 1, 0, 8, 2, 1, 2>), grid, dim3(512), 0, s, q);
#endif
            else if (x3_bd()) DR_LAUNCH((conv_x3_kernel<128, 128, 1, 0, 8, 2, 1>), grid, dim3(512), 0, s, q);
            else DR_LAUNCH((conv_x3_kernel<128, 128, 1, 0, 8>), grid, dim3(512), 0, s, q);
        }
        return 0;
    }
    if (conv_tile_id(p) == KID_CONV_SPLITK) {
        if (p.bf16 && p.Kp % 32) return -1;
        if (p.x_bf16) return -1;                                          // the split-K kernel stages fp32 only
        dim3 grid(dr_ceil_div((int)M, 32), dr_ceil_div(p.Ng > 0 ? p.Ng : p.Np, 32));
        ConvParams q = p;
        q.gx = (int)grid.x; q.gy = (int)grid.y;
        if (p.bf16 && (p.y_bf16 || p.bst_raw_bf16)) DR_LAUNCH((conv_splitk_kernel<1, 0, 1>), grid, dim3(256), 0, s, q);
        else if (p.bf16) DR_LAUNCH((conv_splitk_kernel<1>), grid, dim3(256), 0, s, q);
        else DR_LAUNCH((conv_splitk_kernel<0>), grid, dim3(256), 0, s, q);
        return 0;
    }
    if (p.bf16) {
        if (p.Kp % 32) return -1;
        switch (conv_tile_id(p)) {
            case KID_CONV_128x128: launch_cfg<128, 128, 2, 2, 16, 0, 1>(p, s); break;
            case KID_CONV_64x128: launch_cfg<64, 128, 2, 2, 16, 0, 1>(p, s); break;
            case KID_CONV_128x64: launch_cfg<128, 64, 2, 2, 16, 0, 1>(p, s); break;
            case KID_CONV_64x64: launch_cfg<64, 64, 2, 2, 16, 0, 1>(p, s); break;
            case KID_CONV_128x32: launch_cfg<128, 32, 4, 1, 16, 0, 1>(p, s); break;
            case KID_CONV_64x96: launch_cfg<64, 96, 2, 1, 16, 0, 1, 2>(p, s); break;
            case KID_CONV_64x160: launch_cfg<64, 160, 2, 1, 16, 0, 1, 2>(p, s); break;
            default: return -1;
        }
        return 0;
    }
    switch (conv_tile_id(p)) {
        case KID_CONV_128x128: launch_cfg<128, 128, 2, 2>(p, s); break;
        case KID_CONV_64x128:
            if (conv_use_glds(p)) launch_cfg<64, 128, 2, 2, 16, 1>(p, s); else launch_cfg<64, 128, 2, 2>(p, s);
            break;
        case KID_CONV_128x64: launch_cfg<128, 64, 2, 2>(p, s); break;
        case KID_CONV_64x64:
            if (conv_use_glds(p)) launch_cfg<64, 64, 2, 2, 16, 1>(p, s); else launch_cfg<64, 64, 2, 2>(p, s);
            break;
        case KID_CONV_64x64_K64: launch_cfg<64, 64, 2, 2, 64>(p, s); break;
        case KID_CONV_64x96: launch_cfg<64, 96, 2, 1, 16, 0, 0, 2>(p, s); break;
        case KID_CONV_64x160: launch_cfg<64, 160, 2, 1, 16, 0, 0, 2>(p, s); break;
        case KID_CONV16_64x80: if (p.Cout > 80) return -1; launch_cfg<64, 80, 4, 1, 16, 0, 0, 1, 16>(p, s); break;
        case KID_CONV16_64x144: if (p.Cout > 144) return -1; launch_cfg<64, 144, 4, 1, 16, 0, 0, 1, 16>(p, s); break;
        case KID_CONV16_64x160: if (p.Cout > 160) return -1; launch_cfg<64, 160, 4, 1, 16, 0, 0, 1, 16>(p, s); break;
        default: launch_cfg<128, 32, 4, 1>(p, s); break;
    }
    return 0;
}

// stream of an executor lane: the caller's stream for lane 0, or when lanes are disabled / profiling
static inline hipStream_t lane_of(const dr_handle* h, int lane, hipStream_t main) {
    return (lane == 0 || !h->multi_stream || h->profiling) ? main : h->lane_stream[lane];
}
// event edge `from` -> `to`: everything enqueued on `from` so far happens before what follows on `to`
static inline void lane_edge(dr_handle* h, const Op& op, bool parent_to_child, hipStream_t main) {
    hipStream_t a = lane_of(h, op.lane, main), b = lane_of(h, op.lane2, main);
    if (a == b) return;
    if (!parent_to_child) std::swap(a, b);
    rt::event_record(h->lane_ev[op.ev], a);
    rt::stream_wait_event(b, h->lane_ev[op.ev]);
}

// RAII profiling scope: two events around the launches of one op (only when profiling is on)
struct ProfScope {
    dr_handle* h; hipStream_t s; ProfRecord r;
    ProfScope(dr_handle* h_, hipStream_t s_, int kid, double flops, double bytes) : h(h_), s(s_) {
        if (!h->profiling) return;
        r.kid = kid; r.tag = h->prof_tag; r.flops = flops; r.bytes = bytes;
        r.a = rt::event_create(); r.b = rt::event_create();
        rt::event_record(r.a, s);
    }
    ~ProfScope() {
        if (!h->profiling) return;
        rt::event_record(r.b, s);
        h->prof.push_back(r);
    }
};

}  // namespace dr

// ==============================================================================================
// small device helpers for parameters
// ==============================================================================================
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* w, float* wp, int taps, int Cin, int Cout,
                                                           int Kp, int Np) {
    const long total = (long)taps * Kp * Np;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        // packed layout [Kp/16][tap][Np][16] (conv_igemm.h): K-tile major, 16 k contiguous per output channel
        const int kk = int(i % 16);
        const int n = int((i / 16) % Np);
        const int t = int((i / (16l * Np)) % taps);
        const int k = int(i / (16l * Np * taps)) * 16 + kk;
        wp[i] = (k < Cin && n < Cout) ? w[((long)t * Cin + k) * Cout + n] : 0.f;
    }
}

// bf16 forward packing [Kp/32][tap][Np][32] (conv_igemm.h, BF kernels), round to nearest even
__global__ __launch_bounds__(256) void pack_weights_bf16_kernel(const float* w, __bf16* wp, int taps, int Cin, int Cout, int Kp, int Np) {
    const long total = (long)taps * Kp * Np;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int kk = int(i % 32);
        const int n = int((i / 32) % Np);
        const int t = int((i / (32l * Np)) % taps);
        const int k = int(i / (32l * Np * taps)) * 32 + kk;
        wp[i] = (__bf16)((k < Cin && n < Cout) ? w[((long)t * Cin + k) * Cout + n] : 0.f);
    }
}

// dgrad weights: wpT[t'][k=cout][n=cin] = w[taps-1-t'][cin][cout]
__global__ __launch_bounds__(256) void pack_weights_T_kernel(const float* w, float* wpT, int taps, int Cin, int Cout,
                                                             int KpT, int NpT) {
    const long total = (long)taps * KpT * NpT;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int kk = int(i % 16);
        const int n = int((i / 16) % NpT);
        const int t = int((i / (16l * NpT)) % taps);
        const int k = int(i / (16l * NpT * taps)) * 16 + kk;
        wpT[i] = (k < Cout && n < Cin) ? w[((long)(taps - 1 - t) * Cin + n) * Cout + k] : 0.f;
    }
}

// Every layer's forward and dgrad packing in ONE launch (292 separate ~3 us launches per optimizer step otherwise).
// Segment s covers workgroups [first_block, next first_block) of 256 packed elements each.
struct PackSeg { long w_off; long dst_off; int taps, Cin, Cout, Kp, Np, transposed, first_block; };
// conv_x3.h / conv_p3.h: element i of an fp32 packed buffer [chunk][tap][Np][16] as three bf16 planes [chunk][tap][Np][3][16]
__device__ __forceinline__ void x3_store_planes(__bf16* w3, long i, int Np, float v) {
    (void)Np;
    const __bf16 h0 = (__bf16)v;
    const float r1 = v - (float)h0;
    const __bf16 h1 = (__bf16)r1;
    const __bf16 h2 = (__bf16)(r1 - (float)h1);
    __bf16* d = w3 + (i >> 4) * 48 + (i & 15);
    d[0] = h0; d[16] = h1; d[32] = h2;
}
// a whole fp32 packed buffer -> planes (debug entry points; the handle's weights go through pack_all_kernel)
__global__ __launch_bounds__(256) void pack_x3_kernel(const float* wp, __bf16* w3, long total, int Np) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) x3_store_planes(w3, i, Np, wp[i]);
}

__global__ __launch_bounds__(256) void pack_all_kernel(const float* flat, float* wp, float* wpT, const PackSeg* segs, int nseg, __bf16* wp3, __bf16* wp3T) {
    int lo = 0, hi = nseg - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (segs[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const PackSeg sg = segs[lo];
    const long i = (long)((int)blockIdx.x - sg.first_block) * 256 + threadIdx.x;
    if (i >= (long)sg.taps * sg.Kp * sg.Np) return;
    const float* w = flat + sg.w_off;
    if (sg.transposed == 2) {                                   // bf16 forward packing [Kp/32][tap][Np][32], round to nearest even
        const int kk = int(i % 32);
        const int n = int((i / 32) % sg.Np);
        const int t = int((i / (32l * sg.Np)) % sg.taps);
        const int k = int(i / (32l * sg.Np * sg.taps)) * 32 + kk;
        reinterpret_cast<__bf16*>(wp + sg.dst_off)[i] = (__bf16)((k < sg.Cin && n < sg.Cout) ? w[((long)t * sg.Cin + k) * sg.Cout + n] : 0.f);
        return;
    }
    if (sg.transposed == 3) {                                   // bf16 dgrad packing: k = cout, n = cin, taps flipped
        const int kk = int(i % 32);
        const int n = int((i / 32) % sg.Np);
        const int t = int((i / (32l * sg.Np)) % sg.taps);
        const int k = int(i / (32l * sg.Np * sg.taps)) * 32 + kk;
        reinterpret_cast<__bf16*>(wpT + sg.dst_off)[i] =
            (__bf16)((k < sg.Cout && n < sg.Cin) ? w[((long)(sg.taps - 1 - t) * sg.Cin + n) * sg.Cout + k] : 0.f);
        return;
    }
    const int kk = int(i % 16);
    const int n = int((i / 16) % sg.Np);
    const int t = int((i / (16l * sg.Np)) % sg.taps);
    const int k = int(i / (16l * sg.Np * sg.taps)) * 16 + kk;
    if (!sg.transposed) {
        const float v = (k < sg.Cin && n < sg.Cout) ? w[((long)t * sg.Cin + k) * sg.Cout + n] : 0.f;
        wp[sg.dst_off + i] = v;
        if (wp3) x3_store_planes(wp3 + 3 * sg.dst_off, i, sg.Np, v);
    } else {                                                    // dgrad: k = cout, n = cin, taps flipped
        const float v = (k < sg.Cout && n < sg.Cin) ? w[((long)(sg.taps - 1 - t) * sg.Cin + n) * sg.Cout + k] : 0.f;
        wpT[sg.dst_off + i] = v;
        if (wp3T) x3_store_planes(wp3T + 3 * sg.dst_off, i, sg.Np, v);
    }
}

// eval-mode BatchReNorm fold (ops.py:173-180): scale = gamma*rsqrt(var+eps), shift = beta - mean*scale
__global__ __launch_bounds__(256) void bn_fold_kernel(const float* beta, const float* gamma, const float* mm,
                                                      const float* mv, float* scale, float* shift, int C, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        const float inv = (1.0f / sqrtf(mv[c] + eps)) * gamma[c];
        scale[c] = inv;
        shift[c] = beta[c] - mm[c] * inv;
    }
}

// ==============================================================================================
// graph builder (network/um_v1.py)
// ==============================================================================================
namespace {

struct Builder {
    dr_handle* h;
    int stem_count = 0, root_count = 0;
    bool in_stem = false;
    int cur_lane = 0;                  // lane of the ops being appended (net.h: executor lanes)

    Tensor* new_tensor(int H, int W, int C, const char* tag) {
        auto t = std::make_unique<Tensor>();
        t->id = (int)h->tensors.size();
        t->H = H; t->W = W; t->C = C; t->cs = dr_round_up(C, 4);
        t->tag = tag;
        h->tensors.push_back(std::move(t));
        return h->tensors.back().get();
    }
    TView whole(Tensor* t) { return TView{t, 0, t->C}; }
    TView slice(Tensor* t, int coff, int C) { return TView{t, coff, C}; }

    std::string next_name() {
        int& n = in_stem ? stem_count : root_count;
        std::string base = n == 0 ? "Conv" : "Conv_" + std::to_string(n);
        ++n;
        return (in_stem ? std::string("hg_imgproc/") : std::string()) + base;
    }

    // creates the ConvLayer (name order = creation order); the op is appended by add_conv_op
    int new_conv(int k, int stride, int cin, int cout, bool bn, bool relu, float wd, int H, int W) {
        ConvLayer c;
        c.name = next_name();
        c.k = k; c.stride = stride; c.cin = cin; c.cout = cout; c.bn = bn; c.relu = relu; c.wd = wd;
        c.H = H; c.W = W;
        h->convs.push_back(c);
        h->flops_per_crop += 2.0 * H * W * k * k * cin * cout;
        return (int)h->convs.size() - 1;
    }
    void add_conv_op(int ci, TView in, TView out, TView res = TView(), bool masked = false, int dropout = -1) {
        Op op;
        op.kind = h->convs[ci].k == 7 ? OP_STEM : OP_CONV;
        op.in = in; op.out = out; op.in2 = res; op.conv = ci; op.masked = masked; op.dropout = dropout;
        op.lane = cur_lane;
        h->ops.push_back(op);
    }

    // um_v1.py:18-48
    TView residual(TView ins, int num_out, TView dst = TView(), bool masked = false) {
        const int cin = ins.C;
        if (num_out <= 0) num_out = cin;
        const int half = cin / 2;
        const int H = ins.t->H, W = ins.t->W;
        const int k = h->cfg.kernel_size;
        const int c1 = new_conv(1, 1, cin, half, true, true, 0.0005f, H, W);
        const int c2 = new_conv(k, 1, half, half, true, true, 0.0005f, H, W);
        const int c3 = new_conv(1, 1, half, num_out, true, true, 0.0005f, H, W);
        int cs = -1;
        if (num_out != cin) cs = new_conv(1, 1, cin, num_out, true, true, 0.0005f, H, W);
        TView t1 = whole(new_tensor(H, W, half, "res.c1"));
        TView t2 = whole(new_tensor(H, W, half, "res.c2"));
        if (!dst.valid()) dst = whole(new_tensor(H, W, num_out, "res.out"));
        add_conv_op(c1, ins, t1, TView(), masked);
        add_conv_op(c2, t1, t2);
        TView skip = ins;
        if (cs >= 0) {
            skip = whole(new_tensor(H, W, num_out, "res.skip"));
            add_conv_op(cs, ins, skip, TView(), masked);
        }
        add_conv_op(c3, t2, dst, skip);
        return dst;
    }

    TView pool(TView in, int k) {
        const int Ho = (in.t->H + 1) / 2, Wo = (in.t->W + 1) / 2;
        TView out = whole(new_tensor(Ho, Wo, in.C, "pool"));
        Op op; op.kind = OP_POOL; op.in = in; op.out = out; op.pool_k = k; op.lane = cur_lane;
        h->ops.push_back(op);
        return out;
    }

    void edge(OpKind kind, int parent, int child) {
        Op op; op.kind = kind; op.lane = parent; op.lane2 = child; op.ev = (int)h->lane_ev.size();
        h->lane_ev.push_back(rt::event_create_sync());
        h->ops.push_back(op);
    }

    // um_v1.py:51-69.  Variables (conv names) are created in the reference's order: upper residual, lower
    // residual, inner hourglass, last residual.  The OPS are emitted pool-first so that the fork sits in front of
    // both branches: the upper residual runs on this level's lane, the pooled pyramid on the next lane.  Only this
    // lane touches grad(ins) in the reverse sweep (upper residual, then -- after the child lane joined -- the pool).
    TView hourglass(TView ins, int n, TView dst = TView()) {
        const int parent = cur_lane;
        const int child = std::min(parent + 1, DR_MAX_LANES - 1);
        const bool split = child != parent;
        const int pool_op = (int)h->ops.size();
        TView lower1 = pool(ins, h->cfg.kernel_size);
        if (split) edge(OP_FORK, parent, child);
        TView upper1 = residual(ins, 0);
        cur_lane = child;
        const int first_op = (int)h->ops.size();
        lower1 = residual(lower1, 0);
        TView lower2 = n > 1 ? hourglass(lower1, n - 1) : lower1;
        TView lower3 = residual(lower2, 0);
        if (ins.t->H == 16 && ins.t->W == 16 && n == 3) {      // everything at 8x8 and below: one launch in eval mode (hg_fused.h)
            FusedRegion fr;
            fr.pool_op = pool_op; fr.first_op = first_op; fr.last_op = (int)h->ops.size() - 1;
            fr.in = ins; fr.out = lower3;
            int nc = 0;
            for (int i = first_op; i <= fr.last_op; ++i)
                if (h->ops[i].kind == OP_CONV && nc < 24) fr.conv[nc++] = h->ops[i].conv;
            // hg_tail_eval_kernel hard-codes what `residual` emits today -- eight identity-skip modules of 1x1 F->F/2, 3x3 F/2->F/2,
            // 1x1 F/2->F, BatchReNorm + ReLU on every conv, on F input channels: a region of any other shape stays unfused
            const int F = h->cfg.num_fea;
            bool shape_ok = nc == 24 && ins.C == F && lower3.C == F;
            for (int i = 0; shape_ok && i < 24; ++i) {
                const ConvLayer& cl = h->convs[fr.conv[i]];
                const int k3 = i % 3;
                shape_ok = cl.bn && cl.relu && cl.stride == 1 && cl.k == (k3 == 1 ? 3 : 1) && cl.cin == (k3 == 0 ? F : F / 2) &&
                           cl.cout == (k3 == 2 ? F : F / 2);
            }
            for (int i = first_op; shape_ok && i <= fr.last_op; ++i) {      // the third conv of a module adds the module's own input
                const Op& o = h->ops[i];
                if (o.kind == OP_CONV && h->convs[o.conv].k == 1 && h->convs[o.conv].cout == F) shape_ok = o.in2.valid() && o.in2.C == F;
            }
            if (shape_ok) h->fused.push_back(fr);
        }
        cur_lane = parent;
        if (split) edge(OP_JOIN, parent, child);
        h->n_lanes = std::max(h->n_lanes, child + 1);
        if (!dst.valid()) dst = whole(new_tensor(ins.t->H, ins.t->W, ins.C, "hg.out"));
        Op op; op.kind = OP_UPADD; op.in = upper1; op.in2 = lower3; op.out = dst; op.lane = parent;
        h->ops.push_back(op);
        return dst;
    }

    void copy(TView src, TView dst) {
        Op op; op.kind = OP_COPY; op.in = src; op.out = dst; op.lane = cur_lane;
        h->ops.push_back(op);
    }

    // um_v1.py:71-185
    void detect_net() {
        const dr_config& c = h->cfg;
        const int F = c.num_fea, J = c.num_jnt, hw = c.in_hw, mh = hw / 4;
        h->map_hw = mh;
        const int num_resize = hw == 512 ? 6 : (hw == 256 ? 5 : 4);
        Tensor* input = new_tensor(hw, hw, 1, "input");
        input->needs_grad = false;
        h->input = input;

        in_stem = true;
        const int conv1 = new_conv(7, 2, 1, 32, true, true, 0.0005f, hw / 2, hw / 2);
        TView t_conv1 = whole(new_tensor(hw / 2, hw / 2, 32, "stem.conv1"));
        add_conv_op(conv1, whole(input), t_conv1);
        TView conv_2 = residual(t_conv1, 64);
        TView pool_1 = pool(conv_2, 2);
        TView conv_3 = residual(pool_1, 0);
        TView hg_ins = residual(conv_3, F);
        in_stem = false;

        // uvd channels are written once per destination concat buffer
        std::vector<Tensor*> LLU(c.num_stack), CMB(c.num_stack);
        for (int s = 0; s < c.num_stack; ++s) {
            LLU[s] = new_tensor(mh, mh, F + 3, "ll|uvd");
            CMB[s] = new_tensor(mh, mh, 512 + 3, "comb|uvd");
            Op op; op.kind = OP_UVD; op.in = whole(input);
            op.uvd0 = slice(LLU[s], F, 3); op.uvd1 = slice(CMB[s], 512, 3);
            h->ops.push_back(op);
        }

        for (int s = 0; s < c.num_stack; ++s) {
            Tensor* A = new_tensor(mh, mh, F + 2 * J, "hg|hm|hm3");
            TView hg_outs = hourglass(hg_ins, num_resize, slice(A, 0, F));
            TView ll = residual(hg_outs, 0);
            const int c_ll = new_conv(1, 1, F, F, true, true, 0.0005f, mh, mh);
            TView ll2 = slice(LLU[s], 0, F);
            add_conv_op(c_ll, ll, ll2);
            const int c_hm = new_conv(1, 1, F, J, false, false, 0.0005f, mh, mh);
            Tensor* HM = new_tensor(mh, mh, J, "hm_out");
            add_conv_op(c_hm, ll2, whole(HM));
            copy(whole(HM), slice(A, F, J));
            TView hm3_in = residual(whole(LLU[s]), 128);
            const int c_hm3 = new_conv(1, 1, 128, J, false, false, 0.0005f, mh, mh);
            Tensor* HM3 = new_tensor(mh, mh, J, "hm3_out");
            add_conv_op(c_hm3, hm3_in, whole(HM3));
            copy(whole(HM3), slice(A, F + J, J));

            Tensor* CAT = new_tensor(mh, mh, 512, "um_in|um_in_mask");
            TView um_a = residual(whole(A), 256);
            residual(um_a, 0, slice(CAT, 0, 256));
            TView um_m = residual(whole(A), 256, TView(), /*masked=*/true);
            residual(um_m, 0, slice(CAT, 256, 256));
            residual(whole(CAT), 0, slice(CMB[s], 0, 512));

            const int c_f1 = new_conv(1, 1, 515, 512, false, true, 0.0005f, mh, mh);
            TView f1 = whole(new_tensor(mh, mh, 512, "um_full1"));
            add_conv_op(c_f1, whole(CMB[s]), f1, TView(), false, s * 2 + 0);
            const int c_f2 = new_conv(1, 1, 512, 512, false, true, 0.0005f, mh, mh);
            TView f2 = whole(new_tensor(mh, mh, 512, "um_full2"));
            add_conv_op(c_f2, f1, f2, TView(), false, s * 2 + 1);
            const int c_um = new_conv(1, 1, 512, 3 * J, false, false, 0.0005f, mh, mh);
            Tensor* UM = new_tensor(mh, mh, 3 * J, "um_out");
            add_conv_op(c_um, f2, whole(UM));
            h->hm.push_back(HM); h->hm3.push_back(HM3); h->um.push_back(UM);

            if (s < c.num_stack - 1) {
                Tensor* T = new_tensor(mh, mh, 5 * J, "hm|hm3|um");
                copy(whole(HM), slice(T, 0, J));
                copy(whole(HM3), slice(T, J, J));
                copy(whole(UM), slice(T, 2 * J, 3 * J));
                const int c_t = new_conv(1, 1, 5 * J, F, false, false, 0.f, mh, mh);
                const int c_i = new_conv(1, 1, F, F, false, false, 0.f, mh, mh);
                TView x1 = whole(new_tensor(mh, mh, F, "reinject.tmp"));
                add_conv_op(c_t, whole(T), x1, hg_ins);
                TView x2 = whole(new_tensor(mh, mh, F, "hg_ins"));
                add_conv_op(c_i, ll2, x2, x1);
                hg_ins = x2;
            }
        }
    }
};

void add_param(dr_handle* h, const std::string& name, std::initializer_list<int> dims, bool trainable, ParamKind kind,
               int conv) {
    ParamInfo p;
    p.name = name; p.trainable = trainable; p.kind = kind; p.conv = conv;
    p.ndim = (int)dims.size();
    p.count = 1;
    int i = 0;
    for (int d : dims) { p.dims[i++] = d; p.count *= (size_t)d; }
    h->params.push_back(p);
}

}  // namespace

#include "pipeline.inc"

static void free_all(dr_handle* h) {
    (void)pipeline_drain(h);
    if (h->slot[0].act_arena) bind_slot(h, 0);             // the working fields name slot 0's buffers again: freed below
    free_slot1(h);
    for (void* p : {(void*)h->flat_param, (void*)h->flat_grad, (void*)h->adam_m, (void*)h->adam_v, (void*)h->flat_state, (void*)h->flat_state_next,
                    (void*)h->shadow, (void*)h->wp, (void*)h->wpT, (void*)h->wp3, (void*)h->wp3T, (void*)h->fold, (void*)h->bnc,
                    (void*)h->act_arena, (void*)h->grad_arena, (void*)h->scratch, (void*)h->tiny, (void*)h->tiny_ext, (void*)h->zeros,
                    (void*)h->losses, (void*)h->reg_segs, (void*)h->loss_acc, (void*)h->bn_coef, (void*)h->wg_partial, (void*)h->fold_dev, h->pack_dev, h->zero_dev,
                    (void*)h->g_keep_arena, (void*)h->pool_arg_arena, (void*)h->group_dev, (void*)h->bn_flags})
        if (p) rt::dfree(p);
    for (int l = 1; l < DR_MAX_LANES; ++l) {
        if (h->scratch_l[l]) rt::dfree(h->scratch_l[l]);
        if (h->wg_partial_l[l]) rt::dfree(h->wg_partial_l[l]);
        if (h->bn_coef_l[l]) rt::dfree(h->bn_coef_l[l]);
        rt::stream_destroy(h->lane_stream[l]);
    }
    for (int l = 0; l < DR_MAX_LANES; ++l) {
        if (h->stat_part_l[l]) rt::dfree(h->stat_part_l[l]);
        if (h->stat_part2_l[l]) rt::dfree(h->stat_part2_l[l]);
    }
    if (h->wg_stream) { if (rt::sync_stream_bounded(h->wg_stream) == 0) rt::stream_destroy(h->wg_stream); rt::event_destroy(h->wg_ready); rt::event_destroy(h->wg_done); }
    for (auto& e : h->lane_ev) rt::event_destroy(e);
    for (auto& g : h->graphs) rt::graph_destroy(g.g);
    rt::stream_destroy(h->cap_stream);
}

namespace {
int alloc_training_state(dr_handle* h);
}

// ==============================================================================================
// C ABI
// ==============================================================================================
extern "C" {

int dr_abi_version(void) { return DR_ABI_VERSION; }
const char* dr_backend(void) { return rt::backend_name(); }

const char* dr_last_error(const dr_handle* h) { return h ? h->err.c_str() : g_create_err.c_str(); }

int dr_create(const dr_config* cfg, dr_handle** out) {
    if (!cfg || !out) { g_create_err = "dr_create: null argument"; return DR_E_INVALID; }
    *out = nullptr;
    if (cfg->in_hw != 128 && cfg->in_hw != 256 && cfg->in_hw != 512) {
        g_create_err = "unknown input depth map shape (um_v1.py:106-107): in_hw must be 128, 256 or 512";
        return DR_E_UNSUPPORTED;
    }
    if (cfg->num_stack < 1 || cfg->num_fea < 8 || cfg->num_fea % 8 || cfg->num_jnt < 1 || cfg->max_batch < 1 ||
        cfg->kernel_size != 3) {
        g_create_err = "dr_create: need num_stack>=1, num_fea%8==0, num_jnt>=1, max_batch>=1, kernel_size==3";
        return DR_E_INVALID;
    }
    if (rt::device_count() <= cfg->device || rt::set_device(cfg->device)) {
        g_create_err = "dr_create: no such HIP device (the product path needs a gfx950 GPU; there is no CPU fallback)";
        return DR_E_DEVICE;
    }
    auto* h = new dr_handle();
    h->cfg = *cfg;
    Builder b{h};
    b.detect_net();

    // ---- parameter registry + flat layouts (TF creation order) -------------------------------
    size_t nt = 0, ns = 0, nsh = 0, nwp = 0, nwpT = 0, nfold = 0, nbnc = 0;
    for (int i = 0; i < (int)h->convs.size(); ++i) {
        ConvLayer& c = h->convs[i];
        const int taps = c.k * c.k;
        add_param(h, c.name + "/weights", {c.k, c.k, c.cin, c.cout}, true, PK_WEIGHT, i);
        c.w_off = nt; nt += (size_t)taps * c.cin * c.cout;
        if (c.bn) {
            const std::string bp = c.name + "/BatchReNorm/";
            add_param(h, bp + "beta", {c.cout}, true, PK_BETA, i);   c.beta_off = nt;  nt += c.cout;
            add_param(h, bp + "gamma", {c.cout}, true, PK_GAMMA, i); c.gamma_off = nt; nt += c.cout;
            add_param(h, bp + "moving_mean", {c.cout}, false, PK_MMEAN, i);    c.mm_off = ns; ns += c.cout;
            add_param(h, bp + "moving_variance", {c.cout}, false, PK_MVAR, i); c.mv_off = ns; ns += c.cout;
            add_param(h, bp + "r_max", {1}, false, PK_RMAX, i);
            add_param(h, bp + "d_max", {1}, false, PK_DMAX, i);
            add_param(h, bp + "curr_t", {1}, false, PK_CURRT, i);
            c.shadow_off = nsh; nsh += 2 * (size_t)c.cout;
            c.fold_off = nfold; nfold += 2 * (size_t)c.cout;
            c.bnc_off = nbnc; nbnc += 4 * (size_t)c.cout;
        } else {
            add_param(h, c.name + "/biases", {c.cout}, true, PK_BIAS, i); c.bias_off = nt; nt += c.cout;
        }
        if (c.k != 7) {
            c.Kp = dr_round_up(c.cin, 16);  c.Np = dr_round_up(c.cout, 32);
            c.KpT = dr_round_up(c.cout, 16); c.NpT = dr_round_up(c.cin, 32);
            c.wp_off = nwp;   nwp += (size_t)taps * c.Kp * c.Np;
            c.wpT_off = nwpT; nwpT += (size_t)taps * c.KpT * c.NpT;
        }
    }
    h->n_train = nt; h->n_state = ns; h->n_shadow = nsh; h->n_wp = nwp; h->n_wpT = nwpT; h->n_fold = nfold;
    h->n_bnc = nbnc;

    // ---- allocation ------------------------------------------------------------------------------
    const size_t MB = (size_t)cfg->max_batch;
    size_t nact = 0, max_t = 0;
    for (auto& t : h->tensors) {
        if (t.get() == h->input) continue;
        const size_t n = MB * t->H * t->W * t->cs;
        nact += n;
        max_t = std::max(max_t, n);
    }
    size_t nraw = 0;
    if (cfg->training)
        for (auto& c : h->convs)
            if (c.bn) nraw += MB * c.H * c.W * dr_round_up(c.cout, 4);
    h->n_act = nact + nraw;
    h->n_scratch = max_t;
    bool ok = true;
    auto alloc_f = [&](float*& p, size_t n) { p = (float*)rt::dmalloc(std::max<size_t>(n, 1) * sizeof(float)); ok = ok && p; };
    alloc_f(h->flat_param, nt);
    alloc_f(h->flat_state, ns);
    alloc_f(h->wp, nwp);
    h->wp3 = (__bf16*)rt::dmalloc(std::max<size_t>(nwp, 1) * 3 * sizeof(__bf16)); ok = ok && h->wp3;      // conv_x3.h
    alloc_f(h->fold, nfold * (cfg->training ? (size_t)kMaxGroups : 1));      // training: one copy per micro-batch group (dr_set_groups)
    alloc_f(h->act_arena, h->n_act);
    alloc_f(h->scratch, h->n_scratch);
    h->scratch_l[0] = h->scratch;
    {
        // Lanes are opt-in (DR_MULTI_STREAM=1): measured on MI355X after the small-grid kernels were tightened, the
        // event edges cost more than the overlap returns at every batch size (B=40: train equal, inference -1 %;
        // B=1: -11 %: such steps are bound by the host's launch rate, and every edge is two more API calls).
        const char* multi = getenv("DR_MULTI_STREAM");
        const char* single = getenv("DR_SINGLE_STREAM");
        h->multi_stream = multi && multi[0] == '1' && !(single && single[0] == '1');
        const char* graphs = getenv("DR_GRAPHS");
        h->use_graphs = graphs && graphs[0] == '1';      // opt-in: measured no gain (B=1: 1.87 ms either way, GPU-bound)
        // Created unconditionally, BEFORE the low-priority weight-gradient stream below, although only DR_GRAPHS=1 records on it:
        // measured on this stack (ROCm 7.2, MI355X), a process whose first library stream is the low-priority one runs the
        // training step at 870 crops/s instead of 2050 (the low-priority queue and the caller's are then scheduled one after the
        // other instead of side by side); with a normal-priority stream created first -- or a normal-priority side stream, which
        // measures the same 2050 -- the effect is gone (profiles/r03_experiments.md, visit 18).
        h->cap_stream = rt::stream_create();
        const char* ft = getenv("DR_FUSE_TAIL");
        h->fuse_tail = !(ft && ft[0] == '0');
        const char* fuse = getenv("DR_FUSE_BN_BWD");
        h->fuse_bn_bwd = !(fuse && fuse[0] == '0');
        const char* grp = getenv("DR_GROUP_WGRAD");
        h->group_wgrad = !(grp && grp[0] == '0');
        const char* ws = getenv("DR_WGRAD_STREAM");
        // default on (DR_WGRAD_STREAM=0: every weight gradient inline on the caller's stream; n > 1: additionally release the
        // queue after every (n-1)-th layer -- measured worse: beside the full-resolution kernels the side work only competes:
        // 1 -> 1927, 2 -> 1832, 3 -> 1843, 5 -> 1869, 9 -> 1880 crops/s against 1842 inline)
        h->wgrad_stream = !(ws && (ws[0] < '1' || ws[0] > '9')) && cfg->training && !h->multi_stream;
        h->wg_flush_every = (h->wgrad_stream && ws) ? atoi(ws) - 1 : 0;
        if (h->wgrad_stream) {
            // Measured and dropped (round 2): a CU-masked side stream (hipExtStreamCreateWithCUMask, 96-224 CUs: 31-37 ms per step
            // against 20.6) and side launches cut to ~256-768 workgroups so that every CU keeps free slots (1854-1916 crops/s
            // against 1918-1943 with the full-occupancy plan).
            static const bool wg_low = [] { const char* e = getenv("DR_WG_PRIO"); return !(e && e[0] == '0'); }();   // experiment: 0 = normal priority
            h->wg_stream = wg_low ? rt::stream_create_low_priority() : rt::stream_create();
            h->wg_ready = rt::event_create_sync();
            h->wg_done = rt::event_create_sync();
        }
        const char* a16 = getenv("DR_BF16_ACT");
        h->bf16_act = !(a16 && a16[0] == '0');
        const char* b16 = getenv("DR_BF16_DRAW");
        h->bf16_draw = !(b16 && b16[0] == '0');
        const char* ga16 = getenv("DR_BF16_GACT");
        h->bf16_gact = !(ga16 && ga16[0] == '0');
        const char* r16 = getenv("DR_BF16_RAW");
        h->bf16_raw = !(r16 && r16[0] == '0');
        // opt-in: measured slower on MI355X (train_kernels.h, bn_handoff_wait) -- BatchReNorm 5.5 -> 8.1 ms per step
        const char* lb = getenv("DR_BN_LOOKBACK");
        h->bn_lookback = lb && lb[0] == '1';
    }
    for (int l = 1; l < h->n_lanes; ++l) {
        h->lane_stream[l] = rt::stream_create();
        if (cfg->training) alloc_f(h->scratch_l[l], h->n_scratch);         // 288 GB of HBM: no need to be clever
    }
    alloc_f(h->tiny, MB * h->map_hw * h->map_hw);
    alloc_f(h->tiny_ext, MB * h->map_hw * h->map_hw);
    alloc_f(h->losses, 4 * (size_t)kMaxGroups);
    alloc_f(h->zeros, 64);
    if (cfg->training) {
        alloc_f(h->flat_grad, nt);
        alloc_f(h->adam_m, nt);
        alloc_f(h->adam_v, nt);
        alloc_f(h->shadow, nsh);
        alloc_f(h->flat_state_next, ns);
        alloc_f(h->wpT, nwpT);
        h->wp3T = (__bf16*)rt::dmalloc(std::max<size_t>(nwpT, 1) * 3 * sizeof(__bf16)); ok = ok && h->wp3T;
        alloc_f(h->bnc, nbnc * (size_t)kMaxGroups);
        h->n_gact = nact;
        alloc_f(h->grad_arena, nact);
        if (ok && alloc_training_state(h)) ok = false;
    }
    if (!ok) {
        free_all(h);
        delete h;
        g_create_err = "dr_create: device allocation failed";
        return DR_E_NOMEM;
    }
    size_t off = 0;
    for (auto& t : h->tensors) {
        if (t.get() == h->input) continue;
        t->off = off;
        t->p = h->act_arena + off;
        if (cfg->training) t->g = h->grad_arena + off;
        off += MB * t->H * t->W * t->cs;
    }
    if (cfg->training) {
        for (auto& c : h->convs) {
            if (!c.bn) continue;
            auto t = std::make_unique<Tensor>();
            t->id = (int)h->tensors.size();
            t->H = c.H; t->W = c.W; t->C = c.cout; t->cs = dr_round_up(c.cout, 4); t->tag = "raw";
            t->off = off;
            t->p = h->act_arena + off;
            off += MB * t->H * t->W * t->cs;
            c.raw = t.get();
            h->tensors.push_back(std::move(t));
        }
    }
    // pad channels are never written by the kernels: keep them (and everything else) finite
    rt::memset_async(h->act_arena, 0, h->n_act * sizeof(float), nullptr);
    rt::memset_async(h->zeros, 0, 64 * sizeof(float), nullptr);
    rt::memset_async(h->flat_param, 0, nt * sizeof(float), nullptr);
    rt::memset_async(h->flat_state, 0, std::max<size_t>(ns, 1) * sizeof(float), nullptr);
    if (cfg->training) {
        rt::memset_async(h->flat_grad, 0, nt * sizeof(float), nullptr);
        rt::memset_async(h->adam_m, 0, nt * sizeof(float), nullptr);
        rt::memset_async(h->adam_v, 0, nt * sizeof(float), nullptr);
        rt::memset_async(h->shadow, 0, std::max<size_t>(nsh, 1) * sizeof(float), nullptr);
        rt::memset_async(h->grad_arena, 0, nact * sizeof(float), nullptr);
    }
    rt::sync_stream(nullptr);
    slot_capture0(h);
    *out = h;
    return DR_OK;
}

void dr_destroy(dr_handle* h) {
    if (!h) return;
    rt::sync_stream(nullptr);
    free_all(h);
    delete h;
}

int dr_param_count(const dr_handle* h) { return h ? (int)h->params.size() : 0; }

int dr_param_info(const dr_handle* h, int index, const char** name, int32_t dims[4], int32_t* ndim, int32_t* trainable) {
    if (!h) return DR_E_INVALID;
    if (index < 0 || index >= (int)h->params.size()) DR_FAIL(h, DR_E_INVALID, "dr_param_info: index %d out of range", index);
    const ParamInfo& p = h->params[index];
    if (name) *name = p.name.c_str();
    if (dims) for (int i = 0; i < 4; ++i) dims[i] = p.dims[i];
    if (ndim) *ndim = p.ndim;
    if (trainable) *trainable = p.trainable ? 1 : 0;
    return DR_OK;
}

static const ParamInfo* find_param(const dr_handle* h, const char* name) {
    for (auto& p : h->params)
        if (p.name == name) return &p;
    return nullptr;
}

// Zero-debias slot variables of assign_moving_average ([TF1.3-semantics] moving_averages.py): "<var>/biased" (the
// un-corrected accumulator, same shape as the moving statistic) and "<var>/local_step" (one float, the number of
// updates).  They are not listed by dr_param_info (TF creates them lazily and a Saver checkpoint names them with
// the scope repeated); dr_load_param / dr_read_param accept them by these names so a checkpoint importer can
// carry the exact BatchReNorm state over.  Returns the base variable, *slot = 1 biased, 2 local_step.
static const ParamInfo* find_slot(const dr_handle* h, const char* name, int* slot) {
    const std::string n(name);
    for (int k = 1; k <= 2; ++k) {
        const std::string suf = k == 1 ? "/biased" : "/local_step";
        if (n.size() > suf.size() && n.compare(n.size() - suf.size(), suf.size(), suf) == 0) {
            const ParamInfo* p = find_param(h, n.substr(0, n.size() - suf.size()).c_str());
            if (p && (p->kind == PK_MMEAN || p->kind == PK_MVAR)) { *slot = k; return p; }
        }
    }
    return nullptr;
}

static float* param_dev_ptr(dr_handle* h, const ParamInfo& p) {
    ConvLayer& c = h->convs[p.conv];
    switch (p.kind) {
        case PK_WEIGHT: return h->flat_param + c.w_off;
        case PK_BETA: return h->flat_param + c.beta_off;
        case PK_GAMMA: return h->flat_param + c.gamma_off;
        case PK_BIAS: return h->flat_param + c.bias_off;
        case PK_MMEAN: return h->flat_state + c.mm_off;
        case PK_MVAR: return h->flat_state + c.mv_off;
        default: return nullptr;
    }
}

int dr_load_param(dr_handle* h, const char* name, const float* host, size_t count) {
    if (!h || !name || !host) return DR_E_INVALID;
    if (pipeline_drain(h)) DR_FAIL(h, DR_E_DEVICE, "a micro-step slot's stream did not drain");
    const ParamInfo* p = find_param(h, name);
    if (!p) {
        int slot = 0;
        const ParamInfo* base = find_slot(h, name, &slot);
        if (!base) DR_FAIL(h, DR_E_INVALID, "dr_load_param: unknown variable '%s'", name);
        if (!h->cfg.training) DR_FAIL(h, DR_E_STATE, "dr_load_param: '%s' exists only on a training handle", name);
        ConvLayer& cl = h->convs[base->conv];
        if (slot == 2) {
            if (count != 1) DR_FAIL(h, DR_E_INVALID, "dr_load_param: '%s' is a scalar", name);
            cl.shadow_step = (int)host[0];
            return DR_OK;
        }
        if (count != base->count) DR_FAIL(h, DR_E_INVALID, "dr_load_param: '%s' has %zu elements, got %zu", name, base->count, count);
        float* dst = h->shadow + cl.shadow_off + (base->kind == PK_MVAR ? cl.cout : 0);
        if (rt::h2d(dst, host, count * sizeof(float), nullptr)) DR_FAIL(h, DR_E_DEVICE, "h2d failed");
        rt::sync_stream(nullptr);
        return DR_OK;
    }
    if (count != p->count) DR_FAIL(h, DR_E_INVALID, "dr_load_param: '%s' has %zu elements, got %zu", name, p->count, count);
    ConvLayer& c = h->convs[p->conv];
    if (p->kind == PK_RMAX) { c.r_max = host[0]; return DR_OK; }
    if (p->kind == PK_DMAX) { c.d_max = host[0]; return DR_OK; }
    if (p->kind == PK_CURRT) { c.curr_t = host[0]; return DR_OK; }
    if (rt::h2d(param_dev_ptr(h, *p), host, count * sizeof(float), nullptr)) DR_FAIL(h, DR_E_DEVICE, "h2d failed");
    rt::sync_stream(nullptr);
    if (h->cfg.training && (p->kind == PK_MMEAN || p->kind == PK_MVAR)) {
        // Restoring moving stats WITHOUT the zero-debias slot variables (<var>/biased, <var>/local_step) =
        // a fresh shadow, exactly what TF1.3 does when a checkpoint lacks them (DESIGN.md "BatchReNorm state").
        c.shadow_step = 0;
        rt::memset_async(h->shadow + c.shadow_off, 0, 2 * (size_t)c.cout * sizeof(float), nullptr);
        rt::sync_stream(nullptr);
    }
    h->finalized = false;
    return DR_OK;
}

int dr_read_param(dr_handle* h, const char* name, float* host, size_t count) {
    if (!h || !name || !host) return DR_E_INVALID;
    if (pipeline_drain(h)) DR_FAIL(h, DR_E_DEVICE, "a micro-step slot's stream did not drain");
    const ParamInfo* p = find_param(h, name);
    if (!p) {
        int slot = 0;
        const ParamInfo* base = find_slot(h, name, &slot);
        if (!base) DR_FAIL(h, DR_E_INVALID, "dr_read_param: unknown variable '%s'", name);
        if (!h->cfg.training) DR_FAIL(h, DR_E_STATE, "dr_read_param: '%s' exists only on a training handle", name);
        ConvLayer& cl = h->convs[base->conv];
        if (slot == 2) {
            if (count != 1) DR_FAIL(h, DR_E_INVALID, "dr_read_param: '%s' is a scalar", name);
            host[0] = (float)cl.shadow_step;
            return DR_OK;
        }
        if (count != base->count) DR_FAIL(h, DR_E_INVALID, "dr_read_param: '%s' has %zu elements, got %zu", name, base->count, count);
        rt::sync_stream(nullptr);
        const float* src = h->shadow + cl.shadow_off + (base->kind == PK_MVAR ? cl.cout : 0);
        if (rt::d2h(host, src, count * sizeof(float), nullptr)) DR_FAIL(h, DR_E_DEVICE, "d2h failed");
        rt::sync_stream(nullptr);
        return DR_OK;
    }
    if (count != p->count) DR_FAIL(h, DR_E_INVALID, "dr_read_param: '%s' has %zu elements, got %zu", name, p->count, count);
    ConvLayer& c = h->convs[p->conv];
    if (p->kind == PK_RMAX) { host[0] = c.r_max; return DR_OK; }
    if (p->kind == PK_DMAX) { host[0] = c.d_max; return DR_OK; }
    if (p->kind == PK_CURRT) { host[0] = c.curr_t; return DR_OK; }
    rt::sync_stream(nullptr);
    if (rt::d2h(host, param_dev_ptr(h, *p), count * sizeof(float), nullptr)) DR_FAIL(h, DR_E_DEVICE, "d2h failed");
    rt::sync_stream(nullptr);
    return DR_OK;
}

static int repack_weights(dr_handle* h, hipStream_t s) {
    if (!h->pack_dev) {                                         // the table is fixed per handle: built and uploaded once
        std::vector<PackSeg> segs;
        int blocks = 0;
        for (auto& c : h->convs) {
            if (c.k == 7) continue;
            const int taps = c.k * c.k;
            if (h->precision == 1) {                            // half the bytes per element: always fits the fp32 slot
                const int Kp32 = dr_round_up(c.cin, 32);
                segs.push_back(PackSeg{(long)c.w_off, (long)c.wp_off, taps, c.cin, c.cout, Kp32, c.Np, 2, blocks});
                blocks += dr_ceil_div(taps * Kp32 * c.Np, 256);
                if (h->cfg.training) {
                    const int KpT32 = dr_round_up(c.cout, 32);
                    segs.push_back(PackSeg{(long)c.w_off, (long)c.wpT_off, taps, c.cin, c.cout, KpT32, c.NpT, 3, blocks});
                    blocks += dr_ceil_div(taps * KpT32 * c.NpT, 256);
                }
                continue;
            }
            segs.push_back(PackSeg{(long)c.w_off, (long)c.wp_off, taps, c.cin, c.cout, c.Kp, c.Np, 0, blocks});
            blocks += dr_ceil_div(taps * c.Kp * c.Np, 256);
            if (h->cfg.training) {
                segs.push_back(PackSeg{(long)c.w_off, (long)c.wpT_off, taps, c.cin, c.cout, c.KpT, c.NpT, 1, blocks});
                blocks += dr_ceil_div(taps * c.KpT * c.NpT, 256);
            }
        }
        h->pack_nseg = (int)segs.size();
        h->pack_blocks = blocks;
        h->pack_dev = rt::dmalloc(std::max<size_t>(1, segs.size()) * sizeof(PackSeg));
        if (!h->pack_dev) DR_FAIL(h, DR_E_NOMEM, "pack table");
        if (rt::h2d(h->pack_dev, segs.data(), segs.size() * sizeof(PackSeg), s)) DR_FAIL(h, DR_E_DEVICE, "h2d of the pack table failed");
        rt::sync_stream(s);
    }
    if (h->pack_blocks > 0)
        DR_LAUNCH(pack_all_kernel, dim3(h->pack_blocks), dim3(256), 0, s, (const float*)h->flat_param, h->wp, h->wpT,
                  (const PackSeg*)h->pack_dev, h->pack_nseg, h->precision == 0 ? h->wp3 : (__bf16*)nullptr,
                  h->precision == 0 ? h->wp3T : (__bf16*)nullptr);
    DR_CHECK_LAUNCH(h);
    return DR_OK;
}

static int fold_bn(dr_handle* h, hipStream_t s) {
    for (auto& c : h->convs) {
        if (!c.bn) continue;
        DR_LAUNCH(bn_fold_kernel, dim3(dr_ceil_div(c.cout, 256)), dim3(256), 0, s, (const float*)(h->flat_param + c.beta_off),
                  (const float*)(h->flat_param + c.gamma_off), (const float*)(h->flat_state + c.mm_off),
                  (const float*)(h->flat_state + c.mv_off), h->fold + c.fold_off, h->fold + c.fold_off + c.cout, c.cout,
                  0.001f);
    }
    DR_CHECK_LAUNCH(h);
    return DR_OK;
}

int dr_set_precision(dr_handle* h, int precision) {
    if (!h) return DR_E_INVALID;
    if (precision != DR_PREC_F32 && precision != DR_PREC_BF16) DR_FAIL(h, DR_E_INVALID, "dr_set_precision: unknown precision %d", precision);
    if (precision != h->precision) {
        if (pipeline_drain(h)) DR_FAIL(h, DR_E_DEVICE, "a micro-step slot's stream did not drain");
        rt::sync_stream(nullptr);
        if (h->pack_dev) { rt::dfree(h->pack_dev); h->pack_dev = nullptr; }     // the packing table depends on the element type
        h->precision = precision;
        h->finalized = false;                                                   // weights must be re-packed
        for (auto& t : h->tensors) t->is_bf16 = false;                          // what a forward stored is void with the old precision
        h->last_forward_train = false;
        for (auto& g : h->graphs) rt::graph_destroy(g.g);                       // recorded launches name the old kernels
        h->graphs.clear();
    }
    return DR_OK;
}

int dr_set_fusion(dr_handle* h, int on) {
    if (!h) return DR_E_INVALID;
    if ((on != 0) != h->fuse_tail) {
        for (auto& g : h->graphs) rt::graph_destroy(g.g);                       // recorded launches name the other set of kernels
        h->graphs.clear();
    }
    h->fuse_tail = on != 0;
    return DR_OK;
}

int dr_finalize_params(dr_handle* h, dr_stream stream) {
    if (!h) return DR_E_INVALID;
    DR_ENTER(h);
    if (pipeline_drain(h)) DR_FAIL(h, DR_E_DEVICE, "a micro-step slot's stream did not drain");
    hipStream_t s = (hipStream_t)stream;
    int rc = repack_weights(h, s);
    if (rc) return rc;
    rc = fold_bn(h, s);
    if (rc) return rc;
    if (rt::sync_stream_bounded(s)) DR_FAIL(h, DR_E_DEVICE, "dr_finalize_params: stream sync failed");
    DR_CHECK_LAUNCH(h);
    h->finalized = true;
    h->fold_is_eval = true;
    return DR_OK;
}

// CRC-32C (Castagnoli, reflected 0x82F63B78), the checksum of TensorFlow's tensor-bundle checkpoint files; host code,
// used by the checkpoint importer/exporter (densereg_amd/checkpoint.py) on multi-megabyte tensors.
uint32_t dr_crc32c(uint32_t crc, const void* data, size_t n) {
    static uint32_t table[8][256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
            table[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int t = 1; t < 8; ++t) table[t][i] = (table[t - 1][i] >> 8) ^ table[0][table[t - 1][i] & 0xFF];
        ready = true;
    }
    const unsigned char* p = (const unsigned char*)data;
    uint32_t c = ~crc;
    while (n >= 8) {
        uint32_t lo, hi;
        memcpy(&lo, p, 4); memcpy(&hi, p + 4, 4);
        lo ^= c;
        c = table[7][lo & 0xFF] ^ table[6][(lo >> 8) & 0xFF] ^ table[5][(lo >> 16) & 0xFF] ^ table[4][lo >> 24] ^
            table[3][hi & 0xFF] ^ table[2][(hi >> 8) & 0xFF] ^ table[1][(hi >> 16) & 0xFF] ^ table[0][hi >> 24];
        p += 8; n -= 8;
    }
    while (n--) c = table[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
    return ~c;
}

// ---- input front-end: handle-free, one workgroup per frame (frontend.h) --------------------------------------
int dr_crop_from_pose(int B, const float* frames, int H, int W, const float* pose, int J, const float* cfg, int icvl, float pad,
                      int out_hw, float* crops, float* new_cfg, float* com, dr_stream stream) {
    if (!frames || !pose || !cfg || !crops || !new_cfg || !com) return DR_E_INVALID;
    if (B < 1 || H < 1 || W < 1 || J < 1 || out_hw < 2 || !(pad > 0.f) || (long)H * W >= (1l << 31)) return DR_E_INVALID;
    CropParams p{};
    p.frames = frames; p.H = H; p.W = W; p.pose = pose; p.J = J; p.bbx = nullptr; p.cfg = cfg; p.icvl = icvl; p.pad = pad;
    p.out_hw = out_hw; p.crops = crops; p.new_cfg = new_cfg; p.com = com;
    DR_LAUNCH(crop_com_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, p);
    std::string m;
    return rt::last_error(&m) ? DR_E_DEVICE : DR_OK;
}

int dr_crop_from_bbx(int B, const float* frames, int H, int W, const float* bbx, const float* cfg, int out_hw, float* crops,
                     float* new_cfg, float* com, dr_stream stream) {
    if (!frames || !bbx || !cfg || !crops || !new_cfg || !com) return DR_E_INVALID;
    if (B < 1 || H < 1 || W < 1 || out_hw < 2 || (long)H * W >= (1l << 31)) return DR_E_INVALID;
    CropParams p{};
    p.frames = frames; p.H = H; p.W = W; p.pose = nullptr; p.J = 0; p.bbx = bbx; p.cfg = cfg; p.icvl = 0; p.pad = 20.f;
    p.out_hw = out_hw; p.crops = crops; p.new_cfg = new_cfg; p.com = com;
    DR_LAUNCH(crop_com_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, p);
    std::string m;
    return rt::last_error(&m) ? DR_E_DEVICE : DR_OK;
}

int dr_data_aug(int B, const float* dms, int H, int W, const float* pose, int J, const float* cfg, const float* com,
                const float* draws, float* out_dms, float* out_pose, dr_stream stream) {
    if (!dms || !pose || !cfg || !com || !draws || !out_dms || !out_pose) return DR_E_INVALID;
    if (B < 1 || H < 1 || W < 1 || J < 1 || J > 256 || dms == out_dms) return DR_E_INVALID;
    AugParams p{};
    p.dms = dms; p.H = H; p.W = W; p.pose = pose; p.J = J; p.cfg = cfg; p.com = com; p.draws = draws;
    p.out_dms = out_dms; p.out_pose = out_pose;
    DR_LAUNCH(data_aug_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, p);
    std::string m;
    return rt::last_error(&m) ? DR_E_DEVICE : DR_OK;
}

int dr_png_unfilter(const uint8_t* filtered, int height, int row_bytes, int bpp, uint8_t* out) {
    if (!filtered || !out || height < 1 || row_bytes < 1 || bpp < 1 || bpp > 8) return DR_E_INVALID;
    return png_unfilter_host(filtered, height, row_bytes, bpp, out) ? DR_E_INVALID : DR_OK;
}

int dr_depth_from_samples(const uint8_t* samples_dev, long npix, int mode, float* depth_dev, dr_stream stream) {
    if (!samples_dev || !depth_dev || npix < 1 || (mode != DR_SAMPLES_RGB8_GB && mode != DR_SAMPLES_GREY16_BE)) return DR_E_INVALID;
    if (((uintptr_t)samples_dev & 3) || ((uintptr_t)depth_dev & 15)) return DR_E_INVALID;
    DR_LAUNCH(depth_unpack_kernel, dim3(grid_for((npix + 3) / 4, 256, 1 << 30)), dim3(256), 0, (hipStream_t)stream, samples_dev, npix, mode, depth_dev);
    std::string m;
    return rt::last_error(&m) ? DR_E_DEVICE : DR_OK;
}

int dr_norm_dm(dr_handle* h, int B, const float* dm, const float* com, float* out, dr_stream stream) {
    if (!h || !dm || !com || !out) return DR_E_INVALID;
    if (B < 1) DR_FAIL(h, DR_E_INVALID, "dr_norm_dm: B=%d", B);
    DR_ENTER(h);
    const int npix = h->cfg.in_hw * h->cfg.in_hw;
    DR_LAUNCH(norm_dm_kernel, dim3(grid_for((long)B * npix)), dim3(256), 0, (hipStream_t)stream, dm, com, out, B, npix);
    DR_CHECK_LAUNCH(h);
    return DR_OK;
}

}  // extern "C"

// ==============================================================================================
// forward executor
// ==============================================================================================
static int run_conv_eval(dr_handle* h, const Op& op, int B, hipStream_t s) {
    const ConvLayer& c = h->convs[op.conv];
    if (op.kind == OP_STEM) {
        StemParams sp{};
        sp.x = h->dm_in; sp.B = B; sp.H = h->cfg.in_hw; sp.W = h->cfg.in_hw;
        sp.w = h->flat_param + c.w_off;
        sp.k = c.k; sp.stride = c.stride;
        const int total = std::max((c.H - 1) * c.stride + c.k - sp.H, 0);
        sp.pad_t = total / 2; sp.pad_l = total / 2;
        sp.Ho = c.H; sp.Wo = c.W;
        sp.y = op.out.t->p; sp.y_cs = op.out.t->cs;
        sp.scale = h->fold + c.fold_off; sp.shift = h->fold + c.fold_off + c.cout; sp.relu = c.relu;
        ProfScope ps(h, s, KID_STEM, 2.0 * B * c.H * c.W * c.k * c.k * c.cout,
                     4.0 * B * (sp.H * sp.W + (double)c.H * c.W * c.cout));
        DR_LAUNCH(stem_conv_kernel, dim3(dr_ceil_div(B * c.H * c.W, 64)), dim3(256), 0, s, sp);
        return DR_OK;
    }
    ConvParams p{};
    p.x = op.in.t->p; p.x_cs = op.in.t->cs; p.x_coff = op.in.coff; p.Cin = op.in.C;
    p.B = B; p.H = c.H; p.W = c.W; p.ksize = c.k;
    p.w = h->wp + c.wp_off; p.Kp = c.Kp; p.Np = c.Np;
    p.w3 = h->precision == 0 && h->wp3 ? h->wp3 + 3 * c.wp_off : nullptr;
    if (h->precision == 1) { p.bf16 = 1; p.Kp = dr_round_up(c.cin, 32); }
    p.y = op.out.t->p; p.y_cs = op.out.t->cs; p.y_coff = op.out.coff; p.Cout = c.cout;
    if (c.bn) { p.scale = h->fold + c.fold_off; p.shift = h->fold + c.fold_off + c.cout; }
    else { p.scale = nullptr; p.shift = h->flat_param + c.bias_off; }
    p.relu = c.relu;
    if (op.in2.valid()) { p.res = op.in2.t->p; p.res_cs = op.in2.t->cs; p.res_coff = op.in2.coff; }
    if (op.masked) { p.rowmask = h->tiny; p.mask_thresh = -0.9f; }
    p.zeros = h->zeros;
    ProfScope ps(h, s, conv_tile_id(p), 2.0 * B * c.H * c.W * c.k * c.k * c.cin * c.cout,
                 4.0 * B * c.H * c.W * ((double)c.cin + c.cout + (op.in2.valid() ? c.cout : 0)));
    if (launch_conv_igemm(p, s)) DR_FAIL(h, DR_E_STATE, "conv %s: unsupported layout", c.name.c_str());
    return DR_OK;
}

static int run_simple_op(dr_handle* h, const Op& op, int B, hipStream_t s, bool train = false) {
    const int kid = op.kind == OP_POOL ? KID_POOL : op.kind == OP_UPADD ? KID_UPADD : op.kind == OP_UVD ? KID_UVD : KID_COPY;
    double bytes = 0;
    if (op.kind == OP_POOL) bytes = 4.0 * B * op.in.C * ((double)op.in.t->H * op.in.t->W + (double)op.out.t->H * op.out.t->W);
    if (op.kind == OP_UPADD) bytes = 4.0 * B * op.out.C * (2.25 * op.out.t->H * op.out.t->W);
    if (op.kind == OP_COPY) bytes = 8.0 * B * op.in.C * op.in.t->H * op.in.t->W;
    ProfScope ps(h, s, kid, 0.0, bytes);
    switch (op.kind) {
        case OP_POOL: {
            const Tensor* ti = op.in.t; const Tensor* to = op.out.t;
            const int k = op.pool_k;
            const int total = std::max((to->H - 1) * 2 + k - ti->H, 0);
            DR_LAUNCH(maxpool_kernel, dim3(grid_for((long)B * to->H * to->W * (op.in.C / 4))), dim3(256), 0, s,
                      (const float*)ti->p, ti->cs, op.in.coff, B, ti->H, ti->W, op.in.C, k, total / 2, total / 2, to->p,
                      to->cs, op.out.coff, to->H, to->W, train ? op.pool_arg : (unsigned char*)nullptr);
            break;
        }
        case OP_UPADD: {
            const Tensor* to = op.out.t;
            DR_LAUNCH(upsample_add_kernel, dim3(grid_for((long)B * to->H * to->W * (op.out.C / 4))), dim3(256), 0, s,
                      (const float*)op.in.t->p, op.in.t->cs, op.in.coff, (const float*)op.in2.t->p, op.in2.t->cs,
                      op.in2.coff, to->p, to->cs, op.out.coff, B, to->H, to->W, op.out.C);
            break;
        }
        case OP_UVD: {
            const int mh = h->map_hw;
            DR_LAUNCH(uvd_kernel, dim3(grid_for((long)B * mh * mh)), dim3(256), 0, s, h->dm_in, B, h->cfg.in_hw, h->tiny,
                      op.uvd0.t->p, op.uvd0.t->cs, op.uvd0.coff, op.uvd1.t->p, op.uvd1.t->cs, op.uvd1.coff);
            break;
        }
        case OP_COPY: {
            const long M = (long)B * op.in.t->H * op.in.t->W;
            DR_LAUNCH(copy_channels_kernel, dim3(grid_for(M * op.in.C)), dim3(256), 0, s, (const float*)op.in.t->p,
                      op.in.t->cs, op.in.coff, op.out.t->p, op.out.t->cs, op.out.coff, M, op.in.C, 0);
            break;
        }
        default: break;
    }
    return DR_OK;
}

// the bottom of an hourglass as one launch (hg_fused.h)
static int run_fused_region(dr_handle* h, const FusedRegion& fr, int B, hipStream_t s) {
    HgFusedParams p{};
    p.x = fr.in.t->p; p.x_cs = fr.in.t->cs; p.x_coff = fr.in.coff;
    p.y = fr.out.t->p; p.y_cs = fr.out.t->cs; p.y_coff = fr.out.coff;
    p.B = B; p.F = fr.in.C;
    double flops = 0;
    for (int i = 0; i < 24; ++i) {
        const ConvLayer& c = h->convs[fr.conv[i]];
        p.conv[i].w = h->wp + c.wp_off; p.conv[i].Np = c.Np;
        p.conv[i].scale = h->fold + c.fold_off; p.conv[i].shift = h->fold + c.fold_off + c.cout;
        flops += 2.0 * B * c.H * c.W * c.k * c.k * c.cin * c.cout;
    }
    const size_t lds = (size_t)hg_fused_lds_floats(p.F) * sizeof(float);
    // (four waves per workgroup, one per SIMD; a variant with eight -- the 8x8 levels' row tiles split over two waves per SIMD --
    // measured equal, profiles/r04_experiments.md, and is not instantiated)
    // the > 64 KB dynamic-LDS attribute is a property of (kernel, device): asked once per device, and only for the F that needs it
    if (lds > 64 * 1024) {
        static std::mutex mu;
        static std::map<std::pair<int, int>, bool> granted;
        std::lock_guard<std::mutex> lk(mu);
        const auto key = std::make_pair(h->cfg.device, p.F);
        auto it = granted.find(key);
        if (it == granted.end()) {
            const void* fn = p.F == 96 ? (const void*)hg_tail_eval_kernel<96, 4> : (const void*)hg_tail_eval_kernel<128, 4>;
            it = granted.emplace(key, rt::allow_dyn_lds(fn, lds)).first;
        }
        if (!it->second) return DR_E_UNSUPPORTED;              // (the caller falls back to the unfused ops)
    }
    ProfScope ps(h, s, KID_HG_FUSED, flops, 4.0 * B * p.F * (256.0 + 64.0));
    switch (p.F) {                                              // (hg_fused_supported: multiples of 32 up to 128)
        case 32: DR_LAUNCH((hg_tail_eval_kernel<32, 4>), dim3(B), dim3(256), lds, s, p); break;
        case 64: DR_LAUNCH((hg_tail_eval_kernel<64, 4>), dim3(B), dim3(256), lds, s, p); break;
        case 96: DR_LAUNCH((hg_tail_eval_kernel<96, 4>), dim3(B), dim3(256), lds, s, p); break;
        default: DR_LAUNCH((hg_tail_eval_kernel<128, 4>), dim3(B), dim3(256), lds, s, p); break;
    }
    return DR_OK;
}

static bool fused_tail_usable(const dr_handle* h) {
    return h->fuse_tail && !h->fused.empty() && !h->multi_stream && h->precision == 0 && hg_fused_supported(h->cfg.num_fea) &&
           h->cfg.kernel_size == 3;
}

static int forward_eval_impl(dr_handle* h, int B, const float* dm, hipStream_t s) {
    if (!h->finalized) DR_FAIL(h, DR_E_STATE, "forward before dr_finalize_params");
    if (pipeline_drain(h)) DR_FAIL(h, DR_E_DEVICE, "a micro-step slot's stream did not drain");   // (two micro-step slots: the bound slot's buffers are reused)
    if (B < 1 || B > h->cfg.max_batch) DR_FAIL(h, DR_E_INVALID, "batch %d outside [1, max_batch=%d]", B, h->cfg.max_batch);
    DR_ENTER(h);
    h->dm_in = dm;
    for (auto& t : h->tensors) t->is_bf16 = false;          // every conv epilogue of this pass writes fp32
    if (!h->fold_is_eval) {          // a training forward overwrote the per-layer scale/shift
        int rc = fold_bn(h, s);
        if (rc) return rc;
        h->fold_is_eval = true;
    }
    bool fuse = fused_tail_usable(h);
    size_t next_region = 0;                                  // regions are disjoint and in op order
    for (int i = 0; i < (int)h->ops.size(); ++i) {
        const Op& op = h->ops[i];
        if (fuse && next_region < h->fused.size()) {
            const FusedRegion& fr = h->fused[next_region];
            if (i == fr.pool_op) {                           // the pool and everything in [first_op, last_op]: one launch, here
                h->prof_tag = -1;
                int rc = run_fused_region(h, fr, B, s);
                if (rc == DR_OK) continue;
                if (rc != DR_E_UNSUPPORTED) return rc;
                // the device refused the kernel's LDS: this handle runs the region's ops one by one from here on (nothing of the
                // region has been skipped yet: its pool is this op)
                fprintf(stderr, "densereg: fused hourglass bottom unavailable on device %d, running the unfused ops\n", h->cfg.device);
                h->fuse_tail = false;
                fuse = false;
            }
            if (fuse && i >= fr.first_op && i <= fr.last_op) {
                if (i == fr.last_op) ++next_region;
                continue;
            }
        }
        if (op.kind == OP_FORK || op.kind == OP_JOIN) { lane_edge(h, op, op.kind == OP_FORK, s); continue; }
        h->prof_tag = op.conv;
        hipStream_t ls = lane_of(h, op.lane, s);
        int rc = (op.kind == OP_CONV || op.kind == OP_STEM) ? run_conv_eval(h, op, B, ls) : run_simple_op(h, op, B, ls);
        if (rc) return rc;
    }
    h->prof_tag = -1;
    DR_CHECK_LAUNCH(h);
    h->last_forward_train = false;
    h->last_eval_fused = fuse;
    h->last_B = B;
    return DR_OK;
}

// Replay (or record, the first time) the launches of an inference entry point as ONE executable graph (opt-in,
// DR_GRAPHS=1).  A B=1 forward + vote is ~150 launches; measured on MI355X it is NOT bound by the host's launch rate
// (1.87 ms per step with plain launches and with the graph: each tiny kernel is a 7-20 us dependent chain on the
// GPU), so this only pays where the host is slow or shared.  The graph is
// recorded on a library-owned stream (the caller's may be the legacy default stream, which cannot be captured) and
// launched into the caller's stream.  Keyed by entry point, batch and every caller pointer baked into the launches.
template <typename Fn>
static int run_with_graph(dr_handle* h, int entry, int B, std::initializer_list<const void*> ptrs, hipStream_t s, Fn&& direct) {
    const bool usable = h->use_graphs && !h->profiling && !h->multi_stream && h->cap_stream && h->finalized;
    if (!usable) return direct(s);
    if (B < 1 || B > h->cfg.max_batch) return direct(s);                  // let the direct path report the error
    DR_ENTER(h);
    if (!h->fold_is_eval) {                                                  // outside the recording: it is conditional
        int rc = fold_bn(h, s);
        if (rc) return rc;
        h->fold_is_eval = true;
    }
    dr_handle::GraphEntry key{};
    key.entry = entry; key.B = B;
    int n = 0;
    for (const void* q : ptrs) key.ptr[n++] = q;
    for (auto& g : h->graphs)
        if (g.entry == key.entry && g.B == key.B && !memcmp(g.ptr, key.ptr, sizeof(key.ptr))) {
            if (!rt::graph_launch(g.g, s)) DR_FAIL(h, DR_E_DEVICE, "graph launch failed");
            h->dm_in = (const float*)key.ptr[0];
            h->last_forward_train = false;
            h->last_B = B;
            return DR_OK;
        }
    static const bool dbg = getenv("DR_GRAPH_DEBUG") != nullptr;
    if (h->graphs.size() >= 16 || !rt::capture_begin(h->cap_stream)) {
        if (dbg) fprintf(stderr, "[densereg] graph: capture_begin refused (%zu cached)\n", h->graphs.size());
        return direct(s);
    }
    const int rc = direct(h->cap_stream);
    const bool ok = rt::capture_end(h->cap_stream, &key.g);
    if (dbg) fprintf(stderr, "[densereg] graph: recorded entry %d B=%d rc=%d instantiate=%d\n", entry, B, rc, (int)ok);
    if (rc != DR_OK) { if (ok) rt::graph_destroy(key.g); return rc; }
    if (!ok) return direct(s);                                               // e.g. capture not supported: plain launches
    h->graphs.push_back(key);
    if (!rt::graph_launch(key.g, s)) DR_FAIL(h, DR_E_DEVICE, "graph launch failed");
    return DR_OK;
}


static void copy_out(dr_handle* h, const Tensor* t, int B, float* dst, hipStream_t s) {
    const long M = (long)B * t->H * t->W;
    DR_LAUNCH(copy_channels_kernel, dim3(grid_for(M * t->C)), dim3(256), 0, s, (const float*)t->p, t->cs, 0, dst, t->C, 0, M,
              t->C, 0);
}

static int vote_impl(dr_handle* h, int B, View hm, View hm3, View um, const float* tiny, const float* cfg, const float* com,
                     float* xyz, hipStream_t s) {
    VoteParams vp{};
    vp.hm = hm; vp.hm3 = hm3; vp.um = um; vp.tiny = tiny; vp.cfg = cfg; vp.com = com; vp.xyz_mm = xyz; vp.xyz_norm = nullptr;
    vp.B = B; vp.h = h->map_hw; vp.w = h->map_hw; vp.J = h->cfg.num_jnt;
    const int npix = vp.h * vp.w;
    if (npix > kVoteMaxPix) DR_FAIL(h, DR_E_UNSUPPORTED, "vote: map of %d px exceeds the LDS plan", npix);
    {
        ProfScope ps(h, s, KID_VOTE, 0.0, (double)B * ((5.0 * vp.J + 1.0) * npix * 4.0 + 12.0 * vp.J));
        DR_LAUNCH(vote_kernel, dim3(B, dr_ceil_div(vp.J, kVoteJC)), dim3(256), (size_t)kVoteJC * npix * sizeof(float), s, vp);
    }
    DR_CHECK_LAUNCH(h);
    return DR_OK;
}

extern "C" {

int dr_forward_eval(dr_handle* h, int B, const float* dm, float* hm, float* hm3, float* um, dr_stream stream) {
    if (!h || !dm) return DR_E_INVALID;
    return run_with_graph(h, 2, B, {dm, hm, hm3, um}, (hipStream_t)stream, [&](hipStream_t s) {
        int rc = forward_eval_impl(h, B, dm, s);
        if (rc) return rc;
        return dr_read_maps(h, B, h->cfg.num_stack - 1, hm, hm3, um, (dr_stream)s);
    });
}

int dr_read_maps(dr_handle* h, int B, int stack, float* hm, float* hm3, float* um, dr_stream stream) {
    if (!h) return DR_E_INVALID;
    if (stack < 0 || stack >= h->cfg.num_stack) DR_FAIL(h, DR_E_INVALID, "dr_read_maps: stack %d", stack);
    if (B < 1 || B > h->last_B) DR_FAIL(h, DR_E_INVALID, "dr_read_maps: B=%d but the last forward ran %d", B, h->last_B);
    hipStream_t s = (hipStream_t)stream;
    if (h->pipe_depth == 2 && h->slot[h->cur_slot].fwd_recorded) rt::stream_wait_event(s, h->slot[h->cur_slot].fwd_done);
    if (hm) copy_out(h, h->hm[stack], B, hm, s);
    if (hm3) copy_out(h, h->hm3[stack], B, hm3, s);
    if (um) copy_out(h, h->um[stack], B, um, s);
    DR_CHECK_LAUNCH(h);
    return DR_OK;
}

int dr_vote(dr_handle* h, int B, const float* hm, const float* hm3, const float* um, const float* dm, const float* cfg,
            const float* com, float* xyz, dr_stream stream) {
    if (!h || !hm || !hm3 || !um || !dm || !cfg || !com || !xyz) return DR_E_INVALID;
    if (B < 1 || B > h->cfg.max_batch) DR_FAIL(h, DR_E_INVALID, "batch %d outside [1, max_batch=%d]", B, h->cfg.max_batch);
    DR_ENTER(h);
    hipStream_t s = (hipStream_t)stream;
    const int J = h->cfg.num_jnt, mh = h->map_hw;
    DR_LAUNCH(uvd_kernel, dim3(grid_for((long)B * mh * mh)), dim3(256), 0, s, dm, B, h->cfg.in_hw, h->tiny_ext, (float*)nullptr,
              0, 0, (float*)nullptr, 0, 0);
    return vote_impl(h, B, View{(float*)hm, J, 0, J}, View{(float*)hm3, J, 0, J}, View{(float*)um, 3 * J, 0, 3 * J},
                     h->tiny_ext, cfg, com, xyz, s);
}

int dr_infer(dr_handle* h, int B, const float* dm, const float* cfg, const float* com, float* xyz, dr_stream stream) {
    if (!h || !dm || !cfg || !com || !xyz) return DR_E_INVALID;
    return run_with_graph(h, 1, B, {dm, cfg, com, xyz}, (hipStream_t)stream, [&](hipStream_t s) {
        int rc = forward_eval_impl(h, B, dm, s);
        if (rc) return rc;
        const int S = h->cfg.num_stack - 1;
        TView a{h->hm[S], 0, h->hm[S]->C}, b{h->hm3[S], 0, h->hm3[S]->C}, c{h->um[S], 0, h->um[S]->C};
        return vote_impl(h, B, a.fwd(), b.fwd(), c.fwd(), h->tiny, cfg, com, xyz, s);
    });
}

int dr_read_activation(dr_handle* h, const char* scope, int B, float* host, size_t count) {
    if (!h || !scope || !host) return DR_E_INVALID;
    if (pipeline_drain(h)) DR_FAIL(h, DR_E_DEVICE, "a micro-step slot's stream did not drain");
    // "<scope>#raw": the layer's output BEFORE BatchReNorm (kept by a training forward: the backward pass needs it); "<scope>#fold":
    // [scale | shift], 2 * cout floats, the multiply-add that training forward applied to it (micro-batch group 0).  What a ReLU of
    // that forward decided is then exact on the host: raw * scale + shift > 0 (tests/test_train_parity.py, the switch-free gradient test).
    std::string want(scope);
    int part = 0;
    if (want.size() > 4 && want.compare(want.size() - 4, 4, "#raw") == 0) { part = 1; want.resize(want.size() - 4); }
    else if (want.size() > 5 && want.compare(want.size() - 5, 5, "#fold") == 0) { part = 2; want.resize(want.size() - 5); }
    for (size_t oi = 0; oi < h->ops.size(); ++oi) {
        const Op& op = h->ops[oi];
        if ((op.kind != OP_CONV && op.kind != OP_STEM) || h->convs[op.conv].name != want) continue;
        if (part) {
            const ConvLayer& c = h->convs[op.conv];
            if (!c.bn || !h->last_forward_train || !c.raw || c.raw->is_bf16) DR_FAIL(h, DR_E_STATE, "dr_read_activation: %s needs a BatchReNorm layer after an fp32 training forward", scope);
            if (part == 2) {
                if (count != 2 * (size_t)c.cout) DR_FAIL(h, DR_E_INVALID, "dr_read_activation: %s has %d elements, got %zu", scope, 2 * c.cout, count);
                rt::sync_stream(nullptr);
                rt::d2h(host, h->fold + c.fold_off, count * sizeof(float), nullptr);
                rt::sync_stream(nullptr);
                DR_CHECK_LAUNCH(h);
                return DR_OK;
            }
            const long Mr = (long)B * c.H * c.W;
            if (count != (size_t)Mr * c.cout) DR_FAIL(h, DR_E_INVALID, "dr_read_activation: %s has %zu elements, got %zu", scope, (size_t)Mr * c.cout, count);
            if (count > h->n_scratch) DR_FAIL(h, DR_E_STATE, "scratch too small");
            DR_LAUNCH(copy_channels_kernel, dim3(grid_for(Mr * c.cout)), dim3(256), 0, (hipStream_t) nullptr, (const float*)c.raw->p,
                      c.raw->cs, 0, h->scratch, c.cout, 0, Mr, c.cout, 0);
            rt::sync_stream(nullptr);
            rt::d2h(host, h->scratch, count * sizeof(float), nullptr);
            rt::sync_stream(nullptr);
            DR_CHECK_LAUNCH(h);
            return DR_OK;
        }
        if (!h->last_forward_train && h->last_eval_fused)
            for (const FusedRegion& fr : h->fused)
                if ((int)oi >= fr.first_op && (int)oi < fr.last_op)          // (the region's last conv IS written: its output tensor)
                    DR_FAIL(h, DR_E_STATE, "dr_read_activation: %s lives only in LDS when the hourglass bottom runs as one launch; dr_set_fusion(h, 0) keeps every layer's output", scope);
        const Tensor* t = op.out.t;
        const long M = (long)B * t->H * t->W;
        const size_t need = (size_t)M * op.out.C;
        if (count != need) DR_FAIL(h, DR_E_INVALID, "dr_read_activation: %s has %zu elements, got %zu", scope, need, count);
        if (need > h->n_scratch) DR_FAIL(h, DR_E_STATE, "scratch too small");
        if (t->is_bf16)                     // stored as bf16 by the training forward of the bf16 path (its only reader is a conv): widen
            DR_LAUNCH(copy_channels_from_bf16_kernel, dim3(grid_for(M * op.out.C)), dim3(256), 0, (hipStream_t) nullptr,
                      reinterpret_cast<const __bf16*>(t->p), t->cs, op.out.coff, h->scratch, op.out.C, 0, M, op.out.C);
        else
            DR_LAUNCH(copy_channels_kernel, dim3(grid_for(M * op.out.C)), dim3(256), 0, (hipStream_t) nullptr, (const float*)t->p,
                      t->cs, op.out.coff, h->scratch, op.out.C, 0, M, op.out.C, 0);
        rt::sync_stream(nullptr);
        rt::d2h(host, h->scratch, need * sizeof(float), nullptr);
        rt::sync_stream(nullptr);
        DR_CHECK_LAUNCH(h);
        return DR_OK;
    }
    DR_FAIL(h, DR_E_INVALID, "dr_read_activation: unknown conv scope '%s'", scope);
}

double dr_conv_flops_per_crop(const dr_handle* h) { return h ? h->flops_per_crop : 0.0; }

}  // extern "C"

#include "train_exec.inc"

// ==============================================================================================
// test hooks (include/densereg_debug.h): compiled into libdensereg_hip_dbg.so and the emulator build only
// ==============================================================================================
#include "../../include/densereg_profile.h"
#if defined(DR_DEBUG_HOOKS)
#include "../../include/densereg_debug.h"
#include "debug_hooks.inc"
#endif

// diagnostic of the opt-in BatchReNorm look-back hand-off (DR_BN_LOOKBACK=1): bounded waits that ran out; 0 otherwise
extern "C" int dr_lookback_expired(dr_handle* h) {
    if (!h || !h->bn_flags) return 0;
    rt::sync_stream(nullptr);
    std::vector<int> f(h->convs.size() * 4, 0);
    rt::d2h(f.data(), h->bn_flags, f.size() * sizeof(int), nullptr);
    rt::sync_stream(nullptr);
    int n = 0;
    for (size_t i = 0; i < h->convs.size(); ++i) n += f[4 * i + 1] + f[4 * i + 3];
    return n;
}

extern "C" int dr_profile_enable(dr_handle* h, int on) {
    if (!h) return DR_E_INVALID;
    if (pipeline_drain(h)) DR_FAIL(h, DR_E_DEVICE, "a micro-step slot's stream did not drain");
    rt::sync_stream(nullptr);
    for (auto& r : h->prof) { rt::event_destroy(r.a); rt::event_destroy(r.b); }
    h->prof.clear();
    h->profiling = on != 0;
    return DR_OK;
}

// one row per (kernel, conv layer): name = "<kernel>:<scope> k<k> <cin>-><cout> @<H>"; does not reset
extern "C" int dr_profile_detail(dr_handle* h, dr_kernel_stat* out, int max_out, int* n_out) {
    if (!h || !out || !n_out) return DR_E_INVALID;
    rt::sync_stream(nullptr);
    std::map<std::pair<int, int>, dr_kernel_stat> agg;
    for (auto& r : h->prof) {
        auto key = std::make_pair(r.kid, r.tag);
        auto it = agg.find(key);
        if (it == agg.end()) {
            dr_kernel_stat st;
            memset(&st, 0, sizeof(st));
            if (r.tag >= 0) {
                const ConvLayer& c = h->convs[r.tag];
                snprintf(st.name, sizeof(st.name), "%s:%s k%d %d->%d @%d", kKernelNames[r.kid], c.name.c_str(), c.k, c.cin, c.cout, c.H);
            } else {
                snprintf(st.name, sizeof(st.name), "%s", kKernelNames[r.kid]);
            }
            it = agg.emplace(key, st).first;
        }
        it->second.launches += 1;
        it->second.total_ms += rt::event_elapsed_ms(r.a, r.b);
        it->second.flops += r.flops;
        it->second.bytes += r.bytes;
    }
    int n = 0;
    for (auto& kv : agg)
        if (n < max_out) out[n++] = kv.second;
    *n_out = n;
    return DR_OK;
}

extern "C" int dr_profile_read(dr_handle* h, dr_kernel_stat* out, int max_out, int* n_out) {
    if (!h || !out || !n_out) return DR_E_INVALID;
    rt::sync_stream(nullptr);
    dr_kernel_stat agg[KID_COUNT];
    memset(agg, 0, sizeof(agg));
    for (int k = 0; k < KID_COUNT; ++k) snprintf(agg[k].name, sizeof(agg[k].name), "%s", kKernelNames[k]);
    for (auto& r : h->prof) {
        agg[r.kid].launches += 1;
        agg[r.kid].total_ms += rt::event_elapsed_ms(r.a, r.b);
        agg[r.kid].flops += r.flops;
        agg[r.kid].bytes += r.bytes;
        rt::event_destroy(r.a); rt::event_destroy(r.b);
    }
    h->prof.clear();
    int n = 0;
    for (int k = 0; k < KID_COUNT && n < max_out; ++k)
        if (agg[k].launches) out[n++] = agg[k];
    *n_out = n;
    return DR_OK;
}
