// Weight-resident persistent 3-D convolution (k = 3, pad 1) for Cin = 32, Cout = 64 - layer 2 of
// the voxel encoder, the kernel bench.py's roofline line is about.  bf16 MFMA, fp32 accumulate,
// channels-last (NDHWC) bf16 in, channels-last bf16 out + per-channel BatchNorm sums.
//
//   Y[b, v, n] = bias[n] + sum_{tap, c} X[b, v + off(tap), c] * W[n, tap, c]      (implicit GEMM
//   M = B*D*H*W rows, N = 64, K = 27 * 32 = 864; 2*M*N*K FLOPs, algorithmic bytes = 64 M in
//   + 128 M out + 108 KiB of weights)
//
// One workgroup per CU keeps all 64 x 27 x 32 bf16 weights (108 KiB) in LDS for its lifetime and
// walks 4 x 8 x 8 output tiles (256 GEMM rows; wave w owns depth slice w = 64 rows x 64 columns =
// 2 x 2 MFMA 32x32x16 tiles).  The (6 x 10 x 10)-voxel input halo of a tile is staged once into
// LDS; every tap reads its A fragments from it at a row offset, so the im2col matrix only ever
// exists as LDS addresses.
//
// What changed against round 1's kernel, and why (profiles/r02_*):
//  * output is bf16 (16.8 MB at C2 instead of 33.5 MB of fp32): the pre-BatchNorm tensor is read
//    back by the pooling pass and the backward in bf16, the statistics are taken here from the fp32
//    accumulators.  Channel n of MFMA column-tile j, lane column lr is 2*lr + j, so a lane packs
//    its two tiles' values of one voxel into ONE dword and a half-wave store is one 128-byte voxel row.
//  * LDS addressing costs no VALU in the K loop: rows are unpadded 64-byte rows whose 16-byte
//    slots are XOR-swizzled with a key that is separable from the tap offset - (h + kh) & 3 for
//    the halo (h = the lane's row in the tile face), (row >> 2) & 3 for the weights - so six A and
//    eight B per-lane base registers plus immediate offsets address all 54 x 4 fragment reads
//    (round 1's key, (row >> 2) & 3 of the shifted halo row, needed ~6 VALU per K-step).
//    GEMM row -> voxel stays (h, w) = (lr >> 3, (lr & 3) + 4 * parity(lr >> 2)) on a w-pitch of
//    12 rows: each ds_read_b128 lane group ({0-3,12-15,20-27}, ...) covers a 4 x 4 voxel patch =
//    4 distinct w (4 R mod 16) x 4 distinct keys = 16 distinct 16-byte slots: conflict-free.
//  * tile -> workgroup map is XCD-aware: workgroups b and b + 8 share an XCD (round-robin
//    dispatch), so XCD x takes the x-th eighth of the (b, d, h, w)-ordered tile list and its
//    workgroups walk it interleaved: neighbouring tiles' halos are fetched into that XCD's L2
//    once instead of once per XCD (2.1x -> ~1.0x input traffic).  Placement is a speed
//    assumption only; any placement computes the same result.
//  * the previous tile's stores are spread evenly over the 54 K-steps of the current one.
#include "conv3d_args.h"
#ifdef WRES_ASM_INC                              // ablation builds (tools/abl_build.sh) substitute a variant stream
#include WRES_ASM_INC
#else
#include "conv3d_wres_asm.inc"
#endif

#include <mutex>

namespace {

constexpr int CIN = 32;
constexpr int BN = 64;
constexpr int TD = 4;                            // tile depth: 4 x 8 x 8 = 256 GEMM rows
constexpr int HB = 10;                           // halo edge of an 8-wide tile face
constexpr int WP = 12;                           // halo w-pitch in LDS rows (10 used)
constexpr int DP = HB * WP;                      // halo d-pitch (120 rows)
constexpr int HROWS = (TD + 2) * HB * HB;        // 600 rows fetched per tile
constexpr int LROWS = (TD + 2) * DP;             // 720 LDS rows
constexpr int HREGS = (HROWS * 4 + 255) / 256;   // 16-byte chunks per thread per halo (10)
constexpr int ROWB = CIN * 2;                    // bytes per LDS row (64)
constexpr int W_BYTES = 27 * BN * ROWB;          // 110 592
constexpr int H_OFF = W_BYTES;
constexpr int H_BYTES = LROWS * ROWB;            // 46 080
constexpr int S_OFF = H_OFF + H_BYTES;           // per-wave BatchNorm partial sums [4][2][64] fp32
constexpr int LDS_BYTES = S_OFF + 4 * 2 * BN * 4;
static_assert(LDS_BYTES <= 160 * 1024, "LDS");

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct Tile { int b, d0, h0, w0; };
template <int V> struct Int { static constexpr int value = V; };

__device__ __forceinline__ int parity3(int v) { return __builtin_popcount(v & 7) & 1; }

__global__ __launch_bounds__(256) void conv3d_wres_kernel(Conv3dArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef WRES_STAMPS      // diagnostic builds only (tools/abl_build.sh s*): shader-clock timeline of the workgroup
    const long long t_begin = __builtin_readcyclecounter(), r_begin = wall_clock64();
    long long kcyc = 0;
    float tl[12] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#define WR_TL(i) tl[i] = (float)(__builtin_readcyclecounter() - t_begin);
#define WR_T0 const long long t0_ = __builtin_readcyclecounter();
#define WR_T1 kcyc += __builtin_readcyclecounter() - t0_;
#else
#define WR_TL(i)
#define WR_T0
#define WR_T1
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform: tile addressing stays in SGPRs
    const int lr = lane & 31, lh = lane >> 5;
    // ---- weights: global -> LDS by LDS-DMA (no registers, no ds_write, lands while the wave does something else),
    // issued before anything else.  LDS image [tap][rho][slot] x 16 B; a DMA writes wave-uniform base + 16 * lane,
    // so wave w, instruction k of a kd plane p covers tap 9 p + k, rows rho = 16 w + (lane >> 2), slot = lane & 3:
    // the permutation (channel n = 2 (rho & 31) + (rho >> 5), channel segment = slot ^ key(rho)) sits in the SOURCE
    // address, which is affine in the tap (+64 bytes).
    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;
    const bf16* wlane;
    {
        const int rho = wave * 16 + (lane >> 2), slot = lane & 3;
        const int n = 2 * (rho & 31) + (rho >> 5);
        wlane = a.w + (size_t)n * 27 * CIN + ((slot ^ ((rho >> 2) & 3)) << 3);
    }
    auto dma_plane = [&](int p) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < 9; ++k)
            __builtin_amdgcn_global_load_lds((gptr_t*)(wlane + (9 * p + k) * CIN), (lptr_t*)(smem + (9 * p + k) * BN * ROWB + wave * 1024),
                                             16, 0, 0);
    };
    dma_plane(0);
    dma_plane(1);
    dma_plane(2);
    const int tw = (a.W + 7) / 8, th = (a.H + 7) / 8, td = (a.D + TD - 1) / TD;
    const int ntiles = a.B * td * th * tw;

    // ---- XCD-aware tile list: XCD x (= blockIdx % 8 under round-robin dispatch) owns tiles [lo, hi)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int nper = (gridDim.x - xcd + 7) >> 3;                    // workgroups of this XCD
    const int lo = (int)((long)ntiles * xcd / 8), hi = (int)((long)ntiles * (xcd + 1) / 8);
    int tile = lo + slot;
    if (tile >= hi) return;                                        // uniform: nothing to do

    // tile coordinates advance incrementally (one runtime division chain per workgroup, not per tile)
    auto coords = [&](int t) __attribute__((always_inline)) {
        Tile c;
        c.w0 = (t % tw) * 8; t /= tw;
        c.h0 = (t % th) * 8; t /= th;
        c.d0 = (t % td) * TD; t /= td;
        c.b = t;
        return c;
    };
    const Tile stepT = coords(nper);                                // nper decomposed in the same mixed radix
    auto advance = [&](Tile c) __attribute__((always_inline)) {
        c.w0 += stepT.w0; if (c.w0 >= tw * 8) { c.w0 -= tw * 8; c.h0 += 8; }
        c.h0 += stepT.h0; if (c.h0 >= th * 8) { c.h0 -= th * 8; c.d0 += TD; }
        c.d0 += stepT.d0; if (c.d0 >= td * TD) { c.d0 -= td * TD; c.b += 1; }
        c.b += stepT.b;
        return c;
    };
    // ---- the halo as 16-byte chunks: chunk s = tid + 256 i is halo row r = s >> 2 = (hd, hh, hw), channel
    // segment s & 3.  Everything that does not depend on the tile is computed once: the global offset relative to
    // the tile's first voxel, the swizzled LDS address, and a one-hot selector (1 << hd | 1 << (6 + hh) |
    // 1 << (16 + hw)) that is tested against the tile's in-volume mask - two VALU instructions per chunk and tile
    // instead of ~40 (div / mod by 10 and six compares; PMC showed ~800 VALU per tile boundary).
    int goff[HREGS], ldso[HREGS];
    unsigned sel[HREGS];
    {   // row r = (tid >> 2) + 64 i: one small division for i = 0, then +64 rows = +6 h-rows +4 w with carries
        const int q = tid >> 2, sg = tid & 3;
        int hh = (q * 205) >> 11, hw = q - hh * HB, hd = 0;          // q < 64: exact
#pragma unroll
        for (int i = 0; i < HREGS; ++i) {
            goff[i] = ((((hd - 1) * a.H + (hh - 1)) * a.W + (hw - 1)) * CIN + sg * 8) * 2;       // bytes
            ldso[i] = H_OFF + ((hd * HB + hh) * WP + hw) * ROWB + ((sg ^ (hh & 3)) << 4);
            sel[i] = (i < HREGS - 1 || q + 64 * i < HROWS) ? (1u << hd) | (1u << (6 + hh)) | (1u << (16 + hw)) : 0x80000000u;
            hw += 4; hh += 6;
            if (hw >= HB) { hw -= HB; hh += 1; }
            if (hh >= HB) { hh -= HB; hd += 1; }
        }
    }
    auto range_mask = [](int lo_, int hi_, int n) __attribute__((always_inline)) {      // bits [max(lo,0), min(hi,n))
        lo_ = lo_ < 0 ? 0 : lo_;
        hi_ = hi_ > n ? n : hi_;
        return hi_ > lo_ ? ((1u << hi_) - 1u) & ~((1u << lo_) - 1u) : 0u;
    };
    // buffer loads: a chunk outside the volume gets an out-of-range offset and the hardware returns zeros - no
    // exec masking, no zero-initialised registers (the whole input is < 4 GiB: checked on the host)
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16*>(a.x), 0, (unsigned)((size_t)a.B * a.D * a.H * a.W * CIN * 2), 0x00020000);
    auto load_halo = [&](const Tile& c, u32x4 (&regs)[HREGS]) __attribute__((always_inline)) {
        const unsigned toff = (unsigned)((((c.b * a.D + c.d0) * a.H + c.h0) * a.W + c.w0) * CIN * 2);   // uniform, bytes
        // halo index k is inside the volume iff 0 <= c0 + k - 1 < extent
        const unsigned M = range_mask(1 - c.d0, a.D - c.d0 + 1, TD + 2) | (range_mask(1 - c.h0, a.H - c.h0 + 1, HB) << 6) |
                           (range_mask(1 - c.w0, a.W - c.w0 + 1, HB) << 16);
#pragma unroll
        for (int i = 0; i < HREGS; ++i) {
            const unsigned off = ((M & sel[i]) == sel[i]) ? toff + (unsigned)goff[i] : 0xFFFFFFF0u;
            regs[i] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, off, 0, 0);
        }
    };
    auto store_halo = [&](const u32x4 (&regs)[HREGS]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < HREGS; ++i)
            if (i < HREGS - 1 || (int)sel[i] >= 0) *reinterpret_cast<u32x4*>(smem + ldso[i]) = regs[i];
    };

    // ---- per-lane fragment bases (bytes into smem).  GEMM row m = wave*64 + i*32 + lr  <->  voxel
    // (d, h, w) = (wave, 4 i + (lr >> 3), wl); tap (kd, kh, kw) adds (kd*120 + kh*12 + kw) rows.
    const int hl = lr >> 3;
    const int wl = (lr & 3) + 4 * parity3(lr >> 2);
    // the hand-written streams address LDS absolutely: byte offset of the dynamic segment (0 with no static LDS)
    const int lds0 = (int)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int abase = lds0 + ((wave * HB + hl) * WP + wl) * ROWB + H_OFF;
    int aoff[3][2], boff[2][2][2];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) aoff[kh][ks] = abase + (((2 * ks + lh) ^ ((hl + kh) & 3)) << 4);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int rho = j * 32 + lr;                           // LDS weight row of (column tile j, lane column lr)
            boff[j][ks][0] = lds0 + rho * ROWB + (((2 * ks + lh) ^ ((rho >> 2) & 3)) << 4);
            boff[j][ks][1] = boff[j][ks][0] + 13 * BN * ROWB;       // taps 13..26: ds offsets are 16-bit
        }
    // output channel of (j, lr) is 2 lr + j: the weight rows are de-interleaved on their way into LDS
    const float sh0 = a.shift ? a.shift[2 * lr] : 0.f, sh1 = a.shift ? a.shift[2 * lr + 1] : 0.f;

    u32x4 nxt[HREGS];
    Tile curT = coords(tile);                                        // the tile whose halo sits in `nxt`
    WR_TL(0)
    load_halo(curT, nxt);
    WR_TL(1)

    // ---- the K loop and the previous tile's epilogue are hand-scheduled instruction streams
    // (tools/gen_wres_asm.py -> conv3d_wres_asm.inc): accumulator sets X = a[48:111], Y = a[112:175] and the
    // rotating fragment sets a[0:47] live in the accumulator file and are only ever touched by these statements.
    Tile pt = {0, 0, 0, 0};
    const unsigned pitch_b = __builtin_amdgcn_readfirstlane((unsigned)a.W * BN * 2);   // one h step of the output, in bytes
    // accumulator register r of a lane holds GEMM row (r & 3) + 8 (r >> 2) + 4 lh, i.e. voxel
    // (h, w) = (r >> 2, (r & 3) + 4 (parity(r >> 2) ^ lh)): odd-parity registers swap the halves
    // (byte offset of the even-parity rows: lh * 512 + 4 lr; the odd-parity ones are that ^ 512)
    const unsigned voff_e0 = (4 * lh * BN + 2 * lr) * 2;
    float s10 = 0.f, s11 = 0.f, s20 = 0.f, s21 = 0.f;                // BatchNorm sums of channels 2 lr (.0) and 2 lr + 1 (.1)

#define WR_K_OPERANDS                                                                                              \
    [ab0] "v"(aoff[0][0]), [ab1] "v"(aoff[0][1]), [ab2] "v"(aoff[1][0]), [ab3] "v"(aoff[1][1]), [ab4] "v"(aoff[2][0]),  \
    [ab5] "v"(aoff[2][1]), [bb0] "v"(boff[0][0][0]), [bb1] "v"(boff[0][0][1]), [bb2] "v"(boff[0][1][0]),              \
    [bb3] "v"(boff[0][1][1]), [bb4] "v"(boff[1][0][0]), [bb5] "v"(boff[1][0][1]), [bb6] "v"(boff[1][1][0]),           \
    [bb7] "v"(boff[1][1][1])
#define WR_EPI_OUT                                                                                                  \
    [s10] "=&v"(p10), [s11] "=&v"(p11), [s20] "=&v"(p20), [s21] "=&v"(p21), [voffe] "=&v"(ve), [voffo] "=&v"(vo),       \
    [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2)
#define WR_EPI_IN [sh0] "v"(sh0), [sh1] "v"(sh1), [pbase] "s"(pbase), [pitch] "s"(pitch_b), [voff0] "v"(voff_e0)
    auto tile_base = [&](const Tile& c) __attribute__((always_inline)) {                           // wave-uniform: (b, d0 + wave, h0, w0, 0)
        bf16* p = a.out_bf16 + ((((size_t)c.b * a.D + c.d0 + wave) * a.H + c.h0) * a.W + c.w0) * BN;
        // provably uniform for the "s" operand of the hand-written stores (cdna_hip_programming.md T20)
        const unsigned long long u = reinterpret_cast<unsigned long long>(p);
        const unsigned lo_ = __builtin_amdgcn_readfirstlane((unsigned)u), hi_ = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
        return reinterpret_cast<bf16*>(((unsigned long long)hi_ << 32) | lo_);
    };
    // K loop of one tile into set CUR; EPI: the pending interior tile in the other set is packed, stored and
    // summed in the shadow of the first 32 K-steps
    auto k_plain = [&](int cur) __attribute__((always_inline)) {
        if (cur == 0) asm volatile(WRES_K_X_ALL : : WR_K_OPERANDS : "memory", WRES_AGPR_CLOBBERS);
        else asm volatile(WRES_K_Y_ALL : : WR_K_OPERANDS : "memory", WRES_AGPR_CLOBBERS);
    };
    auto k_epi = [&](int cur, const bf16* pbase) __attribute__((always_inline)) {
        unsigned ve, vo, t2;
        float t0, t1, p10, p11, p20, p21;                            // this tile's BatchNorm partial sums
        if (cur == 0)
            asm volatile(WRES_K_X_ALL_EPI
                         : WR_EPI_OUT
                         : WR_K_OPERANDS, WR_EPI_IN
                         : "memory", WRES_AGPR_CLOBBERS);
        else
            asm volatile(WRES_K_Y_ALL_EPI
                         : WR_EPI_OUT
                         : WR_K_OPERANDS, WR_EPI_IN
                         : "memory", WRES_AGPR_CLOBBERS);
        s10 += p10; s11 += p11; s20 += p20; s21 += p21;
    };
    auto flush = [&](int set, const bf16* pbase) __attribute__((always_inline)) {                  // the last tile's stores have nothing to hide behind
        unsigned ve, vo, t2;
        float t0, t1, p10, p11, p20, p21;
        if (set == 0)
            asm volatile(WRES_FLUSH_X
                         : WR_EPI_OUT
                         : WR_EPI_IN
                         : "memory", WRES_AGPR_CLOBBERS);
        else
            asm volatile(WRES_FLUSH_Y
                         : WR_EPI_OUT
                         : WR_EPI_IN
                         : "memory", WRES_AGPR_CLOBBERS);
        s10 += p10; s11 += p11; s20 += p20; s21 += p21;
    };
    // a ragged tile (volume edge) leaves the accumulator file through sixteen "=v" operands per MFMA tile and is
    // stored by ordinary code with per-voxel predicates, right after its K loop (exposed; edge tiles only)
#define WR_X16(v) "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]), "=v"(v[4]), "=v"(v[5]), "=v"(v[6]), "=v"(v[7]), "=v"(v[8]), \
                  "=v"(v[9]), "=v"(v[10]), "=v"(v[11]), "=v"(v[12]), "=v"(v[13]), "=v"(v[14]), "=v"(v[15])
    auto store_ragged = [&](int set, const Tile& c) __attribute__((always_inline)) {
        float v[4][16];
        if (set == 0) {
            asm volatile(WRES_EXTRACT_X_0 : WR_X16(v[0]) : : WRES_AGPR_CLOBBERS);
            asm volatile(WRES_EXTRACT_X_1 : WR_X16(v[1]) : : WRES_AGPR_CLOBBERS);
            asm volatile(WRES_EXTRACT_X_2 : WR_X16(v[2]) : : WRES_AGPR_CLOBBERS);
            asm volatile(WRES_EXTRACT_X_3 : WR_X16(v[3]) : : WRES_AGPR_CLOBBERS);
        } else {
            asm volatile(WRES_EXTRACT_Y_0 : WR_X16(v[0]) : : WRES_AGPR_CLOBBERS);
            asm volatile(WRES_EXTRACT_Y_1 : WR_X16(v[1]) : : WRES_AGPR_CLOBBERS);
            asm volatile(WRES_EXTRACT_Y_2 : WR_X16(v[2]) : : WRES_AGPR_CLOBBERS);
            asm volatile(WRES_EXTRACT_Y_3 : WR_X16(v[3]) : : WRES_AGPR_CLOBBERS);
        }
        bf16* pbase = tile_base(c);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int hh = i * 4 + (r >> 2), par = __builtin_popcount(r >> 2) & 1;
                const int ww = (r & 3) + 4 * (par ^ lh);
                if ((c.d0 + wave < a.D) && (c.h0 + hh < a.H) && (c.w0 + ww < a.W)) {
                    const float v0 = v[2 * i][r] + sh0, v1 = v[2 * i + 1][r] + sh1;
                    bf16x2 pk = {(bf16)v0, (bf16)v1};
                    *reinterpret_cast<bf16x2*>(pbase + ((size_t)hh * a.W + ww) * BN + 2 * lr) = pk;
                    s10 += v0; s20 += v0 * v0;
                    s11 += v1; s21 += v1 * v1;
                }
            }
    };
    auto is_full = [&](const Tile& c) __attribute__((always_inline)) { return c.d0 + TD <= a.D && c.h0 + 8 <= a.H && c.w0 + 8 <= a.W; };
    auto prefetch_next = [&](int t) __attribute__((always_inline)) {  // next halo -> registers, in flight during the MFMAs
        curT = advance(curT);
        if (t + nper < hi) load_halo(curT, nxt);
    };
    auto begin_tile = [&]() __attribute__((always_inline)) {                                        // halo -> LDS
        __syncthreads();                                            // previous tile's LDS reads are done
        store_halo(nxt);
        __syncthreads();
    };
    bool pending = false;                                           // an interior tile waits in the set not being computed
    // one tile into set CUR (0 = X, 1 = Y); the pending tile, if any, is in the other set
    bool first_boundary = true;
    auto next_tile = [&](int cur, int t) __attribute__((always_inline)) {
        if (first_boundary) { WR_TL(7) }
        begin_tile();
        if (first_boundary) { WR_TL(8) }
        const Tile me = curT;
        prefetch_next(t);
        if (first_boundary) { WR_TL(9) }
        first_boundary = false;
        WR_T0
        if (pending) k_epi(cur, tile_base(pt));
        else k_plain(cur);
        WR_T1
        pt = me;
        pending = is_full(pt);
        if (!pending) store_ragged(cur, pt);
    };

    // ---- first tile: its halo and the weights were the first things this workgroup asked for
    begin_tile();                                                   // (the barriers also cover the weight DMA: vmcnt(0) first)
    WR_TL(3)
    pt = curT;
    prefetch_next(tile);
    WR_TL(5)
    k_plain(0);
    WR_TL(6)
    pending = is_full(pt);
    if (!pending) store_ragged(0, pt);
    bool last_in_x = true;
    for (tile += nper; tile < hi; tile += 2 * nper) {
        next_tile(1, tile);
        last_in_x = false;
        const int t2 = tile + nper;
        if (t2 >= hi) break;
        next_tile(0, t2);
        last_in_x = true;
    }
    if (pending) {
        if (last_in_x) flush(0, tile_base(pt));
        else flush(1, tile_base(pt));
    }
#ifdef WRES_STAMPS
    if (a.stats && tid == 0) {          // [32][2][64] statistics, then per workgroup {K-loop cycles, kernel cycles, kernel 100 MHz ticks, tiles, 12 timeline stamps}
        float* o = a.stats + MM_REPL * 2 * BN + blockIdx.x * 16;
        o[0] = (float)kcyc; o[1] = (float)(__builtin_readcyclecounter() - t_begin);
        o[2] = (float)(wall_clock64() - r_begin); o[3] = (float)((hi - lo - slot + nper - 1) / nper);
#pragma unroll
        for (int i = 0; i < 12; ++i) o[4 + i] = tl[i];
    }
#endif
    float st1[2] = {s10, s11}, st2[2] = {s20, s21};
    if (a.stats) {
        float* sstat = reinterpret_cast<float*>(smem + S_OFF);
        // lanes l and l+32 hold the same two channels (rows differ)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            st1[j] += __shfl_xor(st1[j], 32);
            st2[j] += __shfl_xor(st2[j], 32);
        }
        if (lh == 0) {                                   // every wave parks its channel sums: no LDS atomics
            float* mine = sstat + wave * 2 * BN;
            mine[2 * lr] = st1[0];
            mine[2 * lr + 1] = st1[1];
            mine[BN + 2 * lr] = st2[0];
            mine[BN + 2 * lr + 1] = st2[1];
        }
        __syncthreads();
        float* rep = a.stats + (size_t)(blockIdx.x % MM_REPL) * 2 * BN;
        if (tid < 2 * BN)
            atomicAdd(&rep[tid], (sstat[tid] + sstat[2 * BN + tid]) + (sstat[4 * BN + tid] + sstat[6 * BN + tid]));
    }
#undef WR_K_OPERANDS
#undef WR_EPI_OUT
#undef WR_EPI_IN
#undef WR_X16
}

}  // namespace

bool conv3d_wres_applies(const Conv3dArgs& a) {
    const long tiles = (long)a.B * ceil_div(a.D, TD) * ceil_div(a.H, 8) * ceil_div(a.W, 8);
    const size_t in_bytes = (size_t)a.B * a.D * a.H * a.W * CIN * 2;   // halo chunks are fetched with 32-bit buffer offsets
    return a.Cin == CIN && a.Cout == BN && a.out_bf16 && !a.out_f32 && tiles >= 64 && in_bytes < 0xFFFF0000ull;
}

int launch3d_wres(const Conv3dArgs& a, hipStream_t st) {
    auto kern = conv3d_wres_kernel;
    static std::once_flag once;                                     // the only process-wide state: an immutable kernel attribute
    std::call_once(once, [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    });
    const int ntiles = a.B * ceil_div(a.D, TD) * ceil_div(a.H, 8) * ceil_div(a.W, 8);
    const int grid = ntiles < 256 ? (ntiles & ~7) : 256;            // a multiple of 8: every XCD list has its workgroups
    if (grid < 8) return mm_fail(MM_ERR_UNSUPPORTED, "conv3d_fwd_wres: %d tiles", ntiles);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS_BYTES, st, a);
    return mm_check_launch("conv3d_fwd_wres");
}
