// Weight-resident persistent 3-D convolution (k = 3, pad 1) for Cin = 32, Cout = 64 - layer 2 of
// the voxel encoder, the kernel bench.py's roofline line is about.  bf16 MFMA, fp32 accumulate,
// channels-last (NDHWC) bf16 in, channels-last bf16 out + per-channel BatchNorm sums.
//
//   Y[b, v, n] = bias[n] + sum_{tap, c} X[b, v + off(tap), c] * W[n, tap, c]      (implicit GEMM
//   M = B*D*H*W rows, N = 64, K = 27 * 32 = 864; 2*M*N*K FLOPs, algorithmic bytes = 64 M in
//   + 128 M out + 108 KiB of weights)
//
// One workgroup per CU keeps all 64 x 27 x 32 bf16 weights (108 KiB) in LDS for its lifetime and
// walks 4 x 8 x 8 output tiles (256 GEMM rows; wave w owns depth slice w = 64 rows x 64 columns).
// The (6 x 10 x 10)-voxel input halo of a tile is staged once into LDS; every tap reads its A fragments
// from it at a row offset, so the im2col matrix only ever exists as LDS addresses.
//
// Design points (measurements: profiles/r02_*):
//  * MFMA 16x16x32 (one tap = one K-step of all 32 input channels; 4 x 4 MFMA tiles per wave).  Same FLOP
//    per pipe cycle as 32x32x16, but the chip holds a ~14 % higher clock on it under this kernel's load.
//    MFMA row m of M-tile i <-> voxel (h, w) = (4 (i >> 1) + (m >> 2), 4 (i & 1) + (m & 3)): 4 x 4 patches.
//  * the per-tile instruction streams are hand-scheduled (tools/gen_wres_asm.py -> conv3d_wres_asm.inc): the
//    K loop with the next tap's fragment reads, the previous tile's pack / store / BatchNorm-sum and the next
//    tile's halo prefetch in the MFMA gaps, and the tile boundary (halo registers -> LDS between two barriers).
//    Accumulators, fragments, the prefetched halo and the per-lane halo constants live in the accumulator
//    file (a0-a251), which only those statements touch (tests/test_abi_and_host.py audits the compiled ISA).
//  * LDS rows are unpadded 64-byte rows whose 16-byte slots are XOR-swizzled with 2 * (row-of-patch parity):
//    slot = segment ^ (2 * ((h + kh) & 1)) for the halo (h = the lane's patch row), segment ^ (2 * ((rho >> 2)
//    & 1)) for the weights.  Each ds_read_b128 lane group then touches 16 distinct 16-byte slots for every tap
//    shift (brute-forced over the lane groups {0-3,12-15,20-27}, ...), and the key is separable from the tap
//    offset: two A and two B per-lane base registers + immediate offsets address every fragment read.
//  * output: channel n of column tile j, lane column c is 4 c + j, so a lane holds four consecutive channels
//    of a voxel: one global_store_dwordx2 per MFMA-tile row, a wave instruction = four 128-byte voxel rows;
//    write-through (sc0 sc1) so that the kernel boundary does not pay a 16.8 MB L2 write-back burst.
//  * weights arrive by LDS-DMA (global_load_lds_dwordx4: no registers, no ds_write), issued first.
//  * tile -> workgroup map is XCD-aware: workgroups b and b + 8 share an XCD (round-robin dispatch), so XCD x
//    takes the x-th eighth of the (b, d, h, w)-ordered tile list and its workgroups walk it interleaved:
//    neighbouring tiles' halos are fetched into that XCD's L2 once (PMC: 1.06x the algorithmic bytes; round 1:
//    2.0x).  Placement is a speed assumption only; any placement computes the same result.
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

struct Tile { int b, d0, h0, w0; };

__global__ __launch_bounds__(256) void conv3d_wres_kernel(Conv3dArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef WRES_STAMPS      // diagnostic builds only (tools/abl_build.sh s*): shader-clock timeline of the workgroup
    const long long t_begin = __builtin_readcyclecounter(), r_begin = wall_clock64();
    long long kcyc = 0;
    float tl[12] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#define WR_TL(i) tl[i] = (float)(__builtin_readcyclecounter() - t_begin);
#define WR_T0 const long long t0_ = __builtin_readcyclecounter();
#define WR_T1 kcyc += __builtin_readcyclecounter() - t0_;
#define WR_TLK if (tl[4] == 0.f) { WR_TL(4) } else if (tl[5] == 0.f) { WR_TL(5) }
#else
#define WR_TLK
#define WR_TL(i)
#define WR_T0
#define WR_T1
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform: tile addressing stays in SGPRs
    const int lc = lane & 15, lg = lane >> 4;                       // MFMA lane = (row / column 0-15, group 0-3)
    // ---- weights: global -> LDS by LDS-DMA, issued before anything else.  LDS image [tap][rho][slot] x 16 B; a
    // DMA writes wave-uniform base + 16 * lane, so wave w, instruction k covers tap k, rows rho = 16 w + (lane >> 2),
    // slot = lane & 3.  LDS row rho holds output channel n = 4 (rho & 15) + (rho >> 4) (column tile j = rho >> 4,
    // lane column c = rho & 15 <-> channel 4 c + j); the channel segment in slot s is s ^ 2 ((rho >> 2) & 1).  Both
    // permutations sit in the SOURCE address, which is affine in the tap (+64 bytes).
    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;
    const bf16* wlane;
    {
        const int rho = wave * 16 + (lane >> 2), slot = lane & 3;
        const int n = 4 * (rho & 15) + (rho >> 4);
        wlane = a.w + (size_t)n * 27 * CIN + ((slot ^ (2 * ((rho >> 2) & 1))) << 3);
    }
    // three batches of nine, interleaved with the set-up arithmetic below: a DMA is bandwidth-bound (~64 B/clk per
    // CU), the wave that issues 27 in a row just stalls on the full queue for ~1 700 cycles
    auto dma_batch = [&](int p) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 9 * p; k < 9 * p + 9; ++k)
            __builtin_amdgcn_global_load_lds((gptr_t*)(wlane + k * CIN), (lptr_t*)(smem + k * BN * ROWB + wave * 1024), 16, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    dma_batch(0);
    WR_TL(1)
    const int tw = (a.W + 7) / 8, th = (a.H + 7) / 8, td = (a.D + TD - 1) / TD;
    const int ntiles = a.B * td * th * tw;

    // ---- XCD-aware tile list: XCD x (= blockIdx % 8 under round-robin dispatch) owns tiles [lo, hi)
    const int xcd = blockIdx.x & 7, slot_ = blockIdx.x >> 3;
    const int nper = (gridDim.x - xcd + 7) >> 3;                    // workgroups of this XCD
    const int lo = (int)((long)ntiles * xcd / 8), hi = (int)((long)ntiles * (xcd + 1) / 8);
    int tile = lo + slot_;
    const bool has_work = tile < hi;                                // (uniform) a workgroup without tiles still drains its DMA

    // tile coordinates advance incrementally (one runtime division chain per workgroup, not per tile)
    auto coords = [&](int t) __attribute__((always_inline)) {       // divisions by host-made reciprocals (t < 2^24)
        Tile c;
        int q = (int)__umulhi((unsigned)t, a.mtw); c.w0 = (t - q * tw) * 8; t = q;
        q = (int)__umulhi((unsigned)t, a.mth); c.h0 = (t - q * th) * 8; t = q;
        q = (int)__umulhi((unsigned)t, a.mtd); c.d0 = (t - q * td) * TD;
        c.b = q;
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
    // ---- the halo as 16-byte chunks: chunk q of a thread is halo row r = (tid >> 2) + 64 q = (hd, hh, hw), channel
    // segment tid & 3.  What does not depend on the tile is computed once and parked in the accumulator file: the
    // global byte offset relative to the tile's first voxel and a one-hot selector (1 << hd | 1 << (6 + hh) |
    // 1 << (16 + hw)) that is tested against the tile's in-volume mask.  The swizzled LDS addresses stay in VGPRs.
    int ldso[HREGS];
    {
        int goff[HREGS];
        unsigned sel[HREGS];
        const int q0 = tid >> 2, sg = tid & 3;
        int hh = (q0 * 205) >> 11, hw = q0 - hh * HB, hd = 0;        // q0 < 64: exact division by 10
#pragma unroll
        for (int i = 0; i < HREGS; ++i) {
            const bool real = i < HREGS - 1 || q0 + 64 * i < HROWS;
            goff[i] = ((((hd - 1) * a.H + (hh - 1)) * a.W + (hw - 1)) * CIN + sg * 8) * 2;       // bytes
            // rows past the halo (the last chunk of threads 96-255) park in an unused pitch column of row 0
            ldso[i] = H_OFF + (real ? ((hd * HB + hh) * WP + hw) * ROWB + ((sg ^ (2 * (hh & 1))) << 4) : (HB + (q0 & 1)) * ROWB + (sg << 4));
            sel[i] = real ? (1u << hd) | (1u << (6 + hh)) | (1u << (16 + hw)) : 0x80000000u;
            hw += 4; hh += 6;
            if (hw >= HB) { hw -= HB; hh += 1; }
            if (hh >= HB) { hh -= HB; hd += 1; }
        }
        asm volatile(WRES_INIT : : [g0] "v"(goff[0]), [g1] "v"(goff[1]), [g2] "v"(goff[2]), [g3] "v"(goff[3]), [g4] "v"(goff[4]),
                     [g5] "v"(goff[5]), [g6] "v"(goff[6]), [g7] "v"(goff[7]), [g8] "v"(goff[8]), [g9] "v"(goff[9]),
                     [s0] "v"(sel[0]), [s1] "v"(sel[1]), [s2] "v"(sel[2]), [s3] "v"(sel[3]), [s4] "v"(sel[4]), [s5] "v"(sel[5]),
                     [s6] "v"(sel[6]), [s7] "v"(sel[7]), [s8] "v"(sel[8]), [s9] "v"(sel[9]) : WRES_CLOBBERS);
    }
    WR_TL(2)
    dma_batch(1);
    auto range_mask = [](int lo_, int hi_, int n) __attribute__((always_inline)) {      // bits [max(lo,0), min(hi,n))
        lo_ = lo_ < 0 ? 0 : lo_;
        hi_ = hi_ > n ? n : hi_;
        return hi_ > lo_ ? ((1u << hi_) - 1u) & ~((1u << lo_) - 1u) : 0u;
    };
    // buffer loads: a chunk outside the volume gets an out-of-range offset and the hardware returns zeros
    // (the whole input is < 4 GiB: checked on the host)
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16*>(a.x), 0, (unsigned)((size_t)a.B * a.D * a.H * a.W * CIN * 2), 0x00020000);
    struct Pf { unsigned toff, mask; };
    auto pf_args = [&](const Tile& c, bool valid) __attribute__((always_inline)) {      // uniform
        Pf p;
        p.toff = (unsigned)((((c.b * a.D + c.d0) * a.H + c.h0) * a.W + c.w0) * CIN * 2);
        // halo index k is inside the volume iff 0 <= c0 + k - 1 < extent; no next tile: nothing is in-volume
        p.mask = valid ? range_mask(1 - c.d0, a.D - c.d0 + 1, TD + 2) | (range_mask(1 - c.h0, a.H - c.h0 + 1, HB) << 6) |
                             (range_mask(1 - c.w0, a.W - c.w0 + 1, HB) << 16) : 0u;
        p.toff = __builtin_amdgcn_readfirstlane(p.toff);
        p.mask = __builtin_amdgcn_readfirstlane(p.mask);
        return p;
    };

    Tile curT = coords(has_work ? tile : 0);                         // the tile whose halo is in flight / in the halo registers
    {   // first halo: everything else of the set-up overlaps its latency and the weight DMA
        const Pf pf = pf_args(curT, has_work);
        asm volatile(WRES_PREFETCH : : [vmask] "s"(pf.mask), [toff] "s"(pf.toff), [rsrc] "s"(xrsrc) : "memory", WRES_CLOBBERS);
    }
    WR_TL(0)
    dma_batch(2);

    // ---- per-lane fragment bases (absolute LDS byte addresses).  Lane (lc, lg) of MFMA tile i reads the halo row
    // of voxel (d, h, w) = (wave, 4 (i >> 1) + (lc >> 2), 4 (i & 1) + (lc & 3)) shifted by the tap, channel segment lg;
    // of column tile j the weight row rho = 16 j + lc.
    const int lds0 = (int)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int arow = ((wave * HB + (lc >> 2)) * WP + (lc & 3)) * ROWB + H_OFF + lds0;
    const int ab0 = arow + ((lg ^ (2 * (((lc >> 2) + 0) & 1))) << 4);        // taps with kh even
    const int ab1 = arow + ((lg ^ (2 * (((lc >> 2) + 1) & 1))) << 4);        // kh odd
    const int bb0 = lds0 + lc * ROWB + ((lg ^ (2 * ((lc >> 2) & 1))) << 4);
    const int bb1 = bb0 + 13 * BN * ROWB;                                     // taps 13..26: ds offsets are 16-bit
    // a lane's four channels: 4 lc + j
    float sh[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) sh[j] = a.shift ? a.shift[4 * lc + j] : 0.f;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};       // BatchNorm sums of those channels
    const unsigned pitch_b = (unsigned)a.W * BN * 2;                 // one h step of the output, in bytes
    const unsigned pitch4 = __builtin_amdgcn_readfirstlane(4 * pitch_b);
    // accumulator register r of lane (lc, lg) in MFMA tile (i, j) is voxel (h, w) = (4 (i >> 1) + lg, 4 (i & 1) + r)
    const unsigned voff0 = lg * pitch_b + lc * 8;

    auto tile_base = [&](const Tile& c) __attribute__((always_inline)) {                           // wave-uniform: (b, d0 + wave, h0, w0, 0)
        bf16* p = a.out_bf16 + ((((size_t)c.b * a.D + c.d0 + wave) * a.H + c.h0) * a.W + c.w0) * BN;
        // provably uniform for the "s" operand of the hand-written stores (cdna_hip_programming.md T20)
        const unsigned long long u = reinterpret_cast<unsigned long long>(p);
        const unsigned lo_ = __builtin_amdgcn_readfirstlane((unsigned)u), hi_ = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
        return reinterpret_cast<bf16*>(((unsigned long long)hi_ << 32) | lo_);
    };
#define WR_BASES [ab0] "v"(ab0), [ab1] "v"(ab1), [bb0] "v"(bb0), [bb1] "v"(bb1)
#define WR_K_IN WR_BASES, [vmask] "s"(pf.mask), [toff] "s"(pf.toff), [rsrc] "s"(xrsrc)
#define WR_EPI_OUT [s10] "=&v"(p1[0]), [s11] "=&v"(p1[1]), [s12] "=&v"(p1[2]), [s13] "=&v"(p1[3]), [s20] "=&v"(p2[0]),      \
                   [s21] "=&v"(p2[1]), [s22] "=&v"(p2[2]), [s23] "=&v"(p2[3])
#define WR_EPI_IN [sh0] "v"(sh[0]), [sh1] "v"(sh[1]), [sh2] "v"(sh[2]), [sh3] "v"(sh[3]), [pbase] "s"(pbase), [pitch4] "s"(pitch4), \
                  [voff0] "v"(voff0)
    // K loop of one tile into set CUR (0 = X, 1 = Y) with the halo prefetch described by pf in its last taps
    auto k_plain = [&](int cur, const Pf& pf) __attribute__((always_inline)) {
        if (cur == 0) asm volatile(WRES_K_X : : WR_K_IN : "memory", WRES_CLOBBERS);
        else asm volatile(WRES_K_Y : : WR_K_IN : "memory", WRES_CLOBBERS);
    };
    // ... and the pending interior tile in the other set packed, stored and summed in the first 16 taps
    auto k_epi = [&](int cur, const Pf& pf, const bf16* pbase) __attribute__((always_inline)) {
        float p1[4], p2[4];                                          // this tile's BatchNorm partial sums
        if (cur == 0) asm volatile(WRES_K_X_EPI : WR_EPI_OUT : WR_K_IN, WR_EPI_IN : "memory", WRES_CLOBBERS);
        else asm volatile(WRES_K_Y_EPI : WR_EPI_OUT : WR_K_IN, WR_EPI_IN : "memory", WRES_CLOBBERS);
#pragma unroll
        for (int j = 0; j < 4; ++j) { s1[j] += p1[j]; s2[j] += p2[j]; }
    };
    auto flush = [&](int set, const bf16* pbase) __attribute__((always_inline)) {                  // the last tile's stores have nothing to hide behind
        float p1[4], p2[4];
        if (set == 0) asm volatile(WRES_FLUSH_X : WR_EPI_OUT : WR_EPI_IN : "memory", WRES_CLOBBERS, WRES_FLUSH_CLOBBERS);
        else asm volatile(WRES_FLUSH_Y : WR_EPI_OUT : WR_EPI_IN : "memory", WRES_CLOBBERS, WRES_FLUSH_CLOBBERS);
#pragma unroll
        for (int j = 0; j < 4; ++j) { s1[j] += p1[j]; s2[j] += p2[j]; }
    };
    // a ragged tile (volume edge) leaves the accumulator file through sixteen "=v" operands per MFMA-tile row and is
    // stored by ordinary code with per-voxel predicates, right after its K loop (exposed; edge tiles only)
#define WR_X16(v) "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]), "=v"(v[4]), "=v"(v[5]), "=v"(v[6]), "=v"(v[7]), "=v"(v[8]), \
                  "=v"(v[9]), "=v"(v[10]), "=v"(v[11]), "=v"(v[12]), "=v"(v[13]), "=v"(v[14]), "=v"(v[15])
    auto store_ragged = [&](int set, const Tile& c) __attribute__((always_inline)) {
        float v[4][16];                                              // [i][4 j + r]
        if (set == 0) {
            asm volatile(WRES_EXTRACT_X_0 : WR_X16(v[0]) : : WRES_CLOBBERS);
            asm volatile(WRES_EXTRACT_X_1 : WR_X16(v[1]) : : WRES_CLOBBERS);
            asm volatile(WRES_EXTRACT_X_2 : WR_X16(v[2]) : : WRES_CLOBBERS);
            asm volatile(WRES_EXTRACT_X_3 : WR_X16(v[3]) : : WRES_CLOBBERS);
        } else {
            asm volatile(WRES_EXTRACT_Y_0 : WR_X16(v[0]) : : WRES_CLOBBERS);
            asm volatile(WRES_EXTRACT_Y_1 : WR_X16(v[1]) : : WRES_CLOBBERS);
            asm volatile(WRES_EXTRACT_Y_2 : WR_X16(v[2]) : : WRES_CLOBBERS);
            asm volatile(WRES_EXTRACT_Y_3 : WR_X16(v[3]) : : WRES_CLOBBERS);
        }
        bf16* pbase = tile_base(c);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int hh = 4 * (i >> 1) + lg, ww = 4 * (i & 1) + r;
                if ((c.d0 + wave < a.D) && (c.h0 + hh < a.H) && (c.w0 + ww < a.W)) {
                    float o[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        o[j] = v[i][4 * j + r] + sh[j];
                        s1[j] += o[j]; s2[j] += o[j] * o[j];
                    }
                    bf16x4 pk = {(bf16)o[0], (bf16)o[1], (bf16)o[2], (bf16)o[3]};
                    *reinterpret_cast<bf16x4*>(pbase + ((size_t)hh * a.W + ww) * BN + 4 * lc) = pk;
                }
            }
    };
    auto is_full = [&](const Tile& c) __attribute__((always_inline)) { return c.d0 + TD <= a.D && c.h0 + 8 <= a.H && c.w0 + 8 <= a.W; };
#define WR_LDSO [l0] "v"(ldso[0]), [l1] "v"(ldso[1]), [l2] "v"(ldso[2]), [l3] "v"(ldso[3]), [l4] "v"(ldso[4]), [l5] "v"(ldso[5]),   \
                [l6] "v"(ldso[6]), [l7] "v"(ldso[7]), [l8] "v"(ldso[8]), [l9] "v"(ldso[9])
    // prefetched halo -> LDS, between two barriers.  after_epi: the statement before was a K loop with an epilogue,
    // whose 16 stores were issued after the halo loads and may stay in flight (counted vmcnt); otherwise wait for all
    auto boundary = [&](bool after_epi) __attribute__((always_inline)) {
        // (the statement ends with the fragment reads of the new tile's first tap: they fly during the scalar set-up)
        if (after_epi) asm volatile(WRES_BOUNDARY_EPI : : WR_LDSO, WR_BASES : "memory", WRES_CLOBBERS);
        else asm volatile(WRES_BOUNDARY_ALL : : WR_LDSO, WR_BASES : "memory", WRES_CLOBBERS);
    };

    Tile pt = curT;
    bool pending = false;                                           // an interior tile waits in the set not being computed
    bool last_in_x = true, epi_before = false;
    if (has_work) {
        int cur = 0;
        for (; tile < hi; tile += nper, cur ^= 1) {
            boundary(epi_before);                                    // (the first one also waits for the weight DMA)
            if (cur == 0 && !epi_before) { WR_TL(3) }
            epi_before = pending;                                    // this tile's K loop carries an epilogue iff one is pending
            const Tile me = curT;
            curT = advance(curT);
            const Pf pf = pf_args(curT, tile + nper < hi);
            WR_T0
            if (cur == 0) {
                if (pending) k_epi(0, pf, tile_base(pt));
                else k_plain(0, pf);
            } else {
                if (pending) k_epi(1, pf, tile_base(pt));
                else k_plain(1, pf);
            }
            WR_T1
            WR_TLK
            pt = me;
            pending = is_full(pt);
            if (!pending) { store_ragged(cur, pt); epi_before = false; }   // compiler stores in between: wait for all
            last_in_x = cur == 0;
        }
        WR_TL(6)
        if (pending) {
            if (last_in_x) flush(0, tile_base(pt));
            else flush(1, tile_base(pt));
        }
        WR_TL(7)
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the weight DMA targets this workgroup's LDS
    }
#ifdef WRES_STAMPS
    if (a.stats && tid == 0) {          // [32][2][64] statistics, then per workgroup {K-loop cycles, kernel cycles, kernel 100 MHz ticks, tiles, 12 timeline stamps}
        float* o = a.stats + MM_REPL * 2 * BN + blockIdx.x * 16;
        o[0] = (float)kcyc; o[1] = (float)(__builtin_readcyclecounter() - t_begin);
        o[2] = (float)(wall_clock64() - r_begin); o[3] = (float)((hi - lo - slot_ + nper - 1) / nper);
#pragma unroll
        for (int i = 0; i < 12; ++i) o[4 + i] = tl[i];
    }
#endif
    if (a.stats) {
        float* sstat = reinterpret_cast<float*>(smem + S_OFF);
        // the four lane groups hold the same four channels (rows differ)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s1[j] += __shfl_xor(s1[j], 16); s1[j] += __shfl_xor(s1[j], 32);
            s2[j] += __shfl_xor(s2[j], 16); s2[j] += __shfl_xor(s2[j], 32);
        }
        __syncthreads();                                 // the last tile's LDS reads are done (sstat is its own region anyway)
        if (lg == 0) {                                   // every wave parks its channel sums: no LDS atomics
            float* mine = sstat + wave * 2 * BN;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                mine[4 * lc + j] = s1[j];
                mine[BN + 4 * lc + j] = s2[j];
            }
        }
        __syncthreads();
        float* rep = a.stats + (size_t)(blockIdx.x % MM_REPL) * 2 * BN;
        if (tid < 2 * BN)
            atomicAdd(&rep[tid], (sstat[tid] + sstat[2 * BN + tid]) + (sstat[4 * BN + tid] + sstat[6 * BN + tid]));
    }
#undef WR_LDSO
#undef WR_BASES
#undef WR_K_IN
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
    if (ntiles >= (1 << 24)) return mm_fail(MM_ERR_UNSUPPORTED, "conv3d_fwd_wres: %d tiles", ntiles);
    Conv3dArgs k = a;
    auto magic = [](int d) { return (unsigned)(0x100000000ull / (unsigned)d) + 1u; };
    k.mtw = magic(ceil_div(a.W, 8)); k.mth = magic(ceil_div(a.H, 8)); k.mtd = magic(ceil_div(a.D, TD));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS_BYTES, st, k);
    return mm_check_launch("conv3d_fwd_wres");
}
