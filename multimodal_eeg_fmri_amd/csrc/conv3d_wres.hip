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
//    K loop with the next tap's fragment reads, the previous tile's pack / store / BatchNorm-sum, the next
//    tile's halo prefetch and its way into LDS in the MFMA gaps.  Accumulators, fragments, the prefetched halo,
//    per-lane constants, the BatchNorm sums and the bias live in the accumulator file (a0-a255), which only
//    those statements touch (tests/test_abi_and_host.py audits the compiled ISA).
//  * the halo is a ring of six depth planes and a workgroup walks its tiles down the (b, h, w) columns: the
//    next tile shares two planes and its four new ones replace planes that die during this tile's kd phases,
//    one barrier per phase inside the K loop - no boundary between the tiles of a column, 1/3 less halo traffic.
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
constexpr int LROWS = (TD + 2) * DP;             // 720 LDS rows: six plane slots of 120
constexpr int PCHUNKS = HB * HB * 4;             // 16-byte chunks of one halo plane (400): two per thread
constexpr int ROWB = CIN * 2;                    // bytes per LDS row (64)
constexpr int W_BYTES = 27 * BN * ROWB;          // 110 592
constexpr int H_OFF = W_BYTES;
constexpr int H_BYTES = LROWS * ROWB;            // 46 080
constexpr int SLOTB = DP * ROWB;                 // 7 680 bytes per plane slot
constexpr int S_OFF = H_OFF + H_BYTES;           // per-wave BatchNorm partial sums [4][2][64] fp32
constexpr int LDS_BYTES = S_OFF + 4 * 2 * BN * 4;
static_assert(LDS_BYTES <= 160 * 1024, "LDS");

struct Tile { int b, d, h0, w0; };               // d = tile index along the depth (first output depth = 4 d)

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
#define WR_TL(i)
#define WR_T0
#define WR_T1
#define WR_TLK
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform: tile addressing stays in SGPRs
    const int lc = lane & 15, lg = lane >> 4;                       // MFMA lane = (row / column 0-15, group 0-3)
    // ---- start-up order (profiles/r03_wres_*): what the first MFMA waits for longest is the first tile's halo (a cold
    // HBM read, ~2 000 cycles), so its loads are issued before anything else that touches memory; then the weight DMA;
    // the bias (four global loads) is parked last - everything is waited for together at the first boundary.
    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;
    const int tw = (a.W + 7) / 8, th = (a.H + 7) / 8, td = (a.D + TD - 1) / TD;
    const int ntiles = a.B * td * th * tw;

    // ---- tile list: depth index fastest, so that consecutive tiles walk down a (b, h, w) column.  XCD x (=
    // blockIdx % 8 under round-robin dispatch) owns the x-th eighth of the list, each of its workgroups a contiguous run.
    const int xcd = blockIdx.x & 7, slot_ = blockIdx.x >> 3;
    const int nper = (gridDim.x - xcd + 7) >> 3;                    // workgroups of this XCD
    const int xlo = (int)((long)ntiles * xcd / 8), xn = (int)((long)ntiles * (xcd + 1) / 8) - xlo;
    const int per = xn / nper, rem = xn - per * nper;
    const int lo = xlo + slot_ * per + (slot_ < rem ? slot_ : rem), hi = lo + per + (slot_ < rem ? 1 : 0);
    const bool has_work = lo < hi;                                  // (uniform) a workgroup without tiles still drains its DMA
    auto coords = [&](int t) __attribute__((always_inline)) {       // divisions by host-made reciprocals (t < 2^24)
        Tile c;                                                      // a divisor of 1 has no 32-bit reciprocal (2^32 + 1): q = t
        int q = td == 1 ? t : (int)__umulhi((unsigned)t, a.mtd); c.d = t - q * td; t = q;
        q = tw == 1 ? t : (int)__umulhi((unsigned)t, a.mtw); c.w0 = (t - q * tw) * 8; t = q;
        q = th == 1 ? t : (int)__umulhi((unsigned)t, a.mth); c.h0 = (t - q * th) * 8;
        c.b = q;
        return c;
    };
    auto next_of = [&](Tile c) __attribute__((always_inline)) {     // successor in list order
        if (++c.d == td) {
            c.d = 0; c.w0 += 8;
            if (c.w0 >= tw * 8) { c.w0 = 0; c.h0 += 8; if (c.h0 >= th * 8) { c.h0 = 0; c.b += 1; } }
        }
        return c;
    };
    // ---- a halo plane (10 x 10 voxels x 32 channels) as 16-byte chunks: chunk j of a thread is c = tid + 256 j <
    // 400, row c >> 2 = (hh, hw), channel segment c & 3.  Per-lane constants, computed once: the byte offset relative
    // to the plane's voxel (h0, w0), a one-hot (hh, hw) selector tested against the column's in-volume mask, and the
    // swizzled byte offset inside a ring slot (chunks past 400 park in an unused pitch column).
    int ldsp[2];
    {
        int goffp[2];
        unsigned selp[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = tid + 256 * j;
            const bool real = c < PCHUNKS;
            const int r = real ? c >> 2 : 0, sg = c & 3;
            const int hh = (r * 205) >> 11, hw = r - hh * HB;       // r < 100: exact division by 10
            goffp[j] = (((hh - 1) * a.W + (hw - 1)) * CIN + sg * 8) * 2;
            selp[j] = real ? (1u << hh) | (1u << (10 + hw)) : 0x80000000u;
            ldsp[j] = real ? (hh * WP + hw) * ROWB + ((sg ^ (2 * (hh & 1))) << 4) : (HB + (tid & 1)) * ROWB + (sg << 4);
        }
        asm volatile(WRES_INIT : : [g0] "v"(goffp[0]), [g1] "v"(goffp[1]), [s0] "v"(selp[0]), [s1] "v"(selp[1]) : WRES_CLOBBERS);
    }
    WR_TL(2)
    auto range_mask = [](int lo_, int hi_, int n) __attribute__((always_inline)) {      // bits [max(lo,0), min(hi,n))
        lo_ = lo_ < 0 ? 0 : lo_;
        hi_ = hi_ > n ? n : hi_;
        return hi_ > lo_ ? ((1u << hi_) - 1u) & ~((1u << lo_) - 1u) : 0u;
    };
    // buffer loads: a chunk outside the volume gets an out-of-range offset and the hardware returns zeros
    // (the whole input is < 4 GiB: checked on the host)
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16*>(a.x), 0, (unsigned)((size_t)a.B * a.D * a.H * a.W * CIN * 2), 0x00020000);
    const unsigned planeb = __builtin_amdgcn_readfirstlane((unsigned)(a.H * a.W * CIN * 2));   // one depth step of the input, bytes
    // planes p0 .. p0 + n - 1 of column c (plane p holds input depth p - 1): byte offset of plane p0's voxel (h0, w0),
    // which planes lie inside the volume, and which (hh, hw) of the column do
    struct Pf { unsigned off0, pvalid, mhw; };
    auto pf_args = [&](const Tile& c, int p0, int n, bool any) __attribute__((always_inline)) {      // uniform
        Pf p;
        p.off0 = (unsigned)((((c.b * a.D + (p0 - 1)) * a.H + c.h0) * a.W + c.w0) * CIN * 2);
        p.pvalid = any ? range_mask(1 - p0, a.D + 1 - p0, n) : 0u;
        p.mhw = range_mask(1 - c.h0, a.H - c.h0 + 1, HB) | (range_mask(1 - c.w0, a.W - c.w0 + 1, HB) << 10);
        p.off0 = __builtin_amdgcn_readfirstlane(p.off0);
        p.pvalid = __builtin_amdgcn_readfirstlane(p.pvalid);
        p.mhw = __builtin_amdgcn_readfirstlane(p.mhw);
        return p;
    };
    Tile curT = coords(has_work ? lo : 0);
    {   // the first tile's six planes: everything else of the set-up overlaps their latency
        const Pf pf = pf_args(curT, 4 * curT.d, 6, has_work);
        asm volatile(WRES_PREFETCH : : [off0] "s"(pf.off0), [planeb] "s"(planeb), [pvalid] "s"(pf.pvalid), [mhw] "s"(pf.mhw),
                     [rsrc] "s"(xrsrc) : "memory", WRES_CLOBBERS);
    }
    WR_TL(0)
    // ---- weights: global -> LDS by LDS-DMA.  LDS image [tap][rho][slot] x 16 B; a DMA writes wave-uniform base +
    // 16 * lane, so wave w, instruction k covers tap k, rows rho = 16 w + (lane >> 2), slot = lane & 3.  LDS row rho
    // holds output channel n = 4 (rho & 15) + (rho >> 4) (column tile j = rho >> 4, lane column c = rho & 15 <->
    // channel 4 c + j); the channel segment in slot s is s ^ 2 ((rho >> 2) & 1).  Both permutations sit in the
    // SOURCE address, which is affine in the tap (+64 bytes).
    const bf16* wsrc9;                                                // this lane's source of tap 9
    {
        const int rho = wave * 16 + (lane >> 2), slot = lane & 3;
        const int n = 4 * (rho & 15) + (rho >> 4);
        const bf16* wlane = a.w + (size_t)n * 27 * CIN + ((slot ^ (2 * ((rho >> 2) & 1))) << 3);
        auto dma_taps = [&](int k0, int k1) __attribute__((always_inline)) {
#pragma unroll
            for (int k = k0; k < k1; ++k)
                __builtin_amdgcn_global_load_lds((gptr_t*)(wlane + k * CIN), (lptr_t*)(smem + k * BN * ROWB + wave * 1024), 16, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        // taps 0-8 (kd = 0) and the bias: the first tile starts once the halo and this first batch are in LDS; taps 9-26
        // are issued in the MFMA gaps of its first taps (WRES_K_X_*_FIRST) and land before tap 9 needs them
        dma_taps(0, 9);
        if (a.shift && wave == 0) {                                  // 64 floats -> the (still unused) statistics area
            __builtin_amdgcn_global_load_lds((gptr_t*)(a.shift + lane), (lptr_t*)(smem + S_OFF), 4, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        wsrc9 = wlane + 9 * CIN;
    }
    WR_TL(1)
    float sh[4] = {0.f, 0.f, 0.f, 0.f};                              // a lane's four channels 4 lc + j: read from LDS after the first boundary

    // ---- per-lane fragment bases (absolute LDS byte addresses).  Lane (lc, lg) of MFMA tile i reads the halo row
    // of voxel (h, w) = (4 (i >> 1) + (lc >> 2), 4 (i & 1) + (lc & 3)) shifted by (kh, kw), channel segment lg, in the
    // ring slot of plane d0 + kd + wave; of column tile j the weight row rho = 16 j + lc.
    const int lds0 = (int)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int hbase = lds0 + H_OFF;
    const int arow = hbase + ((lc >> 2) * WP + (lc & 3)) * ROWB;
    const int lane_a0 = arow + ((lg ^ (2 * (((lc >> 2) + 0) & 1))) << 4);    // taps with kh even
    const int lane_a1 = arow + ((lg ^ (2 * (((lc >> 2) + 1) & 1))) << 4);    // kh odd
    const int bb0 = lds0 + lc * ROWB + ((lg ^ (2 * ((lc >> 2) & 1))) << 4);
    const int bb1 = bb0 + 13 * BN * ROWB;                                     // taps 13..26: ds offsets are 16-bit
    const int wlds9 = __builtin_amdgcn_readfirstlane(lds0 + 9 * BN * ROWB + wave * 1024);      // this wave's LDS piece of tap 9
    const unsigned pitch_b = (unsigned)a.W * BN * 2;                 // one h step of the output, in bytes
    const unsigned pitch4 = __builtin_amdgcn_readfirstlane(4 * pitch_b);
    // accumulator register r of lane (lc, lg) in MFMA tile (i, j) is voxel (h, w) = (4 (i >> 1) + lg, 4 (i & 1) + r)
    const unsigned voff0 = lg * pitch_b + lc * 8;
    float rs1[4] = {0.f, 0.f, 0.f, 0.f}, rs2[4] = {0.f, 0.f, 0.f, 0.f};     // BatchNorm sums of ragged tiles (ordinary code)
    auto slot_of = [&](int p) __attribute__((always_inline)) { return (p % 6) * SLOTB; };       // p >= 0, uniform

    auto tile_base = [&](const Tile& c) __attribute__((always_inline)) {                           // wave-uniform: (b, 4 d + wave, h0, w0, 0)
        bf16* p = a.out_bf16 + ((((size_t)c.b * a.D + 4 * c.d + wave) * a.H + c.h0) * a.W + c.w0) * BN;
        // provably uniform for the "s" operand of the hand-written stores (cdna_hip_programming.md T20)
        const unsigned long long u = reinterpret_cast<unsigned long long>(p);
        const unsigned lo_ = __builtin_amdgcn_readfirstlane((unsigned)u), hi_ = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
        return reinterpret_cast<bf16*>(((unsigned long long)hi_ << 32) | lo_);
    };
#define WR_AB [ab00] "v"(ab[0][0]), [ab01] "v"(ab[0][1]), [ab10] "v"(ab[1][0]), [ab11] "v"(ab[1][1]), [ab20] "v"(ab[2][0]), \
              [ab21] "v"(ab[2][1]), [bb0] "v"(bb0), [bb1] "v"(bb1)
#define WR_PF [off0] "s"(pf.off0), [planeb] "s"(planeb), [pvalid] "s"(pf.pvalid), [mhw] "s"(pf.mhw), [rsrc] "s"(xrsrc)
#define WR_MARCH [abn] "v"(abn), [sb0] "s"(sb[0]), [sb1] "s"(sb[1]), [sb2] "s"(sb[2]), [sb3] "s"(sb[3]), [ldsp0] "v"(ldsp[0]),   \
                 [ldsp1] "v"(ldsp[1])
#define WR_EPI [pbase] "s"(pbase), [pitch4] "s"(pitch4), [voff0] "v"(voff0)
#define WR_W9 [wlo] "v"((unsigned)(reinterpret_cast<unsigned long long>(wsrc9))), [whi] "v"((unsigned)(reinterpret_cast<unsigned long long>(wsrc9) >> 32)), \
              [wlds] "s"(wlds9)
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
                if ((4 * c.d + wave < a.D) && (c.h0 + hh < a.H) && (c.w0 + ww < a.W)) {
                    float o[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        o[j] = v[i][4 * j + r] + sh[j];
                        rs1[j] += o[j]; rs2[j] += o[j] * o[j];
                    }
                    bf16x4 pk = {(bf16)o[0], (bf16)o[1], (bf16)o[2], (bf16)o[3]};
                    *reinterpret_cast<bf16x4*>(pbase + ((size_t)hh * a.W + ww) * BN + 4 * lc) = pk;
                }
            }
    };

    bool pending = false;                                           // an interior tile waits in the set not being computed
    bool last_in_x = true;
    const bf16* base_prev = a.out_bf16;                              // output base of the pending tile (wave-uniform)
    if (has_work) {
        // the first tile's six planes -> ring slots, between two barriers (the vmcnt(0) in front also covers the weight DMA)
        auto full_boundary = [&](const Tile& c, bool counted) __attribute__((always_inline)) {
            int sb[6];
#pragma unroll
            for (int pp = 0; pp < 6; ++pp) sb[pp] = __builtin_amdgcn_readfirstlane(hbase + slot_of(4 * c.d + pp));
            const int abn = lane_a0 + slot_of(4 * c.d + wave);
            // counted: the statement before was a K loop with an epilogue, whose 16 stores were issued after the plane
            // loads and may stay in flight; otherwise wait for everything
            if (counted)
                asm volatile(WRES_BOUNDARY_EPI : : [sb0] "s"(sb[0]), [sb1] "s"(sb[1]), [sb2] "s"(sb[2]), [sb3] "s"(sb[3]), [sb4] "s"(sb[4]),
                             [sb5] "s"(sb[5]), [ldsp0] "v"(ldsp[0]), [ldsp1] "v"(ldsp[1]), [abn] "v"(abn), [bb0] "v"(bb0) : "memory", WRES_CLOBBERS);
            else
                asm volatile(WRES_BOUNDARY_ALL : : [sb0] "s"(sb[0]), [sb1] "s"(sb[1]), [sb2] "s"(sb[2]), [sb3] "s"(sb[3]), [sb4] "s"(sb[4]),
                             [sb5] "s"(sb[5]), [ldsp0] "v"(ldsp[0]), [ldsp1] "v"(ldsp[1]), [abn] "v"(abn), [bb0] "v"(bb0) : "memory", WRES_CLOBBERS);
        };
        {   // first tile: wait for the halo, taps 0-8 of the weights and the bias only
            int sb[6];
#pragma unroll
            for (int pp = 0; pp < 6; ++pp) sb[pp] = __builtin_amdgcn_readfirstlane(hbase + slot_of(4 * curT.d + pp));
            const int abn = lane_a0 + slot_of(4 * curT.d + wave);
            asm volatile(WRES_BOUNDARY_FIRST : : [sb0] "s"(sb[0]), [sb1] "s"(sb[1]), [sb2] "s"(sb[2]), [sb3] "s"(sb[3]), [sb4] "s"(sb[4]),
                         [sb5] "s"(sb[5]), [ldsp0] "v"(ldsp[0]), [ldsp1] "v"(ldsp[1]), [abn] "v"(abn), [bb0] "v"(bb0) : "memory", WRES_CLOBBERS);
        }
        if (a.shift) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(smem + S_OFF + 16 * lc);
            sh[0] = b4[0]; sh[1] = b4[1]; sh[2] = b4[2]; sh[3] = b4[3];
        }
        asm volatile(WRES_INIT_BIAS : : [sh0] "v"(sh[0]), [sh1] "v"(sh[1]), [sh2] "v"(sh[2]), [sh3] "v"(sh[3]) : WRES_CLOBBERS);
        WR_TL(3)
        int cur = 0, tile = lo;
        while (tile < hi) {
            // ---- a column segment: tiles `tile` .. `tile + nseg - 1` walk down one (b, h, w) column.  What changes
            // from tile to tile is kept as running scalars (no per-tile divisions, masks or multiplications):
            //   so[k]  ring-slot byte address (LDS) of plane d0 + k, k = 0..5;  sw[k] = the same for plane d0 + k + wave
            //   offn   byte offset (input) of plane d0 + 6's voxel (h0, w0);  base_me  output base of the tile
            const Tile col = curT;
            const int nseg = (td - col.d) < (hi - tile) ? (td - col.d) : (hi - tile);
            int d0 = 4 * col.d;
            int so[6], sw[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                so[k] = __builtin_amdgcn_readfirstlane(hbase + slot_of(d0 + k));
                sw[k] = __builtin_amdgcn_readfirstlane(slot_of(d0 + k + wave));
            }
            unsigned offn = __builtin_amdgcn_readfirstlane((unsigned)((((col.b * a.D + d0 + 5) * a.H + col.h0) * a.W + col.w0) * CIN * 2));
            const unsigned mhw = __builtin_amdgcn_readfirstlane(range_mask(1 - col.h0, a.H - col.h0 + 1, HB) |
                                                                (range_mask(1 - col.w0, a.W - col.w0 + 1, HB) << 10));
            const bool hw_full = col.h0 + 8 <= a.H && col.w0 + 8 <= a.W;
            const bf16* base_me = tile_base(col);
            const size_t out_step = (size_t)4 * a.H * a.W * BN;      // four output depths, in elements
            for (int k = 0; k < nseg; ++k, ++tile, cur ^= 1) {
                const bool last = k == nseg - 1;                     // last tile of the segment: column change or end of the run
                int ab[3][2];
#pragma unroll
                for (int kd = 0; kd < 3; ++kd) { ab[kd][0] = lane_a0 + sw[kd]; ab[kd][1] = lane_a1 + sw[kd]; }
                const bf16* pbase = base_prev;
                const bool me_full = hw_full && d0 + TD <= a.D;
                WR_T0
                if (!last) {
                    // new planes d0 + 6 .. d0 + 9 take the slots of planes d0 .. d0 + 3 as those die
                    int nv = a.D - (d0 + 5);                         // how many of the four new planes lie inside the volume
                    nv = nv < 0 ? 0 : (nv > 4 ? 4 : nv);
                    Pf pf;
                    pf.off0 = offn; pf.pvalid = __builtin_amdgcn_readfirstlane((1u << nv) - 1u); pf.mhw = mhw;
                    const int abn = lane_a0 + sw[4];                 // the next tile's kd = 0 plane of this wave: d0 + 4 + wave
                    const int sb[4] = {__builtin_amdgcn_readfirstlane(so[0]), __builtin_amdgcn_readfirstlane(so[1]),
                                       __builtin_amdgcn_readfirstlane(so[2]), __builtin_amdgcn_readfirstlane(so[3])};
                    if (cur == 0) {
                        if (pending) asm volatile(WRES_K_X_EPI_MARCH : : WR_AB, WR_PF, WR_MARCH, WR_EPI : "memory", WRES_CLOBBERS);
                        else if (tile == lo) asm volatile(WRES_K_X_MARCH_FIRST : : WR_AB, WR_PF, WR_MARCH, WR_W9 : "memory", WRES_CLOBBERS);
                        else asm volatile(WRES_K_X_MARCH : : WR_AB, WR_PF, WR_MARCH : "memory", WRES_CLOBBERS);
                    } else {
                        if (pending) asm volatile(WRES_K_Y_EPI_MARCH : : WR_AB, WR_PF, WR_MARCH, WR_EPI : "memory", WRES_CLOBBERS);
                        else asm volatile(WRES_K_Y_MARCH : : WR_AB, WR_PF, WR_MARCH : "memory", WRES_CLOBBERS);
                    }
                } else {
                    // six planes of the next tile's column (end of the run: nothing valid to fetch)
                    Tile nx = col; nx.d = col.d + k;
                    nx = next_of(nx);
                    const bool has_next = tile + 1 < hi;
                    const Pf pf = pf_args(nx, 4 * nx.d, 6, has_next);
                    const bool epi = pending;
                    if (cur == 0) {
                        if (pending) asm volatile(WRES_K_X_EPI_COL : : WR_AB, WR_PF, WR_EPI : "memory", WRES_CLOBBERS);
                        else if (tile == lo) asm volatile(WRES_K_X_COL_FIRST : : WR_AB, WR_PF, WR_W9 : "memory", WRES_CLOBBERS);
                        else asm volatile(WRES_K_X_COL : : WR_AB, WR_PF : "memory", WRES_CLOBBERS);
                    } else {
                        if (pending) asm volatile(WRES_K_Y_EPI_COL : : WR_AB, WR_PF, WR_EPI : "memory", WRES_CLOBBERS);
                        else asm volatile(WRES_K_Y_COL : : WR_AB, WR_PF : "memory", WRES_CLOBBERS);
                    }
                    curT = nx;
                    if (!me_full) {                                  // ordinary stores before the boundary: it waits for all
                        Tile me = col; me.d = col.d + k;
                        store_ragged(cur, me);
                    }
                    if (has_next) full_boundary(nx, epi && me_full);
                }
                WR_T1
                WR_TLK
                if (!last && !me_full) {
                    Tile me = col; me.d = col.d + k;
                    store_ragged(cur, me);
                }
                pending = me_full;
                last_in_x = cur == 0;
                base_prev = base_me;
                // ---- down the column: d0 += 4, the ring turns by four slots
                d0 += 4;
                offn += 4 * planeb;
                base_me += out_step;
                const int t0 = so[0], t1 = so[1], u0 = sw[0], u1 = sw[1];
                so[0] = so[4]; so[1] = so[5]; so[4] = so[2]; so[5] = so[3]; so[2] = t0; so[3] = t1;
                sw[0] = sw[4]; sw[1] = sw[5]; sw[4] = sw[2]; sw[5] = sw[3]; sw[2] = u0; sw[3] = u1;
            }
        }
        WR_TL(6)
        if (pending) {
            const bf16* pbase = base_prev;
            if (last_in_x) asm volatile(WRES_FLUSH_X : : WR_EPI : "memory", WRES_CLOBBERS);
            else asm volatile(WRES_FLUSH_Y : : WR_EPI : "memory", WRES_CLOBBERS);
        }
        WR_TL(7)
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the weight DMA targets this workgroup's LDS
    }
#ifdef WRES_STAMPS
    if (a.stats && tid == 0) {          // [32][2][64] statistics, then per workgroup {K-loop cycles, kernel cycles, kernel 100 MHz ticks, tiles, 12 timeline stamps}
        float* o = a.stats + MM_REPL * 2 * BN + blockIdx.x * 16;
        o[0] = (float)kcyc; o[1] = (float)(__builtin_readcyclecounter() - t_begin);
        o[2] = (float)(wall_clock64() - r_begin); o[3] = (float)(hi - lo);
#pragma unroll
        for (int i = 0; i < 12; ++i) o[4 + i] = tl[i];
        o[12] = (float)(r_begin & 0xFFFFFF); o[13] = (float)(wall_clock64() & 0xFFFFFF);     // absolute 100 MHz ticks: start stagger and span over workgroups
    }
#endif
    if (a.stats) {
        float* sstat = reinterpret_cast<float*>(smem + S_OFF);
        float st[8];
        asm volatile(WRES_STATS_OUT : "=v"(st[0]), "=v"(st[1]), "=v"(st[2]), "=v"(st[3]), "=v"(st[4]), "=v"(st[5]), "=v"(st[6]), "=v"(st[7])
                     : : WRES_CLOBBERS);
        float s1[4], s2[4];
        // the four lane groups hold the same four channels (rows differ)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s1[j] = st[j] + rs1[j]; s2[j] = st[4 + j] + rs2[j];
            s1[j] += __shfl_xor(s1[j], 16); s1[j] += __shfl_xor(s1[j], 32);
            s2[j] += __shfl_xor(s2[j], 16); s2[j] += __shfl_xor(s2[j], 32);
        }
        __syncthreads();                                 // (sstat is its own LDS region; the barrier orders it after the last tile anyway)
        if (lg == 0) {                                   // every wave parks its channel sums: no LDS atomics
            float* mine = sstat + wave * 2 * BN;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                mine[4 * lc + j] = s1[j];
                mine[BN + 4 * lc + j] = s2[j];
            }
        }
        __syncthreads();
        mm_acc_t* rep = acc_rep(a.stats, blockIdx.x % MM_ACC_REPL, 2 * BN);
        if (tid < 2 * BN)
            acc_add<MM_ACC_STAT>(&rep[tid], (sstat[tid] + sstat[2 * BN + tid]) + (sstat[4 * BN + tid] + sstat[6 * BN + tid]));
    }
#undef WR_AB
#undef WR_PF
#undef WR_MARCH
#undef WR_EPI
#undef WR_W9
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
    // floor(2^32 / d) + 1; d = 1 would need 2^32 + 1 and is special-cased in the kernel's coords() (the value is unused)
    auto magic = [](int d) { return d == 1 ? 0u : (unsigned)(0x100000000ull / (unsigned)d) + 1u; };
    k.mtw = magic(ceil_div(a.W, 8)); k.mth = magic(ceil_div(a.H, 8)); k.mtd = magic(ceil_div(a.D, TD));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS_BYTES, st, k);
    return mm_check_launch("conv3d_fwd_wres");
}
