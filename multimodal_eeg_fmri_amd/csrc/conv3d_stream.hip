// Weight-streaming 3-D convolution (k = 3, pad 1) for the channel-heavy layers of the voxel encoder:
// layer 3 forward (64 -> 128), its data gradient (128 -> 64) and layer 2's data gradient (64 -> 32).
// bf16 MFMA 16x16x32, fp32 accumulate, channels-last in and out.
//
//   Y[b, v, n] = bias[n] + sum_{tap, c} X[b, v + off(tap), c] * W[n, tap, c]
//
// Here the weights (27 * Cin * Cout * 2 B = 110-442 KB) do not fit beside the halo, or there are too few
// output tiles to amortise loading them (8^3 volumes).  A workgroup owns TWO 8 x 8 faces of output voxels
// (2 x 8 x 8 = 128 GEMM rows) x NW groups of 32 output channels, stages the 4 x 10 x 10 x Cin halo in LDS once,
// and the weights never touch LDS: every wave loads its B operands (its 32 columns, its K-group's 32-channel
// slice of a tap = one "unit" = one MFMA K-step) straight from L2 into registers, DEPTH units ahead of the
// MFMAs, in exactly the lane layout the MFMA wants; mm_prep_conv_weight writes the image of these shapes in that
// order (common.h: conv_image_index), so a wave load is 1 KB contiguous.
//
// Why this shape (measured, profiles/r02_stream_*): per CU the weights have to arrive at up to 64 B/clk - the
// vector-memory path's peak - and every CU of an XCD pulls the SAME bytes from its L2 at the same time.
//  * an LDS-DMA ring (first version) adds an LDS write per weight byte on top of the fragment reads: 384 LDS cycles
//    per 256 MFMA cycles, measured 504;
//  * 64 rows x all columns per workgroup asks the L2 for 100 % of its bandwidth (435 cycles per unit measured);
//    128 rows x half the columns is the same MFMA work per workgroup with half the weight bytes, and a wave tile
//    of 128 rows x 32 columns (8 x 2 MFMA tiles) needs 2 KB of weights per 16 MFMAs: the vector-memory path runs at
//    half rate and the LDS pipe (8 A-fragment reads per 16 MFMAs and wave) exactly as busy as the MFMA pipes,
//    as in conv3d_wres.hip.
//  * the four waves split the workgroup's NW column groups first and K second (WK = 4 / NW K-groups: unit g of the
//    27 * Cin / 32 belongs to K-group g % WK); K-groups are summed through LDS (the dead halo) after the loop,
//    each wave keeping 8 / WK row tiles for the epilogue.  No barrier inside the K loop.
//  * the halo image is that of conv3d_wres.hip per 32-channel slice: unpadded 64-byte rows, 16-byte slots
//    XOR-swizzled with 2 * (patch-row parity) - conflict-free ds_read_b128 for every tap shift.  MFMA row m of row
//    tile i <-> face i >> 2, voxel (h, w) = (4 ((i >> 1) & 1) + (m >> 2), 4 (i & 1) + (m & 3)).
//    Column c of column tile j of group wn is channel 32 wn + 2 c + j: a lane holds two adjacent channels.
//  * the A-fragment reads are inline asm with explicit lgkmcnt waits so that they stay where the software
//    pipeline puts them (one unit ahead); the B loads are ordinary loads, the compiler counts vmcnt.
#include "conv3d_args.h"

#ifndef STREAM_ABL      // diagnostic builds (tools/abl_stream.sh): 1 no B loads in the loop, 2 no MFMAs, 4 no fragment reads
#define STREAM_ABL 0
#endif

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int STD = 2;                  // faces (depth planes) per workgroup
constexpr int SHB = 10;                 // halo edge of an 8-wide face
constexpr int SWP = 12;                 // halo w-pitch in LDS rows
constexpr int SDP = SHB * SWP;          // rows per halo plane
constexpr int SROWB = 64;               // bytes per LDS row (32 channels)
constexpr int HKS = (STD + 2) * SDP * SROWB;    // bytes of one 32-channel halo image (30 720)

template <int CIN, int COUT, int NW_, int DEPTH_>
struct StreamCfg {
    static constexpr int KS = CIN / 32;                 // K-steps (units) per tap
    static constexpr int NW = NW_;                      // column groups (32 channels) per workgroup = waves along N
    static constexpr int WK = 4 / NW;                   // K-groups
    static constexpr int NU = 27 * KS;                  // units in all
    static constexpr int NUW = (NU + WK - 1) / WK;      // units per wave
    static constexpr int DEPTH = DEPTH_;                // B operands in flight, in units
    static constexpr int OWN = 8 / WK;                  // row tiles per wave after the K-group sum
    static constexpr int H_BYTES = KS * HKS;
    static constexpr int PARK = WK > 1 ? 4 * 16 * 1024 : 0;         // [wave][i][j][lane] x 16 B, overlays the halo
    static constexpr int S_OFF = H_BYTES > PARK ? H_BYTES : PARK;   // [4 waves][2][32] fp32 BatchNorm partials
    static constexpr int LDS = S_OFF + 4 * 2 * 32 * 4;
    static_assert((NW == 1 || NW == 2 || NW == 4) && (COUT / 32) % NW == 0 && DEPTH <= NUW, "shape");
    static_assert(LDS <= 160 * 1024, "LDS");
};

__device__ __forceinline__ int tap_off(int tap) {       // byte offset of a tap inside a 32-channel halo image
    const int kd = tap / 9, r = tap - 9 * kd, kh = r / 3, kw = r - 3 * kh;
    return (kd * SDP + kh * SWP + kw) * SROWB;
}

__device__ __forceinline__ u32x4 lds_read128(int addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}

template <int CIN, int COUT, int NW_, int DEPTH_>
__global__ __launch_bounds__(256) void conv3d_stream_kernel(Conv3dArgs a) {
    using C = StreamCfg<CIN, COUT, NW_, DEPTH_>;
    constexpr int KS = C::KS, NW = C::NW, WK = C::WK, NU = C::NU, NUW = C::NUW, DEPTH = C::DEPTH, OWN = C::OWN;
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef STREAM_STAMPS
    const long long t_begin = __builtin_readcyclecounter(), r_begin = wall_clock64();
    float tl[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#define ST_TL(i) tl[i] = (float)(__builtin_readcyclecounter() - t_begin);
#else
#define ST_TL(i)
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % NW, kg = wave / NW;
    const int lc = lane & 15, lg = lane >> 4;
    const int lds0 = (int)(size_t)(__attribute__((address_space(3))) char*)smem;

    // ---- tile: XCD x (= blockIdx.x % 8 under round-robin dispatch) takes the x-th eighth of the (b, d, h, w)-ordered list
    const int tw = (a.W + 7) / 8, th = (a.H + 7) / 8, td = (a.D + STD - 1) / STD;
    int t = blockIdx.x;
    if ((gridDim.x & 7) == 0) t = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int w0 = (t % tw) * 8; t /= tw;
    const int h0 = (t % th) * 8; t /= th;
    const int d0 = (t % td) * STD;
    const int b = t / td;
    const int grp = blockIdx.y * NW + wn;                      // this wave's column group: channels 32 grp .. + 31

    // ---- B operands: unit u of this wave is global unit g = u WK + kg = (tap, ks); the image (common.h:
    // conv_image_index) holds, per unit and column group, 2 x 1 KB in lane order: column tile j of lane (lc, lg) is
    // weight row n0 + j, channels 32 ks + 8 lg .. + 7 of the tap
    const int n0 = 32 * grp + 2 * lc;
    const bf16* wlane = a.w + (grp * 2 * 64 + lane) * 8;
    auto load_b = [&](int u, u32x4 (&dst)[2]) __attribute__((always_inline)) {
        int g = u * WK + kg;
        if (u * WK + WK > NU) g = g < NU ? g : NU - 1;         // a K-group without a last unit re-reads a valid one (unused)
        const bf16* p = wlane + g * (COUT * 32);
        dst[0] = *reinterpret_cast<const u32x4*>(p);
        dst[1] = *reinterpret_cast<const u32x4*>(p + 512);
    };
    u32x4 bq[DEPTH][2];
    float sh[2];

    // ---- halo: (STD + 2) x 10 x 10 voxels x CIN channels.  A "piece" is one halo row (10 voxels) x 64 channels =
    // 80 chunks of 16 bytes, contiguous per voxel in memory; three pieces per pass of the 256 threads (240 busy), so a
    // thread's voxel column and channel segment never change and a pass costs a handful of integer instructions.
    // The first B operands and the bias are requested right behind the first batch of halo loads (loads return in
    // order: the halo's wait does not include them), so their latency hides behind the LDS writes and the barrier.
    {
        constexpr int PPR = CIN / 64;                          // pieces per halo row
        constexpr int NP = (STD + 2) * SHB * PPR;              // 40 / 80 pieces
        constexpr int PASSES = (NP + 2) / 3;                   // 14 / 27
        constexpr int BATCH = PASSES > 14 ? (PASSES + 1) / 2 : PASSES;
        const int rs = tid / 80, ch = tid - 80 * rs;           // piece of the pass (3 = idle), chunk of the piece
        const int hw = ch >> 3, sg8 = ch & 7;
        const int w = w0 + hw - 1;
        const bool wok = rs < 3 && w >= 0 && w < a.W;
        const bf16* xb = a.x + (size_t)b * a.D * a.H * a.W * CIN + w * CIN + sg8 * 8;
        const int ldst = hw * SROWB;
#pragma unroll
        for (int q0 = 0; q0 < PASSES; q0 += BATCH) {
            u32x4 v[BATCH];
#pragma unroll
            for (int q = 0; q < BATCH; ++q) {
                const int p = 3 * (q0 + q) + rs;
                const int row = p / PPR, half = p % PPR;
                const int hd = row / SHB, hh = row - SHB * hd;
                const int d = d0 + hd - 1, h = h0 + hh - 1;
                v[q] = u32x4{0u, 0u, 0u, 0u};
                if (q0 + q < PASSES && p < NP && wok && d >= 0 && d < a.D && h >= 0 && h < a.H)
                    v[q] = *reinterpret_cast<const u32x4*>(xb + (d * a.H + h) * a.W * CIN + half * 64);
            }
            if (q0 == 0) {
#pragma unroll
                for (int u = 0; u < DEPTH; ++u) load_b(u, bq[u]);
#pragma unroll
                for (int j = 0; j < 2; ++j) sh[j] = a.shift ? a.shift[n0 + j] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < BATCH; ++q) {
                const int p = 3 * (q0 + q) + rs;
                const int row = p / PPR, half = p % PPR;
                const int hd = row / SHB, hh = row - SHB * hd;
                const int seg = half * 8 + sg8;
                if (q0 + q < PASSES && p < NP && rs < 3)
                    *reinterpret_cast<u32x4*>(smem + (seg >> 2) * HKS + (hd * SDP + hh * SWP) * SROWB + ldst +
                                              (((seg & 3) ^ (2 * (hh & 1))) << 4)) = v[q];
            }
        }
    }

    // ---- per-lane A-fragment bases (absolute LDS byte addresses)
    const int arow = ((lc >> 2) * SWP + (lc & 3)) * SROWB;
    const int lane_a0 = lds0 + arow + ((lg ^ (2 * (((lc >> 2) + 0) & 1))) << 4);     // taps with kh even
    const int lane_a1 = lds0 + arow + ((lg ^ (2 * (((lc >> 2) + 1) & 1))) << 4);     // kh odd

    f32x4 acc[8][2];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // unit u of this wave -> global unit g = u WK + kg -> (tap, ks), written so that what is a compile-time constant
    // after unrolling stays one (kg < WK is not known to the compiler)
    auto a_addr = [&](int u, int i) __attribute__((always_inline)) {
        int tap, ks;
        if constexpr (KS % WK == 0) { tap = (u * WK) / KS; ks = (u * WK) % KS + kg; }       // the K-groups share a tap
        else { tap = u * (WK / KS) + kg / KS; ks = kg % KS; }                                 // WK = 4, KS = 2
        const int kh = (tap / 3) % 3;
        return ((kh & 1) ? lane_a1 : lane_a0) + ks * HKS + tap_off(tap) +
               ((i >> 2) * SDP + (4 * ((i >> 1) & 1)) * SWP + 4 * (i & 1)) * SROWB;
    };

    ST_TL(0)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                              // the halo (all waves' writes) is in LDS
    ST_TL(1)

    u32x4 fa[2][8];                                            // A fragments: unit u lives in fa[u & 1], no copies
#pragma unroll
    for (int i = 0; i < 8; ++i) fa[0][i] = lds_read128(a_addr(0, i));

    // Software pipeline: group i of unit u requests fragment i of unit u + 1, then runs the two MFMAs of fragment i of
    // unit u - which was requested eight reads earlier, so "at most 8 LDS reads outstanding" is exactly "it has
    // arrived" and every read has a whole unit (256 MFMA cycles) to come back.  (One wait per unit at lgkmcnt(0) left
    // the last read of a unit ~30 cycles: 470 cycles per unit; eight reads in a burst in front of the MFMAs: 500.)
#pragma unroll
    for (int u = 0; u < NUW; ++u) {
        u32x4 (&cur)[8] = fa[u & 1];
        u32x4 (&nxt)[8] = fa[(u + 1) & 1];
        const bool valid = (u * WK + WK <= NU) ? true : (u * WK + kg < NU);
        u32x4 cb[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) cb[j] = bq[u % DEPTH][j];
        if (u + DEPTH < NUW && !(STREAM_ABL & 1)) load_b(u + DEPTH, bq[u % DEPTH]);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (u + 1 < NUW && !(STREAM_ABL & 4)) {
                nxt[i] = lds_read128(a_addr(u + 1, i));
                asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(cur[i]));
            } else {
                switch (i) {                                   // last unit: reads i .. 7 of it are the only ones outstanding
                    case 0: asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(cur[i])); break;
                    case 1: asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(cur[i])); break;
                    case 2: asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(cur[i])); break;
                    case 3: asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(cur[i])); break;
                    case 4: asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(cur[i])); break;
                    case 5: asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(cur[i])); break;
                    case 6: asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(cur[i])); break;
                    default: asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur[i])); break;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (valid && !(STREAM_ABL & 2)) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, cur[i]), __builtin_bit_cast(bf16x8, cb[j]),
                                                                         acc[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    ST_TL(2)
    // ---- K-groups: every wave parks its partial wave tile, wave (wn, kg) then owns row tiles i = kg OWN .. + OWN - 1
    f32x4 res[OWN][2];
    if constexpr (WK == 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) res[i][j] = acc[i][j];
    } else {
        __syncthreads();                                       // every wave is done with the halo
        f32x4* park = reinterpret_cast<f32x4*>(smem);          // [wave][i][j][lane]
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) park[((wave * 8 + i) * 2 + j) * 64 + lane] = acc[i][j];
        __syncthreads();
#pragma unroll
        for (int o = 0; o < OWN; ++o)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x4 t4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < WK; ++k) t4 += park[(((wn + NW * k) * 8 + kg * OWN + o) * 2 + j) * 64 + lane];   // fixed order
                res[o][j] = t4;
            }
    }

    ST_TL(3)
    // ---- epilogue: register r of lane (lc, lg) in row tile i is face i >> 2, voxel (h, w) = (4 ((i >> 1) & 1) + lg,
    // 4 (i & 1) + r), channels n0 and n0 + 1
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
    const size_t obase = ((((size_t)b * a.D + d0) * a.H + h0) * a.W + w0) * COUT + n0;
    float* of = a.out_f32 ? a.out_f32 + obase : nullptr;
    bf16* ob = a.out_bf16 ? a.out_bf16 + obase : nullptr;
    const int fstride = a.H * a.W * COUT;                      // one depth step of the output
#pragma unroll
    for (int o = 0; o < OWN; ++o) {
        const int i = kg * OWN + o;
        const int f = i >> 2, hh = 4 * ((i >> 1) & 1) + lg;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ww = 4 * (i & 1) + r;
            if (d0 + f < a.D && h0 + hh < a.H && w0 + ww < a.W) {
                const float v0 = res[o][0][r] + sh[0], v1 = res[o][1][r] + sh[1];
                s1[0] += v0; s2[0] += v0 * v0; s1[1] += v1; s2[1] += v1 * v1;
                const int off = f * fstride + (hh * a.W + ww) * COUT;
                if (of) *reinterpret_cast<float2*>(of + off) = float2{v0, v1};
                if (ob) *reinterpret_cast<bf16x2*>(ob + off) = bf16x2{(bf16)v0, (bf16)v1};
            }
        }
    }
    ST_TL(4)
#ifdef STREAM_STAMPS
    if (a.stats && tid == 0 && blockIdx.y == 0) {
        float* o = a.stats + MM_REPL * 2 * COUT + blockIdx.x * 8;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        o[0] = tl[0]; o[1] = tl[1]; o[2] = tl[2]; o[3] = tl[3]; o[4] = tl[4];
        o[5] = (float)(__builtin_readcyclecounter() - t_begin);
        o[6] = (float)(r_begin & 0xFFFFFF); o[7] = (float)(wall_clock64() & 0xFFFFFF);
    }
#endif
    if (a.stats) {
        float* sstat = reinterpret_cast<float*>(smem + C::S_OFF);       // [wave][2][32]
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            s1[j] += __shfl_xor(s1[j], 16); s1[j] += __shfl_xor(s1[j], 32);
            s2[j] += __shfl_xor(s2[j], 16); s2[j] += __shfl_xor(s2[j], 32);
        }
        if (lg == 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                sstat[(wave * 2 + 0) * 32 + 2 * lc + j] = s1[j];
                sstat[(wave * 2 + 1) * 32 + 2 * lc + j] = s2[j];
            }
        }
        __syncthreads();
        mm_acc_t* rep = acc_rep(a.stats, (blockIdx.x + blockIdx.y) % MM_ACC_REPL, 2 * COUT);
        if (tid < 2 * 32 * NW) {
            const int which = tid / (32 * NW), nl = tid % (32 * NW);           // nl: channel inside the workgroup's NW groups
            float tsum = 0.f;
#pragma unroll
            for (int k = 0; k < WK; ++k) tsum += sstat[(((nl >> 5) + NW * k) * 2 + which) * 32 + (nl & 31)];
            acc_add<MM_ACC_STAT>(&rep[which * COUT + blockIdx.y * NW * 32 + nl], tsum);
        }
    }
}

template <int CIN, int COUT, int NW_, int DEPTH_>
int launch_stream(const Conv3dArgs& a, hipStream_t st) {
    using C = StreamCfg<CIN, COUT, NW_, DEPTH_>;
    auto kern = conv3d_stream_kernel<CIN, COUT, NW_, DEPTH_>;
    if (C::LDS > 64 * 1024) {
        static const hipError_t attr =                              // the only process-wide state: an immutable kernel attribute
            hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
        (void)attr;
    }
    const int ntiles = a.B * ceil_div(a.D, STD) * ceil_div(a.H, 8) * ceil_div(a.W, 8);
    hipLaunchKernelGGL(kern, dim3(ntiles, COUT / 32 / NW_), dim3(256), C::LDS, st, a);
    return mm_check_launch("conv3d_stream");
}

}  // namespace

bool conv3d_stream_applies(const Conv3dArgs& a) {
    // offsets inside a sample are 32-bit
    return conv3d_stream_shape(a.Cout, a.Cin) && (size_t)a.D * a.H * a.W * a.Cin < (1u << 31) &&
           (size_t)a.D * a.H * a.W * a.Cout < (1u << 31);
}

int launch3d_stream(const Conv3dArgs& a, hipStream_t st) {
    // few tiles (8^3 volumes: 128): split the columns over workgroups so that every CU has one
    const int ntiles = a.B * ceil_div(a.D, STD) * ceil_div(a.H, 8) * ceil_div(a.W, 8);
    if (a.Cin == 64 && a.Cout == 128) return ntiles >= 256 ? launch_stream<64, 128, 4, 6>(a, st) : launch_stream<64, 128, 2, 6>(a, st);
    if (a.Cin == 128 && a.Cout == 64) return ntiles >= 256 ? launch_stream<128, 64, 2, 6>(a, st) : launch_stream<128, 64, 1, 6>(a, st);
    if (a.Cin == 64 && a.Cout == 32) return launch_stream<64, 32, 1, 6>(a, st);
    return mm_fail(MM_ERR_UNSUPPORTED, "conv3d_stream: Cin %d Cout %d", a.Cin, a.Cout);
}
