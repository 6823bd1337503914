// Weight-streaming 3-D convolution (k = 3, pad 1) for the channel-heavy layers of the voxel encoder:
// layer 3 forward (64 -> 128), its data gradient (128 -> 64) and layer 2's data gradient (64 -> 32).
// bf16 MFMA 16x16x32, fp32 accumulate, channels-last in and out.
//
//   Y[b, v, n] = bias[n] + sum_{tap, c} X[b, v + off(tap), c] * W[n, tap, c]
//
// Here the weights (27 * Cin * Cout * 2 B = 110-442 KB) do not fit beside the halo, or there are too few
// output tiles to amortise loading them (8^3 volumes: one tile per CU).  One workgroup owns ONE 1 x 8 x 8
// face of output voxels (64 GEMM rows) x all Cout columns and stages the 3 x 10 x 10 x Cin halo in LDS once;
// the weights never touch LDS: every wave loads its B operands (its column group, its K-group's 32-channel
// slice of a tap = one "unit" = one MFMA K-step) straight from L2 into registers, DEPTH units ahead of the
// MFMAs, in exactly the lane layout the MFMA wants (lane (c, g) <- 16 bytes of weight row n(c), channels
// 8 g .. 8 g + 7); mm_prep_conv_weight writes the image of these shapes in that order (common.h:
// conv_image_index), so a wave load is 1 KB contiguous - from the plain [n][tap][c] image the same load touched
// 16 cache lines and ran at a quarter of the rate.  Per CU the kernel needs 64 B of weights per clock at full MFMA rate - the vector-memory
// path's peak - so an LDS hop on top (LDS-DMA ring, first version: 384 LDS cycles per 256 MFMA cycles, measured
// 504) is what it cannot afford; A fragments alone keep the LDS pipe half busy.
//
//  * wave tile = 64 rows x 64 columns (4 x 4 MFMA tiles: 4 A reads from LDS + 4 B loads from L2 per 16 MFMAs),
//    64 x 32 for Cout = 32.  The four waves split the column groups first and K second (WK = 4 / WN K-groups:
//    unit g of the 27 * Cin / 32 belongs to K-group g % WK); K-groups are summed through LDS (the dead halo)
//    after the loop, each wave keeping 4 / WK row tiles for the epilogue.  No barrier inside the K loop.
//  * the halo image is that of conv3d_wres.hip per 32-channel slice: unpadded 64-byte rows, 16-byte slots
//    XOR-swizzled with 2 * (patch-row parity) - conflict-free ds_read_b128 for every tap shift - and
//    MFMA row m of row tile i <-> voxel (h, w) = (4 (i >> 1) + (m >> 2), 4 (i & 1) + (m & 3)).
//    Column c of column tile j of group wn is channel WTN wn + TJ c + j: a lane holds TJ adjacent channels
//    (one 16-byte fp32 / 8-byte bf16 store per voxel).
//  * the A-fragment reads are inline asm with explicit lgkmcnt waits so that they stay where the software
//    pipeline puts them (one unit ahead); the B loads are ordinary loads, the compiler counts vmcnt.
#include "conv3d_args.h"

#ifndef STREAM_ABL      // diagnostic builds (tools/abl_stream.sh): 1 no B loads in the loop, 2 no MFMAs, 4 no fragment reads
#define STREAM_ABL 0
#endif

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int SHB = 10;                 // halo edge of an 8-wide face
constexpr int SWP = 12;                 // halo w-pitch in LDS rows
constexpr int SDP = SHB * SWP;          // rows per halo plane
constexpr int SROWB = 64;               // bytes per LDS row (32 channels)
constexpr int HKS = 3 * SDP * SROWB;    // bytes of one 32-channel halo image (23 040)

template <int CIN, int COUT, int DEPTH_>
struct StreamCfg {
    static constexpr int KS = CIN / 32;                 // K-steps (units) per tap
    static constexpr int WTN = COUT >= 64 ? 64 : 32;    // wave tile columns
    static constexpr int TJ = WTN / 16;                 // MFMA column tiles per wave
    static constexpr int WN = COUT / WTN;               // column groups = waves along N
    static constexpr int WK = 4 / WN;                   // K-groups
    static constexpr int NU = 27 * KS;                  // units in all
    static constexpr int NUW = (NU + WK - 1) / WK;      // units per wave
    static constexpr int DEPTH = DEPTH_;                // B operands in flight, in units
    static constexpr int H_BYTES = KS * HKS;
    static constexpr int PARK = WK > 1 ? 4 * 4 * TJ * 1024 : 0;     // [wave][i][j][lane] x 16 B, overlays the halo
    static constexpr int S_OFF = H_BYTES > PARK ? H_BYTES : PARK;   // [4 waves][2][WTN] fp32 BatchNorm partials
    static constexpr int LDS = S_OFF + 4 * 2 * WTN * 4;
    static_assert(WN >= 1 && WN <= 4 && WN * WK == 4 && DEPTH <= NUW, "shape");
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

template <int CIN, int COUT, int DEPTH_>
__global__ __launch_bounds__(256) void conv3d_stream_kernel(Conv3dArgs a) {
    using C = StreamCfg<CIN, COUT, DEPTH_>;
    constexpr int KS = C::KS, WN = C::WN, WK = C::WK, NU = C::NU, NUW = C::NUW, DEPTH = C::DEPTH, WTN = C::WTN, TJ = C::TJ;
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
    const int wn = wave % WN, kg = wave / WN;
    const int lc = lane & 15, lg = lane >> 4;
    const int lds0 = (int)(size_t)(__attribute__((address_space(3))) char*)smem;

    // ---- B operands: unit u of this wave is global unit g = u WK + kg = (tap, ks); the image (common.h:
    // conv_image_index) holds, per unit and column group, TJ x 1 KB in lane order: column tile j of lane (lc, lg) is
    // weight row n0 + j, channels 32 ks + 8 lg .. + 7 of the tap
    const int n0 = WTN * wn + TJ * lc;
    const bf16* wlane = a.w + (wn * TJ * 64 + lane) * 8;
    auto load_b = [&](int u, u32x4 (&dst)[TJ]) __attribute__((always_inline)) {
        int g = u * WK + kg;
        g = g < NU ? g : NU - 1;                               // a K-group without a last unit re-reads a valid one (unused)
        const bf16* p = wlane + g * (COUT * 32);
#pragma unroll
        for (int j = 0; j < TJ; ++j) dst[j] = *reinterpret_cast<const u32x4*>(p + j * 512);
    };
    u32x4 bq[DEPTH][TJ];
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) load_b(u, bq[u]);
    float sh[TJ];
#pragma unroll
    for (int j = 0; j < TJ; ++j) sh[j] = a.shift ? a.shift[n0 + j] : 0.f;

    // ---- tile: XCD x (= blockIdx % 8 under round-robin dispatch) takes the x-th eighth of the (b, d, h, w)-ordered list
    const int tw = (a.W + 7) / 8, th = (a.H + 7) / 8;
    int t = blockIdx.x;
    if ((gridDim.x & 7) == 0) t = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int w0 = (t % tw) * 8; t /= tw;
    const int h0 = (t % th) * 8; t /= th;
    const int d0 = t % a.D;
    const int b = t / a.D;

    // ---- halo: 3 x 10 x 10 voxels x CIN channels as 16-byte chunks, all loads of a batch in flight before its LDS writes
    {
        constexpr int CPR = CIN / 8;                           // chunks per voxel row
        constexpr int NCH = 3 * SHB * SHB * CPR;
        constexpr int PER = (NCH + 255) / 256;
        constexpr int BATCH = PER > 10 ? (PER + 1) / 2 : PER;
        const bf16* xb = a.x + (size_t)b * a.D * a.H * a.W * CIN;
        for (int c0 = 0; c0 < PER; c0 += BATCH) {
            u32x4 v[BATCH];
#pragma unroll
            for (int q = 0; q < BATCH; ++q) {
                const int c = (c0 + q) * 256 + tid;
                const int r = c / CPR, seg = c % CPR;
                const int hw = r % SHB, hh = (r / SHB) % SHB, hd = r / (SHB * SHB);
                const int d = d0 + hd - 1, h = h0 + hh - 1, w = w0 + hw - 1;
                v[q] = u32x4{0u, 0u, 0u, 0u};
                if (c0 + q < PER && c < NCH && d >= 0 && d < a.D && h >= 0 && h < a.H && w >= 0 && w < a.W)
                    v[q] = *reinterpret_cast<const u32x4*>(xb + ((d * a.H + h) * a.W + w) * CIN + seg * 8);
            }
#pragma unroll
            for (int q = 0; q < BATCH; ++q) {
                const int c = (c0 + q) * 256 + tid;
                const int r = c / CPR, seg = c % CPR;
                const int hw = r % SHB, hh = (r / SHB) % SHB, hd = r / (SHB * SHB);
                if (c0 + q < PER && c < NCH)
                    *reinterpret_cast<u32x4*>(smem + (seg >> 2) * HKS + ((hd * SHB + hh) * SWP + hw) * SROWB +
                                              (((seg & 3) ^ (2 * (hh & 1))) << 4)) = v[q];
            }
        }
    }

    // ---- per-lane A-fragment bases (absolute LDS byte addresses)
    const int arow = ((lc >> 2) * SWP + (lc & 3)) * SROWB;
    const int lane_a0 = lds0 + arow + ((lg ^ (2 * (((lc >> 2) + 0) & 1))) << 4);     // taps with kh even
    const int lane_a1 = lds0 + arow + ((lg ^ (2 * (((lc >> 2) + 1) & 1))) << 4);     // kh odd

    f32x4 acc[4][TJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // unit u of this wave -> (tap, ks); the tap is a compile-time constant whenever WK divides KS
    auto a_addr = [&](int u, int i) __attribute__((always_inline)) {
        const int g = u * WK + kg;
        const int tap = g / KS, ks = g % KS;
        const int kh = (tap / 3) % 3;
        return ((kh & 1) ? lane_a1 : lane_a0) + ks * HKS + tap_off(tap) + ((4 * (i >> 1)) * SWP + 4 * (i & 1)) * SROWB;
    };

    ST_TL(0)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                              // the halo (all waves' writes) is in LDS
    ST_TL(1)

    u32x4 fa[4];                                               // A fragments of the unit about to run
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = lds_read128(a_addr(0, i));

#pragma unroll
    for (int u = 0; u < NUW; ++u) {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]));   // requested one unit ago
        const bool valid = (u * WK + WK <= NU) ? true : (u * WK + kg < NU);
        u32x4 ca[4], cb[TJ];
#pragma unroll
        for (int i = 0; i < 4; ++i) ca[i] = fa[i];
#pragma unroll
        for (int j = 0; j < TJ; ++j) cb[j] = bq[u % DEPTH][j];
        if (u + 1 < NUW && !(STREAM_ABL & 4)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = lds_read128(a_addr(u + 1, i));
        }
        if (u + DEPTH < NUW && !(STREAM_ABL & 1)) load_b(u + DEPTH, bq[u % DEPTH]);
        __builtin_amdgcn_sched_barrier(0);
        if (valid && !(STREAM_ABL & 2)) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ca[i]), __builtin_bit_cast(bf16x8, cb[j]),
                                                                         acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    ST_TL(2)
    // ---- K-groups: every wave parks its 64 x 32 partial tile, wave (wn, kg) then owns row tiles i = kg (4 / WK) ..
    constexpr int OWN = 4 / WK;                                // row tiles per wave after the reduction
    f32x4 res[OWN][TJ];
    if constexpr (WK == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) res[i][j] = acc[i][j];
    } else {
        __syncthreads();                                       // every wave is done with the halo
        f32x4* park = reinterpret_cast<f32x4*>(smem);          // [wave][i][j][lane]
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) park[((wave * 4 + i) * TJ + j) * 64 + lane] = acc[i][j];
        __syncthreads();
#pragma unroll
        for (int o = 0; o < OWN; ++o)
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                f32x4 t4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < WK; ++k) t4 += park[(((wn + WN * k) * 4 + kg * OWN + o) * TJ + j) * 64 + lane];   // fixed order
                res[o][j] = t4;
            }
    }

    ST_TL(3)
    // ---- epilogue: register r of lane (lc, lg) in row tile i is voxel (h, w) = (4 (i >> 1) + lg, 4 (i & 1) + r), channels
    // n0 .. n0 + TJ - 1
    float s1[TJ], s2[TJ];
#pragma unroll
    for (int j = 0; j < TJ; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    const size_t obase = ((((size_t)b * a.D + d0) * a.H + h0) * a.W + w0) * COUT + n0;
    float* of = a.out_f32 ? a.out_f32 + obase : nullptr;
    bf16* ob = a.out_bf16 ? a.out_bf16 + obase : nullptr;
#pragma unroll
    for (int o = 0; o < OWN; ++o) {
        const int i = kg * OWN + o;
        const int hh = 4 * (i >> 1) + lg;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ww = 4 * (i & 1) + r;
            if (h0 + hh < a.H && w0 + ww < a.W) {
                float v[TJ];
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    v[j] = res[o][j][r] + sh[j];
                    s1[j] += v[j]; s2[j] += v[j] * v[j];
                }
                const int off = (hh * a.W + ww) * COUT;
                if constexpr (TJ == 4) {
                    if (of) *reinterpret_cast<f32x4*>(of + off) = f32x4{v[0], v[1], v[2], v[3]};
                    if (ob) *reinterpret_cast<bf16x4*>(ob + off) = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                } else {
                    if (of) *reinterpret_cast<float2*>(of + off) = float2{v[0], v[1]};
                    if (ob) *reinterpret_cast<bf16x2*>(ob + off) = bf16x2{(bf16)v[0], (bf16)v[1]};
                }
            }
        }
    }
    ST_TL(4)
#ifdef STREAM_STAMPS
    if (a.stats && tid == 0) {
        float* o = a.stats + MM_REPL * 2 * COUT + blockIdx.x * 8;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        o[0] = tl[0]; o[1] = tl[1]; o[2] = tl[2]; o[3] = tl[3]; o[4] = tl[4];
        o[5] = (float)(__builtin_readcyclecounter() - t_begin);
        o[6] = (float)(r_begin & 0xFFFFFF); o[7] = (float)(wall_clock64() & 0xFFFFFF);
    }
#endif
    if (a.stats) {
        float* sstat = reinterpret_cast<float*>(smem + C::S_OFF);       // [wave][2][WTN]
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            s1[j] += __shfl_xor(s1[j], 16); s1[j] += __shfl_xor(s1[j], 32);
            s2[j] += __shfl_xor(s2[j], 16); s2[j] += __shfl_xor(s2[j], 32);
        }
        if (lg == 0) {
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                sstat[(wave * 2 + 0) * WTN + TJ * lc + j] = s1[j];
                sstat[(wave * 2 + 1) * WTN + TJ * lc + j] = s2[j];
            }
        }
        __syncthreads();
        float* rep = a.stats + (size_t)(blockIdx.x % MM_REPL) * 2 * COUT;
        if (tid < 2 * COUT) {
            const int which = tid / COUT, n = tid % COUT;
            float tsum = 0.f;
#pragma unroll
            for (int k = 0; k < WK; ++k) tsum += sstat[(((n / WTN) + WN * k) * 2 + which) * WTN + (n % WTN)];
            atomicAdd(&rep[which * COUT + n], tsum);
        }
    }
}

template <int CIN, int COUT, int DEPTH_>
int launch_stream(const Conv3dArgs& a, hipStream_t st) {
    using C = StreamCfg<CIN, COUT, DEPTH_>;
    auto kern = conv3d_stream_kernel<CIN, COUT, DEPTH_>;
    if (C::LDS > 64 * 1024) {
        static const hipError_t attr =                              // the only process-wide state: an immutable kernel attribute
            hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
        (void)attr;
    }
    const int ntiles = a.B * a.D * ceil_div(a.H, 8) * ceil_div(a.W, 8);
    hipLaunchKernelGGL(kern, dim3(ntiles), dim3(256), C::LDS, st, a);
    return mm_check_launch("conv3d_stream");
}

}  // namespace

bool conv3d_stream_applies(const Conv3dArgs& a) {
    // offsets inside a sample are 32-bit
    return ((a.Cin == 64 && a.Cout == 128) || (a.Cin == 128 && a.Cout == 64) || (a.Cin == 64 && a.Cout == 32)) &&
           (size_t)a.D * a.H * a.W * a.Cin < (1u << 31) && (size_t)a.D * a.H * a.W * a.Cout < (1u << 31);
}

int launch3d_stream(const Conv3dArgs& a, hipStream_t st) {
    if (a.Cin == 64 && a.Cout == 128) return launch_stream<64, 128, 4>(a, st);
    if (a.Cin == 128 && a.Cout == 64) return launch_stream<128, 64, 4>(a, st);
    if (a.Cin == 64 && a.Cout == 32) return launch_stream<64, 32, 7>(a, st);
    return mm_fail(MM_ERR_UNSUPPORTED, "conv3d_stream: Cin %d Cout %d", a.Cin, a.Cout);
}
