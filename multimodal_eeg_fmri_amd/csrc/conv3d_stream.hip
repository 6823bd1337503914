// Weight-streaming 3-D convolution (k = 3, pad 1) for the channel-heavy layers of the voxel encoder:
// layer 3 forward (64 -> 128), its data gradient (128 -> 64) and layer 2's data gradient (64 -> 32).
// bf16 MFMA 16x16x32, fp32 accumulate, channels-last in and out.
//
//   Y[b, v, n] = bias[n] + sum_{tap, c} X[b, v + off(tap), c] * W[n, tap, c]
//
// Here the weights (27 * Cin * Cout * 2 B = 110-442 KB) do not fit beside the halo, or there are too few
// output tiles to amortise loading them (8^3 volumes: one tile per CU).  So one workgroup owns ONE 1 x 8 x 8
// face of output voxels (64 GEMM rows) x all Cout columns, stages the 3 x 10 x 10 x Cin halo once, and the
// weights stream through a ring of NST stage buffers by LDS-DMA (global_load_lds_dwordx4: no registers, no
// ds_write), NST - 2 stages ahead of the MFMAs.  A "unit" is one tap x 32 input channels (one MFMA K-step);
// a stage is U units.
//
//  * wave tile = 64 rows x 64 columns (4 x 4 MFMA tiles: 8 fragment reads per 16 MFMAs - the LDS pipe is then
//    exactly as busy as the MFMA pipes; a 64 x 32 wave tile measured LDS-bound) or 64 x 32 for Cout = 32.
//    The four waves split the column groups first and the units of a stage (K) second (WK = 4 / WN K-groups:
//    2 for 64 -> 128, 4 for the two data gradients); K-groups are summed through LDS (the dead ring) after the
//    loop, each wave keeping 4 / WK row tiles for the epilogue.
//  * LDS images are those of conv3d_wres.hip per 32-channel slice: unpadded 64-byte rows, 16-byte slots
//    XOR-swizzled with 2 * (patch-row parity) - conflict-free ds_read_b128 for every tap shift - and
//    MFMA row m of row tile i <-> voxel (h, w) = (4 (i >> 1) + (m >> 2), 4 (i & 1) + (m & 3)).
//    Column c of column tile j of group wn is channel WTN wn + TJ c + j: a lane holds TJ adjacent channels.
//  * no barrier in the K loop: a wave's DMA share of a stage is exactly the slice it reads itself (its K-group's
//    unit, its column group's rows), so the ring is private to the wave - counted vmcnt before the fragment
//    reads of the next stage, and a buffer is refilled right after its fragments have arrived in registers.
//    The waves drift apart and fill each other's LDS / MFMA gaps (with a barrier per stage all four issued their
//    reads at the same moment: 450 cycles per stage against 256 of MFMA work).  The fragment reads are inline
//    asm with explicit lgkmcnt waits: the compiler would otherwise order every LDS read behind ALL outstanding
//    LDS-DMA (it cannot tell the ring buffers apart) and serialise the pipeline.
#include "conv3d_args.h"

#include <mutex>

#ifndef STREAM_ABL      // diagnostic builds (tools/abl_stream.sh): 1 no DMA in the loop, 2 no MFMAs, 4 no fragment reads
#define STREAM_ABL 0
#endif

namespace {

typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int SHB = 10;                 // halo edge of an 8-wide face
constexpr int SWP = 12;                 // halo w-pitch in LDS rows
constexpr int SDP = SHB * SWP;          // rows per halo plane
constexpr int SROWB = 64;               // bytes per LDS row (32 channels)
constexpr int HKS = 3 * SDP * SROWB;    // bytes of one 32-channel halo image (23 040)

template <int CIN, int COUT, int NST_>
struct StreamCfg {
    static constexpr int KS = CIN / 32;                 // K-steps (units) per tap
    static constexpr int WTN = COUT >= 64 ? 64 : 32;    // wave tile columns
    static constexpr int TJ = WTN / 16;                 // MFMA column tiles per wave
    static constexpr int WN = COUT / WTN;               // column groups = waves along N
    static constexpr int WK = 4 / WN;                   // K-groups
    static constexpr int U = KS > WK ? KS : WK;         // units per stage
    static constexpr int UPG = U / WK;                  // units per K-group per stage
    static constexpr int NU = 27 * KS;                  // units in all
    static constexpr int NS = (NU + U - 1) / U;         // stages
    static constexpr int NST = NST_;                    // ring depth
    static constexpr int UNITB = COUT * SROWB;          // bytes of a unit's weight image
    static constexpr int STAGEB = U * UNITB;
    static constexpr int DPW = STAGEB / 4096;           // DMA instructions (1 KB each) per wave per stage
    static constexpr int H_BYTES = KS * HKS;
    static constexpr int R_OFF = H_BYTES;
    static constexpr int S_OFF = R_OFF + NST * STAGEB;  // [4 waves][2][WTN] fp32 BatchNorm partials
    static constexpr int LDS = S_OFF + 4 * 2 * WTN * 4;
    static_assert(WN >= 1 && WN <= 4 && WN * WK == 4 && U % WK == 0 && STAGEB % 4096 == 0, "shape");
    static_assert(NST * STAGEB >= 4 * TJ * 4096 || WK == 1, "the ring doubles as the K-group reduction buffer");
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

template <int CIN, int COUT, int NST_>
__global__ __launch_bounds__(256) void conv3d_stream_kernel(Conv3dArgs a) {
    using C = StreamCfg<CIN, COUT, NST_>;
    constexpr int KS = C::KS, WN = C::WN, WK = C::WK, U = C::U, UPG = C::UPG, NU = C::NU, NS = C::NS, NST = C::NST;
    constexpr int UNITB = C::UNITB, STAGEB = C::STAGEB, DPW = C::DPW, WTN = C::WTN, TJ = C::TJ;
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

    // ---- weight DMA: every wave fetches exactly the slice it reads itself - unit(s) kg UPG + e / TJ of a stage, rows
    // rho = WTN wn + 16 (e % TJ) + (lane >> 2), slot lane & 3 - so the ring needs no barrier: a wave orders its own
    // DMA (counted vmcnt) against its own fragment reads.  Row rho = WTN wn + 16 j + c holds channel WTN wn + TJ c + j;
    // slot s holds channel segment s ^ 2 ((rho >> 2) & 1).  Both permutations sit in the source address.
    static_assert(DPW == UPG * TJ, "a wave's DMA share is its own slice");
    int wsrc[TJ];
#pragma unroll
    for (int e = 0; e < TJ; ++e) {
        const int rho = wn * WTN + e * 16 + (lane >> 2), slot = lane & 3;
        const int n = (rho / WTN) * WTN + TJ * (rho & 15) + ((rho % WTN) >> 4);
        wsrc[e] = n * 27 * CIN + ((slot ^ (2 * ((rho >> 2) & 1))) << 3);
    }
    auto dma_stage = [&](int s) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < UPG; ++u) {
            int g = s * U + kg * UPG + u;
            g = g < NU ? g : NU - 1;                          // a short last stage still issues DPW loads (counted vmcnt)
            const int tap = g / KS, ks = g % KS;
            char* dst = smem + C::R_OFF + (s % NST) * STAGEB + (kg * UPG + u) * UNITB + wn * WTN * SROWB;
#pragma unroll
            for (int e = 0; e < TJ; ++e)
                __builtin_amdgcn_global_load_lds((gptr_t*)(a.w + wsrc[e] + tap * CIN + ks * 32), (lptr_t*)(dst + e * 1024), 16, 0, 0);
        }
    };
#pragma unroll
    for (int s = 0; s < NST; ++s)
        if (s < NS) dma_stage(s);

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
                    v[q] = *reinterpret_cast<const u32x4*>(xb + (((size_t)d * a.H + h) * a.W + w) * CIN + seg * 8);
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

    // ---- per-lane fragment bases (absolute LDS byte addresses)
    const int arow = ((lc >> 2) * SWP + (lc & 3)) * SROWB;
    const int lane_a0 = lds0 + arow + ((lg ^ (2 * (((lc >> 2) + 0) & 1))) << 4);     // taps with kh even
    const int lane_a1 = lds0 + arow + ((lg ^ (2 * (((lc >> 2) + 1) & 1))) << 4);     // kh odd
    const int lane_b = lds0 + C::R_OFF + (wn * WTN + lc) * SROWB + ((lg ^ (2 * ((lc >> 2) & 1))) << 4);

    f32x4 acc[4][TJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // unit (s, e) of this wave's K-group: g = s U + kg UPG + e -> (tap, ks).  Compile-time when WK == 1.
    auto a_addr = [&](int s, int e, int i) __attribute__((always_inline)) {
        const int g = s * U + kg * UPG + e;
        const int tap = g / KS, ks = g % KS;
        const int kh = (tap / 3) % 3;
        return ((kh & 1) ? lane_a1 : lane_a0) + ks * HKS + tap_off(tap) + ((4 * (i >> 1)) * SWP + 4 * (i & 1)) * SROWB;
    };
    auto b_addr = [&](int s, int e, int j) __attribute__((always_inline)) {
        return lane_b + (s % NST) * STAGEB + (kg * UPG + e) * UNITB + j * 16 * SROWB;
    };
    auto unit_valid = [&](int s, int e) __attribute__((always_inline)) { return s * U + kg * UPG + e < NU; };

    ST_TL(0)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                              // the halo (all waves' writes) and this wave's stages 0 .. NST - 1 are in LDS

    ST_TL(1)
    u32x4 fa[4], fb[TJ];                                        // fragments of the unit about to run
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = lds_read128(a_addr(0, 0, i));
#pragma unroll
    for (int j = 0; j < TJ; ++j) fb[j] = lds_read128(b_addr(0, 0, j));

#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
        for (int e = 0; e < UPG; ++e) {
            // the fragments of (s, e) were requested one unit ago
            if constexpr (TJ == 4)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3]));
            else
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fb[0]), "+v"(fb[1]));
            const bool valid = (s * U + U <= NU) ? true : unit_valid(s, e);
            u32x4 ca[4], cb[TJ];
#pragma unroll
            for (int i = 0; i < 4; ++i) ca[i] = fa[i];
#pragma unroll
            for (int j = 0; j < TJ; ++j) cb[j] = fb[j];
            if (e == 0 && s > 0 && s + NST - 1 < NS && !(STREAM_ABL & 1)) dma_stage(s + NST - 1);   // into stage s - 1's buffer: its reads are done
            const int s2 = e + 1 < UPG ? s : s + 1, e2 = e + 1 < UPG ? e + 1 : 0;
            if (s2 < NS && !(STREAM_ABL & 4)) {
                if (e2 == 0) {
                    // this wave's stage s + 1 has landed; stages s + 2 .. s + NST - 1 may still be in flight
                    const int later = (NS - 2 - s) < (NST - 2) ? (NS - 2 - s) : (NST - 2);
                    if (later * DPW == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                    else if (later * DPW == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else if (later * DPW == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                    else if (later * DPW == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    else if (later * DPW == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                    else if (later * DPW == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                // request the next unit's fragments: they overlap this unit's MFMAs
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[i] = lds_read128(a_addr(s2, e2, i));
#pragma unroll
                for (int j = 0; j < TJ; ++j) fb[j] = lds_read128(b_addr(s2, e2, j));
            }
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
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                          // every wave is done with the ring
        f32x4* park = reinterpret_cast<f32x4*>(smem + C::R_OFF);    // [wave][i][j][lane]
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
    // n0 = WTN wn + TJ lc .. n0 + TJ - 1
    const int n0 = WTN * wn + TJ * lc;
    float sh[TJ], s1[TJ], s2[TJ];
#pragma unroll
    for (int j = 0; j < TJ; ++j) { sh[j] = a.shift ? a.shift[n0 + j] : 0.f; s1[j] = 0.f; s2[j] = 0.f; }
    const size_t plane = ((size_t)b * a.D + d0) * a.H;
#pragma unroll
    for (int o = 0; o < OWN; ++o) {
        const int i = kg * OWN + o;
        const int h = h0 + 4 * (i >> 1) + lg;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int w = w0 + 4 * (i & 1) + r;
            if (h < a.H && w < a.W) {
                float v[TJ];
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    v[j] = res[o][j][r] + sh[j];
                    s1[j] += v[j]; s2[j] += v[j] * v[j];
                }
                const size_t off = ((plane + h) * a.W + w) * COUT + n0;
                if constexpr (TJ == 4) {
                    if (a.out_f32) *reinterpret_cast<f32x4*>(a.out_f32 + off) = f32x4{v[0], v[1], v[2], v[3]};
                    if (a.out_bf16) *reinterpret_cast<bf16x4*>(a.out_bf16 + off) = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                } else {
                    if (a.out_f32) *reinterpret_cast<float2*>(a.out_f32 + off) = float2{v[0], v[1]};
                    if (a.out_bf16) *reinterpret_cast<bf16x2*>(a.out_bf16 + off) = bf16x2{(bf16)v[0], (bf16)v[1]};
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

template <int CIN, int COUT, int NST_>
int launch_stream(const Conv3dArgs& a, hipStream_t st) {
    using C = StreamCfg<CIN, COUT, NST_>;
    auto kern = conv3d_stream_kernel<CIN, COUT, NST_>;
    static std::once_flag once;
    std::call_once(once, [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
    });
    const int ntiles = a.B * a.D * ceil_div(a.H, 8) * ceil_div(a.W, 8);
    hipLaunchKernelGGL(kern, dim3(ntiles), dim3(256), C::LDS, st, a);
    return mm_check_launch("conv3d_stream");
}

}  // namespace

bool conv3d_stream_applies(const Conv3dArgs& a) {
    return (a.Cin == 64 && a.Cout == 128) || (a.Cin == 128 && a.Cout == 64) || (a.Cin == 64 && a.Cout == 32);
}

int launch3d_stream(const Conv3dArgs& a, hipStream_t st) {
    if (a.Cin == 64 && a.Cout == 128) return launch_stream<64, 128, 5>(a, st);
    if (a.Cin == 128 && a.Cout == 64) return launch_stream<128, 64, 4>(a, st);
    if (a.Cin == 64 && a.Cout == 32) return launch_stream<64, 32, 4>(a, st);
    return mm_fail(MM_ERR_UNSUPPORTED, "conv3d_stream: Cin %d Cout %d", a.Cin, a.Cout);
}
