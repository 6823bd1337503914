// Weight gradient of the 3-D voxel convolution (k = 3, pad 1), bf16 MFMA 32x32x16, fp32 accumulate:
//
//   dW[n, tap, c] = sum_{b, v} dY[b, v, n] * X[b, v + off(tap), c]       (GEMM M = Cout, N = 27 Cin, K = voxels)
//
// Both operands are K-strided in memory (channels-last), so the dY tile [64 voxels][64 n] and the X halo plane
// [10 x 10 voxels][64 c] are staged row-major and read with ds_read_b64_tr_b16 (hardware transpose); the row stride
// of 192 B puts the four rows a lane group touches on distinct banks (PMC: 0 conflicts).  grid.z = kd plane x
// 64-channel block of Cin, grid.y = 64-channel block of Cout, grid.x = chunk of 1 x 8 x 8 output tiles: a workgroup
// accumulates the 9 taps of its kd plane for a 64 (n) x 64 (c) block (wave tile 32 x 32 per tap, 9 x 16 accumulator
// registers) over its tiles and leaves them in slot blockIdx.x (plain stores; mm_wgrad_scatter sums the slots) or
// adds them atomically.
//
// Round-2 rewrite.  The round-1 kernel prefetched the next tile into registers; its rolled tile loop made the
// compiler drain vmcnt to 0 at the top of every iteration, which left one tile (0.5 us of MFMA work) of cover for a
// ~2 us global round trip: 125 k cycles per workgroup of which 41 k in the MFMA phases.  Now
//  * the tiles arrive by LDS-DMA (buffer_load_dwordx4 ... lds: no registers, no ds_write, out-of-volume voxels
//    zero-filled by the buffer's range check) into a ring of W3_ST stages, W3_ST - 1 tiles ahead of the MFMAs, with
//    counted vmcnt and ONE barrier per tile; a DMA writes 1 KB contiguous per wave, so a lane's chunk is chosen on
//    the source side: chunk ci = 64 E + lane of a stage is row ci / 12, 16-byte column ci % 12 of the padded image
//    (columns 8-11 are padding: those lanes fetch out of range);
//  * inside a tile the 36 (k-step, tap) units are software-pipelined by hand: the B fragment of unit q + W3_LA is
//    requested before the MFMA of unit q, under counted lgkmcnt (left to the compiler the reads sat right in front
//    of their MFMAs).  The reads are inline asm: the compiler cannot tell the ring stages apart and would order
//    every LDS read behind ALL outstanding DMA.
#include "common.h"

#include <type_traits>

#ifndef W3_ABL           // diagnostic builds (ABL_FILE=conv3d_wgrad ABL_MACRO=W3_ABL tools/abl_stream.sh): 1 no MFMA phase, 2 no DMA in the loop
#define W3_ABL 0
#endif

namespace {

constexpr int HB = 10;             // halo edge of an 8-wide tile
constexpr int W3_LD = 96;          // LDS row stride (elements) == 192 B
constexpr int W3_ROWB = W3_LD * 2;
constexpr int W3_ST = 4;           // stages of the LDS ring
constexpr int W3_LA = 6;           // B fragments of look-ahead inside a tile (2 LDS reads each: 14 of lgkmcnt's 15 with an A pair)
constexpr int W3_YB = 64 * W3_ROWB;                 // 12 288 B: dY tile image = 12 DMA instructions
constexpr int W3_XB = 20 * 1024;                    // X plane image: 100 rows x 192 B = 19 200 B inside 20 DMA instructions
constexpr int W3_STAGE = W3_YB + W3_XB;             // 32 KB = 32 DMA instructions, 8 per wave
constexpr int W3_LDS = W3_ST * W3_STAGE;

struct Wgrad3dArgs {
    const bf16* dy; const bf16* x; float* dw; float* dbias;
    int B, D, H, W, Cin, Cout, Cin_real, tiles_per_wg, nrep;
    long sn, sc, stap, rep_stride;
    int slot_mode;            // 1: tile-chunk x stores its partial dW into slot blockIdx.x (no atomics)
};

typedef unsigned long long u64;
typedef __attribute__((address_space(3))) void lptr_t;
union Frag { u64 u[2]; bf16x8 v; };

template <int OFF>
__device__ __forceinline__ u64 tr_read(int addr) {
    u64 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int N>
__device__ __forceinline__ void wait_lgkm(Frag& f) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f.u[0]), "+v"(f.u[1]) : "n"(N));
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
// LDS reads issued after the B fragment of unit q when unit q's MFMA is about to run (units q + 1 .. q + W3_LA,
// two reads each, plus the A pair in front of every ninth unit)
constexpr int reads_after(int q, int nq) {
    int n = 0;
    for (int s = q + 1; s <= q + W3_LA; ++s)
        if (s < nq) n += 2 + ((s % 9 == 0) ? 2 : 0);
    return n;
}

// KSPLIT (Cin <= 32: a second 32-column block of c would be all padding): the waves of a pair split the tile's four
// k-steps instead of the columns and are summed through LDS at the end.
template <bool KSPLIT>
__global__ __launch_bounds__(256) void conv3d_wgrad_kernel(Wgrad3dArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef STREAM_STAMPS       // diagnostic builds (tools/abl_stream.sh s0): cycle stamps into the first floats of the slot
    const long long t_begin = __builtin_readcyclecounter();
    float tl[4] = {0.f, 0.f, 0.f, 0.f};
#define W3_TL(i) tl[i] = (float)(__builtin_readcyclecounter() - t_begin);
#else
#define W3_TL(i)
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wc = KSPLIT ? 0 : wave & 1, kh = KSPLIT ? wave & 1 : 0;
    const int kd = blockIdx.z % 3, cblk = blockIdx.z / 3;
    const int n0 = blockIdx.y * 64, c0 = cblk * 64;
    const int tw = (a.W + 7) / 8, th = (a.H + 7) / 8;
    const int tiles_total = a.B * a.D * th * tw;
    const int tbeg = blockIdx.x * a.tiles_per_wg;
    const int tend = min(tiles_total, tbeg + a.tiles_per_wg);
    const int li = lane & 15, g = lane >> 4;
    const int lds0 = (int)(size_t)(__attribute__((address_space(3))) char*)smem;

    // ---- DMA plan: instruction E = 8 wave + e of a stage moves chunks ci = 64 E + lane; E < 12 is the dY image
    // (row = voxel (row >> 3, row & 7) of the tile), the rest the X plane image (row = halo voxel (row / 10, row % 10)).
    // Per lane and instruction: the byte offset relative to the tile's first voxel and the voxel's tile coordinates.
    const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16*>(a.dy), 0, (unsigned)((size_t)a.B * a.D * a.H * a.W * a.Cout * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16*>(a.x), 0, (unsigned)((size_t)a.B * a.D * a.H * a.W * a.Cin * 2), 0x00020000);
    // Per lane and instruction: byte offset relative to the tile's first voxel (-1: padding chunk or a channel past the
    // tensor) and a one-hot (row, column) selector of the voxel's tile coordinates, tested per tile against the
    // wave-uniform mask of coordinates that lie inside the volume.
    int rel[8];
    unsigned sel[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int E = wave * 8 + e;
        const bool isy = E < 12;                                 // wave-uniform
        const int ci = 64 * E + lane - (isy ? 0 : 768);
        const int row = ci / 12, ch = ci - 12 * row;
        int vh, vw;
        bool ok;
        if (isy) {
            vh = row >> 3; vw = row & 7;
            rel[e] = ((vh * a.W + vw) * a.Cout + 8 * ch) * 2;
            ok = ch < 8 && n0 + 8 * ch < a.Cout;
        } else {
            vh = row / HB; vw = row - HB * vh;
            rel[e] = ((vh * a.W + vw) * a.Cin + 8 * ch) * 2;
            ok = ch < 8 && row < HB * HB && c0 + 8 * ch < a.Cin;
        }
        sel[e] = ok ? (1u << vh) | (1u << (16 + vw)) : 0x80000000u;      // bit 31 is never allowed
    }
    auto range_bits = [](int lo, int hi, int n) __attribute__((always_inline)) {                // bits [max(lo, 0), min(hi, n))
        lo = lo < 0 ? 0 : lo;
        hi = hi > n ? n : hi;
        return hi > lo ? ((1u << hi) - 1u) & ~((1u << lo) - 1u) : 0u;
    };
    // the DMA cursor walks the chunk's tiles in order: coordinates by carry, not by division
    int cw0, ch0, cd, cb;
    {
        int q = tbeg;
        cw0 = (q % tw) * 8; q /= tw;
        ch0 = (q % th) * 8; q /= th;
        cd = q % a.D; cb = q / a.D;
    }
    int ctile = tbeg;
    auto issue = [&]() __attribute__((always_inline)) {
        // always eight loads per wave (counted vmcnt): a tile past the chunk's end fetches out of range = zeros
        const bool live = ctile < tend;
        const int dd = cd + kd - 1;
        const int ybase = ((((cb * a.D + cd) * a.H + ch0) * a.W + cw0) * a.Cout + n0) * 2;
        const int xbase = ((((cb * a.D + dd) * a.H + ch0 - 1) * a.W + cw0 - 1) * a.Cin + c0) * 2;
        const unsigned ymask = live ? range_bits(0, a.H - ch0, 8) | (range_bits(0, a.W - cw0, 8) << 16) : 0u;
        const unsigned xmask = (live && dd >= 0 && dd < a.D) ? range_bits(1 - ch0, a.H + 1 - ch0, HB) | (range_bits(1 - cw0, a.W + 1 - cw0, HB) << 16) : 0u;
        char* stage = smem + (ctile % W3_ST) * W3_STAGE + wave * 8 * 1024;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool isy = wave * 8 + e < 12;
            const unsigned allowed = __builtin_amdgcn_readfirstlane(isy ? ymask : xmask);
            const int base = __builtin_amdgcn_readfirstlane(isy ? ybase : xbase);
            const int off = (sel[e] & ~allowed) == 0u ? base + rel[e] : -1;
            if (isy) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_y, (lptr_t*)(stage + e * 1024), 16, off, 0, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lptr_t*)(stage + e * 1024), 16, off, 0, 0, 0);
        }
        ++ctile;
        cw0 += 8;
        if (cw0 >= tw * 8) { cw0 = 0; ch0 += 8; if (ch0 >= th * 8) { ch0 = 0; if (++cd == a.D) { cd = 0; ++cb; } } }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bsum = 0.f;
    const bool do_bias = a.dbias && kd == 0 && cblk == 0 && wc == 0;      // KSPLIT: each wave of the pair sums its own k-steps

#pragma unroll
    for (int p = 0; p < W3_ST - 1; ++p) issue();

    // ---- per-lane fragment addresses inside a stage.  k rows supplied by this lane in k-step kk: kA = 16 kk + kbase and
    // kA + 4 (kbase = 8 (g >> 1) + (li >> 2)); dY row = k, X row = halo voxel (k >> 3, k & 7) + tap offset.
    const int kbase = 8 * (g >> 1) + (li >> 2) + 32 * kh, kb4 = kbase + 4;     // KSPLIT: wave kh takes k-steps 2 kh, 2 kh + 1
    const int colb = ((g & 1) * 16 + 4 * (li & 3)) * 2;
    const int ya = kbase * W3_ROWB + wn * 64 + colb;                                    // + 4 rows = + 768 B for the second half
    const int xa = W3_YB + ((kbase >> 3) * HB + (kbase & 7)) * W3_ROWB + wc * 64 + colb;    // (20 rows per k-step: 2 halo rows of 10)
    const int xb = W3_YB + ((kb4 >> 3) * HB + (kb4 & 7)) * W3_ROWB + wc * 64 + colb;

    W3_TL(0)
    for (int tile = tbeg; tile < tend; ++tile) {
        // this wave's share of `tile` has landed (the W3_ST - 2 tiles behind it may still be in flight)
        asm volatile("s_waitcnt vmcnt(%0)" : : "n"((W3_ST - 2) * 8) : "memory");
        __builtin_amdgcn_s_barrier();                      // everybody's share has; everybody is done with tile - 1's stage
        if (!(W3_ABL & 2)) issue();                        // tile + W3_ST - 1, into the stage tile - 1 just left
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (W3_ABL & 1) continue;
        const int sbase = lds0 + (tile % W3_ST) * W3_STAGE;
        const int pya = sbase + ya, pxa = sbase + xa, pxb = sbase + xb;
        Frag af[2], bfq[W3_LA + 1];
        // unit q = 9 kk + t9: A fragment of k-step kk (rows + 16 kk), B fragment at halo rows + 20 kk + tap offset
        auto issue_a = [&](auto KK) __attribute__((always_inline)) {
            constexpr int kk = decltype(KK)::value;
            af[kk & 1].u[0] = tr_read<16 * kk * W3_ROWB>(pya);
            af[kk & 1].u[1] = tr_read<(16 * kk + 4) * W3_ROWB>(pya);
        };
        auto issue_b = [&](auto Q) __attribute__((always_inline)) {
            constexpr int q = decltype(Q)::value;
            constexpr int kk = q / 9, t9 = q % 9;
            constexpr int off = (20 * kk + (t9 / 3) * HB + (t9 % 3)) * W3_ROWB;
            bfq[q % (W3_LA + 1)].u[0] = tr_read<off>(pxa);
            bfq[q % (W3_LA + 1)].u[1] = tr_read<off>(pxb);
        };
        constexpr int NQ = KSPLIT ? 18 : 36;               // (k-step, tap) units per tile and wave
        issue_a(std::integral_constant<int, 0>{});
        static_for<0, W3_LA>([&](auto Q) { issue_b(Q); });
        static_for<0, NQ>([&](auto Q) {
            constexpr int q = decltype(Q)::value;
            if constexpr (q + W3_LA < NQ) {
                if constexpr ((q + W3_LA) % 9 == 0) issue_a(std::integral_constant<int, (q + W3_LA) / 9>{});
                issue_b(std::integral_constant<int, q + W3_LA>{});
            }
            wait_lgkm<reads_after(q, NQ)>(bfq[q % (W3_LA + 1)]);
            __builtin_amdgcn_sched_barrier(0);
            if (q % 9 == 0 && do_bias) {
#pragma unroll
                for (int j = 0; j < 8; ++j) bsum += (float)af[(q / 9) & 1].v[j];
            }
            acc[q % 9] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[(q / 9) & 1].v, bfq[q % (W3_LA + 1)].v, acc[q % 9], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    W3_TL(1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the ring's last (out-of-range) fills target this workgroup's LDS
    if (do_bias) {
        bsum += __shfl_xor(bsum, 32, 64);
        const int n = n0 + wn * 32 + (lane & 31);
        if ((lane >> 5) == 0 && n < a.Cout) atomicAdd(a.dbias + (size_t)(blockIdx.x % MM_REPL) * a.Cout + n, bsum);
    }
    if constexpr (KSPLIT) {                                  // wave (wn, 1) parks its sums, wave (wn, 0) adds them (fixed order)
        __syncthreads();
        f32x16* park = reinterpret_cast<f32x16*>(smem) + (wn * 9) * 64 + lane;   // [wn][t9][lane] x 64 B
        if (kh == 1) {
#pragma unroll
            for (int t = 0; t < 9; ++t) park[t * 64] = acc[t];
        }
        __syncthreads();
        if (kh == 1) return;
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t] += park[t * 64];
    }

    W3_TL(2)
    // register r of lane (lc, lh) is row n = n0 + 32 wn + 4 lh + (r & 3) + 8 (r >> 2), column c: the lane's pointer is
    // formed once, the (tap, r) part of every address is wave-uniform (with the strides multiplied per store in 64 bits
    // this epilogue was 17 k cycles of address arithmetic)
    const int c = c0 + wc * 32 + (lane & 31);
    const int nl = n0 + wn * 32 + 4 * (lane >> 5);
    float* lane_p = a.dw + (size_t)(a.slot_mode ? blockIdx.x : blockIdx.x % a.nrep) * a.rep_stride + (long)nl * a.sn + (long)c * a.sc +
                    (long)(kd * 9) * a.stap;
    if (c < a.Cin_real) {
        const bool full = n0 + 64 <= a.Cout;                 // (uniform) no row predicate inside
#pragma unroll
        for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dn = (r & 3) + 8 * (r >> 2);
                const long uoff = __builtin_amdgcn_readfirstlane((int)(t9 * a.stap + dn * a.sn));   // < 2^31 elements: checked on the host
                if (full || nl + dn < a.Cout) {
                    if (a.slot_mode) lane_p[uoff] = acc[t9][r];
                    else atomicAdd(lane_p + uoff, acc[t9][r]);
                }
            }
    }
#ifdef STREAM_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0 && kd == 2) {
        float* o = a.dw + (size_t)blockIdx.x * a.rep_stride;
        o[0] = tl[0]; o[1] = tl[1]; o[2] = tl[2]; o[3] = (float)(__builtin_readcyclecounter() - t_begin);
    }
#endif
}

// tiles per workgroup: one workgroup per CU (128 KB of LDS ring), every one at least W3_ST tiles deep so that the
// ring fills; fewer, longer workgroups also mean fewer slots for the scatter to sum.
int wgrad3d_tiles_per_wg(int B, int D, int H, int W, int Cin, int Cout) {
    const int tiles_total = B * D * ceil_div(H, 8) * ceil_div(W, 8);
    const int par = ceil_div(Cout, 64) * 3 * ceil_div(Cin, 64);
    int chunks = 256 / par;
    if (chunks > tiles_total / 6) chunks = tiles_total / 6;
    if (chunks > tiles_total) chunks = tiles_total;
    if (chunks < 1) chunks = 1;
    return ceil_div(tiles_total, chunks);
}

}  // namespace

extern "C" {

int mm_conv3d_wgrad_slots(int B, int D, int H, int W, int Cin, int Cout, int* slots_host, hipStream_t) {
    MM_REQUIRE(slots_host && B > 0 && D > 0 && H > 0 && W > 0, "conv3d_wgrad_slots: bad args");
    const int tiles_total = B * D * ceil_div(H, 8) * ceil_div(W, 8);
    *slots_host = ceil_div(tiles_total, wgrad3d_tiles_per_wg(B, D, H, W, Cin, Cout));
    return 0;
}

int mm_conv3d_wgrad(const void* dy, const void* x, float* dw, float* dbias, int B, int D, int H, int W, int Cin,
                    int Cout, int Cin_real, int64_t sn, int64_t sc, int64_t stap, int nrep, int64_t rep_stride,
                    int slot_mode, hipStream_t st) {
    MM_REQUIRE(dy && x && dw && B > 0, "conv3d_wgrad: null/invalid");
    MM_REQUIRE(nrep >= 1 && (slot_mode || nrep <= 64), "conv3d_wgrad: nrep");
    MM_REQUIRE(Cin % 8 == 0 && Cout % 8 == 0 && Cin_real > 0 && Cin_real <= Cin, "conv3d_wgrad: channels");
    const size_t vox = (size_t)B * D * H * W;
    MM_REQUIRE(vox * Cout * 2 < 0x7FFF0000ull && vox * Cin * 2 < 0x7FFF0000ull, "conv3d_wgrad: tensors past 2 GiB (32-bit buffer offsets)");
    MM_REQUIRE(9 * stap + 64 * sn < 0x7FFFFFFFll && stap >= 0 && sn >= 0, "conv3d_wgrad: strides");
    Wgrad3dArgs a;
    a.dy = (const bf16*)dy; a.x = (const bf16*)x; a.dw = dw; a.dbias = dbias;
    a.B = B; a.D = D; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.Cin_real = Cin_real;
    a.sn = sn; a.sc = sc; a.stap = stap; a.nrep = nrep; a.rep_stride = rep_stride; a.slot_mode = slot_mode;
    const int tiles_total = B * D * ceil_div(H, 8) * ceil_div(W, 8);
    a.tiles_per_wg = wgrad3d_tiles_per_wg(B, D, H, W, Cin, Cout);
    dim3 grid(ceil_div(tiles_total, a.tiles_per_wg), ceil_div(Cout, 64), 3 * ceil_div(Cin, 64));
    MM_REQUIRE(!slot_mode || nrep >= (int)grid.x, "conv3d_wgrad: slot mode needs %d slots, got %d", (int)grid.x, nrep);
    static const hipError_t attr0 =                                 // the only process-wide state: immutable kernel attributes
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_wgrad_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, W3_LDS);
    static const hipError_t attr1 =
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_wgrad_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, W3_LDS);
    (void)attr0; (void)attr1;
    if (Cin <= 32) hipLaunchKernelGGL(conv3d_wgrad_kernel<true>, grid, dim3(256), W3_LDS, st, a);
    else hipLaunchKernelGGL(conv3d_wgrad_kernel<false>, grid, dim3(256), W3_LDS, st, a);
    return mm_check_launch("conv3d_wgrad");
}

}  // extern "C"
