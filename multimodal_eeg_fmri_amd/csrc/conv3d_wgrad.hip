// Weight gradient of the 3-D voxel convolution (k = 3, pad 1), bf16 MFMA 32x32x16, fp32 accumulate:
//
//   dW[n, tap, c] = sum_{b, v} dY[b, v, n] * X[b, v + off(tap), c]       (GEMM M = Cout, N = 27 Cin, K = voxels)
//
// Both operands are K-strided in memory (channels-last), so a face of dY [64 voxels][32 n] and the X halo plane
// [10 x 10 voxels][32 c] are staged row-major and read with ds_read_b64_tr_b16 (hardware transpose).  Rows are 64 B,
// UNPADDED: the instruction is served in two groups of 32 lanes, a group touches four consecutive rows x two 32-byte
// halves = eight 8-bank windows at (16 row + 8 half) mod 64, all distinct (round 2's 96-byte rows put row 3 / half 0
// on the banks of row 0 / half 1: 0.45-0.47 of the LDS cycles were conflicts, and a third of the DMA pieces padding).  A workgroup owns ONE 32 (n) x 32 (c) block of
// one kd plane (9 taps: 9 x 16 accumulator registers per wave) - grid.y = 32-channel block of Cout, grid.z = kd x
// 32-channel block of Cin - over a chunk (grid.x) of 1 x 8 x 8 output faces, and its four waves split K: a stage
// holds two faces, wave w takes k-step w (voxels 16 w .. 16 w + 15) of both.  The waves are summed through LDS at
// the end (fixed order) and the block goes to slot blockIdx.x (plain 16-byte stores; mm_wgrad_scatter sums the
// slots) or is added atomically.
//
// Why small output blocks: slot traffic is (chunks) x |dW| and chunks = 256 / (blocks x 3), so 32 x 32 blocks cut
// the fp32 partial sums the kernel writes (and the scatter reads back) from 18 / 33 MB to 9 MB per launch at the
// two C2 shapes - with 64 x 64 blocks the final store burst was a quarter of the kernel (15.9 k of 68.9 k cycles).
//
// Round-2 rewrite.  The round-1 kernel prefetched the next tile into registers; its rolled tile loop made the
// compiler drain vmcnt to 0 at the top of every iteration, which left one tile (0.5 us of MFMA work) of cover for a
// ~2 us global round trip: 125 k cycles per workgroup of which 41 k in the MFMA phases.  Now
//  * the faces arrive by LDS-DMA (buffer_load_dwordx4 ... lds: no registers, no ds_write, out-of-volume voxels
//    zero-filled by the buffer's range check) into a ring of W3_ST stages, W3_ST - 1 stages ahead of the MFMAs, with
//    counted vmcnt and ONE barrier per stage; a DMA writes 1 KB contiguous per wave, so a lane's chunk is chosen on
//    the source side: chunk ci = 64 e + lane of a face image is row ci / 4, 16-byte column ci % 4;
//  * inside a stage the 18 (face, tap) units of a wave are software-pipelined by hand: the B fragment of unit q +
//    W3_LA is requested before the MFMA of unit q, under counted lgkmcnt (left to the compiler the reads sat right
//    in front of their MFMAs).  The reads are inline asm: the compiler cannot tell the ring stages apart and would
//    order every LDS read behind ALL outstanding DMA.
#include "common.h"

#include <cstdlib>
#include <type_traits>


namespace {

constexpr int HB = 10;             // halo edge of an 8-wide face
constexpr int W3_ROWB = 64;        // LDS row stride: 64 B of data (32 channels), no padding
constexpr int W3_ST = 4;           // stages of the LDS ring (96 KB: a 45-52 KB workgroup of the EEG chain still fits on the CU)
constexpr int W3_LA = 5;           // B fragments of look-ahead (2 LDS reads each + an A pair: 12 of lgkmcnt's 15); the ring of W3_LA + 1
                                   // fragments divides the 18 units of a stage, so the pipeline runs on across stages
constexpr int W3_YB = 64 * W3_ROWB;                 // 4 096 B: dY face image = 4 DMA instructions
constexpr int W3_NP = 6;                            // DMA instructions per wave and stage: a face = 4 (dY) + 7 (X plane: 100 rows x 64 B =
                                                    // 6 400 B) pieces of 1 KB + one dummy piece, split over two waves
constexpr int W3_FACE = 2 * W3_NP * 1024;           // 12 KB: the 11 pieces + the dummy's KB
constexpr int W3_STAGE = 2 * W3_FACE;               // two faces
constexpr int W3_LDS = W3_ST * W3_STAGE;
static_assert(W3_LDS >= 4 * 5 * 32 * 36 * 4, "the K-partial images of the epilogue reuse the ring");
constexpr int W3_NQ = 18;                           // (face, tap) units per stage and wave

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
constexpr int reads_after(int q) {
    int n = 0;
    for (int s = q + 1; s <= q + W3_LA; ++s) n += 2 + ((s % 9 == 0) ? 2 : 0);
    return n;
}
static_assert(18 % (W3_LA + 1) == 0 && 2 * W3_LA + 2 <= 15, "fragment ring");

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
    const int kd = blockIdx.z % 3, cblk = blockIdx.z / 3;
    const int n0 = blockIdx.y * 32, c0 = cblk * 32;
    const int tw = (a.W + 7) / 8, th = (a.H + 7) / 8;
    const int tiles_total = a.B * a.D * th * tw;
    const int tbeg = blockIdx.x * a.tiles_per_wg;
    const int tend = min(tiles_total, tbeg + a.tiles_per_wg);
    const int nstage = (tend - tbeg + 1) >> 1;               // two faces per stage
    const int li = lane & 15, g = lane >> 4;
    const int lds0 = (int)(size_t)(__attribute__((address_space(3))) char*)smem;

    // ---- DMA plan.  Waves 0, 1 fill face 0 of a stage, waves 2, 3 face 1; instruction e' = 6 (wave & 1) + e of a face
    // moves chunks ci = 64 e' + lane: e' < 4 is the dY image (row = voxel (row >> 3, row & 7) of the face), 4 <= e' < 11
    // the X plane image (row = halo voxel (row / 10, row % 10)), e' = 11 a dummy that keeps the two waves' instruction
    // counts equal (counted vmcnt).  Per lane and instruction: byte offset relative to the face's first voxel and a
    // one-hot (row, column) selector of the voxel's face coordinates, tested per stage against the wave-uniform mask
    // of coordinates that lie inside the volume (bit 31 = row past the image / channel past the tensor / dummy).
    const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16*>(a.dy), 0, (unsigned)((size_t)a.B * a.D * a.H * a.W * a.Cout * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16*>(a.x), 0, (unsigned)((size_t)a.B * a.D * a.H * a.W * a.Cin * 2), 0x00020000);
    const int face = wave >> 1;
    int rel[W3_NP];
    unsigned sel[W3_NP];
#pragma unroll
    for (int e = 0; e < W3_NP; ++e) {
        const int ep = W3_NP * (wave & 1) + e;
        const bool isy = ep < 4;                                 // wave-uniform
        const int ci = 64 * (isy ? ep : ep - 4) + lane;
        const int row = ci >> 2, ch = ci & 3;
        int vh, vw;
        bool ok;
        if (isy) {
            vh = row >> 3; vw = row & 7;
            rel[e] = ((vh * a.W + vw) * a.Cout + 8 * ch) * 2;
            ok = n0 + 8 * ch < a.Cout;
        } else {
            vh = row / HB; vw = row - HB * vh;
            rel[e] = ((vh * a.W + vw) * a.Cin + 8 * ch) * 2;
            ok = ep < 11 && row < HB * HB && c0 + 8 * ch < a.Cin;
            if (!ok) { vh = 0; vw = 0; rel[e] = 0; }
        }
        sel[e] = ok ? (1u << vh) | (1u << (16 + vw)) : 0x80000000u;
    }
    auto range_bits = [](int lo, int hi, int n) __attribute__((always_inline)) {                // bits [max(lo, 0), min(hi, n))
        lo = lo < 0 ? 0 : lo;
        hi = hi > n ? n : hi;
        return hi > lo ? ((1u << hi) - 1u) & ~((1u << lo) - 1u) : 0u;
    };
    // this wave's DMA cursor: the face it fills in the next stage, coordinates by carry, not by division
    int ctile = tbeg + face, cw0, ch0, cd, cb;
    {
        int q = ctile;
        cw0 = (q % tw) * 8; q /= tw;
        ch0 = (q % th) * 8; q /= th;
        cd = q % a.D; cb = q / a.D;
    }
    int cstage = 0;
    // One stage's fill is W3_NP = 6 instructions per wave.  In the loop they are spread over the MFMA gaps of the stage being
    // computed (issued in one piece in front of it they cost ~600 cycles of scalar / vector issue per stage that nothing hid).
    struct Fill { int ybase, xbase; unsigned ymask, xmask; char* dst; } fill;
    auto fill_begin = [&]() __attribute__((always_inline)) {
        // always W3_NP loads per wave (counted vmcnt): a face past the chunk's end fetches out of range = zeros
        const bool live = ctile < tend;
        const int dd = cd + kd - 1;
        fill.ybase = ((((cb * a.D + cd) * a.H + ch0) * a.W + cw0) * a.Cout + n0) * 2;
        fill.xbase = ((((cb * a.D + dd) * a.H + ch0 - 1) * a.W + cw0 - 1) * a.Cin + c0) * 2;
        fill.ymask = live ? range_bits(0, a.H - ch0, 8) | (range_bits(0, a.W - cw0, 8) << 16) : 0u;
        fill.xmask = (live && dd >= 0 && dd < a.D) ? range_bits(1 - ch0, a.H + 1 - ch0, HB) | (range_bits(1 - cw0, a.W + 1 - cw0, HB) << 16) : 0u;
        fill.dst = smem + (cstage % W3_ST) * W3_STAGE + face * W3_FACE + (wave & 1) * W3_NP * 1024;
        ++cstage;
#pragma unroll
        for (int k = 0; k < 2; ++k) {                          // the cursor moves two faces on
            ++ctile;
            cw0 += 8;
            if (cw0 >= tw * 8) { cw0 = 0; ch0 += 8; if (ch0 >= th * 8) { ch0 = 0; if (++cd == a.D) { cd = 0; ++cb; } } }
        }
    };
    auto fill_one = [&](int e) __attribute__((always_inline)) {
        const bool isy = W3_NP * (wave & 1) + e < 4;
        const unsigned allowed = __builtin_amdgcn_readfirstlane(isy ? fill.ymask : fill.xmask);
        const int base = __builtin_amdgcn_readfirstlane(isy ? fill.ybase : fill.xbase);
        const int off = (sel[e] & ~allowed) == 0u ? base + rel[e] : -1;
        if (isy) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_y, (lptr_t*)(fill.dst + e * 1024), 16, off, 0, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lptr_t*)(fill.dst + e * 1024), 16, off, 0, 0, 0);
    };
    auto issue = [&]() __attribute__((always_inline)) {
        fill_begin();
#pragma unroll
        for (int e = 0; e < W3_NP; ++e) fill_one(e);
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bsum = 0.f;
    const bool do_bias = a.dbias && kd == 0 && cblk == 0;    // every wave sums the dY rows of its own k-steps

#pragma unroll
    for (int p = 0; p < W3_ST - 1; ++p) issue();

    // ---- per-lane fragment addresses inside a face.  k rows supplied by this lane: kA = 16 wave + kbase and kA + 4
    // (kbase = 8 (g >> 1) + (li >> 2)); dY row = k, X row = halo voxel (k >> 3, k & 7) + tap offset.
    const int kA = 16 * wave + 8 * (g >> 1) + (li >> 2), kB = kA + 4;
    const int colb = ((g & 1) * 16 + 4 * (li & 3)) * 2;
    const int ya = kA * W3_ROWB + colb;                                                 // + 4 rows for the second half
    const int xa = W3_YB + ((kA >> 3) * HB + (kA & 7)) * W3_ROWB + colb;
    const int xb = W3_YB + ((kB >> 3) * HB + (kB & 7)) * W3_ROWB + colb;

    // ---- one continuous software pipeline over all stages.  Unit q = 9 f + t9 of a stage (face f, tap t9) requests the
    // B fragment of unit q + W3_LA - of the NEXT stage for the last W3_LA units - and then runs its MFMA; the A pair of a
    // face is requested just before the face's first B fragment.  The stage's barrier sits in the middle (unit 9): it
    // certifies that stage st + 1 has landed (counted vmcnt: one later stage may still be in flight) before unit 13 first
    // reads it, and that every wave has left stage st - 1, whose buffer the W3_NP DMA instructions of stage st + W3_ST - 1 -
    // spread over the gaps of units 10 .. 15 - then refill.  (With the barrier and the fill in front of each stage and the
    // pipeline restarted behind it a stage took 1 640 cycles for 576 of MFMA work.)
    W3_TL(0)
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"((W3_ST - 2) * W3_NP) : "memory");
    __builtin_amdgcn_s_barrier();                          // stage 0 is in LDS
    Frag af[2], bfq[W3_LA + 1];
    int pya = lds0 + ya, pxa = lds0 + xa, pxb = lds0 + xb;  // current stage; *_n: the next one
    int pya_n, pxa_n, pxb_n;
    auto issue_a = [&](auto F, int py) __attribute__((always_inline)) {
        constexpr int f = decltype(F)::value;
        af[f & 1].u[0] = tr_read<f * W3_FACE>(py);
        af[f & 1].u[1] = tr_read<f * W3_FACE + 4 * W3_ROWB>(py);
    };
    auto issue_b = [&](auto Q, int px0, int px1) __attribute__((always_inline)) {
        constexpr int q = decltype(Q)::value;
        constexpr int f = q / 9, t9 = q % 9;
        constexpr int off = f * W3_FACE + ((t9 / 3) * HB + (t9 % 3)) * W3_ROWB;
        bfq[q % (W3_LA + 1)].u[0] = tr_read<off>(px0);
        bfq[q % (W3_LA + 1)].u[1] = tr_read<off>(px1);
    };
    issue_a(std::integral_constant<int, 0>{}, pya);
    static_for<0, W3_LA>([&](auto Q) { issue_b(Q, pxa, pxb); });
    for (int st = 0; st < nstage; ++st) {
        {
            const int nb = lds0 + ((st + 1) % W3_ST) * W3_STAGE;
            pya_n = nb + ya; pxa_n = nb + xa; pxb_n = nb + xb;
        }
        static_for<0, W3_NQ>([&](auto Q) {
            constexpr int q = decltype(Q)::value;
            if constexpr (q == 9) {
                asm volatile("s_waitcnt vmcnt(%0)" : : "n"((W3_ST - 3) * W3_NP) : "memory");
                __builtin_amdgcn_s_barrier();
                fill_begin();
            }
            constexpr int p = q + W3_LA;                       // the unit requested now
            if constexpr (p < W3_NQ) {
                if constexpr (p % 9 == 0) issue_a(std::integral_constant<int, p / 9>{}, pya);
                issue_b(std::integral_constant<int, p>{}, pxa, pxb);
            } else {
                if constexpr (p == W3_NQ) issue_a(std::integral_constant<int, 0>{}, pya_n);
                issue_b(std::integral_constant<int, p - W3_NQ>{}, pxa_n, pxb_n);
            }
            wait_lgkm<reads_after(q)>(bfq[q % (W3_LA + 1)]);
            __builtin_amdgcn_sched_barrier(0);
            if (q % 9 == 0 && do_bias) {
#pragma unroll
                for (int j = 0; j < 8; ++j) bsum += (float)af[(q / 9) & 1].v[j];
            }
            acc[q % 9] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[(q / 9) & 1].v, bfq[q % (W3_LA + 1)].v, acc[q % 9], 0, 0, 0);
            if constexpr (q >= 10 && q < 10 + W3_NP) fill_one(q - 10);
            __builtin_amdgcn_sched_barrier(0);
        });
        pya = pya_n; pxa = pxa_n; pxb = pxb_n;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the look-ahead past the last stage
    W3_TL(1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the ring's last (out-of-range) fills target this workgroup's LDS
    if (do_bias) {
        bsum += __shfl_xor(bsum, 32, 64);
        const int n = n0 + (lane & 31);
        if ((lane >> 5) == 0 && n < a.Cout) acc_add<MM_ACC_GRAD>(acc_rep(a.dbias, blockIdx.x % MM_ACC_REPL, a.Cout) + n, bsum);
    }
    // ---- the four K partials: every wave scatters its accumulators into its own [tap][n][c] image in LDS (row stride 36
    // floats), then all 256 threads sum the four images in wave order (fixed) and store 16-byte pieces.  (A wave's
    // global_store_dword took ~110 cycles whatever it carried: 144 four-byte stores per lane were 16 k cycles of every
    // launch.)  Two rounds of at most five taps: 4 x 5 x 32 x 36 floats = 90 KB of the ring.
    constexpr int PS = 36, IMG = 5 * 32 * PS;
    float* img = reinterpret_cast<float*>(smem);
    const int ec = lane & 31, en = 4 * (lane >> 5);            // register r of this lane: row en + (r & 3) + 8 (r >> 2), column ec
    float* slot = a.dw + (size_t)blockIdx.x * a.rep_stride;
    const bool vec = a.sc == 1 && (a.stap & 3) == 0 && (a.sn & 3) == 0 && c0 + 32 <= a.Cin_real &&
                     ((size_t)slot & 15) == 0;                                               // (uniform)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int t_lo = half * 5, nt = half ? 4 : 5;
        __syncthreads();
#pragma unroll
        for (int tt = 0; tt < 5; ++tt)
            if (tt < nt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    img[wave * IMG + (tt * 32 + en + (r & 3) + 8 * (r >> 2)) * PS + ec] = acc[t_lo + tt][r];
        __syncthreads();
        if (vec) {
            for (int i = tid; i < nt * 32 * 8; i += 256) {                                    // (tap, n, 4 columns)
                const int c4 = i & 7, n = (i >> 3) & 31, tt = i >> 8;
                const float* src = img + (tt * 32 + n) * PS + 4 * c4;
                f32x4 v = *reinterpret_cast<const f32x4*>(src);
                v += *reinterpret_cast<const f32x4*>(src + IMG);
                v += *reinterpret_cast<const f32x4*>(src + 2 * IMG);
                v += *reinterpret_cast<const f32x4*>(src + 3 * IMG);
                if (n0 + n < a.Cout)
                    *reinterpret_cast<f32x4*>(slot + (long)(n0 + n) * a.sn + (long)(kd * 9 + t_lo + tt) * a.stap + c0 + 4 * c4) = v;
            }
        } else {
            for (int i = tid; i < nt * 32 * 32; i += 256) {
                const int cc = i & 31, n = (i >> 5) & 31, tt = i >> 10;
                const float* src = img + (tt * 32 + n) * PS + cc;
                const float v = ((src[0] + src[IMG]) + src[2 * IMG]) + src[3 * IMG];
                if (n0 + n < a.Cout && c0 + cc < a.Cin_real)
                    slot[(long)(n0 + n) * a.sn + (long)(c0 + cc) * a.sc + (long)(kd * 9 + t_lo + tt) * a.stap] = v;
            }
        }
    }
    W3_TL(2)
#ifdef STREAM_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0 && kd == 2) {
        float* o = a.dw + (size_t)blockIdx.x * a.rep_stride;
        o[0] = tl[0]; o[1] = tl[1]; o[2] = tl[2]; o[3] = (float)(__builtin_readcyclecounter() - t_begin);
    }
#endif
}

// faces per workgroup: one workgroup per CU (96 KB of LDS ring), every one at least 2 W3_ST faces deep so that the ring
// fills; the more output blocks there are, the fewer chunks - and slots for the scatter to sum - it takes to get there.
int wgrad3d_tiles_per_wg(int B, int D, int H, int W, int Cin, int Cout) {
    const int tiles_total = B * D * ceil_div(H, 8) * ceil_div(W, 8);
    const int par = ceil_div(Cout, 32) * 3 * ceil_div(Cin, 32);
    // workgroups (= CUs: 96 KB of LDS each) to spread over.  MM_W3_CUS (A/B runs): fewer than 256 leaves whole CUs to the kernels of
    // the other stream, which cannot co-reside with a 96 KB workgroup
    static const int cus = getenv("MM_W3_CUS") ? atoi(getenv("MM_W3_CUS")) : 256;
    int chunks = cus / par;
    if (chunks > tiles_total / (2 * W3_ST)) chunks = tiles_total / (2 * W3_ST);
    if (chunks < 1) chunks = 1;
    int per = ceil_div(tiles_total, chunks);
    return per + (per & 1);                                  // even: a stage is two faces of the same chunk
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
    MM_REQUIRE(slot_mode == 1 && nrep >= 1, "conv3d_wgrad: slot_mode must be 1 (the fp32-atomics mode is gone: results are order-free)");
    MM_REQUIRE(Cin % 8 == 0 && Cout % 8 == 0 && Cin_real > 0 && Cin_real <= Cin, "conv3d_wgrad: channels");
    const size_t vox = (size_t)B * D * H * W;
    MM_REQUIRE(vox * Cout * 2 < 0x7FFF0000ull && vox * Cin * 2 < 0x7FFF0000ull, "conv3d_wgrad: tensors past 2 GiB (32-bit buffer offsets)");
    MM_REQUIRE(9 * stap + 32 * sn < 0x7FFFFFFFll && stap >= 0 && sn >= 0, "conv3d_wgrad: strides");
    Wgrad3dArgs a;
    a.dy = (const bf16*)dy; a.x = (const bf16*)x; a.dw = dw; a.dbias = dbias;
    a.B = B; a.D = D; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.Cin_real = Cin_real;
    a.sn = sn; a.sc = sc; a.stap = stap; a.nrep = nrep; a.rep_stride = rep_stride; a.slot_mode = slot_mode;
    const int tiles_total = B * D * ceil_div(H, 8) * ceil_div(W, 8);
    a.tiles_per_wg = wgrad3d_tiles_per_wg(B, D, H, W, Cin, Cout);
    dim3 grid(ceil_div(tiles_total, a.tiles_per_wg), ceil_div(Cout, 32), 3 * ceil_div(Cin, 32));
    MM_REQUIRE(!slot_mode || nrep >= (int)grid.x, "conv3d_wgrad: slot mode needs %d slots, got %d", (int)grid.x, nrep);
    static const hipError_t attr =                                  // the only process-wide state: an immutable kernel attribute
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, W3_LDS);
    (void)attr;
    hipLaunchKernelGGL(conv3d_wgrad_kernel, grid, dim3(256), W3_LDS, st, a);
    return mm_check_launch("conv3d_wgrad");
}

}  // extern "C"
