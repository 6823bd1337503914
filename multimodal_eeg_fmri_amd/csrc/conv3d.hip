// 3-D voxel convolution (k = 3, pad = 1, stride 1) as an LDS-staged implicit
// GEMM on bf16 MFMA, channels-last (NDHWC), fp32 accumulate.
//
//   forward : Y[b,v,n]   = sum_{tap,c} X[b, v + off(tap), c] * W[n, tap, c]
//   (data gradient = same kernel on dY with the flipped/transposed image)
//   wgrad   : dW[n,tap,c] = sum_{b,v} dY[b,v,n] * X[b, v + off(tap), c]
//
// One workgroup owns a TD x 8 x 8 block of output voxels (BM = 64*TD GEMM rows)
// of one sample.  The (TD+2) x 10 x 10 input halo block is staged ONCE per Cin
// chunk into LDS as rows of KC channels; every one of the 27 taps then reads
// its A fragments from that block at a row offset kd*100 + kh*10 + kw, so the
// im2col matrix only ever exists as LDS addresses.  The weight slice of one kd
// plane (9 taps) sits beside it and is re-staged three times per chunk.
#include "common.h"

namespace {

constexpr int KC3 = 32;
constexpr int KPAD3 = 8;
constexpr int HB = 10;            // halo edge of an 8-wide tile

struct Conv3dArgs {
    const bf16* x; const bf16* w;
    int B, D, H, W, Cin, Cout;
    const float* shift;           // [Cout] bias (nullptr = 0)
    int dbg;                      // ablation switches for tools/kbench.py (0 in production)
    float* stats;                 // [2][Cout] sum / sumsq of (acc + shift)   (nullptr)
    float* out_f32;               // [B][D][H][W][Cout]
    bf16* out_bf16;
};

template <int TD, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv3d_fwd_kernel(Conv3dArgs a) {
    constexpr int BM = TD * 64;
    constexpr int TM = BM / (WM * 32);
    constexpr int TN = BN / (WN * 32);
    static_assert(WM * WN == 4 && TM >= 1 && TN >= 1, "tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 31, lh = lane >> 5;
    const int tw = (a.W + 7) / 8, th = (a.H + 7) / 8, td = (a.D + TD - 1) / TD;
    int bid = blockIdx.x;
    const int w0 = (bid % tw) * 8; bid /= tw;
    const int h0 = (bid % th) * 8; bid /= th;
    const int d0 = (bid % td) * TD; bid /= td;
    const int b = bid;
    const int n0 = blockIdx.y * BN;
    const int kc = a.Cin < KC3 ? a.Cin : KC3;
    const int AS = kc + KPAD3;
    constexpr int HROWS = (TD + 2) * HB * HB;
    bf16* As = reinterpret_cast<bf16*>(smem);
    bf16* Ws = As + HROWS * AS;
    const int segs = kc / 8;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // halo-row base of this lane's A row for every M sub-tile
    int abase[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = (wm * TM + i) * 32 + lr;
        abase[i] = ((m >> 6) * HB + ((m >> 3) & 7)) * HB + (m & 7);
    }
    const bf16* xb = a.x + (size_t)b * a.D * a.H * a.W * a.Cin;

    for (int c0 = 0; c0 < a.Cin; c0 += kc) {
        for (int s = tid; s < HROWS * segs; s += 256) {
            const int r = s / segs, sg = s - r * segs;
            const int hw = r % HB, hh = (r / HB) % HB, hd = r / (HB * HB);
            const int d = d0 + hd - 1, h = h0 + hh - 1, w = w0 + hw - 1;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (d >= 0 && d < a.D && h >= 0 && h < a.H && w >= 0 && w < a.W)
                v = *reinterpret_cast<const uint4*>(xb + (((size_t)d * a.H + h) * a.W + w) * a.Cin + c0 + sg * 8);
            *reinterpret_cast<uint4*>(As + r * AS + sg * 8) = v;
        }
        for (int kd = 0; kd < 3; ++kd) {
            if (kd) __syncthreads();                       // previous plane's reads done
            for (int s = tid; s < BN * 9 * segs; s += 256) {
                const int r = s / segs, sg = s - r * segs; // r = n_local * 9 + t9
                const int nl = r / 9, t9 = r - nl * 9;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (n0 + nl < a.Cout)
                    v = *reinterpret_cast<const uint4*>(a.w + ((size_t)(n0 + nl) * 27 + kd * 9 + t9) * a.Cin + c0 + sg * 8);
                *reinterpret_cast<uint4*>(Ws + r * AS + sg * 8) = v;
            }
            __syncthreads();
#pragma unroll
            for (int t9 = 0; t9 < 9; ++t9) {
                const int toff = kd * HB * HB + (t9 / 3) * HB + (t9 % 3);
                for (int ks = 0; ks < kc; ks += 16) {
                    bf16x8 af[TM], bfr[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        af[i] = *reinterpret_cast<const bf16x8*>(As + (abase[i] + toff) * AS + ks + lh * 8);
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int nl = (wn * TN + j) * 32 + lr;
                        bfr[j] = *reinterpret_cast<const bf16x8*>(Ws + (nl * 9 + t9) * AS + ks + lh * 8);
                    }
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    float* sstat = reinterpret_cast<float*>(smem);
    if (a.stats) {
        for (int i = tid; i < 2 * BN; i += 256) sstat[i] = 0.f;
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + (wn * TN + j) * 32 + lr;
        const bool nok = n < a.Cout;
        const float sh = (a.shift && nok) ? a.shift[n] : 0.f;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int d = d0 + (m >> 6), h = h0 + ((m >> 3) & 7), w = w0 + (m & 7);
                if (nok && d < a.D && h < a.H && w < a.W) {
                    const float v = acc[i][j][r] + sh;
                    s1 += v; s2 += v * v;
                    const size_t o = ((((size_t)b * a.D + d) * a.H + h) * a.W + w) * a.Cout + n;
                    if (a.out_f32) a.out_f32[o] = v;
                    if (a.out_bf16) a.out_bf16[o] = (bf16)v;
                }
            }
        if (a.stats) {
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (lh == 0) {
                const int nl = (wn * TN + j) * 32 + lr;
                atomicAdd(&sstat[nl], s1);
                atomicAdd(&sstat[BN + nl], s2);
            }
        }
    }
    if (a.stats) {
        __syncthreads();
        float* rep = a.stats + (size_t)(blockIdx.x % MM_REPL) * 2 * a.Cout;
        for (int i = tid; i < BN; i += 256)
            if (n0 + i < a.Cout) {
                atomicAdd(&rep[n0 + i], sstat[i]);
                atomicAdd(&rep[a.Cout + n0 + i], sstat[BN + i]);
            }
    }
}

template <int TD, int BN, int WM, int WN>
int launch3d(const Conv3dArgs& a, hipStream_t st) {
    const int kc = a.Cin < KC3 ? a.Cin : KC3;
    const size_t lds = (size_t)((TD + 2) * HB * HB + BN * 9) * (kc + KPAD3) * sizeof(bf16);
    if (lds > 160 * 1024) return mm_fail(MM_ERR_UNSUPPORTED, "conv3d_fwd: LDS %zu", lds);
    auto kern = conv3d_fwd_kernel<TD, BN, WM, WN>;
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid(a.B * ceil_div(a.D, TD) * ceil_div(a.H, 8) * ceil_div(a.W, 8), ceil_div(a.Cout, BN));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
    return mm_check_launch("conv3d_fwd");
}


// ---------------------------------------------------------------------------
// W-resident persistent variant for layers whose whole weight image fits LDS
// (Cin = 32, Cout <= 64: 64 x 27 x 32 bf16 = 108 KiB).  One workgroup per CU
// keeps ALL taps in LDS for its lifetime and walks 2x8x8 output tiles; the next
// tile's (4x10x10)-row halo is fetched into registers while the current tile
// runs its 27 x 2 x 2 MFMAs per wave, then swapped in behind one barrier pair.
// LDS rows are unpadded 64-B rows; the 16-B chunk index is XOR-swizzled with
// (row >> 2) & 3 (halo) / (n >> 2) & 3 (weights) so the 16-lane groups of
// ds_read_b128 spread over all sixteen 16-B slots of a 256-B bank row.
// ---------------------------------------------------------------------------
constexpr int WR_CIN = 32;
constexpr int WR_BN = 64;
constexpr int WR_TD = 4;                              // tile depth: 4 x 8 x 8 = 256 GEMM rows
constexpr int WR_TM = WR_TD / 2;                      // 32-row sub-tiles per wave (4 waves x 64 rows)
constexpr int WR_HROWS = (WR_TD + 2) * HB * HB;       // 600
constexpr int WR_HREGS = (WR_HROWS * 4 + 255) / 256;  // uint4 per thread per halo tile (10)

__device__ __forceinline__ int swz(int row_key, int seg) { return seg ^ ((row_key >> 2) & 3); }

__global__ __launch_bounds__(256) void conv3d_fwd_wres_kernel(Conv3dArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* Wl = reinterpret_cast<bf16*>(smem);                       // [64*27][32]
    bf16* Hl = Wl + WR_BN * 27 * WR_CIN;                             // [400][32]
    float* sstat = reinterpret_cast<float*>(Hl + WR_HROWS * WR_CIN); // [2][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int tw = (a.W + 7) / 8, th = (a.H + 7) / 8, td = (a.D + WR_TD - 1) / WR_TD;
    const int ntiles = a.B * td * th * tw;

    if (a.stats)
        for (int i = tid; i < 2 * WR_BN; i += 256) sstat[i] = 0.f;

    auto load_halo = [&](int tile, uint4 (&regs)[WR_HREGS]) {
        int q = tile;
        const int w0 = (q % tw) * 8; q /= tw;
        const int h0 = (q % th) * 8; q /= th;
        const int d0 = (q % td) * WR_TD; q /= td;
        const bf16* xb = a.x + (size_t)q * a.D * a.H * a.W * WR_CIN;
#pragma unroll
        for (int i = 0; i < WR_HREGS; ++i) {
            const int s = tid + i * 256;
            const int r = s >> 2, sg = s & 3;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (r < WR_HROWS) {
                const int hw = r % HB, hh = (r / HB) % HB, hd = r / (HB * HB);
                const int d = d0 + hd - 1, h = h0 + hh - 1, w = w0 + hw - 1;
                if (d >= 0 && d < a.D && h >= 0 && h < a.H && w >= 0 && w < a.W)
                    v = *reinterpret_cast<const uint4*>(xb + (((size_t)d * a.H + h) * a.W + w) * WR_CIN + sg * 8);
            }
            regs[i] = v;
        }
    };
    auto store_halo = [&](const uint4 (&regs)[WR_HREGS]) {
#pragma unroll
        for (int i = 0; i < WR_HREGS; ++i) {
            const int s = tid + i * 256;
            const int r = s >> 2, sg = s & 3;
            if (r < WR_HROWS) *reinterpret_cast<uint4*>(Hl + r * WR_CIN + swz(r, sg) * 8) = regs[i];
        }
    };

    int abase[WR_TM];                                               // this lane's A rows in the tile
#pragma unroll
    for (int i = 0; i < WR_TM; ++i) {
        const int m = (wave * WR_TM + i) * 32 + lr;
        abase[i] = ((m >> 6) * HB + ((m >> 3) & 7)) * HB + (m & 7);
    }
    float st1[4] = {0.f, 0.f, 0.f, 0.f}, st2[4] = {0.f, 0.f, 0.f, 0.f}, sh4[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int n = (tid & 15) * 4 + c;
        sh4[c] = (a.shift && n < a.Cout) ? a.shift[n] : 0.f;
    }

    uint4 nxt[WR_HREGS];
    int tile = blockIdx.x;
    if (a.dbg & 16) tile = ntiles;
    if (tile < ntiles) load_halo(tile, nxt);            // in flight while the weights are staged
    // ---- weights: once per workgroup.  All 27 loads of a thread are issued before
    // the first LDS write (a load->store loop would serialise on load latency).
    {
        constexpr int WREGS = WR_BN * 27 * 4 / 256;      // 27
        uint4 wv[WREGS];
        const int wvalid = a.Cout * 27;
#pragma unroll
        for (int i = 0; i < WREGS; ++i) {
            const int s = tid + i * 256, r = s >> 2, sg = s & 3;      // r = n * 27 + tap
            wv[i] = (r < wvalid && !(a.dbg & 8)) ? *reinterpret_cast<const uint4*>(a.w + (size_t)r * WR_CIN + sg * 8) : make_uint4(0, 0, 0, 0);
        }
        if (!(a.dbg & 32))
#pragma unroll
        for (int i = 0; i < WREGS; ++i) {
            const int s = tid + i * 256, r = s >> 2, sg = s & 3;
            *reinterpret_cast<uint4*>(Wl + r * WR_CIN + swz(r / 27, sg) * 8) = wv[i];
        }
    }

    for (; tile < ntiles; tile += gridDim.x) {
        __syncthreads();                                            // previous tile's reads are done
        if (!(a.dbg & 256)) store_halo(nxt);
        __syncthreads();
        const int tnext = tile + gridDim.x;
        if (tnext < ntiles && !(a.dbg & 4)) load_halo(tnext, nxt);  // in flight during the MFMAs below

        f32x16 acc[WR_TM][2];
#pragma unroll
        for (int i = 0; i < WR_TM; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        if (!(a.dbg & 2))
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
            const int toff = (tap / 9) * HB * HB + ((tap / 3) % 3) * HB + (tap % 3);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int sg = ks * 2 + lh;
                bf16x8 af[WR_TM], bfr[2];
#pragma unroll
                for (int i = 0; i < WR_TM; ++i) {
                    const int arow = abase[i] + toff;
                    af[i] = *reinterpret_cast<const bf16x8*>(Hl + arow * WR_CIN + swz(arow, sg) * 8);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int n = j * 32 + lr;
                    bfr[j] = *reinterpret_cast<const bf16x8*>(Wl + (n * 27 + tap) * WR_CIN + swz(n, sg) * 8);
                }
#pragma unroll
                for (int i = 0; i < WR_TM; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
        }
        // ---- epilogue through LDS (the halo region is dead once every wave left the
        // MFMA loop): two 128-row halves of fp32 [128][64+4]; each thread then owns one
        // 4-column group and writes row-contiguous 16-byte vectors (8 per half).
        int q = tile;
        const int w0 = (q % tw) * 8; q /= tw;
        const int h0 = (q % th) * 8; q /= th;
        const int d0 = (q % td) * WR_TD; q /= td;
        const int b = q;
        if (a.dbg & 128) continue;
        const bool full = d0 + WR_TD <= a.D && h0 + 8 <= a.H && w0 + 8 <= a.W;
        float* Cs = reinterpret_cast<float*>(Hl);
        constexpr int LDC = WR_BN + 4;
        const int cg = tid & 15, rr = tid >> 4;                    // column group, first row
        const bool cok = cg * 4 < a.Cout;
        __syncthreads();
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if ((wave >> 1) == half) {
#pragma unroll
                for (int i = 0; i < WR_TM; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = (wave & 1) * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                            Cs[row * LDC + j * 32 + lr] = acc[i][j][r];
                        }
            }
            __syncthreads();
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int row = rr + 16 * c;                      // 0..127 inside the half
                const int mg = half * 128 + row;
                const int d = d0 + (mg >> 6), h = h0 + ((mg >> 3) & 7), w = w0 + (mg & 7);
                if (cok && (full || (d < a.D && h < a.H && w < a.W))) {
                    const float4 t = *reinterpret_cast<const float4*>(Cs + row * LDC + cg * 4);
                    const float v0 = t.x + sh4[0], v1 = t.y + sh4[1], v2 = t.z + sh4[2], v3 = t.w + sh4[3];
                    st1[0] += v0; st1[1] += v1; st1[2] += v2; st1[3] += v3;
                    st2[0] += v0 * v0; st2[1] += v1 * v1; st2[2] += v2 * v2; st2[3] += v3 * v3;
                    const size_t o = ((((size_t)b * a.D + d) * a.H + h) * a.W + w) * a.Cout + cg * 4;
                    if (!(a.dbg & 1)) {
                        if (a.out_f32) *reinterpret_cast<float4*>(a.out_f32 + o) = make_float4(v0, v1, v2, v3);
                        if (a.out_bf16) {
                            bf16x4 ob = {(bf16)v0, (bf16)v1, (bf16)v2, (bf16)v3};
                            *reinterpret_cast<bf16x4*>(a.out_bf16 + o) = ob;
                        }
                    }
                }
            }
            if (half == 0) __syncthreads();
        }
    }
    if (a.stats && !(a.dbg & 64)) {
        __syncthreads();                                            // Cs reads done; sstat lives past Hl
        if ((tid & 15) * 4 < a.Cout)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                atomicAdd(&sstat[(tid & 15) * 4 + c], st1[c]);
                atomicAdd(&sstat[WR_BN + (tid & 15) * 4 + c], st2[c]);
            }
        __syncthreads();
        float* rep = a.stats + (size_t)(blockIdx.x % MM_REPL) * 2 * a.Cout;
        for (int i = tid; i < WR_BN; i += 256)
            if (i < a.Cout) {
                atomicAdd(&rep[i], sstat[i]);
                atomicAdd(&rep[a.Cout + i], sstat[WR_BN + i]);
            }
    }
}

int launch3d_wres(const Conv3dArgs& a, hipStream_t st) {
    const size_t lds = (size_t)(WR_BN * 27 + WR_HROWS) * WR_CIN * sizeof(bf16) + 2 * WR_BN * sizeof(float);
    auto kern = conv3d_fwd_wres_kernel;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const int ntiles = a.B * ceil_div(a.D, WR_TD) * ceil_div(a.H, 8) * ceil_div(a.W, 8);
    const int grid = ntiles < 256 ? ntiles : 256;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);
    return mm_check_launch("conv3d_fwd_wres");
}

// ---------------------------------------------------------------------------
// weight gradient (transposed LDS reads, see igemm1d.hip).  grid.z = kd plane:
// each workgroup accumulates the 9 taps of one kd for a 64(n) x BC(c) block over
// a run of output tiles, then adds them atomically into dW (element strides).
// ---------------------------------------------------------------------------
constexpr int W3_LD = 96;          // LDS row stride (elements) == 192 B (mod 256)

__device__ __forceinline__ bf16x8 tr_frag_rows(const bf16* tile, int rowA, int rowB, int col0, int lane) {
    // rowA/rowB: LDS rows of k = 8*(lane>>5) + (li>>2) and that + 4 (per lane)
    const int li = lane & 15, g = lane >> 4;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const bf16* pa = tile + rowA * W3_LD + col0 + (g & 1) * 16 + 4 * (li & 3);
    const bf16* pb = tile + rowB * W3_LD + col0 + (g & 1) * 16 + 4 * (li & 3);
    union { s16x4 s[2]; bf16x8 v; } u;
    u.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)pa);
    u.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)pb);
    return u.v;
}

struct Wgrad3dArgs {
    const bf16* dy; const bf16* x; float* dw; float* dbias;
    int B, D, H, W, Cin, Cout, Cin_real, tiles_per_wg, nrep;
    long sn, sc, stap, rep_stride;
};

__global__ __launch_bounds__(256) void conv3d_wgrad_kernel(Wgrad3dArgs a) {
    // tile = 1 x 8 x 8 output voxels (64 GEMM-k rows); halo = 3 x 10 x 10
    __shared__ __attribute__((aligned(16))) bf16 Ys[64 * W3_LD];
    __shared__ __attribute__((aligned(16))) bf16 Xs[3 * HB * HB * W3_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wc = wave & 1;
    const int kd = blockIdx.z % 3, cblk = blockIdx.z / 3;
    const int n0 = blockIdx.y * 64, c0 = cblk * 64;
    const int tw = (a.W + 7) / 8, th = (a.H + 7) / 8;
    const int tiles_total = a.B * a.D * th * tw;
    const int tbeg = blockIdx.x * a.tiles_per_wg;
    const int tend = min(tiles_total, tbeg + a.tiles_per_wg);

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bsum = 0.f;
    const int li = lane & 15, g = lane >> 4;

    for (int tile = tbeg; tile < tend; ++tile) {
        int q = tile;
        const int w0 = (q % tw) * 8; q /= tw;
        const int h0 = (q % th) * 8; q /= th;
        const int d = q % a.D; q /= a.D;
        const int b = q;
        __syncthreads();
        for (int s = tid; s < 64 * 8; s += 256) {
            const int r = s >> 3, sg = s & 7;
            const int h = h0 + (r >> 3), w = w0 + (r & 7), n = n0 + sg * 8;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (h < a.H && w < a.W && n < a.Cout)
                v = *reinterpret_cast<const uint4*>(a.dy + ((((size_t)b * a.D + d) * a.H + h) * a.W + w) * a.Cout + n);
            *reinterpret_cast<uint4*>(Ys + r * W3_LD + sg * 8) = v;
        }
        // only the kd-th depth plane of the halo is needed: rows [0, 100)
        for (int s = tid; s < HB * HB * 8; s += 256) {
            const int r = s >> 3, sg = s & 7;
            const int hh = r / HB, hw = r % HB;
            const int dd = d + kd - 1, h = h0 + hh - 1, w = w0 + hw - 1, c = c0 + sg * 8;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (dd >= 0 && dd < a.D && h >= 0 && h < a.H && w >= 0 && w < a.W && c < a.Cin)
                v = *reinterpret_cast<const uint4*>(a.x + ((((size_t)b * a.D + dd) * a.H + h) * a.W + w) * a.Cin + c);
            *reinterpret_cast<uint4*>(Xs + r * W3_LD + sg * 8) = v;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 64; kk += 16) {
            // k rows supplied by this lane: kA = kk + 8*(g>>1) + (li>>2), kB = kA + 4
            const int kA = kk + 8 * (g >> 1) + (li >> 2), kB = kA + 4;
            const bf16x8 af = tr_frag_rows(Ys, kA, kB, wn * 32, lane);
            if (a.dbias && kd == 0 && cblk == 0 && wc == 0)
#pragma unroll
                for (int j = 0; j < 8; ++j) bsum += (float)af[j];
            const int hA = (kA >> 3) * HB + (kA & 7), hB = (kB >> 3) * HB + (kB & 7);
#pragma unroll
            for (int t9 = 0; t9 < 9; ++t9) {
                const int toff = (t9 / 3) * HB + (t9 % 3);
                const bf16x8 bfr = tr_frag_rows(Xs, hA + toff, hB + toff, wc * 32, lane);
                acc[t9] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[t9], 0, 0, 0);
            }
        }
    }
    const int c = c0 + wc * 32 + (lane & 31);
    float* dwr = a.dw + (size_t)(blockIdx.x % a.nrep) * a.rep_stride;
    if (c < a.Cin_real) {
#pragma unroll
        for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (n < a.Cout) atomicAdd(dwr + n * a.sn + c * a.sc + (kd * 9 + t9) * a.stap, acc[t9][r]);
            }
    }
    if (a.dbias && kd == 0 && cblk == 0 && wc == 0) {
        bsum += __shfl_xor(bsum, 32, 64);
        const int n = n0 + wn * 32 + (lane & 31);
        if ((lane >> 5) == 0 && n < a.Cout) atomicAdd(a.dbias + (size_t)(blockIdx.x % MM_REPL) * a.Cout + n, bsum);
    }
}

// ---------------------------------------------------------------------------
// (B,1,D,H,W) fp32 -> (B,D,H,W,Cp) bf16 with channel 0 = value, rest zero
// ---------------------------------------------------------------------------
__global__ void pack_vol_kernel(const float* __restrict__ x, bf16* __restrict__ y, size_t nvox, int Cp) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvox * (Cp / 8); i += (size_t)gridDim.x * blockDim.x) {
        const size_t v = i / (Cp / 8);
        const int sg = (int)(i % (Cp / 8));
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16)0.f;
        if (sg == 0) o[0] = (bf16)x[v];
        *reinterpret_cast<bf16x8*>(y + v * Cp + sg * 8) = o;
    }
}

// ---------------------------------------------------------------------------
// BN/act/2x2x2-maxpool/dropout on channels-last volumes (fwd + bwd), the 3-D
// sibling of bn_act_* in elementwise.hip.
// ---------------------------------------------------------------------------
struct Pool3Args {
    const float* y; const float* out4; const bf16* dout; const float* sums;
    bf16* out; float* sums_out; bf16* dy;
    int B, D, H, W, N, act, train;
    uint32_t thresh, seed; float inv_keep, inv_count;
    const uint32_t* epoch;
};

template <int MODE>   // 0 fwd, 1 bwd-reduce, 2 bwd-apply
__global__ void pool3_bn_act_kernel(Pool3Args a) {
    a.seed = mm_eff_seed(a.seed, a.epoch);
    const int nv = a.N / 4;
    const int Do = a.D / 2, Ho = a.H / 2, Wo = a.W / 2;
    const size_t nrows = (size_t)a.B * Do * Ho * Wo;
    const int rows_per_blk = 256 / nv > 0 ? 256 / nv : 1;
    const int vi = threadIdx.x % nv, ri = threadIdx.x / nv;
    const bool active = ri < rows_per_blk;
    const int n4 = vi * 4;
    float sc[4], sh[4], mu[4], rs[4], c0[4], c1[4], s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        sc[q] = a.out4[n4 + q]; sh[q] = a.out4[a.N + n4 + q];
        mu[q] = a.out4[2 * a.N + n4 + q]; rs[q] = a.out4[3 * a.N + n4 + q];
        c0[q] = (MODE == 2 && a.train) ? a.sums[n4 + q] * a.inv_count : 0.f;    // compact [2][N] sums
        c1[q] = (MODE == 2 && a.train) ? a.sums[a.N + n4 + q] * a.inv_count : 0.f;
    }
    if (active)
        for (size_t row = (size_t)blockIdx.x * rows_per_blk + ri; row < nrows; row += (size_t)gridDim.x * rows_per_blk) {
            size_t q = row;
            const int ow = (int)(q % Wo); q /= Wo;
            const int oh = (int)(q % Ho); q /= Ho;
            const int od = (int)(q % Do); q /= Do;
            const size_t b = q;
            float yv[8][4], av[8][4];
            size_t idx[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int d = 2 * od + (j >> 2), h = 2 * oh + ((j >> 1) & 1), w = 2 * ow + (j & 1);
                idx[j] = ((((size_t)b * a.D + d) * a.H + h) * a.W + w) * a.N + n4;
                const float4 t = *reinterpret_cast<const float4*>(a.y + idx[j]);
                yv[j][0] = t.x; yv[j][1] = t.y; yv[j][2] = t.z; yv[j][3] = t.w;
#pragma unroll
                for (int c = 0; c < 4; ++c) av[j][c] = apply_act(yv[j][c] * sc[c] + sh[c], a.act);
            }
            const size_t oidx = row * a.N + n4;
            int arg[4];
            float mx[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                mx[c] = av[0][c]; arg[c] = 0;
#pragma unroll
                for (int j = 1; j < 8; ++j)
                    if (av[j][c] > mx[c]) { mx[c] = av[j][c]; arg[c] = j; }
            }
            if (MODE == 0) {
                bf16x4 o;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float v = mx[c];
                    if (a.thresh) v *= dropout_scale(a.seed, (uint32_t)(oidx + c), a.thresh, a.inv_keep);
                    o[c] = (bf16)v;
                }
                *reinterpret_cast<bf16x4*>(a.out + oidx) = o;
            } else {
                const bf16x4 gv = *reinterpret_cast<const bf16x4*>(a.dout + oidx);
                float dz[8][4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float g = (float)gv[c];
                    if (a.thresh) g *= dropout_scale(a.seed, (uint32_t)(oidx + c), a.thresh, a.inv_keep);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float z = yv[j][c] * sc[c] + sh[c];
                        dz[j][c] = (j == arg[c]) ? g * act_grad(z, a.act) : 0.f;
                        const float xh = (yv[j][c] - mu[c]) * rs[c];
                        if (MODE == 1) { s0[c] += dz[j][c]; s1[c] += dz[j][c] * xh; }
                        else dz[j][c] = a.train ? sc[c] * (dz[j][c] - c0[c] - xh * c1[c]) : sc[c] * dz[j][c];
                    }
                }
                if (MODE == 2)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        bf16x4 o = {(bf16)dz[j][0], (bf16)dz[j][1], (bf16)dz[j][2], (bf16)dz[j][3]};
                        *reinterpret_cast<bf16x4*>(a.dy + idx[j]) = o;
                    }
            }
        }
    if (MODE == 1) {
        __shared__ float red[2][1024];
        for (int i = threadIdx.x; i < 2048; i += 256) (&red[0][0])[i] = 0.f;
        __syncthreads();
        if (active)
#pragma unroll
            for (int c = 0; c < 4; ++c) { atomicAdd(&red[0][n4 + c], s0[c]); atomicAdd(&red[1][n4 + c], s1[c]); }
        __syncthreads();
        float* rep = a.sums_out + (size_t)(blockIdx.x % MM_REPL) * 2 * a.N;
        for (int i = threadIdx.x; i < a.N; i += 256) {
            atomicAdd(&rep[i], red[0][i]);
            atomicAdd(&rep[a.N + i], red[1][i]);
        }
    }
}

inline uint32_t thresh3(float p) { return p > 0.f ? (uint32_t)((double)p * 4294967296.0) : 0u; }

int pool3_launch(int mode, const float* y, const float* out4, const void* dout, const float* sums, void* out,
                 float* sums_out, void* dy, int B, int D, int H, int W, int N, int act, float drop_p, uint32_t seed,
                 const uint32_t* seed_epoch, int train, hipStream_t st) {
    MM_REQUIRE(y && out4 && B > 0 && D % 2 == 0 && H % 2 == 0 && W % 2 == 0, "pool3d_bn_act: dims must be even");
    MM_REQUIRE(N % 4 == 0 && N <= 1024, "pool3d_bn_act: N");
    Pool3Args a;
    a.y = y; a.out4 = out4; a.dout = (const bf16*)dout; a.sums = sums; a.out = (bf16*)out; a.sums_out = sums_out;
    a.dy = (bf16*)dy; a.B = B; a.D = D; a.H = H; a.W = W; a.N = N; a.act = act; a.train = train;
    a.thresh = thresh3(drop_p); a.seed = seed; a.inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    a.inv_count = 1.f / ((float)B * D * H * W);
    a.epoch = seed_epoch;
    const int rpb = 256 / (N / 4) > 0 ? 256 / (N / 4) : 1;
    const size_t rows = (size_t)B * (D / 2) * (H / 2) * (W / 2);
    int grid = (int)((rows + rpb - 1) / rpb);
    if (grid > 2048) grid = 2048;
    if (mode == 0) hipLaunchKernelGGL(pool3_bn_act_kernel<0>, dim3(grid), dim3(256), 0, st, a);
    else if (mode == 1) hipLaunchKernelGGL(pool3_bn_act_kernel<1>, dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(pool3_bn_act_kernel<2>, dim3(grid), dim3(256), 0, st, a);
    return mm_check_launch("pool3d_bn_act");
}

}  // namespace

static int g_dbg = 0;

extern "C" {

int mm_debug_flags(int flags, hipStream_t) { g_dbg = flags; return 0; }

int mm_pack_volume_bf16(const float* x, void* y, int64_t nvox, int Cp, hipStream_t st) {
    MM_REQUIRE(x && y && nvox > 0 && Cp % 8 == 0, "pack_volume: bad args");
    size_t n = (size_t)nvox * (Cp / 8);
    int grid = (int)((n + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(pack_vol_kernel, dim3(grid), dim3(256), 0, st, x, (bf16*)y, (size_t)nvox, Cp);
    return mm_check_launch("pack_volume");
}

int mm_conv3d_fwd(const void* x, const void* w, int B, int D, int H, int W, int Cin, int Cout, const float* shift,
                  float* stats, float* out_f32, void* out_bf16, hipStream_t st) {
    MM_REQUIRE(x && w && (out_f32 || out_bf16) && B > 0 && D > 0 && H > 0 && W > 0, "conv3d_fwd: null/invalid");
    MM_REQUIRE(Cin == 16 || Cin % 32 == 0, "conv3d_fwd: Cin=%d must be 16 or a multiple of 32", Cin);
    Conv3dArgs a{(const bf16*)x, (const bf16*)w, B, D, H, W, Cin, Cout, shift, g_dbg, stats, out_f32, (bf16*)out_bf16};
    const long tiles2 = (long)B * ceil_div(D, 2) * ceil_div(H, 8) * ceil_div(W, 8);
    if (Cin == WR_CIN && Cout <= WR_BN && Cout > 32 && Cout % 4 == 0 && tiles2 >= 512) return launch3d_wres(a, st);
    if (Cout <= 32) return launch3d<2, 32, 4, 1>(a, st);
    if (Cout <= 64) {
        if (tiles2 >= 256 && D % 2 == 0) return launch3d<2, 64, 4, 1>(a, st);
        return launch3d<1, 64, 2, 2>(a, st);
    }
    if (tiles2 * ceil_div(Cout, 128) >= 256 && D % 2 == 0) return launch3d<2, 128, 2, 2>(a, st);
    return launch3d<1, 64, 2, 2>(a, st);
}

int mm_conv3d_wgrad(const void* dy, const void* x, float* dw, float* dbias, int B, int D, int H, int W, int Cin,
                    int Cout, int Cin_real, int64_t sn, int64_t sc, int64_t stap, int nrep, int64_t rep_stride,
                    hipStream_t st) {
    MM_REQUIRE(dy && x && dw && B > 0, "conv3d_wgrad: null/invalid");
    MM_REQUIRE(nrep >= 1 && nrep <= 64, "conv3d_wgrad: nrep");
    MM_REQUIRE(Cin % 8 == 0 && Cout % 8 == 0 && Cin_real > 0 && Cin_real <= Cin, "conv3d_wgrad: channels");
    Wgrad3dArgs a;
    a.dy = (const bf16*)dy; a.x = (const bf16*)x; a.dw = dw; a.dbias = dbias;
    a.B = B; a.D = D; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.Cin_real = Cin_real;
    a.sn = sn; a.sc = sc; a.stap = stap; a.nrep = nrep; a.rep_stride = rep_stride;
    const int tiles_total = B * D * ceil_div(H, 8) * ceil_div(W, 8);
    const int par = ceil_div(Cout, 64) * 3 * ceil_div(Cin, 64);
    int chunks = ceil_div(384, par);
    if (chunks > tiles_total) chunks = tiles_total;
    if (chunks < 1) chunks = 1;
    a.tiles_per_wg = ceil_div(tiles_total, chunks);
    dim3 grid(ceil_div(tiles_total, a.tiles_per_wg), ceil_div(Cout, 64), 3 * ceil_div(Cin, 64));
    hipLaunchKernelGGL(conv3d_wgrad_kernel, grid, dim3(256), 0, st, a);
    return mm_check_launch("conv3d_wgrad");
}

int mm_pool3d_bn_act_fwd(const float* y, const float* out4, void* out_bf16, int B, int D, int H, int W, int N,
                         int act, float drop_p, uint32_t seed, const uint32_t* seed_epoch, hipStream_t st) {
    MM_REQUIRE(out_bf16, "pool3d_bn_act_fwd: null out");
    return pool3_launch(0, y, out4, nullptr, nullptr, out_bf16, nullptr, nullptr, B, D, H, W, N, act, drop_p, seed,
                        seed_epoch, 0, st);
}

int mm_pool3d_bn_act_bwd_reduce(const float* y, const float* out4, const void* dout_bf16, float* sums_out, int B,
                                int D, int H, int W, int N, int act, float drop_p, uint32_t seed,
                                const uint32_t* seed_epoch, hipStream_t st) {
    MM_REQUIRE(dout_bf16 && sums_out, "pool3d_bn_act_bwd_reduce: null");
    return pool3_launch(1, y, out4, dout_bf16, nullptr, nullptr, sums_out, nullptr, B, D, H, W, N, act, drop_p, seed,
                        seed_epoch, 1, st);
}

int mm_pool3d_bn_act_bwd_apply(const float* y, const float* out4, const void* dout_bf16, const float* sums, void* dy,
                               int B, int D, int H, int W, int N, int act, float drop_p, uint32_t seed,
                               const uint32_t* seed_epoch, int train, hipStream_t st) {
    MM_REQUIRE(dout_bf16 && dy && (!train || sums), "pool3d_bn_act_bwd_apply: null");
    return pool3_launch(2, y, out4, dout_bf16, sums, nullptr, nullptr, dy, B, D, H, W, N, act, drop_p, seed,
                        seed_epoch, train, st);
}

}  // extern "C"
