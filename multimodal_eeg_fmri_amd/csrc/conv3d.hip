// 3-D voxel convolution (k = 3, pad = 1, stride 1) as an LDS-staged implicit
// GEMM on bf16 MFMA, channels-last (NDHWC), fp32 accumulate.
//
//   forward : Y[b,v,n]   = sum_{tap,c} X[b, v + off(tap), c] * W[n, tap, c]
//   (data gradient = same kernel on dY with the flipped/transposed image)
//   wgrad   : dW[n,tap,c] = sum_{b,v} dY[b,v,n] * X[b, v + off(tap), c]
//
// One workgroup owns a TD x 8 x 8 block of output voxels (BM = 64*TD GEMM rows)
// of one sample.  The (TD+2) x 10 x 10 input halo block is staged ONCE per Cin
// chunk into LDS as rows of KC channels (w-pitch 12 rows); every one of the 27
// taps then reads its A fragments from that block at a row offset
// kd*120 + kh*12 + kw, so the im2col matrix only ever exists as LDS addresses.
// The weight slice of one kd plane (9 taps) sits beside it and is re-staged three
// times per chunk.  GEMM row -> voxel is (h, w) = (lr >> 3, (lr & 3) + 4 *
// parity(lr >> 2)) so that every ds_read_b128 lane group ({0-3,12-15,20-27}, ...)
// reads a 4 x 4 voxel patch: 16 rows that are distinct mod 16 on the pitch of 12,
// i.e. conflict-free with the 80-byte row stride (see the W-resident kernel below).
#include "conv3d_args.h"

namespace {

constexpr int KC3 = 32;
constexpr int KPAD3 = 8;
constexpr int HB = 10;            // halo edge of an 8-wide tile
constexpr int GWP = 12;           // halo w-pitch in LDS rows (generic kernel)
constexpr int GDP = HB * GWP;     // halo d-pitch

// GEMM row m (0..63 inside a depth slice) <-> voxel (h, w) of the 8 x 8 tile face
__device__ __forceinline__ int row_h(int m) { return (m >> 3) & 7; }
__device__ __forceinline__ int row_w(int m) { return (m & 3) + 4 * (__builtin_popcount((m >> 2) & 7) & 1); }


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
    const int kc = a.kc;
    const int AS = kc + KPAD3;
    constexpr int HROWS = (TD + 2) * HB * HB;            // rows fetched
    constexpr int LROWS = (TD + 2) * GDP;                 // rows of the LDS image
    bf16* As = reinterpret_cast<bf16*>(smem);
    bf16* Ws = As + LROWS * AS;
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
        abase[i] = ((m >> 6) * HB + row_h(m)) * GWP + row_w(m);
    }
    const bf16* xb = a.x + (size_t)b * a.D * a.H * a.W * a.Cin;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr int SB = 8;                                 // 16-byte chunks per thread per staging batch

    // a weight plane is BN * 9 rows of <= 4 segments: ONE batch of SBW loads per thread (with the halo's
    // batch size of 8 the 9 chunks of a 64-column plane took two global round trips per plane)
    constexpr int SBW = (BN * 9 * 4 + 255) / 256;
    auto load_halo = [&](int s, int c0) {
        const int r = s / segs, sg = s - r * segs;
        const int hw = r % HB, hh = (r / HB) % HB, hd = r / (HB * HB);
        const int d = d0 + hd - 1, h = h0 + hh - 1, w = w0 + hw - 1;
        u32x4 v = u32x4{0u, 0u, 0u, 0u};
        if (s < HROWS * segs && d >= 0 && d < a.D && h >= 0 && h < a.H && w >= 0 && w < a.W)
            v = *reinterpret_cast<const u32x4*>(xb + (((size_t)d * a.H + h) * a.W + w) * a.Cin + c0 + sg * 8);
        return v;
    };
    auto store_halo = [&](int s, const u32x4& v) {
        const int r = s / segs, sg = s - r * segs;
        if (s < HROWS * segs) *reinterpret_cast<u32x4*>(As + ((r / HB) * GWP + r % HB) * AS + sg * 8) = v;
    };
    auto load_w = [&](int s, int c0, int kd) {
        const int r = s / segs, sg = s - r * segs;         // r = n_local * 9 + t9
        const int nl = r / 9, t9 = r - nl * 9;
        u32x4 v = u32x4{0u, 0u, 0u, 0u};
        if (s < BN * 9 * segs && n0 + nl < a.Cout)
            v = *reinterpret_cast<const u32x4*>(a.w + ((size_t)(n0 + nl) * 27 + kd * 9 + t9) * a.Cin + c0 + sg * 8);
        return v;
    };
    auto store_w = [&](int s, const u32x4& v) {
        if (s < BN * 9 * segs) *reinterpret_cast<u32x4*>(Ws + (s / segs) * AS + (s % segs) * 8) = v;
    };
    constexpr bool ONE_HALO_BATCH = HROWS * 4 <= 256 * SB;    // the halo fits one batch: issue it with weight plane 0
    for (int c0 = 0; c0 < a.Cin; c0 += kc) {
        // staging in batches: all loads of a batch are in flight before its first LDS write
        if constexpr (ONE_HALO_BATCH) {
            u32x4 vh[SB], vw[SBW];
#pragma unroll
            for (int q = 0; q < SB; ++q) vh[q] = load_halo(q * 256 + tid, c0);
#pragma unroll
            for (int q = 0; q < SBW; ++q) vw[q] = load_w(q * 256 + tid, c0, 0);
#pragma unroll
            for (int q = 0; q < SB; ++q) store_halo(q * 256 + tid, vh[q]);
#pragma unroll
            for (int q = 0; q < SBW; ++q) store_w(q * 256 + tid, vw[q]);
        } else {
            for (int s0 = 0; s0 < HROWS * segs; s0 += 256 * SB) {
                u32x4 v[SB];
#pragma unroll
                for (int q = 0; q < SB; ++q) v[q] = load_halo(s0 + q * 256 + tid, c0);
#pragma unroll
                for (int q = 0; q < SB; ++q) store_halo(s0 + q * 256 + tid, v[q]);
            }
        }
        if constexpr (!ONE_HALO_BATCH) {
            u32x4 v[SBW];
#pragma unroll
            for (int q = 0; q < SBW; ++q) v[q] = load_w(q * 256 + tid, c0, 0);
#pragma unroll
            for (int q = 0; q < SBW; ++q) store_w(q * 256 + tid, v[q]);
        }
        for (int kd = 0; kd < 3; ++kd) {
            __syncthreads();                               // plane kd (and the halo) are in LDS
            // the NEXT plane's global loads are in flight during this plane's 9 x kc/16 MFMA steps
            u32x4 vn[SBW];
            if (kd < 2)
#pragma unroll
                for (int q = 0; q < SBW; ++q) vn[q] = load_w(q * 256 + tid, c0, kd + 1);
#pragma unroll
            for (int t9 = 0; t9 < 9; ++t9) {
                const int toff = kd * GDP + (t9 / 3) * GWP + (t9 % 3);
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
            if (kd < 2) {
                __syncthreads();                           // every wave is done reading plane kd
#pragma unroll
                for (int q = 0; q < SBW; ++q) store_w(q * 256 + tid, vn[q]);
            }
        }
        __syncthreads();
    }

    float* sstat = reinterpret_cast<float*>(smem);            // [WM][2][BN] per-wave-row partial sums (plain stores)
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
                const int d = d0 + (m >> 6), h = h0 + row_h(m), w = w0 + row_w(m);
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
                sstat[(wm * 2 + 0) * BN + nl] = s1;
                sstat[(wm * 2 + 1) * BN + nl] = s2;
            }
        }
    }
    if (a.stats) {
        __syncthreads();
        mm_acc_t* rep = acc_rep(a.stats, blockIdx.x % MM_ACC_REPL, 2 * (size_t)a.Cout);
        for (int i = tid; i < 2 * BN; i += 256) {
            const int which = i / BN, col = i % BN;
            if (n0 + col < a.Cout) {
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < WM; ++w) t += sstat[(w * 2 + which) * BN + col];
                acc_add<MM_ACC_STAT>(&rep[which * a.Cout + n0 + col], t);
            }
        }
    }
}

template <int TD, int BN, int WM, int WN>
int launch3d(Conv3dArgs a, hipStream_t st) {
    // channels per chunk stay at 32: 64-channel chunks halve the re-stagings but the bigger image
    // costs residency (measured: L3 forward 30 -> 45 us, L2 data gradient 39 -> 78 us)
    auto lds_for = [](int kc) { return (size_t)((TD + 2) * GDP + BN * 9) * (kc + KPAD3) * sizeof(bf16); };
    const int kc = a.Cin < KC3 ? a.Cin : KC3;
    a.kc = kc;
    const size_t lds = lds_for(kc);
    if (lds > 160 * 1024) return mm_fail(MM_ERR_UNSUPPORTED, "conv3d_fwd: LDS %zu", lds);
    auto kern = conv3d_fwd_kernel<TD, BN, WM, WN>;
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid(a.B * ceil_div(a.D, TD) * ceil_div(a.H, 8) * ceil_div(a.W, 8), ceil_div(a.Cout, BN));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
    return mm_check_launch("conv3d_fwd");
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
// ---------------------------------------------------------------------------
// BatchNorm3d -> act -> MaxPool3d(2) -> dropout and its backward, on the fp32
// pre-BN volume [B][D][H][W][N] (HBM-bound: 33.5 MB in at layer 2).
//
// Every supported activation is quasi-convex in z (GELU falls to z = -0.75 and
// rises after; the others are monotone), so max_j act(z_j) over a 2x2x2 window is
// act(max z) or act(min z): two activations per pooled element instead of eight.
// The forward keeps the winner's pre-BN value (`ysel`, fp32) and window index
// (`arg`, one byte), so that
//   * the BN-gradient reduction reads 6 MB of pooled data instead of the volume
//     (gradients are zero everywhere but at the winners), and
//   * the apply pass evaluates one act' per pooled element and never re-derives
//     the argmax.
// One thread = 4 channels of one pooled voxel (8 x 16-byte loads).
// ---------------------------------------------------------------------------
struct Pool3Args {
    const bf16* y; const float* out4; const bf16* dout; const float* sums;
    bf16* out; float* sums_out; bf16* dy;
    bf16* ysel; uint8_t* arg;                  // pooled [B][D/2][H/2][W/2][N]
    int B, D, H, W, N, act, train;
    uint32_t thresh, seed; float inv_keep, inv_count;
    const uint32_t* epoch;
    int sums_nrep;                             // apply: 1 = compact fp32 sums, MM_REPL = the reduce pass's accumulator workspace
    MmBnFin fin;                               // forward with FIN: the layer's BatchNorm finalize runs in the prologue (out4 is then written, not read)
};

// One thread = 8 channels of one pooled voxel: eight 16-byte loads of the bf16 pre-BatchNorm volume.
// FIN (forward): the train-mode BatchNorm finalize as the kernel's prologue (csrc/common.h: bn_fin_channel; N <= 256)
template <int MODE, int ACT = -1, bool FIN = false>   // MODE 0 fwd, 2 bwd-apply; ACT >= 0: compiled for that activation (no per-element switch)
__global__ __launch_bounds__(256) void pool3_bn_act_kernel(Pool3Args a) {
    if (ACT >= 0) a.act = ACT;
    a.seed = mm_eff_seed(a.seed, a.epoch);
    __shared__ float s_fin[FIN ? 4 : 1][FIN ? 256 : 1];
    if (FIN) {
        for (int n = threadIdx.x; n < a.N; n += 256)
            bn_fin_channel(a.fin, n, blockIdx.x == 0, s_fin[0][n], s_fin[1][n], s_fin[2][n], s_fin[3][n]);
        __syncthreads();
    }
    const int nv = a.N / 8;
    const int Do = a.D / 2, Ho = a.H / 2, Wo = a.W / 2;
    const size_t nrows = (size_t)a.B * Do * Ho * Wo;
    const int rows_per_blk = 256 / nv > 0 ? 256 / nv : 1;
    const int vi = threadIdx.x % nv, ri = threadIdx.x / nv;
    __shared__ float csum[MODE == 2 ? 2048 : 1];       // sum dz | sum dz * xhat per channel (N <= 1024)
    if (MODE == 2 && a.train) {
        // the workspace's replicas are summed here (a separate compaction launch sat between the two passes)
        for (int i = threadIdx.x; i < 2 * a.N; i += 256)
            csum[i] = a.sums_nrep == 1 ? a.sums[i] : acc_val<MM_ACC_GRAD>(acc_sum(a.sums, 2 * (size_t)a.N, i));
        __syncthreads();
    }
    if (ri >= rows_per_blk) return;
    const int n8 = vi * 8;
    float sc[8], sh[8], mu[8], rs[8], c0[8], c1[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        if (FIN) {
            sc[q] = s_fin[0][n8 + q]; sh[q] = s_fin[1][n8 + q]; mu[q] = s_fin[2][n8 + q]; rs[q] = s_fin[3][n8 + q];
        } else {
            sc[q] = a.out4[n8 + q]; sh[q] = a.out4[a.N + n8 + q];
            mu[q] = a.out4[2 * a.N + n8 + q]; rs[q] = a.out4[3 * a.N + n8 + q];
        }
        c0[q] = (MODE == 2 && a.train) ? csum[n8 + q] * a.inv_count : 0.f;
        c1[q] = (MODE == 2 && a.train) ? csum[a.N + n8 + q] * a.inv_count : 0.f;
    }
    for (size_t row = (size_t)blockIdx.x * rows_per_blk + ri; row < nrows; row += (size_t)gridDim.x * rows_per_blk) {
        unsigned q = (unsigned)row;                                 // 32-bit divisions (rows < 2^31: checked on the host)
        const int ow = (int)(q % (unsigned)Wo); q /= (unsigned)Wo;
        const int oh = (int)(q % (unsigned)Ho); q /= (unsigned)Ho;
        const int od = (int)(q % (unsigned)Do); q /= (unsigned)Do;
        const size_t base = ((((size_t)q * a.D + 2 * od) * a.H + 2 * oh) * a.W + 2 * ow) * a.N + n8;
        const size_t sw = a.N, shh = (size_t)a.W * a.N, sd = (size_t)a.H * a.W * a.N;
        bf16x8 t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            t[j] = *reinterpret_cast<const bf16x8*>(a.y + base + (j >> 2) * sd + ((j >> 1) & 1) * shh + (j & 1) * sw);
        const size_t oidx = row * a.N + n8;
        if (MODE == 0) {
            bf16x8 o, ys;
            uint32_t args[2] = {0u, 0u};
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                float zmax = -INFINITY, zmin = INFINITY;
                bf16 ymax = (bf16)0.f, ymin = (bf16)0.f;
                int jmax = 0, jmin = 0;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bf16 yv = t[j][c];
                    const float z = (float)yv * sc[c] + sh[c];
                    if (z > zmax) { zmax = z; jmax = j; ymax = yv; }       // strict: first occurrence wins, as PyTorch
                    if (z < zmin) { zmin = z; jmin = j; ymin = yv; }
                }
                // GELU: with zmax >= 0 the largest pre-activation holds the largest activation (conv3d_l1.hip); the second
                // evaluation is needed for an all-negative window only
                const float amax = apply_act(zmax, a.act);
                bool lo = false;
                float v = amax;
                if (a.act != MM_ACT_GELU || zmax < 0.f) {
                    const float amin = apply_act(zmin, a.act);
                    lo = amin > amax;
                    v = lo ? amin : amax;
                }
                ys[c] = lo ? ymin : ymax;
                args[c >> 2] |= (uint32_t)(lo ? jmin : jmax) << (8 * (c & 3));
                if (a.thresh) v *= dropout_scale(a.seed, (uint32_t)(oidx + c), a.thresh, a.inv_keep);
                o[c] = (bf16)v;
            }
            *reinterpret_cast<bf16x8*>(a.out + oidx) = o;
            if (a.ysel) {
                *reinterpret_cast<bf16x8*>(a.ysel + oidx) = ys;
                *reinterpret_cast<uint2*>(a.arg + oidx) = make_uint2(args[0], args[1]);
            }
        } else {
            const bf16x8 gv = *reinterpret_cast<const bf16x8*>(a.dout + oidx);
            const uint2 args = *reinterpret_cast<const uint2*>(a.arg + oidx);
            float dzs[8];
            int arg[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                arg[c] = ((c < 4 ? args.x : args.y) >> (8 * (c & 3))) & 7;
                float ysel = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) ysel = (j == arg[c]) ? (float)t[j][c] : ysel;
                float g = (float)gv[c];
                if (a.thresh) g *= dropout_scale(a.seed, (uint32_t)(oidx + c), a.thresh, a.inv_keep);
                dzs[c] = g * act_grad(ysel * sc[c] + sh[c], a.act);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                bf16x8 o;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float dz = (j == arg[c]) ? dzs[c] : 0.f;
                    const float xh = ((float)t[j][c] - mu[c]) * rs[c];
                    o[c] = (bf16)(a.train ? sc[c] * (dz - c0[c] - xh * c1[c]) : sc[c] * dz);
                }
                *reinterpret_cast<bf16x8*>(a.dy + base + (j >> 2) * sd + ((j >> 1) & 1) * shh + (j & 1) * sw) = o;
            }
        }
    }
}

// BN-gradient partial sums from the pooled winners only:  sums[0][n] += dz, sums[1][n] += dz * xhat
template <int ACT = -1>
__global__ __launch_bounds__(256) void pool3_bwd_reduce_kernel(Pool3Args a) {
    if (ACT >= 0) a.act = ACT;
    a.seed = mm_eff_seed(a.seed, a.epoch);
    const int nv = a.N / 8;
    const size_t nrows = (size_t)a.B * (a.D / 2) * (a.H / 2) * (a.W / 2);
    const int rows_per_blk = 256 / nv > 0 ? 256 / nv : 1;
    const int vi = threadIdx.x % nv, ri = threadIdx.x / nv;
    const bool active = ri < rows_per_blk;
    const int n8 = vi * 8;
    float s0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (active) {
        float sc[8], sh[8], mu[8], rs[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            sc[q] = a.out4[n8 + q]; sh[q] = a.out4[a.N + n8 + q];
            mu[q] = a.out4[2 * a.N + n8 + q]; rs[q] = a.out4[3 * a.N + n8 + q];
        }
        for (size_t row = (size_t)blockIdx.x * rows_per_blk + ri; row < nrows; row += (size_t)gridDim.x * rows_per_blk) {
            const size_t oidx = row * a.N + n8;
            const bf16x8 ys = *reinterpret_cast<const bf16x8*>(a.ysel + oidx);
            const bf16x8 gv = *reinterpret_cast<const bf16x8*>(a.dout + oidx);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float yv = (float)ys[c];
                float g = (float)gv[c];
                if (a.thresh) g *= dropout_scale(a.seed, (uint32_t)(oidx + c), a.thresh, a.inv_keep);
                const float dz = g * act_grad(yv * sc[c] + sh[c], a.act);
                s0[c] += dz;
                s1[c] += dz * (yv - mu[c]) * rs[c];
            }
        }
    }
    // plain stores + column walk (LDS float atomics with rows_per_blk-way same-address conflicts are slow)
    __shared__ __attribute__((aligned(16))) float part[4096];        // [rows_per_blk][2][N]
    if (active) {
        float* dst = part + (size_t)ri * 2 * a.N + n8;
#pragma unroll
        for (int c = 0; c < 8; ++c) { dst[c] = s0[c]; dst[a.N + c] = s1[c]; }
    }
    __syncthreads();
    mm_acc_t* rep = acc_rep(a.sums_out, blockIdx.x % MM_ACC_REPL, 2 * (size_t)a.N);
    for (int i = threadIdx.x; i < 2 * a.N; i += 256) {
        float s = 0.f;
        for (int r = 0; r < rows_per_blk; ++r) s += part[r * 2 * a.N + i];
        acc_add<MM_ACC_GRAD>(&rep[i], s);
    }
}

inline uint32_t thresh3(float p) { return p > 0.f ? (uint32_t)((double)p * 4294967296.0) : 0u; }

int pool3_launch(int mode, Pool3Args a, float drop_p, hipStream_t st, bool with_fin = false) {
    MM_REQUIRE(a.out4 && a.B > 0 && a.D % 2 == 0 && a.H % 2 == 0 && a.W % 2 == 0, "pool3d_bn_act: dims must be even");
    MM_REQUIRE(a.N % 8 == 0 && a.N <= 1024, "pool3d_bn_act: N must be a multiple of 8");
    a.thresh = thresh3(drop_p); a.inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    a.inv_count = 1.f / ((float)a.B * a.D * a.H * a.W);
    const int rpb = 256 / (a.N / 8) > 0 ? 256 / (a.N / 8) : 1;
    const size_t rows = (size_t)a.B * (a.D / 2) * (a.H / 2) * (a.W / 2);
    MM_REQUIRE(rows < (1ull << 31), "pool3d_bn_act: %zu pooled voxels (32-bit indices)", rows);
    int grid = (int)((rows + rpb - 1) / rpb);
    if (mode == 1) {
        if (grid > 512) grid = 512;
        if (a.act == MM_ACT_GELU) hipLaunchKernelGGL((pool3_bwd_reduce_kernel<MM_ACT_GELU>), dim3(grid), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((pool3_bwd_reduce_kernel<>), dim3(grid), dim3(256), 0, st, a);
    } else {
        if (grid > 4096) grid = 4096;
        const bool gelu = a.act == MM_ACT_GELU;
        if (with_fin) {                                  // (mode 0, GELU, N <= 256: checked by the caller) few, longer workgroups:
            if (grid > 768) grid = 768;                  // each re-reads the statistics workspace in its prologue: three per CU
            hipLaunchKernelGGL((pool3_bn_act_kernel<0, MM_ACT_GELU, true>), dim3(grid), dim3(256), 0, st, a);
        } else if (mode == 0 && gelu) hipLaunchKernelGGL((pool3_bn_act_kernel<0, MM_ACT_GELU>), dim3(grid), dim3(256), 0, st, a);
        else if (mode == 0) hipLaunchKernelGGL((pool3_bn_act_kernel<0>), dim3(grid), dim3(256), 0, st, a);
        else if (gelu) hipLaunchKernelGGL((pool3_bn_act_kernel<2, MM_ACT_GELU>), dim3(grid), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((pool3_bn_act_kernel<2>), dim3(grid), dim3(256), 0, st, a);
    }
    return mm_check_launch("pool3d_bn_act");
}

}  // namespace

namespace {
__global__ void stamp_kernel(long long* buf, int idx) { buf[idx] = wall_clock64(); }
}

extern "C" {

// one 100 MHz wall-clock stamp written in stream order (a graph node like any other kernel):
// the only way to see where a hipGraph-replayed step spends its time without a profiler attached
int mm_debug_stamp(void* buf, int idx, hipStream_t st) {
    MM_REQUIRE(buf && idx >= 0, "debug_stamp: bad args");
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, st, (long long*)buf, idx);
    return mm_check_launch("debug_stamp");
}

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
    Conv3dArgs a{(const bf16*)x, (const bf16*)w, B, D, H, W, Cin, Cout, shift, 0, stats, out_f32, (bf16*)out_bf16, 0u, 0u, 0u};
    const long tiles2 = (long)B * ceil_div(D, 2) * ceil_div(H, 8) * ceil_div(W, 8);
    if (conv3d_wres_applies(a)) return launch3d_wres(a, st);          // Cin 32 -> Cout 64, bf16 out: conv3d_wres.hip
    if (conv3d_stream_applies(a)) return launch3d_stream(a, st);      // 64 -> 128, 128 -> 64, 64 -> 32: conv3d_stream.hip
    // mm_prep_conv_weight stores the weight image of these shapes in the streaming kernel's lane order (common.h:
    // conv_image_index keys on the shape alone); the generic kernel below reads the plain [n][tap][c] image and would
    // silently compute with scrambled weights
    if (conv3d_stream_shape(Cout, Cin))
        return mm_fail(MM_ERR_UNSUPPORTED, "conv3d_fwd: Cin=%d Cout=%d volumes of %d x %d x %d voxels exceed the streaming kernel's "
                                           "32-bit per-sample offsets (its weight image is in that kernel's lane order)", Cin, Cout, D, H, W);
    if (Cout <= 32) return launch3d<2, 32, 4, 1>(a, st);
    if (Cout <= 64) {
        if (tiles2 >= 256 && D % 2 == 0) return launch3d<2, 64, 4, 1>(a, st);
        return launch3d<1, 64, 2, 2>(a, st);
    }
    if (tiles2 * ceil_div(Cout, 128) >= 256 && D % 2 == 0) return launch3d<2, 128, 2, 2>(a, st);
    return launch3d<1, 64, 2, 2>(a, st);
}

int mm_pool3d_bn_act_fwd(const void* y, const float* out4, void* out_bf16, void* ysel, void* arg, int B, int D,
                         int H, int W, int N, int act, float drop_p, uint32_t seed, const uint32_t* seed_epoch,
                         hipStream_t st) {
    MM_REQUIRE(y && out_bf16 && (!ysel == !arg), "pool3d_bn_act_fwd: null out / ysel and arg go together");
    Pool3Args a{};
    a.y = (const bf16*)y; a.out4 = out4; a.out = (bf16*)out_bf16; a.ysel = (bf16*)ysel; a.arg = (uint8_t*)arg;
    a.B = B; a.D = D; a.H = H; a.W = W; a.N = N; a.act = act; a.seed = seed; a.epoch = seed_epoch;
    return pool3_launch(0, a, drop_p, st);
}

int mm_pool3d_bn_act_fwd_fin(const void* y, const void* bn_fin_host, void* out_bf16, void* ysel, void* arg, int B, int D,
                             int H, int W, int N, int act, float drop_p, uint32_t seed, const uint32_t* seed_epoch,
                             hipStream_t st) {
    MM_REQUIRE(y && out_bf16 && (!ysel == !arg), "pool3d_bn_act_fwd_fin: null out / ysel and arg go together");
    MM_REQUIRE(N <= 256 && act == MM_ACT_GELU, "pool3d_bn_act_fwd_fin: N=%d (<= 256), GELU only", N);
    Pool3Args a{};
    MM_REQUIRE(bn_fin_from_host(a.fin, bn_fin_host, N), "pool3d_bn_act_fwd_fin: incomplete mm_bn_fin_t (null pointer or count < 1)");
    a.y = (const bf16*)y; a.out4 = a.fin.out4; a.out = (bf16*)out_bf16; a.ysel = (bf16*)ysel; a.arg = (uint8_t*)arg;
    a.B = B; a.D = D; a.H = H; a.W = W; a.N = N; a.act = act; a.seed = seed; a.epoch = seed_epoch;
    return pool3_launch(0, a, drop_p, st, true);
}

int mm_pool3d_bn_act_bwd_reduce(const void* ysel, const float* out4, const void* dout_bf16, float* sums_out, int B,
                                int D, int H, int W, int N, int act, float drop_p, uint32_t seed,
                                const uint32_t* seed_epoch, hipStream_t st) {
    MM_REQUIRE(ysel && dout_bf16 && sums_out, "pool3d_bn_act_bwd_reduce: null");
    Pool3Args a{};
    a.ysel = (bf16*)const_cast<void*>(ysel); a.out4 = out4; a.dout = (const bf16*)dout_bf16; a.sums_out = sums_out;
    a.B = B; a.D = D; a.H = H; a.W = W; a.N = N; a.act = act; a.seed = seed; a.epoch = seed_epoch; a.train = 1;
    return pool3_launch(1, a, drop_p, st);
}

int mm_pool3d_bn_act_bwd_apply(const void* y, const void* arg, const float* out4, const void* dout_bf16,
                               const float* sums, void* dy, int B, int D, int H, int W, int N, int act, float drop_p,
                               uint32_t seed, const uint32_t* seed_epoch, int train, int sums_nrep, hipStream_t st) {
    MM_REQUIRE(y && arg && dout_bf16 && dy && (!train || sums), "pool3d_bn_act_bwd_apply: null");
    MM_REQUIRE(sums_nrep == 1 || sums_nrep == MM_REPL, "pool3d_bn_act_bwd_apply: sums_nrep = 1 (compact fp32) or %d (the reduce pass's workspace)", MM_REPL);
    MM_REQUIRE(N <= 1024, "pool3d_bn_act_bwd_apply: N=%d > 1024", N);
    Pool3Args a{};
    a.y = (const bf16*)y; a.arg = (uint8_t*)const_cast<void*>(arg); a.out4 = out4; a.dout = (const bf16*)dout_bf16; a.sums = sums;
    a.dy = (bf16*)dy; a.B = B; a.D = D; a.H = H; a.W = W; a.N = N; a.act = act; a.seed = seed; a.epoch = seed_epoch;
    a.train = train; a.sums_nrep = sums_nrep;
    return pool3_launch(2, a, drop_p, st);
}

}  // extern "C"
