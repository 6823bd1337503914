// First layer of the 3-D voxel encoder: Conv3d(1 -> 32, k3, p1) + BatchNorm3d +
// GELU + MaxPool3d(2) [+ Dropout] on single-channel fp32 volumes, fused.
//
// With Cin = 1 the GEMM K is just the 27 taps, so the layer is HBM/VALU-bound,
// not MFMA-bound.  The pre-BN activation (32x the input, 134 MB fp32 at the C2
// config) is never written: every pass recomputes the convolution from the
// 4 MB input (8 bf16 MFMAs per 128 voxels) and keeps only per-channel sums:
//   mode 0  stats        : sum / sumsq of y = conv + bias           (train fwd 1)
//   mode 1  apply        : BN -> GELU -> 2x2x2 max -> dropout -> bf16 (fwd 2 / eval)
//   mode 2  bwd reduce   : S1 = sum dz, S2 = sum dz*xhat
//   mode 3  bwd apply    : dy = BN'(dz) for every voxel, dW[tap][n] += x^T dy,
//                          dbias += sum dy     (second MFMA: the dy accumulator
//                          tile is the B operand, the im2col gather the A operand)
//   mode 4  bwd, one pass: BN' is linear in the two sums of mode 2, so the weight gradient is
//                          dW = sc (A1 - c0 S - c1 A3) with A1 = x^T dz, A3 = x^T xhat,
//                          S[tap] = sum_v x[v + tap] (l1_tapsum_kernel), c0 = S1/M, c1 = S2/M.
//                          One recompute of the convolution yields S1, S2, A1 and A3 (the two
//                          products share their gathered A fragments); l1_combine_kernel finishes.
//                          Replaces modes 2 + 3 (two recomputes) in training.
// One wave owns a 2 x 8 x 8 block of conv outputs (= 1 x 4 x 4 pooled voxels) x
// 32 channels: all 8 members of every pooling window sit in the same lane.
// GELU is evaluated once per window when max(z) >= 0 (GELU is monotone on
// [-0.7518, inf) and negative left of 0, so the window max is GELU(max z));
// otherwise on all 8 members.  Both branches are exact.
#include "common.h"

namespace {

constexpr int HP = 36;                 // halo row pitch (34 used)
constexpr int HROWS1 = 4 * 10;         // (2+2) depth x (8+2) height rows
constexpr int HSZ = HROWS1 * HP;

struct L1Args {
    const float* x;        // [B][D][H][W]
    const bf16* wimg;      // [32][32] (n, tap; taps 27..31 zero)
    const float* bias;     // [32] or nullptr (eval: folded into out4 shift)
    const float* out4;     // [4][32] scale, shift, mean, rstd  (modes 1-3)
    const bf16* dout;      // [B][D/2][H/2][W/2][32]            (modes 2-3)
    const float* sums;     // [2][32]                           (mode 3)
    float* stats;          // mode 0: [2][32];  mode 2: sums_out
    bf16* out;             // mode 1
    uint8_t* arg;          // mode 1, optional: the pooling window's winner j = (dd << 2) | (hh << 1) | ww per output element
    float* dw;             // mode 3: [27][32] (tap-major, channel-contiguous atomics); mode 4: A1
    float* dw3;            // mode 4: A3, same layout
    float* dbias;          // mode 3
    int B, D, H, W, train;
    uint32_t thresh, seed; float inv_keep, inv_count;
    const uint32_t* epoch;
};

__device__ __forceinline__ int tap_off(int tap) {          // halo offset of tap (0..26), pads -> 0
    if (tap >= 27) return 0;
    const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
    return kd * 10 * HP + kh * HP + kw;
}

// FULLT: H % 8 == 0 and W % 32 == 0, i.e. every 2 x 8 x 32 tile lies inside the volume: the per-voxel bounds tests
// (three compares and the index arithmetic behind them, per voxel and channel) are compiled out
template <int MODE, bool FULLT = false>
__global__ __launch_bounds__(256, 2) void conv3d_l1_kernel(L1Args a) {     // two waves per SIMD: <= 256 registers
    a.seed = mm_eff_seed(a.seed, a.epoch);
    __shared__ __attribute__((aligned(16))) unsigned short halo[HSZ];
    __shared__ float red[4][32];
    __shared__ float wred[4][27][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int tw = (a.W + 31) / 32, th = (a.H + 7) / 8, td = a.D / 2;
    const int ntiles = a.B * td * th * tw;
    const int Do = a.D / 2, Ho = a.H / 2, Wo = a.W / 2;

    // B fragments of the forward product: W[n = lr][k = 16s + 8lh + j]
    bf16x8 wf[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) wf[s] = *reinterpret_cast<const bf16x8*>(a.wimg + lr * 32 + 16 * s + 8 * lh);
    // gather offsets for the forward A fragments (row = voxel, k = tap)
    int foff[2][8];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) foff[s][j] = tap_off(16 * s + 8 * lh + j);
    const float bias = a.bias ? a.bias[lr] : 0.f;
    float sc = 1.f, sh = 0.f, mu = 0.f, rs = 1.f, c0 = 0.f, c1 = 0.f;
    if (MODE >= 1) {
        sc = a.out4[lr]; sh = a.out4[32 + lr]; mu = a.out4[64 + lr]; rs = a.out4[96 + lr];
        if (MODE == 3 && a.train) { c0 = a.sums[lr] * a.inv_count; c1 = a.sums[32 + lr] * a.inv_count; }   // compact sums
    }
    float acc1 = 0.f, acc2 = 0.f;          // per-lane channel sums (modes 0, 2, 4) / dbias (mode 3)
    f32x16 dwacc, dwacc3;                  // modes 3, 4: D[tap][n]
#pragma unroll
    for (int r = 0; r < 16; ++r) { dwacc[r] = 0.f; dwacc3[r] = 0.f; }
    const int my_tap_off = tap_off(lr);    // mode 3: A row = tap lr

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int q = tile;
        const int w0 = (q % tw) * 32; q /= tw;
        const int h0 = (q % th) * 8; q /= th;
        const int d0 = (q % td) * 2; q /= td;
        const int b = q;
        __syncthreads();
        {   // all loads of the halo are in flight before the first LDS write (a load -> store loop
            // pays one global round trip per pass: six per tile)
            constexpr int NH = (HROWS1 * 34 + 255) / 256;
            float hv[NH];
#pragma unroll
            for (int q = 0; q < NH; ++q) {
                const int i = tid + q * 256;
                const int hw = i % 34, hr = i / 34;
                const int hd = hr / 10, hh = hr % 10;
                const int d = d0 + hd - 1, h = h0 + hh - 1, w = w0 + hw - 1;
                hv[q] = 0.f;
                if (i < HROWS1 * 34 && d >= 0 && d < a.D && h >= 0 && h < a.H && w >= 0 && w < a.W)
                    hv[q] = a.x[(((size_t)b * a.D + d) * a.H + h) * a.W + w];
            }
#pragma unroll
            for (int q = 0; q < NH; ++q) {
                const int i = tid + q * 256;
                const bf16 hb = (bf16)hv[q];
                if (i < HROWS1 * 34) halo[(i / 34) * HP + i % 34] = *reinterpret_cast<const unsigned short*>(&hb);
            }
        }
        __syncthreads();
        const int wbase = 8 * wave;                         // this wave's w-block inside the tile
        // ---- conv: acc[i] (rows m = 32 i + ..., voxel = (m>>6, (m>>3)&7, m&7))
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
            const int m = 32 * i + lr;
            const int vb = ((m >> 6) * 10 + ((m >> 3) & 7)) * HP + (m & 7) + wbase;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                union { unsigned short u[8]; bf16x8 v; } fr;
#pragma unroll
                for (int j = 0; j < 8; ++j) fr.u[j] = halo[vb + foff[s][j]];
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr.v, wf[s], acc[i], 0, 0, 0);
            }
        }
        // lane owns channel n = lr; register r of tile i is voxel
        //   dz = i >> 1, hy = 4 (i & 1) + (r >> 2), wx = (r & 3) + 4 lh
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int d = d0 + (i >> 1), h = h0 + 4 * (i & 1) + (r >> 2), w = w0 + wbase + (r & 3) + 4 * lh;
                    if (FULLT || (d < a.D && h < a.H && w < a.W)) {
                        const float y = acc[i][r] + bias;
                        acc1 += y; acc2 += y * y;
                    }
                }
            continue;
        }
        // ---- pooled windows: (ip, ra, rb) -> regs {r0, r0+1, r0+4, r0+5} of tiles ip and ip+2
        // dy (mode 3) / dz and xhat (mode 4) fragments of the two conv tiles ip, ip + 2 that one pass of
        // the ip loop completes; their weight-gradient MFMAs run at the end of that pass, so only two
        // tiles' fragments are ever live (all four: 316 VGPRs in mode 4 = one wave per SIMD)
        bf16x8 dyf[2][2];
        bf16x8 xhf[2][2];
#pragma unroll
        for (int ip = 0; ip < 2; ++ip) {
#pragma unroll
            for (int ra = 0; ra < 2; ++ra)
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    const int r0 = 8 * ra + 2 * rb;
                    const int oh = (h0 >> 1) + 2 * ip + ra, ow = (w0 >> 1) + 4 * wave + rb + 2 * lh, od = d0 >> 1;
                    const bool ok = FULLT || (oh < Ho && ow < Wo);
                    float y[8], z[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {           // j = (dd << 2) | (hh << 1) | ww
                        const int ti = ip + 2 * (j >> 2), r = r0 + 4 * ((j >> 1) & 1) + (j & 1);
                        y[j] = acc[ti][r] + bias;
                        z[j] = y[j] * sc + sh;
                    }
                    // GELU falls on (-inf, -0.75] and rises after it, so the window's largest activation sits at its largest
                    // or at its smallest pre-activation (as pool3_bn_act).  With zmax >= 0 it is the largest: GELU(zmax) >= 0
                    // and anything below it is smaller (rising branch) or negative.  Only an all-negative window (1 in 256
                    // for unit-normal pre-activations) needs the two evaluations - the backward modes then evaluate none
                    // to find the winner, the forward one.
                    // The winner is the FIRST member that equals the extreme value (as PyTorch's max-pool): a max3 tree and
                    // eight equality tests whose first-hit bookkeeping is lane-mask (scalar) work - tracking value, y and
                    // index through eight compare / select steps for both extremes was a third of the VALU stream.
                    float zsel = fmaxf(fmaxf(fmaxf(z[0], z[1]), fmaxf(z[2], z[3])), fmaxf(fmaxf(z[4], z[5]), fmaxf(z[6], z[7])));
                    float best = MODE == 1 ? gelu_erf(zsel) : 0.f;
                    if (zsel < 0.f) {
                        const float zmin = fminf(fminf(fminf(z[0], z[1]), fminf(z[2], z[3])), fminf(fminf(z[4], z[5]), fminf(z[6], z[7])));
                        if (MODE != 1) best = gelu_erf(zsel);
                        const float amin = gelu_erf(zmin);
                        if (amin > best) { best = amin; zsel = zmin; }
                    }
                    bool hit[8], found = false;
                    float ys = y[0];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        hit[j] = !found && z[j] == zsel;
                        found = found || hit[j];
                        if (hit[j]) ys = y[j];
                    }
                    const float zs = zsel;
                    const size_t oidx = ((((size_t)b * Do + od) * Ho + oh) * Wo + ow) * 32 + lr;
                    if (MODE == 1) {
                        if (ok) {
                            if (a.thresh) best *= dropout_scale(a.seed, (uint32_t)oidx, a.thresh, a.inv_keep);
                            a.out[oidx] = (bf16)best;
                            if (a.arg) {                        // inspection output (parity tests): which member won
                                int js = 0;
#pragma unroll
                                for (int j = 1; j < 8; ++j) js = hit[j] ? j : js;
                                a.arg[oidx] = (uint8_t)js;
                            }
                        }
                    } else {
                        float g = ok ? (float)a.dout[oidx] : 0.f;
                        if (a.thresh) g *= dropout_scale(a.seed, (uint32_t)oidx, a.thresh, a.inv_keep);
                        const float dzs = g * gelu_erf_grad(zs);
                        if (MODE == 2 || MODE == 4) {
                            acc1 += dzs;
                            acc2 += dzs * (ys - mu) * rs;
                        }
                        if (MODE == 4) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                const int ti = ip + 2 * (j >> 2), r = r0 + 4 * ((j >> 1) & 1) + (j & 1);
                                const int d = d0 + (ti >> 1), h = h0 + 4 * (ti & 1) + (r >> 2), w = w0 + wbase + (r & 3) + 4 * lh;
                                const bool in = FULLT || (d < a.D && h < a.H && w < a.W);
                                dyf[ti >> 1][r >> 3][r & 7] = (bf16)((in && hit[j]) ? dzs : 0.f);
                                xhf[ti >> 1][r >> 3][r & 7] = (bf16)(in ? (y[j] - mu) * rs : 0.f);
                            }
                        } else if (MODE == 3) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                const int ti = ip + 2 * (j >> 2), r = r0 + 4 * ((j >> 1) & 1) + (j & 1);
                                const float dzj = hit[j] ? dzs : 0.f;
                                float dy = a.train ? sc * (dzj - c0 - (y[j] - mu) * rs * c1) : sc * dzj;
                                const int d = d0 + (ti >> 1), h = h0 + 4 * (ti & 1) + (r >> 2), w = w0 + wbase + (r & 3) + 4 * lh;
                                if (!FULLT && !(d < a.D && h < a.H && w < a.W)) dy = 0.f;
                                acc1 += dy;
                                dyf[ti >> 1][r >> 3][r & 7] = (bf16)dy;
                            }
                        }
                    }
                }
            if (MODE == 3 || MODE == 4) {
                // dW[tap][n] += sum_v xcol[tap][v] * dy[v][n]: A row = tap lr, k-th element of
                // half lh is voxel row 16 s + 8 (j >> 2) + 4 lh + (j & 3) of tile ti
#pragma unroll
                for (int tq = 0; tq < 2; ++tq)
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const int ti = ip + 2 * tq;
                        union { unsigned short u[8]; bf16x8 v; } fr;
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int m = 32 * ti + 16 * s + 8 * (j >> 2) + 4 * lh + (j & 3);
                            const int vb = ((m >> 6) * 10 + ((m >> 3) & 7)) * HP + (m & 7) + wbase;
                            fr.u[j] = halo[vb + my_tap_off];
                        }
                        dwacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr.v, dyf[tq][s], dwacc, 0, 0, 0);
                        if (MODE == 4) dwacc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr.v, xhf[tq][s], dwacc3, 0, 0, 0);
                    }
            }
        }
    }
    // ---------------------------------------------------------------- reductions
    if (MODE == 0 || MODE >= 2) {
        acc1 += __shfl_xor(acc1, 32, 64);
        acc2 += __shfl_xor(acc2, 32, 64);
        __syncthreads();
        if (lh == 0) red[wave][lr] = acc1;
        __syncthreads();
        // MODE 0: forward statistics (activation scale); MODE >= 2: sums of gradients
        constexpr int KS = MODE == 0 ? MM_ACC_STAT : MM_ACC_GRAD;
        const int rep = blockIdx.x % MM_ACC_REPL;
        if (tid < 32) {
            const float s = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
            if (MODE == 3) { if (a.dbias) acc_add<MM_ACC_GRAD>(acc_rep(a.dbias, rep, 32) + tid, s); }
            else acc_add<KS>(acc_rep(a.stats, rep, 64) + tid, s);
        }
        if (MODE != 3) {
            __syncthreads();
            if (lh == 0) red[wave][lr] = acc2;
            __syncthreads();
            if (tid < 32) acc_add<KS>(acc_rep(a.stats, rep, 64) + 32 + tid, red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]);
        }
    }
    if (MODE == 3 || MODE == 4) {
#pragma unroll
        for (int pass = 0; pass < (MODE == 4 ? 2 : 1); ++pass) {
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) {                            // every wave parks its tile: no LDS atomics
                const int tap = (r & 3) + 8 * (r >> 2) + 4 * lh;      // D row
                if (tap < 27) wred[wave][tap][lr] = pass ? dwacc3[r] : dwacc[r];
            }
            __syncthreads();
            mm_acc_t* dwr = acc_rep(pass ? a.dw3 : a.dw, blockIdx.x % MM_ACC_REPL, 27 * 32);
            for (int i = tid; i < 27 * 32; i += 256)
                acc_add<MM_ACC_GRAD>(&dwr[i], ((&wred[0][0][0])[i] + (&wred[1][0][0])[i]) + ((&wred[2][0][0])[i] + (&wred[3][0][0])[i]));
        }
    }
}

// S[tap] = sum over output voxels v of x~[v + tap - 1] (zero padded, bf16-rounded as the conv sees it):
// input voxel u = (d, h, w) feeds tap (kd, kh, kw) iff u - (k - 1) is inside the volume, i.e. the
// indicator factorises per axis.  Per (b, d, h) row: three row sums (all w, all but the last, all but
// the first) and nine conditional adds of them - 27 adds per ROW, not per voxel.
// W % 4 == 0: eight lanes share a row (one float4 each per 32 voxels: a wave reads 8 rows x 128 contiguous bytes
// per instruction; a thread per row read 64 different cache lines per instruction and took 12 us for 4 MB), the
// row sum is a 3-step shuffle and lane 0 of the group keeps the tap sums.  Otherwise: a thread per row.
__global__ __launch_bounds__(256) void l1_tapsum_kernel(const float* __restrict__ x, float* __restrict__ out /* [REPL][32] */,
                                                        int B, int D, int H, int W) {
    const size_t nrows = (size_t)B * D * H;
    float s[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) s[t] = 0.f;
    auto add_row = [&](size_t row, float all, float first, float last) __attribute__((always_inline)) {
        const unsigned rq = (unsigned)row / (unsigned)H;            // 32-bit divisions: rows = B D H < 2^31 (host-checked)
        const int h = (int)((unsigned)row - rq * (unsigned)H), d = (int)(rq % (unsigned)D);
        const float rw[3] = {all - last, all, all - first};         // kw = 0, 1, 2
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) {
            const bool okd = (kd == 0) ? d <= D - 2 : (kd == 2 ? d >= 1 : true);
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const bool okh = (kh == 0) ? h <= H - 2 : (kh == 2 ? h >= 1 : true);
                if (okd && okh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) s[(kd * 3 + kh) * 3 + kw] += rw[kw];
            }
        }
    };
    if ((W & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        const int sub = threadIdx.x & 7, W4 = W >> 2;
        const size_t g0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3, gstep = ((size_t)gridDim.x * blockDim.x) >> 3;
        const size_t nloop = (nrows + gstep - 1) / gstep;           // same trip count for the 8 lanes of a group AND the wave (shuffles)
        for (size_t it = 0; it < nloop; ++it) {
            const size_t row = g0 + it * gstep;
            const bool ok = row < nrows;
            const float4* xr = reinterpret_cast<const float4*>(x + (ok ? row : 0) * W);
            float part = 0.f, first = 0.f, last = 0.f;
            for (int w4 = sub; w4 < W4; w4 += 8) {
                const float4 v = xr[w4];
                const float a = (float)(bf16)v.x, b = (float)(bf16)v.y, c = (float)(bf16)v.z, e = (float)(bf16)v.w;
                part += (a + b) + (c + e);
                if (w4 == 0) first = a;
                if (w4 == W4 - 1) last = e;
            }
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) {
                part += __shfl_xor(part, o, 64);
                first += __shfl_xor(first, o, 64);                  // one owner each, zeros elsewhere
                last += __shfl_xor(last, o, 64);
            }
            if (ok && sub == 0) add_row(row, part, first, last);
        }
    } else {
        for (size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x; row < nrows; row += (size_t)gridDim.x * blockDim.x) {
            const float* xr = x + row * W;
            float all = 0.f;
            for (int w = 0; w < W; ++w) all += (float)(bf16)xr[w];
            add_row(row, all, (float)(bf16)xr[0], (float)(bf16)xr[W - 1]);
        }
    }
    // block sum of the 27 x (32 or 256) partials through LDS in a fixed order (27 wave_sum()s were 162 dependent
    // ds_bpermute round trips per wave: ~9 us, the whole kernel, whatever the grid)
    const bool fast = (W & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    const int holders = fast ? 32 : 256;               // fast path: lane 0 of every 8-lane group holds the sums
    __shared__ float red[256 * 28];
    __shared__ float red2[8][28];
    if (!fast || (threadIdx.x & 7) == 0) {
        float* dst = red + (fast ? threadIdx.x >> 3 : threadIdx.x) * 28;
#pragma unroll
        for (int t = 0; t < 27; ++t) dst[t] = s[t];
    }
    __syncthreads();
    if (threadIdx.x < 27 * 8) {
        const int tap = threadIdx.x % 27, part = threadIdx.x / 27, per = holders / 8;
        float a = 0.f;
        for (int i = 0; i < per; ++i) a += red[(part * per + i) * 28 + tap];
        red2[part][tap] = a;
    }
    __syncthreads();
    if (threadIdx.x < 27) {
        float a = 0.f;
#pragma unroll
        for (int part = 0; part < 8; ++part) a += red2[part][threadIdx.x];
        acc_add<MM_ACC_STAT>(acc_rep(out, blockIdx.x % MM_ACC_REPL, 32) + threadIdx.x, a);
    }
}

// dW[n][tap] += sc (A1 - c0 S - c1 A3);  dbias[n] += train ? 0 : sc S1   (all inputs: fixed-point accumulators x MM_ACC_REPL)
__global__ void l1_combine_kernel(const float* __restrict__ a1, const float* __restrict__ a3, const float* __restrict__ tapsum,
                                  const float* __restrict__ sums, const float* __restrict__ out4, float* __restrict__ dw,
                                  float* __restrict__ dbias, float inv_count, int train) {
    // one output per 16 lanes, one replica per lane: a single load round trip + (integer) shuffle sums
    // (the serial replica loop was five dependent-latency chains: 11 us on 4 workgroups)
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, r = threadIdx.x & 15;
    if (i >= 32 * 27) return;
    const int n = i / 27, tap = i % 27;
    const mm_acc_t *q1 = reinterpret_cast<const mm_acc_t*>(a1), *q3 = reinterpret_cast<const mm_acc_t*>(a3),
                   *qt = reinterpret_cast<const mm_acc_t*>(tapsum), *qs = reinterpret_cast<const mm_acc_t*>(sums);
    mm_acc_t iA1 = q1[(size_t)r * 864 + tap * 32 + n], iA3 = q3[(size_t)r * 864 + tap * 32 + n];
    mm_acc_t iSt = qt[r * 32 + tap], is0 = qs[r * 64 + n], is1 = qs[r * 64 + 32 + n];
    iA1 = acc_sum_lanes16(iA1); iA3 = acc_sum_lanes16(iA3); iSt = acc_sum_lanes16(iSt);
    is0 = acc_sum_lanes16(is0); is1 = acc_sum_lanes16(is1);
    if (r) return;
    const float A1 = acc_val<MM_ACC_GRAD>(iA1), A3 = acc_val<MM_ACC_GRAD>(iA3), St = acc_val<MM_ACC_STAT>(iSt);
    const float s0 = acc_val<MM_ACC_GRAD>(is0), s1 = acc_val<MM_ACC_GRAD>(is1);
    const float sc = out4[n];
    const float c0 = train ? s0 * inv_count : 0.f, c1 = train ? s1 * inv_count : 0.f;
    dw[i] += sc * (A1 - c0 * St - c1 * A3);
    if (tap == 0 && dbias && !train) dbias[n] += sc * s0;      // train: sum dy == 0 identically
}

// dst[c][r] += sum_rep acc[rep][r][c]   (acc: MM_ACC_REPL fixed-point gradient accumulators)
__global__ void transpose_add_kernel(const float* __restrict__ src, float* __restrict__ dst, int R, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < R * C) {
        const int r = i / C, c = i % C;
        dst[(size_t)c * R + r] += acc_val<MM_ACC_GRAD>(acc_sum(src, (size_t)R * C, i));
    }
}

inline uint32_t thresh_l1(float p) { return p > 0.f ? (uint32_t)((double)p * 4294967296.0) : 0u; }

}  // namespace

extern "C" {

int mm_conv3d_l1(int mode, const float* x, const void* wimg, const float* bias, const float* out4,
                 const void* dout, const float* sums, float* stats, void* out, float* dw_tapmajor, float* dbias,
                 int B, int D, int H, int W, int train, float drop_p, uint32_t seed, const uint32_t* seed_epoch,
                 hipStream_t st) {
    MM_REQUIRE(x && wimg && B > 0 && D > 0 && H > 0 && W > 0, "conv3d_l1: null/invalid");
    MM_REQUIRE(D % 2 == 0 && H % 2 == 0 && W % 2 == 0, "conv3d_l1: D,H,W must be even (MaxPool3d(2))");
    MM_REQUIRE(mode >= 0 && mode <= 3, "conv3d_l1: mode");  /* mode 4 has its own entry point */
    MM_REQUIRE(mode == 0 ? stats != nullptr : out4 != nullptr, "conv3d_l1: stats/out4");
    MM_REQUIRE(mode != 1 || out, "conv3d_l1: out");
    MM_REQUIRE(mode < 2 || dout, "conv3d_l1: dout");
    MM_REQUIRE(mode != 2 || stats, "conv3d_l1: sums_out");
    MM_REQUIRE(mode != 3 || (dw_tapmajor && (!train || sums)), "conv3d_l1: dw/sums");
    L1Args a;
    a.x = x; a.wimg = (const bf16*)wimg; a.bias = bias; a.out4 = out4; a.dout = (const bf16*)dout; a.sums = sums;
    a.stats = stats; a.out = (bf16*)out; a.arg = nullptr; a.dw = dw_tapmajor; a.dw3 = nullptr; a.dbias = dbias;
    a.B = B; a.D = D; a.H = H; a.W = W; a.train = train;
    a.thresh = thresh_l1(drop_p); a.seed = seed; a.inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    a.inv_count = 1.f / ((float)B * D * H * W);
    a.epoch = seed_epoch;
    const int ntiles = B * (D / 2) * ceil_div(H, 8) * ceil_div(W, 32);
    const int grid = ntiles < 1024 ? ntiles : 1024;
    const bool full = H % 8 == 0 && W % 32 == 0;
    switch (mode * 2 + (full ? 1 : 0)) {
        case 0: hipLaunchKernelGGL((conv3d_l1_kernel<0, false>), dim3(grid), dim3(256), 0, st, a); break;
        case 1: hipLaunchKernelGGL((conv3d_l1_kernel<0, true>), dim3(grid), dim3(256), 0, st, a); break;
        case 2: hipLaunchKernelGGL((conv3d_l1_kernel<1, false>), dim3(grid), dim3(256), 0, st, a); break;
        case 3: hipLaunchKernelGGL((conv3d_l1_kernel<1, true>), dim3(grid), dim3(256), 0, st, a); break;
        case 4: hipLaunchKernelGGL((conv3d_l1_kernel<2, false>), dim3(grid), dim3(256), 0, st, a); break;
        case 5: hipLaunchKernelGGL((conv3d_l1_kernel<2, true>), dim3(grid), dim3(256), 0, st, a); break;
        case 6: hipLaunchKernelGGL((conv3d_l1_kernel<3, false>), dim3(grid), dim3(256), 0, st, a); break;
        default: hipLaunchKernelGGL((conv3d_l1_kernel<3, true>), dim3(grid), dim3(256), 0, st, a); break;
    }
    return mm_check_launch("conv3d_l1");
}

int mm_conv3d_l1_fwd_winners(const float* x, const void* wimg, const float* bias, const float* out4, void* out, void* arg,
                             int B, int D, int H, int W, int train, float drop_p, uint32_t seed,
                             const uint32_t* seed_epoch, hipStream_t st) {
    MM_REQUIRE(x && wimg && out4 && out && arg && B > 0 && D > 0 && H > 0 && W > 0, "conv3d_l1_fwd_winners: null/invalid");
    MM_REQUIRE(D % 2 == 0 && H % 2 == 0 && W % 2 == 0, "conv3d_l1_fwd_winners: D,H,W must be even (MaxPool3d(2))");
    L1Args a;
    a.x = x; a.wimg = (const bf16*)wimg; a.bias = bias; a.out4 = out4; a.dout = nullptr; a.sums = nullptr;
    a.stats = nullptr; a.out = (bf16*)out; a.arg = (uint8_t*)arg; a.dw = nullptr; a.dw3 = nullptr; a.dbias = nullptr;
    a.B = B; a.D = D; a.H = H; a.W = W; a.train = train;
    a.thresh = thresh_l1(drop_p); a.seed = seed; a.inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    a.inv_count = 1.f / ((float)B * D * H * W);
    a.epoch = seed_epoch;
    const int ntiles = B * (D / 2) * ceil_div(H, 8) * ceil_div(W, 32);
    const int grid = ntiles < 1024 ? ntiles : 1024;
    if (H % 8 == 0 && W % 32 == 0) hipLaunchKernelGGL((conv3d_l1_kernel<1, true>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv3d_l1_kernel<1, false>), dim3(grid), dim3(256), 0, st, a);
    return mm_check_launch("conv3d_l1_fwd_winners");
}

static int l1_tapsum_grid(int B, int D, int H, int W) {
    const long rows = (long)B * D * H, threads = (W & 3) == 0 ? rows * 8 : rows;
    const long g = (threads + 255) / 256;
    return (int)(g < 1024 ? g : 1024);
}

int mm_conv3d_l1_tapsum(const float* x, float* tapsum, int B, int D, int H, int W, hipStream_t st) {
    MM_REQUIRE(x && tapsum && B > 0 && D > 0 && H > 0 && W > 0, "conv3d_l1_tapsum: null/invalid");
    hipLaunchKernelGGL(l1_tapsum_kernel, dim3(l1_tapsum_grid(B, D, H, W)), dim3(256), 0, st,
                       x, tapsum, B, D, H, W);
    return mm_check_launch("conv3d_l1_tapsum");
}

int mm_conv3d_l1_bwd(const float* x, const void* wimg, const float* bias, const float* out4, const void* dout,
                     float* sums_out, float* a1, float* a3, float* tapsum, int tapsum_ready, float* dw, float* dbias, int B,
                     int D, int H, int W, int train, float drop_p, uint32_t seed, const uint32_t* seed_epoch,
                     hipStream_t st) {
    MM_REQUIRE(x && wimg && out4 && dout && sums_out && a1 && a3 && tapsum && dw && B > 0, "conv3d_l1_bwd: null/invalid");
    MM_REQUIRE(D % 2 == 0 && H % 2 == 0 && W % 2 == 0, "conv3d_l1_bwd: D,H,W must be even (MaxPool3d(2))");
    L1Args a;
    a.x = x; a.wimg = (const bf16*)wimg; a.bias = bias; a.out4 = out4; a.dout = (const bf16*)dout; a.sums = nullptr;
    a.stats = sums_out; a.out = nullptr; a.arg = nullptr; a.dw = a1; a.dw3 = a3; a.dbias = nullptr;
    a.B = B; a.D = D; a.H = H; a.W = W; a.train = train;
    a.thresh = thresh_l1(drop_p); a.seed = seed; a.inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    a.inv_count = 1.f / ((float)B * D * H * W);
    a.epoch = seed_epoch;
    const int ntiles = B * (D / 2) * ceil_div(H, 8) * ceil_div(W, 32);
    if (!tapsum_ready)
        hipLaunchKernelGGL(l1_tapsum_kernel, dim3(l1_tapsum_grid(B, D, H, W)), dim3(256), 0,
                           st, x, tapsum, B, D, H, W);
    if (H % 8 == 0 && W % 32 == 0) hipLaunchKernelGGL((conv3d_l1_kernel<4, true>), dim3(ntiles < 1024 ? ntiles : 1024), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv3d_l1_kernel<4, false>), dim3(ntiles < 1024 ? ntiles : 1024), dim3(256), 0, st, a);
    hipLaunchKernelGGL(l1_combine_kernel, dim3(ceil_div(32 * 27 * 16, 256)), dim3(256), 0, st, a1, a3, tapsum, sums_out, out4, dw,
                       dbias, a.inv_count, train);
    return mm_check_launch("conv3d_l1_bwd");
}

int mm_transpose_add(const float* src, float* dst, int R, int C, int nrep, hipStream_t st) {
    MM_REQUIRE(src && dst && R > 0 && C > 0, "transpose_add: null");
    MM_REQUIRE(nrep == MM_ACC_REPL, "transpose_add: src is an accumulator workspace of %d replicas", MM_ACC_REPL);
    hipLaunchKernelGGL(transpose_add_kernel, dim3(ceil_div(R * C, 256)), dim3(256), 0, st, src, dst, R, C);
    return mm_check_launch("transpose_add");
}

}  // extern "C"
