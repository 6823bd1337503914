// First layer of the 3-D voxel encoder: Conv3d(1 -> 32, k3, p1) + BatchNorm3d +
// GELU + MaxPool3d(2) [+ Dropout] on single-channel fp32 volumes, fused.
//
// With Cin = 1 the GEMM K is just the 27 taps: 8 bf16 MFMAs per 128 voxels, i.e. the layer is bound by its
// VALU epilogue and by load latency, not by MFMA or HBM (profiles/r04_pmc_l1.summary.txt: VALU 50-65 % of the
// kernel time, waves waiting 37-55 %, 10-33 MB of HBM traffic).  The pre-BN activation (32x the input, 134 MB
// fp32 at the C2 config) is never written: every pass recomputes the convolution from the 4 MB input.
//
// Training forward:
//   gram   G = Xcol^T Xcol over all output voxels (Xcol = the im2col matrix, 27 taps + a column of ones): ONE
//          32x32 MFMA accumulator per wave for the whole kernel, no epilogue.  Everything the layer needs to know
//          about the input beyond the convolution itself is in G:
//            sum_v y_n   = w_n . S + M b_n                       (S[t] = G[t][ones], M = G[ones][ones])
//            sum_v y_n^2 = w_n^T G w_n + 2 b_n w_n . S + M b_n^2     -> BatchNorm statistics (formed per workgroup: linear in G)
//            A3[t][n] = sum_v xcol[v][t] xhat[v][n] = rstd_n ((G w_n)[t] + (b_n - mean_n) S[t])   (backward)
//          It replaces the statistics pass (mode 0: a full recompute with a per-channel epilogue) AND the tap-sum
//          kernel, and takes the second MFMA product out of the backward.
//   mode 1 apply        : BN -> GELU -> 2x2x2 max -> dropout -> bf16
// Training backward (one recompute pass, mode 4): BatchNorm's backward is linear in S1 = sum dz, S2 = sum dz xhat, so
//   dW = sc (A1 - c0 S - c1 A3),  A1 = x^T dz,  c0 = S1 / M, c1 = S2 / M: the pass yields S1, S2 and A1 (one MFMA
//   product with the sparse dz fragments), l1_combine_kernel finishes from the Gram workspace.
// Kept for callers of the C ABI / frozen-weight paths: mode 0 (statistics by recompute), mode 2 (S1, S2 only),
// mode 3 (two-pass weight gradient), l1_tapsum_kernel.
//
// One wave owns a 2 x 8 x 8 block of conv outputs (= 1 x 4 x 4 pooled voxels) x 32 channels: all 8 members of
// every pooling window sit in the same lane.  A workgroup (4 waves) walks tiles of 2 x 8 x 32 voxels persistently
// through a double-buffered LDS halo (one barrier per tile); the per-thread halo addressing (6 elements) is computed
// once per kernel, not per tile (it was ~250 of mode 1's ~860 VALU instructions per tile).  Fetching the halo of tile
// i + 1 into registers during tile i was measured and is off: two other workgroups per CU already cover the load.
// GELU is evaluated once per window when max(z) >= 0 (GELU is monotone on [-0.7518, inf) and negative left of 0,
// so the window max is GELU(max z)); otherwise at the largest and the smallest member.  Both branches are exact.
#include "common.h"

// ablation builds (tools/abl_stream.sh with ABL_FILE=conv3d_l1 ABL_MACRO=L1_ABL; product = 0; profiles/r04_l1_ablation.txt):
// 1 wave-uniform branch (ballot) in front of the all-negative-window path, 2 fmaxf / fminf trees instead of v_max3 / v_min3,
// 4 dz fragments built member by member, 8 halo of tile i + 1 prefetched into registers during tile i.
// Measured on one box, alternating: 1 and 8 make the backward 0.7 / 1.5 us SLOWER (and are off), 2 and 4 are what the
// product does NOT do (they cost 0.3 / 1.3 us).
#ifndef L1_BWD_WAVES
#define L1_BWD_WAVES 2          // workgroups per CU (= waves per SIMD) of the backward kernels (A/B builds: 3)
#endif
#ifndef L1_ABL
#define L1_ABL 0
#endif

namespace {

constexpr int HP = 36;                 // halo row pitch (34 used)
constexpr int HROWS1 = 4 * 10;         // (2+2) depth x (8+2) height rows
constexpr int HSZ = HROWS1 * HP;
constexpr int NH = (HROWS1 * 34 + 255) / 256;      // halo elements per thread (6)
constexpr int GN = 28;                 // Gram matrix order: 27 taps + the column of ones

struct L1Args {
    const float* x;        // [B][D][H][W]
    const bf16* wimg;      // [32][32] (n, tap; taps 27..31 zero)
    const float* bias;     // [32] or nullptr (eval: folded into out4 shift)
    const float* out4;     // [4][32] scale, shift, mean, rstd  (modes 1-4)
    const bf16* dout;      // [B][D/2][H/2][W/2][32]            (modes 2-4)
    const float* sums;     // [2][32]                           (mode 3)
    float* stats;          // mode 0: [2][32];  modes 2, 4: sums_out
    bf16* out;             // mode 1
    uint8_t* arg;          // mode 1 with ARG: the pooling window's winner j = (dd << 2) | (hh << 1) | ww per output element
    float* dw;             // mode 3: [27][32] (tap-major, channel-contiguous); mode 4: A1
    float* dbias;          // mode 3
    float* gram;           // gram kernel: accumulator workspace [32][32] (tap, tap)
    MmBnFin fin;           // mode 1 with FIN: the layer's BatchNorm finalize in the kernel's prologue (out4 written, not read)
    int B, D, H, W, train;
    uint32_t thresh, seed; float inv_keep, inv_count;
    const uint32_t* epoch;
};

__device__ __forceinline__ int tap_off(int tap) {          // halo offset of tap (0..26), pads -> 0
    if (tap >= 27) return 0;
    const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
    return kd * 10 * HP + kh * HP + kw;
}

// v_max3_f32 / v_min3_f32: the window extreme of 8 finite values in 4 instructions.  (fmaxf() in IEEE mode first
// canonicalises every operand - one v_max_f32 v, v, v per member: 72 of mode 1's ~700 VALU instructions per tile.)
__device__ __forceinline__ float max3f(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float min3f(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float max8f(const float (&z)[8]) {
    if (L1_ABL & 2) return fmaxf(fmaxf(fmaxf(z[0], z[1]), fmaxf(z[2], z[3])), fmaxf(fmaxf(z[4], z[5]), fmaxf(z[6], z[7])));
    return max3f(max3f(z[0], z[1], z[2]), max3f(z[3], z[4], z[5]), max3f(z[6], z[7], z[7]));
}
__device__ __forceinline__ float min8f(const float (&z)[8]) {
    if (L1_ABL & 2) return fminf(fminf(fminf(z[0], z[1]), fminf(z[2], z[3])), fminf(fminf(z[4], z[5]), fminf(z[6], z[7])));
    return min3f(min3f(z[0], z[1], z[2]), min3f(z[3], z[4], z[5]), min3f(z[6], z[7], z[7]));
}

// ---- halo staging: the (2+2) x (8+2) x (32+2) input block of a tile, bf16, zero outside the volume -----------------
// per-thread plan, computed once: element i = tid + 256 q -> offset relative to the tile's first voxel and a packed
// descriptor (bits 0-11 LDS index, 12-13 depth row, 14-17 height row, 18-23 column; sign bit: no such element)
struct HaloPlan { int roff[NH]; int pk[NH]; };

__device__ __forceinline__ void halo_plan(HaloPlan& p, int tid, int H, int W) {
#pragma unroll
    for (int q = 0; q < NH; ++q) {
        const int i = tid + q * 256;
        const int hw = i % 34, hr = i / 34, hd = hr / 10, hh = hr % 10;
        p.roff[q] = ((hd - 1) * H + (hh - 1)) * W + (hw - 1);
        p.pk[q] = i < HROWS1 * 34 ? ((hr * HP + hw) | (hd << 12) | (hh << 14) | (hw << 18)) : (int)0x80000000;
    }
}
// all loads of a tile's halo in flight at once, branch-free (an out-of-volume element reads the tile's first voxel and
// is replaced by zero): xb = &x[b][d0][h0][w0]
__device__ __forceinline__ void halo_load(float (&hv)[NH], const float* __restrict__ xb, const HaloPlan& p,
                                          int d0, int h0, int w0, int D, int H, int W) {
#pragma unroll
    for (int q = 0; q < NH; ++q) {
        const int pk = p.pk[q];
        const unsigned d = (unsigned)(d0 - 1 + ((pk >> 12) & 3)), h = (unsigned)(h0 - 1 + ((pk >> 14) & 15)),
                       w = (unsigned)(w0 - 1 + ((pk >> 18) & 63));
        const bool ok = pk >= 0 && d < (unsigned)D && h < (unsigned)H && w < (unsigned)W;
        const float v = xb[ok ? p.roff[q] : 0];
        hv[q] = ok ? v : 0.f;
    }
}
__device__ __forceinline__ void halo_store(unsigned short* __restrict__ hb, const HaloPlan& p, const float (&hv)[NH]) {
#pragma unroll
    for (int q = 0; q < NH; ++q) {
        const bf16 b = (bf16)hv[q];
        if (p.pk[q] >= 0) hb[p.pk[q] & 0xFFF] = *reinterpret_cast<const unsigned short*>(&b);
    }
}
struct TileCoord { int b, d0, h0, w0; };
__device__ __forceinline__ TileCoord tile_coord(int tile, int tw, int th, int td) {
    TileCoord c;
    int q = tile;
    c.w0 = (q % tw) * 32; q /= tw;
    c.h0 = (q % th) * 8; q /= th;
    c.d0 = (q % td) * 2; q /= td;
    c.b = q;
    return c;
}
__device__ __forceinline__ const float* tile_ptr(const float* x, const TileCoord& c, int D, int H, int W) {
    return x + ((((size_t)c.b * D + c.d0) * H + c.h0) * W + c.w0);
}

// FULLT: H % 8 == 0 and W % 32 == 0, i.e. every 2 x 8 x 32 tile lies inside the volume: the per-voxel bounds tests
// (three compares and the index arithmetic behind them, per voxel and channel) are compiled out.
// ARG (mode 1): also write the window winners (inspection output of the parity tests, mm_conv3d_l1_fwd_winners).
// FIN (mode 1): the train-mode BatchNorm finalize as the prologue (csrc/common.h: bn_fin_channel).
template <int MODE, bool FULLT = false, bool ARG = false, bool FIN = false>
__global__ __launch_bounds__(256, MODE <= 1 ? 3 : L1_BWD_WAVES) void conv3d_l1_kernel(L1Args a) {   // statistics / forward: three waves per SIMD (<= 168 registers), backward two
    a.seed = mm_eff_seed(a.seed, a.epoch);
    __shared__ __attribute__((aligned(16))) unsigned short halo[2][HSZ];
    __shared__ float red[4][32];
    __shared__ float wred[(MODE == 3 || MODE == 4) ? 4 : 1][27][32];
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
    if (FIN) {
        __shared__ float s_fin[2][32];
        if (tid < 32) {
            float m_, r_;
            bn_fin_channel(a.fin, tid, blockIdx.x == 0, s_fin[0][tid], s_fin[1][tid], m_, r_);
        }
        __syncthreads();
        sc = s_fin[0][lr]; sh = s_fin[1][lr];
    } else if (MODE >= 1) {
        sc = a.out4[lr]; sh = a.out4[32 + lr]; mu = a.out4[64 + lr]; rs = a.out4[96 + lr];
        if (MODE == 3 && a.train) { c0 = a.sums[lr] * a.inv_count; c1 = a.sums[32 + lr] * a.inv_count; }   // compact sums
    }
    // z = (acc + bias) sc + sh = acc sc + shb: ONE FMA per voxel and channel in every pass (forward and backward must
    // form z identically - the window winner is decided on it)
    const float shb = fmaf(bias, sc, sh);
    float acc1 = 0.f, acc2 = 0.f;          // per-lane channel sums (modes 0, 2, 4) / dbias (mode 3)
    f32x16 dwacc;                          // modes 3, 4: D[tap][n]
#pragma unroll
    for (int r = 0; r < 16; ++r) dwacc[r] = 0.f;
    const int my_tap_off = tap_off(lr);    // modes 3, 4: A row = tap lr

    HaloPlan plan;
    halo_plan(plan, tid, a.H, a.W);
    float hv[NH];
    int tile = blockIdx.x, buf = 0;
    if (tile < ntiles) {
        const TileCoord c = tile_coord(tile, tw, th, td);
        halo_load(hv, tile_ptr(a.x, c, a.D, a.H, a.W), plan, c.d0, c.h0, c.w0, a.D, a.H, a.W);
    }
    for (; tile < ntiles; tile += gridDim.x, buf ^= 1) {
        const TileCoord tc = tile_coord(tile, tw, th, td);
        const int b = tc.b, d0 = tc.d0, h0 = tc.h0, w0 = tc.w0;
        unsigned short* hb = halo[buf];
        if (!(L1_ABL & 8) && tile != (int)blockIdx.x) halo_load(hv, tile_ptr(a.x, tc, a.D, a.H, a.W), plan, d0, h0, w0, a.D, a.H, a.W);
        halo_store(hb, plan, hv);
        __syncthreads();                                    // (the other buffer's readers passed the previous barrier)
        if ((L1_ABL & 8) && tile + (int)gridDim.x < ntiles) {    // ablation: next tile's halo in flight during this tile's compute
            const TileCoord c = tile_coord(tile + gridDim.x, tw, th, td);
            halo_load(hv, tile_ptr(a.x, c, a.D, a.H, a.W), plan, c.d0, c.h0, c.w0, a.D, a.H, a.W);
        }
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
                for (int j = 0; j < 8; ++j) fr.u[j] = hb[vb + foff[s][j]];
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
        // dy (mode 3) / dz (mode 4) fragments of the two conv tiles ip, ip + 2 that one pass of the ip loop completes;
        // their weight-gradient MFMAs run at the end of that pass, so only two tiles' fragments are ever live
        union DyFrag { bf16x8 v; uint32_t u[4]; } dyf[2][2];
        // flat index of this lane's first pooled output of the tile (32-bit: the host checks the tensor size)
        const uint32_t obase = ((((uint32_t)b * Do + (d0 >> 1)) * Ho + (h0 >> 1)) * Wo + (w0 >> 1) + 4 * wave + 2 * lh) * 32 + lr;
#pragma unroll
        for (int ip = 0; ip < 2; ++ip) {
#pragma unroll
            for (int ra = 0; ra < 2; ++ra)
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    const int r0 = 8 * ra + 2 * rb;
                    const int oh = (h0 >> 1) + 2 * ip + ra, ow = (w0 >> 1) + 4 * wave + rb + 2 * lh;
                    const bool ok = FULLT || (oh < Ho && ow < Wo);
                    float z[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {           // j = (dd << 2) | (hh << 1) | ww
                        const int ti = ip + 2 * (j >> 2), r = r0 + 4 * ((j >> 1) & 1) + (j & 1);
                        z[j] = fmaf(acc[ti][r], sc, shb);
                    }
                    // GELU falls on (-inf, -0.75] and rises after it, so the window's largest activation sits at its largest
                    // or at its smallest pre-activation (as pool3_bn_act).  With zmax >= 0 it is the largest: GELU(zmax) >= 0
                    // and anything below it is smaller (rising branch) or negative.  Only an all-negative window (1 in 256
                    // for unit-normal pre-activations) needs the two evaluations - the backward modes then evaluate none
                    // to find the winner, the forward one.
                    // The winner is the FIRST member that equals the extreme value (as PyTorch's max-pool).
                    // (A wave-uniform branch in front of the all-negative case - no lane of the wave in ~78 % of the windows -
                    // measured 0.7 us slower than letting the compiler predicate it: L1_ABL bit 1.)
                    float zsel = max8f(z);
                    float best = MODE == 1 ? gelu_erf(zsel) : 0.f;
                    if (!(L1_ABL & 1) || __builtin_amdgcn_ballot_w64(zsel < 0.f) != 0) {
                        if (zsel < 0.f) {
                            const float zmin = min8f(z);
                            if (MODE != 1) best = gelu_erf(zsel);
                            const float amin = gelu_erf(zmin);
                            if (amin > best) { best = amin; zsel = zmin; }
                        }
                    }
                    const uint32_t oidx = obase + ((uint32_t)(2 * ip + ra) * Wo + rb) * 32;
                    if (MODE == 1) {
                        if (ok) {
                            if (a.thresh) best *= dropout_scale(a.seed, oidx, a.thresh, a.inv_keep);
                            a.out[oidx] = (bf16)best;
                            if (ARG) {                          // which member won (first hit)
                                int js = 7;
#pragma unroll
                                for (int j = 6; j >= 0; --j) js = z[j] == zsel ? j : js;
                                a.arg[oidx] = (uint8_t)js;
                            }
                        }
                        continue;
                    }
                    // backward modes: the winner's index (first hit, as the forward) and its pre-BatchNorm value
                    int js = 7;
#pragma unroll
                    for (int j = 6; j >= 0; --j) js = z[j] == zsel ? j : js;
                    float g = ok ? (float)a.dout[oidx] : 0.f;
                    if (a.thresh) g *= dropout_scale(a.seed, oidx, a.thresh, a.inv_keep);
                    const float dzs = g * gelu_erf_grad(zsel);
                    if (MODE == 2 || MODE == 4) {
                        // xhat of the winner: its accumulator value (+ bias) is at hand; going back from z would need a division.
                        // Two-level select on (js >> 1, js & 1): the four pair masks are shared with the dz fragments below.
                        const int jp = js >> 1;
                        const int t1 = ip + 2;                                   // members j = 4..7 live in conv tile ip + 2
                        const float lo = jp == 0 ? acc[ip][r0] : jp == 1 ? acc[ip][r0 + 4] : jp == 2 ? acc[t1][r0] : acc[t1][r0 + 4];
                        const float hi = jp == 0 ? acc[ip][r0 + 1] : jp == 1 ? acc[ip][r0 + 5] : jp == 2 ? acc[t1][r0 + 1] : acc[t1][r0 + 5];
                        const float aw = (js & 1) ? hi : lo;
                        acc1 += dzs;
                        acc2 += dzs * (aw + bias - mu) * rs;
                    }
                    if (MODE == 4) {
                        // the window's 8 members are 4 packed bf16 pairs of the dz fragments - (dd, hh) -> register rb + 2 hh
                        // of fragment [dd][ra], low / high half = ww - and exactly one member is non-zero (a window that
                        // is not in the volume has g = 0): one converted value, shifted to its half, selected into its pair
                        if (L1_ABL & 4) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                const int ti = ip + 2 * (j >> 2), r = r0 + 4 * ((j >> 1) & 1) + (j & 1);
                                dyf[ti >> 1][r >> 3].v[r & 7] = (bf16)(js == j ? dzs : 0.f);
                            }
                        } else {
                            const bf16 db = (bf16)dzs;
                            const uint32_t pairval = (uint32_t)(*reinterpret_cast<const unsigned short*>(&db)) << ((js & 1) << 4);
                            const int jp = js >> 1;
#pragma unroll
                            for (int k = 0; k < 4; ++k) dyf[k >> 1][ra].u[rb + 2 * (k & 1)] = jp == k ? pairval : 0u;
                        }
                    } else if (MODE == 3) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int ti = ip + 2 * (j >> 2), r = r0 + 4 * ((j >> 1) & 1) + (j & 1);
                            const float dzj = js == j ? dzs : 0.f;
                            const float y = acc[ti][r] + bias;
                            float dy = a.train ? sc * (dzj - c0 - (y - mu) * rs * c1) : sc * dzj;
                            const int d = d0 + (ti >> 1), h = h0 + 4 * (ti & 1) + (r >> 2), w = w0 + wbase + (r & 3) + 4 * lh;
                            if (!FULLT && !(d < a.D && h < a.H && w < a.W)) dy = 0.f;
                            acc1 += dy;
                            dyf[ti >> 1][r >> 3].v[r & 7] = (bf16)dy;
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
                            fr.u[j] = hb[vb + my_tap_off];
                        }
                        dwacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr.v, dyf[tq][s].v, dwacc, 0, 0, 0);
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
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) {                            // every wave parks its tile: no LDS atomics
            const int tap = (r & 3) + 8 * (r >> 2) + 4 * lh;      // D row
            if (tap < 27) wred[wave][tap][lr] = dwacc[r];
        }
        __syncthreads();
        mm_acc_t* dwr = acc_rep(a.dw, blockIdx.x % MM_ACC_REPL, 27 * 32);
        for (int i = tid; i < 27 * 32; i += 256)
            acc_add<MM_ACC_GRAD>(&dwr[i], ((&wred[0][0][0])[i] + (&wred[1][0][0])[i]) + ((&wred[2][0][0])[i] + (&wred[3][0][0])[i]));
    }
}

// G[t][t'] = sum over output voxels v of xcol[v][t] xcol[v][t'] for the 27 taps and the constant column 27 (ones):
// the Gram matrix of the im2col matrix of the zero-padded, bf16-rounded volume.  A wave owns 2 x 8 x 8 voxels per tile
// = 8 MFMA K-steps of 16 voxels; the A operand (row = tap lr, k = 8 consecutive voxels of one row) and the B operand
// (k = voxel, column = tap lr) of G += Xcol^T Xcol are the SAME registers.  One accumulator for the whole kernel.
// Every workgroup also turns ITS partial Gram matrix into partial BatchNorm sums (they are linear in G): 64 fixed-point
// adds into the ordinary statistics workspace, so that mm_bn_finalize follows as after any other convolution and no
// kernel has to wait for the whole of G.
template <bool FULLT>
__global__ __launch_bounds__(256, 3) void l1_gram_kernel(L1Args a) {
    __shared__ __attribute__((aligned(16))) unsigned short halo[2][HSZ];
    __shared__ __attribute__((aligned(16))) float gred[4][GN][32];
    __shared__ float wsh[32][28];
    __shared__ double part[8][2][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int tw = (a.W + 31) / 32, th = (a.H + 7) / 8, td = a.D / 2;
    const int ntiles = a.B * td * th * tw;
    const int my_tap_off = tap_off(lr);
    for (int i = tid; i < 32 * 27; i += 256)                    // the bf16 weights the convolution multiplies with
        wsh[i / 27][i % 27] = (float)a.wimg[(i / 27) * 32 + i % 27];
    f32x16 g;
#pragma unroll
    for (int r = 0; r < 16; ++r) g[r] = 0.f;
    HaloPlan plan;
    halo_plan(plan, tid, a.H, a.W);
    float hv[NH];
    int tile = blockIdx.x, buf = 0;
    if (tile < ntiles) {
        const TileCoord c = tile_coord(tile, tw, th, td);
        halo_load(hv, tile_ptr(a.x, c, a.D, a.H, a.W), plan, c.d0, c.h0, c.w0, a.D, a.H, a.W);
    }
    const uint32_t keep = lr < 27 ? 0xFFFFFFFFu : 0u, orv = lr == 27 ? 0x3F803F80u : 0u;   // bf16(1.0) pairs for the ones row
    for (; tile < ntiles; tile += gridDim.x, buf ^= 1) {
        const TileCoord tc = tile_coord(tile, tw, th, td);
        unsigned short* hb = halo[buf];
        if (!(L1_ABL & 8) && tile != (int)blockIdx.x) halo_load(hv, tile_ptr(a.x, tc, a.D, a.H, a.W), plan, tc.d0, tc.h0, tc.w0, a.D, a.H, a.W);
        halo_store(hb, plan, hv);
        __syncthreads();
        if ((L1_ABL & 8) && tile + (int)gridDim.x < ntiles) {
            const TileCoord c = tile_coord(tile + gridDim.x, tw, th, td);
            halo_load(hv, tile_ptr(a.x, c, a.D, a.H, a.W), plan, c.d0, c.h0, c.w0, a.D, a.H, a.W);
        }
        const int wbase = 8 * wave;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            // K-step s: voxels m = 16 s + 8 lh + j, j = 0..7: depth s >> 2, row (2 s + lh) & 7, columns wbase + j
            const int hrow = (2 * s + lh) & 7;
            const int vb = ((s >> 2) * 10 + hrow) * HP + wbase;
            union { uint32_t u[4]; bf16x8 v; } fr;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                uint32_t lo = hb[vb + 2 * jj + my_tap_off], hi = hb[vb + 2 * jj + 1 + my_tap_off];   // (rows 27..31 read the voxel itself)
                if (!FULLT) {                                   // an output voxel outside the volume contributes nothing
                    const bool inh = tc.h0 + hrow < a.H;
                    lo = (inh && tc.w0 + wbase + 2 * jj < a.W) ? lo : 0u;
                    hi = (inh && tc.w0 + wbase + 2 * jj + 1 < a.W) ? hi : 0u;
                    fr.u[jj] = ((lo | (hi << 16)) & keep) | ((inh && tc.w0 + wbase + 2 * jj < a.W) ? (orv & 0xFFFFu) : 0u)
                               | ((inh && tc.w0 + wbase + 2 * jj + 1 < a.W) ? (orv & 0xFFFF0000u) : 0u);
                } else {
                    fr.u[jj] = ((lo | (hi << 16)) & keep) | orv;    // rows 27..31: the column of ones / zero rows (v_and_or_b32)
                }
            }
            g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr.v, fr.v, g, 0, 0, 0);
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {                                // D[m][n]: column n = lr, row m = (r & 3) + 8 (r >> 2) + 4 lh
        const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < GN) gred[wave][m][lr] = g[r];
    }
    __syncthreads();
    mm_acc_t* gw = acc_rep(a.gram, blockIdx.x % MM_ACC_REPL, 32 * 32);
    for (int i = tid; i < GN * 32; i += 256) {                    // G is symmetric: the upper triangle only (half the atomics)
        const int m = i >> 5, n = i & 31;
        const float v = (gred[0][m][n] + gred[1][m][n]) + (gred[2][m][n] + gred[3][m][n]);
        gred[0][m][n] = v;                                        // this workgroup's partial G, whole (same thread reads and writes)
        if (n < GN && n >= m) acc_add<MM_ACC_STAT>(&gw[m * 32 + n], v);
    }
    __syncthreads();
    // partial sum_v y_n = w_n . S + M b_n and sum_v y_n^2 = w_n^T G w_n + 2 b_n w_n . S + M b_n^2 of this workgroup's voxels:
    // thread (channel n, row group p) takes rows t = p, p + 8, ...; its channel's 27 weights sit in registers and a row of
    // G is read as seven float4 (every lane of a row group reads the same addresses: LDS broadcasts).  fp32 products,
    // summed per row in fp32 and across rows / groups / workgroups in double / fixed point.
    {
        const int n = tid & 31, p = tid >> 5;
        float w[28];
#pragma unroll
        for (int u = 0; u < 27; ++u) w[u] = wsh[n][u];
        w[27] = 0.f;
        double q = 0.0, sd = 0.0;
        for (int t = p; t < 27; t += 8) {
            const float4* grow = reinterpret_cast<const float4*>(&gred[0][t][0]);
            float row = 0.f;
#pragma unroll
            for (int u4 = 0; u4 < 7; ++u4) {
                const float4 gv = grow[u4];
                row = fmaf(gv.x, w[4 * u4], row); row = fmaf(gv.y, w[4 * u4 + 1], row);
                row = fmaf(gv.z, w[4 * u4 + 2], row); row = fmaf(gv.w, w[4 * u4 + 3], row);      // (u = 27: w = 0)
            }
            const float wt = wsh[n][t];
            q += (double)(wt * row);
            sd += (double)(wt * gred[0][t][27]);
        }
        part[p][0][n] = q; part[p][1][n] = sd;
    }
    __syncthreads();
    if (tid < 32) {
        const int n = tid;
        double q = 0.0, sd = 0.0;
#pragma unroll
        for (int p = 0; p < 8; ++p) { q += part[p][0][n]; sd += part[p][1][n]; }
        const double b = a.bias ? (double)a.bias[n] : 0.0, M = (double)gred[0][27][27];
        mm_acc_t* st = acc_rep(a.stats, blockIdx.x % MM_ACC_REPL, 64);
        acc_add<MM_ACC_STAT>(st + n, (float)(sd + M * b));
        acc_add<MM_ACC_STAT>(st + 32 + n, (float)(q + 2.0 * b * sd + M * b * b));
    }
}

// S[tap] = sum over output voxels v of x~[v + tap - 1] (zero padded, bf16-rounded as the conv sees it):
// input voxel u = (d, h, w) feeds tap (kd, kh, kw) iff u - (k - 1) is inside the volume, i.e. the
// indicator factorises per axis.  Per (b, d, h) row: three row sums (all w, all but the last, all but
// the first) and nine conditional adds of them - 27 adds per ROW, not per voxel.  (The training path takes S from the
// Gram matrix; this kernel serves callers of mm_conv3d_l1_tapsum.)
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

// dW[n][tap] += sc (A1 - c0 S - c1 A3);  dbias[n] += train ? 0 : sc S1, with S[tap] = G[tap][27] and
// A3[tap][n] = rstd_n ((G w_n)[tap] + (b_n - mean_n) S[tap]) from the Gram accumulator workspace of the forward pass
// (upper triangle, MM_ACC_REPL replicas); a1 / sums: fixed-point accumulators x MM_ACC_REPL.  One output per 16 lanes,
// one REPLICA per lane: lane r forms its replica's share of (G w_n)[tap] and S[tap] in double (27 independent loads), the
// 16 shares are summed by a fixed butterfly.  gram may be null when train == 0 (c0 = c1 = 0: frozen BatchNorm).
__global__ void l1_combine_kernel(const float* __restrict__ a1, const float* __restrict__ gram, const bf16* __restrict__ wimg,
                                  const float* __restrict__ bias, const float* __restrict__ sums, const float* __restrict__ out4,
                                  float* __restrict__ dw, float* __restrict__ dbias, float inv_count, int train) {
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, r = threadIdx.x & 15;
    if (i >= 32 * 27) return;
    const int n = i / 27, tap = i % 27;
    const mm_acc_t *q1 = reinterpret_cast<const mm_acc_t*>(a1), *qs = reinterpret_cast<const mm_acc_t*>(sums);
    mm_acc_t iA1 = q1[(size_t)r * 864 + tap * 32 + n], is0 = qs[r * 64 + n], is1 = qs[r * 64 + 32 + n];
    double gw = 0.0, St = 0.0;
    unsigned bad = 0;
    if (train) {
        const mm_acc_t* gr = reinterpret_cast<const mm_acc_t*>(gram) + (size_t)r * 1024;
        mm_acc_t v[28];
#pragma unroll
        for (int u = 0; u < 28; ++u) v[u] = gr[(u < tap ? u : tap) * 32 + (u < tap ? tap : u)];      // symmetric: upper triangle stored
#pragma unroll
        for (int u = 0; u < 28; ++u) bad |= (unsigned long long)(v[u] + (1ll << 61)) >= (1ull << 62) ? 1u : 0u;   // poisoned replica
#pragma unroll
        for (int u = 0; u < 27; ++u) gw += (double)v[u] * (double)(float)wimg[n * 32 + u];
        St = (double)v[27];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            gw += __shfl_xor(gw, o, 64); St += __shfl_xor(St, o, 64); bad |= __shfl_xor(bad, o, 64);
        }
        const double k = 1.0 / (double)(1ull << MM_ACC_STAT);
        gw *= k; St *= k;
    }
    iA1 = acc_sum_lanes16(iA1); is0 = acc_sum_lanes16(is0); is1 = acc_sum_lanes16(is1);
    if (r) return;
    const float A1 = acc_val<MM_ACC_GRAD>(iA1);
    const float s0 = acc_val<MM_ACC_GRAD>(is0), s1 = acc_val<MM_ACC_GRAD>(is1);
    const float sc = out4[n], mu = out4[64 + n], rs = out4[96 + n];
    float corr = 0.f;
    if (train) {
        const float b = bias ? bias[n] : 0.f;
        const float Stf = bad ? __builtin_nanf("") : (float)St;
        const float A3 = rs * ((float)gw + (b - mu) * Stf);
        corr = s0 * inv_count * Stf + s1 * inv_count * A3;
    }
    dw[i] += sc * (A1 - corr);
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

inline void l1_fill(L1Args& a, const float* x, const void* wimg, const float* bias, const float* out4, int B, int D, int H,
                    int W, int train, float drop_p, uint32_t seed, const uint32_t* seed_epoch) {
    a.x = x; a.wimg = (const bf16*)wimg; a.bias = bias; a.out4 = out4; a.dout = nullptr; a.sums = nullptr;
    a.stats = nullptr; a.out = nullptr; a.arg = nullptr; a.dw = nullptr; a.dbias = nullptr; a.gram = nullptr; a.fin = MmBnFin{};
    a.B = B; a.D = D; a.H = H; a.W = W; a.train = train;
    a.thresh = thresh_l1(drop_p); a.seed = seed; a.inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    a.inv_count = 1.f / ((float)B * D * H * W);
    a.epoch = seed_epoch;
}
inline int l1_tiles(int B, int D, int H, int W) { return B * (D / 2) * ceil_div(H, 8) * ceil_div(W, 32); }
// persistent grids: `per_cu` resident workgroups on each of the 256 CUs (the register budget of the instance decides)
inline int l1_grid(int ntiles, int per_cu) { const int cap = 256 * per_cu; return ntiles < cap ? ntiles : cap; }
// pooled output indices are formed in 32 bits (they also seed the dropout hash)
inline bool l1_fits(int B, int D, int H, int W) { return (double)B * D * H * W * 4.0 < 2147483648.0; }

}  // namespace

extern "C" {

int mm_conv3d_l1(int mode, const float* x, const void* wimg, const float* bias, const float* out4,
                 const void* dout, const float* sums, float* stats, void* out, float* dw_tapmajor, float* dbias,
                 int B, int D, int H, int W, int train, float drop_p, uint32_t seed, const uint32_t* seed_epoch,
                 hipStream_t st) {
    MM_REQUIRE(x && wimg && B > 0 && D > 0 && H > 0 && W > 0, "conv3d_l1: null/invalid");
    MM_REQUIRE(D % 2 == 0 && H % 2 == 0 && W % 2 == 0, "conv3d_l1: D,H,W must be even (MaxPool3d(2))");
    MM_REQUIRE(l1_fits(B, D, H, W), "conv3d_l1: more than 2^31 pooled output elements");
    MM_REQUIRE(mode >= 0 && mode <= 3, "conv3d_l1: mode");  /* mode 4 has its own entry point */
    MM_REQUIRE(mode == 0 ? stats != nullptr : out4 != nullptr, "conv3d_l1: stats/out4");
    MM_REQUIRE(mode != 1 || out, "conv3d_l1: out");
    MM_REQUIRE(mode < 2 || dout, "conv3d_l1: dout");
    MM_REQUIRE(mode != 2 || stats, "conv3d_l1: sums_out");
    MM_REQUIRE(mode != 3 || (dw_tapmajor && (!train || sums)), "conv3d_l1: dw/sums");
    L1Args a;
    l1_fill(a, x, wimg, bias, out4, B, D, H, W, train, drop_p, seed, seed_epoch);
    a.dout = (const bf16*)dout; a.sums = sums; a.stats = stats; a.out = (bf16*)out; a.dw = dw_tapmajor; a.dbias = dbias;
    const int ntiles = l1_tiles(B, D, H, W);
    const bool full = H % 8 == 0 && W % 32 == 0;
    const int g3 = l1_grid(ntiles, 3), g2 = l1_grid(ntiles, 2);
    switch (mode * 2 + (full ? 1 : 0)) {
        case 0: hipLaunchKernelGGL((conv3d_l1_kernel<0, false>), dim3(g3), dim3(256), 0, st, a); break;
        case 1: hipLaunchKernelGGL((conv3d_l1_kernel<0, true>), dim3(g3), dim3(256), 0, st, a); break;
        case 2: hipLaunchKernelGGL((conv3d_l1_kernel<1, false>), dim3(g3), dim3(256), 0, st, a); break;
        case 3: hipLaunchKernelGGL((conv3d_l1_kernel<1, true>), dim3(g3), dim3(256), 0, st, a); break;
        case 4: hipLaunchKernelGGL((conv3d_l1_kernel<2, false>), dim3(g2), dim3(256), 0, st, a); break;
        case 5: hipLaunchKernelGGL((conv3d_l1_kernel<2, true>), dim3(g2), dim3(256), 0, st, a); break;
        case 6: hipLaunchKernelGGL((conv3d_l1_kernel<3, false>), dim3(g2), dim3(256), 0, st, a); break;
        default: hipLaunchKernelGGL((conv3d_l1_kernel<3, true>), dim3(g2), dim3(256), 0, st, a); break;
    }
    return mm_check_launch("conv3d_l1");
}

int mm_conv3d_l1_fwd_winners(const float* x, const void* wimg, const float* bias, const float* out4, void* out, void* arg,
                             int B, int D, int H, int W, int train, float drop_p, uint32_t seed,
                             const uint32_t* seed_epoch, hipStream_t st) {
    MM_REQUIRE(x && wimg && out4 && out && arg && B > 0 && D > 0 && H > 0 && W > 0, "conv3d_l1_fwd_winners: null/invalid");
    MM_REQUIRE(D % 2 == 0 && H % 2 == 0 && W % 2 == 0, "conv3d_l1_fwd_winners: D,H,W must be even (MaxPool3d(2))");
    MM_REQUIRE(l1_fits(B, D, H, W), "conv3d_l1_fwd_winners: more than 2^31 pooled output elements");
    L1Args a;
    l1_fill(a, x, wimg, bias, out4, B, D, H, W, train, drop_p, seed, seed_epoch);
    a.out = (bf16*)out; a.arg = (uint8_t*)arg;
    const int grid = l1_grid(l1_tiles(B, D, H, W), 3);
    if (H % 8 == 0 && W % 32 == 0) hipLaunchKernelGGL((conv3d_l1_kernel<1, true, true>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv3d_l1_kernel<1, false, true>), dim3(grid), dim3(256), 0, st, a);
    return mm_check_launch("conv3d_l1_fwd_winners");
}

int mm_conv3d_l1_fwd_fin(const float* x, const void* wimg, const float* bias, const void* bn_fin_host, void* out, int B, int D,
                         int H, int W, float drop_p, uint32_t seed, const uint32_t* seed_epoch, hipStream_t st) {
    MM_REQUIRE(x && wimg && out && B > 0 && D > 0 && H > 0 && W > 0, "conv3d_l1_fwd_fin: null/invalid");
    MM_REQUIRE(D % 2 == 0 && H % 2 == 0 && W % 2 == 0, "conv3d_l1_fwd_fin: D,H,W must be even (MaxPool3d(2))");
    MM_REQUIRE(l1_fits(B, D, H, W), "conv3d_l1_fwd_fin: more than 2^31 pooled output elements");
    L1Args a;
    l1_fill(a, x, wimg, bias, nullptr, B, D, H, W, 1, drop_p, seed, seed_epoch);
    MM_REQUIRE(bn_fin_from_host(a.fin, bn_fin_host, 32), "conv3d_l1_fwd_fin: incomplete mm_bn_fin_t (null pointer or count < 1)");
    a.out4 = a.fin.out4; a.out = (bf16*)out;
    const int grid = l1_grid(l1_tiles(B, D, H, W), 3);
    if (H % 8 == 0 && W % 32 == 0) hipLaunchKernelGGL((conv3d_l1_kernel<1, true, false, true>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv3d_l1_kernel<1, false, false, true>), dim3(grid), dim3(256), 0, st, a);
    return mm_check_launch("conv3d_l1_fwd_fin");
}

int mm_conv3d_l1_gram(const float* x, const void* wimg, const float* bias, float* gram, float* stats, int B, int D, int H,
                      int W, hipStream_t st) {
    MM_REQUIRE(x && wimg && gram && stats && B > 0 && D > 0 && H > 0 && W > 0, "conv3d_l1_gram: null/invalid");
    MM_REQUIRE(D % 2 == 0 && H % 2 == 0 && W % 2 == 0, "conv3d_l1_gram: D,H,W must be even (MaxPool3d(2))");
    L1Args a;
    l1_fill(a, x, wimg, bias, nullptr, B, D, H, W, 1, 0.f, 0, nullptr);
    a.gram = gram; a.stats = stats;
    const int grid = l1_grid(l1_tiles(B, D, H, W), 3);
    if (H % 8 == 0 && W % 32 == 0) hipLaunchKernelGGL((l1_gram_kernel<true>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((l1_gram_kernel<false>), dim3(grid), dim3(256), 0, st, a);
    return mm_check_launch("conv3d_l1_gram");
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
                     float* sums_out, float* a1, const float* gram, float* dw, float* dbias, int B,
                     int D, int H, int W, int train, float drop_p, uint32_t seed, const uint32_t* seed_epoch,
                     hipStream_t st) {
    MM_REQUIRE(x && wimg && out4 && dout && sums_out && a1 && dw && B > 0, "conv3d_l1_bwd: null/invalid");
    MM_REQUIRE(!train || gram, "conv3d_l1_bwd: train-mode BatchNorm needs the Gram workspace of the forward pass");
    MM_REQUIRE(D % 2 == 0 && H % 2 == 0 && W % 2 == 0, "conv3d_l1_bwd: D,H,W must be even (MaxPool3d(2))");
    MM_REQUIRE(l1_fits(B, D, H, W), "conv3d_l1_bwd: more than 2^31 pooled output elements");
    L1Args a;
    l1_fill(a, x, wimg, bias, out4, B, D, H, W, train, drop_p, seed, seed_epoch);
    a.dout = (const bf16*)dout; a.stats = sums_out; a.dw = a1;
    const int grid = l1_grid(l1_tiles(B, D, H, W), L1_BWD_WAVES);
    if (H % 8 == 0 && W % 32 == 0) hipLaunchKernelGGL((conv3d_l1_kernel<4, true>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv3d_l1_kernel<4, false>), dim3(grid), dim3(256), 0, st, a);
    hipLaunchKernelGGL(l1_combine_kernel, dim3(ceil_div(32 * 27 * 16, 256)), dim3(256), 0, st, a1, gram,
                       (const bf16*)wimg, bias, sums_out, out4, dw, dbias, a.inv_count, train);
    return mm_check_launch("conv3d_l1_bwd");
}

int mm_transpose_add(const float* src, float* dst, int R, int C, int nrep, hipStream_t st) {
    MM_REQUIRE(src && dst && R > 0 && C > 0, "transpose_add: null");
    MM_REQUIRE(nrep == MM_ACC_REPL, "transpose_add: src is an accumulator workspace of %d replicas", MM_ACC_REPL);
    hipLaunchKernelGGL(transpose_add_kernel, dim3(ceil_div(R * C, 256)), dim3(256), 0, st, src, dst, R, C);
    return mm_check_launch("transpose_add");
}

}  // extern "C"
