// 1-D implicit-GEMM family on bf16 MFMA (fp32 accumulate), channels-last.
//
//   forward : Y[b,t,n]  = sum_{tap,c} X[b, t+tap-pad, c] * W[n, tap, c]
//   (a Linear layer is the taps == 1 case; data-gradient is the same kernel
//    run on dY with the flipped/transposed weight image)
//   wgrad   : dW[n,tap,c] = sum_{b,t} dY[b,t,n] * X[b, t+tap-pad, c]
//
// Layouts: X [B][T][Cin] bf16 (Cin % 16 == 0), W [Cout][taps][Cin] bf16,
// Y [B][T/pool][Cout].  One workgroup = 256 threads = 4 waves computes a
// BM x BN output tile of ONE batch item: the (BM + taps - 1) x KC halo tile of
// X is staged once per Cin-chunk into LDS and every tap reads it at a row
// offset (the im2col matrix is never materialised); W for the chunk sits
// beside it.  Rows are padded by 16 B so the 16-lane groups of ds_read_b128
// hit 16 distinct 16-B slots (row stride = odd multiple of 16 B).
#include "common.h"

#include <cstdio>
#include <cstdlib>

namespace {


constexpr int KPAD = 8;       // +16 B per LDS row
constexpr int A2S = 128 + KPAD; // row stride (elements) of the second GEMM's LDS operand tile (32 rows of 128 bf16)

// BatchNorm-backward reduce pass of the layer BELOW, fused behind the data-gradient GEMM that produces that layer's
// d(out) (epilogue_bn_reduce): the tile's bf16 d(out) values are routed through dropout / pool / act' exactly as
// elementwise.hip's bn_act_bwd_kernel<false> does and summed into the same accumulator workspace
struct BnRed {
    const float* y = nullptr;      // [B][T * pool][N] pre-BatchNorm activations of that layer (nullptr = off)
    const float* out4 = nullptr;   // [4][N] scale, shift, mean, rstd
    float* sums = nullptr;         // [MM_REPL][2][N] workspace (zeroed by the caller): sum dz | sum dz * xhat
    int act = 0, pool = 1, drop_first = 0;
    uint32_t thresh = 0, seed = 0;
    float inv_keep = 1.f;
    const uint32_t* epoch = nullptr;
    uint32_t thresh2 = 0, seed2 = 0;   // the dropout BEHIND the block (PositionalEncoding's), applied to d(out) first:
    float inv_keep2 = 1.f;             // epilogue_ln_bwd only (fp32 d(out))
};

struct EpiArgs {
    const float* scale;       // [N] multiply (nullptr = 1)
    const float* shift;       // [N] add (bias / folded BN shift) (nullptr = 0)
    const float* residual;    // [M][N] fp32 added after activation (nullptr)
    const float* pe;          // [>=T][N] fp32 positional table added per t (nullptr)
    float* stats;             // [2][N] sum / sum-of-squares of v (atomics) (nullptr)
    float* out_f32;           // [M/pool][N]
    bf16* out_bf16;           // [M/pool][N]
    bf16* out_pre;            // [M][N] pre-activation copy (nullptr)
    int act;
    int pool;                 // 1 or 2 (max over adjacent t pairs, after act)
    uint32_t drop_thresh;     // 0 = no dropout
    uint32_t drop_seed;
    float drop_inv_keep;
    const uint32_t* drop_epoch;
    const bf16* gradz;        // backward fusion: v *= act'(gradz[idx]) (nullptr = off)
    int gradz_act;
    // LayerNorm-128 backward fused behind a data-gradient GEMM (ln_x != nullptr): the tile rows are
    // d(LN output); residual = gradient of the skip path; out_f32 / out_bf16 = d(LN input) (bf16 copy
    // carries the consumer's dropout mask); ln_dgb = [REPL][2][128] {dgamma, dbeta} replicas
    const float* ln_x;
    const float* ln_stat;     // [M][2] mean, rstd
    const float* ln_gamma;
    float* ln_dgb;
    // mean over groups of pool_rows consecutive output rows, fused: pool_out[row / pool_rows][n] += out * pool_scale
    float* pool_out;
    int pool_rows;
    float pool_scale;
    // LayerNorm-128 of every finished output row (the NEXT sub-layer's pre-norm), fused: lnf_out bf16 rows,
    // lnf_stat [M][2] mean / rstd (nullable)
    bf16* lnf_out;
    float* lnf_stat;
    const float* lnf_gamma;
    const float* lnf_beta;
    float lnf_eps;
    BnRed bn;
    // second GEMM behind epilogue_ln_bwd (BM = 32, BN = 128): out2 (M, 128) bf16 = out_bf16 rows @ w2 (a 128 x 128 data-
    // gradient weight image) - the data gradient of the Linear whose output, after dropout, was added to this LayerNorm's
    // input (the attention out-projection under norm2): its operand never leaves the workgroup
    const bf16* w2 = nullptr;
    bf16* out2 = nullptr;
    const float* bias2 = nullptr;   // (n2) added to the second GEMM's columns (nullptr = 0)
    int n2 = 128;                   // its output width: w2 is n2 rows of 128 (a multiple of 128)
    int act2 = 0;                   // activation of the second GEMM's output, then dropout (thresh2 / seed2 / inv_keep2, index
    uint32_t thresh2 = 0, seed2 = 0; float inv_keep2 = 1.f;      // row * n2 + column as a launch of its own would use)
    bf16* pre2 = nullptr;           // pre-activation copy (M, n2) bf16 (nullptr = none)
    int res_rows = 0;            // epilogue_ln_bwd: > 0 = `residual` is (M / res_rows, 128): one row for res_rows consecutive rows
};

struct ConvArgs {
    const bf16* x;
    const bf16* w;
    int B, T, Cin, Cout, taps, pad;
    EpiArgs e;
    // split-K (few output tiles, long reduction: config #5's 192-channel k = 7 convolution over ~6 000 input channels is
    // 96 tiles of 98 chunks): workgroup z reduces input channels [z * csplit, (z + 1) * csplit) and stores its raw fp32
    // tile to partial[z][b][t][n]; conv1d_splitk_epilogue_kernel adds the slices in order and runs the epilogue
    float* partial = nullptr;
    int csplit = 0;
};

// Epilogue feature mask of a launch: which of epilogue_rows' optional steps it needs, the activation in bits 16-19 and
// the fused activation derivative in bits 20-23.  The kernel is compiled once per tile shape with every step behind a
// run-time test (FEAT = EF_ANY) and once more for each combination the training step uses, with the unused steps and
// the activation switch compiled out: the generic epilogue cost ~2.4 us per million outputs in branches and dead work
// (FFN-1 forward, 8.4 M outputs: 25.7 us generic, 17.5 us specialised).
enum : unsigned { EF_RES = 1, EF_PE = 2, EF_PRE = 4, EF_GRADZ = 8, EF_STATS = 16, EF_POOLOUT = 32, EF_LNF = 64, EF_POOL2 = 128,
                  EF_DROP = 256, EF_SCALE = 512, EF_F32 = 1024, EF_BF16 = 2048, EF_SHIFT = 4096, EF_LNBWD = 8192,
                  EF_BNRED = 16384, EF_BNPOOL2 = 32768 /* bits 24-27: the fused BatchNorm-backward's activation */,
                  EF_GEMM2 = 1u << 28, EF_ANY = 0xFFFFFFFFu };
static unsigned epi_mask(const EpiArgs& e) {
    return (e.residual ? EF_RES : 0) | (e.pe ? EF_PE : 0) | (e.out_pre ? EF_PRE : 0) | (e.gradz ? EF_GRADZ : 0) | (e.stats ? EF_STATS : 0) |
           (e.pool_out ? EF_POOLOUT : 0) | (e.lnf_out ? EF_LNF : 0) | (e.pool == 2 ? EF_POOL2 : 0) | (e.drop_thresh ? EF_DROP : 0) |
           (e.scale ? EF_SCALE : 0) | (e.out_f32 ? EF_F32 : 0) | (e.out_bf16 ? EF_BF16 : 0) | (e.shift ? EF_SHIFT : 0) | (e.ln_x ? EF_LNBWD : 0) |
           ((unsigned)e.act << 16) | ((unsigned)(e.gradz ? e.gradz_act : 0) << 20) |
           (e.bn.y ? (EF_BNRED | (e.bn.pool == 2 ? EF_BNPOOL2 : 0) | ((unsigned)e.bn.act << 24)) : 0) | (e.w2 ? EF_GEMM2 : 0);
}


// Epilogue through LDS: the accumulator tile is parked as fp32 [BM][BN+4], then
// every thread owns one 4-column group (fixed per thread) and walks rows, so
// residual / positional loads and all stores are 16-byte, row-contiguous.
template <int BM, int BN, unsigned FEAT>
__device__ __forceinline__ void epilogue_rows(const float* Cs, const EpiArgs& e, int tid, int b, int t0, int T,
                                              int n0, int N, float* sstat, bf16* a2 = nullptr) {
    // no implicit FMA contraction in here: which multiply-adds get fused would depend on what a specialisation folds
    // away, and the variants of one op must agree bit for bit (tests compare them); the two intended FMAs are explicit
#pragma clang fp contract(off)
    constexpr bool ANY = FEAT == EF_ANY;
#define EF_ON(bit, runtime) (ANY ? (bool)(runtime) : ((FEAT & (bit)) != 0))
    const int act = ANY ? e.act : (int)((FEAT >> 16) & 15u);
    const int gradz_act = ANY ? e.gradz_act : (int)((FEAT >> 20) & 15u);
    constexpr int LDC = BN + 4;
    constexpr int CG = BN / 4;                 // column groups
    constexpr int RPP = 256 / CG;              // rows per pass
    const int cg = tid % CG, rr = tid / CG;
    const int n = n0 + cg * 4;
    const bool nok = n < N;                    // N % 4 == 0 is required
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (nok) {
        if (EF_ON(EF_SCALE, e.scale)) sc = *reinterpret_cast<const float4*>(e.scale + n);
        if (EF_ON(EF_SHIFT, e.shift)) sh = *reinterpret_cast<const float4*>(e.shift + n);
    }
    const float scs[4] = {sc.x, sc.y, sc.z, sc.w}, shs[4] = {sh.x, sh.y, sh.z, sh.w};
    float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0}, pp[4] = {0, 0, 0, 0};
    const bool drop = EF_ON(EF_DROP, e.drop_thresh);
    const uint32_t dseed = drop ? mm_eff_seed(e.drop_seed, e.drop_epoch) : 0u;
    const int step = ANY ? e.pool : ((FEAT & EF_POOL2) ? 2 : 1);      // rows consumed per item
    const int To = T / step;
    for (int r0 = rr * step; r0 < BM; r0 += RPP * step) {
        float o[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        bool any = false;
        for (int q = 0; q < step; ++q) {
            const int row = r0 + q, t = t0 + row;
            if (t >= T || !nok) continue;
            any = true;
            const float4 a4 = *reinterpret_cast<const float4*>(Cs + row * LDC + cg * 4);
            float v[4] = {a4.x, a4.y, a4.z, a4.w};
            const size_t idx = ((size_t)b * T + t) * N + n;
            float4 res = make_float4(0.f, 0.f, 0.f, 0.f), pe = res;
            if (EF_ON(EF_RES, e.residual)) res = *reinterpret_cast<const float4*>(e.residual + idx);
            if (EF_ON(EF_PE, e.pe)) pe = *reinterpret_cast<const float4*>(e.pe + (size_t)t * N + n);
            const float rs[4] = {res.x, res.y, res.z, res.w}, ps[4] = {pe.x, pe.y, pe.z, pe.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float val = __builtin_fmaf(v[c], scs[c], shs[c]);
                if (EF_ON(EF_STATS, e.stats)) { s1[c] += val; s2[c] = __builtin_fmaf(val, val, s2[c]); }
                v[c] = val;
            }
            if (EF_ON(EF_PRE, e.out_pre)) {
                bf16x4 pv = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                *reinterpret_cast<bf16x4*>(e.out_pre + idx) = pv;
            }
            if (EF_ON(EF_GRADZ, e.gradz)) {
                const bf16x4 zz = *reinterpret_cast<const bf16x4*>(e.gradz + idx);
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] *= act_grad((float)zz[c], gradz_act);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float val = apply_act(v[c], act);
                const float rp = rs[c] + ps[c];
                if (drop) val = __builtin_fmaf(val, dropout_scale(dseed, (uint32_t)(idx + c), e.drop_thresh, e.drop_inv_keep), rp);
                else val += rp;
                o[c] = fmaxf(o[c], val);
            }
        }
        if (!any) continue;
        if (EF_ON(EF_POOLOUT, e.pool_out))
#pragma unroll
            for (int c = 0; c < 4; ++c) pp[c] += o[c];
        const int t = t0 + r0;
        const size_t oi = ((size_t)b * To + t / step) * N + n;
        if constexpr (BN == 128) {
            if (EF_ON(EF_LNF, e.lnf_out)) {                       // host guarantees N == 128, pool == 1: 32 lanes hold this row
                float sm = (o[0] + o[1]) + (o[2] + o[3]);
                sm = half32_sum(sm);
                const float mean = sm * (1.f / 128.f);
                const float d0 = o[0] - mean, d1 = o[1] - mean, d2 = o[2] - mean, d3 = o[3] - mean;
                float sq = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                sq = half32_sum(sq);
                const float rstd = rsqrtf(sq * (1.f / 128.f) + e.lnf_eps);
                const float4 g4 = *reinterpret_cast<const float4*>(e.lnf_gamma + n);
                const float4 b4 = *reinterpret_cast<const float4*>(e.lnf_beta + n);
                bf16x4 hv = {(bf16)(d0 * rstd * g4.x + b4.x), (bf16)(d1 * rstd * g4.y + b4.y),
                             (bf16)(d2 * rstd * g4.z + b4.z), (bf16)(d3 * rstd * g4.w + b4.w)};
                *reinterpret_cast<bf16x4*>(e.lnf_out + oi) = hv;
                if (a2) *reinterpret_cast<bf16x4*>(a2 + r0 * A2S + cg * 4) = hv;              // operand tile of the second GEMM
                if (e.lnf_stat && cg == 0) {
                    const size_t m = (size_t)b * To + t;
                    e.lnf_stat[2 * m] = mean; e.lnf_stat[2 * m + 1] = rstd;
                }
            }
        }
        if (EF_ON(EF_F32, e.out_f32)) *reinterpret_cast<float4*>(e.out_f32 + oi) = make_float4(o[0], o[1], o[2], o[3]);
        if (EF_ON(EF_BF16, e.out_bf16)) {
            bf16x4 ov = {(bf16)o[0], (bf16)o[1], (bf16)o[2], (bf16)o[3]};
            *reinterpret_cast<bf16x4*>(e.out_bf16 + oi) = ov;
        }
    }
    if (EF_ON(EF_POOLOUT, e.pool_out)) {
        // fused mean over rows (all rows of this tile belong to one group: pool_rows % BM == 0, host-checked)
        __syncthreads();
        float* part = const_cast<float*>(Cs);              // [RPP][BN]
        *reinterpret_cast<float4*>(part + rr * BN + cg * 4) = make_float4(pp[0], pp[1], pp[2], pp[3]);
        __syncthreads();
        const size_t grp = ((size_t)b * T + t0) / e.pool_rows;
        for (int i = tid; i < BN; i += 256)
            if (n0 + i < N) {
                float s = 0.f;
#pragma unroll
                for (int r = 0; r < RPP; ++r) s += part[r * BN + i];
                acc_add<MM_ACC_GRAD>(reinterpret_cast<mm_acc_t*>(e.pool_out) + grp * N + n0 + i, s * e.pool_scale);
            }
        if (EF_ON(EF_STATS, e.stats)) __syncthreads();
    }
    if (EF_ON(EF_STATS, e.stats)) {
        // block reduction of the per-thread column sums: plain stores into the (now dead) C tile, then a
        // column walk.  LDS float atomics with RPP-way same-address conflicts cost ~2 us per workgroup.
        __syncthreads();                                   // every thread is done reading Cs
        float* part = const_cast<float*>(Cs);              // [RPP][2][BN]  (RPP * 2 * BN = 2048 floats <= BM * LDC)
        static_assert(RPP * 2 * BN <= BM * LDC, "partials fit the C tile");
        *reinterpret_cast<float4*>(part + (rr * 2 + 0) * BN + cg * 4) = make_float4(s1[0], s1[1], s1[2], s1[3]);
        *reinterpret_cast<float4*>(part + (rr * 2 + 1) * BN + cg * 4) = make_float4(s2[0], s2[1], s2[2], s2[3]);
        __syncthreads();
        mm_acc_t* rep = acc_rep(e.stats, blockIdx.x % MM_ACC_REPL, 2 * (size_t)N);
        for (int i = tid; i < 2 * BN; i += 256) {
            const int which = i / BN, col = i % BN;
            if (n0 + col < N) {
                float s = 0.f;
#pragma unroll
                for (int r = 0; r < RPP; ++r) s += part[(r * 2 + which) * BN + col];
                acc_add<MM_ACC_STAT>(&rep[which * N + n0 + col], s);
            }
        }
    }
#undef EF_ON
}

// dgrad GEMM -> bf16 d(out) of the layer below + that layer's BatchNorm-backward reduce pass (host: no scale / shift /
// activation / pooling of this GEMM's own, Cout == the BatchNorm's channel count).  Thread layout as epilogue_rows: one
// 4-column group per thread, BM / RPP rows; the rows' pre-BN values are fetched before the first row is touched.
struct BnDz { int act, pool, drop_first; uint32_t thresh, seed; float inv_keep; };
template <int BM, int BN, unsigned FEAT>
__device__ __forceinline__ void epilogue_bn_reduce(const float* Cs, const EpiArgs& e, int tid, int b, int t0, int T,
                                                   int n0, int N) {
#pragma clang fp contract(off)
    constexpr bool ANY = FEAT == EF_ANY;
    constexpr int LDC = BN + 4, CG = BN / 4, RPP = 256 / CG, NR = BM / RPP;
    const int cg = tid % CG, rr = tid / CG;
    const int n = n0 + cg * 4;
    const bool nok = n < N;
    BnDz bn;
    bn.act = ANY ? e.bn.act : (int)((FEAT >> 24) & 15u);
    bn.pool = ANY ? e.bn.pool : ((FEAT & EF_BNPOOL2) ? 2 : 1);
    bn.drop_first = e.bn.drop_first; bn.thresh = e.bn.thresh; bn.inv_keep = e.bn.inv_keep;
    bn.seed = mm_eff_seed(e.bn.seed, e.bn.epoch);
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 y0[NR], y1[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const int t = t0 + rr + k * RPP;
        const bool ok = nok && t < T;                    // rows / columns outside the tensor read element 0 (unused below)
        const size_t in0 = ok ? ((size_t)b * T + t) * bn.pool * N + n : 0;
        y0[k] = *reinterpret_cast<const float4*>(e.bn.y + in0);
        y1[k] = *reinterpret_cast<const float4*>(e.bn.y + in0 + (bn.pool == 2 ? N : 0));
    }
    float4 c4[4] = {z4, z4, z4, z4};
    if (nok)
#pragma unroll
        for (int q = 0; q < 4; ++q) c4[q] = *reinterpret_cast<const float4*>(e.bn.out4 + (size_t)q * N + n);
    const float scs[4] = {c4[0].x, c4[0].y, c4[0].z, c4[0].w}, shs[4] = {c4[1].x, c4[1].y, c4[1].z, c4[1].w};
    const float mus[4] = {c4[2].x, c4[2].y, c4[2].z, c4[2].w}, rss[4] = {c4[3].x, c4[3].y, c4[3].z, c4[3].w};
    float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const int row = rr + k * RPP, t = t0 + row;
        if (!nok || t >= T) continue;
        const float4 a4 = *reinterpret_cast<const float4*>(Cs + row * LDC + cg * 4);
        const size_t oi = ((size_t)b * T + t) * N + n;
        const size_t in0 = ((size_t)b * T + t) * bn.pool * N + n;
        const bf16x4 ov = {(bf16)a4.x, (bf16)a4.y, (bf16)a4.z, (bf16)a4.w};
        *reinterpret_cast<bf16x4*>(e.out_bf16 + oi) = ov;
        const float y0s[4] = {y0[k].x, y0[k].y, y0[k].z, y0[k].w}, y1s[4] = {y1[k].x, y1[k].y, y1[k].z, y1[k].w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float d0, d1;                                   // the stand-alone pass reads the bf16 d(out): so does this one
            bn_dz_pair<-1, 0>(bn, y0s[c], y1s[c], scs[c], shs[c], (float)ov[c], (uint32_t)(in0 + c), (uint32_t)(in0 + N + c),
                              (uint32_t)(oi + c), d0, d1);
            const float xh0 = (y0s[c] - mus[c]) * rss[c], xh1 = (y1s[c] - mus[c]) * rss[c];
            s1[c] += d0 + d1;
            s2[c] += d0 * xh0 + d1 * xh1;
        }
    }
    __syncthreads();                                       // every thread is done reading Cs
    float* part = const_cast<float*>(Cs);                  // [RPP][2][BN]
    static_assert(RPP * 2 * BN <= BM * LDC, "partials fit the C tile");
    *reinterpret_cast<float4*>(part + (rr * 2 + 0) * BN + cg * 4) = make_float4(s1[0], s1[1], s1[2], s1[3]);
    *reinterpret_cast<float4*>(part + (rr * 2 + 1) * BN + cg * 4) = make_float4(s2[0], s2[1], s2[2], s2[3]);
    __syncthreads();
    mm_acc_t* rep = acc_rep(e.bn.sums, blockIdx.x % MM_ACC_REPL, 2 * (size_t)N);
    for (int i = tid; i < 2 * BN; i += 256) {
        const int which = i / BN, col = i % BN;
        if (n0 + col < N) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < RPP; ++r) s += part[(r * 2 + which) * BN + col];
            acc_add<MM_ACC_GRAD>(&rep[which * N + n0 + col], s);
        }
    }
}

// dgrad GEMM -> LayerNorm backward in one pass (N == BN == 128, T % BM == 0: checked on the host).
// 32 lanes own one row (4 columns each): the two row means are 5-step half-wave shuffles; every
// thread keeps its 4 columns' dgamma / dbeta partial sums over the rows it walks.
template <int BM, int BN, unsigned FEAT>
__device__ __forceinline__ void epilogue_ln_bwd(const float* Cs, const EpiArgs& e, int tid, int b, int t0, int T,
                                                float* sstat, bf16* a2 = nullptr) {
    static_assert(BN == 128, "LayerNorm-128 epilogue");
    constexpr bool ANY = FEAT == EF_ANY;
#define EF_ON(bit, runtime) (ANY ? (bool)(runtime) : ((FEAT & (bit)) != 0))
    constexpr int LDC = BN + 4;
    const int cg = tid & 31, rr = tid >> 5;
    const float4 gg = *reinterpret_cast<const float4*>(e.ln_gamma + cg * 4);
    const float gam[4] = {gg.x, gg.y, gg.z, gg.w};
    float sh[4] = {0.f, 0.f, 0.f, 0.f};
    if (EF_ON(EF_SHIFT, e.shift)) {
        const float4 s4 = *reinterpret_cast<const float4*>(e.shift + cg * 4);
        sh[0] = s4.x; sh[1] = s4.y; sh[2] = s4.z; sh[3] = s4.w;
    }
    const bool drop = EF_ON(EF_DROP, e.drop_thresh);
    const uint32_t dseed = drop ? mm_eff_seed(e.drop_seed, e.drop_epoch) : 0u;
    float ag[4] = {0, 0, 0, 0}, ab[4] = {0, 0, 0, 0};
    // BatchNorm-backward reduce of the conv block whose output (+ positional table, dropout) IS this LayerNorm's input:
    // the rows leaving here are that block's fp32 d(out) (EnhancedERPEncoder: conv block 3 under the first transformer block)
    const bool bnred = EF_ON(EF_BNRED, e.bn.y);
    BnDz bn;
    bn.act = ANY ? e.bn.act : (int)((FEAT >> 24) & 15u);
    bn.pool = 1; bn.drop_first = 1; bn.thresh = e.bn.thresh; bn.inv_keep = e.bn.inv_keep;
    bn.seed = bnred ? mm_eff_seed(e.bn.seed, e.bn.epoch) : 0u;
    const uint32_t bseed2 = bnred ? mm_eff_seed(e.bn.seed2, e.bn.epoch) : 0u;
    float4 by[BM / 8];
    float bsc[4] = {0, 0, 0, 0}, bsh[4] = {0, 0, 0, 0}, bmu[4] = {0, 0, 0, 0}, brs[4] = {0, 0, 0, 0};
    float t1[4] = {0, 0, 0, 0}, t2[4] = {0, 0, 0, 0};
    if (bnred) {
#pragma unroll
        for (int k = 0; k < BM / 8; ++k)
            by[k] = *reinterpret_cast<const float4*>(e.bn.y + ((size_t)b * T + t0 + rr + 8 * k) * 128 + cg * 4);
        const float4 c0 = *reinterpret_cast<const float4*>(e.bn.out4 + cg * 4), c1 = *reinterpret_cast<const float4*>(e.bn.out4 + 128 + cg * 4);
        const float4 c2 = *reinterpret_cast<const float4*>(e.bn.out4 + 256 + cg * 4), c3 = *reinterpret_cast<const float4*>(e.bn.out4 + 384 + cg * 4);
        bsc[0] = c0.x; bsc[1] = c0.y; bsc[2] = c0.z; bsc[3] = c0.w; bsh[0] = c1.x; bsh[1] = c1.y; bsh[2] = c1.z; bsh[3] = c1.w;
        bmu[0] = c2.x; bmu[1] = c2.y; bmu[2] = c2.z; bmu[3] = c2.w; brs[0] = c3.x; brs[1] = c3.y; brs[2] = c3.z; brs[3] = c3.w;
    }
#pragma unroll
    for (int row = rr; row < BM; row += 8) {
        const size_t m = (size_t)b * T + t0 + row;
        const size_t base = m * 128 + cg * 4;
        const float4 a4 = *reinterpret_cast<const float4*>(Cs + row * LDC + cg * 4);
        const float4 xv = *reinterpret_cast<const float4*>(e.ln_x + base);
        float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (EF_ON(EF_RES, e.residual))
            rv = *reinterpret_cast<const float4*>(e.residual + (e.res_rows ? (size_t)((unsigned)m / (unsigned)e.res_rows) * 128 + cg * 4 : base));
        const float2 st = *reinterpret_cast<const float2*>(e.ln_stat + 2 * m);
        const float dyv[4] = {a4.x + sh[0], a4.y + sh[1], a4.z + sh[2], a4.w + sh[3]};
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, rs[4] = {rv.x, rv.y, rv.z, rv.w};
        float xh[4], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            xh[c] = (xs[c] - st.x) * st.y;
            const float gh = dyv[c] * gam[c];
            s1 += gh; s2 += gh * xh[c];
            ag[c] += dyv[c] * xh[c]; ab[c] += dyv[c];
        }
        s1 = half32_sum(s1); s2 = half32_sum(s2);
        s1 *= (1.f / 128.f); s2 *= (1.f / 128.f);
        float o4[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) o4[c] = st.y * (dyv[c] * gam[c] - s1 - xh[c] * s2) + rs[c];
        if (EF_ON(EF_F32, e.out_f32)) *reinterpret_cast<float4*>(e.out_f32 + base) = make_float4(o4[0], o4[1], o4[2], o4[3]);
        if (EF_ON(EF_BF16, e.out_bf16)) {
            bf16x4 ob;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                ob[c] = (bf16)(drop ? o4[c] * dropout_scale(dseed, (uint32_t)(base + c), e.drop_thresh, e.drop_inv_keep)
                                             : o4[c]);
            *reinterpret_cast<bf16x4*>(e.out_bf16 + base) = ob;
            if (a2) *reinterpret_cast<bf16x4*>(a2 + row * A2S + cg * 4) = ob;
        }
        if (bnred) {
            const float4 yv = by[(row - rr) / 8];
            const float ys[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float g = o4[c];
                if (e.bn.thresh2) g *= dropout_scale(bseed2, (uint32_t)(base + c), e.bn.thresh2, e.bn.inv_keep2);
                float d0, d1;
                bn_dz_pair<-1, 1>(bn, ys[c], ys[c], bsc[c], bsh[c], g, (uint32_t)(base + c), 0u, 0u, d0, d1);
                t1[c] += d0;
                t2[c] += d0 * ((ys[c] - bmu[c]) * brs[c]);
            }
        }
    }
    if (bnred) {
        __syncthreads();                                   // every thread is done reading Cs
        float* part = const_cast<float*>(Cs);              // [8 row groups][sum dz 128 | sum dz xhat 128]
        *reinterpret_cast<float4*>(part + rr * 256 + cg * 4) = make_float4(t1[0], t1[1], t1[2], t1[3]);
        *reinterpret_cast<float4*>(part + rr * 256 + 128 + cg * 4) = make_float4(t2[0], t2[1], t2[2], t2[3]);
        __syncthreads();
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) s += part[r * 256 + tid];
        acc_add<MM_ACC_GRAD>(acc_rep(e.bn.sums, blockIdx.x % MM_ACC_REPL, 256) + tid, s);
    }
    if (e.ln_dgb) {
        __syncthreads();                                   // every thread is done reading Cs
        float* part = const_cast<float*>(Cs);              // [8 row groups][dgamma 128 | dbeta 128]
        static_assert(8 * 256 <= BM * LDC, "partials fit the C tile");
        *reinterpret_cast<float4*>(part + rr * 256 + cg * 4) = make_float4(ag[0], ag[1], ag[2], ag[3]);
        *reinterpret_cast<float4*>(part + rr * 256 + 128 + cg * 4) = make_float4(ab[0], ab[1], ab[2], ab[3]);
        __syncthreads();
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) s += part[r * 256 + tid];
        acc_add<MM_ACC_GRAD>(acc_rep(e.ln_dgb, blockIdx.x % MM_ACC_REPL, 256) + tid, s);
    }
#undef EF_ON
}

// out2[32 rows][n2] = a2[32][128] (bf16 rows this workgroup has just finished, in LDS) x w2 (+ bias2): wave wn owns columns
// 128 j + 32 wn .. of every 128-column group j; B fragments straight from the L2-resident weight image, k ascending as the
// main loop's, fp32 accumulate, one rounding to bf16 - bit-identical to a launch of its own on the same rows.
__device__ __forceinline__ void second_gemm(const bf16* a2, const EpiArgs& e, size_t row0, int wn, int lr, int lh) {
#pragma clang fp contract(off)
    bf16x8 af[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) af[ks] = *reinterpret_cast<const bf16x8*>(a2 + lr * A2S + ks * 16 + lh * 8);
    for (int j = 0; j < e.n2 / 128; ++j) {
        const int n = 128 * j + 32 * wn + lr;
        const bf16* wrow = e.w2 + (size_t)n * 128 + lh * 8;
        bf16x8 bfr[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) bfr[ks] = *reinterpret_cast<const bf16x8*>(wrow + ks * 16);
        f32x16 c2;
#pragma unroll
        for (int r = 0; r < 16; ++r) c2[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], bfr[ks], c2, 0, 0, 0);
        const float bias = e.bias2 ? e.bias2[n] : 0.f;
        bf16* orow = e.out2 + row0 * e.n2 + n;
        if (!e.act2 && !e.thresh2 && !e.pre2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                orow[(size_t)row * e.n2] = (bf16)(e.bias2 ? c2[r] + bias : c2[r]);
            }
        } else {
            // bias -> pre-activation copy -> activation -> dropout, the arithmetic of epilogue_rows (FFN-1 forward)
            const uint32_t dseed = e.thresh2 ? mm_eff_seed(e.seed2, e.drop_epoch) : 0u;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const size_t idx = (row0 + row) * e.n2 + n;
                const float v = e.bias2 ? c2[r] + bias : c2[r];
                if (e.pre2) e.pre2[idx] = (bf16)v;
                float val = apply_act(v, e.act2);
                if (e.thresh2) val = __builtin_fmaf(val, dropout_scale(dseed, (uint32_t)idx, e.thresh2, e.inv_keep2), 0.f);
                orow[(size_t)row * e.n2] = (bf16)val;
            }
        }
    }
}

// FEAT: epilogue combination (EF_ANY = all run-time); TAPS > 0: compiled for that tap count (1 = the Linear layers)
template <int BM, int BN, int WM, int WN, int KCT, unsigned FEAT, int TAPS>
__global__ __launch_bounds__(256, 2) void conv1d_fwd_kernel(ConvArgs a) {
    if (TAPS > 0) a.taps = TAPS;
    constexpr int TM = BM / (WM * 32);
    constexpr int TN = BN / (WN * 32);
    static_assert(WM * WN == 4, "4 waves");
    constexpr int AS = KCT + KPAD;                // LDS row stride (elements)
    constexpr int SEGS = KCT / 8;                 // 16-B segments per row
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tilesT = (a.T + BM - 1) / BM;
    const int b = blockIdx.x / tilesT;
    const int t0 = (blockIdx.x % tilesT) * BM;
    const int n0 = blockIdx.y * BN;
    const int arows = BM + a.taps - 1;
    bf16* As = reinterpret_cast<bf16*>(smem);
    bf16* Ws = As + arows * AS;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const bf16* xb = a.x + (size_t)b * a.T * a.Cin;
    const int lr = lane & 31, lh = lane >> 5;
    const int wrows = BN * a.taps;
    const int wvalid = (a.Cout - n0) * a.taps;    // rows >= wvalid are beyond Cout -> zeros

    // staging goes through register batches: all loads of a batch are in flight before the first LDS
    // write (a load->store loop serialises on load latency).  The first activation batch and the first
    // weight batch are issued TOGETHER (one global round trip per K chunk for the Linear layers instead of
    // two); whatever does not fit (k > 1 convs) follows in further batches.
    // (only the 32-row Linear tile: at 64 rows the 12 staging registers sets cost the third wave per SIMD)
    constexpr bool MERGE = BM == 32;
    constexpr int NA = MERGE ? 2 : 4, NB = 8;
    const bf16* wb = a.w + (size_t)n0 * a.taps * a.Cin;
    const int nA = arows * SEGS, nW = wrows * SEGS;
    auto load_a = [&](int s, int c0) {
        const int r = s / SEGS, sg = s % SEGS;
        const int t = t0 - a.pad + r;
        return (s < nA && t >= 0 && t < a.T) ? *reinterpret_cast<const uint4*>(xb + (size_t)t * a.Cin + c0 + sg * 8)
                                             : make_uint4(0, 0, 0, 0);
    };
    auto load_w = [&](int s, int c0) {
        const int r = s / SEGS, sg = s % SEGS;             // r = n_local * taps + tap
        return (s < nW && r < wvalid) ? *reinterpret_cast<const uint4*>(wb + (size_t)r * a.Cin + c0 + sg * 8)
                                      : make_uint4(0, 0, 0, 0);
    };
    const int c_lo = a.partial ? (int)blockIdx.z * a.csplit : 0;
    const int c_hi = a.partial ? min(a.Cin, c_lo + a.csplit) : a.Cin;
    for (int c0 = c_lo; c0 < c_hi; c0 += KCT) {
        if (c0 != c_lo) __syncthreads();
        if constexpr (MERGE) {
            uint4 va[NA], vw[NB];
#pragma unroll
            for (int i = 0; i < NA; ++i) va[i] = load_a(i * 256 + tid, c0);
#pragma unroll
            for (int i = 0; i < NB; ++i) vw[i] = load_w(i * 256 + tid, c0);
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int s = i * 256 + tid;
                if (s < nA) *reinterpret_cast<uint4*>(As + (s / SEGS) * AS + (s % SEGS) * 8) = va[i];
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int s = i * 256 + tid;
                if (s < nW) *reinterpret_cast<uint4*>(Ws + (s / SEGS) * AS + (s % SEGS) * 8) = vw[i];
            }
        }
        for (int base = MERGE ? NA * 256 : 0; base < nA; base += NA * 256) {
            uint4 v[NA];
#pragma unroll
            for (int i = 0; i < NA; ++i) v[i] = load_a(base + i * 256 + tid, c0);
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int s = base + i * 256 + tid;
                if (s < nA) *reinterpret_cast<uint4*>(As + (s / SEGS) * AS + (s % SEGS) * 8) = v[i];
            }
        }
        for (int base = MERGE ? NB * 256 : 0; base < nW; base += NB * 256) {
            uint4 v[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) v[i] = load_w(base + i * 256 + tid, c0);
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int s = base + i * 256 + tid;
                if (s < nW) *reinterpret_cast<uint4*>(Ws + (s / SEGS) * AS + (s % SEGS) * 8) = v[i];
            }
        }
        __syncthreads();
        for (int tap = 0; tap < a.taps; ++tap) {
#pragma unroll
            for (int ks = 0; ks < KCT; ks += 16) {
                bf16x8 af[TM], bfr[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int row = (wm * TM + i) * 32 + lr + tap;
                    af[i] = *reinterpret_cast<const bf16x8*>(As + row * AS + ks + lh * 8);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int nl = (wn * TN + j) * 32 + lr;
                    bfr[j] = *reinterpret_cast<const bf16x8*>(Ws + (nl * a.taps + tap) * AS + ks + lh * 8);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
        }
    }
    __syncthreads();                               // staging LDS is dead: reuse as the C tile
    constexpr int LDC = BN + 4;
    float* Cs = reinterpret_cast<float*>(smem);
    float* sstat = Cs + BM * LDC;                  // [2][BN]
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                Cs[row * LDC + (wn * TN + j) * 32 + lr] = acc[i][j][r];
            }
    if (a.partial) {                               // split-K: the raw tile of this channel range, no epilogue
        __syncthreads();
        float* dst = a.partial + ((size_t)blockIdx.z * a.B + b) * a.T * a.Cout;
        for (int i = tid; i < BM * (BN / 4); i += 256) {
            const int row = i / (BN / 4), n = n0 + (i % (BN / 4)) * 4;
            if (t0 + row < a.T && n < a.Cout)
                *reinterpret_cast<float4*>(dst + (size_t)(t0 + row) * a.Cout + n) = *reinterpret_cast<const float4*>(Cs + row * LDC + (n - n0));
        }
        return;
    }
    if (a.e.stats || a.e.ln_dgb)
        for (int i = tid; i < 2 * BN; i += 256) sstat[i] = 0.f;
    __syncthreads();
    if constexpr (BM == 32 && BN == 128) {
        if ((FEAT == EF_ANY && a.e.ln_x) || (FEAT != EF_ANY && (FEAT & EF_LNBWD))) {
            const bool gemm2 = FEAT == EF_ANY ? a.e.w2 != nullptr : (FEAT & EF_GEMM2) != 0;
            bf16* a2 = gemm2 ? reinterpret_cast<bf16*>(smem + (BM * LDC + 2 * BN) * sizeof(float)) : nullptr;   // behind the C tile
            epilogue_ln_bwd<BM, BN, FEAT>(Cs, a.e, tid, b, t0, a.T, sstat, a2);
            if (gemm2) {
                __syncthreads();
                second_gemm(a2, a.e, (size_t)b * a.T + t0, wn, lr, lh);
            }
            return;
        }
    }
    if constexpr (BM == 64 && BN == 64) {
        if ((FEAT == EF_ANY && a.e.bn.y) || (FEAT != EF_ANY && (FEAT & EF_BNRED))) {
            epilogue_bn_reduce<BM, BN, FEAT>(Cs, a.e, tid, b, t0, a.T, n0, a.Cout);
            return;
        }
    }
    if constexpr (BM == 32 && BN == 128) {
        // forward: the LayerNorm rows of the fused next pre-norm (EF_LNF) feed a second GEMM (the next block's QKV projection)
        const bool gemm2 = FEAT == EF_ANY ? a.e.w2 != nullptr : (FEAT & EF_GEMM2) != 0;
        if (gemm2) {
            bf16* a2 = reinterpret_cast<bf16*>(smem + (BM * LDC + 2 * BN) * sizeof(float));
            epilogue_rows<BM, BN, FEAT>(Cs, a.e, tid, b, t0, a.T, n0, a.Cout, sstat, a2);
            __syncthreads();
            second_gemm(a2, a.e, (size_t)b * a.T + t0, wn, lr, lh);
            return;
        }
    }
    epilogue_rows<BM, BN, FEAT>(Cs, a.e, tid, b, t0, a.T, n0, a.Cout, sstat);
}

// second half of a split-K launch: the slices are added in slice order (same bits every run) into the C tile, then the
// ordinary epilogue runs on it (generic form: every step behind its run-time test, bit-equal to the compiled-in forms)
template <int BM, int BN>
__global__ __launch_bounds__(256) void conv1d_splitk_epilogue_kernel(ConvArgs a, int nsplit) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int LDC = BN + 4;
    float* Cs = reinterpret_cast<float*>(smem);
    float* sstat = Cs + BM * LDC;
    const int tid = threadIdx.x;
    const int tilesT = (a.T + BM - 1) / BM;
    const int b = blockIdx.x / tilesT, t0 = (blockIdx.x % tilesT) * BM, n0 = blockIdx.y * BN;
    const size_t slice = (size_t)a.B * a.T * a.Cout;
    const float* src = a.partial + (size_t)b * a.T * a.Cout;
    for (int i = tid; i < BM * (BN / 4); i += 256) {
        const int row = i / (BN / 4), n = n0 + (i % (BN / 4)) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t0 + row < a.T && n < a.Cout) {
            const float* p = src + (size_t)(t0 + row) * a.Cout + n;
            v = *reinterpret_cast<const float4*>(p);
            for (int z = 1; z < nsplit; ++z) {
                const float4 u = *reinterpret_cast<const float4*>(p + z * slice);
                v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
            }
        }
        *reinterpret_cast<float4*>(Cs + row * LDC + (n - n0)) = v;
    }
    if (a.e.stats || a.e.ln_dgb)
        for (int i = tid; i < 2 * BN; i += 256) sstat[i] = 0.f;
    __syncthreads();
    epilogue_rows<BM, BN, EF_ANY>(Cs, a.e, tid, b, t0, a.T, n0, a.Cout, sstat);
}

template <int BM, int BN, int WM, int WN, int KCT, unsigned FEAT, int TAPS = 0>
int launch_fwd_feat(const ConvArgs& a, hipStream_t st) {
    const size_t stage = (size_t)(BM + a.taps - 1 + BN * a.taps) * (KCT + KPAD) * sizeof(bf16);
    const size_t ctile = (size_t)(BM * (BN + 4) + 2 * BN) * sizeof(float);
    size_t need = stage > ctile ? stage : ctile;
    if (a.e.w2 && need < ctile + (size_t)BM * A2S * sizeof(bf16)) need = ctile + (size_t)BM * A2S * sizeof(bf16);
    if (need > 160 * 1024) return mm_fail(MM_ERR_UNSUPPORTED, "conv1d_fwd: LDS %zu B > 160 KiB", need);
    auto kern = conv1d_fwd_kernel<BM, BN, WM, WN, KCT, FEAT, TAPS>;
    if (need > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)need);
    dim3 grid(a.B * ceil_div(a.T, BM), ceil_div(a.Cout, BN));
    if (a.partial) {
        const int nsplit = ceil_div(a.Cin, a.csplit);
        grid.z = nsplit;
        hipLaunchKernelGGL(kern, grid, dim3(256), need, st, a);
        int rc = mm_check_launch("conv1d_fwd(split-K)");
        if (rc) return rc;
        grid.z = 1;
        hipLaunchKernelGGL((conv1d_splitk_epilogue_kernel<BM, BN>), grid, dim3(256), ctile, st, a, nsplit);
        return mm_check_launch("conv1d_fwd(split-K epilogue)");
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), need, st, a);
    return mm_check_launch("conv1d_fwd");
}

// the combinations of the contrastive training step (logged with MM_EPI_LOG=1), each on the tile shape it runs on;
// everything else takes the generic epilogue
template <int BM, int BN, int WM, int WN, int KCT>
int launch_fwd(const ConvArgs& a, hipStream_t st) {
    const unsigned m = epi_mask(a.e);
    if (getenv("MM_EPI_LOG")) fprintf(stderr, "EPI %d %d %d mask %06x K=%d N=%d taps=%d\n", BM, BN, KCT, m, a.Cin, a.Cout, a.taps);
    if (getenv("MM_EPI_GENERIC") || a.partial) return launch_fwd_feat<BM, BN, WM, WN, KCT, EF_ANY>(a, st);      // tests: generic vs compiled-in epilogues; split-K
#define EPI_CASE(mask) case mask: return launch_fwd_feat<BM, BN, WM, WN, KCT, mask, LT>(a, st);
    if constexpr (BM == 64 && BN == 128 && KCT == 128) {
        constexpr int LT = 1;                 // the Linear layers: one tap
        if (a.taps == 1) switch (m) {
            EPI_CASE(0x001800u)          // QKV projection: bias, bf16 out
            EPI_CASE(0x011904u)          // FFN-1 forward: bias, GELU, dropout, pre-activation copy, bf16 out
            EPI_CASE(0x100908u)          // FFN-2 data gradient: GELU', dropout mask, bf16 out
            default: break;
        }
    } else if constexpr (BM == 32 && BN == 128 && KCT == 128) {
        constexpr int LT = 1;
        if (a.taps == 1) switch (m) {
            EPI_CASE(0x001541u)          // out-proj / FFN-2 forward: bias, dropout, residual, fp32 out, LayerNorm of the result
            EPI_CASE(0x10001541u)        // ... and the next block's QKV projection of those LayerNorm rows (second GEMM)
            EPI_CASE(0x001521u)          // last FFN-2 forward: ... and the mean over tokens instead of the LayerNorm
            EPI_CASE(0x000800u)          // plain data gradient, bf16 out
            EPI_CASE(0x002d01u)          // data gradient + LayerNorm backward: skip gradient in, fp32 and masked bf16 out
            EPI_CASE(0x10002d01u)        // ... and the data gradient of the Linear under that LayerNorm's skip path (second GEMM)
            EPI_CASE(0x002401u)          // the same without the bf16 copy (first block)
            EPI_CASE(0x1006401u)         // ... + the BatchNorm-backward reduce of the conv block below the stack (GELU)
            default: break;
        }
    } else if constexpr (BM == 64 && BN == 64 && KCT == 64) {
        constexpr int LT = 0;                 // k = 3, 5, 7 convolutions: tap count at run time
        switch (m) {
            EPI_CASE(0x001410u)          // conv block forward: bias, BatchNorm sums, fp32 out
            EPI_CASE(0x000800u)          // conv data gradient, bf16 out
            EPI_CASE(0x1004800u)         // ... + the BatchNorm-backward reduce of the layer below (GELU)
            EPI_CASE(0x100c800u)         // ... the same below a MaxPool1d(2)
            default: break;
        }
    }
#undef EPI_CASE
    return launch_fwd_feat<BM, BN, WM, WN, KCT, EF_ANY>(a, st);
}

// ------------------------------------------------------------------ packers
// (B, C, T) fp32  ->  (B, T, Cp) bf16, channels zero-padded to Cp
__global__ void pack_nct_kernel(const float* __restrict__ x, bf16* __restrict__ y, int C, int T, int Cp) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 256 threads: 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, t = t0 + tx;
        tile[i][tx] = (c < C && t < T) ? x[((size_t)b * C + c) * T + t] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int t = t0 + i, c = c0 + tx;
        if (t < T && c < Cp) y[((size_t)b * T + t) * Cp + c] = (bf16)tile[tx][i];
    }
}

// A step's inputs into the static buffers of a captured step, ONE launch: the EEG batch (B, C, T) fp32 is packed straight
// into the channels-last bf16 operand (B, T, Cp) of the first convolution (and, optionally, copied as fp32), the fMRI
// batch is copied.  Workgroups [0, npack) are pack_nct tiles, the rest copy.
__global__ void stage_inputs_kernel(const float* __restrict__ x, bf16* __restrict__ y, float* __restrict__ x_copy, int B, int C,
                                    int T, int Cp, int npack, float4* __restrict__ d1, const float4* __restrict__ s1, size_t n1) {
    __shared__ float tile[32][33];
    if ((int)blockIdx.x >= npack) {
        const size_t nb = gridDim.x - npack;
        for (size_t i = (size_t)(blockIdx.x - npack) * blockDim.x + threadIdx.x; i < n1; i += nb * blockDim.x) d1[i] = s1[i];
        return;
    }
    const int tt = (T + 31) / 32, tc = (Cp + 31) / 32;
    const int b = blockIdx.x / (tt * tc), rem = blockIdx.x % (tt * tc);
    const int t0 = (rem % tt) * 32, c0 = (rem / tt) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, t = t0 + tx;
        const float v = (c < C && t < T) ? x[((size_t)b * C + c) * T + t] : 0.f;
        tile[i][tx] = v;
        if (x_copy && c < C && t < T) x_copy[((size_t)b * C + c) * T + t] = v;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int t = t0 + i, c = c0 + tx;
        if (t < T && c < Cp) y[((size_t)b * T + t) * Cp + c] = (bf16)tile[tx][i];
    }
}

// (B, T, Cp) (bf16 grads) -> (B, C, T) fp32  (input-gradient un-pack)
__global__ void unpack_ntc_kernel(const bf16* __restrict__ g, float* __restrict__ dx, int C, int T, int Cp) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int t = t0 + i, c = c0 + tx;
        tile[i][tx] = (t < T && c < Cp) ? (float)g[((size_t)b * T + t) * Cp + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, t = t0 + tx;
        if (c < C && t < T) dx[((size_t)b * C + c) * T + t] = tile[tx][i];
    }
}

// conv weight (Cout, Cin, k) fp32 -> forward image [Cout][k][Cinp] bf16 and
// data-gradient image [Cinp16][k (flipped)][Coutp] bf16 (Coutp = Cout padded to 16)
__global__ void prep_weight_kernel(const float* __restrict__ w, bf16* __restrict__ wf, bf16* __restrict__ wd,
                                   int Cout, int Cin, int k, int Cinp, int Coutp) {
    const int total_f = Cout * k * Cinp;
    const int CinRows = Cinp;
    const int total_d = wd ? CinRows * k * Coutp : 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total_f + total_d; i += gridDim.x * blockDim.x) {
        if (i < total_f) {
            const int c = i % Cinp, tap = (i / Cinp) % k, n = i / (Cinp * k);
            wf[conv_image_index(Cout, k, Cinp, n, tap, c)] = (bf16)(c < Cin ? w[((size_t)n * Cin + c) * k + tap] : 0.f);
        } else {
            const int d = i - total_f;
            const int n = d % Coutp, tap = (d / Coutp) % k, c = d / (Coutp * k);
            const float v = (c < Cin && n < Cout) ? w[((size_t)n * Cin + c) * k + (k - 1 - tap)] : 0.f;
            wd[conv_image_index(CinRows, k, Coutp, c, tap, n)] = (bf16)v;
        }
    }
}


// every weight image of a model in one launch (blockIdx.y = tensor): a training step repacks ~40
// small tensors after each optimizer update, and as separate ~5 us nodes they sat on the critical
// path of the step's graph
struct PrepDesc { const float* w; bf16* wf; bf16* wd; int Cout, Cin, k, Cinp, Coutp, pad_; };
constexpr int PM_MAX = 64;
// first[t] = first workgroup of tensor t: workgroups are dealt out in proportion to the elements (PM_EPB per
// workgroup).  128 workgroups per tensor left the step's two largest images (13 elements per thread, gathered with
// a stride of k floats) as a 12 us tail on the chain while the small ones idled.
constexpr int PM_EPB = 1024;
struct PrepTable { PrepDesc d[PM_MAX]; int first[PM_MAX + 1]; };          // by value, as ReduceTable
__global__ void prep_many_kernel(PrepTable tab, int ndesc, float4* __restrict__ zero, long nzero4) {
    if ((int)blockIdx.x >= tab.first[ndesc]) {
        // the step's accumulator arena is zeroed by the same launch (a fill node of its own cost ~5 us on the chain)
        const long b = blockIdx.x - tab.first[ndesc], nb = gridDim.x - tab.first[ndesc];
        for (long i = b * blockDim.x + threadIdx.x; i < nzero4; i += nb * blockDim.x) zero[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    int t = 0;
    while (t + 1 < ndesc && (int)blockIdx.x >= tab.first[t + 1]) ++t;      // (uniform: <= 64 scalar compares)
    const PrepDesc d = tab.d[t];
    const int blk = blockIdx.x - tab.first[t], nblk = tab.first[t + 1] - tab.first[t];
    // one item = 8 consecutive elements of an image's innermost index (c of the forward image, n of the data-gradient
    // image; both padded widths are multiples of 16 and both layouts keep an aligned group of 8 contiguous): two integer
    // divisions and one 16-byte store per 8 elements (three divisions and a 2-byte store per ELEMENT made this launch -
    // the first of the step, in front of both streams - VALU-bound at 10 us)
    const int cg = d.Cinp / 8, ng = d.wd ? d.Coutp / 8 : 0;
    const int items_f = d.Cout * d.k * cg, items_d = d.wd ? d.Cinp * d.k * ng : 0;
    for (int i = blk * blockDim.x + threadIdx.x; i < items_f + items_d; i += nblk * blockDim.x) {
        bf16x8 v;
        if (i < items_f) {
            const int c0 = (i % cg) * 8, r = i / cg, tap = r % d.k, n = r / d.k;
            const float* src = d.w + ((size_t)n * d.Cin + c0) * d.k + tap;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (bf16)(c0 + j < d.Cin ? src[(size_t)j * d.k] : 0.f);
            *reinterpret_cast<bf16x8*>(d.wf + conv_image_index(d.Cout, d.k, d.Cinp, n, tap, c0)) = v;
        } else {
            const int e = i - items_f;
            const int n0 = (e % ng) * 8, r = e / ng, tap = r % d.k, c = r / d.k;
            const float* src = d.w + ((size_t)n0 * d.Cin + c) * d.k + (d.k - 1 - tap);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (bf16)((c < d.Cin && n0 + j < d.Cout) ? src[(size_t)j * d.Cin * d.k] : 0.f);
            *reinterpret_cast<bf16x8*>(d.wd + conv_image_index(d.Cinp, d.k, d.Coutp, c, tap, n0)) = v;
        }
    }
}

// ---------------------------------------------------------------------------
// weight gradient:  dW[n][tap][c] += sum_{t in chunk} dY[b,t,n] * X[b,t+tap-pad,c]
// MFMA view: D[i=n][j=c] = sum_k A[i][k] B[k][j] with k = t, so both operands
// are k-strided in memory.  The dY tile [64 t][64 n] and the X halo tile
// [64+taps-1][64 c] are staged row-major and read with ds_read_b64_tr_b16
// (hardware transpose): each 16-lane group fetches a 4(t) x 16(col) block and
// every lane receives its column's 4 consecutive t values.  Row stride 192 B
// (== 192 mod 256) puts the 4 rows x 64 B a half-wave touches on 64 distinct
// banks.  Partial sums leave the workgroup as fp32 atomics straight into the
// parameter-gradient tensor (arbitrary element strides sn/sc/stap).
// ---------------------------------------------------------------------------
constexpr int WG_MK = 64;          // t rows per LDS tile
constexpr int WG_LD = 96;          // LDS row stride in elements (192 B)

__device__ __forceinline__ bf16x8 tr_frag(const bf16* tile, int row0, int col0, int lane) {
    // rows row0 + 8*(lane>>5) + {0..7}, column col0 + (lane & 31)
    const int li = lane & 15, g = lane >> 4;
    const bf16* p = tile + (row0 + 8 * (g >> 1) + (li >> 2)) * WG_LD + col0 + (g & 1) * 16 + 4 * (li & 3);
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * WG_LD));
    union { s16x4 s[2]; bf16x8 v; } u;
    u.s[0] = lo; u.s[1] = hi;
    return u.v;
}

struct WgradArgs {
    const bf16* dy; const bf16* x; float* dw; float* dbias;
    int B, T, Cin, Cout, pad, Cin_real, rows_per_wg, nrep;
    long sn, sc, stap, rep_stride;
    int slot_mode;            // 1: workgroup x stores its partial tile into slot blockIdx.x (no atomics)
    int bgroup;               // samples one workgroup accumulates over (> 1 only when a sample is a single row chunk)
};

template <int TAPS>
__device__ __forceinline__ void conv1d_wgrad_body(const WgradArgs& a, const int bx, const int by, const int bz) {
    __shared__ __attribute__((aligned(16))) bf16 Ys[WG_MK * WG_LD];
    __shared__ __attribute__((aligned(16))) bf16 Xs[(WG_MK + TAPS - 1) * WG_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wc = wave & 1;
    const int chunksT = (a.T + a.rows_per_wg - 1) / a.rows_per_wg;
    // samples [b0, b1) of this workgroup: one, or - short sequences with many output tiles (config #5: 33 frames x 6 272
    // channels) - a group of them, so that the number of SLOTS (each a full weight-shaped fp32 image that the flush has to
    // sum: 33.7 MB there) does not grow with the batch
    const int b0 = (bx / chunksT) * a.bgroup, b1 = min(a.B, b0 + a.bgroup);
    const int tbeg = (bx % chunksT) * a.rows_per_wg;
    const int tend = min(a.T, tbeg + a.rows_per_wg);
    const int n0 = by * 64, c0 = bz * 64;

    f32x16 acc[TAPS];
#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tp][r] = 0.f;
    float bsum = 0.f;

    // tiles are fetched into registers one work item (sample, row tile) ahead (all loads in flight
    // before any LDS write, and in flight during the previous tile's MFMAs)
    constexpr int YREG = WG_MK * 8 / 256;                           // 2
    constexpr int XREG = ((WG_MK + TAPS - 1) * 8 + 255) / 256;      // 3
    uint4 yv[YREG], xv[XREG];
    auto fetch = [&](int b, int t0) {
        const bf16* dyb = a.dy + (size_t)b * a.T * a.Cout;
        const bf16* xb = a.x + (size_t)b * a.T * a.Cin;
#pragma unroll
        for (int i = 0; i < YREG; ++i) {
            const int s = tid + i * 256, r = s >> 3, sg = s & 7;
            const int t = t0 + r, n = n0 + sg * 8;
            yv[i] = (t < tend && n < a.Cout) ? *reinterpret_cast<const uint4*>(dyb + (size_t)t * a.Cout + n) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < XREG; ++i) {
            const int s = tid + i * 256, r = s >> 3, sg = s & 7;
            const int t = t0 - a.pad + r, c = c0 + sg * 8;
            xv[i] = (s < (WG_MK + TAPS - 1) * 8 && t >= 0 && t < a.T && c < a.Cin)
                        ? *reinterpret_cast<const uint4*>(xb + (size_t)t * a.Cin + c) : make_uint4(0, 0, 0, 0);
        }
    };
    int b = b0, t0 = tbeg;
    if (b < b1 && tbeg < tend) fetch(b, t0);
    while (b < b1 && tbeg < tend) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < YREG; ++i) {
            const int s = tid + i * 256;
            *reinterpret_cast<uint4*>(Ys + (s >> 3) * WG_LD + (s & 7) * 8) = yv[i];
        }
#pragma unroll
        for (int i = 0; i < XREG; ++i) {
            const int s = tid + i * 256;
            if (s < (WG_MK + TAPS - 1) * 8) *reinterpret_cast<uint4*>(Xs + (s >> 3) * WG_LD + (s & 7) * 8) = xv[i];
        }
        __syncthreads();
        int nb = b, nt = t0 + WG_MK;                                // next work item
        if (nt >= tend) { ++nb; nt = tbeg; }
        if (nb < b1) fetch(nb, nt);
#pragma unroll
        for (int kk = 0; kk < WG_MK; kk += 16) {
            const bf16x8 af = tr_frag(Ys, kk, wn * 32, lane);
            if (a.dbias && bz == 0 && wc == 0)
#pragma unroll
                for (int j = 0; j < 8; ++j) bsum += (float)af[j];
#pragma unroll
            for (int tp = 0; tp < TAPS; ++tp) {
                const bf16x8 bfr = tr_frag(Xs, kk + tp, wc * 32, lane);
                acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[tp], 0, 0, 0);
            }
        }
        b = nb; t0 = nt;
    }
    // D[i = n][j = c]: lane owns column c, rows n = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int c = c0 + wc * 32 + (lane & 31);
    float* dwr = a.dw + (size_t)bx * a.rep_stride;
    if (c < a.Cin_real) {
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (n < a.Cout) dwr[n * a.sn + c * a.sc + tp * a.stap] = acc[tp][r];   // this (slot, tile) element has one writer
            }
    }
    if (a.dbias && bz == 0 && wc == 0) {
        bsum += __shfl_xor(bsum, 32, 64);
        const int n = n0 + wn * 32 + (lane & 31);
        if ((lane >> 5) == 0 && n < a.Cout) acc_add<MM_ACC_GRAD>(acc_rep(a.dbias, bx % MM_ACC_REPL, a.Cout) + n, bsum);
    }
}

template <int TAPS>
__global__ __launch_bounds__(256) void conv1d_wgrad_kernel(WgradArgs a) {
    conv1d_wgrad_body<TAPS>(a, blockIdx.x, blockIdx.y, blockIdx.z);
}

// several independent Linear (taps = 1) weight gradients in ONE launch: workgroup id -> (problem,
// its own 3-D block index).  The transformer blocks' eight weight-gradient GEMMs have nothing waiting
// on them but the final slot sum, so a trainer collects them and issues them once, off the chain.
constexpr int WM_MAX = 12;
struct WgradTable { WgradArgs a[WM_MAX]; int first[WM_MAX + 1]; int gx[WM_MAX], gy[WM_MAX]; int n; };
// Linear (taps = 1) weight gradient on a 128 (n) x 128 (c) workgroup tile: wave (wn, wc) owns 64 x 64 = 2 x 2
// MFMA tiles, so a k-step is 4 transposed LDS fragment reads for 4 MFMAs (the 64 x 64 tile: 2 for 1) and
// the operands are fetched from global memory half as often.  Each operand tile lives in LDS as two
// 64-column halves with the 192-byte row stride tr_frag is laid out for.  Slot mode only.
__device__ __forceinline__ void linear_wgrad128_body(const WgradArgs& a, const int bx, const int by, const int bz) {
    __shared__ __attribute__((aligned(16))) bf16 Ys[2][WG_MK * WG_LD];
    __shared__ __attribute__((aligned(16))) bf16 Xs[2][WG_MK * WG_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wc = wave & 1;
    const int chunksT = (a.T + a.rows_per_wg - 1) / a.rows_per_wg;
    const int b = bx / chunksT;
    const int tbeg = (bx % chunksT) * a.rows_per_wg;
    const int tend = min(a.T, tbeg + a.rows_per_wg);
    const int n0 = by * 128, c0 = bz * 128;
    const bf16* dyb = a.dy + (size_t)b * a.T * a.Cout;
    const bf16* xb = a.x + (size_t)b * a.T * a.Cin;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float bsum[2] = {0.f, 0.f};

    constexpr int NREG = WG_MK * 16 / 256;                          // 4 x 16-byte chunks per operand per thread
    uint4 yv[NREG], xv[NREG];
    auto fetch = [&](int t0) {
#pragma unroll
        for (int i = 0; i < NREG; ++i) {
            const int s = tid + i * 256, r = s >> 4, sg = s & 15;
            const int t = t0 + r, n = n0 + sg * 8, c = c0 + sg * 8;
            yv[i] = (t < tend && n < a.Cout) ? *reinterpret_cast<const uint4*>(dyb + (size_t)t * a.Cout + n) : make_uint4(0, 0, 0, 0);
            xv[i] = (t < tend && c < a.Cin) ? *reinterpret_cast<const uint4*>(xb + (size_t)t * a.Cin + c) : make_uint4(0, 0, 0, 0);
        }
    };
    if (tbeg < tend) fetch(tbeg);
    for (int t0 = tbeg; t0 < tend; t0 += WG_MK) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NREG; ++i) {
            const int s = tid + i * 256, r = s >> 4, sg = s & 15;
            *reinterpret_cast<uint4*>(Ys[sg >> 3] + r * WG_LD + (sg & 7) * 8) = yv[i];
            *reinterpret_cast<uint4*>(Xs[sg >> 3] + r * WG_LD + (sg & 7) * 8) = xv[i];
        }
        __syncthreads();
        if (t0 + WG_MK < tend) fetch(t0 + WG_MK);
#pragma unroll
        for (int kk = 0; kk < WG_MK; kk += 16) {
            bf16x8 af[2], bfr[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = tr_frag(Ys[wn], kk, i * 32, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) bfr[j] = tr_frag(Xs[wc], kk, j * 32, lane);
            if (a.dbias && bz == 0 && wc == 0)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) bsum[i] += (float)af[i][j];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    }
    // D[i = n][j = c]: lane owns column c, rows n = (r&3) + 8*(r>>2) + 4*(lane>>5); every (slot, element) has one writer
    float* dwr = a.dw + (size_t)bx * a.rep_stride;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = c0 + wc * 64 + j * 32 + (lane & 31);
        if (c >= a.Cin_real) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wn * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (n < a.Cout) dwr[n * a.sn + c * a.sc] = acc[i][j][r];
            }
    }
    if (a.dbias && bz == 0 && wc == 0)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float v = bsum[i] + __shfl_xor(bsum[i], 32, 64);
            const int n = n0 + wn * 64 + i * 32 + (lane & 31);
            if ((lane >> 5) == 0 && n < a.Cout) acc_add<MM_ACC_GRAD>(acc_rep(a.dbias, bx % MM_ACC_REPL, a.Cout) + n, v);
        }
}

__global__ __launch_bounds__(256) void conv1d_wgrad_many_kernel(WgradTable tab) {
    int p = 0;
    while (p + 1 < tab.n && (int)blockIdx.x >= tab.first[p + 1]) ++p;
    const int local = blockIdx.x - tab.first[p];
    const int gx = tab.gx[p], gy = tab.gy[p];
    linear_wgrad128_body(tab.a[p], local % gx, (local / gx) % gy, local / (gx * gy));
}

// dw[n][c][tap] += sum_rep ws[rep][n][tap][c]   (replicated contiguous-atomics workspace -> PyTorch layout)
__global__ void wgrad_scatter_kernel(const float* __restrict__ ws, float* __restrict__ dw, int Cout, int Cin, int taps,
                                     int Cinp, int nrep) {
    const size_t rstride = (size_t)Cout * taps * Cinp;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < rstride; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cinp);
        if (c >= Cin) continue;
        const int tap = (int)((i / Cinp) % taps);
        const int n = (int)(i / ((size_t)Cinp * taps));
        float s = 0.f;
        int r = 0;
        for (; r + 8 <= nrep; r += 8) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = ws[(r + q) * rstride + i];      // coalesced along c
#pragma unroll
            for (int q = 0; q < 8; ++q) s += v[q];
        }
        for (; r < nrep; ++r) s += ws[r * rstride + i];
        dw[((size_t)n * Cin + c) * taps + tap] += s;
    }
}

// every conv weight-gradient workspace of a backward pass in one launch (blockIdx.y = tensor)
// taps: bits 0-7 the taps of the gradient tensor, bits 8-15 the first workspace tap it takes, bits 16-23 the taps of the
// workspace rows (0 = the same): a descriptor may take a WINDOW of the workspace's taps (the k = 3 / 5 branches of
// EnhancedPowerEncoder's merged k = 7 convolution: their gradients are the centre taps of their 64 output channels)
struct ScatterDesc { const float* ws; float* dw; int Cout, Cin, taps, Cinp, nrep, cout_all; };
__device__ __host__ inline int scatter_taps(const ScatterDesc& d) { return d.taps & 255; }
__device__ __host__ inline int scatter_tap0(const ScatterDesc& d) { return (d.taps >> 8) & 255; }
__device__ __host__ inline int scatter_ws_taps(const ScatterDesc& d) { return (d.taps >> 16) & 255 ? (d.taps >> 16) & 255 : (d.taps & 255); }
static bool scatter_desc_ok(const ScatterDesc& d) {
    return d.ws && d.dw && d.Cout > 0 && d.Cin > 0 && scatter_taps(d) > 0 && d.Cinp >= d.Cin && d.nrep >= 1 &&
           scatter_tap0(d) + scatter_taps(d) <= scatter_ws_taps(d) && (d.cout_all == 0 || d.cout_all >= d.Cout);
}
// cout_all: 0, or the output channels of the WHOLE workspace when the descriptor covers a slice of them (replica stride)
constexpr int SM_MAX = 64;
struct ScatterTable { ScatterDesc d[SM_MAX]; };
// wide layers (Cin >= 256: config #5's merged convolution has 6 272 input channels, 8.4 M weights): the strided
// read-modify-write of the plain form below ran at 0.6 TB/s (168 us).  Here a workgroup takes one output channel x 256
// input channels, reads the workspace rows of every tap coalesced, turns the [tap][c] block into [c][tap] through LDS and
// adds it to a CONTIGUOUS range of the gradient.  Same replica order as the plain form: same bits.
__device__ __forceinline__ void scatter_body_tiled(const ScatterDesc& d, int blk, int nblk) {
    __shared__ float tile[256 * 9];
    const int taps = scatter_taps(d), tap0 = scatter_tap0(d), tws = scatter_ws_taps(d);
    const size_t rstride = (size_t)(d.cout_all ? d.cout_all : d.Cout) * tws * d.Cinp;
    const int cch = (d.Cin + 255) / 256, items = d.Cout * cch, ts = taps | 1, tid = threadIdx.x;
    for (int item = blk; item < items; item += nblk) {
        const int n = item / cch, c0 = (item - n * cch) * 256;
        const int cn = min(256, d.Cin - c0);
        if (tid < cn)
            for (int tap = 0; tap < taps; ++tap) {
                const float* src = d.ws + ((size_t)n * tws + tap0 + tap) * d.Cinp + c0 + tid;
                float s = 0.f;
                int r = 0;
                for (; r + 8 <= d.nrep; r += 8) {
                    float v[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = src[(r + q) * rstride];
#pragma unroll
                    for (int q = 0; q < 8; ++q) s += v[q];
                }
                for (; r < d.nrep; ++r) s += src[r * rstride];
                tile[tid * ts + tap] = s;
            }
        __syncthreads();
        float* dst = d.dw + ((size_t)n * d.Cin + c0) * taps;
        for (int j = tid; j < cn * taps; j += 256) {
            const int cl = j / taps;
            dst[j] += tile[cl * ts + (j - cl * taps)];
        }
        __syncthreads();
    }
}

__device__ __forceinline__ void scatter_body(const ScatterDesc& d, int blk, int nblk) {
    const int taps = scatter_taps(d), tap0 = scatter_tap0(d), tws = scatter_ws_taps(d);
    if (taps > 1 && taps <= 8 && d.Cin >= 256) return scatter_body_tiled(d, blk, nblk);      // (uniform per descriptor)
    // walk the workspace in ITS order (channel-contiguous: the nrep replica reads coalesce) and
    // scatter one strided write per element, not nrep strided reads
    const size_t rstride = (size_t)(d.cout_all ? d.cout_all : d.Cout) * tws * d.Cinp;
    const size_t count = (size_t)d.Cout * taps * d.Cinp;
    for (size_t i = (size_t)blk * blockDim.x + threadIdx.x; i < count; i += (size_t)nblk * blockDim.x) {
        const int c = (int)(i % d.Cinp);
        if (c >= d.Cin) continue;
        const int tap = (int)((i / d.Cinp) % taps);
        const int n = (int)(i / ((size_t)d.Cinp * taps));
        const size_t e = ((size_t)n * tws + tap0 + tap) * d.Cinp + c;      // (= i for a whole-kernel descriptor)
        float s = 0.f;
        int r = 0;
        for (; r + 8 <= d.nrep; r += 8) {                   // eight independent loads at a time, not a latency chain
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = d.ws[(r + q) * rstride + e];
#pragma unroll
            for (int q = 0; q < 8; ++q) s += v[q];
        }
        for (; r < d.nrep; ++r) s += d.ws[r * rstride + e];
        d.dw[((size_t)n * d.Cin + c) * taps + tap] += s;
    }
}
__global__ void scatter_many_kernel(ScatterTable tab) { scatter_body(tab.d[blockIdx.y], blockIdx.x, gridDim.x); }

// dst[k] += sum_rep src[rep][k]
// one replica per lane (32 lanes per output), one shuffle reduction: a single
// load round trip instead of a 32-deep dependent chain
__global__ void reduce_replicas_kernel(const float* __restrict__ src, float* __restrict__ dst, int K, int nrep,
                                       long rep_stride) {
    const int k = blockIdx.x * 8 + (threadIdx.x >> 5);
    const int r0 = threadIdx.x & 31;
    float s = 0.f;
    if (k < K)
        for (int r = r0; r < nrep; r += 32) s += src[(size_t)r * rep_stride + k];
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (k < K && r0 == 0) dst[k] += s;
}

// dst[k] += 2^-MM_ACC_GRAD * sum_rep acc[rep][k]: fixed-point accumulator workspace (common.h) -> fp32.
// One replica per lane (16 lanes per output), integer shuffle reduction.
__device__ __forceinline__ void acc_reduce_rows(const mm_acc_t* __restrict__ src, float* __restrict__ dst, long K,
                                                long rep_stride, long kfirst, long kstep) {
    const int r0 = threadIdx.x & 15;
    for (long k = kfirst + (threadIdx.x >> 4); k < ((K + 15) / 16) * 16; k += kstep) {
        mm_acc_t s = k < K ? src[r0 * rep_stride + k] : 0;
        s = acc_sum_lanes16(s);
        if (k < K && r0 == 0) dst[k] += acc_val<MM_ACC_GRAD>(s);
    }
}
__global__ void acc_reduce_kernel(const mm_acc_t* __restrict__ src, float* __restrict__ dst, int K, long rep_stride) {
    acc_reduce_rows(src, dst, K, rep_stride, (long)blockIdx.x * 16, (long)gridDim.x * 16);
}

template <int TAPS>
int launch_wgrad(const WgradArgs& a, hipStream_t st) {
    const int chunksT = ceil_div(a.T, a.rows_per_wg);
    dim3 grid(ceil_div(a.B, a.bgroup) * chunksT, ceil_div(a.Cout, 64), ceil_div(a.Cin, 64));
    hipLaunchKernelGGL(conv1d_wgrad_kernel<TAPS>, grid, dim3(256), 0, st, a);
    return mm_check_launch("conv1d_wgrad");
}

}  // namespace

// ============================================================================
// C ABI (declared in include/mmeeg_hip.h)
// ============================================================================
extern "C" {

int mm_pack_nct_bf16(const float* x, void* y, int B, int C, int T, int Cp, hipStream_t st) {
    MM_REQUIRE(x && y && B > 0 && C > 0 && T > 0 && Cp >= C && Cp % 16 == 0, "pack_nct: bad args");
    dim3 grid(ceil_div(T, 32), ceil_div(Cp, 32), B);
    hipLaunchKernelGGL(pack_nct_kernel, grid, dim3(256), 0, st, x, (bf16*)y, C, T, Cp);
    return mm_check_launch("pack_nct");
}

int mm_stage_inputs(const float* eeg, void* eeg_packed_bf16, float* eeg_copy, int B, int C, int T, int Cp, float* fmri_dst,
                    const float* fmri_src, int64_t fmri_n, hipStream_t st) {
    MM_REQUIRE(eeg && eeg_packed_bf16 && B > 0 && C > 0 && T > 0 && Cp >= C && Cp % 16 == 0, "stage_inputs: bad EEG args");
    MM_REQUIRE(fmri_dst && fmri_src && fmri_n > 0 && fmri_n % 4 == 0 && (((uintptr_t)fmri_dst | (uintptr_t)fmri_src) & 15) == 0,
               "stage_inputs: fMRI copy needs 16-byte alignment and a multiple of 4 floats");
    const int npack = B * ceil_div(T, 32) * ceil_div(Cp, 32);
    const long n4 = fmri_n / 4;
    const int ncopy = (int)((n4 + 1023) / 1024 < 1024 ? (n4 + 1023) / 1024 : 1024);
    hipLaunchKernelGGL(stage_inputs_kernel, dim3(npack + ncopy), dim3(256), 0, st, eeg, (bf16*)eeg_packed_bf16, eeg_copy, B, C, T, Cp,
                       npack, reinterpret_cast<float4*>(fmri_dst), reinterpret_cast<const float4*>(fmri_src), (size_t)n4);
    return mm_check_launch("stage_inputs");
}

int mm_unpack_ntc_f32(const void* g, float* dx, int B, int C, int T, int Cp, hipStream_t st) {
    MM_REQUIRE(g && dx && B > 0 && C > 0 && T > 0 && Cp >= C, "unpack_ntc: bad args");
    dim3 grid(ceil_div(T, 32), ceil_div(Cp, 32), B);
    hipLaunchKernelGGL(unpack_ntc_kernel, grid, dim3(256), 0, st, (const bf16*)g, dx, C, T, Cp);
    return mm_check_launch("unpack_ntc");
}

int mm_prep_conv_weight(const float* w, void* w_fwd, void* w_dgrad, int Cout, int Cin, int k,
                        int Cinp, int Coutp, hipStream_t st) {
    MM_REQUIRE(w && w_fwd && Cinp % 16 == 0 && Cinp >= Cin && (!w_dgrad || (Coutp % 16 == 0 && Coutp >= Cout)),
               "prep_conv_weight: bad args");
    const int total = Cout * k * Cinp + (w_dgrad ? Cinp * k * Coutp : 0);
    hipLaunchKernelGGL(prep_weight_kernel, dim3(ceil_div(total, 256) < 1024 ? ceil_div(total, 256) : 1024), dim3(256),
                       0, st, w, (bf16*)w_fwd, (bf16*)w_dgrad, Cout, Cin, k, Cinp, Coutp);
    return mm_check_launch("prep_conv_weight");
}

int mm_prep_many_zero(const void* desc_host, int ndesc, float* zero, int64_t nzero, hipStream_t st) {
    MM_REQUIRE(desc_host && ndesc > 0, "prep_many: bad args");
    MM_REQUIRE(nzero >= 0 && (zero || !nzero) && nzero % 4 == 0 && ((uintptr_t)zero & 15) == 0,
               "prep_many_zero: the zeroed range must be 16-byte aligned and a multiple of 4 floats");
    const PrepDesc* src = (const PrepDesc*)desc_host;
    for (int base = 0; base < ndesc; base += PM_MAX) {
        PrepTable tab;
        const int n = ndesc - base < PM_MAX ? ndesc - base : PM_MAX;
        for (int i = 0; i < n; ++i) {
            const PrepDesc& d = src[base + i];
            MM_REQUIRE(d.w && d.wf && d.Cinp % 16 == 0 && d.Cinp >= d.Cin && d.Cout > 0 && d.k > 0 &&
                       (!d.wd || (d.Coutp % 16 == 0 && d.Coutp >= d.Cout)), "prep_many: descriptor %d", base + i);
            tab.d[i] = d;
        }
        static_assert(sizeof(PrepTable) + 8 <= 4096, "kernel arguments");
        int nblocks = 0;
        for (int i = 0; i < n; ++i) {
            const PrepDesc& d = tab.d[i];
            const long total = (long)d.Cout * d.k * d.Cinp + (d.wd ? (long)d.Cinp * d.k * d.Coutp : 0);
            MM_REQUIRE(total < (1l << 31), "prep_many: descriptor %d too large", base + i);
            tab.first[i] = nblocks;
            nblocks += (int)((total + PM_EPB - 1) / PM_EPB);
        }
        tab.first[n] = nblocks;
        const bool last = base + PM_MAX >= ndesc;                     // the fill rides in the last launch
        const long nz4 = last ? nzero / 4 : 0;
        const int zblocks = (int)((nz4 + 2047) / 2048 < 1024 ? (nz4 + 2047) / 2048 : 1024);
        hipLaunchKernelGGL(prep_many_kernel, dim3(nblocks + zblocks), dim3(256), 0, st, tab, n, reinterpret_cast<float4*>(zero), nz4);
    }
    return mm_check_launch("prep_many");
}

int mm_prep_many(const void* desc_host, int ndesc, hipStream_t st) { return mm_prep_many_zero(desc_host, ndesc, nullptr, 0, st); }

static int conv1d_dispatch(const ConvArgs& a, hipStream_t st) {
    // tile / chunk choice: full-K staging for linears (taps == 1), 64-wide chunks
    // for the k>1 convs with BN = 64 so that two workgroups fit one CU's LDS
    const int taps = a.taps, Cin = a.Cin, Cout = a.Cout, B = a.B, T = a.T;
    const int kct = (taps == 1 && Cin % 128 == 0) ? 128 : (Cin % 64 == 0 ? 64 : (Cin % 32 == 0 ? 32 : 16));
    const bool narrow = Cout <= 64 || taps > 1;
#define MM_FWD(BM_, BN_, WM_, WN_)                                               \
    switch (kct) {                                                               \
        case 16: return launch_fwd<BM_, BN_, WM_, WN_, 16>(a, st);               \
        case 32: return launch_fwd<BM_, BN_, WM_, WN_, 32>(a, st);               \
        case 64: return launch_fwd<BM_, BN_, WM_, WN_, 64>(a, st);               \
        default: return launch_fwd<BM_, BN_, WM_, WN_, 128>(a, st);              \
    }
    if (narrow) { MM_FWD(64, 64, 2, 2) }
    // few row tiles (M <= 16k): halve BM so that >= 2 workgroups share a CU and overlap
    if ((long)B * ceil_div(T, 64) * ceil_div(Cout, 128) <= 512) { MM_FWD(32, 128, 1, 4) }
    MM_FWD(64, 128, 2, 2)
#undef MM_FWD
}

// split-K plan of a forward launch: only the k > 1 convolutions on the 64 x 64 x 64 tile, when the output tiles do not
// fill the chip and the reduction is long.  -> number of channel slices (1 = run it whole)
static int conv1d_splitk_slices(int B, int T, int Cin, int Cout, int taps) {
    if (taps == 1 || Cin % 64) return 1;
    const long tiles = (long)B * ceil_div(T, 64) * ceil_div(Cout, 64);
    const int chunks = Cin / 64;
    if (tiles >= 192 || chunks < 16) return 1;
    long n = 512 / tiles;                       // ~2 workgroups per CU
    if (n > chunks / 8) n = chunks / 8;         // >= 8 chunks per slice
    if (n > 8) n = 8;
    return n < 2 ? 1 : (int)n;
}

int mm_conv1d_fwd_splitk_plan(int B, int T, int Cin, int Cout, int taps, int* nsplit_host, int64_t* ws_floats_host, hipStream_t) {
    MM_REQUIRE(nsplit_host && ws_floats_host && B > 0 && T > 0 && Cin > 0 && Cout > 0 && taps >= 1, "conv1d_fwd_splitk_plan: bad args");
    const int n = conv1d_splitk_slices(B, T, Cin, Cout, taps);
    *nsplit_host = n;
    *ws_floats_host = n > 1 ? (int64_t)n * B * T * Cout : 0;
    return 0;
}

static int conv1d_fwd_args(ConvArgs& a, const void* x, const void* w, int B, int T, int Cin, int Cout, int taps, int pad,
                           const float* scale, const float* shift, int act, const float* residual, const float* pe,
                           int pool, float* stats, float* out_f32, void* out_bf16, void* out_pre,
                           float drop_p, uint32_t drop_seed, const uint32_t* seed_epoch, const void* gradz, int gradz_act);

int mm_conv1d_fwd_splitk(const void* x, const void* w, int B, int T, int Cin, int Cout, int taps, int pad,
                         const float* scale, const float* shift, int act, const float* residual, const float* pe,
                         int pool, float* stats, float* out_f32, void* out_bf16, void* out_pre,
                         float drop_p, uint32_t drop_seed, const uint32_t* seed_epoch, const void* gradz, int gradz_act,
                         float* ws, int nsplit, hipStream_t st) {
    ConvArgs a;
    int rc = conv1d_fwd_args(a, x, w, B, T, Cin, Cout, taps, pad, scale, shift, act, residual, pe, pool, stats, out_f32, out_bf16,
                             out_pre, drop_p, drop_seed, seed_epoch, gradz, gradz_act);
    if (rc) return rc;
    MM_REQUIRE(ws && nsplit >= 2 && nsplit <= 64, "conv1d_fwd_splitk: workspace / nsplit=%d", nsplit);
    MM_REQUIRE(taps > 1 && Cin % 64 == 0, "conv1d_fwd_splitk: k > 1 convolutions with Cin %% 64 == 0 only (taps=%d Cin=%d)", taps, Cin);
    const int chunks = Cin / 64;
    a.csplit = ceil_div(chunks, nsplit) * 64;
    MM_REQUIRE(ceil_div(Cin, a.csplit) >= 2, "conv1d_fwd_splitk: nsplit=%d leaves one slice", nsplit);
    a.partial = ws;                               // ceil(Cin / csplit) <= nsplit slices of B * T * Cout floats
    return conv1d_dispatch(a, st);
}

// Generic forward implicit GEMM.  See include/mmeeg_hip.h for the contract.
int mm_conv1d_fwd(const void* x, const void* w, int B, int T, int Cin, int Cout, int taps, int pad,
                  const float* scale, const float* shift, int act, const float* residual, const float* pe,
                  int pool, float* stats, float* out_f32, void* out_bf16, void* out_pre,
                  float drop_p, uint32_t drop_seed, const uint32_t* seed_epoch, const void* gradz, int gradz_act,
                  hipStream_t st) {
    ConvArgs a;
    int rc = conv1d_fwd_args(a, x, w, B, T, Cin, Cout, taps, pad, scale, shift, act, residual, pe, pool, stats, out_f32, out_bf16,
                             out_pre, drop_p, drop_seed, seed_epoch, gradz, gradz_act);
    return rc ? rc : conv1d_dispatch(a, st);
}

static int conv1d_fwd_args(ConvArgs& a, const void* x, const void* w, int B, int T, int Cin, int Cout, int taps, int pad,
                           const float* scale, const float* shift, int act, const float* residual, const float* pe,
                           int pool, float* stats, float* out_f32, void* out_bf16, void* out_pre,
                           float drop_p, uint32_t drop_seed, const uint32_t* seed_epoch, const void* gradz, int gradz_act) {
    MM_REQUIRE(x && w, "conv1d_fwd: null operand");
    MM_REQUIRE(B > 0 && T > 0 && Cout > 0 && taps >= 1 && taps <= 9 && pad >= 0 && pad < taps, "conv1d_fwd: bad dims");
    MM_REQUIRE(Cin > 0 && Cin % 16 == 0, "conv1d_fwd: Cin=%d must be a multiple of 16", Cin);
    MM_REQUIRE(pool == 1 || (pool == 2 && T % 2 == 0), "conv1d_fwd: pool=%d T=%d", pool, T);
    MM_REQUIRE(out_f32 || out_bf16 || out_pre, "conv1d_fwd: no output");
    MM_REQUIRE(Cout % 4 == 0, "conv1d_fwd: Cout=%d must be a multiple of 4", Cout);
    MM_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "conv1d_fwd: drop_p");
    a.x = (const bf16*)x; a.w = (const bf16*)w;
    a.B = B; a.T = T; a.Cin = Cin; a.Cout = Cout; a.taps = taps; a.pad = pad;
    a.e.scale = scale; a.e.shift = shift; a.e.residual = residual; a.e.pe = pe; a.e.stats = stats;
    a.e.out_f32 = out_f32; a.e.out_bf16 = (bf16*)out_bf16; a.e.out_pre = (bf16*)out_pre;
    a.e.act = act; a.e.pool = pool;
    a.e.drop_thresh = drop_p > 0.f ? (uint32_t)((double)drop_p * 4294967296.0) : 0u;
    a.e.drop_seed = drop_seed;
    a.e.drop_epoch = seed_epoch;
    a.e.gradz = (const bf16*)gradz; a.e.gradz_act = gradz_act;
    a.e.drop_inv_keep = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.f;
    a.e.ln_x = nullptr; a.e.ln_stat = nullptr; a.e.ln_gamma = nullptr; a.e.ln_dgb = nullptr;
    a.e.pool_out = nullptr; a.e.pool_rows = 0; a.e.pool_scale = 0.f;
    a.e.lnf_out = nullptr; a.e.lnf_stat = nullptr; a.e.lnf_gamma = nullptr; a.e.lnf_beta = nullptr; a.e.lnf_eps = 0.f;
    return 0;
}

// Data-gradient convolution of a conv block (dy (B, T, Cin) bf16 x that block's dgrad weight image -> dx (B, T, Cout)
// bf16) with the BatchNorm-backward REDUCE pass of the block below as its epilogue: dx is that block's d(out), and its
// sums (sum dz | sum dz * xhat over the B * T * pool pre-BN rows y_below) land in sums_below exactly as
// mm_bn_act_bwd_reduce(y_below, out4_below, dx, nullptr, sums_below, B, T * pool, Cout, ...) would leave them.
int mm_conv1d_dgrad_bn_reduce(const void* dy, const void* w_dgrad, int B, int T, int Cin, int Cout, int taps, int pad,
                              void* dx_bf16, const float* y_below, const float* out4_below, float* sums_below, int act,
                              int pool, int drop_first, float drop_p, uint32_t seed, const uint32_t* seed_epoch,
                              hipStream_t st) {
    MM_REQUIRE(dy && w_dgrad && dx_bf16 && y_below && out4_below && sums_below, "conv1d_dgrad_bn_reduce: null");
    MM_REQUIRE(B > 0 && T > 0 && Cout > 0 && taps >= 1 && taps <= 9 && pad >= 0 && pad < taps, "conv1d_dgrad_bn_reduce: bad dims");
    MM_REQUIRE(Cin > 0 && Cin % 16 == 0 && Cout % 4 == 0, "conv1d_dgrad_bn_reduce: Cin=%d (x16) Cout=%d (x4)", Cin, Cout);
    MM_REQUIRE(taps > 1 || Cout <= 64, "conv1d_dgrad_bn_reduce: the fused reduce runs on the 64 x 64 tile (taps > 1 or Cout <= 64)");
    MM_REQUIRE(pool == 1 || pool == 2, "conv1d_dgrad_bn_reduce: pool=%d", pool);
    MM_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "conv1d_dgrad_bn_reduce: drop_p");
    MM_REQUIRE((size_t)B * T * pool * Cout < (1ull << 32), "conv1d_dgrad_bn_reduce: 32-bit dropout indices");
    ConvArgs a;
    a.x = (const bf16*)dy; a.w = (const bf16*)w_dgrad;
    a.B = B; a.T = T; a.Cin = Cin; a.Cout = Cout; a.taps = taps; a.pad = pad;
    a.e.scale = nullptr; a.e.shift = nullptr; a.e.residual = nullptr; a.e.pe = nullptr; a.e.stats = nullptr;
    a.e.out_f32 = nullptr; a.e.out_bf16 = (bf16*)dx_bf16; a.e.out_pre = nullptr;
    a.e.act = 0; a.e.pool = 1;
    a.e.drop_thresh = 0u; a.e.drop_seed = 0u; a.e.drop_epoch = nullptr; a.e.drop_inv_keep = 1.f;
    a.e.gradz = nullptr; a.e.gradz_act = 0;
    a.e.ln_x = nullptr; a.e.ln_stat = nullptr; a.e.ln_gamma = nullptr; a.e.ln_dgb = nullptr;
    a.e.pool_out = nullptr; a.e.pool_rows = 0; a.e.pool_scale = 0.f;
    a.e.lnf_out = nullptr; a.e.lnf_stat = nullptr; a.e.lnf_gamma = nullptr; a.e.lnf_beta = nullptr; a.e.lnf_eps = 0.f;
    a.e.bn.y = y_below; a.e.bn.out4 = out4_below; a.e.bn.sums = sums_below;
    a.e.bn.act = act; a.e.bn.pool = pool; a.e.bn.drop_first = drop_first;
    a.e.bn.thresh = drop_p > 0.f ? (uint32_t)((double)drop_p * 4294967296.0) : 0u;
    a.e.bn.seed = seed; a.e.bn.inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    a.e.bn.epoch = seed_epoch;
    return conv1d_dispatch(a, st);
}

// y = dropout(x W^T + b) + residual, fp32 rows of width 128 (a transformer sub-layer's output), with up to two
// fused consumers of the finished rows: the mean over each group of rows_per_group rows (the encoder's pooling
// step, pool_out zeroed by the caller) and LayerNorm-128 (the next sub-layer's pre-norm: bf16 rows + mean/rstd).
static int linear128_fwd(const void* x, const void* w, int M, int K, const float* bias, const float* residual,
                         float* out_f32, float drop_p, uint32_t seed, const uint32_t* seed_epoch, float* pool_out,
                         int rows_per_group, const float* ln_gamma, const float* ln_beta, float ln_eps, void* ln_out,
                         float* ln_stat, hipStream_t st, const void* w2 = nullptr, const float* bias2 = nullptr, int n2 = 0,
                         void* out2 = nullptr, int act2 = 0, float drop2_p = 0.f, uint32_t seed2 = 0, void* pre2 = nullptr) {
    MM_REQUIRE(x && w && out_f32 && M > 0 && M % 32 == 0 && K > 0 && K % 16 == 0, "linear128_fwd: M=%d (x32) K=%d (x16)", M, K);
    MM_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "linear128_fwd: drop_p");
    MM_REQUIRE(!pool_out || (rows_per_group > 0 && rows_per_group % 32 == 0 && M % rows_per_group == 0),
               "linear128_fwd: rows_per_group=%d must be a multiple of 32 dividing M=%d", rows_per_group, M);
    MM_REQUIRE(!ln_out || (ln_gamma && ln_beta), "linear128_fwd: LayerNorm parameters");
    ConvArgs a;
    a.x = (const bf16*)x; a.w = (const bf16*)w;
    a.B = 1; a.T = M; a.Cin = K; a.Cout = 128; a.taps = 1; a.pad = 0;
    a.e.scale = nullptr; a.e.shift = bias; a.e.residual = residual; a.e.pe = nullptr; a.e.stats = nullptr;
    a.e.out_f32 = out_f32; a.e.out_bf16 = nullptr; a.e.out_pre = nullptr;
    a.e.act = 0; a.e.pool = 1;
    a.e.drop_thresh = drop_p > 0.f ? (uint32_t)((double)drop_p * 4294967296.0) : 0u;
    a.e.drop_seed = seed; a.e.drop_epoch = seed_epoch;
    a.e.drop_inv_keep = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.f;
    a.e.gradz = nullptr; a.e.gradz_act = 0;
    a.e.ln_x = nullptr; a.e.ln_stat = nullptr; a.e.ln_gamma = nullptr; a.e.ln_dgb = nullptr;
    a.e.pool_out = pool_out; a.e.pool_rows = pool_out ? rows_per_group : 0;
    a.e.pool_scale = pool_out ? 1.f / (float)rows_per_group : 0.f;
    a.e.lnf_out = (bf16*)ln_out; a.e.lnf_stat = ln_stat; a.e.lnf_gamma = ln_gamma; a.e.lnf_beta = ln_beta; a.e.lnf_eps = ln_eps;
    if (w2) {
        MM_REQUIRE(ln_out && out2 && n2 > 0 && n2 % 128 == 0, "linear128_fwd: the second GEMM needs the LayerNorm rows, an output and n2 %% 128 == 0 (n2=%d)", n2);
        MM_REQUIRE(drop2_p >= 0.f && drop2_p < 1.f && (size_t)M * n2 < (1ull << 32), "linear128_fwd: second GEMM dropout / 32-bit indices");
        a.e.w2 = (const bf16*)w2; a.e.bias2 = bias2; a.e.n2 = n2; a.e.out2 = (bf16*)out2;
        a.e.act2 = act2; a.e.pre2 = (bf16*)pre2;
        a.e.thresh2 = drop2_p > 0.f ? (uint32_t)((double)drop2_p * 4294967296.0) : 0u;
        a.e.seed2 = seed2; a.e.inv_keep2 = drop2_p > 0.f ? 1.0f / (1.0f - drop2_p) : 1.f;
    }
    const int kct = (K % 128 == 0) ? 128 : (K % 64 == 0 ? 64 : (K % 32 == 0 ? 32 : 16));
    switch (kct) {
        case 16: return launch_fwd<32, 128, 1, 4, 16>(a, st);
        case 32: return launch_fwd<32, 128, 1, 4, 32>(a, st);
        case 64: return launch_fwd<32, 128, 1, 4, 64>(a, st);
        default: return launch_fwd<32, 128, 1, 4, 128>(a, st);
    }
}

int mm_linear_fwd_meanpool(const void* x, const void* w, int M, int K, const float* bias, const float* residual,
                           float* out_f32, float drop_p, uint32_t seed, const uint32_t* seed_epoch, float* pool_out,
                           int rows_per_group, hipStream_t st) {
    MM_REQUIRE(pool_out, "linear_fwd_meanpool: null pool_out");
    return linear128_fwd(x, w, M, K, bias, residual, out_f32, drop_p, seed, seed_epoch, pool_out, rows_per_group, nullptr,
                         nullptr, 0.f, nullptr, nullptr, st);
}

int mm_linear_fwd_ln(const void* x, const void* w, int M, int K, const float* bias, const float* residual, float* out_f32,
                     float drop_p, uint32_t seed, const uint32_t* seed_epoch, const float* ln_gamma, const float* ln_beta,
                     float ln_eps, void* ln_out_bf16, float* ln_stat, hipStream_t st) {
    MM_REQUIRE(ln_out_bf16, "linear_fwd_ln: null ln_out");
    return linear128_fwd(x, w, M, K, bias, residual, out_f32, drop_p, seed, seed_epoch, nullptr, 0, ln_gamma, ln_beta, ln_eps,
                         ln_out_bf16, ln_stat, st);
}

// mm_linear_fwd_ln followed, inside the launch, by out2 = ln_out @ w2^T + bias2 (M x 128 x n2): the projection that
// consumes the fused LayerNorm's rows (the next TemporalTransformerBlock's in_proj: n2 = 384).  w2 = that Linear's forward
// weight image (n2 rows of 128); bit-identical to mm_conv1d_fwd(ln_out, w2, 1, M, 128, n2, 1, 0, NULL, bias2, ..., bf16 out).
int mm_linear_fwd_ln_gemm2(const void* x, const void* w, int M, int K, const float* bias, const float* residual, float* out_f32,
                           float drop_p, uint32_t seed, const uint32_t* seed_epoch, const float* ln_gamma, const float* ln_beta,
                           float ln_eps, void* ln_out_bf16, float* ln_stat, const void* w2, const float* bias2, int n2,
                           void* out2_bf16, hipStream_t st) {
    MM_REQUIRE(ln_out_bf16 && w2 && out2_bf16, "linear_fwd_ln_gemm2: null");
    return linear128_fwd(x, w, M, K, bias, residual, out_f32, drop_p, seed, seed_epoch, nullptr, 0, ln_gamma, ln_beta, ln_eps,
                         ln_out_bf16, ln_stat, st, w2, bias2, n2, out2_bf16);
}

// mm_linear_fwd_ln_gemm2 with an epilogue on the second GEMM: out2 = dropout(act(ln_out @ w2^T + bias2)), pre2 (nullable) =
// the bf16 pre-activation - the first FFN Linear (128 -> n2 = 512, GELU, Dropout) on the rows of the norm2 that the attention
// out-projection's launch has just formed.  Bit-identical to mm_conv1d_fwd(ln_out, w2, 1, M, 128, n2, 1, 0, NULL, bias2, act,
// ..., out_bf16 = out2, out_pre = pre2, drop2_p, seed2, seed_epoch, ...).
int mm_linear_fwd_ln_gemm2_act(const void* x, const void* w, int M, int K, const float* bias, const float* residual,
                               float* out_f32, float drop_p, uint32_t seed, const uint32_t* seed_epoch, const float* ln_gamma,
                               const float* ln_beta, float ln_eps, void* ln_out_bf16, float* ln_stat, const void* w2,
                               const float* bias2, int n2, void* out2_bf16, void* pre2_bf16, int act2, float drop2_p,
                               uint32_t seed2, hipStream_t st) {
    MM_REQUIRE(ln_out_bf16 && w2 && out2_bf16, "linear_fwd_ln_gemm2_act: null");
    return linear128_fwd(x, w, M, K, bias, residual, out_f32, drop_p, seed, seed_epoch, nullptr, 0, ln_gamma, ln_beta, ln_eps,
                         ln_out_bf16, ln_stat, st, w2, bias2, n2, out2_bf16, act2, drop2_p, seed2, pre2_bf16);
}

// dx = LayerNorm128_backward(dy @ W^T) + dres in one launch: the data-gradient GEMM of the Linear that
// consumed LN(x) (dy (M, K) bf16, w = that Linear's dgrad image (128 rows of K)) with the LayerNorm
// backward as its epilogue.  Same results as mm_conv1d_fwd followed by mm_layernorm_bwd, except that the
// d(LN output) rows stay fp32 instead of a bf16 round trip.
static int linear_dgrad_ln_bwd(const void* dy, const void* w, int M, int K, const float* x, const float* stat,
                               const float* gamma, const float* dres, float* dx, void* dx_bf16, float* dgb_repl,
                               float drop_p, uint32_t seed, const uint32_t* seed_epoch, const BnRed* bn, hipStream_t st,
                               const void* w2 = nullptr, void* out2 = nullptr, int res_rows = 0) {
    MM_REQUIRE(dy && w && x && stat && gamma && (dx || dx_bf16), "linear_dgrad_ln_bwd: null");
    MM_REQUIRE(!w2 || (out2 && dx_bf16), "linear_dgrad_ln_bwd: the second GEMM needs the bf16 rows and an output");
    MM_REQUIRE(M > 0 && M % 32 == 0 && K > 0 && K % 16 == 0, "linear_dgrad_ln_bwd: M=%d (multiple of 32) K=%d (multiple of 16)", M, K);
    MM_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "linear_dgrad_ln_bwd: drop_p");
    ConvArgs a;
    a.x = (const bf16*)dy; a.w = (const bf16*)w;
    a.B = 1; a.T = M; a.Cin = K; a.Cout = 128; a.taps = 1; a.pad = 0;
    a.e.scale = nullptr; a.e.shift = nullptr; a.e.residual = dres; a.e.pe = nullptr; a.e.stats = nullptr;
    a.e.out_f32 = dx; a.e.out_bf16 = (bf16*)dx_bf16; a.e.out_pre = nullptr;
    a.e.act = 0; a.e.pool = 1;
    a.e.drop_thresh = drop_p > 0.f ? (uint32_t)((double)drop_p * 4294967296.0) : 0u;
    a.e.drop_seed = seed; a.e.drop_epoch = seed_epoch;
    a.e.drop_inv_keep = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.f;
    a.e.gradz = nullptr; a.e.gradz_act = 0;
    a.e.ln_x = x; a.e.ln_stat = stat; a.e.ln_gamma = gamma; a.e.ln_dgb = dgb_repl;
    a.e.pool_out = nullptr; a.e.pool_rows = 0; a.e.pool_scale = 0.f;
    a.e.lnf_out = nullptr; a.e.lnf_stat = nullptr; a.e.lnf_gamma = nullptr; a.e.lnf_beta = nullptr; a.e.lnf_eps = 0.f;
    if (bn) a.e.bn = *bn;
    a.e.w2 = (const bf16*)w2; a.e.out2 = (bf16*)out2;
    MM_REQUIRE(res_rows >= 0 && (!res_rows || (dres && M % res_rows == 0 && (size_t)M < (1ull << 32))), "linear_dgrad_ln_bwd: res_rows=%d", res_rows);
    a.e.res_rows = res_rows;
    const int kct = (K % 128 == 0) ? 128 : (K % 64 == 0 ? 64 : (K % 32 == 0 ? 32 : 16));
    switch (kct) {
        case 16: return launch_fwd<32, 128, 1, 4, 16>(a, st);
        case 32: return launch_fwd<32, 128, 1, 4, 32>(a, st);
        case 64: return launch_fwd<32, 128, 1, 4, 64>(a, st);
        default: return launch_fwd<32, 128, 1, 4, 128>(a, st);
    }
}

int mm_linear_dgrad_ln_bwd(const void* dy, const void* w, int M, int K, const float* x, const float* stat,
                           const float* gamma, const float* dres, float* dx, void* dx_bf16, float* dgb_repl,
                           float drop_p, uint32_t seed, const uint32_t* seed_epoch, hipStream_t st) {
    return linear_dgrad_ln_bwd(dy, w, M, K, x, stat, gamma, dres, dx, dx_bf16, dgb_repl, drop_p, seed, seed_epoch, nullptr, st);
}

// mm_linear_dgrad_ln_bwd followed, inside the launch, by do = dx_bf16 @ w2 (M x 128 x 128): the data gradient of the
// Linear(128 -> 128) whose dropped-out output entered this LayerNorm's input through the residual add (the attention
// out-projection: x1 = x0 + drop(o Wo^T + bo), norm2(x1)) - dx_bf16 carries exactly that dropout mask (drop_p, seed).
// w2 = that Linear's data-gradient weight image (128 rows of 128); do (M, 128) bf16, bit-identical to
// mm_conv1d_fwd(dx_bf16, w2, ...) with a bf16 output.  dres_rows_per_sample > 0: dres is (M / that, 128) - ONE skip-gradient
// row for that many consecutive rows (the backward of a mean over a sample's tokens, mm_pooled_head_bwd_rows).
int mm_linear_dgrad_ln_bwd_gemm2(const void* dy, const void* w, int M, int K, const float* x, const float* stat,
                                 const float* gamma, const float* dres, float* dx, void* dx_bf16, float* dgb_repl,
                                 float drop_p, uint32_t seed, const uint32_t* seed_epoch, const void* w2, void* do_bf16,
                                 int dres_rows_per_sample, hipStream_t st) {
    MM_REQUIRE(w2 && do_bf16 && dx_bf16, "linear_dgrad_ln_bwd_gemm2: null");
    return linear_dgrad_ln_bwd(dy, w, M, K, x, stat, gamma, dres, dx, dx_bf16, dgb_repl, drop_p, seed, seed_epoch, nullptr, st,
                               w2, do_bf16, dres_rows_per_sample);
}

// mm_linear_dgrad_ln_bwd whose rows dx are the fp32 d(out) of a 128-channel, un-pooled conv block (Conv1d -> BatchNorm1d
// -> act -> Dropout(p) -> + positional table -> Dropout(p2)): that block's BatchNorm-backward reduce pass rides in the same
// launch.  y_below (M, 128) fp32, out4_below, sums_below (zeroed [32][2][128] workspace) and act / drop_p / seed / drop2_p /
// seed2 as in mm_bn_act_bwd_reduce(y_below, out4_below, NULL, dx, sums_below, 1, M, 128, act, 1, 1, ...).
int mm_linear_dgrad_ln_bwd_bn_reduce(const void* dy, const void* w, int M, int K, const float* x, const float* stat,
                                     const float* gamma, const float* dres, float* dx, float* dgb_repl,
                                     const uint32_t* seed_epoch, const float* y_below, const float* out4_below,
                                     float* sums_below, int act, float bn_drop_p, uint32_t bn_seed, float bn_drop2_p,
                                     uint32_t bn_seed2, hipStream_t st) {
    MM_REQUIRE(dx && y_below && out4_below && sums_below, "linear_dgrad_ln_bwd_bn_reduce: null");
    MM_REQUIRE(bn_drop_p >= 0.f && bn_drop_p < 1.f && bn_drop2_p >= 0.f && bn_drop2_p < 1.f, "linear_dgrad_ln_bwd_bn_reduce: drop_p");
    MM_REQUIRE((size_t)M * 128 < (1ull << 32), "linear_dgrad_ln_bwd_bn_reduce: 32-bit dropout indices");
    BnRed bn;
    bn.y = y_below; bn.out4 = out4_below; bn.sums = sums_below; bn.act = act; bn.pool = 1; bn.drop_first = 1;
    bn.thresh = bn_drop_p > 0.f ? (uint32_t)((double)bn_drop_p * 4294967296.0) : 0u;
    bn.seed = bn_seed; bn.inv_keep = bn_drop_p > 0.f ? 1.f / (1.f - bn_drop_p) : 1.f;
    bn.thresh2 = bn_drop2_p > 0.f ? (uint32_t)((double)bn_drop2_p * 4294967296.0) : 0u;
    bn.seed2 = bn_seed2; bn.inv_keep2 = bn_drop2_p > 0.f ? 1.f / (1.f - bn_drop2_p) : 1.f;
    bn.epoch = seed_epoch;
    return linear_dgrad_ln_bwd(dy, w, M, K, x, stat, gamma, dres, dx, nullptr, dgb_repl, 0.f, 0u, seed_epoch, &bn, st);
}

// rows of T per workgroup.  Atomic mode: every workgroup ends with 64 x 64 x taps fp32 atomics, so for
// the k > 1 convs few, long workgroups win (sweep on the three EEG convs: 32 / 30 / 20 us at 384
// workgroups, 16 / 19 / 13 us at ~100).  Slot mode has no atomics: parallelism alone decides.
static int wgrad_rows_per_wg(int B, int T, int Cin, int Cout, int taps, int slot_mode) {
    const int tiles = ceil_div(Cout, 64) * ceil_div(Cin, 64);
    const int tilesT = ceil_div(T, WG_MK);
    static const int slot_target = getenv("MM_WG_TARGET") ? atoi(getenv("MM_WG_TARGET")) : 128;   // slot mode, k > 1: 384 1.116, 192 1.107, 128 1.106, 64 1.111 ms/step
    int want_chunks = ceil_div((taps > 1 && !slot_mode) ? 112 : (taps > 1 ? slot_target : 384), tiles * B);
    if (want_chunks < 1) want_chunks = 1;
    if (want_chunks > tilesT) want_chunks = tilesT;
    return ceil_div(tilesT, want_chunks) * WG_MK;
}

// samples per workgroup (conv1d_wgrad_body): 1 unless a sample is a single row chunk AND the output tiles alone fill the chip
static int wgrad_bgroup(int B, int T, int Cin, int Cout, int taps, int slot_mode) {
    const int rows = wgrad_rows_per_wg(B, T, Cin, Cout, taps, slot_mode);
    if (ceil_div(T, rows) != 1) return 1;
    const int tiles = ceil_div(Cout, 64) * ceil_div(Cin, 64);
    int nsl = 384 / tiles;                                          // slots wanted: ~384 workgroups in all
    if (nsl < 1) nsl = 1;
    if (nsl > B) nsl = B;
    return ceil_div(B, nsl);
}

// grouped launches (mm_conv1d_wgrad_many) get their parallelism from the number of problems, so each
// problem is cut into far fewer row chunks: ~32 workgroups per problem instead of 384 (round-1 sweep: 384 1.127, 128 1.124,
// 64 1.120, 32 1.158 ms/step; end of round 2, with the launch on the side stream beside the chain: 128 0.880, 96 0.873,
// 64 0.870, 48 0.866, 32 0.865, 24 0.864, 16 0.887 - fewer workgroups also leave more of the chip to the chain), i.e. 12x less
// slot memory to write and to sum afterwards
static int wgrad_many_target() {
    static const int t = getenv("MM_WGM_TARGET") ? atoi(getenv("MM_WGM_TARGET")) : 32;
    return t;
}
static int wgrad_many_rows_per_wg(int T, int Cin, int Cout) {
    const int tiles = ceil_div(Cout, 128) * ceil_div(Cin, 128);
    const int tilesT = ceil_div(T, WG_MK);
    int want_chunks = ceil_div(wgrad_many_target(), tiles);
    if (want_chunks < 1) want_chunks = 1;
    if (want_chunks > tilesT) want_chunks = tilesT;
    return ceil_div(tilesT, want_chunks) * WG_MK;
}

int mm_conv1d_wgrad_many_slots(int B, int T, int Cin, int Cout, int* slots_host, hipStream_t) {
    MM_REQUIRE(slots_host && B > 0 && T > 0 && Cin > 0 && Cout > 0, "conv1d_wgrad_many_slots: bad args");
    *slots_host = B * ceil_div(T, wgrad_many_rows_per_wg(T, Cin, Cout));
    return 0;
}

int mm_conv1d_wgrad_slots(int B, int T, int Cin, int Cout, int taps, int* slots_host, hipStream_t) {
    MM_REQUIRE(slots_host && B > 0 && T > 0 && Cin > 0 && Cout > 0, "conv1d_wgrad_slots: bad args");
    *slots_host = ceil_div(B, wgrad_bgroup(B, T, Cin, Cout, taps, 1)) * ceil_div(T, wgrad_rows_per_wg(B, T, Cin, Cout, taps, 1));
    return 0;
}

int mm_conv1d_wgrad(const void* dy, const void* x, float* dw, float* dbias, int B, int T, int Cin, int Cout,
                    int taps, int pad, int Cin_real, int64_t sn, int64_t sc, int64_t stap, int nrep,
                    int64_t rep_stride, int slot_mode, hipStream_t st) {
    MM_REQUIRE(dy && x && dw && B > 0 && T > 0, "conv1d_wgrad: null/invalid");
    MM_REQUIRE(slot_mode == 1 && nrep >= 1, "conv1d_wgrad: slot_mode must be 1 (the fp32-atomics mode is gone: results are order-free)");
    MM_REQUIRE(Cin % 8 == 0 && Cout % 8 == 0, "conv1d_wgrad: Cin=%d Cout=%d must be multiples of 8", Cin, Cout);
    MM_REQUIRE(Cin_real > 0 && Cin_real <= Cin, "conv1d_wgrad: Cin_real");
    WgradArgs a;
    a.dy = (const bf16*)dy; a.x = (const bf16*)x; a.dw = dw; a.dbias = dbias;
    a.B = B; a.T = T; a.Cin = Cin; a.Cout = Cout; a.pad = pad; a.Cin_real = Cin_real;
    a.sn = sn; a.sc = sc; a.stap = stap; a.nrep = nrep; a.rep_stride = rep_stride; a.slot_mode = slot_mode;
    a.rows_per_wg = wgrad_rows_per_wg(B, T, Cin, Cout, taps, slot_mode);
    a.bgroup = wgrad_bgroup(B, T, Cin, Cout, taps, slot_mode);
    MM_REQUIRE(!slot_mode || nrep >= ceil_div(B, a.bgroup) * ceil_div(T, a.rows_per_wg),
               "conv1d_wgrad: slot mode needs %d slots (mm_conv1d_wgrad_slots), got %d", ceil_div(B, a.bgroup) * ceil_div(T, a.rows_per_wg), nrep);
    switch (taps) {
        case 1: return launch_wgrad<1>(a, st);
        case 3: return launch_wgrad<3>(a, st);
        case 5: return launch_wgrad<5>(a, st);
        case 7: return launch_wgrad<7>(a, st);
        default: return mm_fail(MM_ERR_UNSUPPORTED, "conv1d_wgrad: taps=%d (1,3,5,7)", taps);
    }
}

// desc (host, 64 bytes each): {dy, x, dw(workspace), dbias (nullable)} pointers, then int B, T, Cin, Cout,
// Cin_real, nslots, 2 x pad.  Linear layers only (taps 1, pad 0), slot mode, workspace layout [slot][n][c].
struct WgradManyDesc { const void* dy; const void* x; float* dw; float* dbias; int B, T, Cin, Cout, Cin_real, nslots, p0, p1; };
int mm_conv1d_wgrad_many(const void* desc_host, int n, hipStream_t st) {
    MM_REQUIRE(desc_host && n > 0, "conv1d_wgrad_many: bad args");
    const WgradManyDesc* d = (const WgradManyDesc*)desc_host;
    for (int base = 0; base < n; base += WM_MAX) {
        WgradTable tab;
        tab.n = (n - base < WM_MAX) ? n - base : WM_MAX;
        int total = 0;
        for (int i = 0; i < tab.n; ++i) {
            const WgradManyDesc& q = d[base + i];
            MM_REQUIRE(q.dy && q.x && q.dw && q.B > 0 && q.T > 0, "conv1d_wgrad_many: null/invalid");
            MM_REQUIRE(q.Cin % 8 == 0 && q.Cout % 8 == 0 && q.Cin_real > 0 && q.Cin_real <= q.Cin,
                       "conv1d_wgrad_many: Cin=%d Cout=%d", q.Cin, q.Cout);
            WgradArgs& a = tab.a[i];
            a.dy = (const bf16*)q.dy; a.x = (const bf16*)q.x; a.dw = q.dw; a.dbias = q.dbias;
            a.B = q.B; a.T = q.T; a.Cin = q.Cin; a.Cout = q.Cout; a.pad = 0; a.Cin_real = q.Cin_real;
            a.sn = q.Cin; a.sc = 1; a.stap = q.Cin; a.nrep = q.nslots; a.rep_stride = (long)q.Cout * q.Cin; a.slot_mode = 1;
            a.rows_per_wg = wgrad_many_rows_per_wg(q.T, q.Cin, q.Cout);
            a.bgroup = 1;
            const int chunks = q.B * ceil_div(q.T, a.rows_per_wg);
            MM_REQUIRE(q.nslots == chunks, "conv1d_wgrad_many: needs exactly %d slots (mm_conv1d_wgrad_many_slots), got %d",
                       chunks, q.nslots);
            tab.first[i] = total;
            tab.gx[i] = chunks; tab.gy[i] = ceil_div(q.Cout, 128);
            total += chunks * tab.gy[i] * ceil_div(q.Cin, 128);
        }
        tab.first[tab.n] = total;
        hipLaunchKernelGGL(conv1d_wgrad_many_kernel, dim3(total), dim3(256), 0, st, tab);
        const int rc = mm_check_launch("conv1d_wgrad_many");
        if (rc) return rc;
    }
    return 0;
}

// many independent reductions into parameter gradients in one launch: desc[i] = {src, dst, K, nrep, stride};
// nrep = MM_ACC_REPL: src is a fixed-point accumulator workspace (stride in 64-bit elements);
// nrep = 1: src is a compact fp32 vector (plain dst[k] += src[k])
struct ReduceDesc { const void* src; float* dst; long K, nrep, stride; };
constexpr int RM_MAX = 64;
struct ReduceTable { ReduceDesc d[RM_MAX]; };      // passed BY VALUE (kernel argument): no memcpy node,
                                                   // so the launch can be recorded in a hipGraph
__device__ __forceinline__ void reduce_body(const ReduceDesc& d, int blk, int nblk) {
    if (d.nrep == 1) {
        const float* src = reinterpret_cast<const float*>(d.src);
        for (long k = (long)blk * 256 + threadIdx.x; k < d.K; k += (long)nblk * 256) d.dst[k] += src[k];
        return;
    }
    acc_reduce_rows(reinterpret_cast<const mm_acc_t*>(d.src), d.dst, d.K, d.stride, (long)blk * 16, (long)nblk * 16);
}
__global__ void reduce_many_kernel(ReduceTable tab) { reduce_body(tab.d[blockIdx.y], blockIdx.x, gridDim.x); }

// the slot sums AND the accumulator reductions of one gradient flush in ONE launch (they are independent; two graph
// nodes cost ~5 us of latency each on the stream that flushes): blocks [0, 256 ns) scatter, the rest reduce
constexpr int FM_MAX = 48, FM_SB = 256, FM_RB = 16;
struct FlushTable { ScatterDesc s[FM_MAX]; ReduceDesc r[FM_MAX]; int ns, nr; };
__global__ void flush_many_kernel(FlushTable tab) {
    const int b = blockIdx.x;
    if (b < tab.ns * FM_SB) scatter_body(tab.s[b / FM_SB], b % FM_SB, FM_SB);
    else reduce_body(tab.r[(b - tab.ns * FM_SB) / FM_RB], (b - tab.ns * FM_SB) % FM_RB, FM_RB);
}

int mm_reduce_many(const void* desc_host, int ndesc, hipStream_t st) {
    MM_REQUIRE(desc_host && ndesc > 0, "reduce_many: bad args");
    const ReduceDesc* src = (const ReduceDesc*)desc_host;
    for (int base = 0; base < ndesc; base += RM_MAX) {
        ReduceTable tab;
        const int n = ndesc - base < RM_MAX ? ndesc - base : RM_MAX;
        for (int i = 0; i < n; ++i) {
            tab.d[i] = src[base + i];
            MM_REQUIRE(tab.d[i].src && tab.d[i].dst && tab.d[i].K > 0 && tab.d[i].stride >= tab.d[i].K &&
                           (tab.d[i].nrep == 1 || (tab.d[i].nrep == MM_ACC_REPL && ((uintptr_t)tab.d[i].src & 7) == 0)),
                       "reduce_many: descriptor %d (nrep = 1 fp32 vector, or %d accumulator replicas)", base + i, MM_ACC_REPL);
        }
        hipLaunchKernelGGL(reduce_many_kernel, dim3(16, n), dim3(256), 0, st, tab);
    }
    return mm_check_launch("reduce_many");
}

int mm_reduce_replicas(const float* src, float* dst, int K, int nrep, int64_t rep_stride, hipStream_t st) {
    MM_REQUIRE(src && dst && K > 0 && nrep >= 1 && rep_stride >= K, "reduce_replicas: bad args");
    hipLaunchKernelGGL(reduce_replicas_kernel, dim3(ceil_div(K, 8)), dim3(256), 0, st, src, dst, K, nrep, (long)rep_stride);
    return mm_check_launch("reduce_replicas");
}

int mm_flush_many(const void* scatter_desc_host, int nscatter, const void* reduce_desc_host, int nreduce, hipStream_t st) {
    MM_REQUIRE(nscatter >= 0 && nreduce >= 0 && nscatter + nreduce > 0 && (scatter_desc_host || !nscatter) &&
                   (reduce_desc_host || !nreduce), "flush_many: bad args");
    const ScatterDesc* sd_in = (const ScatterDesc*)scatter_desc_host;
    const ReduceDesc* rd = (const ReduceDesc*)reduce_desc_host;
    static_assert(sizeof(FlushTable) <= 4096, "kernel arguments");
    // every scatter descriptor gets FM_SB workgroups: a big workspace (config #5's merged convolution: 8.4 M weights) is dealt
    // out as up to 8 descriptors over slices of its output channels, so that it gets 8 x the workgroups
    constexpr int EXP_MAX = 1024;
    static thread_local ScatterDesc expanded[EXP_MAX];
    int nexp = 0;
    for (int i = 0; i < nscatter; ++i) {
        const ScatterDesc& d = sd_in[i];
        MM_REQUIRE(scatter_desc_ok(d), "flush_many: scatter descriptor %d", i);
        const int tws = scatter_ws_taps(d);
        const size_t elems = (size_t)d.Cout * scatter_taps(d) * d.Cinp;
        int parts = (int)((elems + (1u << 20) - 1) >> 20);
        if (parts > 8) parts = 8;
        if (parts > d.Cout) parts = d.Cout;
        if (parts < 1) parts = 1;
        MM_REQUIRE(nexp + parts <= EXP_MAX, "flush_many: too many scatter descriptors");
        for (int q = 0; q < parts; ++q) {
            const int o0 = (int)((long)d.Cout * q / parts), o1 = (int)((long)d.Cout * (q + 1) / parts);
            ScatterDesc e = d;
            e.ws = d.ws + (size_t)o0 * tws * d.Cinp;
            e.dw = d.dw + (size_t)o0 * d.Cin * scatter_taps(d);
            e.Cout = o1 - o0;
            e.cout_all = d.cout_all ? d.cout_all : d.Cout;
            expanded[nexp++] = e;
        }
    }
    const ScatterDesc* sd = expanded;
    nscatter = nexp;
    for (int sb = 0, rb = 0; sb < nscatter || rb < nreduce; sb += FM_MAX, rb += FM_MAX) {
        FlushTable tab;
        tab.ns = nscatter - sb > FM_MAX ? FM_MAX : (nscatter - sb > 0 ? nscatter - sb : 0);
        tab.nr = nreduce - rb > FM_MAX ? FM_MAX : (nreduce - rb > 0 ? nreduce - rb : 0);
        for (int i = 0; i < tab.ns; ++i) tab.s[i] = sd[sb + i];
        for (int i = 0; i < tab.nr; ++i) {
            const ReduceDesc& d = rd[rb + i];
            MM_REQUIRE(d.src && d.dst && d.K > 0 && d.stride >= d.K && (d.nrep == 1 || (d.nrep == MM_ACC_REPL && ((uintptr_t)d.src & 7) == 0)),
                       "flush_many: reduce descriptor %d (nrep = 1 fp32 vector, or %d accumulator replicas)", rb + i, MM_ACC_REPL);
            tab.r[i] = d;
        }
        hipLaunchKernelGGL(flush_many_kernel, dim3(tab.ns * FM_SB + tab.nr * FM_RB), dim3(256), 0, st, tab);
    }
    return mm_check_launch("flush_many");
}

int mm_acc_reduce(const float* acc, float* dst, int K, int64_t rep_stride, hipStream_t st) {
    MM_REQUIRE(acc && dst && K > 0 && rep_stride >= K, "acc_reduce: bad args");
    MM_REQUIRE(((uintptr_t)acc & 7) == 0, "acc_reduce: workspace must be 8-byte aligned");
    hipLaunchKernelGGL(acc_reduce_kernel, dim3(ceil_div(K, 16)), dim3(256), 0, st, reinterpret_cast<const mm_acc_t*>(acc), dst, K,
                       (long)rep_stride);
    return mm_check_launch("acc_reduce");
}

int mm_wgrad_scatter(const float* ws, float* dw, int Cout, int Cin, int taps, int Cinp, int nrep, hipStream_t st) {
    MM_REQUIRE(ws && dw && Cout > 0 && Cin > 0 && taps > 0 && Cinp >= Cin && nrep >= 1, "wgrad_scatter: bad args");
    const size_t total = (size_t)Cout * Cin * taps;
    int grid = (int)((total + 255) / 256);
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(wgrad_scatter_kernel, dim3(grid), dim3(256), 0, st, ws, dw, Cout, Cin, taps, Cinp, nrep);
    return mm_check_launch("wgrad_scatter");
}

int mm_scatter_many(const void* desc_host, int ndesc, hipStream_t st) {
    MM_REQUIRE(desc_host && ndesc > 0, "scatter_many: bad args");
    const ScatterDesc* src = (const ScatterDesc*)desc_host;
    for (int base = 0; base < ndesc; base += SM_MAX) {
        ScatterTable tab;
        const int n = ndesc - base < SM_MAX ? ndesc - base : SM_MAX;
        for (int i = 0; i < n; ++i) {
            const ScatterDesc& d = src[base + i];
            MM_REQUIRE(scatter_desc_ok(d), "scatter_many: descriptor %d", base + i);
            tab.d[i] = d;
        }
        hipLaunchKernelGGL(scatter_many_kernel, dim3(256, n), dim3(256), 0, st, tab);
    }
    return mm_check_launch("scatter_many");
}

}  // extern "C"
