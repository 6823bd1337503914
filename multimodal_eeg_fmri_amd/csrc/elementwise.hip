// HBM-bound row/channel kernels: BatchNorm finalize + BN/act/pool apply (fwd,
// bwd), LayerNorm (fwd, bwd), mean-pool, column sums, casts.  All statistics,
// normalisation and activation math is fp32; bf16 only at rest.
#include "common.h"

namespace {

// ---------------------------------------------------------------------------
// BatchNorm finalize.  mode 0 (train): stats = {sum, sumsq} over `count`
// samples -> mean / rstd, scale = g*rstd, shift = b - mean*scale, and the
// running-stat update (momentum, unbiased variance) of nn.BatchNorm*d.
// mode 1 (eval): fold running stats (+ optional conv bias) into scale/shift.
// ---------------------------------------------------------------------------
__global__ void bn_finalize_kernel(const float* stats, const float* gamma, const float* beta,
                                   float* run_mean, float* run_var, const float* conv_bias,
                                   float* out /* [4][N]: scale, shift, mean, rstd */, int N,
                                   float count, float momentum, float eps, int mode, long long* tracked) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    if (mode == 0) {                                     // the shared train-mode finalize (csrc/common.h): same bits as the *_fin consumers
        MmBnFin f{stats, gamma, beta, run_mean, run_var, out, tracked, count, momentum, eps, N};
        float sc, sh, mean, rstd;
        bn_fin_channel(f, n, true, sc, sh, mean, rstd);
        return;
    }
    const float mean = run_mean[n], var = run_var[n];
    const float rstd = rsqrtf(var + eps);
    const float sc = gamma[n] * rstd;
    const float cb = (mode == 1 && conv_bias) ? conv_bias[n] : 0.f;
    out[n] = sc;
    out[N + n] = beta[n] + (cb - mean) * sc;
    out[2 * N + n] = mean;
    out[3 * N + n] = rstd;
}

// ---------------------------------------------------------------------------
// y (fp32 [R][S][N], S = pooled axis length) -> act(y*scale+shift) -> pool ->
// dropout -> (+pe) -> bf16 / fp32.   R = batch, S = T (1-D).  pool in {1,2}.
// drop_first: dropout before the pool (Lite encoders) or after it (ERP conv2).
// ---------------------------------------------------------------------------
struct BnActArgs {
    const float* y; const float* scale; const float* shift; const float* pe;
    bf16* out_bf16; float* out_f32;
    int R, S, N, act, pool, drop_first;
    uint32_t thresh, seed; float inv_keep;
    uint32_t thresh2, seed2; float inv_keep2;
    const uint32_t* epoch;
    // LayerNorm-128 of every finished row (the transformer stack's first norm1), N == 128 and pool == 1 only:
    // 32 consecutive lanes hold one row
    const float* ln_gamma = nullptr; const float* ln_beta = nullptr; float ln_eps = 0.f;
    bf16* ln_out = nullptr; float* ln_stat = nullptr;
    MmBnFin fin = {};            // FIN kernels: the BatchNorm finalize runs in the prologue (scale / shift are then unused)
};

template <int ACT>
__device__ __forceinline__ float bnact_one(const BnActArgs& a, float y, float sc, float sh, uint32_t idx, bool drop_here) {
    float v = apply_act(y * sc + sh, ACT >= 0 ? ACT : a.act);
    if (drop_here && a.thresh) v *= dropout_scale(a.seed, idx, a.thresh, a.inv_keep);
    return v;
}

// ACT >= 0 / POOL > 0: compiled for that activation / pool size (GELU with pool 1 and 2: every BatchNorm of the encoders)
// FIN: the train-mode BatchNorm finalize of the layer is this kernel's prologue (every workgroup forms scale / shift of all
// N <= 256 channels in LDS from the statistics workspace; workgroup 0 also writes out4 and the running statistics): the
// one-workgroup mm_bn_finalize launch between the convolution and this pass - ~5 us + a graph node on the chain - is gone.
template <int ACT = -1, int POOL = 0, bool FIN = false>
__global__ void bn_act_fwd_kernel(BnActArgs a) {
    a.seed = mm_eff_seed(a.seed, a.epoch);
    a.seed2 = mm_eff_seed(a.seed2, a.epoch);
    if (POOL > 0) a.pool = POOL;
    __shared__ __attribute__((aligned(16))) float s_sc[FIN ? 256 : 4], s_sh[FIN ? 256 : 4];
    if (FIN) {
        for (int n = threadIdx.x; n < a.N; n += blockDim.x) {
            float sc, sh, mean, rstd;
            bn_fin_channel(a.fin, n, blockIdx.x == 0, sc, sh, mean, rstd);
            s_sc[n] = sc; s_sh[n] = sh;
        }
        __syncthreads();
    }
    const int So = a.S / a.pool;
    const unsigned nv = a.N / 4;
    const unsigned total = (unsigned)((size_t)a.R * So * nv);       // < 2^31 elements / 4: checked on the host
    // index arithmetic in 32 bits and without divisions where the layout allows it: output element 4 i, input row
    // (i / nv) * pool.  (Four 64-bit divisions per four elements made these passes 4x slower than their HBM traffic.)
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const unsigned rs = i / nv;                                 // r * So + so
        const int n4 = (int)(i - rs * nv) * 4;
        const int so = a.pe ? (int)(rs % (unsigned)So) : 0;         // only the positional table needs the position itself
        const float4 sc = FIN ? *reinterpret_cast<const float4*>(s_sc + n4) : *reinterpret_cast<const float4*>(a.scale + n4);
        const float4 sh = FIN ? *reinterpret_cast<const float4*>(s_sh + n4) : *reinterpret_cast<const float4*>(a.shift + n4);
        float o[4];
        const size_t in0 = (size_t)rs * a.pool * a.N + n4;           // r * S + so * pool = (r * So + so) * pool
        const float4 y0 = *reinterpret_cast<const float4*>(a.y + in0);
        const float scs[4] = {sc.x, sc.y, sc.z, sc.w}, shs[4] = {sh.x, sh.y, sh.z, sh.w};
        const float y0s[4] = {y0.x, y0.y, y0.z, y0.w};
        const size_t oidx = (size_t)i * 4;                           // (r * So + so) * N + n4
        if (a.pool == 2) {
            const float4 y1 = *reinterpret_cast<const float4*>(a.y + in0 + a.N);
            const float y1s[4] = {y1.x, y1.y, y1.z, y1.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float m;
                const int act = ACT >= 0 ? ACT : a.act;
                const float z0 = y0s[q] * scs[q] + shs[q], z1 = y1s[q] * scs[q] + shs[q];
                if (act == MM_ACT_GELU && !(a.drop_first && a.thresh) && fmaxf(z0, z1) >= 0.f) {
                    m = gelu_erf(fmaxf(z0, z1));        // the larger pre-activation, if >= 0, holds the larger GELU: one evaluation
                } else {
                    const float v0 = bnact_one<ACT>(a, y0s[q], scs[q], shs[q], (uint32_t)(in0 + q), a.drop_first);
                    const float v1 = bnact_one<ACT>(a, y1s[q], scs[q], shs[q], (uint32_t)(in0 + a.N + q), a.drop_first);
                    m = fmaxf(v0, v1);
                }
                if (!a.drop_first && a.thresh) m *= dropout_scale(a.seed, (uint32_t)(oidx + q), a.thresh, a.inv_keep);
                o[q] = m;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = bnact_one<ACT>(a, y0s[q], scs[q], shs[q], (uint32_t)(in0 + q), true);
        }
        if (a.pe) {
            const float4 p = *reinterpret_cast<const float4*>(a.pe + (size_t)so * a.N + n4);
            o[0] += p.x; o[1] += p.y; o[2] += p.z; o[3] += p.w;
        }
        if (a.thresh2)
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] *= dropout_scale(a.seed2, (uint32_t)(oidx + q), a.thresh2, a.inv_keep2);
        if (a.out_f32) *reinterpret_cast<float4*>(a.out_f32 + oidx) = make_float4(o[0], o[1], o[2], o[3]);
        if (a.out_bf16) {
            bf16x4 b = {(bf16)o[0], (bf16)o[1], (bf16)o[2], (bf16)o[3]};
            *reinterpret_cast<bf16x4*>(a.out_bf16 + oidx) = b;
        }
        if (a.ln_out) {
            // (host: N == 128, pool == 1, total a multiple of 32: every 32-lane group runs this together on one row;
            //  arithmetic as epilogue_rows' fused LayerNorm in igemm1d.hip)
            const float mean = half32_sum((o[0] + o[1]) + (o[2] + o[3])) * (1.f / 128.f);
            const float d0 = o[0] - mean, d1 = o[1] - mean, d2 = o[2] - mean, d3 = o[3] - mean;
            const float rstd = rsqrtf(half32_sum((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) * (1.f / 128.f) + a.ln_eps);
            const float4 g4 = *reinterpret_cast<const float4*>(a.ln_gamma + n4);
            const float4 b4 = *reinterpret_cast<const float4*>(a.ln_beta + n4);
            bf16x4 hv = {(bf16)(d0 * rstd * g4.x + b4.x), (bf16)(d1 * rstd * g4.y + b4.y),
                         (bf16)(d2 * rstd * g4.z + b4.z), (bf16)(d3 * rstd * g4.w + b4.w)};
            *reinterpret_cast<bf16x4*>(a.ln_out + oidx) = hv;
            if (a.ln_stat && n4 == 0) { a.ln_stat[2 * (size_t)rs] = mean; a.ln_stat[2 * (size_t)rs + 1] = rstd; }
        }
    }
}

// ---------------------------------------------------------------------------
// backward of the above.  dz = dOut routed through dropout/pool/act'.
//   reduce : sums[0][n] += dz, sums[1][n] += dz * xhat          (train BN)
//   apply  : dy = scale * (dz - sums0/M - xhat * sums1/M)   (train)
//            dy = scale * dz                                 (frozen stats)
// ---------------------------------------------------------------------------
struct BnBwdArgs {
    const float* y; const float* scale; const float* shift; const float* mean; const float* rstd;
    const bf16* dout_bf16; const float* dout_f32; const float* sums;
    float* sums_out; bf16* dy; float* dy_f32;
    int R, S, N, act, pool, drop_first, train;
    uint32_t thresh, seed; float inv_keep, inv_count;
    uint32_t thresh2, seed2; float inv_keep2;
    const uint32_t* epoch;
    int sums_nrep;               // apply: sums is [sums_nrep][2][N]; the block reduces the replicas itself
    // d(out) that is the same row for every position of a sample (the backward of a mean over positions): dout_f32 is
    // [R][N] and every element is multiplied by bcast_scale (1 / positions); pool == 1 only
    int bcast = 0;
    float bcast_scale = 1.f;
};

template <bool APPLY, int ACT = -1, int POOL = 0>
__global__ void bn_act_bwd_kernel(BnBwdArgs a) {
    a.seed = mm_eff_seed(a.seed, a.epoch);
    a.seed2 = mm_eff_seed(a.seed2, a.epoch);
    if (POOL > 0) a.pool = POOL;
    // block = 256 threads = (N/4 channel-vectors) x rows; grid-stride over pooled rows
    const int nv = a.N / 4;
    const int So = a.S / a.pool;
    const int rows_per_blk = 256 / nv > 0 ? 256 / nv : 1;
    const int vi = threadIdx.x % nv, ri = threadIdx.x / nv;
    const bool active = ri < rows_per_blk;
    const int n4 = vi * 4;
    const size_t nrows = (size_t)a.R * So;
    float s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
    float scs[4], shs[4], mus[4], rss[4], c0[4], c1[4];
    __shared__ float csum[APPLY ? 2048 : 1];           // sum dz | sum dz*xhat per channel (N <= 1024)
    // One iteration of input loads stays in flight ahead of the arithmetic, and the first one is issued BEFORE the
    // prologue below: a workgroup used to pay its global round trips one after the other (replica sums, BatchNorm
    // constants, first rows) - with four workgroups per CU all starting together nothing hid them (10 us for 17 MB).
    struct In { float4 y0, y1, gf; bf16x4 gb; };
    const size_t rstep = (size_t)gridDim.x * rows_per_blk;
    size_t row = (size_t)blockIdx.x * rows_per_blk + ri;
    auto load = [&](size_t r, In& v) __attribute__((always_inline)) {
        const size_t in0 = r * a.pool * a.N + n4, oidx = r * a.N + n4;
        if (a.bcast) v.gf = *reinterpret_cast<const float4*>(a.dout_f32 + (size_t)((unsigned)r / (unsigned)a.S) * a.N + n4);
        else if (a.dout_f32) v.gf = *reinterpret_cast<const float4*>(a.dout_f32 + oidx);
        else v.gb = *reinterpret_cast<const bf16x4*>(a.dout_bf16 + oidx);
        v.y0 = *reinterpret_cast<const float4*>(a.y + in0);
        if (a.pool == 2) v.y1 = *reinterpret_cast<const float4*>(a.y + in0 + a.N);
    };
    In cur, nxt;
    bool have = active && row < nrows;
    if (have) load(row, cur);
    if (APPLY && a.train) {
        // replica reduction in the prologue: a separate 5 us compaction launch sat between the two passes
        for (int i = threadIdx.x; i < 2 * a.N; i += 256) {
            // the accumulator workspace as written by the reduce pass (all replica loads in flight at once), or compact fp32
            const float s = a.sums_nrep == 1 ? a.sums[i] : acc_val<MM_ACC_GRAD>(acc_sum(a.sums, 2 * (size_t)a.N, i));
            csum[i] = s * a.inv_count;
        }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        scs[q] = a.scale[n4 + q]; shs[q] = a.shift[n4 + q];
        mus[q] = a.mean ? a.mean[n4 + q] : 0.f; rss[q] = a.rstd ? a.rstd[n4 + q] : 1.f;
        c0[q] = (APPLY && a.train) ? csum[n4 + q] : 0.f;
        c1[q] = (APPLY && a.train) ? csum[a.N + n4 + q] : 0.f;
    }
        for (; have; row += rstep) {
            const bool hn = row + rstep < nrows;
            if (hn) load(row + rstep, nxt);
            const size_t in0 = row * a.pool * a.N + n4;             // (r * S + so * pool) = row * pool: no division
            const size_t oidx = row * a.N + n4;
            float g[4];
            if (a.bcast) { g[0] = cur.gf.x * a.bcast_scale; g[1] = cur.gf.y * a.bcast_scale; g[2] = cur.gf.z * a.bcast_scale; g[3] = cur.gf.w * a.bcast_scale; }
            else if (a.dout_f32) { g[0] = cur.gf.x; g[1] = cur.gf.y; g[2] = cur.gf.z; g[3] = cur.gf.w; }
            else { g[0] = (float)cur.gb[0]; g[1] = (float)cur.gb[1]; g[2] = (float)cur.gb[2]; g[3] = (float)cur.gb[3]; }
            if (a.thresh2)
#pragma unroll
                for (int q = 0; q < 4; ++q) g[q] *= dropout_scale(a.seed2, (uint32_t)(oidx + q), a.thresh2, a.inv_keep2);
            const float4 y0 = cur.y0;
            const float4 y1 = a.pool == 2 ? cur.y1 : y0;
            const float y0s[4] = {y0.x, y0.y, y0.z, y0.w}, y1s[4] = {y1.x, y1.y, y1.z, y1.w};
            float d0[4], d1[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                bn_dz_pair<ACT, POOL>(a, y0s[q], y1s[q], scs[q], shs[q], g[q], (uint32_t)(in0 + q),
                           (uint32_t)(in0 + a.N + q), (uint32_t)(oidx + q), d0[q], d1[q]);
                const float xh0 = (y0s[q] - mus[q]) * rss[q], xh1 = (y1s[q] - mus[q]) * rss[q];
                if (APPLY) {
                    if (a.train) {
                        d0[q] = scs[q] * (d0[q] - c0[q] - xh0 * c1[q]);
                        d1[q] = scs[q] * (d1[q] - c0[q] - xh1 * c1[q]);
                    } else {
                        d0[q] *= scs[q]; d1[q] *= scs[q];
                    }
                } else {
                    s0[q] += d0[q] + d1[q];
                    s1[q] += d0[q] * xh0 + d1[q] * xh1;
                }
            }
            if (APPLY) {
                if (a.dy) {
                    bf16x4 b0 = {(bf16)d0[0], (bf16)d0[1], (bf16)d0[2], (bf16)d0[3]};
                    *reinterpret_cast<bf16x4*>(a.dy + in0) = b0;
                    if (a.pool == 2) {
                        bf16x4 b1 = {(bf16)d1[0], (bf16)d1[1], (bf16)d1[2], (bf16)d1[3]};
                        *reinterpret_cast<bf16x4*>(a.dy + in0 + a.N) = b1;
                    }
                }
                if (a.dy_f32) {
                    *reinterpret_cast<float4*>(a.dy_f32 + in0) = make_float4(d0[0], d0[1], d0[2], d0[3]);
                    if (a.pool == 2) *reinterpret_cast<float4*>(a.dy_f32 + in0 + a.N) = make_float4(d1[0], d1[1], d1[2], d1[3]);
                }
            }
            cur = nxt;
            have = hn;
        }
    if (!APPLY) {
        // block reduction through plain LDS stores + a column walk.  (LDS float atomics with the
        // 8..16-way same-address conflicts this layout has cost ~2 us per workgroup: half the kernel.)
        __shared__ __attribute__((aligned(16))) float part[2048];     // [rows_per_blk][2][N], 2 * N * rows_per_blk <= 2048
        if (active) {
            float* dst = part + (size_t)ri * 2 * a.N + n4;
            *reinterpret_cast<float4*>(dst) = make_float4(s0[0], s0[1], s0[2], s0[3]);
            *reinterpret_cast<float4*>(dst + a.N) = make_float4(s1[0], s1[1], s1[2], s1[3]);
        }
        __syncthreads();
        mm_acc_t* rep = acc_rep(a.sums_out, blockIdx.x % MM_ACC_REPL, 2 * (size_t)a.N);
        for (int i = threadIdx.x; i < 2 * a.N; i += 256) {
            float s = 0.f;
            for (int r = 0; r < rows_per_blk; ++r) s += part[r * 2 * a.N + i];
            acc_add<MM_ACC_GRAD>(&rep[i], s);
        }
    }
}

// ---------------------------------------------------------------------------
// LayerNorm over the last dim (D % 64 == 0, D <= 1024): one wave per row.
// ---------------------------------------------------------------------------
template <int VPL>   // values per lane = D / 64
__global__ void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b,
                                     bf16* __restrict__ out_bf16, float* __restrict__ out_f32,
                                     float* __restrict__ stat /* [M][2] mean,rstd */, int M, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= M) return;
    constexpr int D = VPL * 64;
    float v[VPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) { v[i] = x[(size_t)row * D + i * 64 + lane]; s += v[i]; }
    const float mean = wave_sum(s) * (1.f / D);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) { const float d = v[i] - mean; q += d * d; }
    const float rstd = rsqrtf(wave_sum(q) * (1.f / D) + eps);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = i * 64 + lane;
        const float o = (v[i] - mean) * rstd * g[c] + b[c];
        if (out_bf16) out_bf16[(size_t)row * D + c] = (bf16)o;
        if (out_f32) out_f32[(size_t)row * D + c] = o;
    }
    if (stat && lane == 0) { stat[2 * row] = mean; stat[2 * row + 1] = rstd; }
}

// dx = dres + rstd * (gh - mean(gh) - xhat * mean(gh * xhat)),  gh = dy * gamma
// dgamma += sum_rows dy * xhat ; dbeta += sum_rows dy       (block partial + atomics)
// optional second output: bf16(dx * dropout_mask) for the next GEMM operand.
template <int VPL>
__global__ void layernorm_bwd_kernel(const bf16* __restrict__ dy_bf16, const float* __restrict__ dy_f32,
                                     const float* __restrict__ x, const float* __restrict__ stat,
                                     const float* __restrict__ g, const float* __restrict__ dres,
                                     float* __restrict__ dx, bf16* __restrict__ dx_bf16,
                                     float* __restrict__ dgb, int M, int rows_per_wave, uint32_t thresh,
                                     uint32_t seed, float inv_keep, const uint32_t* epoch) {
    seed = mm_eff_seed(seed, epoch);
    constexpr int D = VPL * 64;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wpb = blockDim.x >> 6;
    float ag[VPL], ab[VPL], gam[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) { ag[i] = 0.f; ab[i] = 0.f; gam[i] = g[i * 64 + lane]; }
    const int row0 = (blockIdx.x * wpb + wave) * rows_per_wave;
    for (int rr = 0; rr < rows_per_wave; ++rr) {
        const int row = row0 + rr;
        if (row >= M) break;
        const float mean = stat[2 * row], rstd = stat[2 * row + 1];
        float dyv[VPL], xh[VPL];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const size_t idx = (size_t)row * D + i * 64 + lane;
            dyv[i] = dy_bf16 ? (float)dy_bf16[idx] : dy_f32[idx];
            xh[i] = (x[idx] - mean) * rstd;
            const float gh = dyv[i] * gam[i];
            s1 += gh; s2 += gh * xh[i];
            ag[i] += dyv[i] * xh[i]; ab[i] += dyv[i];
        }
        s1 = wave_sum(s1) * (1.f / D);
        s2 = wave_sum(s2) * (1.f / D);
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const size_t idx = (size_t)row * D + i * 64 + lane;
            float o = rstd * (dyv[i] * gam[i] - s1 - xh[i] * s2);
            if (dres) o += dres[idx];
            if (dx) dx[idx] = o;
            if (dx_bf16) dx_bf16[idx] = (bf16)(thresh ? o * dropout_scale(seed, (uint32_t)idx, thresh, inv_keep) : o);
        }
    }
    // dgb = accumulator workspace [MM_ACC_REPL][2][D] (gamma row, beta row).  Every wave parks its partials
    // (plain stores) and the four rows are summed in wave order: no LDS atomics, a fixed summation order.
    if (dgb) {
        __shared__ float red[4][2][D];
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            red[wave][0][i * 64 + lane] = ag[i];
            red[wave][1][i * 64 + lane] = ab[i];
        }
        __syncthreads();
        mm_acc_t* rep = acc_rep(dgb, blockIdx.x % MM_ACC_REPL, 2 * D);
        for (int i = threadIdx.x; i < 2 * D; i += blockDim.x) {
            const int which = i / D, col = i % D;
            acc_add<MM_ACC_GRAD>(&rep[i], (red[0][which][col] + red[1][which][col]) + (red[2][which][col] + red[3][which][col]));
        }
    }
}

// ---------------------------------------------------------------------------
// D == 128 fast path (the transformer width): one 32-lane half-wave per row, one
// 16-byte vector per lane, two rows per wave in flight, 5-step shuffle reductions.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float half_sum(float v) { return half32_sum(v); }

__global__ void layernorm128_fwd_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                        const float* __restrict__ b, bf16* __restrict__ out_bf16,
                                        float* __restrict__ out_f32, float* __restrict__ stat, int M, float eps) {
    const int lane = threadIdx.x & 31;
    const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    if (row >= M) return;
    const float4 v = *reinterpret_cast<const float4*>(x + (size_t)row * 128 + lane * 4);
    const float mean = half_sum(v.x + v.y + v.z + v.w) * (1.f / 128.f);
    const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
    const float rstd = rsqrtf(half_sum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) * (1.f / 128.f) + eps);
    const float4 gg = *reinterpret_cast<const float4*>(g + lane * 4);
    const float4 bb = *reinterpret_cast<const float4*>(b + lane * 4);
    const float o0 = d0 * rstd * gg.x + bb.x, o1 = d1 * rstd * gg.y + bb.y, o2 = d2 * rstd * gg.z + bb.z, o3 = d3 * rstd * gg.w + bb.w;
    if (out_f32) *reinterpret_cast<float4*>(out_f32 + (size_t)row * 128 + lane * 4) = make_float4(o0, o1, o2, o3);
    if (out_bf16) {
        bf16x4 ob = {(bf16)o0, (bf16)o1, (bf16)o2, (bf16)o3};
        *reinterpret_cast<bf16x4*>(out_bf16 + (size_t)row * 128 + lane * 4) = ob;
    }
    if (stat && lane == 0) { stat[2 * row] = mean; stat[2 * row + 1] = rstd; }
}

__global__ void layernorm128_bwd_kernel(const bf16* __restrict__ dy_bf16, const float* __restrict__ dy_f32,
                                        const float* __restrict__ x, const float* __restrict__ stat,
                                        const float* __restrict__ g, const float* __restrict__ dres,
                                        float* __restrict__ dx, bf16* __restrict__ dx_bf16, float* __restrict__ dgb,
                                        int M, int rows_per_half, uint32_t thresh, uint32_t seed, float inv_keep,
                                        const uint32_t* epoch) {
    seed = mm_eff_seed(seed, epoch);
    const int lane = threadIdx.x & 31;
    const int half = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    const float4 gg = *reinterpret_cast<const float4*>(g + lane * 4);
    const float gam[4] = {gg.x, gg.y, gg.z, gg.w};
    float ag[4] = {0, 0, 0, 0}, ab[4] = {0, 0, 0, 0};
    const int row0 = half * rows_per_half;
    for (int rr = 0; rr < rows_per_half; ++rr) {
        const int row = row0 + rr;
        if (row >= M) break;
        const size_t base = (size_t)row * 128 + lane * 4;
        float dyv[4];
        if (dy_bf16) {
            const bf16x4 t = *reinterpret_cast<const bf16x4*>(dy_bf16 + base);
            dyv[0] = (float)t[0]; dyv[1] = (float)t[1]; dyv[2] = (float)t[2]; dyv[3] = (float)t[3];
        } else {
            const float4 t = *reinterpret_cast<const float4*>(dy_f32 + base);
            dyv[0] = t.x; dyv[1] = t.y; dyv[2] = t.z; dyv[3] = t.w;
        }
        const float4 xv = *reinterpret_cast<const float4*>(x + base);
        float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (dres) rv = *reinterpret_cast<const float4*>(dres + base);
        const float mean = stat[2 * row], rstd = stat[2 * row + 1];
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, rs[4] = {rv.x, rv.y, rv.z, rv.w};
        float xh[4], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            xh[c] = (xs[c] - mean) * rstd;
            const float gh = dyv[c] * gam[c];
            s1 += gh; s2 += gh * xh[c];
            ag[c] += dyv[c] * xh[c]; ab[c] += dyv[c];
        }
        s1 = half_sum(s1) * (1.f / 128.f);
        s2 = half_sum(s2) * (1.f / 128.f);
        float o[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = rstd * (dyv[c] * gam[c] - s1 - xh[c] * s2) + rs[c];
        if (dx) *reinterpret_cast<float4*>(dx + base) = make_float4(o[0], o[1], o[2], o[3]);
        if (dx_bf16) {
            bf16x4 ob;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                ob[c] = (bf16)(thresh ? o[c] * dropout_scale(seed, (uint32_t)(base + c), thresh, inv_keep) : o[c]);
            *reinterpret_cast<bf16x4*>(dx_bf16 + base) = ob;
        }
    }
    if (dgb) {                                       // 8 half-waves: plain stores + column walk, no LDS atomics
        __shared__ __attribute__((aligned(16))) float part[8][256];
        const int hw = threadIdx.x >> 5;
        *reinterpret_cast<float4*>(&part[hw][lane * 4]) = make_float4(ag[0], ag[1], ag[2], ag[3]);
        *reinterpret_cast<float4*>(&part[hw][128 + lane * 4]) = make_float4(ab[0], ab[1], ab[2], ab[3]);
        __syncthreads();
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) s += part[r][threadIdx.x];
        acc_add<MM_ACC_GRAD>(acc_rep(dgb, blockIdx.x % MM_ACC_REPL, 256) + threadIdx.x, s);
    }
}

// column sums of a bf16/fp32 [M][N] matrix into fp32 [N] (atomics; bias grads)
__global__ void colsum_kernel(const bf16* __restrict__ a_bf16, const float* __restrict__ a_f32,
                              float* __restrict__ out, int M, int N, int rows_per_blk) {
    const int m0 = blockIdx.x * rows_per_blk;
    const int m1 = min(M, m0 + rows_per_blk);
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        float s = 0.f;
        for (int m = m0; m < m1; ++m) s += a_bf16 ? (float)a_bf16[(size_t)m * N + n] : a_f32[(size_t)m * N + n];
        acc_add<MM_ACC_GRAD>(acc_rep(out, blockIdx.x % MM_ACC_REPL, N) + n, s);
    }
}

// mean over L of fp32 [B][L][D] -> [B][D] fp32 + bf16
// 1024 threads per (b, 64-column group): 16 waves stride over L with 8 loads in flight each
__global__ __launch_bounds__(1024) void meanpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ out_f32,
                                                            bf16* __restrict__ out_bf16, int L, int D) {
    const int b = blockIdx.x;
    const int d = blockIdx.y * 64 + (threadIdx.x & 63);
    const int part = threadIdx.x >> 6, parts = blockDim.x >> 6;
    __shared__ float red[16][64];
    float s = 0.f;
    if (d < D) {
        const float* p = x + (size_t)b * L * D + d;
        int l = part;
        for (; l + 7 * parts < L; l += 8 * parts) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = p[(size_t)(l + i * parts) * D];
#pragma unroll
            for (int i = 0; i < 8; ++i) s += v[i];
        }
        for (; l < L; l += parts) s += p[(size_t)l * D];
    }
    red[part][threadIdx.x & 63] = s;
    __syncthreads();
    if (part == 0 && d < D) {
        float t = 0.f;
        for (int p = 0; p < parts; ++p) t += red[p][threadIdx.x & 63];
        t /= (float)L;
        if (out_f32) out_f32[(size_t)b * D + d] = t;
        if (out_bf16) out_bf16[(size_t)b * D + d] = (bf16)t;
    }
}

// dx[b,l,d] = g[b,d] / L  (fp32)
__global__ void meanpool_bwd_kernel(const float* __restrict__ g, float* __restrict__ dx, int B, int L, int D) {
    const size_t total = (size_t)B * L * D;
    const float inv = 1.f / (float)L;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int d = (int)(i % D);
        const size_t b = i / ((size_t)L * D);
        dx[i] = g[b * D + d] * inv;
    }
}

__global__ void cast_bf16_kernel(const float* __restrict__ x, bf16* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = (bf16)x[i];
}
__global__ void cast_f32_kernel(const bf16* __restrict__ x, float* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = (float)x[i];
}

// dz = g * dropout_mask * act'(z): elementwise gradient through act+dropout
__global__ void act_bwd_kernel(const float* __restrict__ g_f32, const bf16* __restrict__ g_bf16,
                               const bf16* __restrict__ z, bf16* __restrict__ out, size_t n, int act,
                               uint32_t thresh, uint32_t seed, float inv_keep, const uint32_t* epoch) {
    seed = mm_eff_seed(seed, epoch);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float g = g_f32 ? g_f32[i] : (float)g_bf16[i];
        if (thresh) g *= dropout_scale(seed, (uint32_t)i, thresh, inv_keep);
        if (z) g *= act_grad((float)z[i], act);
        out[i] = (bf16)g;
    }
}

// PositionalEncoding.forward as a stand-alone op: out[b][l][d] = (x[b][l][d] + pe[l][d]) * dropout_mask.
// pe == nullptr is its backward (dx = dout * the same mask).  float4 per thread (D % 4 == 0).
__global__ void add_pe_kernel(const float* __restrict__ x, const float* __restrict__ pe, float* __restrict__ out_f32,
                              bf16* __restrict__ out_bf16, size_t n4, int LD4, uint32_t thresh, uint32_t seed,
                              float inv_keep, const uint32_t* epoch) {
    seed = mm_eff_seed(seed, epoch);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = reinterpret_cast<const float4*>(x)[i];
        if (pe) {
            const float4 t = reinterpret_cast<const float4*>(pe)[i % (size_t)LD4];
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        if (thresh) {
            const uint32_t e = (uint32_t)(i * 4);
            v.x *= dropout_scale(seed, e, thresh, inv_keep);
            v.y *= dropout_scale(seed, e + 1, thresh, inv_keep);
            v.z *= dropout_scale(seed, e + 2, thresh, inv_keep);
            v.w *= dropout_scale(seed, e + 3, thresh, inv_keep);
        }
        if (out_f32) reinterpret_cast<float4*>(out_f32)[i] = v;
        if (out_bf16) {
            bf16x4 o = {(bf16)v.x, (bf16)v.y, (bf16)v.z, (bf16)v.w};
            reinterpret_cast<bf16x4*>(out_bf16)[i] = o;
        }
    }
}

inline int grid_for(size_t n, int block = 256, int cap = 4096) {
    size_t g = (n + block - 1) / block;
    return (int)(g < (size_t)cap ? (g ? g : 1) : cap);
}
inline uint32_t thresh_of(float p) { return p > 0.f ? (uint32_t)((double)p * 4294967296.0) : 0u; }

}  // namespace

extern "C" {

int mm_bn_finalize(const float* stats, const float* gamma, const float* beta, float* run_mean, float* run_var,
                   const float* conv_bias, float* out4, int N, float count, float momentum, float eps, int mode,
                   void* batches_tracked, hipStream_t st) {
    MM_REQUIRE(gamma && beta && run_mean && run_var && out4 && N > 0, "bn_finalize: null/invalid");
    MM_REQUIRE(mode == 1 || (stats && count >= 1.f), "bn_finalize: train mode needs stats");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(ceil_div(N, 128)), dim3(128), 0, st, stats, gamma, beta, run_mean,
                       run_var, conv_bias, out4, N, count, momentum, eps, mode, (long long*)batches_tracked);
    return mm_check_launch("bn_finalize");
}

static int bn_act_fwd_common(const float* y, const float* scale, const float* shift, const float* pe, void* out_bf16,
                             float* out_f32, int R, int S, int N, int act, int pool, int drop_first, float drop_p,
                             uint32_t seed, float drop2_p, uint32_t seed2, const uint32_t* seed_epoch,
                             const float* ln_gamma, const float* ln_beta, float ln_eps, void* ln_out, float* ln_stat,
                             hipStream_t st, const void* bn_fin_host = nullptr) {
    MmBnFin fin{};
    const bool with_fin = bn_fin_host != nullptr;
    if (with_fin) {
        MM_REQUIRE(bn_fin_from_host(fin, bn_fin_host, N), "bn_act_fwd_fin: incomplete mm_bn_fin_t (null pointer or count < 1)");
        MM_REQUIRE(N <= 256 && act == MM_ACT_GELU, "bn_act_fwd_fin: N=%d (<= 256), GELU only", N);
        scale = fin.out4; shift = fin.out4 + N;           // (what the non-fused form would read; unused by the FIN kernels)
    }
    MM_REQUIRE(y && scale && shift && (out_bf16 || out_f32), "bn_act_fwd: null");
    MM_REQUIRE(N % 4 == 0 && (pool == 1 || (pool == 2 && S % 2 == 0)), "bn_act_fwd: N%%4, pool");
    BnActArgs a{y, scale, shift, pe, (bf16*)out_bf16, out_f32, R, S, N, act, pool, drop_first,
                thresh_of(drop_p), seed, drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f,
                thresh_of(drop2_p), seed2, drop2_p > 0.f ? 1.f / (1.f - drop2_p) : 1.f, seed_epoch};
    a.ln_gamma = ln_gamma; a.ln_beta = ln_beta; a.ln_eps = ln_eps; a.ln_out = (bf16*)ln_out; a.ln_stat = ln_stat;
    a.fin = fin;
    const size_t total = (size_t)R * (S / pool) * (N / 4);
    MM_REQUIRE(total < (1ull << 31), "bn_act_fwd: %zu vectors (32-bit indices)", total);
    // FIN: every workgroup re-reads the statistics workspace (256 N bytes) in its prologue: few, longer workgroups
    // with the finalize in the prologue every workgroup re-reads the statistics workspace, so few and longer ones:
    // three per CU (256 / 512 / 768 / 1 024 / 2 048: 0.788 / 0.770 / 0.766 / 0.771 / 0.770 ms per step, profiles/r04_fin_fold_and_third_stream_ab.txt)
    int grid = grid_for(total, 256, with_fin ? 768 : 4096);
    if (ln_out) {
        MM_REQUIRE(N == 128 && pool == 1 && ln_gamma && ln_beta, "bn_act_fwd_ln: N=%d (128) pool=%d (1)", N, pool);
        // every 32-lane group must walk its rows together (cross-lane sums): total is a multiple of 32; a grid that
        // covers it in whole passes keeps the loop trip count uniform inside a group
        const size_t blocks = (total + 255) / 256;
        if ((size_t)grid > blocks) grid = (int)blocks;
    }
    if (with_fin && pool == 1) hipLaunchKernelGGL((bn_act_fwd_kernel<MM_ACT_GELU, 1, true>), dim3(grid), dim3(256), 0, st, a);
    else if (with_fin) hipLaunchKernelGGL((bn_act_fwd_kernel<MM_ACT_GELU, 2, true>), dim3(grid), dim3(256), 0, st, a);
    else if (act == MM_ACT_GELU && pool == 1) hipLaunchKernelGGL((bn_act_fwd_kernel<MM_ACT_GELU, 1>), dim3(grid), dim3(256), 0, st, a);
    else if (act == MM_ACT_GELU && pool == 2) hipLaunchKernelGGL((bn_act_fwd_kernel<MM_ACT_GELU, 2>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((bn_act_fwd_kernel<>), dim3(grid), dim3(256), 0, st, a);
    return mm_check_launch("bn_act_fwd");
}

int mm_bn_act_fwd(const float* y, const float* scale, const float* shift, const float* pe, void* out_bf16,
                  float* out_f32, int R, int S, int N, int act, int pool, int drop_first, float drop_p,
                  uint32_t seed, float drop2_p, uint32_t seed2, const uint32_t* seed_epoch, hipStream_t st) {
    return bn_act_fwd_common(y, scale, shift, pe, out_bf16, out_f32, R, S, N, act, pool, drop_first, drop_p, seed, drop2_p,
                             seed2, seed_epoch, nullptr, nullptr, 0.f, nullptr, nullptr, st);
}

int mm_bn_act_fwd_fin(const float* y, const void* bn_fin_host, const float* pe, void* out_bf16, float* out_f32, int R, int S,
                      int N, int act, int pool, int drop_first, float drop_p, uint32_t seed, float drop2_p, uint32_t seed2,
                      const uint32_t* seed_epoch, hipStream_t st) {
    MM_REQUIRE(bn_fin_host, "bn_act_fwd_fin: null descriptor");
    return bn_act_fwd_common(y, nullptr, nullptr, pe, out_bf16, out_f32, R, S, N, act, pool, drop_first, drop_p, seed, drop2_p,
                             seed2, seed_epoch, nullptr, nullptr, 0.f, nullptr, nullptr, st, bn_fin_host);
}

int mm_bn_act_fwd_ln_fin(const float* y, const void* bn_fin_host, const float* pe, float* out_f32, int R, int S, int act,
                         float drop_p, uint32_t seed, float drop2_p, uint32_t seed2, const uint32_t* seed_epoch,
                         const float* ln_gamma, const float* ln_beta, float ln_eps, void* ln_out_bf16, float* ln_stat,
                         hipStream_t st) {
    MM_REQUIRE(bn_fin_host && out_f32 && ln_out_bf16, "bn_act_fwd_ln_fin: null descriptor / output");
    return bn_act_fwd_common(y, nullptr, nullptr, pe, nullptr, out_f32, R, S, 128, act, 1, 1, drop_p, seed, drop2_p, seed2,
                             seed_epoch, ln_gamma, ln_beta, ln_eps, ln_out_bf16, ln_stat, st, bn_fin_host);
}

int mm_bn_act_fwd_ln(const float* y, const float* scale, const float* shift, const float* pe, float* out_f32, int R, int S,
                     int act, float drop_p, uint32_t seed, float drop2_p, uint32_t seed2, const uint32_t* seed_epoch,
                     const float* ln_gamma, const float* ln_beta, float ln_eps, void* ln_out_bf16, float* ln_stat,
                     hipStream_t st) {
    MM_REQUIRE(out_f32 && ln_out_bf16, "bn_act_fwd_ln: null output");
    return bn_act_fwd_common(y, scale, shift, pe, nullptr, out_f32, R, S, 128, act, 1, 1, drop_p, seed, drop2_p, seed2,
                             seed_epoch, ln_gamma, ln_beta, ln_eps, ln_out_bf16, ln_stat, st);
}

static int bn_bwd_common(bool apply, const float* y, const float* out4, const void* dout_bf16, const float* dout_f32,
                         const float* sums_in, float* sums_out, void* dy, float* dy_f32, int R, int S, int N, int act,
                         int pool, int drop_first, float drop_p, uint32_t seed, float drop2_p, uint32_t seed2,
                         const uint32_t* seed_epoch, int train, int sums_nrep, hipStream_t st, int bcast = 0,
                         float bcast_scale = 1.f) {
    MM_REQUIRE(y && out4 && (dout_bf16 || dout_f32), "bn_act_bwd: null");
    MM_REQUIRE(!bcast || (dout_f32 && pool == 1 && (size_t)R * S < (1ull << 32)), "bn_act_bwd: broadcast d(out) needs fp32 rows and pool 1");
    MM_REQUIRE(sums_nrep == 1 || sums_nrep == MM_REPL, "bn_act_bwd: sums_nrep = 1 (compact fp32) or %d (the reduce pass's workspace)", MM_REPL);
    MM_REQUIRE(N % 4 == 0 && N <= 1024 && (N / 4) <= 256, "bn_act_bwd: N");
    BnBwdArgs a;
    a.y = y; a.scale = out4; a.shift = out4 + N; a.mean = out4 + 2 * N; a.rstd = out4 + 3 * N;
    a.dout_bf16 = (const bf16*)dout_bf16; a.dout_f32 = dout_f32; a.sums = sums_in; a.sums_out = sums_out;
    a.dy = (bf16*)dy; a.dy_f32 = dy_f32; a.R = R; a.S = S; a.N = N; a.act = act; a.pool = pool; a.drop_first = drop_first;
    a.train = train; a.thresh = thresh_of(drop_p); a.seed = seed;
    a.inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    a.inv_count = 1.f / ((float)R * (float)S);
    a.thresh2 = thresh_of(drop2_p); a.seed2 = seed2; a.inv_keep2 = drop2_p > 0.f ? 1.f / (1.f - drop2_p) : 1.f;
    a.epoch = seed_epoch;
    a.sums_nrep = sums_nrep;
    a.bcast = bcast; a.bcast_scale = bcast_scale;
    const int rpb = 256 / (N / 4) > 0 ? 256 / (N / 4) : 1;
    const size_t rows = (size_t)R * (S / pool);
    int grid = (int)((rows + rpb - 1) / rpb);
    { static const int cap = getenv("MM_BN_GRID") ? atoi(getenv("MM_BN_GRID")) : 768; if (grid > cap) grid = cap; }   // three workgroups per CU (sweep 256..1024: profiles/r03_second_half_ab.txt)
    // GELU with pool 1 / 2 (every BatchNorm of the encoders on the training path) is compiled in; anything else is generic
    if (act == MM_ACT_GELU && pool == 1) {
        if (apply) hipLaunchKernelGGL((bn_act_bwd_kernel<true, MM_ACT_GELU, 1>), dim3(grid), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((bn_act_bwd_kernel<false, MM_ACT_GELU, 1>), dim3(grid), dim3(256), 0, st, a);
    } else if (act == MM_ACT_GELU && pool == 2) {
        if (apply) hipLaunchKernelGGL((bn_act_bwd_kernel<true, MM_ACT_GELU, 2>), dim3(grid), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((bn_act_bwd_kernel<false, MM_ACT_GELU, 2>), dim3(grid), dim3(256), 0, st, a);
    } else if (apply) hipLaunchKernelGGL((bn_act_bwd_kernel<true>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((bn_act_bwd_kernel<false>), dim3(grid), dim3(256), 0, st, a);
    return mm_check_launch("bn_act_bwd");
}

int mm_bn_act_bwd_reduce(const float* y, const float* out4, const void* dout_bf16, const float* dout_f32,
                         float* sums_out, int R, int S, int N, int act, int pool, int drop_first, float drop_p,
                         uint32_t seed, float drop2_p, uint32_t seed2, const uint32_t* seed_epoch, hipStream_t st) {
    MM_REQUIRE(sums_out, "bn_act_bwd_reduce: null sums");
    return bn_bwd_common(false, y, out4, dout_bf16, dout_f32, nullptr, sums_out, nullptr, nullptr, R, S, N, act,
                         pool, drop_first, drop_p, seed, drop2_p, seed2, seed_epoch, 1, 1, st);
}

int mm_bn_act_bwd_apply(const float* y, const float* out4, const void* dout_bf16, const float* dout_f32,
                        const float* sums, void* dy, float* dy_f32, int R, int S, int N, int act, int pool,
                        int drop_first, float drop_p, uint32_t seed, float drop2_p, uint32_t seed2,
                        const uint32_t* seed_epoch, int train, int sums_nrep, hipStream_t st) {
    MM_REQUIRE((dy || dy_f32) && (!train || sums), "bn_act_bwd_apply: null");
    return bn_bwd_common(true, y, out4, dout_bf16, dout_f32, sums, nullptr, dy, dy_f32, R, S, N, act, pool,
                         drop_first, drop_p, seed, drop2_p, seed2, seed_epoch, train, sums_nrep, st);
}

// The two passes for a block whose d(out) is the backward of a mean over its S positions (AdaptiveAvgPool -> Linear head
// on top of the last conv block of the voxel encoder): dout_rows (R, N) fp32 holds ONE row per sample, every position's
// d(out) is dout_rows[r] * scale (scale = 1 / S).  The (R, S, N) broadcast tensor is never written or read.
int mm_bn_act_bwd_reduce_bcast(const float* y, const float* out4, const float* dout_rows, float scale, float* sums_out, int R,
                               int S, int N, int act, float drop_p, uint32_t seed, const uint32_t* seed_epoch, hipStream_t st) {
    MM_REQUIRE(sums_out && dout_rows, "bn_act_bwd_reduce_bcast: null");
    return bn_bwd_common(false, y, out4, nullptr, dout_rows, nullptr, sums_out, nullptr, nullptr, R, S, N, act, 1, 1, drop_p, seed,
                         0.f, 0u, seed_epoch, 1, 1, st, 1, scale);
}

int mm_bn_act_bwd_apply_bcast(const float* y, const float* out4, const float* dout_rows, float scale, const float* sums,
                              void* dy, int R, int S, int N, int act, float drop_p, uint32_t seed, const uint32_t* seed_epoch,
                              int train, int sums_nrep, hipStream_t st) {
    MM_REQUIRE(dy && dout_rows && (!train || sums), "bn_act_bwd_apply_bcast: null");
    return bn_bwd_common(true, y, out4, nullptr, dout_rows, sums, nullptr, dy, nullptr, R, S, N, act, 1, 1, drop_p, seed, 0.f, 0u,
                         seed_epoch, train, sums_nrep, st, 1, scale);
}

#define LN_DISPATCH(D, CALL)                                   \
    switch ((D) / 64) {                                        \
        case 1: { constexpr int V = 1; CALL; } break;          \
        case 2: { constexpr int V = 2; CALL; } break;          \
        case 4: { constexpr int V = 4; CALL; } break;          \
        case 8: { constexpr int V = 8; CALL; } break;          \
        case 16: { constexpr int V = 16; CALL; } break;        \
        default: return mm_fail(MM_ERR_UNSUPPORTED, "layernorm: D=%d (need 64,128,256,512,1024)", (D)); \
    }

int mm_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* out_bf16, float* out_f32,
                     float* stat, int M, int D, float eps, hipStream_t st) {
    MM_REQUIRE(x && gamma && beta && (out_bf16 || out_f32) && M > 0, "layernorm_fwd: null");
    if (D == 128) {
        hipLaunchKernelGGL(layernorm128_fwd_kernel, dim3(ceil_div(M, 8)), dim3(256), 0, st, x, gamma, beta,
                           (bf16*)out_bf16, out_f32, stat, M, eps);
        return mm_check_launch("layernorm128_fwd");
    }
    const dim3 grid(ceil_div(M, 4)), block(256);
    LN_DISPATCH(D, hipLaunchKernelGGL(layernorm_fwd_kernel<V>, grid, block, 0, st, x, gamma, beta, (bf16*)out_bf16,
                                      out_f32, stat, M, eps));
    return mm_check_launch("layernorm_fwd");
}

int mm_layernorm_bwd(const void* dy_bf16, const float* dy_f32, const float* x, const float* stat, const float* gamma,
                     const float* dres, float* dx, void* dx_bf16, float* dgb_repl, int M, int D, float drop_p,
                     uint32_t seed, const uint32_t* seed_epoch, hipStream_t st) {
    MM_REQUIRE((dy_bf16 || dy_f32) && x && stat && gamma && (dx || dx_bf16), "layernorm_bwd: null");
    if (D == 128) {
        const int rph = M >= 8192 ? 4 : 1;                 // rows per half-wave
        hipLaunchKernelGGL(layernorm128_bwd_kernel, dim3(ceil_div(M, 8 * rph)), dim3(256), 0, st, (const bf16*)dy_bf16,
                           dy_f32, x, stat, gamma, dres, dx, (bf16*)dx_bf16, dgb_repl, M, rph, thresh_of(drop_p), seed,
                           drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f, seed_epoch);
        return mm_check_launch("layernorm128_bwd");
    }
    const int rpw = M >= 8192 ? 8 : (M >= 1024 ? 2 : 1);
    const dim3 grid(ceil_div(M, 4 * rpw)), block(256);
    LN_DISPATCH(D, hipLaunchKernelGGL(layernorm_bwd_kernel<V>, grid, block, 0, st, (const bf16*)dy_bf16, dy_f32, x,
                                      stat, gamma, dres, dx, (bf16*)dx_bf16, dgb_repl, M, rpw, thresh_of(drop_p), seed,
                                      drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f, seed_epoch));
    return mm_check_launch("layernorm_bwd");
}

int mm_colsum(const void* a_bf16, const float* a_f32, float* out, int M, int N, hipStream_t st) {
    MM_REQUIRE((a_bf16 || a_f32) && out && M > 0 && N > 0, "colsum: null");
    const int rpb = M >= 4096 ? 64 : 16;
    hipLaunchKernelGGL(colsum_kernel, dim3(ceil_div(M, rpb)), dim3(N >= 256 ? 256 : (N >= 128 ? 128 : 64)), 0, st,
                       (const bf16*)a_bf16, a_f32, out, M, N, rpb);
    return mm_check_launch("colsum");
}

int mm_meanpool_fwd(const float* x, float* out_f32, void* out_bf16, int B, int L, int D, hipStream_t st) {
    MM_REQUIRE(x && (out_f32 || out_bf16) && B > 0 && L > 0 && D > 0, "meanpool_fwd: null");
    hipLaunchKernelGGL(meanpool_fwd_kernel, dim3(B, ceil_div(D, 64)), dim3(1024), 0, st, x, out_f32, (bf16*)out_bf16, L, D);
    return mm_check_launch("meanpool_fwd");
}

int mm_meanpool_bwd(const float* g, float* dx, int B, int L, int D, hipStream_t st) {
    MM_REQUIRE(g && dx, "meanpool_bwd: null");
    hipLaunchKernelGGL(meanpool_bwd_kernel, dim3(grid_for((size_t)B * L * D)), dim3(256), 0, st, g, dx, B, L, D);
    return mm_check_launch("meanpool_bwd");
}

int mm_cast_bf16(const float* x, void* y, int64_t n, hipStream_t st) {
    MM_REQUIRE(x && y && n > 0, "cast_bf16: null");
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid_for((size_t)n)), dim3(256), 0, st, x, (bf16*)y, (size_t)n);
    return mm_check_launch("cast_bf16");
}

int mm_cast_f32(const void* x, float* y, int64_t n, hipStream_t st) {
    MM_REQUIRE(x && y && n > 0, "cast_f32: null");
    hipLaunchKernelGGL(cast_f32_kernel, dim3(grid_for((size_t)n)), dim3(256), 0, st, (const bf16*)x, y, (size_t)n);
    return mm_check_launch("cast_f32");
}

int mm_act_bwd(const float* g_f32, const void* g_bf16, const void* z, void* out, int64_t n, int act, float drop_p,
               uint32_t seed, const uint32_t* seed_epoch, hipStream_t st) {
    MM_REQUIRE((g_f32 || g_bf16) && out && n > 0, "act_bwd: null");
    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for((size_t)n)), dim3(256), 0, st, g_f32, (const bf16*)g_bf16,
                       (const bf16*)z, (bf16*)out, (size_t)n, act, thresh_of(drop_p), seed,
                       drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f, seed_epoch);
    return mm_check_launch("act_bwd");
}

int mm_add_pe(const float* x, const float* pe, float* out_f32, void* out_bf16, int B, int L, int D, float drop_p,
              uint32_t seed, const uint32_t* seed_epoch, hipStream_t st) {
    MM_REQUIRE(x && (out_f32 || out_bf16) && B > 0 && L > 0 && D > 0, "add_pe: null/invalid");
    MM_REQUIRE(D % 4 == 0, "add_pe: D=%d must be a multiple of 4", D);
    MM_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "add_pe: drop_p");
    const size_t n4 = (size_t)B * L * D / 4;
    hipLaunchKernelGGL(add_pe_kernel, dim3(grid_for(n4)), dim3(256), 0, st, x, pe, out_f32, (bf16*)out_bf16, n4,
                       L * D / 4, thresh_of(drop_p), seed, drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f, seed_epoch);
    return mm_check_launch("add_pe");
}

}  // extern "C"
