// Small fp32 row kernels for the projection bridge and the tabular models:
// dense layer (fwd/bwd), elementwise act, L2-normalise, column statistics, and
// the fused batch-pairwise cosine-similarity / symmetric InfoNCE loss with its
// gradients.  Batches here are tens of rows: these kernels are latency-bound,
// kept in fp32 end-to-end (they sit right before the parity-checked outputs)
// and use wave-level shuffles for every row reduction.
#include "common.h"

namespace {

// y[b][n] = dropout(act(x[b][:] . W[n][:] + bias[n])); one wave per (b, 64 outputs)
__global__ void small_linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                        const float* __restrict__ bias, const float* __restrict__ scale,
                                        const float* __restrict__ shift, float* __restrict__ y,
                                        float* __restrict__ pre, int B, int K, int N, int act,
                                        uint32_t thresh, uint32_t seed, float inv_keep, const uint32_t* epoch) {
    seed = mm_eff_seed(seed, epoch);
    extern __shared__ float xs[];                   // one input row
    const int b = blockIdx.x;
    for (int k = threadIdx.x; k < K; k += blockDim.x) xs[k] = x[(size_t)b * K + k];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int n = blockIdx.y * nw + wave; n < N; n += gridDim.y * nw) {
        float s = 0.f;
        for (int k = lane; k < K; k += 64) s += xs[k] * W[(size_t)n * K + k];
        s = wave_sum(s);
        if (lane == 0) {
            s += bias ? bias[n] : 0.f;
            if (scale) s = s * scale[n] + shift[n];
            const size_t idx = (size_t)b * N + n;
            if (pre) pre[idx] = s;
            s = apply_act(s, act);
            if (thresh) s *= dropout_scale(seed, (uint32_t)idx, thresh, inv_keep);
            y[idx] = s;
        }
    }
}

// dx[b][k] = sum_n dy[b][n] W[n][k]           (grid.x = B, threads over k)
__global__ void small_linear_dx_kernel(const float* __restrict__ dy, const float* __restrict__ W,
                                       float* __restrict__ dx, int B, int K, int N) {
    extern __shared__ float ds[];
    const int b = blockIdx.x;
    for (int n = threadIdx.x; n < N; n += blockDim.x) ds[n] = dy[(size_t)b * N + n];
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += ds[n] * W[(size_t)n * K + k];
        dx[(size_t)b * K + k] = s;
    }
}

// dW[n][k] += sum_b dy[b][n] x[b][k];  db[n] += sum_b dy[b][n]   (unique owner per element)
__global__ void small_linear_dw_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                       float* __restrict__ dW, float* __restrict__ db, int B, int K, int N) {
    const int n = blockIdx.x;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dy[(size_t)b * N + n] * x[(size_t)b * K + k];
        dW[(size_t)n * K + k] += s;
    }
    if (db && threadIdx.x == 0) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dy[(size_t)b * N + n];
        db[n] += s;
    }
}

__global__ void act_f32_kernel(const float* __restrict__ z, float* __restrict__ y, size_t n, int act,
                               uint32_t thresh, uint32_t seed, float inv_keep, const uint32_t* epoch) {
    seed = mm_eff_seed(seed, epoch);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float v = apply_act(z[i], act);
        if (thresh) v *= dropout_scale(seed, (uint32_t)i, thresh, inv_keep);
        y[i] = v;
    }
}
__global__ void act_bwd_f32_kernel(const float* __restrict__ g, const float* __restrict__ z, float* __restrict__ out,
                                   size_t n, int act, uint32_t thresh, uint32_t seed, float inv_keep,
                                   const uint32_t* epoch) {
    seed = mm_eff_seed(seed, epoch);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float v = g[i];
        if (thresh) v *= dropout_scale(seed, (uint32_t)i, thresh, inv_keep);
        if (z) v *= act_grad(z[i], act);
        out[i] = v;
    }
}

// column sum / sum-of-squares of fp32 [B][N]  (BatchNorm1d over a (B, N) batch)
__global__ void colstats_kernel(const float* __restrict__ x, float* __restrict__ stats, int B, int N) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float s = 0.f, q = 0.f;
    for (int b = 0; b < B; ++b) { const float v = x[(size_t)b * N + n]; s += v; q += v * v; }
    // replica 0 of a statistics accumulator workspace (the input of mm_bn_finalize); the other replicas stay zero
    mm_acc_t* acc = reinterpret_cast<mm_acc_t*>(stats);
    acc[n] = acc_encode<MM_ACC_STAT>(s);
    acc[N + n] = acc_encode<MM_ACC_STAT>(q);
}

// z = h / max(||h||, eps); one wave per row
__global__ void l2norm_fwd_kernel(const float* __restrict__ h, float* __restrict__ z, float* __restrict__ nrm, int B, int N, int ldz) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= B) return;
    float q = 0.f;
    for (int n = lane; n < N; n += 64) { const float v = h[(size_t)row * N + n]; q += v * v; }
    const float nr = fmaxf(sqrtf(wave_sum(q)), 1e-12f);
    for (int n = lane; n < N; n += 64) z[(size_t)row * ldz + n] = h[(size_t)row * N + n] / nr;
    if (lane == 0) nrm[row] = nr;
}
// dh = (dz - z (z . dz)) / ||h||
__global__ void l2norm_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ z, const float* __restrict__ nrm,
                                  float* __restrict__ dh, int B, int N, int ldz) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= B) return;
    float d = 0.f;
    for (int n = lane; n < N; n += 64) d += dz[(size_t)row * ldz + n] * z[(size_t)row * ldz + n];
    d = wave_sum(d);
    const float inv = 1.f / nrm[row];
    for (int n = lane; n < N; n += 64)
        dh[(size_t)row * N + n] = (dz[(size_t)row * ldz + n] - z[(size_t)row * ldz + n] * d) * inv;
}


// ---------------------------------------------------------------------------
// Both projection heads of the contrastive bridge in ONE launch each way
// (bridge_utils.py:34-45: Linear -> LayerNorm -> GELU -> Dropout, then F.normalize).
// At B = 32 the eleven separate launches of this chain each way were ~130 us of
// pure launch latency on the step's critical path.  grid = (B rows, 2 heads).
// ---------------------------------------------------------------------------
struct HeadSide {
    const float* x; const float* W; const float* bias; const float* gamma; const float* beta;
    float* dx; float* dW; float* dbias; float* dgamma; float* dbeta;
    int K; uint32_t seed;
};
struct HeadsArgs {
    HeadSide s[2];
    float* z1; float* hn; float* stat;        // [2][B][N], [2][B][N], [2][B][2]   saved for backward
    float* z; float* nrm;                     // [B][2N] packed embeddings, [2][B]
    const float* dz;                          // [B][2N]
    int B, N; float eps; uint32_t thresh; float inv_keep; const uint32_t* epoch;
};
constexpr int HEAD_MAXK = 1024, HEAD_MAXN = 256;

__device__ __forceinline__ float block_sum256(float v, float* red /* [4] */) {
    v = wave_sum(v);
    __syncthreads();                          // red may still be read from the previous call
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void proj_heads_fwd_kernel(HeadsArgs a) {
    __shared__ float xs[HEAD_MAXK], v[HEAD_MAXN], red[4];
    const int b = blockIdx.x, m = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const HeadSide s = a.s[m];
    const int N = a.N, K = s.K;
    for (int k = tid; k < K; k += 256) xs[k] = s.x[(size_t)b * K + k];
    __syncthreads();
    // T = 256 / N lanes share one output; each issues its float4 loads of the weight row back to
    // back (a wave-per-output loop waits out one global-load round trip per output: 36 us at N = 128)
    const int T = N <= 64 ? 4 : (N <= 128 ? 2 : 1);
    {
        const int n = tid / T, part = tid % T;
        float acc = 0.f;
        if (n < N) {
            const float* wr = s.W + (size_t)n * K;
            if ((K & 3) == 0) {
#pragma unroll 8
                for (int k = part * 4; k < K; k += 4 * T) {
                    const float4 w4 = *reinterpret_cast<const float4*>(wr + k);
                    acc += w4.x * xs[k] + w4.y * xs[k + 1] + w4.z * xs[k + 2] + w4.w * xs[k + 3];
                }
            } else {
                for (int k = part; k < K; k += T) acc += wr[k] * xs[k];
            }
        }
        if (T >= 2) acc += __shfl_xor(acc, 1, 64);
        if (T >= 4) acc += __shfl_xor(acc, 2, 64);
        if (n < N && part == 0) v[n] = acc + (s.bias ? s.bias[n] : 0.f);
    }
    (void)lane; (void)wave;
    __syncthreads();
    const bool on = tid < N;
    const float x1 = on ? v[tid] : 0.f;
    const float mean = block_sum256(x1, red) / N;
    const float dlt = on ? x1 - mean : 0.f;
    const float rstd = rsqrtf(block_sum256(dlt * dlt, red) / N + a.eps);
    float act = 0.f, hn = 0.f;
    const size_t o = ((size_t)m * a.B + b) * N + tid;
    if (on) {
        hn = dlt * rstd * s.gamma[tid] + s.beta[tid];
        act = gelu_erf(hn);
        if (a.thresh) act *= dropout_scale(mm_eff_seed(s.seed, a.epoch), (uint32_t)(b * N + tid), a.thresh, a.inv_keep);
        if (a.z1) { a.z1[o] = x1; a.hn[o] = hn; }
    }
    const float nr = fmaxf(sqrtf(block_sum256(act * act, red)), 1e-12f);
    if (on) a.z[(size_t)b * 2 * N + m * N + tid] = act / nr;
    if (tid == 0) {
        a.nrm[m * a.B + b] = nr;
        if (a.stat) { a.stat[((size_t)m * a.B + b) * 2] = mean; a.stat[((size_t)m * a.B + b) * 2 + 1] = rstd; }
    }
}

// Backward of both heads, bit-reproducible: no gradient element has more than one writer.  grid = (B, 2 heads),
// 16 waves.  EVERY block recomputes d z1 of all rows (a wave per row, two rows in flight per wave: a few hundred
// flops each, the same bits in every block), then block j writes dx of row j and the j-th slice of dW, each element
// summed over the rows in order; block 0 adds the bias and LayerNorm-parameter gradients (per-wave partials combined
// in wave order).  NE = ceil(N / 64) elements per lane.
constexpr int HB_RC = 32;                                   // rows per LDS chunk (= 2 per wave)
template <int NE>
__global__ __launch_bounds__(1024) void proj_heads_bwd_kernel(HeadsArgs a) {
    __shared__ float d1[HB_RC][HEAD_MAXN];                  // 32 KB; re-used for the LayerNorm partials at the end
    __shared__ float px[1024];                              // dx partials [part][k]
    const int j = blockIdx.x, m = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const HeadSide s = a.s[m];
    const int N = a.N, K = s.K, B = a.B, G = gridDim.x;
    const int NK = N * K, S = (NK + G - 1) / G;
    const uint32_t seed = mm_eff_seed(s.seed, a.epoch);
    float ag[NE], ab[NE], gam[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        ag[e] = 0.f; ab[e] = 0.f;
        gam[e] = lane + 64 * e < N ? s.gamma[lane + 64 * e] : 0.f;
    }
    for (int b0 = 0; b0 < B; b0 += HB_RC) {
        const int nb = B - b0 < HB_RC ? B - b0 : HB_RC;
        __syncthreads();                                    // the previous chunk is consumed
        // rows wave and wave + 16 of the chunk: every load of both rows is issued before the first use
        float dzv[2][NE], zv[2][NE], hnv[2][NE], z1v[2][NE], mean[2], rstd[2], nr[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int r = wave + 16 * q;
            const bool row_on = r < nb;
            const int b = row_on ? b0 + r : b0;
            const size_t o = ((size_t)m * B + b) * N, oz = (size_t)b * 2 * N + m * N;
            mean[q] = a.stat[((size_t)m * B + b) * 2]; rstd[q] = a.stat[((size_t)m * B + b) * 2 + 1];
            nr[q] = a.nrm[m * B + b];
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const int n = lane + 64 * e;
                const bool on = row_on && n < N;
                dzv[q][e] = on ? a.dz[oz + n] : 0.f;
                zv[q][e] = on ? a.z[oz + n] : 0.f;
                hnv[q][e] = on ? a.hn[o + n] : 0.f;
                z1v[q][e] = on ? a.z1[o + n] : 0.f;
            }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int r = wave + 16 * q;
            if (r >= nb) continue;                          // (wave-uniform)
            const int b = b0 + r;
            float dot = 0.f;
#pragma unroll
            for (int e = 0; e < NE; ++e) dot += dzv[q][e] * zv[q][e];
            dot = wave_sum(dot);
            float gd[NE], xh[NE], s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const int n = lane + 64 * e;
                gd[e] = 0.f; xh[e] = 0.f;
                if (n < N) {
                    // F.normalize backward: da = (dz - z (z . dz)) / ||a||
                    float g = (dzv[q][e] - zv[q][e] * dot) / nr[q];
                    if (a.thresh) g *= dropout_scale(seed, (uint32_t)(b * N + n), a.thresh, a.inv_keep);
                    const float dh = g * gelu_erf_grad(hnv[q][e]);
                    xh[e] = (z1v[q][e] - mean[q]) * rstd[q];
                    gd[e] = dh * gam[e];
                    ag[e] += dh * xh[e];
                    ab[e] += dh;
                }
                s1 += gd[e]; s2 += gd[e] * xh[e];
            }
            const float m1 = wave_sum(s1) / N, m2 = wave_sum(s2) / N;
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const int n = lane + 64 * e;
                if (n < N) d1[r][n] = rstd[q] * (gd[e] - m1 - xh[e] * m2);
            }
        }
        __syncthreads();
        // dx of the rows this block owns: 1024 / K threads share one output (strided over n), partials summed in order
        if (s.dx) {
            const int nparts = 1024 / K > 0 ? 1024 / K : 1;                 // K <= 1024 (host-checked)
            for (int b = j; b < b0 + nb; b += G) {
                if (b < b0) continue;
                const int k = tid % K, part = tid / K;
                if (part < nparts) {
                    float acc = 0.f;
#pragma unroll 4
                    for (int n = part; n < N; n += nparts) acc += d1[b - b0][n] * s.W[(size_t)n * K + k];
                    px[part * K + k] = acc;
                }
                __syncthreads();
                if (tid < K) {
                    float acc = 0.f;
                    for (int q = 0; q < nparts; ++q) acc += px[q * K + tid];
                    s.dx[(size_t)b * K + tid] = acc;
                }
                __syncthreads();
            }
        }
        if (s.dW) {
            const int hi = (j + 1) * S < NK ? (j + 1) * S : NK;
            for (int i = j * S + tid; i < hi; i += 1024) {
                const int n = i / K, k = i % K;
                const float* xc = s.x + (size_t)b0 * K + k;
                float acc = 0.f;
#pragma unroll 8
                for (int r = 0; r < nb; ++r) acc += d1[r][n] * xc[(size_t)r * K];
                s.dW[i] += acc;
            }
        }
        if (j == 0 && s.dbias && tid < N) {
            float acc = 0.f;
            for (int r = 0; r < nb; ++r) acc += d1[r][tid];
            s.dbias[tid] += acc;
        }
    }
    if (j == 0 && (s.dgamma || s.dbeta)) {
        __syncthreads();
        float (*pg)[2][HEAD_MAXN] = reinterpret_cast<float (*)[2][HEAD_MAXN]>(&d1[0][0]);     // [16 waves][dgamma | dbeta][N]
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int n = lane + 64 * e;
            if (n < N) { pg[wave][0][n] = ag[e]; pg[wave][1][n] = ab[e]; }
        }
        __syncthreads();
        if (tid < 2 * N) {
            const int which = tid / N, n = tid % N;
            float acc = 0.f;
            for (int w = 0; w < 16; ++w) acc += pg[w][which][n];
            float* dst = which ? s.dbeta : s.dgamma;
            if (dst) dst[n] += acc;
        }
    }
}

// ---------------------------------------------------------------------------
// symmetric InfoNCE over the gathered batch, bit-reproducible (no float atomics).
//   C[r][j] = ze_r . zf_j  (cosines; Bg x Bg),  s = exp(logit_scale)
//   e->f problem = row softmax of s C, f->e problem = column softmax of s C
//   loss_r = 0.5 * (CE(row r, target r) + CE(column r, target r))
// This rank owns rows/columns [row0, row0 + B).  Its step loss is mean_r loss_r over its own r; the
// gradient it needs is that of the SUM over all ranks' losses w.r.t. its own embeddings (what a
// reduce-scatter of every rank's d loss_rank / d z_all would deliver; AdamW applies the 1/world):
//   dL/dC[r][j] = 0.5/B * s * (P_row[r][j] + P_col[r][j] - 2 delta_rj)
//   dze_r = sum_j dL/dC[r][j] zf_j        dzf_r = sum_j dL/dC[j][r] ze_j
// P_col[r][j] needs column j's normaliser, P_row[j][r] row j's: a global dependency, so two launches:
//   clip_lse_kernel  (one workgroup per GLOBAL row r): row r and column r of C -> their log-sum-exp,
//                     the row's loss / top-1 flags / d loss_r / d logit_scale            -> ws[6][Bg]
//   clip_rows_kernel (one workgroup per OWN row): recomputes its row and column of C, forms dL/dC from
//                     ws, writes dz[own row] with plain stores; workgroup 0 also sums the own rows'
//                     scalars in a fixed order into scal[4] = {loss, top1 e->f, top1 f->e, d/d logit_scale}.
// Every sum runs in a fixed order (lane-strided partials, xor-shuffle trees, fixed wave order).
// ---------------------------------------------------------------------------
struct ClipShared {
    float *qe, *qf, *cr, *cc, *red;
};
__device__ __forceinline__ ClipShared clip_shared(float* sm, int N, int Bg) {
    ClipShared s;
    s.qe = sm; s.qf = sm + N; s.cr = s.qf + N; s.cc = s.cr + Bg; s.red = s.cc + Bg;
    return s;
}
// cr[j] = ze_r . zf_j, cc[j] = ze_j . zf_r for all j (8 lanes per column, float4 strides); returns the
// two maxima in every thread
__device__ __forceinline__ void clip_cosines(const float* __restrict__ z_all, int r, int Bg, int N, const ClipShared& sh,
                                             float& mxr, float& mxc) {
    const int LD = 2 * N, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int n = tid; n < N; n += 256) { sh.qe[n] = z_all[(size_t)r * LD + n]; sh.qf[n] = z_all[(size_t)r * LD + N + n]; }
    __syncthreads();
    float a_mx = -INFINITY, c_mx = -INFINITY;
    const int sub = tid & 7, n32 = N & ~31;                // the float4 sweep covers whole 32-element chunks only
    for (int j = tid >> 3; j < Bg; j += 32) {
        float a = 0.f, c = 0.f;
        const float* re = z_all + (size_t)j * LD;
        const float* rf = re + N;
        for (int n = sub * 4; n < n32; n += 32) {
            const float4 vf = *reinterpret_cast<const float4*>(rf + n);
            const float4 ve = *reinterpret_cast<const float4*>(re + n);
            a += sh.qe[n] * vf.x + sh.qe[n + 1] * vf.y + sh.qe[n + 2] * vf.z + sh.qe[n + 3] * vf.w;
            c += sh.qf[n] * ve.x + sh.qf[n + 1] * ve.y + sh.qf[n + 2] * ve.z + sh.qf[n + 3] * ve.w;
        }
        for (int n = n32 + sub; n < N; n += 8) { a += sh.qe[n] * rf[n]; c += sh.qf[n] * re[n]; }   // N % 32 tail
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); c += __shfl_xor(c, o, 64); }
        if (sub == 0) { sh.cr[j] = a; sh.cc[j] = c; }
        a_mx = fmaxf(a_mx, a); c_mx = fmaxf(c_mx, c);
    }
    a_mx = wave_max(a_mx); c_mx = wave_max(c_mx);
    if (lane == 0) { sh.red[wave] = a_mx; sh.red[4 + wave] = c_mx; }
    __syncthreads();
    mxr = fmaxf(fmaxf(sh.red[0], sh.red[1]), fmaxf(sh.red[2], sh.red[3]));
    mxc = fmaxf(fmaxf(sh.red[4], sh.red[5]), fmaxf(sh.red[6], sh.red[7]));
    __syncthreads();
}
// fixed-order workgroup sum of two values (4 waves)
__device__ __forceinline__ void clip_sum2(float& a, float& c, float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    a = wave_sum(a); c = wave_sum(c);
    if (lane == 0) { red[wave] = a; red[4 + wave] = c; }
    __syncthreads();
    a = (red[0] + red[1]) + (red[2] + red[3]);
    c = (red[4] + red[5]) + (red[6] + red[7]);
    __syncthreads();
}

__global__ __launch_bounds__(256) void clip_lse_kernel(const float* __restrict__ z_all, const float* __restrict__ logit_scale,
                                                       float* __restrict__ ws, int Bg, int N) {
    extern __shared__ float sm[];
    const ClipShared sh = clip_shared(sm, N, Bg);
    const int r = blockIdx.x, tid = threadIdx.x;
    const float s = __expf(logit_scale[0]);
    float mxr, mxc;
    clip_cosines(z_all, r, Bg, N, sh, mxr, mxc);
    float se = 0.f, sf = 0.f, ee = 0.f, ef = 0.f;           // sum exp, sum exp * cos
    for (int j = tid; j < Bg; j += 256) {
        const float a = sh.cr[j], c = sh.cc[j];
        const float pa = __expf(s * (a - mxr)), pc = __expf(s * (c - mxc));
        se += pa; sf += pc; ee += pa * a; ef += pc * c;
    }
    clip_sum2(se, sf, sh.red);
    clip_sum2(ee, ef, sh.red);
    if (tid == 0) {
        const float diag = sh.cr[r];
        const float lse_r = s * mxr + __logf(se), lse_c = s * mxc + __logf(sf);
        ws[r] = lse_r;
        ws[Bg + r] = lse_c;
        ws[2 * Bg + r] = 0.5f * ((lse_r - s * diag) + (lse_c - s * sh.cc[r]));
        ws[3 * Bg + r] = diag >= mxr ? 1.f : 0.f;
        ws[4 * Bg + r] = sh.cc[r] >= mxc ? 1.f : 0.f;
        // d loss_r / d logit_scale = s * 0.5 * (E_row[cos] - cos_rr + E_col[cos] - cos_rr)
        ws[5 * Bg + r] = s * 0.5f * ((ee / se - diag) + (ef / sf - sh.cc[r]));
    }
}

__global__ __launch_bounds__(256) void clip_rows_kernel(const float* __restrict__ z_all, const float* __restrict__ logit_scale,
                                                        const float* __restrict__ ws, float* __restrict__ scal,
                                                        float* __restrict__ dz, int B, int Bg, int N, int row0) {
    extern __shared__ float sm[];
    const ClipShared sh = clip_shared(sm, N, Bg);
    const int i = blockIdx.x, gi = row0 + i, tid = threadIdx.x, LD = 2 * N;
    const float s = __expf(logit_scale[0]);
    const float invB = 1.f / (float)B;
    if (i == 0 && tid < 64) {                               // the own rows' scalars, summed in a fixed order
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int r0 = 0; r0 < B; r0 += 64) {
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = (r0 + tid < B) ? ws[(size_t)(2 + q) * Bg + row0 + r0 + tid] : 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] += wave_sum(v[q]);
        }
        if (tid < 4) scal[tid] = (tid == 0 ? acc[0] : tid == 1 ? acc[1] : tid == 2 ? acc[2] : acc[3]) * invB;
    }
    if (!dz) return;
    float mxr, mxc;
    clip_cosines(z_all, gi, Bg, N, sh, mxr, mxc);
    // dL/dC[gi][j] -> cr[j],  dL/dC[j][gi] -> cc[j]
    const float lse_rg = ws[gi], lse_cg = ws[Bg + gi];
    const float k = 0.5f * invB * s;
    for (int j = tid; j < Bg; j += 256) {
        const float a = s * sh.cr[j], c = s * sh.cc[j];
        float ga = __expf(a - lse_rg) + __expf(a - ws[Bg + j]);          // P_row[gi][j] + P_col[gi][j]
        float gc = __expf(c - ws[j]) + __expf(c - lse_cg);               // P_row[j][gi] + P_col[j][gi]
        if (j == gi) { ga -= 2.f; gc -= 2.f; }
        sh.cr[j] = k * ga; sh.cc[j] = k * gc;
    }
    __syncthreads();
    float* orow = dz + (size_t)i * LD;
    for (int n = tid; n < 2 * N; n += 256) {                // first half: dze (columns of zf), second half: dzf
        const bool first = n < N;
        const float* g = first ? sh.cr : sh.cc;
        const float* col = z_all + (first ? N + n : n - N);
        float acc = 0.f;
        int j = 0;
        for (; j + 8 <= Bg; j += 8) {                       // 8 loads in flight, summed in order
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = col[(size_t)(j + q) * LD];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc += g[j + q] * v[q];
        }
        for (; j < Bg; ++j) acc += g[j] * col[(size_t)j * LD];
        orow[n] = acc;
    }
}

// ---------------------------------------------------------------------------
// fused AdamW (decoupled weight decay) + global-norm clip on a flat fp32 bucket.
// state[0] = step counter (float, incremented on device so the launch can live
// in a hipGraph), state[2] = learning rate (host-updatable), state[3] = last clip
// coefficient, state[4] = last gradient norm, state[8 .. 8+1024) = per-block partial
// sums of squared gradients.  The partials are summed in a FIXED order (no float
// atomics), so data-parallel ranks holding the same all-reduced gradient compute
// bit-identical clip coefficients and their parameters never drift apart.
// ---------------------------------------------------------------------------
constexpr int SUMSQ_SLOTS = 1024;

__global__ void sumsq_kernel(const float* __restrict__ g, float* __restrict__ state, size_t n) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += g[i] * g[i];
    s = wave_sum(s);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) state[8 + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    if (blockIdx.x == 0)
        for (int i = gridDim.x + threadIdx.x; i < SUMSQ_SLOTS; i += blockDim.x) state[8 + i] = 0.f;
}

// every 256-thread block gets the same total, added in the same order
__device__ inline float sumsq_total(const float* __restrict__ state) {
    __shared__ float tot[4];
    const int t = threadIdx.x;
    float s = (state[8 + t] + state[8 + 256 + t]) + (state[8 + 512 + t] + state[8 + 768 + t]);
    s = wave_sum(s);
    if ((t & 63) == 0) tot[t >> 6] = s;
    __syncthreads();
    return (tot[0] + tot[1]) + (tot[2] + tot[3]);
}

// (The bookkeeping below - step counter, last norm / clip coefficient, the dropout epoch word of the next step - stays a
// one-workgroup launch of its own.  Folding it into the update kernel's LAST-ARRIVING workgroup was tried: 2 048 arrivals on
// one counter serialise at the L2 and the update went from 10 to 30 us, profiles/r04_step_kernel_summary.txt history.)
__global__ void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, const float* __restrict__ state, size_t n, float beta1,
                             float beta2, float eps, float wd, float max_norm, float grad_scale, int zero_grad) {
    const float step = state[0] + 1.f;
    const float lr = state[2];
    const float gn = sqrtf(sumsq_total(state)) * grad_scale;
    const float clip = (max_norm > 0.f) ? fminf(1.f, max_norm / (gn + 1e-6f)) : 1.f;
    const float gs = grad_scale * clip;
    const float bc1 = 1.f - powf(beta1, step), bc2 = 1.f - powf(beta2, step);
    const float step_size = lr / bc1, inv_sqrt_bc2 = rsqrtf(bc2);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * gs;
        float pi = p[i] * (1.f - lr * wd);
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        pi -= step_size * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
        p[i] = pi; m[i] = mi; v[i] = vi;
        if (zero_grad) g[i] = 0.f;                 // the next step's zero_grad(), for free
    }
}

__global__ void adamw_finish_kernel(float* __restrict__ state, float max_norm, float grad_scale, uint32_t* epoch) {
    const float ss = sumsq_total(state);
    if (threadIdx.x != 0) return;
    if (epoch) epoch[0] += 1;                      // dropout epoch word of the NEXT step (hipGraph replays)
    const float gn = sqrtf(ss) * grad_scale;
    state[3] = (max_norm > 0.f) ? fminf(1.f, max_norm / (gn + 1e-6f)) : 1.f;
    state[4] = gn;
    state[0] += 1.f;
    state[1] = ss;
}


// ---------------------------------------------------------------------------
// EnhancedPowerEncoder's three parallel Conv1d(C -> 64, k = 3 | 5 | 7) + BatchNorm1d(64) branches
// (enhanced_models_v4.py:210-234) run as ONE Conv1d(C -> 192, k = 7) + BatchNorm1d(192).  The merged tensors are built
// from the parts, their running statistics handed back, and their gradients added back into the parts' - each in one
// launch (the host glue was ~55 tiny torch launches per training step: pads, cats, slice copies, slice adds).
//   mode 0: parts -> merged      W[o][c][t] = w_i[o % 64][c][t - lo_i] inside the branch's taps, else 0  (i = o / 64,
//                                 lo_i = (7 - k_i) / 2: the shorter kernels sit around the centre tap);  vectors concatenated
//   mode 1: merged running mean / var -> the parts';  batches_tracked += 1
//   mode 2: parts' gradient sinks += their slices of the merged gradients (null part = frozen parameter: skipped)
//   mode 3: as mode 0, but the merged WEIGHT is written as the forward kernel's bf16 image [192][7][cinp] (what
//           mm_prep_conv_weight would make of the fp32 merged weight, which is then never materialised: 33 MB written and
//           read back per step at config #5); W points to that image
// ---------------------------------------------------------------------------
struct PowerMergeArgs {
    float* w[3]; float* b[3]; float* gamma[3]; float* beta[3]; float* run_mean[3]; float* run_var[3];
    long long* tracked[3];
    float* W; float* B; float* Gamma; float* Beta; float* Run_mean; float* Run_var;
    int cin, k[3], cinp, reserved;
};

template <int MODE>
__global__ void power_merge_kernel(PowerMergeArgs a) {
    const size_t per_o = (size_t)a.cin * 7, total = 192 * per_o;
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid < 192) {                                          // the per-channel vectors
        const int i = (int)gid / 64, j = (int)gid % 64;
        if (MODE == 0 || MODE == 3) {
            a.B[gid] = a.b[i][j]; a.Gamma[gid] = a.gamma[i][j]; a.Beta[gid] = a.beta[i][j];
            a.Run_mean[gid] = a.run_mean[i][j]; a.Run_var[gid] = a.run_var[i][j];
        } else if (MODE == 1) {
            a.run_mean[i][j] = a.Run_mean[gid]; a.run_var[i][j] = a.Run_var[gid];
            if (j == 0 && a.tracked[i]) a.tracked[i][0] += 1;
        } else {
            if (a.b[i] && a.B) a.b[i][j] += a.B[gid];
            if (a.gamma[i] && a.Gamma) a.gamma[i][j] += a.Gamma[gid];
            if (a.beta[i] && a.Beta) a.beta[i][j] += a.Beta[gid];
        }
    }
    if (MODE == 1 || (MODE == 2 && !a.W)) return;
    if (MODE == 3) {
        // image element (o, t, c): channel-contiguous stores; the fp32 reads of a branch stride by its kernel size
        bf16* img = reinterpret_cast<bf16*>(a.W);
        const size_t per_oi = (size_t)7 * a.cinp, itotal = 192 * per_oi;
        for (size_t e = gid; e < itotal; e += (size_t)gridDim.x * blockDim.x) {
            const int o = (int)(e / per_oi);
            const int r = (int)(e - (size_t)o * per_oi);
            const int t = r / a.cinp, c = r - t * a.cinp;
            const int i = o / 64, k = a.k[i], lo = (7 - k) >> 1;
            const bool in = c < a.cin && t >= lo && t < lo + k;
            img[e] = (bf16)(in ? a.w[i][((size_t)(o - 64 * i) * a.cin + c) * k + (t - lo)] : 0.f);
        }
        return;
    }
    for (size_t e = gid; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int o = (int)(e / per_o);
        const int r = (int)(e - (size_t)o * per_o);
        const int c = r / 7, t = r - 7 * c;
        const int i = o / 64, k = a.k[i], lo = (7 - k) >> 1;
        const bool in = t >= lo && t < lo + k;
        const size_t pe = ((size_t)(o - 64 * i) * a.cin + c) * k + (t - lo);
        if (MODE == 0) a.W[e] = in ? a.w[i][pe] : 0.f;
        else if (in && a.w[i]) a.w[i][pe] += a.W[e];
    }
}

// ---------------------------------------------------------------------------
// tiny fused tails of the tabular / bridge models (forward)
// ---------------------------------------------------------------------------
// out[b] = [ w0 * a[b][:Ha] | w1 * c[b][:Hc] ],  (w0, w1) = softmax(pa[0], pc[0])   (fmri_utils.py:93-96)
__global__ void softmax2_concat_kernel(const float* __restrict__ a, const float* __restrict__ c,
                                       const float* __restrict__ pa, const float* __restrict__ pc,
                                       float* __restrict__ out, int B, int Ha, int Hc) {
    const float m = fmaxf(pa[0], pc[0]);
    const float ea = __expf(pa[0] - m), ec = __expf(pc[0] - m);
    const float w0 = ea / (ea + ec), w1 = ec / (ea + ec);
    const int H = Ha + Hc;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * H; i += gridDim.x * blockDim.x) {
        const int b = i / H, j = i % H;
        out[i] = j < Ha ? w0 * a[(size_t)b * Ha + j] : w1 * c[(size_t)b * Hc + (j - Ha)];
    }
}

// LearnedFusionModule tail (enhanced_models_v4.py:468-484): w = 0.5 softmax(logits/T) +
// 0.5 softmax(dyn[b]/T); fused[b] = sum_m w[b][m] feat_m[b].   M <= 4, one wave per row.
__global__ void learned_fusion_kernel(const float* __restrict__ f0, const float* __restrict__ f1,
                                      const float* __restrict__ f2, const float* __restrict__ dyn,
                                      const float* __restrict__ logits, const float* __restrict__ temp,
                                      float* __restrict__ fused, float* __restrict__ wout, int B, int H, int M) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= B) return;
    const float T = temp[0];
    float st[4], dy[4], w[4];
    float ms = -INFINITY, md = -INFINITY;
    for (int m = 0; m < M; ++m) {
        st[m] = logits[m] / T; dy[m] = dyn[(size_t)row * M + m] / T;
        ms = fmaxf(ms, st[m]); md = fmaxf(md, dy[m]);
    }
    float ss = 0.f, sd = 0.f;
    for (int m = 0; m < M; ++m) { st[m] = __expf(st[m] - ms); dy[m] = __expf(dy[m] - md); ss += st[m]; sd += dy[m]; }
    for (int m = 0; m < M; ++m) w[m] = 0.5f * st[m] / ss + 0.5f * dy[m] / sd;
    const float* fs[3] = {f0, f1, f2};
    for (int h = lane; h < H; h += 64) {
        float acc = 0.f;
        for (int m = 0; m < M; ++m) acc += w[m] * fs[m][(size_t)row * H + h];
        fused[(size_t)row * H + h] = acc;
    }
    if (lane < M && wout) wout[(size_t)row * M + lane] = w[lane];
}

// bridge cross-attention core (bridge_utils.py:75-82): one query (EEG token) over two
// keys [EEG, fMRI], nhead heads of dh.  pe / pf = in_proj outputs [B][3E] (q|k|v) of the
// two tokens.  ctx [B][E], attw [B][2] = head-averaged probabilities.
__global__ void attn_1x2_kernel(const float* __restrict__ pe, const float* __restrict__ pf,
                                float* __restrict__ ctx, float* __restrict__ attw, int B, int E, int nhead) {
    const int b = blockIdx.x;
    const int dh = E / nhead;
    __shared__ float p0s[16], p1s[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* q = pe + (size_t)b * 3 * E;
    const float* ke = q + E; const float* ve = q + 2 * E;
    const float* kf = pf + (size_t)b * 3 * E + E; const float* vf = kf + E;
    for (int h = wave; h < nhead; h += (blockDim.x >> 6)) {
        float s0 = 0.f, s1 = 0.f;
        for (int d = lane; d < dh; d += 64) { s0 += q[h * dh + d] * ke[h * dh + d]; s1 += q[h * dh + d] * kf[h * dh + d]; }
        s0 = wave_sum(s0) * rsqrtf((float)dh); s1 = wave_sum(s1) * rsqrtf((float)dh);
        const float m = fmaxf(s0, s1);
        const float e0 = __expf(s0 - m), e1 = __expf(s1 - m);
        const float p0 = e0 / (e0 + e1), p1 = e1 / (e0 + e1);
        for (int d = lane; d < dh; d += 64) ctx[(size_t)b * E + h * dh + d] = p0 * ve[h * dh + d] + p1 * vf[h * dh + d];
        if (lane == 0) { p0s[h] = p0; p1s[h] = p1; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a0 = 0.f, a1 = 0.f;
        for (int h = 0; h < nhead; ++h) { a0 += p0s[h]; a1 += p1s[h]; }
        attw[2 * b] = a0 / nhead; attw[2 * b + 1] = a1 / nhead;
    }
}

// backward of attn_1x2 (attention-probability dropout p applied to the two
// probabilities before mixing, recomputed from (seed, b, h, key)):
//   d proj_e [B][3E] = [dq | dk_e | dv_e],  d proj_f [B][3E] = [0 | dk_f | dv_f]
__global__ void attn_1x2_fused_kernel(const float* __restrict__ pe, const float* __restrict__ pf,
                                      const float* __restrict__ dctx, float* __restrict__ ctx,
                                      float* __restrict__ attw, float* __restrict__ dpe, float* __restrict__ dpf,
                                      int B, int E, int nhead, uint32_t thresh, uint32_t seed, float inv_keep,
                                      const uint32_t* epoch, int backward) {
    seed = mm_eff_seed(seed, epoch);
    const int b = blockIdx.x;
    const int dh = E / nhead;
    __shared__ float p0s[16], p1s[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* q = pe + (size_t)b * 3 * E;
    const float* ke = q + E; const float* ve = q + 2 * E;
    const float* kf = pf + (size_t)b * 3 * E + E; const float* vf = kf + E;
    const float isq = rsqrtf((float)dh);
    for (int h = wave; h < nhead; h += (blockDim.x >> 6)) {
        float s0 = 0.f, s1 = 0.f;
        for (int d = lane; d < dh; d += 64) { s0 += q[h * dh + d] * ke[h * dh + d]; s1 += q[h * dh + d] * kf[h * dh + d]; }
        s0 = wave_sum(s0) * isq; s1 = wave_sum(s1) * isq;
        const float m = fmaxf(s0, s1);
        const float e0 = __expf(s0 - m), e1 = __expf(s1 - m);
        const float p0 = e0 / (e0 + e1), p1 = e1 / (e0 + e1);
        float k0 = 1.f, k1 = 1.f;
        if (thresh) {
            k0 = dropout_scale(seed, (uint32_t)((b * nhead + h) * 2), thresh, inv_keep);
            k1 = dropout_scale(seed, (uint32_t)((b * nhead + h) * 2 + 1), thresh, inv_keep);
        }
        if (!backward) {
            for (int d = lane; d < dh; d += 64)
                ctx[(size_t)b * E + h * dh + d] = p0 * k0 * ve[h * dh + d] + p1 * k1 * vf[h * dh + d];
            if (lane == 0) { p0s[h] = p0; p1s[h] = p1; }
        } else {
            const float* dc = dctx + (size_t)b * E + h * dh;
            float dp0 = 0.f, dp1 = 0.f;
            for (int d = lane; d < dh; d += 64) { dp0 += dc[d] * ve[h * dh + d]; dp1 += dc[d] * vf[h * dh + d]; }
            dp0 = wave_sum(dp0) * k0; dp1 = wave_sum(dp1) * k1;
            const float dot = p0 * dp0 + p1 * dp1;
            const float ds0 = p0 * (dp0 - dot) * isq, ds1 = p1 * (dp1 - dot) * isq;
            float* dq = dpe + (size_t)b * 3 * E;
            float* dkf_ = dpf + (size_t)b * 3 * E;
            for (int d = lane; d < dh; d += 64) {
                const int i = h * dh + d;
                dq[i] = ds0 * ke[i] + ds1 * kf[i];
                dq[E + i] = ds0 * q[i];
                dq[2 * E + i] = p0 * k0 * dc[d];
                dkf_[i] = 0.f;
                dkf_[E + i] = ds1 * q[i];
                dkf_[2 * E + i] = p1 * k1 * dc[d];
            }
        }
    }
    if (!backward) {
        __syncthreads();
        if (threadIdx.x == 0 && attw) {
            float a0 = 0.f, a1 = 0.f;
            for (int h = 0; h < nhead; ++h) { a0 += p0s[h]; a1 += p1s[h]; }
            attw[2 * b] = a0 / nhead; attw[2 * b + 1] = a1 / nhead;
        }
    }
}


// nn.MultiheadAttention core with ONE query token and K <= 4 key/value tokens per sample (the
// modality-level cross attention of the V4 classifiers: crossmodal_v4_enhancements.py:366-372 K = 3,
// :448-456 K = 2).  p[j] = in_proj(token_j) [B][3E] = [q | k | v]; the query is token 0's q.
// Attention-probability dropout as in the 1x2 kernel (index (b * nhead + h) * K + j).
struct Attn1xKArgs {
    const float* p[4]; float* dp[4];
    const float* dctx; float* ctx; float* attw;
    int B, E, nhead, K; uint32_t thresh, seed; float inv_keep; const uint32_t* epoch; int backward;
};
__global__ __launch_bounds__(256) void attn_1xk_kernel(Attn1xKArgs a) {
    const uint32_t seed = mm_eff_seed(a.seed, a.epoch);
    const int b = blockIdx.x, E = a.E, K = a.K;
    const int dh = E / a.nhead;
    __shared__ float ps[16][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* q = a.p[0] + (size_t)b * 3 * E;
    const float isq = rsqrtf((float)dh);
    for (int h = wave; h < a.nhead; h += (blockDim.x >> 6)) {
        float s[4], pr[4], keep[4];
        float m = -INFINITY;
        for (int j = 0; j < K; ++j) {
            const float* kj = a.p[j] + (size_t)b * 3 * E + E;
            float acc = 0.f;
            for (int d = lane; d < dh; d += 64) acc += q[h * dh + d] * kj[h * dh + d];
            s[j] = wave_sum(acc) * isq;
            m = fmaxf(m, s[j]);
        }
        float den = 0.f;
        for (int j = 0; j < K; ++j) { pr[j] = __expf(s[j] - m); den += pr[j]; }
        for (int j = 0; j < K; ++j) {
            pr[j] /= den;
            keep[j] = a.thresh ? dropout_scale(seed, (uint32_t)((b * a.nhead + h) * K + j), a.thresh, a.inv_keep) : 1.f;
        }
        if (!a.backward) {
            for (int d = lane; d < dh; d += 64) {
                float acc = 0.f;
                for (int j = 0; j < K; ++j) acc += pr[j] * keep[j] * a.p[j][(size_t)b * 3 * E + 2 * E + h * dh + d];
                a.ctx[(size_t)b * E + h * dh + d] = acc;
            }
            if (lane == 0)
                for (int j = 0; j < K; ++j) ps[h][j] = pr[j];
        } else {
            const float* dc = a.dctx + (size_t)b * E + h * dh;
            float dp[4], dot = 0.f;
            for (int j = 0; j < K; ++j) {
                const float* vj = a.p[j] + (size_t)b * 3 * E + 2 * E + h * dh;
                float acc = 0.f;
                for (int d = lane; d < dh; d += 64) acc += dc[d] * vj[d];
                dp[j] = wave_sum(acc) * keep[j];
                dot += pr[j] * dp[j];
            }
            for (int d = lane; d < dh; d += 64) {
                const int i = h * dh + d;
                float dq = 0.f;
                for (int j = 0; j < K; ++j) {
                    const float ds = pr[j] * (dp[j] - dot) * isq;
                    dq += ds * a.p[j][(size_t)b * 3 * E + E + i];
                    float* o = a.dp[j] + (size_t)b * 3 * E;
                    if (j > 0) o[i] = 0.f;
                    o[E + i] = ds * q[i];
                    o[2 * E + i] = pr[j] * keep[j] * dc[d];
                }
                a.dp[0][(size_t)b * 3 * E + i] = dq;
            }
        }
    }
    if (!a.backward && a.attw) {
        __syncthreads();
        if (threadIdx.x < K) {
            float s = 0.f;
            for (int h = 0; h < a.nhead; ++h) s += ps[h][threadIdx.x];
            a.attw[(size_t)b * K + threadIdx.x] = s / a.nhead;
        }
    }
}

__global__ void add_f32_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) o[i] = a[i] + b[i];
}

// backward of learned_fusion_kernel.  ONE block of 16 waves, a wave walks rows w, w + 16, ...; the parameter
// gradients (logits [M], temperature) are per-wave partials summed in wave order (no atomics: bit-reproducible).
__global__ __launch_bounds__(1024) void learned_fusion_bwd_kernel(const float* __restrict__ f0, const float* __restrict__ f1,
                                          const float* __restrict__ f2, const float* __restrict__ dyn,
                                          const float* __restrict__ logits, const float* __restrict__ temp,
                                          const float* __restrict__ dfused, float* __restrict__ df0,
                                          float* __restrict__ df1, float* __restrict__ df2, float* __restrict__ ddyn,
                                          float* __restrict__ dlogits, float* __restrict__ dtemp, int B, int H, int M) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    const float T = temp[0];
    __shared__ float part[16][5];
    float pl[4] = {0.f, 0.f, 0.f, 0.f}, pT = 0.f;
    for (int row = wave; row < B; row += nwave) {
        float us[4], ud[4], st[4], dy[4], w[4], dw[4] = {0.f, 0.f, 0.f, 0.f};
        float ms = -INFINITY, md = -INFINITY;
        for (int m = 0; m < M; ++m) {
            us[m] = logits[m] / T; ud[m] = dyn[(size_t)row * M + m] / T;
            ms = fmaxf(ms, us[m]); md = fmaxf(md, ud[m]);
        }
        float ss = 0.f, sd = 0.f;
        for (int m = 0; m < M; ++m) { st[m] = __expf(us[m] - ms); dy[m] = __expf(ud[m] - md); ss += st[m]; sd += dy[m]; }
        for (int m = 0; m < M; ++m) { st[m] /= ss; dy[m] /= sd; w[m] = 0.5f * st[m] + 0.5f * dy[m]; }
        const float* fs[3] = {f0, f1, f2};
        float* dfs[3] = {df0, df1, df2};
        for (int h = lane; h < H; h += 64) {
            const float g = dfused[(size_t)row * H + h];
            for (int m = 0; m < M; ++m) {
                dw[m] += g * fs[m][(size_t)row * H + h];
                dfs[m][(size_t)row * H + h] = w[m] * g;
            }
        }
        for (int m = 0; m < M; ++m) dw[m] = wave_sum(dw[m]);
        if (lane == 0) {
            float dots = 0.f, dotd = 0.f;
            for (int m = 0; m < M; ++m) { dots += st[m] * dw[m]; dotd += dy[m] * dw[m]; }
            float dT = 0.f;
            for (int m = 0; m < M; ++m) {
                const float gs = 0.5f * st[m] * (dw[m] - dots);      // d L / d (logits_m / T)
                const float gd = 0.5f * dy[m] * (dw[m] - dotd);      // d L / d (dyn_m / T)
                ddyn[(size_t)row * M + m] = gd / T;
                pl[m] += gs / T;
                dT -= (gs * us[m] + gd * ud[m]) / T;
            }
            pT += dT;
        }
    }
    if (lane == 0) {
        for (int m = 0; m < 4; ++m) part[wave][m] = pl[m];
        part[wave][4] = pT;
    }
    __syncthreads();
    if (threadIdx.x <= M) {                              // threads 0..M-1: dlogits[m]; thread M: dtemp
        const int j = (int)threadIdx.x == M ? 4 : (int)threadIdx.x;
        float s = 0.f;
        for (int wv = 0; wv < nwave; ++wv) s += part[wv][j];
        if ((int)threadIdx.x == M) dtemp[0] += s;
        else dlogits[threadIdx.x] += s;
    }
}

// backward of softmax2_concat: d a, d c and the two scalar weight-logit gradients
__global__ void softmax2_concat_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ a,
                                           const float* __restrict__ c, const float* __restrict__ pa,
                                           const float* __restrict__ pc, float* __restrict__ da, float* __restrict__ dc,
                                           float* __restrict__ dpa, float* __restrict__ dpc, int B, int Ha, int Hc) {
    const float m = fmaxf(pa[0], pc[0]);
    const float ea = __expf(pa[0] - m), ec = __expf(pc[0] - m);
    const float w0 = ea / (ea + ec), w1 = ec / (ea + ec);
    const int H = Ha + Hc;
    float s0 = 0.f, s1 = 0.f;
    for (int i = threadIdx.x; i < B * H; i += blockDim.x) {
        const int b = i / H, j = i % H;
        const float g = dout[i];
        if (j < Ha) { da[(size_t)b * Ha + j] = w0 * g; s0 += g * a[(size_t)b * Ha + j]; }
        else { dc[(size_t)b * Hc + (j - Ha)] = w1 * g; s1 += g * c[(size_t)b * Hc + (j - Ha)]; }
    }
    __shared__ float r0[16], r1[16];
    s0 = wave_sum(s0); s1 = wave_sum(s1);
    if ((threadIdx.x & 63) == 0) { r0[threadIdx.x >> 6] = s0; r1[threadIdx.x >> 6] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float t0 = 0.f, t1 = 0.f;
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) { t0 += r0[k]; t1 += r1[k]; }
        const float dot = w0 * t0 + w1 * t1;
        dpa[0] += w0 * (t0 - dot);
        dpc[0] += w1 * (t1 - dot);
    }
}

// nn.CrossEntropyLoss(weight=w): loss = sum_b w[t_b] nll_b / sum_b w[t_b]   (single block, B small)
__global__ void weighted_ce_kernel(const float* __restrict__ logits, const long long* __restrict__ target,
                                   const float* __restrict__ cw, float* __restrict__ out, float* __restrict__ dlogits,
                                   int B, int C) {
    __shared__ float wsum_s;
    if (threadIdx.x == 0) {
        float ws = 0.f, ls = 0.f;
        for (int b = 0; b < B; ++b) {
            const float* z = logits + (size_t)b * C;
            float m = -INFINITY;
            for (int c = 0; c < C; ++c) m = fmaxf(m, z[c]);
            float se = 0.f;
            for (int c = 0; c < C; ++c) se += __expf(z[c] - m);
            const int t = (int)target[b];
            const float w = cw ? cw[t] : 1.f;
            ws += w; ls += w * (m + __logf(se) - z[t]);
        }
        wsum_s = ws;
        out[0] += ls / ws;
    }
    __syncthreads();
    if (!dlogits) return;
    for (int i = threadIdx.x; i < B * C; i += blockDim.x) {
        const int b = i / C, c = i % C;
        const float* z = logits + (size_t)b * C;
        float m = -INFINITY;
        for (int k = 0; k < C; ++k) m = fmaxf(m, z[k]);
        float se = 0.f;
        for (int k = 0; k < C; ++k) se += __expf(z[k] - m);
        const int t = (int)target[b];
        const float w = cw ? cw[t] : 1.f;
        dlogits[i] = w * (__expf(z[c] - m) / se - (c == t ? 1.f : 0.f)) / wsum_s;
    }
}

// FocalLoss (CrossModal_EEG_scr.ipynb cell 20): ce_b = lse(z_b) - z_b[t_b]; pt = exp(-ce);
// fl_b = alpha (1 - pt)^gamma ce.  out[0] += scale * sum_b fl_b; per_sample[b] = fl_b (optional);
// dlogits[b][c] = d fl_b / d z_bc (un-reduced; the caller applies the reduction's factor).
__global__ void focal_loss_kernel(const float* __restrict__ logits, const long long* __restrict__ target,
                                  float* __restrict__ out, float* __restrict__ per_sample,
                                  float* __restrict__ dlogits, int B, int C, float alpha, float gamma, float scale) {
    __shared__ float red[256];
    float acc = 0.f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const float* z = logits + (size_t)b * C;
        float m = -INFINITY;
        for (int c = 0; c < C; ++c) m = fmaxf(m, z[c]);
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += __expf(z[c] - m);
        const int t = (int)target[b];
        const float ce = m + __logf(se) - z[t];
        const float pt = __expf(-ce), q = fmaxf(1.f - pt, 0.f);
        const float qg = (gamma == 0.f) ? 1.f : powf(q, gamma);
        const float fl = alpha * qg * ce;
        if (per_sample) per_sample[b] = fl;
        acc += fl;
        if (dlogits) {
            const float qg1 = (gamma == 0.f || q <= 0.f) ? 0.f : gamma * powf(q, gamma - 1.f) * pt * ce;
            const float dce = alpha * (qg + qg1);
            for (int c = 0; c < C; ++c)
                dlogits[(size_t)b * C + c] = dce * (__expf(z[c] - m) / se - (c == t ? 1.f : 0.f));
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] += scale * red[0];
}

// HybridFusionModule mix (crossmodal_v4_enhancements.py:787-797):
// gate = softmax(g[b][0:2]); comb[b] = [ gate0*erp + gate1*pw | conn * boost ]
__global__ void gate2_mix_kernel(const float* __restrict__ g, const float* __restrict__ erp, const float* __restrict__ pw,
                                 const float* __restrict__ conn, float* __restrict__ comb, float* __restrict__ gate,
                                 int B, int H, float boost) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * 2 * H; i += gridDim.x * blockDim.x) {
        const int b = i / (2 * H), j = i % (2 * H);
        const float g0 = g[2 * b], g1 = g[2 * b + 1], m = fmaxf(g0, g1);
        const float e0 = __expf(g0 - m), e1 = __expf(g1 - m);
        const float w0 = e0 / (e0 + e1), w1 = e1 / (e0 + e1);
        comb[i] = j < H ? w0 * erp[(size_t)b * H + j] + w1 * pw[(size_t)b * H + j] : conn[(size_t)b * H + (j - H)] * boost;
        if (j == 0 && gate) { gate[2 * b] = w0; gate[2 * b + 1] = w1; }
    }
}

// backward of gate2_mix: given d comb [B][2H] -> d erp, d pw, d conn [B][H], d gate logits [B][2]
__global__ void gate2_mix_bwd_kernel(const float* __restrict__ dcomb, const float* __restrict__ g,
                                     const float* __restrict__ erp, const float* __restrict__ pw,
                                     float* __restrict__ derp, float* __restrict__ dpw, float* __restrict__ dconn,
                                     float* __restrict__ dg, int B, int H, float boost) {
    const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (b >= B) return;
    const float g0 = g[2 * b], g1 = g[2 * b + 1], m = fmaxf(g0, g1);
    const float e0 = __expf(g0 - m), e1 = __expf(g1 - m);
    const float w0 = e0 / (e0 + e1), w1 = e1 / (e0 + e1);
    float a0 = 0.f, a1 = 0.f;
    for (int h = lane; h < H; h += 64) {
        const float dm = dcomb[(size_t)b * 2 * H + h];
        derp[(size_t)b * H + h] = w0 * dm;
        dpw[(size_t)b * H + h] = w1 * dm;
        dconn[(size_t)b * H + h] = boost * dcomb[(size_t)b * 2 * H + H + h];
        a0 += dm * erp[(size_t)b * H + h];
        a1 += dm * pw[(size_t)b * H + h];
    }
    a0 = wave_sum(a0); a1 = wave_sum(a1);
    if (lane == 0) {
        const float dot = w0 * a0 + w1 * a1;
        dg[2 * b] = w0 * (a0 - dot);
        dg[2 * b + 1] = w1 * (a1 - dot);
    }
}

// LabelSmoothingCrossEntropy (crossmodal_v4_enhancements.py:665-677): loss = mean_b[(1-s)*nll + s*mean_c(-logp)]
// out[0] += loss ; dlogits[b][c] = (softmax - (1-s)*onehot - s/C) / B
__global__ void smoothed_ce_kernel(const float* __restrict__ logits, const long long* __restrict__ target,
                                   float* __restrict__ out, float* __restrict__ dlogits, int B, int C, float smoothing) {
    // one block (B is a batch size): per-thread partial sums over rows b = tid, tid + 256, ..., then a
    // fixed-order tree over the 256 partials - the loss is bit-reproducible
    __shared__ float red[256];
    float acc = 0.f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const float* z = logits + (size_t)b * C;
        float m = -INFINITY;
        for (int c = 0; c < C; ++c) m = fmaxf(m, z[c]);
        float se = 0.f, sz = 0.f;
        for (int c = 0; c < C; ++c) { se += __expf(z[c] - m); sz += z[c]; }
        const float lse = m + __logf(se);
        const int t = (int)target[b];
        const float nll = lse - z[t], smooth = lse - sz / (float)C;
        acc += ((1.f - smoothing) * nll + smoothing * smooth) / (float)B;
        if (dlogits)
            for (int c = 0; c < C; ++c) {
                const float p = __expf(z[c] - lse);
                dlogits[(size_t)b * C + c] = (p - (c == t ? 1.f - smoothing : 0.f) - smoothing / (float)C) / (float)B;
            }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] += red[0];
}

// ---------------------------------------------------------------------------
// multi-scale STFT power front-end (extension a-X3; torch.stft(center=True, reflect,
// periodic Hann) semantics): x (B, C, T) fp32 -> power written CHANNELS-LAST as bf16
// out[b][frame][ch_off + c*F + f] = |sum_n w[n] x[b,c,frame*hop + n - nfft/2] e^{-2 pi i f n / nfft}|^2
// so the result is directly the (B, L, Cin) activation of the Power encoder's first
// conv.  One workgroup per (b, c, frame-block); twiddles and the windowed frame in LDS.
// ---------------------------------------------------------------------------
__global__ void stft_power_kernel(const float* __restrict__ x, bf16* __restrict__ out, float* __restrict__ out_f32,
                                  int C, int T, int nfft, int hop, int frames, int ch_off, int ch_total) {
    extern __shared__ float sm[];
    float* tw_c = sm;                  // [nfft]
    float* tw_s = tw_c + nfft;         // [nfft]
    float* win = tw_s + nfft;          // [nfft] periodic Hann window
    float* fr = win + nfft;            // [8][nfft] windowed frames
    const int b = blockIdx.z, c = blockIdx.y;
    const int F = nfft / 2 + 1;
    for (int n = threadIdx.x; n < nfft; n += blockDim.x) {
        float s, co;
        __sincosf(6.283185307179586f * (float)n / (float)nfft, &s, &co);
        tw_c[n] = co; tw_s[n] = s;
        win[n] = 0.5f - 0.5f * __cosf(6.283185307179586f * (float)n / (float)nfft);
    }
    const float* xr = x + ((size_t)b * C + c) * T;
    // the workgroup walks its share of the frame blocks (gridDim.x = 1 for short sequences: 10 240 tiny workgroups - twiddles,
    // window and launch overhead per 8 frames - were 133 us per scale at config #5; one workgroup per (b, c) now)
    for (int f0 = blockIdx.x * 8; f0 < frames; f0 += gridDim.x * 8) {
        __syncthreads();                                    // twiddles ready / the previous block's frames consumed
        for (int i = threadIdx.x; i < 8 * nfft; i += blockDim.x) {
            const int fi = i / nfft, n = i % nfft;
            const int frame = f0 + fi;
            float v = 0.f;
            if (frame < frames) {
                int t = frame * hop + n - nfft / 2;
                if (t < 0) t = -t;                              // reflect padding
                if (t >= T) t = 2 * (T - 1) - t;
                v = xr[t] * win[n];
            }
            fr[i] = v;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 8 * F; i += blockDim.x) {
            const int fi = i / F, f = i % F;
            const int frame = f0 + fi;
            if (frame >= frames) continue;
            float re = 0.f, im = 0.f;
            const float* fv = fr + fi * nfft;
            for (int n = 0; n < nfft; ++n) {
                const int k = (f * n) & (nfft - 1);             // nfft is a power of two
                re += fv[n] * tw_c[k];
                im -= fv[n] * tw_s[k];
            }
            const float p = re * re + im * im;
            const size_t o = ((size_t)b * frames + frame) * ch_total + ch_off + (size_t)c * F + f;
            if (out) out[o] = (bf16)p;
            if (out_f32) out_f32[o] = p;
        }
    }
}

// The same spectra by a radix-2 FFT (nfft <= 256): one workgroup per (b, c), each of its four waves transforms one frame at
// a time in its own LDS scratch (decimation in time on the bit-reversed, windowed frame; log2(nfft) butterfly stages of
// nfft / 2 butterflies, lanes = butterflies).  The direct DFT above costs nfft MACs per (frame, bin) - 705 M MAC pairs at
// config #5 (64 ch x 1024 samples, nfft 64 + 128, hop 32): 145 us per scale; the FFT needs nfft / 2 * log2(nfft) butterflies
// per frame.  fp32 throughout; twiddles from __sincosf (as the DFT's).
template <int LOG2N>
__global__ __launch_bounds__(256) void stft_power_fft_kernel(const float* __restrict__ x, bf16* __restrict__ out,
                                                             float* __restrict__ out_f32, int C, int T, int hop, int frames,
                                                             int ch_off, int ch_total) {
    constexpr int N = 1 << LOG2N, H = N / 2, F = H + 1;
    __shared__ float tw_c[H], tw_s[H], win[N];
    __shared__ float re[4][N], im[4][N];
    const int b = blockIdx.z, c = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int n = tid; n < N; n += 256) {
        float sn, cs;
        __sincosf(6.283185307179586f * (float)n / (float)N, &sn, &cs);
        if (n < H) { tw_c[n] = cs; tw_s[n] = sn; }             // e^{-2 pi i n / N} = tw_c - i tw_s
        win[n] = 0.5f - 0.5f * cs;                              // periodic Hann
    }
    __syncthreads();
    const float* xr = x + ((size_t)b * C + c) * T;
    float* r = re[wave];
    float* q = im[wave];
    // Every wave owns its frame and its own LDS arrays: after the twiddle tables nothing is shared between waves, so the
    // stages are ordered by WAVE-level fences only (a wave's LDS operations execute in issue order; the fence keeps the
    // compiler from moving them).  With a workgroup barrier per stage the four waves advanced in lock-step: ten barriers per
    // frame round in a kernel that is nothing but latency (45 us for nfft = 128 at config #5).
    auto wave_sync = []() __attribute__((always_inline)) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    const int rounds = (frames + 3) / 4;
    // the samples of the NEXT round's frame are requested before this round's butterflies (a frame's global round trip was
    // as long as its whole transform)
    constexpr int PER = (N + 63) / 64;
    float nxt[PER];
    auto fetch = [&](int frame) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int n = lane + 64 * u;
            int t = frame * hop + n - H;
            if (t < 0) t = -t;                                  // reflect padding (torch.stft center = True)
            if (t >= T) t = 2 * (T - 1) - t;
            nxt[u] = (n < N && frame < frames) ? xr[t] : 0.f;
        }
    };
    fetch(wave);
    for (int it = 0; it < rounds; ++it) {
        const int frame = it * 4 + wave;
        const bool live = frame < frames;
        // bit-reversed store of the windowed frame
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int n = lane + 64 * u;
            if (n < N) {
                const int rv = (int)(__brev((unsigned)n) >> (32 - LOG2N));
                r[rv] = nxt[u] * win[n];
                q[rv] = 0.f;
            }
        }
        if (it + 1 < rounds) fetch(frame + 4);
        wave_sync();
#pragma unroll
        for (int sgm = 0; sgm < LOG2N; ++sgm) {
            const int m = 1 << sgm;                             // half size of this stage's butterflies
            for (int j = lane; j < H; j += 64) {
                const int k = j & (m - 1);
                const int i0 = ((j >> sgm) << (sgm + 1)) + k, i1 = i0 + m;
                const int tk = k << (LOG2N - 1 - sgm);          // twiddle index k * N / (2 m)
                const float wc = tw_c[tk], ws = tw_s[tk];
                const float ar = r[i1], ai = q[i1];
                const float tr = ar * wc + ai * ws, ti = ai * wc - ar * ws;     // (ar + i ai) (wc - i ws)
                const float br = r[i0], bi = q[i0];
                r[i0] = br + tr; q[i0] = bi + ti;
                r[i1] = br - tr; q[i1] = bi - ti;
            }
            wave_sync();
        }
        if (live) {
            const size_t o = ((size_t)b * frames + frame) * ch_total + ch_off + (size_t)c * F;
            for (int f = lane; f < F; f += 64) {
                const float p = r[f] * r[f] + q[f] * q[f];
                if (out) out[o + f] = (bf16)p;
                if (out_f32) out_f32[o + f] = p;
            }
        }
        wave_sync();
    }
}

// ---------------------------------------------------------------------------
// normalize_modality (run_training_lite.py:48-51, applied to every sample's power features at :162):
// x[b] <- (x[b] - mean(x[b])) / (std(x[b]) + eps), population std (the reference z-scores numpy arrays: ddof = 0), over ALL elements of sample b, fp32 in ->
// bf16 channels-last out (the Power encoder's first-conv operand).  x [B][rows][ch_total]; only channels
// < ch_valid count (the padding channels stay zero).  One workgroup per sample, two sweeps; fixed-order
// sums (bit-reproducible); the mean is subtracted before squaring (two-pass variance).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void sample_zscore_kernel(const float* __restrict__ x, bf16* __restrict__ out,
                                                             int rows, int ch_valid, int ch_total, float eps) {
    __shared__ float red[16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* xs = x + (size_t)b * rows * ch_total;
    bf16* os = out + (size_t)b * rows * ch_total;
    const size_t n = (size_t)rows * ch_total;
    const float cnt = (float)rows * (float)ch_valid;
    auto block_sum = [&](float v) {
        v = wave_sum(v);
        __syncthreads();
        if (lane == 0) red[wave] = v;
        __syncthreads();
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += red[w];
        return t;
    };
    if (ch_valid == ch_total && (ch_total & 3) == 0 && n < (1ull << 31)) {
        // no padding channels (config #5: 6 272 of 6 272): three float4 sweeps with 32-bit indices.  (The general path's 64-bit
        // modulo per ELEMENT and scalar loads made this one-workgroup-per-sample kernel 200 us at config #5.)
        const unsigned n4 = (unsigned)(n >> 2);
        const float4* x4 = reinterpret_cast<const float4*>(xs);
        float s = 0.f;
        for (unsigned i = tid; i < n4; i += 1024) { const float4 v = x4[i]; s += (v.x + v.y) + (v.z + v.w); }
        const float mean = block_sum(s) / cnt;
        float q = 0.f;
        for (unsigned i = tid; i < n4; i += 1024) {
            const float4 v = x4[i];
            const float a = v.x - mean, c = v.y - mean, d = v.z - mean, e = v.w - mean;
            q += (a * a + c * c) + (d * d + e * e);
        }
        const float inv = 1.f / (sqrtf(block_sum(q) / cnt) + eps);   // population std: the reference z-scores numpy arrays (ddof = 0)
        for (unsigned i = tid; i < n4; i += 1024) {
            const float4 v = x4[i];
            bf16x4 o = {(bf16)((v.x - mean) * inv), (bf16)((v.y - mean) * inv), (bf16)((v.z - mean) * inv), (bf16)((v.w - mean) * inv)};
            *reinterpret_cast<bf16x4*>(os + 4 * (size_t)i) = o;
        }
        return;
    }
    float s = 0.f;
    for (size_t i = tid; i < n; i += 1024) s += ((int)(i % ch_total) < ch_valid) ? xs[i] : 0.f;
    const float mean = block_sum(s) / cnt;
    float q = 0.f;
    for (size_t i = tid; i < n; i += 1024) {
        const float d = ((int)(i % ch_total) < ch_valid) ? xs[i] - mean : 0.f;
        q += d * d;
    }
    const float inv = 1.f / (sqrtf(block_sum(q) / cnt) + eps);       // population std: the reference z-scores numpy arrays (ddof = 0)
    for (size_t i = tid; i < n; i += 1024) os[i] = (bf16)(((int)(i % ch_total) < ch_valid) ? (xs[i] - mean) * inv : 0.f);
}

// The same z-score with the sample's elements dealt out over ZS_CHUNKS workgroups (config #5: 33 frames x 6 272 channels per
// sample - one workgroup per sample was 32 workgroups on 256 CUs, 41 us for 40 MB).  Pass 1: every workgroup leaves the sum
// and the sum of squares of its chunk in DOUBLE precision (the variance is then E[x^2] - mean^2 without the two-pass form's
// second sweep; fp64 keeps the cancellation harmless); pass 2: every workgroup adds the sample's partials in chunk order -
// the same value in all of them, the same bits every run - and writes its chunk.  No-padding layouts only (the fast path above).
constexpr int ZS_CHUNKS = 32;
__global__ __launch_bounds__(256) void zscore_partial_kernel(const float* __restrict__ x, double* __restrict__ part, unsigned n4) {
    __shared__ double red[8];
    const int b = blockIdx.y, c = blockIdx.x, tid = threadIdx.x;
    const float4* x4 = reinterpret_cast<const float4*>(x) + (size_t)b * n4;
    const unsigned lo = (unsigned)((unsigned long long)n4 * c / ZS_CHUNKS), hi = (unsigned)((unsigned long long)n4 * (c + 1) / ZS_CHUNKS);
    double s = 0.0, q = 0.0;
    for (unsigned i = lo + tid; i < hi; i += 256) {
        const float4 v = x4[i];
        s += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
        q += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
    if ((tid & 63) == 0) { red[tid >> 6] = s; red[4 + (tid >> 6)] = q; }
    __syncthreads();
    if (tid == 0) {
        double* p = part + ((size_t)b * ZS_CHUNKS + c) * 2;
        p[0] = (red[0] + red[1]) + (red[2] + red[3]);
        p[1] = (red[4] + red[5]) + (red[6] + red[7]);
    }
}
__global__ __launch_bounds__(256) void zscore_apply_kernel(const float* __restrict__ x, const double* __restrict__ part,
                                                          bf16* __restrict__ out, unsigned n4, float cnt, float eps) {
    const int b = blockIdx.y, c = blockIdx.x, tid = threadIdx.x;
    double s = 0.0, q = 0.0;
    for (int k = 0; k < ZS_CHUNKS; ++k) { s += part[((size_t)b * ZS_CHUNKS + k) * 2]; q += part[((size_t)b * ZS_CHUNKS + k) * 2 + 1]; }
    const double mean_d = s / (double)cnt;
    const double var = fmax(q / (double)cnt - mean_d * mean_d, 0.0);
    const float mean = (float)mean_d, inv = 1.f / ((float)sqrt(var) + eps);
    const float4* x4 = reinterpret_cast<const float4*>(x) + (size_t)b * n4;
    bf16* os = out + (size_t)b * n4 * 4;
    const unsigned lo = (unsigned)((unsigned long long)n4 * c / ZS_CHUNKS), hi = (unsigned)((unsigned long long)n4 * (c + 1) / ZS_CHUNKS);
    for (unsigned i = lo + tid; i < hi; i += 256) {
        const float4 v = x4[i];
        bf16x4 o = {(bf16)((v.x - mean) * inv), (bf16)((v.y - mean) * inv), (bf16)((v.z - mean) * inv), (bf16)((v.w - mean) * inv)};
        *reinterpret_cast<bf16x4*>(os + 4 * (size_t)i) = o;
    }
}

// backward of sample_zscore_kernel: y = (x - mean) / d, d = std + eps (population std over the cnt valid elements):
//   dx_i = (g_i - mean(g)) / d - y_i * mean(g * y) / std        (padding channels: 0)
// g = bf16 gradient w.r.t. the z-scored (bf16) tensor, same layout; dx fp32.  Fixed-order block sums.
__global__ __launch_bounds__(1024) void sample_zscore_bwd_kernel(const float* __restrict__ x, const bf16* __restrict__ g,
                                                                 float* __restrict__ dx, int rows, int ch_valid, int ch_total,
                                                                 float eps) {
    __shared__ float red[16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t n = (size_t)rows * ch_total;
    const float* xs = x + (size_t)b * n;
    const bf16* gs = g + (size_t)b * n;
    float* os = dx + (size_t)b * n;
    const float cnt = (float)rows * (float)ch_valid;
    auto block_sum = [&](float v) {
        v = wave_sum(v);
        __syncthreads();
        if (lane == 0) red[wave] = v;
        __syncthreads();
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += red[w];
        return t;
    };
    auto valid = [&](size_t i) { return (int)(i % ch_total) < ch_valid; };
    float s = 0.f;
    for (size_t i = tid; i < n; i += 1024) s += valid(i) ? xs[i] : 0.f;
    const float mean = block_sum(s) / cnt;
    float q = 0.f;
    for (size_t i = tid; i < n; i += 1024) {
        const float d = valid(i) ? xs[i] - mean : 0.f;
        q += d * d;
    }
    const float sd = sqrtf(block_sum(q) / cnt);
    const float inv = 1.f / (sd + eps);
    float sg = 0.f, sgy = 0.f;
    for (size_t i = tid; i < n; i += 1024)
        if (valid(i)) {
            const float gi = (float)gs[i];
            sg += gi;
            sgy += gi * (xs[i] - mean) * inv;
        }
    const float mg = block_sum(sg) / cnt;
    const float mgy = block_sum(sgy) / cnt;
    const float c2 = mgy / fmaxf(sd, 1e-30f);
    for (size_t i = tid; i < n; i += 1024)
        os[i] = valid(i) ? ((float)gs[i] - mg) * inv - (xs[i] - mean) * inv * c2 : 0.f;
}

// ---------------------------------------------------------------------------
// backward of stft_power_kernel: dx[b, c, t] += sum over the (frame, n) that read sample t (directly or through the
// reflect padding) of win[n] * dv[frame][n],   dv[n] = 2 sum_f gP[f] (re_f cos(2 pi f n / N) - im_f sin(2 pi f n / N)).
// One workgroup per (b, c): frames in blocks of 8 - windowed frames and their DFT recomputed in LDS, the gradient of
// the windowed frame formed by the inverse sum, then every thread GATHERS the contributions to the samples it owns
// in a fixed order (frame ascending; direct, left-reflected, right-reflected position): no atomics, bit-reproducible.
// gP fp32 [B][frames][ch_total] (channels ch_off + c * F + f); dx fp32 [B][C][T] is ADDED to (one launch per scale).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stft_power_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gP,
                                                             float* __restrict__ dx, int C, int T, int nfft, int hop,
                                                             int frames, int ch_off, int ch_total) {
    extern __shared__ float sm[];
    const int F = nfft / 2 + 1;
    float* tw_c = sm;                  // [nfft]
    float* tw_s = tw_c + nfft;         // [nfft]
    float* win = tw_s + nfft;          // [nfft]
    float* fr = win + nfft;            // [8][nfft] windowed frames, then their gradient
    float* cr = fr + 8 * nfft;         // [8][F]  2 gP re
    float* ci = cr + 8 * F;            // [8][F]  2 gP im
    float* acc = ci + 8 * F;           // [T]
    const int b = blockIdx.y, c = blockIdx.x, tid = threadIdx.x;
    for (int n = tid; n < nfft; n += 256) {
        float s, co;
        __sincosf(6.283185307179586f * (float)n / (float)nfft, &s, &co);
        tw_c[n] = co; tw_s[n] = s;
        win[n] = 0.5f - 0.5f * __cosf(6.283185307179586f * (float)n / (float)nfft);
    }
    for (int t = tid; t < T; t += 256) acc[t] = 0.f;
    const float* xr = x + ((size_t)b * C + c) * T;
    __syncthreads();
    for (int f0 = 0; f0 < frames; f0 += 8) {
        for (int i = tid; i < 8 * nfft; i += 256) {
            const int fi = i / nfft, n = i % nfft, frame = f0 + fi;
            float v = 0.f;
            if (frame < frames) {
                int t = frame * hop + n - nfft / 2;
                if (t < 0) t = -t;
                if (t >= T) t = 2 * (T - 1) - t;
                v = xr[t] * win[n];
            }
            fr[i] = v;
        }
        __syncthreads();
        for (int i = tid; i < 8 * F; i += 256) {
            const int fi = i / F, f = i % F, frame = f0 + fi;
            float re = 0.f, im = 0.f;
            if (frame < frames) {
                const float* fv = fr + fi * nfft;
                for (int n = 0; n < nfft; ++n) {
                    const int k = (f * n) & (nfft - 1);
                    re += fv[n] * tw_c[k];
                    im -= fv[n] * tw_s[k];
                }
                const float g2 = 2.f * gP[((size_t)b * frames + frame) * ch_total + ch_off + (size_t)c * F + f];
                re *= g2; im *= g2;
            }
            cr[i] = re; ci[i] = im;
        }
        __syncthreads();
        for (int i = tid; i < 8 * nfft; i += 256) {             // gradient of the windowed frame, times the window
            const int fi = i / nfft, n = i % nfft;
            float dv = 0.f;
            const float *pr = cr + fi * F, *pi = ci + fi * F;
            for (int f = 0; f < F; ++f) {
                const int k = (f * n) & (nfft - 1);
                dv += pr[f] * tw_c[k] - pi[f] * tw_s[k];
            }
            fr[i] = dv * win[n];
        }
        __syncthreads();
        for (int t0 = tid; t0 < T; t0 += 256) {
            float a = acc[t0];
            for (int fi = 0; fi < 8; ++fi) {
                const int frame = f0 + fi;
                if (frame >= frames) break;
                const int base = nfft / 2 - frame * hop;
                int n = t0 + base;                                              // read directly
                if (n >= 0 && n < nfft) a += fr[fi * nfft + n];
                n = -t0 + base;                                                 // read as the left reflection of t = -t0
                if (t0 > 0 && n >= 0 && n < nfft) a += fr[fi * nfft + n];
                n = 2 * (T - 1) - t0 + base;                                    // right reflection
                if (t0 < T - 1 && n >= 0 && n < nfft) a += fr[fi * nfft + n];
            }
            acc[t0] = a;
        }
        __syncthreads();
    }
    float* dr = dx + ((size_t)b * C + c) * T;
    for (int t = tid; t < T; t += 256) dr[t] += acc[t];
}

__global__ void mul_f32_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) o[i] = a[i] * b[i];
}

// ---------------------------------------------------------------------------
// Encoder output head on the pooled token (enhanced_models_v4.py:161-167, 188-191): Linear(D -> N) ->
// act -> Dropout on pooled[b] (the mean over time arrives already reduced), fp32 FMAs, one workgroup
// per sample.  Backward: dz = dout * mask * act'(z); d pooled = dz W; every token of the sample gets
// d pooled / L (the mean's gradient), optionally with a second, dropout-masked bf16 copy for the GEMM
// that consumes it next.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pooled_head_fwd_kernel(const float* __restrict__ pooled, const mm_acc_t* __restrict__ pooled_acc,
                                                             const float* __restrict__ W,
                                                             const float* __restrict__ bias, float* __restrict__ out,
                                                             bf16* __restrict__ z_pre, bf16* __restrict__ pooled_bf16,
                                                             int D, int N, int act, uint32_t thresh, float inv_keep,
                                                             uint32_t seed, const uint32_t* __restrict__ epoch) {
    __shared__ float xs[1024];
    const int b = blockIdx.x;
    for (int k = threadIdx.x; k < D; k += 256) {
        const float v = pooled ? pooled[(size_t)b * D + k] : acc_val<MM_ACC_GRAD>(pooled_acc[(size_t)b * D + k]);
        xs[k] = v;
        if (pooled_bf16) pooled_bf16[(size_t)b * D + k] = (bf16)v;
    }
    __syncthreads();
    seed = mm_eff_seed(seed, epoch);
    for (int n = threadIdx.x; n < N; n += 256) {
        const float* wr = W + (size_t)n * D;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        int k = 0;
        for (; k + 64 <= D; k += 64) {                      // sixteen independent float4 loads per round (see pooled_head_bwd_kernel)
            float4 w4[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) w4[q] = *reinterpret_cast<const float4*>(wr + k + q * 4);
#pragma unroll
            for (int q = 0; q < 16; ++q)
                acc[q & 3] += w4[q].x * xs[k + q * 4] + w4[q].y * xs[k + q * 4 + 1] + w4[q].z * xs[k + q * 4 + 2] +
                              w4[q].w * xs[k + q * 4 + 3];
        }
        for (; k < D; k += 16) {                            // four independent float4 loads per round
            float4 w4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) w4[q] = *reinterpret_cast<const float4*>(wr + k + q * 4);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                acc[q] += w4[q].x * xs[k + q * 4] + w4[q].y * xs[k + q * 4 + 1] + w4[q].z * xs[k + q * 4 + 2] +
                          w4[q].w * xs[k + q * 4 + 3];
        }
        const float z = (acc[0] + acc[1]) + (acc[2] + acc[3]) + (bias ? bias[n] : 0.f);
        const size_t idx = (size_t)b * N + n;
        if (z_pre) z_pre[idx] = (bf16)z;
        float v = apply_act(z, act);
        if (thresh) v *= dropout_scale(seed, (uint32_t)idx, thresh, inv_keep);
        out[idx] = v;
    }
}

__global__ __launch_bounds__(256) void pooled_head_bwd_kernel(const float* __restrict__ dout, const bf16* __restrict__ z_pre,
                                                             const float* __restrict__ W, bf16* __restrict__ dz_bf16,
                                                             float* __restrict__ dx, bf16* __restrict__ dx_bf16, int L, int D,
                                                             int N, int rows_per_wg, int act, uint32_t thresh, float inv_keep,
                                                             uint32_t seed, uint32_t thresh2, float inv_keep2, uint32_t seed2,
                                                             const uint32_t* __restrict__ epoch, float* __restrict__ rows_out) {
    __shared__ float dz[1024];
    __shared__ __attribute__((aligned(16))) float dp[1024];
    const int b = blockIdx.x, l0 = blockIdx.y * rows_per_wg;
    seed = mm_eff_seed(seed, epoch);
    seed2 = mm_eff_seed(seed2, epoch);
    for (int n = threadIdx.x; n < N; n += 256) {
        const size_t idx = (size_t)b * N + n;
        float g = dout[idx] * act_grad((float)z_pre[idx], act);
        if (thresh) g *= dropout_scale(seed, (uint32_t)idx, thresh, inv_keep);
        dz[n] = g;
        if (blockIdx.y == 0 && dz_bf16) dz_bf16[idx] = (bf16)g;
    }
    __syncthreads();
    const float invL = 1.f / (float)L;
    for (int k = threadIdx.x; k < D; k += 256) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        int n = 0;
        // coalesced across k; four chains, SIXTEEN loads in flight per round: this loop is a chain of L2 round trips with one
        // workgroup per sample on the chip (13 us for N = 128 at four per round, alone in the serial heads section of the step)
        for (; n + 16 <= N; n += 16) {
            float w[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) w[q] = W[(size_t)(n + q) * D + k];
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q & 3] += dz[n + q] * w[q];
        }
        for (; n < N; n += 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] += dz[n + q] * W[(size_t)(n + q) * D + k];
        }
        dp[k] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) * invL;
        if (rows_out && blockIdx.y == 0) rows_out[(size_t)b * D + k] = dp[k];     // the ONE row every token of the sample receives
    }
    __syncthreads();
    const int vec = D / 4;                                  // float4 groups per row
    const int lend = min(L, l0 + rows_per_wg);
    for (int i = threadIdx.x; i < (lend - l0) * vec; i += 256) {
        const int l = l0 + i / vec, k4 = (i % vec) * 4;
        const float4 v = *reinterpret_cast<const float4*>(dp + k4);
        const size_t base = ((size_t)b * L + l) * D + k4;
        if (dx) *reinterpret_cast<float4*>(dx + base) = v;
        if (dx_bf16) {
            const float vs[4] = {v.x, v.y, v.z, v.w};
            bf16x4 o;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                o[c] = (bf16)(thresh2 ? vs[c] * dropout_scale(seed2, (uint32_t)(base + c), thresh2, inv_keep2) : vs[c]);
            *reinterpret_cast<bf16x4*>(dx_bf16 + base) = o;
        }
    }
}

// drop_path / stochastic depth (crossmodal_v4_enhancements.py:639-650): sample b is kept with
// probability 1 - p and scaled by 1 / (1 - p); the mask depends on (seed, b) only, so the same
// launch on the upstream gradient is the backward.
__global__ void drop_path_kernel(const float* __restrict__ x, float* __restrict__ o, size_t n, size_t inner,
                                 uint32_t thresh, float inv_keep, uint32_t base, const uint32_t* __restrict__ epoch) {
    const uint32_t seed = mm_eff_seed(base, epoch);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        o[i] = x[i] * dropout_scale(seed, (uint32_t)(i / inner), thresh, inv_keep);
}

// mean over S of bf16 [R][S][N] -> fp32 [R][N]   (AdaptiveAvgPool1d(1) of the Lite encoders)
__global__ void meanpool_bf16_kernel(const bf16* __restrict__ x, float* __restrict__ out, int S, int N) {
    const int r = blockIdx.x;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        float s = 0.f;
        for (int t = 0; t < S; ++t) s += (float)x[((size_t)r * S + t) * N + n];
        out[(size_t)r * N + n] = s / (float)S;
    }
}

inline uint32_t thresh_h(float p) { return p > 0.f ? (uint32_t)((double)p * 4294967296.0) : 0u; }
inline int grid_h(size_t n, int cap = 2048) { size_t g = (n + 255) / 256; return (int)(g < (size_t)cap ? (g ? g : 1) : cap); }

}  // namespace

extern "C" {

int mm_small_linear_fwd(const float* x, const float* W, const float* bias, const float* scale, const float* shift,
                        float* y, float* pre, int B, int K, int N, int act, float drop_p, uint32_t seed,
                        const uint32_t* seed_epoch, hipStream_t st) {
    MM_REQUIRE(x && W && y && B > 0 && K > 0 && N > 0, "small_linear_fwd: null/invalid");
    MM_REQUIRE((scale == nullptr) == (shift == nullptr), "small_linear_fwd: scale/shift come in pairs");
    MM_REQUIRE((size_t)K * 4 <= 64 * 1024, "small_linear_fwd: K=%d too large", K);
    const int gy = ceil_div(N, 4) < 64 ? ceil_div(N, 4) : 64;
    hipLaunchKernelGGL(small_linear_fwd_kernel, dim3(B, gy), dim3(256), K * sizeof(float), st, x, W, bias, scale, shift,
                       y, pre, B, K, N, act, thresh_h(drop_p), seed, drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f, seed_epoch);
    return mm_check_launch("small_linear_fwd");
}

int mm_small_linear_bwd(const float* dy, const float* x, const float* W, float* dx, float* dW, float* db, int B, int K,
                        int N, hipStream_t st) {
    MM_REQUIRE(dy && x && W && B > 0 && K > 0 && N > 0, "small_linear_bwd: null/invalid");
    MM_REQUIRE((size_t)N * 4 <= 64 * 1024, "small_linear_bwd: N=%d too large", N);
    if (dx) hipLaunchKernelGGL(small_linear_dx_kernel, dim3(B), dim3(256), N * sizeof(float), st, dy, W, dx, B, K, N);
    if (dW) hipLaunchKernelGGL(small_linear_dw_kernel, dim3(N), dim3(128), 0, st, dy, x, dW, db, B, K, N);
    return mm_check_launch("small_linear_bwd");
}

int mm_act_f32(const float* z, float* y, int64_t n, int act, float drop_p, uint32_t seed, const uint32_t* seed_epoch,
               hipStream_t st) {
    MM_REQUIRE(z && y && n > 0, "act_f32: null");
    hipLaunchKernelGGL(act_f32_kernel, dim3(grid_h((size_t)n)), dim3(256), 0, st, z, y, (size_t)n, act, thresh_h(drop_p),
                       seed, drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f, seed_epoch);
    return mm_check_launch("act_f32");
}

int mm_act_bwd_f32(const float* g, const float* z, float* out, int64_t n, int act, float drop_p, uint32_t seed,
                   const uint32_t* seed_epoch, hipStream_t st) {
    MM_REQUIRE(g && out && n > 0, "act_bwd_f32: null");
    hipLaunchKernelGGL(act_bwd_f32_kernel, dim3(grid_h((size_t)n)), dim3(256), 0, st, g, z, out, (size_t)n, act,
                       thresh_h(drop_p), seed, drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f, seed_epoch);
    return mm_check_launch("act_bwd_f32");
}

int mm_colstats(const float* x, float* stats, int B, int N, hipStream_t st) {
    MM_REQUIRE(x && stats && B > 0 && N > 0, "colstats: null");
    hipLaunchKernelGGL(colstats_kernel, dim3(ceil_div(N, 64)), dim3(64), 0, st, x, stats, B, N);
    return mm_check_launch("colstats");
}

int mm_proj_heads_fwd(const float* x_e, const float* W_e, const float* b_e, const float* g_e, const float* be_e, int K_e,
                      const float* x_f, const float* W_f, const float* b_f, const float* g_f, const float* be_f, int K_f,
                      float* z1, float* hn, float* stat, float* z, float* nrm, int B, int N, float eps, float drop_p,
                      uint32_t seed_e, uint32_t seed_f, const uint32_t* seed_epoch, hipStream_t st) {
    MM_REQUIRE(x_e && W_e && g_e && be_e && x_f && W_f && g_f && be_f && z && nrm, "proj_heads_fwd: null");
    MM_REQUIRE((!z1 && !hn && !stat) || (z1 && hn && stat), "proj_heads_fwd: z1/hn/stat go together");
    MM_REQUIRE(B > 0 && N > 0 && N <= HEAD_MAXN && K_e > 0 && K_e <= HEAD_MAXK && K_f > 0 && K_f <= HEAD_MAXK,
               "proj_heads_fwd: B=%d N=%d K=%d/%d", B, N, K_e, K_f);
    HeadsArgs a{};
    a.s[0].x = x_e; a.s[0].W = W_e; a.s[0].bias = b_e; a.s[0].gamma = g_e; a.s[0].beta = be_e; a.s[0].K = K_e; a.s[0].seed = seed_e;
    a.s[1].x = x_f; a.s[1].W = W_f; a.s[1].bias = b_f; a.s[1].gamma = g_f; a.s[1].beta = be_f; a.s[1].K = K_f; a.s[1].seed = seed_f;
    a.z1 = z1; a.hn = hn; a.stat = stat; a.z = z; a.nrm = nrm; a.B = B; a.N = N; a.eps = eps;
    a.thresh = thresh_h(drop_p); a.inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f; a.epoch = seed_epoch;
    hipLaunchKernelGGL(proj_heads_fwd_kernel, dim3(B, 2), dim3(256), 0, st, a);
    return mm_check_launch("proj_heads_fwd");
}

int mm_proj_heads_bwd(const float* dz, const float* z, const float* nrm, const float* hn, const float* z1,
                      const float* stat, const float* x_e, const float* W_e, const float* g_e, int K_e,
                      const float* x_f, const float* W_f, const float* g_f, int K_f, float* dx_e, float* dW_e,
                      float* db_e, float* dg_e, float* dbe_e, float* dx_f, float* dW_f, float* db_f, float* dg_f,
                      float* dbe_f, int B, int N, float drop_p, uint32_t seed_e, uint32_t seed_f,
                      const uint32_t* seed_epoch, hipStream_t st) {
    MM_REQUIRE(dz && z && nrm && hn && z1 && stat && x_e && W_e && g_e && x_f && W_f && g_f, "proj_heads_bwd: null");
    MM_REQUIRE(B > 0 && N > 0 && N <= HEAD_MAXN && K_e > 0 && K_e <= HEAD_MAXK && K_f > 0 && K_f <= HEAD_MAXK,
               "proj_heads_bwd: B=%d N=%d K=%d/%d", B, N, K_e, K_f);
    HeadsArgs a{};
    a.s[0].x = x_e; a.s[0].W = W_e; a.s[0].gamma = g_e; a.s[0].K = K_e; a.s[0].seed = seed_e;
    a.s[0].dx = dx_e; a.s[0].dW = dW_e; a.s[0].dbias = db_e; a.s[0].dgamma = dg_e; a.s[0].dbeta = dbe_e;
    a.s[1].x = x_f; a.s[1].W = W_f; a.s[1].gamma = g_f; a.s[1].K = K_f; a.s[1].seed = seed_f;
    a.s[1].dx = dx_f; a.s[1].dW = dW_f; a.s[1].dbias = db_f; a.s[1].dgamma = dg_f; a.s[1].dbeta = dbe_f;
    a.z1 = const_cast<float*>(z1); a.hn = const_cast<float*>(hn); a.stat = const_cast<float*>(stat);
    a.z = const_cast<float*>(z); a.nrm = const_cast<float*>(nrm); a.dz = dz; a.B = B; a.N = N;
    a.thresh = thresh_h(drop_p); a.inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f; a.epoch = seed_epoch;
    switch (ceil_div(N, 64)) {
        case 1: hipLaunchKernelGGL(proj_heads_bwd_kernel<1>, dim3(B, 2), dim3(1024), 0, st, a); break;
        case 2: hipLaunchKernelGGL(proj_heads_bwd_kernel<2>, dim3(B, 2), dim3(1024), 0, st, a); break;
        case 3: hipLaunchKernelGGL(proj_heads_bwd_kernel<3>, dim3(B, 2), dim3(1024), 0, st, a); break;
        default: hipLaunchKernelGGL(proj_heads_bwd_kernel<4>, dim3(B, 2), dim3(1024), 0, st, a); break;
    }
    return mm_check_launch("proj_heads_bwd");
}

int mm_l2norm_fwd(const float* h, float* z, float* nrm, int B, int N, int ldz, hipStream_t st) {
    MM_REQUIRE(h && z && nrm && B > 0 && N > 0 && ldz >= N, "l2norm_fwd: null");
    hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(ceil_div(B, 4)), dim3(256), 0, st, h, z, nrm, B, N, ldz);
    return mm_check_launch("l2norm_fwd");
}

int mm_l2norm_bwd(const float* dz, const float* z, const float* nrm, float* dh, int B, int N, int ldz, hipStream_t st) {
    MM_REQUIRE(dz && z && nrm && dh && B > 0 && N > 0 && ldz >= N, "l2norm_bwd: null");
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(ceil_div(B, 4)), dim3(256), 0, st, dz, z, nrm, dh, B, N, ldz);
    return mm_check_launch("l2norm_bwd");
}

int mm_clip_loss_ws_floats(int B, int Bg, int* floats_host, hipStream_t) {
    MM_REQUIRE(floats_host && B > 0 && Bg >= B, "clip_loss_ws_floats: bad args");
    *floats_host = 6 * Bg;
    return 0;
}

int mm_clip_loss_own_rows(const float* z_all, const float* logit_scale, float* scal4, float* dz_local, float* ws, int B,
                          int Bg, int N, int row0, hipStream_t st) {
    MM_REQUIRE(z_all && logit_scale && scal4 && ws, "clip_loss_own_rows: null");
    MM_REQUIRE(B > 0 && Bg >= B && row0 >= 0 && row0 + B <= Bg && N > 0, "clip_loss_own_rows: B=%d Bg=%d row0=%d", B, Bg, row0);
    MM_REQUIRE(N % 4 == 0, "clip_loss_own_rows: N=%d must be a multiple of 4 (16-byte row loads)", N);
    const size_t lds = (size_t)(2 * N + 2 * Bg + 32) * sizeof(float);
    MM_REQUIRE(lds <= 64 * 1024, "clip_loss_own_rows: N/Bg too large for LDS");
    hipLaunchKernelGGL(clip_lse_kernel, dim3(Bg), dim3(256), lds, st, z_all, logit_scale, ws, Bg, N);
    int rc = mm_check_launch("clip_loss_own_rows(lse)");
    if (rc) return rc;
    hipLaunchKernelGGL(clip_rows_kernel, dim3(B), dim3(256), lds, st, z_all, logit_scale, ws, scal4, dz_local, B, Bg, N, row0);
    return mm_check_launch("clip_loss_own_rows(rows)");
}

int mm_sumsq(const float* g, float* state, int64_t n, hipStream_t st) {
    MM_REQUIRE(g && state && n > 0, "sumsq: null");
    hipLaunchKernelGGL(sumsq_kernel, dim3(grid_h((size_t)n, 1024)), dim3(256), 0, st, g, state, (size_t)n);
    return mm_check_launch("sumsq");
}

int mm_adamw_clip(float* p, float* g, float* m, float* v, float* state, int64_t n, float beta1, float beta2,
                  float eps, float weight_decay, float max_norm, float grad_scale, int zero_grad, uint32_t* seed_epoch,
                  hipStream_t st) {
    MM_REQUIRE(p && g && m && v && state && n > 0, "adamw_clip: null");
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_h((size_t)n)), dim3(256), 0, st, p, g, m, v, state, (size_t)n, beta1,
                       beta2, eps, weight_decay, max_norm, grad_scale, zero_grad);
    hipLaunchKernelGGL(adamw_finish_kernel, dim3(1), dim3(256), 0, st, state, max_norm, grad_scale, seed_epoch);
    return mm_check_launch("adamw_clip");
}

int mm_power_merge(const void* desc_host, int mode, hipStream_t st) {
    MM_REQUIRE(desc_host && mode >= 0 && mode <= 3, "power_merge: null / mode");
    const PowerMergeArgs a = *static_cast<const PowerMergeArgs*>(desc_host);
    MM_REQUIRE(a.cin > 0, "power_merge: cin");
    for (int i = 0; i < 3; ++i) MM_REQUIRE(a.k[i] == 3 || a.k[i] == 5 || a.k[i] == 7, "power_merge: kernel sizes must be 3, 5 or 7");
    if (mode == 0 || mode == 3) {
        for (int i = 0; i < 3; ++i)
            MM_REQUIRE(a.w[i] && a.b[i] && a.gamma[i] && a.beta[i] && a.run_mean[i] && a.run_var[i], "power_merge(0): null part");
        MM_REQUIRE(a.W && a.B && a.Gamma && a.Beta && a.Run_mean && a.Run_var, "power_merge(0): null merged tensor");
        MM_REQUIRE(mode == 0 || (a.cinp >= a.cin && a.cinp % 16 == 0), "power_merge(3): cinp=%d", a.cinp);
    } else if (mode == 1) {
        for (int i = 0; i < 3; ++i) MM_REQUIRE(a.run_mean[i] && a.run_var[i], "power_merge(1): null part");
        MM_REQUIRE(a.Run_mean && a.Run_var, "power_merge(1): null merged statistics");
    }
    const size_t total = (size_t)192 * a.cin * 7;
    const int grid = mode == 1 ? 1 : grid_h(total, 2048);
    if (mode == 3) hipLaunchKernelGGL(power_merge_kernel<3>, dim3(grid), dim3(256), 0, st, a);
    else if (mode == 0) hipLaunchKernelGGL(power_merge_kernel<0>, dim3(grid), dim3(256), 0, st, a);
    else if (mode == 1) hipLaunchKernelGGL(power_merge_kernel<1>, dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(power_merge_kernel<2>, dim3(grid), dim3(256), 0, st, a);
    return mm_check_launch("power_merge");
}

int mm_softmax2_concat(const float* a, const float* c, const float* pa, const float* pc, float* out, int B, int Ha,
                       int Hc, hipStream_t st) {
    MM_REQUIRE(a && c && pa && pc && out && B > 0 && Ha > 0 && Hc > 0, "softmax2_concat: null");
    hipLaunchKernelGGL(softmax2_concat_kernel, dim3(grid_h((size_t)B * (Ha + Hc))), dim3(256), 0, st, a, c, pa, pc, out, B, Ha, Hc);
    return mm_check_launch("softmax2_concat");
}

int mm_learned_fusion(const float* f0, const float* f1, const float* f2, const float* dyn, const float* logits,
                      const float* temperature, float* fused, float* weights, int B, int H, int M, hipStream_t st) {
    MM_REQUIRE(f0 && f1 && dyn && logits && temperature && fused && B > 0 && H > 0, "learned_fusion: null");
    MM_REQUIRE(M >= 2 && M <= 3 && (M == 2 || f2), "learned_fusion: M=%d (2 or 3)", M);
    hipLaunchKernelGGL(learned_fusion_kernel, dim3(ceil_div(B, 4)), dim3(256), 0, st, f0, f1, f2, dyn, logits,
                       temperature, fused, weights, B, H, M);
    return mm_check_launch("learned_fusion");
}

int mm_attn_1x2(const float* proj_e, const float* proj_f, float* ctx, float* attw, int B, int E, int nhead,
                hipStream_t st) {
    MM_REQUIRE(proj_e && proj_f && ctx && attw && B > 0 && nhead > 0 && nhead <= 16 && E % nhead == 0, "attn_1x2: bad args");
    hipLaunchKernelGGL(attn_1x2_kernel, dim3(B), dim3(256), 0, st, proj_e, proj_f, ctx, attw, B, E, nhead);
    return mm_check_launch("attn_1x2");
}

int mm_attn_1x2_train(const float* proj_e, const float* proj_f, const float* dctx, float* ctx, float* attw,
                      float* dproj_e, float* dproj_f, int B, int E, int nhead, float drop_p, uint32_t seed,
                      const uint32_t* seed_epoch, int backward, hipStream_t st) {
    MM_REQUIRE(proj_e && proj_f && B > 0 && nhead > 0 && nhead <= 16 && E % nhead == 0, "attn_1x2_train: bad args");
    MM_REQUIRE(backward ? (dctx && dproj_e && dproj_f) : (ctx != nullptr), "attn_1x2_train: outputs");
    hipLaunchKernelGGL(attn_1x2_fused_kernel, dim3(B), dim3(256), 0, st, proj_e, proj_f, dctx, ctx, attw, dproj_e,
                       dproj_f, B, E, nhead, thresh_h(drop_p), seed, drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f,
                       seed_epoch, backward);
    return mm_check_launch("attn_1x2_train");
}

int mm_learned_fusion_bwd(const float* f0, const float* f1, const float* f2, const float* dyn, const float* logits,
                          const float* temperature, const float* dfused, float* df0, float* df1, float* df2,
                          float* ddyn, float* dlogits, float* dtemp, int B, int H, int M, hipStream_t st) {
    MM_REQUIRE(f0 && f1 && dyn && logits && temperature && dfused && df0 && df1 && ddyn && dlogits && dtemp,
               "learned_fusion_bwd: null");
    MM_REQUIRE(M >= 2 && M <= 3 && (M == 2 || (f2 && df2)), "learned_fusion_bwd: M=%d", M);
    hipLaunchKernelGGL(learned_fusion_bwd_kernel, dim3(1), dim3(1024), 0, st, f0, f1, f2, dyn, logits,
                       temperature, dfused, df0, df1, df2, ddyn, dlogits, dtemp, B, H, M);
    return mm_check_launch("learned_fusion_bwd");
}

int mm_softmax2_concat_bwd(const float* dout, const float* a, const float* c, const float* pa, const float* pc,
                           float* da, float* dc, float* dpa, float* dpc, int B, int Ha, int Hc, hipStream_t st) {
    MM_REQUIRE(dout && a && c && pa && pc && da && dc && dpa && dpc && B > 0, "softmax2_concat_bwd: null");
    hipLaunchKernelGGL(softmax2_concat_bwd_kernel, dim3(1), dim3(1024), 0, st, dout, a, c, pa, pc, da, dc, dpa, dpc, B, Ha, Hc);
    return mm_check_launch("softmax2_concat_bwd");
}

int mm_weighted_ce(const float* logits, const void* target_i64, const float* class_weight, float* loss_out,
                   float* dlogits, int B, int C, hipStream_t st) {
    MM_REQUIRE(logits && target_i64 && loss_out && B > 0 && C > 0, "weighted_ce: bad args");
    hipLaunchKernelGGL(weighted_ce_kernel, dim3(1), dim3(256), 0, st, logits, (const long long*)target_i64, class_weight,
                       loss_out, dlogits, B, C);
    return mm_check_launch("weighted_ce");
}

int mm_focal_loss(const float* logits, const void* target_i64, float* loss_out, float* per_sample, float* dlogits,
                  int B, int C, float alpha, float gamma, float scale, hipStream_t st) {
    MM_REQUIRE(logits && target_i64 && loss_out && B > 0 && C > 0, "focal_loss: bad args");
    hipLaunchKernelGGL(focal_loss_kernel, dim3(1), dim3(256), 0, st, logits, (const long long*)target_i64, loss_out,
                       per_sample, dlogits, B, C, alpha, gamma, scale);
    return mm_check_launch("focal_loss");
}

int mm_gate2_mix(const float* g, const float* erp, const float* pw, const float* conn, float* comb, float* gate, int B,
                 int H, float boost, hipStream_t st) {
    MM_REQUIRE(g && erp && pw && conn && comb && B > 0 && H > 0, "gate2_mix: null");
    hipLaunchKernelGGL(gate2_mix_kernel, dim3(grid_h((size_t)B * 2 * H)), dim3(256), 0, st, g, erp, pw, conn, comb, gate, B, H, boost);
    return mm_check_launch("gate2_mix");
}

int mm_gate2_mix_bwd(const float* dcomb, const float* g, const float* erp, const float* pw, float* derp, float* dpw,
                     float* dconn, float* dg, int B, int H, float boost, hipStream_t st) {
    MM_REQUIRE(dcomb && g && erp && pw && derp && dpw && dconn && dg && B > 0 && H > 0, "gate2_mix_bwd: null");
    hipLaunchKernelGGL(gate2_mix_bwd_kernel, dim3(ceil_div(B, 4)), dim3(256), 0, st, dcomb, g, erp, pw, derp, dpw, dconn, dg, B, H, boost);
    return mm_check_launch("gate2_mix_bwd");
}

int mm_smoothed_ce(const float* logits, const void* target_i64, float* loss_out, float* dlogits, int B, int C,
                   float smoothing, hipStream_t st) {
    MM_REQUIRE(logits && target_i64 && loss_out && B > 0 && C > 0 && smoothing >= 0.f && smoothing < 1.f, "smoothed_ce: bad args");
    hipLaunchKernelGGL(smoothed_ce_kernel, dim3(1), dim3(256), 0, st, logits, (const long long*)target_i64,
                       loss_out, dlogits, B, C, smoothing);
    return mm_check_launch("smoothed_ce");
}

int mm_stft_power(const float* x, void* out_bf16, float* out_f32, int B, int C, int T, int nfft, int hop, int ch_off,
                  int ch_total, hipStream_t st) {
    MM_REQUIRE(x && (out_bf16 || out_f32) && B > 0 && C > 0 && T > 0, "stft_power: null/invalid");
    MM_REQUIRE(nfft >= 8 && nfft <= 1024 && (nfft & (nfft - 1)) == 0 && hop > 0 && T > nfft / 2, "stft_power: nfft=%d hop=%d", nfft, hop);
    const int frames = T / hop + 1;
    const int F = nfft / 2 + 1;
    MM_REQUIRE(ch_off >= 0 && ch_off + C * F <= ch_total, "stft_power: channel window");
    if (nfft <= 256 && !getenv("MM_STFT_DFT")) {                // the FFT form (MM_STFT_DFT=1: the direct DFT, for A/B)
        const dim3 grid(1, C, B);
        switch (nfft) {
            case 8: hipLaunchKernelGGL((stft_power_fft_kernel<3>), grid, dim3(256), 0, st, x, (bf16*)out_bf16, out_f32, C, T, hop, frames, ch_off, ch_total); break;
            case 16: hipLaunchKernelGGL((stft_power_fft_kernel<4>), grid, dim3(256), 0, st, x, (bf16*)out_bf16, out_f32, C, T, hop, frames, ch_off, ch_total); break;
            case 32: hipLaunchKernelGGL((stft_power_fft_kernel<5>), grid, dim3(256), 0, st, x, (bf16*)out_bf16, out_f32, C, T, hop, frames, ch_off, ch_total); break;
            case 64: hipLaunchKernelGGL((stft_power_fft_kernel<6>), grid, dim3(256), 0, st, x, (bf16*)out_bf16, out_f32, C, T, hop, frames, ch_off, ch_total); break;
            case 128: hipLaunchKernelGGL((stft_power_fft_kernel<7>), grid, dim3(256), 0, st, x, (bf16*)out_bf16, out_f32, C, T, hop, frames, ch_off, ch_total); break;
            default: hipLaunchKernelGGL((stft_power_fft_kernel<8>), grid, dim3(256), 0, st, x, (bf16*)out_bf16, out_f32, C, T, hop, frames, ch_off, ch_total); break;
        }
        return mm_check_launch("stft_power");
    }
    const size_t lds = (size_t)(3 * nfft + 8 * nfft) * sizeof(float);
    // (b, c) pairs fill the chip by themselves at the encoder's sizes: one workgroup each walks all its frame blocks; only a
    // small problem is also split over the frame blocks
    const int fblocks = ceil_div(frames, 8);
    const int gx = (long)B * C >= 1024 ? 1 : (fblocks < 8 ? fblocks : 8);
    hipLaunchKernelGGL(stft_power_kernel, dim3(gx, C, B), dim3(256), lds, st, x, (bf16*)out_bf16, out_f32,
                       C, T, nfft, hop, frames, ch_off, ch_total);
    return mm_check_launch("stft_power");
}

int mm_sample_zscore_bf16(const float* x, void* out_bf16, double* ws, int B, int rows, int ch_valid, int ch_total, float eps,
                          hipStream_t st) {
    MM_REQUIRE(x && out_bf16 && B > 0 && rows > 0 && ch_valid > 0 && ch_valid <= ch_total, "sample_zscore: null/invalid");
    const size_t n = (size_t)rows * ch_total;
    if (ws && ch_valid == ch_total && (n & 3) == 0 && n < (1ull << 31) && n >= (1u << 16)) {
        // big unpadded samples: ZS_CHUNKS workgroups per sample, partial sums in ws (MM_ZSCORE_WS_DOUBLES per sample)
        MM_REQUIRE(((uintptr_t)ws & 7) == 0, "sample_zscore: workspace alignment");
        hipLaunchKernelGGL(zscore_partial_kernel, dim3(ZS_CHUNKS, B), dim3(256), 0, st, x, ws, (unsigned)(n >> 2));
        hipLaunchKernelGGL(zscore_apply_kernel, dim3(ZS_CHUNKS, B), dim3(256), 0, st, x, (const double*)ws, (bf16*)out_bf16,
                           (unsigned)(n >> 2), (float)rows * (float)ch_valid, eps);
        return mm_check_launch("sample_zscore(chunked)");
    }
    hipLaunchKernelGGL(sample_zscore_kernel, dim3(B), dim3(1024), 0, st, x, (bf16*)out_bf16, rows, ch_valid, ch_total, eps);
    return mm_check_launch("sample_zscore");
}

int mm_stft_power_bwd(const float* x, const float* g_power, float* dx, int B, int C, int T, int nfft, int hop, int ch_off,
                      int ch_total, hipStream_t st) {
    MM_REQUIRE(x && g_power && dx && B > 0 && C > 0 && T > 0, "stft_power_bwd: null/invalid");
    MM_REQUIRE(nfft >= 8 && nfft <= 1024 && (nfft & (nfft - 1)) == 0 && hop > 0 && T > nfft / 2, "stft_power_bwd: nfft=%d hop=%d", nfft, hop);
    const int frames = T / hop + 1;
    const int F = nfft / 2 + 1;
    MM_REQUIRE(ch_off >= 0 && ch_off + C * F <= ch_total, "stft_power_bwd: channel window");
    const size_t lds = (size_t)(3 * nfft + 8 * nfft + 16 * F + T) * sizeof(float);
    if (lds > 160 * 1024) return mm_fail(MM_ERR_UNSUPPORTED, "stft_power_bwd: T=%d nfft=%d needs %zu bytes of LDS", T, nfft, lds);
    auto kern = stft_power_bwd_kernel;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(C, B), dim3(256), lds, st, x, g_power, dx, C, T, nfft, hop, frames, ch_off, ch_total);
    return mm_check_launch("stft_power_bwd");
}

int mm_sample_zscore_bwd(const float* x, const void* g_bf16, float* dx, int B, int rows, int ch_valid, int ch_total, float eps,
                         hipStream_t st) {
    MM_REQUIRE(x && g_bf16 && dx && B > 0 && rows > 0 && ch_valid > 0 && ch_valid <= ch_total, "sample_zscore_bwd: null/invalid");
    hipLaunchKernelGGL(sample_zscore_bwd_kernel, dim3(B), dim3(1024), 0, st, x, (const bf16*)g_bf16, dx, rows, ch_valid, ch_total, eps);
    return mm_check_launch("sample_zscore_bwd");
}

int mm_attn_1xk(const float* p0, const float* p1, const float* p2, const float* p3, int K, const float* dctx,
                float* ctx, float* attw, float* dp0, float* dp1, float* dp2, float* dp3, int B, int E, int nhead,
                float drop_p, uint32_t seed, const uint32_t* seed_epoch, int backward, hipStream_t st) {
    MM_REQUIRE(K >= 1 && K <= 4 && B > 0 && nhead > 0 && nhead <= 16 && E % nhead == 0, "attn_1xk: K=%d nhead=%d E=%d", K, nhead, E);
    Attn1xKArgs a{};
    const float* p[4] = {p0, p1, p2, p3};
    float* dp[4] = {dp0, dp1, dp2, dp3};
    for (int j = 0; j < K; ++j) {
        MM_REQUIRE(p[j] && (!backward || dp[j]), "attn_1xk: null token %d", j);
        a.p[j] = p[j]; a.dp[j] = dp[j];
    }
    MM_REQUIRE(backward ? dctx != nullptr : ctx != nullptr, "attn_1xk: outputs");
    a.dctx = dctx; a.ctx = ctx; a.attw = attw; a.B = B; a.E = E; a.nhead = nhead; a.K = K;
    a.thresh = thresh_h(drop_p); a.seed = seed; a.inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    a.epoch = seed_epoch; a.backward = backward;
    hipLaunchKernelGGL(attn_1xk_kernel, dim3(B), dim3(256), 0, st, a);
    return mm_check_launch("attn_1xk");
}

int mm_add_f32(const float* a, const float* b, float* out, int64_t n, hipStream_t st) {
    MM_REQUIRE(a && b && out && n > 0, "add_f32: null");
    hipLaunchKernelGGL(add_f32_kernel, dim3(grid_h((size_t)n)), dim3(256), 0, st, a, b, out, (size_t)n);
    return mm_check_launch("add_f32");
}

int mm_mul_f32(const float* a, const float* b, float* out, int64_t n, hipStream_t st) {
    MM_REQUIRE(a && b && out && n > 0, "mul_f32: null");
    hipLaunchKernelGGL(mul_f32_kernel, dim3(grid_h((size_t)n)), dim3(256), 0, st, a, b, out, (size_t)n);
    return mm_check_launch("mul_f32");
}

int mm_pooled_head_fwd(const float* pooled, const float* pooled_acc, const float* W, const float* bias, float* out, void* z_pre_bf16,
                       void* pooled_bf16, int B, int D, int N, int act, float drop_p, uint32_t seed,
                       const uint32_t* seed_epoch, hipStream_t st) {
    MM_REQUIRE((pooled != nullptr) != (pooled_acc != nullptr) && W && out && B > 0, "pooled_head_fwd: null (exactly one of pooled / pooled_acc)");
    MM_REQUIRE(D > 0 && D <= 1024 && D % 16 == 0 && N > 0, "pooled_head_fwd: D=%d (multiple of 16, <= 1024)", D);
    MM_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "pooled_head_fwd: drop_p");
    const uint32_t thresh = drop_p > 0.f ? (uint32_t)((double)drop_p * 4294967296.0) : 0u;
    hipLaunchKernelGGL(pooled_head_fwd_kernel, dim3(B), dim3(256), 0, st, pooled, reinterpret_cast<const mm_acc_t*>(pooled_acc), W, bias, out, (bf16*)z_pre_bf16,
                       (bf16*)pooled_bf16, D, N, act, thresh, drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f, seed, seed_epoch);
    return mm_check_launch("pooled_head_fwd");
}

static int pooled_head_bwd_common(const float* dout, const void* z_pre_bf16, const float* W, void* dz_bf16, float* dx,
                                  void* dx_bf16, float* rows_out, int B, int L, int D, int N, int act, float drop_p,
                                  uint32_t seed, float emit_drop_p, uint32_t emit_seed, const uint32_t* seed_epoch,
                                  hipStream_t st) {
    MM_REQUIRE(dout && z_pre_bf16 && W && (dx || dx_bf16) && B > 0 && L > 0, "pooled_head_bwd: null");
    MM_REQUIRE(D > 0 && D <= 1024 && D % 4 == 0 && N > 0 && N <= 1024 && N % 4 == 0, "pooled_head_bwd: D=%d N=%d", D, N);
    MM_REQUIRE(drop_p >= 0.f && drop_p < 1.f && emit_drop_p >= 0.f && emit_drop_p < 1.f, "pooled_head_bwd: drop_p");
    const int rows = 32;
    const uint32_t t1 = drop_p > 0.f ? (uint32_t)((double)drop_p * 4294967296.0) : 0u;
    const uint32_t t2 = emit_drop_p > 0.f ? (uint32_t)((double)emit_drop_p * 4294967296.0) : 0u;
    hipLaunchKernelGGL(pooled_head_bwd_kernel, dim3(B, (L + rows - 1) / rows), dim3(256), 0, st, dout, (const bf16*)z_pre_bf16, W,
                       (bf16*)dz_bf16, dx, (bf16*)dx_bf16, L, D, N, rows, act, t1, drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f, seed,
                       t2, emit_drop_p > 0.f ? 1.f / (1.f - emit_drop_p) : 1.f, emit_seed, seed_epoch, rows_out);
    return mm_check_launch("pooled_head_bwd");
}

int mm_pooled_head_bwd(const float* dout, const void* z_pre_bf16, const float* W, void* dz_bf16, float* dx, void* dx_bf16,
                       int B, int L, int D, int N, int act, float drop_p, uint32_t seed, float emit_drop_p,
                       uint32_t emit_seed, const uint32_t* seed_epoch, hipStream_t st) {
    return pooled_head_bwd_common(dout, z_pre_bf16, W, dz_bf16, dx, dx_bf16, nullptr, B, L, D, N, act, drop_p, seed, emit_drop_p,
                                  emit_seed, seed_epoch, st);
}

// mm_pooled_head_bwd without the fp32 (B, L, D) token gradients: every token of sample b receives rows_out[b] (B, D), so a
// consumer that takes the row (mm_linear_dgrad_ln_bwd_gemm2's dres_rows_per_sample, mm_bn_act_bwd_*_bcast) needs only that;
// dx_bf16 (B, L, D) = the rows under the consumer's dropout mask (emit_drop_p, emit_seed), as before.
int mm_pooled_head_bwd_rows(const float* dout, const void* z_pre_bf16, const float* W, void* dz_bf16, float* rows_out,
                            void* dx_bf16, int B, int L, int D, int N, int act, float drop_p, uint32_t seed,
                            float emit_drop_p, uint32_t emit_seed, const uint32_t* seed_epoch, hipStream_t st) {
    MM_REQUIRE(rows_out && dx_bf16, "pooled_head_bwd_rows: null");
    return pooled_head_bwd_common(dout, z_pre_bf16, W, dz_bf16, nullptr, dx_bf16, rows_out, B, L, D, N, act, drop_p, seed,
                                  emit_drop_p, emit_seed, seed_epoch, st);
}

int mm_drop_path(const float* x, float* out, int64_t B, int64_t inner, float drop_p, uint32_t seed,
                 const uint32_t* seed_epoch, hipStream_t st) {
    MM_REQUIRE(x && out && B > 0 && inner > 0 && drop_p >= 0.f && drop_p < 1.f, "drop_path: bad args");
    const uint32_t thresh = (uint32_t)fminf(drop_p * 4294967296.f, 4294967295.f);
    hipLaunchKernelGGL(drop_path_kernel, dim3(grid_h((size_t)(B * inner))), dim3(256), 0, st, x, out, (size_t)(B * inner),
                       (size_t)inner, thresh, 1.f / (1.f - drop_p), seed, seed_epoch);
    return mm_check_launch("drop_path");
}

int mm_meanpool_bf16(const void* x, float* out, int R, int S, int N, hipStream_t st) {
    MM_REQUIRE(x && out && R > 0 && S > 0 && N > 0, "meanpool_bf16: null");
    hipLaunchKernelGGL(meanpool_bf16_kernel, dim3(R), dim3(128), 0, st, (const bf16*)x, out, S, N);
    return mm_check_launch("meanpool_bf16");
}

}  // extern "C"
