// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels.
// wave = 64 lanes; MFMA operands bf16, accumulation fp32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define MM_OK 0
#define MM_ERR_ARG (-1)
#define MM_ERR_LAUNCH (-2)
#define MM_ERR_UNSUPPORTED (-3)

// activation codes shared with include/mmeeg_hip.h
#define MM_ACT_NONE 0
#define MM_ACT_GELU 1
#define MM_ACT_RELU 2
#define MM_ACT_TANH 3
#define MM_ACT_SIGMOID 4

int mm_fail(int code, const char* fmt, ...);      // records mm_last_error()
int mm_check_launch(const char* what);            // hipGetLastError -> code

#define MM_REQUIRE(cond, ...)                                   \
    do {                                                        \
        if (!(cond)) return mm_fail(MM_ERR_ARG, __VA_ARGS__);   \
    } while (0)

// exact-erf GELU (nn.GELU() default).  erf by Abramowitz-Stegun 7.1.26 (|abs err| < 1.5e-7,
// i.e. fp32 rounding level): one v_rcp + one v_exp + 5 FMAs instead of ocml's ~30-instruction
// erff; the exp(-x^2/2) is shared with the Gaussian pdf that GELU' needs.
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& pdf_unnorm) {
    const float u = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * u);      // v_rcp_f32 (1 ulp); __frcp_rn() is the ~10-instruction IEEE division
    const float e = __expf(-u * u);                                   // = exp(-x^2 / 2)
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float erfa = 1.0f - poly * e;                               // erf(|x| / sqrt 2)
    cdf = 0.5f * (1.0f + copysignf(erfa, x));
    pdf_unnorm = e;
}
__device__ __forceinline__ float gelu_erf(float x) {
    float cdf, e;
    gelu_parts(x, cdf, e);
    return x * cdf;
}
// d/dx [x * Phi(x)] = Phi(x) + x * phi(x)
__device__ __forceinline__ float gelu_erf_grad(float x) {
    float cdf, e;
    gelu_parts(x, cdf, e);
    return cdf + x * 0.3989422804014327f * e;
}
__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case MM_ACT_GELU: return gelu_erf(v);
        case MM_ACT_RELU: return v > 0.f ? v : 0.f;
        case MM_ACT_TANH: return tanhf(v);
        case MM_ACT_SIGMOID: return 1.0f / (1.0f + __expf(-v));
        default: return v;
    }
}
// derivative w.r.t. the pre-activation z, given z
__device__ __forceinline__ float act_grad(float z, int act) {
    switch (act) {
        case MM_ACT_GELU: return gelu_erf_grad(z);
        case MM_ACT_RELU: return z > 0.f ? 1.f : 0.f;
        case MM_ACT_TANH: { float t = tanhf(z); return 1.f - t * t; }
        case MM_ACT_SIGMOID: { float s = 1.0f / (1.0f + __expf(-z)); return s * (1.f - s); }
        default: return 1.f;
    }
}

// Cross-lane reductions without the LDS crossbar: __shfl_xor() is a ds_bpermute round trip (~100 cycles, and the
// compiler waits for each one), six of them per wave_sum; rows of 16 lanes reduce with four DPP adds (VALU), row
// pairs with v_permlane16_swap, the two halves of the wave with v_permlane32_swap (gfx950).  Every lane of the group
// ends up with the group's total; the pairing is the xor butterfly's (1, 2, 4, 8, 16, 32).
template <int CTRL> __device__ __forceinline__ float dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140;   // quad_perm [1,0,3,2] / [2,3,0,1]
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f32<DPP_XOR1>(v);
    v += dpp_f32<DPP_XOR2>(v);
    v += dpp_f32<DPP_HALF_MIRROR>(v);          // lane i <- lane 7 - i of its half row: the other quad (quads are uniform by now)
    v += dpp_f32<DPP_MIRROR>(v);               // lane i <- lane 15 - i: the other half row
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_f32<DPP_XOR1>(v));
    v = fmaxf(v, dpp_f32<DPP_XOR2>(v));
    v = fmaxf(v, dpp_f32<DPP_HALF_MIRROR>(v));
    v = fmaxf(v, dpp_f32<DPP_MIRROR>(v));
    return v;
}
// sum over each aligned group of 32 lanes (rows 0+1, rows 2+3)
__device__ __forceinline__ float half32_sum(float v) {
    v = row16_sum(v);
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float half32_max(float v) {
    v = row16_max(v);
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float wave_sum(float v) {
    v = half32_sum(v);
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float wave_max(float v) {
    v = half32_max(v);
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// counter-based dropout mask: keep iff hash(seed, idx) >= p * 2^32.
// Same function in forward and backward, so no mask tensor is stored.
// One Weyl multiply (strength-reduced to adds across the consecutive elements a thread handles)
// and ONE xorshift-multiply round: v_mul_lo_u32 is a quarter-rate instruction and the epilogues
// that apply dropout are VALU-bound, so the second round of the usual two-round mixer cost ~15 %
// of them.  Mask statistics (keep rate, lag-1 / lag-32 / row correlations <= 0.005 on 512 x 512
// score tiles at p = 0.1 and 0.3) are at the level of i.i.d. sampling noise.
__device__ __forceinline__ uint32_t mm_hash(uint32_t seed, uint32_t idx) {
    uint32_t x = idx * 0x9E3779B1u + seed;
    x ^= x >> 15;
    x *= 0x2C1B3C6Du;
    x ^= x >> 13;
    return x;
}
__device__ __forceinline__ float dropout_scale(uint32_t seed, uint32_t idx, uint32_t thresh, float inv_keep) {
    return mm_hash(seed, idx) >= thresh ? inv_keep : 0.f;
}

// effective dropout seed: `base` is fixed at launch-record time, `epoch` (device
// word, may be null) changes between replays of a captured hipGraph
__device__ __forceinline__ uint32_t mm_eff_seed(uint32_t base, const uint32_t* epoch) {
    return epoch ? base ^ (epoch[0] * 0x85EBCA6Bu + 0xC2B2AE35u) : base;
}

// BatchNorm + activation [+ pool 2] [+ dropout] backward, element level (elementwise.hip's two passes and the reduce pass
// fused behind a data-gradient GEMM, igemm1d.hip): dz for the (up to) two inputs of one pooled output element.
// Args carries act, pool, drop_first, thresh, seed (already mm_eff_seed'ed), inv_keep.  ACT >= 0 / POOL > 0: compiled for that activation /
// pool size (the per-element switch and the two-way pool logic of the generic form made these passes VALU-bound)
template <int ACT, int POOL, class Args>
__device__ __forceinline__ void bn_dz_pair(const Args& a, float y0, float y1, float sc, float sh, float g,
                                           uint32_t i0, uint32_t i1, uint32_t io, float& dz0, float& dz1) {
    const int act = ACT >= 0 ? ACT : a.act;
    const int pool = POOL > 0 ? POOL : a.pool;
    const float z0 = y0 * sc + sh;
    if (pool == 1) {
        float m = a.thresh ? dropout_scale(a.seed, i0, a.thresh, a.inv_keep) : 1.f;
        dz0 = g * m * act_grad(z0, act);
        dz1 = 0.f;
        return;
    }
    const float z1 = y1 * sc + sh;
    float m0 = 1.f, m1 = 1.f;
    bool first;
    if (a.thresh && a.drop_first) {
        float a0 = apply_act(z0, act), a1 = apply_act(z1, act);
        m0 = dropout_scale(a.seed, i0, a.thresh, a.inv_keep);
        m1 = dropout_scale(a.seed, i1, a.thresh, a.inv_keep);
        a0 *= m0; a1 *= m1;
        first = a0 >= a1;                              // ties -> first (torch max_pool)
    } else {
        if (a.thresh) g *= dropout_scale(a.seed, io, a.thresh, a.inv_keep);
        // GELU rises on [0, inf) and is negative below 0: when the larger pre-activation is >= 0 it holds the larger
        // activation and nothing needs evaluating; only a pair of negative values does (the falling branch can win)
        first = z0 >= z1;
        if (act != MM_ACT_GELU || fmaxf(z0, z1) < 0.f) first = apply_act(z0, act) >= apply_act(z1, act);
    }
    // ONE derivative, at the winner (two selects of act_grad(z0) / act_grad(z1) evaluate both)
    const float d = g * (first ? m0 : m1) * act_grad(first ? z0 : z1, act);
    dz0 = first ? d : 0.f;
    dz1 = first ? 0.f : d;
}


// Per-channel reductions (BN statistics, dgamma/dbeta, dbias) are accumulated across workgroups with
// 64-bit INTEGER atomics on fixed-point values: integer addition is associative, so the sum does not
// depend on the order the workgroups arrive in and a training step is bit-reproducible (fp32 atomics
// are not).  A contribution v (already a block-level partial sum, formed in a fixed order) is added as
// rint(v * 2^K); the consumer converts sum * 2^-K back to fp32 once.
//   K = MM_ACC_STAT (28) for sums of activations / their squares: resolution 3.7e-9, range +-8.6e9
//   K = MM_ACC_GRAD (40) for sums of gradients:                     resolution 9.1e-13, range +-2.1e6
// Hundreds of workgroups adding to the SAME address serialise in L2 (~25 ns each), so every such
// accumulator is replicated: callers allocate (and zero) MM_REPL fp32-sized copies = MM_ACC_REPL
// 64-bit ones; a workgroup adds into replica (blockIdx.x % MM_ACC_REPL) and the consumer sums the
// replicas (as integers: also order-free).  The C ABI passes these workspaces as `float*` of
// MM_REPL * n elements; their content is opaque to the caller (zero it, hand it to the consumer).
#define MM_REPL 32
#define MM_ACC_REPL 16
#define MM_ACC_STAT 28
#define MM_ACC_GRAD 40
typedef long long mm_acc_t;
// Overflow / non-finite contract (fp32 atomics degraded to inf / NaN by themselves; integers wrap): a contribution
// whose fixed-point image is not below 2^55 in magnitude (|v| >= 1.3e8 for STAT, 32 768 for GRAD - or NaN / Inf)
// does not add: it EXCHANGES the accumulator for the poison value 2^62, which no run of in-range contributions can
// move out of the poisoned band (64 full-size ones of one sign per replica would be needed - 1 024 workgroups each
// at the limit, whose true sum is out of range anyway; a wrap would need 256 per replica).  A consumer sums the replicas as
// integers and, beside that, their magnitudes in units of 2^36: when those reach 2^61 a replica is poisoned or the
// integer sum may have wrapped, and the consumer's value is NaN (acc_val of MM_ACC_BAD) - BatchNorm statistics,
// gradients and the loss then go NaN exactly as they would have with floating-point sums, never to wrapped
// garbage.  Exact range of a sum: +-8.6e9 (STAT), +-2.1e6 (GRAD).  tests/test_kernels_gpu.py::test_accumulator_*.
#define MM_ACC_POISON (1ll << 62)
#define MM_ACC_BAD ((mm_acc_t)0x8000000000000000ull)
#define MM_ACC_MAG_LIMIT ((1u << 25) - 16u)
#ifdef __HIPCC__
template <int K> __device__ __forceinline__ void acc_add(mm_acc_t* p, float v) {
    const float f = v * (float)(1ull << K);
    if (fabsf(f) < 36028797018963968.f)                 // 2^55; false for NaN
        atomicAdd(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__float2ll_rn(f));
    else
        atomicExch(reinterpret_cast<unsigned long long*>(p), (unsigned long long)MM_ACC_POISON);
}
// a whole value written by one thread (no accumulation): same contract
template <int K> __device__ __forceinline__ mm_acc_t acc_encode(float v) {
    const float f = v * (float)(1ull << K);
    return fabsf(f) < 2305843009213693952.f ? __float2ll_rn(f) : MM_ACC_POISON;      // 2^61
}
// magnitude of one replica in units of 2^36 (floor), for the guard sums
__device__ __forceinline__ unsigned acc_mag(mm_acc_t v) {
    const int hi = (int)(v >> 32);
    return (unsigned)(hi < 0 ? ~hi : hi) >> 4;
}
__device__ __forceinline__ mm_acc_t acc_guard(mm_acc_t s, unsigned mag_sum) { return mag_sum >= MM_ACC_MAG_LIMIT ? MM_ACC_BAD : s; }
template <int K> __device__ __forceinline__ float acc_val(mm_acc_t s) {
    // a single accumulator (or a guarded sum): outside [-2^61, 2^61) = poisoned / flagged
    const bool bad = (unsigned long long)(s + (1ll << 61)) >= (1ull << 62);
    return bad ? __builtin_nanf("") : (float)s * (1.0f / (float)(1ull << K));
}
// replica r of an accumulator workspace of n values per replica
__device__ __forceinline__ mm_acc_t* acc_rep(float* ws, int r, size_t n) { return reinterpret_cast<mm_acc_t*>(ws) + (size_t)r * n; }
// sum over the MM_ACC_REPL replicas of element i (all loads in flight at once); MM_ACC_BAD if poisoned / out of range
__device__ __forceinline__ mm_acc_t acc_sum(const float* ws, size_t n, size_t i) {
    const mm_acc_t* p = reinterpret_cast<const mm_acc_t*>(ws) + i;
    mm_acc_t v[MM_ACC_REPL];
#pragma unroll
    for (int r = 0; r < MM_ACC_REPL; ++r) v[r] = p[(size_t)r * n];
    mm_acc_t s = 0;
    unsigned g = 0;
#pragma unroll
    for (int r = 0; r < MM_ACC_REPL; ++r) { s += v[r]; g += acc_mag(v[r]); }
    return acc_guard(s, g);
}
// the same sum with one replica per lane: 16 adjacent lanes hold the 16 replicas of one element
__device__ __forceinline__ mm_acc_t acc_sum_lanes16(mm_acc_t v) {
    unsigned g = acc_mag(v);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) { v += __shfl_xor(v, o, 64); g += __shfl_xor(g, o, 64); }
    return acc_guard(v, g);
}
#endif

// ---- BatchNorm finalize (train mode), shared by mm_bn_finalize and by the consumers that carry it in their prologue
// (mm_bn_act_fwd_fin, mm_pool3d_bn_act_fwd_fin, mm_conv3d_l1_fwd_fin): ONE function, so that both forms produce the
// same bits.  Host side: mm_bn_fin_t (include/mmeeg_hip.h) has this layout.
struct MmBnFin {
    const float* stats;        // accumulator workspace [MM_REPL][2][N]: {sum, sumsq}
    const float* gamma; const float* beta;
    float* run_mean; float* run_var;
    float* out4;               // [4][N]: scale, shift, mean, rstd (written by the launch's workgroup 0)
    long long* tracked;        // num_batches_tracked (nullable)
    float count, momentum, eps;
    int N;
};
#ifdef __HIPCC__
// channel n: {scale, shift, mean, rstd} from the fixed-point sums; `writer` (one thread per channel in the whole launch)
// also updates the running statistics and stores the four values
__device__ __forceinline__ void bn_fin_channel(const MmBnFin& f, int n, bool writer, float& sc, float& sh, float& mean, float& rstd) {
    const float s1 = acc_val<MM_ACC_STAT>(acc_sum(f.stats, 2 * (size_t)f.N, n));
    const float s2 = acc_val<MM_ACC_STAT>(acc_sum(f.stats, 2 * (size_t)f.N, (size_t)f.N + n));
    mean = s1 / f.count;
    float var = s2 / f.count - mean * mean;
    var = var < 0.f ? 0.f : var;                         // (not fmaxf: a NaN sum - an accumulator out of range - must stay NaN)
    rstd = rsqrtf(var + f.eps);
    sc = f.gamma[n] * rstd;
    sh = f.beta[n] + (0.f - mean) * sc;
    if (writer) {
        if (n == 0 && f.tracked) f.tracked[0] += 1;
        f.run_mean[n] = (1.f - f.momentum) * f.run_mean[n] + f.momentum * mean;
        const float unb = f.count > 1.f ? var * f.count / (f.count - 1.f) : var;
        f.run_var[n] = (1.f - f.momentum) * f.run_var[n] + f.momentum * unb;
        f.out4[n] = sc; f.out4[f.N + n] = sh; f.out4[2 * f.N + n] = mean; f.out4[3 * f.N + n] = rstd;
    }
}
#endif
// host layout of the descriptor the *_fin entry points read (copied into the kernel's arguments at launch)
struct MmBnFinHost {
    const float* stats; const float* gamma; const float* beta; float* run_mean; float* run_var; float* out4;
    void* batches_tracked; float count, momentum, eps; int reserved;
};
static inline bool bn_fin_from_host(MmBnFin& f, const void* host, int N) {
    if (!host) return false;
    const MmBnFinHost& h = *static_cast<const MmBnFinHost*>(host);
    if (!h.stats || !h.gamma || !h.beta || !h.run_mean || !h.run_var || !h.out4 || !(h.count >= 1.f)) return false;
    f.stats = h.stats; f.gamma = h.gamma; f.beta = h.beta; f.run_mean = h.run_mean; f.run_var = h.run_var; f.out4 = h.out4;
    f.tracked = (long long*)h.batches_tracked; f.count = h.count; f.momentum = h.momentum; f.eps = h.eps; f.N = N;
    return true;
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// ---- layout of a 3-D convolution weight image (k = 27).  Images are made by mm_prep_conv_weight / mm_prep_many and
// consumed by mm_conv3d_fwd only, so the two agree on this function.  Plain: [row n][tap][channel c].  For the shapes
// csrc/conv3d_stream.hip serves - (rows, channels) = (128, 64), (64, 128), (32, 64) - the image is stored in the lane
// order of that kernel's B-operand loads: unit g = tap * (ch / 32) + c / 32, column group grp = n / 32, MFMA column
// tile j = n % 2, lane column lc = (n % 32) / 2, lane group lg = (c % 32) / 8:
//   position = ((((g * (rows / 32) + grp) * 2 + j) * 64 + 16 lg + lc) * 8 + c % 8      (one wave load = 1 KB contiguous)
__host__ __device__ __forceinline__ bool conv3d_stream_shape(int rows, int ch) {
    return (rows == 128 && ch == 64) || (rows == 64 && ch == 128) || (rows == 32 && ch == 64);
}
__host__ __device__ __forceinline__ int conv_image_index(int rows, int k, int ch, int n, int tap, int c) {
    if (k != 27 || !conv3d_stream_shape(rows, ch)) return (n * k + tap) * ch + c;
    const int g = tap * (ch / 32) + c / 32, grp = n / 32, local = n % 32;
    return ((((g * (rows / 32) + grp) * 2 + (local & 1)) * 64 + 16 * ((c % 32) / 8) + (local >> 1)) * 8) + c % 8;
}
