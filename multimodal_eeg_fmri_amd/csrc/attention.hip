// Multi-head self-attention for head_dim 32 on bf16 MFMA, fp32 softmax.
//
// qkv : [B][L][3E] bf16 (nn.MultiheadAttention packed in_proj order q|k|v, head
//       h owns columns h*32 .. h*32+31 of each E-wide third),  E = H*32
// out : [B][L][E]  bf16   (heads concatenated, ready for out_proj)
// lse : [B][H][L]  fp32   log-sum-exp of the scaled scores (for backward)
//
// One workgroup = 4 waves = 128 queries of one (b, h); each wave owns 32
// queries.  Keys/values of the (b, h) pair are staged in LDS in chunks of KCH
// keys: K row-major [key][32] (+16 B pad), V row-major too but read TRANSPOSED by
// ds_read_b64_tr_b16 (tr_frag32), its rows inside each 16-key group permuted so that
// the P^T accumulator tile of the first MFMA is directly the B operand of the second
// (no lane movement):
//     S^T[key][q] = K_tile . Q^T          (A = K rows, B = Q^T from registers)
//     O^T[d][q]  += V^T[d][key] . P^T     (A = V^T rows, B = exp(S^T) as bf16)
// Each lane therefore owns one query column: the row max / row sum are 16
// in-register values plus one exchange with lane^32.
#include "common.h"

namespace {

constexpr int DH = 32;
constexpr int KCH = 128;                 // keys staged per chunk
#ifndef ATTN_PIPE_BWD
#define ATTN_PIPE_BWD 0                  // 1: the backward kernels prefetch the next tile's fragments too (A/B: stand-alone neutral, the STEP 1 % slower)
#endif
constexpr int KS = DH + 8;               // K row stride (elements): 80 B

constexpr int VR = DH;                   // row stride (elements) of a row-major tile read through ds_read_b64_tr_b16: 64 B, NO
                                         // padding.  The instruction is served in two groups of 32 lanes; a group reads four
                                         // rows x two 32-byte column halves, i.e. eight 8-bank windows at (16 row + 8 half)
                                         // mod 64 - all distinct.  (The 96-byte stride of round 2 put row 3 / half 0 on the
                                         // banks of row 0 / half 1: SQ_LDS_BANK_CONFLICT = 0.36 of the LDS cycles,
                                         // profiles/r03_pmc_attn.summary.txt)

// MFMA A operand of a TRANSPOSED product from a row-major LDS tile [k][32] (row stride VR): row index = lane & 31 =
// tile column, k-slots 8 (lane >> 5) + {0..7} = tile rows row0 + 8 (lane >> 5) + {0..7} (hardware transpose, as in
// the weight-gradient kernels) - no scattered 2-byte writes into a transposed copy (they were 0.4 of the LDS cycles)
__device__ __forceinline__ bf16x8 tr_frag32(const bf16* tile, int row0, int lane) {
    const int li = lane & 15, g = lane >> 4;
    const bf16* p = tile + (row0 + 8 * (g >> 1) + (li >> 2)) * VR + (g & 1) * 16 + 4 * (li & 3);
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    union { s16x4 s[2]; bf16x8 v; } u;
    u.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
    u.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * VR));
    return u.v;
}

__device__ __forceinline__ int vperm(int key) {       // swap bits 2 and 3 of the key index
    return (key & ~12) | ((key & 4) << 1) | ((key & 8) >> 1);
}

// attention-probability dropout (nn.MultiheadAttention(dropout=p)): the softmax
// row sum uses the un-dropped probabilities, only the P operand of P.V is masked.
// The mask used to cost a quarter of these kernels (hash ~9 of ~20 VALU instructions per score, its 32-bit multiply
// quarter-rate; the dkv kernel, whose registers run along the queries, could not share the per-key-pair hash of the
// other two and hashed per score).  Now ONE hash serves the 2 x 2 block (queries 2i, 2i + 1) x (keys 2j, 2j + 1): byte
// 2 (q & 1) + (key & 1) of the word decides the score (keep iff byte >= round(p * 256): p is honoured to 1/256 and
// the keep scale is 256 / (256 - t), so the mask stays unbiased for the quantised p).  Every kernel then spends one
// hash per two scores whichever way its registers run: forward / dq lanes own a query and hold the two keys of a
// block in registers r, r + 1; dkv lanes own a key and hold the two queries.  The mixer's multiply is the full-rate
// 24-bit one (v_mul_u32_u24; constants chosen on the mask statistics: keep rate to 7e-4, lag / diagonal / head
// correlations <= 0.0022, row- and column-sum variance 0.97-1.04 of binomial; oracle/dropout_replica.py is the host
// replica).  Block index = (bh * ceil(L / 2) + q / 2) * ceil(L / 2) + key / 2.
__device__ __forceinline__ uint32_t attn_block_hash(uint32_t seed, int bh, int q, int key, int L) {
    const uint32_t Lh = ((uint32_t)L + 1u) >> 1;
    uint32_t x = (((uint32_t)bh * Lh + ((uint32_t)q >> 1)) * Lh + ((uint32_t)key >> 1)) * 0x9E3779B1u + seed;
    x ^= x >> 13;
    x = __umul24(x, 0xB5297Bu);
    x ^= x >> 15;
    return x;
}
// the two scores of a block that ONE lane owns: `mine` = the lane's own index (query in forward / dq, key in dkv),
// `other` = the EVEN index of the register pair (keys 2j, 2j + 1 resp. queries 2i, 2i + 1).  along_keys: the pair runs
// along the keys (forward, dq) or along the queries (dkv).
template <bool ALONG_KEYS>
__device__ __forceinline__ void attn_keep2(uint32_t seed, int bh, int mine, int other, int L, uint32_t thresh8, bool& k0, bool& k1) {
    if (ALONG_KEYS) {
        const uint32_t x = attn_block_hash(seed, bh, mine, other, L) >> (16 * (mine & 1));
        k0 = (x & 0xFFu) >= thresh8;
        k1 = ((x >> 8) & 0xFFu) >= thresh8;
    } else {
        const uint32_t x = attn_block_hash(seed, bh, other, mine, L) >> (8 * (mine & 1));
        k0 = (x & 0xFFu) >= thresh8;
        k1 = ((x >> 16) & 0xFFu) >= thresh8;
    }
}

// exchange between the two halves of a wave (lanes l and l ^ 32) without an LDS round trip: v_permlane32_swap (gfx950)
// leaves the lower half of its first operand / the upper half of its second in both halves
__device__ __forceinline__ float xhalf_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xhalf_sum(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// v_exp_f32 as is: arguments are <= 0 here and a result below 2^-126 may flush to zero (softmax
// weights); exp2f() wraps the instruction in a compare / two selects / add / ldexp for denormal
// results, i.e. 6 extra VALU instructions per score in loops that are VALU-bound.
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// DROP: attention-probability dropout compiled in (no per-score branch); FULL: L is a multiple of
// the key and query chunk sizes, so no score needs a validity mask (3 VALU instructions each)
// MASK: additive fp32 attention mask [L][L] (nn.MultiheadAttention attn_mask; -inf = not allowed), general path only
template <bool DROP, bool FULL, bool MASK = false>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out,
                                                       float* __restrict__ lse, int L, int H, float scale_log2,
                                                       uint32_t dthresh, uint32_t dseed, float dinv,
                                                       const uint32_t* epoch, const float* __restrict__ amask, size_t amask_bh) {
    dseed = mm_eff_seed(dseed, epoch);
    if (MASK) amask += (size_t)(blockIdx.z * H + blockIdx.y) * amask_bh;      // (B*H, L, L) form: one matrix per (batch, head); 0 = shared
    __shared__ __attribute__((aligned(16))) bf16 Ks[KCH * KS];
    __shared__ __attribute__((aligned(16))) bf16 Vs[KCH * VR];      // V row-major, row = vperm(key): read transposed by tr_frag32
    const int E = H * DH, E3 = 3 * E;
    const int b = blockIdx.z, h = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int q = blockIdx.x * 128 + wave * 32 + lr;
    const bf16* base = qkv + (size_t)b * L * E3 + h * DH;

    // Q^T fragments (B operand): lane holds Q[q][16s + 8*lh .. +8]
    bf16x8 qf[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        if (q < L) qf[s] = *reinterpret_cast<const bf16x8*>(base + (size_t)q * E3 + 16 * s + 8 * lh);
        else
#pragma unroll
            for (int j = 0; j < 8; ++j) qf[s][j] = (bf16)0.f;
    }
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // K / V chunks: global -> registers -> LDS, with the NEXT chunk's loads issued before the current chunk's tiles are
    // computed (a chunk is ~2 us of work, a load round trip from L2 / MALL ~1.5 us: exposed four times per workgroup it
    // was a third of the kernel, profiles/r03_attention_ab.txt)
    constexpr int NI = KCH * 4 / 256;
    uint4 kreg[NI], vreg[NI];
    auto load_chunk = [&](int k0) __attribute__((always_inline)) {
        const int kn = min(KCH, L - k0);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int s = tid + i * 256, key = s >> 2, sg = s & 3;
            kreg[i] = make_uint4(0, 0, 0, 0); vreg[i] = make_uint4(0, 0, 0, 0);
            if (key < kn) {
                const bf16* row = base + (size_t)(k0 + key) * E3;
                kreg[i] = *reinterpret_cast<const uint4*>(row + E + sg * 8);
                vreg[i] = *reinterpret_cast<const uint4*>(row + 2 * E + sg * 8);
            }
        }
    };
    load_chunk(0);
    // the query fragments are complete from here on, and the compiler must know it: their first use would otherwise carry
    // an s_waitcnt vmcnt(0) INSIDE the tile loop (the waits in front of the LDS writes sit in divergent blocks), which
    // would also wait for the prefetched chunk
    asm volatile("" : : "v"(qf[0]), "v"(qf[1]));
    for (int k0 = 0; k0 < L; k0 += KCH) {
        const int kn = min(KCH, L - k0);
        const int kn32 = (kn + 31) & ~31;
        __syncthreads();
        // K rows and V rows (zero-padded to a multiple of 32 keys) of this chunk
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int s = tid + i * 256, key = s >> 2, sg = s & 3;
            if (key >= kn32) continue;
            *reinterpret_cast<uint4*>(Ks + key * KS + sg * 8) = kreg[i];
            *reinterpret_cast<uint4*>(Vs + vperm(key) * VR + sg * 8) = vreg[i];
        }
        __syncthreads();
        if (k0 + KCH < L) load_chunk(k0 + KCH);
        // software pipeline over the 32-key tiles (PMC, profiles/r03_pmc_attn.summary.txt: 43-53 % of the wave cycles were
        // spent parked on s_waitcnt - fragment reads issued right in front of their MFMA, at two waves per SIMD): the K
        // fragments of tile t + 1 and the V fragments of tile t are requested before tile t's softmax
        bf16x8 kfr[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) kfr[s] = *reinterpret_cast<const bf16x8*>(Ks + lr * KS + 16 * s + 8 * lh);
        for (int kt = 0; kt < kn32; kt += 32) {
            f32x16 sacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 2; ++s) sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[s], qf[s], sacc, 0, 0, 0);
            bf16x8 vfr[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) vfr[s] = tr_frag32(Vs, kt + 16 * s, lane);
            if (kt + 32 < kn32)
#pragma unroll
                for (int s = 0; s < 2; ++s) kfr[s] = *reinterpret_cast<const bf16x8*>(Ks + (kt + 32 + lr) * KS + 16 * s + 8 * lh);
            // scores: raw MFMA output (the softmax scale rides in the exp's FMA: exp2(s * scale - m * scale)); with an additive
            // mask they are formed in scaled log2 units instead (sc2 = 1).  Padded keys -> -inf.
            const float sc2 = MASK ? 1.f : scale_log2;
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt + (r & 3) + 8 * (r >> 2) + 4 * lh;
                float sc = sacc[r];
                if (MASK) {
                    sc *= scale_log2;
                    if (key < kn) sc += amask[(size_t)min(q, L - 1) * L + k0 + key] * 1.4426950408889634f;
                }
                sacc[r] = (FULL || key < kn) ? sc : -INFINITY;
                mx = fmaxf(mx, sacc[r]);
            }
            mx = xhalf_max(mx);
            // lazy running maximum: the accumulators are rescaled only when some row's maximum grew by more than 2^8 (in the
            // exp's units) - p <= 256 is as exact in bf16 / fp32 as p <= 1 - so after the first tile of a row the 16
            // accumulator reads, multiplies and writes per tile are skipped (wave-uniform branch)
            if (__builtin_amdgcn_ballot_w64(mx * sc2 > m_run + 8.f)) {
                const float m_new = fmaxf(m_run, mx * sc2);
                // a row whose keys so far are all masked out keeps m = -inf: subtract 0 instead (exp2(-inf) = 0)
                const float alpha = fast_exp2(m_run - ((MASK && m_new == -INFINITY) ? 0.f : m_new));  // m_run = -inf first time -> 0
                l_run *= alpha;
                m_run = m_new;
#pragma unroll
                for (int r = 0; r < 16; ++r) o[r] *= alpha;
            }
            const float m_neg = (MASK && m_run == -INFINITY) ? 0.f : -m_run;
            float ps = 0.f;
            union { bf16x8 v[2]; bf16x2 h[8]; } pu;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {                    // registers r, r + 1: keys 2j, 2j + 1
                float p0 = fast_exp2(fmaf(sacc[r], sc2, m_neg)), p1 = fast_exp2(fmaf(sacc[r + 1], sc2, m_neg));
                ps += p0;
                ps += p1;
                if (DROP) {                                      // 0 / 1 mask; the keep scale multiplies the final normaliser
                    bool kp0, kp1;
                    attn_keep2<true>(dseed, b * H + h, q, k0 + kt + (r & 3) + 8 * (r >> 2) + 4 * lh, L, dthresh, kp0, kp1);
                    p0 = kp0 ? p0 : 0.f; p1 = kp1 ? p1 : 0.f;
                }
                typedef __attribute__((ext_vector_type(2))) float f32x2_t;
                pu.h[r >> 1] = __builtin_convertvector((f32x2_t){p0, p1}, bf16x2);      // one v_cvt_pk_bf16_f32
            }
            l_run += ps;
            const bf16x8* pf = pu.v;
#pragma unroll
            for (int s = 0; s < 2; ++s) o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[s], pf[s], o, 0, 0, 0);
        }
    }
    const float l_tot = xhalf_sum(l_run);
    const float inv = (DROP ? dinv : 1.f) / l_tot;
    if (q < L) {
        bf16* orow = out + ((size_t)b * L + q) * E + h * DH;
#pragma unroll
        for (int g = 0; g < 4; ++g) {                       // rows d = 8g + 4*lh + {0..3}
            bf16x4 v = {(bf16)(o[4 * g] * inv), (bf16)(o[4 * g + 1] * inv), (bf16)(o[4 * g + 2] * inv), (bf16)(o[4 * g + 3] * inv)};
            *reinterpret_cast<bf16x4*>(orow + 8 * g + 4 * lh) = v;
        }
        if (lse && lh == 0) lse[((size_t)b * H + h) * L + q] = (m_run + log2f(l_tot)) * 0.6931471805599453f;    // m_run is in scaled log2 units
    }
}

// ---------------------------------------------------------------------------
// Backward.  P = exp(scale*S - lse), dS = P o (dP - delta), delta = rowsum(dO o O)
//   dQ = scale * dS K      dK = scale * dS^T Q      dV = P^T dO
// Two passes, no atomics (bit-reproducible):
//   dq kernel : one WG = 128 queries, sweeps all keys (K, V, K^T in LDS);
//               also writes delta[b][h][q] for the second pass.
//   dkv kernel: one WG = 128 keys (K, V fragments in registers), sweeps all
//               queries (Q, dO, Q^T, dO^T, lse, delta in LDS).
// In both, the first product is oriented so that its accumulator tile is the
// B operand of the following products (rows = reduction index).
// ---------------------------------------------------------------------------
template <bool DROP, bool FULL, bool MASK = false>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ out,
                                                          const bf16* __restrict__ dout, const float* __restrict__ lse,
                                                          bf16* __restrict__ dqkv, float* __restrict__ delta,
                                                          int L, int H, float scale, uint32_t dthresh,
                                                          uint32_t dseed, float dinv, const uint32_t* epoch,
                                                          const float* __restrict__ amask, size_t amask_bh) {
    dseed = mm_eff_seed(dseed, epoch);
    if (MASK) amask += (size_t)(blockIdx.z * H + blockIdx.y) * amask_bh;
    __shared__ __attribute__((aligned(16))) bf16 Ks[KCH * KS];
    __shared__ __attribute__((aligned(16))) bf16 Vs[KCH * KS];
    __shared__ __attribute__((aligned(16))) bf16 Kr[KCH * VR];      // K again, row = vperm(key), for the transposed product (tr_frag32)
    const int E = H * DH, E3 = 3 * E;
    const int b = blockIdx.z, h = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int q = blockIdx.x * 128 + wave * 32 + lr;
    const bool qok = q < L;
    const bf16* base = qkv + (size_t)b * L * E3 + h * DH;
    const float scale_log2 = scale * 1.4426950408889634f;

    bf16x8 qf[2], dof[2];
    float dl = 0.f;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { qf[s][j] = (bf16)0.f; dof[s][j] = (bf16)0.f; }
        if (qok) {
            qf[s] = *reinterpret_cast<const bf16x8*>(base + (size_t)q * E3 + 16 * s + 8 * lh);
            const size_t oi = ((size_t)b * L + q) * E + h * DH + 16 * s + 8 * lh;
            dof[s] = *reinterpret_cast<const bf16x8*>(dout + oi);
            const bf16x8 of = *reinterpret_cast<const bf16x8*>(out + oi);
#pragma unroll
            for (int j = 0; j < 8; ++j) dl += (float)dof[s][j] * (float)of[j];
        }
    }
    dl = xhalf_sum(dl);
    const float lse2 = qok ? lse[((size_t)b * H + h) * L + q] * 1.4426950408889634f : 0.f;
    if (qok && lh == 0) delta[((size_t)b * H + h) * L + q] = dl;

    f32x16 dq;
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[r] = 0.f;

    constexpr int NI = KCH * 4 / 256;                    // next chunk prefetched into registers, as in the forward
    uint4 kreg[NI], vreg[NI];
    auto load_chunk = [&](int k0) __attribute__((always_inline)) {
        const int kn = min(KCH, L - k0);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int s = tid + i * 256, key = s >> 2, sg = s & 3;
            kreg[i] = make_uint4(0, 0, 0, 0); vreg[i] = make_uint4(0, 0, 0, 0);
            if (key < kn) {
                const bf16* row = base + (size_t)(k0 + key) * E3;
                kreg[i] = *reinterpret_cast<const uint4*>(row + E + sg * 8);
                vreg[i] = *reinterpret_cast<const uint4*>(row + 2 * E + sg * 8);
            }
        }
    };
    load_chunk(0);
    asm volatile("" : : "v"(qf[0]), "v"(qf[1]), "v"(dof[0]), "v"(dof[1]), "v"(dl), "v"(lse2));      // (as in the forward)
    for (int k0 = 0; k0 < L; k0 += KCH) {
        const int kn = min(KCH, L - k0);
        const int kn32 = (kn + 31) & ~31;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int s = tid + i * 256, key = s >> 2, sg = s & 3;
            if (key >= kn32) continue;
            *reinterpret_cast<uint4*>(Ks + key * KS + sg * 8) = kreg[i];
            *reinterpret_cast<uint4*>(Vs + key * KS + sg * 8) = vreg[i];
            *reinterpret_cast<uint4*>(Kr + vperm(key) * VR + sg * 8) = kreg[i];
        }
        __syncthreads();
        if (k0 + KCH < L) load_chunk(k0 + KCH);
        // software pipeline as in the forward: K / V fragments of tile t + 1 and the transposed K fragments of tile t are
        // requested before tile t's element-wise work
        bf16x8 kfr[2], vfr[2];
        if (ATTN_PIPE_BWD)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                kfr[s] = *reinterpret_cast<const bf16x8*>(Ks + lr * KS + 16 * s + 8 * lh);
                vfr[s] = *reinterpret_cast<const bf16x8*>(Vs + lr * KS + 16 * s + 8 * lh);
            }
        for (int kt = 0; kt < kn32; kt += 32) {
            f32x16 sacc, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; dp[r] = 0.f; }
            if (!ATTN_PIPE_BWD)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    kfr[s] = *reinterpret_cast<const bf16x8*>(Ks + (kt + lr) * KS + 16 * s + 8 * lh);
                    vfr[s] = *reinterpret_cast<const bf16x8*>(Vs + (kt + lr) * KS + 16 * s + 8 * lh);
                }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[s], qf[s], sacc, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[s], dof[s], dp, 0, 0, 0);
            }
            bf16x8 ktfr[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) ktfr[s] = tr_frag32(Kr, kt + 16 * s, lane);
            if (ATTN_PIPE_BWD && kt + 32 < kn32)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    kfr[s] = *reinterpret_cast<const bf16x8*>(Ks + (kt + 32 + lr) * KS + 16 * s + 8 * lh);
                    vfr[s] = *reinterpret_cast<const bf16x8*>(Vs + (kt + 32 + lr) * KS + 16 * s + 8 * lh);
                }
            bf16x8 dsf[2];
            bool kp[16];
            if (DROP)
#pragma unroll
                for (int r = 0; r < 16; r += 2)                  // registers r, r + 1: keys 2j, 2j + 1 of one block
                    attn_keep2<true>(dseed, b * H + h, q, k0 + kt + (r & 3) + 8 * (r >> 2) + 4 * lh, L, dthresh, kp[r], kp[r + 1]);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt + (r & 3) + 8 * (r >> 2) + 4 * lh;
                float sc = fmaf(sacc[r], scale_log2, -lse2);
                if (MASK && key < kn) sc += amask[(size_t)min(q, L - 1) * L + k0 + key] * 1.4426950408889634f;
                const float p = (FULL || key < kn) ? fast_exp2(sc) : 0.f;
                float dpr = dp[r];
                if (DROP) dpr = fmaf(kp[r] ? dpr : 0.f, dinv, -dl);
                else dpr -= dl;
                dsf[r >> 3][r & 7] = (bf16)(p * dpr);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktfr[s], dsf[s], dq, 0, 0, 0);
        }
    }
    if (qok) {
        bf16* drow = dqkv + ((size_t)b * L + q) * E3 + h * DH;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bf16x4 v = {(bf16)(dq[4 * g] * scale), (bf16)(dq[4 * g + 1] * scale), (bf16)(dq[4 * g + 2] * scale), (bf16)(dq[4 * g + 3] * scale)};
            *reinterpret_cast<bf16x4*>(drow + 8 * g + 4 * lh) = v;
        }
    }
}

constexpr int QCH = 128;                 // queries staged per chunk in the dK/dV pass


template <bool DROP, bool FULL, bool MASK = false>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ dout,
                                                           const float* __restrict__ lse, const float* __restrict__ delta,
                                                           bf16* __restrict__ dqkv, int L, int H, float scale,
                                                           uint32_t dthresh, uint32_t dseed, float dinv,
                                                           const uint32_t* epoch, const float* __restrict__ amask, size_t amask_bh) {
    dseed = mm_eff_seed(dseed, epoch);
    if (MASK) amask += (size_t)(blockIdx.z * H + blockIdx.y) * amask_bh;
    __shared__ __attribute__((aligned(16))) bf16 Qs[QCH * KS];
    __shared__ __attribute__((aligned(16))) bf16 Ds[QCH * KS];
    __shared__ __attribute__((aligned(16))) bf16 Qr[QCH * VR];      // Q and dO again, row = vperm(query), for the transposed
    __shared__ __attribute__((aligned(16))) bf16 Dr[QCH * VR];      // products (tr_frag32)
    __shared__ float Ls[QCH], Dl[QCH];
    const int E = H * DH, E3 = 3 * E;
    const int b = blockIdx.z, h = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int key = blockIdx.x * 128 + wave * 32 + lr;
    const bool kok = key < L;
    const bf16* base = qkv + (size_t)b * L * E3 + h * DH;
    const float scale_log2 = scale * 1.4426950408889634f;

    // K^T / V^T fragments as B operands: lane holds K[key][16s + 8*lh .. +8]
    bf16x8 kf[2], vf[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { kf[s][j] = (bf16)0.f; vf[s][j] = (bf16)0.f; }
        if (kok) {
            kf[s] = *reinterpret_cast<const bf16x8*>(base + (size_t)key * E3 + E + 16 * s + 8 * lh);
            vf[s] = *reinterpret_cast<const bf16x8*>(base + (size_t)key * E3 + 2 * E + 16 * s + 8 * lh);
        }
    }
    f32x16 dk, dv;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[r] = 0.f; dv[r] = 0.f; }

    constexpr int NI = QCH * 4 / 256;                    // next chunk (Q, dO rows, lse, delta) prefetched into registers
    static_assert(QCH <= 256, "one lse / delta value per thread");
    uint4 qreg[NI], dreg[NI];
    float lreg = INFINITY, dlreg = 0.f;
    auto load_chunk = [&](int q0) __attribute__((always_inline)) {
        const int qn = min(QCH, L - q0);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int s = tid + i * 256, qi = s >> 2, sg = s & 3;
            qreg[i] = make_uint4(0, 0, 0, 0); dreg[i] = make_uint4(0, 0, 0, 0);
            if (qi < qn) {
                qreg[i] = *reinterpret_cast<const uint4*>(base + (size_t)(q0 + qi) * E3 + sg * 8);
                dreg[i] = *reinterpret_cast<const uint4*>(dout + ((size_t)b * L + q0 + qi) * E + h * DH + sg * 8);
            }
        }
        const bool ok = tid < qn;
        lreg = ok ? lse[((size_t)b * H + h) * L + q0 + tid] : INFINITY;      // (scaled to log2 units when it is parked: no arithmetic - no wait - here)
        dlreg = ok ? delta[((size_t)b * H + h) * L + q0 + tid] : 0.f;
    };
    load_chunk(0);
    asm volatile("" : : "v"(kf[0]), "v"(kf[1]), "v"(vf[0]), "v"(vf[1]));                              // (as in the forward)
    for (int q0 = 0; q0 < L; q0 += QCH) {
        const int qn = min(QCH, L - q0);
        const int qn32 = (qn + 31) & ~31;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int s = tid + i * 256, qi = s >> 2, sg = s & 3;
            if (qi >= qn32) continue;
            *reinterpret_cast<uint4*>(Qs + qi * KS + sg * 8) = qreg[i];
            *reinterpret_cast<uint4*>(Ds + qi * KS + sg * 8) = dreg[i];
            *reinterpret_cast<uint4*>(Qr + vperm(qi) * VR + sg * 8) = qreg[i];
            *reinterpret_cast<uint4*>(Dr + vperm(qi) * VR + sg * 8) = dreg[i];
        }
        if (tid < qn32) { Ls[tid] = lreg * 1.4426950408889634f; Dl[tid] = dlreg; }
        __syncthreads();
        if (q0 + QCH < L) load_chunk(q0 + QCH);
        // software pipeline as in the forward: the Q / dO fragments of tile t + 1 and the transposed fragments of tile t
        // are requested before tile t's element-wise work
        bf16x8 qar[2], dar[2];
        if (ATTN_PIPE_BWD)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                qar[s] = *reinterpret_cast<const bf16x8*>(Qs + lr * KS + 16 * s + 8 * lh);
                dar[s] = *reinterpret_cast<const bf16x8*>(Ds + lr * KS + 16 * s + 8 * lh);
            }
        for (int qt = 0; qt < qn32; qt += 32) {
            // S[q][key] and dP[q][key]: rows = q (registers), column = this lane's key
            f32x16 sacc, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; dp[r] = 0.f; }
            if (!ATTN_PIPE_BWD)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    qar[s] = *reinterpret_cast<const bf16x8*>(Qs + (qt + lr) * KS + 16 * s + 8 * lh);
                    dar[s] = *reinterpret_cast<const bf16x8*>(Ds + (qt + lr) * KS + 16 * s + 8 * lh);
                }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qar[s], kf[s], sacc, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dar[s], vf[s], dp, 0, 0, 0);
            }
            bf16x8 dtar[2], qtar[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                dtar[s] = tr_frag32(Dr, qt + 16 * s, lane);
                qtar[s] = tr_frag32(Qr, qt + 16 * s, lane);
            }
            if (ATTN_PIPE_BWD && qt + 32 < qn32)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    qar[s] = *reinterpret_cast<const bf16x8*>(Qs + (qt + 32 + lr) * KS + 16 * s + 8 * lh);
                    dar[s] = *reinterpret_cast<const bf16x8*>(Ds + (qt + 32 + lr) * KS + 16 * s + 8 * lh);
                }
            bf16x8 pf[2], dsf[2];
            bool kp[16];
            if (DROP)
#pragma unroll
                for (int r = 0; r < 16; r += 2)                  // registers r, r + 1: queries 2i, 2i + 1 of one block
                    attn_keep2<false>(dseed, b * H + h, key, q0 + qt + (r & 3) + 8 * (r >> 2) + 4 * lh, L, dthresh, kp[r], kp[r + 1]);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qi = qt + (r & 3) + 8 * (r >> 2) + 4 * lh;
                float sc = fmaf(sacc[r], scale_log2, -Ls[qi]);
                if (MASK && kok && qi < qn) sc += amask[(size_t)(q0 + qi) * L + key] * 1.4426950408889634f;
                const float p = (FULL || kok) ? fast_exp2(sc) : 0.f;
                // the keep scale of P's dropout multiplies dV at the end; dS = P o (dP o mask * scale - delta)
                pf[r >> 3][r & 7] = (bf16)((!DROP || kp[r]) ? p : 0.f);
                const float u = DROP ? fmaf(kp[r] ? dp[r] : 0.f, dinv, -Dl[qi]) : dp[r] - Dl[qi];
                dsf[r >> 3][r & 7] = (bf16)(p * u);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dtar[s], pf[s], dv, 0, 0, 0);
                dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtar[s], dsf[s], dk, 0, 0, 0);
            }
        }
    }
    if (kok) {
        bf16* krow = dqkv + ((size_t)b * L + key) * E3 + E + h * DH;
        bf16* vrow = krow + E;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bf16x4 a = {(bf16)(dk[4 * g] * scale), (bf16)(dk[4 * g + 1] * scale), (bf16)(dk[4 * g + 2] * scale), (bf16)(dk[4 * g + 3] * scale)};
            const float dvs = DROP ? dinv : 1.f;
            bf16x4 c = {(bf16)(dv[4 * g] * dvs), (bf16)(dv[4 * g + 1] * dvs), (bf16)(dv[4 * g + 2] * dvs), (bf16)(dv[4 * g + 3] * dvs)};
            *reinterpret_cast<bf16x4*>(krow + 8 * g + 4 * lh) = a;
            *reinterpret_cast<bf16x4*>(vrow + 8 * g + 4 * lh) = c;
        }
    }
}

}  // namespace

extern "C" {

// 8-bit drop threshold of one byte of the block hash (0 = dropout off: p < 1/512), and the keep scale that makes the
// mask unbiased for the quantised probability t / 256
static inline uint32_t attn_thresh(float p) { return p > 0.f ? (uint32_t)((double)p * 256.0 + 0.5) : 0u; }
static inline float attn_keep_scale(uint32_t t) { return t ? 256.f / (256.f - (float)t) : 1.f; }

int mm_attn_fwd(const void* qkv, void* out, float* lse, int B, int L, int H, int head_dim, float scale,
                float drop_p, uint32_t seed, const uint32_t* seed_epoch, const float* attn_mask, int attn_mask_per_head,
                hipStream_t st) {
    MM_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "attn_fwd: drop_p");
    const size_t mask_bh = attn_mask && attn_mask_per_head ? (size_t)L * L : 0;
    MM_REQUIRE(qkv && out && B > 0 && L > 0 && H > 0, "attn_fwd: null/invalid");
    MM_REQUIRE(head_dim == DH, "attn_fwd: head_dim=%d (kernel is specialised for 32)", head_dim);
    dim3 grid(ceil_div(L, 128), H, B);
    const bool full = L % KCH == 0;
    const uint32_t dth = attn_thresh(drop_p);
    auto kern = dth ? (full ? attn_fwd_kernel<true, true> : attn_fwd_kernel<true, false>)
                    : (full ? attn_fwd_kernel<false, true> : attn_fwd_kernel<false, false>);
    if (attn_mask) kern = dth ? attn_fwd_kernel<true, false, true> : attn_fwd_kernel<false, false, true>;
    hipLaunchKernelGGL(kern, grid, dim3(256), 0, st, (const bf16*)qkv, (bf16*)out, lse, L, H,
                       scale * 1.4426950408889634f, dth, seed, attn_keep_scale(dth), seed_epoch, attn_mask, mask_bh);
    return mm_check_launch("attn_fwd");
}

int mm_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* delta_ws,
                int B, int L, int H, int head_dim, float scale, float drop_p, uint32_t seed,
                const uint32_t* seed_epoch, const float* attn_mask, int attn_mask_per_head, hipStream_t st) {
    const uint32_t dth = attn_thresh(drop_p);
    const float dinv = attn_keep_scale(dth);
    const size_t mask_bh = attn_mask && attn_mask_per_head ? (size_t)L * L : 0;
    MM_REQUIRE(qkv && out && dout && lse && dqkv && delta_ws && B > 0 && L > 0 && H > 0, "attn_bwd: null/invalid");
    MM_REQUIRE(head_dim == DH, "attn_bwd: head_dim=%d (kernel is specialised for 32)", head_dim);
    dim3 grid(ceil_div(L, 128), H, B);
    const bool full = L % KCH == 0 && L % QCH == 0;
    auto kdq = dth ? (full ? attn_bwd_dq_kernel<true, true> : attn_bwd_dq_kernel<true, false>)
                   : (full ? attn_bwd_dq_kernel<false, true> : attn_bwd_dq_kernel<false, false>);
    auto kdkv = dth ? (full ? attn_bwd_dkv_kernel<true, true> : attn_bwd_dkv_kernel<true, false>)
                    : (full ? attn_bwd_dkv_kernel<false, true> : attn_bwd_dkv_kernel<false, false>);
    if (attn_mask) {
        kdq = dth ? attn_bwd_dq_kernel<true, false, true> : attn_bwd_dq_kernel<false, false, true>;
        kdkv = dth ? attn_bwd_dkv_kernel<true, false, true> : attn_bwd_dkv_kernel<false, false, true>;
    }
    hipLaunchKernelGGL(kdq, grid, dim3(256), 0, st, (const bf16*)qkv, (const bf16*)out,
                       (const bf16*)dout, lse, (bf16*)dqkv, delta_ws, L, H, scale, dth, seed, dinv, seed_epoch, attn_mask, mask_bh);
    int rc = mm_check_launch("attn_bwd_dq");
    if (rc) return rc;
    hipLaunchKernelGGL(kdkv, grid, dim3(256), 0, st, (const bf16*)qkv, (const bf16*)dout, lse,
                       delta_ws, (bf16*)dqkv, L, H, scale, dth, seed, dinv, seed_epoch, attn_mask, mask_bh);
    return mm_check_launch("attn_bwd_dkv");
}

}  // extern "C"
