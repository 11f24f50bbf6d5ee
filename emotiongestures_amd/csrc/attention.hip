// ScaledDotProductAttention on the matrix cores (Full_model/Modules.py:13-23; called from MultiHeadAttention.forward,
// Full_model/SubLayers.py:30-59): out[b,i,h] = softmax_j((q[b,i,h] / sqrt(dk)) . k[b,j,h]) v[b,j,h], mask = None, eval mode.
//
// One workgroup (4 waves) per (64-row query chunk, head, clip); one wave per 16-query tile.
//   * Q (pre-scaled by 1/sqrt(dk): the reference divides q first), K and V^T of the head are staged once into LDS as fp32:
//     Q/K rows at a 68-float pitch (17 slots of 16 B: the 16 rows of an MFMA operand tile hit 16 different slots), V transposed
//     ([d][k], k contiguous) so that every MFMA operand below is read with ds_read_b128.
//   * S^T = K Q^T per 16x16 tile: A = K rows, B = Q rows (MFMA "swapped" form).  Lane (i = lane & 15, g = lane >> 4) then owns
//     S[q = i][k = 16t + 4g + r], r = 0..3, for every key tile t: the whole softmax row of query i lives in 4 lanes (i, i+16,
//     i+32, i+48) x registers -- max / sum are register reductions plus two cross-lane steps, no LDS round trip for the scores.
//   * O^T = V^T P^T: the accumulator of the first product is already the B operand of the second (cdna_hip_programming.md §3,
//     "an accumulator tile as the next MFMA's operand"): for the 32-deep k-step s the lane holds k = 32s + {4g..4g+3} and
//     32s + 16 + {4g..4g+3}; the A operand (V^T) is gathered in the same permuted k order by two ds_read_b128.  O^T's lane owns
//     4 consecutive d of one query row: one 16-byte store into the head-concatenated output, plus (bf16 modes) the row's
//     bf16 (hi, lo) tile-planar images for the output projection.
// Arithmetic modes as everywhere else: f32 = v_mfma_f32_16x16x4_f32 (exact fp32 products); bf16x3 = three
// v_mfma_f32_16x16x32_bf16 on (hi, lo) splits of both operands; bf16 = one.  exp() is expf on fp32 scores.
// No packed-fp32 VALU on LDS-loaded pairs anywhere (DESIGN.md §5 "co-residency": tools/hazard_probe.hip).
#include "common.h"

namespace {

constexpr int ATT_QC = 64;      // query rows per workgroup
constexpr int ATT_P = 68;       // Q / K row pitch in floats

// DROP (training only, Modules.py:21 `attn = self.dropout(F.softmax(attn, dim=-1))`): the probabilities are masked with the counter-based
// dropout of common.h (counter = linear index of (clip, head, query, key)) before the P V product; `attn` receives the UNMASKED probabilities
// (what the backward kernel needs: it recomputes the mask).
// Optional mask (Modules.py:18-19: `attn = attn.masked_fill(mask == 0, -1e9)` before the softmax): bytes [batch][1 or Lq][Lk], 0 = masked;
// sq = 0 broadcasts one key row over the queries (MultiHeadAttention unsqueezes the head axis, SubLayers.py:44-45).  The path itself always passes None.
struct EgAttMask { const unsigned char* m = nullptr; long long sb = 0; int sq = 0; };
template <int PREC, int KT, bool DROP = false>     // KT = key tiles of 16 (Lk <= 16*KT)
__global__ __launch_bounds__(256) void attention_mfma_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k, int ldk,
                                                             const float* __restrict__ v, int ldv, float* __restrict__ out, int ldo,
                                                             float* __restrict__ attn, int H, int Lq, int Lk, float inv_temp,
                                                             unsigned short* __restrict__ oimg, int rows_total, EgDropout dr, EgAttMask mk) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int LKP = KT * 16;            // padded key count
    constexpr int VP = LKP + 4;             // V^T row pitch (floats): 16-byte aligned rows, consecutive d rows 4 banks apart
    const int qc = blockIdx.x, h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
    const int q0 = qc * ATT_QC, nq = min(ATT_QC, Lq - q0), nqt = (nq + 15) >> 4;
    float* Qs = sm;                         // [ATT_QC][68]
    float* Ks = Qs + ATT_QC * ATT_P;        // [LKP][68]
    float* Vt = Ks + LKP * ATT_P;           // [64][VP]
    const f4 z4 = (f4){0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < nqt * 16 * 16; i += 256) {
        const int r = i >> 4, c = (i & 15) * 4;
        f4 t = z4;
        if (r < nq) t = *reinterpret_cast<const f4*>(q + ((size_t)b * Lq + q0 + r) * ldq + h * 64 + c) * inv_temp;     // q / temperature first (Modules.py:15)
        *reinterpret_cast<f4*>(Qs + r * ATT_P + c) = t;
    }
    for (int i = tid; i < LKP * 16; i += 256) {
        const int r = i >> 4, c = (i & 15) * 4;
        f4 kk = z4, vv = z4;
        if (r < Lk) {
            kk = *reinterpret_cast<const f4*>(k + ((size_t)b * Lk + r) * ldk + h * 64 + c);
            vv = *reinterpret_cast<const f4*>(v + ((size_t)b * Lk + r) * ldv + h * 64 + c);
        }
        *reinterpret_cast<f4*>(Ks + r * ATT_P + c) = kk;
#pragma unroll
        for (int j = 0; j < 4; ++j) Vt[(c + j) * VP + r] = vv[j];       // rows >= Lk are zero: 0 * p stays 0
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
    for (int qt = wave; qt < nqt; qt += 4) {
        // ---- S^T tiles: acc[t][r] = S[q = 16qt + li][k = 16t + 4g + r]
        f4 s[KT];
#pragma unroll
        for (int t = 0; t < KT; ++t) s[t] = z4;
        const float* qrow = Qs + (qt * 16 + li) * ATT_P;
        if constexpr (PREC == EG_PREC_F32) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {           // k-steps (u, r): d = 16u + 4g + r over g  (the same permuted order on both operands)
                const f4 qv = *reinterpret_cast<const f4*>(qrow + 16 * u + 4 * g);
#pragma unroll
                for (int t = 0; t < KT; ++t) {
                    const f4 kv = *reinterpret_cast<const f4*>(Ks + (t * 16 + li) * ATT_P + 16 * u + 4 * g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kv[r], qv[r], s[t], 0, 0, 0);
                }
            }
        } else {
            bf8 qh[2], ql[2];
#pragma unroll
            for (int st = 0; st < 2; ++st)
                split_octet<PREC == EG_PREC_BF16X3>(*reinterpret_cast<const f4*>(qrow + 32 * st + 8 * g),
                                                    *reinterpret_cast<const f4*>(qrow + 32 * st + 8 * g + 4), qh[st], ql[st]);
#pragma unroll
            for (int t = 0; t < KT; ++t) {
                const float* krow = Ks + (t * 16 + li) * ATT_P;
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    bf8 kh, kl;
                    split_octet<PREC == EG_PREC_BF16X3>(*reinterpret_cast<const f4*>(krow + 32 * st + 8 * g),
                                                        *reinterpret_cast<const f4*>(krow + 32 * st + 8 * g + 4), kh, kl);
                    if (PREC == EG_PREC_BF16X3) {
                        s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kl, qh[st], s[t], 0, 0, 0);
                        s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh, ql[st], s[t], 0, 0, 0);
                    }
                    s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh, qh[st], s[t], 0, 0, 0);
                }
            }
        }
        if (mk.m) {         // masked_fill(mask == 0, -1e9): the literal, so that a fully masked row softmaxes to uniform as upstream's does
            const unsigned char* mrow = mk.m + (size_t)b * mk.sb + (size_t)min(q0 + qt * 16 + li, Lq - 1) * mk.sq;
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kk = t * 16 + 4 * g + r;
                    if (kk < Lk && mrow[kk] == 0) s[t][r] = -1.0e9f;
                }
        }
        // ---- softmax over k (torch.softmax(dim=-1), Modules.py:21): row max, exp, row sum, normalise
        float m = -3.0e38f;
#pragma unroll
        for (int t = 0; t < KT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = t * 16 + 4 * g + r < Lk;
                s[t][r] = ok ? s[t][r] : -3.0e38f;
                m = fmaxf(m, s[t][r]);
            }
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < KT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = (t * 16 + 4 * g + r < Lk) ? expf(s[t][r] - m) : 0.f;
                s[t][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
        const int qrow_g = q0 + qt * 16 + li;
        const bool qok = qt * 16 + li < nq;
#pragma unroll
        for (int t = 0; t < KT; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) s[t][r] *= inv;
            if (attn && qok) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kk = t * 16 + 4 * g + r;
                    if (kk < Lk) attn[(((size_t)b * H + h) * Lq + qrow_g) * Lk + kk] = s[t][r];
                }
            }
        }
        if constexpr (DROP) {
            const unsigned int sd = dropout_seed(dr.seed, dr.epoch);
            const unsigned long long base = dr.offset + (((unsigned long long)b * H + h) * Lq + qrow_g) * Lk;
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    s[t][r] = dropout_keep(sd, base + (t * 16 + 4 * g + r), dr.thr) ? s[t][r] * dr.inv_keep : 0.f;
        }
        // ---- O^T = V^T P^T: 4 d-tiles; lane owns out[q = li][d = 16dt + 4g + r]
        f4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = z4;
        if constexpr (PREC == EG_PREC_F32) {
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const f4 vv = *reinterpret_cast<const f4*>(Vt + (dt * 16 + li) * VP + t * 16 + 4 * g);        // k-steps (t, r): k = 16t + 4g + r over g
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[r], s[t][r], o[dt], 0, 0, 0);
                }
        } else {
#pragma unroll
            for (int st = 0; st < (KT + 1) / 2; ++st) {       // 32-deep k-step = key tiles 2st, 2st+1 (an odd last tile pairs with zeros)
                bf8 ph, pl;
                split_octet<PREC == EG_PREC_BF16X3>(s[2 * st], (2 * st + 1 < KT) ? s[2 * st + 1] : z4, ph, pl);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const float* vrow = Vt + (dt * 16 + li) * VP + 32 * st + 4 * g;
                    const f4 v0 = *reinterpret_cast<const f4*>(vrow);
                    const f4 v1 = (2 * st + 1 < KT) ? *reinterpret_cast<const f4*>(vrow + 16) : z4;
                    bf8 vh, vl;
                    split_octet<PREC == EG_PREC_BF16X3>(v0, v1, vh, vl);
                    if (PREC == EG_PREC_BF16X3) {
                        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vl, ph, o[dt], 0, 0, 0);
                        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh, pl, o[dt], 0, 0, 0);
                    }
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh, ph, o[dt], 0, 0, 0);
                }
            }
        }
        if (!qok) continue;
        const int row = b * Lq + qrow_g;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const int kcol = h * 64 + dt * 16 + 4 * g;
            *reinterpret_cast<f4*>(out + (size_t)row * ldo + kcol) = o[dt];
            if (oimg) {         // also emit the head-concatenated row as bf16 (hi, lo) tile-planar images for the output projection
                bf8 h8, l8;
                split_octet<true>(o[dt], z4, h8, l8);
                const int KO = H * 8;
                const size_t slot = (((size_t)(row >> 6) * KO + (kcol >> 3)) * 64 + (row & 63)) * 8 + (kcol & 7);
                const size_t lo_off = (size_t)((rows_total + 63) >> 6) * KO * 512;
                typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                const u32x4_t hh = __builtin_bit_cast(u32x4_t, h8), ll = __builtin_bit_cast(u32x4_t, l8);
                *reinterpret_cast<u32x2*>(oimg + slot) = (u32x2){hh[0], hh[1]};
                *reinterpret_cast<u32x2*>(oimg + lo_off + slot) = (u32x2){ll[0], ll[1]};
            }
        }
    }
}

template <int PREC, int KT, bool DROP = false>
int launch_att(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* out, int ldo, float* attn, void* oimg,
               int batch, int heads, int lq, int lk, float inv_temp, hipStream_t st, EgDropout dr = EgDropout(), EgAttMask mk = EgAttMask()) {
    constexpr size_t smem = sizeof(float) * ((size_t)ATT_QC * ATT_P + (size_t)KT * 16 * ATT_P + (size_t)64 * (KT * 16 + 4));
    auto kern = attention_mfma_kernel<PREC, KT, DROP>;
    if (int rc = eg_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), smem, "eg_attention")) return rc;
    dim3 grid(eg_cdiv(lq, ATT_QC), heads, batch);
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, q, ldq, k, ldk, v, ldv, out, ldo, attn, heads, lq, lk, inv_temp,
                       reinterpret_cast<unsigned short*>(oimg), batch * lq, dr, mk);
    return eg_check_launch("attention");
}

template <int PREC>
int launch_att_kt(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* out, int ldo, float* attn, void* oimg,
                  int batch, int heads, int lq, int lk, float inv_temp, hipStream_t st, EgAttMask mk = EgAttMask()) {
    const EgDropout nd;
    if (lk <= 48) return launch_att<PREC, 3>(q, ldq, k, ldk, v, ldv, out, ldo, attn, oimg, batch, heads, lq, lk, inv_temp, st, nd, mk);
    if (lk <= 64) return launch_att<PREC, 4>(q, ldq, k, ldk, v, ldv, out, ldo, attn, oimg, batch, heads, lq, lk, inv_temp, st, nd, mk);
    if (lk <= 128) return launch_att<PREC, 8>(q, ldq, k, ldk, v, ldv, out, ldo, attn, oimg, batch, heads, lq, lk, inv_temp, st, nd, mk);
    return launch_att<PREC, 16>(q, ldq, k, ldk, v, ldv, out, ldo, attn, oimg, batch, heads, lq, lk, inv_temp, st, nd, mk);
}

}  // namespace

int egi_attention(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* out, int ldo, float* attn,
                  void* oimg, int batch, int heads, int lq, int lk, int dk, int precision, hipStream_t st) {
    EG_REQUIRE(q && k && v && out && batch > 0 && heads > 0 && lq > 0 && lk > 0, EG_ERR_BAD_ARG, "eg_attention: null pointer or empty shape");
    EG_REQUIRE(dk == 64, EG_ERR_UNSUPPORTED, "eg_attention: d_k=%d (64 supported)", dk);
    EG_REQUIRE(lk <= 256, EG_ERR_UNSUPPORTED, "eg_attention: Lk=%d > 256", lk);
    EG_REQUIRE(((ldq | ldk | ldv | ldo) & 3) == 0, EG_ERR_ALIGN, "eg_attention: row strides must be multiples of 4");
    EG_REQUIRE(eg_aligned16(q) && eg_aligned16(k) && eg_aligned16(v) && eg_aligned16(out), EG_ERR_ALIGN, "eg_attention: 16-byte alignment");
    EG_REQUIRE(precision >= 0 && precision <= 2, EG_ERR_BAD_ARG, "eg_attention: precision %d", precision);
    const float inv_temp = 1.0f / sqrtf((float)dk);
    // 2 MAC-products per (query, key, d): QK^T and PV
    EgProfScope prof(6, 4.0 * batch * heads * (double)lq * lk * dk, st);
    if (precision == EG_PREC_F32) return launch_att_kt<EG_PREC_F32>(q, ldq, k, ldk, v, ldv, out, ldo, attn, oimg, batch, heads, lq, lk, inv_temp, st);
    if (precision == EG_PREC_BF16X3) return launch_att_kt<EG_PREC_BF16X3>(q, ldq, k, ldk, v, ldv, out, ldo, attn, oimg, batch, heads, lq, lk, inv_temp, st);
    return launch_att_kt<EG_PREC_BF16>(q, ldq, k, ldk, v, ldv, out, ldo, attn, oimg, batch, heads, lq, lk, inv_temp, st);
}

extern "C" int eg_attention(const float* q, int32_t ldq, const float* k, int32_t ldk, const float* v, int32_t ldv,
                            float* out, int32_t ldo, float* attn, int32_t batch, int32_t heads, int32_t lq, int32_t lk,
                            int32_t dk, int32_t precision, void* stream) {
    return egi_attention(q, ldq, k, ldk, v, ldv, out, ldo, attn, nullptr, batch, heads, lq, lk, dk, precision, (hipStream_t)stream);
}

extern "C" int eg_attention_masked(const float* q, int32_t ldq, const float* k, int32_t ldk, const float* v, int32_t ldv, const uint8_t* mask,
                                   int64_t mask_batch_stride, int32_t mask_query_stride, float* out, int32_t ldo, float* attn, int32_t batch,
                                   int32_t heads, int32_t lq, int32_t lk, int32_t dk, int32_t precision, void* stream) {
    EG_REQUIRE(q && k && v && out && mask && batch > 0 && heads > 0 && lq > 0 && lk > 0, EG_ERR_BAD_ARG, "eg_attention_masked: null pointer or empty shape");
    EG_REQUIRE(dk == 64, EG_ERR_UNSUPPORTED, "eg_attention_masked: d_k=%d (64 supported)", dk);
    EG_REQUIRE(lk <= 256, EG_ERR_UNSUPPORTED, "eg_attention_masked: Lk=%d > 256", lk);
    EG_REQUIRE(((ldq | ldk | ldv | ldo) & 3) == 0, EG_ERR_ALIGN, "eg_attention_masked: row strides must be multiples of 4");
    EG_REQUIRE(eg_aligned16(q) && eg_aligned16(k) && eg_aligned16(v) && eg_aligned16(out), EG_ERR_ALIGN, "eg_attention_masked: 16-byte alignment");
    EG_REQUIRE(precision >= 0 && precision <= 2, EG_ERR_BAD_ARG, "eg_attention_masked: precision %d", precision);
    EG_REQUIRE(mask_query_stride == 0 || mask_query_stride >= lk, EG_ERR_BAD_ARG, "eg_attention_masked: query stride %d (0 = one row per clip, else >= Lk)", mask_query_stride);
    EgAttMask mk;
    mk.m = mask; mk.sb = mask_batch_stride; mk.sq = mask_query_stride;
    const float inv_temp = 1.0f / sqrtf((float)dk);
    hipStream_t st = (hipStream_t)stream;
    EgProfScope prof(6, 4.0 * batch * heads * (double)lq * lk * dk, st);
    if (precision == EG_PREC_F32) return launch_att_kt<EG_PREC_F32>(q, ldq, k, ldk, v, ldv, out, ldo, attn, nullptr, batch, heads, lq, lk, inv_temp, st, mk);
    if (precision == EG_PREC_BF16X3) return launch_att_kt<EG_PREC_BF16X3>(q, ldq, k, ldk, v, ldv, out, ldo, attn, nullptr, batch, heads, lq, lk, inv_temp, st, mk);
    return launch_att_kt<EG_PREC_BF16>(q, ldq, k, ldk, v, ldv, out, ldo, attn, nullptr, batch, heads, lq, lk, inv_temp, st, mk);
}

// ---- training: forward with dropout on the probabilities, MFMA backward -------------------------------------------------------------------
// ScaledDotProductAttention under autograd (Modules.py:13-23 inside MultiHeadAttention, SubLayers.py:30-59), exact fp32 products
// (v_mfma_f32_16x16x4_f32: the gradient-parity arithmetic).  Given P (the forward's UNMASKED probabilities), the dropout mask M (recomputed from
// its counter; M = keep / (1 - p), identity when p = 0) and dO:
//     A = P * M (what multiplied V);   dV = A^T dO;   dA = (dO V^T) * M;   dS = P * (dA - rowsum(dA * P));   dQ = dS K / temp;   dK = dS^T Q / temp.
// One workgroup per (head, clip) walks the queries in chunks of QC: K and V rows stay in LDS for the whole walk, a chunk's Q / dO rows, dS and A
// ([query][key] fp32) are staged per chunk; dK / dV accumulate in registers across the chunks (wave w owns the key tiles t = w, w + 4, ...).
//   * dP^T tile = V dO^T in the forward's S^T form: lane (i = lane & 15, g = lane >> 4) owns dP[q = i][k = 16 t + 4 g + r], the layout P is read in
//     (and was written in by the forward): softmax backward is register arithmetic plus two cross-lane steps per row;
//   * dQ = dS K: the dS registers are the A operand as they stand (k-steps enumerated as (t, r): k = 16 t + 4 g + r over g), B = K rows from LDS;
//   * dK^T / dV^T contract over the QUERIES, i.e. over the lane index of those registers: dS and A take one trip through LDS ([q][k] rows,
//     16-byte writes) and come back with the key on the lane and the query on the k-step -- the one transpose the data flow needs.
// Lk <= 64: QC = 64 (TED 34, BEAT 60); Lk <= 128: QC = 32 (BEAT-long 120).  Deterministic: no atomics, fixed summation order.
namespace {

template <int KT, int QC, bool DROP>
__global__ __launch_bounds__(256) void attention_bwd_mfma_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k, int ldk,
                                                                 const float* __restrict__ v, int ldv, const float* __restrict__ p,
                                                                 const float* __restrict__ dout, int ldo, float* __restrict__ dq, int lddq,
                                                                 float* __restrict__ dk, int lddk, float* __restrict__ dv, int lddv, int H, int Lq,
                                                                 int Lk, float inv_temp, EgDropout dr) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int LKP = KT * 16, KP = LKP + 4;      // padded key count; row pitch of the [query][key] images
    constexpr int NKT = (KT + 3) / 4;               // key tiles a wave accumulates dK / dV for
    constexpr int QT = QC / 16;                     // query tiles per chunk (<= 4 waves)
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
    float* Ks = sm;                         // [LKP][68]
    float* Vs = Ks + LKP * ATT_P;           // [LKP][68]
    float* Qs = Vs + LKP * ATT_P;           // [QC][68]
    float* Ds = Qs + QC * ATT_P;            // dO [QC][68]
    float* Ss = Ds + QC * ATT_P;            // dS [QC][KP]
    float* As = Ss + QC * KP;               // A = P * M [QC][KP]
    const f4 z4 = (f4){0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < LKP * 16; i += 256) {
        const int r = i >> 4, c = (i & 15) * 4;
        f4 kk = z4, vv = z4;
        if (r < Lk) {
            kk = *reinterpret_cast<const f4*>(k + ((size_t)b * Lk + r) * ldk + h * 64 + c);
            vv = *reinterpret_cast<const f4*>(v + ((size_t)b * Lk + r) * ldv + h * 64 + c);
        }
        *reinterpret_cast<f4*>(Ks + r * ATT_P + c) = kk;
        *reinterpret_cast<f4*>(Vs + r * ATT_P + c) = vv;
    }
    f4 acck[NKT][4], accv[NKT][4];          // dK^T / dV^T tiles [key tile][d tile]: lane owns [key = 4 g + r][d = li]
#pragma unroll
    for (int a = 0; a < NKT; ++a)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) acck[a][dt] = accv[a][dt] = z4;
    const unsigned int sd = DROP ? dropout_seed(dr.seed, dr.epoch) : 0u;

    for (int q0 = 0; q0 < Lq; q0 += QC) {
        const int nq = min(QC, Lq - q0);
        __syncthreads();                    // the previous chunk's dK / dV reads of Qs / Ds / Ss / As are done (first pass: K / V staged)
        for (int i = tid; i < QC * 16; i += 256) {
            const int r = i >> 4, c = (i & 15) * 4;
            f4 qq = z4, dd = z4;
            if (r < nq) {
                qq = *reinterpret_cast<const f4*>(q + ((size_t)b * Lq + q0 + r) * ldq + h * 64 + c);
                dd = *reinterpret_cast<const f4*>(dout + ((size_t)b * Lq + q0 + r) * ldo + h * 64 + c);
            }
            *reinterpret_cast<f4*>(Qs + r * ATT_P + c) = qq;
            *reinterpret_cast<f4*>(Ds + r * ATT_P + c) = dd;
        }
        __syncthreads();
        if (wave < QT) {                    // one wave per 16-query tile: dP, softmax backward, dQ
            const int qt = wave;
            const int qrow = q0 + qt * 16 + li;
            const bool qok = qt * 16 + li < nq;
            // dP^T tiles: s[t][r] = dP[q = li][k = 16 t + 4 g + r]  (A = V rows, B = dO rows; k-steps (u, r): d = 16 u + 4 g + r over g)
            f4 s[KT];
#pragma unroll
            for (int t = 0; t < KT; ++t) s[t] = z4;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const f4 dov = *reinterpret_cast<const f4*>(Ds + (qt * 16 + li) * ATT_P + 16 * u + 4 * g);
#pragma unroll
                for (int t = 0; t < KT; ++t) {
                    const f4 vv = *reinterpret_cast<const f4*>(Vs + (t * 16 + li) * ATT_P + 16 * u + 4 * g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[r], dov[r], s[t], 0, 0, 0);
                }
            }
            // P in the same layout, mask, softmax backward
            f4 pr[KT];
            float rs = 0.f;
            const float* prow = p + (((size_t)b * H + h) * Lq + qrow) * Lk;
            const unsigned long long base = dr.offset + (((unsigned long long)b * H + h) * Lq + qrow) * Lk;
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kk = t * 16 + 4 * g + r;
                    const float pv = (qok && kk < Lk) ? prow[kk] : 0.f;
                    float m = 1.f;
                    if (DROP) m = dropout_keep(sd, base + kk, dr.thr) ? dr.inv_keep : 0.f;
                    pr[t][r] = pv;
                    s[t][r] *= m;                       // dA = dP * M
                    rs += s[t][r] * pv;
                    As[(qt * 16 + li) * KP + kk] = pv * m;         // A = P * M (written below as a float4 would need the loop order swapped; 4 B stores, same row)
                }
            rs += __shfl_xor(rs, 16, 64);
            rs += __shfl_xor(rs, 32, 64);
#pragma unroll
            for (int t = 0; t < KT; ++t) {
#pragma unroll
                for (int r = 0; r < 4; ++r) s[t][r] = pr[t][r] * (s[t][r] - rs);       // dS
                *reinterpret_cast<f4*>(Ss + (qt * 16 + li) * KP + t * 16 + 4 * g) = s[t];
            }
            // dQ tile [q rows][d cols] = dS (A operand: lane (q = li, g) supplies k = 16 t + 4 g + r) x K rows (B operand)
            f4 oq[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) oq[dt] = z4;
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float* krow = Ks + (t * 16 + 4 * g + r) * ATT_P + li;
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) oq[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(s[t][r], krow[dt * 16], oq[dt], 0, 0, 0);
                }
            // D[row = q 4 g + r][col = d li]
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qq = qt * 16 + 4 * g + r;
                if (qq < nq)
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) dq[((size_t)b * Lq + q0 + qq) * lddq + h * 64 + dt * 16 + li] = oq[dt][r] * inv_temp;
            }
        } else if (QT < 4) {
            // (waves without a query tile in this chunk have nothing to do in this phase)
        }
        __syncthreads();
        // dK^T / dV^T: contraction over this chunk's queries.  A = dS^T / A^T (row = key li, k-step = query 4 s + g), B = Q / dO rows.
#pragma unroll
        for (int a = 0; a < NKT; ++a) {
            const int t = wave + 4 * a;
            if (t < KT) {
#pragma unroll 4
                for (int sq = 0; sq < QC / 4; ++sq) {
                    const int qq = 4 * sq + g;
                    const float ds = Ss[qq * KP + t * 16 + li], av = As[qq * KP + t * 16 + li];
                    const float* qrow_ = Qs + qq * ATT_P + li;
                    const float* drow_ = Ds + qq * ATT_P + li;
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        acck[a][dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ds, qrow_[dt * 16], acck[a][dt], 0, 0, 0);
                        accv[a][dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, drow_[dt * 16], accv[a][dt], 0, 0, 0);
                    }
                }
            }
        }
    }
    // D[row = key 4 g + r][col = d li]
#pragma unroll
    for (int a = 0; a < NKT; ++a) {
        const int t = wave + 4 * a;
        if (t >= KT) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int kk = t * 16 + 4 * g + r;
            if (kk < Lk)
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    dk[((size_t)b * Lk + kk) * lddk + h * 64 + dt * 16 + li] = acck[a][dt][r] * inv_temp;
                    dv[((size_t)b * Lk + kk) * lddv + h * 64 + dt * 16 + li] = accv[a][dt][r];
                }
        }
    }
}

template <int KT, int QC>
int launch_att_bwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* p, const float* dout, int ldo, float* dq,
                   int lddq, float* dk, int lddk, float* dv, int lddv, int batch, int heads, int lq, int lk, float inv_temp, const EgDropout& dr,
                   hipStream_t st) {
    constexpr size_t smem = sizeof(float) * ((size_t)2 * KT * 16 * ATT_P + (size_t)2 * QC * ATT_P + (size_t)2 * QC * (KT * 16 + 4));
    const dim3 grid(heads, batch);
    if (dr.thr) {
        auto kern = attention_bwd_mfma_kernel<KT, QC, true>;
        if (int rc = eg_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), smem, "eg_attention_backward")) return rc;
        hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, q, ldq, k, ldk, v, ldv, p, dout, ldo, dq, lddq, dk, lddk, dv, lddv, heads, lq, lk, inv_temp, dr);
    } else {
        auto kern = attention_bwd_mfma_kernel<KT, QC, false>;
        if (int rc = eg_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), smem, "eg_attention_backward")) return rc;
        hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, q, ldq, k, ldk, v, ldv, p, dout, ldo, dq, lddq, dk, lddk, dv, lddv, heads, lq, lk, inv_temp, dr);
    }
    return eg_check_launch("attention_backward");
}

int make_dropout(float p, uint32_t seed, uint64_t offset, const int32_t* epoch_dev, EgDropout& dr) {
    EG_REQUIRE(p >= 0.f && p < 1.f, EG_ERR_BAD_ARG, "attention dropout: p=%f", (double)p);
    dr.thr = (unsigned int)((double)p * 4294967296.0);
    dr.inv_keep = 1.0f / (1.0f - p);
    dr.seed = seed; dr.offset = offset; dr.epoch = epoch_dev;
    return EG_OK;
}

}  // namespace

// Training forward: eg_attention in the fp32 arithmetic with nn.Dropout(p) on the probabilities (Modules.py:21); `attn` (required) receives the
// UNMASKED probabilities for eg_attention_backward_train.  p = 0: identical to eg_attention(..., EG_PREC_F32).
extern "C" int eg_attention_train(const float* q, int32_t ldq, const float* k, int32_t ldk, const float* v, int32_t ldv, float* out, int32_t ldo,
                                  float* attn, int32_t batch, int32_t heads, int32_t lq, int32_t lk, int32_t dk, float p, uint32_t seed,
                                  uint64_t offset, const int32_t* epoch_dev, void* stream) {
    EG_REQUIRE(q && k && v && out && attn && batch > 0 && heads > 0 && lq > 0 && lk > 0, EG_ERR_BAD_ARG, "eg_attention_train: null pointer or empty shape");
    EG_REQUIRE(dk == 64 && lk <= 128, EG_ERR_UNSUPPORTED, "eg_attention_train: d_k=%d Lk=%d (64; <= 128)", dk, lk);
    EG_REQUIRE(((ldq | ldk | ldv | ldo) & 3) == 0 && eg_aligned16(q) && eg_aligned16(k) && eg_aligned16(v) && eg_aligned16(out), EG_ERR_ALIGN,
               "eg_attention_train: row strides must be multiples of 4 and the pointers 16-byte aligned");
    EgDropout dr;
    if (int rc = make_dropout(p, seed, offset, epoch_dev, dr)) return rc;
    const float inv_temp = 1.0f / sqrtf((float)dk);
    hipStream_t st = (hipStream_t)stream;
    if (!dr.thr) return egi_attention(q, ldq, k, ldk, v, ldv, out, ldo, attn, nullptr, batch, heads, lq, lk, dk, EG_PREC_F32, st);
    EgProfScope prof(6, 4.0 * batch * heads * (double)lq * lk * dk, st);
    if (lk <= 48) return launch_att<EG_PREC_F32, 3, true>(q, ldq, k, ldk, v, ldv, out, ldo, attn, nullptr, batch, heads, lq, lk, inv_temp, st, dr);
    if (lk <= 64) return launch_att<EG_PREC_F32, 4, true>(q, ldq, k, ldk, v, ldv, out, ldo, attn, nullptr, batch, heads, lq, lk, inv_temp, st, dr);
    return launch_att<EG_PREC_F32, 8, true>(q, ldq, k, ldk, v, ldv, out, ldo, attn, nullptr, batch, heads, lq, lk, inv_temp, st, dr);
}

// Backward of eg_attention / eg_attention_train on the fp32 matrix pipe; (p, seed, offset, epoch_dev) must be the forward's.  Lk <= 128.
extern "C" int eg_attention_backward_train(const float* q, int32_t ldq, const float* k, int32_t ldk, const float* v, int32_t ldv, const float* attn,
                                           const float* dout, int32_t ldo, float* dq, int32_t lddq, float* dk, int32_t lddk, float* dv, int32_t lddv,
                                           int32_t batch, int32_t heads, int32_t lq, int32_t lk, int32_t dk_dim, float p, uint32_t seed, uint64_t offset,
                                           const int32_t* epoch_dev, void* stream) {
    EG_REQUIRE(q && k && v && attn && dout && dq && dk && dv && batch > 0 && heads > 0 && lq > 0 && lk > 0, EG_ERR_BAD_ARG,
               "eg_attention_backward_train: bad argument");
    EG_REQUIRE(dk_dim == 64 && lk <= 128, EG_ERR_UNSUPPORTED, "eg_attention_backward_train: d_k=%d Lk=%d (64; <= 128)", dk_dim, lk);
    EG_REQUIRE(((ldq | ldk | ldv | ldo) & 3) == 0 && eg_aligned16(q) && eg_aligned16(k) && eg_aligned16(v) && eg_aligned16(dout), EG_ERR_ALIGN,
               "eg_attention_backward_train: row strides must be multiples of 4 and the pointers 16-byte aligned");
    EgDropout dr;
    if (int rc = make_dropout(p, seed, offset, epoch_dev, dr)) return rc;
    const float inv_temp = 1.0f / sqrtf((float)dk_dim);
    hipStream_t st = (hipStream_t)stream;
    if (lk <= 48) return launch_att_bwd<3, 64>(q, ldq, k, ldk, v, ldv, attn, dout, ldo, dq, lddq, dk, lddk, dv, lddv, batch, heads, lq, lk, inv_temp, dr, st);
    if (lk <= 64) return launch_att_bwd<4, 64>(q, ldq, k, ldk, v, ldv, attn, dout, ldo, dq, lddq, dk, lddk, dv, lddv, batch, heads, lq, lk, inv_temp, dr, st);
    return launch_att_bwd<8, 32>(q, ldq, k, ldk, v, ldv, attn, dout, ldo, dq, lddq, dk, lddk, dv, lddv, batch, heads, lq, lk, inv_temp, dr, st);
}
