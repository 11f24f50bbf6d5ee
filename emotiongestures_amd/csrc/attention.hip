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

template <int PREC, int KT>     // KT = key tiles of 16 (Lk <= 16*KT)
__global__ __launch_bounds__(256) void attention_mfma_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k, int ldk,
                                                             const float* __restrict__ v, int ldv, float* __restrict__ out, int ldo,
                                                             float* __restrict__ attn, int H, int Lq, int Lk, float inv_temp,
                                                             unsigned short* __restrict__ oimg, int rows_total) {
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

template <int PREC, int KT>
int launch_att(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* out, int ldo, float* attn, void* oimg,
               int batch, int heads, int lq, int lk, float inv_temp, hipStream_t st) {
    constexpr size_t smem = sizeof(float) * ((size_t)ATT_QC * ATT_P + (size_t)KT * 16 * ATT_P + (size_t)64 * (KT * 16 + 4));
    auto kern = attention_mfma_kernel<PREC, KT>;
    if (int rc = eg_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), smem, "eg_attention")) return rc;
    dim3 grid(eg_cdiv(lq, ATT_QC), heads, batch);
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, q, ldq, k, ldk, v, ldv, out, ldo, attn, heads, lq, lk, inv_temp,
                       reinterpret_cast<unsigned short*>(oimg), batch * lq);
    return eg_check_launch("attention");
}

template <int PREC>
int launch_att_kt(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* out, int ldo, float* attn, void* oimg,
                  int batch, int heads, int lq, int lk, float inv_temp, hipStream_t st) {
    if (lk <= 48) return launch_att<PREC, 3>(q, ldq, k, ldk, v, ldv, out, ldo, attn, oimg, batch, heads, lq, lk, inv_temp, st);
    if (lk <= 64) return launch_att<PREC, 4>(q, ldq, k, ldk, v, ldv, out, ldo, attn, oimg, batch, heads, lq, lk, inv_temp, st);
    if (lk <= 128) return launch_att<PREC, 8>(q, ldq, k, ldk, v, ldv, out, ldo, attn, oimg, batch, heads, lq, lk, inv_temp, st);
    return launch_att<PREC, 16>(q, ldq, k, ldk, v, ldv, out, ldo, attn, oimg, batch, heads, lq, lk, inv_temp, st);
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
