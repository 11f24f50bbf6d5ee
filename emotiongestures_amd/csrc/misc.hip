// Small kernels of the generator: LayerNorm, embedding gather, row-periodic add,
// TCN time-axis Linear, prior/memory encoder, CVAE conv1d / convT1d, reparameterise.
#include "common.h"

namespace {

// ---- LayerNorm (Full_model/SubLayers.py:55-57,80-82): one wave per row, two-pass in registers ----------
// SUMP: the row is not read from x but folded from `nparts` partial-sum planes [nparts][rows][D] (x = plane 0) in order, + bias2 + resid: the epilogue
// the fused FFN slab kernel (ffn.hip) leaves to its LayerNorm when the hidden is split over several workgroups.
struct LnSum { int nparts = 0; const float* bias2 = nullptr; const float* resid = nullptr; int ldr = 0; };
template <int NV, bool SUMP = false>   // NV f4 per lane (D <= NV*256)
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                        const float* __restrict__ b, float* __restrict__ y, int rows, int D,
                                                        float eps, unsigned short* __restrict__ img, LnSum ps = LnSum()) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const f4* xr = reinterpret_cast<const f4*>(x + (size_t)row * D);
    const int nq = D >> 2;
    f4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int q = lane + i * 64;
        v[i] = q < nq ? xr[q] : (f4){0.f, 0.f, 0.f, 0.f};
        if (SUMP && q < nq) {
            for (int p = 1; p < ps.nparts; ++p) v[i] += reinterpret_cast<const f4*>(x + ((size_t)p * rows + row) * D)[q];
            v[i] = v[i] + reinterpret_cast<const f4*>(ps.bias2)[q] + reinterpret_cast<const f4*>(ps.resid + (size_t)row * ps.ldr)[q];
        }
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mean = wave_sum(s) / (float)D;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int q = lane + i * 64;
        if (q < nq) {
            const f4 d = v[i] - mean;
            ss += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)D + eps);
    f4* yr = reinterpret_cast<f4*>(y + (size_t)row * D);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int q = lane + i * 64;
        if (q < nq) {
            const f4 o = (v[i] - mean) * rstd * reinterpret_cast<const f4*>(g)[q] + reinterpret_cast<const f4*>(b)[q];
            yr[q] = o;
            if (img) {      // also emit the row as bf16 (hi, lo) tile-planar images for the products that consume it
                const f4 z = (f4){0.f, 0.f, 0.f, 0.f};
                bf8 h8, l8;
                split_octet<true>(o, z, h8, l8);
                const int KO = D >> 3;
                const size_t slot = (((size_t)(row >> 6) * KO + (q >> 1)) * 64 + (row & 63)) * 8 + (q & 1) * 4;
                const size_t lo_off = (size_t)((rows + 63) >> 6) * KO * 512;
                typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                const u32x4_t hh = __builtin_bit_cast(u32x4_t, h8), ll = __builtin_bit_cast(u32x4_t, l8);
                *reinterpret_cast<u32x2*>(img + slot) = (u32x2){hh[0], hh[1]};
                *reinterpret_cast<u32x2*>(img + lo_off + slot) = (u32x2){ll[0], ll[1]};
            }
        }
    }
}

// any D (rows not 16-byte aligned, e.g. the 282-wide encoder of Pose_Discriminator, Models_spatial_memory.py:671-704): scalar accesses, same two-pass order
__global__ __launch_bounds__(256) void layernorm_any_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b,
                                                            float* __restrict__ y, int rows, int D, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (size_t)row * D;
    float s = 0.f;
    for (int i = lane; i < D; i += 64) s += xr[i];
    const float mean = wave_sum(s) / (float)D;
    float ss = 0.f;
    for (int i = lane; i < D; i += 64) { const float d = xr[i] - mean; ss += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)D + eps);
    for (int i = lane; i < D; i += 64) y[(size_t)row * D + i] = (xr[i] - mean) * rstd * g[i] + b[i];
}

__global__ __launch_bounds__(256) void add_rows_any_kernel(const float* __restrict__ a, const float* __restrict__ table, float* __restrict__ out,
                                                           size_t n, int d, int period) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const size_t row = i / d;
        const size_t trow = period > 0 ? row % period : row;
        out[i] = a[i] + table[trow * d + (i - row * d)];
    }
}

// ---- elementwise helpers -------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embedding_kernel(const int64_t* __restrict__ idx, const float* __restrict__ table,
                                                        float* __restrict__ out, int rows, int dim, int ld, int n_words) {
    const int dq = dim >> 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)rows * dq; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / dq), c = (int)(i % dq);
        int64_t w = idx[r];
        w = w < 0 ? 0 : (w >= n_words ? n_words - 1 : w);
        *reinterpret_cast<f4*>(out + (size_t)r * ld + c * 4) = *reinterpret_cast<const f4*>(table + (size_t)w * dim + c * 4);
    }
}

// out = a + b ;  b optionally row-periodic (row % period) -- positional table add (Models_spatial_memory.py:46-48)
__global__ __launch_bounds__(256) void add_kernel(const f4* __restrict__ a, const f4* __restrict__ b, f4* __restrict__ out,
                                                  size_t n4, int row_q, int period) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        size_t j = i;
        if (period) {
            const size_t row = i / row_q;
            j = (row % period) * row_q + (i - row * row_q);
        }
        out[i] = a[i] + b[j];
    }
}

// ---- TextEncoderTCN.fc1: Linear over the time axis (Models_spatial_memory.py:164-166,176) -----------------
// x, y [B, L, C] channels-last:  y[b,t',c] = bias[t'] + sum_t W[t',t] x[b,t,c]
__global__ __launch_bounds__(256) void time_linear_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y, int L, int C, int ld) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Ws = sm;             // [L][L]
    float* Xs = sm + L * L;     // [L][64]
    const int b = blockIdx.y, c0 = blockIdx.x * 64, tid = threadIdx.x;
    for (int i = tid; i < L * L; i += 256) Ws[i] = w[i];
    for (int i = tid; i < L * 64; i += 256) {
        const int t = i >> 6, c = i & 63;
        Xs[i] = (c0 + c < C) ? x[((size_t)b * L + t) * ld + c0 + c] : 0.f;
    }
    __syncthreads();
    const int c = tid & 63;
    if (c0 + c >= C) return;
    for (int tp = tid >> 6; tp < L; tp += 4) {
        float s = bias[tp];
        for (int t = 0; t < L; ++t) s += Ws[tp * L + t] * Xs[t * 64 + c];
        y[((size_t)b * L + tp) * ld + c0 + c] = s;
    }
}

// ---- Prior_MemoryEncoder front half (Models_spatial_memory.py:366-390; Models_memory.py:233-251,282-287) ---
struct PriorArgs {
    const float* prior;         // [B,P,D]
    const float *w1, *b1, *s1, *t1;     // Conv1d(P->PL,k3) raw [PL][P][3], bias, BN scale/shift
    const float *w2, *b2, *s2, *t2;     // Conv1d(PL->PL,k3)
    // memory variant (NULL for spatial): SP_v1 chunk encoder and TM encoders, raw nn.Linear layouts
    const float *sp_w0, *sp_b0, *sp_w1, *sp_b1;         // [D, chunk*D], [D], [D,D], [D]
    const float *tc_w0, *tc_b0, *tc_w1, *tc_b1;         // temporal_chunk_encoder
    const float *tm_w0, *tm_b0, *tm_w1, *tm_b1;         // temporal_memory_encoder [chunk, chunk*D], [chunk], [chunk,chunk], [chunk]
    float* cat;                 // [B,F,Dpad] = cat(prior, pred), zero padded
    float* tm_mem;              // [B,D]
    float* tm_pe;               // [B,chunk]
    int P, PL, D, Dpad, chunk, variant, stage_w, dchunk;
};

__device__ __forceinline__ float block_dot(const float* __restrict__ wrow, const float* __restrict__ xs, int n, int lane) {
    float s = 0.f;
    for (int i = lane; i < n; i += 64) s += wrow[i] * xs[i];
    return wave_sum(s);
}

__global__ __launch_bounds__(256) void prior_pred_kernel(PriorArgs a) {
    // one workgroup = (clip, slice [d0, d0+dc) of the pose axis); the two stacked k=3 convs need a halo of 2 on the input and
    // 1 on the hidden map.  The memory variant (SP_v1 / TM need whole-row inner products) runs with one slice = the whole axis.
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int P = a.P, PL = a.PL, D = a.D;
    const int d0 = blockIdx.y * a.dchunk, dc = min(a.dchunk, D - d0);
    const int XW = dc + 4, HW = dc + 2;
    float* xs = sm;                     // [P][dc+4]   x[d0-2 .. d0+dc+1], zero outside [0, D)
    float* h1 = xs + P * XW;            // [PL][dc+2]  hidden[d0-1 .. d0+dc], zero outside [0, D) (conv2's padding)
    float* h2 = h1 + PL * HW;           // [PL][dc]
    float* v0 = h2 + PL * dc;           // [chunk*D] flat chunk / scratch (memory variant)
    float* v1 = v0 + (a.variant == 1 ? a.chunk * D : 0);
    float* v2 = v1 + (a.variant == 1 ? D : 0);
    float* w1s = v2 + (a.variant == 1 ? D : 0);         // conv weights staged once when they fit
    float* w2s = w1s + PL * P * 3;
    if (a.stage_w) {
        for (int i = tid; i < PL * P * 3; i += 256) w1s[i] = a.w1[i];
        for (int i = tid; i < PL * PL * 3; i += 256) w2s[i] = a.w2[i];
    }
    const float* w1p = a.stage_w ? w1s : a.w1;
    const float* w2p = a.stage_w ? w2s : a.w2;
    for (int i = tid; i < P * XW; i += 256) {
        const int f = i / XW, d = d0 - 2 + (i - f * XW);
        xs[i] = (d >= 0 && d < D) ? a.prior[((size_t)b * P + f) * D + d] : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < PL * HW; i += 256) {
        const int co = i / HW, j = i - co * HW, d = d0 - 1 + j;
        float v = 0.f;
        if (d >= 0 && d < D) {
            float s = a.b1[co];
            for (int ci = 0; ci < P; ++ci) {
                const float* wp = w1p + (co * P + ci) * 3;
                const float* xp = xs + ci * XW + j;           // x[d-1], x[d], x[d+1]
                s += wp[0] * xp[0] + wp[1] * xp[1] + wp[2] * xp[2];
            }
            v = fmaxf(s, 0.f) * a.s1[co] + a.t1[co];
        }
        h1[i] = v;
    }
    __syncthreads();
    for (int i = tid; i < PL * dc; i += 256) {
        const int co = i / dc, j = i - co * dc;
        float s = a.b2[co];
        for (int ci = 0; ci < PL; ++ci) {
            const float* wp = w2p + (co * PL + ci) * 3;
            const float* xp = h1 + ci * HW + j;               // hidden[d-1], [d], [d+1]
            s += wp[0] * xp[0] + wp[1] * xp[1] + wp[2] * xp[2];
        }
        h2[i] = fmaxf(s, 0.f) * a.s2[co] + a.t2[co];
    }
    __syncthreads();
    if (a.variant == 1) {               // d0 == 0, dc == D here
        const int CD = a.chunk * D;
        for (int i = tid; i < CD; i += 256) {               // flat last-chunk prior frames (Models_memory.py:237)
            const int f = i / D, d = i - f * D;
            v0[i] = xs[(P - a.chunk + f) * XW + d + 2];
        }
        __syncthreads();
        // SP_v1: mem = L1(L0(flat)) ; s = sigmoid(<mem, pred_c>) ; pred_c = s*pred_c + (1-s)*mem
        for (int n = wave; n < D; n += 4) {
            const float s = block_dot(a.sp_w0 + (size_t)n * CD, v0, CD, lane);
            if (lane == 0) v1[n] = s + a.sp_b0[n];
        }
        __syncthreads();
        for (int n = wave; n < D; n += 4) {
            const float s = block_dot(a.sp_w1 + (size_t)n * D, v1, D, lane);
            if (lane == 0) v2[n] = s + a.sp_b1[n];
        }
        __syncthreads();
        // TM memory encoding uses the same flat chunk (Models_memory.py:285): compute before v1 is reused
        for (int n = wave; n < D; n += 4) {
            const float s = block_dot(a.tc_w0 + (size_t)n * CD, v0, CD, lane);
            if (lane == 0) v1[n] = s + a.tc_b0[n];
        }
        __syncthreads();
        for (int n = wave; n < D; n += 4) {
            const float s = block_dot(a.tc_w1 + (size_t)n * D, v1, D, lane);
            if (lane == 0) a.tm_mem[(size_t)b * D + n] = s + a.tc_b1[n];
        }
        for (int c = wave; c < a.chunk; c += 4) {           // gate each of the first `chunk` predicted frames
            const float dot = block_dot(v2, h2 + c * D, D, lane);
            const float sg = 1.f / (1.f + expf(-dot));
            for (int d = lane; d < D; d += 64) h2[c * D + d] = sg * h2[c * D + d] + (1.f - sg) * v2[d];
        }
        __syncthreads();
        // TM pred encoding on the gated frames: pe = M1(M0(flat(pred[:chunk])))  (Models_memory.py:286)
        for (int n = wave; n < a.chunk; n += 4) {
            const float s = block_dot(a.tm_w0 + (size_t)n * CD, h2, CD, lane);
            if (lane == 0) v1[n] = s + a.tm_b0[n];
        }
        __syncthreads();
        if (tid < a.chunk) {
            float s = a.tm_b1[tid];
            for (int j = 0; j < a.chunk; ++j) s += a.tm_w1[tid * a.chunk + j] * v1[j];
            a.tm_pe[(size_t)b * a.chunk + tid] = s;
        }
    }
    // cat(prior, pred) rows for this slice; the last slice also zeroes the pad columns [D, Dpad)
    const int F = P + PL;
    const int dend = (d0 + dc >= D) ? a.Dpad : d0 + dc, wcols = dend - d0;
    for (int i = tid; i < F * wcols; i += 256) {
        const int f = i / wcols, d = d0 + (i - f * wcols);
        float v = 0.f;
        if (d < D) v = f < P ? xs[f * XW + (d - d0) + 2] : h2[(f - P) * dc + (d - d0)];
        a.cat[((size_t)b * F + f) * a.Dpad + d] = v;
    }
}

// TM_Memory_Net cross-batch step (Models_memory.py:288-292):  G = mem^T @ pe  [D,chunk]  (sum over the batch)
__global__ __launch_bounds__(256) void tm_gram_kernel(const float* __restrict__ mem, const float* __restrict__ pe,
                                                      float* __restrict__ G, int B, int D, int chunk) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= D * chunk) return;
    const int d = i / chunk, c = i - d * chunk;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += mem[(size_t)b * D + d] * pe[(size_t)b * chunk + c];
    G[i] = s;
}
// score = mem[b] @ G ; w = softmax(score) ; cat[b, P+c, :] *= (1 + w[c])
__global__ __launch_bounds__(64) void tm_apply_kernel(const float* __restrict__ mem, const float* __restrict__ G,
                                                      float* __restrict__ cat, int D, int Dpad, int chunk, int P, int F) {
    __shared__ float sc[64];
    const int b = blockIdx.x, lane = threadIdx.x;
    float mx = -3.0e38f;
    for (int c = 0; c < chunk; ++c) {
        float s = 0.f;
        for (int d = lane; d < D; d += 64) s += mem[(size_t)b * D + d] * G[d * chunk + c];
        s = wave_sum(s);
        if (lane == 0) sc[c] = s;
        mx = fmaxf(mx, s);
    }
    __syncthreads();
    float den = 0.f;
    for (int c = 0; c < chunk; ++c) den += expf(sc[c] - mx);
    for (int c = 0; c < chunk; ++c) {
        const float w = expf(sc[c] - mx) / den;
        float* row = cat + ((size_t)b * F + P + c) * Dpad;
        for (int d = lane; d < D; d += 64) row[d] = row[d] + row[d] * w;
    }
}

// ---- CVAE 1-D convolutions (CAVE/BEAT_CVAE.py:318-332,355-369), layout [n][C][L] ----------------------------
// y[n,co,l] = post( bias[co] + sum_{ci,k} w[co,ci,k] x[n,ci,l*stride + k - pad] );  post = LeakyReLU(0.2) then BN affine
// One workgroup = (sample, 64 output positions, 4*COG output channels): the input tile lives in LDS, each wave owns
// COG output channels (wave-uniform => the weight reads are scalar loads) accumulated in registers, so every LDS read of
// x is reused COG times.
template <int COG>
__global__ __launch_bounds__(256) void conv1d_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, const float* __restrict__ scale,
                                                     const float* __restrict__ shift, float* __restrict__ y, int Cin, int Cout,
                                                     int Lin, int Lout, int K, int stride, int pad, int act) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int n = blockIdx.y, l0 = blockIdx.x * 64, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int span = 63 * stride + K;
    const int in0 = l0 * stride - pad;
    const int cblk = 4 * COG;                        // output channels of this workgroup
    const int cbase = blockIdx.z * cblk;
    float* xs = sm;                                 // [Cin][span]
    float* ws = sm + ((Cin * span + 3) & ~3);       // [Cin*K][cblk]: the COG weights of a wave are COG/4 broadcast b128 reads
    for (int i = tid; i < Cin * span; i += 256) {
        const int ci = i / span, j = i - ci * span, gl = in0 + j;
        xs[i] = (gl >= 0 && gl < Lin) ? x[((size_t)n * Cin + ci) * Lin + gl] : 0.f;
    }
    for (int i = tid; i < Cin * K * cblk; i += 256) {
        const int ck = i / cblk, c = i - ck * cblk, co = cbase + c;
        ws[i] = co < Cout ? w[(size_t)co * Cin * K + ck] : 0.f;
    }
    __syncthreads();
    const int co0 = cbase + wave * COG;
    const int l = l0 + lane;
    f4 acc[COG / 4];
#pragma unroll
    for (int j = 0; j < COG / 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[j][r] = (co0 + j * 4 + r < Cout) ? bias[co0 + j * 4 + r] : 0.f;
    for (int ci = 0; ci < Cin; ++ci) {
        const float* xp = xs + ci * span + lane * stride;
        for (int k = 0; k < K; ++k) {
            const float xv = xp[k];
            const f4* wp = reinterpret_cast<const f4*>(ws + (ci * K + k) * cblk + wave * COG);
#pragma unroll
            for (int j = 0; j < COG / 4; ++j) acc[j] += wp[j] * xv;
        }
    }
    if (l >= Lout) return;
#pragma unroll
    for (int j = 0; j < COG / 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = co0 + j * 4 + r;
            if (co < Cout) {
                float s = acc[j][r];
                if (act) {
                    s = s > 0.f ? s : 0.2f * s;
                    if (scale) s = s * scale[co] + shift[co];
                }
                y[((size_t)n * Cout + co) * Lout + l] = s;
            }
        }
}

// ConvTranspose1d(k=3, stride=2, padding=1, output_padding=1): Lout = 2*Lin; weight [Cin][Cout][3]
// y[co, lo] = bias + sum_ci sum_k [ (lo + 1 - k) even, li = (lo+1-k)/2 in range ] x[ci, li] w[ci, co, k]
__global__ __launch_bounds__(128) void convt1d_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, float* __restrict__ y, int Cin, int Cout,
                                                      int Lin) {
    const int n = blockIdx.y, lo = blockIdx.x * 128 + threadIdx.x, Lout = 2 * Lin;
    if (lo >= Lout) return;
    for (int co = 0; co < Cout; ++co) {
        float s = bias[co];
        for (int ci = 0; ci < Cin; ++ci) {
            const float* xr = x + ((size_t)n * Cin + ci) * Lin;
            const float* wr = w + ((size_t)ci * Cout + co) * 3;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int t = lo + 1 - k;
                if (t >= 0 && !(t & 1) && (t >> 1) < Lin) s += xr[t >> 1] * wr[k];
            }
        }
        s = s > 0.f ? s : 0.2f * s;
        s = s * scale[co] + shift[co];
        y[((size_t)n * Cout + co) * Lout + lo] = s;
    }
}

// tiny dense layer for the CVAE MLPs (<= 512 wide): y[n, o] = bias[o] + sum_i w[o,i] x[n,i]; one wave per output
__global__ __launch_bounds__(256) void small_linear_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y, int ldy,
                                                           int In, int Out) {
    const int n = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int o = blockIdx.x * 4 + wave; o < Out; o += gridDim.x * 4) {
        float s = 0.f;
        for (int i = lane; i < In; i += 64) s += w[(size_t)o * In + i] * x[(size_t)n * ldx + i];
        s = wave_sum(s);
        if (lane == 0) y[(size_t)n * ldy + o] = s + bias[o];
    }
}

__global__ __launch_bounds__(256) void reparam_kernel(const float* __restrict__ mu, const float* __restrict__ logvar,
                                                      const float* __restrict__ eps, float* __restrict__ z, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) z[i] = eps[i] * expf(0.5f * logvar[i]) + mu[i];
}

__global__ __launch_bounds__(256) void copy2d_kernel(const float* __restrict__ src, int lds_, float* __restrict__ dst, int ldd,
                                                     int rows, int cols) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)rows * cols; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / cols), c = (int)(i % cols);
        dst[(size_t)r * ldd + c] = src[(size_t)r * lds_ + c];
    }
}

// out[row] = (a ? a[row] : 0) + b[(row / (rep*period)) * period + row % period]   (per-clip rows broadcast over `rep` draws)
__global__ __launch_bounds__(256) void add_bcast_kernel(const f4* __restrict__ a, const f4* __restrict__ b, f4* __restrict__ out,
                                                        size_t n4, int row_q, int period, int rep) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const size_t row = i / row_q;
        const size_t brow = (row / ((size_t)rep * period)) * period + row % period;
        const f4 bv = b[brow * row_q + (i - row * row_q)];
        out[i] = a ? a[i] + bv : bv;
    }
}

inline int grid_for(size_t n, int cap = 4096) {
    const size_t g = (n + 255) / 256;
    return (int)(g < (size_t)cap ? (g ? g : 1) : cap);
}

}  // namespace

// ================================ C ABI + internal launchers =============================================

int egi_layernorm(const float* x, const float* gamma, const float* beta, float* y, void* img, int rows, int d, float eps, hipStream_t st) {
    EG_REQUIRE(x && gamma && beta && y && rows > 0, EG_ERR_BAD_ARG, "eg_layernorm: null pointer");
    EG_REQUIRE(d > 0 && d <= 2048, EG_ERR_UNSUPPORTED, "eg_layernorm: D=%d", d);
    EG_REQUIRE(!img || (d & 63) == 0, EG_ERR_ALIGN, "layernorm image output needs D %% 64 == 0");
    unsigned short* im = reinterpret_cast<unsigned short*>(img);
    dim3 grid(eg_cdiv(rows, 4)), block(256);
    if (d & 3) {
        hipLaunchKernelGGL(layernorm_any_kernel, grid, block, 0, st, x, gamma, beta, y, rows, d, eps);
        return eg_check_launch("layernorm");
    }
    if (d <= 256) hipLaunchKernelGGL((layernorm_kernel<1>), grid, block, 0, st, x, gamma, beta, y, rows, d, eps, im);
    else if (d <= 512) hipLaunchKernelGGL((layernorm_kernel<2>), grid, block, 0, st, x, gamma, beta, y, rows, d, eps, im);
    else if (d <= 1024) hipLaunchKernelGGL((layernorm_kernel<4>), grid, block, 0, st, x, gamma, beta, y, rows, d, eps, im);
    else hipLaunchKernelGGL((layernorm_kernel<8>), grid, block, 0, st, x, gamma, beta, y, rows, d, eps, im);
    return eg_check_launch("layernorm");
}
extern "C" int eg_layernorm_img(const float* x, const float* gamma, const float* beta, float* y, void* y_images, int32_t rows, int32_t d, float eps,
                                void* stream) {
    return egi_layernorm(x, gamma, beta, y, y_images, rows, d, eps, (hipStream_t)stream);
}
extern "C" int eg_layernorm(const float* x, const float* gamma, const float* beta, float* y, int32_t rows, int32_t d,
                            float eps, void* stream) {
    return egi_layernorm(x, gamma, beta, y, nullptr, rows, d, eps, (hipStream_t)stream);
}


extern "C" int eg_reparameterize(const float* mu, const float* logvar, const float* eps, float* z, int64_t n, void* stream) {
    EG_REQUIRE(mu && logvar && eps && z && n > 0, EG_ERR_BAD_ARG, "eg_reparameterize: null pointer");
    hipLaunchKernelGGL(reparam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mu, logvar, eps, z, n);
    return eg_check_launch("reparameterize");
}

extern "C" int eg_add_rows(const float* a, const float* table, float* out, int64_t rows, int32_t d, int32_t period, void* stream) {
    EG_REQUIRE(a && table && out && rows > 0 && d > 0, EG_ERR_BAD_ARG, "eg_add_rows: null pointer or empty shape");
    if (d & 3) {
        const size_t n = (size_t)rows * d;
        hipLaunchKernelGGL(add_rows_any_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a, table, out, n, d, period);
        return eg_check_launch("add_rows");
    }
    const size_t n4 = (size_t)rows * (d / 4);
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(n4)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const f4*>(a),
                       reinterpret_cast<const f4*>(table), reinterpret_cast<f4*>(out), n4, d / 4, period);
    return eg_check_launch("add_rows");
}

// ---- internal (C++ linkage) launchers used by generator.hip ------------------------------------------------
int egi_embedding(const int64_t* idx, const float* table, float* out, int rows, int dim, int ld, int n_words, hipStream_t st) {
    hipLaunchKernelGGL(embedding_kernel, dim3(grid_for((size_t)rows * dim / 4)), dim3(256), 0, st, idx, table, out, rows, dim, ld, n_words);
    return eg_check_launch("embedding");
}
int egi_add(const float* a, const float* b, float* out, size_t n, int row_len, int period, hipStream_t st) {
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(n / 4)), dim3(256), 0, st, reinterpret_cast<const f4*>(a),
                       reinterpret_cast<const f4*>(b), reinterpret_cast<f4*>(out), n / 4, row_len / 4, period);
    return eg_check_launch("add");
}
int egi_add_bcast(const float* a, const float* b, float* out, size_t rows, int row_len, int period, int rep, hipStream_t st) {
    const size_t n4 = rows * (size_t)(row_len / 4);
    hipLaunchKernelGGL(add_bcast_kernel, dim3(grid_for(n4)), dim3(256), 0, st, reinterpret_cast<const f4*>(a),
                       reinterpret_cast<const f4*>(b), reinterpret_cast<f4*>(out), n4, row_len / 4, period, rep);
    return eg_check_launch("add_bcast");
}
int egi_time_linear(const float* x, const float* w, const float* bias, float* y, int batch, int L, int C, int ld, hipStream_t st) {
    const size_t smem = sizeof(float) * ((size_t)L * L + (size_t)L * 64);
    hipLaunchKernelGGL(time_linear_kernel, dim3(eg_cdiv(C, 64), batch), dim3(256), smem, st, x, w, bias, y, L, C, ld);
    return eg_check_launch("time_linear");
}
int egi_copy2d(const float* src, int lds_, float* dst, int ldd, int rows, int cols, hipStream_t st) {
    hipLaunchKernelGGL(copy2d_kernel, dim3(grid_for((size_t)rows * cols)), dim3(256), 0, st, src, lds_, dst, ldd, rows, cols);
    return eg_check_launch("copy2d");
}

struct EgiPriorW {
    const float *w1, *b1, *s1, *t1, *w2, *b2, *s2, *t2;
    const float *sp_w0, *sp_b0, *sp_w1, *sp_b1, *tc_w0, *tc_b0, *tc_w1, *tc_b1, *tm_w0, *tm_b0, *tm_w1, *tm_b1;
};
int egi_prior_encoder(const float* prior, const EgiPriorW& w, float* cat, float* tm_mem, float* tm_pe, float* tm_gram,
                      int batch, int P, int F, int D, int Dpad, int chunk, int variant, hipStream_t st) {
    PriorArgs a;
    a.prior = prior; a.w1 = w.w1; a.b1 = w.b1; a.s1 = w.s1; a.t1 = w.t1; a.w2 = w.w2; a.b2 = w.b2; a.s2 = w.s2; a.t2 = w.t2;
    a.sp_w0 = w.sp_w0; a.sp_b0 = w.sp_b0; a.sp_w1 = w.sp_w1; a.sp_b1 = w.sp_b1;
    a.tc_w0 = w.tc_w0; a.tc_b0 = w.tc_b0; a.tc_w1 = w.tc_w1; a.tc_b1 = w.tc_b1;
    a.tm_w0 = w.tm_w0; a.tm_b0 = w.tm_b0; a.tm_w1 = w.tm_w1; a.tm_b1 = w.tm_b1;
    a.cat = cat; a.tm_mem = tm_mem; a.tm_pe = tm_pe;
    a.P = P; a.PL = F - P; a.D = D; a.Dpad = Dpad; a.chunk = chunk; a.variant = variant;
    const int PL = F - P;
    a.dchunk = variant == 1 ? D : (D < 64 ? D : 64);
    const int dc = a.dchunk;
    const size_t base = sizeof(float) * ((size_t)P * (dc + 4) + (size_t)PL * (dc + 2) + (size_t)PL * dc +
                                         (variant == 1 ? (size_t)chunk * D + 2 * (size_t)D : 0));
    const size_t wbytes = sizeof(float) * ((size_t)PL * P * 3 + (size_t)PL * PL * 3);
    a.stage_w = (base + wbytes <= 160 * 1024) ? 1 : 0;          // conv weights in LDS when they fit, else read through L1
    const size_t smem = base + (a.stage_w ? wbytes : 0);
    if (smem > 160 * 1024) { eg_set_error("prior encoder: LDS need %zu B", smem); return EG_ERR_UNSUPPORTED; }
    if (int rc = eg_ensure_dynamic_lds(reinterpret_cast<const void*>(prior_pred_kernel), smem, "prior encoder")) return rc;
    hipLaunchKernelGGL(prior_pred_kernel, dim3(batch, eg_cdiv(D, dc)), dim3(256), smem, st, a);
    int rc = eg_check_launch("prior_pred");
    if (rc || variant != 1) return rc;
    hipLaunchKernelGGL(tm_gram_kernel, dim3(eg_cdiv(D * chunk, 256)), dim3(256), 0, st, tm_mem, tm_pe, tm_gram, batch, D, chunk);
    if ((rc = eg_check_launch("tm_gram"))) return rc;
    hipLaunchKernelGGL(tm_apply_kernel, dim3(batch), dim3(64), 0, st, tm_mem, tm_gram, cat, D, Dpad, chunk, P, F);
    return eg_check_launch("tm_apply");
}

int egi_conv1d(const float* x, const float* w, const float* bias, const float* scale, const float* shift, float* y, int n, int cin,
               int cout, int lin, int k, int stride, int pad, int act, hipStream_t st) {
    const int lout = (lin + 2 * pad - k) / stride + 1;
    const int per_wave = eg_cdiv(cout, 4);
    const int cog = per_wave <= 4 ? 4 : (per_wave <= 8 ? 8 : 16);
    const size_t smem = sizeof(float) * ((((size_t)cin * (63 * stride + k) + 3) & ~(size_t)3) + (size_t)cin * k * 4 * cog);
    if (smem > 160 * 1024) { eg_set_error("conv1d: LDS need %zu B", smem); return EG_ERR_UNSUPPORTED; }
    {
        const void* kp = cog == 4 ? reinterpret_cast<const void*>(conv1d_kernel<4>)
                                  : (cog == 8 ? reinterpret_cast<const void*>(conv1d_kernel<8>) : reinterpret_cast<const void*>(conv1d_kernel<16>));
        if (int rc = eg_ensure_dynamic_lds(kp, smem, "conv1d")) return rc;
    }
    dim3 grid(eg_cdiv(lout, 64), n, eg_cdiv(cout, 4 * cog));
    if (cog == 4)
        hipLaunchKernelGGL((conv1d_kernel<4>), grid, dim3(256), smem, st, x, w, bias, scale, shift, y, cin, cout, lin, lout, k, stride, pad, act);
    else if (cog == 8)
        hipLaunchKernelGGL((conv1d_kernel<8>), grid, dim3(256), smem, st, x, w, bias, scale, shift, y, cin, cout, lin, lout, k, stride, pad, act);
    else
        hipLaunchKernelGGL((conv1d_kernel<16>), grid, dim3(256), smem, st, x, w, bias, scale, shift, y, cin, cout, lin, lout, k, stride, pad, act);
    return eg_check_launch("conv1d");
}
extern "C" int eg_conv1d(const float* x, const float* w, const float* bias, const float* scale, const float* shift, float* y, int32_t n,
                         int32_t cin, int32_t cout, int32_t lin, int32_t k, int32_t stride, int32_t pad, int32_t act, void* stream) {
    EG_REQUIRE(x && w && bias && y && n > 0 && cin > 0 && cout > 0 && lin > 0, EG_ERR_BAD_ARG, "eg_conv1d: null pointer or empty shape");
    EG_REQUIRE(k >= 1 && stride >= 1 && pad >= 0 && lin + 2 * pad >= k, EG_ERR_BAD_ARG, "eg_conv1d: k=%d stride=%d pad=%d lin=%d", k, stride, pad, lin);
    EG_REQUIRE((scale == nullptr) == (shift == nullptr), EG_ERR_BAD_ARG, "eg_conv1d: scale and shift come together");
    return egi_conv1d(x, w, bias, scale, shift, y, n, cin, cout, lin, k, stride, pad, act, (hipStream_t)stream);
}
int egi_convt1d(const float* x, const float* w, const float* bias, const float* scale, const float* shift, float* y, int n, int cin,
                int cout, int lin, hipStream_t st) {
    hipLaunchKernelGGL(convt1d_kernel, dim3(eg_cdiv(2 * lin, 128), n), dim3(128), 0, st, x, w, bias, scale, shift, y, cin, cout, lin);
    return eg_check_launch("convt1d");
}
int egi_small_linear(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy, int n, int in, int out,
                     hipStream_t st) {
    hipLaunchKernelGGL(small_linear_kernel, dim3(eg_cdiv(out, 4) < 64 ? eg_cdiv(out, 4) : 64, n), dim3(256), 0, st, x, ldx, w, bias, y,
                       ldy, in, out);
    return eg_check_launch("small_linear");
}


// ---- SoftmaxContrastiveLoss (test_emotion_gesture_diversity_iterative.py:80-127) -------------------------------------------
namespace {
// one workgroup per face row i: cross[i][:] (each thread owns columns j = tid, tid+256, ...), then the row's logsumexp / argmax
__global__ __launch_bounds__(256) void contrastive_rows_kernel(const float* __restrict__ face, const float* __restrict__ audio, int n, int d,
                                                               float* __restrict__ cross, float* __restrict__ row_loss,
                                                               int* __restrict__ row_hit) {
    extern __shared__ float sh[];                 // d normalised face values, then 256 x (max, argmax) / sums
    float* f = sh;
    float* red = sh + d;
    int* redi = reinterpret_cast<int*>(red + 256);
    const int i = blockIdx.x, tid = threadIdx.x;
    float part = 0.f;
    for (int k = tid; k < d; k += 256) { const float v = face[(size_t)i * d + k]; f[k] = v; part += v * v; }
    red[tid] = part;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
    const float finv = 1.f / fmaxf(sqrtf(red[0]), 1e-12f);
    __syncthreads();
    float vmax = -INFINITY, diag = 0.f;
    int amax = 0x7fffffff;
    float* crow = cross ? cross + (size_t)i * n : nullptr;
    // pass 1: values, row maximum and its first index
    for (int j = tid; j < n; j += 256) {
        const float* a = audio + (size_t)j * d;
        float na = 0.f;
        for (int k = 0; k < d; ++k) na += a[k] * a[k];
        const float ainv = 1.f / fmaxf(sqrtf(na), 1e-12f);
        float dist = 0.f;
        for (int k = 0; k < d; ++k) { const float t = f[k] * finv - a[k] * ainv; dist += t * t; }
        const float c = fmaxf(1.f / (sqrtf(dist) + 1e-8f), 1e-8f);
        if (crow) crow[j] = c;
        if (j == i) diag = c;
        if (c > vmax) { vmax = c; amax = j; }
    }
    red[tid] = vmax; redi[tid] = amax;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            const float o = red[tid + s]; const int oi = redi[tid + s];
            if (o > red[tid] || (o == red[tid] && oi < redi[tid])) { red[tid] = o; redi[tid] = oi; }
        }
        __syncthreads();
    }
    const float m = red[0];
    const int am = redi[0];
    __syncthreads();
    // pass 2: sum exp(c - m); recompute c (cross may be absent) exactly as above
    float se = 0.f;
    for (int j = tid; j < n; j += 256) {
        const float* a = audio + (size_t)j * d;
        float na = 0.f;
        for (int k = 0; k < d; ++k) na += a[k] * a[k];
        const float ainv = 1.f / fmaxf(sqrtf(na), 1e-12f);
        float dist = 0.f;
        for (int k = 0; k < d; ++k) { const float t = f[k] * finv - a[k] * ainv; dist += t * t; }
        const float c = fmaxf(1.f / (sqrtf(dist) + 1e-8f), 1e-8f);
        se += expf(c - m);
    }
    red[tid] = se;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
    // the diagonal value lives in exactly one thread
    __shared__ float sdiag;
    if (i % 256 == tid) sdiag = diag;
    __syncthreads();
    if (tid == 0) {
        row_loss[i] = m + logf(red[0]) - sdiag;
        row_hit[i] = (am == i) ? 1 : 0;
    }
}

__global__ __launch_bounds__(256) void contrastive_mean_kernel(const float* __restrict__ row_loss, const int* __restrict__ row_hit, int n,
                                                               float* __restrict__ loss, float* __restrict__ acc) {
    __shared__ float rl[256];
    __shared__ int rh[256];
    const int tid = threadIdx.x;
    float l = 0.f; int h = 0;
    for (int i = tid; i < n; i += 256) { l += row_loss[i]; h += row_hit[i]; }
    rl[tid] = l; rh[tid] = h;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (tid < s) { rl[tid] += rl[tid + s]; rh[tid] += rh[tid + s]; } __syncthreads(); }
    if (tid == 0) { *loss = rl[0] / (float)n; *acc = (float)rh[0] / (float)n; }
}
// ---- backward of the loss above --------------------------------------------------------------------------------------------
// L = mean_i (logsumexp_j c_ij - c_ii), c_ij = max(1 / (D_ij + 1e-8), 1e-8), D_ij = || fh_i - ah_j ||, fh = f / max(||f||, 1e-12) (F.normalize).
//   G_ij = (softmax_j(c_i.) - [i == j]) / n;   dc/dD = -c^2 (0 where the clamp is active);   dD/dfh_i = (fh_i - ah_j) / D (0 at D = 0, as torch.norm)
//   W_ij = -G_ij c_ij^2 / D_ij;   g_fh_i = sum_j W_ij (fh_i - ah_j);   g_ah_j = -sum_i W_ij (fh_i - ah_j);   g_x = (g_xh - xh (xh . g_xh)) / ||x||
// Three launches: the 2n inverse norms; one workgroup per row i (softmax of its row, W row, face gradient); one per column j (audio gradient).
__global__ __launch_bounds__(256) void contrastive_norms_kernel(const float* __restrict__ face, const float* __restrict__ audio, int n, int d,
                                                                float* __restrict__ inv) {
    __shared__ float red[256];
    const int r = blockIdx.x, tid = threadIdx.x;
    const float* x = (r < n ? face + (size_t)r * d : audio + (size_t)(r - n) * d);
    float part = 0.f;
    for (int k = tid; k < d; k += 256) part += x[k] * x[k];
    red[tid] = part;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
    if (tid == 0) inv[r] = 1.f / fmaxf(sqrtf(red[0]), 1e-12f);
}
__device__ __forceinline__ float block_sum256(float v, float* red) {
    const int tid = threadIdx.x;
    __syncthreads();
    red[tid] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
    return red[0];
}
__device__ __forceinline__ float block_max256(float v, float* red) {
    const int tid = threadIdx.x;
    __syncthreads();
    red[tid] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]); __syncthreads(); }
    return red[0];
}
__global__ __launch_bounds__(256) void contrastive_bwd_rows_kernel(const float* __restrict__ face, const float* __restrict__ audio,
                                                                   const float* __restrict__ inv, int n, int d, float* __restrict__ Wm,
                                                                   float* __restrict__ gface) {
    extern __shared__ float sh[];                 // fh_i [d], then 256 reduction slots
    float* f = sh;
    float* red = sh + d;
    const int i = blockIdx.x, tid = threadIdx.x;
    const float finv = inv[i];
    for (int k = tid; k < d; k += 256) f[k] = face[(size_t)i * d + k] * finv;
    __syncthreads();
    float* wrow = Wm + (size_t)i * n;
    // pass 1: c_ij -> wrow, D_ij is recomputed in pass 3 from c (D = 1/c - 1e-8 would lose bits: recompute the distance instead)
    float vmax = -INFINITY;
    for (int j = tid; j < n; j += 256) {
        const float* a = audio + (size_t)j * d;
        const float ainv = inv[n + j];
        float dist = 0.f;
        for (int k = 0; k < d; ++k) { const float t = f[k] - a[k] * ainv; dist += t * t; }
        const float c = fmaxf(1.f / (sqrtf(dist) + 1e-8f), 1e-8f);
        wrow[j] = c;
        vmax = fmaxf(vmax, c);
    }
    const float m = block_max256(vmax, red);
    float se = 0.f;
    for (int j = tid; j < n; j += 256) se += expf(wrow[j] - m);
    const float denom = block_sum256(se, red);
    // pass 3: W_ij
    float wsum = 0.f;
    for (int j = tid; j < n; j += 256) {
        const float* a = audio + (size_t)j * d;
        const float ainv = inv[n + j];
        float dist = 0.f;
        for (int k = 0; k < d; ++k) { const float t = f[k] - a[k] * ainv; dist += t * t; }
        const float D = sqrtf(dist), c = wrow[j];
        const float G = (expf(c - m) / denom - (j == i ? 1.f : 0.f)) / (float)n;
        const bool live = D > 0.f && (1.f / (D + 1e-8f)) > 1e-8f;
        const float w = live ? -G * c * c / D : 0.f;
        wrow[j] = w;
        wsum += w;
    }
    const float Wi = block_sum256(wsum, red);       // also orders the wrow writes before the reads below
    // g_fh[k] = fh[k] * sum_j W_ij - sum_j W_ij ah_j[k]; then through the normalisation
    float dotp = 0.f;
    for (int k = tid; k < d; k += 256) {
        float acc = 0.f;
        for (int j = 0; j < n; ++j) acc += wrow[j] * audio[(size_t)j * d + k] * inv[n + j];
        const float g = f[k] * Wi - acc;
        gface[(size_t)i * d + k] = g;               // g_fh for now
        dotp += f[k] * g;
    }
    const float dot = block_sum256(dotp, red);
    for (int k = tid; k < d; k += 256) gface[(size_t)i * d + k] = (gface[(size_t)i * d + k] - f[k] * dot) * finv;
}
__global__ __launch_bounds__(256) void contrastive_bwd_cols_kernel(const float* __restrict__ face, const float* __restrict__ audio,
                                                                   const float* __restrict__ inv, const float* __restrict__ Wm, int n, int d,
                                                                   float* __restrict__ gaudio) {
    extern __shared__ float sh[];
    float* ah = sh;
    float* red = sh + d;
    const int j = blockIdx.x, tid = threadIdx.x;
    const float ainv = inv[n + j];
    for (int k = tid; k < d; k += 256) ah[k] = audio[(size_t)j * d + k] * ainv;
    float ws = 0.f;
    for (int i = tid; i < n; i += 256) ws += Wm[(size_t)i * n + j];
    const float Wj = block_sum256(ws, red);
    float dotp = 0.f;
    for (int k = tid; k < d; k += 256) {
        float acc = 0.f;
        for (int i = 0; i < n; ++i) acc += Wm[(size_t)i * n + j] * face[(size_t)i * d + k] * inv[i];
        const float g = ah[k] * Wj - acc;           // -sum_i W_ij (fh_i[k] - ah_j[k])
        gaudio[(size_t)j * d + k] = g;
        dotp += ah[k] * g;
    }
    const float dot = block_sum256(dotp, red);
    for (int k = tid; k < d; k += 256) gaudio[(size_t)j * d + k] = (gaudio[(size_t)j * d + k] - ah[k] * dot) * ainv;
}
}  // namespace

extern "C" int64_t eg_contrastive_backward_workspace_bytes(int32_t n) { return n > 0 ? ((int64_t)n * n + 2 * (int64_t)n) * 4 : 0; }
// d loss / d face, d loss / d audio for an upstream gradient of 1 (scale afterwards).
extern "C" int eg_contrastive_loss_backward(const float* face, const float* audio, int32_t n, int32_t d, float* gface, float* gaudio, void* workspace,
                                            int64_t workspace_bytes, void* stream) {
    EG_REQUIRE(face && audio && gface && gaudio && workspace, EG_ERR_BAD_ARG, "eg_contrastive_loss_backward: null pointer");
    EG_REQUIRE(n >= 1 && n <= 4096 && d >= 1 && d <= 8192, EG_ERR_BAD_ARG, "eg_contrastive_loss_backward: n=%d d=%d out of range", n, d);
    EG_REQUIRE(workspace_bytes >= eg_contrastive_backward_workspace_bytes(n), EG_ERR_WORKSPACE, "eg_contrastive_loss_backward: workspace %lld < %lld",
               (long long)workspace_bytes, (long long)eg_contrastive_backward_workspace_bytes(n));
    float* Wm = static_cast<float*>(workspace);
    float* inv = Wm + (size_t)n * n;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)(d + 256) * sizeof(float);
    hipLaunchKernelGGL(contrastive_norms_kernel, dim3(2 * n), dim3(256), 0, st, face, audio, n, d, inv);
    if (int rc = eg_check_launch("contrastive_norms")) return rc;
    hipLaunchKernelGGL(contrastive_bwd_rows_kernel, dim3(n), dim3(256), lds, st, face, audio, inv, n, d, Wm, gface);
    if (int rc = eg_check_launch("contrastive_bwd_rows")) return rc;
    hipLaunchKernelGGL(contrastive_bwd_cols_kernel, dim3(n), dim3(256), lds, st, face, audio, inv, Wm, n, d, gaudio);
    return eg_check_launch("contrastive_bwd_cols");
}

extern "C" int64_t eg_contrastive_workspace_bytes(int32_t n) { return n > 0 ? (int64_t)n * 8 : 0; }

extern "C" int eg_contrastive_loss(const float* face, const float* audio, int32_t n, int32_t d, float* cross, float* loss, float* acc,
                                   void* workspace, int64_t workspace_bytes, void* stream) {
    EG_REQUIRE(face && audio && loss && acc && workspace, EG_ERR_BAD_ARG, "eg_contrastive_loss: null pointer");
    EG_REQUIRE(n >= 1 && n <= 4096 && d >= 1 && d <= 8192, EG_ERR_BAD_ARG, "eg_contrastive_loss: n=%d d=%d out of range", n, d);
    EG_REQUIRE(workspace_bytes >= eg_contrastive_workspace_bytes(n), EG_ERR_WORKSPACE, "eg_contrastive_loss: workspace %lld < %lld",
               (long long)workspace_bytes, (long long)eg_contrastive_workspace_bytes(n));
    float* row_loss = static_cast<float*>(workspace);
    int* row_hit = reinterpret_cast<int*>(row_loss + n);
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)(d + 512) * sizeof(float);
    hipLaunchKernelGGL(contrastive_rows_kernel, dim3(n), dim3(256), lds, st, face, audio, n, d, cross, row_loss, row_hit);
    int rc = eg_check_launch("contrastive_rows");
    if (rc != EG_OK) return rc;
    hipLaunchKernelGGL(contrastive_mean_kernel, dim3(1), dim3(256), 0, st, row_loss, row_hit, n, loss, acc);
    return eg_check_launch("contrastive_mean");
}
