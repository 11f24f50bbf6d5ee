// Training-path primitives (fp32, deterministic: fixed-order two-level reductions, no atomics).
//
// The reference trains with autograd over ATen (train_audio_classifier_K_fold.py:155-175 is its one loop; the generator's
// training script is not in the repo).  These are the backward-side kernels the forward library lacks:
//   * TN GEMM  C[m,n] = sum_k A[k,m] B[k,n]  on v_mfma_f32_16x16x4_f32 (weight gradients: dW = dY^T X, conv wgrad over im2col rows),
//   * transpose (dX = dY W runs on the forward NT GEMM with W^T), im2col / col2im for 3x3 NHWC and for 1-D channels-last convs,
//   * BatchNorm with batch statistics (forward, running-stat update, backward), LayerNorm backward, attention backward,
//   * SE pooling / gating forward+backward pieces, ReLU / LeakyReLU / sigmoid masks, losses (smooth-L1, cross-entropy / focal),
//   * fused Adam with torch.optim.Adam's exact update order (L2 weight decay folded into the gradient).
// Every entry is a plain C-ABI launcher on caller-owned memory; the host sequences them (emotiongestures_amd/train/).
#include "common.h"

namespace {

constexpr int TB = 256;
inline dim3 grid1(size_t n, int cap = 8192) { const size_t b = (n + TB - 1) / TB; return dim3((unsigned)(b < (size_t)cap ? (b ? b : 1) : cap)); }

// ---- transpose: y[c][r] = x[r][c] ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ x, int ldx, int rows, int cols, float* __restrict__ y, int ldy) {
    __shared__ float t[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + tx;
        t[j][tx] = (r < rows && c < cols) ? x[(size_t)r * ldx + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + tx;
        if (c < cols && r < rows) y[(size_t)c * ldy + r] = t[tx][j];
    }
}

// ---- TN GEMM on fp32 MFMA: C[m,n] (+)= sum_k A[k,m] B[k,n];  workgroup tile 64 x 64, K-step 32, split-K over blockIdx.z ----
// LDS images [32 k][80] (pitch 80 floats: the two 16-lane halves of a ds_read_b32 hit disjoint bank halves).
// Operands are swapped (MFMA A = B-tile rows, MFMA B = A-tile rows) so that a lane owns 4 consecutive n of one m.
// The next K-step's global loads are issued into registers before the MFMAs of the current one (register double buffer).
// IMPLICIT: B is never materialised -- B[k = output pixel p][n = tap*Cin + ci] = x[b, oy*s + kh - 1, ox*s + kw - 1, ci] is gathered from
// the NHWC activation while the tile is staged (implicit-GEMM weight gradient of a 3x3 / pad 1 convolution: dW = dY^T im2col(x)).
constexpr int TN_P = 80;
struct ConvGeo { int H, W, Ho, Wo, Cin, S; };
template <bool IMPLICIT>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb,
                                                      float* __restrict__ c, int ldc, int M, int N, long K, long k_per_split,
                                                      float* __restrict__ partial, ConvGeo geo) {
    __shared__ float As[32 * TN_P], Bs[32 * TN_P];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    const long kbeg = (long)blockIdx.z * k_per_split, kend = (kbeg + k_per_split < K) ? kbeg + k_per_split : K;
    f4 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) acc[t][u] = (f4){0.f, 0.f, 0.f, 0.f};
    const bool a_vec = ((lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(a) & 15) == 0);
    const bool b_vec = IMPLICIT || (((ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(b) & 15) == 0));
    // staging role of this thread: rows r (+16), column quad cq
    const int r_ = tid >> 4, cq = (tid & 15) * 4;
    int tap_dy = 0, tap_dx = 0, ci = 0;
    if (IMPLICIT) {                         // this thread's column of the implicit matrix is fixed: (tap, ci..ci+3)
        const int col = n0 + cq, tap = col / geo.Cin;
        ci = col - tap * geo.Cin;
        tap_dy = tap / 3 - 1;
        tap_dx = tap % 3 - 1;
    }
    f4 ra[2], rb[2];
    auto load = [&](long k0) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const long k = k0 + r_ + p * 16;
            f4 va = (f4){0.f, 0.f, 0.f, 0.f}, vb = va;
            if (k < kend) {
                const float* ap = a + (size_t)k * lda + m0 + cq;
                if (a_vec && m0 + cq + 3 < M) va = *reinterpret_cast<const f4*>(ap);
                else
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (m0 + cq + j < M) va[j] = ap[j];
                if (IMPLICIT) {
                    if (n0 + cq < N) {
                        const int hw = geo.Ho * geo.Wo;
                        const int bi = (int)(k / hw), rem = (int)(k - (long)bi * hw), oy = rem / geo.Wo, ox = rem - oy * geo.Wo;
                        const int iy = oy * geo.S + tap_dy, ix = ox * geo.S + tap_dx;
                        if (iy >= 0 && iy < geo.H && ix >= 0 && ix < geo.W)
                            vb = *reinterpret_cast<const f4*>(b + (((size_t)bi * geo.H + iy) * geo.W + ix) * geo.Cin + ci);
                    }
                } else {
                    const float* bp = b + (size_t)k * ldb + n0 + cq;
                    if (b_vec && n0 + cq + 3 < N) vb = *reinterpret_cast<const f4*>(bp);
                    else
#pragma unroll
                        for (int j = 0; j < 4; ++j) if (n0 + cq + j < N) vb[j] = bp[j];
                }
            }
            ra[p] = va;
            rb[p] = vb;
        }
    };
    load(kbeg);
    for (long k0 = kbeg; k0 < kend; k0 += 32) {
        __syncthreads();                                    // the previous step's fragment reads are done
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            *reinterpret_cast<f4*>(As + (r_ + p * 16) * TN_P + cq) = ra[p];
            *reinterpret_cast<f4*>(Bs + (r_ + p * 16) * TN_P + cq) = rb[p];
        }
        __syncthreads();
        if (k0 + 32 < kend) load(k0 + 32);                 // in flight during the MFMAs below
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            float av[2], bv[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) av[t] = As[(kk * 4 + g) * TN_P + wm + t * 16 + li];
#pragma unroll
            for (int u = 0; u < 2; ++u) bv[u] = Bs[(kk * 4 + g) * TN_P + wn + u * 16 + li];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[u], av[t], acc[t][u], 0, 0, 0);
        }
    }
    // D[row = n index (4g + r)][col = m index (li)]: lane owns C[m = li][n = 4g .. 4g+3]
    float* out = partial ? partial + (size_t)blockIdx.z * M * N : c;
    const int ldo = partial ? N : ldc;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int m = m0 + wm + t * 16 + li;
        if (m >= M) continue;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int n = n0 + wn + u * 16 + 4 * g;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (n + r < N) out[(size_t)m * ldo + n + r] = acc[t][u][r];
        }
    }
}
// c[m][n] (+)= sum over z of partial[z][m][n], fixed order: 64 outputs per workgroup, the four waves take z = w, w + 4, ... with eight loads
// in flight per lane, then combine in wave order.  (One thread per output walking all slices took 90-230 us for the small, deeply split
// products: the stem's and the stride-2 convolutions' weight gradients run 57-1024 slices.)
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ partial, float* __restrict__ c, int ldc, int M, int N,
                                                        int splits, int accumulate) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t mn = (size_t)M * N, i = (size_t)blockIdx.x * 64 + lane;
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    if (i < mn) {
        int z = w;
        for (; z + 28 < splits; z += 32) {
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += partial[(size_t)(z + 4 * k) * mn + i];
        }
        for (int k = 0; z < splits; z += 4, ++k) acc[k & 7] += partial[(size_t)z * mn + i];
    }
    red[w][lane] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    __syncthreads();
    if (w == 0 && i < mn) {
        const float s = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        const int m = (int)(i / N), n = (int)(i - (size_t)m * N);
        float* o = c + (size_t)m * ldc + n;
        *o = accumulate ? *o + s : s;
    }
}

// ---- im2col / col2im, 3x3 pad 1, NHWC: col[p][tap*C + c] = x[b, oy*s + kh - 1, ox*s + kw - 1, c] -------------------------
__global__ __launch_bounds__(256) void im2col3x3_kernel(const float* __restrict__ x, float* __restrict__ col, int B, int H, int W, int C,
                                                        int Ho, int Wo, int S) {
    const size_t total = (size_t)B * Ho * Wo * 9 * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        size_t r = i / C;
        const int tap = (int)(r % 9);
        r /= 9;
        const int ox = (int)(r % Wo);
        r /= Wo;
        const int oy = (int)(r % Ho), b = (int)(r / Ho);
        const int iy = oy * S + tap / 3 - 1, ix = ox * S + tap % 3 - 1;
        col[i] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? x[(((size_t)b * H + iy) * W + ix) * C + c] : 0.f;
    }
}
// gather form of the transpose: dx[b,y,x,c] = sum over taps of dcol[(b,oy,ox)][tap*C + c] with oy*s + kh - 1 == y, ox*s + kw - 1 == x
__global__ __launch_bounds__(256) void col2im3x3_kernel(const float* __restrict__ col, float* __restrict__ dx, int B, int H, int W, int C,
                                                        int Ho, int Wo, int S) {
    const size_t total = (size_t)B * H * W * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        size_t r = i / C;
        const int xx = (int)(r % W);
        r /= W;
        const int yy = (int)(r % H), b = (int)(r / H);
        float s = 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int ty = yy + 1 - kh;
            if (ty < 0 || ty % S) continue;
            const int oy = ty / S;
            if (oy >= Ho) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int tx = xx + 1 - kw;
                if (tx < 0 || tx % S) continue;
                const int ox = tx / S;
                if (ox >= Wo) continue;
                s += col[(((size_t)b * Ho + oy) * Wo + ox) * 9 * C + (kh * 3 + kw) * C + c];
            }
        }
        dx[i] = s;
    }
}
// float4 forms (C % 4 == 0, fewer than 2^31 channel quads: 32-bit index arithmetic): one thread = 4 channels of a pixel
__global__ __launch_bounds__(256) void col2im3x3_v4_kernel(const f4* __restrict__ col, f4* __restrict__ dx, int B, int H, int W, int C4, int Ho, int Wo,
                                                           int S) {
    const unsigned total = (unsigned)B * H * W * C4;
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const unsigned c = i % C4;
        unsigned r = i / C4;
        const int xx = (int)(r % W);
        r /= W;
        const int yy = (int)(r % H), b = (int)(r / H);
        f4 s = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int ty = yy + 1 - kh;
            if (ty < 0 || ty % S) continue;
            const int oy = ty / S;
            if (oy >= Ho) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int tx = xx + 1 - kw;
                if (tx < 0 || tx % S) continue;
                const int ox = tx / S;
                if (ox >= Wo) continue;
                s += col[(((size_t)b * Ho + oy) * Wo + ox) * 9 * C4 + (kh * 3 + kw) * C4 + c];
            }
        }
        dx[i] = s;
    }
}
__global__ __launch_bounds__(256) void subsample_v4_kernel(const f4* __restrict__ x, f4* __restrict__ y, int B, int H, int W, int C4, int Ho, int Wo, int S,
                                                           int backward) {
    const unsigned total = backward ? (unsigned)B * H * W * C4 : (unsigned)B * Ho * Wo * C4;
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const unsigned c = i % C4;
        unsigned r = i / C4;
        if (!backward) {
            const int ox = (int)(r % Wo);
            r /= Wo;
            const int oy = (int)(r % Ho), b = (int)(r / Ho);
            y[i] = x[(((size_t)b * H + oy * S) * W + ox * S) * C4 + c];
        } else {
            const int xx = (int)(r % W);
            r /= W;
            const int yy = (int)(r % H), b = (int)(r / H);
            const bool hit = (yy % S == 0) && (xx % S == 0) && (yy / S < Ho) && (xx / S < Wo);
            y[i] = hit ? x[(((size_t)b * Ho + yy / S) * Wo + xx / S) * C4 + c] : (f4){0.f, 0.f, 0.f, 0.f};
        }
    }
}
// strided pixel subsample (1x1 stride-s conv input) and its transpose
__global__ __launch_bounds__(256) void subsample_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, int C, int Ho,
                                                        int Wo, int S, int backward) {
    const size_t total = backward ? (size_t)B * H * W * C : (size_t)B * Ho * Wo * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        size_t r = i / C;
        if (!backward) {
            const int ox = (int)(r % Wo);
            r /= Wo;
            const int oy = (int)(r % Ho), b = (int)(r / Ho);
            y[i] = x[(((size_t)b * H + oy * S) * W + ox * S) * C + c];
        } else {            // x = dy [B,Ho,Wo,C], y = dx [B,H,W,C]
            const int xx = (int)(r % W);
            r /= W;
            const int yy = (int)(r % H), b = (int)(r / H);
            const bool hit = (yy % S == 0) && (xx % S == 0) && (yy / S < Ho) && (xx / S < Wo);
            y[i] = hit ? x[(((size_t)b * Ho + yy / S) * Wo + xx / S) * C + c] : 0.f;
        }
    }
}
// 1-D channels-last: x [B, L, C] -> col [B*Lout, k*C], position l*stride + j*dil - pad_left
__global__ __launch_bounds__(256) void im2col1d_kernel(const float* __restrict__ x, float* __restrict__ col, int B, int L, int C, int Lout,
                                                       int k, int stride, int pad_left, int dil) {
    const size_t total = (size_t)B * Lout * k * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        size_t r = i / C;
        const int j = (int)(r % k);
        r /= k;
        const int lo = (int)(r % Lout), b = (int)(r / Lout);
        const int l = lo * stride + j * dil - pad_left;
        col[i] = (l >= 0 && l < L) ? x[((size_t)b * L + l) * C + c] : 0.f;
    }
}
__global__ __launch_bounds__(256) void col2im1d_kernel(const float* __restrict__ col, float* __restrict__ dx, int B, int L, int C, int Lout,
                                                       int k, int stride, int pad_left, int dil) {
    const size_t total = (size_t)B * L * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        size_t r = i / C;
        const int l = (int)(r % L), b = (int)(r / L);
        float s = 0.f;
        for (int j = 0; j < k; ++j) {
            const int t = l + pad_left - j * dil;
            if (t < 0 || t % stride) continue;
            const int lo = t / stride;
            if (lo >= Lout) continue;
            s += col[(((size_t)b * Lout + lo) * k + j) * C + c];
        }
        dx[i] = s;
    }
}

// ---- column reductions over rows: two deterministic levels ---------------------------------------------------------------
// level 1: block (bx, by) sums rows [by*rows_per, ...) of 64 columns -> part[by][2][C]
//   mode 0: (sum a, sum a*a)      mode 1: (sum a, sum a*b)      mode 2: (sum a, sum a*(b - mean[c]))      mode 3: (sum (a - mean[c])^2, 0)
//   mode 4 (fast path only): mode 2 with a replaced by a * [mask > 0]  (gradient behind a ReLU whose output is `mask`)
// The centred forms (2, 3) keep BatchNorm exact for channels whose mean is large against their spread (post-ReLU maps):
// E[x^2] - mean^2 from fp32 partial sums loses ~eps*mean^2/var there (measured: 3e-3 in the input gradient of one block).
// (generic channel counts -- 34 / 60 / 120 frame channels behind final_conv1, the TCN's widths: the mode is a template parameter and eight rows'
// loads are in flight per thread; with the mode a runtime branch inside a one-load-per-iteration loop a [15 872, 34] map took 21 us, round 6)
template <int MODE>
__global__ __launch_bounds__(256) void col_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ mean,
                                                          float* __restrict__ part, long rows, int C, long rows_per) {
    __shared__ float s0[4][64], s1[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
    const long r0 = (long)blockIdx.y * rows_per, r1 = (r0 + rows_per < rows) ? r0 + rows_per : rows;
    float u = 0.f, v = 0.f;
    if (c < C) {
        const float mu = (MODE >= 2) ? mean[c] : 0.f;
        auto fold = [&](float x, float y) {
            if (MODE == 0) { u += x; v += x * x; }
            else if (MODE == 1) { u += x; v += x * y; }
            else if (MODE == 2) { u += x; v += x * (y - mu); }
            else { const float d = x - mu; u += d * d; }
        };
        constexpr bool HAS_B = (MODE == 1 || MODE == 2);
        long r = r0 + w;
        for (; r + 28 < r1; r += 32) {              // eight rows of this thread's column in flight, folded in row order
            float x[8], y[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                x[k] = a[(size_t)(r + 4 * k) * C + c];
                y[k] = HAS_B ? b[(size_t)(r + 4 * k) * C + c] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) fold(x[k], y[k]);
        }
        for (; r < r1; r += 4) fold(a[(size_t)r * C + c], HAS_B ? b[(size_t)r * C + c] : 0.f);
    }
    s0[w][threadIdx.x & 63] = u;
    s1[w][threadIdx.x & 63] = v;
    __syncthreads();
    if (w == 0 && c < C) {
        const int l = threadIdx.x;
        part[((size_t)blockIdx.y * 2 + 0) * C + c] = (s0[0][l] + s0[1][l]) + (s0[2][l] + s0[3][l]);
        part[((size_t)blockIdx.y * 2 + 1) * C + c] = (s1[0][l] + s1[1][l]) + (s1[2][l] + s1[3][l]);
    }
}
// ReLU mask of a map as one nibble per float4 (bit k = component k > 0), eight float4 indices per 32-bit word (written by se_tail_fwd_kernel)
__device__ __forceinline__ unsigned relu_nibble(const unsigned* __restrict__ bits, size_t i4) { return (bits[i4 >> 3] >> (4 * (unsigned)(i4 & 7))) & 15u; }
// Fast path of level 1 when C divides 1024: the [rows, C] block is read as one float4 stream, thread t always sees the channel quad
// (4t) mod C, 256 threads x 16 B = 4 KB per block-iteration fully coalesced; partials are combined through LDS in a fixed order.
// Segmented: blockIdx.y = segment (a clip for the SE pooling, 1 segment for BatchNorm / bias / LayerNorm-affine sums),
// blockIdx.x = chunk of rows inside the segment; part[seg][chunk][2][C].
// MODE and the mask form are template parameters: with them as runtime arguments the loop body was a forest of uniform branches and the mask loads
// were issued behind them, after the two main streams (round 6: 4.4 TB/s where the apply passes reach 6.2).  MASK: 0 none, 1 a float map, 2 the nibble bits.
template <int MODE, int MASK>
__global__ __launch_bounds__(256) void col_partial_fast_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ mean,
                                                               float* __restrict__ part, long seg_rows, int C, long rows_per,
                                                               const float* __restrict__ mask, const unsigned* __restrict__ mask_bits) {
    constexpr int mode = MODE;
    constexpr bool HAS_B = (MODE == 1 || MODE == 2 || MODE == 4);
    __shared__ f4 s0[256], s1[256];
    const int tid = threadIdx.x, seg = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
    const long r0 = (long)chunk * rows_per, r1 = (r0 + rows_per < seg_rows) ? r0 + rows_per : seg_rows;
    const size_t base = ((size_t)seg * seg_rows + r0) * C;
    const size_t n4 = (r1 > r0) ? (size_t)(r1 - r0) * C / 4 : 0;
    const int cq = (tid * 4) % C;
    f4 mu = (f4){0.f, 0.f, 0.f, 0.f};
    if (mode >= 2) mu = *reinterpret_cast<const f4*>(mean + cq);
    const f4* a4 = reinterpret_cast<const f4*>(a + base);
    const f4* b4 = HAS_B ? reinterpret_cast<const f4*>(b + base) : nullptr;
    const f4* m4 = (MASK == 1) ? reinterpret_cast<const f4*>(mask + base) : nullptr;
    const size_t g4 = base >> 2;                // float4 index of this block's first element in the whole map (mode 4 with mask_bits: the nibble index)
    auto mload = [&](size_t f) -> f4 {
        if constexpr (MASK == 2) {
            const unsigned nb = relu_nibble(mask_bits, g4 + f);
            return (f4){(nb & 1u) ? 1.f : 0.f, (nb & 2u) ? 1.f : 0.f, (nb & 4u) ? 1.f : 0.f, (nb & 8u) ? 1.f : 0.f};
        }
        if constexpr (MASK == 1) return m4[f];
        return (f4){0.f, 0.f, 0.f, 0.f};
    };
    f4 u = (f4){0.f, 0.f, 0.f, 0.f}, v = u;
    const f4 zero4 = (f4){0.f, 0.f, 0.f, 0.f};
    auto fold = [&](f4 x, const f4& y, const f4& m) {         // one float4 of the stream into (u, v); same per-thread order as a plain loop
        if (mode == 4) {
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = m[r] > 0.f ? x[r] : 0.f;
            u += x;
            v += x * (y - mu);
        } else if (mode == 0) { u += x; v += x * x; }
        else if (mode == 1) { u += x; v += x * y; }
        else if (mode == 2) { u += x; v += x * (y - mu); }
        else { const f4 d = x - mu; u += d * d; }
    };
    size_t f = tid;
    // four independent 16-byte loads per stream in flight per thread (the kernel is a pure stream: at one load per iteration and <= 8 waves per
    // CU it ran at 3.7 TB/s), folded in the same order as the scalar loop
    for (; f + 768 < n4; f += 1024) {
        const f4 x0 = a4[f], x1 = a4[f + 256], x2 = a4[f + 512], x3 = a4[f + 768];
        f4 y0 = zero4, y1 = zero4, y2 = zero4, y3 = zero4, m0 = zero4, m1 = zero4, m2 = zero4, m3 = zero4;
        if constexpr (HAS_B) { y0 = b4[f]; y1 = b4[f + 256]; y2 = b4[f + 512]; y3 = b4[f + 768]; }
        if constexpr (MODE == 4) { m0 = mload(f); m1 = mload(f + 256); m2 = mload(f + 512); m3 = mload(f + 768); }
        fold(x0, y0, m0);
        fold(x1, y1, m1);
        fold(x2, y2, m2);
        fold(x3, y3, m3);
    }
    for (; f < n4; f += 256) {
        f4 y = zero4, m = zero4;
        if constexpr (HAS_B) y = b4[f];
        if constexpr (MODE == 4) m = mload(f);
        fold(a4[f], y, m);
    }
    s0[tid] = u;
    s1[tid] = v;
    __syncthreads();
    const int nq = C >> 2;                  // channel quads; 256 / nq contributors each (nq <= 256)
    if (tid < nq) {
        f4 su = (f4){0.f, 0.f, 0.f, 0.f}, sv = su;
        for (int t = tid; t < 256; t += nq) { su += s0[t]; sv += s1[t]; }
        float* o = part + (((size_t)seg * nchunk + chunk) * 2) * C + tid * 4;
        *reinterpret_cast<f4*>(o) = su;
        *reinterpret_cast<f4*>(o + C) = sv;
    }
}

// level 2 for BatchNorm forward, pass 1: mean;  pass 2 (from centred squares): rstd (biased variance) and the running statistics
// (unbiased variance, momentum) exactly as torch.nn.BatchNorm updates them
// level 2: one wave per channel; lanes stride over the partials, the two sums are folded with a fixed-order wave reduction (double)
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ void fold_partials(const float* __restrict__ part, int nblk, int C, int c, double& s, double& q) {
    const int lane = threadIdx.x & 63;
    double a = 0.0, b = 0.0;
    for (int i = lane; i < nblk; i += 64) { a += part[((size_t)i * 2) * C + c]; b += part[((size_t)i * 2 + 1) * C + c]; }
    s = wave_sum_d(a);
    q = wave_sum_d(b);
}
__global__ __launch_bounds__(256) void bn_mean_kernel(const float* __restrict__ part, int nblk, int C, long rows, float* __restrict__ mean) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    double s, q;
    fold_partials(part, nblk, C, c, s, q);
    if ((threadIdx.x & 63) == 0) mean[c] = (float)(s / (double)rows);
}
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ part, int nblk, int C, long rows, float eps, float momentum,
                                                          const float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ run_mean,
                                                          float* __restrict__ run_var) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    double q, unused;
    fold_partials(part, nblk, C, c, q, unused);
    if ((threadIdx.x & 63) != 0) return;
    const double var = q / (double)rows;
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (run_mean) {
        const double unb = rows > 1 ? var * (double)rows / (double)(rows - 1) : var;
        run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mean[c];
        run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)unb;
    }
}
__global__ __launch_bounds__(256) void col_finalize_kernel(const float* __restrict__ part, int nblk, int C, float* __restrict__ o0,
                                                           float* __restrict__ o1, float scale) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), seg = blockIdx.y;
    if (c >= C) return;
    double s, q;
    fold_partials(part + (size_t)seg * nblk * 2 * C, nblk, C, c, s, q);
    if ((threadIdx.x & 63) != 0) return;
    if (o0) o0[(size_t)seg * C + c] = (float)(s * scale);
    if (o1) o1[(size_t)seg * C + c] = (float)(q * scale);
}
// Small inputs (bias / LayerNorm-affine gradients, the BatchNorms of the prior encoder and the CVAE at training batch sizes): both levels in ONE
// launch.  A workgroup owns 64 columns: 16 column quads x 16 row lanes, each thread walks rows rl, rl + 16, ... with float4 loads (a row segment of
// 256 bytes per 16 lanes), the 16 row lanes are combined through LDS in a fixed order.  Same modes as col_partial_kernel (0, 1, 2, 3).
__global__ __launch_bounds__(256) void col_direct_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ mean,
                                                         float* __restrict__ o0, float* __restrict__ o1, long rows, int C, int mode, float scale) {
    __shared__ f4 s0[256], s1[256];
    const int tid = threadIdx.x, cq = tid & 15, rl = tid >> 4;
    const int c = blockIdx.x * 64 + 4 * cq;
    f4 u = (f4){0.f, 0.f, 0.f, 0.f}, v = u;
    if (c < C) {                                    // C % 4 == 0: a quad is inside or outside as a whole
        f4 mu = u;
        if (mode >= 2) mu = *reinterpret_cast<const f4*>(mean + c);
#pragma unroll 4
        for (long r = rl; r < rows; r += 16) {
            const f4 x = *reinterpret_cast<const f4*>(a + (size_t)r * C + c);
            if (mode == 0) { u += x; v += x * x; }
            else if (mode == 1) { u += x; v += x * *reinterpret_cast<const f4*>(b + (size_t)r * C + c); }
            else if (mode == 2) { u += x; v += x * (*reinterpret_cast<const f4*>(b + (size_t)r * C + c) - mu); }
            else { const f4 d = x - mu; u += d * d; }
        }
    }
    s0[tid] = u;
    s1[tid] = v;
    __syncthreads();
    if (tid < 16 && blockIdx.x * 64 + 4 * tid < C) {
        f4 su = (f4){0.f, 0.f, 0.f, 0.f}, sv = su;
#pragma unroll
        for (int j = 0; j < 16; ++j) { su += s0[j * 16 + tid]; sv += s1[j * 16 + tid]; }
        const int cc = blockIdx.x * 64 + 4 * tid;
        if (o0) *reinterpret_cast<f4*>(o0 + cc) = su * scale;
        if (o1) *reinterpret_cast<f4*>(o1 + cc) = sv * scale;
    }
}
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ y,
                                                       size_t total, int C) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        y[i] = (x[i] - mean[c]) * rstd[c] * gamma[c] + beta[c];
    }
}
// dx = gamma*rstd * (dy - sum_dy/R - xhat * sum_dy_xhat/R);  sums arrive as (sum dy, sum dy*(x - mean)): sum dy*xhat = rstd * the latter
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ sum_dy, const float* __restrict__ sum_dyx,
                                                           float* __restrict__ dx, float* __restrict__ dgamma, size_t total, int C, float inv_rows,
                                                           int relu_mask) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const float xv = x[i];
        const float xh = (xv - mean[c]) * rstd[c];
        const float sdyxh = rstd[c] * sum_dyx[c];
        const float d = gamma[c] * rstd[c] * (dy[i] - sum_dy[c] * inv_rows - xh * sdyxh * inv_rows);
        dx[i] = (relu_mask && !(xv > 0.f)) ? 0.f : d;          // x = relu(conv): the ReLU's backward rides on this pass
        if (dgamma && i < (size_t)C) dgamma[i] = rstd[i] * sum_dyx[i];
    }
}

// float4 forms of the two passes above (C % 4 == 0, 16-byte aligned maps and vectors): same per-element operation order, a quarter of the
// load / store instructions and of the channel-index arithmetic (the scalar forms ran at 3.9 / 5.3 TB/s on the tower's maps)
__global__ __launch_bounds__(256) void bn_apply4_kernel(const f4* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta, f4* __restrict__ y, size_t n4,
                                                        int C4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C4) * 4;
        const f4 m = *reinterpret_cast<const f4*>(mean + c), r = *reinterpret_cast<const f4*>(rstd + c), g = *reinterpret_cast<const f4*>(gamma + c),
                 b = *reinterpret_cast<const f4*>(beta + c);
        y[i] = (x[i] - m) * r * g + b;
    }
}
__global__ __launch_bounds__(256) void bn_bwd_apply4_kernel(const f4* __restrict__ x, const f4* __restrict__ dy, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                            const float* __restrict__ sum_dy, const float* __restrict__ sum_dyx, f4* __restrict__ dx,
                                                            float* __restrict__ dgamma, size_t n4, int C4, float inv_rows, int relu_mask) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C4) * 4;
        const f4 m = *reinterpret_cast<const f4*>(mean + c), r = *reinterpret_cast<const f4*>(rstd + c), g = *reinterpret_cast<const f4*>(gamma + c),
                 sd = *reinterpret_cast<const f4*>(sum_dy + c), sx = *reinterpret_cast<const f4*>(sum_dyx + c);
        const f4 xv = x[i], dv = dy[i];
        const f4 xh = (xv - m) * r;
        const f4 sdyxh = r * sx;
        f4 d = g * r * (dv - sd * inv_rows - xh * sdyxh * inv_rows);
        if (relu_mask) {
#pragma unroll
            for (int q = 0; q < 4; ++q) d[q] = xv[q] > 0.f ? d[q] : 0.f;
        }
        dx[i] = d;
        if (dgamma && i < (size_t)C4) *reinterpret_cast<f4*>(dgamma + 4 * i) = r * sx;
    }
}

// ---- fused SE-block pieces of the training path (ResNetBlocks.py:21-37 under autograd) --------------------------------------------
// The conv kernels already emit per-(clip, tile) channel sums of their output (the SE pooling partials of the inference path):
// BatchNorm's mean and the per-clip sums come from those without reading the map again.  One wave per channel.
__global__ __launch_bounds__(256) void clip_sum_from_gap_kernel(const float* __restrict__ gap, int tiles, int C, float* __restrict__ clip_sum) {
    __shared__ float red[256];
    const int b = blockIdx.x, t = threadIdx.x, c = t % C, g = t / C, ng = 256 / C;          // C divides 256: ng tile groups
    float s = 0.f;
    for (int k = g; k < tiles; k += ng) s += gap[((size_t)b * tiles + k) * C + c];
    red[t] = s;
    __syncthreads();
    if (t < C) {
        float a = 0.f;
        for (int k = 0; k < ng; ++k) a += red[k * C + t];
        clip_sum[(size_t)b * C + t] = a;
    }
}
__global__ __launch_bounds__(256) void bn_mean_from_clips_kernel(const float* __restrict__ clip_sum, int B, int C, long rows, float* __restrict__ mean) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    double tot = 0.0;
    for (int b = lane; b < B; b += 64) tot += clip_sum[(size_t)b * C + c];
    tot = wave_sum_d(tot);
    if (lane == 0) mean[c] = (float)(tot / (double)rows);
}

// Train-mode BatchNorm statistics from the producing convolution's per-tile sums AND sums of squares (eg_conv3x3_sq): no pass over the map.
// The per-tile partials are fp32 sums over <= 256 pixels; everything across tiles and clips, and the E[x^2] - mean^2 subtraction, is double:
// the relative error of the variance is ~1e-6 (1 + mean^2 / var), far inside the split-bf16 modes' own error (the fp32 configuration keeps
// the centred pass over the map).
__global__ __launch_bounds__(256) void clip_sums_sq_kernel(const float* __restrict__ gap, const float* __restrict__ gap_sq, int tiles, int C,
                                                           float* __restrict__ clip_sum, double* __restrict__ clip_d) {
    __shared__ float red[256];
    __shared__ double redq[256], reds[256];
    const int b = blockIdx.x, t = threadIdx.x, c = t % C, g = t / C, ng = 256 / C;          // C divides 256: ng tile groups
    float s = 0.f;
    double sd = 0.0, q = 0.0;
    for (int k = g; k < tiles; k += ng) {
        const float v = gap[((size_t)b * tiles + k) * C + c];
        s += v;                                 // the fp32 clip sum the SE pooling uses (same order as clip_sum_from_gap_kernel)
        sd += (double)v;
        q += (double)gap_sq[((size_t)b * tiles + k) * C + c];
    }
    red[t] = s;
    reds[t] = sd;
    redq[t] = q;
    __syncthreads();
    if (t < C) {
        float a = 0.f;
        double ad = 0.0, aq = 0.0;
        for (int k = 0; k < ng; ++k) { a += red[k * C + t]; ad += reds[k * C + t]; aq += redq[k * C + t]; }
        if (clip_sum) clip_sum[(size_t)b * C + t] = a;
        clip_d[((size_t)b * 2) * C + t] = ad;
        clip_d[((size_t)b * 2 + 1) * C + t] = aq;
    }
}
__global__ __launch_bounds__(256) void bn_finalize_sq_kernel(const double* __restrict__ clip_d, int B, int C, long rows, float eps, float momentum,
                                                             float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ run_mean,
                                                             float* __restrict__ run_var, const float* __restrict__ gamma = nullptr,
                                                             const float* __restrict__ beta = nullptr, float* __restrict__ aff_scale = nullptr,
                                                             float* __restrict__ aff_shift = nullptr) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    double s = 0.0, q = 0.0;
    for (int b = lane; b < B; b += 64) { s += clip_d[((size_t)b * 2) * C + c]; q += clip_d[((size_t)b * 2 + 1) * C + c]; }
    s = wave_sum_d(s);
    q = wave_sum_d(q);
    if (lane != 0) return;
    const double m = s / (double)rows;
    double var = q / (double)rows - m * m;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (aff_scale) {            // the apply y = (x - mean) * rstd * gamma + beta as one multiply-add per element, exactly as bn_apply rounds it: see below
        aff_scale[c] = rstd[c] * gamma[c];
        aff_shift[c] = beta[c] - mean[c] * (rstd[c] * gamma[c]);
    }
    if (run_mean) {
        const double unb = rows > 1 ? var * (double)rows / (double)(rows - 1) : var;
        run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)m;
        run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)unb;
    }
}

// SE gate of one clip from the pooled BatchNorm output, pooled[c] = gamma (clip_sum/HW - mean) rstd + beta = mean_hw(bn2(c2)):
// h = relu(W1 pooled + b1), gate = sigmoid(W2 h + b2) (ResNetBlocks.py:92-96).  One workgroup per clip, C <= 256.
__global__ __launch_bounds__(256) void se_gate_train_kernel(const float* __restrict__ clip_sum, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, const float* __restrict__ w1, const float* __restrict__ b1,
                                                            const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ pooled,
                                                            float* __restrict__ h, float* __restrict__ gate, int C, float inv_hw) {
    __shared__ float sp[256], sh[32];
    const int b = blockIdx.x, t = threadIdx.x, Ch = C >> 3;
    if (t < C) {
        const float p = (clip_sum[(size_t)b * C + t] * inv_hw - mean[t]) * rstd[t] * gamma[t] + beta[t];
        sp[t] = p;
        pooled[(size_t)b * C + t] = p;
    }
    __syncthreads();
    if (t < Ch) {
        float a = b1[t];
        for (int c = 0; c < C; ++c) a += w1[(size_t)t * C + c] * sp[c];
        a = fmaxf(a, 0.f);
        sh[t] = a;
        h[(size_t)b * Ch + t] = a;
    }
    __syncthreads();
    if (t < C) {
        float z = b2[t];
        for (int j = 0; j < Ch; ++j) z += w2[(size_t)t * Ch + j] * sh[j];
        gate[(size_t)b * C + t] = 1.f / (1.f + expf(-z));
    }
}

// out = relu(bn2(c2) * gate[b,c] + res): the block's tail in one pass (bn2's output is never written; the backward pass recomputes it).
__global__ __launch_bounds__(256) void se_tail_fwd_kernel(const f4* __restrict__ c2, const f4* __restrict__ res, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ gate, f4* __restrict__ out, size_t n4, int hwq, int cq_n,
                                                          unsigned* __restrict__ relu_bits) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % cq_n) * 4;
        const size_t b = i / (size_t)hwq;
        const f4 x = c2[i], r = res[i];
        const f4 mu = *reinterpret_cast<const f4*>(mean + c), rs = *reinterpret_cast<const f4*>(rstd + c);
        const f4 ga = *reinterpret_cast<const f4*>(gamma + c), be = *reinterpret_cast<const f4*>(beta + c);
        const f4 gt = *reinterpret_cast<const f4*>(gate + b * (size_t)(cq_n * 4) + c);
        f4 v;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = fmaxf(((x[k] - mu[k]) * rs[k] * ga[k] + be[k]) * gt[k] + r[k], 0.f);
        out[i] = v;
        if (relu_bits) {
            // the ReLU mask [out > 0] as one nibble per float4 (bit k = component k), eight float4 indices per word: the backward passes read 1/32 of
            // a map instead of `out`.  n4 is a multiple of 8 and i = tid (mod 8), so the eight lanes of a word are in or out of this iteration together.
            unsigned w = ((v[0] > 0.f ? 1u : 0u) | (v[1] > 0.f ? 2u : 0u) | (v[2] > 0.f ? 4u : 0u) | (v[3] > 0.f ? 8u : 0u)) << (4 * (threadIdx.x & 7));
            w |= __shfl_xor(w, 1, 64);
            w |= __shfl_xor(w, 2, 64);
            w |= __shfl_xor(w, 4, 64);
            if ((threadIdx.x & 7) == 0) relu_bits[i >> 3] = w;
        }
    }
}

// Backward of the SE gate of one clip.  With d_pre = dout * [out > 0], S1 = sum_hw d_pre, S2 = rstd * sum_hw d_pre (c2 - mean):
//   dgate = sum_hw d_pre * bn2 = gamma S2 + beta S1;   dz2 = dgate g (1 - g);   dh = W2^T dz2;   dz1 = dh [h > 0];   dpool = W1^T dz1;
// bn2's element gradient is d_pre * gate + dpool / HW, so its two BatchNorm reductions are sums over clips of
//   u1 = gate S1 + dpool        u2 = gate S2 + dpool * P1 / HW,   P1 = sum_hw xhat = (clip_sum - HW mean) rstd
// and no second reduction pass over the map is needed.
__global__ __launch_bounds__(256) void se_gate_train_bwd_kernel(const float* __restrict__ s1, const float* __restrict__ s2raw,
                                                                const float* __restrict__ clip_sum, const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, const float* __restrict__ gate, const float* __restrict__ h,
                                                                const float* __restrict__ w1, const float* __restrict__ w2, float* __restrict__ dz2,
                                                                float* __restrict__ dz1, float* __restrict__ dgap_hw, float* __restrict__ u1,
                                                                float* __restrict__ u2, int C, float hw) {
    __shared__ float sz2[256], sz1[32];
    const int b = blockIdx.x, t = threadIdx.x, Ch = C >> 3;
    float S1 = 0.f, S2 = 0.f, g = 0.f;
    if (t < C) {
        S1 = s1[(size_t)b * C + t];
        S2 = rstd[t] * s2raw[(size_t)b * C + t];
        g = gate[(size_t)b * C + t];
        const float z = (gamma[t] * S2 + beta[t] * S1) * g * (1.f - g);
        sz2[t] = z;
        dz2[(size_t)b * C + t] = z;
    }
    __syncthreads();
    if (t < Ch) {
        float a = 0.f;
        for (int c = 0; c < C; ++c) a += w2[(size_t)c * Ch + t] * sz2[c];
        a = h[(size_t)b * Ch + t] > 0.f ? a : 0.f;
        sz1[t] = a;
        dz1[(size_t)b * Ch + t] = a;
    }
    __syncthreads();
    if (t < C) {
        float dp = 0.f;
        for (int j = 0; j < Ch; ++j) dp += w1[(size_t)j * C + t] * sz1[j];
        const float P1 = (clip_sum[(size_t)b * C + t] - hw * mean[t]) * rstd[t];
        dgap_hw[(size_t)b * C + t] = dp / hw;
        u1[(size_t)b * C + t] = g * S1 + dp;
        u2[(size_t)b * C + t] = g * S2 + dp * P1 / hw;
    }
}

// Sums over clips (fixed order: lanes stride over clips, butterfly wave reduction): bn2's parameter gradients and reduction means, and the
// SE layer's weight gradients.  One wave per channel c.
__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <int CH>
__global__ __launch_bounds__(64) void se_tail_bwd_finish_kernel(const float* __restrict__ u1, const float* __restrict__ u2, const float* __restrict__ dz2,
                                                                const float* __restrict__ dz1, const float* __restrict__ h, const float* __restrict__ pooled,
                                                                float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ m1,
                                                                float* __restrict__ m2, float* __restrict__ dw1, float* __restrict__ db1,
                                                                float* __restrict__ dw2, float* __restrict__ db2, int B, int C, float inv_rows) {
    const int c = blockIdx.x, lane = threadIdx.x;
    float a2[CH], a1[CH], e1[CH], su1 = 0.f, su2 = 0.f, sz = 0.f;
#pragma unroll
    for (int j = 0; j < CH; ++j) a2[j] = a1[j] = e1[j] = 0.f;
    for (int b = lane; b < B; b += 64) {
        const float z2 = dz2[(size_t)b * C + c], p = pooled[(size_t)b * C + c];
        su1 += u1[(size_t)b * C + c];
        su2 += u2[(size_t)b * C + c];
        sz += z2;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const float z1 = dz1[(size_t)b * CH + j];
            a2[j] += z2 * h[(size_t)b * CH + j];            // dW2[c][j] = sum_b dz2[b,c] h[b,j]
            a1[j] += z1 * p;                                // dW1[j][c] = sum_b dz1[b,j] pooled[b,c]
            e1[j] += z1;
        }
    }
    su1 = wave_sum_f(su1); su2 = wave_sum_f(su2); sz = wave_sum_f(sz);
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        const float x2 = wave_sum_f(a2[j]), x1 = wave_sum_f(a1[j]), y1 = wave_sum_f(e1[j]);
        if (lane == 0) {
            dw2[(size_t)c * CH + j] = x2;
            dw1[(size_t)j * C + c] = x1;
            if (c == 0) db1[j] = y1;
        }
    }
    if (lane == 0) {
        dbeta[c] = su1; m1[c] = su1 * inv_rows;
        dgamma[c] = su2; m2[c] = su2 * inv_rows;
        db2[c] = sz;
    }
}

// dres = d_pre = dout [out > 0];  dc2 = gamma rstd (db2 - m1 - xhat m2),  db2 = d_pre gate + dgap_hw,  xhat = (c2 - mean) rstd.
__global__ __launch_bounds__(256) void se_tail_bwd_apply_kernel(const f4* __restrict__ dout, const f4* __restrict__ out, const f4* __restrict__ c2,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                const float* __restrict__ gamma, const float* __restrict__ gate,
                                                                const float* __restrict__ dgap_hw, const float* __restrict__ m1, const float* __restrict__ m2,
                                                                f4* __restrict__ dc2, f4* __restrict__ dres, size_t n4, int hwq, int cq_n,
                                                                const unsigned* __restrict__ relu_bits) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % cq_n) * 4;
        const size_t b = i / (size_t)hwq;
        const f4 d = dout[i], x = c2[i];
        f4 o;
        if (relu_bits) {
            const unsigned nb = relu_nibble(relu_bits, i);
            o = (f4){(nb & 1u) ? 1.f : 0.f, (nb & 2u) ? 1.f : 0.f, (nb & 4u) ? 1.f : 0.f, (nb & 8u) ? 1.f : 0.f};
        } else {
            o = out[i];
        }
        const f4 mu = *reinterpret_cast<const f4*>(mean + c), rs = *reinterpret_cast<const f4*>(rstd + c), ga = *reinterpret_cast<const f4*>(gamma + c);
        const f4 a1 = *reinterpret_cast<const f4*>(m1 + c), a2 = *reinterpret_cast<const f4*>(m2 + c);
        const f4 gt = *reinterpret_cast<const f4*>(gate + b * (size_t)(cq_n * 4) + c);
        const f4 dg = *reinterpret_cast<const f4*>(dgap_hw + b * (size_t)(cq_n * 4) + c);
        f4 dp, dx;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            dp[k] = o[k] > 0.f ? d[k] : 0.f;
            const float xh = (x[k] - mu[k]) * rs[k];
            dx[k] = ga[k] * rs[k] * (dp[k] * gt[k] + dg[k] - a1[k] - xh * a2[k]);
        }
        if (dres) dres[i] = dp;        // null: the consumer masks dout itself (eg_conv3x3_res_masked reads dout and the same bits)
        dc2[i] = dx;
    }
}

// ---- elementwise ----------------------------------------------------------------------------------------------------------
// op 0: y = max(x,0)            op 1: dx = dy * (x > 0)            op 2: y = x > 0 ? x : s*x        op 3: dx = dy * (x > 0 ? 1 : s)
// op 4: y = a + b               op 5: y = a * s                    op 6: y = sigmoid(x)             op 7: dx = dy * y * (1 - y)  (a = dy, b = y)
// op 8: y = a * b               op 9: y = a + s*b                  op 10: y = exp(s * a)            op 11: y = a * b[0]  (scalar kept on the device)
__global__ __launch_bounds__(256) void ew_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, size_t n, int op,
                                                 float s) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float u = a[i], v = b ? (op == 11 ? b[0] : b[i]) : 0.f;
        float r;
        switch (op) {
            case 11: r = u * v; break;
            case 0: r = fmaxf(u, 0.f); break;
            case 1: r = v > 0.f ? u : 0.f; break;               // a = dy, b = x
            case 2: r = u > 0.f ? u : s * u; break;
            case 3: r = v > 0.f ? u : s * u; break;             // a = dy, b = x
            case 4: r = u + v; break;
            case 5: r = u * s; break;
            case 6: r = 1.f / (1.f + expf(-u)); break;
            case 7: r = u * v * (1.f - v); break;
            case 8: r = u * v; break;
            case 9: r = u + s * v; break;
            default: r = expf(s * u); break;
        }
        y[i] = r;
    }
}

// ---- conv3x3 weight image, built on the device (training: the weights change every step) ----------------------------------------
// OIHW fp32 -> the image eg_conv3x3 reads (conv.hip: launch_conv): fp32 [tap][ci/4][coutp][4], then bf16 hi and lo images
// [tap][ci/8][coutp][8].  flip != 0 packs the filter of the input-gradient convolution instead: w'[ci][co][kh][kw] = w[co][ci][2-kh][2-kw]
// (the result then convolves `cout` input channels into `cin` output channels).  One thread per (tap, channel octet, output channel).
__device__ __forceinline__ void pack_conv3x3_item(const float* __restrict__ w, int cout, int cin, int flip, float* __restrict__ image, int idx, bool skip_f32 = false) {
    const int CI = flip ? cout : cin, CO = flip ? cin : cout, coutp = (CO + 15) / 16 * 16;
    const int total = 9 * (CI / 8) * coutp;
    if (idx >= total) return;
    const int co = idx % coutp, oc = (idx / coutp) % (CI / 8), tap = idx / (coutp * (CI / 8));
    const int st = flip ? 8 - tap : tap;
    f4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
    if (co < CO) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int ci = oc * 8 + j;
            const float x = flip ? w[((size_t)ci * cin + co) * 9 + st] : w[((size_t)co * cin + ci) * 9 + st];
            if (j < 4) v0[j] = x; else v1[j - 4] = x;
        }
    }
    if (!skip_f32) {            // the fp32 head of the image: read by the fp32 kernels (and by eg_se_gate_pre), never by the split-bf16 ones
        f4* f32img = reinterpret_cast<f4*>(image);
        f32img[((size_t)tap * (CI / 4) + oc * 2) * coutp + co] = v0;
        f32img[((size_t)tap * (CI / 4) + oc * 2 + 1) * coutp + co] = v1;
    }
    bf8 hi, lo;
    split_octet<true>(v0, v1, hi, lo);
    bf8* himg = reinterpret_cast<bf8*>(image + (size_t)9 * CI * coutp);
    bf8* limg = himg + (size_t)9 * (CI / 8) * coutp;
    himg[idx] = hi;
    limg[idx] = lo;
}
__global__ __launch_bounds__(256) void pack_conv3x3_kernel(const float* __restrict__ w, int cout, int cin, int flip, float* __restrict__ image) {
    pack_conv3x3_item(w, cout, cin, flip, image, blockIdx.x * 256 + threadIdx.x);
}

// ---- nn.Linear weight image, built on the device --------------------------------------------------------------------------------------
// [n][k] fp32 (row stride ld) -> the image eg_linear's split-bf16 modes read (gemm.hip: fill_common): fp32 [rows][kpad], then tile-planar bf16
// hi and lo images [rows/64][kpad/8][64][8]; rows = n rounded up to 64, kpad = k rounded up to 64, zero padded.  transpose != 0 packs
// the transposed matrix (the `weight` of dX = dY W is W^T): image row r, column q = w[q][r].  One thread per (row tile, k octet, row).
__device__ __forceinline__ void pack_linear_item(const float* __restrict__ w, int ld, int n, int k, int transpose, float* __restrict__ image, int idx,
                                                 bool skip_f32 = false) {
    const int rows = (n + 63) / 64 * 64, kpad = (k + 63) / 64 * 64, KO = kpad / 8;
    if (idx >= rows * KO) return;
    const int r = idx & 63, ko = (idx >> 6) % KO, rt = idx / (64 * KO);
    const int row = rt * 64 + r;
    f4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
    if (row < n) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int q = ko * 8 + j;
            const float x = q < k ? (transpose ? w[(size_t)q * ld + row] : w[(size_t)row * ld + q]) : 0.f;
            if (j < 4) v0[j] = x; else v1[j - 4] = x;
        }
    }
    if (!skip_f32) {
        f4* f32img = reinterpret_cast<f4*>(image + (size_t)row * kpad + ko * 8);
        f32img[0] = v0;
        f32img[1] = v1;
    }
    bf8 hi, lo;
    split_octet<true>(v0, v1, hi, lo);
    bf8* himg = reinterpret_cast<bf8*>(image + (size_t)rows * kpad);
    bf8* limg = himg + (size_t)rows * KO;
    himg[idx] = hi;
    limg[idx] = lo;
}
__global__ __launch_bounds__(256) void pack_linear_kernel(const float* __restrict__ w, int ld, int n, int k, int transpose, float* __restrict__ image) {
    pack_linear_item(w, ld, n, k, transpose, image, blockIdx.x * 256 + threadIdx.x);
}

// All weight images of a training step in ONE launch: a table of (source, image, shape) entries built once (parameters and images live at fixed
// addresses), each entry owning a run of workgroups; a workgroup finds its entry by bisection over the run starts.  (A step packed ~200 weights
// one launch each: 8 % of its launches.)
struct PackEntry { const float* src; float* image; int kind, a, b, c, flag, first_block; };      // kind 0: Linear (n, k, ld, transpose); 1: conv3x3 (cout, cin, -, flip)
__global__ __launch_bounds__(256) void pack_table_kernel(const PackEntry* __restrict__ table, int count) {
    int lo = 0, hi = count - 1;
    const int blk = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].first_block <= blk) lo = mid; else hi = mid - 1;
    }
    const PackEntry e = table[lo];
    const int idx = (blk - e.first_block) * 256 + threadIdx.x;
    // flag bit 0: transpose / flip; bit 1: leave the fp32 head of the image unwritten (a training step in the split-bf16 modes never reads it: half the
    // bytes of the step's image refresh)
    if (e.kind == 0) pack_linear_item(e.src, e.c, e.a, e.b, e.flag & 1, e.image, idx, (e.flag & 2) != 0);
    else pack_conv3x3_item(e.src, e.a, e.b, e.flag & 1, e.image, idx, (e.flag & 2) != 0);
}

// ---- dropout: counter-based mask, nothing stored -- keep(i) = hash(seed, offset + i) >= p; y = keep ? x / (1 - p) : 0.  The backward pass
// is the same kernel on dy with the same (seed, offset).  (nn.Dropout's semantics; the mask stream is this library's own, not torch's.)
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n, float p, float inv_keep,
                                                      unsigned int seed, unsigned long long offset, const int* __restrict__ epoch) {
    const unsigned int thr = (unsigned int)(p * 4294967296.0);
    seed = dropout_seed(seed, epoch);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        y[i] = dropout_keep(seed, offset + i, thr) ? x[i] * inv_keep : 0.f;
}

// ---- SE pieces: x [B, HW, C] --------------------------------------------------------------------------------------------
// segment mean over HW (two-level through col_partial per clip would need B launches; one block per (clip, 64 channels) instead)
__global__ __launch_bounds__(256) void seg_mean_kernel(const float* __restrict__ x, float* __restrict__ out, int HW, int C, float scale) {
    __shared__ float s[4][64];
    const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
    float u = 0.f;
    if (c < C)
        for (int p = w; p < HW; p += 4) u += x[((size_t)b * HW + p) * C + c];
    s[w][threadIdx.x & 63] = u;
    __syncthreads();
    if (w == 0 && c < C) {
        const int l = threadIdx.x;
        out[(size_t)b * C + c] = ((s[0][l] + s[1][l]) + (s[2][l] + s[3][l])) * scale;
    }
}
// mode 0: y = x * gate[b,c];  mode 1: y = a * gate[b,c] + dgap[b,c]  (a = dy: dx of the scaled map plus the pooled branch's gradient);
__global__ __launch_bounds__(256) void se_scale_kernel(const float* __restrict__ a, const float* __restrict__ gate, const float* __restrict__ add,
                                                       float* __restrict__ y, size_t total, int HW, int C) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const size_t b = i / ((size_t)HW * C);
        const float g = gate[b * C + c];
        y[i] = a[i] * g + (add ? add[b * C + c] : 0.f);
    }
}
// dgate[b,c] = sum_hw dy * x
__global__ __launch_bounds__(256) void seg_dot_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ out, int HW,
                                                      int C) {
    __shared__ float s[4][64];
    const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
    float u = 0.f;
    if (c < C)
        for (int p = w; p < HW; p += 4) {
            const size_t i = ((size_t)b * HW + p) * C + c;
            u += dy[i] * x[i];
        }
    s[w][threadIdx.x & 63] = u;
    __syncthreads();
    if (w == 0 && c < C) {
        const int l = threadIdx.x;
        out[(size_t)b * C + c] = (s[0][l] + s[1][l]) + (s[2][l] + s[3][l]);
    }
}

// ---- LayerNorm backward: one wave per row ----------------------------------------------------------------------------------
// dx = rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat));  also writes xhat, so that ONE column reduction gives both affine gradients:
// (dbeta, dgamma) = (sum dy, sum dy * xhat) = eg_colsum(dy, xhat)
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ gamma,
                                                     float* __restrict__ dx, float* __restrict__ t, int rows, int D, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (size_t)row * D;
    const float* dr = dy + (size_t)row * D;
    float s = 0.f;
    for (int i = lane; i < D; i += 64) s += xr[i];
    const float mean = wave_sum(s) / (float)D;
    float ss = 0.f;
    for (int i = lane; i < D; i += 64) { const float d = xr[i] - mean; ss += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)D + eps);
    float a = 0.f, b = 0.f;
    for (int i = lane; i < D; i += 64) {
        const float xh = (xr[i] - mean) * rstd, gd = gamma[i] * dr[i];
        a += gd;
        b += gd * xh;
    }
    a = wave_sum(a) / (float)D;
    b = wave_sum(b) / (float)D;
    for (int i = lane; i < D; i += 64) {
        const float xh = (xr[i] - mean) * rstd;
        dx[(size_t)row * D + i] = rstd * (gamma[i] * dr[i] - a - xh * b);
        t[(size_t)row * D + i] = xh;
    }
}

// LayerNorm backward for the fused transformer blocks: a wave walks RW rows with the row held in registers (NV float4 per lane: the column quads
// lane + 64 i), keeps the two affine-gradient column sums (sum dy, sum dy * xhat) in registers across its rows, and the workgroup's four waves are
// folded through LDS in a fixed order into one partial per workgroup ([blk][2][D], the layout col_finalize_kernel folds).  Optional: a second output,
// dx with the Dropout mask of the branch that precedes the residual add, and that branch gradient as bf16 (hi, lo) tile-planar images for a
// pre-split input-gradient product.
template <int NV>
__global__ __launch_bounds__(256) void ln_bwd_ex_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ gamma,
                                                        float* __restrict__ dx, float* __restrict__ dxm, float* __restrict__ part, int rows, int D,
                                                        int RW, float eps, EgDropout dr, unsigned short* __restrict__ img) {
    extern __shared__ float lsum[];                      // [4 waves][2][D]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + wave) * RW, nq = D >> 2;
    const unsigned int dseed = dr.thr ? dropout_seed(dr.seed, dr.epoch) : 0u;
    const float invD = 1.0f / (float)D;
    const f4 z4 = (f4){0.f, 0.f, 0.f, 0.f};
    f4 sb[NV], sg[NV], gm[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        sb[i] = sg[i] = z4;
        gm[i] = (lane + i * 64 < nq) ? reinterpret_cast<const f4*>(gamma)[lane + i * 64] : z4;
    }
    for (int j = 0; j < RW; ++j) {
        const int row = row0 + j;
        if (row >= rows) break;
        const f4* xr = reinterpret_cast<const f4*>(x + (size_t)row * D);
        const f4* dr_ = reinterpret_cast<const f4*>(dy + (size_t)row * D);
        f4 xv[NV], dv[NV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const bool ok = lane + i * 64 < nq;
            xv[i] = ok ? xr[lane + i * 64] : z4;
            dv[i] = ok ? dr_[lane + i * 64] : z4;
            s += (xv[i][0] + xv[i][1]) + (xv[i][2] + xv[i][3]);
        }
        const float mean = wave_sum(s) * invD;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (lane + i * 64 < nq) {
                const f4 d = xv[i] - mean;
                ss += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
            }
        const float rstd = 1.0f / sqrtf(wave_sum(ss) * invD + eps);
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const f4 xh = (xv[i] - mean) * rstd, gd = gm[i] * dv[i];      // quads past D: gm = dv = 0
            xv[i] = xh;
            a += (gd[0] + gd[1]) + (gd[2] + gd[3]);
            b += (gd[0] * xh[0] + gd[1] * xh[1]) + (gd[2] * xh[2] + gd[3] * xh[3]);
        }
        a = wave_sum(a) * invD;
        b = wave_sum(b) * invD;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int q = lane + i * 64;
            if (q < nq) {
                const f4 g = (gm[i] * dv[i] - a - xv[i] * b) * rstd;
                reinterpret_cast<f4*>(dx + (size_t)row * D)[q] = g;
                f4 br = g;
                if (dxm) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        br[r] = dropout_keep(dseed, dr.offset + (unsigned long long)row * D + 4 * q + r, dr.thr) ? g[r] * dr.inv_keep : 0.f;
                    reinterpret_cast<f4*>(dxm + (size_t)row * D)[q] = br;
                }
                if (img) {
                    bf8 h8, l8;
                    split_octet<true>(br, z4, h8, l8);
                    const int KO = D >> 3;
                    const size_t slot = (((size_t)(row >> 6) * KO + (q >> 1)) * 64 + (row & 63)) * 8 + (q & 1) * 4;
                    const size_t lo_off = (size_t)((rows + 63) >> 6) * KO * 512;
                    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                    const u32x4_t hh = __builtin_bit_cast(u32x4_t, h8), ll = __builtin_bit_cast(u32x4_t, l8);
                    *reinterpret_cast<u32x2*>(img + slot) = (u32x2){hh[0], hh[1]};
                    *reinterpret_cast<u32x2*>(img + lo_off + slot) = (u32x2){ll[0], ll[1]};
                }
                sb[i] += dv[i];
                sg[i] += dv[i] * xv[i];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int q = lane + i * 64;
        if (q < nq) {
            reinterpret_cast<f4*>(lsum + (wave * 2) * D)[q] = sb[i];
            reinterpret_cast<f4*>(lsum + (wave * 2 + 1) * D)[q] = sg[i];
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * D; c += 256) {
        const int which = c / D, col = c - which * D;
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) t += lsum[(w * 2 + which) * D + col];
        part[((size_t)blockIdx.x * 2 + which) * D + col] = t;
    }
}

// ---- attention backward (Full_model/Modules.py:13-23), one workgroup per (head, clip), Lq, Lk <= 64 -------------------------
// given P (the forward's probabilities) and dO:  dV = P^T dO;  dP = dO V^T;  dS = P * (dP - rowsum(dP * P));
// dQ = dS K / temp;  dK = dS^T Q / temp
__global__ __launch_bounds__(256) void attention_bwd_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k, int ldk,
                                                            const float* __restrict__ v, int ldv, const float* __restrict__ p,
                                                            const float* __restrict__ dout, int ldo, float* __restrict__ dq, int lddq,
                                                            float* __restrict__ dk, int lddk, float* __restrict__ dv, int lddv, int H, int Lq,
                                                            int Lk, float inv_temp) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    float* Qs = sm;                     // [Lq][65]
    float* Ks = Qs + Lq * 65;           // [Lk][65]
    float* Vs = Ks + Lk * 65;           // [Lk][65]
    float* Ds = Vs + Lk * 65;           // dO [Lq][65]
    float* Ps = Ds + Lq * 65;           // P  [Lq][Lk+1]
    float* Ss = Ps + Lq * (Lk + 1);     // dS [Lq][Lk+1]
    for (int i = tid; i < Lq * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        Qs[r * 65 + c] = q[((size_t)b * Lq + r) * ldq + h * 64 + c];
        Ds[r * 65 + c] = dout[((size_t)b * Lq + r) * ldo + h * 64 + c];
    }
    for (int i = tid; i < Lk * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        Ks[r * 65 + c] = k[((size_t)b * Lk + r) * ldk + h * 64 + c];
        Vs[r * 65 + c] = v[((size_t)b * Lk + r) * ldv + h * 64 + c];
    }
    for (int i = tid; i < Lq * Lk; i += 256) {
        const int r = i / Lk, c = i - r * Lk;
        Ps[r * (Lk + 1) + c] = p[(((size_t)b * H + h) * Lq + r) * Lk + c];
    }
    __syncthreads();
    // dP
    for (int i = tid; i < Lq * Lk; i += 256) {
        const int r = i / Lk, c = i - r * Lk;
        float s = 0.f;
        for (int d = 0; d < 64; ++d) s += Ds[r * 65 + d] * Vs[c * 65 + d];
        Ss[r * (Lk + 1) + c] = s;
    }
    __syncthreads();
    // dS = P * (dP - sum_k dP*P), one wave per row
    const int wave = tid >> 6, lane = tid & 63;
    for (int r = wave; r < Lq; r += 4) {
        float s = 0.f;
        for (int c = lane; c < Lk; c += 64) s += Ss[r * (Lk + 1) + c] * Ps[r * (Lk + 1) + c];
        s = wave_sum(s);
        for (int c = lane; c < Lk; c += 64) Ss[r * (Lk + 1) + c] = Ps[r * (Lk + 1) + c] * (Ss[r * (Lk + 1) + c] - s);
    }
    __syncthreads();
    for (int i = tid; i < Lq * 64; i += 256) {           // dQ
        const int r = i >> 6, d = i & 63;
        float s = 0.f;
        for (int c = 0; c < Lk; ++c) s += Ss[r * (Lk + 1) + c] * Ks[c * 65 + d];
        dq[((size_t)b * Lq + r) * lddq + h * 64 + d] = s * inv_temp;
    }
    for (int i = tid; i < Lk * 64; i += 256) {           // dK, dV
        const int c = i >> 6, d = i & 63;
        float s = 0.f, t = 0.f;
        for (int r = 0; r < Lq; ++r) {
            s += Ss[r * (Lk + 1) + c] * Qs[r * 65 + d];
            t += Ps[r * (Lk + 1) + c] * Ds[r * 65 + d];
        }
        dk[((size_t)b * Lk + c) * lddk + h * 64 + d] = s * inv_temp;
        dv[((size_t)b * Lk + c) * lddv + h * 64 + d] = t;
    }
}

// ---- losses (one workgroup; sums in fixed order) -----------------------------------------------------------------------------
// smooth-L1 (nn.SmoothL1Loss / F.smooth_l1_loss, mean reduction, beta): loss = mean(|d|<beta ? 0.5 d^2/beta : |d| - 0.5 beta)
__global__ __launch_bounds__(256) void smooth_l1_kernel(const float* __restrict__ pred, const float* __restrict__ target, float* __restrict__ dpred,
                                                        float* __restrict__ partial, size_t n, float beta, float scale) {
    __shared__ float s[256];
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float d = pred[i] - target[i], ad = fabsf(d);
        acc += ad < beta ? 0.5f * d * d / beta : ad - 0.5f * beta;
        if (dpred) dpred[i] = scale * (ad < beta ? d / beta : (d > 0.f ? 1.f : -1.f));
    }
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = s[0];
}
__global__ void sum_small_kernel(const float* __restrict__ partial, int n, float scale, float* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += partial[i];
        out[0] = (float)(s * scale);
    }
}
// cross entropy on logits [B, C] with integer labels; focal: alpha[b] * (1 - pt)^gamma * ce with a PER-SAMPLE weight vector, as
// `self.alpha * (1-pt)**self.gamma * ce_loss` broadcasts it (train_audio_classifier_K_fold.py:95-105),
// gamma < 0 selects plain CE.  per-sample loss -> rowloss; dlogits = d(mean loss * scale)/dlogits
__global__ __launch_bounds__(64) void ce_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels, const float* __restrict__ alpha,
                                                float gamma, float scale, int B, int C, float* __restrict__ rowloss, float* __restrict__ dlogits) {
    const int b = blockIdx.x;
    if (b >= B) return;
    const int lane = threadIdx.x;
    const float* z = logits + (size_t)b * C;
    float m = -3.0e38f;
    for (int c = lane; c < C; c += 64) m = fmaxf(m, z[c]);
    m = wave_max(m);
    float se = 0.f;
    for (int c = lane; c < C; c += 64) se += expf(z[c] - m);
    se = wave_sum(se);
    const int y = (int)labels[b];
    const float logp = z[y] - m - logf(se), ce = -logp, pt = expf(logp);
    float loss = ce, dce = 1.f;      // d loss / d ce (through pt = exp(-ce) as well)
    if (gamma >= 0.f) {
        const float a = alpha ? alpha[b] : 1.f, om = 1.f - pt;
        const float w = powf(om, gamma);
        loss = a * w * ce;
        // d/dce [a (1-e^{-ce})^g ce] = a [ (1-pt)^g + g (1-pt)^{g-1} pt ce ]
        dce = a * (w + (gamma > 0.f ? gamma * powf(om, gamma - 1.f) * pt * ce : 0.f));
    }
    if (lane == 0) rowloss[b] = loss;
    if (dlogits)
        for (int c = lane; c < C; c += 64) {
            const float sm = expf(z[c] - m) / se;
            dlogits[(size_t)b * C + c] = scale * dce * (sm - (c == y ? 1.f : 0.f)) / (float)B;
        }
}

// KL divergence of N(mu, exp(logvar)) from N(0, I), the standard VAE term: loss = scale * mean_b( -0.5 * sum_j (1 + lv - mu^2 - e^lv) )
__global__ __launch_bounds__(256) void kld_kernel(const float* __restrict__ mu, const float* __restrict__ lv, float* __restrict__ dmu,
                                                  float* __restrict__ dlv, float* __restrict__ loss, int n, int d, float scale) {
    __shared__ float s[256];
    float acc = 0.f;
    const float k = scale / (float)n;
    for (int i = threadIdx.x; i < n * d; i += 256) {
        const float m = mu[i], l = lv[i], e = expf(l);
        acc += 1.f + l - m * m - e;
        if (dmu) { dmu[i] = k * m; dlv[i] = k * 0.5f * (e - 1.f); }
    }
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = -0.5f * k * s[0];
}

// ---- Adam (torch.optim.Adam: grad += wd * p; m, v EMA; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)) ---------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                   size_t n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float gi = g[i] + wd * p[i];
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
    }
}

// The same update with the step count read from device memory (a captured hipGraph replays with a count that advances on the device).
__global__ __launch_bounds__(256) void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                       size_t n, float lr, float b1, float b2, float eps, float wd, const int* __restrict__ step) {
    const int t = *step;
    const float bc1 = (float)(1.0 - pow((double)b1, (double)t)), bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, (double)t));
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float gi = g[i] + wd * p[i];
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
    }
}
// The float4 form both entry points launch when the four slices share a 16-byte phase (they do: one offset into four separately allocated flat
// buffers): every thread owns four float4 per stream, all sixteen loads issued before any arithmetic (the scalar grid-stride loop above streamed
// at 3.5 TB/s, round 6); the bias corrections of the device-count form are computed behind the loads.  The up to three elements in front of the
// first 16-byte boundary and the n mod 4 behind the last float4 are updated by the first threads of workgroup 0 (no extra launch).  Same
// per-element expression, same results.
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt) {
    const float gi = g + wd * p;
    const float mi = b1 * m + (1.f - b1) * gi;
    const float vi = b2 * v + (1.f - b2) * gi * gi;
    m = mi;
    v = vi;
    p -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
}
template <bool DEV>
__global__ __launch_bounds__(256) void adam_v4_kernel(float* __restrict__ ps, const float* __restrict__ gs, float* __restrict__ ms, float* __restrict__ vs,
                                                      int head, size_t n4, int tail, float lr, float b1, float b2, float eps, float wd, float bc1,
                                                      float bc2_sqrt, const int* __restrict__ step) {
    f4* p = reinterpret_cast<f4*>(ps + head);
    const f4* g = reinterpret_cast<const f4*>(gs + head);
    f4* m = reinterpret_cast<f4*>(ms + head);
    f4* v = reinterpret_cast<f4*>(vs + head);
    const size_t i0 = (size_t)blockIdx.x * 1024 + threadIdx.x;
    f4 pv[4], gv[4], mv[4], vv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const size_t i = i0 + (size_t)k * 256;
        if (i < n4) { pv[k] = p[i]; gv[k] = g[i]; mv[k] = m[i]; vv[k] = v[i]; }
    }
    if (DEV) {
        const int t = *step;
        bc1 = (float)(1.0 - pow((double)b1, (double)t));
        bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, (double)t));
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const size_t i = i0 + (size_t)k * 256;
        if (i < n4) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float pe = pv[k][r], me = mv[k][r], ve = vv[k][r];
                adam_one(pe, gv[k][r], me, ve, lr, b1, b2, eps, wd, bc1, bc2_sqrt);
                pv[k][r] = pe; mv[k][r] = me; vv[k][r] = ve;
            }
            m[i] = mv[k]; v[i] = vv[k]; p[i] = pv[k];
        }
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < head + tail) {
        const size_t e = (int)threadIdx.x < head ? (size_t)threadIdx.x : (size_t)head + n4 * 4 + (threadIdx.x - head);
        float pe = ps[e], me = ms[e], ve = vs[e];
        adam_one(pe, gs[e], me, ve, lr, b1, b2, eps, wd, bc1, bc2_sqrt);
        ms[e] = me; vs[e] = ve; ps[e] = pe;
    }
}
template <bool DEV>
void launch_adam(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                 const int* step, hipStream_t st) {
    const uintptr_t ph = reinterpret_cast<uintptr_t>(p) & 15;
    const bool same_phase = (reinterpret_cast<uintptr_t>(g) & 15) == ph && (reinterpret_cast<uintptr_t>(m) & 15) == ph &&
                            (reinterpret_cast<uintptr_t>(v) & 15) == ph && (ph & 3) == 0;
    if (!same_phase || n < 8) {
        if (DEV) hipLaunchKernelGGL(adam_dev_kernel, grid1(n, 16384), dim3(256), 0, st, p, g, m, v, n, lr, b1, b2, eps, wd, step);
        else hipLaunchKernelGGL(adam_kernel, grid1(n, 16384), dim3(256), 0, st, p, g, m, v, n, lr, b1, b2, eps, wd, bc1, bc2_sqrt);
        return;
    }
    const int head = (int)(((16 - ph) & 15) / 4);
    const size_t n4 = (n - head) / 4;
    const int tail = (int)(n - head - n4 * 4);
    hipLaunchKernelGGL(adam_v4_kernel<DEV>, dim3((unsigned)((n4 + 1023) / 1024)), dim3(256), 0, st, p, g, m, v, head, n4, tail, lr, b1, b2, eps, wd, bc1,
                       bc2_sqrt, step);
}
__global__ void counter_add_kernel(int* c, int d) { *c += d; }

// ---- Models_memory.py: the two memory nets of Prior_MemoryEncoder under autograd (the MLPs around them are ordinary Linear ops) ----------------
// SP_Memory_Net_v1 (:239-249): for the first `chunk` predicted frames  s = sigmoid(<m_b, p_bc>),  out_bc = s p_bc + (1 - s) m_b;  later frames
// pass through.  One wave per (clip, frame); the gate s is kept for the backward pass.
__global__ __launch_bounds__(64) void sp_gate_fwd_kernel(const float* __restrict__ m, const float* __restrict__ p, float* __restrict__ out,
                                                         float* __restrict__ gate, int P, int D, int chunk) {
    const int b = blockIdx.x / P, c = blockIdx.x - b * P, lane = threadIdx.x;
    const float* pr = p + ((size_t)b * P + c) * D;
    float* o = out + ((size_t)b * P + c) * D;
    if (c >= chunk) {
        for (int d = lane; d < D; d += 64) o[d] = pr[d];
        return;
    }
    const float* mr = m + (size_t)b * D;
    float dot = 0.f;
    for (int d = lane; d < D; d += 64) dot += mr[d] * pr[d];
    dot = wave_sum(dot);
    const float sg = 1.f / (1.f + expf(-dot));
    for (int d = lane; d < D; d += 64) o[d] = sg * pr[d] + (1.f - sg) * mr[d];
    if (lane == 0) gate[b * chunk + c] = sg;
}
// given g = d loss / d out:  ds = <g, p - m>,  dscore = ds s (1 - s);  dp = g s + dscore m;  dm_b = sum_c (g (1 - s) + dscore p).  One wave per clip.
__global__ __launch_bounds__(64) void sp_gate_bwd_kernel(const float* __restrict__ m, const float* __restrict__ p, const float* __restrict__ gate,
                                                         const float* __restrict__ g, float* __restrict__ dp, float* __restrict__ dm, int P, int D,
                                                         int chunk) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const float* mr = m + (size_t)b * D;
    for (int d = lane; d < D; d += 64) dm[(size_t)b * D + d] = 0.f;
    for (int c = 0; c < P; ++c) {
        const float* pr = p + ((size_t)b * P + c) * D;
        const float* gr = g + ((size_t)b * P + c) * D;
        float* dpr = dp + ((size_t)b * P + c) * D;
        if (c >= chunk) {
            for (int d = lane; d < D; d += 64) dpr[d] = gr[d];
            continue;
        }
        const float sg = gate[b * chunk + c];
        float ds = 0.f;
        for (int d = lane; d < D; d += 64) ds += gr[d] * (pr[d] - mr[d]);
        ds = wave_sum(ds);
        const float dscore = ds * sg * (1.f - sg);
        for (int d = lane; d < D; d += 64) {
            dpr[d] = gr[d] * sg + dscore * mr[d];
            dm[(size_t)b * D + d] += gr[d] * (1.f - sg) + dscore * pr[d];         // same lane owns element d in every pass: fixed order
        }
    }
}
// TM_Memory_Net (:289-292) behind its score: w = softmax_c(score_b), out_bc = p_bc (1 + w_bc) for c < chunk; later frames pass through.
__global__ __launch_bounds__(64) void tm_scale_fwd_kernel(const float* __restrict__ score, const float* __restrict__ p, float* __restrict__ out,
                                                          float* __restrict__ w, int P, int D, int chunk) {
    const int b = blockIdx.x / P, c = blockIdx.x - b * P, lane = threadIdx.x;
    const float* pr = p + ((size_t)b * P + c) * D;
    float* o = out + ((size_t)b * P + c) * D;
    float wc = 0.f;
    if (c < chunk) {
        const float* sr = score + (size_t)b * chunk;
        float mx = -3.0e38f;
        for (int j = 0; j < chunk; ++j) mx = fmaxf(mx, sr[j]);
        float se = 0.f;
        for (int j = 0; j < chunk; ++j) se += expf(sr[j] - mx);
        wc = expf(sr[c] - mx) / se;
        if (lane == 0) w[b * chunk + c] = wc;
    }
    for (int d = lane; d < D; d += 64) o[d] = pr[d] * (1.f + wc);
}
// dp = g (1 + w);  dw_c = <g_c, p_c>;  dscore_c = w_c (dw_c - sum_j w_j dw_j).  One wave per clip.
__global__ __launch_bounds__(64) void tm_scale_bwd_kernel(const float* __restrict__ w, const float* __restrict__ p, const float* __restrict__ g,
                                                          float* __restrict__ dp, float* __restrict__ dscore, int P, int D, int chunk) {
    __shared__ float dw[64];
    const int b = blockIdx.x, lane = threadIdx.x;
    for (int c = 0; c < P; ++c) {
        const float* pr = p + ((size_t)b * P + c) * D;
        const float* gr = g + ((size_t)b * P + c) * D;
        float* dpr = dp + ((size_t)b * P + c) * D;
        const float wc = c < chunk ? w[b * chunk + c] : 0.f;
        float acc = 0.f;
        for (int d = lane; d < D; d += 64) {
            dpr[d] = gr[d] * (1.f + wc);
            acc += gr[d] * pr[d];
        }
        if (c < chunk) {
            acc = wave_sum(acc);
            if (lane == 0) dw[c] = acc;
        }
    }
    __syncthreads();
    if (lane < chunk) {
        float dot = 0.f;
        for (int j = 0; j < chunk; ++j) dot += w[b * chunk + j] * dw[j];
        dscore[b * chunk + lane] = w[b * chunk + lane] * (dw[lane] - dot);
    }
}

// gradient-bucket payload conversion (train/optim.GradBuckets, payload "bf16"): fp32 -> bf16 (round to nearest even) and back (x scale)
__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = f32_to_bf16_rne(x[i]);
}
__global__ __launch_bounds__(256) void bf16_to_f32_kernel(const unsigned short* __restrict__ x, float* __restrict__ y, size_t n, float scale) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = bf16_to_f32(x[i]) * scale;
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int eg_transpose(const float* x, int32_t ldx, int32_t rows, int32_t cols, float* y, int32_t ldy, void* stream) {
    EG_REQUIRE(x && y && rows > 0 && cols > 0 && ldx >= cols && ldy >= rows, EG_ERR_BAD_ARG, "eg_transpose: bad argument");
    hipLaunchKernelGGL(transpose_kernel, dim3(eg_cdiv(cols, 32), eg_cdiv(rows, 32)), dim3(256), 0, ST, x, ldx, rows, cols, y, ldy);
    return eg_check_launch("transpose");
}

extern "C" int64_t eg_gemm_tn_workspace_floats(int32_t m, int32_t n, int64_t k) {
    const long tiles = (long)eg_cdiv(m, 64) * eg_cdiv(n, 64);
    long splits = tiles >= 512 ? 1 : (1024 + tiles - 1) / tiles;
    const long max_splits = (k + 255) / 256;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    return splits <= 1 ? 0 : splits * (int64_t)m * n;
}
namespace {
int launch_gemm_tn(bool implicit, const float* a, int lda, const float* b, int ldb, float* c, int ldc, int m, int n, int64_t k, float* workspace,
                   int64_t workspace_floats, int accumulate, ConvGeo geo, hipStream_t st) {
    const int64_t need = eg_gemm_tn_workspace_floats(m, n, k);
    EG_REQUIRE(need == 0 || (workspace && workspace_floats >= need), EG_ERR_WORKSPACE, "eg_gemm_tn: workspace %lld < %lld floats",
               (long long)workspace_floats, (long long)need);
    const int splits = need ? (int)(need / ((int64_t)m * n)) : 1;
    const long kps = ((k + splits - 1) / splits + 31) / 32 * 32;
    const int nz = (int)((k + kps - 1) / kps);
    EgProfScope prof(7, 2.0 * m * (double)n * (double)k, st);
    const bool direct = nz <= 1 && !accumulate;
    float* part = direct ? nullptr : workspace;
    EG_REQUIRE(direct || (part && workspace_floats >= (int64_t)nz * m * n), EG_ERR_WORKSPACE, "eg_gemm_tn: accumulate / split-K needs a workspace");
    dim3 grid(eg_cdiv(m, 64), eg_cdiv(n, 64), direct ? 1 : nz);
    if (implicit) hipLaunchKernelGGL(gemm_tn_kernel<true>, grid, dim3(256), 0, st, a, lda, b, ldb, c, ldc, m, n, (long)k, direct ? (long)k : kps, part, geo);
    else hipLaunchKernelGGL(gemm_tn_kernel<false>, grid, dim3(256), 0, st, a, lda, b, ldb, c, ldc, m, n, (long)k, direct ? (long)k : kps, part, geo);
    if (int rc = eg_check_launch("gemm_tn")) return rc;
    if (direct) return EG_OK;
    hipLaunchKernelGGL(tn_reduce_kernel, dim3(eg_cdiv(m * n, 64)), dim3(256), 0, st, part, c, ldc, m, n, nz, accumulate);
    return eg_check_launch("gemm_tn_reduce");
}
}  // namespace

extern "C" int eg_gemm_tn(const float* a, int32_t lda, const float* b, int32_t ldb, float* c, int32_t ldc, int32_t m, int32_t n, int64_t k,
                          float* workspace, int64_t workspace_floats, int32_t accumulate, void* stream) {
    EG_REQUIRE(a && b && c && m > 0 && n > 0 && k > 0 && lda >= m && ldb >= n && ldc >= n, EG_ERR_BAD_ARG, "eg_gemm_tn: bad argument");
    ConvGeo geo = {0, 0, 0, 0, 0, 0};
    return launch_gemm_tn(false, a, lda, b, ldb, c, ldc, m, n, k, workspace, workspace_floats, accumulate, geo, ST);
}

// Weight gradient of nn.Conv2d(k=3, pad=1, stride s) on NHWC activations as an implicit GEMM (no im2col buffer):
// dw_mat[co][(kh*3 + kw)*Cin + ci] = sum over (b, oy, ox) of dy[b,oy,ox,co] * x[b, oy*s + kh - 1, ox*s + kw - 1, ci].   Cin % 4 == 0.
// workspace >= eg_gemm_tn_workspace_floats(cout, 9*cin, B*Ho*Wo).
extern "C" int eg_conv3x3_wgrad(const float* x, const float* dy, float* dw_mat, int32_t batch, int32_t h, int32_t w, int32_t cin, int32_t cout,
                                int32_t stride, float* workspace, int64_t workspace_floats, void* stream) {
    EG_REQUIRE(x && dy && dw_mat && batch > 0 && h > 0 && w > 0 && cin > 0 && cout > 0 && (stride == 1 || stride == 2), EG_ERR_BAD_ARG,
               "eg_conv3x3_wgrad: bad argument");
    EG_REQUIRE((cin & 3) == 0 && eg_aligned16(x), EG_ERR_ALIGN, "eg_conv3x3_wgrad: Cin %% 4 and a 16-byte aligned activation");
    ConvGeo geo = {h, w, (h + 2 - 3) / stride + 1, (w + 2 - 3) / stride + 1, cin, stride};
    const int64_t k = (int64_t)batch * geo.Ho * geo.Wo;
    return launch_gemm_tn(true, dy, cout, x, 0, dw_mat, 9 * cin, cout, 9 * cin, k, workspace, workspace_floats, 0, geo, ST);
}

extern "C" int eg_im2col3x3(const float* x, float* col, int32_t batch, int32_t h, int32_t w, int32_t c, int32_t stride, int32_t backward,
                            void* stream) {
    EG_REQUIRE(x && col && batch > 0 && h > 0 && w > 0 && c > 0 && (stride == 1 || stride == 2), EG_ERR_BAD_ARG, "eg_im2col3x3: bad argument");
    const int ho = (h + 2 - 3) / stride + 1, wo = (w + 2 - 3) / stride + 1;
    if (!backward)
        hipLaunchKernelGGL(im2col3x3_kernel, grid1((size_t)batch * ho * wo * 9 * c, 65536), dim3(256), 0, ST, x, col, batch, h, w, c, ho, wo, stride);
    else if ((c & 3) == 0 && eg_aligned16(x) && eg_aligned16(col) && (size_t)batch * h * w * c / 4 < (1ull << 31))
        hipLaunchKernelGGL(col2im3x3_v4_kernel, grid1((size_t)batch * h * w * c / 4, 65536), dim3(256), 0, ST, reinterpret_cast<const f4*>(x),
                           reinterpret_cast<f4*>(col), batch, h, w, c / 4, ho, wo, stride);
    else        // x = dcol [B*Ho*Wo, 9C], col = dx [B,H,W,C]
        hipLaunchKernelGGL(col2im3x3_kernel, grid1((size_t)batch * h * w * c, 65536), dim3(256), 0, ST, x, col, batch, h, w, c, ho, wo, stride);
    return eg_check_launch("im2col3x3");
}
extern "C" int eg_subsample(const float* x, float* y, int32_t batch, int32_t h, int32_t w, int32_t c, int32_t stride, int32_t backward, void* stream) {
    EG_REQUIRE(x && y && batch > 0 && h > 0 && w > 0 && c > 0 && stride >= 1, EG_ERR_BAD_ARG, "eg_subsample: bad argument");
    const int ho = (h - 1) / stride + 1, wo = (w - 1) / stride + 1;
    const size_t total = backward ? (size_t)batch * h * w * c : (size_t)batch * ho * wo * c;
    if ((c & 3) == 0 && eg_aligned16(x) && eg_aligned16(y) && (size_t)batch * h * w * c / 4 < (1ull << 31))
        hipLaunchKernelGGL(subsample_v4_kernel, grid1(total / 4, 65536), dim3(256), 0, ST, reinterpret_cast<const f4*>(x), reinterpret_cast<f4*>(y), batch, h, w,
                           c / 4, ho, wo, stride, backward);
    else
        hipLaunchKernelGGL(subsample_kernel, grid1(total, 65536), dim3(256), 0, ST, x, y, batch, h, w, c, ho, wo, stride, backward);
    return eg_check_launch("subsample");
}
extern "C" int eg_im2col1d(const float* x, float* col, int32_t batch, int32_t len, int32_t c, int32_t k, int32_t stride, int32_t pad_left,
                           int32_t dilation, int32_t lout, int32_t backward, void* stream) {
    EG_REQUIRE(x && col && batch > 0 && len > 0 && c > 0 && k > 0 && stride > 0 && dilation > 0 && lout > 0, EG_ERR_BAD_ARG, "eg_im2col1d: bad argument");
    if (!backward)
        hipLaunchKernelGGL(im2col1d_kernel, grid1((size_t)batch * lout * k * c, 65536), dim3(256), 0, ST, x, col, batch, len, c, lout, k, stride,
                           pad_left, dilation);
    else
        hipLaunchKernelGGL(col2im1d_kernel, grid1((size_t)batch * len * c, 65536), dim3(256), 0, ST, x, col, batch, len, c, lout, k, stride, pad_left,
                           dilation);
    return eg_check_launch("im2col1d");
}

namespace {
constexpr long COL_MAX_PART = 2048;       // level-1 partials per reduction: 2048 workgroups x >= 64 KB keep 8 workgroups per CU streaming
// rows = rows per segment; nseg segments back to back; partials part[seg][nblk][2][c] with nseg * nblk <= COL_MAX_PART
int col_reduce(const float* a, const float* b, const float* mean, int64_t rows, int c, int mode, float* part, int* nblk_out, hipStream_t st,
               int nseg = 1, const float* mask = nullptr, const unsigned* mask_bits = nullptr) {
    const bool fast = (c >= 4) && (1024 % c == 0) && eg_aligned16(a) && (!b || eg_aligned16(b));
    long cap = COL_MAX_PART / nseg;
    if (cap < 1) cap = 1;
    long nblk = fast ? (rows * c + 16383) / 16384 : (rows + 127) / 128;         // fast path: >= 64 KB of input per block; generic: 128 rows x 64 columns
    if (nblk > cap) nblk = cap;
    if (nblk < 1) nblk = 1;
    const long rows_per = (rows + nblk - 1) / nblk;
    nblk = (rows + rows_per - 1) / rows_per;
    *nblk_out = (int)nblk;
    if (fast) {
        const dim3 grid((unsigned)nblk, nseg), block(256);
#define CPF(M, K) hipLaunchKernelGGL((col_partial_fast_kernel<M, K>), grid, block, 0, st, a, b, mean, part, (long)rows, c, rows_per, mask, mask_bits)
        if (mode == 4) {
            EG_REQUIRE(b && (mask || mask_bits), EG_ERR_BAD_ARG, "column reduction behind a ReLU mask: needs the second operand and a mask");
            if (mask_bits) CPF(4, 2); else CPF(4, 1);
        } else if (mode == 0 && !b) CPF(0, 0);
        else if (mode == 1 && b) CPF(1, 0);
        else if (mode == 2 && b) CPF(2, 0);
        else if (mode == 3 && !b) CPF(3, 0);
        else { eg_set_error("column reduction: mode %d with%s a second operand", mode, b ? "" : "out"); return EG_ERR_BAD_ARG; }
#undef CPF
        return eg_check_launch("col_partial_fast");
    }
    EG_REQUIRE(mode != 4, EG_ERR_UNSUPPORTED, "column reduction behind a ReLU mask: C=%d must divide 1024 and the operands be 16-byte aligned", c);
    EG_REQUIRE((mode == 0 && !b) || (mode == 1 && b) || (mode == 2 && b) || (mode == 3 && !b), EG_ERR_BAD_ARG, "column reduction: mode %d with%s a second operand", mode,
               b ? "" : "out");
    for (int sgi = 0; sgi < nseg; ++sgi) {       // generic channel counts: one launch per segment
        const size_t off = (size_t)sgi * rows * c;
        const dim3 grid(eg_cdiv(c, 64), (unsigned)nblk), block(256);
        float* po = part + (size_t)sgi * nblk * 2 * c;
#define CPG(M) hipLaunchKernelGGL((col_partial_kernel<M>), grid, block, 0, st, a + off, b ? b + off : nullptr, mean, po, (long)rows, c, rows_per)
        if (mode == 0) CPG(0); else if (mode == 1) CPG(1); else if (mode == 2) CPG(2); else CPG(3);
#undef CPG
        if (int rc = eg_check_launch("col_partial")) return rc;
    }
    return EG_OK;
}
// column sums straight to their final place: one launch for small inputs, partials + finalize otherwise
bool col_direct_ok(const float* a, const float* b, const float* o0, const float* o1, int64_t rows, int c) {
    // rows <= 1024: a workgroup owns 64 columns for ALL rows, so the launch has only C / 64 workgroups -- right for a [544, 512] bias / LayerNorm
    // gradient (8 workgroups x 139 KB), wrong for a long thin input (measured: 67 us for 8192 x 32 on ONE workgroup; and a 1,984-element fp32
    // chain per thread cost 1.6e-4 on a BatchNorm beta gradient over 31,744 rows).  Longer inputs take the 64 KB partials + double-precision fold.
    return (c & 3) == 0 && rows <= 1024 && eg_aligned16(a) && (!b || eg_aligned16(b)) && (!o0 || eg_aligned16(o0)) &&
           (!o1 || eg_aligned16(o1));
}
int col_sums(const float* a, const float* b, const float* mean, int64_t rows, int c, int mode, float* workspace, float* o0, float* o1, float scale,
             hipStream_t st) {
    if (col_direct_ok(a, b, o0, o1, rows, c) && (!mean || eg_aligned16(mean))) {
        hipLaunchKernelGGL(col_direct_kernel, dim3(eg_cdiv(c, 64)), dim3(256), 0, st, a, b, mean, o0, o1, (long)rows, c, mode, scale);
        return eg_check_launch("col_direct");
    }
    int nblk = 0;
    if (int rc = col_reduce(a, b, mean, rows, c, mode, workspace, &nblk, st)) return rc;
    hipLaunchKernelGGL(col_finalize_kernel, dim3(eg_cdiv(c, 4)), dim3(256), 0, st, workspace, nblk, c, o0, o1, scale);
    return eg_check_launch("col_finalize");
}
}  // namespace

// workspace for the column reductions below: partials [<= COL_MAX_PART][2][C] + one scratch column
extern "C" int64_t eg_colreduce_workspace_floats(int32_t c) { return (int64_t)(2 * COL_MAX_PART + 1) * c; }

namespace {
int launch_bn_apply(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta, float* y, int64_t rows, int c, hipStream_t st) {
    if ((c & 3) == 0 && eg_aligned16(x) && eg_aligned16(y) && eg_aligned16(mean) && eg_aligned16(rstd) && eg_aligned16(gamma) && eg_aligned16(beta)) {
        const size_t n4 = (size_t)rows * c / 4;
        hipLaunchKernelGGL(bn_apply4_kernel, grid1(n4), dim3(256), 0, st, reinterpret_cast<const f4*>(x), mean, rstd, gamma, beta, reinterpret_cast<f4*>(y), n4,
                           c / 4);
        return eg_check_launch("bn_apply4");
    }
    hipLaunchKernelGGL(bn_apply_kernel, grid1((size_t)rows * c), dim3(256), 0, st, x, mean, rstd, gamma, beta, y, (size_t)rows * c, c);
    return eg_check_launch("bn_apply");
}
}  // namespace

extern "C" int eg_bn_train_forward(const float* x, const float* gamma, const float* beta, float* y, float* save_mean, float* save_rstd,
                                   float* running_mean, float* running_var, int64_t rows, int32_t c, float momentum, float eps, float* workspace,
                                   void* stream) {
    EG_REQUIRE(x && gamma && beta && y && save_mean && save_rstd && workspace && rows > 0 && c > 0, EG_ERR_BAD_ARG, "eg_bn_train_forward: bad argument");
    int nblk = 0;
    if (int rc = col_reduce(x, nullptr, nullptr, rows, c, 0, workspace, &nblk, ST)) return rc;            // pass 1: mean
    hipLaunchKernelGGL(bn_mean_kernel, dim3(eg_cdiv(c, 4)), dim3(256), 0, ST, workspace, nblk, c, (long)rows, save_mean);
    if (int rc = eg_check_launch("bn_mean")) return rc;
    if (int rc = col_reduce(x, nullptr, save_mean, rows, c, 3, workspace, &nblk, ST)) return rc;           // pass 2: centred squares
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(eg_cdiv(c, 4)), dim3(256), 0, ST, workspace, nblk, c, (long)rows, eps, momentum, save_mean, save_rstd,
                       running_mean, running_var);
    if (int rc = eg_check_launch("bn_finalize")) return rc;
    return launch_bn_apply(x, save_mean, save_rstd, gamma, beta, y, rows, c, ST);
}
extern "C" int eg_bn_train_backward(const float* x, const float* dy, const float* gamma, const float* save_mean, const float* save_rstd, float* dx,
                                    float* dgamma, float* dbeta, int64_t rows, int32_t c, int32_t relu_mask, float* workspace, void* stream) {
    EG_REQUIRE(x && dy && gamma && save_mean && save_rstd && dx && dgamma && dbeta && workspace && rows > 0 && c > 0, EG_ERR_BAD_ARG,
               "eg_bn_train_backward: bad argument");
    float* sum_dyx = workspace + (size_t)2 * COL_MAX_PART * c;           // scratch column behind the partials
    // (sum dy, sum dy*(x - mean)): dbeta receives sum dy directly; sum dy*(x - mean) goes to the scratch column
    if (int rc = col_sums(dy, x, save_mean, rows, c, 2, workspace, dbeta, sum_dyx, 1.0f, ST)) return rc;
    if ((c & 3) == 0 && eg_aligned16(x) && eg_aligned16(dy) && eg_aligned16(dx) && eg_aligned16(save_mean) && eg_aligned16(save_rstd) && eg_aligned16(gamma) &&
        eg_aligned16(dbeta) && eg_aligned16(sum_dyx) && eg_aligned16(dgamma)) {
        const size_t n4 = (size_t)rows * c / 4;
        hipLaunchKernelGGL(bn_bwd_apply4_kernel, grid1(n4), dim3(256), 0, ST, reinterpret_cast<const f4*>(x), reinterpret_cast<const f4*>(dy), save_mean,
                           save_rstd, gamma, dbeta, sum_dyx, reinterpret_cast<f4*>(dx), dgamma, n4, c / 4, 1.0f / (float)rows, relu_mask);
        return eg_check_launch("bn_bwd_apply4");
    }
    hipLaunchKernelGGL(bn_bwd_apply_kernel, grid1((size_t)rows * c), dim3(256), 0, ST, x, dy, save_mean, save_rstd, gamma, dbeta, sum_dyx, dx, dgamma,
                       (size_t)rows * c, c, 1.0f / (float)rows, relu_mask);
    return eg_check_launch("bn_bwd_apply");
}
// BatchNorm (train mode) whose input came out of eg_conv3x3 with gap_partial [batch][tiles][c]: the mean (and the per-clip sums, for the
// SE pooling) are taken from those partials; one centred pass gives the variance.  y == nullptr: statistics only (the fused tail applies it).
extern "C" int eg_bn_train_forward_gap(const float* x, const float* gap_partial, int32_t tiles, int32_t batch, const float* gamma, const float* beta,
                                       float* y, float* save_mean, float* save_rstd, float* clip_sum, float* running_mean, float* running_var,
                                       int64_t rows, int32_t c, float momentum, float eps, float* workspace, void* stream) {
    EG_REQUIRE(x && gap_partial && save_mean && save_rstd && workspace && rows > 0 && c > 0 && tiles > 0 && batch > 0 && (!y || (gamma && beta)),
               EG_ERR_BAD_ARG, "eg_bn_train_forward_gap: bad argument");
    EG_REQUIRE(c <= 256 && 256 % c == 0, EG_ERR_UNSUPPORTED, "eg_bn_train_forward_gap: C=%d must divide 256", c);
    float* cs = clip_sum ? clip_sum : workspace + (size_t)2 * COL_MAX_PART * c - (size_t)batch * c;        // scratch behind the (later) partials: read before they are written
    EG_REQUIRE(clip_sum || batch <= 512, EG_ERR_UNSUPPORTED, "eg_bn_train_forward_gap: batch %d > 512 needs a clip_sum buffer", batch);
    hipLaunchKernelGGL(clip_sum_from_gap_kernel, dim3(batch), dim3(256), 0, ST, gap_partial, tiles, c, cs);
    if (int rc = eg_check_launch("clip_sum_from_gap")) return rc;
    hipLaunchKernelGGL(bn_mean_from_clips_kernel, dim3(eg_cdiv(c, 4)), dim3(256), 0, ST, cs, batch, c, (long)rows, save_mean);
    if (int rc = eg_check_launch("bn_mean_from_clips")) return rc;
    int nblk = 0;
    if (int rc = col_reduce(x, nullptr, save_mean, rows, c, 3, workspace, &nblk, ST)) return rc;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(eg_cdiv(c, 4)), dim3(256), 0, ST, workspace, nblk, c, (long)rows, eps, momentum, save_mean, save_rstd,
                       running_mean, running_var);
    if (int rc = eg_check_launch("bn_finalize")) return rc;
    if (!y) return EG_OK;
    return launch_bn_apply(x, save_mean, save_rstd, gamma, beta, y, rows, c, ST);
}

// The same operator fed by eg_conv3x3_sq's two partial arrays: statistics in two small launches (per-clip folds, then the merge), no pass over x.
// workspace >= eg_colreduce_workspace_floats(c) (holds the per-clip double sums: 4 * batch * c floats).
extern "C" int eg_bn_train_forward_sq(const float* x, const float* gap_partial, const float* gap_sq_partial, int32_t tiles, int32_t batch,
                                      const float* gamma, const float* beta, float* y, float* save_mean, float* save_rstd, float* clip_sum,
                                      float* running_mean, float* running_var, int64_t rows, int32_t c, float momentum, float eps, float* workspace,
                                      void* stream) {
    EG_REQUIRE(gap_partial && gap_sq_partial && save_mean && save_rstd && workspace && rows > 0 && c > 0 && tiles > 0 && batch > 0 &&
                   (!y || (x && gamma && beta)), EG_ERR_BAD_ARG, "eg_bn_train_forward_sq: bad argument");
    EG_REQUIRE(c <= 256 && 256 % c == 0, EG_ERR_UNSUPPORTED, "eg_bn_train_forward_sq: C=%d must divide 256", c);
    EG_REQUIRE((int64_t)4 * batch * c <= eg_colreduce_workspace_floats(c) && (reinterpret_cast<uintptr_t>(workspace) & 7) == 0, EG_ERR_WORKSPACE,
               "eg_bn_train_forward_sq: batch %d does not fit the reduction workspace", batch);
    double* clip_d = reinterpret_cast<double*>(workspace);
    hipLaunchKernelGGL(clip_sums_sq_kernel, dim3(batch), dim3(256), 0, ST, gap_partial, gap_sq_partial, tiles, c, clip_sum, clip_d);
    if (int rc = eg_check_launch("clip_sums_sq")) return rc;
    hipLaunchKernelGGL(bn_finalize_sq_kernel, dim3(eg_cdiv(c, 4)), dim3(256), 0, ST, clip_d, batch, c, (long)rows, eps, momentum, save_mean, save_rstd,
                       running_mean, running_var);
    if (int rc = eg_check_launch("bn_finalize_sq")) return rc;
    if (!y) return EG_OK;
    return launch_bn_apply(x, save_mean, save_rstd, gamma, beta, y, rows, c, ST);
}

// Train-mode BatchNorm statistics only, from the producing convolution's partials, plus the apply folded to one affine per channel
// (aff_scale = gamma * rstd, aff_shift = beta - mean * aff_scale) for the consumer that applies it while staging its operand (eg_conv3x3_sq_in_affine,
// eg_conv3x3_wgrad_mfma_oihw_in_affine): the normalised map is never written.  Running statistics updated as eg_bn_train_forward_sq does.
extern "C" int eg_bn_train_stats_sq(const float* gap_partial, const float* gap_sq_partial, int32_t tiles, int32_t batch, const float* gamma, const float* beta,
                                    float* save_mean, float* save_rstd, float* running_mean, float* running_var, float* aff_scale, float* aff_shift,
                                    int64_t rows, int32_t c, float momentum, float eps, float* workspace, void* stream) {
    EG_REQUIRE(gap_partial && gap_sq_partial && gamma && beta && save_mean && save_rstd && aff_scale && aff_shift && workspace && rows > 0 && batch > 0 && tiles > 0,
               EG_ERR_BAD_ARG, "eg_bn_train_stats_sq: bad argument");
    EG_REQUIRE(c > 0 && c <= 256 && 256 % c == 0, EG_ERR_UNSUPPORTED, "eg_bn_train_stats_sq: C=%d (a divisor of 256)", c);
    EG_REQUIRE((int64_t)4 * batch * c <= eg_colreduce_workspace_floats(c) && (reinterpret_cast<uintptr_t>(workspace) & 7) == 0, EG_ERR_WORKSPACE,
               "eg_bn_train_stats_sq: batch %d does not fit the reduction workspace", batch);
    double* clip_d = reinterpret_cast<double*>(workspace);
    hipLaunchKernelGGL(clip_sums_sq_kernel, dim3(batch), dim3(256), 0, ST, gap_partial, gap_sq_partial, tiles, c, (float*)nullptr, clip_d);
    if (int rc = eg_check_launch("clip_sums_sq")) return rc;
    hipLaunchKernelGGL(bn_finalize_sq_kernel, dim3(eg_cdiv(c, 4)), dim3(256), 0, ST, clip_d, batch, c, (long)rows, eps, momentum, save_mean, save_rstd,
                       running_mean, running_var, gamma, beta, aff_scale, aff_shift);
    return eg_check_launch("bn_finalize_sq");
}

#define SE_TAIL_SHAPE(who) EG_REQUIRE(batch > 0 && hw > 0 && c >= 8 && c <= 256 && c % 8 == 0, EG_ERR_UNSUPPORTED, who ": C=%d (multiple of 8, <= 256)", c)

extern "C" int eg_se_gate_train_forward(const float* clip_sum, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* w1,
                                        const float* b1, const float* w2, const float* b2, float* pooled, float* h, float* gate, int32_t batch, int32_t hw,
                                        int32_t c, void* stream) {
    EG_REQUIRE(clip_sum && mean && rstd && gamma && beta && w1 && b1 && w2 && b2 && pooled && h && gate, EG_ERR_BAD_ARG, "eg_se_gate_train_forward: null pointer");
    SE_TAIL_SHAPE("eg_se_gate_train_forward");
    hipLaunchKernelGGL(se_gate_train_kernel, dim3(batch), dim3(256), 0, ST, clip_sum, mean, rstd, gamma, beta, w1, b1, w2, b2, pooled, h, gate, c, 1.0f / (float)hw);
    return eg_check_launch("se_gate_train");
}

extern "C" int eg_se_tail_forward(const float* c2, const float* res, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                  const float* gate, float* out, uint32_t* relu_bits, int32_t batch, int32_t hw, int32_t c, void* stream) {
    EG_REQUIRE(c2 && res && mean && rstd && gamma && beta && gate && out, EG_ERR_BAD_ARG, "eg_se_tail_forward: null pointer");
    SE_TAIL_SHAPE("eg_se_tail_forward");
    EG_REQUIRE(eg_aligned16(c2) && eg_aligned16(res) && eg_aligned16(out) && eg_aligned16(gate), EG_ERR_ALIGN, "eg_se_tail_forward: 16-byte aligned maps");
    const size_t n4 = (size_t)batch * hw * c / 4;
    EG_REQUIRE(!relu_bits || (n4 & 7) == 0, EG_ERR_UNSUPPORTED, "eg_se_tail_forward: the ReLU bit mask needs batch * hw * c to be a multiple of 32");
    hipLaunchKernelGGL(se_tail_fwd_kernel, grid1(n4), dim3(256), 0, ST, reinterpret_cast<const f4*>(c2), reinterpret_cast<const f4*>(res), mean, rstd, gamma,
                       beta, gate, reinterpret_cast<f4*>(out), n4, hw * (c / 4), c / 4, relu_bits);
    return eg_check_launch("se_tail_fwd");
}

// s1[b,c] = sum_hw dout [out > 0];  s2raw[b,c] = sum_hw dout [out > 0] (c2 - mean[c]).   workspace >= eg_colreduce_workspace_floats(c); batch <= 512.
extern "C" int eg_se_tail_backward_reduce(const float* dout, const float* out, const uint32_t* relu_bits, const float* c2, const float* mean, float* s1,
                                          float* s2raw, int32_t batch, int32_t hw, int32_t c, float* workspace, void* stream) {
    EG_REQUIRE(dout && (out || relu_bits) && c2 && mean && s1 && s2raw && workspace && batch <= 512, EG_ERR_BAD_ARG, "eg_se_tail_backward_reduce: bad argument");
    SE_TAIL_SHAPE("eg_se_tail_backward_reduce");
    int nblk = 0;
    if (int rc = col_reduce(dout, c2, mean, hw, c, 4, workspace, &nblk, ST, batch, relu_bits ? nullptr : out, relu_bits)) return rc;
    hipLaunchKernelGGL(col_finalize_kernel, dim3(eg_cdiv(c, 4), batch), dim3(256), 0, ST, workspace, nblk, c, s1, s2raw, 1.0f);
    return eg_check_launch("se_tail_bwd_reduce");
}

extern "C" int eg_se_gate_train_backward(const float* s1, const float* s2raw, const float* clip_sum, const float* mean, const float* rstd,
                                         const float* gamma, const float* beta, const float* gate, const float* h, const float* w1, const float* w2,
                                         float* dz2, float* dz1, float* dgap_hw, float* u1, float* u2, int32_t batch, int32_t hw, int32_t c, void* stream) {
    EG_REQUIRE(s1 && s2raw && clip_sum && mean && rstd && gamma && beta && gate && h && w1 && w2 && dz2 && dz1 && dgap_hw && u1 && u2, EG_ERR_BAD_ARG,
               "eg_se_gate_train_backward: null pointer");
    SE_TAIL_SHAPE("eg_se_gate_train_backward");
    hipLaunchKernelGGL(se_gate_train_bwd_kernel, dim3(batch), dim3(256), 0, ST, s1, s2raw, clip_sum, mean, rstd, gamma, beta, gate, h, w1, w2, dz2, dz1,
                       dgap_hw, u1, u2, c, (float)hw);
    return eg_check_launch("se_gate_train_bwd");
}

extern "C" int eg_se_tail_backward_finish(const float* u1, const float* u2, const float* dz2, const float* dz1, const float* h, const float* pooled,
                                          float* dgamma, float* dbeta, float* m1, float* m2, float* dw1, float* db1, float* dw2, float* db2,
                                          int32_t batch, int32_t hw, int32_t c, void* stream) {
    EG_REQUIRE(u1 && u2 && dz2 && dz1 && h && pooled && dgamma && dbeta && m1 && m2 && dw1 && db1 && dw2 && db2, EG_ERR_BAD_ARG,
               "eg_se_tail_backward_finish: null pointer");
    SE_TAIL_SHAPE("eg_se_tail_backward_finish");
    const float inv_rows = 1.0f / ((float)batch * (float)hw);
#define FINISH(CH) hipLaunchKernelGGL(se_tail_bwd_finish_kernel<CH>, dim3(c), dim3(64), 0, ST, u1, u2, dz2, dz1, h, pooled, dgamma, dbeta, m1, m2, dw1, db1, \
                                      dw2, db2, batch, c, inv_rows)
    switch (c / 8) {
        case 4: FINISH(4); break;
        case 8: FINISH(8); break;
        case 16: FINISH(16); break;
        case 32: FINISH(32); break;
        default: eg_set_error("eg_se_tail_backward_finish: C=%d (32, 64, 128 or 256)", c); return EG_ERR_UNSUPPORTED;
    }
#undef FINISH
    return eg_check_launch("se_tail_bwd_finish");
}

extern "C" int eg_se_tail_backward_apply(const float* dout, const float* out, const uint32_t* relu_bits, const float* c2, const float* mean,
                                         const float* rstd, const float* gamma, const float* gate, const float* dgap_hw, const float* m1, const float* m2,
                                         float* dc2, float* dres, int32_t batch, int32_t hw, int32_t c, void* stream) {
    EG_REQUIRE(dout && (out || relu_bits) && c2 && mean && rstd && gamma && gate && dgap_hw && m1 && m2 && dc2, EG_ERR_BAD_ARG,
               "eg_se_tail_backward_apply: null pointer");
    EG_REQUIRE(dres || relu_bits, EG_ERR_BAD_ARG, "eg_se_tail_backward_apply: the residual's gradient may be left out only with the ReLU bit mask");
    SE_TAIL_SHAPE("eg_se_tail_backward_apply");
    EG_REQUIRE(eg_aligned16(dout) && (!out || eg_aligned16(out)) && eg_aligned16(c2) && eg_aligned16(dc2) && (!dres || eg_aligned16(dres)), EG_ERR_ALIGN,
               "eg_se_tail_backward_apply: 16-byte aligned maps");
    const size_t n4 = (size_t)batch * hw * c / 4;
    hipLaunchKernelGGL(se_tail_bwd_apply_kernel, grid1(n4), dim3(256), 0, ST, reinterpret_cast<const f4*>(dout), reinterpret_cast<const f4*>(out),
                       reinterpret_cast<const f4*>(c2), mean, rstd, gamma, gate, dgap_hw, m1, m2, reinterpret_cast<f4*>(dc2), reinterpret_cast<f4*>(dres), n4,
                       hw * (c / 4), c / 4, relu_bits);
    return eg_check_launch("se_tail_bwd_apply");
}

// o0[c] = sum_r a[r][c];  o1[c] = sum_r a[r][c]*b[r][c] (b optional: then o1 = sum a^2)
extern "C" int eg_colsum(const float* a, const float* b, float* o0, float* o1, int64_t rows, int32_t c, float* workspace, void* stream) {
    EG_REQUIRE(a && (o0 || o1) && workspace && rows > 0 && c > 0, EG_ERR_BAD_ARG, "eg_colsum: bad argument");
    return col_sums(a, b, nullptr, rows, c, b ? 1 : 0, workspace, o0, o1, 1.0f, ST);
}

extern "C" int eg_elementwise(const float* a, const float* b, float* y, int64_t n, int32_t op, float s, void* stream) {
    EG_REQUIRE(a && y && n > 0 && op >= 0 && op <= 11, EG_ERR_BAD_ARG, "eg_elementwise: bad argument");
    EG_REQUIRE(b || op == 0 || op == 2 || op == 5 || op == 6 || op == 10, EG_ERR_BAD_ARG, "eg_elementwise: op %d needs a second operand", op);
    hipLaunchKernelGGL(ew_kernel, grid1((size_t)n), dim3(256), 0, ST, a, b, y, (size_t)n, op, s);
    return eg_check_launch("elementwise");
}

// per-clip pooled sums through the two-level column reduction (clips = segments): workspace >= eg_colreduce_workspace_floats(c)
extern "C" int eg_pack_conv3x3_device(const float* w_oihw, int32_t cout, int32_t cin, int32_t flip_transpose, float* image, void* stream) {
    EG_REQUIRE(w_oihw && image && cout > 0 && cin > 0, EG_ERR_BAD_ARG, "eg_pack_conv3x3_device: null pointer or empty shape");
    const int ci = flip_transpose ? cout : cin, co = flip_transpose ? cin : cout;
    EG_REQUIRE(ci % 8 == 0 && eg_aligned16(image), EG_ERR_UNSUPPORTED, "eg_pack_conv3x3_device: input channels %d (need a multiple of 8), 16-byte aligned image", ci);
    const int total = 9 * (ci / 8) * (int)eg_round_up(co, 16);
    hipLaunchKernelGGL(pack_conv3x3_kernel, dim3((total + 255) / 256), dim3(256), 0, ST, w_oihw, cout, cin, flip_transpose, image);
    return eg_check_launch("pack_conv3x3");
}

extern "C" int64_t eg_linear_packed_floats(int32_t n, int32_t k) {
    return 2 * (int64_t)eg_round_up(n, 64) * (int64_t)eg_round_up(k, 64);      // fp32 image + (hi, lo) bf16 images (= the same bytes again)
}
extern "C" int eg_pack_linear_device(const float* w, int32_t ld, int32_t n, int32_t k, int32_t transpose, float* image, void* stream) {
    EG_REQUIRE(w && image && n > 0 && k > 0 && ld >= (transpose ? n : k), EG_ERR_BAD_ARG, "eg_pack_linear_device: bad argument");
    EG_REQUIRE(eg_aligned16(image), EG_ERR_ALIGN, "eg_pack_linear_device: image must be 16-byte aligned");
    const int total = (int)(eg_round_up(n, 64) * (eg_round_up(k, 64) / 8));
    hipLaunchKernelGGL(pack_linear_kernel, dim3((total + 255) / 256), dim3(256), 0, ST, w, ld, n, k, transpose, image);
    return eg_check_launch("pack_linear");
}

// Table-driven packing (see pack_table_kernel).  `table` is device memory holding `count` 40-byte entries
//   { const float* src; float* image; int32 kind, a, b, c, flag, first_block; }   kind 0: Linear a = n, b = k, c = ld, flag = transpose;  kind 1: conv3x3 a = cout, b = cin, flag = flip
// with first_block the running sum of eg_pack_table_blocks(...) of the entries before it; total_blocks = that sum over all entries.
extern "C" int32_t eg_pack_table_blocks(int32_t kind, int32_t a, int32_t b, int32_t flag) {
    if (kind == 0) return (int32_t)((eg_round_up(a, 64) * (eg_round_up(b, 64) / 8) + 255) / 256);
    const int ci = (flag & 1) ? a : b, co = (flag & 1) ? b : a;
    return (int32_t)((9 * (ci / 8) * (int)eg_round_up(co, 16) + 255) / 256);
}
extern "C" int eg_pack_table(const void* table, int32_t count, int32_t total_blocks, void* stream) {
    EG_REQUIRE(table && count > 0 && total_blocks > 0, EG_ERR_BAD_ARG, "eg_pack_table: bad argument");
    static_assert(sizeof(PackEntry) == 40, "PackEntry layout is part of the ABI");
    hipLaunchKernelGGL(pack_table_kernel, dim3(total_blocks), dim3(256), 0, ST, reinterpret_cast<const PackEntry*>(table), count);
    return eg_check_launch("pack_table");
}

extern "C" int eg_dropout(const float* x, float* y, int64_t n, float p, uint32_t seed, uint64_t offset, void* stream) {
    return eg_dropout_dev(x, y, n, p, seed, offset, nullptr, stream);
}
extern "C" int eg_dropout_dev(const float* x, float* y, int64_t n, float p, uint32_t seed, uint64_t offset, const int32_t* epoch_dev, void* stream) {
    EG_REQUIRE(x && y && n > 0 && p >= 0.f && p < 1.f, EG_ERR_BAD_ARG, "eg_dropout: bad argument (p=%f)", (double)p);
    hipLaunchKernelGGL(dropout_kernel, grid1((size_t)n), dim3(256), 0, ST, x, y, (size_t)n, p, 1.0f / (1.0f - p), seed, (unsigned long long)offset,
                       epoch_dev);
    return eg_check_launch("dropout");
}

extern "C" int eg_seg_mean(const float* x, float* out, int32_t batch, int32_t hw, int32_t c, float scale, float* workspace, void* stream) {
    EG_REQUIRE(x && out && batch > 0 && hw > 0 && c > 0, EG_ERR_BAD_ARG, "eg_seg_mean: bad argument");
    if (!workspace || batch > 512) {
        hipLaunchKernelGGL(seg_mean_kernel, dim3(eg_cdiv(c, 64), batch), dim3(256), 0, ST, x, out, hw, c, scale);
        return eg_check_launch("seg_mean");
    }
    int nblk = 0;
    if (int rc = col_reduce(x, nullptr, nullptr, hw, c, 0, workspace, &nblk, ST, batch)) return rc;
    hipLaunchKernelGGL(col_finalize_kernel, dim3(eg_cdiv(c, 4), batch), dim3(256), 0, ST, workspace, nblk, c, out, (float*)nullptr, scale);
    return eg_check_launch("seg_mean_finalize");
}
extern "C" int eg_seg_dot(const float* dy, const float* x, float* out, int32_t batch, int32_t hw, int32_t c, float* workspace, void* stream) {
    EG_REQUIRE(dy && x && out && batch > 0 && hw > 0 && c > 0, EG_ERR_BAD_ARG, "eg_seg_dot: bad argument");
    if (!workspace || batch > 512) {
        hipLaunchKernelGGL(seg_dot_kernel, dim3(eg_cdiv(c, 64), batch), dim3(256), 0, ST, dy, x, out, hw, c);
        return eg_check_launch("seg_dot");
    }
    int nblk = 0;
    if (int rc = col_reduce(dy, x, nullptr, hw, c, 1, workspace, &nblk, ST, batch)) return rc;
    hipLaunchKernelGGL(col_finalize_kernel, dim3(eg_cdiv(c, 4), batch), dim3(256), 0, ST, workspace, nblk, c, (float*)nullptr, out, 1.0f);
    return eg_check_launch("seg_dot_finalize");
}
extern "C" int eg_se_scale(const float* a, const float* gate, const float* add, float* y, int32_t batch, int32_t hw, int32_t c, void* stream) {
    EG_REQUIRE(a && gate && y && batch > 0 && hw > 0 && c > 0, EG_ERR_BAD_ARG, "eg_se_scale: bad argument");
    hipLaunchKernelGGL(se_scale_kernel, grid1((size_t)batch * hw * c), dim3(256), 0, ST, a, gate, add, y, (size_t)batch * hw * c, hw, c);
    return eg_check_launch("se_scale");
}

extern "C" int eg_layernorm_backward(const float* x, const float* dy, const float* gamma, float* dx, float* xhat, int32_t rows, int32_t d, float eps,
                                     void* stream) {
    EG_REQUIRE(x && dy && gamma && dx && xhat && rows > 0 && d > 0, EG_ERR_BAD_ARG, "eg_layernorm_backward: bad argument");
    hipLaunchKernelGGL(ln_bwd_kernel, dim3(eg_cdiv(rows, 4)), dim3(256), 0, ST, x, dy, gamma, dx, xhat, rows, d, eps);
    return eg_check_launch("layernorm_backward");
}

namespace {
inline int ln_ex_rows_per_wave(int rows) { const int rw = eg_cdiv(rows, 4 * 256); return rw < 1 ? 1 : (rw > 8 ? 8 : rw); }
}  // namespace
extern "C" int64_t eg_layernorm_backward_ex_workspace_floats(int32_t rows, int32_t d) {
    return (int64_t)eg_cdiv(rows, 4 * ln_ex_rows_per_wave(rows)) * 2 * d;
}
extern "C" int eg_layernorm_backward_ex(const float* x, const float* dy, const float* gamma, float* dx, float* dx_dropped, float* dgamma, float* dbeta,
                                        int32_t rows, int32_t d, float eps, float drop_p, uint32_t drop_seed, uint64_t drop_offset,
                                        const int32_t* epoch_dev, float* workspace, void* branch_images, void* stream) {
    EG_REQUIRE(x && dy && gamma && dx && dgamma && dbeta && workspace && rows > 0, EG_ERR_BAD_ARG, "eg_layernorm_backward_ex: bad argument");
    EG_REQUIRE(d > 0 && (d & 63) == 0 && d <= 1024, EG_ERR_UNSUPPORTED, "eg_layernorm_backward_ex: d=%d (multiple of 64, <= 1024)", d);
    EG_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f) == (dx_dropped == nullptr), EG_ERR_BAD_ARG,
               "eg_layernorm_backward_ex: dx_dropped goes with drop_p > 0 (p=%f)", (double)drop_p);
    EgDropout dr;
    if (drop_p > 0.f) {
        dr.thr = (unsigned int)((double)drop_p * 4294967296.0);
        dr.inv_keep = 1.0f / (1.0f - drop_p);
        dr.seed = drop_seed; dr.offset = drop_offset; dr.epoch = epoch_dev;
    }
    const int rw = ln_ex_rows_per_wave(rows), nblk = eg_cdiv(rows, 4 * rw);
    const size_t smem = (size_t)8 * d * sizeof(float);
    unsigned short* img = reinterpret_cast<unsigned short*>(branch_images);
    if (d <= 512) hipLaunchKernelGGL((ln_bwd_ex_kernel<2>), dim3(nblk), dim3(256), smem, ST, x, dy, gamma, dx, dx_dropped, workspace, rows, d, rw, eps, dr, img);
    else hipLaunchKernelGGL((ln_bwd_ex_kernel<4>), dim3(nblk), dim3(256), smem, ST, x, dy, gamma, dx, dx_dropped, workspace, rows, d, rw, eps, dr, img);
    if (int rc = eg_check_launch("layernorm_backward_ex")) return rc;
    hipLaunchKernelGGL(col_finalize_kernel, dim3(eg_cdiv(d, 4)), dim3(256), 0, ST, workspace, nblk, d, dbeta, dgamma, 1.0f);
    return eg_check_launch("layernorm_backward_ex_fold");
}

extern "C" int eg_attention_backward(const float* q, int32_t ldq, const float* k, int32_t ldk, const float* v, int32_t ldv, const float* attn,
                                     const float* dout, int32_t ldo, float* dq, int32_t lddq, float* dk, int32_t lddk, float* dv, int32_t lddv,
                                     int32_t batch, int32_t heads, int32_t lq, int32_t lk, int32_t dk_dim, void* stream) {
    EG_REQUIRE(q && k && v && attn && dout && dq && dk && dv && batch > 0 && heads > 0 && lq > 0 && lk > 0, EG_ERR_BAD_ARG,
               "eg_attention_backward: bad argument");
    EG_REQUIRE(dk_dim == 64, EG_ERR_UNSUPPORTED, "eg_attention_backward: d_k=%d", dk_dim);
    const size_t smem = sizeof(float) * ((size_t)2 * lq * 65 + (size_t)2 * lk * 65 + (size_t)2 * lq * (lk + 1));
    EG_REQUIRE(smem <= 160 * 1024, EG_ERR_UNSUPPORTED, "eg_attention_backward: Lq=%d Lk=%d needs %zu B of LDS", lq, lk, smem);
    if (int rc = eg_ensure_dynamic_lds(reinterpret_cast<const void*>(attention_bwd_kernel), smem, "eg_attention_backward")) return rc;
    hipLaunchKernelGGL(attention_bwd_kernel, dim3(heads, batch), dim3(256), smem, ST, q, ldq, k, ldk, v, ldv, attn, dout, ldo, dq, lddq, dk, lddk, dv,
                       lddv, heads, lq, lk, 1.0f / sqrtf((float)dk_dim));
    return eg_check_launch("attention_backward");
}

// loss[0] = scale * mean smooth_l1(pred, target; beta);  dpred (optional) = d loss / d pred.  workspace >= 1024 floats.
extern "C" int eg_smooth_l1(const float* pred, const float* target, float* loss, float* dpred, int64_t n, float beta, float scale, float* workspace,
                            void* stream) {
    EG_REQUIRE(pred && target && loss && workspace && n > 0 && beta > 0.f, EG_ERR_BAD_ARG, "eg_smooth_l1: bad argument");
    const int nb = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(smooth_l1_kernel, dim3(nb), dim3(256), 0, ST, pred, target, dpred, workspace, (size_t)n, beta, scale / (float)n);
    if (int rc = eg_check_launch("smooth_l1")) return rc;
    hipLaunchKernelGGL(sum_small_kernel, dim3(1), dim3(64), 0, ST, workspace, nb, (double)scale / (double)n, loss);
    return eg_check_launch("smooth_l1_sum");
}
// loss[0] = scale * mean_b L_b;  L_b = CE (gamma < 0) or focal alpha[y] (1-pt)^gamma CE.  workspace >= B floats.
extern "C" int eg_cross_entropy(const float* logits, const int64_t* labels, const float* alpha, float gamma, float scale, float* loss, float* dlogits,
                                int32_t batch, int32_t classes, float* workspace, void* stream) {
    EG_REQUIRE(logits && labels && loss && workspace && batch > 0 && classes > 0, EG_ERR_BAD_ARG, "eg_cross_entropy: bad argument");
    hipLaunchKernelGGL(ce_kernel, dim3(batch), dim3(64), 0, ST, logits, labels, alpha, gamma, scale, batch, classes, workspace, dlogits);
    if (int rc = eg_check_launch("cross_entropy")) return rc;
    hipLaunchKernelGGL(sum_small_kernel, dim3(1), dim3(64), 0, ST, workspace, batch, (double)scale / (double)batch, loss);
    return eg_check_launch("cross_entropy_sum");
}

extern "C" int eg_kld(const float* mu, const float* logvar, float* loss, float* dmu, float* dlogvar, int32_t n, int32_t d, float scale, void* stream) {
    EG_REQUIRE(mu && logvar && loss && n > 0 && d > 0 && ((dmu == nullptr) == (dlogvar == nullptr)), EG_ERR_BAD_ARG, "eg_kld: bad argument");
    hipLaunchKernelGGL(kld_kernel, dim3(1), dim3(256), 0, ST, mu, logvar, dmu, dlogvar, loss, n, d, scale);
    return eg_check_launch("kld");
}

extern "C" int eg_sp_gate_forward(const float* mem, const float* pred, float* out, float* gate, int32_t batch, int32_t frames, int32_t dim, int32_t chunk,
                                  void* stream) {
    EG_REQUIRE(mem && pred && out && gate && batch > 0 && frames > 0 && dim > 0 && chunk >= 0 && chunk <= frames, EG_ERR_BAD_ARG, "eg_sp_gate_forward: bad argument");
    hipLaunchKernelGGL(sp_gate_fwd_kernel, dim3(batch * frames), dim3(64), 0, ST, mem, pred, out, gate, frames, dim, chunk);
    return eg_check_launch("sp_gate_forward");
}
extern "C" int eg_sp_gate_backward(const float* mem, const float* pred, const float* gate, const float* dout, float* dpred, float* dmem, int32_t batch,
                                   int32_t frames, int32_t dim, int32_t chunk, void* stream) {
    EG_REQUIRE(mem && pred && gate && dout && dpred && dmem && batch > 0 && frames > 0 && dim > 0 && chunk >= 0 && chunk <= frames, EG_ERR_BAD_ARG,
               "eg_sp_gate_backward: bad argument");
    hipLaunchKernelGGL(sp_gate_bwd_kernel, dim3(batch), dim3(64), 0, ST, mem, pred, gate, dout, dpred, dmem, frames, dim, chunk);
    return eg_check_launch("sp_gate_backward");
}
extern "C" int eg_tm_scale_forward(const float* score, const float* pred, float* out, float* w, int32_t batch, int32_t frames, int32_t dim, int32_t chunk,
                                   void* stream) {
    EG_REQUIRE(score && pred && out && w && batch > 0 && frames > 0 && dim > 0 && chunk > 0 && chunk <= frames && chunk <= 64, EG_ERR_BAD_ARG,
               "eg_tm_scale_forward: bad argument");
    hipLaunchKernelGGL(tm_scale_fwd_kernel, dim3(batch * frames), dim3(64), 0, ST, score, pred, out, w, frames, dim, chunk);
    return eg_check_launch("tm_scale_forward");
}
extern "C" int eg_tm_scale_backward(const float* w, const float* pred, const float* dout, float* dpred, float* dscore, int32_t batch, int32_t frames,
                                    int32_t dim, int32_t chunk, void* stream) {
    EG_REQUIRE(w && pred && dout && dpred && dscore && batch > 0 && frames > 0 && dim > 0 && chunk > 0 && chunk <= frames && chunk <= 64, EG_ERR_BAD_ARG,
               "eg_tm_scale_backward: bad argument");
    hipLaunchKernelGGL(tm_scale_bwd_kernel, dim3(batch), dim3(64), 0, ST, w, pred, dout, dpred, dscore, frames, dim, chunk);
    return eg_check_launch("tm_scale_backward");
}

extern "C" int eg_f32_to_bf16(const float* x, uint16_t* y, int64_t n, void* stream) {
    EG_REQUIRE(x && y && n > 0, EG_ERR_BAD_ARG, "eg_f32_to_bf16: bad argument");
    hipLaunchKernelGGL(f32_to_bf16_kernel, grid1((size_t)n), dim3(256), 0, ST, x, y, (size_t)n);
    return eg_check_launch("f32_to_bf16");
}
extern "C" int eg_bf16_to_f32(const uint16_t* x, float* y, int64_t n, float scale, void* stream) {
    EG_REQUIRE(x && y && n > 0, EG_ERR_BAD_ARG, "eg_bf16_to_f32: bad argument");
    hipLaunchKernelGGL(bf16_to_f32_kernel, grid1((size_t)n), dim3(256), 0, ST, x, y, (size_t)n, scale);
    return eg_check_launch("bf16_to_f32");
}

extern "C" int eg_counter_add(int32_t* counter, int32_t delta, void* stream) {
    EG_REQUIRE(counter, EG_ERR_BAD_ARG, "eg_counter_add: null pointer");
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, ST, counter, delta);
    return eg_check_launch("counter_add");
}
extern "C" int eg_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                                float eps, float weight_decay, const int32_t* step_dev, void* stream) {
    EG_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && step_dev, EG_ERR_BAD_ARG, "eg_adam_step_dev: bad argument");
    launch_adam<true>(param, grad, exp_avg, exp_avg_sq, (size_t)n, lr, beta1, beta2, eps, weight_decay, 0.f, 0.f, step_dev, ST);
    return eg_check_launch("adam_step_dev");
}

extern "C" int eg_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2, float eps,
                            float weight_decay, int32_t step, void* stream) {
    EG_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && step > 0, EG_ERR_BAD_ARG, "eg_adam_step: bad argument");
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    launch_adam<false>(param, grad, exp_avg, exp_avg_sq, (size_t)n, lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), nullptr, ST);
    return eg_check_launch("adam_step");
}
