// nn.Linear family on MFMA:  Y[M,N] = epi(X[M,K] . W[N,K]^T)
// Reference call sites: Full_model/SubLayers.py:19-22,67-68 (Q/K/V/O, FFN), Models_spatial_memory.py:105-107,
// 488-536 (audio fc1/fc2, emotion/semantic/fusion projections, classifier header, post_projector), tcn.py:18-24.
//
// Workgroup = 4 waves = 64(M) x 64(N) output tile, K-step 32.  Operands are swapped so that a lane owns 4
// consecutive n of one row m (one 16-byte store into row-major Y):
//   D[n][m] = sum_k W[n][k] X[m][k];  A operand = W rows, B operand = X rows.
// LDS images are k-quad (fp32) / k-octet (bf16) planar: [k/4][row][4] -- 64 rows per plane = 0 mod 16 slots, so the
// ds_read_b128 of 16 consecutive rows is bank-conflict free.
#include "common.h"

namespace {

struct GemmArgs {
    const float* x; const float* w; const float* bias; const float* res1; const float* res2; float* y;
    const unsigned short* whi; const unsigned short* wlo;     // bf16 images [Nrows][ldw] (split-bf16 modes)
    int lda, ldw, ldr, ldc, M, N, K, relu, a_shift, a_seq, k_per_split;
    float* partial;
};

__device__ __forceinline__ f4 load_x_quad(const GemmArgs& a, int m, int k, int kend) {
    f4 v = (f4){0.f, 0.f, 0.f, 0.f};
    if (m < a.M && k < kend) {
        int src = m;
        bool ok = true;
        if (a.a_shift) {
            ok = (m % a.a_seq) >= a.a_shift;
            src = m - a.a_shift;
        }
        if (ok) v = *reinterpret_cast<const f4*>(a.x + (size_t)src * a.lda + k);
    }
    return v;
}


// v = acc + bias + res1; relu; (+res2, relu); lane owns 4 consecutive n of row m (+16 per t), n += 16 per tile
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& a, f4 (&acc)[2][2], int mbase, int nbase) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int m = mbase + t * 16;
        if (m >= a.M) continue;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int nn = nbase + n * 16;
            if (nn >= a.N) continue;
            f4 v = acc[t][n];
            if (a.partial) {            // split-K: raw partial sums [split][M][N]
                float* p = a.partial + ((size_t)blockIdx.z * a.M + m) * a.N + nn;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (nn + r < a.N) p[r] = v[r];
                continue;
            }
            const bool full = (nn + 3 < a.N);
            if (a.bias) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (nn + r < a.N) v[r] += a.bias[nn + r];
            }
            if (a.res1) {
                const float* rp = a.res1 + (size_t)m * a.ldr + nn;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (nn + r < a.N) v[r] += rp[r];
            }
            if (a.relu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            if (a.res2) {
                const float* rp = a.res2 + (size_t)m * a.ldr + nn;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (nn + r < a.N) v[r] = fmaxf(v[r] + rp[r], 0.f);
            }
            float* yp = a.y + (size_t)m * a.ldc + nn;
            if (full && (a.ldc & 3) == 0) {
                *reinterpret_cast<f4*>(yp) = v;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (nn + r < a.N) yp[r] = v[r];
            }
        }
    }
}

template <int PREC>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs a) {
    __shared__ f4 lds[1024];            // fp32: Xs[8][64] | Ws[8][64];  bf16: Xh[4][64] Xl[4][64] Wh[4][64] Wl[4][64] (bf8 = 16 B)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    const int kbeg = blockIdx.z * a.k_per_split;
    const int kend = min(a.K, kbeg + a.k_per_split);

    f4 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[t][n] = (f4){0.f, 0.f, 0.f, 0.f};

    for (int k0 = kbeg; k0 < kend; k0 += 32) {
        if (k0 != kbeg) __syncthreads();
        if (PREC == EG_PREC_F32) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int idx = tid + i * 256;
                const int r = (idx >> 6) * 8 + (idx & 7), q = (idx >> 3) & 7;
                lds[q * 64 + r] = load_x_quad(a, m0 + r, k0 + q * 4, kend);
                f4 wv = (f4){0.f, 0.f, 0.f, 0.f};
                if (n0 + r < a.N && k0 + q * 4 < kend) wv = *reinterpret_cast<const f4*>(a.w + (size_t)(n0 + r) * a.ldw + k0 + q * 4);
                lds[512 + q * 64 + r] = wv;
            }
        } else {
            bf8* l8 = reinterpret_cast<bf8*>(lds);
            const int r = (tid >> 5) * 8 + (tid & 7), o = (tid >> 3) & 3;      // one (row, octet) per thread
            const f4 v0 = load_x_quad(a, m0 + r, k0 + o * 8, kend), v1 = load_x_quad(a, m0 + r, k0 + o * 8 + 4, kend);
            bf8 hi, lo;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float x = j < 4 ? v0[j & 3] : v1[j & 3];
                const unsigned short h = f32_to_bf16_rne(x);
                hi[j] = (short)h;
                lo[j] = (short)f32_to_bf16_rne(x - bf16_to_f32(h));
            }
            l8[o * 64 + r] = hi;
            l8[256 + o * 64 + r] = lo;
            bf8 wh = (bf8){0, 0, 0, 0, 0, 0, 0, 0}, wl = wh;
            if (n0 + r < a.N && k0 + o * 8 < kend) {       // packed images are zero padded to ldw (multiple of 8)
                const size_t off = (size_t)(n0 + r) * a.ldw + k0 + o * 8;
                wh = *reinterpret_cast<const bf8*>(a.whi + off);
                if (PREC == EG_PREC_BF16X3) wl = *reinterpret_cast<const bf8*>(a.wlo + off);
            }
            l8[512 + o * 64 + r] = wh;
            l8[768 + o * 64 + r] = wl;
        }
        __syncthreads();
        if (PREC == EG_PREC_F32) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                f4 wv[2], xv[2];
#pragma unroll
                for (int n = 0; n < 2; ++n) wv[n] = lds[512 + (g * 4 + kq) * 64 + wn + n * 16 + li];
#pragma unroll
                for (int t = 0; t < 2; ++t) xv[t] = lds[(g * 4 + kq) * 64 + wm + t * 16 + li];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int n = 0; n < 2; ++n)
                            acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[n][j], xv[t][j], acc[t][n], 0, 0, 0);
            }
        } else {
            const bf8* l8 = reinterpret_cast<const bf8*>(lds);
            bf8 xh[2], xl[2], wh[2], wl[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                xh[t] = l8[kq * 64 + wm + t * 16 + li];
                xl[t] = l8[256 + kq * 64 + wm + t * 16 + li];
            }
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                wh[n] = l8[512 + kq * 64 + wn + n * 16 + li];
                wl[n] = l8[768 + kq * 64 + wn + n * 16 + li];
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    if (PREC == EG_PREC_BF16X3) {
                        acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[n], xh[t], acc[t][n], 0, 0, 0);
                        acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[n], xl[t], acc[t][n], 0, 0, 0);
                    }
                    acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[n], xh[t], acc[t][n], 0, 0, 0);
                }
        }
    }

    gemm_epilogue(a, acc, m0 + wm + li, n0 + wn + kq * 4);
}

// ---- split-bf16 path: 64x64 tile, K-step 64, both operands double-buffered in LDS ------------------------------
//  * W: pre-split on the host into tile-planar bf16 images [n/64][k/8][64 rows][8] (hi image, lo image): the 8 octets of
//    one K-step are 8 contiguous 1-KiB pieces, copied by global_load_lds (no VGPR/VALU) into a 2-deep ring;
//  * X (fp32 activations): next step's rows are loaded to registers before this step's MFMAs, split to (hi, lo) bf16 with
//    v_cvt_pk_bf16_f32 and written to the other ring slot after them; one barrier per K-step.
//  LDS image of both operands: [k/8][row] bf8 => the 16 rows of an MFMA tile are 16 consecutive 16-B slots (conflict free).
template <int TERMS>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmArgs a) {
    constexpr int NIMG = (TERMS == 3) ? 2 : 1;
    constexpr int IMG = 8 * 64;                     // bf8 slots of one operand image of one step (8 octets x 64 rows)
    __shared__ bf8 lds[2 * 2 * NIMG * IMG];         // [buf][X|W][hi|lo][oct][row]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    const int kbeg = blockIdx.z * a.k_per_split;
    const int kend = min(a.K, kbeg + a.k_per_split);
    const int nsteps = (kend - kbeg + 63) / 64;
    const int KO = a.ldw >> 3;                      // octets per packed weight row

    // X staging role: row = tid>>2, octets 2*(tid&3), 2*(tid&3)+1 (64 contiguous bytes per lane, 256 B per 4 lanes)
    const int xr = tid >> 2, xo = (tid & 3) * 2;
    const int xm = m0 + xr;
    bool xok = xm < a.M;
    int xsrc = xm;
    if (a.a_shift) {
        xok = xok && (xm % a.a_seq) >= a.a_shift;
        xsrc = xm - a.a_shift;
    }
    const float* xrow = a.x + (size_t)(xok ? xsrc : 0) * a.lda;
    f4 xv[4];
    auto load_x = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + xo * 8 + j * 4;
            xv[j] = (xok && k < kend) ? *reinterpret_cast<const f4*>(xrow + k) : (f4){0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store_x = [&](int buf) {
        bf8* X = lds + buf * (2 * NIMG * IMG);
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            bf8 hi, lo;
            split_octet<TERMS == 3>(xv[2 * o], xv[2 * o + 1], hi, lo);
            X[(xo + o) * 64 + xr] = hi;
            if (TERMS == 3) X[IMG + (xo + o) * 64 + xr] = lo;
        }
    };
    auto issue_w = [&](int k0, int buf) {
        bf8* W = lds + buf * (2 * NIMG * IMG) + NIMG * IMG;
        const size_t gbase = ((size_t)blockIdx.y * KO + (k0 >> 3)) * 64;        // bf8 slots: [n/64][k/8][64]
#pragma unroll
        for (int img = 0; img < NIMG; ++img) {
            const bf8* src = reinterpret_cast<const bf8*>(img ? a.wlo : a.whi) + gbase;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int piece = p * 4 + wave;                                 // octet index 0..7
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 64 + lane),
                                                 (__attribute__((address_space(3))) void*)(W + img * IMG + piece * 64), 16, 0, 0);
            }
        }
    };

    f4 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[t][n] = (f4){0.f, 0.f, 0.f, 0.f};

    issue_w(kbeg, 0);
    load_x(kbeg);
    store_x(0);
    __syncthreads();
#pragma unroll 1
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        const bool more = s + 1 < nsteps;
        if (more) {
            issue_w(kbeg + (s + 1) * 64, buf ^ 1);
            load_x(kbeg + (s + 1) * 64);
        }
        const bf8* X = lds + buf * (2 * NIMG * IMG);
        const bf8* W = X + NIMG * IMG;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            bf8 xh[2], xl[2], wh[2], wl[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                xh[t] = X[(g * 4 + kq) * 64 + wm + t * 16 + li];
                if (TERMS == 3) xl[t] = X[IMG + (g * 4 + kq) * 64 + wm + t * 16 + li];
            }
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                wh[n] = W[(g * 4 + kq) * 64 + wn + n * 16 + li];
                if (TERMS == 3) wl[n] = W[IMG + (g * 4 + kq) * 64 + wn + n * 16 + li];
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    if (TERMS == 3) {
                        acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[n], xh[t], acc[t][n], 0, 0, 0);
                        acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[n], xl[t], acc[t][n], 0, 0, 0);
                    }
                    acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[n], xh[t], acc[t][n], 0, 0, 0);
                }
        }
        if (more) store_x(buf ^ 1);
        __syncthreads();
    }
    gemm_epilogue(a, acc, m0 + wm + li, n0 + wn + kq * 4);
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ bias,
                                                            float* __restrict__ y, int ldc, int M, int N, int splits, int relu) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M * N) return;
    const int m = i / N, n = i - m * N;
    float s = bias ? bias[n] : 0.f;
    for (int z = 0; z < splits; ++z) s += partial[(size_t)z * M * N + i];
    if (relu) s = fmaxf(s, 0.f);
    y[(size_t)m * ldc + n] = s;
}

int launch_gemm(GemmArgs& a, int splits, int precision, hipStream_t st) {
    dim3 grid(eg_cdiv(a.M, 64), eg_cdiv(a.N, 64), splits), block(256);
    if (precision == EG_PREC_F32) hipLaunchKernelGGL((gemm_kernel<EG_PREC_F32>), grid, block, 0, st, a);
    else if (precision == EG_PREC_BF16X3) hipLaunchKernelGGL((gemm_bf16_kernel<3>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((gemm_bf16_kernel<1>), grid, block, 0, st, a);
    return eg_check_launch("gemm");
}

int fill_common(GemmArgs& a, const float* x, int lda, const float* w, int ldw, int m, int n, int k, int precision,
                const char* who) {
    EG_REQUIRE(x && w && m > 0 && n > 0 && k > 0, EG_ERR_BAD_ARG, "%s: null pointer or empty shape", who);
    EG_REQUIRE(eg_aligned16(x) && eg_aligned16(w), EG_ERR_ALIGN, "%s: X and W must be 16-byte aligned", who);
    EG_REQUIRE((k & 3) == 0 && (lda & 3) == 0 && (ldw & 3) == 0, EG_ERR_ALIGN, "%s: K, lda, ldw must be multiples of 4 (K=%d lda=%d ldw=%d)", who, k, lda, ldw);
    EG_REQUIRE(precision >= 0 && precision <= 2, EG_ERR_BAD_ARG, "%s: precision %d", who, precision);
    a.x = x; a.w = w; a.lda = lda; a.ldw = ldw; a.M = m; a.N = n; a.K = k;
    a.whi = a.wlo = nullptr;
    if (precision != EG_PREC_F32) {
        // packed weight = fp32 image [rows][ldw] followed by tile-planar bf16 hi and lo images [rows/64][ldw/8][64][8];
        // rows = N rounded up to 64, ldw = K rounded up to 64 (zero padded)
        EG_REQUIRE((ldw & 63) == 0, EG_ERR_ALIGN, "%s: split-bf16 modes need ldw %% 64 == 0 (EG_PACK_LINEAR weights)", who);
        const size_t rows = (size_t)eg_round_up(n, 64);
        a.whi = reinterpret_cast<const unsigned short*>(w + rows * ldw);
        a.wlo = a.whi + rows * ldw;
    }
    return EG_OK;
}

}  // namespace

extern "C" int eg_linear(const float* x, int32_t lda, const float* w, int32_t ldw, const float* bias,
                         const float* res1, const float* res2, int32_t ldr, float* y, int32_t ldc,
                         int32_t m, int32_t n, int32_t k, int32_t relu, int32_t a_shift, int32_t a_seq,
                         int32_t precision, void* stream) {
    GemmArgs a;
    int rc = fill_common(a, x, lda, w, ldw, m, n, k, precision, "eg_linear");
    if (rc) return rc;
    EG_REQUIRE(y, EG_ERR_BAD_ARG, "eg_linear: null output");
    EG_REQUIRE(a_shift == 0 || a_seq > 0, EG_ERR_BAD_ARG, "eg_linear: a_seq must be positive when a_shift is set");
    a.bias = bias; a.res1 = res1; a.res2 = res2; a.ldr = ldr; a.y = y; a.ldc = ldc; a.relu = relu;
    a.a_shift = a_shift; a.a_seq = a_seq > 0 ? a_seq : 1; a.k_per_split = (int)eg_round_up(k, 64); a.partial = nullptr;
    EgProfScope prof(2, 2.0 * m * (double)n * k, (hipStream_t)stream);
    return launch_gemm(a, 1, precision, (hipStream_t)stream);
}

extern "C" int eg_linear_splitk(const float* x, int32_t lda, const float* w, int32_t ldw, const float* bias, float* y,
                                int32_t ldc, int32_t m, int32_t n, int32_t k, int32_t relu, int32_t splits,
                                float* partial, int32_t precision, void* stream) {
    GemmArgs a;
    int rc = fill_common(a, x, lda, w, ldw, m, n, k, precision, "eg_linear_splitk");
    if (rc) return rc;
    EG_REQUIRE(y && partial && splits > 0, EG_ERR_BAD_ARG, "eg_linear_splitk: null output/partial");
    a.bias = nullptr; a.res1 = a.res2 = nullptr; a.ldr = 0; a.y = y; a.ldc = ldc; a.relu = 0;
    a.a_shift = 0; a.a_seq = 1; a.partial = partial;
    a.k_per_split = (int)eg_round_up(eg_cdiv(k, splits), 64);
    const int nsplit = eg_cdiv(k, a.k_per_split);
    rc = launch_gemm(a, nsplit, precision, (hipStream_t)stream);
    if (rc) return rc;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(eg_cdiv(m * n, 256)), dim3(256), 0, (hipStream_t)stream, partial, bias, y, ldc,
                       m, n, nsplit, relu);
    return eg_check_launch("splitk_reduce");
}
