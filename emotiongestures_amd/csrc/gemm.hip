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
#include <stdlib.h>
#include <string.h>

namespace {

struct GemmArgs {
    const float* x; const float* w; const float* bias; const float* res1; const float* res2; float* y;
    const unsigned short* whi; const unsigned short* wlo;     // bf16 images [Nrows][ldw] (split-bf16 modes)
    int lda, ldw, ldr, ldc, M, N, K, relu, a_shift, a_seq, k_per_split;
    float* partial;
    // optional second output: Y split to bf16 (hi, lo) tile-planar images [ceil(M/64)][yKO][64][8] for a downstream product
    unsigned short* yimg = nullptr;
    int yKO = 0, yoct0 = 0;
    int xoct0 = 0;              // pre-split X: first octet of this product's K range inside the X images
    // training epilogue (eg_linear_ex): ReLU-backward gate and nn.Dropout on the product, both before the residual add
    const float* gate = nullptr; int ldg = 0;       // v = gate[m][n] > 0 ? v : 0
    EgDropout drop;                                 // thr = 0: off; counter = offset + m * N + n (the flat index eg_dropout uses on a [M, N] tensor)
};

// v = acc + bias, then the two training masks: ReLU backward by the saved activation, Dropout from the counter hash (shared by the GEMM epilogue and
// the split-K fold)
__device__ __forceinline__ float epi_masks(const GemmArgs& a, float v, int m, int n, unsigned int dseed) {
    if (a.gate && !(a.gate[(size_t)m * a.ldg + n] > 0.f)) v = 0.f;
    if (a.drop.thr) v = dropout_keep(dseed, a.drop.offset + (unsigned long long)m * a.N + n, a.drop.thr) ? v * a.drop.inv_keep : 0.f;
    return v;
}

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


// v = acc + bias; (gate, dropout: training); + res1; relu; (+res2, relu); lane owns 4 consecutive n of row m (+16 per t), n += 16 per tile
// TRAIN: compile the two training masks in (the fp32-input kernels that carry the training path); the pre-split inference kernels leave them out.
template <int MT, int NT, bool TRAIN = false>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& a, f4 (&acc)[MT][NT], int mbase, int nbase) {
    const bool masks = TRAIN && (a.gate != nullptr || a.drop.thr != 0) && !a.partial;
    const unsigned int dseed = (masks && a.drop.thr) ? dropout_seed(a.drop.seed, a.drop.epoch) : 0u;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int m = mbase + t * 16;
        if (m >= a.M) continue;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
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
            if (masks) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (nn + r < a.N) v[r] = epi_masks(a, v[r], m, nn + r, dseed);
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
            if (a.yimg) {           // 4 consecutive k of the downstream product = half an octet: two 8-byte stores
                const f4 z = (f4){0.f, 0.f, 0.f, 0.f};
                bf8 h8, l8;
                split_octet<true>(v, z, h8, l8);
                const size_t mt_total = (size_t)((a.M + 63) >> 6);
                const size_t slot = (((size_t)(m >> 6) * a.yKO + a.yoct0 + (nn >> 3)) * 64 + (m & 63)) * 8 + (nn & 7);
                typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                const u32x4_t hh = __builtin_bit_cast(u32x4_t, h8), ll = __builtin_bit_cast(u32x4_t, l8);
                *reinterpret_cast<u32x2*>(a.yimg + slot) = (u32x2){hh[0], hh[1]};
                *reinterpret_cast<u32x2*>(a.yimg + mt_total * a.yKO * 512 + slot) = (u32x2){ll[0], ll[1]};
                if (!a.y) continue;
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
    int bx, by;
    xcd_tile(bx, by);
    const int m0 = bx * 64, n0 = by * 64;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    const int kbeg = blockIdx.z * a.k_per_split;
    const int kend = min(a.K, kbeg + a.k_per_split);

    f4 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[t][n] = (f4){0.f, 0.f, 0.f, 0.f};

    // f32 mode: the next K-step's operands are loaded into registers before the MFMAs of the current one (register double
    // buffer), so the global latency hides under the matrix work (this kernel also carries the training path's fp32 products)
    f4 px[2], pw[2];
    auto prefetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + i * 256;
            const int r = (idx >> 6) * 8 + (idx & 7), q = (idx >> 3) & 7;
            px[i] = load_x_quad(a, m0 + r, k0 + q * 4, kend);
            pw[i] = (f4){0.f, 0.f, 0.f, 0.f};
            if (n0 + r < a.N && k0 + q * 4 < kend) pw[i] = *reinterpret_cast<const f4*>(a.w + (size_t)(n0 + r) * a.ldw + k0 + q * 4);
        }
    };
    if (PREC == EG_PREC_F32) prefetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += 32) {
        if (k0 != kbeg) __syncthreads();
        if (PREC == EG_PREC_F32) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int idx = tid + i * 256;
                const int r = (idx >> 6) * 8 + (idx & 7), q = (idx >> 3) & 7;
                lds[q * 64 + r] = px[i];
                lds[512 + q * 64 + r] = pw[i];
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
            if (k0 + 32 < kend) prefetch(k0 + 32);
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

    gemm_epilogue<2, 2, true>(a, acc, m0 + wm + li, n0 + wn + kq * 4);
}

// ---- split-bf16 path: 64x64 tile, K-step 64 -----------------------------------------------------------------
//  * W: pre-split on the host into tile-planar bf16 images [n/64][k/8][64 rows][8] (hi image, lo image): the 8 octets of
//    one K-step are 8 contiguous 1-KiB pieces, copied by global_load_lds (no VGPR/VALU) into a 2-deep LDS ring; LDS image
//    [k/8][row] bf8 => the 16 rows of an MFMA tile are 16 consecutive 16-B slots (conflict-free ds_read_b128);
//  * X (fp32 activations) never touches LDS: each wave owns 16 rows, a lane loads its own MFMA B-operand octets
//    (row = lane&15, k = 8*(lane>>4)..+7: 32 contiguous bytes) one K-step ahead, splits them to (hi, lo) bf16 in registers
//    with v_cvt_pk_bf16_f32 and feeds the MFMAs directly.  (The r01b profile showed the LDS-staged X path LDS-bound:
//    SQ_LDS_BANK_CONFLICT = 50 % of SQ_LDS_IDX_ACTIVE from the ds_write_b128 of the split tile.)
//  One barrier per K-step, for the weight ring only.
template <int TERMS>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmArgs a) {
    constexpr int NIMG = (TERMS == 3) ? 2 : 1;
    constexpr int IMG = 8 * 64;                     // bf8 slots of one weight image of one step (8 octets x 64 rows)
    __shared__ bf8 lds[2 * NIMG * IMG];             // [buf][hi|lo][oct][row]
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx, by;
    xcd_tile(bx, by);
    const int m0 = bx * 64, n0 = by * 64;
    const int kbeg = blockIdx.z * a.k_per_split;
    const int kend = min(a.K, kbeg + a.k_per_split);
    const int nsteps = (kend - kbeg + 63) / 64;
    const int KO = a.ldw >> 3;                      // octets per packed weight row

    const int xm = m0 + wave * 16 + li;
    bool xok = xm < a.M;
    int xsrc = xm;
    if (a.a_shift) {
        xok = xok && (xm % a.a_seq) >= a.a_shift;
        xsrc = xm - a.a_shift;
    }
    const float* xrow = a.x + (size_t)(xok ? xsrc : 0) * a.lda + kq * 8;
    f4 xv[2][2], xn[2][2];
    auto load_x = [&](int k0, f4 (&v)[2][2]) {
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = k0 + g * 32 + kq * 8 + h * 4;
                v[g][h] = (xok && k < kend) ? *reinterpret_cast<const f4*>(xrow + (k0 + g * 32 + h * 4)) : (f4){0.f, 0.f, 0.f, 0.f};
            }
    };
    auto issue_w = [&](int k0, int buf) {
        bf8* W = lds + buf * (NIMG * IMG);
        const size_t gbase = ((size_t)by * KO + (k0 >> 3)) * 64;        // bf8 slots: [n/64][k/8][64]
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

    f4 acc[1][4];
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[0][n] = (f4){0.f, 0.f, 0.f, 0.f};

    issue_w(kbeg, 0);
    load_x(kbeg, xv);
    __syncthreads();
#pragma unroll 1
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        const bool more = s + 1 < nsteps;
        if (more) {
            issue_w(kbeg + (s + 1) * 64, buf ^ 1);
            load_x(kbeg + (s + 1) * 64, xn);
        }
        const bf8* W = lds + buf * (NIMG * IMG);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            bf8 xh, xl;
            split_octet<TERMS == 3>(xv[g][0], xv[g][1], xh, xl);
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const bf8 wh = W[(g * 4 + kq) * 64 + n * 16 + li];
                if (TERMS == 3) {
                    const bf8 wl = W[IMG + (g * 4 + kq) * 64 + n * 16 + li];
                    acc[0][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh, acc[0][n], 0, 0, 0);
                    acc[0][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl, acc[0][n], 0, 0, 0);
                }
                acc[0][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh, acc[0][n], 0, 0, 0);
            }
        }
        if (more) {
#pragma unroll
            for (int g = 0; g < 2; ++g) { xv[g][0] = xn[g][0]; xv[g][1] = xn[g][1]; }
        }
        __syncthreads();
    }
    gemm_epilogue<1, 4>(a, acc, m0 + wave * 16 + li, n0 + kq * 4);
}


// ---- split-bf16 path, all-LDS-DMA pipeline (default when there is no causal row shift) --------------------------------
// Both operands are copied global -> LDS by global_load_lds into a 4-slot ring, 3 K-steps (of 32) ahead of the MFMAs; the
// only waits are a counted s_waitcnt vmcnt (the two youngest step groups stay in flight) and one raw s_barrier per step.
// (Mixing ordinary register loads of X with the LDS-DMA of W makes hipcc wait vmcnt(0) at the first use of the register
//  operand, which exposed a full L2 round trip in every K-step of the kernel above: r01c profile, 12.8 us for a 13 %-MFMA
//  busy 2176x512x512 product.)
//  * X stays fp32 in LDS: image [64 rows][8 quads] (128 B per row), filled 8 rows x 128 contiguous bytes per piece
//    (fully coalesced), with the quad position XOR-swizzled on the SOURCE side, q -> q ^ ((r ^ r>>1) & 7), so that the
//    fragment reads (16 consecutive rows, same octet) are bank-conflict free; split to (hi, lo) bf16 in registers.
//  * W: tile-planar bf16 images as above (4 octets per step = 4 contiguous 1-KiB pieces per image).
//  Rows >= M are clamped (their results are never stored); k >= K is clamped to finite data (the packed weights are zero there).
// WN = MFMA tiles of a wave along N: 2 -> 64 (M) x 64 (N) workgroup tile (static 64 KB of LDS, two workgroups per CU); 4 -> 64 x 128 (two
// 64-row weight image tiles; a wave owns 32 x 64): every X fragment a wave splits now feeds 8 MFMA triples instead of 4 -- the in-kernel split
// (7-8 vector instructions per MFMA in the 64 x 64 shape) is what paces this kernel -- and X is re-read by half as many workgroups.
template <int TERMS, int WN = 2>
__global__ __launch_bounds__(256, 2) void gemm_glds_kernel(GemmArgs a) {
    constexpr int NIMG = (TERMS == 3) ? 2 : 1;
    constexpr int TN = WN / 2;                                                     // 64-row weight image tiles per workgroup
    // Ring depth: a 2-slot ring (48 KB: three workgroups per CU) is 7-17 % faster in a stand-alone loop at 4352 rows, where the operands stay in L2,
    // and SLOWER inside the training step (+ 0.15 ms at 128 clips, + 0.3 ms at 16: profiles/r06_train_ab.txt) -- there every product meets operands the
    // previous kernel has just written or weights from HBM, and the two-step prefetch distance is what hides that
    constexpr int RING = (WN == 2) ? 4 : 3, XS = 64 * 8, WIMG = 4 * 64, WS = TN * NIMG * WIMG, SLOT = XS + WS;      // 16-byte slots
    constexpr int G = 2 + TN * NIMG;                                               // LDS-DMA instructions per wave per step
    extern __shared__ __attribute__((aligned(16))) bf8 lds_raw[];                  // RING * SLOT slots of 16 bytes (bf8 and f4 alike)
    f4* const lds = reinterpret_cast<f4*>(lds_raw);
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx, by;
    xcd_tile(bx, by);
    const int m0 = bx * 64, n0 = by * 64 * TN;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * (16 * WN);
    const int kbeg = blockIdx.z * a.k_per_split;
    const int kend = min(a.K, kbeg + a.k_per_split);
    const int nsteps = (kend - kbeg + 31) / 32;
    const int KO = a.ldw >> 3;
    const int nt_last = ((a.N + 63) >> 6) - 1;

    // X copy role of this lane: rows (p*4 + wave)*8 + (lane>>3), swizzled quad
    const float* xsrc[2];
    int xq[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int r = (p * 4 + wave) * 8 + (lane >> 3);
        const int m = min(m0 + r, a.M - 1);
        xsrc[p] = a.x + (size_t)m * a.lda;
        xq[p] = ((lane & 7) ^ ((r ^ (r >> 1)) & 7)) * 4;
    }
    const int klast = kend - 4;
    auto issue = [&](int step, int slot) {
        const int k0 = kbeg + step * 32;
        f4* X = lds + slot * SLOT;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int k = min(k0 + xq[p], klast);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xsrc[p] + k),
                                             (__attribute__((address_space(3))) void*)(X + (p * 4 + wave) * 64), 16, 0, 0);
        }
        bf8* W = reinterpret_cast<bf8*>(X + XS);
#pragma unroll
        for (int t = 0; t < TN; ++t) {            // weight image tile t of this workgroup (past the last tile: re-read it, those columns are never stored)
            const size_t gbase = ((size_t)min(by * TN + t, nt_last) * KO + (k0 >> 3)) * 64 + wave * 64 + lane;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const bf8*>(a.whi) + gbase),
                                             (__attribute__((address_space(3))) void*)(W + t * NIMG * WIMG + wave * 64), 16, 0, 0);
            if (TERMS == 3)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const bf8*>(a.wlo) + gbase),
                                                 (__attribute__((address_space(3))) void*)(W + t * NIMG * WIMG + WIMG + wave * 64), 16, 0, 0);
        }
    };
    auto wait_groups = [&](int groups_in_flight) {      // all but the youngest `groups_in_flight` step groups have landed
        if (groups_in_flight >= 2) wait_vmcnt_imm<2 * G>();
        else if (groups_in_flight == 1) wait_vmcnt_imm<G>();
        else wait_vmcnt_imm<0>();
    };

    f4 acc[2][WN];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int n = 0; n < WN; ++n) acc[t][n] = (f4){0.f, 0.f, 0.f, 0.f};

    const int pre = min(nsteps, RING - 1);
    for (int s = 0; s < pre; ++s) issue(s, s);
    wait_groups(pre - 1);
    wg_barrier();
    const int fsw = (li ^ (li >> 1)) & 7;
#pragma unroll 1
    for (int s = 0; s < nsteps; ++s) {
        if (s + RING - 1 < nsteps) issue(s + RING - 1, (s + RING - 1) % RING);
        const f4* X = lds + (s % RING) * SLOT;
        const bf8* W = reinterpret_cast<const bf8*>(X + XS);
        bf8 xh[2], xl[2], wh[WN], wl[WN];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const f4* xr = X + (wm + t * 16 + li) * 8;
            const f4 q0 = xr[(2 * kq) ^ fsw], q1 = xr[(2 * kq + 1) ^ fsw];
            split_octet<TERMS == 3>(q0, q1, xh[t], xl[t]);
        }
#pragma unroll
        for (int n = 0; n < WN; ++n) {
            const int col = wn + n * 16, tile = col >> 6, r = (col & 63) + li;
            wh[n] = W[tile * NIMG * WIMG + kq * 64 + r];
            if (TERMS == 3) wl[n] = W[tile * NIMG * WIMG + WIMG + kq * 64 + r];
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int n = 0; n < WN; ++n) {
                if (TERMS == 3) {
                    acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[n], xh[t], acc[t][n], 0, 0, 0);
                    acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[n], xl[t], acc[t][n], 0, 0, 0);
                }
                acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[n], xh[t], acc[t][n], 0, 0, 0);
            }
        // next step's group must have landed; groups beyond it stay in flight across the barrier
        const int newest = min(s + RING - 1, nsteps - 1);
        wait_groups(max(0, newest - (s + 1)));
        __builtin_amdgcn_s_waitcnt(0xC07F);                     // lgkmcnt(0): this step's LDS reads are done before the slot is reused
        wg_barrier();
    }
    gemm_epilogue<2, WN, true>(a, acc, m0 + wm + li, n0 + wn + kq * 4);
}

template <int TERMS, int WN>
int launch_glds(const GemmArgs& a, int splits, hipStream_t st) {
    constexpr int NIMG = (TERMS == 3) ? 2 : 1, TN = WN / 2, RING = (WN == 2) ? 4 : 3;
    constexpr size_t LDS_BYTES = (size_t)RING * (64 * 8 + TN * NIMG * 4 * 64) * 16;
    auto kern = gemm_glds_kernel<TERMS, WN>;
    if (int rc = eg_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), LDS_BYTES, "gemm_glds")) return rc;
    dim3 grid(eg_cdiv(a.M, 64), eg_cdiv(a.N, 64 * TN), splits);
    hipLaunchKernelGGL(kern, grid, dim3(256), LDS_BYTES, st, a);
    return eg_check_launch("gemm_glds");
}

// ---- split-bf16 path with PRE-SPLIT activations ---------------------------------------------------------------------
// X arrives already split to bf16 (hi, lo) in the same tile-planar layout as the weights ([m/64][k/8][64 rows][8]); it is
// produced by the upstream kernel's epilogue (or by split_tile_kernel), once, instead of being re-split by every one of
// the N/64 workgroups that consume it.  The K loop then has no VALU work at all: LDS-DMA copies of both operands 3 steps
// ahead, fragments of step s+1 read from LDS while the MFMAs of step s run (register double buffer), one counted vmcnt and
// one raw barrier per 32-deep step.

// BK = K elements per barrier interval, RING = LDS slots, WM = MFMA tiles of a wave along M: 2 -> 64 (M) x 64 (N) workgroup tile,
// 4 -> 128 x 64 (two 64-row X image tiles; the wave grid stays 2 x 2, a wave owns 64 x 32).  The 64 x 64 tile streams 16 KB of operands
// from L2 into LDS per 48 MFMAs; the 128 x 64 tile 24 KB per 96 (3/4 of the operand stream per MFMA, 6 instead of 4 LDS-DMA pieces and
// 12 instead of 8 ds_read_b128 per wave for twice the MFMAs), at half the workgroups.
template <int TERMS, int BK, int RING, int WM = 2, bool TRAIN = false>
__global__ __launch_bounds__(256, (WM == 2 || RING <= 3) ? 2 : 1) void gemm_presplit_kernel(GemmArgs a, const bf8* __restrict__ xhi, const bf8* __restrict__ xlo, int xKO) {
    constexpr int NIMG = (TERMS == 3) ? 2 : 1;
    constexpr int KG = BK / 32;                                           // MFMA k-groups per step
    constexpr int TM = WM / 2;                                            // 64-row X image tiles per workgroup
    constexpr int IMG = KG * 256;                                         // bf8 slots of one (tile, image): KG*4 octets x 64 rows
    constexpr int XO = TM * NIMG * IMG, SLOT = XO + NIMG * IMG;           // slot: X [tile][hi|lo] then W [hi|lo]
    constexpr int G = (TM + 1) * NIMG * KG;                               // LDS-DMA instructions per wave per step
    static_assert(WM == 2 || (WM == 4 && BK == 32), "wave tile: 32 x 32 or 64 x 32");
    extern __shared__ __attribute__((aligned(16))) bf8 lds[];             // RING * SLOT
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx, by;
    xcd_tile(bx, by);
    const int m0 = bx * 64 * TM, n0 = by * 64;
    const int wmt = wave >> 1;                                            // WM == 2: 32-row half of the tile; WM == 4: which 64-row tile
    const int wm = (TM == 1) ? wmt * 32 : 0, wn = (wave & 1) * 32;
    const int xfrag0 = (TM == 1) ? 0 : wmt * NIMG * IMG;                  // this wave's X fragments start here inside a slot
    const int kbeg = blockIdx.z * a.k_per_split;
    const int kend = min(a.K, kbeg + a.k_per_split);
    const int nsteps = (kend - kbeg + BK - 1) / BK;
    const int KO = a.ldw >> 3;
    const int mt_last = ((a.M + 63) >> 6) - 1;
    // X image tile t of this workgroup (a 128-row workgroup past an odd tile count re-reads the last tile: those rows are never stored)
    const int xt0 = bx * TM, xt1 = min(bx * TM + TM - 1, mt_last);
    auto issue = [&](int step, int slot) {
        const int ko = (kbeg + step * BK) >> 3;
        bf8* S = lds + slot * SLOT;
#pragma unroll
        for (int p = 0; p < KG; ++p) {                                    // piece = octet (p*4 + wave) of this step
            const int d = (p * 4 + wave) * 64;
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                const size_t gx = ((size_t)(t ? xt1 : xt0) * xKO + a.xoct0 + ko + p * 4 + wave) * 64 + lane;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xhi + gx),
                                                 (__attribute__((address_space(3))) void*)(S + t * NIMG * IMG + d), 16, 0, 0);
                if (TERMS == 3)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xlo + gx),
                                                     (__attribute__((address_space(3))) void*)(S + t * NIMG * IMG + IMG + d), 16, 0, 0);
            }
            const size_t gw = ((size_t)by * KO + ko + p * 4 + wave) * 64 + lane;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const bf8*>(a.whi) + gw),
                                             (__attribute__((address_space(3))) void*)(S + XO + d), 16, 0, 0);
            if (TERMS == 3)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const bf8*>(a.wlo) + gw),
                                                 (__attribute__((address_space(3))) void*)(S + XO + IMG + d), 16, 0, 0);
        }
    };
    auto wait_groups = [&](int n) {                 // all but the youngest n step groups of this wave have landed
        switch (n) {
            case 0: wait_vmcnt_imm<0>(); break;
            case 1: wait_vmcnt_imm<G>(); break;
            case 2: wait_vmcnt_imm<2 * G>(); break;
            case 3: wait_vmcnt_imm<3 * G>(); break;
            case 4: wait_vmcnt_imm<4 * G>(); break;
            default: wait_vmcnt_imm<5 * G>(); break;
        }
    };
    static_assert(RING <= 8 && (RING - 3) * G <= 63, "wait_groups covers RING - 3 groups in flight");
    struct Frags { bf8 xh[KG][WM], xl[KG][WM], wh[KG][2], wl[KG][2]; };
    auto read_frags = [&](Frags& f, int slot) {
        const bf8* S = lds + slot * SLOT + li;
#pragma unroll
        for (int g = 0; g < KG; ++g) {
            const int o = (g * 4 + kq) * 64;
#pragma unroll
            for (int t = 0; t < WM; ++t) {
                f.xh[g][t] = S[xfrag0 + o + wm + t * 16];
                if (TERMS == 3) f.xl[g][t] = S[xfrag0 + IMG + o + wm + t * 16];
            }
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                f.wh[g][n] = S[XO + o + wn + n * 16];
                if (TERMS == 3) f.wl[g][n] = S[XO + IMG + o + wn + n * 16];
            }
        }
    };
    f4 acc[WM][2];
#pragma unroll
    for (int t = 0; t < WM; ++t)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[t][n] = (f4){0.f, 0.f, 0.f, 0.f};
    auto mfma_step = [&](const Frags& f) {
#pragma unroll
        for (int g = 0; g < KG; ++g)
#pragma unroll
            for (int t = 0; t < WM; ++t)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    if (TERMS == 3) {
                        acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.wl[g][n], f.xh[g][t], acc[t][n], 0, 0, 0);
                        acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.wh[g][n], f.xl[g][t], acc[t][n], 0, 0, 0);
                    }
                    acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.wh[g][n], f.xh[g][t], acc[t][n], 0, 0, 0);
                }
    };

    // prologue: groups 0..RING-2 in flight; groups 0 and 1 landed before the loop
    const int pre = min(nsteps, RING - 1);
    for (int s = 0; s < pre; ++s) issue(s, s);
    wait_groups(max(0, pre - 2));
    wg_barrier();
    Frags fa, fb;
    read_frags(fa, 0);
    __builtin_amdgcn_s_waitcnt(0xC07F);       // nothing pending at loop entry, else the compiler waits (lgkmcnt 0) in front of the first MFMAs of every pair
    // steps are processed in pairs so that the two fragment sets have static names
    auto step = [&](int s, Frags& cur, Frags& nxt) {
        if (s + RING - 1 < nsteps) issue(s + RING - 1, (s + RING - 1) % RING);
        if (s + 1 < nsteps) read_frags(nxt, (s + 1) % RING);
        mfma_step(cur);
        // at the next step's start, group s+2 must be landed (its fragments are read then); younger ones may stay in flight
        const int newest = min(s + RING - 1, nsteps - 1);
        wait_groups(max(0, min(RING - 3, newest - (s + 2))));
        __builtin_amdgcn_s_waitcnt(0xC07F);       // lgkmcnt(0) as a real instruction, so the compiler's own wait insertion sees it
        wg_barrier();
    };
    int s0 = 0;
    if constexpr (BK == 32) {
        // Steady state, RING steps per iteration: every step issues a copy (group s+RING-1), so there are no conditionals, the
        // ring slots are compile-time constants and the source addresses are running pointers (one 64-bit add each per
        // step).  A copy has RING-2 steps to land.
        constexpr int AHEAD = (RING - 1) * 4;                           // octets between this step and the group it issues
        const size_t xoff = (size_t)(a.xoct0 + (kbeg >> 3) + AHEAD + wave) * 64 + lane;
        const bf8* px = xhi + (size_t)xt0 * xKO * 64 + xoff;
        const bf8* pxl = xlo + (size_t)xt0 * xKO * 64 + xoff;
        const bf8* px1 = xhi + (size_t)xt1 * xKO * 64 + xoff;          // second X tile (WM == 4 only)
        const bf8* pxl1 = xlo + (size_t)xt1 * xKO * 64 + xoff;
        const bf8* pw = reinterpret_cast<const bf8*>(a.whi) + ((size_t)by * KO + (kbeg >> 3) + AHEAD + wave) * 64 + lane;
        const bf8* pwl = reinterpret_cast<const bf8*>(a.wlo) + ((size_t)by * KO + (kbeg >> 3) + AHEAD + wave) * 64 + lane;
        auto fast = [&](const int j, Frags& cur, Frags& nxt) {     // j = step index mod RING (static)
            bf8* S = lds + ((j + RING - 1) % RING) * SLOT + wave * 64;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)px, (__attribute__((address_space(3))) void*)S, 16, 0, 0);
            if (TERMS == 3)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pxl, (__attribute__((address_space(3))) void*)(S + IMG), 16, 0, 0);
            if (TM == 2) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)px1, (__attribute__((address_space(3))) void*)(S + NIMG * IMG), 16, 0, 0);
                if (TERMS == 3)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pxl1, (__attribute__((address_space(3))) void*)(S + NIMG * IMG + IMG), 16, 0, 0);
                px1 += 256; pxl1 += 256;
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pw, (__attribute__((address_space(3))) void*)(S + XO), 16, 0, 0);
            if (TERMS == 3)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pwl, (__attribute__((address_space(3))) void*)(S + XO + IMG), 16, 0, 0);
            px += 256; pxl += 256; pw += 256; pwl += 256;
            read_frags(nxt, (j + 1) % RING);
            mfma_step(cur);
            wait_vmcnt_imm<(RING - 3) * G>();       // group s+2 landed; the RING-3 younger copies stay in flight
            wait_lgkmcnt0();
            wg_barrier();
        };
        constexpr int UNR = (RING % 2) ? 2 * RING : RING;               // whole ring turns and an even number of steps (fa / fb)
#pragma unroll 1
        for (; s0 + UNR + RING - 1 <= nsteps; s0 += UNR) {
#pragma unroll
            for (int j = 0; j < UNR; j += 2) {
                fast(j % RING, fa, fb);
                fast((j + 1) % RING, fb, fa);
            }
        }
    }
#pragma unroll 1
    for (int s = s0; s < nsteps; s += 2) {          // head of short products and the last steps (copies run out): generic path
        step(s, fa, fb);
        if (s + 1 < nsteps) step(s + 1, fb, fa);
    }
    gemm_epilogue<WM, 2, TRAIN>(a, acc, m0 + ((TM == 1) ? wm : wmt * 64) + li, n0 + wn + kq * 4);
}

template <int TERMS, int BK, int RING, int WM = 2, bool TRAIN = false>
int launch_presplit(const GemmArgs& a, const bf8* xhi, const bf8* xlo, int xko, dim3 grid, hipStream_t st) {
    constexpr int NIMG = (TERMS == 3) ? 2 : 1;
    constexpr size_t LDS_BYTES = (size_t)RING * (WM / 2 + 1) * NIMG * (BK / 32) * 4 * 64 * 16;
    auto kern = gemm_presplit_kernel<TERMS, BK, RING, WM, TRAIN>;
    if (int rc = eg_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), LDS_BYTES, "gemm_presplit")) return rc;
    if (WM == 4) grid.x = (grid.x + 1) / 2;          // grid.x arrives as the number of 64-row tiles
    hipLaunchKernelGGL(kern, grid, dim3(256), LDS_BYTES, st, a, xhi, xlo, xko);
    return eg_check_launch("gemm_presplit");
}

// ---- pre-split product, 128 x 128 tile (large M) -----------------------------------------------------------------------
// The 64 x 64-tile kernel above streams 16 KB from L2 into LDS for every 48 MFMAs of a workgroup; at M = B*draws*frames rows
// (BASELINE cfg 5: 69 632) that stream, 13-15 TB/s chip-wide, is what bounds it (200-250 TFLOP/s algorithmic).  This variant
// doubles the reuse: 512 threads = 8 waves as 2 (M) x 4 (N), wave tile 64 x 32 (4 x 2 MFMA tiles, 24 MFMAs per step in bf16x3),
// workgroup tile 128 x 128 = two 64-row image tiles of X and of W, 32 KB per 32-deep step, 3-slot ring (96 KB: one workgroup
// = two waves per SIMD).  Schedule as in the convolution: one barrier per step between the two MFMA halves, counted vmcnt
// (the copy of step s+2 stays in flight), fragments of step s+1 read after the barrier under the second half.
template <int TERMS, int RING = 3, bool TRAIN = false>
__global__ __launch_bounds__(512, 1) void gemm_presplit128_kernel(GemmArgs a, const bf8* __restrict__ xhi, const bf8* __restrict__ xlo, int xKO,
                                                                  int m_tiles, int n_tiles) {
    constexpr int NIMG = (TERMS == 3) ? 2 : 1;
    constexpr int OPI = 8 * 64;                         // bf8 slots of one image of one operand: [tile(2)][octet(4)][64 rows]
    constexpr int OP = NIMG * OPI, SLOT = 2 * OP;
    constexpr int G = 2 * NIMG;                         // LDS-DMA instructions per wave per step (one 1-KiB piece each)
    extern __shared__ __attribute__((aligned(16))) bf8 lds[];
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx, by;
    xcd_tile(bx, by);
    const int m0 = bx * 128, n0 = by * 128;
    const int wm = wave >> 2, wn = wave & 3;            // wave tile: rows [wm*64, +64), columns [wn*32, +32)
    const int nsteps = (a.K + 31) / 32;
    const int KO = a.ldw >> 3;
    // copy role: wave w moves piece w of each image: tile = w>>2, octet = w&3.  A 128-row workgroup whose second 64-row
    // tile does not exist (odd tile counts) re-reads the last tile; those rows / columns are never stored.
    const int pt = wave >> 2, po = wave & 3;
    const size_t gx = ((size_t)min(bx * 2 + pt, m_tiles - 1) * xKO + a.xoct0 + po) * 64 + lane;
    const size_t gw = ((size_t)min(by * 2 + pt, n_tiles - 1) * KO + po) * 64 + lane;
    const bf8* whi = reinterpret_cast<const bf8*>(a.whi);
    const bf8* wlo = reinterpret_cast<const bf8*>(a.wlo);
    auto issue = [&](int step, int slot) {
        bf8* S = lds + slot * SLOT + wave * 64;
        const size_t o = (size_t)step * 4 * 64;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xhi + gx + o), (__attribute__((address_space(3))) void*)S, 16, 0, 0);
        if (TERMS == 3)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xlo + gx + o), (__attribute__((address_space(3))) void*)(S + OPI), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(whi + gw + o), (__attribute__((address_space(3))) void*)(S + OP), 16, 0, 0);
        if (TERMS == 3)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wlo + gw + o), (__attribute__((address_space(3))) void*)(S + OP + OPI), 16, 0, 0);
    };
    struct Frags { bf8 xh[4], xl[4], wh[2], wl[2]; };
    auto read_frags = [&](Frags& f, int slot) {
        const bf8* X = lds + slot * SLOT + (wm * 4 + kq) * 64 + li;
        const bf8* W = lds + slot * SLOT + OP + ((wn >> 1) * 4 + kq) * 64 + (wn & 1) * 32 + li;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f.xh[t] = X[t * 16];
            if (TERMS == 3) f.xl[t] = X[OPI + t * 16];
        }
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            f.wh[n] = W[n * 16];
            if (TERMS == 3) f.wl[n] = W[OPI + n * 16];
        }
    };
    f4 acc[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[t][n] = (f4){0.f, 0.f, 0.f, 0.f};
    auto mfma_half = [&](const Frags& f, int half) {
#pragma unroll
        for (int t = half * 2; t < half * 2 + 2; ++t)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                if (TERMS == 3) {
                    acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.wl[n], f.xh[t], acc[t][n], 0, 0, 0);
                    acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.wh[n], f.xl[t], acc[t][n], 0, 0, 0);
                }
                acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.wh[n], f.xh[t], acc[t][n], 0, 0, 0);
            }
    };
    static_assert(RING == 3 || RING == 4, "ring depth");
    auto wait_younger = [&](int n) {                // all but the youngest n groups of this wave have landed
        if (n <= 0) wait_vmcnt_imm<0>();
        else if (n == 1) wait_vmcnt_imm<G>();
        else wait_vmcnt_imm<2 * G>();
    };
    const int pre = min(nsteps, RING - 1);
    for (int s = 0; s < pre; ++s) issue(s, s);
    wait_younger(pre - 1);                          // group 0 landed
    wg_barrier();
    Frags fa, fb;
    read_frags(fa, 0);
    auto step = [&](int s, Frags& cur, Frags& nxt) {
        if (s + RING - 1 < nsteps) issue(s + RING - 1, (s + RING - 1) % RING);      // slot of step s-1: its fragments were read before the barrier of step s-1
        mfma_half(cur, 0);
        __builtin_amdgcn_sched_barrier(0);
        wait_younger(min(s + RING - 1, nsteps - 1) - (s + 1));                      // group s+1 landed; younger copies stay in flight
        wait_lgkmcnt0();
        wg_barrier();
        if (s + 1 < nsteps) read_frags(nxt, (s + 1) % RING);
        mfma_half(cur, 1);
    };
#pragma unroll 1
    for (int s = 0; s < nsteps; s += 2) {
        step(s, fa, fb);
        if (s + 1 < nsteps) step(s + 1, fb, fa);
    }
    gemm_epilogue<4, 2, TRAIN>(a, acc, m0 + wm * 64 + li, n0 + wn * 32 + kq * 4);
}

template <int TERMS, int RING = 3, bool TRAIN = false>
int launch_presplit128(const GemmArgs& a, const bf8* xhi, const bf8* xlo, int xko, int m_tiles, int n_tiles, hipStream_t st) {
    constexpr int NIMG = (TERMS == 3) ? 2 : 1;
    constexpr size_t LDS_BYTES = (size_t)RING * 2 * NIMG * 8 * 64 * 16;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    auto kern = gemm_presplit128_kernel<TERMS, RING, TRAIN>;
    if (int rc = eg_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), LDS_BYTES, "gemm_presplit128")) return rc;
    dim3 grid(eg_cdiv(m_tiles, 2), eg_cdiv(n_tiles, 2), 1);
    hipLaunchKernelGGL(kern, grid, dim3(512), LDS_BYTES, st, a, xhi, xlo, xko, m_tiles, n_tiles);
    return eg_check_launch("gemm_presplit128");
}

// Tile choice of the pre-split product.  Default policy (measured on the 4-lane headline and on tools/bench_ops.py):
//   the policy of `presplit_tile_default` below (stand-alone: 64 x 64 up to 1024 workgroups; with the caller's shared-chip hint: 128 x 128 from 64).
// EG_GEMM_TILE overrides it for A/B runs: "64" | "128x64" (4-slot ring, one workgroup per CU) | "128x64r3" (3-slot ring, two per CU) | "128".
enum PresplitTile { TILE_64 = 0, TILE_128x64 = 1, TILE_128x64_R3 = 2, TILE_128 = 3, TILE_64_R8 = 4, TILE_128x64_R6 = 5, TILE_128_R4 = 6, TILE_AUTO = -1 };
int presplit_tile_override() {          // read per call (a tool / test switches it between launches; a captured graph keeps what it was captured with)
    const char* e = getenv("EG_GEMM_TILE");
    if (!e || !e[0]) return TILE_AUTO;
    if (!strcmp(e, "64")) return TILE_64;
    if (!strcmp(e, "128x64")) return TILE_128x64;
    if (!strcmp(e, "128x64r3")) return TILE_128x64_R3;
    if (!strcmp(e, "128")) return TILE_128;
    if (!strcmp(e, "64r8")) return TILE_64_R8;            // deeper rings: more operand bytes in flight per CU (one workgroup per CU)
    if (!strcmp(e, "128x64r6")) return TILE_128x64_R6;
    if (!strcmp(e, "128r4")) return TILE_128_R4;
    return TILE_AUTO;
}
int presplit_tile_default(int m, int n, int shared_chip) {
    // Stand-alone (one stream, nothing else resident) the 64 x 64 tile is the faster one below ~1000 workgroups: at the headline's 2176-row products
    // 9.3 vs 13.3 us (N = 512), 25.6 vs 39.2 us (K = 2048) -- 68 workgroups of 128 x 128 leave 188 CUs idle (profiles/r04b_gemm_tile_sweep.txt).
    // The 128 x 128 tile (8 waves, 32 KB of operands per 192 MFMAs) wins from ~1024 workgroups up (diversity sampling, M = 69 632) -- and, from 64
    // workgroups up, when the CALLER says other work fills the chip (`shared_chip`: ClipPipeline with several batches in flight sets it through
    // EgGeneratorConfig.reserved[4]): a product then costs CU time, not latency, and the large tile holds a quarter of the CUs for half the
    // operand traffic -- same-box A/B of the 4-lane bench line, twice each: 25 537 / 25 605 clips/s against 25 311 / 25 288 (profiles/r04c_gemm_tile_ab.txt).
    const long wgs = (long)eg_cdiv(m, 128) * eg_cdiv(n, 128);
    if (wgs >= (shared_chip ? 64 : 1024)) return TILE_128;
    return TILE_64;
}
int dispatch_presplit(const GemmArgs& a, const bf8* xhi, const bf8* xlo, int xko, int precision, hipStream_t st, EgProfScope* prof = nullptr, int shared_chip = 0) {
    const int mt = eg_cdiv(a.M, 64), nt = eg_cdiv(a.N, 64);
    int tile = presplit_tile_override();
    if (tile == TILE_AUTO) tile = presplit_tile_default(a.M, a.N, shared_chip);
    if (prof) {
        const int tm = (tile == TILE_64 || tile == TILE_64_R8) ? 1 : 2, tn = (tile == TILE_128 || tile == TILE_128_R4) ? 2 : 1;
        prof->workgroups(eg_cdiv(mt, tm) * eg_cdiv(nt, tn));
    }
    const bool x3 = precision == EG_PREC_BF16X3;
    dim3 grid(mt, nt, 1);
    if (a.gate || a.drop.thr) {             // the training epilogue (ReLU-backward gate / Dropout): split-bf16 only, the two default tiles
        if (!x3) { eg_set_error("pre-split product with a training epilogue: bf16x3 only"); return EG_ERR_UNSUPPORTED; }
        if (tile == TILE_128) return launch_presplit128<3, 3, true>(a, xhi, xlo, xko, mt, nt, st);
        return launch_presplit<3, 32, 4, 2, true>(a, xhi, xlo, xko, grid, st);
    }
    switch (tile) {
        case TILE_128: return x3 ? launch_presplit128<3>(a, xhi, xlo, xko, mt, nt, st) : launch_presplit128<1>(a, xhi, xlo, xko, mt, nt, st);
        case TILE_128x64: return x3 ? launch_presplit<3, 32, 4, 4>(a, xhi, xlo, xko, grid, st) : launch_presplit<1, 32, 4, 4>(a, xhi, xlo, xko, grid, st);
        case TILE_128_R4: return x3 ? launch_presplit128<3, 4>(a, xhi, xlo, xko, mt, nt, st) : launch_presplit128<1, 4>(a, xhi, xlo, xko, mt, nt, st);
        case TILE_64_R8: return x3 ? launch_presplit<3, 32, 8>(a, xhi, xlo, xko, grid, st) : launch_presplit<1, 32, 8>(a, xhi, xlo, xko, grid, st);
        case TILE_128x64_R6: return x3 ? launch_presplit<3, 32, 6, 4>(a, xhi, xlo, xko, grid, st) : launch_presplit<1, 32, 6, 4>(a, xhi, xlo, xko, grid, st);
        case TILE_128x64_R3: return x3 ? launch_presplit<3, 32, 3, 4>(a, xhi, xlo, xko, grid, st) : launch_presplit<1, 32, 3, 4>(a, xhi, xlo, xko, grid, st);
        default: break;
    }
    // 64 x 64: 32-deep steps, 4 slots (64 KB: two workgroups per CU).  Measured and dropped in round 1 (2176-row products, bf16x3):
    // 64-deep steps with 4 slots (16 % slower), 5 slots (no change), 8 slots (one workgroup per CU: the 272-tile grids
    // then need two passes, 50 % slower), register-staged copies instead of LDS-DMA (30 % slower).
    return x3 ? launch_presplit<3, 32, 4>(a, xhi, xlo, xko, grid, st) : launch_presplit<1, 32, 4>(a, xhi, xlo, xko, grid, st);
}

// ---- products of few rows (one clip: M <= 64 by default) ------------------------------------------------------------------
// With few row tiles the tiled kernels above run 16-64 barriered K steps on a handful of workgroups (a 512 -> 512 product of 34 rows: 16 workgroups,
// ~10 us; of 544 rows: 72 workgroups, ~11 us -- almost all of it LDS-DMA latency and barriers).  Here a workgroup owns 16 output columns of a 64-row
// block; its four waves take the K steps round robin (wave w: steps w, w + 4, ...), every operand goes global -> registers as 16-byte loads (the X rows
// are L2 resident and L2 feeds the 32 column workgroups that re-read them; a weight element is used once per row tile: nothing to share through LDS),
// four steps of loads are in flight before the first MFMA, and the four partial accumulators are folded through LDS in a fixed order (wave 0 .. 3) by
// the wave that then runs the row tile's epilogue.  (N / 16) x (M / 64) workgroups, no barrier in the K loop, no LDS-DMA.  A row's result depends on
// K only (not on M): bitwise the same whatever batch the row travels in, as long as the batch stays in this kernel's range.
// XF32: X is fp32 rows (split in registers; a_shift honoured) instead of pre-split images.  TRAIN: the training epilogue's masks compiled in.
template <int RT, bool XF32, bool TRAIN>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmArgs a, const bf8* __restrict__ xhi, const bf8* __restrict__ xlo, int xKO) {
    constexpr int B = 4;                                   // K steps per batch of loads
    __shared__ f4 red[4][RT][64];
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 64, KO = a.ldw >> 3;
    const int nsteps = (a.K + 31) >> 5;
    const bf8* __restrict__ wh = reinterpret_cast<const bf8*>(a.whi) + ((size_t)(n0 >> 6) * KO) * 64 + (n0 & 63) + li;
    const bf8* __restrict__ wl = reinterpret_cast<const bf8*>(a.wlo) + ((size_t)(n0 >> 6) * KO) * 64 + (n0 & 63) + li;
    const size_t xtile = (size_t)blockIdx.y * xKO * 64;    // this row block's X image tile
    struct Frag { bf8 wh, wl, xh[RT], xl[RT]; };
    auto load = [&](Frag& f, int step) {                   // step < nsteps
        const int ko = step * 4 + kq;
        f.wh = wh[(size_t)ko * 64];
        f.wl = wl[(size_t)ko * 64];
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            if constexpr (XF32) {
                const int m = m0 + t * 16 + li, k = ko * 8;
                split_octet<true>(load_x_quad(a, m, k, a.K), load_x_quad(a, m, k + 4, a.K), f.xh[t], f.xl[t]);
            } else {
                const size_t gx = xtile + (size_t)(a.xoct0 + ko) * 64 + t * 16 + li;
                f.xh[t] = xhi[gx];
                f.xl[t] = xlo[gx];
            }
        }
    };
    f4 acc[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) acc[t] = (f4){0.f, 0.f, 0.f, 0.f};
    auto mfma = [&](const Frag& f) {
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.wl, f.xh[t], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.wh, f.xl[t], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.wh, f.xh[t], acc[t], 0, 0, 0);
        }
    };
    // this wave's steps: wave, wave + 4, ...; batches of B with the next batch's loads issued before the current batch's MFMAs
    const int mine = (nsteps - wave + 3) >> 2;
    Frag fa[B], fb[B];
    auto load_batch = [&](Frag (&f)[B], int first) {       // steps first .. first + B - 1 of this wave's list
#pragma unroll
        for (int j = 0; j < B; ++j)
            if (first + j < mine) load(f[j], wave + 4 * (first + j));
    };
    auto mfma_batch = [&](const Frag (&f)[B], int first) {
#pragma unroll
        for (int j = 0; j < B; ++j)
            if (first + j < mine) mfma(f[j]);
    };
    load_batch(fa, 0);
#pragma unroll 1
    for (int s = 0; s < mine; s += 2 * B) {
        if (s + B < mine) load_batch(fb, s + B);
        mfma_batch(fa, s);
        if (s + 2 * B < mine) load_batch(fa, s + 2 * B);
        mfma_batch(fb, s + B);
    }
#pragma unroll
    for (int t = 0; t < RT; ++t) red[wave][t][lane] = acc[t];
    __syncthreads();
    // row tile t is finished by wave t (RT <= 4: one tile per wave)
    if (wave < RT) {
        f4 v[1][1];
        v[0][0] = (red[0][wave][lane] + red[1][wave][lane]) + (red[2][wave][lane] + red[3][wave][lane]);
        gemm_epilogue<1, 1, TRAIN>(a, v, m0 + wave * 16 + li, n0 + kq * 4);
    }
}
template <bool XF32, bool TRAIN>
int launch_skinny_t(const GemmArgs& a, const bf8* xhi, const bf8* xlo, int xko, hipStream_t st) {
    const dim3 grid(eg_cdiv(a.N, 16), eg_cdiv(a.M, 64)), block(256);
    switch (a.M > 64 ? 4 : eg_cdiv(a.M, 16)) {
        case 1: hipLaunchKernelGGL((gemm_skinny_kernel<1, XF32, TRAIN>), grid, block, 0, st, a, xhi, xlo, xko); break;
        case 2: hipLaunchKernelGGL((gemm_skinny_kernel<2, XF32, TRAIN>), grid, block, 0, st, a, xhi, xlo, xko); break;
        case 3: hipLaunchKernelGGL((gemm_skinny_kernel<3, XF32, TRAIN>), grid, block, 0, st, a, xhi, xlo, xko); break;
        default: hipLaunchKernelGGL((gemm_skinny_kernel<4, XF32, TRAIN>), grid, block, 0, st, a, xhi, xlo, xko); break;
    }
    return eg_check_launch("gemm (few rows)");
}
template <bool XF32>
int launch_skinny(const GemmArgs& a, const bf8* xhi, const bf8* xlo, int xko, hipStream_t st) {
    if (a.gate || a.drop.thr) return launch_skinny_t<XF32, true>(a, xhi, xlo, xko, st);
    return launch_skinny_t<XF32, false>(a, xhi, xlo, xko, st);
}
// few rows, split-bf16, no split-K request: the skinny kernel (EG_GEMM_SKINNY=0, read per call, keeps the tiled kernels).
// One clip (<= 64 rows) is where it wins (1.26 -> 1.10 ms per clip end to end).  With more row blocks every one of the N / 16 column workgroups re-reads
// and re-splits the block's X rows: at the 16-clip training step's 544 rows the step went 7.69 -> 8.42 ms, at 1088 rows 10.4 -> 12.9 ms (same box,
// bench.py --train, twice each) -- the LDS-tiled kernels keep everything above one row block.
constexpr int SKINNY_ROWS = 64;
bool skinny_ok(const GemmArgs& a, int precision) {
    if (precision != EG_PREC_BF16X3 || a.partial) return false;
    int limit = SKINNY_ROWS;
    if (const char* e = getenv("EG_GEMM_SKINNY")) { if (e[0] == '0') limit = 0; }
    return a.M <= limit;
}

// fp32 [M, K] (row stride lda) -> bf16 (hi, lo) tile-planar images [ceil(M/64)][Kpad/8][64][8]; rows >= M and k >= K are zero.
__global__ __launch_bounds__(256) void split_tile_kernel(const float* __restrict__ x, int lda, int M, int K, int KO,
                                                         bf8* __restrict__ hi, bf8* __restrict__ lo) {
    const size_t total = (size_t)gridDim.y * 64 * KO;       // one thread per (row, octet)
    const int mt = blockIdx.y;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < 64 * KO; i += gridDim.x * 256) {
        const int o = i >> 6, r = i & 63, m = mt * 64 + r, k = o * 8;
        f4 v0 = (f4){0.f, 0.f, 0.f, 0.f}, v1 = v0;
        if (m < M) {
            if (k < K) v0 = *reinterpret_cast<const f4*>(x + (size_t)m * lda + k);
            if (k + 4 < K) v1 = *reinterpret_cast<const f4*>(x + (size_t)m * lda + k + 4);
        }
        bf8 h, l;
        split_octet<true>(v0, v1, h, l);
        hi[((size_t)mt * KO + o) * 64 + r] = h;
        lo[((size_t)mt * KO + o) * 64 + r] = l;
    }
    (void)total;
}

// fold of the split-K partials with the full epilogue of gemm_epilogue (bias, gate, dropout, res1, relu, res2), fixed order over the splits
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmArgs a, int splits) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.M * a.N) return;
    const int m = i / a.N, n = i - m * a.N;
    float s = a.bias ? a.bias[n] : 0.f;
    for (int z = 0; z < splits; ++z) s += a.partial[(size_t)z * a.M * a.N + i];
    if (a.gate || a.drop.thr) s = epi_masks(a, s, m, n, a.drop.thr ? dropout_seed(a.drop.seed, a.drop.epoch) : 0u);
    if (a.res1) s += a.res1[(size_t)m * a.ldr + n];
    if (a.relu) s = fmaxf(s, 0.f);
    if (a.res2) s = fmaxf(s + a.res2[(size_t)m * a.ldr + n], 0.f);
    a.y[(size_t)m * a.ldc + n] = s;
}

int launch_gemm(GemmArgs& a, int splits, int precision, hipStream_t st) {
    dim3 grid(eg_cdiv(a.M, 64), eg_cdiv(a.N, 64), splits), block(256);
    if (precision == EG_PREC_F32) hipLaunchKernelGGL((gemm_kernel<EG_PREC_F32>), grid, block, 0, st, a);
    else if (a.a_shift) {       // causal row shift needs zero rows: register-staged X path
        if (precision == EG_PREC_BF16X3) hipLaunchKernelGGL((gemm_bf16_kernel<3>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((gemm_bf16_kernel<1>), grid, block, 0, st, a);
    } else {
        // 64 x 128 workgroup tile once the launch has >= 256 workgroups of that shape (one per CU): 128-clip training steps (M = 4352); below that
        // the 64 x 64 tile's doubled workgroup count wins.  Same-box A/B of the 128-clip training step, alternating: 28.36 / 28.44 ms (wide) vs
        // 28.61 / 28.72 ms (64 x 64).  EG_GLDS_TILE = "0" | "1" forces one (A/B runs).  Same K order per output element: bitwise-identical results.
        bool wide = (long)eg_cdiv(a.M, 64) * eg_cdiv(a.N, 128) * splits >= 256;
        if (const char* e = getenv("EG_GLDS_TILE")) wide = e[0] == '1';
        if (precision == EG_PREC_BF16X3) return wide ? launch_glds<3, 4>(a, splits, st) : launch_glds<3, 2>(a, splits, st);
        return wide ? launch_glds<1, 4>(a, splits, st) : launch_glds<1, 2>(a, splits, st);
    }
    return eg_check_launch("gemm");
}

// split-K: raw partial sums from `splits` K slices, then one fold that applies the whole epilogue (fixed order: deterministic)
int launch_splitk(GemmArgs& a, int splits, int precision, hipStream_t st) {
    a.k_per_split = (int)eg_round_up(eg_cdiv(a.K, splits), 64);
    const int nsplit = eg_cdiv(a.K, a.k_per_split);
    if (int rc = launch_gemm(a, nsplit, precision, st)) return rc;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(eg_cdiv(a.M * a.N, 256)), dim3(256), 0, st, a, nsplit);
    return eg_check_launch("splitk_reduce");
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

// Internal (C++ linkage) product with every option; generator.hip uses it to chain products through pre-split images.
int egi_linear(const EgiLinear& p, hipStream_t st) {
    GemmArgs a;
    EG_REQUIRE((p.x || p.ximg) && p.w && (p.y || p.yimg) && p.m > 0 && p.n > 0 && p.k > 0, EG_ERR_BAD_ARG, "egi_linear: null pointer or empty shape");
    EG_REQUIRE((p.k & 3) == 0 && (p.ldw & 3) == 0, EG_ERR_ALIGN, "egi_linear: K, ldw %% 4");
    a.x = p.x; a.w = p.w; a.lda = p.lda; a.ldw = p.ldw; a.M = p.m; a.N = p.n; a.K = p.k;
    a.whi = a.wlo = nullptr;
    if (p.precision != EG_PREC_F32) {
        EG_REQUIRE((p.ldw & 63) == 0, EG_ERR_ALIGN, "egi_linear: packed weights need ldw %% 64 == 0");
        const size_t rows = (size_t)eg_round_up(p.n, 64);
        a.whi = reinterpret_cast<const unsigned short*>(p.w + rows * p.ldw);
        a.wlo = a.whi + rows * p.ldw;
    }
    a.bias = p.bias; a.res1 = p.res1; a.res2 = p.res2; a.ldr = p.ldr; a.y = p.y; a.ldc = p.ldc; a.relu = p.relu;
    a.a_shift = p.a_shift; a.a_seq = p.a_seq > 0 ? p.a_seq : 1; a.k_per_split = (int)eg_round_up(p.k, 64); a.partial = nullptr;
    if (p.yimg) {
        EG_REQUIRE((p.n & 3) == 0 && (p.yK & 63) == 0 && (p.yk0 & 7) == 0, EG_ERR_ALIGN, "egi_linear: image output alignment");
        a.yimg = reinterpret_cast<unsigned short*>(p.yimg); a.yKO = p.yK >> 3; a.yoct0 = p.yk0 >> 3;
    }
    EgProfScope prof(p.precision == EG_PREC_F32 ? 5 : ((p.ximg) ? 3 : (p.a_shift ? 4 : 2)), 2.0 * p.m * (double)p.n * p.k, st);
    if (p.ximg && p.precision != EG_PREC_F32) {
        EG_REQUIRE((p.xK & 63) == 0 && (p.xk0 & 31) == 0 && (p.k & 31) == 0, EG_ERR_ALIGN, "egi_linear: pre-split X needs K %% 32 == 0");
        const int xko = p.xK >> 3, mt = eg_cdiv(p.m, 64);
        const bf8* xhi = reinterpret_cast<const bf8*>(p.ximg);
        const bf8* xlo = xhi + (size_t)mt * xko * 64;
        a.xoct0 = p.xk0 >> 3;
        if (skinny_ok(a, p.precision)) {        // one clip: 16 columns per workgroup, K over the waves, no LDS staging (replaces the split-K request too)
            prof.workgroups(eg_cdiv(a.N, 16) * eg_cdiv(a.M, 64));
            return launch_skinny<false>(a, xhi, xlo, xko, st);
        }
        if (p.splits > 1) {
            // a single clip (one or two row tiles) and a deep K: 64 serial K steps on a handful of workgroups.  Split K over workgroup slices (64 x 64
            // tile, blockIdx.z) and fold the partials in a fixed order with the whole epilogue (bias, residual, ReLU): shorter wall time for one more launch.
            EG_REQUIRE(p.partial && !p.yimg && p.y, EG_ERR_BAD_ARG, "egi_linear: split-K needs a partial buffer and an fp32 output (no image output)");
            a.partial = p.partial;
            a.k_per_split = (int)eg_round_up(eg_cdiv(a.K, p.splits), 64);
            const int nsplit = eg_cdiv(a.K, a.k_per_split);
            prof.workgroups(mt * eg_cdiv(a.N, 64) * nsplit);
            dim3 grid(mt, eg_cdiv(a.N, 64), nsplit);
            const int rc = (p.precision == EG_PREC_BF16X3) ? launch_presplit<3, 32, 4>(a, xhi, xlo, xko, grid, st) : launch_presplit<1, 32, 4>(a, xhi, xlo, xko, grid, st);
            if (rc) return rc;
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3(eg_cdiv(a.M * a.N, 256)), dim3(256), 0, st, a, nsplit);
            return eg_check_launch("splitk_reduce");
        }
        return dispatch_presplit(a, xhi, xlo, xko, p.precision, st, &prof, p.shared_chip);
    }
    EG_REQUIRE(p.x && (p.lda & 3) == 0, EG_ERR_BAD_ARG, "egi_linear: fp32 input missing");
    if (skinny_ok(a, p.precision)) {
        prof.workgroups(eg_cdiv(a.N, 16) * eg_cdiv(a.M, 64));
        return launch_skinny<true>(a, nullptr, nullptr, 0, st);
    }
    prof.workgroups(eg_cdiv(p.m, 64) * eg_cdiv(p.n, 64));
    return launch_gemm(a, 1, p.precision, st);
}
int egi_split_tiles(const float* x, int lda, int m, int k, void* images, hipStream_t st) { return eg_split_tiles(x, lda, m, k, images, st); }

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
    EgProfScope prof(precision == EG_PREC_F32 ? 5 : (a_shift ? 4 : 2), 2.0 * m * (double)n * k, (hipStream_t)stream);
    if (skinny_ok(a, precision)) return launch_skinny<true>(a, nullptr, nullptr, 0, (hipStream_t)stream);      // one clip's rows
    return launch_gemm(a, 1, precision, (hipStream_t)stream);
}

extern "C" int eg_linear_splitk(const float* x, int32_t lda, const float* w, int32_t ldw, const float* bias, float* y,
                                int32_t ldc, int32_t m, int32_t n, int32_t k, int32_t relu, int32_t splits,
                                float* partial, int32_t precision, void* stream) {
    GemmArgs a;
    int rc = fill_common(a, x, lda, w, ldw, m, n, k, precision, "eg_linear_splitk");
    if (rc) return rc;
    EG_REQUIRE(y && partial && splits > 0, EG_ERR_BAD_ARG, "eg_linear_splitk: null output/partial");
    a.bias = bias; a.res1 = a.res2 = nullptr; a.ldr = 0; a.y = y; a.ldc = ldc; a.relu = relu;
    a.a_shift = 0; a.a_seq = 1; a.partial = partial;
    return launch_splitk(a, splits, precision, (hipStream_t)stream);
}

// The extended product of the training path (see include/emogest.h: EgLinearArgs).
extern "C" int eg_linear_ex(const EgLinearArgs* p, void* stream) {
    EG_REQUIRE(p, EG_ERR_BAD_ARG, "eg_linear_ex: null argument block");
    GemmArgs a;
    int rc = fill_common(a, p->x_images ? reinterpret_cast<const float*>(p->x_images) : p->x, p->x_images ? 4 : p->lda, p->w, p->ldw, p->m, p->n, p->k, p->precision,
                         "eg_linear_ex");
    if (rc) return rc;
    if (p->x_images) a.x = nullptr;
    EG_REQUIRE(p->y, EG_ERR_BAD_ARG, "eg_linear_ex: null output");
    EG_REQUIRE(p->drop_p >= 0.f && p->drop_p < 1.f, EG_ERR_BAD_ARG, "eg_linear_ex: drop_p=%f", (double)p->drop_p);
    a.bias = p->bias; a.res1 = p->res1; a.res2 = p->res2; a.ldr = p->ldr; a.y = p->y; a.ldc = p->ldc; a.relu = p->relu;
    a.a_shift = 0; a.a_seq = 1; a.k_per_split = (int)eg_round_up(p->k, 64); a.partial = nullptr;
    a.gate = p->gate_src; a.ldg = p->ldg;
    if (p->drop_p > 0.f) {
        a.drop.thr = (unsigned int)((double)p->drop_p * 4294967296.0);
        a.drop.inv_keep = 1.0f / (1.0f - p->drop_p);
        a.drop.seed = p->drop_seed; a.drop.offset = p->drop_offset; a.drop.epoch = p->drop_epoch;
    }
    if (p->y_images) {                  // second output: Y as bf16 (hi, lo) tile-planar images of width y_k for a downstream pre-split product
        EG_REQUIRE(p->precision != EG_PREC_F32 && (p->n & 3) == 0 && (p->y_k & 63) == 0 && p->y_k >= p->n && p->splits < 2, EG_ERR_BAD_ARG,
                   "eg_linear_ex: image output needs a bf16 mode, n %% 4 == 0, y_k %% 64 == 0 and no split-K");
        a.yimg = reinterpret_cast<unsigned short*>(p->y_images); a.yKO = p->y_k >> 3; a.yoct0 = 0;
    }
    if (p->splits >= 2) {
        EG_REQUIRE(p->partial && !p->x_images, EG_ERR_BAD_ARG, "eg_linear_ex: split-K needs a partial buffer of splits*M*N floats (and fp32 X)");
        a.partial = p->partial;
        return launch_splitk(a, p->splits, p->precision, (hipStream_t)stream);
    }
    if (p->x_images) {                  // X arrives pre-split (the producing kernel's epilogue wrote the images): no in-kernel split, both operands by LDS-DMA
        EG_REQUIRE(p->precision != EG_PREC_F32 && (p->k_x & 63) == 0 && (p->k & 31) == 0 && p->k <= p->k_x, EG_ERR_BAD_ARG,
                   "eg_linear_ex: pre-split X needs a bf16 mode, k %% 32 == 0 and images of width k_x %% 64 == 0 >= k");
        const int xko = p->k_x >> 3, mt = eg_cdiv(p->m, 64);
        const bf8* xhi = reinterpret_cast<const bf8*>(p->x_images);
        const bf8* xlo = xhi + (size_t)mt * xko * 64;
        a.xoct0 = 0;
        EgProfScope prof(3, 2.0 * p->m * (double)p->n * p->k, (hipStream_t)stream);
        if (skinny_ok(a, p->precision)) return launch_skinny<false>(a, xhi, xlo, xko, (hipStream_t)stream);
        return dispatch_presplit(a, xhi, xlo, xko, p->precision, (hipStream_t)stream, &prof, 0);
    }
    EgProfScope prof(p->precision == EG_PREC_F32 ? 5 : 2, 2.0 * p->m * (double)p->n * p->k, (hipStream_t)stream);
    if (skinny_ok(a, p->precision)) return launch_skinny<true>(a, nullptr, nullptr, 0, (hipStream_t)stream);      // few rows (small training steps)
    return launch_gemm(a, 1, p->precision, (hipStream_t)stream);
}

// X -> pre-split tile-planar bf16 images (workspace-resident; consumed by eg_linear_presplit).  images must hold
// 2 * ceil(M/64)*64 * Kpad bf16 (hi image then lo image), Kpad = K rounded up to 64.
extern "C" int eg_split_tiles(const float* x, int32_t lda, int32_t m, int32_t k, void* images, void* stream) {
    EG_REQUIRE(x && images && m > 0 && k > 0 && (k & 3) == 0 && (lda & 3) == 0, EG_ERR_BAD_ARG, "eg_split_tiles: bad argument");
    const int kpad = (int)eg_round_up(k, 64), ko = kpad / 8, mt = eg_cdiv(m, 64);
    bf8* hi = reinterpret_cast<bf8*>(images);
    bf8* lo = hi + (size_t)mt * ko * 64;
    hipLaunchKernelGGL(split_tile_kernel, dim3(eg_cdiv(64 * ko, 256), mt), dim3(256), 0, (hipStream_t)stream, x, lda, m, k, ko, hi, lo);
    return eg_check_launch("split_tiles");
}

extern "C" int eg_linear_presplit(const void* x_images, int32_t k_x, const float* w, int32_t ldw, const float* bias,
                                  const float* res1, const float* res2, int32_t ldr, float* y, int32_t ldc,
                                  int32_t m, int32_t n, int32_t k, int32_t relu, int32_t precision, void* stream) {
    GemmArgs a;
    EG_REQUIRE(x_images && w && y && m > 0 && n > 0 && k > 0, EG_ERR_BAD_ARG, "eg_linear_presplit: null pointer or empty shape");
    EG_REQUIRE(precision == EG_PREC_BF16X3 || precision == EG_PREC_BF16, EG_ERR_BAD_ARG, "eg_linear_presplit: bf16 modes only");
    EG_REQUIRE((ldw & 63) == 0, EG_ERR_ALIGN, "eg_linear_presplit: ldw %% 64");
    a.x = nullptr; a.w = w; a.lda = 0; a.ldw = ldw; a.M = m; a.N = n; a.K = k;
    const size_t rows = (size_t)eg_round_up(n, 64);
    a.whi = reinterpret_cast<const unsigned short*>(w + rows * ldw);
    a.wlo = a.whi + rows * ldw;
    a.bias = bias; a.res1 = res1; a.res2 = res2; a.ldr = ldr; a.y = y; a.ldc = ldc; a.relu = relu;
    a.a_shift = 0; a.a_seq = 1; a.k_per_split = (int)eg_round_up(k, 64); a.partial = nullptr;
    const int xko = (int)eg_round_up(k_x, 64) / 8, mt = eg_cdiv(m, 64);
    const bf8* xhi = reinterpret_cast<const bf8*>(x_images);
    const bf8* xlo = xhi + (size_t)mt * xko * 64;
    EgProfScope prof(3, 2.0 * m * (double)n * k, (hipStream_t)stream);
    return dispatch_presplit(a, xhi, xlo, xko, precision, (hipStream_t)stream, &prof);
}
