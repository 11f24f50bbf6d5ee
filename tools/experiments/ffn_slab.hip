// PositionwiseFeedForward (Full_model/SubLayers.py:74-84) of the inference transformer as ONE clip-slab kernel in the split-bf16 arithmetic:
//   pre = w_2(relu(w_1 x + b_1)) + b_2 + x   on a 64-row slab per workgroup; the d_inner-wide hidden is produced and consumed in 128-column chunks
//   through LDS and never written to memory (two pre-split GEMM launches write and re-read 17.8 MB of hidden images per FFN at 2176 rows).
// LayerNorm stays the existing kernel (it needs whole rows and emits the next block's images).
//
//   workgroup = 256 threads (4 waves, one per SIMD), one 64-row tile of the X images; grid = ceil(rows / 64).
//   for chunk c (128 hidden columns):
//     phase 1, 16 steps of K = 32:  Hc[64 x 128] += X[64 x 32] . W1[c*128.., 32]^T     X step 8 KB + W1 step 16 KB by LDS-DMA into a 4-slot ring
//        a wave owns 64 rows x 32 hidden columns (4 x 2 MFMA tiles x 3 terms = 24 MFMA 16x16x32 per step); + b_1, ReLU, split -> Hc (32 KB of LDS)
//     phase 2, 16 steps (4 output blocks of 128 columns x 4 K-steps of 32):  Y[64 x 512] += Hc[64 x 32] . W2[j*128.., c*128 + i*32..]^T  (W2 step 16 KB)
//        Y lives in 128 accumulator registers per lane for the whole kernel
//   epilogue: + b_2 + x (fp32 residual) -> pre.
// Every output element accumulates its products in ascending k with the same three terms per 32-deep step as gemm_presplit_kernel, and the hidden is
// split by the same function as the GEMM epilogue's image output: the result is BITWISE the two-launch result (tools/ffn_fused_probe.hip checks it;
// tests/test_gpu_kernels.py holds the product path to it).  Measured (profiles/r05_ffn_fused_probe.txt, 2176 rows): 177 us on 34 CUs = 6 030 CU.us
// against 12 117 CU.us for the two launches with the 128 x 128 tile (14 692 with 64 x 64): half the CU time, at 3 x the stand-alone latency --
// which is what the step pays for with several batches in flight.  d_model is fixed at 512 (the accumulator layout); d_inner % 128 == 0.
#include "common.h"
#include <stdlib.h>

namespace {
constexpr int D = 512, HC = 128, RING = 4;
constexpr int KO1 = D / 8;                                      // octets per row of the X / W1 images
constexpr int HCS = 2 * (HC / 8) * 64;                          // Hc: [img][octet 16][row 64] bf8 slots
constexpr int XS = 2 * 4 * 64, WS = 2 * 2 * 4 * 64, SLOT = XS + WS;      // ring slot: X [img][octet 4][row 64] | W [img][tile 2][octet 4][row 64]
constexpr size_t LDS_BYTES = (size_t)(HCS + RING * SLOT) * 16;

struct FfnArgs {
    const bf8* xhi; const bf8* xlo;             // X images [rows/64][KO1][64]
    const bf8* w1hi; const bf8* w1lo;           // W1 images [H/64][KO1][64]
    const bf8* w2hi; const bf8* w2lo;           // W2 images [D/64][KO2][64]
    const float* b1; const float* b2; const float* x; float* pre; int rows, nch, ko2, ldx, ldp;
    int cps;                                    // hidden chunks per workgroup (blockIdx.y selects the chunk range); < nch: pre receives raw partial sums [split][rows][ldp]
};

__global__ __launch_bounds__(256, 1) void ffn_slab_kernel(FfnArgs a) {
    const int KO2 = a.ko2, NSTEPS = a.cps * 32, c0 = blockIdx.y * a.cps;
    extern __shared__ __attribute__((aligned(16))) bf8 lds[];
    bf8* const Hc = lds;
    bf8* const ring = lds + HCS;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mt = blockIdx.x;
    const int wt = wave >> 1, wr = (wave & 1) * 32;             // this wave's weight rows inside a step's 128: image tile wt, rows wr..wr+32

    // ---- the operand stream: global step g = c*32 + s;  s < 16: phase 1 (K-step s of chunk c);  s >= 16: phase 2, q = s - 16: block j = q>>2, K-step i = q&3
    auto issue = [&](int g) {
        const int c = c0 + (g >> 5), s = g & 31;
        bf8* S = ring + (g % RING) * SLOT;
        const bf8 *whi, *wlo;
        size_t wbase;
        if (s < 16) {
            const size_t gx = ((size_t)mt * KO1 + s * 4 + wave) * 64 + lane;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.xhi + gx), (__attribute__((address_space(3))) void*)(S + wave * 64), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.xlo + gx), (__attribute__((address_space(3))) void*)(S + 256 + wave * 64), 16, 0, 0);
            whi = a.w1hi; wlo = a.w1lo;
            wbase = ((size_t)(2 * c) * KO1 + s * 4 + wave) * 64 + lane;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(whi + wbase + (size_t)t * KO1 * 64),
                                                 (__attribute__((address_space(3))) void*)(S + XS + (t * 4 + wave) * 64), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wlo + wbase + (size_t)t * KO1 * 64),
                                                 (__attribute__((address_space(3))) void*)(S + XS + 512 + (t * 4 + wave) * 64), 16, 0, 0);
            }
        } else {
            const int q = s - 16, j = q >> 2, i = q & 3;
            whi = a.w2hi; wlo = a.w2lo;
            wbase = ((size_t)(2 * j) * KO2 + c * 16 + i * 4 + wave) * 64 + lane;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(whi + wbase + (size_t)t * KO2 * 64),
                                                 (__attribute__((address_space(3))) void*)(S + XS + (t * 4 + wave) * 64), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wlo + wbase + (size_t)t * KO2 * 64),
                                                 (__attribute__((address_space(3))) void*)(S + XS + 512 + (t * 4 + wave) * 64), 16, 0, 0);
            }
        }
    };
    struct Frags { bf8 bh[4], bl[4], wh[2], wl[2]; };
    auto read_w = [&](Frags& f, int g) {
        const bf8* W = ring + (g % RING) * SLOT + XS + (wt * 4 + kq) * 64 + wr + li;
#pragma unroll
        for (int n = 0; n < 2; ++n) { f.wh[n] = W[n * 16]; f.wl[n] = W[512 + n * 16]; }
    };
    auto read_b = [&](Frags& f, int g) {
        const int s = g & 31;
        const bf8* B = (s < 16) ? ring + (g % RING) * SLOT + kq * 64 + li : Hc + (((s - 16) & 3) * 4 + kq) * 64 + li;
        const int lo_off = (s < 16) ? 256 : (HC / 8) * 64;
#pragma unroll
        for (int t = 0; t < 4; ++t) { f.bh[t] = B[t * 16]; f.bl[t] = B[lo_off + t * 16]; }
    };
    f4 yacc[4][4][2], hacc[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int n = 0; n < 2; ++n) yacc[j][t][n] = (f4){0.f, 0.f, 0.f, 0.f};
    auto mfma_half = [&](const Frags& f, f4 (&acc)[4][2], int half) {
#pragma unroll
        for (int t = half * 2; t < half * 2 + 2; ++t)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.wl[n], f.bh[t], acc[t][n], 0, 0, 0);
                acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.wh[n], f.bl[t], acc[t][n], 0, 0, 0);
                acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.wh[n], f.bh[t], acc[t][n], 0, 0, 0);
            }
    };
    auto mfma_step_half = [&](const Frags& f, int s, int half) {
        if (s < 16) mfma_half(f, hacc, half);
        else {
            switch ((s - 16) >> 2) {                        // static after unrolling by 32
                case 0: mfma_half(f, yacc[0], half); break;
                case 1: mfma_half(f, yacc[1], half); break;
                case 2: mfma_half(f, yacc[2], half); break;
                default: mfma_half(f, yacc[3], half); break;
            }
        }
    };
    for (int g = 0; g < RING - 1; ++g) issue(g);
    wait_vmcnt_imm<(RING - 2) * 6>();           // group 0 landed (the younger groups, six copies each, may still fly)
    wg_barrier();
    Frags fa, fb;
    read_w(fa, 0);
    read_b(fa, 0);
    wait_lgkmcnt0();
    // one step, the schedule of the convolution / 128 x 128 GEMM kernels: issue the copy RING-1 steps ahead; first half of the MFMAs; counted wait +
    // the step's one barrier (they complete under the first half's MFMAs still in the pipe); read the NEXT step's fragments; second half
    auto step = [&](int g, Frags& cur, Frags& nxt) {
        const int s = g & 31;
        if (g + RING - 1 < NSTEPS) issue(g + RING - 1);
        if (s == 0) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int n = 0; n < 2; ++n) hacc[t][n] = (f4){0.f, 0.f, 0.f, 0.f};
        }
        const bool more = g + 1 < NSTEPS;
        mfma_step_half(cur, s, 0);
        __builtin_amdgcn_sched_barrier(0);
        // step g+1's copies must have landed; younger groups (>= 4 copies each) may stay in flight -- RING-2 of them in the steady state, fewer at the
        // very end, where no copy is issued any more (a count that assumes the steady state would let the last groups through unlanded)
        if (g + RING - 1 < NSTEPS) wait_vmcnt_imm<(RING - 2) * 4>();
        else if (g + 2 < NSTEPS) wait_vmcnt_imm<4>();
        else wait_vmcnt_imm<0>();
        wait_lgkmcnt0();
        wg_barrier();
        if (more) {
            read_w(nxt, g + 1);
            if (s != 15) read_b(nxt, g + 1);                // the first phase-2 step reads Hc, which this step is about to write
        }
        mfma_step_half(cur, s, 1);
        if (s == 15) {                                      // hidden chunk: + b_1, ReLU, split -> Hc (the layout of a B operand: [octet][row])
            const int c = c0 + (g >> 5);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int col = wave * 32 + n * 16 + kq * 4;
                    const f4 b = *reinterpret_cast<const f4*>(a.b1 + c * HC + col);
                    f4 v = hacc[t][n];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r] + b[r], 0.f);
                    const f4 z = (f4){0.f, 0.f, 0.f, 0.f};
                    bf8 h8, l8;
                    split_octet<true>(v, z, h8, l8);
                    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                    const u32x4_t hh = __builtin_bit_cast(u32x4_t, h8), ll = __builtin_bit_cast(u32x4_t, l8);
                    unsigned short* dst = reinterpret_cast<unsigned short*>(Hc) + (((size_t)(col >> 3) * 64 + t * 16 + li) * 8 + (col & 7));
                    *reinterpret_cast<u32x2*>(dst) = (u32x2){hh[0], hh[1]};
                    *reinterpret_cast<u32x2*>(dst + (HC / 8) * 64 * 8) = (u32x2){ll[0], ll[1]};
                }
            wait_lgkmcnt0();
            wg_barrier();                                   // one extra barrier per chunk: Hc complete before its first fragments are read
            if (more) { read_b(nxt, g + 1); wait_lgkmcnt0(); }
        }
    };
#pragma unroll 1
    for (int c = 0; c < a.cps; ++c) {
#pragma unroll
        for (int s = 0; s < 32; s += 2) {
            step(c * 32 + s, fa, fb);
            step(c * 32 + s + 1, fb, fa);
        }
    }
    // ---- epilogue: the whole hidden in this workgroup: + b_2 + x -> pre;  a chunk range: the raw partial sums (egi_layernorm_sum folds them in order)
    const bool whole = a.cps == a.nch;
    float* dstbase = whole ? a.pre : a.pre + (size_t)blockIdx.y * a.rows * a.ldp;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int m = mt * 64 + t * 16 + li;
            if (m >= a.rows) continue;
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int col = j * 128 + wave * 32 + n * 16 + kq * 4;
                f4 v = yacc[j][t][n];
                if (whole) {
                    const f4 b = *reinterpret_cast<const f4*>(a.b2 + col);
                    const f4 r = *reinterpret_cast<const f4*>(a.x + (size_t)m * a.ldx + col);
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = v[q] + b[q] + r[q];
                }
                *reinterpret_cast<f4*>(dstbase + (size_t)m * a.ldp + col) = v;
            }
        }
}

}  // namespace

// EG_FFN_FUSED (read per call, like EG_GEMM_TILE; a captured graph keeps what it was captured with): unset / "0" = two pre-split launches (the
// default: the slab kernel halves the FFN's CU time but does not shorten the step -- DESIGN.md §10, profiles/r05_ffn_fused_probe.txt), "1" = one
// workgroup per slab (bitwise the two launches), "2" / "4" = the hidden split over that many workgroups per slab (partial sums folded by the LayerNorm).
static int ffn_mode() {
    const char* e = getenv("EG_FFN_FUSED");
    return (e && e[0]) ? atoi(e) : 0;
}
int egi_ffn_fused_splits(int d_model, int d_inner, int precision, int rows) {
    const int mode = ffn_mode();
    if (mode <= 0 || precision != EG_PREC_BF16X3 || d_model != D || d_inner < HC || d_inner % HC) return 0;
    if (rows < 64) return 0;            // a single clip (34 rows): one slab of mostly padding
    int sp = mode;
    const int nch = d_inner / HC;
    while (sp > 1 && (nch % sp || (long)sp * D > d_inner)) sp >>= 1;            // chunk ranges of equal size; the partials fit the (unused) fp32 hidden buffer
    return sp < 1 ? 1 : sp;
}

// x_images: the bf16 (hi, lo) tile-planar images of X [rows, 512] (hi image, then lo);  x: the same rows as fp32 (the residual), row stride ldx;
// w1 / w2: EG_PACK_LINEAR images of w_1 [d_inner, 512] and w_2 [512, d_inner].  splits == 1: pre [rows, ldp] = w_2(relu(w_1 x + b_1)) + b_2 + x;
// splits > 1: pre [splits][rows][ldp] receives the partial sums over each workgroup's hidden-chunk range (no bias, no residual).
int egi_ffn_fused(const void* x_images, const float* x, int ldx, const float* w1, const float* b1, const float* w2, const float* b2, float* pre, int ldp,
                  int rows, int d_inner, int splits, hipStream_t st) {
    EG_REQUIRE(x_images && x && w1 && b1 && w2 && b2 && pre && rows > 0, EG_ERR_BAD_ARG, "egi_ffn_fused: null pointer or empty shape");
    EG_REQUIRE(d_inner >= HC && d_inner % HC == 0 && (ldx & 3) == 0 && (ldp & 3) == 0, EG_ERR_UNSUPPORTED, "egi_ffn_fused: d_inner=%d ldx=%d ldp=%d", d_inner, ldx, ldp);
    const int nch = d_inner / HC;
    EG_REQUIRE(splits >= 1 && nch % splits == 0, EG_ERR_BAD_ARG, "egi_ffn_fused: %d hidden chunks do not split %d ways", nch, splits);
    const int mt = eg_cdiv(rows, 64);
    FfnArgs a;
    a.xhi = reinterpret_cast<const bf8*>(x_images);
    a.xlo = a.xhi + (size_t)mt * KO1 * 64;
    const size_t r1 = (size_t)d_inner * D, r2 = (size_t)D * d_inner;            // fp32 part of a packed weight [rows][ldw]; then the hi and the lo image
    a.w1hi = reinterpret_cast<const bf8*>(w1 + r1); a.w1lo = a.w1hi + r1 / 8;
    a.w2hi = reinterpret_cast<const bf8*>(w2 + r2); a.w2lo = a.w2hi + r2 / 8;
    a.b1 = b1; a.b2 = b2; a.x = x; a.pre = pre; a.rows = rows; a.nch = nch; a.ko2 = d_inner / 8; a.ldx = ldx; a.ldp = ldp; a.cps = nch / splits;
    if (int rc = eg_ensure_dynamic_lds(reinterpret_cast<const void*>(ffn_slab_kernel), LDS_BYTES, "ffn_slab")) return rc;
    EgProfScope prof(7, 4.0 * rows * (double)D * d_inner, st);
    prof.workgroups(mt * splits);
    hipLaunchKernelGGL(ffn_slab_kernel, dim3(mt, splits), dim3(256), LDS_BYTES, st, a);
    return eg_check_launch("ffn_slab");
}
