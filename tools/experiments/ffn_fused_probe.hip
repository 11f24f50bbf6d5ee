// Go / no-go probe (round-4 verdict item 3): PositionwiseFeedForward (Full_model/SubLayers.py:74-84) of the INFERENCE transformer as ONE clip-slab
// kernel instead of two pre-split GEMM launches -- pre = w_2(relu(w_1 x + b_1)) + b_2 + x on a 64-row slab per workgroup, the 2048-wide hidden
// produced and consumed in 128-column chunks through LDS, never written to memory.  (LayerNorm stays the existing kernel: it needs the whole row.)
//
//   workgroup = 256 threads (4 waves, one per SIMD), one 64-row slab of X (a 64-row tile of the bf16 (hi, lo) tile-planar images the producing
//   LayerNorm already emits), grid = rows / 64 (34 workgroups at the headline's 2176 rows).
//   for chunk c = 0..15 (128 hidden columns):
//     phase 1, 16 steps of K = 32:   Hc[64 x 128] += X[64 x 32] . W1[c*128.., 32]^T       X step 8 KB + W1 step 16 KB by LDS-DMA into a 4-slot ring
//        a wave owns 64 rows x 32 hidden columns: 4 x 2 MFMA tiles x 3 terms = 24 MFMA 16x16x32 per step; + b_1, ReLU, split to (hi, lo) -> Hc in LDS (32 KB)
//     phase 2, 16 steps (4 output blocks of 128 columns x 4 K-steps of 32):   Y[64 x 512] += Hc[64 x 32] . W2[j*128.., c*128 + i*32..]^T   W2 step 16 KB
//        a wave owns 64 rows x 32 columns of each block: Y lives in 128 accumulator registers for the whole kernel
//   epilogue: + b_2 + x (fp32 residual) -> pre [rows, 512] fp32.
//   Every output element accumulates its K = 512 / K = 2048 products in ascending k with the same three terms per 32-deep step as the pre-split GEMMs:
//   the result must be BITWISE the two-launch result (checked below against eg_linear_presplit x 2).
//
// Judged by CU time per FFN (workgroups x kernel time vs the two launches' duration x their share of the 256 CUs), not by stand-alone latency.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Xclang -target-feature -Xclang -packed-fp32-ops -I include -I emotiongestures_amd/csrc \
//        tools/ffn_fused_probe.hip -L emotiongestures_amd -lemogest_hip -Wl,-rpath,'$ORIGIN/../emotiongestures_amd' -o tools/ffn_fused_probe
#include "common.h"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {
#ifndef PROBE_RING
#define PROBE_RING 4
#endif
constexpr int D = 512, H = 2048, HC = 128, NCH = H / HC, RING = PROBE_RING;
constexpr int KO1 = D / 8, KO2 = H / 8;                         // octets per row of the W1 / W2 (and X / hidden) images
constexpr int HCS = 2 * (HC / 8) * 64;                          // Hc: [img][octet 16][row 64] bf8 slots
constexpr int XS = 2 * 4 * 64, WS = 2 * 2 * 4 * 64, SLOT = XS + WS;      // ring slot: X [img][octet 4][row 64] | W [img][tile 2][octet 4][row 64]
constexpr size_t LDS_BYTES = (size_t)(HCS + RING * SLOT) * 16;

struct FfnArgs {
    const bf8* xhi; const bf8* xlo;             // X images [rows/64][KO1][64]
    const bf8* w1hi; const bf8* w1lo;           // W1 images [H/64][KO1][64]
    const bf8* w2hi; const bf8* w2lo;           // W2 images [D/64][KO2][64]
    const float* b1; const float* b2; const float* x; float* pre; int rows;
};

__global__ __launch_bounds__(256, 1) void ffn_fused_kernel(FfnArgs a) {
    extern __shared__ __attribute__((aligned(16))) bf8 lds[];
    bf8* const Hc = lds;
    bf8* const ring = lds + HCS;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mt = blockIdx.x;
    const int wt = wave >> 1, wr = (wave & 1) * 32;             // this wave's weight rows inside a step's 128: image tile wt, rows wr..wr+32

    // ---- the operand stream: global step g = c*32 + s;  s < 16: phase 1 (K-step s of chunk c);  s >= 16: phase 2, q = s - 16: block j = q>>2, K-step i = q&3
    auto issue = [&](int g) {
        const int c = g >> 5, s = g & 31;
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
    constexpr int NSTEPS = NCH * 32;
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
            const int c = g >> 5;
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
    for (int c = 0; c < NCH; ++c) {
#pragma unroll
        for (int s = 0; s < 32; s += 2) {
            step(c * 32 + s, fa, fb);
            step(c * 32 + s + 1, fb, fa);
        }
    }
    // ---- epilogue: + b_2 + x -> pre
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int m = mt * 64 + t * 16 + li;
            if (m >= a.rows) continue;
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int col = j * 128 + wave * 32 + n * 16 + kq * 4;
                const f4 b = *reinterpret_cast<const f4*>(a.b2 + col);
                const f4 r = *reinterpret_cast<const f4*>(a.x + (size_t)m * D + col);
                f4 v = yacc[j][t][n];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = v[q] + b[q] + r[q];
                *reinterpret_cast<f4*>(a.pre + (size_t)m * D + col) = v;
            }
        }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define EG(x) do { int r_ = (x); if (r_) { printf("library call failed (%d) at line %d: %s\n", r_, __LINE__, eg_last_error()); exit(1); } } while (0)

float unit(uint32_t i, uint32_t seed) {            // deterministic pseudo-random in [-1, 1)
    uint32_t h = i * 0x9E3779B9u + seed * 0x85EBCA6Bu;
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    return (float)(h >> 8) / 8388608.0f - 1.0f;
}
}  // namespace

int main(int argc, char** argv) {
    const int rows = argc > 1 ? atoi(argv[1]) : 2176, reps = 50;
    const int mt = (rows + 63) / 64;
    std::vector<float> hx((size_t)rows * D), hw1((size_t)H * D), hb1(H), hw2((size_t)D * H), hb2(D);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = unit((uint32_t)i, 1);
    for (size_t i = 0; i < hw1.size(); ++i) hw1[i] = 0.05f * unit((uint32_t)i, 2);
    for (size_t i = 0; i < hw2.size(); ++i) hw2[i] = 0.03f * unit((uint32_t)i, 3);
    for (int i = 0; i < H; ++i) hb1[i] = 0.1f * unit(i, 4);
    for (int i = 0; i < D; ++i) hb2[i] = 0.1f * unit(i, 5);
    float *x, *w1, *w2, *b1, *b2, *w1img, *w2img, *h, *ref, *pre;
    void *ximg, *himg;
    CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&w1, hw1.size() * 4)); CK(hipMalloc(&w2, hw2.size() * 4)); CK(hipMalloc(&b1, H * 4)); CK(hipMalloc(&b2, D * 4));
    CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(w1, hw1.data(), hw1.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(w2, hw2.data(), hw2.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(b1, hb1.data(), H * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(b2, hb2.data(), D * 4, hipMemcpyHostToDevice));
    const size_t f1 = (size_t)eg_linear_packed_floats(H, D), f2 = (size_t)eg_linear_packed_floats(D, H);
    CK(hipMalloc(&w1img, f1 * 4)); CK(hipMalloc(&w2img, f2 * 4));
    CK(hipMalloc(&ximg, (size_t)4 * mt * 64 * D)); CK(hipMalloc(&himg, (size_t)4 * mt * 64 * H));
    CK(hipMalloc(&h, (size_t)rows * H * 4)); CK(hipMalloc(&ref, (size_t)rows * D * 4)); CK(hipMalloc(&pre, (size_t)rows * D * 4));
    EG(eg_pack_linear_device(w1, D, H, D, 0, w1img, nullptr));
    EG(eg_pack_linear_device(w2, H, D, H, 0, w2img, nullptr));
    EG(eg_split_tiles(x, D, rows, D, ximg, nullptr));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timed = [&](auto&& fn) {
        for (int i = 0; i < 5; ++i) fn();
        CK(hipEventRecord(e0, nullptr));
        for (int i = 0; i < reps; ++i) fn();
        CK(hipEventRecord(e1, nullptr));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        return ms * 1e3f / reps;
    };
    // ---- today: two pre-split launches (the hidden goes through memory; its split pass is what the GEMM epilogue does in the product path: not timed)
    printf("rows %d: FFN 512 -> 2048 -> 512, bf16x3\n", rows);
    for (const char* tile : {"64", "128"}) {
        setenv("EG_GEMM_TILE", tile, 1);
        const float t1 = timed([&] { EG(eg_linear_presplit(ximg, D, w1img, D, b1, nullptr, nullptr, 0, h, H, rows, H, D, 1, EG_PREC_BF16X3, nullptr)); });
        EG(eg_split_tiles(h, H, rows, H, himg, nullptr));
        const float t2 = timed([&] { EG(eg_linear_presplit(himg, H, w2img, H, b2, x, nullptr, D, ref, D, rows, D, H, 0, EG_PREC_BF16X3, nullptr)); });
        const int tw = strcmp(tile, "64") ? 128 : 64;
        const int wg1 = ((rows + tw - 1) / tw) * (H / tw), wg2 = ((rows + tw - 1) / tw) * (D / tw);
        const float cu1 = t1 * (wg1 < 256 ? wg1 : 256), cu2 = t2 * (wg2 < 256 ? wg2 : 256);
        printf("  two launches, %3s x %-3s tile: %7.2f + %7.2f us stand-alone = %7.2f us;  CU time %8.0f + %8.0f = %8.0f CU.us (%d / %d workgroups)\n", tile, tile, t1, t2,
               t1 + t2, cu1, cu2, cu1 + cu2, wg1, wg2);
    }
    unsetenv("EG_GEMM_TILE");
    // ---- fused slab kernel
    FfnArgs a;
    const size_t xi = (size_t)mt * KO1 * 64;
    a.xhi = reinterpret_cast<const bf8*>(ximg); a.xlo = a.xhi + xi;
    const size_t r1 = (size_t)H * D, r2 = (size_t)D * H;                // fp32 part of a packed weight: [rows][ldw], then the hi and lo images
    a.w1hi = reinterpret_cast<const bf8*>(w1img + r1); a.w1lo = a.w1hi + r1 / 8;
    a.w2hi = reinterpret_cast<const bf8*>(w2img + r2); a.w2lo = a.w2hi + r2 / 8;
    a.b1 = b1; a.b2 = b2; a.x = x; a.pre = pre; a.rows = rows;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(ffn_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
    CK(hipMemset(pre, 0, (size_t)rows * D * 4));
    const float tf = timed([&] { hipLaunchKernelGGL(ffn_fused_kernel, dim3(mt), dim3(256), LDS_BYTES, nullptr, a); });
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    std::vector<float> got((size_t)rows * D), want((size_t)rows * D);
    CK(hipMemcpy(got.data(), pre, got.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(want.data(), ref, want.size() * 4, hipMemcpyDeviceToHost));
    size_t diff = 0;
    double maxd = 0.0, nrm = 0.0;
    for (size_t i = 0; i < got.size(); ++i) {
        if (memcmp(&got[i], &want[i], 4)) ++diff;
        const double d = fabs((double)got[i] - want[i]);
        if (d > maxd) maxd = d;
        nrm += (double)want[i] * want[i];
    }
    printf("  fused slab kernel          : %7.2f us stand-alone, %d workgroups (one per CU, %zu KB LDS);  CU time %8.0f CU.us\n", tf, mt, LDS_BYTES >> 10, tf * mt);
    printf("  agreement with the two-launch result: %zu of %zu elements differ bitwise, max |d| %.3g (rms of the reference %.3g)\n", diff, got.size(), maxd,
           sqrt(nrm / got.size()));
    const double flop = 2.0 * 2.0 * rows * (double)D * H;
    printf("  algorithmic rate of the fused kernel: %.1f TFLOP/s on %d CUs = %.2f TFLOP/s per CU (bf16x3 ceiling 3.26 per CU)\n", flop / tf * 1e-6, mt, flop / tf * 1e-6 / mt);
    return 0;
}
