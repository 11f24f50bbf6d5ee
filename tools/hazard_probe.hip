// Root-cause probe for the round-1 finding "attention results depend on co-resident work" (DESIGN.md §5).
//
// The victim is the round-1 attention kernel's P.V loop (Full_model/Modules.py:13-23, after softmax) in four variants that
// separate the two suspects -- (a) v_pk_fma_f32 with op_sel broadcast out of an LDS-loaded register pair, and (b) consuming
// ds_read_b128 results under progressive counted waits (s_waitcnt lgkmcnt(7), (6), ... while younger reads are in flight):
//   mode 0  float4 FMAs, unrolled x8: hipcc forms v_pk_fma_f32 + progressive lgkmcnt(N)            (the failing round-1 code shape)
//   mode 1  float4 FMAs, unrolled x8, but a full s_waitcnt lgkmcnt(0) in front of the first FMA    (packed ops, no partial waits)
//   mode 2  scalar v_fmac_f32 (inline asm), unrolled x8: progressive lgkmcnt(N), no packed op      (partial waits, no packed ops)
//   mode 3  scalar v_fmac_f32, not unrolled, lgkmcnt(0) per iteration                               (the round-1 fix)
//   mode 4  inline-asm v_pk_fma_f32 WITHOUT op_sel (the probability duplicated into a register pair), lgkmcnt(0) first
//   mode 5  inline-asm v_pk_fma_f32 WITH op_sel / op_sel_hi broadcast out of a pair, lgkmcnt(0) + s_nop 4 first
//   mode 6  inline-asm v_pk_mul_f32 (op_sel_hi broadcast) + v_pk_add_f32, lgkmcnt(0) first
// Aggressors: the library's 128->128 convolution (MFMA + LDS-DMA + LDS), plain LDS traffic, MFMA only (no memory), LDS-DMA only.
// Each mode is launched alone (quiet reference), then repeatedly while the aggressor -- the library's 128->128 MFMA convolution
// (LDS-DMA weight ring, 74 KB LDS, raw s_barrier) -- runs on a second stream; every output word is compared with the quiet run.
//
// build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/hazard_probe.hip -o gpurun_out/hazard_probe \
//               -L emotiongestures_amd -lemogest_hip -Wl,-rpath,$PWD/emotiongestures_amd
// run:    gpurun_out/hazard_probe [rounds=20]        (prints one line per mode: launches differing / launches)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "emogest.h"

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef short bf8 __attribute__((ext_vector_type(8)));
constexpr int P = 68;           // LDS row pitch in floats (round-1 kernel)

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void pv_kernel(const float* __restrict__ v, const float* __restrict__ p, float* __restrict__ out,
                                                 int Lq, int Lk) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int LkP = (Lk + 7) & ~7;
    float* Vs = sm;                 // [LkP][68]  (rows >= Lk zero)
    float* Ss = sm + LkP * P;       // [Lq][LkP]  (columns >= Lk zero)
    const int blk = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < LkP * 16; i += 256) {
        const int r = i >> 4, c = (i & 15) * 4;
        f4 t = (f4){0.f, 0.f, 0.f, 0.f};
        if (r < Lk) t = *reinterpret_cast<const f4*>(v + ((size_t)blk * Lk + r) * 64 + c);
        *reinterpret_cast<f4*>(Vs + r * P + c) = t;
    }
    for (int i = tid; i < Lq * LkP; i += 256) {
        const int r = i / LkP, c = i - r * LkP;
        Ss[i] = c < Lk ? p[((size_t)blk * Lq + r) * Lk + c] : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < Lq * 16; i += 256) {
        const int r = i >> 4, d4 = i & 15;
        const float* vp = Vs + d4 * 4;
        const float* pp = Ss + r * LkP;
        f4 o = (f4){0.f, 0.f, 0.f, 0.f};
        if (MODE == 0 || MODE == 1) {
            for (int c = 0; c < LkP; c += 8) {
                f4 vv[8];
                const f4 p0 = *reinterpret_cast<const f4*>(pp + c), p1 = *reinterpret_cast<const f4*>(pp + c + 4);
#pragma unroll
                for (int j = 0; j < 8; ++j) vv[j] = *reinterpret_cast<const f4*>(vp + (c + j) * P);
                if (MODE == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int j = 0; j < 8; ++j) o += vv[j] * (j < 4 ? p0[j & 3] : p1[j & 3]);
            }
        } else if (MODE == 2) {
            float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
            for (int c = 0; c < LkP; c += 8) {
                f4 vv[8];
                const f4 p0 = *reinterpret_cast<const f4*>(pp + c), p1 = *reinterpret_cast<const f4*>(pp + c + 4);
#pragma unroll
                for (int j = 0; j < 8; ++j) vv[j] = *reinterpret_cast<const f4*>(vp + (c + j) * P);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float pj = j < 4 ? p0[j & 3] : p1[j & 3];
                    asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(o0) : "v"(vv[j][0]), "v"(pj));
                    asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(o1) : "v"(vv[j][1]), "v"(pj));
                    asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(o2) : "v"(vv[j][2]), "v"(pj));
                    asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(o3) : "v"(vv[j][3]), "v"(pj));
                }
            }
            o = (f4){o0, o1, o2, o3};
        } else if (MODE >= 4) {
            f2 oa = (f2){0.f, 0.f}, ob = (f2){0.f, 0.f};
            for (int c = 0; c < LkP; c += 8) {
                f4 vv[8];
                const f4 p0 = *reinterpret_cast<const f4*>(pp + c), p1 = *reinterpret_cast<const f4*>(pp + c + 4);
#pragma unroll
                for (int j = 0; j < 8; ++j) vv[j] = *reinterpret_cast<const f4*>(vp + (c + j) * P);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const f2 va = (f2){vv[j][0], vv[j][1]}, vb = (f2){vv[j][2], vv[j][3]};
                    const float pj = j < 4 ? p0[j & 3] : p1[j & 3];
                    if (MODE == 4) {
                        const f2 pd = (f2){pj, pj};
                        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(oa) : "v"(va), "v"(pd));
                        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(ob) : "v"(vb), "v"(pd));
                    } else if (MODE == 5) {
                        // pair (p_j, p_j+1) as the LDS read delivered it; even j: both halves take the low word, odd j: the high word
                        const f2 pr = (j < 4) ? ((j & 2) ? (f2){p0[2], p0[3]} : (f2){p0[0], p0[1]}) : ((j & 2) ? (f2){p1[2], p1[3]} : (f2){p1[0], p1[1]});
                        if (j & 1) {
                            asm volatile("s_nop 4\n\tv_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(oa) : "v"(va), "v"(pr));
                            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(ob) : "v"(vb), "v"(pr));
                        } else {
                            asm volatile("s_nop 4\n\tv_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(oa) : "v"(va), "v"(pr));
                            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(ob) : "v"(vb), "v"(pr));
                        }
                    } else {
                        const f2 pd = (f2){pj, pj};
                        f2 ta, tb;
                        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(ta) : "v"(va), "v"(pd));
                        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(tb) : "v"(vb), "v"(pd));
                        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(oa) : "v"(ta));
                        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(ob) : "v"(tb));
                    }
                }
            }
            o = (f4){oa[0], oa[1], ob[0], ob[1]};
        } else {
            float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
#pragma unroll 1
            for (int c = 0; c < Lk; ++c) {
                const f4 vv = *reinterpret_cast<const f4*>(vp + c * P);
                const float pj = pp[c];
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(o0) : "v"(vv[0]), "v"(pj));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(o1) : "v"(vv[1]), "v"(pj));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(o2) : "v"(vv[2]), "v"(pj));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(o3) : "v"(vv[3]), "v"(pj));
            }
            o = (f4){o0, o1, o2, o3};
        }
        *reinterpret_cast<f4*>(out + ((size_t)blk * Lq + r) * 64 + d4 * 4) = o;
    }
}

// A second aggressor with no LDS-DMA and no MFMA: plain LDS read/write traffic, to tell "any LDS-heavy neighbour" from "MFMA + LDS-DMA".
__global__ __launch_bounds__(256) void lds_noise_kernel(float* sink, int iters) {
    __shared__ f4 buf[4096];        // 64 KB
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += 256) buf[i] = (f4){(float)i, 1.f, 2.f, 3.f};
    __syncthreads();
    f4 acc = (f4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll 8
        for (int j = 0; j < 16; ++j) acc += buf[(tid * 17 + j * 256 + it) & 4095];
        buf[(tid + it * 256) & 4095] = acc;
    }
    if (acc[0] == 123.456f) sink[0] = acc[1];
}

// MFMA only: no LDS, no memory traffic inside the loop
__global__ __launch_bounds__(256) void mfma_noise_kernel(float* sink, int iters) {
    f4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f4){0.f, 0.f, 0.f, 0.f};
    bf8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (short)(0x3C00 + threadIdx.x + j); b[j] = (short)(0x3D00 + j); }
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 123.456f) sink[0] = 1.f;
}
// LDS-DMA only (global_load_lds into a 64 KB LDS array), no MFMA
__global__ __launch_bounds__(256) void ldsdma_noise_kernel(const float* src, float* sink, int iters) {
    __shared__ f4 buf[4096];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < 8; ++p)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const f4*>(src) + ((it * 8 + p) & 1023) * 64 + lane),
                                             (__attribute__((address_space(3))) void*)(buf + (wave * 8 + p) * 64), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (buf[threadIdx.x][0] == 123.456f) sink[0] = 1.f;
}

typedef void (*PvKern)(const float*, const float*, float*, int, int);

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 20;
    const int nblk = 256, Lq = 34, Lk = 34;          // 256 (clip, head) pairs as in a B=32 batch
    const int LkP = (Lk + 7) & ~7;
    const size_t smem = sizeof(float) * ((size_t)LkP * P + (size_t)Lq * LkP);
    std::vector<float> hv((size_t)nblk * Lk * 64), hp((size_t)nblk * Lq * Lk);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 65536.0f; };
    for (auto& x : hv) x = rnd() * 2.f - 1.f;
    for (auto& x : hp) x = rnd() / Lk;
    float *dv, *dp, *dout, *dref;
    const size_t obytes = sizeof(float) * (size_t)nblk * Lq * 64;
    CK(hipMalloc(&dv, hv.size() * 4)); CK(hipMalloc(&dp, hp.size() * 4)); CK(hipMalloc(&dout, obytes * 40)); CK(hipMalloc(&dref, obytes));
    CK(hipMemcpy(dv, hv.data(), hv.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dp, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
    // aggressor A: the library's 128->128 bf16x3 convolution, B=16 at 32x31
    const int CB = 16, CH = 32, CW = 31, CC = 128;
    float *cx, *cy, *cw;
    const size_t xs = (size_t)CB * CH * CW * CC;
    const int64_t wfl = eg_conv3x3_packed_floats(CC, CC);
    CK(hipMalloc(&cx, xs * 4)); CK(hipMalloc(&cy, xs * 4)); CK(hipMalloc(&cw, (size_t)wfl * 4));
    {
        std::vector<float> hx(xs);
        for (auto& x : hx) x = rnd() - 0.5f;
        CK(hipMemcpy(cx, hx.data(), xs * 4, hipMemcpyHostToDevice));
        std::vector<unsigned short> hw((size_t)wfl * 2, (unsigned short)0x3C23);       // ~0.01 as bf16 (and ~0.00996 as fp32 pairs)
        CK(hipMemcpy(cw, hw.data(), (size_t)wfl * 4, hipMemcpyHostToDevice));
    }
    float* sink;
    CK(hipMalloc(&sink, 64));
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    PvKern kerns[7] = {pv_kernel<0>, pv_kernel<1>, pv_kernel<2>, pv_kernel<3>, pv_kernel<4>, pv_kernel<5>, pv_kernel<6>};
    std::vector<float> href(obytes / 4), hout(obytes / 4);
    const char* anames[4] = {"conv128_mfma_ldsdma", "plain_lds_traffic", "mfma_only", "ldsdma_only"};
    for (int aggr = 0; aggr < 4; ++aggr) {
        for (int mode = 0; mode < 7; ++mode) {
            hipLaunchKernelGGL(kerns[mode], dim3(nblk), dim3(256), smem, sa, dv, dp, dref, Lq, Lk);
            CK(hipStreamSynchronize(sa));
            CK(hipMemcpy(href.data(), dref, obytes, hipMemcpyDeviceToHost));
            int bad = 0, total = 0;
            long words_bad = 0;
            for (int rep = 0; rep < rounds; ++rep) {
                for (int i = 0; i < 120; ++i) {
                    if (aggr == 0) {
                        if (eg_conv3x3(cx, cw, nullptr, nullptr, nullptr, cy, nullptr, CB, CH, CW, CC, CC, 1, 1, 0, EG_PREC_BF16X3, sb) != EG_OK) {
                            fprintf(stderr, "conv: %s\n", eg_last_error());
                            return 2;
                        }
                    } else if (aggr == 1) {
                        hipLaunchKernelGGL(lds_noise_kernel, dim3(512), dim3(256), 0, sb, sink, 200);
                    } else if (aggr == 2) {
                        hipLaunchKernelGGL(mfma_noise_kernel, dim3(512), dim3(256), 0, sb, sink, 2000);
                    } else {
                        hipLaunchKernelGGL(ldsdma_noise_kernel, dim3(512), dim3(256), 0, sb, cx, sink, 200);
                    }
                }
                for (int i = 0; i < 40; ++i)
                    hipLaunchKernelGGL(kerns[mode], dim3(nblk), dim3(256), smem, sa, dv, dp, dout + (size_t)i * (obytes / 4), Lq, Lk);
                CK(hipDeviceSynchronize());
                for (int i = 0; i < 40; ++i) {
                    CK(hipMemcpy(hout.data(), dout + (size_t)i * (obytes / 4), obytes, hipMemcpyDeviceToHost));
                    long wb = 0;
                    for (size_t j = 0; j < hout.size(); ++j) wb += memcmp(&hout[j], &href[j], 4) != 0;
                    bad += wb != 0;
                    words_bad += wb;
                    ++total;
                }
            }
            printf("aggressor=%s mode=%d: %d/%d launches differ from the quiet run (%ld words)\n", anames[aggr],
                   mode, bad, total, words_bad);
            fflush(stdout);
        }
    }
    return 0;
}
