// LDS-fill probe (MI355X): how fast can one CU stream L2-resident operand panels into LDS with LDS-DMA (global_load_lds, 16 B per lane = 1 KiB per
// wave-instruction), alone and with MFMAs issued beside it?  This is the operand stream of the pre-split GEMM kernels (csrc/gemm.hip):
//   64 x 64 tile:   4 waves x 4 pieces = 16 KiB per step for 12 MFMAs per wave (bf16x3)
//   128 x 64 tile:  4 waves x 6 pieces = 24 KiB per step for 24 MFMAs per wave
//   128 x 128 tile: 8 waves x 4 pieces = 32 KiB per step for 24 MFMAs per wave
// Build: hipcc --offload-arch=gfx950 -O3 tools/ldsfill_probe.hip -o tools/ldsfill_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

template <int N> __device__ __forceinline__ void wait_vmcnt_imm() { __builtin_amdgcn_s_waitcnt((N & 15) | 0x70 | 0xF00 | ((N >> 4) << 14)); }
__device__ __forceinline__ void wg_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// WAVES waves, PIECES 1-KiB LDS-DMA pieces per wave and step, RING slots, MFMAS MFMAs per wave and step (register operands), LREADS ds_read_b128 per
// wave and step (from the slot that landed).  SHARED: every workgroup reads the same panel (a weight panel), else a panel of its own (activations).
template <int WAVES, int PIECES, int RING, int MFMAS, int LREADS>
__global__ __launch_bounds__(WAVES * 64) void fill_kernel(const bf8* __restrict__ src, size_t panel_bf8, int shared, int nsteps, float* sink) {
    extern __shared__ __attribute__((aligned(16))) bf8 lds[];
    constexpr int STEP = WAVES * PIECES * 64;                  // bf8 slots per step
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t base = shared ? 0 : ((size_t)blockIdx.x * 8 * STEP) % panel_bf8;
    f4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    bf8 fa = src[lane], fb = src[64 + lane];
    auto issue = [&](int s) {
        bf8* S = lds + (s % RING) * STEP + wave * PIECES * 64;
        size_t o = (base + (size_t)s * STEP + wave * PIECES * 64) % panel_bf8 + lane;
#pragma unroll
        for (int p = 0; p < PIECES; ++p)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + o + p * 64), (__attribute__((address_space(3))) void*)(S + p * 64), 16, 0, 0);
    };
    for (int s = 0; s < RING - 1; ++s) issue(s);
#pragma unroll 1
    for (int s = 0; s < nsteps; ++s) {
        issue(s + RING - 1);
        wait_vmcnt_imm<(RING - 1) * PIECES>();                 // group s landed
        wg_barrier();
        if (LREADS > 0) {
            const bf8* S = lds + (s % RING) * STEP + lane;
#pragma unroll
            for (int r = 0; r < LREADS; ++r) {
                bf8 v = S[((r * 4 + wave) % (WAVES * PIECES)) * 64];
                if (r & 1) fa = v; else fb = v;
            }
        }
#pragma unroll
        for (int m = 0; m < MFMAS; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[m & 3], 0, 0, 0);
        wg_barrier();                                          // slot s may be overwritten by the copy issued at step s+1 only after every wave read it
    }
    wait_vmcnt_imm<0>();
    f4 t = acc[0] + acc[1] + acc[2] + acc[3];
    if (t[0] == 123.456f) sink[0] = t[1];
}

template <int WAVES, int PIECES, int RING, int MFMAS, int LREADS>
void run(const bf8* src, size_t panel_bytes, int shared, int wg_per_cu, float* sink) {
    auto kern = fill_kernel<WAVES, PIECES, RING, MFMAS, LREADS>;
    const size_t lds_bytes = (size_t)RING * WAVES * PIECES * 1024;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) { printf("lds attr failed\n"); return; }
    int occ = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, WAVES * 64, lds_bytes);
    const int nsteps = 4000, grid = 256 * wg_per_cu;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e30f;
    for (int it = 0; it < 4; ++it) {
        hipEventRecord(a);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds_bytes, 0, src, panel_bytes / 16, shared, nsteps, sink);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (it && ms < best) best = ms;
    }
    const double bytes = (double)grid * nsteps * WAVES * PIECES * 1024.0;
    const double step_ns = best * 1e6 / nsteps / wg_per_cu;   // per workgroup-step on a CU (workgroups of a CU interleave)
    printf("waves %d pieces %d (%2d KiB/step) ring %d wg/CU %d (occupancy %d) mfma/wave/step %2d lds reads %2d %s panel %5.1f MB: %7.1f GB/s per CU  %6.2f TB/s chip  %6.0f ns per WG-step  MFMA rate %6.1f TFLOP/s\n",
           WAVES, PIECES, WAVES * PIECES, RING, wg_per_cu, occ, MFMAS, LREADS, shared ? "shared" : "own   ", panel_bytes / 1e6, bytes / best / 1e6 / 256, bytes / best / 1e9,
           step_ns, (double)grid * nsteps * WAVES * MFMAS * 16384.0 / best / 1e9);
    fflush(stdout);
}

int main() {
    const size_t cap = 64u << 20;
    bf8* src; float* sink;
    if (hipMalloc(&src, cap) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
    (void)hipMemset(src, 0, cap);
    for (int shared = 1; shared >= 0; --shared) {
        const size_t panel = shared ? (1u << 20) : (32u << 20);      // 1 MB weight panel (L2-resident per XCD) | 32 MB of activations (L2 + Infinity Cache)
        printf("---- %s ----\n", shared ? "every workgroup reads the same 1 MB panel" : "every workgroup walks its own part of a 32 MB buffer");
        // fill only
        run<4, 4, 4, 0, 0>(src, panel, shared, 1, sink);
        run<4, 4, 4, 0, 0>(src, panel, shared, 2, sink);
        run<4, 6, 4, 0, 0>(src, panel, shared, 1, sink);
        run<4, 6, 3, 0, 0>(src, panel, shared, 2, sink);
        run<8, 4, 3, 0, 0>(src, panel, shared, 1, sink);
        run<8, 4, 2, 0, 0>(src, panel, shared, 2, sink);
        run<4, 8, 4, 0, 0>(src, panel, shared, 1, sink);
        // with the kernels' MFMA and LDS-read counts
        run<4, 4, 4, 12, 8>(src, panel, shared, 2, sink);           // 64 x 64 today
        run<4, 6, 4, 24, 12>(src, panel, shared, 1, sink);          // 128 x 64, 4-slot ring
        run<4, 6, 3, 24, 12>(src, panel, shared, 2, sink);          // 128 x 64, 3-slot ring, two per CU
        run<8, 4, 3, 24, 12>(src, panel, shared, 1, sink);          // 128 x 128
        run<4, 8, 4, 48, 16>(src, panel, shared, 1, sink);          // 128 x 128 with 4 waves (64 x 64 wave tiles)
        // MFMA only (no fill): the issue ceiling of the same loop
        run<4, 0, 4, 12, 0>(src, panel, shared, 2, sink);
        run<4, 0, 4, 24, 0>(src, panel, shared, 2, sink);
    }
    return 0;
}
