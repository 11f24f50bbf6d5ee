// Go / no-go probe for Winograd F(2x2, 3x3) on the 128 -> 128 tower convolution in split-bf16 arithmetic (MI355X; DESIGN.md §10, round-4 verdict item 2).
//
// The probe runs the INNER LOOP of the one formulation that fits a CU's 160 KB of LDS -- same instruction mix, same LDS images, same operand stream;
// the arithmetic runs on whatever bytes the buffers hold -- and reports the loop time per workgroup tile, from which the kernel's floor follows.
//
//   Formulation (4 running output accumulators + 1 product tile, V formed on the fly from the fp32 patch, as the verdict prescribes):
//     workgroup = 32 Winograd tiles (4 x 8 tiles = 8 x 16 output pixels; fp32 input patch 10 x 18 pixels x 128 ci = 92 160 B of LDS) x 128 co
//     4 waves = 2 tile groups (16 tiles = one MFMA N) x 2 co halves (64 co = 4 MFMA M tiles)
//     for each of the 16 transform positions (xi, nu):            product tile temp[4] (16 registers) = 0
//        for each 32-deep ci chunk (4):                           U(xi,nu) chunk: 128 co x 32 ci x (hi, lo) = 16 KB by LDS-DMA, 3-slot ring (48 KB)
//            V fragment: 4 patch pixels x 8 ci per lane = 8 ds_read_b128, 24 adds (B^T d B restricted to this position), split to (hi, lo)
//            4 co tiles: 2 ds_read_b128 (U hi, lo) + 3 MFMA 16x16x32 each
//        output transform: acc[o] += / -= temp for the 1, 2 or 4 outputs this position feeds (A^T rows; 36 tile adds over the 16 positions)
//     => 64 steps per workgroup tile; per wave and step: 12 MFMA, 16 ds_read_b128 (8 patch + 8 U), ~60 VALU; a larger workgroup tile does not fit
//        (64 tiles need a 166 KB patch), more co per wave needs the U chunk ring to grow past what is left beside the patch.
//   Work: 64 clips x 16 x 16 tiles = 16 384 tiles = 512 workgroup tiles = 2 per CU.
//
// Build: hipcc --offload-arch=gfx950 -O3 tools/wino_probe.hip -o tools/wino_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int N> __device__ __forceinline__ void wait_vmcnt_imm() { __builtin_amdgcn_s_waitcnt((N & 15) | 0x70 | 0xF00 | ((N >> 4) << 14)); }
__device__ __forceinline__ void wg_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// fp32 x 8 -> bf16 hi (round to nearest even) and lo = bf16(x - hi): the split the convolution kernels use (csrc/common.h split_octet)
__device__ __forceinline__ void split8(const f4& a, const f4& b, bf8& hi, bf8& lo) {
    unsigned int h[8], l[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float x = i < 4 ? a[i] : b[i - 4];
        unsigned int u = __float_as_uint(x);
        u += 0x7FFFu + ((u >> 16) & 1u);
        h[i] = u >> 16;
        const float r = x - __uint_as_float(h[i] << 16);
        unsigned int v = __float_as_uint(r);
        v += 0x7FFFu + ((v >> 16) & 1u);
        l[i] = v >> 16;
    }
    u32x4 ph = {h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
    u32x4 pl = {l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16)};
    hi = __builtin_bit_cast(bf8, ph);
    lo = __builtin_bit_cast(bf8, pl);
}

constexpr int PATCH_F4 = 10 * 18 * 32;            // fp32 patch: 180 pixels x 128 ci = 32 f4 per pixel
constexpr int USLOT = 1024;                       // bf8 slots of one U chunk: [hi|lo][4 ci octets][128 co]
constexpr int RING = 3;

template <bool WITH_V, bool WITH_OUT>
__global__ __launch_bounds__(256, 1) void wino_loop_kernel(const float* __restrict__ x, const bf8* __restrict__ u, float* __restrict__ y, int wg_tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f4* patch = reinterpret_cast<f4*>(smem);                                   // 92 160 B
    bf8* ur = reinterpret_cast<bf8*>(smem + PATCH_F4 * 16);                    // 3 x 16 KB
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tg = wave >> 1, ch = wave & 1;
    // this lane's tile inside the 4 x 8 tile block, top-left patch pixel (ty*2, tx*2)
    const int t = tg * 16 + li, ty = t >> 3, tx = t & 7;
    auto issue = [&](int step) {                  // one 16 KB U chunk: 16 pieces of 1 KB, 4 per wave
        bf8* S = ur + (step % RING) * USLOT + wave * 256;
        const bf8* g = u + (size_t)(step & 63) * USLOT + wave * 256 + lane;
#pragma unroll
        for (int p = 0; p < 4; ++p)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + p * 64), (__attribute__((address_space(3))) void*)(S + p * 64), 16, 0, 0);
    };
    for (int wt = blockIdx.x; wt < wg_tiles; wt += gridDim.x) {
        // patch load: 92 KB of fp32 per workgroup tile (coalesced; the real kernel reads 10 x 18 pixel rows of the NHWC map)
        const f4* src = reinterpret_cast<const f4*>(x) + (size_t)wt * PATCH_F4;
        for (int i = tid; i < PATCH_F4; i += 256) patch[i] = src[i];
        f4 acc[4][4];
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[o][c] = (f4){0.f, 0.f, 0.f, 0.f};
        issue(0); issue(1);
        __syncthreads();
        int step = 0;
#pragma unroll 1
        for (int pos = 0; pos < 16; ++pos) {
            const int xi = pos >> 2, nu = pos & 3;
            // B^T rows: row 0 = d0 - d2, row 1 = d1 + d2, row 2 = d2 - d1, row 3 = d1 - d3: two source rows / columns each, signs folded below
            const int r0 = (xi == 0) ? 0 : 1, r1 = (xi == 3) ? 3 : 2, c0 = (nu == 0) ? 0 : 1, c1 = (nu == 3) ? 3 : 2;
            const float sr = (xi == 1) ? 1.f : -1.f, sc = (nu == 1) ? 1.f : -1.f;
            f4 temp[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) temp[c] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
            for (int kc = 0; kc < 4; ++kc, ++step) {
                issue(step + 2);
                wait_vmcnt_imm<2 * 4>();                  // chunk `step` landed, two younger ones in flight
                wg_barrier();
                bf8 vh, vl;
                if (WITH_V) {
                    // V(xi,nu) for this lane's tile, channels kc*32 + kq*8 .. +7: (d[r0][c0] + sc*d[r0][c1]) + sr*(d[r1][c0] + sc*d[r1][c1])
                    const int cq = kc * 8 + kq * 2;
                    const f4* p00 = patch + ((ty * 2 + r0) * 18 + tx * 2 + c0) * 32 + cq;
                    const f4* p01 = patch + ((ty * 2 + r0) * 18 + tx * 2 + c1) * 32 + cq;
                    const f4* p10 = patch + ((ty * 2 + r1) * 18 + tx * 2 + c0) * 32 + cq;
                    const f4* p11 = patch + ((ty * 2 + r1) * 18 + tx * 2 + c1) * 32 + cq;
                    const f4 a = (p00[0] + sc * p01[0]) + sr * (p10[0] + sc * p11[0]);
                    const f4 b = (p00[1] + sc * p01[1]) + sr * (p10[1] + sc * p11[1]);
                    split8(a, b, vh, vl);
                } else {
                    vh = ur[(step % RING) * USLOT + lane];
                    vl = vh;
                }
                const bf8* U = ur + (step % RING) * USLOT + kq * 128 + ch * 64 + li;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const bf8 wh = U[c * 16], wl = U[512 + c * 16];
                    temp[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, vh, temp[c], 0, 0, 0);
                    temp[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, vl, temp[c], 0, 0, 0);
                    temp[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, vh, temp[c], 0, 0, 0);
                }
                wg_barrier();                              // the slot is overwritten by the copy issued at the next step
            }
            if (WITH_OUT) {
                // A^T = [[1,1,1,0],[0,1,-1,-1]]: output row i takes xi in {0,1,2} (i = 0) / {1,2,3} (i = 1), same for columns
                const bool i0 = xi <= 2, i1 = xi >= 1, j0 = nu <= 2, j1 = nu >= 1;
                const float si1 = (xi >= 2) ? -1.f : 1.f, sj1 = (nu >= 2) ? -1.f : 1.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (i0 && j0) acc[0][c] += temp[c];
                    if (i0 && j1) acc[1][c] += sj1 * temp[c];
                    if (i1 && j0) acc[2][c] += si1 * temp[c];
                    if (i1 && j1) acc[3][c] += (si1 * sj1) * temp[c];
                }
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[pos & 3][c] += temp[c];
            }
        }
        wait_vmcnt_imm<0>();
        // epilogue: 2 x 2 outputs x 4 co tiles per lane (the real kernel adds BN / ReLU / gate / residual here)
        float* yo = y + ((size_t)wt * 256 + tid) * 64;
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int c = 0; c < 4; ++c) *reinterpret_cast<f4*>(yo + (o * 4 + c) * 4) = acc[o][c];
        __syncthreads();
    }
}

template <bool WITH_V, bool WITH_OUT>
void run(const float* x, const bf8* u, float* y, const char* what) {
    auto kern = wino_loop_kernel<WITH_V, WITH_OUT>;
    const size_t lds = (size_t)PATCH_F4 * 16 + RING * USLOT * 16;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { printf("lds attr failed\n"); return; }
    const int wg_tiles = 512;                      // 64 clips x 256 tiles / 32
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float best = 1e30f;
    for (int it = 0; it < 6; ++it) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(kern, dim3(256), dim3(256), lds, 0, x, u, y, wg_tiles);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        if (it && ms < best) best = ms;
    }
    printf("%-78s %7.1f us per launch (64 clips x 32 x 31 x 128 -> 128; the direct split-bf16 kernel: 52.5 us; go <= 40 us)\n", what, best * 1e3);
    fflush(stdout);
}

int main() {
    float *x, *y; bf8* u;
    const size_t xb = (size_t)512 * PATCH_F4 * 16, ub = (size_t)64 * USLOT * 16, yb = (size_t)512 * 256 * 64 * 4;
    if (hipMalloc(&x, xb) != hipSuccess || hipMalloc(&u, ub) != hipSuccess || hipMalloc(&y, yb) != hipSuccess) return 1;
    (void)hipMemset(x, 0, xb); (void)hipMemset(u, 0, ub);
    printf("Winograd F(2x2,3x3) inner-loop probe, 128 -> 128 channels, split-bf16 (3 MFMA per product), LDS %zu B per workgroup, 1 workgroup per CU\n",
           (size_t)PATCH_F4 * 16 + RING * USLOT * 16);
    run<true, true>(x, u, y, "full loop: patch load, V on the fly (+ split), U ring, MFMA, output transform");
    run<false, true>(x, u, y, "without the V build (fragment read from LDS instead): MFMA + U reads + output transform");
    run<true, false>(x, u, y, "without the output-transform sign / scatter logic");
    run<false, false>(x, u, y, "MFMA + U stream only");
    return 0;
}
