// Memory-pipeline probe for the NHWC fp32 access patterns of the 32-channel convolution stage (MI355X).  One "pixel" = 128 bytes (32 fp32 channels).
// Build: hipcc --offload-arch=gfx950 -O3 tools/mem_pattern_probe.hip -o tools/mem_pattern_probe ; run on the GPU box.
// Loads: each wave-instruction moves 1 KiB; what differs is how its 64 lanes' 16-byte pieces are spread over 128-byte lines.
// Stores: the epilogue writes 32 channels of 16 pixels per (wave, tile) as two instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int PAT>
__global__ __launch_bounds__(256) void load_probe(const float* __restrict__ x, float* __restrict__ sink, size_t npix) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    // a wave consumes 16 pixels (2 KiB) per step with two instructions
    const size_t nstep = npix / 16;
    for (size_t s = (size_t)blockIdx.x * 4 + wave; s < nstep; s += (size_t)gridDim.x * 4) {
        const float* base = x + s * 16 * 32;
        size_t o0, o1;
        if (PAT == 0) { o0 = lane * 4; o1 = 256 + lane * 4; }                                                  // contiguous 1 KiB per instruction
        else if (PAT == 1) { const int pix = (lane & 7) + 8 * (lane >> 5), oc = (lane >> 3) & 3; o0 = pix * 32 + oc * 8; o1 = o0 + 4; }   // conv staging today
        else { const int pix = lane >> 3, c = lane & 7; o0 = pix * 32 + c * 4; o1 = (pix + 8) * 32 + c * 4; } // 8 lanes = one full line
        acc += *reinterpret_cast<const f4*>(base + o0);
        acc += *reinterpret_cast<const f4*>(base + o1);
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = acc[0];
}

template <int PAT>
__global__ __launch_bounds__(256) void store_probe(float* __restrict__ y, size_t npix) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f4 v = {1.f, 2.f, 3.f, (float)lane};
    const size_t nstep = npix / 16;
    for (size_t s = (size_t)blockIdx.x * 4 + wave; s < nstep; s += (size_t)gridDim.x * 4) {
        float* base = y + s * 16 * 32;
        size_t o0, o1;
        if (PAT == 0) { o0 = lane * 4; o1 = 256 + lane * 4; }                                                  // contiguous
        else if (PAT == 1) { const int li = lane & 15, kq = lane >> 4; o0 = li * 32 + kq * 4; o1 = o0 + 16; }  // conv epilogue today: 64-byte halves of 16 lines
        else if (PAT == 2) { const int li = lane & 15, kq = lane >> 4; o0 = li * 32 + kq * 8; o1 = o0 + 4; }   // channel-permuted: 16 B at 32-byte stride
        else { const int pix = lane >> 3, c = lane & 7; o0 = pix * 32 + c * 4; o1 = (pix + 8) * 32 + c * 4; }  // full lines
        *reinterpret_cast<f4*>(base + o0) = v;
        *reinterpret_cast<f4*>(base + o1) = v;
    }
}

template <typename F> float timeit(F f, int iters = 20) {
    for (int i = 0; i < 3; ++i) f();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / iters * 1e3f;
}

int main() {
    const size_t npix = (size_t)64 * 128 * 124;         // one 32-channel activation map at B = 64: 130 MB
    float *x, *y, *sink;
    hipMalloc(&x, npix * 128); hipMalloc(&y, npix * 128); hipMalloc(&sink, 64);
    hipMemset(x, 0, npix * 128);
    const double mb = npix * 128 / 1e6;
    for (int grid : {512, 1024, 2048}) {
        printf("grid %d x 256 threads, %.0f MB per pass\n", grid, mb);
        float t;
        t = timeit([&] { hipLaunchKernelGGL(load_probe<0>, dim3(grid), dim3(256), 0, 0, x, sink, npix); }); printf("  load  contiguous            %7.1f us  %6.2f TB/s\n", t, mb / t / 1e0 * 1e-6 * 1e6 / 1e6);
        t = timeit([&] { hipLaunchKernelGGL(load_probe<1>, dim3(grid), dim3(256), 0, 0, x, sink, npix); }); printf("  load  conv staging (today)  %7.1f us  %6.2f TB/s\n", t, mb / t);
        t = timeit([&] { hipLaunchKernelGGL(load_probe<2>, dim3(grid), dim3(256), 0, 0, x, sink, npix); }); printf("  load  full line per 8 lanes %7.1f us  %6.2f TB/s\n", t, mb / t);
        t = timeit([&] { hipLaunchKernelGGL(store_probe<0>, dim3(grid), dim3(256), 0, 0, y, npix); }); printf("  store contiguous            %7.1f us  %6.2f TB/s\n", t, mb / t);
        t = timeit([&] { hipLaunchKernelGGL(store_probe<1>, dim3(grid), dim3(256), 0, 0, y, npix); }); printf("  store conv epilogue (today) %7.1f us  %6.2f TB/s\n", t, mb / t);
        t = timeit([&] { hipLaunchKernelGGL(store_probe<2>, dim3(grid), dim3(256), 0, 0, y, npix); }); printf("  store channel-permuted      %7.1f us  %6.2f TB/s\n", t, mb / t);
        t = timeit([&] { hipLaunchKernelGGL(store_probe<3>, dim3(grid), dim3(256), 0, 0, y, npix); }); printf("  store full line per 8 lanes %7.1f us  %6.2f TB/s\n", t, mb / t);
    }
    return 0;
}
