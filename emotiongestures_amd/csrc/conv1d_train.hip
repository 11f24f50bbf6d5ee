// Training-path kernels of the small 1-D convolutions: the emotion CVAE's Conv1d / ConvTranspose1d stacks (CAVE/BEAT_CVAE.py:318-332,355-369)
// and the prior encoder's pred_conv (Full_model/Models_spatial_memory.py:224-231).  Channels-last activations [B, L, C]; weights in the
// reference's own layouts.  These layers are tiny (<= 34 channels, 34 positions: ~2 MFLOP per clip batch of 16) -- one launch per product on the
// fp32 FMA pipe replaces the im2col + padded-GEMM + permute chain (13 launches per layer and step) the training step used for them before.
//
//   forward          y[b, lo, co]  = bias[co] + sum_{ci, j} x[b, lo*s - p + j*d, ci] * w[co][ci][j]
//   backward_input   dx[b, li, ci] = (bias[ci]) + sum_{co, j : li + p - j*d = lo*s} dy[b, lo, co] * w[co][ci][j]
//   backward_weight  dw[co][ci][j] = sum_{b, lo} dy[b, lo, co] * x[b, lo*s - p + j*d, ci];  db_dy[co] = sum dy;  db_x[ci] = sum x
//
// nn.ConvTranspose1d(weight [Cin][Cout][k]) is the adjoint: its forward is backward_input (with the layer's bias), its input gradient is forward
// and its weight gradient is backward_weight with the roles of x and dy exchanged (db_x is then the layer's bias gradient).
// All sums run in a fixed order (no atomics): results are bitwise reproducible.
#include "common.h"

namespace {

struct C1Args {
    const float* x; const float* w; const float* bias; const float* dy;
    float* y; float* dw; float* db_dy; float* db_x;
    int B, L, Ci, Lo, Co, k, stride, pad, dil;
};

__global__ __launch_bounds__(256) void conv1d_cl_fwd_kernel(C1Args a) {
    const size_t total = (size_t)a.B * a.Lo * a.Co;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int co = (int)(idx % a.Co), lo = (int)((idx / a.Co) % a.Lo), b = (int)(idx / ((size_t)a.Co * a.Lo));
    float s = a.bias ? a.bias[co] : 0.f;
    const float* wr = a.w + (size_t)co * a.Ci * a.k;
    for (int j = 0; j < a.k; ++j) {
        const int li = lo * a.stride - a.pad + j * a.dil;
        if (li < 0 || li >= a.L) continue;
        const float* xr = a.x + ((size_t)b * a.L + li) * a.Ci;
        for (int ci = 0; ci < a.Ci; ++ci) s = fmaf(xr[ci], wr[ci * a.k + j], s);
    }
    a.y[idx] = s;
}

__global__ __launch_bounds__(256) void conv1d_cl_bwd_input_kernel(C1Args a) {
    const size_t total = (size_t)a.B * a.L * a.Ci;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int ci = (int)(idx % a.Ci), li = (int)((idx / a.Ci) % a.L), b = (int)(idx / ((size_t)a.Ci * a.L));
    float s = a.bias ? a.bias[ci] : 0.f;
    const size_t wco = (size_t)a.Ci * a.k;
    for (int j = 0; j < a.k; ++j) {
        const int t = li + a.pad - j * a.dil;
        if (t < 0 || t % a.stride) continue;
        const int lo = t / a.stride;
        if (lo >= a.Lo) continue;
        const float* dr = a.dy + ((size_t)b * a.Lo + lo) * a.Co;
        const float* wr = a.w + (size_t)ci * a.k + j;
        for (int co = 0; co < a.Co; ++co) s = fmaf(dr[co], wr[co * wco], s);
    }
    a.y[idx] = s;
}

constexpr int C1_KMAX = 8;

// one workgroup per (co, ci): the B * Lo rows are strided over 256 threads, the k tap sums (and the bias sums) are reduced wave -> workgroup in
// a fixed order
__global__ __launch_bounds__(256) void conv1d_cl_bwd_weight_kernel(C1Args a) {
    __shared__ float red[4][C1_KMAX + 2];
    const int ci = blockIdx.x, co = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float acc[C1_KMAX];
#pragma unroll
    for (int j = 0; j < C1_KMAX; ++j) acc[j] = 0.f;
    float sdy = 0.f, sx = 0.f;
    const int rows = a.B * a.Lo;
    for (int r = tid; r < rows; r += 256) {
        const int b = r / a.Lo, lo = r - b * a.Lo;
        const float g = a.dy[(size_t)r * a.Co + co];
        sdy += g;
        const int base = lo * a.stride - a.pad;
#pragma unroll
        for (int j = 0; j < C1_KMAX; ++j) {
            if (j < a.k) {
                const int li = base + j * a.dil;
                if (li >= 0 && li < a.L) acc[j] = fmaf(g, a.x[((size_t)b * a.L + li) * a.Ci + ci], acc[j]);
            }
        }
    }
    if (a.db_x && co == 0)
        for (int r = tid; r < a.B * a.L; r += 256) sx += a.x[(size_t)r * a.Ci + ci];
#pragma unroll
    for (int j = 0; j < C1_KMAX; ++j) acc[j] = wave_sum(acc[j]);
    sdy = wave_sum(sdy);
    sx = wave_sum(sx);
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < C1_KMAX; ++j) red[wave][j] = acc[j];
        red[wave][C1_KMAX] = sdy;
        red[wave][C1_KMAX + 1] = sx;
    }
    __syncthreads();
    if (tid < C1_KMAX + 2) {
        const float s = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
        if (tid < a.k) a.dw[((size_t)co * a.Ci + ci) * a.k + tid] = s;
        else if (tid == C1_KMAX && a.db_dy && ci == 0) a.db_dy[co] = s;
        else if (tid == C1_KMAX + 1 && a.db_x && co == 0) a.db_x[ci] = s;
    }
}

// ---- LDS-tiled forms (the CVAE's layers run over L = 512 positions: 65,536 rows at 128 clips, where the per-output kernels above spend their
// time re-reading x and w through L1).  A workgroup stages its input rows (pitch C + 1: conflict-free position-strided reads) and the whole
// filter in LDS once; a thread owns one position and CPW channels of its wave's channel group, the filter reads are wave-uniform broadcasts.
constexpr int C1_TL = 64;          // positions per workgroup (forward / input gradient)
constexpr int C1_TR = 128;         // output rows per workgroup (weight gradient)

template <int CPW>
__global__ __launch_bounds__(256) void conv1d_cl_fwd_tiled_kernel(C1Args a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, p = tid & 63, g = tid >> 6, b = blockIdx.y, lo0 = blockIdx.x * C1_TL;
    const int span = (C1_TL - 1) * a.stride + (a.k - 1) * a.dil + 1, pitch = a.Ci + 1, li0 = lo0 * a.stride - a.pad;
    float* xs = sm;                                         // [span][Ci + 1]
    float* ws = sm + ((span * pitch + 3) & ~3);             // [Ci * k][4 * CPW]
    for (int i = tid; i < span * a.Ci; i += 256) {
        const int r = i / a.Ci, c = i - r * a.Ci, li = li0 + r;
        xs[r * pitch + c] = (li >= 0 && li < a.L) ? a.x[((size_t)b * a.L + li) * a.Ci + c] : 0.f;
    }
    const int nck = a.Ci * a.k;
    for (int i = tid; i < nck * 4 * CPW; i += 256) {
        const int ck = i / (4 * CPW), co = i - ck * (4 * CPW);
        ws[i] = co < a.Co ? a.w[(size_t)co * nck + ck] : 0.f;
    }
    __syncthreads();
    float acc[CPW];
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int co = g * CPW + c;
        acc[c] = (a.bias && co < a.Co) ? a.bias[co] : 0.f;
    }
    const float* xp = xs + p * a.stride * pitch;
    for (int ci = 0; ci < a.Ci; ++ci)
        for (int j = 0; j < a.k; ++j) {
            const float xv = xp[j * a.dil * pitch + ci];
            const float* wp = ws + (ci * a.k + j) * (4 * CPW) + g * CPW;
#pragma unroll
            for (int c = 0; c < CPW; ++c) acc[c] = fmaf(xv, wp[c], acc[c]);
        }
    const int lo = lo0 + p;
    if (lo >= a.Lo) return;
    float* yo = a.y + ((size_t)b * a.Lo + lo) * a.Co;
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int co = g * CPW + c;
        if (co < a.Co) yo[co] = acc[c];
    }
}

// dx[b, li, ci]: the dy rows lo_min .. lo_max that reach this workgroup's 64 input positions are staged; a thread owns one li and CPW input channels
template <int CPW>
__global__ __launch_bounds__(256) void conv1d_cl_bwd_input_tiled_kernel(C1Args a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, p = tid & 63, g = tid >> 6, b = blockIdx.y, li0 = blockIdx.x * C1_TL;
    const int tmin = li0 + a.pad - (a.k - 1) * a.dil, tmax = li0 + C1_TL - 1 + a.pad;
    const int lo_min = tmin > 0 ? (tmin + a.stride - 1) / a.stride : 0;
    int lo_max = tmax / a.stride;
    if (lo_max > a.Lo - 1) lo_max = a.Lo - 1;
    const int nrow = lo_max >= lo_min ? lo_max - lo_min + 1 : 0, pitch = a.Co + 1;
    const int rows_cap = (C1_TL - 1 + (a.k - 1) * a.dil) / a.stride + 2;
    float* ds = sm;                                         // [rows_cap][Co + 1]
    float* ws = sm + ((rows_cap * pitch + 3) & ~3);         // [Co * k][4 * CPW]: w[co][ci][j] at [(co * k + j)][ci]
    for (int i = tid; i < nrow * a.Co; i += 256) {
        const int r = i / a.Co, c = i - r * a.Co;
        ds[r * pitch + c] = a.dy[((size_t)b * a.Lo + lo_min + r) * a.Co + c];
    }
    const int nok = a.Co * a.k;
    for (int i = tid; i < nok * 4 * CPW; i += 256) {
        const int ok = i / (4 * CPW), ci = i - ok * (4 * CPW), co = ok / a.k, j = ok - co * a.k;
        ws[i] = ci < a.Ci ? a.w[((size_t)co * a.Ci + ci) * a.k + j] : 0.f;
    }
    __syncthreads();
    const int li = li0 + p;
    float acc[CPW];
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int ci = g * CPW + c;
        acc[c] = (a.bias && ci < a.Ci) ? a.bias[ci] : 0.f;
    }
    for (int j = 0; j < a.k; ++j) {
        const int t = li + a.pad - j * a.dil;
        const int lo = t / a.stride;
        const bool ok = t >= 0 && t - lo * a.stride == 0 && lo <= lo_max && lo >= lo_min;
        const float* dp = ds + (ok ? lo - lo_min : 0) * pitch;
        for (int co = 0; co < a.Co; ++co) {
            const float dv = ok ? dp[co] : 0.f;            // (a select, not a product: unstaged LDS may hold NaN patterns)
            const float* wp = ws + (co * a.k + j) * (4 * CPW) + g * CPW;
#pragma unroll
            for (int c = 0; c < CPW; ++c) acc[c] = fmaf(dv, wp[c], acc[c]);
        }
    }
    if (li >= a.L) return;
    float* o = a.y + ((size_t)b * a.L + li) * a.Ci;
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int ci = g * CPW + c;
        if (ci < a.Ci) o[ci] = acc[c];
    }
}

// Weight gradient, level 1: a workgroup owns C1_TR output rows of one clip.  Thread -> (column cij of [Ci * k + 1], row group rg): the last
// column is the bias (x := 1).  It walks its rows with COB accumulators (one per output channel; the dy reads are near-uniform broadcasts),
// the row groups are folded through LDS in a fixed order and the workgroup's [Co][Ci * k + 1] partial goes to part[workgroup].
template <int COB>
__global__ __launch_bounds__(256) void conv1d_cl_bwd_weight_tiled_kernel(C1Args a, float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, b = blockIdx.y, lo0 = blockIdx.x * C1_TR;
    const int rows = (a.Lo - lo0 < C1_TR) ? a.Lo - lo0 : C1_TR;
    const int span = (C1_TR - 1) * a.stride + (a.k - 1) * a.dil + 1, pitch = a.Ci + 1, li0 = lo0 * a.stride - a.pad;
    float* xs = sm;                                         // [span][Ci + 1]
    float* ds = sm + ((span * pitch + 3) & ~3);             // [C1_TR][COB]
    for (int i = tid; i < span * a.Ci; i += 256) {
        const int r = i / a.Ci, c = i - r * a.Ci, li = li0 + r;
        xs[r * pitch + c] = (li >= 0 && li < a.L) ? a.x[((size_t)b * a.L + li) * a.Ci + c] : 0.f;
    }
    for (int i = tid; i < C1_TR * COB; i += 256) {
        const int r = i / COB, c = i - r * COB;
        ds[i] = (r < rows && c < a.Co) ? a.dy[((size_t)b * a.Lo + lo0 + r) * a.Co + c] : 0.f;
    }
    __syncthreads();
    const int ncol = a.Ci * a.k + 1, G = 256 / ncol;
    const int cij = tid % ncol, rg = tid / ncol;
    const bool active = rg < G;
    const bool is_bias = cij == ncol - 1;
    const int ci = is_bias ? 0 : cij / a.k, j = is_bias ? 0 : cij - ci * a.k;
    float acc[COB];
#pragma unroll
    for (int c = 0; c < COB; ++c) acc[c] = 0.f;
    if (active)
        for (int r = rg; r < rows; r += G) {
            const float xv = is_bias ? 1.f : xs[(r * a.stride + j * a.dil) * pitch + ci];
            const float* dp = ds + r * COB;
#pragma unroll
            for (int c = 0; c < COB; ++c) acc[c] = fmaf(dp[c], xv, acc[c]);
        }
    __syncthreads();                    // the staged tiles are dead: their memory becomes the fold buffer [G * ncol][COB + 1]
    float* fold = sm;
    if (active) {
#pragma unroll
        for (int c = 0; c < COB; ++c) fold[tid * (COB + 1) + c] = acc[c];
    }
    __syncthreads();
    if (tid < ncol) {
        float* o = part + ((size_t)(b * gridDim.x + blockIdx.x) * a.Co) * ncol + tid;
        for (int c = 0; c < a.Co; ++c) {
            float s = 0.f;
            for (int q = 0; q < G; ++q) s += fold[(q * ncol + tid) * (COB + 1) + c];
            o[(size_t)c * ncol] = s;
        }
    }
}
// level 2: dw[co][ci][j] / db[co] = sum over the workgroups' partials, fixed order
__global__ __launch_bounds__(256) void conv1d_cl_bwd_weight_fold_kernel(const float* __restrict__ part, int nwg, int Co, int ncol, float* __restrict__ dw,
                                                                        float* __restrict__ db) {
    // 64 outputs per workgroup; wave w sums the partials w, w + 4, ... (four independent chains each), the four waves are folded in order
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, i = blockIdx.x * 64 + lane, n = Co * ncol;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < n) {
        int q = w;
        for (; q + 12 < nwg; q += 16) {
            s0 += part[(size_t)q * n + i];
            s1 += part[(size_t)(q + 4) * n + i];
            s2 += part[(size_t)(q + 8) * n + i];
            s3 += part[(size_t)(q + 12) * n + i];
        }
        for (; q < nwg; q += 4) s0 += part[(size_t)q * n + i];
    }
    red[w][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (w != 0 || i >= n) return;
    const float s = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
    const int co = i / ncol, col = i - co * ncol;
    if (col < ncol - 1) dw[(size_t)co * (ncol - 1) + col] = s;
    else if (db) db[co] = s;
}

int cpw_of(int channels) {             // channels per wave (4 waves), rounded up to an instantiated count; 0: too wide for the tiled kernels
    const int need = (channels + 3) / 4;
    const int opts[] = {1, 2, 4, 8, 12, 16};
    for (int o : opts)
        if (need <= o) return o;
    return 0;
}
constexpr size_t C1_LDS_CAP = 64 * 1024;

// [R, K] -> [R, Kp] zero padded (the GEMM reads 16-byte row pieces): one launch instead of a fill and a strided copy
__global__ __launch_bounds__(256) void pad_cols_kernel(const float* __restrict__ x, float* __restrict__ y, size_t rows, int k, int kp) {
    const size_t total = rows * (size_t)kp;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t r = i / kp;
        const int c = (int)(i - r * kp);
        y[i] = c < k ? x[r * k + c] : 0.f;
    }
}

int check_c1(const C1Args& a, const char* who) {
    EG_REQUIRE(a.B > 0 && a.L > 0 && a.Ci > 0 && a.Lo > 0 && a.Co > 0, EG_ERR_BAD_ARG, "%s: empty shape", who);
    EG_REQUIRE(a.k >= 1 && a.k <= C1_KMAX && a.stride >= 1 && a.pad >= 0 && a.dil >= 1, EG_ERR_BAD_ARG, "%s: k=%d (1..%d) stride=%d pad=%d dilation=%d", who, a.k,
               C1_KMAX, a.stride, a.pad, a.dil);
    EG_REQUIRE((a.Lo - 1) * a.stride - a.pad + (a.k - 1) * a.dil <= a.L - 1 + a.pad, EG_ERR_BAD_ARG,
               "%s: Lo=%d does not fit L=%d (k=%d stride=%d pad=%d dilation=%d)", who, a.Lo, a.L, a.k, a.stride, a.pad, a.dil);
    return EG_OK;
}

C1Args make_c1(int B, int L, int Ci, int Lo, int Co, int k, int stride, int pad, int dil) {
    C1Args a;
    a.x = a.w = a.bias = a.dy = nullptr;
    a.y = a.dw = a.db_dy = a.db_x = nullptr;
    a.B = B; a.L = L; a.Ci = Ci; a.Lo = Lo; a.Co = Co; a.k = k; a.stride = stride; a.pad = pad; a.dil = dil;
    return a;
}

}  // namespace

extern "C" int eg_conv1d_cl_forward(const float* x, const float* w, const float* bias, float* y, int32_t batch, int32_t len, int32_t cin, int32_t len_out,
                                    int32_t cout, int32_t k, int32_t stride, int32_t pad, int32_t dilation, void* stream) {
    EG_REQUIRE(x && w && y, EG_ERR_BAD_ARG, "eg_conv1d_cl_forward: null pointer");
    C1Args a = make_c1(batch, len, cin, len_out, cout, k, stride, pad, dilation);
    if (int rc = check_c1(a, "eg_conv1d_cl_forward")) return rc;
    a.x = x; a.w = w; a.bias = bias; a.y = y;
    hipStream_t st = (hipStream_t)stream;
    const int cpw = cpw_of(cout);
    const int span = (C1_TL - 1) * stride + (k - 1) * dilation + 1;
    const size_t lds = sizeof(float) * ((size_t)((span * (cin + 1) + 3) & ~3) + (size_t)cin * k * 4 * cpw);
    if (cpw && lds <= C1_LDS_CAP && batch <= 65535) {
        const dim3 grid(eg_cdiv(len_out, C1_TL), batch);
        switch (cpw) {
            case 1: hipLaunchKernelGGL((conv1d_cl_fwd_tiled_kernel<1>), grid, dim3(256), lds, st, a); break;
            case 2: hipLaunchKernelGGL((conv1d_cl_fwd_tiled_kernel<2>), grid, dim3(256), lds, st, a); break;
            case 4: hipLaunchKernelGGL((conv1d_cl_fwd_tiled_kernel<4>), grid, dim3(256), lds, st, a); break;
            case 8: hipLaunchKernelGGL((conv1d_cl_fwd_tiled_kernel<8>), grid, dim3(256), lds, st, a); break;
            case 12: hipLaunchKernelGGL((conv1d_cl_fwd_tiled_kernel<12>), grid, dim3(256), lds, st, a); break;
            default: hipLaunchKernelGGL((conv1d_cl_fwd_tiled_kernel<16>), grid, dim3(256), lds, st, a); break;
        }
        return eg_check_launch("conv1d_cl_forward (tiled)");
    }
    const size_t total = (size_t)batch * len_out * cout;
    hipLaunchKernelGGL(conv1d_cl_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a);
    return eg_check_launch("conv1d_cl_forward");
}

extern "C" int eg_conv1d_cl_backward_input(const float* dy, const float* w, const float* bias, float* dx, int32_t batch, int32_t len, int32_t cin,
                                           int32_t len_out, int32_t cout, int32_t k, int32_t stride, int32_t pad, int32_t dilation, void* stream) {
    EG_REQUIRE(dy && w && dx, EG_ERR_BAD_ARG, "eg_conv1d_cl_backward_input: null pointer");
    C1Args a = make_c1(batch, len, cin, len_out, cout, k, stride, pad, dilation);
    if (int rc = check_c1(a, "eg_conv1d_cl_backward_input")) return rc;
    a.dy = dy; a.w = w; a.bias = bias; a.y = dx;
    hipStream_t st = (hipStream_t)stream;
    const int cpw = cpw_of(cin);
    const int rows_cap = (C1_TL - 1 + (k - 1) * dilation) / stride + 2;
    const size_t lds = sizeof(float) * ((size_t)((rows_cap * (cout + 1) + 3) & ~3) + (size_t)cout * k * 4 * cpw);
    // pad <= (k - 1) * dilation: the staged dy rows of a tile then fit rows_cap (any Conv1d / ConvTranspose1d of the path; else the untiled kernel)
    if (cpw && lds <= C1_LDS_CAP && batch <= 65535 && pad <= (k - 1) * dilation) {
        const dim3 grid(eg_cdiv(len, C1_TL), batch);
        switch (cpw) {
            case 1: hipLaunchKernelGGL((conv1d_cl_bwd_input_tiled_kernel<1>), grid, dim3(256), lds, st, a); break;
            case 2: hipLaunchKernelGGL((conv1d_cl_bwd_input_tiled_kernel<2>), grid, dim3(256), lds, st, a); break;
            case 4: hipLaunchKernelGGL((conv1d_cl_bwd_input_tiled_kernel<4>), grid, dim3(256), lds, st, a); break;
            case 8: hipLaunchKernelGGL((conv1d_cl_bwd_input_tiled_kernel<8>), grid, dim3(256), lds, st, a); break;
            case 12: hipLaunchKernelGGL((conv1d_cl_bwd_input_tiled_kernel<12>), grid, dim3(256), lds, st, a); break;
            default: hipLaunchKernelGGL((conv1d_cl_bwd_input_tiled_kernel<16>), grid, dim3(256), lds, st, a); break;
        }
        return eg_check_launch("conv1d_cl_backward_input (tiled)");
    }
    const size_t total = (size_t)batch * len * cin;
    hipLaunchKernelGGL(conv1d_cl_bwd_input_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a);
    return eg_check_launch("conv1d_cl_backward_input");
}

namespace {
struct C1WPlan { int cob; size_t lds; int nchunk; };
C1WPlan plan_c1w(int cin, int len_out, int cout, int k, int stride, int dilation) {
    C1WPlan p;
    p.cob = cout <= 8 ? 8 : (cout <= 16 ? 16 : (cout <= 36 ? 36 : 0));
    const int span = (C1_TR - 1) * stride + (k - 1) * dilation + 1, ncol = cin * k + 1;
    const size_t tiles = (size_t)((span * (cin + 1) + 3) & ~3) + (size_t)C1_TR * p.cob;
    const size_t fold = ncol <= 256 ? (size_t)(256 / ncol) * ncol * (p.cob + 1) : 0;
    p.lds = sizeof(float) * (tiles > fold ? tiles : fold);
    p.nchunk = eg_cdiv(len_out, C1_TR);
    if (ncol > 256 || p.lds > C1_LDS_CAP) p.cob = 0;
    return p;
}
}  // namespace

// partials of the tiled weight gradient: [batch * ceil(len_out / 128)][cout][cin * k + 1]; 0 when the shape takes the one-launch kernel
extern "C" int64_t eg_conv1d_cl_backward_weight_workspace_floats(int32_t batch, int32_t cin, int32_t len_out, int32_t cout, int32_t k, int32_t stride,
                                                                 int32_t dilation) {
    if (batch <= 0 || cin <= 0 || len_out <= 0 || cout <= 0 || k <= 0 || k > C1_KMAX || stride <= 0 || dilation <= 0) return 0;
    const C1WPlan p = plan_c1w(cin, len_out, cout, k, stride, dilation);
    if (!p.cob || batch > 65535 || (int64_t)batch * len_out < 512) return 0;
    return (int64_t)batch * p.nchunk * cout * (cin * k + 1);
}

extern "C" int eg_conv1d_cl_backward_weight(const float* x, const float* dy, float* dw, float* db_dy, float* db_x, int32_t batch, int32_t len, int32_t cin,
                                            int32_t len_out, int32_t cout, int32_t k, int32_t stride, int32_t pad, int32_t dilation, float* workspace,
                                            int64_t workspace_floats, void* stream) {
    EG_REQUIRE(x && dy && dw, EG_ERR_BAD_ARG, "eg_conv1d_cl_backward_weight: null pointer");
    C1Args a = make_c1(batch, len, cin, len_out, cout, k, stride, pad, dilation);
    if (int rc = check_c1(a, "eg_conv1d_cl_backward_weight")) return rc;
    a.x = x; a.dy = dy; a.dw = dw; a.db_dy = db_dy; a.db_x = db_x;
    hipStream_t st = (hipStream_t)stream;
    const int64_t need = eg_conv1d_cl_backward_weight_workspace_floats(batch, cin, len_out, cout, k, stride, dilation);
    if (need > 0 && !db_x) {
        EG_REQUIRE(workspace && workspace_floats >= need, EG_ERR_WORKSPACE, "eg_conv1d_cl_backward_weight: workspace %lld < %lld floats",
                   (long long)workspace_floats, (long long)need);
        const C1WPlan p = plan_c1w(cin, len_out, cout, k, stride, dilation);
        const dim3 grid(p.nchunk, batch);
        if (p.cob == 8) hipLaunchKernelGGL((conv1d_cl_bwd_weight_tiled_kernel<8>), grid, dim3(256), p.lds, st, a, workspace);
        else if (p.cob == 16) hipLaunchKernelGGL((conv1d_cl_bwd_weight_tiled_kernel<16>), grid, dim3(256), p.lds, st, a, workspace);
        else hipLaunchKernelGGL((conv1d_cl_bwd_weight_tiled_kernel<36>), grid, dim3(256), p.lds, st, a, workspace);
        if (int rc = eg_check_launch("conv1d_cl_backward_weight (tiled)")) return rc;
        const int ncol = cin * k + 1;
        hipLaunchKernelGGL(conv1d_cl_bwd_weight_fold_kernel, dim3(eg_cdiv(cout * ncol, 64)), dim3(256), 0, st, workspace, batch * p.nchunk, cout, ncol, dw,
                           db_dy);
        return eg_check_launch("conv1d_cl_backward_weight (fold)");
    }
    hipLaunchKernelGGL(conv1d_cl_bwd_weight_kernel, dim3(cin, cout), dim3(256), 0, st, a);
    return eg_check_launch("conv1d_cl_backward_weight");
}

extern "C" int eg_pad_cols(const float* x, float* y, int64_t rows, int32_t k, int32_t k_padded, void* stream) {
    EG_REQUIRE(x && y && rows > 0 && k > 0 && k_padded >= k, EG_ERR_BAD_ARG, "eg_pad_cols: bad argument");
    const size_t total = (size_t)rows * k_padded;
    const size_t nb = (total + 255) / 256;
    hipLaunchKernelGGL(pad_cols_kernel, dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(256), 0, (hipStream_t)stream, x, y, (size_t)rows, k, k_padded);
    return eg_check_launch("pad_cols");
}
