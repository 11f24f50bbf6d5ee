// Weight (and bias) gradient of nn.Linear on the split-bf16 matrix pipe (training path, F.set_precision("bf16x3")):
//
//     dW[n][k] = sum_r dY[r][n] * X[r][k]          db[n] = sum_r dY[r][n]            (dY [R, N], X [R, K] row-major fp32)
//
// (F.linear's weight / bias gradient under autograd; every nn.Linear of Full_model/Models_spatial_memory.py and SubLayers.py:30-84.)
// The contraction runs over ROWS, so both MFMA operands need 8 consecutive rows of one column per lane while the activations are
// row-major: the operands are transposed on the way out of LDS by the hardware (ds_read_b64_tr_b16), not by the staging pass.
//   * a workgroup owns a 64 (n) x 64 (k) tile of dW and a slice of the rows; 4 waves as 2 x 2, a wave owns 32 x 32 = 2 x 2 MFMA tiles;
//   * a stage = 64 rows: every thread loads 4 + 4 float4 (dY and X, coalesced 256-byte row segments), splits them to bf16 (hi, lo) and
//     writes four ds_write_b64 per operand into [row][64 columns] images laid out in 8-row x 32-column subtiles with the XOR of
//     cdna_hip_programming.md T10 image (a): the transposed reads of two 16-lane groups 8 rows apart are bank-conflict free;
//   * a 32-row K group = 2 transposed reads per operand tile and image: the lane receives 4 + 4 consecutive rows of its column;
//     D = X^T(k rows) x dY(n cols): acc += Xl*DYh + Xh*DYl + Xh*DYh (v_mfma_f32_16x16x32_bf16, fp32 accumulate).  With X as the A operand
//     the lane owns dW[n = lane & 15][4 consecutive k]: one 16-byte store per tile;
//   * two LDS buffers: the next stage's global loads are in flight during the MFMAs, one barrier per stage;
//   * the bias gradient rides along: the staging threads of the k-tile-0 workgroups add up the fp32 dY values they load (fixed order:
//     16 row lanes per column quad, combined through LDS), so no separate column-sum launches are needed for db;
//   * rows are split over grid.z when the tile count alone cannot fill the chip; partials are summed in a fixed order by a second
//     launch (deterministic, no atomics), directly written when one slice suffices.
#include "common.h"

namespace {

typedef short s4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

constexpr int LG_IMG = 64 * 64 * 2;                 // bytes of one [64 rows][64 columns] bf16 image
constexpr int LG_STAGE = 4 * LG_IMG;                // dY hi | dY lo | X hi | X lo

// byte offset of 16-byte chunk `ch` (8 columns) of row `row` inside an image (T10 image (a) with 64-column rows)
__device__ __forceinline__ int img_off(int row, int ch) {
    return 1024 * (row >> 3) + 512 * (ch >> 2) + 64 * (row & 7) + 16 * ((ch & 3) ^ ((row >> 2) & 3));
}

struct LinGradArgs {
    const float* dy; const float* x; float* dw; float* db; float* part; float* dbpart;
    int ldy, ldx, lddw, R, N, K, rows_per, S;
    int xa = 0, xb = 0;     // XCD blocking of the tile grid: xa x xb = 8 blocks of (grid.x / xa) x (grid.y / xb) tiles, one per XCD; 0: launch order
    // CONVX: X is never materialised -- X[r = output pixel (b, oy, ox)][k = tap * Cin + ci] = x[b, oy * S + kh - 1, ox * S + kw - 1, ci] is gathered from the
    // NHWC activation while the stage is loaded (the weight gradient of a 3x3 / pad 1 convolution of any stride: dW = dY^T im2col(x))
    int cH = 0, cW = 0, cHo = 0, cWo = 0, cCin = 0, cS = 0;
};

__device__ __attribute__((aligned(64))) float eg_lingrad_zero_line[16];       // what out-of-image taps of the gathered operand read

template <bool CONVX>
__global__ __launch_bounds__(256, 2) void linear_wgrad_bf16_kernel(LinGradArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];       // 2 stages + bias scratch [16][64] floats
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
    // Workgroups are dealt round-robin to the 8 XCDs in launch order (x fastest), each XCD with its own L2.  In launch order XCD c owns ONE k tile and
    // every n tile: its X column slice is fetched once, but all dY slices by every XCD (memory-side reads 4 x the algorithmic bytes, round 6's traffic
    // table).  Remapped, XCD c owns a near-square block of tiles: (bw + bh) slices fetched per XCD instead of (1 + grid.y).
    int tx = blockIdx.x, ty = blockIdx.y;
    if (a.xa) {
        const int gx = gridDim.x, gy = gridDim.y;
        const int lin = ty * gx + tx;                  // gx * gy is a multiple of 8 here: the XCD of this workgroup is lin & 7 in every row slice z
        const int xcd = lin & 7, j = lin >> 3;
        const int bw = gx / a.xa, bh = gy / a.xb;
        tx = (xcd % a.xa) * bw + j % bw;
        ty = (xcd / a.xa) * bh + j / bw;
    }
    const int k0 = tx * 64, n0 = ty * 64, sl = blockIdx.z;
    const int r_beg = sl * a.rows_per, r_end = min(a.R, r_beg + a.rows_per);
    const int wn = (wave >> 1) * 32, wk = (wave & 1) * 32;
    const bool want_db = a.db != nullptr && tx == 0;
    // staging role: column quad cq (4 columns), rows rr + 16 i
    const int cq = tid & 15, rr = tid >> 4;
    const bool dy_vec = ((a.ldy & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.dy) & 15) == 0) && (n0 + 64 <= a.N);
    const bool x_vec = ((a.ldx & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.x) & 15) == 0) && (k0 + 64 <= a.K);
    f4 rd[4], rx[4];
    f4 bsum = (f4){0.f, 0.f, 0.f, 0.f};
    // CONVX: this thread's column quad of the implicit matrix is one tap and four consecutive input channels (Cin % 4 == 0), fixed for the whole walk;
    // its row -> pixel decomposition is carried along incrementally (rows advance by 16 inside a stage and by 64 between stages: a few compares, no division)
    int tdy = 0, tdx = 0, tci = 0, pb = 0, py = 0, px = 0;
    bool kval = true;
    if (CONVX) {
        const int kcol = k0 + 4 * cq;
        kval = kcol < a.K;
        const int tap = kval ? kcol / a.cCin : 0;
        tci = kcol - tap * a.cCin;
        tdy = tap / 3 - 1;
        tdx = tap % 3 - 1;
        const int r = r_beg + rr, hw = a.cHo * a.cWo;
        pb = r / hw;
        const int rem = r - pb * hw;
        py = rem / a.cWo;
        px = rem - py * a.cWo;
    }
    auto advance = [&](int& b_, int& y_, int& x_, int d) {          // d <= 64 rows on: at most 4 row wraps when Wo >= 16 (every map of the path); any Wo works
        x_ += d;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (x_ >= a.cWo) { x_ -= a.cWo; ++y_; }
        while (x_ >= a.cWo) { x_ -= a.cWo; ++y_; }
        while (y_ >= a.cHo) { y_ -= a.cHo; ++b_; }
    };
    auto load = [&](int r0) {
        int qb = pb, qy = py, qx = px;             // pixel of row r0 + rr
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0 + rr + 16 * i;
            f4 vd = (f4){0.f, 0.f, 0.f, 0.f}, vx = vd;
            if (r < r_end) {
                const float* dp = a.dy + (size_t)r * a.ldy + n0 + 4 * cq;
                if (dy_vec) vd = *reinterpret_cast<const f4*>(dp);
                else
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (n0 + 4 * cq + j < a.N) vd[j] = dp[j];
                if (CONVX) {
                    const int iy = qy * a.cS + tdy, ix = qx * a.cS + tdx;
                    const bool ok = kval && iy >= 0 && iy < a.cH && ix >= 0 && ix < a.cW;
                    const float* xp = ok ? a.x + (((size_t)qb * a.cH + iy) * a.cW + ix) * a.cCin + tci : eg_lingrad_zero_line;
                    vx = *reinterpret_cast<const f4*>(xp);
                } else {
                    const float* xp = a.x + (size_t)r * a.ldx + k0 + 4 * cq;
                    if (x_vec) vx = *reinterpret_cast<const f4*>(xp);
                    else
#pragma unroll
                        for (int j = 0; j < 4; ++j) if (k0 + 4 * cq + j < a.K) vx[j] = xp[j];
                }
            }
            rd[i] = vd;
            rx[i] = vx;
            if (CONVX) advance(qb, qy, qx, 16);
        }
        if (CONVX) advance(pb, py, px, 64);        // the next stage starts 64 rows further
    };
    auto store = [&](int buf) {
        unsigned char* base = lds + buf * LG_STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = rr + 16 * i;
            const int off = img_off(row, cq >> 1) + 8 * (cq & 1);
            unsigned h0, l0, h1, l1;
            {
                const f32x2_t p0 = {rd[i][0], rd[i][1]}, p1 = {rd[i][2], rd[i][3]};
                h0 = __builtin_bit_cast(unsigned, __builtin_convertvector(p0, bf16x2_t));
                h1 = __builtin_bit_cast(unsigned, __builtin_convertvector(p1, bf16x2_t));
                const f32x2_t q0 = {p0[0] - __uint_as_float(h0 << 16), p0[1] - __uint_as_float(h0 & 0xffff0000u)};
                const f32x2_t q1 = {p1[0] - __uint_as_float(h1 << 16), p1[1] - __uint_as_float(h1 & 0xffff0000u)};
                l0 = __builtin_bit_cast(unsigned, __builtin_convertvector(q0, bf16x2_t));
                l1 = __builtin_bit_cast(unsigned, __builtin_convertvector(q1, bf16x2_t));
            }
            *reinterpret_cast<u32x2_t*>(base + off) = (u32x2_t){h0, h1};
            *reinterpret_cast<u32x2_t*>(base + LG_IMG + off) = (u32x2_t){l0, l1};
            {
                const f32x2_t p0 = {rx[i][0], rx[i][1]}, p1 = {rx[i][2], rx[i][3]};
                h0 = __builtin_bit_cast(unsigned, __builtin_convertvector(p0, bf16x2_t));
                h1 = __builtin_bit_cast(unsigned, __builtin_convertvector(p1, bf16x2_t));
                const f32x2_t q0 = {p0[0] - __uint_as_float(h0 << 16), p0[1] - __uint_as_float(h0 & 0xffff0000u)};
                const f32x2_t q1 = {p1[0] - __uint_as_float(h1 << 16), p1[1] - __uint_as_float(h1 & 0xffff0000u)};
                l0 = __builtin_bit_cast(unsigned, __builtin_convertvector(q0, bf16x2_t));
                l1 = __builtin_bit_cast(unsigned, __builtin_convertvector(q1, bf16x2_t));
            }
            *reinterpret_cast<u32x2_t*>(base + 2 * LG_IMG + off) = (u32x2_t){h0, h1};
            *reinterpret_cast<u32x2_t*>(base + 3 * LG_IMG + off) = (u32x2_t){l0, l1};
            if (want_db) bsum += rd[i];
        }
    };
    // transposed fragment of the 16-column tile starting at chunk c0, K group s of the stage: rows 32 s + 8 g .. + 7 of column li
    auto frag = [&](const unsigned char* img, int c0, int s) -> bf8 {
        const int q = li >> 2, p = li & 3;
        const int r0 = 32 * s + 8 * g + q;
        const s4v lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s4v*)(img + img_off(r0, c0 + (p >> 1)) + 8 * (p & 1)));
        const s4v hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s4v*)(img + img_off(r0 + 4, c0 + (p >> 1)) + 8 * (p & 1)));
        return (bf8){lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
    };
    f4 acc[2][2];          // [k tile][n tile]
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) acc[t][u] = (f4){0.f, 0.f, 0.f, 0.f};

    const int nstage = (r_end - r_beg + 63) / 64;
    if (nstage > 0) {
        load(r_beg);
        store(0);
    }
    __syncthreads();
    for (int stg = 0; stg < nstage; ++stg) {
        const int buf = stg & 1;
        if (stg + 1 < nstage) load(r_beg + (stg + 1) * 64);           // in flight during the MFMAs below
        const unsigned char* base = lds + buf * LG_STAGE;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf8 dh[2], dl[2], xh[2], xl[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                dh[u] = frag(base, (wn >> 3) + 2 * u, s);
                dl[u] = frag(base + LG_IMG, (wn >> 3) + 2 * u, s);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                xh[t] = frag(base + 2 * LG_IMG, (wk >> 3) + 2 * t, s);
                xl[t] = frag(base + 3 * LG_IMG, (wk >> 3) + 2 * t, s);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl[t], dh[u], acc[t][u], 0, 0, 0);
                    acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh[t], dl[u], acc[t][u], 0, 0, 0);
                    acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh[t], dh[u], acc[t][u], 0, 0, 0);
                }
        }
        if (stg + 1 < nstage) store(buf ^ 1);            // the other buffer: its last reads finished before the previous barrier
        __syncthreads();
    }
    // D[row = k index 4g + r][col = n index li]: lane owns dW[n = li][k = 4g .. 4g + 3] of every (k tile, n tile)
    float* out = a.part ? a.part + (size_t)sl * a.N * a.lddw : a.dw;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int n = n0 + wn + 16 * u + li, k = k0 + wk + 16 * t + 4 * g;
            if (n >= a.N) continue;
            float* o = out + (size_t)n * a.lddw + k;
            if (k + 3 < a.K && ((a.lddw & 3) == 0)) *reinterpret_cast<f4*>(o) = acc[t][u];
            else
#pragma unroll
                for (int r = 0; r < 4; ++r) if (k + r < a.K) o[r] = acc[t][u][r];
        }
    if (want_db) {          // column sums of dY over this slice: 16 row lanes per column quad, combined in a fixed order
        f4* red = reinterpret_cast<f4*>(lds + 2 * LG_STAGE);
        red[rr * 16 + cq] = bsum;
        __syncthreads();
        if (tid < 16) {
            f4 s = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 16; ++j) s += red[j * 16 + tid];
            float* o = (a.part ? a.dbpart + (size_t)sl * a.N : a.db) + n0 + 4 * tid;
#pragma unroll
            for (int r = 0; r < 4; ++r) if (n0 + 4 * tid + r < a.N) o[r] = s[r];
        }
    }
}

// dw[n][k] = sum_s part[s][n][k] (partials at row pitch ldp, output at lddw); db[n] = sum_s dbpart[s][n]; fixed order
__global__ __launch_bounds__(256) void linear_wgrad_reduce_kernel(const float* __restrict__ part, const float* __restrict__ dbpart, float* __restrict__ dw,
                                                                  float* __restrict__ db, int N, int K, int ldp, int lddw, int S) {
    // one thread per 4 consecutive k of a row (the partials' pitch ldp is a multiple of 4: 16-byte reads); 16-byte stores when dw allows them
    const int q4 = ldp >> 2;
    const size_t total = (size_t)N * q4, slice = (size_t)N * ldp;
    const bool wide = (lddw & 3) == 0 && (reinterpret_cast<uintptr_t>(dw) & 15) == 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int n = (int)(i / q4), k = (int)(i - (size_t)n * q4) * 4;
        const float* p = part + (size_t)n * ldp + k;
        f4 s = *reinterpret_cast<const f4*>(p);
        for (int z = 1; z < S; ++z) s += *reinterpret_cast<const f4*>(p + (size_t)z * slice);
        float* o = dw + (size_t)n * lddw + k;
        if (wide && k + 4 <= K) *reinterpret_cast<f4*>(o) = s;
        else
            for (int j = 0; j < 4 && k + j < K; ++j) o[j] = s[j];
    }
    if (db && blockIdx.x == 0)
        for (int n = threadIdx.x; n < N; n += 256) {
            float s = 0.f;
            for (int z = 0; z < S; ++z) s += dbpart[(size_t)z * N + n];
            db[n] = s;
        }
}

struct LgPlan { int S, rows_per; };
LgPlan plan_lingrad(int rows, int n, int k) {
    const long tiles = (long)eg_cdiv(n, 64) * eg_cdiv(k, 64);
    int S = 1;
    if (rows > 640 && tiles < 384) {                    // few tiles and many rows: split the rows until ~1.5 workgroups per CU exist
        S = (int)((384 + tiles - 1) / tiles);
        const int smax = eg_cdiv(rows, 256);            // at least 4 stages per slice
        if (S > smax) S = smax;
        if (S < 1) S = 1;
    }
    int rows_per = (int)eg_round_up(eg_cdiv(rows, S), 64);
    S = eg_cdiv(rows, rows_per);
    return {S, rows_per};
}

}  // namespace

extern "C" int64_t eg_linear_wgrad_mfma_workspace_floats(int32_t rows, int32_t n, int32_t k) {
    if (rows <= 0 || n <= 0 || k <= 0) return 0;
    const LgPlan p = plan_lingrad(rows, n, k);
    return p.S > 1 ? (int64_t)p.S * n * (int64_t)eg_round_up(k, 4) + (int64_t)p.S * n : 0;
}

namespace { int lingrad_launch(const float* dy, int32_t ldy, const float* x, int32_t ldx, float* dw, int32_t lddw, float* db, int32_t rows, int32_t n, int32_t k,
                               float* workspace, int64_t workspace_floats, void* stream, const int* geo); }

extern "C" int eg_linear_wgrad_mfma(const float* dy, int32_t ldy, const float* x, int32_t ldx, float* dw, int32_t lddw, float* db, int32_t rows, int32_t n,
                                    int32_t k, float* workspace, int64_t workspace_floats, void* stream) {
    EG_REQUIRE(dy && x && dw && rows > 0 && n > 0 && k > 0 && ldy >= n && ldx >= k && lddw >= k, EG_ERR_BAD_ARG, "eg_linear_wgrad_mfma: bad argument");
    return lingrad_launch(dy, ldy, x, ldx, dw, lddw, db, rows, n, k, workspace, workspace_floats, stream, nullptr);
}

// Weight gradient of nn.Conv2d(cin -> cout, k = 3, pad = 1, ANY stride) on the split-bf16 matrix pipe as dW = dY^T im2col(x) with the im2col matrix gathered
// while the stage is loaded (the stride-2 entry convolutions of the tower, Full_model/ResNetSE34V2.py:40-55; the stride-1 body convolutions keep their
// dedicated kernel, wgrad.hip).  dw_mat [cout][9 * cin] with (kh, kw, ci) fastest to slowest, as eg_conv3x3_wgrad writes it.  cin % 4 == 0.
// workspace: eg_linear_wgrad_mfma_workspace_floats(batch * ho * wo, cout, 9 * cin).
extern "C" int eg_conv3x3_wgrad_gather_mfma(const float* x, const float* dy, float* dw_mat, int32_t batch, int32_t h, int32_t w, int32_t cin, int32_t cout,
                                            int32_t stride, float* workspace, int64_t workspace_floats, void* stream) {
    EG_REQUIRE(x && dy && dw_mat && batch > 0 && h > 0 && w > 0 && cin > 0 && cout > 0 && (stride == 1 || stride == 2), EG_ERR_BAD_ARG,
               "eg_conv3x3_wgrad_gather_mfma: bad argument");
    EG_REQUIRE(cin % 4 == 0 && eg_aligned16(x), EG_ERR_UNSUPPORTED, "eg_conv3x3_wgrad_gather_mfma: cin %% 4 == 0 and a 16-byte aligned map");
    const int ho = (h - 1) / stride + 1, wo = (w - 1) / stride + 1;
    EG_REQUIRE((int64_t)batch * ho * wo < (1ll << 31), EG_ERR_UNSUPPORTED, "eg_conv3x3_wgrad_gather_mfma: too many output pixels");
    const int geo[6] = {h, w, ho, wo, cin, stride};
    return lingrad_launch(dy, cout, x, 9 * cin, dw_mat, 9 * cin, nullptr, batch * ho * wo, cout, 9 * cin, workspace, workspace_floats, stream, geo);
}

namespace {
int lingrad_launch(const float* dy, int32_t ldy, const float* x, int32_t ldx, float* dw, int32_t lddw, float* db, int32_t rows, int32_t n, int32_t k,
                   float* workspace, int64_t workspace_floats, void* stream, const int* geo) {
    const LgPlan p = plan_lingrad(rows, n, k);
    const int64_t need = eg_linear_wgrad_mfma_workspace_floats(rows, n, k);
    EG_REQUIRE(p.S == 1 || (workspace && workspace_floats >= need), EG_ERR_WORKSPACE, "eg_linear_wgrad_mfma: workspace %lld < %lld floats",
               (long long)workspace_floats, (long long)need);
    hipStream_t st = (hipStream_t)stream;
    LinGradArgs a;
    a.dy = dy; a.x = x; a.dw = dw; a.db = db; a.ldy = ldy; a.ldx = ldx; a.R = rows; a.N = n; a.K = k; a.rows_per = p.rows_per; a.S = p.S;
    a.part = nullptr; a.dbpart = nullptr; a.lddw = lddw;
    if (p.S > 1) {
        a.lddw = (int)eg_round_up(k, 4);
        a.part = workspace;
        a.dbpart = workspace + (size_t)p.S * n * a.lddw;
    }
    {   // XCD blocking: xa | grid.x, xb | grid.y, xa * xb = 8, the block with the smallest half perimeter (slices fetched per XCD)
        const int gx = eg_cdiv(k, 64), gy = eg_cdiv(n, 64);
        if ((gx * gy) % 8 == 0) {                          // same-box A/B against launch order: 128-clip step 25.32 -> 25.12 ms (profiles/r06_train_ab.txt)
            int best = 1 << 30;
            for (int xa = 1; xa <= 8; xa *= 2) {
                const int xb = 8 / xa;
                if (gx % xa || gy % xb) continue;
                const int cost = gx / xa + gy / xb;
                if (cost < best) { best = cost; a.xa = xa; a.xb = xb; }
            }
        }
    }
    EgProfScope prof(8, 2.0 * rows * (double)n * (double)k, st);
    constexpr size_t LDS_BYTES = 2 * LG_STAGE + 16 * 16 * sizeof(f4);
    if (geo) {
        a.cH = geo[0]; a.cW = geo[1]; a.cHo = geo[2]; a.cWo = geo[3]; a.cCin = geo[4]; a.cS = geo[5];
        if (int rc = eg_ensure_dynamic_lds(reinterpret_cast<const void*>(linear_wgrad_bf16_kernel<true>), LDS_BYTES, "eg_conv3x3_wgrad_gather_mfma")) return rc;
        hipLaunchKernelGGL(linear_wgrad_bf16_kernel<true>, dim3(eg_cdiv(k, 64), eg_cdiv(n, 64), p.S), dim3(256), LDS_BYTES, st, a);
    } else {
        if (int rc = eg_ensure_dynamic_lds(reinterpret_cast<const void*>(linear_wgrad_bf16_kernel<false>), LDS_BYTES, "eg_linear_wgrad_mfma")) return rc;
        hipLaunchKernelGGL(linear_wgrad_bf16_kernel<false>, dim3(eg_cdiv(k, 64), eg_cdiv(n, 64), p.S), dim3(256), LDS_BYTES, st, a);
    }
    if (int rc = eg_check_launch("linear_wgrad_mfma")) return rc;
    if (p.S == 1) return EG_OK;
    const size_t total = (size_t)n * (a.lddw >> 2);
    const int nb = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(linear_wgrad_reduce_kernel, dim3(nb), dim3(256), 0, st, a.part, a.dbpart, dw, db, n, k, a.lddw, lddw, p.S);
    return eg_check_launch("linear_wgrad_reduce");
}
}  // namespace
