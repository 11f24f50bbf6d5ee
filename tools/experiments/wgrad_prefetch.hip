// Weight gradient of nn.Conv2d(k=3, pad=1, stride 1) on the split-bf16 matrix pipe (training path, F.set_precision("bf16x3")).
//
//   dW[co][(kh*3 + kw)*Cin + ci] = sum over (b, oy, ox) of dy[b,oy,ox,co] * x[b, oy + kh - 1, ox + kw - 1, ci]        (NHWC fp32 activations)
//
// (F.conv2d's weight gradient, the conv weights of Full_model/ResNetBlocks.py:12,14 under autograd.)  The contraction runs over
// PIXELS, so both MFMA operands need 8 consecutive pixels of one channel per lane while the activations are pixel-major: the
// staging pass transposes.  Design:
//   * a workgroup owns a (32 output channels) x (64 or 32 input channels) x 9 taps block of dW and walks a list of work units,
//     unit = (clip, 32-pixel column strip, row chunk); its accumulators (72 VGPRs per lane at 32 x 64) live in registers for
//     the whole walk and are written once, as a partial that a fixed-order pass sums (deterministic, no atomics);
//   * one step = one output row of the strip = 32 contraction elements = one K group of v_mfma_f32_16x16x32_bf16 per k-quarter;
//   * rows are staged once and reused by all 9 taps: a ring of 4 x rows (3 in use + the one being written) and 3 dy rows in LDS
//     (loaded two steps ahead into registers: three register sets),
//     each element split to (hi, lo) bf16 by the thread that loaded it (one thread = 2 adjacent pixels x 8 channels -> eight
//     ds_write_b32 per image), layout [slot of 8 pixels][channel position] (6 / 4 slots per ring row, no padding: see the kernel);
//   * the three kw taps of a row come from the aligned 16-byte read of their k-quarter's slot plus one dword of each neighbouring slot,
//     funnel-shifted in registers (v_alignbit_b32): kw = 0 and kw = 2 share three of their four dwords;
//   * per step and wave: 22 ds_read_b128 feed 54 MFMAs (9 taps x 2 channel tiles x 3 split terms);
//     acc += x_hi*dy_lo + x_lo*dy_hi + x_hi*dy_hi, fp32 accumulation.
// Channel positions: LDS position p holds channel (p % NOCT)*8 + p / NOCT (NOCT = channel octets of the block), so that the eight
// channels a staging thread holds land at stride NOCT positions and the lanes of a staging wave at consecutive positions; an MFMA
// tile is 16 consecutive positions, and the epilogue maps positions back to channels.
#include "common.h"

namespace {

struct WgradArgs {
    const float* x;
    const float* dy;
    float* part;            // [S][Co][9*Ci]
    int B, H, W, Ci, Co;
    int strips, chunks, R;  // units = B * strips * chunks; chunk = R rows
    int units, upw;         // units per workgroup (consecutive)
    int n_cit, nct;         // channel tiles: nct = (Ci / CI_T) * (Co / CO_T), input tile fastest
    const float* in_scale = nullptr; const float* in_shift = nullptr;      // x' = x * in_scale[ci] + in_shift[ci] (in-image pixels) while staging: the folded BatchNorm
};

// Staging lanes whose pixel lies outside the image (or whose stage is outside the unit's rows) load from this line instead of branching around the
// loads and zero-filling their registers: ~200 VALU + ~125 SALU per step against 54 MFMAs (profiles/r06_wgrad_pmc.txt), a quarter of them the
// predication forest and register zero-fills of load_stage -> 158 + 112 with this line; same-box A/B: 190 -> 182 / 138.6 -> 135 / 138 -> 134 us per call
// at 128 clips (32 / 64 / 128 channels), bitwise the same gradients (profiles/r06_train_ab.txt).
__device__ __attribute__((aligned(64))) float eg_wgrad_zero_line[16];

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split_pair(float a0, float a1, unsigned& hi, unsigned& lo) {
    const f32x2 v = {a0, a1};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    const f32x2 r = {a0 - __uint_as_float(hi << 16), a1 - __uint_as_float(hi & 0xffff0000u)};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
}

template <int CI_T, int CO_T, int WCI, int WCO>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_bf16_kernel(WgradArgs a) {
    constexpr int NOX = CI_T / 8, NOD = CO_T / 8;           // channel octets of the x / dy block
    constexpr int XU = 18 * NOX, DU = 16 * NOD;             // staging threads: x has 18 pixel pairs (columns -2 .. 33), dy 16
    // LDS image of one ring row: [slot of 8 pixels][channel position], 16 B per entry, no padding.  A ds_read_b128 is serviced in 16-lane groups that MIX
    // two k-quarters ({0-3, 12-15} of one with {4-11} of the next, MI355X_MICROARCH.md section LDS): here the 16 positions of one k-quarter are 256
    // contiguous bytes (every bank once) and the next k-quarter's lie a multiple of 256 B further, so each group is conflict free.  (Rounds 2-5 kept
    // [position][slot] with 7- / 5-slot padding: conflict free for 16 CONSECUTIVE lanes, 2-way conflicted for the hardware's groups -- 55 % of the kernel's
    // LDS-array cycles were conflict cycles, profiles/r06_wgrad_pmc.txt.)
    constexpr int XS = 6, DS = 4;                           // slots (8 pixels, 16 B) per ring row and position
    constexpr int XROW = CI_T * XS, DROW = CO_T * DS;       // bf8 slots of one ring row of one image
    constexpr int XIMG = 4 * XROW, DIMG = 3 * DROW;           // x: 3 rows in use + the one being written; dy: the row in use, the prefetched one, the written one
    constexpr int NCI = CI_T / 16 / WCI, NCO = CO_T / 16 / WCO;
    static_assert(WCI * WCO == 4 && NCI >= 1 && NCO >= 1 && XU + DU <= 256, "wave tiling / staging roles");
    extern __shared__ __attribute__((aligned(16))) bf8 lds[];       // X hi | X lo | dY hi | dY lo
    bf8* Xh = lds;
    bf8* Xl = lds + XIMG;
    bf8* Dh = lds + 2 * XIMG;
    bf8* Dl = Dh + DIMG;

    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wci = wave % WCI, wco = wave / WCI;
    // workgroups of one unit list (same s, all channel tiles) sit on one XCD and are adjacent in dispatch order: they read the same rows
    int s, ct;
    {
        const int lin = blockIdx.x, xcd = lin & 7, idx = lin >> 3;
        s = (idx / a.nct) * 8 + xcd;
        ct = idx % a.nct;
    }
    const int cit = ct % a.n_cit, cot = ct / a.n_cit;

    // staging role
    const bool is_x = tid < XU, is_d = !is_x && tid < XU + DU;
    const int t2 = is_x ? tid : tid - XU;
    const int NO = is_x ? NOX : NOD;
    const int so = t2 % NO, sq = t2 / NO - (is_x ? 1 : 0);         // channel octet, pixel pair (x: -1 .. 16, dy: 0 .. 15)
    const int sC = is_x ? a.Ci : a.Co;
    const int sch = (is_x ? cit * CI_T : cot * CO_T) + so * 8;
    const float* sbase = is_x ? a.x : a.dy;
    // LDS destination (dword index inside one ring row of one image) of channel j: (slot * positions + j * NO + so) * 4 + pair dword inside the slot
    const int spair = is_x ? sq + 4 : sq;                   // pixel pair index inside the row image (x: pairs -2 .. 33 -> dwords 3 .. 20)
    const int sdw = ((spair >> 2) * (is_x ? CI_T : CO_T) + so) * 4 + (spair & 3);
    const int sjstride = NO * 4;

    f4 isc0 = (f4){1.f, 1.f, 1.f, 1.f}, isc1 = isc0, ish0 = (f4){0.f, 0.f, 0.f, 0.f}, ish1 = ish0;
    const bool affine = a.in_scale != nullptr && is_x;
    if (affine) {
        isc0 = *reinterpret_cast<const f4*>(a.in_scale + sch); isc1 = *reinterpret_cast<const f4*>(a.in_scale + sch + 4);
        ish0 = *reinterpret_cast<const f4*>(a.in_shift + sch); ish1 = *reinterpret_cast<const f4*>(a.in_shift + sch + 4);
    }
    f4 acc[9][NCI][NCO];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < NCI; ++i)
#pragma unroll
            for (int o = 0; o < NCO; ++o) acc[t][i][o] = (f4){0.f, 0.f, 0.f, 0.f};

    const int u_end = min(a.units, (s + 1) * a.upw);
    for (int u = s * a.upw; u < u_end; ++u) {
        const int chunk = u % a.chunks, strip = (u / a.chunks) % a.strips, b = u / (a.chunks * a.strips);
        const int r0 = chunk * a.R, r1 = min(a.H, r0 + a.R), n = r1 - r0, ox0 = strip * 32;
        const int col0 = ox0 + 2 * sq;
        const bool v0 = col0 >= 0 && col0 < a.W, v1 = col0 + 1 >= 0 && col0 + 1 < a.W;
        const float* sp = sbase + ((size_t)b * a.H * a.W + col0) * sC + sch;       // + row * W * C
        // Software pipeline over "stages" (stage j = x row r0 - 1 + j and dy row r0 + j - 1): loaded into one of three register
        // sets at iteration j, split and written to the ring at the end of iteration j + 2 (two iterations of MFMAs hide the HBM
        // latency).  Iteration i computes output row r0 + i - 5 from the x stages i-5 .. i-3 (ring slots & 3) and the dy stage i-4 (slot % 3)
        // while stage i-2 is written.  The dy fragments and the kh = 0 row's x slots of the NEXT output row are visible one barrier early
        // (stages i-3 and i-4): they are read into registers at the END of iteration i, in front of its barrier, so that the MFMAs of iteration
        // i + 1 start right behind the barrier instead of behind an LDS round trip (round 6: the kernel is latency bound, two waves per SIMD).
        f4 pv[3][4];
        auto load_stage = [&](f4 (&dst)[4], int j) {
            const int row = r0 - 1 + j;
            const bool rv = is_x ? (row >= 0 && row < a.H && j <= n + 1) : (is_d && j >= 1 && row < r1);
            const float* p = sp + (size_t)row * a.W * sC;
            const bool in0 = rv && v0, in1 = rv && v1;
            const float* p0 = in0 ? p : eg_wgrad_zero_line;                 // branch-free: out-of-image lanes read zeros (see eg_wgrad_zero_line)
            const float* p1 = in1 ? p + sC : eg_wgrad_zero_line;
            dst[0] = *reinterpret_cast<const f4*>(p0); dst[1] = *reinterpret_cast<const f4*>(p0 + 4);
            dst[2] = *reinterpret_cast<const f4*>(p1); dst[3] = *reinterpret_cast<const f4*>(p1 + 4);
            if (affine) {               // in-image pixels only: what lies outside stays the zero padding of the normalised map
                if (in0) { dst[0] = dst[0] * isc0 + ish0; dst[1] = dst[1] * isc1 + ish1; }
                if (in1) { dst[2] = dst[2] * isc0 + ish0; dst[3] = dst[3] * isc1 + ish1; }
            }
        };
        auto write_stage = [&](const f4 (&src)[4], int j, int dslot) {        // dslot = j % 3
            if (is_x || is_d) {
                unsigned* hi = reinterpret_cast<unsigned*>(is_x ? Xh + (j & 3) * XROW : Dh + dslot * DROW) + sdw;
                unsigned* lo = reinterpret_cast<unsigned*>(is_x ? Xl + (j & 3) * XROW : Dl + dslot * DROW) + sdw;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    unsigned h, l;
                    split_pair(c < 4 ? src[0][c] : src[1][c - 4], c < 4 ? src[2][c] : src[3][c - 4], h, l);
                    hi[c * sjstride] = h;
                    lo[c * sjstride] = l;
                }
            }
        };
        // operands of the next output row fetched ahead of the barrier: dy fragments, and the three raw slots of the kh = 0 row per channel tile and image
        bf8 p_dyh[NCO], p_dyl[NCO];
        u32x4_t p_c[NCI][2];
        unsigned p_left[NCI][2], p_right[NCI][2];
        auto prefetch = [&](int i, int dslot) {          // for compute(i): dy stage i - 4 (dslot = (i - 4) % 3), x stage i - 5
#pragma unroll
            for (int o = 0; o < NCO; ++o) {
                const int pos = (wco * NCO + o) * 16 + li;
                p_dyh[o] = Dh[dslot * DROW + kq * CO_T + pos];
                p_dyl[o] = Dl[dslot * DROW + kq * CO_T + pos];
            }
            const int ring = (i - 5) & 3;
#pragma unroll
            for (int ci = 0; ci < NCI; ++ci) {
                const int pos = (wci * NCI + ci) * 16 + li;
#pragma unroll
                for (int img = 0; img < 2; ++img) {
                    const bf8* row = (img ? Xl : Xh) + ring * XROW + kq * CI_T + pos;
                    p_c[ci][img] = __builtin_bit_cast(u32x4_t, row[CI_T]);
                    p_left[ci][img] = __builtin_bit_cast(u32x4_t, row[0])[3];
                    p_right[ci][img] = __builtin_bit_cast(u32x4_t, row[2 * CI_T])[0];
                }
            }
        };
        auto compute = [&](int i) {
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int ring = (i - 5 + kh) & 3;
#pragma unroll
                for (int ci = 0; ci < NCI; ++ci) {
                    const int pos = (wci * NCI + ci) * 16 + li;
                    bf8 xf[2][3];                    // [image][kw]
#pragma unroll
                    for (int img = 0; img < 2; ++img) {
                        // three aligned 16-byte reads (slots kq, kq + 1, kq + 2 of this position): the centre slot and the one dword of each neighbour
                        // that the two shifted taps need -- as full b128 reads they are conflict free, the two lone dwords were 4-way conflicted
                        u32x4_t c;
                        unsigned left, right;
                        if (kh == 0) {
                            c = p_c[ci][img]; left = p_left[ci][img]; right = p_right[ci][img];
                        } else {
                            const bf8* row = (img ? Xl : Xh) + ring * XROW + kq * CI_T + pos;
                            c = __builtin_bit_cast(u32x4_t, row[CI_T]);
                            left = __builtin_bit_cast(u32x4_t, row[0])[3];
                            right = __builtin_bit_cast(u32x4_t, row[2 * CI_T])[0];
                        }
                        const unsigned m01 = __builtin_amdgcn_alignbit(c[1], c[0], 16), m12 = __builtin_amdgcn_alignbit(c[2], c[1], 16),
                                       m23 = __builtin_amdgcn_alignbit(c[3], c[2], 16);
                        const u32x4_t k0 = {__builtin_amdgcn_alignbit(c[0], left, 16), m01, m12, m23};
                        const u32x4_t k2 = {m01, m12, m23, __builtin_amdgcn_alignbit(right, c[3], 16)};
                        xf[img][0] = __builtin_bit_cast(bf8, k0);
                        xf[img][1] = __builtin_bit_cast(bf8, c);
                        xf[img][2] = __builtin_bit_cast(bf8, k2);
                    }
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                        for (int o = 0; o < NCO; ++o) {
                            f4& c = acc[kh * 3 + kw][ci][o];
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[1][kw], p_dyh[o], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[0][kw], p_dyl[o], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[0][kw], p_dyh[o], c, 0, 0, 0);
                        }
                }
            }
        };
        // k3 = i % 3 (a literal at the three call sites): the dy ring slots of this iteration follow from it
        auto step = [&](f4 (&ld)[4], const f4 (&wr)[4], int i, int k3) {
            load_stage(ld, i);
            if (i >= 5) compute(i);
            if (i >= 2) write_stage(wr, i - 2, (k3 + 1) % 3);
            if (i >= 4 && i < n + 4) prefetch(i + 1, k3);         // (i + 1 - 4) % 3 = k3: written at iteration i - 1, visible since its barrier
            __syncthreads();
        };
        const int last = n + 4;
        for (int i = 0; i <= last; i += 3) {
            step(pv[0], pv[1], i, 0);
            if (i + 1 <= last) step(pv[1], pv[2], i + 1, 1);
            if (i + 2 <= last) step(pv[2], pv[0], i + 2, 2);
        }
    }

    // ---- partial block: lane (li, kq) of a tile holds D[ci position kq*4 + r][co position li] ----
    float* part = a.part + (size_t)s * a.Co * 9 * a.Ci;
#pragma unroll
    for (int o = 0; o < NCO; ++o) {
        const int pco = (wco * NCO + o) * 16 + li;
        const int co = cot * CO_T + (pco % NOD) * 8 + pco / NOD;
        float* prow = part + (size_t)co * 9 * a.Ci + cit * CI_T;
#pragma unroll
        for (int ci = 0; ci < NCI; ++ci)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int pci = (wci * NCI + ci) * 16 + kq * 4 + r;
                const int cin = (pci % NOX) * 8 + pci / NOX;
#pragma unroll
                for (int t = 0; t < 9; ++t) prow[t * a.Ci + cin] = acc[t][ci][o][r];
            }
    }
}

// dw[i] = sum over s of part[s][i], fixed order: 64 outputs per workgroup, the four waves take s = w, w + 4, ... (eight loads in
// flight per lane), then combine in wave order.  (One thread per output walking all S partials was 100+ us at S = 512.)
// ci > 0: the output is written in the reference's OIHW order (dw[co][ci][tap]) instead of the partials' [co][tap][ci]: the permute that
// autograd's caller would otherwise run as a strided device copy per convolution.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int n, int S, int ci) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    if (i < n) {
        int s = w;
        for (; s + 28 < S; s += 32) {
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += part[(size_t)(s + 4 * k) * n + i];
        }
        for (int k = 0; s < S; s += 4, ++k) acc[k & 7] += part[(size_t)s * n + i];
    }
    red[w][lane] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    __syncthreads();
    if (w == 0 && i < n) {
        int o = i;
        if (ci > 0) {
            const int co = i / (9 * ci), rem = i - co * 9 * ci, tap = rem / ci, c = rem - tap * ci;
            o = (co * ci + c) * 9 + tap;
        }
        dw[o] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    }
}

struct Plan { int ci_t, strips, R, chunks, units, S, upw, nct, n_cit; };

Plan plan_wgrad(int B, int H, int W, int Ci, int Co) {
    Plan p;
    p.ci_t = (Ci % 64 == 0) ? 64 : 32;
    p.n_cit = Ci / p.ci_t;
    p.nct = p.n_cit * (Co / 32);
    p.strips = eg_cdiv(W, 32);
    const int slots = 512;                                        // co-resident workgroups: 2 per CU (launch bounds; 56 KB LDS at 32 x 64)
    const int want = slots / p.nct > 8 ? slots / p.nct : 8;       // unit lists (each runs nct workgroups)
    p.R = H;
    // 16-row chunks at least: 8-row chunks (twice the unit lists and partials at 16 clips) were measured slower, 7.92 vs 7.77 ms per step
    // (profiles/r06_train_ab.txt)
    while (B * p.strips * eg_cdiv(H, p.R) < want && p.R > 16) p.R = (p.R + 1) / 2;
    p.chunks = eg_cdiv(H, p.R);
    p.units = B * p.strips * p.chunks;
    p.upw = eg_cdiv(p.units, want);
    p.S = (int)eg_round_up(eg_cdiv(p.units, p.upw), 8);
    return p;
}

}  // namespace

extern "C" int64_t eg_conv3x3_wgrad_mfma_workspace_floats(int32_t batch, int32_t h, int32_t w, int32_t cin, int32_t cout) {
    if (batch <= 0 || h <= 0 || w <= 0 || cin % 32 || cout % 32) return 0;
    const Plan p = plan_wgrad(batch, h, w, cin, cout);
    return (int64_t)p.S * cout * 9 * cin;
}

namespace { int wgrad_mfma(const float* x, const float* dy, float* dw_mat, int32_t batch, int32_t h, int32_t w, int32_t cin, int32_t cout, float* workspace,
                           int64_t workspace_floats, int oihw, void* stream, const float* in_scale = nullptr, const float* in_shift = nullptr); }

// dw_mat [cout][9*cin] ((kh, kw, ci) fastest to slowest as eg_conv3x3_wgrad writes it); stride 1 only, cin % 32 == 0, cout % 32 == 0.
extern "C" int eg_conv3x3_wgrad_mfma(const float* x, const float* dy, float* dw_mat, int32_t batch, int32_t h, int32_t w, int32_t cin, int32_t cout,
                                     float* workspace, int64_t workspace_floats, void* stream) {
    return wgrad_mfma(x, dy, dw_mat, batch, h, w, cin, cout, workspace, workspace_floats, 0, stream);
}
// the same gradient written as dw [cout][cin][3][3] (nn.Conv2d's weight layout: no permute pass behind it)
extern "C" int eg_conv3x3_wgrad_mfma_oihw(const float* x, const float* dy, float* dw, int32_t batch, int32_t h, int32_t w, int32_t cin, int32_t cout,
                                          float* workspace, int64_t workspace_floats, void* stream) {
    return wgrad_mfma(x, dy, dw, batch, h, w, cin, cout, workspace, workspace_floats, 1, stream);
}
// the same for a convolution whose input was x' = x * in_scale[ci] + in_shift[ci] (eg_conv3x3_sq_in_affine): the affine is re-applied while x is staged
extern "C" int eg_conv3x3_wgrad_mfma_oihw_in_affine(const float* x, const float* in_scale, const float* in_shift, const float* dy, float* dw, int32_t batch,
                                                    int32_t h, int32_t w, int32_t cin, int32_t cout, float* workspace, int64_t workspace_floats, void* stream) {
    EG_REQUIRE(in_scale && in_shift && eg_aligned16(in_scale) && eg_aligned16(in_shift), EG_ERR_BAD_ARG, "eg_conv3x3_wgrad_mfma_oihw_in_affine: affine vectors");
    return wgrad_mfma(x, dy, dw, batch, h, w, cin, cout, workspace, workspace_floats, 1, stream, in_scale, in_shift);
}

namespace {
int wgrad_mfma(const float* x, const float* dy, float* dw_mat, int32_t batch, int32_t h, int32_t w, int32_t cin, int32_t cout, float* workspace,
               int64_t workspace_floats, int oihw, void* stream, const float* in_scale, const float* in_shift) {
    EG_REQUIRE(x && dy && dw_mat && workspace && batch > 0 && h > 0 && w > 0, EG_ERR_BAD_ARG, "eg_conv3x3_wgrad_mfma: null pointer or empty shape");
    EG_REQUIRE(cin > 0 && cout > 0 && cin % 32 == 0 && cout % 32 == 0, EG_ERR_UNSUPPORTED, "eg_conv3x3_wgrad_mfma: channels %d -> %d (multiples of 32)", cin, cout);
    EG_REQUIRE(eg_aligned16(x) && eg_aligned16(dy), EG_ERR_ALIGN, "eg_conv3x3_wgrad_mfma: activations must be 16-byte aligned");
    const Plan p = plan_wgrad(batch, h, w, cin, cout);
    EG_REQUIRE(workspace_floats >= (int64_t)p.S * cout * 9 * cin, EG_ERR_WORKSPACE, "eg_conv3x3_wgrad_mfma: workspace %lld < %lld floats",
               (long long)workspace_floats, (long long)p.S * cout * 9 * cin);
    hipStream_t st = (hipStream_t)stream;
    WgradArgs a;
    a.x = x; a.dy = dy; a.part = workspace; a.B = batch; a.H = h; a.W = w; a.Ci = cin; a.Co = cout;
    a.strips = p.strips; a.chunks = p.chunks; a.R = p.R; a.units = p.units; a.upw = p.upw; a.n_cit = p.n_cit; a.nct = p.nct;
    a.in_scale = in_scale; a.in_shift = in_shift;
    EgProfScope prof((int64_t)cin * 1000000 + (int64_t)cout * 1000 + 7, 2.0 * 9 * cin * cout * (double)h * w * batch, st);
    const dim3 grid(p.S * p.nct);
    if (p.ci_t == 64) {
        constexpr size_t LDS_BYTES = 16 * (size_t)(2 * 4 * 64 * 6 + 2 * 3 * 32 * 4);
        auto kern = conv3x3_wgrad_bf16_kernel<64, 32, 4, 1>;
        if (int rc = eg_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), LDS_BYTES, "conv3x3_wgrad")) return rc;
        hipLaunchKernelGGL(kern, grid, dim3(256), LDS_BYTES, st, a);
    } else {
        constexpr size_t LDS_BYTES = 16 * (size_t)(2 * 4 * 32 * 6 + 2 * 3 * 32 * 4);
        auto kern = conv3x3_wgrad_bf16_kernel<32, 32, 2, 2>;
        if (int rc = eg_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), LDS_BYTES, "conv3x3_wgrad")) return rc;
        hipLaunchKernelGGL(kern, grid, dim3(256), LDS_BYTES, st, a);
    }
    if (int rc = eg_check_launch("conv3x3_wgrad_mfma")) return rc;
    const int n = cout * 9 * cin;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((n + 63) / 64), dim3(256), 0, st, workspace, dw_mat, n, p.S, oihw ? cin : 0);
    return eg_check_launch("wgrad_reduce");
}
}  // namespace
