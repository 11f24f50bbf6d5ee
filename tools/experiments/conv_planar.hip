// First stage of the audio tower (32 channels at full spectrogram resolution) on producer-split activations.
// Reference semantics: Full_model/ResNetSE34V2.py:64-67 (stem), Full_model/ResNetBlocks.py:21-37 (the three 32-channel SEBasicBlocks).
//
// Why a second layout.  The persistent 32 -> 32 convolution of conv.hip keeps fp32 NHWC activations in HBM and splits every halo pixel to
// bf16 (hi, lo) while staging it: in-kernel stamps (profiles/r03i_conv32_phase_stamps.txt) and the SQ counters (4.8 non-MFMA VALU instructions
// per MFMA) say that kernel is bound by the SIMDs' instruction issue, not by HBM (3.4 TB/s at the memory-side counters, exactly the algorithmic
// bytes) and not by the matrix pipe (34 % busy).  Half of those instructions are the staging pass: address / bounds arithmetic, the split, the
// LDS writes.  Here the PRODUCER splits once, in its epilogue, and the activations of the stage live in HBM as two bf16 images per clip,
//
//     "P32":  [clip][hi | lo][channel octet 0..3][pixel][8 bf16]          16-byte slots, 128 bytes per pixel -- the same bytes as fp32 NHWC,
//
// which is exactly the channel-octet planar image the MFMA fragment reads want in LDS.  The consumer stages a halo tile with LDS-DMA
// (global_load_lds, 16 bytes per lane, no VGPRs, no VALU beyond one address per instruction): the LDS image is the lane-linear list of
// (image, octet, halo pixel) slots, each lane's source address points at that pixel's slot in HBM or -- outside the map -- at a 16-byte
// block of zeros.  With no register staging the next tile needs its own LDS buffer: one workgroup of 8 waves per CU owns two 8-row tile
// buffers (2 x 44 KB) plus the resident weights (36 KB) and walks 16 tiles; per tile ONE barrier: wait for this tile's copies -> barrier ->
// issue the next tile's copies (they land during the 9 taps) -> taps -> epilogue.  The residual of the SE tail comes from the block input's
// (hi, lo) images (hi + lo: 2^-17 relative to the fp32 value), prefetched into registers one tile ahead like the gate.
// Arithmetic and summation order of the products are those of conv.hip (same fragments, same tap order): only the residual's last bits differ.
#include "common.h"

namespace {

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __attribute__((aligned(16))) unsigned int g_zero_slot[16];       // zero-initialised: the source of every out-of-map halo slot
__device__ __attribute__((aligned(16))) float g_ones4[4] = {1.f, 1.f, 1.f, 1.f};

constexpr int P_TH = 8, P_IH = P_TH + 2, P_IW = 34, P_NPIX = P_IH * P_IW, P_PL = 352;      // halo tile 10 x 34, plane pitch 352 slots (0 mod 16)
constexpr int P_WTAP = 4 * 32, P_WIMG = 9 * P_WTAP;

struct PlanarArgs {
    const bf8* x;           // input planes
    const bf8* res;         // residual planes (block input) or nullptr
    const float* w;         // packed conv weights (fp32 image, then hi / lo images) as eg_conv3x3 takes them
    const float* bias; const float* scale; const float* shift; const float* gate;
    bf8* yp;                // output planes, or ...
    float* yf;              // ... fp32 NHWC output
    float* gap;             // per-(clip, tile) channel sums of the output or nullptr
    int H, W, relu, relu2, tiles_x, tiles_y, tiles, total_tiles;
};

__device__ __forceinline__ float bf_lo(unsigned int u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned int u) { return __uint_as_float(u & 0xffff0000u); }

template <int TERMS, bool OUT_PLANAR>
__global__ __launch_bounds__(512, 2) void conv3x3_c32_planar_kernel(PlanarArgs a) {
    constexpr int NIMG = (TERMS == 3) ? 2 : 1;
    constexpr int TILE = NIMG * 4 * P_PL;                  // slots of one tile buffer
    constexpr int NINSTR = TILE / 64;                      // LDS-DMA instructions per tile (44 / 22)
    constexpr int NPW = (NINSTR + 7) / 8;                  // per wave
    constexpr int MT = 2, NT = 2;
    static_assert(TILE % 64 == 0, "tile buffer = whole LDS-DMA instructions");
    extern __shared__ __attribute__((aligned(16))) bf8 lds[];      // tile A | tile B | weights (hi, lo) | gap scratch [2][8][32]
    bf8* wl = lds + 2 * TILE;
    float* sred = reinterpret_cast<float*>(lds + 2 * TILE + NIMG * P_WIMG);

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, kq = lane >> 4;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int HW = a.H * a.W;
    const size_t CLIP = (size_t)8 * HW;                    // slots per clip: 2 images x 4 octets x HW
    // XCD-aware: workgroups are dealt round-robin to the 8 XCDs; XCD k walks a contiguous eighth of the tile list, all resident workgroups
    // work on consecutive tiles at any time (conv.hip, persistent kernel)
    int wg = blockIdx.x;
    if ((gridDim.x & 7) == 0) wg = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int NWG = gridDim.x;
    if (wg >= a.total_tiles) return;

    // weights -> LDS once
    {
        const size_t f32_floats = (size_t)9 * 32 * 32;
        const bf8* whi = reinterpret_cast<const bf8*>(a.w + f32_floats);
        const bf8* wlo = whi + (size_t)9 * 4 * 32;
#pragma unroll
        for (int img = 0; img < NIMG; ++img) {
            const bf8* src = img ? wlo : whi;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const int piece = p * 8 + wave_u;
                if (piece < P_WIMG / 64)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 64 + lane),
                                                     (__attribute__((address_space(3))) void*)(wl + img * P_WIMG + piece * 64), 16, 0, 0);
            }
        }
    }
    // staging roles (tile independent): instruction j = wave + 8 k covers LDS slots 64 j .. 64 j + 63 of a tile buffer
    int rel[NPW], pyx[NPW];          // slot offset inside the clip relative to the tile origin; (iy << 8 | ix), or -1 for pad slots
#pragma unroll
    for (int k = 0; k < NPW; ++k) {
        const int slot = (wave + 8 * k) * 64 + lane;
        const int img = slot / (4 * P_PL), rem = slot - img * 4 * P_PL, oct = rem / P_PL, p = rem - oct * P_PL;
        const int iy = p / P_IW, ix = p - iy * P_IW;
        rel[k] = (img * 4 + oct) * HW + iy * a.W + ix;
        pyx[k] = (p < P_NPIX && slot < TILE) ? ((iy << 8) | ix) : -1;
    }
    struct Coord { int b, ty, tx; };
    const int adv_b = NWG / a.tiles, adv_r = NWG - adv_b * a.tiles, adv_ty = adv_r / a.tiles_x, adv_tx = adv_r - adv_ty * a.tiles_x;
    auto advance = [&](Coord& c) {
        c.tx += adv_tx;
        if (c.tx >= a.tiles_x) { c.tx -= a.tiles_x; ++c.ty; }
        c.ty += adv_ty;
        if (c.ty >= a.tiles_y) { c.ty -= a.tiles_y; ++c.b; }
        c.b += adv_b;
    };
    const bf8* zero = reinterpret_cast<const bf8*>(g_zero_slot);
    const float* ones4 = g_ones4;
    auto stage = [&](const Coord& c, int buf) {
        const int iy0 = c.ty * P_TH - 1, ix0 = c.tx * 32 - 1;
        const bf8* xb = a.x + (size_t)c.b * CLIP + (iy0 * a.W + ix0);
#pragma unroll
        for (int k = 0; k < NPW; ++k) {
            const int j = wave_u + 8 * k;
            if (j < NINSTR) {
                const int iy = pyx[k] >> 8, ix = pyx[k] & 255;
                const int gy = iy0 + iy, gx = ix0 + ix;
                const bool in = pyx[k] >= 0 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                const bf8* src = in ? xb + rel[k] : zero;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(lds + buf * TILE + j * 64), 16, 0, 0);
            }
        }
    };
    // this lane's output elements: pixel tile id = wave * MT + t -> (row id >> 1, column half id & 1), pixel li; channels n * 16 + kq * 4 .. + 3,
    // i.e. half (kq & 1) of octet 2 n + (kq >> 1)
    int pbase[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int id = wave * MT + t;
        pbase[t] = (id >> 1) * P_IW + (id & 1) * 16 + li;
    }
    struct Frags { bf8 wh[NT], wlf[NT], xh[MT], xl[MT]; };
    auto read_frags = [&](Frags& f, const bf8* tile, int tap) {
        const bf8* Wh = wl + tap * P_WTAP + kq * 32 + li;
        const int kh = tap / 3, kw = tap - kh * 3, toff = kh * P_IW + kw;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            f.wh[n] = Wh[n * 16];
            if (TERMS == 3) f.wlf[n] = Wh[P_WIMG + n * 16];
        }
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            f.xh[t] = tile[kq * P_PL + pbase[t] + toff];
            if (TERMS == 3) f.xl[t] = tile[(4 + kq) * P_PL + pbase[t] + toff];
        }
    };
    f4 bi[NT], sc[NT], sh[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int co = n * 16 + kq * 4;
        bi[n] = a.bias ? *reinterpret_cast<const f4*>(a.bias + co) : (f4){0.f, 0.f, 0.f, 0.f};
        sc[n] = a.scale ? *reinterpret_cast<const f4*>(a.scale + co) : (f4){1.f, 1.f, 1.f, 1.f};
        sh[n] = a.shift ? *reinterpret_cast<const f4*>(a.shift + co) : (f4){0.f, 0.f, 0.f, 0.f};
    }
    // residual (hi, lo halves of this lane's 4-channel groups) and gate of a tile, prefetched one tile ahead
    u32x2 rnh[MT][NT], rnl[MT][NT];
    f4 gtn[NT];
    const bool has_res = a.res != nullptr;          // uniform (a kernel argument)
    auto load_side = [&](const Coord& c) {
        // No lane-divergent branch or pointer select around these loads: a lane whose pixel lies outside the map reads a clamped (valid) pixel --
        // its value is never stored.  (A per-lane "valid ? slot : zero block" select was turned into two branches by hipcc, each with its own load
        // and a `s_waitcnt vmcnt(0)` at the merge: the wave then waited for the next tile's LDS-DMA right after issuing it.)
        if (has_res) {
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const int id = wave * MT + t;
                const int oy = min(c.ty * P_TH + (id >> 1), a.H - 1), ox = min(c.tx * 32 + (id & 1) * 16 + li, a.W - 1);
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const size_t slot = (size_t)c.b * CLIP + (size_t)(2 * n + (kq >> 1)) * HW + oy * a.W + ox;
                    const unsigned int* ph = reinterpret_cast<const unsigned int*>(a.res + slot) + (kq & 1) * 2;
                    rnh[t][n] = *reinterpret_cast<const u32x2*>(ph);
                    rnl[t][n] = *reinterpret_cast<const u32x2*>(ph + (size_t)4 * HW * 4);       // the lo image: 4 octet planes of HW slots further
                }
            }
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const float* gp = a.gate ? a.gate + (size_t)c.b * 32 + n * 16 + kq * 4 : ones4;
            gtn[n] = *reinterpret_cast<const f4*>(gp);
        }
    };
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int n = 0; n < NT; ++n) rnh[t][n] = rnl[t][n] = (u32x2){0u, 0u};

    Coord cur;
    cur.b = wg / a.tiles;
    cur.ty = (wg - cur.b * a.tiles) / a.tiles_x;
    cur.tx = wg - cur.b * a.tiles - cur.ty * a.tiles_x;
    Coord nxt = cur;
    stage(cur, 0);
    load_side(cur);
    int it = 0, prev_b = 0, prev_tile = 0;
    for (int L = wg; L < a.total_tiles; L += NWG, cur = nxt, ++it) {
        const int buf = it & 1;
        const int b = cur.b, tile_id = cur.ty * a.tiles_x + cur.tx;
        const int oy0 = cur.ty * P_TH, ox0 = cur.tx * 32;
        advance(nxt);
        // this tile's copies (issued one tile ago; the weights the first time) and its residual / gate have landed
        wait_vmcnt_imm<0>();
        wait_lgkmcnt0();                        // this wave's gap-scratch writes of the previous tile
        wg_barrier();                           // ... for every wave; and every wave is done reading the other tile buffer
        if (a.gap && it > 0 && tid < 32) {      // pooling partials of the previous tile (scratch [(it - 1) & 1])
            const float* sp = sred + ((it - 1) & 1) * 256;
            float sm = 0.f;
#pragma unroll
            for (int m = 0; m < 8; ++m) sm += sp[m * 32 + tid];
            a.gap[((size_t)prev_b * a.tiles + prev_tile) * 32 + tid] = sm;
        }
        u32x2 rh[MT][NT], rl[MT][NT];
        f4 gt[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            gt[n] = gtn[n];
#pragma unroll
            for (int t = 0; t < MT; ++t) { rh[t][n] = rnh[t][n]; rl[t][n] = rnl[t][n]; }
        }
        if (L + NWG < a.total_tiles) {          // the next tile: its halo into the other buffer, its residual / gate into registers
            stage(nxt, buf ^ 1);
            load_side(nxt);
        }
        const bf8* tile = lds + buf * TILE;
        f4 acc[MT][NT];
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[t][n] = (f4){0.f, 0.f, 0.f, 0.f};
        Frags fr[2];
        read_frags(fr[0], tile, 0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (tap < 8) read_frags(fr[(tap + 1) & 1], tile, tap + 1);
            const Frags& f = fr[tap & 1];
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    if (TERMS == 3) {
                        acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.wlf[n], f.xh[t], acc[t][n], 0, 0, 0);
                        acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.wh[n], f.xl[t], acc[t][n], 0, 0, 0);
                    }
                    acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.wh[n], f.xh[t], acc[t][n], 0, 0, 0);
                }
        }
        // epilogue: v = acc + bias; relu; v * scale + shift; (* gate + residual; relu); store; pooling partials
        f4 gsum[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) gsum[n] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const int id = wave * MT + t;
            const int oy = oy0 + (id >> 1), ox = ox0 + (id & 1) * 16 + li;
            const bool ok = oy < a.H && ox < a.W;
            const int pix = oy * a.W + ox;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                f4 v = acc[t][n] + bi[n];
                if (a.relu) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                }
                v = v * sc[n] + sh[n];
                if (a.gate) v = v * gt[n];
                if (has_res) {
                    v[0] += bf_lo(rh[t][n][0]) + bf_lo(rl[t][n][0]);
                    v[1] += bf_hi(rh[t][n][0]) + bf_hi(rl[t][n][0]);
                    v[2] += bf_lo(rh[t][n][1]) + bf_lo(rl[t][n][1]);
                    v[3] += bf_hi(rh[t][n][1]) + bf_hi(rl[t][n][1]);
                }
                if (a.relu2) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                }
                if (ok) {
                    if constexpr (OUT_PLANAR) {
                        const f32x2_t p0 = {v[0], v[1]}, p1 = {v[2], v[3]};
                        const unsigned h0 = __builtin_bit_cast(unsigned, __builtin_convertvector(p0, bf16x2_t));
                        const unsigned h1 = __builtin_bit_cast(unsigned, __builtin_convertvector(p1, bf16x2_t));
                        const f32x2_t q0 = {v[0] - bf_lo(h0), v[1] - bf_hi(h0)}, q1 = {v[2] - bf_lo(h1), v[3] - bf_hi(h1)};
                        const unsigned l0 = __builtin_bit_cast(unsigned, __builtin_convertvector(q0, bf16x2_t));
                        const unsigned l1 = __builtin_bit_cast(unsigned, __builtin_convertvector(q1, bf16x2_t));
                        const size_t slot = (size_t)b * CLIP + (size_t)(2 * n + (kq >> 1)) * HW + pix;
                        unsigned int* ph = reinterpret_cast<unsigned int*>(a.yp + slot) + (kq & 1) * 2;
                        *reinterpret_cast<u32x2*>(ph) = (u32x2){h0, h1};
                        *reinterpret_cast<u32x2*>(ph + (size_t)4 * HW * 4) = (u32x2){l0, l1};
                    } else {
                        *reinterpret_cast<f4*>(a.yf + ((size_t)b * HW + pix) * 32 + n * 16 + kq * 4) = v;
                    }
                    gsum[n] += v;
                }
            }
        }
        if (a.gap) {
            float* sp = sred + buf * 256 + wave * 32;
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float sm = gsum[n][r];
                    sm += __shfl_xor(sm, 1, 64); sm += __shfl_xor(sm, 2, 64);
                    sm += __shfl_xor(sm, 4, 64); sm += __shfl_xor(sm, 8, 64);
                    if (li == 0) sp[n * 16 + kq * 4 + r] = sm;
                }
        }
        prev_b = b;
        prev_tile = tile_id;
    }
    if (a.gap && it > 0) {
        wait_lgkmcnt0();
        wg_barrier();
        if (tid < 32) {
            const float* sp = sred + ((it - 1) & 1) * 256;
            float sm = 0.f;
#pragma unroll
            for (int m = 0; m < 8; ++m) sm += sp[m * 32 + tid];
            a.gap[((size_t)prev_b * a.tiles + prev_tile) * 32 + tid] = sm;
        }
    }
}

// ---- stem: Conv2d(1 -> 32, 3x3, bias) -> ReLU -> BN, spectrogram [B,H,W] -> P32 planes -------------------------------------------------------
// One workgroup per output row; thread = (pixel, channel octet): its 9 x 8 tap weights, bias and BN affine stay in registers.
__global__ __launch_bounds__(256) void stem_conv_planar_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                               const float* __restrict__ scale, const float* __restrict__ shift, bf8* __restrict__ y,
                                                               int H, int W) {
    extern __shared__ float rows[];                     // [3][W + 2]
    const int tid = threadIdx.x;
    const int b = blockIdx.x / H, oy = blockIdx.x - b * H, WP = W + 2, HW = H * W;
    const float* xb = x + (size_t)b * HW;
    for (int i = tid; i < 3 * WP; i += 256) {
        const int r = i / WP, c = i - r * WP, gy = oy + r - 1, gx = c - 1;
        rows[i] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? xb[gy * W + gx] : 0.f;
    }
    const int oc = tid & 3, px0 = tid >> 2;            // 64 pixels x 4 octets per pass
    f4 wv[9][2], bi[2], sc[2], sh[2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf) wv[t][hlf] = *reinterpret_cast<const f4*>(w + t * 32 + oc * 8 + hlf * 4);
#pragma unroll
    for (int hlf = 0; hlf < 2; ++hlf) {
        bi[hlf] = *reinterpret_cast<const f4*>(bias + oc * 8 + hlf * 4);
        sc[hlf] = *reinterpret_cast<const f4*>(scale + oc * 8 + hlf * 4);
        sh[hlf] = *reinterpret_cast<const f4*>(shift + oc * 8 + hlf * 4);
    }
    __syncthreads();
    bf8* yh = y + (size_t)b * 8 * HW + (size_t)oc * HW + (size_t)oy * W;
    bf8* yl = yh + (size_t)4 * HW;
    for (int ox = px0; ox < W; ox += 64) {
        f4 v[2] = {bi[0], bi[1]};
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const float xv = rows[kh * WP + ox + kw];
                v[0] += wv[kh * 3 + kw][0] * xv;
                v[1] += wv[kh * 3 + kw][1] * xv;
            }
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[hlf][r] = fmaxf(v[hlf][r], 0.f);
            v[hlf] = v[hlf] * sc[hlf] + sh[hlf];
        }
        bf8 hi, lo;
        split_octet<true>(v[0], v[1], hi, lo);
        yh[ox] = hi;
        yl[ox] = lo;
    }
}

// ---- SE gate from the moments of conv1's output held as planes (conv.hip: se_gate_pre_kernel, same algebra and summation structure) ----------
constexpr int GPP_T = 1024;
__device__ __forceinline__ float plane_at(const bf8* __restrict__ tb, int HW, int pix, int c) {
    const unsigned short* h = reinterpret_cast<const unsigned short*>(tb + (size_t)(c >> 3) * HW + pix) + (c & 7);
    return bf16_to_f32(h[0]) + bf16_to_f32(h[(size_t)4 * HW * 8]);
}
__global__ __launch_bounds__(GPP_T) void se_gate_pre_planar_kernel(const bf8* __restrict__ t1, const float* __restrict__ gap, int tiles,
                                                                   const float* __restrict__ w2img, const float* __restrict__ scale2,
                                                                   const float* __restrict__ shift2, const float* __restrict__ w1,
                                                                   const float* __restrict__ b1, const float* __restrict__ wf2,
                                                                   const float* __restrict__ bf2, float* __restrict__ gate, int H, int W) {
    constexpr int C = 32, G = GPP_T / C, R = C >> 3;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int b = blockIdx.x, t = threadIdx.x, c = t % C, g = t / C, HW = H * W;
    float* part = sm;                   // [5][GPP_T]
    float* S = sm + 5 * GPP_T;          // [9][C]
    float* zp = S + 9 * C;              // [G][C]
    float* m = zp + GPP_T;              // [C]
    float* hbuf = m + C;                // [R]
    const bf8* tb = t1 + (size_t)b * 8 * HW;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
    for (int i = g; i < tiles; i += G) s0 += gap[((size_t)b * tiles + i) * C + c];
    for (int x = g; x < W; x += G) {
        s1 += plane_at(tb, HW, x, c);
        s2 += plane_at(tb, HW, (H - 1) * W + x, c);
    }
    for (int y = g; y < H; y += G) {
        s3 += plane_at(tb, HW, y * W, c);
        s4 += plane_at(tb, HW, y * W + W - 1, c);
    }
    part[t] = s0; part[GPP_T + t] = s1; part[2 * GPP_T + t] = s2; part[3 * GPP_T + t] = s3; part[4 * GPP_T + t] = s4;
    __syncthreads();
    if (t < C) {
        float T = 0.f, R0 = 0.f, RL = 0.f, C0 = 0.f, CL = 0.f;
        for (int j = 0; j < G; ++j) {
            T += part[j * C + t]; R0 += part[GPP_T + j * C + t]; RL += part[2 * GPP_T + j * C + t]; C0 += part[3 * GPP_T + j * C + t];
            CL += part[4 * GPP_T + j * C + t];
        }
        const float c00 = plane_at(tb, HW, 0, t), c0L = plane_at(tb, HW, W - 1, t), cL0 = plane_at(tb, HW, (H - 1) * W, t),
                    cLL = plane_at(tb, HW, (H - 1) * W + W - 1, t);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const float rex = kh == 0 ? RL : (kh == 2 ? R0 : 0.f);
                const float cex = kw == 0 ? CL : (kw == 2 ? C0 : 0.f);
                float corner = 0.f;
                if (kh == 0 && kw == 0) corner = cLL;
                if (kh == 0 && kw == 2) corner = cL0;
                if (kh == 2 && kw == 0) corner = c0L;
                if (kh == 2 && kw == 2) corner = c00;
                S[(kh * 3 + kw) * C + t] = T - rex - cex + corner;
            }
    }
    __syncthreads();
    {
        const f4* w4 = reinterpret_cast<const f4*>(w2img);
        const int nq = 9 * (C >> 2);
        float z = 0.f;
        for (int q = g; q < nq; q += G) {
            const int tap = q / (C >> 2), cq = q - tap * (C >> 2);
            const f4 wv = w4[(size_t)q * C + c];
            const float* sp = S + tap * C + cq * 4;
            z += (wv[0] * sp[0] + wv[1] * sp[1]) + (wv[2] * sp[2] + wv[3] * sp[3]);
        }
        zp[g * C + c] = z;
    }
    __syncthreads();
    if (t < C) {
        float z = 0.f;
        for (int j = 0; j < G; ++j) z += zp[j * C + t];
        m[t] = z / (float)(H * W) * scale2[t] + shift2[t];
    }
    __syncthreads();
    if (t < R) {
        float s = b1[t];
        for (int cc = 0; cc < C; ++cc) s += w1[t * C + cc] * m[cc];
        hbuf[t] = fmaxf(s, 0.f);
    }
    __syncthreads();
    if (t < C) {
        float s = bf2[t];
        for (int j = 0; j < R; ++j) s += wf2[t * R + j] * hbuf[j];
        gate[(size_t)b * C + t] = 1.f / (1.f + expf(-s));
    }
}

// P32 <-> fp32 NHWC (tests / taps)
__global__ __launch_bounds__(256) void planar_to_nhwc_kernel(const bf8* __restrict__ p, float* __restrict__ y, int HW, size_t total) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i & 31);
        const size_t px = i >> 5, b = px / HW;
        y[i] = plane_at(p + b * 8 * HW, HW, (int)(px - b * HW), c);
    }
}
__global__ __launch_bounds__(256) void nhwc_to_planar_kernel(const float* __restrict__ x, bf8* __restrict__ p, int HW, size_t slots) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < slots; i += (size_t)gridDim.x * 256) {       // i = (b, octet, pixel)
        const size_t b = i / ((size_t)4 * HW), rem = i - b * 4 * HW;
        const int oc = (int)(rem / HW), pix = (int)(rem - (size_t)oc * HW);
        const float* src = x + ((size_t)b * HW + pix) * 32 + oc * 8;
        bf8 hi, lo;
        split_octet<true>(*reinterpret_cast<const f4*>(src), *reinterpret_cast<const f4*>(src + 4), hi, lo);
        p[b * 8 * HW + (size_t)oc * HW + pix] = hi;
        p[b * 8 * HW + (size_t)(4 + oc) * HW + pix] = lo;
    }
}

template <int TERMS, bool OUTP>
int launch_planar(const PlanarArgs& a, int grid, hipStream_t st) {
    constexpr int NIMG = (TERMS == 3) ? 2 : 1;
    constexpr size_t LDS_BYTES = sizeof(bf8) * (size_t)(2 * NIMG * 4 * P_PL + NIMG * P_WIMG) + 2 * 8 * 32 * sizeof(float);
    auto kern = conv3x3_c32_planar_kernel<TERMS, OUTP>;
    if (int rc = eg_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), LDS_BYTES, "eg_conv3x3_c32_planar")) return rc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS_BYTES, st, a);
    return EG_OK;
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int32_t eg_conv3x3_c32_planar_gap_tiles(int32_t h, int32_t wdt) { return eg_cdiv(h, P_TH) * eg_cdiv(wdt, 32); }

extern "C" int eg_conv3x3_c32_planar(const void* x_planes, const float* w_packed, const float* bias, const float* scale, const float* shift,
                                     const float* gate, const void* residual_planes, void* y_planes, float* y_nhwc, float* gap_partial, int32_t batch,
                                     int32_t h, int32_t wdt, int32_t relu, int32_t precision, void* stream) {
    EG_REQUIRE(x_planes && w_packed && batch > 0 && h > 0 && wdt > 0 && ((y_planes == nullptr) != (y_nhwc == nullptr)), EG_ERR_BAD_ARG,
               "eg_conv3x3_c32_planar: null pointer, empty shape, or not exactly one output");
    EG_REQUIRE(precision == EG_PREC_BF16X3 || precision == EG_PREC_BF16, EG_ERR_UNSUPPORTED, "eg_conv3x3_c32_planar: split-bf16 / bf16 arithmetic only");
    EG_REQUIRE(!gate || residual_planes, EG_ERR_BAD_ARG, "eg_conv3x3_c32_planar: a gate needs a residual");
    EG_REQUIRE(eg_aligned16(x_planes) && eg_aligned16(w_packed) && (!y_planes || eg_aligned16(y_planes)) && (!y_nhwc || eg_aligned16(y_nhwc)) &&
                   (!residual_planes || eg_aligned16(residual_planes)) && residual_planes != y_planes && x_planes != y_planes,
               EG_ERR_ALIGN, "eg_conv3x3_c32_planar: 16-byte alignment; the output must not alias an input");
    EG_REQUIRE((int64_t)h * wdt * 8 < ((int64_t)1 << 31), EG_ERR_UNSUPPORTED, "eg_conv3x3_c32_planar: map too large");
    PlanarArgs a;
    a.x = reinterpret_cast<const bf8*>(x_planes); a.res = reinterpret_cast<const bf8*>(residual_planes); a.w = w_packed;
    a.bias = bias; a.scale = scale; a.shift = shift; a.gate = gate;
    a.yp = reinterpret_cast<bf8*>(y_planes); a.yf = y_nhwc; a.gap = gap_partial;
    a.H = h; a.W = wdt; a.relu = relu; a.relu2 = gate ? 1 : 0;
    a.tiles_x = eg_cdiv(wdt, 32); a.tiles_y = eg_cdiv(h, P_TH); a.tiles = a.tiles_x * a.tiles_y; a.total_tiles = a.tiles * batch;
    hipStream_t st = ST;
    EgProfScope prof((int64_t)32 * 1000000 + 32 * 1000 + 100 + 1, 2.0 * 9 * 32 * 32 * (double)h * wdt * batch, st);
    int grid = a.total_tiles < 256 ? a.total_tiles : 256;                 // one 8-wave workgroup per CU
    const int tpw = eg_cdiv(a.total_tiles, grid);
    grid = eg_cdiv(a.total_tiles, tpw);
    if (grid >= 8) grid = (int)eg_round_up(grid, 8);
    int rc;
    if (precision == EG_PREC_BF16X3) rc = y_planes ? launch_planar<3, true>(a, grid, st) : launch_planar<3, false>(a, grid, st);
    else rc = y_planes ? launch_planar<1, true>(a, grid, st) : launch_planar<1, false>(a, grid, st);
    if (rc) return rc;
    return eg_check_launch("conv3x3 (32 -> 32, planar)");
}

extern "C" int eg_stem_conv_planar(const float* x, const float* w9xc, const float* bias, const float* scale, const float* shift, void* y_planes,
                                   int32_t batch, int32_t h, int32_t wdt, void* stream) {
    EG_REQUIRE(x && w9xc && bias && scale && shift && y_planes && batch > 0 && h > 0 && wdt > 0, EG_ERR_BAD_ARG, "eg_stem_conv_planar: null pointer");
    hipLaunchKernelGGL(stem_conv_planar_kernel, dim3(batch * h), dim3(256), 3 * (wdt + 2) * sizeof(float), ST, x, w9xc, bias, scale, shift,
                       reinterpret_cast<bf8*>(y_planes), h, wdt);
    return eg_check_launch("stem_conv_planar");
}

extern "C" int eg_se_gate_pre_planar(const void* t1_planes, const float* gap_partial, int32_t tiles, const float* conv2_w, const float* scale2,
                                     const float* shift2, const float* w1, const float* b1, const float* w2, const float* b2, float* gate,
                                     int32_t batch, int32_t h, int32_t wdt, void* stream) {
    EG_REQUIRE(t1_planes && gap_partial && conv2_w && scale2 && shift2 && w1 && b1 && w2 && b2 && gate && batch > 0 && h > 1 && wdt > 1, EG_ERR_BAD_ARG,
               "eg_se_gate_pre_planar: null pointer or empty shape");
    const size_t smem = sizeof(float) * (5 * GPP_T + 9 * 32 + GPP_T + 32 + 4 + 4);
    hipLaunchKernelGGL(se_gate_pre_planar_kernel, dim3(batch), dim3(GPP_T), smem, ST, reinterpret_cast<const bf8*>(t1_planes), gap_partial, tiles, conv2_w,
                       scale2, shift2, w1, b1, w2, b2, gate, h, wdt);
    return eg_check_launch("se_gate_pre_planar");
}

extern "C" int eg_planar32_to_nhwc(const void* planes, float* y, int32_t batch, int32_t h, int32_t wdt, void* stream) {
    EG_REQUIRE(planes && y && batch > 0 && h > 0 && wdt > 0, EG_ERR_BAD_ARG, "eg_planar32_to_nhwc: bad argument");
    const size_t total = (size_t)batch * h * wdt * 32;
    hipLaunchKernelGGL(planar_to_nhwc_kernel, dim3((unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192)), dim3(256), 0, ST,
                       reinterpret_cast<const bf8*>(planes), y, h * wdt, total);
    return eg_check_launch("planar32_to_nhwc");
}
extern "C" int eg_nhwc_to_planar32(const float* x, void* planes, int32_t batch, int32_t h, int32_t wdt, void* stream) {
    EG_REQUIRE(x && planes && batch > 0 && h > 0 && wdt > 0 && eg_aligned16(x), EG_ERR_BAD_ARG, "eg_nhwc_to_planar32: bad argument");
    const size_t slots = (size_t)batch * 4 * h * wdt;
    hipLaunchKernelGGL(nhwc_to_planar_kernel, dim3((unsigned)((slots + 255) / 256 < 8192 ? (slots + 255) / 256 : 8192)), dim3(256), 0, ST, x,
                       reinterpret_cast<bf8*>(planes), h * wdt, slots);
    return eg_check_launch("nhwc_to_planar32");
}
