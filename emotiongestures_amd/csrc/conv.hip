// Audio tower kernels: stem conv, 3x3 implicit-GEMM conv on MFMA, SE gate, SE tail (+1x1 downsample).
// Reference semantics: Full_model/ResNetSE34V2.py:62-74, Full_model/ResNetBlocks.py:21-37,81-96.
//
// Data layout in HBM: activations NHWC fp32 (channel innermost, 128 B per pixel at C=32), so that
// the implicit-GEMM K axis (tap, channel) is contiguous per pixel and every global access is a
// 16-byte-per-lane coalesced load/store.
//
// conv3x3 kernel shape (one workgroup = 4 waves = TH x 32 output pixels x all output channels):
//   D[co][pix] = sum_{tap,ci} Wt[tap][ci][co] * X[pix + tap][ci]            (swapped operands)
//   A operand = weights  (lane: co = lane&15, k-quad = lane>>4), streamed from L2 (every workgroup
//                          reads the same <= 590 KB, so they stay cache resident),
//   B operand = pixels   (lane: pix = lane&15, k-quad = lane>>4), read from an LDS halo tile kept in a
//                          channel-quad planar image [ci/4][pixel][4]: a ds_read_b128 per lane, bank-conflict
//                          free for 16 consecutive pixels (plane stride = 0 mod 16 slots, stride 1; odd, stride 2),
//   D: each lane owns 4 consecutive output channels of one pixel -> one 16-byte NHWC store per tile.
#include <type_traits>
#include "common.h"

// (Measured and dropped in round 1: s_setprio around the MFMA cluster costs 20-50 % here -- hipcc stops interleaving the LDS
//  reads; a persistent 32->32 kernel with register-resident weights was 35 % slower than the tiled one.  Round 2's persistent 32->32
//  kernel below keeps the weights in LDS and the next tile's pixels in registers instead.)

namespace {

struct ConvArgs {
    const float* x; const float* w; const float* bias; const float* scale; const float* shift;
    float* y; float* gap;
    float* gap2 = nullptr;              // training forward: per-(clip, tile) channel sums of v*v beside `gap` (BatchNorm's variance without a pass over y)
    int H, W, Ho, Wo, cout, relu, nchw, tiles_x, tiles;
    // fused SE tail (SEBasicBlock.forward, ResNetBlocks.py:28-36, identity shortcut): v = relu(v * gate[b, co] + res[pixel, co])
    // applied after the BatchNorm affine; gate comes from se_gate_pre_kernel (computed BEFORE this convolution runs)
    const float* gate = nullptr; const float* res = nullptr; int relu2 = 0;
    // training: a per-INPUT-channel affine applied while the halo tile is staged (the BatchNorm in front of this convolution folded into its operand
    // staging: x' = x * in_scale[ci] + in_shift[ci] for pixels inside the image, zero padding stays zero) -- split-bf16 kernels only
    const float* in_scale = nullptr; const float* in_shift = nullptr;
    int wrow = 0;                       // channel-split launches: row length of the weight images (the convolution's padded output channels)
    const unsigned* res_bits = nullptr; // residual behind a ReLU mask: one bit per element of `res` (bit e & 31 of word e >> 5 = the nibble-per-float4 layout
                                        // se_tail_fwd_kernel writes): v += bit ? res : 0 -- the SE block's identity shortcut gradient dout * (out > 0), never stored
    const float* res_q = nullptr;       // DG2 only: a gradient on the dy grid ([B][H][W][cout]) added to the (even, even) phase -- the stride-2 1x1 shortcut's
    unsigned int* dbg = nullptr;        // diagnostic build only (EG_CONV32_STAMP=1): per-wave phase cycle sums of the persistent 32->32 kernel
};
// a quad of the residual, behind its ReLU bit mask when there is one (e = element index of the quad's first float, a multiple of 4)
__device__ __forceinline__ f4 residual_quad(const float* __restrict__ res, const unsigned* __restrict__ bits, size_t e) {
    f4 r = *reinterpret_cast<const f4*>(res + e);
    if (bits) {
        const unsigned nb = bits[e >> 5] >> (unsigned)(e & 28);
        r[0] = (nb & 1u) ? r[0] : 0.f;
        r[1] = (nb & 2u) ? r[1] : 0.f;
        r[2] = (nb & 4u) ? r[2] : 0.f;
        r[3] = (nb & 8u) ? r[3] : 0.f;
    }
    return r;
}

template <int S, int TH> struct ConvGeom {
    static constexpr int IH = (TH - 1) * S + 3;
    static constexpr int IW = 31 * S + 3;
    static constexpr int NPIX = IH * IW;
    static constexpr int PL = (S == 1) ? ((NPIX + 15) / 16 * 16) : (NPIX | 1);
    static constexpr int MT = TH * 2 / 4;
};

// Input gradient of a STRIDE-2 convolution (conv3x3_bf16_kernel, DG2): the tile is TH x 32 pixels of dy; a base pixel (i, j) owns the four dx pixels
// (2i + py, 2j + px) and needs dy at (i, j), (i, j + 1), (i + 1, j), (i + 1, j + 1): halo of one row / column on the HIGH side only.
template <int TH> struct ConvGeomDG2 {
    static constexpr int IH = TH + 1;
    static constexpr int IW = 33;
    static constexpr int NPIX = IH * IW;
    static constexpr int PL = (NPIX + 15) / 16 * 16;
};

// ---- fp32 MFMA path -------------------------------------------------------------------------
template <int CIN, int NT, int S, int TH>
__global__ __launch_bounds__(256) void conv3x3_f32_kernel(ConvArgs a) {
    using G = ConvGeom<S, TH>;
    constexpr int COUTP = NT * 16, IW = G::IW, NPIX = G::NPIX, PL = G::PL, MT = G::MT;
    __shared__ f4 tile[8 * PL];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, kq = lane >> 4;
    // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs, so give each XCD a contiguous run of tiles
    // (whole clips at B >= 8); halo rows / columns shared by neighbouring tiles then hit in that XCD's L2.
    int tile_id = blockIdx.x, b = blockIdx.y;
    {
        const int total = gridDim.x * gridDim.y;
        if ((total & 7) == 0) {
            const int lin = blockIdx.y * gridDim.x + blockIdx.x;
            const int log = (lin & 7) * (total >> 3) + (lin >> 3);
            b = log / (int)gridDim.x;
            tile_id = log - b * (int)gridDim.x;
        }
    }
    const int ty = tile_id / a.tiles_x, tx = tile_id - ty * a.tiles_x;
    const int oy0 = ty * TH, ox0 = tx * 32;
    const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;

    int pbase[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int id = wave * MT + t;
        pbase[t] = ((id >> 1) * S) * IW + ((id & 1) * 16 + li) * S;
    }
    f4 acc[MT][NT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[t][n] = (f4){0.f, 0.f, 0.f, 0.f};

    const f4* __restrict__ w4 = reinterpret_cast<const f4*>(a.w);
    const float* __restrict__ xb = a.x + (size_t)b * a.H * a.W * CIN;

    for (int chunk = 0; chunk < CIN / 32; ++chunk) {
        if (chunk) __syncthreads();
        // stage the (IH x IW) x 32-channel halo tile: 8 consecutive lanes = 8 consecutive pixels of one
        // channel quad (conflict-free ds_write_b128); a wave still covers 8 pixels x 128 contiguous bytes.
        for (int idx = tid; idx < ((NPIX + 7) / 8) * 64; idx += 256) {
            const int p = (idx >> 6) * 8 + (idx & 7), cq = (idx >> 3) & 7;
            if (p < NPIX) {
                const int iy = p / IW, ix = p - iy * IW;
                const int gy = iy0 + iy, gx = ix0 + ix;
                f4 v = (f4){0.f, 0.f, 0.f, 0.f};
                if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                    v = *reinterpret_cast<const f4*>(xb + ((size_t)gy * a.W + gx) * CIN + chunk * 32 + cq * 4);
                tile[cq * PL + p] = v;
            }
        }
        __syncthreads();
        const f4* wp = w4 + (size_t)(chunk * 8 + kq) * COUTP + li;
#pragma unroll 1
        for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int tap = kh * 3 + kw, toff = kh * IW + kw;
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    f4 wv[NT], xv[MT];
#pragma unroll
                    for (int n = 0; n < NT; ++n) wv[n] = wp[(size_t)(tap * (CIN / 4) + g * 4) * COUTP + n * 16];
#pragma unroll
                    for (int t = 0; t < MT; ++t) xv[t] = tile[(g * 4 + kq) * PL + pbase[t] + toff];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int t = 0; t < MT; ++t)
#pragma unroll
                            for (int n = 0; n < NT; ++n)
                                acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[n][j], xv[t][j], acc[t][n], 0, 0, 0);
                }
            }
        }
    }

    // ---- epilogue: v = acc + bias; relu?; v*scale + shift; NHWC f4 store / NCHW scalar store; SE sums
    f4 gsum[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) gsum[n] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int co = n * 16 + kq * 4;
        const f4 bi = a.bias ? *reinterpret_cast<const f4*>(a.bias + co) : (f4){0.f, 0.f, 0.f, 0.f};
        const f4 sc = a.scale ? *reinterpret_cast<const f4*>(a.scale + co) : (f4){1.f, 1.f, 1.f, 1.f};
        const f4 sh = a.shift ? *reinterpret_cast<const f4*>(a.shift + co) : (f4){0.f, 0.f, 0.f, 0.f};
        const f4 gt = (a.gate && co < a.cout) ? *reinterpret_cast<const f4*>(a.gate + (size_t)b * a.cout + co) : (f4){1.f, 1.f, 1.f, 1.f};
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const int id = wave * MT + t;
            const int oy = oy0 + (id >> 1), ox = ox0 + (id & 1) * 16 + li;
            const bool valid = (oy < a.Ho) && (ox < a.Wo);
            f4 v = acc[t][n] + bi;
            if (a.relu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            v = v * sc + sh;
            if (a.gate) v = v * gt;
            if (a.res && valid && co < a.cout) v += residual_quad(a.res, a.res_bits, (((size_t)b * a.Ho + oy) * a.Wo + ox) * a.cout + co);
            if (a.relu2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            if (valid) {
                if (!a.nchw) {
                    if (co < a.cout)
                        *reinterpret_cast<f4*>(a.y + (((size_t)b * a.Ho + oy) * a.Wo + ox) * a.cout + co) = v;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (co + r < a.cout)
                            a.y[((size_t)b * a.cout + co + r) * a.Ho * a.Wo + (size_t)oy * a.Wo + ox] = v[r];
                }
                gsum[n] += v;
            }
        }
    }
    if (a.gap) {
        float* sred = reinterpret_cast<float*>(tile);
        __syncthreads();                                      // all waves are done reading the halo tile
#pragma unroll
        for (int n = 0; n < NT; ++n) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s = gsum[n][r];
                s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64);
                s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64);
                if (li == 0) sred[wave * COUTP + n * 16 + kq * 4 + r] = s;
            }
        }
        __syncthreads();
        if (tid < a.cout) {
            const float s = (sred[tid] + sred[COUTP + tid]) + (sred[2 * COUTP + tid] + sred[3 * COUTP + tid]);
            a.gap[((size_t)b * a.tiles + tile_id) * a.cout + tid] = s;
        }
    }
}

// ---- split-bf16 MFMA path (EG_PREC_BF16X3 / EG_PREC_BF16) --------------------------------------
// Same tiling, but BOTH operands come from LDS and the weight stream is asynchronous:
//  * halo tile: split while it is staged, hi = bf16(x), lo = bf16(x - hi), kept as two channel-octet planar images
//    [ci/8][pixel][8 bf16] (same 16-byte slot structure as the fp32 image => conflict-free ds_read_b128);
//  * weights: pre-split on the host into [tap][ci/8][co][8] (hi image, lo image).  The 4 octets of one (tap, 32-channel
//    chunk) are one contiguous 64*COUTP-byte run, copied by global_load_lds (16 B/lane, no VGPR, no VALU) into a
//    3-slot LDS ring two steps ahead; one barrier per step (placed mid-step, counted vmcnt: see the schedule comment
//    in the kernel) retires the copy of step s+1 while the copy of step s+2 stays in flight;
//  * waves tile the workgroup's TH x 32 pixels x COUTP channels as WM (pixels) x WN (channels); per step and
//    (pixel tile, channel tile) pair:  acc += Whi*Xhi + Whi*Xlo + Wlo*Xhi   (3 x v_mfma_f32_16x16x32_bf16).
// Without the LDS weight ring every wave re-read all weights from L1/L2 (170 B/clk/CU demanded at C=128 vs 64 B/clk
// of L1): the r01a profile's 115 us per 128->128 launch.

// SPLIT: the workgroup computes the NTT * 16 output channels starting at blockIdx.z * NTT * 16 of a convolution with a.cout channels in all (its
// weight images' row length `a.wrow`): small batches of the 64- and 128-channel stages have fewer pixel tiles than the chip has CUs (one clip: 8 tiles of
// 128 -> 128), so the channels are spread over workgroups too.  Same pixel -> (wave, tile, lane) map (WM, MT) and the same K order as the unsplit
// instantiation: every output element and every pooling partial is bitwise the same, whichever the launch function picks.
//
// DG2: the INPUT GRADIENT of a stride-2 convolution (F.conv2d's dgrad for the two `_make_layer` entry convolutions, ResNetSE34V2.py:40-55) on the same
// machinery, phase-decomposed: x here is dy [B][H][W][CIN] (CIN = the convolution's output channels: the contraction), y is dx [B][Ho][Wo][cout],
// Ho = the convolution's input height.  dx[2i + py][2j + px] only receives the taps whose parity matches: (even, even) 1 tap, (even, odd) / (odd, even)
// 2, (odd, odd) 4 -- 9 tap-products per FOUR output pixels instead of the 36 a zero-upsampled stride-1 convolution would issue.  One step is still one
// (tap, 32-channel chunk) with the same weight ring; it accumulates into the phase the tap belongs to (4 x the accumulators, hence the smaller
// tiles of the launch table).  The weight images are those of the rotated, transposed filter the stride-1 input gradients use (pack flip):
// image tap t' is the filter's tap (2 - t'/3, 2 - t'%3), which reads dy at (i + (t'/3 == 2), j + (t'%3 == 2)).
template <int CIN, int NTT, int S, int TH, int WM, int WN, int TERMS, int RING, bool SPLIT = false, bool DG2 = false>
__global__ __launch_bounds__(256, 2) void conv3x3_bf16_kernel(ConvArgs a, const bf8* __restrict__ whi,
                                                           const bf8* __restrict__ wlo) {
    static_assert(!DG2 || (S == 1 && !SPLIT), "DG2 tiles the dy grid with unit stride");
    using G = typename std::conditional<DG2, ConvGeomDG2<TH>, ConvGeom<S, TH>>::type;
    constexpr int PH = DG2 ? 4 : 1;                 // output phases (accumulator sets)
    constexpr int COUTP = NTT * 16, IW = G::IW, NPIX = G::NPIX, PL = G::PL;
    constexpr int MT = TH * 2 / WM, NT = NTT / WN;
    constexpr int NIMG = (TERMS == 3) ? 2 : 1;
    constexpr int TILE = NIMG * 4 * PL;             // bf8 slots of the halo tile
    constexpr int WIMG = 4 * COUTP;                 // bf8 slots of one weight image of one step
    constexpr int WBUF = NIMG * WIMG;
    constexpr int NSTEP = (CIN / 32) * 9;
    static_assert(WM * WN == 4 && MT * WM == TH * 2 && NT * WN == NTT, "wave tiling");
    extern __shared__ __attribute__((aligned(16))) bf8 lds[];      // TILE + RING * WBUF slots (dynamic: can exceed 64 KB)
    bf8* tile = lds;
    bf8* wring = lds + TILE;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, kq = lane >> 4;
    const int wm = wave / WN, wn = wave % WN;
    // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs, so give each XCD a contiguous run of tiles
    // (whole clips at B >= 8); halo rows / columns shared by neighbouring tiles then hit in that XCD's L2.
    int tile_id = blockIdx.x, b = blockIdx.y;
    const int c0 = SPLIT ? (int)blockIdx.z * COUTP : 0;        // first output channel of this workgroup
    if (!SPLIT) {
        const int total = gridDim.x * gridDim.y;
        if ((total & 7) == 0) {
            const int lin = blockIdx.y * gridDim.x + blockIdx.x;
            const int log = (lin & 7) * (total >> 3) + (lin >> 3);
            b = log / (int)gridDim.x;
            tile_id = log - b * (int)gridDim.x;
        }
    }
    const int ty = tile_id / a.tiles_x, tx = tile_id - ty * a.tiles_x;
    const int oy0 = ty * TH, ox0 = tx * 32;
    const int iy0 = DG2 ? oy0 : oy0 * S - 1, ix0 = DG2 ? ox0 : ox0 * S - 1;

    int pbase[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int id = wm * MT + t;
        pbase[t] = ((id >> 1) * S) * IW + ((id & 1) * 16 + li) * S;
    }
    f4 acc[PH][MT][NT];
#pragma unroll
    for (int ph = 0; ph < PH; ++ph)
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[ph][t][n] = (f4){0.f, 0.f, 0.f, 0.f};
    const float* __restrict__ xb = a.x + (size_t)b * a.H * a.W * CIN;

    // one step's weights = NIMG runs of WIMG slots; 64-slot (1 KiB) pieces are dealt round-robin to the 4 waves
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    constexpr int PIECES = WIMG / 64;
    // (image, piece) pairs are dealt round-robin to the 4 waves as one flat list, so that every wave issues the same number
    // of copies per step whenever NIMG * PIECES is a multiple of 4 (then the per-step wait can be a counted vmcnt)
    constexpr int GTOT = NIMG * PIECES;
    constexpr bool COUNTED = (GTOT % 4 == 0);
    constexpr int GW = GTOT / 4;
    auto issue_weights = [&](int step, int buf) {
        const int chunk = step / 9, tap = step - chunk * 9;
        if constexpr (SPLIT) {
            // the slice's 4 octet rows of COUTP slots are not contiguous in the image (row length a.wrow): per-lane source addresses, the LDS side
            // stays one contiguous KiB per wave-instruction
            static_assert(COUNTED, "split instantiations deal whole pieces to the four waves");
            const size_t rbase = ((size_t)tap * (CIN / 8) + chunk * 4) * a.wrow + c0;
#pragma unroll
            for (int p = 0; p < GW; ++p) {
                const int flat = p * 4 + wave_u;
                const int img = flat / PIECES, piece = flat - img * PIECES;
                const int slot = piece * 64 + lane, oct = slot / COUTP, co = slot - oct * COUTP;
                const bf8* src = (img ? wlo : whi) + rbase + (size_t)oct * a.wrow + co;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(wring + buf * WBUF + img * WIMG + piece * 64), 16, 0, 0);
            }
            return;
        }
        const size_t gbase = ((size_t)tap * (CIN / 8) + chunk * 4) * COUTP;
        if constexpr (PIECES % 4 == 0) {        // every wave copies PIECES / 4 pieces of each image (image known at compile time)
#pragma unroll
            for (int img = 0; img < NIMG; ++img) {
                const bf8* src = (img ? wlo : whi) + gbase;
#pragma unroll
                for (int p = 0; p < PIECES / 4; ++p) {
                    const int piece = p * 4 + wave_u;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 64 + lane),
                                                     (__attribute__((address_space(3))) void*)(wring + buf * WBUF + img * WIMG + piece * 64),
                                                     16, 0, 0);
                }
            }
        } else {                                // few pieces: deal the flat (image, piece) list
#pragma unroll
            for (int p = 0; p < (GTOT + 3) / 4; ++p) {
                const int flat = p * 4 + wave_u;
                if (COUNTED || flat < GTOT) {
                    const int img = flat / PIECES, piece = flat - img * PIECES;
                    const bf8* src = (img ? wlo : whi) + gbase;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 64 + lane),
                                                     (__attribute__((address_space(3))) void*)(wring + buf * WBUF + img * WIMG + piece * 64),
                                                     16, 0, 0);
                }
            }
        }
    };
    // halo tile staging, split in two so the HBM latency of chunk c+1 hides under the 9 taps of chunk c:
    //   load_tile: (IH x IW) pixels x 32 channels -> registers (one lane = one pixel x 8 channels per iteration)
    //   store_tile: split to (hi, lo) bf16 and write the channel-octet planar LDS image
    constexpr int NIT = (((NPIX + 7) / 8) * 32 + 255) / 256;
    f4 pv[NIT][2];
    // per-lane staging roles are chunk independent: compute the global element offset (32-bit; -1 = zero padding / unused)
    // and the LDS slot once, so the per-chunk loops carry no address arithmetic
    int goff[NIT], lslot[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = tid + it * 256;
        const int p = (idx >> 5) * 8 + (idx & 7), oc = (idx >> 3) & 3;
        const int iy = p / IW, ix = p - iy * IW;
        const int gy = iy0 + iy, gx = ix0 + ix;
        const bool inside = p < NPIX && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        goff[it] = inside ? (gy * a.W + gx) * CIN + oc * 8 : -1;
        lslot[it] = p < NPIX ? oc * PL + p : -1;
    }
    auto load_tile = [&](int chunk) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            pv[it][0] = pv[it][1] = (f4){0.f, 0.f, 0.f, 0.f};
            if (goff[it] >= 0) {
                const float* src = xb + goff[it] + chunk * 32;
                pv[it][0] = *reinterpret_cast<const f4*>(src);
                pv[it][1] = *reinterpret_cast<const f4*>(src + 4);
            }
        }
    };
    const int oc8s = ((tid >> 3) & 3) * 8;          // this thread's channel octet inside a 32-channel chunk (the same for all its iterations)
    auto store_tile = [&](int chunk) {
        f4 s0 = (f4){1.f, 1.f, 1.f, 1.f}, s1 = s0, h0 = (f4){0.f, 0.f, 0.f, 0.f}, h1 = h0;
        if (a.in_scale) {
            s0 = *reinterpret_cast<const f4*>(a.in_scale + chunk * 32 + oc8s); s1 = *reinterpret_cast<const f4*>(a.in_scale + chunk * 32 + oc8s + 4);
            h0 = *reinterpret_cast<const f4*>(a.in_shift + chunk * 32 + oc8s); h1 = *reinterpret_cast<const f4*>(a.in_shift + chunk * 32 + oc8s + 4);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (lslot[it] >= 0) {
                if (a.in_scale && goff[it] >= 0) {          // inside the image only: the zero padding of the normalised map stays zero
                    pv[it][0] = pv[it][0] * s0 + h0;
                    pv[it][1] = pv[it][1] * s1 + h1;
                }
                bf8 hi, lo;
                split_octet<TERMS == 3>(pv[it][0], pv[it][1], hi, lo);
                tile[lslot[it]] = hi;
                if (TERMS == 3) tile[4 * PL + lslot[it]] = lo;
            }
        }
    };

    // glds instructions this wave issues per step (pieces are dealt round-robin, so it can differ by wave)
    // Schedule (3-slot weight ring, fragments double-buffered in registers, taps fully unrolled):
    //   step s:  issue the weight copy of step s+2  ->  read step s+1's fragments from LDS (in flight during ...)
    //            ... the MFMAs of step s  ->  vmcnt(0) + barrier (copies of s+1, s+2 landed; slot of s is free)
    // so neither the LDS fragment reads nor the global->LDS weight copies sit on the MFMA critical path.
    struct Frags { bf8 wh[NT], wl[NT], xh[MT], xl[MT]; };
    auto read_w = [&](Frags& f, int tap) {
        const bf8* Wh = wring + (tap % 3) * WBUF + kq * COUTP + wn * NT * 16 + li;      // step % 3 == tap % 3 (9 taps per chunk)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            f.wh[n] = Wh[n * 16];
            if (TERMS == 3) f.wl[n] = Wh[WIMG + n * 16];
        }
    };
    auto read_x = [&](Frags& f, int tap) {
        const int kh = tap / 3, kw = tap - kh * 3;
        const int toff = DG2 ? (kh == 2 ? IW : 0) + (kw == 2 ? 1 : 0) : kh * IW + kw;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            f.xh[t] = tile[kq * PL + pbase[t] + toff];
            if (TERMS == 3) f.xl[t] = tile[(4 + kq) * PL + pbase[t] + toff];
        }
    };
    static_assert(MT % 2 == 0, "a step's MFMAs are issued as two halves (pixel tiles)");
    auto mfma_half = [&](const Frags& f, int half, int tap) {
        const int ph = DG2 ? (tap / 3 != 1 ? 2 : 0) + (tap % 3 != 1 ? 1 : 0) : 0;      // (py, px): a centre row / column tap lands on even rows / columns
#pragma unroll
        for (int t = half * (MT / 2); t < (half + 1) * (MT / 2); ++t)
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                if (TERMS == 3) {
                    acc[ph][t][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.wl[n], f.xh[t], acc[ph][t][n], 0, 0, 0);
                    acc[ph][t][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.wh[n], f.xl[t], acc[ph][t][n], 0, 0, 0);
                }
                acc[ph][t][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.wh[n], f.xh[t], acc[ph][t][n], 0, 0, 0);
            }
    };
    static_assert(RING == 3, "schedule below assumes a 3-slot ring");
    // Schedule of step s (one tap of one 32-channel chunk), one barrier per step, placed between the two MFMA halves:
    //   issue copy W(s+2) -> slot (s+2)%3        (its previous content, W(s-1), was last read before the barrier of step s-1)
    //   read the pixel fragments of step s+1     (hidden under ...)
    //   MFMAs of step s, first half
    //   wait: W(s+1) landed [counted vmcnt: W(s+2) stays in flight], LDS reads so far complete; barrier
    //   read the weight fragments of step s+1    (hidden under ...)
    //   MFMAs of step s, second half
    // so a weight copy has 1.5 steps of MFMAs to hide under and the fragment reads half a step.
    auto mid_sync = [&](bool copy_in_flight) {
        if constexpr (COUNTED) {
            __builtin_amdgcn_sched_barrier(0);  // keep the first-half MFMAs in front of the wait (the scheduler would sink them)
            if (copy_in_flight) wait_vmcnt_imm<GW>(); else wait_vmcnt_imm<0>();
            wait_lgkmcnt0();
            wg_barrier();
        } else {
            __syncthreads();
        }
    };
    issue_weights(0, 0);
    if (NSTEP > 1) issue_weights(1, 1);
    load_tile(0);
    Frags fr[2];
#pragma unroll 1
    for (int chunk = 0; chunk < CIN / 32; ++chunk) {
        store_tile(chunk);                      // the tile is free: its last reads completed before the barrier of the previous tap 8
        if (COUNTED && chunk > 0) {              // W of this chunk's tap 0 landed at that barrier too; only W of tap 1 is in flight
            wait_vmcnt_imm<GW>();
            wait_lgkmcnt0();                    // tile writes visible
            wg_barrier();
        } else {
            __syncthreads();                    // tile visible; every weight copy issued so far has landed
        }
        if (chunk + 1 < CIN / 32) load_tile(chunk + 1);
        read_w(fr[0], 0);
        read_x(fr[0], 0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int step = chunk * 9 + tap;
            if (step + 2 < NSTEP) issue_weights(step + 2, (tap + 2) % 3);
            mfma_half(fr[tap & 1], 0, tap);
            mid_sync(step + 2 < NSTEP);
            if (tap < 8) { read_w(fr[(tap + 1) & 1], tap + 1); read_x(fr[(tap + 1) & 1], tap + 1); }
            mfma_half(fr[tap & 1], 1, tap);
        }
    }

    if constexpr (DG2) {
        // four phases: dx[2 * by + py][2 * bx + px] for the base pixel (by, bx) of the dy grid; the (even, even) phase also takes the gradient that
        // arrives on the dy grid (the stride-2 1x1 shortcut reads exactly those pixels of x)
        const size_t ohw = (size_t)a.Ho * a.Wo;
        float* __restrict__ yb = a.y + (size_t)b * ohw * a.cout;
        const float* __restrict__ rq = a.res_q ? a.res_q + (size_t)b * a.H * a.W * a.cout : nullptr;
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
            const int py = ph >> 1, px = ph & 1;
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const int id = wm * MT + t;
                const int by = oy0 + (id >> 1), bx = ox0 + (id & 1) * 16 + li;
                const int oy = 2 * by + py, ox = 2 * bx + px;
                if (by < a.H && bx < a.W && oy < a.Ho && ox < a.Wo) {
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        const int co = (wn * NT + n) * 16 + kq * 4;
                        if (co < a.cout) {
                            f4 v = acc[ph][t][n];
                            if (ph == 0 && rq) v += *reinterpret_cast<const f4*>(rq + ((size_t)by * a.W + bx) * a.cout + co);
                            *reinterpret_cast<f4*>(yb + ((size_t)oy * a.Wo + ox) * a.cout + co) = v;
                        }
                    }
                }
            }
        }
        return;
    }

    // epilogue: v = acc + bias; relu; v*scale + shift.  Pixel offsets are 32-bit and computed once per pixel tile; the
    // per-image base is a scalar.  NHWC: one 16-byte store per (pixel tile, channel tile); NCHW (final_conv1 only): 4 stores.
    f4 gsum[NT], gsq[NT];
    int pixo[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int id = wm * MT + t;
        const int oy = oy0 + (id >> 1), ox = ox0 + (id & 1) * 16 + li;
        pixo[t] = (oy < a.Ho && ox < a.Wo) ? oy * a.Wo + ox : -1;
    }
    const int hw = a.Ho * a.Wo;
    float* __restrict__ yb = a.y + (size_t)b * hw * a.cout;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        gsum[n] = (f4){0.f, 0.f, 0.f, 0.f};
        gsq[n] = gsum[n];
        const int co = c0 + (wn * NT + n) * 16 + kq * 4;
        const f4 bi = a.bias ? *reinterpret_cast<const f4*>(a.bias + co) : (f4){0.f, 0.f, 0.f, 0.f};
        const f4 sc = a.scale ? *reinterpret_cast<const f4*>(a.scale + co) : (f4){1.f, 1.f, 1.f, 1.f};
        const f4 sh = a.shift ? *reinterpret_cast<const f4*>(a.shift + co) : (f4){0.f, 0.f, 0.f, 0.f};
        const f4 gt = (a.gate && co < a.cout) ? *reinterpret_cast<const f4*>(a.gate + (size_t)b * a.cout + co) : (f4){1.f, 1.f, 1.f, 1.f};
        const bool rb = a.res != nullptr;
        const size_t rbase = (size_t)b * hw * a.cout;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            f4 v = acc[0][t][n] + bi;
            if (a.relu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            v = v * sc + sh;
            if (a.gate) v = v * gt;
            if (rb && pixo[t] >= 0 && co < a.cout) v += residual_quad(a.res, a.res_bits, rbase + (size_t)pixo[t] * a.cout + co);
            if (a.relu2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            if (pixo[t] >= 0) {
                if (!a.nchw) {
                    if (co < a.cout) *reinterpret_cast<f4*>(yb + pixo[t] * a.cout + co) = v;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (co + r < a.cout) yb[(co + r) * hw + pixo[t]] = v[r];
                }
                gsum[n] += v;
                if (a.gap2) gsq[n] += v * v;
            }
        }
    }
    if (a.gap) {            // no LDS reads follow the last step's barrier: the LDS is free for the reduction
        float* sred = reinterpret_cast<float*>(lds);        // [WM][COUTP] sums, then [WM][COUTP] sums of squares
#pragma unroll
        for (int n = 0; n < NT; ++n) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s = gsum[n][r];
                s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64);
                s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64);
                if (li == 0) sred[wm * COUTP + (wn * NT + n) * 16 + kq * 4 + r] = s;
                if (a.gap2) {
                    float q = gsq[n][r];
                    q += __shfl_xor(q, 1, 64); q += __shfl_xor(q, 2, 64);
                    q += __shfl_xor(q, 4, 64); q += __shfl_xor(q, 8, 64);
                    if (li == 0) sred[(WM + wm) * COUTP + (wn * NT + n) * 16 + kq * 4 + r] = q;
                }
            }
        }
        __syncthreads();
        if (tid < (SPLIT ? COUTP : a.cout)) {
            float s = 0.f;
#pragma unroll
            for (int m = 0; m < WM; ++m) s += sred[m * COUTP + tid];
            a.gap[((size_t)b * a.tiles + tile_id) * a.cout + c0 + tid] = s;
            if (a.gap2) {
                float q = 0.f;
#pragma unroll
                for (int m = 0; m < WM; ++m) q += sred[(WM + m) * COUTP + tid];
                a.gap2[((size_t)b * a.tiles + tile_id) * a.cout + c0 + tid] = q;
            }
        }
    }
}

// ---- 32 -> 32 channels, stride 1: persistent variant (the HBM-bound first stage) -------------------------------------------------
// The tiled kernel above gives a 32-channel workgroup 108 MFMAs per wave between a cold halo load and its stores: its lifetime is
// almost all memory latency, hidden only by co-resident workgroups (3.3-3.8 TB/s).  Here a workgroup walks a run of consecutive
// 4 x 32-pixel tiles: all 9 taps' weights (36 KB of hi/lo images) are copied to LDS once, the NEXT tile's halo pixels (and this tile's
// residual, for the fused SE tail) are in flight in registers while the 9 taps of the current tile run, and the taps need no barrier
// (weights and tile are static): two barriers per tile instead of ten.  Same arithmetic and summation order as the tiled kernel.
// STAMP: s_memtime stamps around the six phases of a tile (a separate diagnostic instantiation; the production kernel executes none)
template <int TERMS, int TH = 4, bool STAMP = false>
__global__ __launch_bounds__(256, 2) void conv3x3_c32_persistent_kernel(ConvArgs a, const bf8* __restrict__ whi, const bf8* __restrict__ wlo,
                                                                     int total_tiles, int tiles_per_wg) {
    using G = ConvGeom<1, TH>;
    constexpr int CIN = 32, IW = G::IW, NPIX = G::NPIX, PL = G::PL, MT = TH / 2, NT = 2;
    constexpr int NIMG = (TERMS == 3) ? 2 : 1;
    constexpr int TILE = NIMG * 4 * PL, WTAP = 4 * 32, WIMG = 9 * WTAP;
    static_assert(PL - NPIX >= 8 || TH == 4, "the gap scratch of the 8-row variant lives in the planes' padding slots");
    extern __shared__ __attribute__((aligned(16))) bf8 lds[];      // tile | weights (hi [tap][octet][co], lo) | gap scratch
    bf8* tile = lds;
    bf8* wl = lds + TILE;
    // gap scratch [4 waves][32 floats] (+ the same again for the sums of squares): behind the weights (4-row tiles), or -- 8-row tiles: tile + weights fill exactly half a CU's LDS -- in the
    // padding slots NPIX .. PL-1 of the first four channel-octet planes, which no staging write and no fragment read touches
    float* sred_base = reinterpret_cast<float*>(lds + TILE + NIMG * WIMG);
    const int sred_pitch = (TH == 4) ? 32 : PL * 4;               // floats between two waves' scratch rows
    float* sred = (TH == 4) ? sred_base : reinterpret_cast<float*>(tile + NPIX);

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, kq = lane >> 4;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // XCD-aware: workgroups are dealt round-robin to the 8 XCDs; XCD k walks a contiguous eighth of the tile list
    int wg = blockIdx.x;
    if ((gridDim.x & 7) == 0) wg = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    // Walk order: workgroup w takes tiles w, w + G, w + 2G, ... (G = grid size), so that at any moment the G resident workgroups work on G
    // CONSECUTIVE tiles (each XCD on a contiguous run of G / 8): the halo rows shared by vertically adjacent tiles are fetched by neighbours at
    // about the same time and hit in that XCD's L2.  (A contiguous run per workgroup re-read them from memory 4 tiles later: 260 MB read per
    // launch at the memory-side counters against 196 MB for the tiled kernel.)
    const int NWG = gridDim.x;
    const int t_begin = wg, t_end = total_tiles;
    (void)tiles_per_wg;
    if (t_begin >= t_end) return;

    // all weights -> LDS, once: NIMG x 18 pieces of 64 slots, dealt round-robin to the 4 waves
#pragma unroll
    for (int img = 0; img < NIMG; ++img) {
        const bf8* src = img ? wlo : whi;
#pragma unroll
        for (int p = 0; p < 5; ++p) {
            const int piece = p * 4 + wave_u;
            if (piece < WIMG / 64)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 64 + lane),
                                                 (__attribute__((address_space(3))) void*)(wl + img * WIMG + piece * 64), 16, 0, 0);
        }
    }

    // staging roles (tile independent): pixel p of the 6 x 34 halo tile, channel octet oc
    constexpr int NIT = (((NPIX + 7) / 8) * 32 + 255) / 256;
    int piy[NIT], pix[NIT], lslot[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = tid + it * 256;
        const int p = (idx >> 5) * 8 + (idx & 7), oc = (idx >> 3) & 3;
        piy[it] = p / IW;
        pix[it] = p - piy[it] * IW;
        lslot[it] = p < NPIX ? oc * PL + p : -1;
    }
    const int oc8 = ((tid >> 3) & 3) * 8;
    f4 isc[2] = {(f4){1.f, 1.f, 1.f, 1.f}, (f4){1.f, 1.f, 1.f, 1.f}}, ish[2] = {(f4){0.f, 0.f, 0.f, 0.f}, (f4){0.f, 0.f, 0.f, 0.f}};
    if (a.in_scale) {
        isc[0] = *reinterpret_cast<const f4*>(a.in_scale + oc8); isc[1] = *reinterpret_cast<const f4*>(a.in_scale + oc8 + 4);
        ish[0] = *reinterpret_cast<const f4*>(a.in_shift + oc8); ish[1] = *reinterpret_cast<const f4*>(a.in_shift + oc8 + 4);
    }
    f4 pv[NIT][2];
    // tile coordinates advance incrementally along the walk (an integer division per use costs ~30 VALU instructions; this kernel
    // issues 4 non-MFMA VALU instructions per MFMA as it is: profiles/r02k_pmc_conv32_persistent.txt)
    struct Coord { int b, ty, tx; };
    const int tiles_y = a.tiles / a.tiles_x;
    const int adv_b = NWG / a.tiles, adv_r = NWG - adv_b * a.tiles, adv_ty = adv_r / a.tiles_x, adv_tx = adv_r - adv_ty * a.tiles_x;
    auto advance = [&](Coord& c) {                // + NWG tiles, without a division
        c.tx += adv_tx;
        if (c.tx >= a.tiles_x) { c.tx -= a.tiles_x; ++c.ty; }
        c.ty += adv_ty;
        if (c.ty >= tiles_y) { c.ty -= tiles_y; ++c.b; }
        c.b += adv_b;
    };
    auto load_tile = [&](const Coord& c) {
        const int b = c.b;
        const int iy0 = c.ty * TH - 1, ix0 = c.tx * 32 - 1;
        const float* __restrict__ xb = a.x + (size_t)b * a.H * a.W * CIN;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            pv[it][0] = pv[it][1] = (f4){0.f, 0.f, 0.f, 0.f};
            const int gy = iy0 + piy[it], gx = ix0 + pix[it];
            if (lslot[it] >= 0 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
                const float* src = xb + (gy * a.W + gx) * CIN + oc8;
                pv[it][0] = *reinterpret_cast<const f4*>(src);
                pv[it][1] = *reinterpret_cast<const f4*>(src + 4);
                if (a.in_scale) {                   // the BatchNorm in front of this convolution, folded into the staging (in-image pixels only)
                    pv[it][0] = pv[it][0] * isc[0] + ish[0];
                    pv[it][1] = pv[it][1] * isc[1] + ish[1];
                }
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (lslot[it] >= 0) {
                bf8 hi, lo;
                split_octet<TERMS == 3>(pv[it][0], pv[it][1], hi, lo);
                tile[lslot[it]] = hi;
                if (TERMS == 3) tile[4 * PL + lslot[it]] = lo;
            }
        }
    };
    int pbase[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int id = wave * MT + t;
        pbase[t] = (id >> 1) * IW + (id & 1) * 16 + li;
    }
    // (Keeping the hi weight fragments of all 9 taps in registers -- 72 VGPRs, a quarter less LDS fragment traffic -- measured no faster
    //  in the step and 7 % slower alone: the fragment reads are not what bounds this kernel.)
    struct Frags { bf8 wh[NT], wlf[NT], xh[MT], xl[MT]; };
    auto read_frags = [&](Frags& f, int tap) {
        const bf8* Wh = wl + tap * WTAP + kq * 32 + li;
        const int kh = tap / 3, kw = tap - kh * 3, toff = kh * IW + kw;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            f.wh[n] = Wh[n * 16];
            if (TERMS == 3) f.wlf[n] = Wh[WIMG + n * 16];
        }
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            f.xh[t] = tile[kq * PL + pbase[t] + toff];
            if (TERMS == 3) f.xl[t] = tile[(4 + kq) * PL + pbase[t] + toff];
        }
    };

    const int hw = a.Ho * a.Wo;
    // channel-wise epilogue constants of this lane's 2 x 4 output channels
    f4 bi[NT], sc[NT], sh[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int co = n * 16 + kq * 4;
        bi[n] = a.bias ? *reinterpret_cast<const f4*>(a.bias + co) : (f4){0.f, 0.f, 0.f, 0.f};
        sc[n] = a.scale ? *reinterpret_cast<const f4*>(a.scale + co) : (f4){1.f, 1.f, 1.f, 1.f};
        sh[n] = a.shift ? *reinterpret_cast<const f4*>(a.shift + co) : (f4){0.f, 0.f, 0.f, 0.f};
    }

    f4 rsn[MT][NT];
    auto load_res = [&](const Coord& c) {
        const int ty = c.ty, tx = c.tx;
        const size_t rbase = (size_t)c.b * hw * 32;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const int id = wave * MT + t;
            const int oy = ty * TH + (id >> 1), ox = tx * 32 + (id & 1) * 16 + li;
            const bool ok = a.res && oy < a.Ho && ox < a.Wo;
#pragma unroll
            for (int n = 0; n < NT; ++n)
                rsn[t][n] = ok ? residual_quad(a.res, a.res_bits, rbase + (size_t)(oy * a.Wo + ox) * 32 + n * 16 + kq * 4) : (f4){0.f, 0.f, 0.f, 0.f};
        }
    };
    Coord cur;
    cur.b = t_begin / a.tiles;
    cur.ty = (t_begin - cur.b * a.tiles) / a.tiles_x;
    cur.tx = t_begin - cur.b * a.tiles - cur.ty * a.tiles_x;
    Coord nxt = cur;
    load_tile(cur);
    load_res(cur);
    wait_vmcnt_imm<0>();                       // the weight copies have landed (this wave's); the barrier below publishes all of them
    unsigned int ph[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long tk = 0;
    auto stamp = [&](int i) {
        if constexpr (STAMP) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            if (i >= 0) ph[i] += (unsigned int)(now - tk);
            tk = now;
        }
    };
    for (int L = t_begin; L < t_end; L += NWG, cur = nxt) {
        const int b = cur.b, tile_id = cur.ty * a.tiles_x + cur.tx;
        const int oy0 = cur.ty * TH, ox0 = cur.tx * 32;
        advance(nxt);
        stamp(-1);
        store_tile();
        stamp(0);
        __syncthreads();                        // tile (and, first time, weights) visible
        stamp(1);
        // halo AND residual of the NEXT tile: in flight during this tile's 9 taps and epilogue
        int pixo[MT];
        f4 rs[MT][NT];
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const int id = wave * MT + t;
            const int oy = oy0 + (id >> 1), ox = ox0 + (id & 1) * 16 + li;
            pixo[t] = (oy < a.Ho && ox < a.Wo) ? oy * a.Wo + ox : -1;
#pragma unroll
            for (int n = 0; n < NT; ++n) rs[t][n] = rsn[t][n];
        }
        const bool has_res = a.res != nullptr;
        if (L + NWG < t_end) load_res(nxt);
        if (L + NWG < t_end) load_tile(nxt);
        stamp(2);

        f4 acc[MT][NT];
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[t][n] = (f4){0.f, 0.f, 0.f, 0.f};
        Frags fr[2];
        read_frags(fr[0], 0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (tap < 8) read_frags(fr[(tap + 1) & 1], tap + 1);
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

        if constexpr (STAMP) {
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int n = 0; n < NT; ++n) asm volatile("" : "+v"(acc[t][n]));        // the stamp must not move above the last MFMA's result
        }
        stamp(3);
        float* __restrict__ yb = a.y + (size_t)b * hw * 32;
        f4 gsum[NT], gsq[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            gsum[n] = (f4){0.f, 0.f, 0.f, 0.f};
            gsq[n] = gsum[n];
            const int co = n * 16 + kq * 4;
            const f4 gt = a.gate ? *reinterpret_cast<const f4*>(a.gate + (size_t)b * 32 + co) : (f4){1.f, 1.f, 1.f, 1.f};
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                f4 v = acc[t][n] + bi[n];
                if (a.relu) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                }
                v = v * sc[n] + sh[n];
                if (a.gate) v = v * gt;
                if (has_res && pixo[t] >= 0) v += rs[t][n];
                if (a.relu2) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                }
                if (pixo[t] >= 0) {
                    *reinterpret_cast<f4*>(yb + pixo[t] * 32 + co) = v;
                    gsum[n] += v;
                    if (a.gap2) gsq[n] += v * v;
                }
            }
        }
        if (a.gap) {
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float sm = gsum[n][r];
                    sm += __shfl_xor(sm, 1, 64); sm += __shfl_xor(sm, 2, 64);
                    sm += __shfl_xor(sm, 4, 64); sm += __shfl_xor(sm, 8, 64);
                    if (li == 0) sred[wave * sred_pitch + n * 16 + kq * 4 + r] = sm;
                    if (a.gap2) {               // 4-row tiles only (launch check): the squares' scratch follows the sums'
                        float q = gsq[n][r];
                        q += __shfl_xor(q, 1, 64); q += __shfl_xor(q, 2, 64);
                        q += __shfl_xor(q, 4, 64); q += __shfl_xor(q, 8, 64);
                        if (li == 0) sred[128 + wave * 32 + n * 16 + kq * 4 + r] = q;
                    }
                }
        }
        stamp(4);
        __syncthreads();                        // every wave is done with the tile image (and the gap scratch is complete)
        stamp(5);
        if (a.gap && tid < 32) {
            float sm = 0.f;
#pragma unroll
            for (int m = 0; m < 4; ++m) sm += sred[m * sred_pitch + tid];
            a.gap[((size_t)b * a.tiles + tile_id) * 32 + tid] = sm;
            if (a.gap2) a.gap2[((size_t)b * a.tiles + tile_id) * 32 + tid] = (sred[128 + tid] + sred[160 + tid]) + (sred[192 + tid] + sred[224 + tid]);
        }
    }
    if constexpr (STAMP) {
        if (lane == 0 && a.dbg)
#pragma unroll
            for (int i = 0; i < 6; ++i) a.dbg[((size_t)blockIdx.x * 4 + wave) * 8 + i] = ph[i];
    }
}

// ---- stem: Conv2d(1->C, 3x3, bias) -> ReLU -> BN, x [B,H,W] -> y NHWC ----------------------------
__global__ __launch_bounds__(256) void stem_conv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, float* __restrict__ y,
                                                        int B, int H, int W, int C) {
    // One workgroup per output row: the three input rows (zero padded) are staged in LDS once, thread = (pixel, channel quad) with the
    // quad fixed per thread (its 9 taps + bias + BN affine stay in registers), pixels strided by 256 / (C/4).  No per-element index
    // division (the first version spent four 64-bit divisions per output quad: 27 M VALU instructions per launch, VALU bound at 53 us).
    extern __shared__ float rows[];                     // [3][W + 2]
    const int cq_n = C >> 2, tid = threadIdx.x;
    const int b = blockIdx.x / H, oy = blockIdx.x - b * H, WP = W + 2;
    const float* xb = x + (size_t)b * H * W;
    for (int i = tid; i < 3 * WP; i += 256) {
        const int r = i / WP, c = i - r * WP, gy = oy + r - 1, gx = c - 1;
        rows[i] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? xb[gy * W + gx] : 0.f;
    }
    const int cq = tid % cq_n, px0 = tid / cq_n, pstep = 256 / cq_n;
    f4 wv[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wv[t] = *reinterpret_cast<const f4*>(w + t * C + cq * 4);
    const f4 bi = *reinterpret_cast<const f4*>(bias + cq * 4);
    const f4 sc = *reinterpret_cast<const f4*>(scale + cq * 4), sh = *reinterpret_cast<const f4*>(shift + cq * 4);
    __syncthreads();
    float* yrow = y + ((size_t)b * H + oy) * W * C + cq * 4;
    for (int ox = px0; ox < W; ox += pstep) {
        f4 acc = bi;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) acc += wv[kh * 3 + kw] * rows[kh * WP + ox + kw];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = fmaxf(acc[r], 0.f);
        *reinterpret_cast<f4*>(yrow + (size_t)ox * C) = acc * sc + sh;
    }
}

// ---- SE gate: GAP finish + FC -> ReLU -> FC -> sigmoid, one workgroup per clip ---------------------
__global__ __launch_bounds__(256) void se_gate_kernel(const float* __restrict__ gap, int tiles, const float* __restrict__ w1,
                                                      const float* __restrict__ b1, const float* __restrict__ w2,
                                                      const float* __restrict__ b2, float* __restrict__ gate, int C, float inv_hw) {
    __shared__ float m[256];
    __shared__ float h[32];
    const int b = blockIdx.x, t = threadIdx.x, R = C >> 3;
    {   // fixed-order (deterministic) two-level sum of the per-tile partials: blockDim/C groups x C channels
        const int c = t % C, g = t / C, G = blockDim.x / C;
        float s = 0.f;
        for (int i = g; i < tiles; i += G) s += gap[((size_t)b * tiles + i) * C + c];
        m[t] = s;
        __syncthreads();
        float tot = 0.f;
        if (t < C)
            for (int j = 0; j < G; ++j) tot += m[j * C + t];
        __syncthreads();
        if (t < C) m[t] = tot * inv_hw;
    }
    __syncthreads();
    if (t < R) {
        float s = b1[t];
        for (int c = 0; c < C; ++c) s += w1[t * C + c] * m[c];
        h[t] = fmaxf(s, 0.f);
    }
    __syncthreads();
    if (t < C) {
        float s = b2[t];
        for (int j = 0; j < R; ++j) s += w2[t * R + j] * h[j];
        gate[(size_t)b * C + t] = 1.f / (1.f + expf(-s));
    }
}

// ---- SE gate computed BEFORE conv2 runs ("gate from the input's moments") ------------------------------------------------------
// gate = sigmoid(W2 relu(W1 mean_hw(y) + b1) + b2), y = BN2(conv2(t1)) (ResNetBlocks.py:28-30,92-96).  The spatial mean of a 3x3 / pad 1 /
// stride 1 convolution is linear in shifted-window sums of its INPUT:
//     mean_hw(conv2(t1))[co] = 1/HW * sum_{tap, ci} W[co][ci][tap] * S[tap][ci],   S[(kh,kw)][ci] = sum of t1[.., ci] over the rows / columns
//     the tap reaches: everything, minus the last (kh = 0) or first (kh = 2) row, minus the last (kw = 0) or first (kw = 2) column, plus
//     the doubly removed corner.
// The total comes from conv1's per-tile channel sums (its `gap` output), the four border lines and corners are read from t1 itself.
// With the gate known up front, conv2's epilogue applies relu(y * gate + x) directly: the block's y tensor is never written and the
// separate tail pass (2 reads + 1 write of the activation) disappears.  One workgroup per clip; fixed-order sums (deterministic).
constexpr int GP_T = 1024;      // threads per clip: G = 1024 / C thread groups share every sum (many loads in flight: the kernel is latency bound)
__global__ __launch_bounds__(GP_T) void se_gate_pre_kernel(const float* __restrict__ t1, const float* __restrict__ gap, int tiles,
                                                           const float* __restrict__ w2img, const float* __restrict__ scale2,
                                                           const float* __restrict__ shift2, const float* __restrict__ w1, const float* __restrict__ b1,
                                                           const float* __restrict__ wf2, const float* __restrict__ bf2, float* __restrict__ gate,
                                                           int H, int W, int C) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int b = blockIdx.x, t = threadIdx.x, G = GP_T / C, c = t % C, g = t / C, R = C >> 3;
    float* part = sm;                   // [5][GP_T]: total, first row, last row, first column, last column (per thread)
    float* S = sm + 5 * GP_T;           // [9][C]
    float* zp = S + 9 * C;              // [G][C]
    float* m = zp + GP_T;               // [C]
    float* hbuf = m + C;                // [R]
    const float* tb = t1 + (size_t)b * H * W * C;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
#pragma unroll 4
    for (int i = g; i < tiles; i += G) s0 += gap[((size_t)b * tiles + i) * C + c];
#pragma unroll 4
    for (int x = g; x < W; x += G) {
        s1 += tb[(size_t)x * C + c];
        s2 += tb[((size_t)(H - 1) * W + x) * C + c];
    }
#pragma unroll 4
    for (int y = g; y < H; y += G) {
        s3 += tb[((size_t)y * W) * C + c];
        s4 += tb[((size_t)y * W + W - 1) * C + c];
    }
    part[t] = s0; part[GP_T + t] = s1; part[2 * GP_T + t] = s2; part[3 * GP_T + t] = s3; part[4 * GP_T + t] = s4;
    __syncthreads();
    if (t < C) {
        float T = 0.f, R0 = 0.f, RL = 0.f, C0 = 0.f, CL = 0.f;
        for (int j = 0; j < G; ++j) {
            T += part[j * C + t]; R0 += part[GP_T + j * C + t]; RL += part[2 * GP_T + j * C + t]; C0 += part[3 * GP_T + j * C + t];
            CL += part[4 * GP_T + j * C + t];
        }
        const float c00 = tb[t], c0L = tb[(size_t)(W - 1) * C + t], cL0 = tb[((size_t)(H - 1) * W) * C + t], cLL = tb[((size_t)(H - 1) * W + W - 1) * C + t];
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
    {   // z[co] = sum over (tap, ci) of W[tap][ci/4][co][4] * S[tap][ci]; the 9*C/4 quads are dealt to the G thread groups
        const f4* w4 = reinterpret_cast<const f4*>(w2img);
        const int nq = 9 * (C >> 2);
        float z = 0.f;
#pragma unroll 4
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
    {   // hidden layer: one wave per unit, lanes stride the C inputs, fixed-order wave reduction (R threads each walking a row of W1 was a
        // 2.5 us serial chain at C = 128: stamps in profiles/r06_gate_pre_stamps.txt)
        const int wv = t >> 6, lane = t & 63;
        for (int j = wv; j < R; j += GP_T / 64) {
            float s = 0.f;
            for (int cc = lane; cc < C; cc += 64) s += w1[j * C + cc] * m[cc];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            if (lane == 0) hbuf[j] = fmaxf(s + b1[j], 0.f);
        }
    }
    __syncthreads();
    if (t < C) {
        float s = bf2[t];
        for (int j = 0; j < R; ++j) s += wf2[t * R + j] * hbuf[j];
        gate[(size_t)b * C + t] = 1.f / (1.f + expf(-s));
    }
}

// ---- SE tail: out = relu(y*gate + residual) ----------------------------------------------------------
__global__ __launch_bounds__(256) void se_tail_identity_kernel(const f4* __restrict__ y, const float* __restrict__ gate,
                                                               const f4* __restrict__ res, f4* __restrict__ out,
                                                               size_t n4, int hw_cq, int cq_n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const int cq = (int)(i % cq_n);
        const size_t b = i / hw_cq;
        const f4 g = *reinterpret_cast<const f4*>(gate + b * cq_n * 4 + cq * 4);
        f4 v = y[i] * g + res[i];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        out[i] = v;
    }
}

// residual = BN(conv1x1 stride s (x_in)); weights [CIN][COUT] staged in LDS; one thread = 1 pixel x 4 couts
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void se_tail_downsample_kernel(const float* __restrict__ y, const float* __restrict__ gate,
                                                                 const float* __restrict__ xin, const float* __restrict__ dsw,
                                                                 const float* __restrict__ dss, const float* __restrict__ dsh,
                                                                 float* __restrict__ out, int B, int Ho, int Wo, int Hin, int Win, int S) {
    extern __shared__ __attribute__((aligned(16))) f4 wl[];         // CIN * COUT / 4 (128 KB at 128 -> 256)
    for (int i = threadIdx.x; i < CIN * COUT / 4; i += 256) wl[i] = reinterpret_cast<const f4*>(dsw)[i];
    __syncthreads();
    constexpr int CQ = COUT / 4;
    // rows of the output map are dealt to the workgroups; inside a row the index splits by shifts (CQ is a power of two): no per-element
    // 64-bit division (three per output quad in the first version)
    for (int row = blockIdx.x; row < B * Ho; row += gridDim.x) {
      const int b = row / Ho, oy = row - b * Ho;
      for (int t = threadIdx.x; t < Wo * CQ; t += 256) {
        const int cq = t % CQ, ox = t / CQ;
        const size_t p = (size_t)row * Wo + ox;
        const float* xp = xin + (((size_t)b * Hin + (size_t)oy * S) * Win + (size_t)ox * S) * CIN;
        f4 acc = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int ci = 0; ci < CIN; ci += 4) {
            const f4 xv = *reinterpret_cast<const f4*>(xp + ci);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc += wl[(ci + j) * CQ + cq] * xv[j];
        }
        const f4 res = acc * *reinterpret_cast<const f4*>(dss + cq * 4) + *reinterpret_cast<const f4*>(dsh + cq * 4);
        const f4 g = *reinterpret_cast<const f4*>(gate + (size_t)b * COUT + cq * 4);
        f4 v = *reinterpret_cast<const f4*>(y + p * COUT + cq * 4) * g + res;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        *reinterpret_cast<f4*>(out + p * COUT + cq * 4) = v;
      }
    }
}

template <int CIN, int NT, int S, int TH, int WM, int WN, int TERMS>
int launch_conv_bf16(const ConvArgs& a, const bf8* whi, const bf8* wlo, dim3 grid, hipStream_t st) {
    // weight-ring depth: 3 where the K loop is long (C >= 64, stride 1); 2 for the HBM-bound C=32 layer and the
    // stride-2 entries (smaller LDS footprint => one more workgroup per CU)
    constexpr int RING = 3;
    using G = ConvGeom<S, TH>;
    constexpr int NIMG = (TERMS == 3) ? 2 : 1;
    constexpr size_t LDS_BYTES = sizeof(bf8) * (size_t)(NIMG * 4 * G::PL + RING * NIMG * 4 * NT * 16);
    auto kern = conv3x3_bf16_kernel<CIN, NT, S, TH, WM, WN, TERMS, RING>;
    if (int rc = eg_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), LDS_BYTES, "conv3x3")) return rc;
    hipLaunchKernelGGL(kern, grid, dim3(256), LDS_BYTES, st, a, whi, wlo);
    return eg_check_launch("conv3x3");
}

// Channel-split launch of a stride-1 body convolution: NTS * 16 output channels per workgroup, grid.z = cout / (NTS * 16) (conv3x3_bf16_kernel, SPLIT).
template <int CIN, int NTS, int TH, int WM, int WN, int TERMS>
int launch_conv_split_t(ConvArgs a, int batch, const bf8* whi, const bf8* wlo, hipStream_t st) {
    constexpr int RING = 3;
    using G = ConvGeom<1, TH>;
    constexpr int NIMG = (TERMS == 3) ? 2 : 1;
    constexpr size_t LDS_BYTES = sizeof(bf8) * (size_t)(NIMG * 4 * G::PL + RING * NIMG * 4 * NTS * 16);
    auto kern = conv3x3_bf16_kernel<CIN, NTS, 1, TH, WM, WN, TERMS, RING, true>;
    if (int rc = eg_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), LDS_BYTES, "conv3x3 (channel split)")) return rc;
    a.wrow = a.cout;
    hipLaunchKernelGGL(kern, dim3(a.tiles, batch, a.cout / (NTS * 16)), dim3(256), LDS_BYTES, st, a, whi, wlo);
    return eg_check_launch("conv3x3 (channel split)");
}
// Input gradient of a stride-2 convolution (conv3x3_bf16_kernel, DG2): KCH = the convolution's output channels (the contraction), NTN * 16 = its input
// channels; grid over TH x 32 tiles of the dy grid.
template <int KCH, int NTN, int TH, int WM, int WN>
int launch_dgrad_s2_t(const ConvArgs& a, int batch, const bf8* whi, const bf8* wlo, hipStream_t st) {
    constexpr int RING = 3;
    using G = ConvGeomDG2<TH>;
    constexpr size_t LDS_BYTES = sizeof(bf8) * (size_t)(2 * 4 * G::PL + RING * 2 * 4 * NTN * 16);
    auto kern = conv3x3_bf16_kernel<KCH, NTN, 1, TH, WM, WN, 3, RING, false, true>;
    if (int rc = eg_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), LDS_BYTES, "conv3x3 (stride-2 input gradient)")) return rc;
    dim3 grid(eg_cdiv(a.W, 32) * eg_cdiv(a.H, TH), batch);
    hipLaunchKernelGGL(kern, grid, dim3(256), LDS_BYTES, st, a, whi, wlo);
    return eg_check_launch("conv3x3 (stride-2 input gradient)");
}
// how many ways to split the channels of a C -> C body convolution with `wgs` pixel-tile workgroups: keep the launch near one workgroup per CU
int conv_channel_split(int wgs, int max_split) {
    const char* e = getenv("EG_CONV_SPLIT");            // A/B switch, read per call: 1 = never, 2, 4 (a captured graph keeps what it was captured with)
    const int forced = (e && e[0]) ? atoi(e) : -1;
    if (forced >= 0) return (forced == 2 || forced == 4) && forced <= max_split ? forced : 1;
    // measured (tools/conv_b1_probe.py, us per launch, unsplit / 2 / 4): 128 ch. 64 tiles 26.9 / 18.9 / 15.3, 128 tiles 28.6 / 20.7 / 20.8, 256 tiles
    // 31.1 / 30.7 / 37.0; 64 ch. 128 tiles 18.4 / 13.9, 256 tiles 20.1 / 19.6, 512 tiles 29.7 / 37.6 -- split while the launch stays within two workgroups per CU
    int split = 1;
    while (split < max_split && wgs * split * 2 <= 512) split *= 2;
    return split;
}

template <int CIN, int NT, int S, int TH, int WM, int WN>
int launch_conv(const ConvArgs& a, int batch, int precision, hipStream_t st) {
    dim3 grid(a.tiles, batch), block(256);
    if (precision == EG_PREC_F32) {
        hipLaunchKernelGGL((conv3x3_f32_kernel<CIN, NT, S, TH>), grid, block, 0, st, a);
        return eg_check_launch("conv3x3");
    }
    // packed bf16 weights follow the fp32 image in the arena: [hi image][lo image], each 9*CIN*COUTP bf16
    const size_t f32_floats = (size_t)9 * CIN * NT * 16;
    const bf8* whi = reinterpret_cast<const bf8*>(a.w + f32_floats);
    const bf8* wlo = whi + (size_t)9 * (CIN / 8) * NT * 16;
    if (precision == EG_PREC_BF16X3) return launch_conv_bf16<CIN, NT, S, TH, WM, WN, 3>(a, whi, wlo, grid, st);
    return launch_conv_bf16<CIN, NT, S, TH, WM, WN, 1>(a, whi, wlo, grid, st);
}

bool conv32_persistent_enabled() {
    static const bool on = [] { const char* e = getenv("EG_CONV32_PERSISTENT"); return !(e && e[0] == '0'); }();
    return on;
}

template <int TERMS, int TH>
int launch_conv32_persistent_t(const ConvArgs& a, int batch, const bf8* whi, const bf8* wlo, hipStream_t st) {
    using G = ConvGeom<1, TH>;
    constexpr int NIMG = (TERMS == 3) ? 2 : 1;
    constexpr size_t LDS_BYTES = sizeof(bf8) * (size_t)(NIMG * 4 * G::PL + NIMG * 9 * 128) + (TH == 4 ? 2 * 4 * 32 * sizeof(float) : 0);   // + sums / squares scratch
    auto kern = conv3x3_c32_persistent_kernel<TERMS, TH>;
    if (int rc = eg_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), LDS_BYTES, "conv3x3 (32 -> 32, persistent)")) return rc;
    const int total = a.tiles * batch;
    // 512 = 2 workgroups per CU, each walking total / 512 tiles (more, shorter-lived workgroups were measured slower alone and under four lanes:
    // profiles/r05_conv32_experiments.txt).
    const int cap = 512;
    int grid = total < cap ? total : cap;
    int tpw = eg_cdiv(total, grid);
    grid = eg_cdiv(total, tpw);
    if (grid >= 8) grid = (int)eg_round_up(grid, 8);             // XCD remap needs a multiple of 8 (surplus workgroups exit at once)
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS_BYTES, st, a, whi, wlo, total, tpw);
    return eg_check_launch("conv3x3 (32 -> 32, persistent)");
}
int launch_conv32_persistent(const ConvArgs& a, int batch, int precision, int th, hipStream_t st) {
    const size_t f32_floats = (size_t)9 * 32 * 32;
    const bf8* whi = reinterpret_cast<const bf8*>(a.w + f32_floats);
    const bf8* wlo = whi + (size_t)9 * 4 * 32;
    static const bool stamp = [] { const char* e = getenv("EG_CONV32_STAMP"); return e && e[0] == '1'; }();
    if (stamp && precision == EG_PREC_BF16X3 && th == 4) {
        // diagnostic: per-phase cycle sums of every wave, averaged and printed after the launch (synchronises: never use while timing)
        static unsigned int* dbg = nullptr;
        const int nwords = 512 * 4 * 8;
        if (!dbg && hipMalloc(&dbg, nwords * sizeof(unsigned int)) != hipSuccess) return EG_ERR_HIP;
        (void)hipMemsetAsync(dbg, 0, nwords * sizeof(unsigned int), st);
        ConvArgs b = a;
        b.dbg = dbg;
        constexpr size_t LDS_BYTES = sizeof(bf8) * (size_t)(2 * 4 * ConvGeom<1, 4>::PL + 2 * 9 * 128) + 2 * 4 * 32 * sizeof(float);   // sums + squares (a.gap2), as the production launch
        auto kern = conv3x3_c32_persistent_kernel<3, 4, true>;
        if (int rc = eg_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), LDS_BYTES, "conv3x3 (stamps)")) return rc;
        const int total = a.tiles * batch;
        int grid = total < 512 ? total : 512;
        grid = eg_cdiv(total, eg_cdiv(total, grid));
        if (grid >= 8) grid = (int)eg_round_up(grid, 8);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS_BYTES, st, b, whi, wlo, total, 0);
        static unsigned int host[512 * 4 * 8];
        if (hipMemcpyAsync(host, dbg, sizeof(host), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return EG_ERR_HIP;
        double sum[6] = {0, 0, 0, 0, 0, 0};
        int nw = 0;
        for (int w = 0; w < grid * 4; ++w) {
            if (host[w * 8 + 3] == 0) continue;
            ++nw;
            for (int i = 0; i < 6; ++i) sum[i] += host[w * 8 + i];
        }
        const double tiles_per_wave = (double)total / grid;
        fprintf(stderr, "conv32 stamps (cycles per tile and wave, %d waves, %s residual): store_tile %.0f  barrier1 %.0f  issue_loads %.0f  taps %.0f  epilogue %.0f  barrier2 %.0f\n",
                nw, a.res ? "with" : "no", sum[0] / nw / tiles_per_wave, sum[1] / nw / tiles_per_wave, sum[2] / nw / tiles_per_wave, sum[3] / nw / tiles_per_wave,
                sum[4] / nw / tiles_per_wave, sum[5] / nw / tiles_per_wave);
        return eg_check_launch("conv3x3 (32 -> 32, stamps)");
    }
    if (th == 8) {
        if (precision == EG_PREC_BF16X3) return launch_conv32_persistent_t<3, 8>(a, batch, whi, wlo, st);
        return launch_conv32_persistent_t<1, 8>(a, batch, whi, wlo, st);
    }
    if (precision == EG_PREC_BF16X3) return launch_conv32_persistent_t<3, 4>(a, batch, whi, wlo, st);
    return launch_conv32_persistent_t<1, 4>(a, batch, whi, wlo, st);
}

int conv_tile_rows(int cin, int cout, int stride) {
    if (stride == 2 || cout >= 256) return 2;
    if (cin == 32 && cout == 32) {               // 4-row tiles: smaller LDS footprint, more workgroups in flight (HBM-bound layer)
        static const int th = [] { const char* e = getenv("EG_CONV32_TH"); return (e && atoi(e) == 8) ? 8 : 4; }();      // experiment hook
        return th;
    }
    return (cout >= 128 || cin >= 128) ? 4 : 8;
}

}  // namespace

extern "C" int32_t eg_conv3x3_gap_tiles(int32_t h, int32_t wdt, int32_t cin, int32_t cout, int32_t stride) {
    const int ho = (h + 2 - 3) / stride + 1, wo = (wdt + 2 - 3) / stride + 1;
    const int th = conv_tile_rows(cin, cout, stride);
    return eg_cdiv(ho, th) * eg_cdiv(wo, 32);
}

// Packed weight size in floats for one 3x3 conv: fp32 image + (hi, lo) bf16 images.
extern "C" int64_t eg_conv3x3_packed_floats(int32_t cin, int32_t cout_pad) {
    return (int64_t)9 * cin * cout_pad * 2;     // 9*cin*coutp fp32 + 2 * 9*cin*coutp bf16 (= same bytes again)
}

extern "C" int eg_conv3x3(const float* x, const float* w, const float* bias, const float* scale, const float* shift,
                          float* y, float* gap_partial, int32_t batch, int32_t h, int32_t wdt, int32_t cin, int32_t cout,
                          int32_t stride, int32_t relu, int32_t nchw_out, int32_t precision, void* stream) {
    return eg_conv3x3_se(x, w, bias, scale, shift, nullptr, nullptr, y, gap_partial, batch, h, wdt, cin, cout, stride, relu, nchw_out, precision, stream);
}

namespace {
int conv3x3_dispatch(const float* x, const float* w, const float* bias, const float* scale, const float* shift, const float* gate, const float* residual,
                     float* y, float* gap_partial, float* gap_sq, int32_t batch, int32_t h, int32_t wdt, int32_t cin, int32_t cout, int32_t stride,
                     int32_t relu, int32_t nchw_out, int32_t precision, void* stream, const float* in_scale = nullptr, const float* in_shift = nullptr,
                     const uint32_t* res_bits = nullptr);
}
extern "C" int eg_conv3x3_se(const float* x, const float* w, const float* bias, const float* scale, const float* shift, const float* gate,
                             const float* residual, float* y, float* gap_partial, int32_t batch, int32_t h, int32_t wdt, int32_t cin,
                             int32_t cout, int32_t stride, int32_t relu, int32_t nchw_out, int32_t precision, void* stream) {
    return conv3x3_dispatch(x, w, bias, scale, shift, gate, residual, y, gap_partial, nullptr, batch, h, wdt, cin, cout, stride, relu, nchw_out, precision, stream);
}
// y = conv(x) + (bit ? residual : 0): the input gradient of an SE block's first convolution with the identity shortcut's gradient
// dout * (out > 0) (ResNetBlocks.py:33-36) added in the epilogue straight from dout and the tail's ReLU bits (csrc/train.hip: se_tail_fwd_kernel writes one bit
// per element, bit e & 31 of word e >> 5) -- the masked map is never stored.  Stride 1, NHWC, cout % 4 == 0; x is the upstream gradient, w the flipped image.
extern "C" int eg_conv3x3_res_masked(const float* x, const float* w, const float* residual, const uint32_t* res_bits, float* y, int32_t batch, int32_t h,
                                     int32_t wdt, int32_t cin, int32_t cout, int32_t precision, void* stream) {
    EG_REQUIRE(residual && res_bits, EG_ERR_BAD_ARG, "eg_conv3x3_res_masked: the residual and its bit mask are required");
    return conv3x3_dispatch(x, w, nullptr, nullptr, nullptr, nullptr, residual, y, nullptr, nullptr, batch, h, wdt, cin, cout, 1, 0, 0, precision, stream, nullptr,
                            nullptr, res_bits);
}
// Training forward: y = [relu](conv(x) + bias), plus the per-(clip, tile) channel sums of y AND of y*y (both [batch][tiles][cout]): train-mode
// BatchNorm takes its mean and variance from them (csrc/train.hip: eg_bn_train_forward_sq) without reading y again.  Split-bf16 modes only.
extern "C" int eg_conv3x3_sq(const float* x, const float* w, const float* bias, float* y, float* gap_partial, float* gap_sq_partial, int32_t batch, int32_t h,
                             int32_t wdt, int32_t cin, int32_t cout, int32_t stride, int32_t relu, int32_t precision, void* stream) {
    EG_REQUIRE(gap_partial && gap_sq_partial, EG_ERR_BAD_ARG, "eg_conv3x3_sq: both partial buffers are required");
    EG_REQUIRE(precision != EG_PREC_F32, EG_ERR_UNSUPPORTED, "eg_conv3x3_sq: split-bf16 modes only (the fp32 kernel emits sums only)");
    EG_REQUIRE(!(cin == 32 && cout == 32 && stride == 1) || conv_tile_rows(cin, cout, stride) == 4, EG_ERR_UNSUPPORTED,
               "eg_conv3x3_sq: the 8-row experiment tiles of the 32 -> 32 kernel have no room for the squares scratch");
    return conv3x3_dispatch(x, w, bias, nullptr, nullptr, nullptr, nullptr, y, gap_partial, gap_sq_partial, batch, h, wdt, cin, cout, stride, relu, 0, precision,
                            stream);
}
// eg_conv3x3_sq on x' = x * in_scale[ci] + in_shift[ci] (per INPUT channel, in-image pixels; the zero padding stays zero): the train-mode BatchNorm in
// front of the convolution folded into its operand staging, so that the normalised map is never written (ResNetBlocks.py:26-27: bn1 -> conv2).
extern "C" int eg_conv3x3_sq_in_affine(const float* x, const float* in_scale, const float* in_shift, const float* w, const float* bias, float* y,
                                       float* gap_partial, float* gap_sq_partial, int32_t batch, int32_t h, int32_t wdt, int32_t cin, int32_t cout,
                                       int32_t stride, int32_t relu, int32_t precision, void* stream) {
    EG_REQUIRE(in_scale && in_shift, EG_ERR_BAD_ARG, "eg_conv3x3_sq_in_affine: both affine vectors are required");
    EG_REQUIRE(!gap_partial == !gap_sq_partial, EG_ERR_BAD_ARG, "eg_conv3x3_sq_in_affine: both partial buffers or none");
    return conv3x3_dispatch(x, w, bias, nullptr, nullptr, nullptr, nullptr, y, gap_partial, gap_sq_partial, batch, h, wdt, cin, cout, stride, relu, 0, precision,
                            stream, in_scale, in_shift);
}
namespace {
int conv3x3_dispatch(const float* x, const float* w, const float* bias, const float* scale, const float* shift, const float* gate, const float* residual,
                     float* y, float* gap_partial, float* gap_sq, int32_t batch, int32_t h, int32_t wdt, int32_t cin, int32_t cout, int32_t stride,
                     int32_t relu, int32_t nchw_out, int32_t precision, void* stream, const float* in_scale, const float* in_shift, const uint32_t* res_bits) {
    EG_REQUIRE(x && w && y && batch > 0 && h > 0 && wdt > 0, EG_ERR_BAD_ARG, "eg_conv3x3: null pointer or empty shape");
    EG_REQUIRE(!res_bits || (residual && !gate && ((size_t)batch * h * wdt * cout) % 32 == 0), EG_ERR_BAD_ARG,
               "eg_conv3x3_res_masked: the bit mask needs a residual, no gate and a map of a multiple of 32 elements");
    EG_REQUIRE((in_scale == nullptr) == (in_shift == nullptr) && (!in_scale || (precision != EG_PREC_F32 && cin % 32 == 0 && eg_aligned16(in_scale) &&
               eg_aligned16(in_shift))), EG_ERR_BAD_ARG, "eg_conv3x3: the input affine needs both vectors, a split-bf16 mode and cin %% 32 == 0");
    // gate + residual: relu(v * gate + residual) (the fused SE tail); residual alone: v + residual, no ReLU (the training path's fused fan-in add:
    // an input gradient that lands on a tensor with a second consumer, train/functional.py conv3x3(passthrough=True))
    EG_REQUIRE(!gate || residual, EG_ERR_BAD_ARG, "eg_conv3x3_se: a gate needs a residual");
    EG_REQUIRE(!residual || (!nchw_out && stride == 1 && cout % 4 == 0 && (!gate || eg_aligned16(gate)) && eg_aligned16(residual) && residual != y),
               EG_ERR_BAD_ARG, "eg_conv3x3_se: the fused residual needs NHWC output, stride 1, cout %% 4 == 0 and a residual buffer distinct from y");
    EG_REQUIRE(eg_aligned16(x) && eg_aligned16(w) && eg_aligned16(y), EG_ERR_ALIGN, "eg_conv3x3: pointers must be 16-byte aligned");
    EG_REQUIRE(stride == 1 || stride == 2, EG_ERR_UNSUPPORTED, "eg_conv3x3: stride %d", stride);
    EG_REQUIRE(precision >= 0 && precision <= 2, EG_ERR_BAD_ARG, "eg_conv3x3: precision %d", precision);
    EG_REQUIRE(nchw_out || (cout % 4 == 0), EG_ERR_UNSUPPORTED, "eg_conv3x3: NHWC output needs cout %% 4 == 0");
    ConvArgs a;
    a.x = x; a.w = w; a.bias = bias; a.scale = scale; a.shift = shift; a.y = y; a.gap = gap_partial; a.gap2 = gap_sq;
    a.gate = gate; a.res = residual; a.relu2 = gate ? 1 : 0;
    a.res_bits = res_bits;
    a.in_scale = in_scale; a.in_shift = in_shift;
    a.H = h; a.W = wdt; a.Ho = (h + 2 - 3) / stride + 1; a.Wo = (wdt + 2 - 3) / stride + 1;
    a.cout = cout; a.relu = relu; a.nchw = nchw_out;
    const int th = conv_tile_rows(cin, cout, stride);
    a.tiles_x = eg_cdiv(a.Wo, 32);
    a.tiles = a.tiles_x * eg_cdiv(a.Ho, th);
    hipStream_t st = (hipStream_t)stream;
    const int coutp = (int)eg_round_up(cout, 16);
    EgProfScope prof((int64_t)cin * 1000000 + (int64_t)cout * 1000 + stride * 100 + 1,
                     2.0 * 9 * cin * cout * (double)a.Ho * a.Wo * batch, st);
    if (cin == 32 && coutp == 32 && cout == 32 && stride == 1 && precision != EG_PREC_F32 && !nchw_out && conv32_persistent_enabled())
        return launch_conv32_persistent(a, batch, precision, th, st);
    if (cin == 32 && coutp == 32 && stride == 1 && th == 4) return launch_conv<32, 2, 1, 4, 4, 1>(a, batch, precision, st);
    if (cin == 32 && coutp == 32 && stride == 1) return launch_conv<32, 2, 1, 8, 4, 1>(a, batch, precision, st);
    if (cin == 32 && coutp == 64 && stride == 2) return launch_conv<32, 4, 2, 2, 2, 2>(a, batch, precision, st);
    if ((cin == 64 || cin == 128) && cout == cin && stride == 1 && precision == EG_PREC_BF16X3 && !nchw_out) {
        // few pixel tiles (small batches): spread the output channels over workgroups as well -- bitwise the unsplit kernels below
        const int split = conv_channel_split(a.tiles * batch, cin == 128 ? 4 : 2);
        const size_t f32_floats = (size_t)9 * cin * cout;
        const bf8* whi = reinterpret_cast<const bf8*>(a.w + f32_floats);
        const bf8* wlo = whi + (size_t)9 * (cin / 8) * cout;
        if (cin == 64 && split == 2) return launch_conv_split_t<64, 2, 8, 4, 1, 3>(a, batch, whi, wlo, st);
        if (cin == 128 && split == 2) return launch_conv_split_t<128, 4, 4, 2, 2, 3>(a, batch, whi, wlo, st);
        if (cin == 128 && split == 4) return launch_conv_split_t<128, 2, 4, 2, 2, 3>(a, batch, whi, wlo, st);
    }
    if (cin == 64 && coutp == 64 && stride == 1) return launch_conv<64, 4, 1, 8, 4, 1>(a, batch, precision, st);
    if (cin == 64 && coutp == 128 && stride == 2) return launch_conv<64, 8, 2, 2, 2, 2>(a, batch, precision, st);
    if (cin == 64 && coutp == 128 && stride == 1) return launch_conv<64, 8, 1, 4, 2, 2>(a, batch, precision, st);      // training: input gradient of final_conv1 (dy padded to 64 channels)
    if (cin == 128 && coutp == 128 && stride == 1) return launch_conv<128, 8, 1, 4, 2, 2>(a, batch, precision, st);
    // 256-channel stage of the audio emotion classifier (model/audio_emotion_classifer.py:20-22): 2-row tiles, waves split the channels
    if (cin == 128 && coutp == 256 && stride == 2) return launch_conv<128, 16, 2, 2, 1, 4>(a, batch, precision, st);
    if (cin == 256 && coutp == 256 && stride == 1) return launch_conv<256, 16, 1, 2, 1, 4>(a, batch, precision, st);
    if (cin == 128 && coutp <= 64 && stride == 1) {       // final_conv1: 128 -> frames (34 -> 48, 60 -> 64)
        if (coutp <= 48) return launch_conv<128, 3, 1, 4, 4, 1>(a, batch, precision, st);
        return launch_conv<128, 4, 1, 4, 4, 1>(a, batch, precision, st);
    }
    // final_conv1 with 65..128 frames (BEAT-long, 120): the 128-wide body kernel on weights zero-padded to 128 channels
    if (cin == 128 && coutp <= 128 && stride == 1) return launch_conv<128, 8, 1, 4, 2, 2>(a, batch, precision, st);
    eg_set_error("eg_conv3x3: unsupported channels cin=%d cout=%d stride=%d", cin, cout, stride);
    return EG_ERR_UNSUPPORTED;
}
}  // namespace

// dx = F.conv2d's input gradient of nn.Conv2d(cin -> cout, k = 3, pad = 1, stride = 2) (the `_make_layer` entry convolutions, ResNetSE34V2.py:40-55):
// dy [batch][ho][wo][cout], w_flip = the packed images of the rotated, transposed filter (cout -> cin: the image the stride-1 input gradients use),
// dx [batch][h][w][cin] with ho = (h - 1) / 2 + 1, wo = (w - 1) / 2 + 1.  res_q (optional): a gradient [batch][ho][wo][cin] that belongs to the
// pixels (2i, 2j) of dx -- the stride-2 1x1 shortcut's input gradient (:43-47) -- added in the epilogue instead of scattered into a zero map.
// Split-bf16 (EG_PREC_BF16X3) only; (cin, cout) in {(32, 64), (64, 128), (128, 256)}.  Every element of dx is written.
extern "C" int eg_conv3x3_dgrad_s2(const float* dy, const float* w_flip, const float* res_q, float* dx, int32_t batch, int32_t h, int32_t wdt, int32_t cin,
                                   int32_t cout, int32_t precision, void* stream) {
    EG_REQUIRE(dy && w_flip && dx && batch > 0 && h > 0 && wdt > 0, EG_ERR_BAD_ARG, "eg_conv3x3_dgrad_s2: null pointer or empty shape");
    EG_REQUIRE(precision == EG_PREC_BF16X3, EG_ERR_UNSUPPORTED, "eg_conv3x3_dgrad_s2: split-bf16 arithmetic only (precision %d)", precision);
    EG_REQUIRE(eg_aligned16(dy) && eg_aligned16(w_flip) && eg_aligned16(dx) && (!res_q || eg_aligned16(res_q)), EG_ERR_ALIGN,
               "eg_conv3x3_dgrad_s2: pointers must be 16-byte aligned");
    ConvArgs a;
    a.x = dy; a.w = w_flip; a.bias = nullptr; a.scale = nullptr; a.shift = nullptr; a.y = dx; a.gap = nullptr;
    a.H = (h - 1) / 2 + 1; a.W = (wdt - 1) / 2 + 1; a.Ho = h; a.Wo = wdt;          // the kernel's "input" is dy, its output dx
    a.cout = cin; a.relu = 0; a.nchw = 0; a.tiles_x = eg_cdiv(a.W, 32); a.tiles = 0;
    a.res_q = res_q;
    hipStream_t st = (hipStream_t)stream;
    const size_t f32_floats = (size_t)9 * cout * cin;                              // the flipped image: cout "input" channels -> cin "output" channels
    const bf8* whi = reinterpret_cast<const bf8*>(w_flip + f32_floats);
    const bf8* wlo = whi + (size_t)9 * (cout / 8) * cin;
    EgProfScope prof((int64_t)cin * 1000000 + (int64_t)cout * 1000 + 200 + 8, 2.0 * 9 * cin * cout * (double)a.H * a.W * batch, st);
    if (cin == 32 && cout == 64) return launch_dgrad_s2_t<64, 2, 4, 4, 1>(a, batch, whi, wlo, st);
    if (cin == 64 && cout == 128) return launch_dgrad_s2_t<128, 4, 2, 2, 2>(a, batch, whi, wlo, st);
    if (cin == 128 && cout == 256) return launch_dgrad_s2_t<256, 8, 1, 1, 4>(a, batch, whi, wlo, st);
    eg_set_error("eg_conv3x3_dgrad_s2: unsupported channels %d -> %d", cin, cout);
    return EG_ERR_UNSUPPORTED;
}

extern "C" int eg_stem_conv(const float* x, const float* w9xc, const float* bias, const float* scale, const float* shift,
                            float* y, int32_t batch, int32_t h, int32_t wdt, int32_t c, void* stream) {
    EG_REQUIRE(x && w9xc && bias && scale && shift && y && batch > 0, EG_ERR_BAD_ARG, "eg_stem_conv: null pointer");
    EG_REQUIRE(c % 4 == 0 && c <= 128 && 256 % (c / 4) == 0, EG_ERR_UNSUPPORTED, "eg_stem_conv: C=%d", c);
    // 256 % (C/4) == 0: a thread keeps its channel quad; one workgroup per output row
    hipLaunchKernelGGL(stem_conv_kernel, dim3(batch * h), dim3(256), 3 * (wdt + 2) * sizeof(float), (hipStream_t)stream, x, w9xc, bias, scale, shift, y,
                       batch, h, wdt, c);
    return eg_check_launch("stem_conv");
}

extern "C" int eg_se_gate(const float* gap_partial, int32_t tiles, const float* w1, const float* b1, const float* w2,
                          const float* b2, float* gate, int32_t batch, int32_t c, int32_t hw, void* stream) {
    EG_REQUIRE(gap_partial && w1 && b1 && w2 && b2 && gate && batch > 0, EG_ERR_BAD_ARG, "eg_se_gate: null pointer");
    EG_REQUIRE(c % 8 == 0 && c <= 256 && (c <= 128 ? 128 % c : 256 % c) == 0, EG_ERR_UNSUPPORTED, "eg_se_gate: C=%d", c);
    hipLaunchKernelGGL(se_gate_kernel, dim3(batch), dim3(c <= 128 ? 128 : 256), 0, (hipStream_t)stream, gap_partial, tiles, w1, b1, w2, b2,
                       gate, c, 1.0f / (float)hw);
    return eg_check_launch("se_gate");
}

extern "C" int eg_se_gate_pre(const float* t1, const float* gap_partial, int32_t tiles, const float* conv2_w, const float* scale2,
                              const float* shift2, const float* w1, const float* b1, const float* w2, const float* b2, float* gate,
                              int32_t batch, int32_t h, int32_t wdt, int32_t c, void* stream) {
    EG_REQUIRE(t1 && gap_partial && conv2_w && scale2 && shift2 && w1 && b1 && w2 && b2 && gate && batch > 0 && h > 1 && wdt > 1, EG_ERR_BAD_ARG,
               "eg_se_gate_pre: null pointer or empty shape");
    EG_REQUIRE(c % 8 == 0 && c <= 256 && 256 % c == 0 && eg_aligned16(conv2_w), EG_ERR_UNSUPPORTED, "eg_se_gate_pre: C=%d", c);
    const size_t smem = sizeof(float) * (5 * GP_T + 9 * (size_t)c + GP_T + c + (c >> 3) + 4);
    hipLaunchKernelGGL(se_gate_pre_kernel, dim3(batch), dim3(GP_T), smem, (hipStream_t)stream, t1, gap_partial, tiles, conv2_w, scale2, shift2, w1,
                       b1, w2, b2, gate, h, wdt, c);
    return eg_check_launch("se_gate_pre");
}

extern "C" int eg_se_residual_relu(const float* y, const float* gate, const float* x_in, const float* ds_w,
                                   const float* ds_scale, const float* ds_shift, float* out, int32_t batch, int32_t ho,
                                   int32_t wo, int32_t c, int32_t h_in, int32_t w_in, int32_t cin, int32_t stride, void* stream) {
    EG_REQUIRE(y && gate && x_in && out && batch > 0, EG_ERR_BAD_ARG, "eg_se_residual_relu: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (!ds_w) {
        EG_REQUIRE(cin == c && h_in == ho && w_in == wo, EG_ERR_BAD_ARG, "eg_se_residual_relu: identity shortcut shape mismatch");
        const size_t n4 = (size_t)batch * ho * wo * c / 4;
        const int blocks = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
        hipLaunchKernelGGL(se_tail_identity_kernel, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const f4*>(y), gate,
                           reinterpret_cast<const f4*>(x_in), reinterpret_cast<f4*>(out), n4, ho * wo * (c / 4), c / 4);
        return eg_check_launch("se_tail");
    }
    EG_REQUIRE(ds_scale && ds_shift, EG_ERR_BAD_ARG, "eg_se_residual_relu: downsample BN missing");
    const size_t total = (size_t)batch * ho * wo * (c / 4);
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (cin == 32 && c == 64)
        hipLaunchKernelGGL((se_tail_downsample_kernel<32, 64>), dim3(blocks), dim3(256), 32 * 64 * 4, st, y, gate, x_in, ds_w, ds_scale,
                           ds_shift, out, batch, ho, wo, h_in, w_in, stride);
    else if (cin == 64 && c == 128)
        hipLaunchKernelGGL((se_tail_downsample_kernel<64, 128>), dim3(blocks), dim3(256), 64 * 128 * 4, st, y, gate, x_in, ds_w, ds_scale,
                           ds_shift, out, batch, ho, wo, h_in, w_in, stride);
    else if (cin == 128 && c == 256) {
        if (int rc = eg_ensure_dynamic_lds(reinterpret_cast<const void*>(se_tail_downsample_kernel<128, 256>), 128 * 256 * 4, "eg_se_residual_relu")) return rc;
        hipLaunchKernelGGL((se_tail_downsample_kernel<128, 256>), dim3(blocks < 1024 ? blocks : 1024), dim3(256), 128 * 256 * 4, st, y, gate,
                           x_in, ds_w, ds_scale, ds_shift, out, batch, ho, wo, h_in, w_in, stride);
    } else {
        eg_set_error("eg_se_residual_relu: unsupported downsample %d->%d", cin, c);
        return EG_ERR_UNSUPPORTED;
    }
    return eg_check_launch("se_tail_downsample");
}
