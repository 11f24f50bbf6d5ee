// EXPERIMENT, NOT BUILT (moved out of csrc/conv.hip in round 6; measured no-go in round 5: profiles/r05_conv32_experiments.txt).
// A fragment of csrc/conv.hip: it uses that file's ConvArgs / ConvGeom / bf8 / staging helpers and compiles only when pasted back after
// conv3x3_c32_persistent_kernel (kernel), before launch_conv32_persistent (launcher), and at the top of launch_conv32_persistent (switch).

// ======== kernel ========
// ---- the same kernel with TEAMS x 4 waves per workgroup sharing ONE copy of the weights (round-4 verdict item 4) -----------------------------------
// The persistent kernel above is latency bound at two waves per SIMD, and LDS caps its occupancy: 64 KB per workgroup, 36.8 KB of it the private copy of
// the 9 taps' weights.  Here a workgroup is TEAMS teams of 4 waves; a team is exactly one workgroup of the kernel above (own tile image, own walk
// through the tile list, own gap scratch), the weight images exist once: 3 teams = 12 waves = 3 per SIMD in 36.8 + 3 x 27.6 = 120 KB.  The barriers stay
// workgroup-wide (the teams run their phases in lockstep; what the third wave per SIMD adds is more loads in flight per CU inside each phase).  The
// residual is loaded for the CURRENT tile behind the first barrier (one register set instead of two: 3 waves per SIMD leave 168 VGPRs).
// Same arithmetic and summation order per output element: bitwise the kernel above.  EG_CONV32_TEAMS = 2 | 3 selects it (A/B; default: off).
template <int TERMS, int TEAMS>
__global__ __launch_bounds__(256 * TEAMS, 1) void conv3x3_c32_teams_kernel(ConvArgs a, const bf8* __restrict__ whi, const bf8* __restrict__ wlo,
                                                                     int total_tiles, int tiles_per_wg) {
    constexpr int TH = 4;
    constexpr bool STAMP = false;
    using G = ConvGeom<1, TH>;
    constexpr int CIN = 32, IW = G::IW, NPIX = G::NPIX, PL = G::PL, MT = TH / 2, NT = 2;
    constexpr int NIMG = (TERMS == 3) ? 2 : 1;
    constexpr int TILE = NIMG * 4 * PL, WTAP = 4 * 32, WIMG = 9 * WTAP;
    static_assert(PL - NPIX >= 8 || TH == 4, "the gap scratch of the 8-row variant lives in the planes' padding slots");
    extern __shared__ __attribute__((aligned(16))) bf8 lds[];      // TEAMS tiles | weights (hi [tap][octet][co], lo), ONE copy | TEAMS gap scratches
    const int team = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
    bf8* tile = lds + team * TILE;
    bf8* wl = lds + TEAMS * TILE;
    // gap scratch [4 waves][32 floats] (+ the same again for the sums of squares): behind the weights (4-row tiles), or -- 8-row tiles: tile + weights fill exactly half a CU's LDS -- in the
    // padding slots NPIX .. PL-1 of the first four channel-octet planes, which no staging write and no fragment read touches
    float* sred = reinterpret_cast<float*>(lds + TEAMS * TILE + NIMG * WIMG) + team * 256;
    constexpr int sred_pitch = 32;                                // floats between two waves' scratch rows

    const int tid = threadIdx.x & 255, wave = tid >> 6, lane = tid & 63, li = lane & 15, kq = lane >> 4;       // roles inside the team
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // XCD-aware: workgroups are dealt round-robin to the 8 XCDs; XCD k walks a contiguous eighth of the tile list
    int wg = blockIdx.x;
    if ((gridDim.x & 7) == 0) wg = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    // Walk order: workgroup w takes tiles w, w + G, w + 2G, ... (G = grid size), so that at any moment the G resident workgroups work on G
    // CONSECUTIVE tiles (each XCD on a contiguous run of G / 8): the halo rows shared by vertically adjacent tiles are fetched by neighbours at
    // about the same time and hit in that XCD's L2.  (A contiguous run per workgroup re-read them from memory 4 tiles later: 260 MB read per
    // launch at the memory-side counters against 196 MB for the tiled kernel.)
    const int NWG = gridDim.x * TEAMS;                             // tiles in flight chip-wide: team t of workgroup w walks w*TEAMS + t, + NWG, ...
    const int t_begin = wg * TEAMS + team, t_end = total_tiles;
    (void)tiles_per_wg;
    if (wg * TEAMS >= t_end) return;                               // the whole workgroup is surplus (uniform)
    // every team runs the same number of iterations (the barriers are workgroup-wide); a team past the end idles through them
    const int iters = (t_end - wg * TEAMS + NWG - 1) / NWG;

    // all weights -> LDS, once: NIMG x 18 pieces of 64 slots, dealt round-robin to the 4 waves
#pragma unroll
    for (int img = 0; img < NIMG; ++img) {
        const bf8* src = img ? wlo : whi;
#pragma unroll
        for (int p = 0; p < (WIMG / 64 + 4 * TEAMS - 1) / (4 * TEAMS); ++p) {
            const int piece = p * 4 * TEAMS + wave_all;
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
        const float* __restrict__ rb = a.res ? a.res + (size_t)c.b * hw * 32 : nullptr;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const int id = wave * MT + t;
            const int oy = ty * TH + (id >> 1), ox = tx * 32 + (id & 1) * 16 + li;
            const bool ok = rb && oy < a.Ho && ox < a.Wo;
#pragma unroll
            for (int n = 0; n < NT; ++n)
                rsn[t][n] = ok ? *reinterpret_cast<const f4*>(rb + (oy * a.Wo + ox) * 32 + n * 16 + kq * 4) : (f4){0.f, 0.f, 0.f, 0.f};
        }
    };
    Coord cur;
    cur.b = t_begin / a.tiles;
    cur.ty = (t_begin - cur.b * a.tiles) / a.tiles_x;
    cur.tx = t_begin - cur.b * a.tiles - cur.ty * a.tiles_x;
    Coord nxt = cur;
    if (t_begin < t_end) load_tile(cur);
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
    int L = t_begin;
    for (int it = 0; it < iters; ++it, L += NWG, cur = nxt) {
        const bool active = L < t_end;
        const int b = cur.b, tile_id = cur.ty * a.tiles_x + cur.tx;
        const int oy0 = cur.ty * TH, ox0 = cur.tx * 32;
        advance(nxt);
        if (active) store_tile();
        __syncthreads();                        // tile (and, first time, weights) visible
        if (!active) { __syncthreads(); continue; }
        // halo AND residual of the NEXT tile: in flight during this tile's 9 taps and epilogue
        int pixo[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const int id = wave * MT + t;
            const int oy = oy0 + (id >> 1), ox = ox0 + (id & 1) * 16 + li;
            pixo[t] = (oy < a.Ho && ox < a.Wo) ? oy * a.Wo + ox : -1;
        }
        const bool has_res = a.res != nullptr;
        load_res(cur);                          // this tile's residual: in flight during the 9 taps (one register set: 3 waves per SIMD need <= 168 VGPRs)
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
                if (has_res && pixo[t] >= 0) v += rsn[t][n];
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
}

// ======== launcher ========
template <int TERMS, int TEAMS>
int launch_conv32_teams_t(const ConvArgs& a, int batch, const bf8* whi, const bf8* wlo, hipStream_t st) {
    using G = ConvGeom<1, 4>;
    constexpr int NIMG = (TERMS == 3) ? 2 : 1;
    constexpr size_t LDS_BYTES = sizeof(bf8) * (size_t)(TEAMS * NIMG * 4 * G::PL + NIMG * 9 * 128) + (size_t)TEAMS * 256 * sizeof(float);
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    auto kern = conv3x3_c32_teams_kernel<TERMS, TEAMS>;
    if (int rc = eg_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), LDS_BYTES, "conv3x3 (32 -> 32, teams)")) return rc;
    const int total = a.tiles * batch;
    int grid = eg_cdiv(total, TEAMS);
    if (grid > 256) grid = 256;                                  // one workgroup (TEAMS x 4 waves) per CU
    if (grid >= 8) grid = (int)eg_round_up(grid, 8);             // XCD remap needs a multiple of 8 (surplus workgroups exit at once)
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256 * TEAMS), LDS_BYTES, st, a, whi, wlo, total, 0);
    return eg_check_launch("conv3x3 (32 -> 32, teams)");
}
// ======== switch (top of launch_conv32_persistent) ========
    {   // A/B switch, read per call (a captured graph keeps what it was captured with): teams of 4 waves sharing one weight copy
        const char* e = getenv("EG_CONV32_TEAMS");
        const int teams = (e && e[0]) ? atoi(e) : 0;
        if (th == 4 && precision == EG_PREC_BF16X3 && teams == 2 && !a.in_scale) return launch_conv32_teams_t<3, 2>(a, batch, whi, wlo, st);
        if (th == 4 && precision == EG_PREC_BF16X3 && teams == 3 && !a.in_scale) return launch_conv32_teams_t<3, 3>(a, batch, whi, wlo, st);
    }
