// Fused front half of an EfficientNet inverted-residual block:
//
//   expand 1x1 conv (MFMA) -> BN1 -> SiLU -> depthwise k x k (stride 1|2, TF-SAME) -> BN2 -> SiLU
//   (+ per-tile partial sums of the squeeze-excite global average pool)
//
// replaces timm's InvertedResidual.conv_pw/bn1/act1/conv_dw/bn2/act2 (reached from
// effdet/efficientdet.py:837).  The 6x-expanded activation - by far the largest tensor of the network -
// lives only in LDS: a workgroup owns a TH x TW output tile of one image, loads the input halo tile
// [(TH-1)*s+k] x [(TW-1)*s+k] x Cin once, and walks the expanded channels 64 at a time:
//     W1 chunk -> LDS;  E^T = W1_chunk * X^T by 16x16 MFMA tiles (channels on the accumulator rows, so
//     every lane holds 4 consecutive channels of one pixel and writes them with one 8-byte LDS store);
//     BN1 + SiLU in registers, zero outside the image (the depthwise conv pads the EXPANDED map);
//     depthwise taps out of LDS, BN2 + SiLU, 16-byte NHWC stores, pool partials.
// HBM traffic per tile = input halo + output tile (+ weights from L2).
#include "common.h"
// Phase-ablation switch of tools/ablate_gpu.sh: exists only in the separate -DEFFDET_ABLATE build
// (libeffdet_hip_ablate.so, `make ablate`); the product library has no run-time work-skipping switch.
#ifdef EFFDET_ABLATE
#include <cstdlib>
#define MB_DBG(p) ((p).dbg)
#else
#define MB_DBG(p) 0
#endif

namespace {

struct MbArgs {
    const void* X; void* Y;
    const float* in_gate;                       // optional [B][Cin]: X is multiplied by it (rounded to T) while the tile is loaded
    const void* W1;                             // [mid][Cin]  (T)
    const float* s1; const float* t1;           // [mid]
    const float* taps;                          // [k*k][mid]
    const float* s2; const float* t2;           // [mid]
    float* pool_partial;                        // [B][tiles][mid] or null
    int B, H, W, Cin, mid, Ho, Wo, k, stride, pad_t, pad_l;
    int TH, TW, IH, IW, HP, HPpad, tiles_x, tiles_y;
    int arow;                                   // LDS pitch of X / W1 rows (bytes)
    int e_bytes;                                // size of the expanded tile (also hosts the pool scratch)
    FastDiv fd_ppr, fd_iw, fd_tx;               // / (16-byte pieces per input row), / IW, / tiles_x
    int tw_shift;                               // TW = 1 << tw_shift
    int dbg;                                    // phase-ablation mask; only read in the -DEFFDET_ABLATE build
    int sym;                                    // padding convention (host side: forwarded to the rolling-window launchers)
};

constexpr int MC = 64;                          // expanded channels per pass
// Row pitch of the expanded tile in LDS: +16 bytes so that the 16 lanes which write one 4-channel column
// of 16 different pixels spread over the banks (a 128-byte pitch puts all of them on one bank: 16-way conflict)
template <typename T> struct ERow { static constexpr int value = MC + 16 / (int)sizeof(T); };

template <typename T> struct Pack4;
template <> struct Pack4<bf16_t> { typedef unsigned long long type; };
template <> struct Pack4<float> { typedef f32x4 type; };


// Spatial-tile form.  512 threads (8 waves) per workgroup: with the LDS footprint allowing two workgroups per
// CU this keeps 4 waves per SIMD in flight, which the VALU-heavy epilogues (SiLU on every expanded element)
// need to fill their issue slots (6 waves per workgroup measured 1.5x slower).  Expanded channels are processed
// SM_MC = 48 at a time: mid = 6 * Cin with Cin a multiple of 8, so 48 always divides mid and no pass runs half
// empty.  In the bf16 depthwise phase waves 0-5 own (channel tile w % 3, half w / 3 of the pixel tiles), so each
// builds its diagonal MFMA operands once per pass; waves 6-7 sit that phase out.
constexpr int SM_T = 512;                       // threads
constexpr int SM_NJ = 3;                        // 16-channel MFMA tiles per pass
constexpr int SM_MC = 16 * SM_NJ;               // 48 expanded channels per pass
constexpr int SM_CG = SM_MC / 8;                // 6 channel groups in the depthwise phase
constexpr int SM_PG = SM_T / SM_CG;             // 85 pixel-group threads per channel group (float32 depthwise; 510 threads work)
template <typename T> struct ERowS { static constexpr int value = SM_MC + 16 / (int)sizeof(T); };

template <typename T, int KS, int S>
__global__ __launch_bounds__(SM_T, 4) void mbconv_front_kernel(MbArgs p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 15, fpiece = lane >> 4;
    const int b = blockIdx.y;
    const int tile = blockIdx.x;
    const int tile_y = fdiv(tile, p.fd_tx);
    const int oy0 = tile_y * p.TH, ox0 = (tile - tile_y * p.tiles_x) * p.TW;
    const int iy0 = oy0 * S - p.pad_t, ix0 = ox0 * S - p.pad_l;
    const int Cin = p.Cin, mid = p.mid;
    const int cbytes = Cin * (int)sizeof(T);
    const int nkc = (cbytes + 63) / 64;
    const int arow = p.arow;
    constexpr int EROW = ERowS<T>::value;
    constexpr int NPAR = 4 + KS * KS;
    // LDS carve
    char* At = lds;                                         // [HPpad][arow]
    char* Wc = At + p.HPpad * arow;                         // [SM_MC][arow]
    T* E = reinterpret_cast<T*>(Wc + SM_MC * arow);         // [HP][EROW]
    float* red = reinterpret_cast<float*>(E);               // [SM_T][8] pool scratch: reuses E after the depthwise pass
    float* cpar = reinterpret_cast<float*>(reinterpret_cast<char*>(E) + p.e_bytes);   // [s1|t1|s2|t2][SM_MC] fp32, then taps
    float* ctap = cpar + 4 * SM_MC;                         // [KS*KS][SM_MC] fp32: the depthwise loop is VALU-bound, so no converts
    float* msk = cpar + NPAR * SM_MC;                       // [HPpad] 1 inside the image, 0 outside / padding rows
    const char* zslot = reinterpret_cast<const char*>(msk + p.HPpad);   // 16 zero bytes: operand of lanes past the row end
    if (tid < 4) reinterpret_cast<unsigned*>(msk + p.HPpad)[tid] = 0u;

    // ---- input halo tile -> LDS (zero rows outside the image); loads batched ahead of their LDS stores
    const T* X = reinterpret_cast<const T*>(p.X) + (long long)b * p.H * p.W * Cin;
    const int ppr = cbytes / 16;                            // 16-byte pieces per LDS row
    constexpr int AB = 3;                                   // pieces in flight per thread: one round trip covers every tile of d0..d4
    for (int i0 = tid; i0 < p.HPpad * ppr; i0 += AB * SM_T) {
        u32x4 v[AB];
#pragma unroll
        for (int u = 0; u < AB; ++u) {
            const int i = i0 + SM_T * u;
            v[u] = u32x4{0u, 0u, 0u, 0u};
            if (i < p.HPpad * ppr) {
                const int hp = fdiv(i, p.fd_ppr), piece = i - hp * ppr;
                bool inside = false;
                if (hp < p.HP) {
                    const int hy = fdiv(hp, p.fd_iw);
                    const int y = iy0 + hy, x = ix0 + hp - hy * p.IW;
                    inside = y >= 0 && y < p.H && x >= 0 && x < p.W;
                    if (inside) {
                        v[u] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(X + ((long long)y * p.W + x) * Cin) + piece * 16);
                    }
                }
                if (piece == 0) msk[hp] = inside ? 1.f : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < AB; ++u) {
            const int i = i0 + SM_T * u;
            if (i < p.HPpad * ppr) {
                const int hp = fdiv(i, p.fd_ppr), piece = i - hp * ppr;
                if (p.in_gate != nullptr) {
                    // SE gate of the producing block (its project conv was folded into W1), applied here - after every
                    // load of the batch has been issued - and rounded to T like the reference's gated tensor; zero rows stay zero
                    const float* g = p.in_gate + (long long)b * Cin + piece * (16 / (int)sizeof(T));
                    if constexpr (sizeof(T) == 2) {
                        bf16x8 xv = __builtin_bit_cast(bf16x8, v[u]);
#pragma unroll
                        for (int e = 0; e < 8; ++e) xv[e] = (bf16_t)((float)xv[e] * g[e]);
                        v[u] = __builtin_bit_cast(u32x4, xv);
                    } else {
                        f32x4 xv = __builtin_bit_cast(f32x4, v[u]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) xv[e] *= g[e];
                        v[u] = __builtin_bit_cast(u32x4, xv);
                    }
                }
                *reinterpret_cast<u32x4*>(At + hp * arow + piece * 16) = v[u];
            }
        }
    }

    T* Y = reinterpret_cast<T*>(p.Y) + (long long)b * p.Ho * p.Wo * mid;
    const int n_msub = p.HPpad / 16;

    // W1 and the per-channel constants of pass c+1 are fetched into registers at the start of pass c and
    // committed to LDS at the start of pass c+1: their latency hides behind the expand + depthwise work.
    constexpr int WPC = 1, PPC = (KS * KS * SM_MC + SM_T - 1) / SM_T;
    const bool w_pref = SM_MC * ppr <= SM_T * WPC;
    u32x4 wpre[WPC];
    float ppre[PPC], bnpre;
    auto fetch = [&](int c0n) {
        if (w_pref) {
#pragma unroll
            for (int q = 0; q < WPC; ++q) {
                const int i = tid + SM_T * q;
                wpre[q] = u32x4{0u, 0u, 0u, 0u};
                if (i < SM_MC * ppr) {
                    const int r = fdiv(i, p.fd_ppr);
                    wpre[q] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.W1) + (c0n + r) * cbytes + (i - r * ppr) * 16);
                }
            }
        }
        // taps: one base pointer + a 32-bit offset per lane; the four BN vectors: one predicated load each
#pragma unroll
        for (int q = 0; q < PPC; ++q) {
            const int i = tid + SM_T * q;
            ppre[q] = i < KS * KS * SM_MC ? p.taps[(i / SM_MC) * mid + c0n + i % SM_MC] : 0.f;
        }
        {
            const int r = tid / SM_MC, c = c0n + tid % SM_MC;
            float v = 0.f;
            if (r == 0) v = p.s1[c];
            if (r == 1) v = p.t1[c];
            if (r == 2) v = p.s2[c];
            if (r == 3) v = p.t2[c];
            bnpre = v;
        }
    };
    auto commit = [&](int c0n) {
        if (w_pref) {
#pragma unroll
            for (int q = 0; q < WPC; ++q) {
                const int i = tid + SM_T * q;
                if (i < SM_MC * ppr) { const int r = fdiv(i, p.fd_ppr); *reinterpret_cast<u32x4*>(Wc + r * arow + (i - r * ppr) * 16) = wpre[q]; }
            }
        } else {
            for (int i = tid; i < SM_MC * ppr; i += SM_T)
                *reinterpret_cast<u32x4*>(Wc + (i / ppr) * arow + (i % ppr) * 16) =
                    *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.W1) + (long long)(c0n + i / ppr) * cbytes + (i % ppr) * 16);
        }
#pragma unroll
        for (int q = 0; q < PPC; ++q) {
            const int i = tid + SM_T * q;
            if (i < KS * KS * SM_MC) ctap[i] = ppre[q];
        }
        if (tid < 4 * SM_MC) cpar[tid] = bnpre;
    };
    fetch(0);

    // Operand addresses of the expand GEMM are lane-invariant: a lane whose 16-byte piece lies past the end of
    // the (tight) row reads the zero slot instead, so the loop carries no per-lane branches.
    constexpr int HK = KS == 3 ? 2 : 1;                                   // K-chunks whose W fragments are held in registers
    const bool hoist = nkc <= HK;
    const int fo = fpiece * 16;
    const char* xa[HK];
    int xs[HK];
#pragma unroll
    for (int kc = 0; kc < HK; ++kc) {
        const bool ok = kc * 64 + fo < cbytes;
        xa[kc] = ok ? At + (16 * wave + frow) * arow + kc * 64 + fo : zslot;
        xs[kc] = ok ? 16 * (SM_T / 64) * arow : 0;
    }

    // bf16 depthwise on the matrix cores (see mbconv_deep_kernel): lane constants of the diagonal operands
    constexpr bool MF = sizeof(T) == 2;
    constexpr int NTAP = KS * KS, NPAIR = (NTAP + 1) / 2;
    constexpr int NH = 2;                                   // waves per channel tile (waves >= NH * SM_NJ idle in that phase)
    const int dj = wave % SM_NJ, dhalf = wave / SM_NJ;
    const int dhi = fpiece >> 1;
    const bool dactive = (fpiece & 1) == (frow >> 3);
    const int ddq = (frow & 7) >> 1;
    // byte offset of tap t inside the expanded tile; per pair a lane picks the even or the odd tap's (uniform) offset
    auto tap_off = [&](int pr) {
        const int t0 = 2 * pr, t1 = 2 * pr + 1 < NTAP ? 2 * pr + 1 : 0;
        const int o0 = ((t0 / KS) * p.IW + (t0 % KS)) * EROW * (int)sizeof(T);
        const int o1 = ((t1 / KS) * p.IW + (t1 % KS)) * EROW * (int)sizeof(T);
        return dhi ? o1 : o0;
    };

    for (int c0 = 0; c0 < mid; c0 += SM_MC) {
        __syncthreads();                                    // previous pass done with Wc / E / red / cpar
        commit(c0);
        __syncthreads();
        if (c0 + SM_MC < mid) fetch(c0 + SM_MC);
        // ---- expand: rows of the accumulator = channels, columns = halo pixels
        Frag<T> wreg[HK][SM_NJ];
        if (hoist) {
#pragma unroll
            for (int kc = 0; kc < HK; ++kc)
#pragma unroll
                for (int j = 0; j < SM_NJ; ++j)
                    wreg[kc][j] = ld_frag<T>(kc * 64 + fo < cbytes ? Wc + (16 * j + frow) * arow + kc * 64 + fo : zslot);
        }
        const char* xp[HK];
#pragma unroll
        for (int kc = 0; kc < HK; ++kc) xp[kc] = xa[kc];
        for (int ms = wave; ms < ((MB_DBG(p) & 1) ? 0 : n_msub); ms += SM_T / 64) {
            f32x4 acc[SM_NJ];
#pragma unroll
            for (int j = 0; j < SM_NJ; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int hp = 16 * ms + frow;
            const float m = msk[hp];
            if (hoist) {
#pragma unroll
                for (int kc = 0; kc < HK; ++kc) {
                    if (kc < nkc) {
                        const Frag<T> xf = ld_frag<T>(xp[kc]);
#pragma unroll
                        for (int j = 0; j < SM_NJ; ++j) mma_chunk(wreg[kc][j], xf, acc[j]);
                    }
                    xp[kc] += xs[kc];
                }
            } else {
                for (int kc = 0; kc < nkc; ++kc) {
                    const int off = kc * 64 + fo;
                    const Frag<T> xf = ld_frag<T>(off < cbytes ? At + hp * arow + off : zslot);
#pragma unroll
                    for (int j = 0; j < SM_NJ; ++j) {
                        const Frag<T> wf = ld_frag<T>(off < cbytes ? Wc + (16 * j + frow) * arow + off : zslot);
                        mma_chunk(wf, xf, acc[j]);
                    }
                }
            }
            // BN + SiLU on all 12 values at once (independent chains), zeroed outside the image by the mask
#pragma unroll
            for (int j = 0; j < SM_NJ; ++j) {
                const f32x4 sc = *reinterpret_cast<const f32x4*>(cpar + 16 * j + 4 * fpiece);
                const f32x4 sh = *reinterpret_cast<const f32x4*>(cpar + SM_MC + 16 * j + 4 * fpiece);
                const f32x4 v = bn_silu4<T>(acc[j], sc, sh) * m;
                store4<T>(E + hp * EROW + 16 * j + 4 * fpiece, v[0], v[1], v[2], v[3]);
            }
        }
        __syncthreads();
        if constexpr (MF) {
            // ---- depthwise on the matrix cores: wave (dj, dhalf) owns channel tile dj and every NH-th pixel tile
            float plr[4] = {0.f, 0.f, 0.f, 0.f};
            if (wave < NH * SM_NJ && !(MB_DBG(p) & 2)) {
                // the lane's single non-zero dword of each diagonal operand; expanded to the 16-byte fragment at use
                unsigned abits[NPAIR];
#pragma unroll
                for (int pr = 0; pr < NPAIR; ++pr) {
                    const int t = 2 * pr + dhi;
                    const bool on = dactive && t < NTAP;
                    const float wv = on ? ctap[(t < NTAP ? t : 0) * SM_MC + 16 * dj + frow] : 0.f;
                    abits[pr] = (unsigned)__builtin_bit_cast(unsigned short, (bf16_t)wv) << (16 * (frow & 1));
                }
                const f32x4 s2v = *reinterpret_cast<const f32x4*>(cpar + 2 * SM_MC + 16 * dj + 4 * fpiece);
                const f32x4 t2v = *reinterpret_cast<const f32x4*>(cpar + 3 * SM_MC + 16 * dj + 4 * fpiece);
                const char* Eb = reinterpret_cast<const char*>(E) + (16 * dj + 8 * (fpiece & 1)) * (int)sizeof(T);
                const int npix = p.TH * p.TW;
                float pl[4] = {0.f, 0.f, 0.f, 0.f};
                // two pixel tiles per step share every expanded diagonal operand (the expansion is the larger part of
                // the per-MFMA vector work)
                constexpr int UT = 2;
                for (int q0 = 16 * dhalf; q0 < npix; q0 += 16 * NH * UT) {
                    const char* base[UT];
                    int oy[UT], ox[UT];
                    bool ok[UT];
                    f32x4 acc[UT];
#pragma unroll
                    for (int u = 0; u < UT; ++u) {
                        const int q = q0 + 16 * NH * u + frow;
                        const int qq = q < npix ? q : 0;
                        const int ty = qq >> p.tw_shift, tx = qq & (p.TW - 1);
                        oy[u] = oy0 + ty; ox[u] = ox0 + tx;
                        ok[u] = q < npix && oy[u] < p.Ho && ox[u] < p.Wo;
                        base[u] = Eb + ((ty * S) * p.IW + tx * S) * (EROW * (int)sizeof(T));
                        acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                    const bool second = q0 + 16 * NH < npix;       // uniform
#pragma unroll
                    for (int pr = 0; pr < NPAIR; ++pr) {
                        unsigned bits = abits[pr];
                        asm volatile("" : "+v"(bits));           // keep the expansion here: hoisted, the fragments would not fit the registers
                        const u32x4 fr = {ddq == 0 ? bits : 0u, ddq == 1 ? bits : 0u, ddq == 2 ? bits : 0u, ddq == 3 ? bits : 0u};
                        Frag<T> af; af.v = __builtin_bit_cast(bf16x8, fr);
                        const int to = tap_off(pr);
                        // both pixel tiles unconditionally (an absent second tile reads pixel 0's operands and is dropped at the
                        // store): no exec-mask branch between the two LDS loads, so they are in flight together
                        const Frag<T> b0 = ld_frag<T>(base[0] + to), b1 = ld_frag<T>(base[1] + to);
                        mma_chunk(af, b0, acc[0]);
                        mma_chunk(af, b1, acc[1]);
                    }
#pragma unroll
                    for (int u = 0; u < UT; ++u) {
                        if (u == 1 && !second) break;
                        float o[4];
                        const f32x4 ov = bn_silu4<T>(acc[u], s2v, t2v);
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = to_f<T>(from_f<T>(ov[r]));   // SE averages what the next layer reads
                        if (ok[u]) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) pl[r] += o[r];
                            store4<T>(Y + ((long long)oy[u] * p.Wo + ox[u]) * mid + c0 + 16 * dj + 4 * fpiece, o[0], o[1], o[2], o[3]);
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = pl[r];
                    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
                    plr[r] = v;
                }
            }
            if (p.pool_partial != nullptr) {
                __syncthreads();                            // every wave is done reading E: `red` may overwrite it
                if (dhalf > 0 && dhalf < NH && frow == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[(dhalf - 1) * SM_MC + 16 * dj + 4 * fpiece + r] = plr[r];
                }
                __syncthreads();
                if (dhalf == 0 && frow == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = plr[r];
#pragma unroll
                        for (int h = 1; h < NH; ++h) v += red[(h - 1) * SM_MC + 16 * dj + 4 * fpiece + r];
                        p.pool_partial[((long long)b * (p.tiles_x * p.tiles_y) + tile) * mid + c0 + 16 * dj + 4 * fpiece + r] = v;
                    }
                }
            }
        } else {
            // ---- float32: depthwise on the vector ALU out of LDS, a thread owns 8 channels of one output pixel at a time
            F8 pool = f8_zero();
            const int cg = tid % SM_CG, pg0 = tid / SM_CG;
            if (pg0 < SM_PG && !(MB_DBG(p) & 2)) {
                for (int px = pg0; px < p.TH * p.TW; px += SM_PG) {
                    const int ty = px >> p.tw_shift, tx = px & (p.TW - 1);
                    const int oy = oy0 + ty, ox = ox0 + tx;
                    if (oy >= p.Ho || ox >= p.Wo) continue;
                    F8 acc = f8_zero();
#pragma unroll 1
                    for (int ky = 0; ky < KS; ++ky) {      // one tap / one LDS vector at a time: small register footprint
                        const T* erow = E + ((ty * S + ky) * p.IW + tx * S) * EROW + cg * 8;
                        const float* wrow = ctap + (ky * KS) * SM_MC + cg * 8;
#pragma unroll
                        for (int kx = 0; kx < KS; ++kx) {
                            const F8 e = load8<T>(erow + kx * EROW);
                            const F8 w = load8<float>(wrow + kx * SM_MC);
#pragma unroll
                            for (int q = 0; q < 8; ++q) acc.v[q] = fmaf(e.v[q], w.v[q], acc.v[q]);
                        }
                    }
                    const F8 s2 = load8<float>(cpar + 2 * SM_MC + cg * 8), t2 = load8<float>(cpar + 3 * SM_MC + cg * 8);
                    F8 o;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        o.v[q] = silu_t<T>(acc.v[q] * s2.v[q] + t2.v[q]);
                        pool.v[q] += o.v[q];
                    }
                    store8<T>(Y + ((long long)oy * p.Wo + ox) * mid + c0 + cg * 8, o);
                }
            }
            if (p.pool_partial != nullptr) {
                __syncthreads();                            // every thread is done reading E
                const float tot = pool_reduce<SM_T>(pool, red, red + SM_T * 8, tid, SM_CG, SM_CG * SM_PG);
                if (tid < SM_MC) p.pool_partial[((long long)b * (p.tiles_x * p.tiles_y) + tile) * mid + c0 + tid] = tot;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// "Deep" decomposition for the late stages (small maps, many channels): a workgroup owns
// (image, band of output rows, slice of 64 expanded channels).  The W1 slice is read once into LDS,
// the input rows of the band are streamed from L2 straight into MFMA operand registers (they are
// shared by all channel slices of the band), the expanded band [rows x W x 64] lives in LDS and the
// depthwise conv walks it with explicit border checks (no x halo is stored).
// ------------------------------------------------------------------------------------------------
struct MbDeepArgs {
    const void* X; void* Y; const void* W1;
    const float* s1; const float* t1; const float* taps; const float* s2; const float* t2;
    float* pool_partial;
    int B, H, W, Cin, mid, Ho, Wo, pad_t, pad_l;
    int band_rows, nbands, nchunks, arow, e_rows_max;
    int dbg;
    int we;                                      // bf16: columns of the zero-haloed expanded band
    FastDiv fd_w, fd_wo;
};

// NTH: 512 threads in bf16 (two workgroups per CU -> four waves per SIMD to cover LDS / MFMA latencies)
template <typename T, int KS, int S, int PPT, int NTH>
__global__ __launch_bounds__(NTH, NTH == 512 ? 4 : 2) void mbconv_deep_kernel(MbDeepArgs p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 15, fpiece = lane >> 4;
    // blocks are dealt round-robin over the 8 XCDs: image = (round, xcd), so that all channel slices / bands of an image run on
    // ONE XCD and its L2 fetches that image's X once (it was fetched by up to 8 L2s: 2.6x read amplification in the counters)
    const int xcd_ = blockIdx.x & 7, rr_ = blockIdx.x >> 3;
    const int per_image_ = p.nchunks * p.nbands;
    const int b = (rr_ / per_image_) * 8 + xcd_;
    if (b >= p.B) return;
    const int bx_ = rr_ % per_image_;
    const int chunk = bx_ % p.nchunks, band = bx_ / p.nchunks;
    const int Cin = p.Cin, mid = p.mid, W = p.W;
    const int cbytes = Cin * (int)sizeof(T);
    const int nkc = (cbytes + 63) / 64;
    const int arow = p.arow;
    const int c0 = chunk * MC;
    const int cn = (mid - c0) < MC ? (mid - c0) : MC;
    const int oy_b = band * p.band_rows;
    const int oy_e = min(p.Ho, oy_b + p.band_rows);
    const int iy_lo = max(0, oy_b * S - p.pad_t);
    const int iy_hi = min(p.H, (oy_e - 1) * S - p.pad_t + KS);
    const int npx = (iy_hi - iy_lo) * W;

    char* Wc = lds;                                            // [MC][arow]; reused as pool scratch
    float* red = reinterpret_cast<float*>(lds);
    const int wc_bytes = (MC * arow > 256 * 9 * 4) ? MC * arow : 256 * 9 * 4;
    constexpr int EROW = ERow<T>::value;
    T* E = reinterpret_cast<T*>(lds + wc_bytes);               // fp32: [npx][EROW]; bf16: [rows][we][EROW], zero halo
    // bf16 runs the depthwise taps on the matrix cores, which want branch-free operand addresses: the band is
    // stored with its TF-SAME zero padding (rows above / below the image and the left / right columns)
    constexpr bool MF = sizeof(T) == 2;
    const int iy_top = oy_b * S - p.pad_t;                     // first (possibly virtual) input row of the band
    if constexpr (MF) {
        const int e_rows = (oy_e - 1 - oy_b) * S + KS;
        const int n16 = e_rows * p.we * EROW * (int)sizeof(T) / 16;
        for (int i = tid; i < n16; i += NTH) reinterpret_cast<u32x4*>(E)[i] = u32x4{0u, 0u, 0u, 0u};
    }

    const int ppr = nkc * 4;
    {   // W1 slice -> LDS: all of a thread's pieces are loaded before the first LDS store (ppr <= 16 pieces per row)
        constexpr int WPT_MAX = 4;
        for (int i0 = tid; i0 < MC * ppr; i0 += NTH * WPT_MAX) {
            u32x4 v[WPT_MAX];
#pragma unroll
            for (int u = 0; u < WPT_MAX; ++u) {
                const int i = i0 + NTH * u;
                v[u] = u32x4{0u, 0u, 0u, 0u};
                if (i < MC * ppr) {
                    const int row = i / ppr, piece = i % ppr;
                    if (row < cn && piece * 16 < cbytes)
                        v[u] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.W1) + (long long)(c0 + row) * cbytes + piece * 16);
                }
            }
#pragma unroll
            for (int u = 0; u < WPT_MAX; ++u) {
                const int i = i0 + NTH * u;
                if (i < MC * ppr) *reinterpret_cast<u32x4*>(Wc + (i / ppr) * arow + (i % ppr) * 16) = v[u];
            }
        }
    }
    f32x4 sc[4], sh[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (16 * j + 4 * fpiece < cn) {
            sc[j] = *reinterpret_cast<const f32x4*>(p.s1 + c0 + 16 * j + 4 * fpiece);
            sh[j] = *reinterpret_cast<const f32x4*>(p.t1 + c0 + 16 * j + 4 * fpiece);
        } else {
            sc[j] = f32x4{0.f, 0.f, 0.f, 0.f}; sh[j] = sc[j];
        }
    }
    // depthwise constants of this channel slice: fetched now, parked in the W slot once the expand is done
    constexpr int NPAR = 2 + KS * KS;                         // s2 | t2 | taps
    constexpr int PPC = (NPAR * MC + NTH - 1) / NTH;
    float ppre[PPC], bnpre;
#pragma unroll
    for (int q = 0; q < PPC; ++q) {
        const int i = tid + NTH * q;
        float v = 0.f;
        if (i < NPAR * MC) {
            const int r = i / MC, c = i % MC;
            if (c < cn) v = (r == 0 ? p.s2 : r == 1 ? p.t2 : p.taps + (long long)(r - 2) * mid)[c0 + c];
        }
        ppre[q] = v;
    }
    __syncthreads();
    // ---- expand the band: two 16-pixel sub-tiles per step share every W fragment read
    const char* Xb = reinterpret_cast<const char*>(p.X) + ((long long)b * p.H * W + (long long)iy_lo * W) * cbytes;
    const int n_pair = (npx + 31) / 32;
    for (int mp = wave; mp < ((MB_DBG(p) & 1) ? 0 : n_pair); mp += NTH / 64) {
        f32x4 acc[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[u][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int hp0 = 32 * mp + frow, hp1 = hp0 + 16;
        const char* x0 = Xb + (long long)(hp0 < npx ? hp0 : 0) * cbytes;
        const char* x1 = Xb + (long long)(hp1 < npx ? hp1 : 0) * cbytes;
        // X fragments of four K-chunks are fetched together (one exposed L2 round trip per four chunks
        // instead of one per chunk), then consumed by the MFMAs
        for (int kc0 = 0; kc0 < nkc; kc0 += 4) {
            Frag<T> xa[4], xb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int off = (kc0 + u) * 64 + fpiece * 16;
                if (kc0 + u < nkc && off < cbytes) { xa[u] = ld_frag<T>(x0 + off); xb[u] = ld_frag<T>(x1 + off); }
                else { xa[u].v = decltype(xa[u].v){}; xb[u].v = decltype(xb[u].v){}; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (kc0 + u < nkc) {
                    const int off = (kc0 + u) * 64 + fpiece * 16;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const Frag<T> wf = ld_frag<T>(Wc + (16 * j + frow) * arow + off);
                        mma_chunk(wf, xa[u], acc[0][j]);
                        mma_chunk(wf, xb[u], acc[1][j]);
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int hp = u == 0 ? hp0 : hp1;
            if (hp < npx) {
                int ei = hp;
                if constexpr (MF) {
                    const int r = fdiv(hp, p.fd_w);
                    ei = (iy_lo - iy_top + r) * p.we + (hp - r * W) + p.pad_l;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 v = bn_silu4<T>(acc[u][j], sc[j], sh[j]);
                    store4<T>(E + ei * EROW + 16 * j + 4 * fpiece, v[0], v[1], v[2], v[3]);
                }
            }
        }
    }
    __syncthreads();
    float* cpar = reinterpret_cast<float*>(lds);               // reuses the W slot: [s2 | t2 | taps][MC]
#pragma unroll
    for (int q = 0; q < PPC; ++q) {
        const int i = tid + NTH * q;
        if (i < NPAR * MC) cpar[i] = ppre[q];
    }
    __syncthreads();
    T* Y = reinterpret_cast<T*>(p.Y) + (long long)b * p.Ho * p.Wo * mid;
    if constexpr (MF) {
        // ---- depthwise on the matrix cores.  For a pair of taps (t0, t1) and a 16-channel tile:
        //   A[m][k] (16 x 32) = diag(w[t0]) | diag(w[t1])      (k < 16: tap t0, k >= 16: tap t1)
        //   B[n][k]           = E[pixel n shifted by t0][16 ch] | E[pixel n shifted by t1][16 ch]
        // so one 16x16x32 MFMA accumulates two taps of 16 channels x 16 pixels.  The accumulator holds 4
        // channels x 1 pixel per lane: BN + SiLU, the SE pool sums and the 8-byte store all stay in registers.
        // Wave w owns channel tile w for the whole band, so its diagonal operands are built once.
        constexpr int NH = NTH / 256;                              // waves that share one channel tile
        const int j = wave & 3, half = wave >> 2;
        float* pl_x = reinterpret_cast<float*>(lds) + (NPAR * MC);  // [NH-1][64] pool sums handed to the first wave of a tile
        float plr[4] = {0.f, 0.f, 0.f, 0.f};
        const bool ch_ok = 16 * j + 4 * fpiece < cn;
        if (16 * j < cn && !(MB_DBG(p) & 2)) {
            constexpr int NTAP = KS * KS, NPAIR = (NTAP + 1) / 2;
            const int hi = fpiece >> 1;
            const bool active = (fpiece & 1) == (frow >> 3);
            const int dq = (frow & 7) >> 1;
            Frag<T> afr[NPAIR];
            int toff[NPAIR];
#pragma unroll
            for (int pr = 0; pr < NPAIR; ++pr) {
                const int t = 2 * pr + hi;                                   // this lane's tap of the pair
                const bool on = active && t < NTAP;
                const float wv = on ? cpar[(2 + (t < NTAP ? t : 0)) * MC + 16 * j + frow] : 0.f;
                const unsigned bits = (unsigned)__builtin_bit_cast(unsigned short, (bf16_t)wv) << (16 * (frow & 1));
                const u32x4 fr = {dq == 0 ? bits : 0u, dq == 1 ? bits : 0u, dq == 2 ? bits : 0u, dq == 3 ? bits : 0u};
                afr[pr].v = __builtin_bit_cast(bf16x8, fr);
                const int tt = t < NTAP ? t : 0;
                toff[pr] = ((tt / KS) * p.we + (tt % KS)) * EROW * (int)sizeof(T);
            }
            const f32x4 s2v = *reinterpret_cast<const f32x4*>(cpar + 16 * j + 4 * fpiece);
            const f32x4 t2v = *reinterpret_cast<const f32x4*>(cpar + MC + 16 * j + 4 * fpiece);
            float pl[4] = {0.f, 0.f, 0.f, 0.f};
            const int n_out = (oy_e - oy_b) * p.Wo;
            const char* Eb = reinterpret_cast<const char*>(E) + (16 * j + 8 * (fpiece & 1)) * (int)sizeof(T);
            for (int q0 = 16 * half; q0 < n_out; q0 += 16 * NH) {
                const int q = q0 + frow;
                const bool qv = q < n_out;
                const int qq = qv ? q : 0;
                const int oyr = fdiv(qq, p.fd_wo), ox = qq - oyr * p.Wo;
                const char* base = Eb + ((oyr * S) * p.we + ox * S) * (EROW * (int)sizeof(T));
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int pr = 0; pr < NPAIR; ++pr) mma_chunk(afr[pr], ld_frag<T>(base + toff[pr]), acc);
                float o[4];
                const f32x4 ov = bn_silu4<T>(acc, s2v, t2v);
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = to_f<T>(from_f<T>(ov[r]));
                if (qv && ch_ok) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) pl[r] += o[r];
                    store4<T>(Y + ((long long)(oy_b + oyr) * p.Wo + ox) * mid + c0 + 16 * j + 4 * fpiece, o[0], o[1], o[2], o[3]);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = pl[r];
                v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
                plr[r] = v;
            }
        }
        if (p.pool_partial != nullptr) {
            if constexpr (NH > 1) {                                // second-half waves hand their sums over through LDS
                __syncthreads();                                   // every wave is done reading the constants
                if (half > 0 && frow == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) pl_x[(half - 1) * 64 + 16 * j + 4 * fpiece + r] = plr[r];
                }
                __syncthreads();
                if (half == 0) {
#pragma unroll
                    for (int h = 1; h < NH; ++h)
#pragma unroll
                        for (int r = 0; r < 4; ++r) plr[r] += pl_x[(h - 1) * 64 + 16 * j + 4 * fpiece + r];
                }
            }
            if (half == 0 && frow == 0 && ch_ok && 16 * j < cn) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    p.pool_partial[((long long)b * p.nbands + band) * mid + c0 + 16 * j + 4 * fpiece + r] = plr[r];
            }
        }
        return;
    }
    // ---- float32: depthwise on the vector ALU over the band, borders by index checks
    const int cgn = cn / 8;
    F8 pool = f8_zero();
    const int cg = tid & 7;
    if (cg < cgn && !(MB_DBG(p) & 2)) {
        const F8 s2 = load8<float>(cpar + cg * 8), t2 = load8<float>(cpar + MC + cg * 8);
        const int gpr = (p.Wo + PPT - 1) / PPT;
        for (int pg = tid >> 3; pg < (oy_e - oy_b) * gpr; pg += NTH / 8) {
            const int oy = oy_b + pg / gpr, ox0 = (pg % gpr) * PPT;
            F8 acc[PPT];
#pragma unroll
            for (int pi = 0; pi < PPT; ++pi) acc[pi] = f8_zero();
#pragma unroll 1
            for (int ky = 0; ky < KS; ++ky) {
                const int iy = oy * S - p.pad_t + ky;
                if (iy < 0 || iy >= p.H) continue;
                F8 w[KS];
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) w[kx] = load8<float>(cpar + (2 + ky * KS + kx) * MC + cg * 8);
                const T* erow = E + (long long)(iy - iy_lo) * W * EROW + cg * 8;
                const int ix0 = ox0 * S - p.pad_l;
#pragma unroll
                for (int c = 0; c < (PPT - 1) * S + KS; ++c) {
                    const int ix = ix0 + c;
                    if (ix < 0 || ix >= W) continue;
                    const F8 e = load8<T>(erow + ix * EROW);
#pragma unroll
                    for (int pi = 0; pi < PPT; ++pi) {
                        const int kx = c - pi * S;
                        if (kx >= 0 && kx < KS) {
#pragma unroll
                            for (int q = 0; q < 8; ++q) acc[pi].v[q] = fmaf(e.v[q], w[kx].v[q], acc[pi].v[q]);
                        }
                    }
                }
            }
#pragma unroll
            for (int pi = 0; pi < PPT; ++pi) {
                const int ox = ox0 + pi;
                if (ox >= p.Wo) continue;
                F8 o;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float v = silu_t<T>(acc[pi].v[q] * s2.v[q] + t2.v[q]);
                    o.v[q] = to_f<T>(from_f<T>(v));
                    pool.v[q] += o.v[q];
                }
                store8<T>(Y + ((long long)oy * p.Wo + ox) * mid + c0 + cg * 8, o);
            }
        }
    }
    if (p.pool_partial != nullptr) {
        __syncthreads();                                       // every wave is done reading the W slot
        const float tot = pool_reduce<NTH>(pool, red, red + NTH * 8, tid, 8, NTH);    // cg = tid & 7; idle groups hold zeros
        if (tid < cn) p.pool_partial[((long long)b * p.nbands + band) * mid + c0 + tid] = tot;
    }
}

struct DeepGeometry { bool use; int band_rows, nbands, nchunks, arow, e_rows_max, we; size_t lds; };

template <typename T>
DeepGeometry pick_deep_budget(int H, int W, int Cin, int mid, int k, int stride, size_t budget) {
    DeepGeometry g{};
    const int Ho = same_out(H, stride);
    const int nkc = (Cin * (int)sizeof(T) + 63) / 64;
    g.arow = nkc * 64 + 16;
    g.nchunks = (mid + MC - 1) / MC;
    const size_t wc = (size_t)MC * g.arow > 9216 ? (size_t)MC * g.arow : 9216;
    g.use = false;
    // bf16 stores the band with its zero padding: (Wo-1)*stride + k columns and unclipped rows
    const bool mf = sizeof(T) == 2;
    const int Wo = same_out(W, stride);
    g.we = mf ? (Wo - 1) * stride + k : W;
    const size_t row_bytes = (size_t)g.we * ERow<T>::value * sizeof(T);
    if (Cin * (int)sizeof(T) < 128 || wc + (size_t)k * row_bytes > budget) return g;   // wide inputs, narrow maps only
    for (int rows = Ho; rows >= 1; --rows) {
        int in_rows = (rows - 1) * stride + k;
        if (!mf && in_rows > H) in_rows = H;
        const size_t lds = wc + (size_t)in_rows * row_bytes;
        if (lds <= budget) { g.band_rows = rows; g.e_rows_max = in_rows; g.lds = lds; break; }
        if (rows == 1) return g;
    }
    // equal bands: no nearly empty last band
    g.nbands = (Ho + g.band_rows - 1) / g.band_rows;
    g.band_rows = (Ho + g.nbands - 1) / g.nbands;
    {
        int in_rows = (g.band_rows - 1) * stride + k;
        if (!mf && in_rows > H) in_rows = H;
        g.e_rows_max = in_rows; g.lds = wc + (size_t)in_rows * row_bytes;
    }
    if (g.band_rows < 3 && g.band_rows < Ho) return g;          // too much halo recompute: use spatial tiles
    g.nbands = (Ho + g.band_rows - 1) / g.band_rows;
    g.use = true;
    return g;
}

// Two workgroups per CU (78 KiB each) where a useful band fits; very wide inputs (the W1 slice of a 448-channel
// block alone is 58 KiB) take the whole CU rather than falling back to tiny spatial tiles.
template <typename T>
DeepGeometry pick_deep(int H, int W, int Cin, int mid, int k, int stride) {
    DeepGeometry g = pick_deep_budget<T>(H, W, Cin, mid, k, stride, 78 * 1024);
    if (!g.use && Cin * (int)sizeof(T) >= 512) g = pick_deep_budget<T>(H, W, Cin, mid, k, stride, 156 * 1024);
    return g;
}

struct Geometry { int TH, TW, IH, IW, HP, HPpad, arow, e_bytes; size_t lds; };

template <typename T>
Geometry pick_tile(int Ho, int Wo, int Cin, int k, int stride) {
    const int cand[][2] = {{16, 16}, {8, 16}, {8, 8}, {4, 16}, {4, 8}, {4, 4}, {2, 4}};     // TW is a power of two >= 4
    Geometry best{};
    for (auto& c : cand) {
        Geometry g;
        g.TH = c[0]; g.TW = c[1];
        g.IH = (g.TH - 1) * stride + k; g.IW = (g.TW - 1) * stride + k;
        g.HP = g.IH * g.IW; g.HPpad = (g.HP + 15) / 16 * 16;
        g.arow = Cin * (int)sizeof(T) + 16;
        g.e_bytes = g.HPpad * ERowS<T>::value * (int)sizeof(T);
        if (g.e_bytes < SM_T * 9 * 4) g.e_bytes = SM_T * 9 * 4;       // also hosts the pool-reduction scratch
        g.e_bytes = (g.e_bytes + 15) / 16 * 16;
        g.lds = (size_t)g.HPpad * g.arow + (size_t)SM_MC * g.arow + (size_t)g.e_bytes + (size_t)(4 + k * k) * SM_MC * 4
              + (size_t)g.HPpad * 4 + 16;               // + inside-mask + zero slot
        best = g;
        // a tile much larger than the map wastes the workgroup; keep two workgroups per CU (<= 76 KiB each)
        const bool fits_map = (g.TH <= Ho || g.TH == 2) && (g.TW <= 2 * Wo);
        if (g.lds <= 78 * 1024 && fits_map) return g;
    }
    return best;
}

template <typename T>
int launch_deep(hipStream_t st, const MbArgs& a, const DeepGeometry& g) {
    MbDeepArgs d{a.X, a.Y, a.W1, a.s1, a.t1, a.taps, a.s2, a.t2, a.pool_partial, a.B, a.H, a.W, a.Cin, a.mid, a.Ho, a.Wo,
                 a.pad_t, a.pad_l, g.band_rows, g.nbands, g.nchunks, g.arow, g.e_rows_max, a.dbg, g.we,
                 make_fastdiv(a.W), make_fastdiv(a.Wo)};
    void (*kern)(MbDeepArgs) = nullptr;
    constexpr int NTH = sizeof(T) == 2 ? 512 : 256;
    if (a.k == 3) kern = a.stride == 1 ? mbconv_deep_kernel<T, 3, 1, 4, NTH> : mbconv_deep_kernel<T, 3, 2, 4, NTH>;
    else kern = a.stride == 1 ? mbconv_deep_kernel<T, 5, 1, 4, NTH> : mbconv_deep_kernel<T, 5, 2, 4, NTH>;
    if (g.lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return EFFDET_ELAUNCH;
    }
    hipLaunchKernelGGL(kern, dim3(((a.B + 7) / 8) * g.nchunks * g.nbands * 8), dim3(NTH), g.lds, st, d);
    return effdet_check_launch();
}

template <typename T>
int launch_mb(hipStream_t st, MbArgs& a) {
    if constexpr (sizeof(T) == 2) {
#ifdef MB_PREFER_WIDE    /* experiment (variant builds only) */
        if (!a.in_gate && effdet_mbconv_wide_parts(a.H, a.W, a.Cin, a.mid, a.k, a.stride) > 0)
            return effdet_mbconv_wide_launch(st, a.X, a.Y, a.W1, a.s1, a.t1, a.taps, a.s2, a.t2, a.pool_partial,
                                             a.B, a.H, a.W, a.Cin, a.mid, a.k, a.stride, 0, a.sym);
#endif
        // bf16: the rolling-window form (mbconv_roll.hip) wherever its geometry applies
        if (effdet_mbconv_roll_parts(a.H, a.W, a.Cin, a.mid, a.k, a.stride) > 0)
            return effdet_mbconv_roll_launch(st, a.X, a.in_gate, a.Y, a.W1, a.s1, a.t1, a.taps, a.s2, a.t2, a.pool_partial,
                                             a.B, a.H, a.W, a.Cin, a.mid, a.k, a.stride, 0, a.sym);
        // wider inputs: the rolling-window form with the X rows shared through LDS (mbconv_wide.hip)
        if (!a.in_gate && effdet_mbconv_wide_parts(a.H, a.W, a.Cin, a.mid, a.k, a.stride) > 0)
            return effdet_mbconv_wide_launch(st, a.X, a.Y, a.W1, a.s1, a.t1, a.taps, a.s2, a.t2, a.pool_partial,
                                             a.B, a.H, a.W, a.Cin, a.mid, a.k, a.stride, 0, a.sym);
    }
    const DeepGeometry dg = pick_deep<T>(a.H, a.W, a.Cin, a.mid, a.k, a.stride);
    if (dg.use && !a.in_gate) return launch_deep<T>(st, a, dg);                    // gated inputs always take the spatial form
    const Geometry g = pick_tile<T>(a.Ho, a.Wo, a.Cin, a.k, a.stride);
    if (g.lds > 160 * 1024) return EFFDET_EINVAL;
    a.TH = g.TH; a.TW = g.TW; a.IH = g.IH; a.IW = g.IW; a.HP = g.HP; a.HPpad = g.HPpad; a.arow = g.arow; a.e_bytes = g.e_bytes;
    a.tiles_x = (a.Wo + g.TW - 1) / g.TW; a.tiles_y = (a.Ho + g.TH - 1) / g.TH;
    a.fd_ppr = make_fastdiv(a.Cin * (int)sizeof(T) / 16); a.fd_iw = make_fastdiv(g.IW); a.fd_tx = make_fastdiv(a.tiles_x);
    a.tw_shift = 0; while ((1 << a.tw_shift) < g.TW) ++a.tw_shift;             // candidates have TW in {4, 8, 16}
    dim3 grid(a.tiles_x * a.tiles_y, a.B), block(SM_T);
    if (a.mid % SM_MC) return EFFDET_EINVAL;               // mid = 6 * Cin: always a multiple of 48
    void (*kern)(MbArgs) = nullptr;
#define MB_PICK(K_, S_) mbconv_front_kernel<T, K_, S_>
    if (a.k == 3) kern = a.stride == 1 ? MB_PICK(3, 1) : MB_PICK(3, 2);
    else kern = a.stride == 1 ? MB_PICK(5, 1) : MB_PICK(5, 2);
#undef MB_PICK
    if (g.lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return EFFDET_ELAUNCH;
    }
    hipLaunchKernelGGL(kern, grid, block, g.lds, st, a);
    return effdet_check_launch();
}

}  // namespace

extern "C" int effdet_mbconv_tiles_per_image(int dtype, int H, int W, int Cin, int mid, int k, int stride) {
    (void)take_pad_flag(dtype);                          // (no geometry depends on the padding convention)
    if (H <= 0 || W <= 0 || Cin <= 0 || mid <= 0 || (k != 3 && k != 5) || (stride != 1 && stride != 2) || dtype < 0 || dtype > 2) return EFFDET_EINVAL;
    const int Ho = same_out(H, stride), Wo = same_out(W, stride);
    if (dtype == 2) {                                   // two-term bf16: the two rolling-window forms only
        int parts = effdet_mbconv_roll_parts(H, W, Cin, mid, k, stride, 1);
        if (parts > 0) return parts;
        parts = effdet_mbconv_wide_parts(H, W, Cin, mid, k, stride, 1);
        return parts > 0 ? parts : EFFDET_EINVAL;
    }
    if (dtype == 1) {
#ifdef MB_PREFER_WIDE
        { const int pw_ = effdet_mbconv_wide_parts(H, W, Cin, mid, k, stride); if (pw_ > 0) return pw_; }
#endif
        int parts = effdet_mbconv_roll_parts(H, W, Cin, mid, k, stride);
        if (parts > 0) return parts;
        parts = effdet_mbconv_wide_parts(H, W, Cin, mid, k, stride);
        if (parts > 0) return parts;
    }
    const DeepGeometry dg = dtype == 0 ? pick_deep<float>(H, W, Cin, mid, k, stride) : pick_deep<bf16_t>(H, W, Cin, mid, k, stride);
    if (dg.use) return dg.nbands;
    const Geometry g = dtype == 0 ? pick_tile<float>(Ho, Wo, Cin, k, stride) : pick_tile<bf16_t>(Ho, Wo, Cin, k, stride);
    if (g.lds > 160 * 1024) return EFFDET_EINVAL;      // no fused geometry fits: the caller runs expand GEMM + depthwise
    return ((Wo + g.TW - 1) / g.TW) * ((Ho + g.TH - 1) / g.TH);
}

extern "C" int effdet_mbconv_gated_tiles_per_image(int dtype, int H, int W, int Cin, int mid, int k, int stride) {
    (void)take_pad_flag(dtype);
    if (H <= 0 || W <= 0 || Cin <= 0 || mid <= 0 || (k != 3 && k != 5) || (stride != 1 && stride != 2) || dtype < 0 || dtype > 2) return EFFDET_EINVAL;
    const int Ho = same_out(H, stride), Wo = same_out(W, stride);
    if (dtype == 2) {
        const int parts = effdet_mbconv_roll_parts(H, W, Cin, mid, k, stride, 1);
        return parts > 0 ? parts : EFFDET_EINVAL;
    }
    if (dtype == 1) {
        const int parts = effdet_mbconv_roll_parts(H, W, Cin, mid, k, stride);
        if (parts > 0) return parts;
    }
    const Geometry g = dtype == 0 ? pick_tile<float>(Ho, Wo, Cin, k, stride) : pick_tile<bf16_t>(Ho, Wo, Cin, k, stride);
    if (g.lds > 160 * 1024 || mid % SM_MC) return EFFDET_EINVAL;
    return ((Wo + g.TW - 1) / g.TW) * ((Ho + g.TH - 1) / g.TH);
}

static int mbconv_common(void* stream, int dtype, const void* X, const float* in_gate, void* Y, const void* W1,
                         const float* s1, const float* t1, const float* taps,
                         const float* s2, const float* t2, float* pool_partial,
                         int B, int H, int W, int Cin, int mid, int k, int stride) {
    if (!X || !Y || !W1 || !s1 || !t1 || !taps || !s2 || !t2 || B <= 0 || H <= 0 || W <= 0) return EFFDET_EINVAL;
    const int sym = take_pad_flag(dtype);
    if (Cin <= 0 || Cin % 8 || mid <= 0 || mid % 8 || (k != 3 && k != 5) || (stride != 1 && stride != 2) || dtype < 0 || dtype > 2) return EFFDET_EINVAL;
    if (dtype == 2) {
        // two-term bf16 (the "accurate" mode): the rolling-window forms with every operand in two terms; geometries outside them
        // (no such layer in tf_efficientdet_d0 ... d2 at their sizes) are rejected - the caller then runs expand GEMM + depthwise
        if (reinterpret_cast<uintptr_t>(X) % 16 || reinterpret_cast<uintptr_t>(Y) % 16 || reinterpret_cast<uintptr_t>(W1) % 16) return EFFDET_EINVAL;
        hipStream_t st2 = reinterpret_cast<hipStream_t>(stream);
        if (effdet_mbconv_roll_parts(H, W, Cin, mid, k, stride, 1) > 0)
            return effdet_mbconv_roll_launch(st2, X, in_gate, Y, W1, s1, t1, taps, s2, t2, pool_partial, B, H, W, Cin, mid, k, stride, 1, sym);
        if (!in_gate && effdet_mbconv_wide_parts(H, W, Cin, mid, k, stride, 1) > 0)
            return effdet_mbconv_wide_launch(st2, X, Y, W1, s1, t1, taps, s2, t2, pool_partial, B, H, W, Cin, mid, k, stride, 1, sym);
        return EFFDET_EINVAL;
    }
    MbArgs a;
    a.X = X; a.in_gate = in_gate; a.Y = Y; a.W1 = W1; a.s1 = s1; a.t1 = t1; a.taps = taps; a.s2 = s2; a.t2 = t2; a.pool_partial = pool_partial;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.mid = mid; a.k = k; a.stride = stride;
    a.Ho = same_out(H, stride); a.Wo = same_out(W, stride);
    a.pad_t = pad_before(H, k, stride, sym); a.pad_l = pad_before(W, k, stride, sym); a.sym = sym;
#ifdef EFFDET_ABLATE
    a.dbg = getenv("EFFDET_DEBUG_SKIP") ? atoi(getenv("EFFDET_DEBUG_SKIP")) : 0;
#else
    a.dbg = 0;
#endif
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    return dtype == 0 ? launch_mb<float>(st, a) : launch_mb<bf16_t>(st, a);
}

extern "C" int effdet_mbconv_expand_dw(void* stream, int dtype, const void* X, void* Y, const void* W1,
                                       const float* s1, const float* t1, const float* taps,
                                       const float* s2, const float* t2, float* pool_partial,
                                       int B, int H, int W, int Cin, int mid, int k, int stride) {
    EFFDET_ENTER();
    return mbconv_common(stream, dtype, X, nullptr, Y, W1, s1, t1, taps, s2, t2, pool_partial, B, H, W, Cin, mid, k, stride);
}

extern "C" int effdet_mbconv_expand_dw_gated(void* stream, int dtype, const void* X, const float* in_gate, void* Y, const void* W1,
                                             const float* s1, const float* t1, const float* taps,
                                             const float* s2, const float* t2, float* pool_partial,
                                             int B, int H, int W, int Cin, int mid, int k, int stride) {
    EFFDET_ENTER();
    if (!in_gate) return EFFDET_EINVAL;
    return mbconv_common(stream, dtype, X, in_gate, Y, W1, s1, t1, taps, s2, t2, pool_partial, B, H, W, Cin, mid, k, stride);
}
