// Fused network entry: conv_stem 3x3/s2 (TF-SAME) + bn1 + SiLU  ->  stage-0 depthwise 3x3/s1 + bn + SiLU
// (+ per-tile partial sums for that block's squeeze-excite average).
//
// Replaces timm's conv_stem/bn1/act1 and blocks.0.0.conv_dw/bn1/act1 (reached from
// effdet/efficientdet.py:837).  The stem output (6.5 MB/image at 640 px, written once and read ~3x by the
// stand-alone depthwise) never leaves LDS.
//
// A workgroup owns a TH x TW tile of the depthwise output:
//   phase 0  input patch [3][2TH+5][2TW+5] (NCHW source, any float dtype) -> LDS, zero outside the image
//   phase 1  stem conv as an im2col GEMM on the matrix cores: K = 27 taps (padded to 32), rows of the
//            accumulator = output channels, columns = the (TH+2) x (TW+2) halo pixels; every lane gathers its
//            8 (bf16) / 4 (f32) im2col elements of one pixel with scalar LDS reads at offsets fixed per lane.
//            BN + SiLU in registers, zero outside the stem output map (the depthwise conv pads THAT map),
//            4 consecutive channels per lane -> one LDS store.
//   phase 2  depthwise 3x3 out of LDS (4 x-adjacent outputs per thread, sliding window), BN + SiLU, NHWC stores
//   phase 3  pool partials (fixed summation order)
#include "common.h"

namespace {

struct SdArgs {
    const void* X; int in_dtype;                 // NCHW; 0 = f32, 1 = bf16, 2 = uint8 normalised on the fly
    float nmean[3], nstd[3];
    const void* Wk;                              // [Cpad][32] im2col weights (T), k = (ky*3+kx)*3+ci, zero padded
    const float* s1; const float* t1;            // stem BN fold [C]
    const float* taps;                           // depthwise [9][C]
    const float* s2; const float* t2;            // depthwise BN fold [C]
    void* Y; float* pool_partial;                // [B,Ho,Wo,C], [B][tiles][C]
    int B, H, W, C, Ho, Wo, pad_t, pad_l, tiles_x, tiles_y;
    int vec_in;                                  // bf16 input, even W and pad_l, 4-byte aligned: the patch is loaded two pixels at a time
};

constexpr int SD_TH = 16, SD_TW = 16;
constexpr int SD_HH = SD_TH + 2, SD_HW = SD_TW + 2;          // stem-output halo tile
constexpr int SD_HP = SD_HH * SD_HW;                         // 324
constexpr int SD_HPPAD = (SD_HP + 15) / 16 * 16;             // 336
constexpr int SD_IH = 2 * SD_TH + 5, SD_IW = 2 * SD_TW + 5;  // input patch 37 x 37
constexpr int SD_IWP = SD_IW + 1;                            // LDS row pitch (even: rows can be filled two elements at a time)

// IN: input dtype at compile time (0 = f32, 1 = bf16, 2 = uint8 normalised on the fly): keeps the patch gather a
// straight batch of loads
template <typename T, int IN>
__global__ __launch_bounds__(256) void stem_dw_kernel(SdArgs p) {
    constexpr int EPC = VecTraits<T>::EPC;                   // im2col elements per lane per MFMA chunk
    constexpr int KPC = 64 / (int)sizeof(T);                 // k values per chunk (32 bf16 / 16 f32)
    constexpr int NKC = 32 / KPC;                            // chunks to cover K = 32
    constexpr int WROW = 32 * (int)sizeof(T) + 16;           // weight row pitch (bytes)
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int C = p.C;
    const int cpad = (C + 15) / 16 * 16;
    const int erow = C + 16 / (int)sizeof(T);                // E row pitch in elements (+16 B: bank spread)
    T* In = reinterpret_cast<T*>(lds);                                       // [3][IH][IW] (+ zero slot)
    constexpr int IN_ELEMS = 3 * SD_IH * SD_IWP + 8;
    char* Wl = lds + ((IN_ELEMS * (int)sizeof(T) + 15) / 16) * 16;           // [cpad][WROW]
    T* E = reinterpret_cast<T*>(Wl + cpad * WROW);                           // [HP][erow]
    float* red = reinterpret_cast<float*>(reinterpret_cast<char*>(E) + ((SD_HP * erow * (int)sizeof(T) + 15) / 16) * 16);
    float* par = red + 256 * 9;                                              // [13][C] per-channel constants

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 15, fpiece = lane >> 4;
    const int b = blockIdx.y, tile = blockIdx.x;
    const int oy0 = (tile / p.tiles_x) * SD_TH, ox0 = (tile % p.tiles_x) * SD_TW;
    // stem-output coordinate of halo pixel (0,0) and input coordinate of patch element (0,0)
    const int sy0 = oy0 - 1, sx0 = ox0 - 1;
    const int iy0 = 2 * sy0 - p.pad_t, ix0 = 2 * sx0 - p.pad_l;

    // ---- phase 0: input patch.  All loads of a thread are issued before the first LDS store (a load -> store
    // loop with a runtime trip count is not pipelined by the compiler: every iteration would expose a full
    // memory round trip).
    const long long plane = (long long)p.H * p.W;
    constexpr int IN_REAL = 3 * SD_IH * SD_IWP;
    constexpr int IN_PER_THREAD = (IN_ELEMS + 255) / 256;
    if (IN == 1 && sizeof(T) == 2 && p.vec_in) {
        // bf16 -> bf16: whole dwords (two pixels of a row); ix0 is even, so a pair never straddles the image border
        constexpr int DPR = SD_IWP / 2;                      // dwords per patch row
        constexpr int NDW = 3 * SD_IH * DPR;
        constexpr int DW_PER_THREAD = (NDW + 255) / 256;
        unsigned dv[DW_PER_THREAD];
#pragma unroll
        for (int q = 0; q < DW_PER_THREAD; ++q) {
            const int i = tid + 256 * q;
            unsigned v = 0u;
            if (i < NDW) {
                const int row = i / DPR, d = i % DPR;
                const int ci = row / SD_IH, y = iy0 + row % SD_IH, x = ix0 + 2 * d;
                if (y >= 0 && y < p.H && x >= 0 && x < p.W)
                    v = *reinterpret_cast<const unsigned*>(reinterpret_cast<const bf16_t*>(p.X) + ((long long)b * 3 + ci) * plane + (long long)y * p.W + x);
            }
            dv[q] = v;
        }
#pragma unroll
        for (int q = 0; q < DW_PER_THREAD; ++q) {
            const int i = tid + 256 * q;
            if (i < NDW) reinterpret_cast<unsigned*>(In)[i] = dv[q];
        }
        if (tid < 4) reinterpret_cast<unsigned*>(In)[NDW + tid] = 0u;       // the zero slot of the k >= 27 im2col lanes
    } else {
        float vin[IN_PER_THREAD];
#pragma unroll
        for (int q = 0; q < IN_PER_THREAD; ++q) {
            const int i = tid + 256 * q;
            float v = 0.f;
            if (i < IN_REAL) {
                const int ci = i / (SD_IH * SD_IWP), rem = i % (SD_IH * SD_IWP);
                const int y = iy0 + rem / SD_IWP, x = ix0 + rem % SD_IWP;
                if (y >= 0 && y < p.H && x >= 0 && x < p.W && rem % SD_IWP < SD_IW) {
                    const long long off = ((long long)b * 3 + ci) * plane + (long long)y * p.W + x;
                    if constexpr (IN == 0) v = reinterpret_cast<const float*>(p.X)[off];
                    else if constexpr (IN == 1) v = (float)reinterpret_cast<const bf16_t*>(p.X)[off];
                    else {
                        const float m = ci == 0 ? p.nmean[0] : ci == 1 ? p.nmean[1] : p.nmean[2];
                        const float sd = ci == 0 ? p.nstd[0] : ci == 1 ? p.nstd[1] : p.nstd[2];
                        v = ((float)reinterpret_cast<const unsigned char*>(p.X)[off] - m) / sd;
                    }
                }
            }
            vin[q] = v;
        }
#pragma unroll
        for (int q = 0; q < IN_PER_THREAD; ++q) {
            const int i = tid + 256 * q;
            if (i < IN_ELEMS) In[i] = from_f<T>(vin[q]);
        }
    }
    for (int i = tid; i < cpad * (WROW / 16); i += 256) {
        const int row = i / (WROW / 16), piece = i % (WROW / 16);
        u32x4 v = {0u, 0u, 0u, 0u};
        if (row < C && piece * 16 < 32 * (int)sizeof(T))
            v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.Wk) + row * 32 * sizeof(T) + piece * 16);
        *reinterpret_cast<u32x4*>(Wl + row * WROW + piece * 16) = v;
    }
    // per-channel constants of both convs -> LDS once: taps [9][C], s1, t1, s2, t2 (13 * C <= 832 values:
    // up to four per thread, all loaded before the first LDS store)
    {
        float pv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid + 256 * q;
            float v = 0.f;
            if (i < 13 * C) {
                if (i < 9 * C) v = p.taps[i];
                else if (i < 10 * C) v = p.s1[i - 9 * C];
                else if (i < 11 * C) v = p.t1[i - 10 * C];
                else if (i < 12 * C) v = p.s2[i - 11 * C];
                else v = p.t2[i - 12 * C];
            }
            pv[q] = v;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid + 256 * q;
            if (i < 13 * C) par[i] = pv[q];
        }
    }
    // per-lane im2col offsets (relative to the pixel's top-left input element); k >= 27 -> the zero slot
    int koff[NKC][EPC];
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc)
#pragma unroll
        for (int j = 0; j < EPC; ++j) {
            const int k = kc * KPC + fpiece * EPC + j;
            const int tap = k / 3, ci = k % 3;
            koff[kc][j] = k < 27 ? ci * SD_IH * SD_IWP + (tap / 3) * SD_IWP + (tap % 3) : -1;
        }
    __syncthreads();

    // ---- phase 1: stem conv on the halo tile
    const int n_ct = cpad / 16;
    for (int ms = wave; ms < SD_HPPAD / 16; ms += 4) {
        const int hp = 16 * ms + frow;
        const int hy = hp / SD_HW, hx = hp % SD_HW;
        const int base = (hp < SD_HP) ? (2 * hy) * SD_IWP + 2 * hx : 0;
        Frag<T> xf[NKC];
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc)
#pragma unroll
            for (int j = 0; j < EPC; ++j)
                xf[kc].v[j] = In[koff[kc][j] >= 0 ? base + koff[kc][j] : 3 * SD_IH * SD_IWP];
        const int sy = sy0 + hy, sx = sx0 + hx;
        const bool inside = hp < SD_HP && sy >= 0 && sy < p.Ho && sx >= 0 && sx < p.Wo;
        const float inside_m = inside ? 1.f : 0.f;
        for (int j = 0; j < n_ct; ++j) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kc = 0; kc < NKC; ++kc) {
                const Frag<T> wf = ld_frag<T>(Wl + (16 * j + frow) * WROW + kc * 64 + fpiece * 16);
                mma_chunk(wf, xf[kc], acc);
            }
            const int ch = 16 * j + 4 * fpiece;
            if (hp < SD_HP && ch < C) {
                const f32x4 sc = *reinterpret_cast<const f32x4*>(par + 9 * C + ch);
                const f32x4 sh = *reinterpret_cast<const f32x4*>(par + 10 * C + ch);
                const f32x4 v = bn_silu4<T>(acc, sc, sh) * inside_m;                                // a mask, not a branch per element
                if constexpr (sizeof(T) == 2) {
                    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                    *reinterpret_cast<bf16x4*>(E + hp * erow + ch) = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                } else {
                    *reinterpret_cast<f32x4*>(E + hp * erow + ch) = f32x4{v[0], v[1], v[2], v[3]};
                }
            }
        }
    }
    __syncthreads();

    // ---- phase 2: depthwise 3x3 (stride 1) out of LDS
    const int cgn = C / 8;
    const int PG = 256 / cgn;                                // pixel-group threads; thread t < cgn*PG works
    T* Y = reinterpret_cast<T*>(p.Y) + (long long)b * p.Ho * p.Wo * C;
    F8 pool = f8_zero();
    const int cg = tid % cgn, pg0 = tid / cgn;
    if (pg0 < PG) {
        const F8 s2 = load8<float>(par + 11 * C + cg * 8), t2 = load8<float>(par + 12 * C + cg * 8);
        constexpr int GPRW = SD_TW / 4;
        for (int pg = pg0; pg < SD_TH * GPRW; pg += PG) {
            const int ty = pg / GPRW, tx0 = (pg % GPRW) * 4;
            const int oy = oy0 + ty;
            if (oy >= p.Ho || ox0 + tx0 >= p.Wo) continue;
            F8 acc[4];
#pragma unroll
            for (int pi = 0; pi < 4; ++pi) acc[pi] = f8_zero();
#pragma unroll 1
            for (int ky = 0; ky < 3; ++ky) {
                F8 w[3];
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) w[kx] = load8<float>(par + (ky * 3 + kx) * C + cg * 8);
                const T* er = E + ((ty + ky) * SD_HW + tx0) * erow + cg * 8;
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    const F8 e = load8<T>(er + c * erow);
#pragma unroll
                    for (int pi = 0; pi < 4; ++pi) {
                        const int kx = c - pi;
                        if (kx >= 0 && kx < 3) {
#pragma unroll
                            for (int q = 0; q < 8; ++q) acc[pi].v[q] = fmaf(e.v[q], w[kx].v[q], acc[pi].v[q]);
                        }
                    }
                }
            }
#pragma unroll
            for (int pi = 0; pi < 4; ++pi) {
                const int ox = ox0 + tx0 + pi;
                if (ox >= p.Wo) continue;
                F8 o;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float v = silu_t<T>(acc[pi].v[q] * s2.v[q] + t2.v[q]);
                    o.v[q] = to_f<T>(from_f<T>(v));
                    pool.v[q] += o.v[q];
                }
                store8<T>(Y + ((long long)oy * p.Wo + ox) * C + cg * 8, o);
            }
        }
    }
    if (p.pool_partial != nullptr) {
        const float tot = pool_reduce<256>(pool, red, red + 256 * 8, tid, cgn, cgn * PG);
        if (tid < C) p.pool_partial[((long long)b * (p.tiles_x * p.tiles_y) + tile) * C + tid] = tot;
    }
}

template <typename T>
size_t sd_lds_bytes(int C) {
    const int cpad = (C + 15) / 16 * 16;
    const int erow = C + 16 / (int)sizeof(T);
    const int WROW = 32 * (int)sizeof(T) + 16;
    size_t n = ((3 * SD_IH * SD_IWP + 8) * sizeof(T) + 15) / 16 * 16;
    n += (size_t)cpad * WROW;
    n += ((size_t)SD_HP * erow * sizeof(T) + 15) / 16 * 16;
    n += 256 * 9 * 4;
    n += (size_t)13 * C * 4;
    return n;
}

}  // namespace

// SE pool partial rows per image of the kernel effdet_stem_dw_fused[_u8] will run for (dtype, H, W, C)
extern "C" int effdet_stem_dw_parts(int dtype, int H, int W, int C) {
    const int sym = take_pad_flag(dtype);
    if (H <= 0 || W <= 0 || C <= 0 || dtype < 0 || dtype > 2) return EFFDET_EINVAL;
    if (dtype == 2) {                                                // two-term bf16: the rolling-window form only
        const int parts = effdet_stem_roll_parts(H, W, C, 1, sym);
        return parts > 0 ? parts : EFFDET_EINVAL;
    }
    if (dtype == 1) {
        const int parts = effdet_stem_roll_parts(H, W, C, 0, sym);
        if (parts > 0) return parts;
    }
    const int Ho = same_out(H, 2), Wo = same_out(W, 2);
    return ((Ho + SD_TH - 1) / SD_TH) * ((Wo + SD_TW - 1) / SD_TW);
}

extern "C" int effdet_stem_dw_tiles_per_image(int H, int W) {
    if (H <= 0 || W <= 0) return EFFDET_EINVAL;
    const int Ho = same_out(H, 2), Wo = same_out(W, 2);
    return ((Ho + SD_TH - 1) / SD_TH) * ((Wo + SD_TW - 1) / SD_TW);
}

static int stem_dw_common(void* stream, int in_dtype, int dtype, const void* X, const float* mean, const float* stdv, const void* Wk,
                          const float* s1, const float* t1, const float* taps,
                          const float* s2, const float* t2, void* Y, float* pool_partial,
                          int B, int H, int W, int C) {
    if (!X || !Wk || !s1 || !t1 || !taps || !s2 || !t2 || !Y || B <= 0 || H <= 0 || W <= 0) return EFFDET_EINVAL;
    const int sym = take_pad_flag(dtype);
    if (C <= 0 || C % 8 || C > 64 || in_dtype < 0 || in_dtype > 2 || dtype < 0 || dtype > 2 || (in_dtype == 2 && (!mean || !stdv))) return EFFDET_EINVAL;
    if (dtype == 2) {
        // two-term bf16 (the "accurate" mode): Wk is float32 [C][32], Y two-term; float32 or uint8 images; C = 32 and an even width
        if (effdet_stem_roll_parts(H, W, C, 1, sym) <= 0 || reinterpret_cast<uintptr_t>(Y) % 16) return EFFDET_EINVAL;
        return effdet_stem_roll_launch(reinterpret_cast<hipStream_t>(stream), in_dtype, X, mean, stdv, Wk, s1, t1, taps, s2, t2, Y, pool_partial, B, H, W, C, 1, sym);
    }
    SdArgs a;
    for (int i = 0; i < 3; ++i) { a.nmean[i] = in_dtype == 2 ? mean[i] : 0.f; a.nstd[i] = in_dtype == 2 ? stdv[i] : 1.f; }
    a.X = X; a.in_dtype = in_dtype; a.Wk = Wk; a.s1 = s1; a.t1 = t1; a.taps = taps; a.s2 = s2; a.t2 = t2;
    a.Y = Y; a.pool_partial = pool_partial; a.B = B; a.H = H; a.W = W; a.C = C;
    a.Ho = same_out(H, 2); a.Wo = same_out(W, 2);
    a.pad_t = pad_before(H, 3, 2, sym); a.pad_l = pad_before(W, 3, 2, sym);
    a.tiles_x = (a.Wo + SD_TW - 1) / SD_TW; a.tiles_y = (a.Ho + SD_TH - 1) / SD_TH;
    a.vec_in = in_dtype == 1 && dtype == 1 && W % 2 == 0 && a.pad_l % 2 == 0 && reinterpret_cast<uintptr_t>(X) % 4 == 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == 1 && effdet_stem_roll_parts(H, W, C, 0, sym) > 0)          // bf16: the rolling-window form (stem_roll.hip) where it applies
        return effdet_stem_roll_launch(st, in_dtype, X, mean, stdv, Wk, s1, t1, taps, s2, t2, Y, pool_partial, B, H, W, C, 0, sym);
    dim3 grid(a.tiles_x * a.tiles_y, B), block(256);
    const size_t lds = dtype == 0 ? sd_lds_bytes<float>(C) : sd_lds_bytes<bf16_t>(C);
    void (*kern)(SdArgs) = nullptr;
    if (dtype == 0) kern = in_dtype == 0 ? stem_dw_kernel<float, 0> : in_dtype == 1 ? stem_dw_kernel<float, 1> : stem_dw_kernel<float, 2>;
    else kern = in_dtype == 0 ? stem_dw_kernel<bf16_t, 0> : in_dtype == 1 ? stem_dw_kernel<bf16_t, 1> : stem_dw_kernel<bf16_t, 2>;
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return EFFDET_ELAUNCH;
    }
    hipLaunchKernelGGL(kern, grid, block, lds, st, a);
    return effdet_check_launch();
}

extern "C" int effdet_stem_dw_fused(void* stream, int in_dtype, int dtype, const void* X, const void* Wk,
                                    const float* s1, const float* t1, const float* taps,
                                    const float* s2, const float* t2, void* Y, float* pool_partial,
                                    int B, int H, int W, int C) {
    EFFDET_ENTER();
    if (in_dtype & ~1) return EFFDET_EINVAL;
    return stem_dw_common(stream, in_dtype, dtype, X, nullptr, nullptr, Wk, s1, t1, taps, s2, t2, Y, pool_partial, B, H, W, C);
}

extern "C" int effdet_stem_dw_fused_u8(void* stream, int dtype, const unsigned char* X, const float* mean, const float* stdv,
                                       const void* Wk, const float* s1, const float* t1, const float* taps,
                                       const float* s2, const float* t2, void* Y, float* pool_partial,
                                       int B, int H, int W, int C) {
    EFFDET_ENTER();
    return stem_dw_common(stream, 2, dtype, X, mean, stdv, Wk, s1, t1, taps, s2, t2, Y, pool_partial, B, H, W, C);
}
