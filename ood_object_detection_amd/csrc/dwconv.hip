// Backbone spatial kernels (NHWC):
//   effdet_stem_conv      3x3 stride-2 full conv 3 -> Cout (TF-SAME), folded BN + SiLU, NCHW in -> NHWC out
//   effdet_dwconv_bn_act  depthwise k x k (k = 3 | 5, stride 1 | 2, TF-SAME), folded BN + SiLU, and the
//                         per-block partial sums of the squeeze-excite global average pool
//   effdet_se_gate        SE: finish the average, fc-reduce -> SiLU -> fc-expand -> sigmoid
//   effdet_maxpool_same   3x3 stride-2 max pool with TF-SAME (-inf) padding (BiFPN P6 / P7)
//
// These replace the timm EfficientNet conv_stem / conv_dw / SqueezeExcite modules and
// `create_pool2d('max', 3, 2, 'same')` reached from effdet/efficientdet.py:837 and :165-166.
// timm is absent from the reference tree; semantics are restated in DESIGN.md.
//
// Mapping: one thread owns 8 consecutive channels (one 16-byte bf16 piece) of an output pixel;
// consecutive threads walk the channel groups of a pixel first, then pixels along x, so every
// wave-instruction reads and writes contiguous NHWC bytes.  Overlapping input windows of
// neighbouring pixels are served by the CU's L1.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------ stem
struct StemArgs {
    const void* X; int in_dtype;            // NCHW, 0 = f32, 1 = bf16, 2 = uint8 (normalised on the fly)
    float nmean[3], nstd[3];                // uint8 input: y = (x - mean) / std, then rounded to the model dtype
    const float* Wt;                        // [27][Cout] tap-major (ky, kx, ci)
    const float* scale; const float* shift;
    void* Y;                                // NHWC [B, Ho, Wo, Cout]
    int B, H, W, Cout, Ho, Wo, pad_t, pad_l;
};

template <typename T>
__global__ __launch_bounds__(256) void stem_kernel(StemArgs p) {
    extern __shared__ __attribute__((aligned(16))) float wl[];   // 27 * Cout weights + 2 * Cout affine
    const int C = p.Cout;
    for (int i = threadIdx.x; i < 27 * C; i += blockDim.x) wl[i] = p.Wt[i];
    for (int i = threadIdx.x; i < C; i += blockDim.x) { wl[27 * C + i] = p.scale[i]; wl[28 * C + i] = p.shift[i]; }
    __syncthreads();
    const int CG = C / 8;
    const long long total = (long long)p.B * p.Ho * p.Wo * CG;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int cg = (int)(idx % CG);
    long long pix = idx / CG;
    const int ox = (int)(pix % p.Wo); pix /= p.Wo;
    const int oy = (int)(pix % p.Ho);
    const int b = (int)(pix / p.Ho);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    const long long plane = (long long)p.H * p.W;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy * 2 + ky - p.pad_t;
        if (iy < 0 || iy >= p.H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox * 2 + kx - p.pad_l;
            if (ix < 0 || ix >= p.W) continue;
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) {
                const long long off = ((long long)b * 3 + ci) * plane + (long long)iy * p.W + ix;
                float x;
                if (p.in_dtype == 0) x = reinterpret_cast<const float*>(p.X)[off];
                else if (p.in_dtype == 1) x = (float)reinterpret_cast<const bf16_t*>(p.X)[off];
                else {
                    x = ((float)reinterpret_cast<const unsigned char*>(p.X)[off] - p.nmean[ci]) / p.nstd[ci];
                    if constexpr (!IsPair<T>::value) x = to_f<T>(from_f<T>(x));       // (two-term mode: the float32 value is the input)
                }
                const float* w = wl + ((ky * 3 + kx) * 3 + ci) * C + cg * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = fmaf(x, w[e], acc[e]);
            }
        }
    }
    F8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o.v[e] = silu_t<T>(acc[e] * wl[27 * C + cg * 8 + e] + wl[28 * C + cg * 8 + e]);
    T* dst = reinterpret_cast<T*>(p.Y) + (((long long)b * p.Ho + oy) * p.Wo + ox) * C + cg * 8;
    store8<T>(dst, o);
}

// ---------------------------------------------------------------------------------------- dwconv
struct DwArgs {
    const void* X; void* Y;
    const float* Wt;                        // [k*k][C] tap-major, fp32
    const float* scale; const float* shift;
    float* pool_partial;                    // [B, blocks_per_image, C] or null
    int B, H, W, C, Ho, Wo, k, stride, pad_t, pad_l, act;
    int CG, PT, pix_per_block, blocks_per_image;   // CG: channel groups handled by one workgroup (grid.z slices)
};

template <typename T, int KS>
__global__ __launch_bounds__(256) void dwconv_kernel(DwArgs p) {
    extern __shared__ __attribute__((aligned(16))) float red[];   // [PT][C] pool staging
    const int tid = threadIdx.x;
    const int cg = tid % p.CG;
    const int pt = tid / p.CG;
    const int b = blockIdx.y;
    const int cbase = blockIdx.z * p.CG * 8;             // first channel of this slice
    const int npix = p.Ho * p.Wo;
    const int pix0 = blockIdx.x * p.pix_per_block;
    const int pix1 = min(pix0 + p.pix_per_block, npix);
    const int c0 = cbase + cg * 8;
    const T* X = reinterpret_cast<const T*>(p.X) + (long long)b * p.H * p.W * p.C;
    T* Y = reinterpret_cast<T*>(p.Y) + (long long)b * npix * p.C;

    F8 sc = load8<float>(p.scale + c0), sh = load8<float>(p.shift + c0);
    F8 pool = f8_zero();
    for (int pix = pix0 + pt; pix < pix1; pix += p.PT) {
        const int oy = pix / p.Wo, ox = pix % p.Wo;
        F8 acc = f8_zero();
        const int iy0 = oy * p.stride - p.pad_t, ix0 = ox * p.stride - p.pad_l;
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
            const int iy = iy0 + ky;
            if (iy < 0 || iy >= p.H) continue;
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                const int ix = ix0 + kx;
                if (ix < 0 || ix >= p.W) continue;
                const F8 x = load8<T>(X + ((long long)iy * p.W + ix) * p.C + c0);
                const F8 w = load8<float>(p.Wt + (ky * KS + kx) * p.C + c0);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc.v[e] = fmaf(x.v[e], w.v[e], acc.v[e]);
            }
        }
        F8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = acc.v[e] * sc.v[e] + sh.v[e];
            if (p.act == 1) v = silu_t<T>(v);
            // the SE average is taken over what the next layer will read: the T-rounded value (two-term: 16 bits - below the tolerance)
            if constexpr (IsPair<T>::value) o.v[e] = v; else o.v[e] = to_f<T>(from_f<T>(v));
            pool.v[e] += o.v[e];
        }
        store8<T>(Y + (long long)pix * p.C + c0, o);
    }
    if (p.pool_partial != nullptr) {
        const int CS = p.CG * 8;                          // channels of this slice
        store8<float>(red + pt * CS + cg * 8, pool);
        __syncthreads();
        for (int c = tid; c < CS; c += blockDim.x) {
            float s = 0.f;
            for (int q = 0; q < p.PT; ++q) s += red[q * CS + c];
            p.pool_partial[((long long)b * p.blocks_per_image + blockIdx.x) * p.C + cbase + c] = s;
        }
    }
}

// --------------------------------------------------------------------------------------- SE gate
struct SeArgs {
    const float* partial; int nblk; float inv_hw;
    const float* W1; const float* b1;       // [R][C], [R]
    const float* W2t; const float* b2;      // [R][C] (conv_expand weight transposed: coalesced over channels), [C]
    float* gate;                            // [B, C]
    int C, R, S;                            // S: slices that split the partial-sum rows
    float* pool_out;                        // optional [B, C]: the pooled SUMS (the training backward reads them)
};

// One workgroup per image.  Every step is a short dependent chain, so the kernel is all latency: loads are
// issued in batches of 8 before they are consumed and each thread sums only ~nblk / S partial rows.
__global__ __launch_bounds__(1024) void se_gate_kernel(SeArgs p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];    // pooled[C] + r[R] + slice sums [S][C]
    float* pooled = sm;
    float* red = sm + p.C;
    float* wsum = red + p.R;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = p.C, S = p.S;
    // partial sums: slice sl takes rows sl, sl+S, ... (fixed order -> bitwise reproducible)
    for (int idx = tid; idx < C * S; idx += 1024) {
        const int sl = idx / C, c = idx - sl * C;
        const float* src = p.partial + (long long)b * p.nblk * C + c;
        float s = 0.f;
        int q = sl;
        for (; q + 7 * S < p.nblk; q += 8 * S) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(long long)(q + u * S) * C];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; q < p.nblk; q += S) s += src[(long long)q * C];
        wsum[idx] = s;
    }
    __syncthreads();
    for (int c = tid; c < C; c += 1024) {
        float s = 0.f;
        for (int w = 0; w < S; ++w) s += wsum[w * C + c];
        pooled[c] = s * p.inv_hw;
        if (p.pool_out) p.pool_out[(long long)b * C + c] = s;
    }
    __syncthreads();
    for (int j = wave; j < p.R; j += 16) {
        const float* w1 = p.W1 + (long long)j * C;
        float s = 0.f;
        int c = lane;
        for (; c + 3 * 64 < C; c += 4 * 64) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = w1[c + 64 * u];
#pragma unroll
            for (int u = 0; u < 4; ++u) s = fmaf(v[u], pooled[c + 64 * u], s);
        }
        for (; c < C; c += 64) s = fmaf(w1[c], pooled[c], s);
        s = wave_reduce_sum(s);
        if (lane == 0) red[j] = silu_f(s + p.b1[j]);
    }
    __syncthreads();
    for (int c = tid; c < C; c += 1024) {
        float s = p.b2[c];
        int j = 0;
        for (; j + 7 < p.R; j += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p.W2t[(long long)(j + u) * C + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) s = fmaf(v[u], red[j + u], s);
        }
        for (; j < p.R; ++j) s = fmaf(p.W2t[(long long)j * C + c], red[j], s);
        p.gate[(long long)b * C + c] = sigmoid_f(s);
    }
}

// --------------------------------------------------------------------------------------- maxpool
struct PoolArgs {
    const void* X; void* Y;
    long long x_image_stride, y_image_stride;     // elements
    int B, H, W, C, Ho, Wo, pad_t, pad_l;
};

template <typename T>
__global__ __launch_bounds__(256) void maxpool_kernel(PoolArgs p) {
    const int CG = p.C / 8;
    const long long total = (long long)p.B * p.Ho * p.Wo * CG;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int cg = (int)(idx % CG);
    long long pix = idx / CG;
    const int ox = (int)(pix % p.Wo); pix /= p.Wo;
    const int oy = (int)(pix % p.Ho);
    const int b = (int)(pix / p.Ho);
    const T* X = reinterpret_cast<const T*>(p.X) + (long long)b * p.x_image_stride;
    F8 m = f8_fill(-INFINITY);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy * 2 + ky - p.pad_t;
        if (iy < 0 || iy >= p.H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox * 2 + kx - p.pad_l;
            if (ix < 0 || ix >= p.W) continue;
            const F8 x = load8<T>(X + ((long long)iy * p.W + ix) * p.C + cg * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) m.v[e] = fmaxf(m.v[e], x.v[e]);
        }
    }
    T* dst = reinterpret_cast<T*>(p.Y) + (long long)b * p.y_image_stride + ((long long)oy * p.Wo + ox) * p.C + cg * 8;
    store8<T>(dst, m);
}

}  // namespace

static int stem_conv_common(void* stream, int in_dtype, int out_dtype, const void* X, const float* mean, const float* stdv,
                            const float* Wt, const float* scale, const float* shift, void* Y, int B, int H, int W, int Cout) {
    const int sym = take_pad_flag(out_dtype);
    if (!X || !Wt || !scale || !shift || !Y || B <= 0 || H <= 0 || W <= 0 || Cout <= 0 || Cout % 8) return EFFDET_EINVAL;
    if (in_dtype < 0 || in_dtype > 2 || out_dtype < 0 || out_dtype > 2 || (in_dtype == 2 && (!mean || !stdv))) return EFFDET_EINVAL;
    if (out_dtype == 2 && (in_dtype == 1 || reinterpret_cast<uintptr_t>(Y) % 16)) return EFFDET_EINVAL;   // two-term: float32 / uint8 images
    StemArgs a{X, in_dtype, {0.f, 0.f, 0.f}, {1.f, 1.f, 1.f}, Wt, scale, shift, Y, B, H, W, Cout, same_out(H, 2), same_out(W, 2),
               pad_before(H, 3, 2, sym), pad_before(W, 3, 2, sym)};
    if (in_dtype == 2) for (int i = 0; i < 3; ++i) { a.nmean[i] = mean[i]; a.nstd[i] = stdv[i]; }
    const long long total = (long long)B * a.Ho * a.Wo * (Cout / 8);
    const long long blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffLL) return EFFDET_EINVAL;
    const size_t sh = (size_t)29 * Cout * sizeof(float);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (out_dtype == 0) hipLaunchKernelGGL(stem_kernel<float>, dim3((unsigned)blocks), dim3(256), sh, st, a);
    else if (out_dtype == 1) hipLaunchKernelGGL(stem_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), sh, st, a);
    else hipLaunchKernelGGL(stem_kernel<bf16p_t>, dim3((unsigned)blocks), dim3(256), sh, st, a);
    return effdet_check_launch();
}

extern "C" int effdet_stem_conv(void* stream, int in_dtype, int out_dtype,
                                const void* X, const float* Wt, const float* scale, const float* shift,
                                void* Y, int B, int H, int W, int Cout) {
    EFFDET_ENTER();
    if (in_dtype & ~1) return EFFDET_EINVAL;
    return stem_conv_common(stream, in_dtype, out_dtype, X, nullptr, nullptr, Wt, scale, shift, Y, B, H, W, Cout);
}

extern "C" int effdet_stem_conv_u8(void* stream, int out_dtype, const unsigned char* X, const float* mean, const float* stdv,
                                   const float* Wt, const float* scale, const float* shift,
                                   void* Y, int B, int H, int W, int Cout) {
    EFFDET_ENTER();
    return stem_conv_common(stream, 2, out_dtype, X, mean, stdv, Wt, scale, shift, Y, B, H, W, Cout);
}

namespace {
struct NormArgs { const unsigned char* X; void* Y; float mean[4], stdv[4]; int C; long long hw, total; };

// PrefetchLoader normalisation: 16 pixels of one plane per thread (one 16-byte load)
template <typename T>
__global__ __launch_bounds__(256) void normalize_u8_kernel(NormArgs p) {
    const long long i0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 16;
    if (i0 >= p.total) return;
    const int c = (int)((i0 / p.hw) % p.C);                   // hw % 16 == 0 on this path: a piece never straddles planes
    const float m = p.mean[c], s = p.stdv[c];
    const u32x4 raw = *reinterpret_cast<const u32x4*>(p.X + i0);
    T* dst = reinterpret_cast<T*>(p.Y) + i0;
    F8 o[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const float x = (float)((raw[e >> 2] >> (8 * (e & 3))) & 0xFFu);
        o[e >> 3].v[e & 7] = (x - m) / s;
    }
    store8<T>(dst, o[0]);
    store8<T>(dst + 8, o[1]);
}

template <typename T>
__global__ __launch_bounds__(256) void normalize_u8_scalar_kernel(NormArgs p) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= p.total) return;
    const int c = (int)((i / p.hw) % p.C);
    reinterpret_cast<T*>(p.Y)[i] = from_f<T>(((float)p.X[i] - p.mean[c]) / p.stdv[c]);
}
}  // namespace

extern "C" int effdet_normalize_u8(void* stream, int out_dtype, const unsigned char* X, const float* mean, const float* stdv,
                                   void* Y, int B, int C, long long hw) {
    EFFDET_ENTER();
    if (!X || !Y || !mean || !stdv || B <= 0 || C <= 0 || C > 4 || hw <= 0 || (out_dtype & ~1)) return EFFDET_EINVAL;
    NormArgs a; a.X = X; a.Y = Y; a.C = C; a.hw = hw; a.total = (long long)B * C * hw;
    for (int i = 0; i < C; ++i) { if (stdv[i] == 0.f) return EFFDET_EINVAL; a.mean[i] = mean[i]; a.stdv[i] = stdv[i]; }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const bool vec = hw % 16 == 0 && reinterpret_cast<uintptr_t>(X) % 16 == 0 && reinterpret_cast<uintptr_t>(Y) % 16 == 0;
    const long long items = vec ? a.total / 16 : a.total;
    const long long blocks = (items + 255) / 256;
    if (blocks > 0x7fffffffLL) return EFFDET_EINVAL;
    if (vec) {
        if (out_dtype == 0) hipLaunchKernelGGL(normalize_u8_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(normalize_u8_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, st, a);
    } else {
        if (out_dtype == 0) hipLaunchKernelGGL(normalize_u8_scalar_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(normalize_u8_scalar_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, st, a);
    }
    return effdet_check_launch();
}

// Geometry helper shared with the host: how many partial-sum blocks per image the depthwise kernel
// uses for an output of Ho x Wo pixels and C channels.
// channel slices (grid.z) so that one workgroup handles at most 256 groups of 8 channels
static inline int dw_slices(int C) {
    const int cg = C / 8;
    for (int nz = 1; nz <= cg; ++nz) if (cg % nz == 0 && cg / nz <= 256) return nz;
    return 1;
}

extern "C" int effdet_dwconv_blocks_per_image(int Ho, int Wo, int C) {
    if (Ho <= 0 || Wo <= 0 || C <= 0 || C % 8) return EFFDET_EINVAL;
    const int CG = C / 8 / dw_slices(C), PT = 256 / CG;
    const int npix = Ho * Wo;
    int ppb = PT * 8;                                     // 8 pixels per thread
    if (ppb > npix) ppb = ((npix + PT - 1) / PT) * PT;
    return (npix + ppb - 1) / ppb;
}

static int launch_dwconv(void* stream, int dtype, const void* X, void* Y, const float* Wt,
                         const float* scale, const float* shift, int act, float* pool_partial,
                         int B, int H, int W, int C, int k, int stride) {
    const int sym = take_pad_flag(dtype);
    if (!X || !Y || !Wt || !scale || !shift || B <= 0 || H <= 0 || W <= 0) return EFFDET_EINVAL;
    if (C <= 0 || C % 8 || (k != 3 && k != 5) || (stride != 1 && stride != 2)) return EFFDET_EINVAL;
    if (dtype < 0 || dtype > 2 || (act & ~1)) return EFFDET_EINVAL;
    if (dtype == 2 && (reinterpret_cast<uintptr_t>(X) % 16 || reinterpret_cast<uintptr_t>(Y) % 16)) return EFFDET_EINVAL;
    DwArgs a;
    a.X = X; a.Y = Y; a.Wt = Wt; a.scale = scale; a.shift = shift; a.pool_partial = pool_partial;
    a.B = B; a.H = H; a.W = W; a.C = C; a.k = k; a.stride = stride; a.act = act;
    a.Ho = same_out(H, stride); a.Wo = same_out(W, stride);
    a.pad_t = pad_before(H, k, stride, sym); a.pad_l = pad_before(W, k, stride, sym);
    const int nz = dw_slices(C);
    a.CG = C / 8 / nz; a.PT = 256 / a.CG;
    const int npix = a.Ho * a.Wo;
    int ppb = a.PT * 8;
    if (ppb > npix) ppb = ((npix + a.PT - 1) / a.PT) * a.PT;
    a.pix_per_block = ppb;
    a.blocks_per_image = (npix + ppb - 1) / ppb;
    dim3 grid(a.blocks_per_image, B, nz), block(a.CG * a.PT);
    const size_t sh = pool_partial ? (size_t)a.PT * a.CG * 8 * sizeof(float) : 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == 0) {
        if (k == 3) hipLaunchKernelGGL((dwconv_kernel<float, 3>), grid, block, sh, st, a);
        else hipLaunchKernelGGL((dwconv_kernel<float, 5>), grid, block, sh, st, a);
    } else if (dtype == 1) {
        if (k == 3) hipLaunchKernelGGL((dwconv_kernel<bf16_t, 3>), grid, block, sh, st, a);
        else hipLaunchKernelGGL((dwconv_kernel<bf16_t, 5>), grid, block, sh, st, a);
    } else {                                     // two-term bf16: the unfused fallback of the accurate mode (wide / odd geometries)
        if (k == 3) hipLaunchKernelGGL((dwconv_kernel<bf16p_t, 3>), grid, block, sh, st, a);
        else hipLaunchKernelGGL((dwconv_kernel<bf16p_t, 5>), grid, block, sh, st, a);
    }
    return effdet_check_launch();
}

extern "C" int effdet_dwconv_bn_act(void* stream, int dtype, const void* X, void* Y, const float* Wt,
                                    const float* scale, const float* shift, int act,
                                    float* pool_partial,
                                    int B, int H, int W, int C, int k, int stride) {
    EFFDET_ENTER();
    return launch_dwconv(stream, dtype, X, Y, Wt, scale, shift, act, pool_partial, B, H, W, C, k, stride);
}


static int launch_se_gate(void* stream, const float* partial, int nblk, int hw,
                          const float* W1, const float* b1, const float* W2t, const float* b2,
                          float* gate, float* pool_out, int B, int C, int R) {
    if (!partial || !W1 || !b1 || !W2t || !b2 || !gate || nblk <= 0 || hw <= 0 || B <= 0 || C <= 0 || R <= 0) return EFFDET_EINVAL;
    int S = 1024 / C; if (S < 1) S = 1; if (S > 16) S = 16; if (S > nblk) S = nblk;
    while (S > 1 && (size_t)((S + 1) * C + R) * sizeof(float) > 64 * 1024) --S;
    SeArgs a{partial, nblk, 1.0f / (float)hw, W1, b1, W2t, b2, gate, C, R, S, pool_out};
    const size_t sh = (size_t)((S + 1) * C + R) * sizeof(float);
    if (sh > 64 * 1024) return EFFDET_EINVAL;
    hipLaunchKernelGGL(se_gate_kernel, dim3(B), dim3(1024), sh, reinterpret_cast<hipStream_t>(stream), a);
    return effdet_check_launch();
}

extern "C" int effdet_se_gate(void* stream, const float* partial, int nblk, int hw,
                              const float* W1, const float* b1, const float* W2t, const float* b2,
                              float* gate, int B, int C, int R) {
    EFFDET_ENTER();
    return launch_se_gate(stream, partial, nblk, hw, W1, b1, W2t, b2, gate, nullptr, B, C, R);
}

// the same gate, also writing the pooled sums [B][C] that effdet_train_se_bwd reads
extern "C" int effdet_train_se_gate(void* stream, const float* partial, int nblk, int hw, const float* W1, const float* b1,
                                    const float* W2t, const float* b2, float* gate, float* pool_sum, int B, int C, int R) {
    EFFDET_ENTER();
    if (!pool_sum) return EFFDET_EINVAL;
    return launch_se_gate(stream, partial, nblk, hw, W1, b1, W2t, b2, gate, pool_sum, B, C, R);
}

extern "C" int effdet_maxpool_same(void* stream, int dtype, const void* X, long long x_image_stride,
                                   void* Y, long long y_image_stride, int B, int H, int W, int C) {
    EFFDET_ENTER();
    const int sym = take_pad_flag(dtype);
    if (!X || !Y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 || dtype < 0 || dtype > 2) return EFFDET_EINVAL;
    if (dtype == 2 && (reinterpret_cast<uintptr_t>(X) % 16 || reinterpret_cast<uintptr_t>(Y) % 16 || x_image_stride % 4 || y_image_stride % 4)) return EFFDET_EINVAL;
    PoolArgs a{X, Y, x_image_stride, y_image_stride, B, H, W, C, same_out(H, 2), same_out(W, 2),
               pad_before(H, 3, 2, sym), pad_before(W, 3, 2, sym)};
    if (a.x_image_stride <= 0) a.x_image_stride = (long long)H * W * C;
    if (a.y_image_stride <= 0) a.y_image_stride = (long long)a.Ho * a.Wo * C;
    const long long total = (long long)B * a.Ho * a.Wo * (C / 8);
    const long long blocks = (total + 255) / 256;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == 0) hipLaunchKernelGGL(maxpool_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, a);
    else if (dtype == 1) hipLaunchKernelGGL(maxpool_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(maxpool_kernel<bf16p_t>, dim3((unsigned)blocks), dim3(256), 0, st, a);   // two-term bf16: the maximum of stored values is a stored value
    return effdet_check_launch();
}

// ------------------------------------------------------------------------------------------------
// ResizePad (effdet/data/transforms.py:75-107) = Pillow's 8-bit BILINEAR resample + paste on a fill-colour canvas.
// The integer coefficient tables (Pillow's precompute_coeffs + normalize_coeffs_8bpc, 22 fractional bits) are built on
// the host in double precision exactly as Pillow does; the device only does the integer passes, so results are
// bit-identical to `Image.resize(.., Image.BILINEAR)`: horizontal pass -> 8-bit temporary -> vertical pass.
// ------------------------------------------------------------------------------------------------
namespace {

struct ResizeArgs {
    const unsigned char* src; unsigned char* tmp; unsigned char* dst;
    int h, w, sw, sh, S;
    const int* bx; const int* kx; int ksx;       // [sw][2], [sw][ksx]
    const int* by; const int* ky; int ksy;       // [sh][2], [sh][ksy]
    int fill[3];
};

DEV unsigned char clip8(int v) { v >>= 22; return (unsigned char)(v < 0 ? 0 : v > 255 ? 255 : v); }

// src HWC [h][w][3] -> tmp HWC [h][sw][3]
__global__ __launch_bounds__(256) void resize_h_kernel(ResizeArgs p) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)p.h * p.sw) return;
    const int y = (int)(i / p.sw), xx = (int)(i % p.sw);
    const int xmin = p.bx[2 * xx], cnt = p.bx[2 * xx + 1];
    const int* k = p.kx + (long long)xx * p.ksx;
    int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
    const unsigned char* row = p.src + ((long long)y * p.w + xmin) * 3;
    for (int x = 0; x < cnt; ++x) { const int c = k[x]; s0 += row[3 * x] * c; s1 += row[3 * x + 1] * c; s2 += row[3 * x + 2] * c; }
    unsigned char* o = p.tmp + ((long long)y * p.sw + xx) * 3;
    o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
}

// tmp HWC [h][sw][3] -> dst CHW [3][S][S] (top-left paste, fill elsewhere)
__global__ __launch_bounds__(256) void resize_v_paste_kernel(ResizeArgs p) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)p.S * p.S) return;
    const int yy = (int)(i / p.S), xx = (int)(i % p.S);
    int v0 = p.fill[0], v1 = p.fill[1], v2 = p.fill[2];
    if (yy < p.sh && xx < p.sw) {
        const int ymin = p.by[2 * yy], cnt = p.by[2 * yy + 1];
        const int* k = p.ky + (long long)yy * p.ksy;
        int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
        for (int y = 0; y < cnt; ++y) {
            const unsigned char* px = p.tmp + ((long long)(ymin + y) * p.sw + xx) * 3;
            const int c = k[y];
            s0 += px[0] * c; s1 += px[1] * c; s2 += px[2] * c;
        }
        v0 = clip8(s0); v1 = clip8(s1); v2 = clip8(s2);
    }
    const long long plane = (long long)p.S * p.S;
    p.dst[i] = (unsigned char)v0; p.dst[plane + i] = (unsigned char)v1; p.dst[2 * plane + i] = (unsigned char)v2;
}

}  // namespace

extern "C" int effdet_resize_pad_u8(void* stream, const unsigned char* src, int h, int w, unsigned char* dst, int S, int sw, int sh,
                                    const int* bounds_x, const int* coef_x, int ksize_x,
                                    const int* bounds_y, const int* coef_y, int ksize_y,
                                    const int* fill_rgb, unsigned char* workspace) {
    EFFDET_ENTER();
    if (!src || !dst || !bounds_x || !coef_x || !bounds_y || !coef_y || !fill_rgb || !workspace) return EFFDET_EINVAL;
    if (h <= 0 || w <= 0 || S <= 0 || sw <= 0 || sh <= 0 || sw > S || sh > S || ksize_x <= 0 || ksize_y <= 0) return EFFDET_EINVAL;
    ResizeArgs a{src, workspace, dst, h, w, sw, sh, S, bounds_x, coef_x, ksize_x, bounds_y, coef_y, ksize_y,
                 {fill_rgb[0], fill_rgb[1], fill_rgb[2]}};
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const long long n1 = (long long)h * sw, n2 = (long long)S * S;
    hipLaunchKernelGGL(resize_h_kernel, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, st, a);
    hipLaunchKernelGGL(resize_v_paste_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, st, a);
    return effdet_check_launch();
}
