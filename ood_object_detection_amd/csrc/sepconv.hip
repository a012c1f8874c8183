// Fused "combine -> SiLU -> depthwise 3x3 -> pointwise 1x1 -> affine -> (SiLU)" kernel.
//
// One launch implements
//   * a BiFPN node: FpnCombine (weighted fusion of 2-3 inputs, each taken as-is, nearest-upsampled x2
//     or 3x3/s2 TF-SAME max-pooled on the fly) -> Swish -> SeparableConv2d -> BN
//     (effdet/efficientdet.py:224-245, :281-292, :76-83, :165-177), or
//   * one HeadNet layer for ALL pyramid levels at once: SeparableConv2d -> per-level BN -> Swish, or the
//     predict SeparableConv2d (+bias) (effdet/efficientdet.py:438-452) - conv weights are shared by
//     the levels, only the affine differs - with, for the class head, the per-anchor OOD scores
//     energy = -logsumexp_c(z) and max_logit = max_c(z) reduced in the epilogue while the logits tile
//     is still in LDS.
//
// A workgroup owns a TH x TW pixel tile of one level of one image:
//   phase 1  fused+activated (TH+2) x (TW+2) halo tile -> LDS, 64 channels at a time
//   phase 2  depthwise 3x3 out of LDS -> A tile [TH*TW][F] in LDS
//   phase 3  A x Wpw^T by 16x16 MFMA tiles, BN output columns at a time; accumulators staged through
//            LDS, affine/activation applied, whole 16-byte row pieces stored to HBM
// so every feature map is read once and written once per node/layer.
#include "common.h"

namespace {

struct SepInput {
    const void* ptr; long long image_stride;   // elements between images
    int H, W; int mode;                        // 0 same size, 1 nearest x2 up, 2 maxpool 3x3/s2 SAME
    int pad_t, pad_l;
};
struct SepLevel {
    int H, W, tiles_x, tiles_y, tile_begin, affine_row;
    SepInput in[3];
    void* out; long long out_image_stride;
    long long ood_off;
};
struct SepArgs {
    int nlevels; SepLevel lv[5];
    int n_in, fuse_mode;                       // fuse_mode 0: single input; 1: (x*w)/den; 2: x*w
    float fw[3]; float fden;
    int pre_act, post_act;
    const float* dw_w;                         // [9][F]
    const void* pw_w;                          // [N][F]
    const float* scale; const float* shift;    // [rows][N]; scale may be null
    int F, N;
    int ood_classes, num_anchors;              // > 0: column chunks are cut per anchor
    float* ood_energy; float* ood_maxlogit; long long ood_image_stride;
};

constexpr int FC = 64;    // channels per halo pass

// (m, s) <- log-sum-exp merge with (om, os); a -inf maximum carries a zero sum
DEV void merge_lse(float& m, float& s, float om, float os) {
    const float nm = fmaxf(m, om);
    const float s1 = (m == -INFINITY) ? 0.f : s * expf(m - nm);
    const float s2 = (om == -INFINITY) ? 0.f : os * expf(om - nm);
    m = nm; s = s1 + s2;
}

template <typename T>
DEV F8 fetch_input(const SepInput& in, int b, int y, int x, int F, int c) {
    const T* base = reinterpret_cast<const T*>(in.ptr) + (long long)b * in.image_stride;
    if (in.mode == 0) return load8<T>(base + ((long long)y * in.W + x) * F + c);
    if (in.mode == 1) return load8<T>(base + ((long long)(y >> 1) * in.W + (x >> 1)) * F + c);
    F8 m = f8_fill(-INFINITY);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = 2 * y + ky - in.pad_t;
        if (iy < 0 || iy >= in.H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = 2 * x + kx - in.pad_l;
            if (ix < 0 || ix >= in.W) continue;
            const F8 v = load8<T>(base + ((long long)iy * in.W + ix) * F + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) m.v[e] = fmaxf(m.v[e], v.v[e]);
        }
    }
    return m;
}

template <typename T, int TH, int TW, int BN>
__global__ __launch_bounds__(256) void sepconv_kernel(SepArgs p) {
    constexpr int BM = TH * TW;
    constexpr int HW_ = (TH + 2) * (TW + 2);
    constexpr int WPT = BM / 64;                 // 16-row MFMA tiles per wave (BM/4 rows per wave)
    constexpr int NT = BN / 16;
    constexpr int SROW = BN + 12;                // + 8 columns the alignment shift can spill into, + 4 bank spread
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int F = p.F, N = p.N;
    const int fbytes = F * (int)sizeof(T);
    const int nkc = (fbytes + 63) / 64;
    const int arow = nkc * 64 + 16;              // A / W row pitch in bytes
    // LDS carve (all multiples of 16)
    constexpr int HALO_BYTES = HW_ * FC * (int)sizeof(T);
    constexpr int STAGE_BYTES = BM * SROW * 4 + BM * 8;
    constexpr int R0 = HALO_BYTES > STAGE_BYTES ? HALO_BYTES : STAGE_BYTES;
    char* halo = lds;                            // phase 1/2
    float* S = reinterpret_cast<float*>(lds);    // phase 3 staging (aliases halo)
    float* run_m = S + BM * SROW;                // running max / sum-exp per row (OOD)
    float* run_s = run_m + BM;
    char* At = lds + R0;
    char* Wt = At + BM * arow;
    float* dww = reinterpret_cast<float*>(Wt + BN * arow);   // [9][F]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    int li = 0;
#pragma unroll
    for (int q = 1; q < 5; ++q) if (q < p.nlevels && (int)blockIdx.x >= p.lv[q].tile_begin) li = q;
    const SepLevel& L = p.lv[li];
    const int t = blockIdx.x - L.tile_begin;
    const int y0 = (t / L.tiles_x) * TH, x0 = (t % L.tiles_x) * TW;
    const int H = L.H, W = L.W;

    for (int i = tid; i < 9 * F; i += 256) dww[i] = p.dw_w[i];
    // zero the K padding of the A tile rows once (columns [fbytes, nkc*64))
    if (nkc * 64 > fbytes) {
        const int padb = nkc * 64 - fbytes;
        for (int i = tid; i < BM * (padb / 16); i += 256) {
            const int row = i / (padb / 16), piece = i % (padb / 16);
            *reinterpret_cast<u32x4*>(At + row * arow + fbytes + piece * 16) = u32x4{0u, 0u, 0u, 0u};
        }
    }

    // ------------------------------------------------------------------ phases 1 + 2 per 64 channels
    for (int fc0 = 0; fc0 < F; fc0 += FC) {
        const int fcn = (F - fc0) < FC ? (F - fc0) : FC;
        const int fcg = fcn / 8;
        __syncthreads();
        for (int it = tid; it < HW_ * fcg; it += 256) {
            const int cg = it % fcg, hp = it / fcg;
            const int y = y0 + hp / (TW + 2) - 1, x = x0 + hp % (TW + 2) - 1;
            F8 v = f8_zero();
            if (y >= 0 && y < H && x >= 0 && x < W) {
                const int c = fc0 + cg * 8;
                if (p.fuse_mode == 0) {
                    v = fetch_input<T>(L.in[0], b, y, x, F, c);
                } else {
                    for (int i = 0; i < p.n_in; ++i) {
                        const F8 xi = fetch_input<T>(L.in[i], b, y, x, F, c);
                        if (p.fuse_mode == 1) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v.v[e] += (xi.v[e] * p.fw[i]) / p.fden;
                        } else {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v.v[e] += xi.v[e] * p.fw[i];
                        }
                    }
                }
                if (p.pre_act) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v.v[e] = silu_t<T>(v.v[e]);
                }
            }
            store8<T>(reinterpret_cast<T*>(halo) + hp * FC + cg * 8, v);
        }
        __syncthreads();
        for (int it = tid; it < BM * fcg; it += 256) {
            const int cg = it % fcg, px = it / fcg;
            const int ty = px / TW, tx = px % TW;
            F8 acc = f8_zero();
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const F8 xv = load8<T>(reinterpret_cast<const T*>(halo) + ((ty + ky) * (TW + 2) + tx + kx) * FC + cg * 8);
                    const float* w = dww + (ky * 3 + kx) * F + fc0 + cg * 8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc.v[e] = fmaf(xv.v[e], w[e], acc.v[e]);
                }
            store8<T>(reinterpret_cast<T*>(At + px * arow) + fc0 + cg * 8, acc);
        }
    }
    __syncthreads();

    // ------------------------------------------------------------------ phase 3: column chunks
    const int frow = lane & 15, fpiece = lane >> 4;
    const int C = p.ood_classes;
    const bool ood = C > 0;
    const int subs = ood ? (C + BN - 1) / BN : 1;
    const int nchunks = ood ? p.num_anchors * subs : (N + BN - 1) / BN;
    const float* scale = p.scale ? p.scale + (long long)L.affine_row * N : nullptr;
    const float* shift = p.shift + (long long)L.affine_row * N;
    T* out = reinterpret_cast<T*>(L.out) + (long long)b * L.out_image_stride;
    float* cs = dww + 9 * F;                           // per-chunk scale[BN], shift[BN]
    const int ppr = nkc * 4;                           // 16-byte pieces per W row
    constexpr int WPC = 4;                             // W pieces a thread may prefetch (BN * ppr <= 1024)
    u32x4 wpre[WPC];

    auto chunk_range = [&](int ch, int& n_begin, int& n_count) {
        if (ood) {
            const int a = ch / subs, sc = ch % subs;
            n_begin = a * C + sc * BN;
            n_count = C - sc * BN; if (n_count > BN) n_count = BN;
        } else {
            n_begin = ch * BN;
            n_count = N - n_begin; if (n_count > BN) n_count = BN;
        }
    };
    const bool prefetch = BN * ppr <= 256 * WPC;
    auto w_fetch = [&](int ch) {                        // global -> registers (in flight across the epilogue)
        int n_begin, n_count;
        chunk_range(ch, n_begin, n_count);
#pragma unroll
        for (int q = 0; q < WPC; ++q) {
            const int i = tid + 256 * q;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (i < BN * ppr) {
                const int row = i / ppr, piece = i % ppr;
                if (row < n_count && piece * 16 < fbytes)
                    v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.pw_w) +
                                                        (long long)(n_begin + row) * fbytes + piece * 16);
            }
            wpre[q] = v;
        }
    };
    auto w_commit = [&]() {
#pragma unroll
        for (int q = 0; q < WPC; ++q) {
            const int i = tid + 256 * q;
            if (i < BN * ppr) *reinterpret_cast<u32x4*>(Wt + (i / ppr) * arow + (i % ppr) * 16) = wpre[q];
        }
    };
    if (prefetch) w_fetch(0);

    for (int ch = 0; ch < nchunks; ++ch) {
        int n_begin, n_count;
        chunk_range(ch, n_begin, n_count);
        if (prefetch) {
            w_commit();
        } else {
            for (int i = tid; i < BN * ppr; i += 256) {
                const int row = i / ppr, piece = i % ppr;
                u32x4 v = {0u, 0u, 0u, 0u};
                if (row < n_count && piece * 16 < fbytes)
                    v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.pw_w) +
                                                        (long long)(n_begin + row) * fbytes + piece * 16);
                *reinterpret_cast<u32x4*>(Wt + row * arow + piece * 16) = v;
            }
        }
        if (tid < BN) {
            cs[tid] = (scale && tid < n_count) ? scale[n_begin + tid] : 1.0f;
            cs[BN + tid] = tid < n_count ? shift[n_begin + tid] : 0.0f;
        }
        if (ood && (ch % subs) == 0) {
            for (int i = tid; i < BM; i += 256) { run_m[i] = -INFINITY; run_s[i] = 0.f; }
        }
        __syncthreads();
        if (prefetch && ch + 1 < nchunks) w_fetch(ch + 1);

        f32x4 acc[WPT][NT];
#pragma unroll
        for (int i = 0; i < WPT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kc = 0; kc < nkc; ++kc) {
            Frag<T> a[WPT];
#pragma unroll
            for (int i = 0; i < WPT; ++i)
                a[i] = ld_frag<T>(At + (16 * WPT * wave + 16 * i + frow) * arow + kc * 64 + fpiece * 16);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                Frag<T> bf = ld_frag<T>(Wt + (16 * j + frow) * arow + kc * 64 + fpiece * 16);
#pragma unroll
                for (int i = 0; i < WPT; ++i) mma_chunk(a[i], bf, acc[i][j]);
            }
        }
        // Stage the finished values (affine + activation applied here, in MFMA layout) into LDS.  Row `row` is
        // stored shifted right by delta(row) columns, chosen so that staging column 8k of the row is the element
        // that sits on a 16-byte boundary IN MEMORY (a row may start at any element offset, e.g. the 1620-byte
        // class rows): the store pass then reads two aligned float4 per thread and writes whole 16-byte pieces.
        constexpr int ALIGN_E = 16 / (int)sizeof(T);       // elements per 16 bytes
        {
            int delta[WPT][4];
#pragma unroll
            for (int i = 0; i < WPT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * WPT * wave + 16 * i + 4 * fpiece + r;
                    const T* drow = out + ((long long)(y0 + row / TW) * W + (x0 + row % TW)) * N + n_begin;
                    const int e0 = (int)((reinterpret_cast<uintptr_t>(drow) / sizeof(T)) % ALIGN_E);
                    delta[i][r] = (8 - (ALIGN_E - e0) % ALIGN_E) % 8;
                }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const float csc = cs[16 * j + frow], csh = cs[BN + 16 * j + frow];
#pragma unroll
                for (int i = 0; i < WPT; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = acc[i][j][r] * csc + csh;
                        if (p.post_act) v = silu_t<T>(v);
                        S[(16 * WPT * wave + 16 * i + 4 * fpiece + r) * SROW + 16 * j + frow + delta[i][r]] = v;
                    }
            }
        }
        __syncthreads();

        // Store pass: GPR threads per row, thread cg owns staging columns [8cg, 8cg+8) (+ the last thread of the
        // row the 9th window that the shift can spill into).
        constexpr int GPR = BN / 8;
        for (int g = tid; g < BM * GPR; g += 256) {        // BM*GPR is a multiple of 256: no divergence
            const int row = g / GPR, cg = g % GPR;
            const int y = y0 + row / TW, x = x0 + row % TW;
            const bool inside = (y < H) && (x < W);
            T* drow = out + ((long long)y * W + x) * N + n_begin;
            const int e0 = (int)((reinterpret_cast<uintptr_t>(drow) / sizeof(T)) % ALIGN_E);
            const int dl = (8 - (ALIGN_E - e0) % ALIGN_E) % 8;
            float tm = -INFINITY, ts = 0.f;
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const int s0 = pass == 0 ? cg * 8 : BN;        // staging column of this window
                if (pass == 1 && (cg != GPR - 1 || dl == 0)) continue;
                const f32x4 va = *reinterpret_cast<const f32x4*>(S + row * SROW + s0);
                const f32x4 vb = *reinterpret_cast<const f32x4*>(S + row * SROW + s0 + 4);
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = va[e]; v[4 + e] = vb[e]; }
                const int c_lo = s0 - dl;                      // chunk column of v[0]
                int lo = c_lo < 0 ? -c_lo : 0;                 // valid element range [lo, hi) inside the window
                int hi = n_count - c_lo; hi = hi > 8 ? 8 : hi;
                if (lo >= hi) continue;
                if (inside) {
                    T* dst = drow + c_lo;
                    if (lo == 0 && hi == 8) {
                        F8 o;
#pragma unroll
                        for (int e = 0; e < 8; ++e) o.v[e] = v[e];
                        store8<T>(dst, o);
                    } else {
                        for (int e = lo; e < hi; ++e) dst[e] = from_f<T>(v[e]);
                    }
                }
                if (ood) {
                    float wm = -INFINITY, wsum = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) if (e >= lo && e < hi) wm = fmaxf(wm, v[e]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) if (e >= lo && e < hi) wsum += exp_t<T>(v[e] - wm);
                    merge_lse(tm, ts, wm, wsum);
                }
            }
            if (ood) {
#pragma unroll
                for (int o = 1; o < GPR; o <<= 1) {
                    const float om = __shfl_xor(tm, o, 64), os = __shfl_xor(ts, o, 64);
                    merge_lse(tm, ts, om, os);
                }
                if (cg == 0) {
                    float pm = run_m[row], ps = run_s[row];
                    merge_lse(pm, ps, tm, ts);
                    run_m[row] = pm; run_s[row] = ps;
                    if ((ch % subs) == subs - 1 && inside) {
                        const int a = ch / subs;
                        const long long idx = (long long)b * p.ood_image_stride + L.ood_off +
                                              ((long long)y * W + x) * p.num_anchors + a;
                        p.ood_energy[idx] = -(pm + logf(ps));
                        p.ood_maxlogit[idx] = pm;
                    }
                }
            }
        }
        __syncthreads();
    }
}

template <typename T, int TH, int TW, int BN>
size_t sep_lds_bytes(int F) {
    constexpr int BM = TH * TW;
    constexpr int HW_ = (TH + 2) * (TW + 2);
    constexpr int SROW = BN + 12;
    const int nkc = (F * (int)sizeof(T) + 63) / 64;
    const int arow = nkc * 64 + 16;
    const size_t halo = (size_t)HW_ * FC * sizeof(T);
    const size_t stage = (size_t)BM * SROW * 4 + BM * 8;
    return (halo > stage ? halo : stage) + (size_t)BM * arow + (size_t)BN * arow + (size_t)9 * F * 4 + (size_t)2 * BN * 4;
}

template <typename T, int TH, int TW, int BN>
int launch_sep(hipStream_t st, SepArgs& a, int B) {
    int tiles = 0;
    for (int i = 0; i < a.nlevels; ++i) {
        a.lv[i].tiles_x = (a.lv[i].W + TW - 1) / TW;
        a.lv[i].tiles_y = (a.lv[i].H + TH - 1) / TH;
        a.lv[i].tile_begin = tiles;
        tiles += a.lv[i].tiles_x * a.lv[i].tiles_y;
    }
    const size_t lds = sep_lds_bytes<T, TH, TW, BN>(a.F);
    if (lds > 160 * 1024) return EFFDET_EINVAL;
    auto kern = sepconv_kernel<T, TH, TW, BN>;
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return EFFDET_ELAUNCH;
    }
    hipLaunchKernelGGL(kern, dim3(tiles, B), dim3(256), lds, st, a);
    return effdet_check_launch();
}

}  // namespace

// Flat C-ABI descriptor (mirrors SepArgs; arrays are per level / per input)
extern "C" int effdet_sepconv_fused(
    void* stream, int dtype, int B, int nlevels,
    const int* level_hw,                  // [nlevels][2]  output H, W
    int n_in,
    const void* const* in_ptr,            // [nlevels][n_in]
    const long long* in_image_stride,     // [nlevels][n_in]
    const int* in_hw,                     // [nlevels][n_in][2]
    const int* in_mode,                   // [nlevels][n_in]
    int fuse_mode, const float* fuse_w, float fuse_den, int pre_act,
    const float* dw_w, const void* pw_w, const float* scale, const float* shift,
    const int* affine_row,                // [nlevels]
    int post_act, int F, int N,
    void* const* out_ptr, const long long* out_image_stride,   // [nlevels]
    int ood_classes, int num_anchors, float* ood_energy, float* ood_maxlogit,
    long long ood_image_stride, const long long* ood_level_off) {
    EFFDET_ENTER();
    if (nlevels < 1 || nlevels > 5 || n_in < 1 || n_in > 3 || B <= 0) return EFFDET_EINVAL;
    if (!level_hw || !in_ptr || !in_image_stride || !in_hw || !in_mode || !dw_w || !pw_w || !shift || !affine_row ||
        !out_ptr || !out_image_stride) return EFFDET_EINVAL;
    if (F <= 0 || F % 8 || N <= 0 || (dtype & ~1) || fuse_mode < 0 || fuse_mode > 2) return EFFDET_EINVAL;
    if (fuse_mode != 0 && !fuse_w) return EFFDET_EINVAL;
    if (fuse_mode == 0 && n_in != 1) return EFFDET_EINVAL;
    if (ood_classes > 0) {
        if (!ood_energy || !ood_maxlogit || !ood_level_off || num_anchors <= 0 || num_anchors * ood_classes != N) return EFFDET_EINVAL;
    }
    SepArgs a;
    a.nlevels = nlevels; a.n_in = n_in; a.fuse_mode = fuse_mode; a.fden = fuse_den;
    for (int i = 0; i < 3; ++i) a.fw[i] = (fuse_mode != 0 && i < n_in) ? fuse_w[i] : 0.f;
    a.pre_act = pre_act; a.post_act = post_act; a.dw_w = dw_w; a.pw_w = pw_w; a.scale = scale; a.shift = shift;
    a.F = F; a.N = N; a.ood_classes = ood_classes > 0 ? ood_classes : 0; a.num_anchors = num_anchors;
    a.ood_energy = ood_energy; a.ood_maxlogit = ood_maxlogit; a.ood_image_stride = ood_image_stride;
    for (int l = 0; l < nlevels; ++l) {
        SepLevel& L = a.lv[l];
        L.H = level_hw[2 * l]; L.W = level_hw[2 * l + 1];
        if (L.H <= 0 || L.W <= 0) return EFFDET_EINVAL;
        L.affine_row = affine_row[l];
        L.out = out_ptr[l]; L.out_image_stride = out_image_stride[l];
        L.ood_off = (ood_classes > 0) ? ood_level_off[l] : 0;
        if (!L.out) return EFFDET_EINVAL;
        for (int i = 0; i < n_in; ++i) {
            SepInput& I = L.in[i];
            I.ptr = in_ptr[l * n_in + i]; I.image_stride = in_image_stride[l * n_in + i];
            I.H = in_hw[(l * n_in + i) * 2]; I.W = in_hw[(l * n_in + i) * 2 + 1];
            I.mode = in_mode[l * n_in + i];
            I.pad_t = I.pad_l = 0;
            if (!I.ptr) return EFFDET_EINVAL;
            if (I.mode == 0) { if (I.H != L.H || I.W != L.W) return EFFDET_EINVAL; }
            else if (I.mode == 1) { if (I.H * 2 != L.H || I.W * 2 != L.W) return EFFDET_EINVAL; }
            else if (I.mode == 2) {
                if (same_out(I.H, 2) != L.H || same_out(I.W, 2) != L.W) return EFFDET_EINVAL;
                I.pad_t = same_pad_before(I.H, 3, 2); I.pad_l = same_pad_before(I.W, 3, 2);
            } else return EFFDET_EINVAL;
        }
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == 0) return launch_sep<float, 8, 8, 64>(st, a, B);
    return launch_sep<bf16_t, 8, 16, 64>(st, a, B);
}
