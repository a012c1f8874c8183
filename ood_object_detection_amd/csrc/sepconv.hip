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
#include <type_traits>

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

template <typename T, int TH, int TW, int BN, bool OOD, int NTH>
__global__ __launch_bounds__(NTH, NTH == 512 ? 4 : ((OOD || sizeof(T) == 4) ? 2 : 3)) void sepconv_kernel(SepArgs p) {
    // staging element: fp32 when the OOD reduction reads it back, else the output dtype (half the LDS)
    typedef typename std::conditional<OOD, float, T>::type ST;
    constexpr int BM = TH * TW;
    constexpr int HW_ = (TH + 2) * (TW + 2);
    constexpr int NWAVE = NTH / 64;
    constexpr int WPT = BM / (16 * NWAVE);       // 16-row MFMA tiles per wave
    constexpr int NT = BN / 16;
    constexpr int SROW = BN + 24;                // + 8 columns the alignment shift can spill into, + bank spread
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int F = p.F, N = p.N;
    const int fbytes = F * (int)sizeof(T);
    const int nkc = (fbytes + 63) / 64;
    const int arow = nkc * 64 + 16;              // A / W row pitch in bytes
    // LDS carve (all multiples of 16)
    constexpr int HALO_BYTES = HW_ * FC * (int)sizeof(T);
    constexpr int STAGE_BYTES = BM * SROW * (int)sizeof(ST) + (OOD ? BM * 8 : 0);
    constexpr int R0 = HALO_BYTES > STAGE_BYTES ? HALO_BYTES : STAGE_BYTES;
    char* halo = lds;                            // phase 1/2
    ST* S = reinterpret_cast<ST*>(lds);          // phase 3 staging (aliases halo)
    float* run_m = reinterpret_cast<float*>(lds + BM * SROW * sizeof(ST));   // running max / sum-exp per row (OOD)
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

    // depthwise taps: fetched into registers now, committed to LDS after the halo loads have been issued
    constexpr int DPC = (9 * 128 + NTH - 1) / NTH;        // taps per thread when F <= 128 (else copied later)
    const bool d_pref = 9 * F <= DPC * NTH;
    float dpre[DPC];
#pragma unroll
    for (int q = 0; q < DPC; ++q) {
        const int i = tid + NTH * q;
        dpre[q] = (d_pref && i < 9 * F) ? p.dw_w[i] : 0.f;
    }
    // zero the K padding of the A tile rows once (columns [fbytes, nkc*64))
    if (nkc * 64 > fbytes) {
        const int padb = nkc * 64 - fbytes;
        for (int i = tid; i < BM * (padb / 16); i += NTH) {
            const int row = i / (padb / 16), piece = i % (padb / 16);
            *reinterpret_cast<u32x4*>(At + row * arow + fbytes + piece * 16) = u32x4{0u, 0u, 0u, 0u};
        }
    }

    // ------------------------------------------------------------------ phases 1 + 2 per 64 channels
    for (int fc0 = 0; fc0 < F; fc0 += FC) {
        const int fcn = (F - fc0) < FC ? (F - fc0) : FC;
        const int fcg = fcn / 8;
        __syncthreads();
        // two halo items per step: their input loads are all issued before the first use (the loop has a runtime
        // trip count, so the compiler would otherwise expose one memory round trip per item)
        constexpr int HU = NTH == 512 ? 1 : 2;              // halo items in flight per thread
        for (int it0 = tid; it0 < HW_ * fcg; it0 += HU * NTH) {
            F8 xin[HU][3];
            bool ok[HU];
#pragma unroll
            for (int u = 0; u < HU; ++u) {
                const int it = it0 + NTH * u;
                const int cgh = it % fcg, hp = it / fcg;
                const int y = y0 + hp / (TW + 2) - 1, x = x0 + hp % (TW + 2) - 1;
                ok[u] = it < HW_ * fcg && y >= 0 && y < H && x >= 0 && x < W;
                if (ok[u]) {
                    const int c = fc0 + cgh * 8;
#pragma unroll
                    for (int i = 0; i < 3; ++i)
                        if (i < p.n_in) xin[u][i] = fetch_input<T>(L.in[i], b, y, x, F, c);
                }
            }
#pragma unroll
            for (int u = 0; u < HU; ++u) {
                const int it = it0 + NTH * u;
                if (it >= HW_ * fcg) continue;
                const int cgh = it % fcg, hp = it / fcg;
                F8 v = f8_zero();
                if (ok[u]) {
                    if (p.fuse_mode == 0) {
                        v = xin[u][0];
                    } else {
#pragma unroll
                        for (int i = 0; i < 3; ++i) {
                            if (i < p.n_in) {
                                if (p.fuse_mode == 1) {
#pragma unroll
                                    for (int e = 0; e < 8; ++e) v.v[e] += (xin[u][i].v[e] * p.fw[i]) / p.fden;
                                } else {
#pragma unroll
                                    for (int e = 0; e < 8; ++e) v.v[e] += xin[u][i].v[e] * p.fw[i];
                                }
                            }
                        }
                    }
                    if (p.pre_act) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v.v[e] = silu_t<T>(v.v[e]);
                    }
                }
                store8<T>(reinterpret_cast<T*>(halo) + hp * FC + cgh * 8, v);
            }
        }
        if (fc0 == 0) {
            if (d_pref) {
#pragma unroll
                for (int q = 0; q < DPC; ++q) {
                    const int i = tid + NTH * q;
                    if (i < 9 * F) dww[i] = dpre[q];
                }
            } else {
                for (int i = tid; i < 9 * F; i += NTH) dww[i] = p.dw_w[i];
            }
        }
        __syncthreads();
        for (int it = tid; it < BM * fcg; it += NTH) {
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
    const int C = OOD ? p.ood_classes : 0;
    constexpr bool ood = OOD;
    const int subs = ood ? (C + BN - 1) / BN : 1;
    const int nchunks = ood ? p.num_anchors * subs : (N + BN - 1) / BN;
    const float* scale = p.scale ? p.scale + (long long)L.affine_row * N : nullptr;
    const float* shift = p.shift + (long long)L.affine_row * N;
    T* out = reinterpret_cast<T*>(L.out) + (long long)b * L.out_image_stride;
    float* cs = dww + 9 * F;                           // per-chunk scale[BN], shift[BN]
    const int ppr = nkc * 4;                           // 16-byte pieces per W row
    constexpr int WPC = 4;                             // W pieces a thread may prefetch (BN * ppr <= 1024)
    u32x4 wpre[WPC];
    float cpre_s = 1.0f, cpre_t = 0.0f;                 // next chunk's scale / shift for column `tid`

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
    const bool prefetch = BN * ppr <= NTH * WPC;
    auto w_fetch = [&](int ch) {                        // global -> registers (in flight across the epilogue)
        int n_begin, n_count;
        chunk_range(ch, n_begin, n_count);
#pragma unroll
        for (int q = 0; q < WPC; ++q) {
            const int i = tid + NTH * q;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (i < BN * ppr) {
                const int row = i / ppr, piece = i % ppr;
                if (row < n_count && piece * 16 < fbytes)
                    v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.pw_w) +
                                                        (long long)(n_begin + row) * fbytes + piece * 16);
            }
            wpre[q] = v;
        }
        if (tid < BN) {
            cpre_s = (scale && tid < n_count) ? scale[n_begin + tid] : 1.0f;
            cpre_t = tid < n_count ? shift[n_begin + tid] : 0.0f;
        }
    };
    auto w_commit = [&]() {
#pragma unroll
        for (int q = 0; q < WPC; ++q) {
            const int i = tid + NTH * q;
            if (i < BN * ppr) *reinterpret_cast<u32x4*>(Wt + (i / ppr) * arow + (i % ppr) * 16) = wpre[q];
        }
    };
    if (prefetch) w_fetch(0);

    // Per-row constants of this tile, computed once: element offset of the row inside the image's output
    // (pixel * N) and its position on the 16-byte grid.  `rowmod + n_begin` (mod ALIGN_E) is the shift that
    // puts staging column 8k on a 16-byte boundary in memory.
    constexpr int ALIGN_E = 16 / (int)sizeof(T);       // elements per 16 bytes
    constexpr int GPR = BN / 8;
    constexpr int RPT = BM * GPR / NTH;                // rows per thread in the store pass
    const int base_mod = (int)((reinterpret_cast<uintptr_t>(out) / sizeof(T)) % ALIGN_E);
    int st_mod[WPT][4];                                // staging rows of this lane (MFMA layout)
#pragma unroll
    for (int i = 0; i < WPT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * WPT * wave + 16 * i + 4 * fpiece + r;
            const int pix = (y0 + row / TW) * W + (x0 + row % TW);
            st_mod[i][r] = (base_mod + (int)(((long long)pix * N) % ALIGN_E)) % ALIGN_E;
        }
    int sp_off[RPT], sp_mod[RPT];                      // store-pass rows of this thread
    bool sp_in[RPT];
    const int cg = tid % GPR;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int row = (tid + NTH * k) / GPR;
        const int y = y0 + row / TW, x = x0 + row % TW;
        sp_in[k] = (y < H) && (x < W);
        sp_off[k] = (y * W + x) * N;
        sp_mod[k] = (base_mod + (int)(((long long)(y * W + x) * N) % ALIGN_E)) % ALIGN_E;
    }

    for (int ch = 0; ch < nchunks; ++ch) {
        int n_begin, n_count;
        chunk_range(ch, n_begin, n_count);
        if (prefetch) {
            w_commit();
        } else {
            for (int i = tid; i < BN * ppr; i += NTH) {
                const int row = i / ppr, piece = i % ppr;
                u32x4 v = {0u, 0u, 0u, 0u};
                if (row < n_count && piece * 16 < fbytes)
                    v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.pw_w) +
                                                        (long long)(n_begin + row) * fbytes + piece * 16);
                *reinterpret_cast<u32x4*>(Wt + row * arow + piece * 16) = v;
            }
        }
        if (tid < BN) {
            if (prefetch) { cs[tid] = cpre_s; cs[BN + tid] = cpre_t; }
            else {
                cs[tid] = (scale && tid < n_count) ? scale[n_begin + tid] : 1.0f;
                cs[BN + tid] = tid < n_count ? shift[n_begin + tid] : 0.0f;
            }
        }
        if (ood && (ch % subs) == 0) {
            for (int i = tid; i < BM; i += NTH) { run_m[i] = -INFINITY; run_s[i] = 0.f; }
        }
        __syncthreads();
        if (prefetch && ch + 1 < nchunks) w_fetch(ch + 1);

        const int njt = (n_count + 15) / 16;               // 16-column tiles that hold real columns
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
                if (j < njt) {
                    Frag<T> bf = ld_frag<T>(Wt + (16 * j + frow) * arow + kc * 64 + fpiece * 16);
#pragma unroll
                    for (int i = 0; i < WPT; ++i) mma_chunk(a[i], bf, acc[i][j]);
                }
            }
        }
        // Stage the finished values (affine + activation applied here, in MFMA layout).  Row `row` is stored
        // shifted right by delta(row) columns so that staging column 8k is the element on a 16-byte boundary
        // IN MEMORY (rows may start anywhere, e.g. the 1620-byte class rows): the store pass then reads aligned
        // 8-element windows and writes whole 16-byte pieces.
        const int nb_mod = n_begin % ALIGN_E;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if (j < njt) {
                const float csc = cs[16 * j + frow], csh = cs[BN + 16 * j + frow];
#pragma unroll
                for (int i = 0; i < WPT; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = acc[i][j][r] * csc + csh;
                        if (p.post_act) v = silu_t<T>(v);
                        const int delta = (st_mod[i][r] + nb_mod) % ALIGN_E;
                        S[(16 * WPT * wave + 16 * i + 4 * fpiece + r) * SROW + 16 * j + frow + delta] = (ST)v;
                    }
            }
        }
        __syncthreads();

        // Store pass: GPR threads per row; thread cg owns staging columns [8cg, 8cg+8); the last thread of the
        // row also owns the window the shift spills into.
        constexpr float LOG2E = 1.4426950408889634f;
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int row = (tid + NTH * k) / GPR;
            const int dl = (sp_mod[k] + nb_mod) % ALIGN_E;
            T* drow = out + sp_off[k] + n_begin;
            const int c_lo = cg * 8 - dl;
            float v[8];
            if constexpr (sizeof(ST) == 4) {
                const f32x4 va = *reinterpret_cast<const f32x4*>(S + row * SROW + cg * 8);
                const f32x4 vb = *reinterpret_cast<const f32x4*>(S + row * SROW + cg * 8 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = va[e]; v[4 + e] = vb[e]; }
            } else {
                const bf16x8 va = *reinterpret_cast<const bf16x8*>(S + row * SROW + cg * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (float)va[e];
            }
            const bool full = (c_lo >= 0) && (c_lo + 8 <= n_count);
            float rm = -INFINITY, rs = 0.f;
            if (full) {                                        // the common case: no masks
                if (sp_in[k]) {
                    F8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o.v[e] = v[e];
                    store8<T>(drow + c_lo, o);
                }
                if constexpr (OOD) {
                    rm = fmaxf(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])), fmaxf(fmaxf(v[4], v[5]), fmaxf(v[6], v[7])));
                }
            } else {
                const int lo = c_lo < 0 ? -c_lo : 0;
                int hi = n_count - c_lo; hi = hi > 8 ? 8 : hi;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const bool ok = e >= lo && e < hi;
                    if (ok && sp_in[k]) drow[c_lo + e] = from_f<T>(v[e]);
                    if (!ok) v[e] = -INFINITY;
                    if constexpr (OOD) rm = fmaxf(rm, v[e]);
                }
            }
            float v2[8];
            bool spill = false;
            if (cg == GPR - 1 && dl > 0) {                     // the spill window [BN, BN + dl)
                spill = true;
                const int c2 = BN - dl;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const bool ok = e < dl && c2 + e < n_count;
                    v2[e] = ok ? (float)S[row * SROW + BN + e] : -INFINITY;
                    if (ok && sp_in[k]) drow[c2 + e] = from_f<T>(v2[e]);
                    if constexpr (OOD) rm = fmaxf(rm, v2[e]);
                }
            }
            if constexpr (OOD) {
#pragma unroll
                for (int o = 1; o < GPR; o <<= 1) rm = fmaxf(rm, __shfl_xor(rm, o, 64));
                const float t = (rm == -INFINITY) ? 0.f : rm;
                if constexpr (sizeof(T) == 2) {
                    const float tl = t * LOG2E;
#pragma unroll
                    for (int e = 0; e < 8; ++e) rs += __builtin_amdgcn_exp2f(fmaf(v[e], LOG2E, -tl));   // exp2(-inf) = 0
                    if (spill) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) rs += __builtin_amdgcn_exp2f(fmaf(v2[e], LOG2E, -tl));
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) rs += (v[e] == -INFINITY) ? 0.f : expf(v[e] - t);
                    if (spill) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) rs += (v2[e] == -INFINITY) ? 0.f : expf(v2[e] - t);
                    }
                }
#pragma unroll
                for (int o = 1; o < GPR; o <<= 1) rs += __shfl_xor(rs, o, 64);
                if (cg == 0) {
                    float pm = run_m[row], ps = run_s[row];
                    if (rm != -INFINITY) {
                        const float nm = fmaxf(pm, rm);
                        ps = (pm == -INFINITY ? 0.f : ps * exp_t<T>(pm - nm)) + rs * exp_t<T>(rm - nm);
                        pm = nm;
                    }
                    run_m[row] = pm; run_s[row] = ps;
                    if ((ch % subs) == subs - 1 && sp_in[k]) {
                        const int a = ch / subs;
                        const long long idx = (long long)b * p.ood_image_stride + L.ood_off +
                                              (long long)(sp_off[k] / N) * p.num_anchors + a;
                        p.ood_energy[idx] = -(pm + logf(ps));
                        p.ood_maxlogit[idx] = pm;
                    }
                }
            }
        }
        __syncthreads();
    }
}

template <typename T, int TH, int TW, int BN, bool OOD>
size_t sep_lds_bytes(int F) {
    constexpr int BM = TH * TW;
    constexpr int HW_ = (TH + 2) * (TW + 2);
    constexpr int SROW = BN + 24;
    const int nkc = (F * (int)sizeof(T) + 63) / 64;
    const int arow = nkc * 64 + 16;
    const size_t halo = (size_t)HW_ * FC * sizeof(T);
    const size_t stage = (size_t)BM * SROW * (OOD ? 4 : sizeof(T)) + (OOD ? BM * 8 : 0);
    return (halo > stage ? halo : stage) + (size_t)BM * arow + (size_t)BN * arow + (size_t)9 * F * 4 + (size_t)2 * BN * 4;
}

template <typename T, int TH, int TW, int BN, bool OOD, int NTH>
int launch_sep(hipStream_t st, SepArgs& a, int B) {
    int tiles = 0;
    for (int i = 0; i < a.nlevels; ++i) {
        a.lv[i].tiles_x = (a.lv[i].W + TW - 1) / TW;
        a.lv[i].tiles_y = (a.lv[i].H + TH - 1) / TH;
        a.lv[i].tile_begin = tiles;
        tiles += a.lv[i].tiles_x * a.lv[i].tiles_y;
    }
    const size_t lds = sep_lds_bytes<T, TH, TW, BN, OOD>(a.F);
    if (lds > 160 * 1024) return EFFDET_EINVAL;
    auto kern = sepconv_kernel<T, TH, TW, BN, OOD, NTH>;
    if (lds > 64 * 1024) {
        static bool attr_done = false;           // one per template instantiation
        if (!attr_done) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return EFFDET_ELAUNCH;
            attr_done = true;
        }
    }
    hipLaunchKernelGGL(kern, dim3(tiles, B), dim3(NTH), lds, st, a);
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
    const bool ood = a.ood_classes > 0;
    if (dtype == 0) return ood ? launch_sep<float, 8, 8, 64, true, 256>(st, a, B) : launch_sep<float, 8, 8, 64, false, 256>(st, a, B);
    // bf16: 512 threads per 8x16 tile - the LDS footprint allows two workgroups per CU, so this doubles the waves
    // that share the VALU-heavy staging / store passes
    return ood ? launch_sep<bf16_t, 8, 16, 64, true, 512>(st, a, B) : launch_sep<bf16_t, 8, 16, 64, false, 512>(st, a, B);
}
