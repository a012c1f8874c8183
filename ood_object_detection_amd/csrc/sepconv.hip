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
//   phase 3  Wpw x A^T by 16x16 MFMA tiles, BN output channels at a time.  The accumulator rows are output
//            channels and its columns pixels, and the W rows are permuted on their way into LDS so that one lane
//            ends up with 8 CONSECUTIVE channels of one pixel per pair of tiles: affine / activation / the OOD
//            reduction all happen in registers and each lane stores 16 bytes straight to HBM (no staging pass).
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
    int in_affine_row;                         // META: row of in_scale / in_shift for this level
    void* dw_out; long long dw_out_image_stride;   // META: optional copy of the depthwise output [B, H*W, F]
};
struct SepArgs {
    int nlevels; SepLevel lv[5];
    int n_in, fuse_mode;                       // fuse_mode 0: single input; 1: (x*w)/den; 2: x*w
    float fw[3]; float fden;
    float fwn[3];                              // fw / fden (throughput mode multiplies instead of dividing)
    int vec_ok;                                // output rows are dword aligned: 8-channel pieces go out as vectors
    int out_f32;                               // outputs are written as float32 whatever T is (box regressions of a bf16 model)
    int pre_act, post_act;
    const float* dw_w;                         // [9][F]
    const void* pw_w;                          // [N][F]
    const float* scale; const float* shift;    // [rows][N]; scale may be null
    int F, N;
    int ood_classes, num_anchors;              // > 0: column chunks are cut per anchor
    float* ood_energy; float* ood_maxlogit; long long ood_image_stride;
    // META (MetaHead, effdet/efficientdet.py:569-695): batch-statistics BN of the previous layer arrives as a per-level
    // affine applied to the input before the activation, and this layer's outputs are summed for the next one's statistics
    const float* in_scale; const float* in_shift;   // [rows][F] or null
    float* stat_partial;                            // [B][tiles][2][N] per-workgroup sums / sums of squares, or null
    int tiles_total;
};

constexpr int FC = 64;    // channels per halo pass



// 8 channels as loaded (16 bytes of bf16 / 32 bytes of f32): halo inputs wait in this form, so that a thread can keep
// every load of a pass in flight without holding 8 converted floats per piece
template <typename T> struct Raw8 { u32x4 v[sizeof(T) == 2 ? 1 : 2]; };
template <typename T> DEV Raw8<T> load_raw8(const T* p) {
    Raw8<T> r;
    r.v[0] = *reinterpret_cast<const u32x4*>(p);
    if constexpr (sizeof(T) == 4) r.v[1] = *(reinterpret_cast<const u32x4*>(p) + 1);
    return r;
}
template <typename T> DEV F8 unpack8(const Raw8<T>& r) {
    F8 o;
    if constexpr (IsPair<T>::value) {
        return pair_join8(r.v[0], r.v[1]);
    } else if constexpr (sizeof(T) == 2) {
        const bf16x8 a = __builtin_bit_cast(bf16x8, r.v[0]);
#pragma unroll
        for (int e = 0; e < 8; ++e) o.v[e] = (float)a[e];
    } else {
        const f32x4 a = __builtin_bit_cast(f32x4, r.v[0]), b = __builtin_bit_cast(f32x4, r.v[1]);
#pragma unroll
        for (int e = 0; e < 4; ++e) { o.v[e] = a[e]; o.v[4 + e] = b[e]; }
    }
    return o;
}
template <typename T> DEV Raw8<T> pack8(const F8& f) {
    Raw8<T> r;
    if constexpr (IsPair<T>::value) {
        pair_split8(f, r.v[0], r.v[1]);
    } else if constexpr (sizeof(T) == 2) {
        bf16x8 a;
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = (bf16_t)f.v[e];
        r.v[0] = __builtin_bit_cast(u32x4, a);
    } else {
        r.v[0] = __builtin_bit_cast(u32x4, f32x4{f.v[0], f.v[1], f.v[2], f.v[3]});
        r.v[1] = __builtin_bit_cast(u32x4, f32x4{f.v[4], f.v[5], f.v[6], f.v[7]});
    }
    return r;
}

// 8 channels of input `in` (already offset to the image) at output-grid position (y, x), which the caller has clamped into the
// map.  Every load here is UNCONDITIONAL: same-size and nearest-upsampled inputs differ by a shift only, and the 3x3/s2 max pool
// reads clamped taps and replaces the out-of-map ones by -inf afterwards.  (Loads behind exec-mask branches made the compiler
// wait for each one on the spot - `s_waitcnt vmcnt(0)` after every 16 bytes of the halo; rounds 1 - 2.)
template <typename T>
DEV Raw8<T> fetch_input(const T* base, const SepInput& in, int y, int x, int F, int c) {
    if (in.mode != 2) {                                           // uniform per input
        const int sh = in.mode == 1 ? 1 : 0;
        return load_raw8<T>(base + ((y >> sh) * in.W + (x >> sh)) * F + c);
    }
    F8 m = f8_fill(-INFINITY);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = 2 * y + ky - in.pad_t;
        const int iyc = iy < 0 ? 0 : (iy >= in.H ? in.H - 1 : iy);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = 2 * x + kx - in.pad_l;
            const int ixc = ix < 0 ? 0 : (ix >= in.W ? in.W - 1 : ix);
            const F8 v = load8<T>(base + (iyc * in.W + ixc) * F + c);
            const bool inside = iy == iyc && ix == ixc;
#pragma unroll
            for (int e = 0; e < 8; ++e) m.v[e] = fmaxf(m.v[e], inside ? v.v[e] : -INFINITY);
        }
    }
    return pack8<T>(m);                          // a maximum of stored values: exactly representable
}

// 8 consecutive output channels -> memory.  Rows are only guaranteed dword aligned (e.g. 1620-byte class
// rows), which global_store_dwordx4 accepts.
template <typename T>
DEV void store_piece(T* dst, const float (&v)[8], int nvalid, bool vec_ok) {
    if constexpr (IsPair<T>::value) {                            // whole groups only (N % 8 == 0 is checked on the host)
        typedef unsigned u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
        F8 f;
#pragma unroll
        for (int e = 0; e < 8; ++e) f.v[e] = v[e];
        u32x4 hi, lo;
        pair_split8(f, hi, lo);
        if (nvalid >= 8) {
            *reinterpret_cast<u32x4_a4*>(dst) = hi;
            *(reinterpret_cast<u32x4_a4*>(dst) + 1) = lo;
        }
    } else
    if (vec_ok && nvalid >= 8) {
        if constexpr (sizeof(T) == 2) {
            typedef unsigned u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
            bf16x8 a;
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = (bf16_t)v[e];
            *reinterpret_cast<u32x4_a4*>(dst) = __builtin_bit_cast(u32x4, a);
        } else {
            typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
            *reinterpret_cast<f32x4_a4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4_a4*>(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
        }
    } else {
        if constexpr (!IsPair<T>::value) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (e < nvalid) dst[e] = from_f<T>(v[e]);
        }
    }
}

// Timing experiments on the class predict's chunk loop: variant builds only (`make variant TAG=.. UNIT=sepconv VDEFS=-DSEP_ABLATE=n`,
// never loaded by the package).  1: no logit stores  2: no max / sum-exp / score stores  4: no epilogue at all (accumulators are kept
// alive by one store that never happens)
#ifndef SEP_ABLATE
#define SEP_ABLATE 0
#endif
// FT: the channel count when known at compile time (all index arithmetic folds), 0 = read it from the arguments
// BST (class predict, bf16, dword-aligned rows): every vector-memory operation of the chunk loop is unconditional - W pieces and
// affine constants are fetched from clamped addresses and masked at their LDS store, logits and OOD scores leave through buffer
// stores whose out-of-range lanes / tail dwords get an offset beyond num_records (dropped by the hardware) - so the wait for the
// next chunk's W prefetch is a counted `vmcnt(N)` that does not drain the chunk's own stores (the exec-mask branches around them
// made it `vmcnt(0)`: every chunk waited for its 884 MB share to be acknowledged before the next could start).
template <typename T, int TH, int TW, int BN, bool OOD, int NTH, int FT, bool META = false, int NIN = 3, bool BST = false>
// (experiment switches, round 4: six waves per SIMD for the class predict - 80 registers, 2 spilled - ran 0.386 against 0.355 ms;
// 16 x 16 pixel tiles - two MFMA tiles per wave and W chunk, half the barriers per pixel - 0.459 against 0.380 ms.  Defaults = kept.)
#ifndef SEP_BST_WAVES
#define SEP_BST_WAVES 4
#endif
__global__ __launch_bounds__(NTH, NTH == 512 ? (FT == 64 && !OOD && !META && !IsPair<T>::value ? 8 : (BST && FT == 64 ? SEP_BST_WAVES : 4)) : 2) void sepconv_kernel(SepArgs p) {
    constexpr int BM = TH * TW;
    constexpr int HW_ = (TH + 2) * (TW + 2);
    constexpr int NWAVE = NTH / 64;
    constexpr int WPT = BM / (16 * NWAVE);       // 16-pixel MFMA tiles per wave
    constexpr int NT = BN / 16;                  // 16-channel tiles per chunk
    constexpr int NP = NT / 2;                   // tile pairs = 32-channel groups
    static_assert(NT % 2 == 0 && WPT >= 1, "tile shape");
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int F = FT ? FT : p.F, N = p.N;
    const int fbytes = F * (int)sizeof(T);
    constexpr int PB = OpGeom<T>::PIECE, CHB = OpGeom<T>::CHUNK;   // bytes of a lane's operand piece / of a K-chunk
    const int nkc = (fbytes + CHB - 1) / CHB;
    const int arow = nkc * CHB + 16;             // A / W row pitch in bytes
    // LDS carve (all multiples of 16): the halo tile (phases 1-2) and the W chunk (phase 3) share a region
    // bf16 runs the depthwise taps on the matrix cores (see mbconv.hip): 16 pixels x 16 bytes per operand read, so the
    // halo rows get 16 bytes of padding to spread the pixels over the LDS banks
    constexpr bool MF = IsFast<T>::value && TW == 16 && NTH == 512;
    constexpr int HROW = FC + (MF ? 16 / (int)sizeof(T) : 0);          // halo row pitch in elements (+16 bytes)
    constexpr int HALO_BYTES = HW_ * HROW * (int)sizeof(T);
    // 64-channel layers (one halo pass): the depthwise output waits in registers until every wave has read the halo and
    // then overwrites it, with the W chunk behind it -> 30 KB instead of 47 KB per workgroup, four workgroups per CU
    constexpr bool CP = MF && FT == 64 && !META;
    const int r0 = CP ? (HALO_BYTES > (BM + BN) * arow ? HALO_BYTES : (BM + BN) * arow)
                      : (HALO_BYTES > BN * arow ? HALO_BYTES : BN * arow);
    char* halo = lds;
    char* Wt = CP ? lds + BM * arow : lds;
    char* At = CP ? lds : lds + r0;
    float* dww = reinterpret_cast<float*>(CP ? lds + r0 : At + BM * arow);   // [9][F]
    float* cs = dww + 9 * F;                                  // [2][BN] scale | shift of the current chunk

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
    if (nkc * CHB > fbytes) {
        const int padb = nkc * CHB - fbytes;
        for (int i = tid; i < BM * (padb / 16); i += NTH) {
            const int row = i / (padb / 16), piece = i % (padb / 16);
            *reinterpret_cast<u32x4*>(At + row * arow + fbytes + piece * 16) = u32x4{0u, 0u, 0u, 0u};
        }
    }
    const T* ib[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {                         // inputs a node does not have alias input 0 with weight 0 (their loads stay unconditional)
        const int ii = i < p.n_in ? i : 0;
        ib[i] = reinterpret_cast<const T*>(L.in[ii].ptr) + (long long)b * L.in[ii].image_stride;
    }
    float fwm[3];                                         // per-input multiplier of the fused sum
#pragma unroll
    for (int i = 0; i < 3; ++i) fwm[i] = i < p.n_in ? ((IsFast<T>::value && p.fuse_mode == 1) ? p.fwn[i] : p.fw[i]) : 0.f;
    const bool divide = !IsFast<T>::value && p.fuse_mode == 1;   // parity mode keeps the reference's (x*w)/sum order

    // ------------------------------------------------------------------ phases 1 + 2 per 64 channels
    for (int fc0 = 0; fc0 < F; fc0 += FC) {
        const int fcn = (F - fc0) < FC ? (F - fc0) : FC;
        const int fcg = fcn / 8;
        __syncthreads();
        // HU halo items per step: their input loads are all issued before the first use (the loop has a runtime
        // trip count, so the compiler would otherwise expose one memory round trip per item)
        // bf16 tile: all 1440 items of a 64-channel pass in one batch; the 64-register variant (four workgroups per CU) keeps two
        // batches when a node fuses several inputs
        constexpr int HU = NTH == 512 ? ((FT == 64 && !OOD && !META && NIN > 1) ? 1 : ((FT == 0 && NIN > 1) ? 2 : 3)) : 2;
        for (int it0 = tid; it0 < HW_ * fcg; it0 += HU * NTH) {
            Raw8<T> xin[HU][NIN];
            unsigned okm[HU];
#pragma unroll
            for (int u = 0; u < HU; ++u) {
                // every item loads - pixels outside the map from the clamped position, surplus items the last item again - and
                // is masked afterwards: no exec-mask branch around a load, so all HU x NIN loads of the batch are in flight together
                const int it_ = it0 + NTH * u;
                const int it = it_ < HW_ * fcg ? it_ : HW_ * fcg - 1;
                const int cgh = it % fcg, hp = it / fcg;
                const int y = y0 + hp / (TW + 2) - 1, x = x0 + hp % (TW + 2) - 1;
                const int yc = y < 0 ? 0 : (y >= H ? H - 1 : y), xc = x < 0 ? 0 : (x >= W ? W - 1 : x);
                okm[u] = (it_ < HW_ * fcg && y == yc && x == xc) ? 0xFFFFFFFFu : 0u;
                const int c = fc0 + cgh * 8;
#pragma unroll
                for (int i = 0; i < NIN; ++i) xin[u][i] = fetch_input<T>(ib[i], L.in[i < p.n_in ? i : 0], yc, xc, F, c);
            }
#pragma unroll
            for (int u = 0; u < HU; ++u) {
                const int it = it0 + NTH * u;
                const int cgh = it % fcg, hp = it / fcg;
                const u32x4 km = {okm[u], okm[u], okm[u], okm[u]};
                F8 v = f8_zero();
                {
                    if (p.fuse_mode == 0) {
                        Raw8<T> r0 = xin[u][0];
                        r0.v[0] = r0.v[0] & km;
                        if constexpr (sizeof(T) == 4) r0.v[1] = r0.v[1] & km;
                        v = unpack8<T>(r0);
                    } else {
#pragma unroll
                        for (int i = 0; i < NIN; ++i) {
                            Raw8<T> ri = xin[u][i];
                            ri.v[0] = ri.v[0] & km;
                            if constexpr (sizeof(T) == 4) ri.v[1] = ri.v[1] & km;
                            const F8 xi = unpack8<T>(ri);
                            if (divide) {
#pragma unroll
                                for (int e = 0; e < 8; ++e) v.v[e] += (xi.v[e] * fwm[i]) / p.fden;
                            } else {
#pragma unroll
                                for (int e = 0; e < 8; ++e) v.v[e] = fmaf(xi.v[e], fwm[i], v.v[e]);
                            }
                        }
                    }
                    if constexpr (META) {
                        if (p.in_scale != nullptr) {
                            const F8 is = load8<float>(p.in_scale + (long long)L.in_affine_row * F + fc0 + cgh * 8);
                            const F8 it = load8<float>(p.in_shift + (long long)L.in_affine_row * F + fc0 + cgh * 8);
#pragma unroll
                            for (int e = 0; e < 8; ++e) v.v[e] = okm[u] ? v.v[e] * is.v[e] + it.v[e] : 0.f;
                        }
                    }
                    if (p.pre_act) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v.v[e] = silu_t<T>(v.v[e]);
                    }
                }
                if (it < HW_ * fcg) store8<T>(reinterpret_cast<T*>(halo) + hp * HROW + cgh * 8, v);
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
        if constexpr (MF) {
            // depthwise 3x3 on the matrix cores: wave (j, part) owns channel tile j of this pass and half of the
            // 16-pixel rows of the tile; A = diag(w[t0]) | diag(w[t1]) per pair of taps, B = the halo pixels shifted
            // by those taps; the accumulator (4 channels x 1 pixel per lane) goes straight into the A tile.
            const int frow_ = lane & 15, fp_ = lane >> 4;
            const int j = wave & 3, part = wave >> 2;
            f32x4 keep[CP ? TH / 2 : 1];
            if (16 * j < fcn) {
                const int hi = fp_ >> 1;
                const bool active = (fp_ & 1) == (frow_ >> 3);
                const int dq = (frow_ & 7) >> 1;
                Frag<T> afr[5];
#pragma unroll
                for (int pr = 0; pr < 5; ++pr) {
                    const int t = 2 * pr + hi;
                    const bool on = active && t < 9 && 16 * j + frow_ < fcn;
                    const float wv = on ? dww[(t < 9 ? t : 0) * F + fc0 + 16 * j + frow_] : 0.f;
                    const bf16_t wh = (bf16_t)wv;
                    const unsigned bits = (unsigned)__builtin_bit_cast(unsigned short, wh) << (16 * (frow_ & 1));
                    const u32x4 fr = {dq == 0 ? bits : 0u, dq == 1 ? bits : 0u, dq == 2 ? bits : 0u, dq == 3 ? bits : 0u};
                    if constexpr (IsPair<T>::value) {
                        const unsigned bl = (unsigned)__builtin_bit_cast(unsigned short, (bf16_t)(wv - (float)wh)) << (16 * (frow_ & 1));
                        const u32x4 fl = {dq == 0 ? bl : 0u, dq == 1 ? bl : 0u, dq == 2 ? bl : 0u, dq == 3 ? bl : 0u};
                        afr[pr].h = __builtin_bit_cast(bf16x8, fr);
                        afr[pr].l = __builtin_bit_cast(bf16x8, fl);
                    } else {
                        afr[pr].v = __builtin_bit_cast(bf16x8, fr);
                    }
                }
                const char* hb = halo + (frow_ * HROW + 16 * j + 8 * (fp_ & 1)) * (int)sizeof(T);
#pragma unroll
                for (int pt = 0; pt < TH / 2; ++pt) {
                    const int ty = part * (TH / 2) + pt;             // TW == 16: a pixel tile is one row of the output tile
                    const char* base = hb + ty * (TW + 2) * HROW * (int)sizeof(T);
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int pr = 0; pr < 5; ++pr) {
                        constexpr int RB = (TW + 2) * HROW * (int)sizeof(T), CB = HROW * (int)sizeof(T);
                        const int t0 = 2 * pr, t1 = 2 * pr + 1 < 9 ? 2 * pr + 1 : 0;
                        const int off = hi ? (t1 / 3) * RB + (t1 % 3) * CB : (t0 / 3) * RB + (t0 % 3) * CB;
                        mma_chunk(afr[pr], ld_frag<T>(base + off), acc);
                    }
                    if constexpr (CP) keep[pt] = acc;
                    else if (16 * j + 4 * fp_ < fcn)
                        row_store4<T>(At + (16 * ty + frow_) * arow, fc0 + 16 * j + 4 * fp_, acc);
                }
            }
            if constexpr (CP) {
                __syncthreads();                                     // every wave has read its halo rows: the A tile may overwrite them
                if (16 * j < fcn) {
#pragma unroll
                    for (int pt = 0; pt < TH / 2; ++pt) {
                        const int ty = part * (TH / 2) + pt;
                        row_store4<T>(At + (16 * ty + frow_) * arow, fc0 + 16 * j + 4 * fp_, keep[pt]);
                    }
                }
            }
        } else {
        for (int it = tid; it < BM * fcg; it += NTH) {
            const int cg = it % fcg, px = it / fcg;
            const int ty = px / TW, tx = px % TW;
            F8 acc = f8_zero();
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const F8 xv = load8<T>(reinterpret_cast<const T*>(halo) + ((ty + ky) * (TW + 2) + tx + kx) * HROW + cg * 8);
                    const F8 w = load8<float>(dww + (ky * 3 + kx) * F + fc0 + cg * 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc.v[e] = fmaf(xv.v[e], w.v[e], acc.v[e]);
                }
            store8<T>(reinterpret_cast<T*>(At + px * arow) + fc0 + cg * 8, acc);
        }
        }
    }

    // ------------------------------------------------------------------ phase 3: channel chunks
    const int frow = lane & 15, fpiece = lane >> 4;
    const int C = OOD ? p.ood_classes : 0;
    const int subs = OOD ? (C + BN - 1) / BN : 1;
    const int nchunks = OOD ? p.num_anchors * subs : (N + BN - 1) / BN;
    const float* scale = p.scale ? p.scale + (long long)L.affine_row * N : nullptr;
    const float* shift = p.shift + (long long)L.affine_row * N;
    T* out = reinterpret_cast<T*>(L.out) + (long long)b * L.out_image_stride;
    const int ppr = nkc * (CHB / 16);                  // 16-byte pieces per W row
    constexpr int WPC = (BN * (CHB / 8) + NTH - 1) / NTH;      // W pieces a thread prefetches when a row has <= 8 (two-term: 16) pieces
    u32x4 wpre[WPC];
    float cpre_s = 1.0f, cpre_t = 0.0f;                // next chunk's scale / shift for channel `tid`

    auto chunk_range = [&](int ch, int& n_begin, int& n_count) {
        if (OOD) {
            const int a = ch / subs, sc = ch % subs;
            n_begin = a * C + sc * BN;
            n_count = C - sc * BN; if (n_count > BN) n_count = BN;
        } else {
            n_begin = ch * BN;
            n_count = N - n_begin; if (n_count > BN) n_count = BN;
        }
    };
    // channel offset co of the chunk -> LDS row: tile 2J+jj, row 4*fp+r holds channel 32J + 8fp + 4jj + r
    auto lds_row = [](int co) { return 16 * (2 * (co >> 5) + ((co >> 2) & 1)) + 4 * ((co >> 3) & 3) + (co & 3); };
    const bool prefetch = BN * ppr <= NTH * WPC;
    auto w_fetch = [&](int ch) {                        // global -> registers (in flight across the epilogue)
        int n_begin, n_count;
        chunk_range(ch, n_begin, n_count);
        if constexpr (BST) {
#pragma unroll
            for (int q = 0; q < WPC; ++q) {
                const int i = tid + NTH * q < BN * ppr ? tid + NTH * q : BN * ppr - 1;
                const int co = i / ppr, piece = i % ppr;
                const bool ok = co < n_count && piece * 16 < fbytes;
                wpre[q] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.pw_w) +
                                                          (long long)(n_begin + (ok ? co : 0)) * fbytes + (ok ? piece * 16 : 0));
            }
            const int tc = tid < n_count ? tid : 0;
            cpre_t = shift[n_begin + tc];
            cpre_s = scale ? scale[n_begin + tc] : 1.0f;      // uniform
            return;
        }
#pragma unroll
        for (int q = 0; q < WPC; ++q) {
            const int i = tid + NTH * q;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (i < BN * ppr) {
                const int co = i / ppr, piece = i % ppr;
                if (co < n_count && piece * 16 < fbytes)
                    v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.pw_w) +
                                                        (long long)(n_begin + co) * fbytes + piece * 16);
            }
            wpre[q] = v;
        }
        if (tid < BN) {
            cpre_s = (scale && tid < n_count) ? scale[n_begin + tid] : 1.0f;
            cpre_t = tid < n_count ? shift[n_begin + tid] : 0.0f;
        }
    };
    if (prefetch) w_fetch(0);

    // this lane's pixels (one per 16-pixel tile of the wave)
    int pix_off[WPT];
    bool pix_in[WPT];
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
        const int prow = 16 * WPT * wave + 16 * i + frow;
        const int y = y0 + prow / TW, x = x0 + prow % TW;
        pix_in[i] = y < H && x < W;
        pix_off[i] = (y * W + x) * N;
    }
    // BST: buffer descriptors of this image's level rows (logits) and of its OOD score rows
    __amdgpu_buffer_rsrc_t ors, ers, mrs;
    constexpr int OB = IsPair<T>::value ? 4 : 2;          // BST: bytes per stored logit
    if constexpr (BST) {
        // (two-term mode: the predict conv writes float32 logits)
        ors = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(L.out) + (long long)b * L.out_image_stride * OB, 0, H * W * N * OB, 0x00020000);
        ers = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(p.ood_energy + (long long)b * p.ood_image_stride + L.ood_off), 0,
                                                H * W * p.num_anchors * 4, 0x00020000);
        mrs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(p.ood_maxlogit + (long long)b * p.ood_image_stride + L.ood_off), 0,
                                                H * W * p.num_anchors * 4, 0x00020000);
    }
    // BST: store offsets of the lane's pieces, the same for every anchor chunk (the chunk's first byte travels in the buffer
    // operation's scalar offset): 16-byte piece of group J, the three tail dwords of the one group the class count cuts (J* = C / 32,
    // the same for every lane), the lane's OOD score slot.  A lane outside the map / past the classes holds an out-of-range offset.
    int o128[BST ? WPT : 1][NP], otail[BST ? WPT : 1][3], oscore[BST ? WPT : 1];
    const int jstar = C / 32;
    if constexpr (BST) {
        constexpr int OOB = 0x7FFFFFF0;
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int sb = pix_off[i] * OB + fpiece * 8 * OB;
#pragma unroll
            for (int J = 0; J < NP; ++J) o128[i][J] = (!(SEP_ABLATE & 1) && pix_in[i] && C - (32 * J + 8 * fpiece) >= 8) ? sb + 32 * OB * J : OOB;
            const int nvt = C - (32 * jstar + 8 * fpiece);
#pragma unroll
            for (int d = 0; d < 3; ++d)                // the cut group's tail, one pair of logits (a dword; float32: two) per store
                otail[i][d] = (!(SEP_ABLATE & 1) && pix_in[i] && nvt > 0 && nvt < 8 && 2 * d < nvt) ? sb + 32 * OB * jstar + 2 * OB * d : OOB;
            oscore[i] = (pix_in[i] && fpiece == 0) ? (pix_off[i] / N) * p.num_anchors * 4 : OOB;
        }
    }
    float run_m[WPT], run_s[WPT];                      // running max / sum-exp over the sub-chunks of an anchor
    constexpr float LOG2E = 1.4426950408889634f;

    for (int ch = 0; ch < nchunks; ++ch) {
        int n_begin, n_count;
        chunk_range(ch, n_begin, n_count);
        __syncthreads();                               // halo / previous W chunk no longer read
        if constexpr (META) {
            if (ch == 0 && L.dw_out != nullptr) {       // x_pred of the reference: the depthwise output before the pointwise conv
                T* dwo = reinterpret_cast<T*>(L.dw_out) + (long long)b * L.dw_out_image_stride;
                const int ppf = fbytes / 16;
                for (int i = tid; i < BM * ppf; i += NTH) {
                    const int row = i / ppf, piece = i % ppf;
                    const int y = y0 + row / TW, x = x0 + row % TW;
                    if (y < H && x < W)
                        *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(dwo + ((long long)y * W + x) * F) + piece * 16) =
                            *reinterpret_cast<const u32x4*>(At + row * arow + piece * 16);
                }
            }
        }
        if (BST) {
            // the clamped fetch is masked here (a bitwise AND keeps the load unconditional; surplus threads repeat the last piece)
#pragma unroll
            for (int q = 0; q < WPC; ++q) {
                const int i = tid + NTH * q < BN * ppr ? tid + NTH * q : BN * ppr - 1;
                const int co = i / ppr, piece = i % ppr;
                const unsigned keep = (co < n_count && piece * 16 < fbytes) ? 0xFFFFFFFFu : 0u;
                *reinterpret_cast<u32x4*>(Wt + lds_row(co) * arow + piece * 16) = wpre[q] & u32x4{keep, keep, keep, keep};
            }
            // (channels beyond the anchor's classes get a shift of -inf: they drop out of the max and the sum-exp without a mask)
            if (tid < BN) cs[BN + tid] = tid < n_count ? cpre_t : -INFINITY;
        } else if (prefetch) {
#pragma unroll
            for (int q = 0; q < WPC; ++q) {
                const int i = tid + NTH * q;
                if (i < BN * ppr) *reinterpret_cast<u32x4*>(Wt + lds_row(i / ppr) * arow + (i % ppr) * 16) = wpre[q];
            }
            if (tid < BN) { cs[tid] = cpre_s; cs[BN + tid] = cpre_t; }
        } else {
            for (int i = tid; i < BN * ppr; i += NTH) {
                const int co = i / ppr, piece = i % ppr;
                u32x4 v = {0u, 0u, 0u, 0u};
                if (co < n_count && piece * 16 < fbytes)
                    v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.pw_w) +
                                                        (long long)(n_begin + co) * fbytes + piece * 16);
                *reinterpret_cast<u32x4*>(Wt + lds_row(co) * arow + piece * 16) = v;
            }
            if (tid < BN) {
                cs[tid] = (scale && tid < n_count) ? scale[n_begin + tid] : 1.0f;
                cs[BN + tid] = tid < n_count ? shift[n_begin + tid] : 0.0f;
            }
        }
        __syncthreads();
        if (prefetch && ch + 1 < nchunks) w_fetch(ch + 1);

        const int njp = (n_count + 31) / 32;               // 32-channel groups that hold real channels
        f32x4 acc[WPT][NT];
#pragma unroll
        for (int i = 0; i < WPT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kc = 0; kc < nkc; ++kc) {
            Frag<T> xb[WPT];
#pragma unroll
            for (int i = 0; i < WPT; ++i)
                xb[i] = ld_frag<T>(At + (16 * WPT * wave + 16 * i + frow) * arow + kc * CHB + fpiece * PB);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if (j < 2 * njp) {
                    const Frag<T> wf = ld_frag<T>(Wt + (16 * j + frow) * arow + kc * CHB + fpiece * PB);
#pragma unroll
                    for (int i = 0; i < WPT; ++i) mma_chunk(wf, xb[i], acc[i][j]);
                }
            }
        }

        if constexpr (BST) {
            // Class predict, branch-free (round 4: the general epilogue below ran 407 vector instructions per wave and anchor here,
            // most of them selects and compares around its store tails and -inf masks).  One chunk = one anchor (classes <= 96), no
            // scale, no activation: logits = acc + bias as packed adds, 16-byte stores at loop-invariant lane offsets (+ the chunk's
            // byte offset in the scalar operand), the tail dwords of the one cut group selected by the wave-uniform J*, max / sum-exp
            // over all 24 lane values (-inf biases beyond the classes), energy from the hardware log2.
            constexpr float LN2 = 0.6931471805599453f;
            const int so = n_begin * OB, sa = ch * 4;
#pragma unroll
            for (int i = 0; i < WPT; ++i) {
                if constexpr ((SEP_ABLATE & 4) != 0) {
                    f32x4 t_ = acc[i][0];
#pragma unroll
                    for (int j = 1; j < NT; ++j) t_ += acc[i][j];
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, t_[0] + t_[1] + t_[2] + t_[3]), ers, 0x7FFFFFF0, 0, 0);
                    continue;
                }
                f32x4 v[NT];
#pragma unroll
                for (int J = 0; J < NP; ++J) {
                    const int cb = 32 * J + 8 * fpiece;
                    v[2 * J] = acc[i][2 * J] + *reinterpret_cast<const f32x4*>(cs + BN + cb);
                    v[2 * J + 1] = acc[i][2 * J + 1] + *reinterpret_cast<const f32x4*>(cs + BN + cb + 4);
                }
                if constexpr (IsPair<T>::value) {
                    // float32 logits: two 16-byte stores per group (an out-of-range offset + 16 is out of range too)
#pragma unroll
                    for (int J = 0; J < NP; ++J) {
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[2 * J]), ors, o128[i][J], so, 0);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[2 * J + 1]), ors, o128[i][J] + 16, so, 0);
                    }
                    f32x4 ta = v[0], tb = v[1];
#pragma unroll
                    for (int J = 1; J < NP; ++J) { ta = jstar == J ? v[2 * J] : ta; tb = jstar == J ? v[2 * J + 1] : tb; }      // wave-uniform select
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, f32x2{ta[0], ta[1]}), ors, otail[i][0], so, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, f32x2{ta[2], ta[3]}), ors, otail[i][1], so, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, f32x2{tb[0], tb[1]}), ors, otail[i][2], so, 0);
                } else {
                    u32x4 pk[NP];
#pragma unroll
                    for (int J = 0; J < NP; ++J) {
                        const bf16x8 a8 = {(bf16_t)v[2 * J][0], (bf16_t)v[2 * J][1], (bf16_t)v[2 * J][2], (bf16_t)v[2 * J][3],
                                           (bf16_t)v[2 * J + 1][0], (bf16_t)v[2 * J + 1][1], (bf16_t)v[2 * J + 1][2], (bf16_t)v[2 * J + 1][3]};
                        pk[J] = __builtin_bit_cast(u32x4, a8);
                        __builtin_amdgcn_raw_buffer_store_b128(pk[J], ors, o128[i][J], so, 0);
                    }
                    u32x4 pt = pk[0];
#pragma unroll
                    for (int J = 1; J < NP; ++J) pt = jstar == J ? pk[J] : pt;          // wave-uniform select
#pragma unroll
                    for (int d = 0; d < 3; ++d) __builtin_amdgcn_raw_buffer_store_b32(pt[d], ors, otail[i][d], so, 0);
                }
                if constexpr ((SEP_ABLATE & 2) != 0) continue;
                float m = -INFINITY;
#pragma unroll
                for (int j = 0; j < NT; ++j) m = fmaxf(m, fmaxf(fmaxf(v[j][0], v[j][1]), fmaxf(v[j][2], v[j][3])));
                m = fmaxf(m, __shfl_xor(m, 16, 64));
                m = fmaxf(m, __shfl_xor(m, 32, 64));
                const float tl = m * LOG2E;
                f32x2 s2 = {0.f, 0.f};
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const f32x2 e0 = f32x2{v[j][0], v[j][1]} * LOG2E - tl, e1 = f32x2{v[j][2], v[j][3]} * LOG2E - tl;
                    s2 += f32x2{__builtin_amdgcn_exp2f(e0[0]), __builtin_amdgcn_exp2f(e0[1])};      // exp2(-inf) = 0
                    s2 += f32x2{__builtin_amdgcn_exp2f(e1[0]), __builtin_amdgcn_exp2f(e1[1])};
                }
                float ssum = s2[0] + s2[1];
                ssum += __shfl_xor(ssum, 16, 64);
                ssum += __shfl_xor(ssum, 32, 64);
                const float energy = IsPair<T>::value ? -(m + logf(ssum)) : -(m + LN2 * __builtin_amdgcn_logf(ssum));
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, energy), ers, oscore[i], sa, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, m), mrs, oscore[i], sa, 0);
            }
            continue;
        }
        if constexpr ((SEP_ABLATE & 8) != 0 && !OOD && !META) {      // 8: the general kernel without its epilogue (timing only)
            f32x4 t_ = acc[0][0];
#pragma unroll
            for (int i = 0; i < WPT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) t_ += acc[i][j];
            if (t_[0] + t_[1] + t_[2] + t_[3] == 12345.678f) *reinterpret_cast<unsigned*>(out) = 1u;
            continue;
        }
        // Epilogue in registers: per 32-channel group J this lane holds channels [32J + 8fp, +8) of its pixel
        float st1[META ? NP : 1][8], st2[META ? NP : 1][8];
        if constexpr (META) {
#pragma unroll
            for (int J = 0; J < NP; ++J)
#pragma unroll
                for (int e = 0; e < 8; ++e) { st1[J][e] = 0.f; st2[J][e] = 0.f; }
        }
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            float m = -INFINITY, ssum = 0.f;
            float vals[NP][8];
#pragma unroll
            for (int J = 0; J < NP; ++J) {
                if (J < njp) {
                    const int cb = 32 * J + 8 * fpiece;
                    const f32x4 t0 = *reinterpret_cast<const f32x4*>(cs + BN + cb);
                    const f32x4 t1 = *reinterpret_cast<const f32x4*>(cs + BN + cb + 4);
                    f32x4 s0 = {1.f, 1.f, 1.f, 1.f}, s1 = {1.f, 1.f, 1.f, 1.f};
                    if (scale) {
                        s0 = *reinterpret_cast<const f32x4*>(cs + cb);
                        s1 = *reinterpret_cast<const f32x4*>(cs + cb + 4);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float a0 = fmaf(acc[i][2 * J][r], s0[r], t0[r]);
                        float a1 = fmaf(acc[i][2 * J + 1][r], s1[r], t1[r]);
                        if (p.post_act) { a0 = silu_t<T>(a0); a1 = silu_t<T>(a1); }
                        vals[J][r] = a0; vals[J][4 + r] = a1;
                    }
                    const bool group_full = 32 * J + 32 <= n_count;     // uniform: only the last group of a chunk has a tail
                    const int nvalid = group_full ? 8 : n_count - cb;
                    if constexpr (BST) {
                        // 8 channels = 16 bytes at a dword-aligned offset of this level's rows; a lane outside the map, a piece past
                        // the chunk's channels or a tail dword gets the out-of-range offset and is dropped
                        constexpr int OOB = 0x7FFFFFF0;
                        const int boff = (pix_off[i] + n_begin + cb) * 2;
                        bf16x8 a8;
#pragma unroll
                        for (int e = 0; e < 8; ++e) a8[e] = (bf16_t)vals[J][e];
                        const u32x4 pk = __builtin_bit_cast(u32x4, a8);
                        __builtin_amdgcn_raw_buffer_store_b128(pk, ors, (pix_in[i] && nvalid >= 8) ? boff : OOB, 0, 0);
#pragma unroll
                        for (int d = 0; d < 3; ++d)             // tail of the chunk: up to three whole dwords (the class count is even)
                            __builtin_amdgcn_raw_buffer_store_b32(pk[d], ors, (pix_in[i] && nvalid < 8 && 2 * d < nvalid) ? boff + 4 * d : OOB, 0, 0);
                    } else
#ifndef SEP_ABLATE_NOSTORE            /* variant build for timing only (make variant ... VDEFS=-DSEP_ABLATE_NOSTORE) */
                    if (pix_in[i] && nvalid > 0) {
#else
                    if (pix_in[i] && nvalid > 0 && !OOD) {
#endif
                        if (IsFast<T>::value && p.out_f32)
                            store_piece<float>(reinterpret_cast<float*>(L.out) + (long long)b * L.out_image_stride + pix_off[i] + n_begin + cb,
                                               vals[J], nvalid, true);
                        else
                            store_piece<T>(out + pix_off[i] + n_begin + cb, vals[J], nvalid, p.vec_ok != 0);
                    }
                    if constexpr (META) {
                        if (p.stat_partial != nullptr && pix_in[i]) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                float q = vals[J][e];
                                if constexpr (!IsPair<T>::value) q = to_f<T>(from_f<T>(vals[J][e]));   // statistics of what the next layer reads
                                st1[J][e] += q; st2[J][e] += q * q;
                            }
                        }
                    }
                    if constexpr (OOD) {
                        if (!group_full) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) if (e >= nvalid) vals[J][e] = -INFINITY;
                        }
                        m = fmaxf(m, fmaxf(fmaxf(fmaxf(vals[J][0], vals[J][1]), fmaxf(vals[J][2], vals[J][3])),
                                           fmaxf(fmaxf(vals[J][4], vals[J][5]), fmaxf(vals[J][6], vals[J][7]))));
                    }
                }
            }
            if constexpr (OOD) {
                m = fmaxf(m, __shfl_xor(m, 16, 64));
                m = fmaxf(m, __shfl_xor(m, 32, 64));
                const float tm = (m == -INFINITY) ? 0.f : m;
#pragma unroll
                for (int J = 0; J < NP; ++J) {
                    if (J < njp) {
                        if constexpr (IsFast<T>::value) {
                            const float tl = tm * LOG2E;
#pragma unroll
                            for (int e = 0; e < 8; ++e) ssum += __builtin_amdgcn_exp2f(fmaf(vals[J][e], LOG2E, -tl));   // exp2(-inf) = 0
                        } else {
#pragma unroll
                            for (int e = 0; e < 8; ++e) ssum += (vals[J][e] == -INFINITY) ? 0.f : expf(vals[J][e] - tm);
                        }
                    }
                }
                ssum += __shfl_xor(ssum, 16, 64);
                ssum += __shfl_xor(ssum, 32, 64);
                const int sc = ch % subs;
                float pm = sc == 0 ? -INFINITY : run_m[i], ps = sc == 0 ? 0.f : run_s[i];
                if (m != -INFINITY) {
                    const float nm = fmaxf(pm, m);
                    ps = (pm == -INFINITY ? 0.f : ps * exp_t<T>(pm - nm)) + ssum * exp_t<T>(m - nm);
                    pm = nm;
                }
                run_m[i] = pm; run_s[i] = ps;
                if constexpr (BST) {
                    constexpr int OOB = 0x7FFFFFF0;
                    const int a = ch / subs;
                    const int eo = (sc == subs - 1 && pix_in[i] && fpiece == 0) ? ((pix_off[i] / N) * p.num_anchors + a) * 4 : OOB;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, -(pm + logf(ps))), ers, eo, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pm), mrs, eo, 0, 0);
                } else
                if (sc == subs - 1 && pix_in[i] && fpiece == 0) {
                    const int a = ch / subs;
                    const long long idx = (long long)b * p.ood_image_stride + L.ood_off +
                                          (long long)(pix_off[i] / N) * p.num_anchors + a;
                    p.ood_energy[idx] = -(pm + logf(ps));
                    p.ood_maxlogit[idx] = pm;
                }
            }
        }
        if constexpr (META) {
            if (p.stat_partial != nullptr) {
                // per-channel sums over this workgroup's pixels: 16 lanes (pixels) by shuffles, the waves through LDS in a
                // fixed order, one row of the partial table per workgroup
                float* sst = cs + 2 * BN;                      // [NWAVE][2][BN]
#pragma unroll
                for (int J = 0; J < NP; ++J)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float a = st1[J][e], q = st2[J][e];
#pragma unroll
                        for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o, 64); q += __shfl_xor(q, o, 64); }
                        if (frow == 0) { sst[(wave * 2 + 0) * BN + 32 * J + 8 * fpiece + e] = a; sst[(wave * 2 + 1) * BN + 32 * J + 8 * fpiece + e] = q; }
                    }
                __syncthreads();
                if (tid < 2 * BN) {
                    const int which = tid / BN, n = tid % BN;
                    float tot = 0.f;
                    for (int w = 0; w < NWAVE; ++w) tot += sst[(w * 2 + which) * BN + n];
                    if (n < n_count)
                        p.stat_partial[(((long long)b * p.tiles_total + blockIdx.x) * 2 + which) * N + n_begin + n] = tot;
                }
            }
        }
    }
}

template <typename T, int TH, int TW, int BN>
size_t sep_lds_bytes(int F, bool cp = false) {
    constexpr int BM = TH * TW;
    constexpr int HW_ = (TH + 2) * (TW + 2);
    const int nkc = (F * (int)sizeof(T) + OpGeom<T>::CHUNK - 1) / OpGeom<T>::CHUNK;
    const int arow = nkc * OpGeom<T>::CHUNK + 16;
    const size_t halo = (size_t)HW_ * (FC + (IsFast<T>::value && TW == 16 ? 16 / (int)sizeof(T) : 0)) * sizeof(T);
    const size_t wt = (size_t)BN * arow;
    if (cp) return (halo > wt + (size_t)BM * arow ? halo : wt + (size_t)BM * arow) + (size_t)9 * F * 4 + (size_t)2 * BN * 4;
    return (halo > wt ? halo : wt) + (size_t)BM * arow + (size_t)9 * F * 4 + (size_t)2 * BN * 4;
}

template <typename T, int TH, int TW, int BN, bool OOD, int NTH, int FT, bool META = false, int NIN = 3, bool BST = false>
int launch_sep(hipStream_t st, SepArgs& a, int B) {
    int tiles = 0;
    for (int i = 0; i < a.nlevels; ++i) {
        a.lv[i].tiles_x = (a.lv[i].W + TW - 1) / TW;
        a.lv[i].tiles_y = (a.lv[i].H + TH - 1) / TH;
        a.lv[i].tile_begin = tiles;
        tiles += a.lv[i].tiles_x * a.lv[i].tiles_y;
    }
    const size_t lds = sep_lds_bytes<T, TH, TW, BN>(a.F, IsFast<T>::value && TW == 16 && NTH == 512 && FT == 64 && !META) + (META ? (size_t)(NTH / 64) * 2 * BN * 4 : 0);   // + statistics scratch
    if (lds > 160 * 1024) return EFFDET_EINVAL;
    a.tiles_total = tiles;
    auto kern = sepconv_kernel<T, TH, TW, BN, OOD, NTH, FT, META, NIN, BST>;
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

#ifndef SEP_BST_TH
#define SEP_BST_TH 8
#endif
constexpr int SEP_BST_TH_ = SEP_BST_TH;
template <typename T, int TH, int TW, int NTH, int FT>
int dispatch_sep_f(hipStream_t st, SepArgs& a, int B) {
    if (a.ood_classes > 0) {
        // class predict: one 96-channel chunk per anchor; widths whose 96-row W chunk no longer fits beside the A tile (float32
        // d5, 288 channels) take 64-row chunks - two sub-chunks per anchor, the running max / sum-exp carries over
        if constexpr (FT == 0) {
            if (sep_lds_bytes<T, TH, TW, 96>(a.F) > 160 * 1024) return launch_sep<T, TH, TW, 64, true, NTH, FT>(st, a, B);
        }
        if constexpr (IsPair<T>::value && NTH == 512) {
            // two-term mode (float32 logits): the same branch-free chunk loop
            if (a.out_f32 && a.ood_classes % 2 == 0 && a.ood_classes <= 96 && a.F <= 64 && a.scale == nullptr && !a.post_act)
                return launch_sep<T, SEP_BST_TH_, TW, 96, true, NTH, FT, false, 3, true>(st, a, B);
        }
        if constexpr (sizeof(T) == 2 && NTH == 512) {
            // bf16 with dword-aligned class rows (an even class count): the branch-free chunk loop
            // (and a W chunk every thread can prefetch in two pieces: up to 64 channels - wider heads stay on the general loop)
            // (no BN scale, no activation behind the predict conv: the branch-free epilogue below builds on all of that)
            if (a.vec_ok && !a.out_f32 && a.ood_classes % 2 == 0 && a.ood_classes <= 96 && a.F <= 64 && a.scale == nullptr && !a.post_act)
                return launch_sep<T, SEP_BST_TH, TW, 96, true, NTH, FT, false, 3, true>(st, a, B);
        }
        return launch_sep<T, TH, TW, 96, true, NTH, FT>(st, a, B);
    }
    // head layers read one input: a variant without the registers of the other two
    if (FT == 64 && a.n_in == 1) return launch_sep<T, TH, TW, 64, false, NTH, FT, false, 1>(st, a, B);
    return launch_sep<T, TH, TW, 64, false, NTH, FT>(st, a, B);
}

// the BiFPN widths of tf_efficientdet_d0..d4 get compile-time channel counts; anything else the generic kernel
template <typename T, int TH, int TW, int NTH>
int dispatch_sep(hipStream_t st, SepArgs& a, int B) {
    switch (a.F) {
        case 64:  return dispatch_sep_f<T, TH, TW, NTH, 64>(st, a, B);
        case 88:  return dispatch_sep_f<T, TH, TW, NTH, 88>(st, a, B);
        case 112: return dispatch_sep_f<T, TH, TW, NTH, 112>(st, a, B);
        case 160: return dispatch_sep_f<T, TH, TW, NTH, 160>(st, a, B);
        case 224: return dispatch_sep_f<T, TH, TW, NTH, 224>(st, a, B);
        default:  return dispatch_sep_f<T, TH, TW, NTH, 0>(st, a, B);
    }
}

}  // namespace

// Flat C-ABI descriptor (mirrors SepArgs; arrays are per level / per input)
namespace {
struct MetaExtra {
    const float* in_scale; const float* in_shift; const int* in_affine_row;
    float* stat_partial;
    void* const* dw_out; const long long* dw_out_image_stride;
};
}

static int sepconv_common(
    const MetaExtra* meta,
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
    if (nlevels < 1 || nlevels > 5 || n_in < 1 || n_in > 3 || B <= 0) return EFFDET_EINVAL;
    if (!level_hw || !in_ptr || !in_image_stride || !in_hw || !in_mode || !dw_w || !pw_w || !shift || !affine_row ||
        !out_ptr || !out_image_stride) return EFFDET_EINVAL;
    const int sym = take_pad_flag(dtype);                // padding convention of the on-the-fly 3x3 / s2 max pool (mode 2 inputs)
    // dtype 3 (= 1 | 2): bfloat16 compute, float32 outputs; dtype 6 (= 2 | 4): two-term bf16 compute, float32 outputs
    if (dtype != 0 && dtype != 1 && dtype != 2 && dtype != 3 && dtype != 6) return EFFDET_EINVAL;
    const int out_f32 = (dtype == 3 || dtype == 6) ? 1 : 0;
    dtype = dtype == 3 ? 1 : (dtype == 6 ? 2 : dtype);
    if (dtype == 2 && (meta || (!out_f32 && N % 8))) return EFFDET_EINVAL;      // two-term outputs: whole 8-channel groups; no MetaHead form
    if (F <= 0 || F % 8 || N <= 0 || fuse_mode < 0 || fuse_mode > 2) return EFFDET_EINVAL;
    if (fuse_mode != 0 && !fuse_w) return EFFDET_EINVAL;
    if (fuse_mode == 0 && n_in != 1) return EFFDET_EINVAL;
    if (ood_classes > 0) {
        if (!ood_energy || !ood_maxlogit || !ood_level_off || num_anchors <= 0 || num_anchors * ood_classes != N) return EFFDET_EINVAL;
    }
    SepArgs a;
    a.nlevels = nlevels; a.n_in = n_in; a.fuse_mode = fuse_mode; a.fden = fuse_den;
    for (int i = 0; i < 3; ++i) a.fw[i] = (fuse_mode != 0 && i < n_in) ? fuse_w[i] : 0.f;
    for (int i = 0; i < 3; ++i) a.fwn[i] = fuse_mode == 1 ? a.fw[i] / fuse_den : a.fw[i];
    const size_t esz = (dtype == 0 || out_f32) ? 4 : 2;
    a.out_f32 = out_f32;
    a.vec_ok = ((size_t)N * esz) % 4 == 0 && (ood_classes <= 0 || ((size_t)ood_classes * esz) % 4 == 0);
    a.pre_act = pre_act; a.post_act = post_act; a.dw_w = dw_w; a.pw_w = pw_w; a.scale = scale; a.shift = shift;
    a.F = F; a.N = N; a.ood_classes = ood_classes > 0 ? ood_classes : 0; a.num_anchors = num_anchors;
    a.ood_energy = ood_energy; a.ood_maxlogit = ood_maxlogit; a.ood_image_stride = ood_image_stride;
    a.in_scale = meta ? meta->in_scale : nullptr; a.in_shift = meta ? meta->in_shift : nullptr;
    a.stat_partial = meta ? meta->stat_partial : nullptr; a.tiles_total = 0;
    if (meta && ((meta->in_scale != nullptr) != (meta->in_shift != nullptr) || (meta->in_scale && !meta->in_affine_row))) return EFFDET_EINVAL;
    for (int l = 0; l < nlevels; ++l) {
        SepLevel& L = a.lv[l];
        L.in_affine_row = (meta && meta->in_affine_row) ? meta->in_affine_row[l] : 0;
        L.dw_out = (meta && meta->dw_out) ? meta->dw_out[l] : nullptr;
        L.dw_out_image_stride = (meta && meta->dw_out_image_stride) ? meta->dw_out_image_stride[l] : 0;
        if (L.dw_out && reinterpret_cast<uintptr_t>(L.dw_out) % 16) return EFFDET_EINVAL;
        L.H = level_hw[2 * l]; L.W = level_hw[2 * l + 1];
        if (L.H <= 0 || L.W <= 0) return EFFDET_EINVAL;
        L.affine_row = affine_row[l];
        L.out = out_ptr[l]; L.out_image_stride = out_image_stride[l];
        L.ood_off = (ood_classes > 0) ? ood_level_off[l] : 0;
        if (!L.out) return EFFDET_EINVAL;
        if (reinterpret_cast<uintptr_t>(L.out) % 16 || ((size_t)L.out_image_stride * esz) % 4) a.vec_ok = 0;
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
                I.pad_t = pad_before(I.H, 3, 2, sym); I.pad_l = pad_before(I.W, 3, 2, sym);
            } else return EFFDET_EINVAL;
        }
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (meta) {                                  // MetaHead layers: generic-width kernels with the extra inputs / outputs
        if (ood_classes > 0 || fuse_mode != 0 || out_f32) return EFFDET_EINVAL;
        return dtype == 0 ? launch_sep<float, 8, 8, 64, false, 256, 0, true>(st, a, B)
                          : launch_sep<bf16_t, 8, 16, 64, false, 512, 0, true>(st, a, B);
    }
    if (dtype == 0) return dispatch_sep<float, 8, 8, 256>(st, a, B);
    if (dtype == 2) {
        for (int l = 0; l < nlevels; ++l) {                      // 16-byte aligned groups
            if (!out_f32 && (reinterpret_cast<uintptr_t>(a.lv[l].out) % 16 || a.lv[l].out_image_stride % 4)) return EFFDET_EINVAL;
            for (int i = 0; i < n_in; ++i)
                if (reinterpret_cast<uintptr_t>(a.lv[l].in[i].ptr) % 16 || a.lv[l].in[i].image_stride % 4) return EFFDET_EINVAL;
        }
        return dispatch_sep<bf16p_t, 8, 16, 512>(st, a, B);
    }
    // bf16: 512 threads per 8x16 tile - the LDS footprint allows two workgroups per CU, so four waves per SIMD
    // share the VALU-heavy halo and epilogue phases
    return dispatch_sep<bf16_t, 8, 16, 512>(st, a, B);
}

extern "C" int effdet_sepconv_fused(
    void* stream, int dtype, int B, int nlevels, const int* level_hw, int n_in,
    const void* const* in_ptr, const long long* in_image_stride, const int* in_hw, const int* in_mode,
    int fuse_mode, const float* fuse_w, float fuse_den, int pre_act,
    const float* dw_w, const void* pw_w, const float* scale, const float* shift, const int* affine_row,
    int post_act, int F, int N, void* const* out_ptr, const long long* out_image_stride,
    int ood_classes, int num_anchors, float* ood_energy, float* ood_maxlogit,
    long long ood_image_stride, const long long* ood_level_off) {
    EFFDET_ENTER();
    return sepconv_common(nullptr, stream, dtype, B, nlevels, level_hw, n_in, in_ptr, in_image_stride, in_hw, in_mode,
                          fuse_mode, fuse_w, fuse_den, pre_act, dw_w, pw_w, scale, shift, affine_row, post_act, F, N,
                          out_ptr, out_image_stride, ood_classes, num_anchors, ood_energy, ood_maxlogit, ood_image_stride, ood_level_off);
}

// One MetaHead layer for all levels (effdet/efficientdet.py:655-676): the previous layer's batch-statistics BN arrives as
// in_scale / in_shift [rows][F] (row in_affine_row[level]; null for the first layer) applied before the SiLU, the conv
// output (+ bias) is written raw, and its per-channel sums / sums of squares go to stat_partial
// [B][effdet_sepconv_tiles][2][N] for effdet_bn_batch_stats.  dw_out (optional, per level [B, H*W, F]) receives the
// depthwise output (`x_pred`, what ret_activs returns).
extern "C" int effdet_sepconv_meta(
    void* stream, int dtype, int B, int nlevels, const int* level_hw,
    const void* const* in_ptr, const long long* in_image_stride,
    const float* in_scale, const float* in_shift, const int* in_affine_row, int pre_act,
    const float* dw_w, const void* pw_w, const float* bias, int F, int N,
    void* const* out_ptr, const long long* out_image_stride,
    float* stat_partial, void* const* dw_out, const long long* dw_out_image_stride) {
    EFFDET_ENTER();
    if (nlevels < 1 || nlevels > 5 || !level_hw) return EFFDET_EINVAL;
    int in_hw[10], in_mode[5], arow[5];
    for (int l = 0; l < nlevels; ++l) { in_hw[2 * l] = level_hw[2 * l]; in_hw[2 * l + 1] = level_hw[2 * l + 1]; in_mode[l] = 0; arow[l] = 0; }
    MetaExtra m{in_scale, in_shift, in_affine_row, stat_partial, dw_out, dw_out_image_stride};
    return sepconv_common(&m, stream, dtype, B, nlevels, level_hw, 1, in_ptr, in_image_stride, in_hw, in_mode,
                          0, nullptr, 1.f, pre_act, dw_w, pw_w, nullptr, bias, arow, 0, F, N,
                          out_ptr, out_image_stride, 0, 0, nullptr, nullptr, 0, nullptr);
}

// Tiles per image of a multi-level launch (rows of stat_partial per image) and the first tile of every level
extern "C" int effdet_sepconv_tiles(int dtype, int nlevels, const int* level_hw, int* level_tile_begin) {
    if (nlevels < 1 || nlevels > 5 || !level_hw || dtype < 0 || dtype > 2) return EFFDET_EINVAL;
    const int TH = 8, TW = dtype == 0 ? 8 : 16;
    int tiles = 0;
    for (int l = 0; l < nlevels; ++l) {
        if (level_tile_begin) level_tile_begin[l] = tiles;
        tiles += ((level_hw[2 * l + 1] + TW - 1) / TW) * ((level_hw[2 * l] + TH - 1) / TH);
    }
    return tiles;
}

namespace {
struct BnStatArgs {
    const float* partial; int B, tiles_total, N;
    int nlevels; int tile_begin[6]; int count[5];       // pixels per level (B*H*W)
    const float* weight; const float* bias; int param_row[5];
    float eps; float* out_scale; float* out_shift;       // [nlevels][N]
};

// F.batch_norm(training=True) of one (level, layer): mean / biased variance over B*H*W from the per-workgroup sums
// (fixed order, double accumulation), folded with the affine parameters into scale / shift for the next layer's load
__global__ __launch_bounds__(256) void bn_batch_stats_kernel(BnStatArgs p) {
    const int l = blockIdx.x;
    for (int n = threadIdx.x; n < p.N; n += 256) {
        double s = 0.0, q = 0.0;
        for (int b = 0; b < p.B; ++b)
            for (int t = p.tile_begin[l]; t < p.tile_begin[l + 1]; ++t) {
                const float* row = p.partial + (((long long)b * p.tiles_total + t) * 2) * p.N;
                s += (double)row[n]; q += (double)row[p.N + n];
            }
        const double mean = s / p.count[l];
        double var = q / p.count[l] - mean * mean; if (var < 0.0) var = 0.0;
        const float w = p.weight[(long long)p.param_row[l] * p.N + n], bb = p.bias[(long long)p.param_row[l] * p.N + n];
        const float sc = w / sqrtf((float)var + p.eps);
        p.out_scale[(long long)l * p.N + n] = sc;
        p.out_shift[(long long)l * p.N + n] = bb - (float)mean * sc;
    }
}
}  // namespace

extern "C" int effdet_bn_batch_stats(void* stream, int dtype, const float* partial, int B, int nlevels, const int* level_hw, int N,
                                     const float* weight, const float* bias, const int* param_row, float eps,
                                     float* out_scale, float* out_shift) {
    EFFDET_ENTER();
    if (!partial || !level_hw || !weight || !bias || !param_row || !out_scale || !out_shift || B <= 0 || N <= 0 || nlevels < 1 || nlevels > 5) return EFFDET_EINVAL;
    BnStatArgs a; a.partial = partial; a.B = B; a.N = N; a.nlevels = nlevels; a.weight = weight; a.bias = bias; a.eps = eps;
    a.out_scale = out_scale; a.out_shift = out_shift;
    int tb[5];
    const int tiles = effdet_sepconv_tiles(dtype, nlevels, level_hw, tb);
    if (tiles <= 0) return EFFDET_EINVAL;
    a.tiles_total = tiles;
    for (int l = 0; l < nlevels; ++l) { a.tile_begin[l] = tb[l]; a.count[l] = B * level_hw[2 * l] * level_hw[2 * l + 1]; a.param_row[l] = param_row[l]; }
    a.tile_begin[nlevels] = tiles;
    hipLaunchKernelGGL(bn_batch_stats_kernel, dim3(nlevels), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    return effdet_check_launch();
}
