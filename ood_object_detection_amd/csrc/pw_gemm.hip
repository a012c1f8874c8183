// Pointwise (1x1) convolution as an MFMA GEMM with fused per-channel affine (folded BN / bias),
// optional SiLU, optional SE gate on the input channels and optional residual add.
//
//   C[m, n] = act( (sum_k A[m,k] * gate[img(m),k] * W[n,k]) * scale[n] + shift[n] ) + R[m,n]      act: none / SiLU / ReLU
//
// Replaces the 1x1 `create_conv2d` + BatchNorm2d (+ Swish) call sites of the reference:
//   timm EfficientNet conv_pw / conv_pwl (effdet/efficientdet.py:837) and the BiFPN lateral
//   ConvBnAct2d (effdet/efficientdet.py:155-158).
//
// A is the NHWC activation viewed as [M = B*H*W, K]; W is the conv weight [N = Cout, K = Cin].
// These GEMMs have a tiny N (16..320) and stream a large A exactly once, so the design is built around
// keeping many A bytes in flight:
//   * a workgroup owns 128 pixels of ONE image x BN output channels; each of its 4 waves owns 32 pixels;
//   * the activations never touch LDS: every lane loads its 16-byte MFMA operand pieces straight from
//     HBM, PF pipeline stages (128 bytes of K each) ahead of their use;
//   * the (small, L2-resident) weight chunk goes through a double-buffered LDS image; the SE gate is folded
//     into it on the way in (W[n,k] * gate[img,k]), so no vector instruction ever touches A;
//   * accumulator rows are output channels, columns pixels, and the W rows are permuted in LDS so that a lane
//     ends up with 8 consecutive channels of one pixel per pair of tiles: affine / SiLU / residual happen in
//     registers and every lane stores 16 bytes straight to HBM.
#include "common.h"

namespace {

struct PwArgs {
    const void* A; long long M; int K;
    const void* W; int N;
    const float* scale; const float* shift; int act;
    const void* res;
    const float* gate; int rows_per_image;
    void* C; long long c_image_stride; long long ldc;
    int tiles_per_image, n_tiles, vec_ok;
};

template <int V> struct PwInt { static constexpr int value = V; };

#ifndef PW_PF_SMALL
#define PW_PF_SMALL 2            /* measured on d0 blocks 5.x: four stages 0.042 ms, two stages 0.035 ms */
#endif
constexpr int PW_PIX = 128;                      // pixels per workgroup: 4 waves x 2 MFMA tiles, or 8 waves x 1 (small maps)
#ifndef PW_KCH_SMALL
#define PW_KCH_SMALL 2
#endif

// PT 16-pixel tiles per wave, NTH threads: (2, 256) normally; (1, 512) when the launch has fewer than two workgroups per
// CU (20x20 maps) - twice the waves per SIMD to hide the per-stage latencies, at the price of reading each W
// fragment from LDS once per 16 instead of once per 32 pixels
template <typename T, int BN, int PT, int NTH, bool GATED>
DEV void pw_gemm_body(const PwArgs& p, const int bid) {
    // 64-byte K-chunks per pipeline stage (PW_KCH_SMALL: the 512-thread small-map form; 4 in a variant build for A/B timing)
    // (two-term bf16: ONE 128-byte chunk = 32 K values in two terms per stage - the same bytes per pixel and stage, and the same
    // number of operand registers in the A ring, as the two 64-byte chunks of the other dtypes)
    constexpr bool PAIR = IsPair<T>::value;
    constexpr int KCH = PAIR ? 1 : ((PT == 1 && NTH == 512) ? PW_KCH_SMALL : 2);
    constexpr int EPC = VecTraits<T>::EPC;          // elements per operand piece
    constexpr int KPC = OpGeom<T>::KPC;             // elements per K-chunk
    constexpr int PB = OpGeom<T>::PIECE, CB = OpGeom<T>::CHUNK;   // bytes of a lane's operand piece / of a K-chunk (16 / 64; two-term: 32 / 128)
    constexpr int NT = BN / 16, NP = NT / 2;
    static_assert(NT % 2 == 0, "tile pairs");
    // A stages (128 bytes of K per pixel) in flight ahead of the one being multiplied: as many as the registers
    // left over by the accumulators allow - the late layers are latency bound, not bandwidth bound
    // two stages everywhere; PW_PF_SMALL = 4 (a variant build) deepens the ring of the 512-thread small-map form - measured slower
    constexpr int PF = (PT == 1 && NTH == 512) ? PW_PF_SMALL : 2;
    constexpr int ROWB = KCH * CB + 16;             // bytes per LDS row: one stage of K + 16 pad
    constexpr int W_BYTES = BN * ROWB;
    __shared__ __attribute__((aligned(16))) char lds[2 * W_BYTES];
    extern __shared__ __attribute__((aligned(16))) float gate_lds[];   // GATED: this image's SE gate, all K channels (dynamic: K floats)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 15, fpiece = lane >> 4;
    const int K = p.K, N = p.N;
    const int nt = bid % p.n_tiles;
    const int mt = bid / p.n_tiles;
    const int img = mt / p.tiles_per_image;
    const int pix0 = (mt - img * p.tiles_per_image) * PW_PIX;
    const int n0 = nt * BN;
    const int n_count = (N - n0) < BN ? (N - n0) : BN;
    const long long pitch = (long long)K * (long long)sizeof(T);
    const int nkc = (K + KPC - 1) / KPC;            // K-chunks
    const int nst = (nkc + KCH - 1) / KCH;          // pipeline stages
    const int kbytes = K * (int)sizeof(T);

    // ---- this lane's two pixels (B operand columns): pointers into A, clamped inside the image
    const char* arow[PT];
    int pix[PT];
    bool pix_ok[PT];
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        pix[i] = pix0 + 16 * PT * wave + 16 * i + frow;
        pix_ok[i] = pix[i] < p.rows_per_image;
        const long long m = (long long)img * p.rows_per_image + (pix_ok[i] ? pix[i] : 0);
        arow[i] = reinterpret_cast<const char*>(p.A) + m * pitch;
    }
    // A ring: PF + 1 stages x KCH chunks x 2 pixel tiles, straight from memory into MFMA operand registers
    Frag<T> areg[PF + 1][KCH][PT];
    auto a_load = [&](int stg, int slot) {
#pragma unroll
        for (int sub = 0; sub < KCH; ++sub) {
            const int off = (stg * KCH + sub) * CB + fpiece * PB;
            // a lane past the end of K re-reads the row start (always in bounds, finite activations): its products meet the zeroed
            // W pieces of the K tail.  No select on the loaded value here - it would make the stage wait for its own prefetch.
            const int offc = off < kbytes ? off : 0;
#pragma unroll
            for (int i = 0; i < PT; ++i) areg[slot][sub][i] = ld_frag<T>(arow[i] + offc);
        }
    };

    // ---- W staging: BN rows x 8 pieces per stage; gate folded in
    const char* Wb = reinterpret_cast<const char*>(p.W);
    const float* gate = GATED ? p.gate + (long long)img * K : nullptr;
    constexpr int PPR = KCH * 4;
    constexpr int W_PER_THREAD = (BN * PPR + NTH - 1) / NTH;
    auto lds_row = [](int co) { return 16 * (2 * (co >> 5) + ((co >> 2) & 1)) + 4 * ((co >> 3) & 3) + (co & 3); };
    constexpr int WV = PB / 16;                      // 16-byte loads per piece
    u32x4 w_reg[W_PER_THREAD][WV];
    if constexpr (GATED) {                          // the gate goes to LDS once (published by the barrier in the prologue)
        for (int k = tid * 4; k < K; k += NTH * 4) *reinterpret_cast<f32x4*>(gate_lds + k) = *reinterpret_cast<const f32x4*>(gate + k);
    }
    // Every load of the stage loop is UNCONDITIONAL (out-of-range pieces read a clamped, valid address and are zeroed by a
    // select afterwards; stages past the end re-read the last one): with no exec-mask branch around a vector-memory operation
    // the compiler counts them, and the wait for the W pieces of the next stage (`s_waitcnt vmcnt(N)` in front of their LDS
    // store) leaves the younger A prefetch in flight.  vmcnt retires in issue order, so the W loads are issued BEFORE the A
    // loads of the same stage; round 2 had them the other way round behind branches and drained everything (`vmcnt(0)`) once
    // per stage - the A ring was never more than one stage deep (64 - 76 % of the wave cycles waiting on vector memory).
    auto w_load = [&](int stg) {                     // issues loads only
#pragma unroll
        for (int q = 0; q < W_PER_THREAD; ++q) {
            const int idx = tid + NTH * q < BN * PPR ? tid + NTH * q : BN * PPR - 1;    // surplus threads duplicate the last piece
            const int co = idx / PPR, piece = idx % PPR;
            const int ke = stg * KCH * KPC + piece * EPC;
            const bool ok = co < n_count && ke < K;
            const int coc = ok ? co : 0, kec = ok ? ke : 0;
            const u32x4* src = reinterpret_cast<const u32x4*>(Wb + (long long)(n0 + coc) * pitch + (long long)kec * sizeof(T));
#pragma unroll
            for (int h = 0; h < WV; ++h) w_reg[q][h] = src[h];
        }
    };
    auto w_store = [&](int buf, int stg) {            // the out-of-range pieces (N tail rows, K tail) become zeros HERE, not at the load
        char* Wd = lds + buf * W_BYTES;
#pragma unroll
        for (int q = 0; q < W_PER_THREAD; ++q) {
            const int idx = tid + NTH * q < BN * PPR ? tid + NTH * q : BN * PPR - 1;
            {
                // (a bitwise AND, not a select: a select lets the compiler sink the LOAD into the in-range branch, and a load behind an
                // exec-mask branch ends every counted wait)
                const unsigned keep = (idx / PPR < n_count && stg * KCH * KPC + (idx % PPR) * EPC < K) ? 0xFFFFFFFFu : 0u;
                u32x4 v[WV];
#pragma unroll
                for (int h = 0; h < WV; ++h) v[h] = w_reg[q][h] & u32x4{keep, keep, keep, keep};
                if constexpr (GATED) {
                    const int kg_ = stg * KCH * KPC + (idx % PPR) * EPC;
                    const int kgc = kg_ < K ? kg_ : 0;
                    const f32x4 g0 = *reinterpret_cast<const f32x4*>(gate_lds + kgc);
                    if constexpr (PAIR) {
                        // two-term weights: gate the VALUE (hi + lo) in float32 and split again - gating the terms separately would
                        // round each to 8 bits
                        const f32x4 g1 = *reinterpret_cast<const f32x4*>(gate_lds + kgc + 4);
                        F8 x = pair_join8(v[0], v[1]);
#pragma unroll
                        for (int e = 0; e < 8; ++e) x.v[e] *= (e < 4 ? g0[e] : g1[e - 4]);
                        pair_split8(x, v[0], v[1]);
                    } else if constexpr (sizeof(T) == 4) {
                        f32x4 x = __builtin_bit_cast(f32x4, v[0]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) x[e] *= g0[e];
                        v[0] = __builtin_bit_cast(u32x4, x);
                    } else {
                        const f32x4 g1 = *reinterpret_cast<const f32x4*>(gate_lds + kgc + 4);
                        bf16x8 x = __builtin_bit_cast(bf16x8, v[0]);
#pragma unroll
                        for (int e = 0; e < 8; ++e) x[e] = (bf16_t)((float)x[e] * (e < 4 ? g0[e] : g1[e - 4]));
                        v[0] = __builtin_bit_cast(u32x4, x);
                    }
                }
#pragma unroll
                for (int h = 0; h < WV; ++h) *reinterpret_cast<u32x4*>(Wd + lds_row(idx / PPR) * ROWB + (idx % PPR) * PB + h * 16) = v[h];
            }
        }
    };

    f32x4 acc[PT][NT];
#pragma unroll
    for (int i = 0; i < PT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // prologue: W stage 0 in LDS, PF stages of A in flight
    w_load(0);
#pragma unroll
    for (int s = 0; s < PF; ++s) a_load(s < nst ? s : nst - 1, s);
    if constexpr (GATED) __syncthreads();           // the gate's LDS copy is complete before the first W stage is folded with it
    w_store(0, 0);
    __syncthreads();

    const int njp = (n_count + 31) / 32;               // 32-channel groups that hold real channels
    // One pipeline stage; u = stage % (PF + 1) is the A ring slot (compile-time).  Branch-free: W pieces of stage + 1 first, then
    // the A prefetch of stage + PF (in-order vmcnt: the wait for the former leaves the latter in flight), the MFMAs of this
    // stage, the LDS copy of the next W stage, one barrier.
    auto stage = [&](int stg, auto UC) {
        constexpr int u = decltype(UC)::value;
        w_load(stg + 1 < nst ? stg + 1 : nst - 1);
        a_load(stg + PF < nst ? stg + PF : nst - 1, (u + PF) % (PF + 1));
        const char* Ws = lds + (stg & 1) * W_BYTES;
#pragma unroll
        for (int sub = 0; sub < KCH; ++sub) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const Frag<T> wf = ld_frag<T>(Ws + (16 * j + frow) * ROWB + sub * CB + fpiece * PB);
#pragma unroll
                for (int i = 0; i < PT; ++i) mma_chunk(wf, areg[u][sub][i], acc[i][j]);
            }
        }
        w_store((stg + 1) & 1, stg + 1 < nst ? stg + 1 : nst - 1);   // (after the last stage: a copy nobody reads)
        __syncthreads();
    };
    // whole groups of PF + 1 stages without a single guard (so every wait in them is counted), then at most PF tail stages
    static_assert(PF == 2 || PF == 4, "the stage groups below are written out for three- and five-slot A rings");
    int stg = 0;
    for (; stg + PF + 1 <= nst; stg += PF + 1) {
        stage(stg, PwInt<0>{});
        stage(stg + 1, PwInt<1>{});
        stage(stg + 2, PwInt<2>{});
        if constexpr (PF == 4) {
            stage(stg + 3, PwInt<3>{});
            stage(stg + 4, PwInt<4>{});
        }
    }
    if (stg < nst) {
        stage(stg, PwInt<0>{});
        if (stg + 1 < nst) {
            stage(stg + 1, PwInt<1>{});
            if constexpr (PF == 4) {
                if (stg + 2 < nst) {
                    stage(stg + 2, PwInt<2>{});
                    if (stg + 3 < nst) stage(stg + 3, PwInt<3>{});
                }
            }
        }
    }

    // ---- epilogue in registers: per 32-channel group J this lane holds channels [32J + 8*fpiece, +8) of its pixels
    T* Cb = reinterpret_cast<T*>(p.C) + (long long)img * p.c_image_stride;
    const T* Rb = reinterpret_cast<const T*>(p.res);
#pragma unroll
    for (int J = 0; J < NP; ++J) {
        if (J < njp) {
            const int cb = 32 * J + 8 * fpiece;
            const int nvalid = n_count - cb;
            if (nvalid > 0) {
                float sc[8], sh[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int n = n0 + cb + (e < nvalid ? e : 0);
                    sc[e] = p.scale ? p.scale[n] : 1.0f;
                    sh[e] = p.shift[n];
                }
#pragma unroll
                for (int i = 0; i < PT; ++i) {
                    if (!pix_ok[i]) continue;
                    float v[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = fmaf(acc[i][2 * J][r], sc[r], sh[r]);
                        v[4 + r] = fmaf(acc[i][2 * J + 1][r], sc[4 + r], sh[4 + r]);
                    }
                    if (p.act == 1) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = silu_t<T>(v[e]);
                    } else if (p.act == 2) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                    T* dst = Cb + (long long)pix[i] * p.ldc + n0 + cb;
                    if (Rb != nullptr) {
                        const T* rsrc = Rb + ((long long)img * p.rows_per_image + pix[i]) * N + n0 + cb;
                        if (p.vec_ok && nvalid >= 8) {
                            const F8 rv = load8<T>(rsrc);
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] += rv.v[e];
                        } else if constexpr (!PAIR) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) if (e < nvalid) v[e] += to_f<T>(rsrc[e]);
                        }
                    }
                    if (p.vec_ok && nvalid >= 8) {
                        F8 o;
#pragma unroll
                        for (int e = 0; e < 8; ++e) o.v[e] = v[e];
                        store8<T>(dst, o);
                    } else if constexpr (!PAIR) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) if (e < nvalid) dst[e] = from_f<T>(v[e]);
                    }
                }
            }
        }
    }
}

template <typename T, int BN, int PT, int NTH, bool GATED>
__global__ __launch_bounds__(NTH, 2) void pw_gemm_kernel(PwArgs p) {
    pw_gemm_body<T, BN, PT, NTH, GATED>(p, blockIdx.x);
}

// Several independent GEMMs of one tile shape in ONE launch (the BiFPN's lateral 1x1 convs of the backbone features: six
// small problems that would otherwise each leave most of the chip idle)
constexpr int PW_GROUP_MAX = 8;
struct PwGroupArgs { int n; int first[PW_GROUP_MAX + 1]; PwArgs p[PW_GROUP_MAX]; };

template <typename T, int BN, int PT, int NTH>
__global__ __launch_bounds__(NTH, 2) void pw_gemm_group_kernel(PwGroupArgs g) {
    int i = 0;
#pragma unroll
    for (int q = 1; q < PW_GROUP_MAX; ++q) if (q < g.n && (int)blockIdx.x >= g.first[q]) i = q;
    pw_gemm_body<T, BN, PT, NTH, false>(g.p[i], (int)blockIdx.x - g.first[i]);
}

template <typename T>
int pw_prepare(PwArgs& a, int& bn, long long& blocks) {
    int ntl = (a.N + 191) / 192;
    bn = ((a.N + ntl - 1) / ntl + 31) / 32 * 32;
    a.tiles_per_image = (a.rows_per_image + PW_PIX - 1) / PW_PIX;
    const long long images = a.M / a.rows_per_image;
    a.n_tiles = (a.N + bn - 1) / bn;
    blocks = images * a.tiles_per_image * a.n_tiles;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return EFFDET_EINVAL;
    const size_t esz = sizeof(T);
    a.vec_ok = ((size_t)a.N * esz) % 16 == 0 && ((size_t)a.ldc * esz) % 16 == 0 && ((size_t)a.c_image_stride * esz) % 16 == 0 &&
               reinterpret_cast<uintptr_t>(a.C) % 16 == 0 && (a.res == nullptr || reinterpret_cast<uintptr_t>(a.res) % 16 == 0);
    return EFFDET_OK;
}

template <typename T>
int launch_pw_group(hipStream_t st, PwGroupArgs& g) {
    int bn0 = 0;
    long long total = 0;
    for (int i = 0; i < g.n; ++i) {
        int bn; long long blocks;
        const int rc = pw_prepare<T>(g.p[i], bn, blocks);
        if (rc) return rc;
        if (i == 0) bn0 = bn; else if (bn != bn0) return EFFDET_EINVAL;       // one tile shape per launch
        g.first[i] = (int)total;
        total += blocks;
        if (total > 0x7fffffffLL) return EFFDET_EINVAL;
    }
    g.first[g.n] = (int)total;
    dim3 grid((unsigned)total);
    const bool small = total < 512;
#define PWG_LAUNCH(BN_) do { if (small && BN_ != 128) hipLaunchKernelGGL((pw_gemm_group_kernel<T, BN_ == 128 ? 96 : BN_, 1, 512>), grid, dim3(512), 0, st, g); \
                             else hipLaunchKernelGGL((pw_gemm_group_kernel<T, BN_, 2, 256>), grid, dim3(256), 0, st, g); } while (0)
    switch (bn0) {
        case 32:  PWG_LAUNCH(32); break;
        case 64:  PWG_LAUNCH(64); break;
        case 96:  PWG_LAUNCH(96); break;
        default: return EFFDET_EINVAL;                                          // lateral convs are 64-88 wide (d0-d2); wider: one launch each
    }
#undef PWG_LAUNCH
    return effdet_check_launch();
}

template <typename T>
int launch_pw(hipStream_t st, PwArgs& a) {
    // output-channel tile: the smallest of {32, 64, 96, 128, 160, 192} that covers N, else equal tiles <= 192
    int bn; long long blocks;
    const int rc0 = pw_prepare<T>(a, bn, blocks);
    if (rc0) return rc0;
    dim3 grid((unsigned)blocks);
    const bool small = blocks < 512;                 // fewer than two workgroups per CU: 8 waves x 16 pixels each
    const bool gated = a.gate != nullptr;
    const size_t dyn = gated ? ((size_t)a.K * 4 + 15) / 16 * 16 : 0;          // the SE gate's LDS copy
#define PW_LAUNCH_G(BN_, G_) do { if (small && BN_ != 128) hipLaunchKernelGGL((pw_gemm_kernel<T, BN_ == 128 ? 96 : BN_, 1, 512, G_>), grid, dim3(512), dyn, st, a); /* the (128, 1, 512) instantiation spills */ \
                            else hipLaunchKernelGGL((pw_gemm_kernel<T, BN_, 2, 256, G_>), grid, dim3(256), dyn, st, a); } while (0)
#define PW_LAUNCH(BN_) do { if (gated) PW_LAUNCH_G(BN_, true); else PW_LAUNCH_G(BN_, false); } while (0)
    switch (bn) {
        case 32:  PW_LAUNCH(32); break;
        case 64:  PW_LAUNCH(64); break;
        case 96:  PW_LAUNCH(96); break;
        case 128: PW_LAUNCH(128); break;
        case 160: PW_LAUNCH(160); break;
        case 192: PW_LAUNCH(192); break;
        default: return EFFDET_EINVAL;
    }
#undef PW_LAUNCH
#undef PW_LAUNCH_G
    return effdet_check_launch();
}

}  // namespace

extern "C" int effdet_pw_gemm_bn_act(void* stream, int dtype,
                                     const void* A, long long M, int K,
                                     const void* W, int N,
                                     const float* scale, const float* shift, int act,
                                     const void* residual,
                                     const float* gate, int rows_per_image,
                                     void* C, long long c_image_stride, long long ldc) {
    EFFDET_ENTER();
    if (!A || !W || !C || !shift || M <= 0 || K <= 0 || N <= 0) return EFFDET_EINVAL;
    if (K % 8 != 0 || M > 0x7fffffffLL) return EFFDET_EINVAL;   // 16-byte pieces along K; 32-bit row arithmetic
    if (act < 0 || act > 2) return EFFDET_EINVAL;                  // 0 none, 1 SiLU, 2 ReLU
    if (rows_per_image <= 0) { rows_per_image = (int)(M > 0x7fffffffLL ? 0x7fffffff : M); }
    if ((M % rows_per_image) != 0) return EFFDET_EINVAL;         // workgroups never straddle images
    if (ldc <= 0) ldc = N;
    if (c_image_stride <= 0) c_image_stride = (long long)rows_per_image * ldc;
    PwArgs a{A, M, K, W, N, scale, shift, act, residual, gate, rows_per_image, C, c_image_stride, ldc, 0, 0, 0};
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == 0) return launch_pw<float>(st, a);
    if (dtype == 1) return launch_pw<bf16_t>(st, a);
    if (dtype == 2) {                                              // two-term bf16: whole 8-channel groups, 32-byte aligned rows
        if (N % 8 || ldc % 8 || c_image_stride % 8 || reinterpret_cast<uintptr_t>(A) % 32 || reinterpret_cast<uintptr_t>(W) % 32 ||
            reinterpret_cast<uintptr_t>(C) % 32 || (residual && reinterpret_cast<uintptr_t>(residual) % 32)) return EFFDET_EINVAL;
        return launch_pw<bf16p_t>(st, a);
    }
    return EFFDET_EINVAL;
}

// n independent GEMMs (no gate, no residual, dense outputs) in one launch; all N must select the same output tile (<= 96).
// Arrays are per problem.  Used for the BiFPN's lateral 1x1 convs (effdet/efficientdet.py:155-158 via ResampleFeatureMap).
extern "C" int effdet_pw_gemm_group(void* stream, int dtype, int n, const void* const* A, const long long* M, const int* K,
                                    const void* const* W, const int* N, const float* const* scale, const float* const* shift,
                                    int act, void* const* C) {
    EFFDET_ENTER();
    if (n < 1 || n > PW_GROUP_MAX || !A || !M || !K || !W || !N || !scale || !shift || !C || act < 0 || act > 2) return EFFDET_EINVAL;
    PwGroupArgs g;
    g.n = n;
    for (int i = 0; i < n; ++i) {
        if (!A[i] || !W[i] || !C[i] || !shift[i] || M[i] <= 0 || M[i] > 0x7fffffffLL || K[i] <= 0 || K[i] % 8 || N[i] <= 0) return EFFDET_EINVAL;
        g.p[i] = PwArgs{A[i], M[i], K[i], W[i], N[i], scale[i], shift[i], act, nullptr, nullptr, (int)M[i], C[i], M[i] * N[i], N[i], 0, 0, 0};
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == 0) return launch_pw_group<float>(st, g);
    if (dtype == 1) return launch_pw_group<bf16_t>(st, g);
    if (dtype == 2) {
        for (int i = 0; i < n; ++i)
            if (N[i] % 8 || reinterpret_cast<uintptr_t>(A[i]) % 32 || reinterpret_cast<uintptr_t>(W[i]) % 32 || reinterpret_cast<uintptr_t>(C[i]) % 32)
                return EFFDET_EINVAL;
        return launch_pw_group<bf16p_t>(st, g);
    }
    return EFFDET_EINVAL;
}
