// Pointwise (1x1) convolution as an MFMA GEMM with fused per-channel affine (folded BN / bias),
// optional SiLU, optional SE gate on the A operand and optional residual add.
//
//   C[m, n] = act( (sum_k A[m,k] * gate[img(m),k] * W[n,k]) * scale[n] + shift[n] ) + R[m,n]
//
// Replaces the 1x1 `create_conv2d` + BatchNorm2d (+ Swish) call sites of the reference:
//   timm EfficientNet conv_pw / conv_pwl (effdet/efficientdet.py:837) and the BiFPN lateral
//   ConvBnAct2d (effdet/efficientdet.py:155-158).
//
// A is the NHWC activation viewed as [M = B*H*W, K]; W is the conv weight [N = Cout, K = Cin].
// The workgroup tile is 128 rows x BN columns; K is walked in 64-byte chunks (32 bf16 / 16 f32)
// through a double-buffered LDS image whose rows are padded to 80 bytes.  Each of the 4 waves owns
// 32 rows x BN columns of 16x16 MFMA tiles.  The accumulator tile is staged through LDS so that
// HBM stores are whole 16-byte pieces of a row.
#include "common.h"

namespace {

struct PwArgs {
    const void* A; long long M; int K;
    const void* W; int N;
    const float* scale; const float* shift; int act;
    const void* res;
    const float* gate; int rows_per_image;
    void* C; long long c_image_stride; long long ldc;
};

constexpr int BM = 128;

template <typename T> struct Chunk { u32x4 raw; };

template <typename T>
DEV u32x4 apply_gate(u32x4 raw, const float* g) {
    if constexpr (sizeof(T) == 4) {
        f32x4 v = __builtin_bit_cast(f32x4, raw);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] *= g[i];
        return __builtin_bit_cast(u32x4, v);
    } else {
        bf16x8 v = __builtin_bit_cast(bf16x8, raw);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (bf16_t)((float)v[i] * g[i]);
        return __builtin_bit_cast(u32x4, v);
    }
}

template <typename T, int BN>
__global__ __launch_bounds__(256) void pw_gemm_kernel(PwArgs p) {
    constexpr int EPC = VecTraits<T>::EPC;          // elements per 16-byte piece
    constexpr int KPC = 64 / (int)sizeof(T);        // elements per 64-byte K-chunk
    constexpr int NT = BN / 16;                     // 16-wide column tiles per wave
    // 64-byte K-chunks per pipeline stage: two for the wide tile (more bytes in flight at 2 workgroups/CU),
    // one for the narrow tiles, whose small LDS footprint already gives 4-6 workgroups per CU
    constexpr int KCH = BN >= 128 ? 2 : 1;
    constexpr int ROWB = KCH * 64 + 16;             // bytes per LDS row: stage of K + 16 pad
    constexpr int A_BYTES = BM * ROWB;
    constexpr int W_BYTES = BN * ROWB;
    constexpr int TILE_BYTES = 2 * (A_BYTES + W_BYTES);
    constexpr int SROW = BN + 4;                    // staging row stride in floats
    constexpr int STAGE_BYTES = BM * SROW * 4;
    constexpr int LDS_BYTES = TILE_BYTES > STAGE_BYTES ? TILE_BYTES : STAGE_BYTES;
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int n_tiles = (p.N + BN - 1) / BN;
    const long long bid = blockIdx.x;
    const int nt = (int)(bid % n_tiles);
    const long long mt = bid / n_tiles;
    const long long m0 = mt * BM;
    const int n0 = nt * BN;
    const int K = p.K;
    const long long pitch = (long long)K * (long long)sizeof(T);
    const int nkc = (K + KPC - 1) / KPC;            // 64-byte chunks
    const int nst = (nkc + KCH - 1) / KCH;          // pipeline stages

    const char* Ab = reinterpret_cast<const char*>(p.A);
    const char* Wb = reinterpret_cast<const char*>(p.W);

    // staging assignment: per stage A has 128 rows x PPR pieces, W has BN x PPR pieces (16 bytes each)
    constexpr int PPR = KCH * 4;
    constexpr int A_PER_THREAD = BM * PPR / 256;
    constexpr int W_PIECES = BN * PPR;
    constexpr int W_PER_THREAD = (W_PIECES + 255) / 256;

    u32x4 a_reg[A_PER_THREAD];
    u32x4 w_reg[W_PER_THREAD];
    f32x4 g_reg[A_PER_THREAD][2];                     // SE gate of the A pieces (8 floats each), applied at store time

    // per-thread constants of its A rows, hoisted out of the K loop (the image index needs an integer division)
    const char* a_src[A_PER_THREAD];
    const float* a_gate[A_PER_THREAD];
    bool a_ok[A_PER_THREAD];
#pragma unroll
    for (int q = 0; q < A_PER_THREAD; ++q) {
        const int idx = tid + 256 * q;
        const long long m = m0 + idx / PPR;
        a_ok[q] = m < p.M;
        a_src[q] = Ab + (a_ok[q] ? m : 0) * pitch;
        a_gate[q] = p.gate != nullptr ? p.gate + (long long)((unsigned int)(a_ok[q] ? m : 0) / (unsigned int)p.rows_per_image) * K : nullptr;
    }

    // load_stage only ISSUES loads (A piece, its gate vector, W piece): nothing here may consume them, or the
    // prefetch would stall on its own data instead of overlapping the MFMA work of the current stage
    auto load_stage = [&](int stg) {
#pragma unroll
        for (int q = 0; q < A_PER_THREAD; ++q) {
            const int idx = tid + 256 * q;
            const int piece = idx % PPR;
            const int ke = stg * KCH * KPC + piece * EPC;    // element offset along K
            u32x4 v = {0u, 0u, 0u, 0u};
            f32x4 g0 = {1.f, 1.f, 1.f, 1.f}, g1 = g0;
            if (a_ok[q] && ke < K) {
                v = *reinterpret_cast<const u32x4*>(a_src[q] + (long long)ke * sizeof(T));
                if (p.gate != nullptr) {
                    g0 = *reinterpret_cast<const f32x4*>(a_gate[q] + ke);
                    if constexpr (sizeof(T) == 2) g1 = *reinterpret_cast<const f32x4*>(a_gate[q] + ke + 4);
                }
            }
            a_reg[q] = v; g_reg[q][0] = g0; g_reg[q][1] = g1;
        }
#pragma unroll
        for (int q = 0; q < W_PER_THREAD; ++q) {
            const int idx = tid + 256 * q;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (idx < W_PIECES) {
                const int piece = idx % PPR, row = idx / PPR;
                const int n = n0 + row;
                const int ke = stg * KCH * KPC + piece * EPC;
                if (n < p.N && ke < K)
                    v = *reinterpret_cast<const u32x4*>(Wb + (long long)n * pitch + (long long)ke * sizeof(T));
            }
            w_reg[q] = v;
        }
    };
    auto store_stage = [&](int buf) {
        char* Ad = lds + buf * (A_BYTES + W_BYTES);
        char* Wd = Ad + A_BYTES;
#pragma unroll
        for (int q = 0; q < A_PER_THREAD; ++q) {
            const int idx = tid + 256 * q;
            u32x4 v = a_reg[q];
            if (p.gate != nullptr) {
                float g[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) { g[e] = g_reg[q][0][e]; g[4 + e] = g_reg[q][1][e]; }
                v = apply_gate<T>(v, g);
            }
            *reinterpret_cast<u32x4*>(Ad + (idx / PPR) * ROWB + (idx % PPR) * 16) = v;
        }
#pragma unroll
        for (int q = 0; q < W_PER_THREAD; ++q) {
            const int idx = tid + 256 * q;
            if (idx < W_PIECES)
                *reinterpret_cast<u32x4*>(Wd + (idx / PPR) * ROWB + (idx % PPR) * 16) = w_reg[q];
        }
    };

    f32x4 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    load_stage(0);
    store_stage(0);
    __syncthreads();

    const int frow = lane & 15, fpiece = lane >> 4;
    for (int stg = 0; stg < nst; ++stg) {
        const int buf = stg & 1;
        if (stg + 1 < nst) load_stage(stg + 1);
        const char* As = lds + buf * (A_BYTES + W_BYTES);
        const char* Ws = As + A_BYTES;
#pragma unroll
        for (int sub = 0; sub < KCH; ++sub) {
            if (stg * KCH + sub < nkc) {
                Frag<T> a0 = ld_frag<T>(As + (32 * wave + frow) * ROWB + sub * 64 + fpiece * 16);
                Frag<T> a1 = ld_frag<T>(As + (32 * wave + 16 + frow) * ROWB + sub * 64 + fpiece * 16);
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    Frag<T> b = ld_frag<T>(Ws + (16 * j + frow) * ROWB + sub * 64 + fpiece * 16);
                    mma_chunk(a0, b, acc[0][j]);
                    mma_chunk(a1, b, acc[1][j]);
                }
            }
        }
        if (stg + 1 < nst) store_stage(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: accumulators -> LDS staging (fp32) -> affine/act/residual -> HBM
    float* S = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                S[(32 * wave + 16 * i + 4 * fpiece + r) * SROW + 16 * j + frow] = acc[i][j][r];
    __syncthreads();

    constexpr int GPR = BN / 8;                      // 8-column groups per row
    T* Cb = reinterpret_cast<T*>(p.C);
    const T* Rb = reinterpret_cast<const T*>(p.res);
    for (int g = tid; g < BM * GPR; g += 256) {
        const int row = g / GPR, cg = g % GPR;
        const long long m = m0 + row;
        const int n = n0 + cg * 8;
        if (m >= p.M || n >= p.N) continue;
        const unsigned int bimg = (unsigned int)m / (unsigned int)p.rows_per_image;      // M < 2^31: 32-bit division
        const unsigned int pix = (unsigned int)m - bimg * (unsigned int)p.rows_per_image;
        T* dst = Cb + (long long)bimg * p.c_image_stride + (long long)pix * p.ldc + n;
        const int nvalid = (p.N - n) < 8 ? (p.N - n) : 8;
        float v[8];
        const f32x4 va = *reinterpret_cast<const f32x4*>(S + row * SROW + cg * 8);        // aligned: SROW % 4 == 0
        const f32x4 vb = *reinterpret_cast<const f32x4*>(S + row * SROW + cg * 8 + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float x = e < 4 ? va[e] : vb[e - 4];
            if (e < nvalid) {
                const float sc = p.scale ? p.scale[n + e] : 1.0f;
                x = x * sc + p.shift[n + e];
                if (p.act == 1) x = silu_t<T>(x);
                if (Rb) x += to_f<T>(Rb[m * p.N + n + e]);
            }
            v[e] = x;
        }
        if (nvalid == 8 && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
            F8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o.v[e] = v[e];
            store8<T>(dst, o);
        } else {
            for (int e = 0; e < nvalid; ++e) dst[e] = from_f<T>(v[e]);
        }
    }
}

template <typename T>
int launch_pw(hipStream_t st, const PwArgs& a) {
    const long long mt = (a.M + BM - 1) / BM;
    int bn = a.N <= 16 ? 16 : a.N <= 32 ? 32 : a.N <= 64 ? 64 : 128;
    const long long ntile = (a.N + bn - 1) / bn;
    const long long blocks = mt * ntile;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return EFFDET_EINVAL;
    dim3 grid((unsigned)blocks), block(256);
    switch (bn) {
        case 16: hipLaunchKernelGGL((pw_gemm_kernel<T, 16>), grid, block, 0, st, a); break;
        case 32: hipLaunchKernelGGL((pw_gemm_kernel<T, 32>), grid, block, 0, st, a); break;
        case 64: hipLaunchKernelGGL((pw_gemm_kernel<T, 64>), grid, block, 0, st, a); break;
        default: hipLaunchKernelGGL((pw_gemm_kernel<T, 128>), grid, block, 0, st, a); break;
    }
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
    if (act != 0 && act != 1) return EFFDET_EINVAL;
    if (rows_per_image <= 0) { rows_per_image = (int)(M > 0x7fffffffLL ? 0x7fffffff : M); }
    if (gate && (M % rows_per_image) != 0) return EFFDET_EINVAL;
    if (ldc <= 0) ldc = N;
    if (c_image_stride <= 0) c_image_stride = (long long)rows_per_image * ldc;
    PwArgs a{A, M, K, W, N, scale, shift, act, residual, gate, rows_per_image, C, c_image_stride, ldc};
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == 0) return launch_pw<float>(st, a);
    if (dtype == 1) return launch_pw<bf16_t>(st, a);
    return EFFDET_EINVAL;
}
