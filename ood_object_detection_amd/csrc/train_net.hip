// Training-path kernels of the pretrain step (pretrain.py:226-236: forward with saved activations,
// `qry_loss.backward()` through heads, BiFPN and backbone).  float32 only (the reference trains in fp32), NHWC.
//
// The step is a sequence of these generic operators (host side: ood_object_detection_amd/train_engine.py):
//   effdet_train_gemm_nt     C[M,N] = A[M,K] W[N,K]^T + bias      1x1 conv forward, and dX = dY W (W passed transposed)
//   effdet_train_gemm_tn     out[N,K+1] = dY[M,N]^T [X[M,K] | 1]   weight gradient and per-channel sum of dY in one pass
//   effdet_train_dwconv_bwd_dx / _dw                               depthwise k x k (TF-SAME, stride 1|2) backward
//   effdet_train_ew          element-wise family (SiLU fwd/bwd, affine, SE gate fwd/bwd, BN-train bwd, fusion, add)
//   effdet_train_col_reduce  per-channel reductions over pixels (BN statistics, SE pool, dot products)
//   effdet_train_spatial     nearest x2 upsample fwd/bwd, 3x3/s2 max-pool backward
//   effdet_train_im2col_stem conv_stem input patches (the stem then runs on the two GEMMs above)
//   effdet_train_se_bwd      SqueezeExcite FC backward (per image) + its parameter-gradient partials
//   effdet_train_reduce_mid  fixed-order second stage of every two-stage reduction (bitwise reproducible)
// Every reduction is two-stage with a fixed order: gradients are bitwise reproducible run to run.
// MFMA 16x16x4 f32 carries the GEMMs; operands are streamed from HBM/L2 straight into registers (the weight side is
// small and L2 resident); D holds 4 consecutive output channels of one pixel per lane -> 16-byte stores.
#include "common.h"

namespace {

// row m -> (m / rpi) * img_stride + (m % rpi) * ld;  with nlev > 0 the rows are those of a level-major packed pyramid (level,
// image, pixel) and the tensor is the image-major one of the head outputs, [B][sum_l H_l W_l][ld]: row -> image * img_stride +
// (lq0[level] + pixel) * ld   (effdet_train_gemm_*_levels)
constexpr int MAXLEV = 8;
struct RowMap { long long rpi, img_stride, ld; int nlev; long long lrow0[MAXLEV + 1], lq0[MAXLEV], lhw[MAXLEV]; };
DEV long long row_off(const RowMap& r, long long m) {
    if (r.nlev > 0) {                                        // rows < 2^31 (make_levels_rowmap checks): 32-bit division
        const unsigned mu = (unsigned)m;
        unsigned row0 = 0, hw = (unsigned)r.lhw[0], q0 = 0;
#pragma unroll
        for (int i = 1; i < MAXLEV; ++i) {
            const bool in = i < r.nlev && mu >= (unsigned)r.lrow0[i];
            row0 = in ? (unsigned)r.lrow0[i] : row0;
            hw = in ? (unsigned)r.lhw[i] : hw;
            q0 = in ? (unsigned)r.lq0[i] : q0;
        }
        const unsigned local = mu - row0;
        const unsigned b = local / hw;
        return (long long)b * r.img_stride + (long long)(q0 + (local - b * hw)) * r.ld;
    }
    if (r.img_stride == 0) return m * r.ld;
    const long long q = m / r.rpi;
    return q * r.img_stride + (m - q * r.rpi) * r.ld;
}
inline RowMap make_rowmap(long long rpi, long long img_stride, long long ld, long long M, long long cols) {
    RowMap r;
    r.nlev = 0;
    for (int l = 0; l < MAXLEV; ++l) { r.lrow0[l] = 0; r.lq0[l] = 0; r.lhw[l] = 1; }
    r.lrow0[MAXLEV] = 0;
    if (rpi <= 0 || img_stride <= 0) { r.rpi = M > 0 ? M : 1; r.img_stride = 0; r.ld = ld > 0 ? ld : cols; }
    else { r.rpi = rpi; r.img_stride = img_stride; r.ld = ld > 0 ? ld : cols; }
    return r;
}
// -> number of rows (B * sum H W), or < 0
inline long long make_levels_rowmap(RowMap& r, int B, int L, const int* Hs, const int* Ws, long long img_stride, long long ld) {
    if (B <= 0 || L <= 0 || L > MAXLEV || !Hs || !Ws || img_stride <= 0 || ld <= 0) return -1;
    r = make_rowmap(0, 0, ld, 1, ld);
    r.nlev = L; r.img_stride = img_stride;
    long long q = 0;
    for (int l = 0; l < L; ++l) {
        if (Hs[l] <= 0 || Ws[l] <= 0) return -1;
        r.lhw[l] = (long long)Hs[l] * Ws[l];
        r.lq0[l] = q;
        q += r.lhw[l];
        r.lrow0[l + 1] = r.lrow0[l] + (long long)B * r.lhw[l];
    }
    if (q * ld > img_stride || r.lrow0[L] >= 0x7fffffffLL) return -1;
    return r.lrow0[L];
}

// ------------------------------------------------------------------------------------------------------------
// C = A W^T + bias
// ------------------------------------------------------------------------------------------------------------
struct GemmNtArgs {
    const float* A; RowMap am; const float* W; const float* bias; float* C; RowMap cm;
    long long M; int K, N, accumulate, vec_out;
    float* C2;                                 // optional second output with the row map of C: silu(C)
    const float* R;                            // optional residual with the row map of C, added after the bias
    const float* a_scale; long long a_scale_rpi;   // optional per-image scale of A's columns (SE gate [img][K]), img = row / a_scale_rpi
};

// VW: widest aligned load the rows allow (4 = 16 bytes, 2 = 8 bytes: e.g. the 810-channel class head, 1 = scalar)
template <int VW>
DEV Frag<float> ld_k4(const float* row, int k, int K) {
    Frag<float> f;
    if constexpr (VW == 4) {
        if (k + 3 < K) f.v = *reinterpret_cast<const f32x4*>(row + k);
        else f.v = f32x4{0.f, 0.f, 0.f, 0.f};
    } else if constexpr (VW == 2) {
        const f32x2 lo = (k + 1 < K) ? *reinterpret_cast<const f32x2*>(row + k) : f32x2{0.f, 0.f};
        const f32x2 hi = (k + 3 < K) ? *reinterpret_cast<const f32x2*>(row + k + 2) : f32x2{0.f, 0.f};
        f.v = f32x4{lo[0], lo[1], hi[0], hi[1]};
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) f.v[i] = (k + i < K) ? row[k + i] : 0.f;
    }
    return f;
}

// A wave owns 32 rows x 64 columns: every W fragment it fetches feeds two row tiles, and the fragments of the next 16 k are
// requested before the current ones go to the matrix cores (the loop has a runtime trip count; without the explicit
// double buffer every step would expose one L2 round trip).  The k order per output element is unchanged: sequential.
// KS = 4: the four waves of a workgroup share ONE 32-row tile and each takes a quarter of K; the partial accumulators are
// added in wave order through LDS (fixed association).  For the deep-K convs on small maps (M = a few thousand rows, K up to
// 1152) this gives 4x the workgroups and a quarter of the serial K chain.
// FAST (VEC = 4, K % 4 == 0, dense A rows): the k loop has no branch around a load and no use of a loaded value before the next
// step - operands are fetched TWO steps ahead into two register stages that ping-pong (addresses clamped to the last quad of K,
// the out-of-range lanes zeroed where the A fragment is built; the SE gate travels with the stage), so the waits are counted
// and a wave keeps 12 - 16 KB on the wire.  The general loop below issues every load under `k + 3 < K` and waits for it at once.
template <int VEC, int KS, bool FAST = false>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmNtArgs p) {
    constexpr int RT = 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    // the column blocks of one row group are consecutive workgroups (n fastest): they run at the same time, so a wide row (the
    // 810-column class head) is completed in L2 before it is evicted and its A rows are fetched from HBM once
    const unsigned nby = (unsigned)((p.N + 63) / 64);
    const unsigned bx = blockIdx.x / nby, by = blockIdx.x - bx * nby;
    const long long m0 = KS == 1 ? ((long long)bx * 4 + wave) * (16 * RT) : (long long)bx * (16 * RT);
    const int n0 = by * 64;
    if (m0 >= p.M) return;                                   // uniform per wave (KS = 1) or per workgroup (KS = 4)
    int kb = 0, ke = p.K;
    if constexpr (KS > 1) {
        const int kchunk = (p.K + 16 * KS - 1) / (16 * KS) * 16;
        kb = wave * kchunk;
        ke = kb + kchunk < p.K ? kb + kchunk : p.K;
    }
    const float* arow[RT];
    const float* grow[RT];
    bool mv[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        long long m = m0 + 16 * i + r16;
        mv[i] = m < p.M;
        if (!mv[i]) m = p.M - 1;
        arow[i] = p.A + (FAST ? m * p.am.ld : row_off(p.am, m));
        grow[i] = p.a_scale ? p.a_scale + (long long)((unsigned)m / (unsigned)p.a_scale_rpi) * p.K : nullptr;    // rows < 2^31 (launcher)
    }
    auto lda = [&](int i, int k) {                            // A fragment, times the image's gate where there is one
        Frag<float> f = ld_k4<VEC>(arow[i], k, p.K);
        if (p.a_scale) f.v *= ld_k4<VEC>(grow[i], k, p.K).v;
        return f;
    };
    const float* wrow[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        int n = n0 + 16 * t + r16;
        if (n >= p.N) n = p.N - 1;
        wrow[t] = p.W + (long long)n * p.K;
    }
    f32x4 acc[RT][4];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (FAST) {
        struct Stage { f32x4 a[RT], w[4], gt[RT]; };
        const bool scaled = p.a_scale != nullptr;             // uniform
        auto fetch = [&](int k0) {
            int k = k0 + 4 * g;
            k = k < p.K - 3 ? k : p.K - 4;
            Stage sg;
#pragma unroll
            for (int t = 0; t < 4; ++t) sg.w[t] = *reinterpret_cast<const f32x4*>(wrow[t] + k);
#pragma unroll
            for (int i = 0; i < RT; ++i) sg.a[i] = *reinterpret_cast<const f32x4*>(arow[i] + k);
#pragma unroll
            for (int i = 0; i < RT; ++i) sg.gt[i] = *reinterpret_cast<const f32x4*>((scaled ? grow[i] : arow[i]) + k);
            return sg;
        };
        auto step = [&](Stage& sg, int k0) {
            const bool kv = k0 + 4 * g < ke;
            Frag<float> fa[RT], fw[4];
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                f32x4 v = scaled ? sg.a[i] * sg.gt[i] : sg.a[i];
                fa[i].v = kv ? v : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) fw[t].v = sg.w[t];
            __builtin_amdgcn_sched_barrier(0);
            sg = fetch(k0 + 32);                              // this stage again two steps on (clamped past the end of K)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < RT; ++i) mma_chunk(fw[t], fa[i], acc[i][t]);
        };
        Stage s0 = fetch(kb), s1 = fetch(kb + 16);
        int k0 = kb;
        for (; k0 + 16 < ke; k0 += 32) { step(s0, k0); step(s1, k0 + 16); }
        if (k0 < ke) step(s0, k0);
    } else {
    Frag<float> bc[RT], ac[4];
#pragma unroll
    for (int i = 0; i < RT; ++i) bc[i] = lda(i, kb + 4 * g);
#pragma unroll
    for (int t = 0; t < 4; ++t) ac[t] = ld_k4<VEC>(wrow[t], kb + 4 * g, p.K);
    for (int k0 = kb; k0 < ke; k0 += 16) {
        const int kn = k0 + 16 + 4 * g;                       // past the end of K: ld_k4 returns zeros without touching memory
        Frag<float> bn[RT], an[4];
#pragma unroll
        for (int i = 0; i < RT; ++i) bn[i] = lda(i, kn);
#pragma unroll
        for (int t = 0; t < 4; ++t) an[t] = ld_k4<VEC>(wrow[t], kn, p.K);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < RT; ++i) mma_chunk(ac[t], bc[i], acc[i][t]);
#pragma unroll
        for (int i = 0; i < RT; ++i) bc[i] = bn[i];
#pragma unroll
        for (int t = 0; t < 4; ++t) ac[t] = an[t];
    }
    }
    if constexpr (KS > 1) {
        __shared__ float red[KS - 1][RT * 16][64];
        if (wave > 0) {
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[wave - 1][(i * 4 + t) * 4 + r][lane] = acc[i][t][r];
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int w = 0; w < KS - 1; ++w)
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][t][r] += red[w][(i * 4 + t) * 4 + r][lane];
    }
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        if (!mv[i]) continue;
        const long long coff = row_off(p.cm, m0 + 16 * i + r16);
        float* crow = p.C + coff;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int n = n0 + 16 * t + 4 * g;
            if (p.vec_out == 1) {                                // N % 4 == 0 and 16-byte aligned rows: one store per lane and tile
                if (n < p.N) {
                    f32x4 v = acc[i][t];
                    if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
                    if (p.R) v += *reinterpret_cast<const f32x4*>(p.R + coff + n);
                    if (p.accumulate) v += *reinterpret_cast<const f32x4*>(crow + n);
                    *reinterpret_cast<f32x4*>(crow + n) = v;
                    if (p.C2) {
                        f32x4 q;
#pragma unroll
                        for (int r = 0; r < 4; ++r) q[r] = silu_train(v[r]);
                        *reinterpret_cast<f32x4*>(p.C2 + coff + n) = q;
                    }
                }
                continue;
            }
            if (p.vec_out == 2) {                                // N even, 8-byte aligned rows (the 810-wide class head): two stores
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (n + 2 * h < p.N) {
                        f32x2 v = f32x2{acc[i][t][2 * h], acc[i][t][2 * h + 1]};
                        if (p.bias) v += *reinterpret_cast<const f32x2*>(p.bias + n + 2 * h);
                        if (p.R) v += *reinterpret_cast<const f32x2*>(p.R + coff + n + 2 * h);
                        if (p.accumulate) v += *reinterpret_cast<const f32x2*>(crow + n + 2 * h);
                        *reinterpret_cast<f32x2*>(crow + n + 2 * h) = v;
                        if (p.C2) *reinterpret_cast<f32x2*>(p.C2 + coff + n + 2 * h) = f32x2{silu_train(v[0]), silu_train(v[1])};
                    }
                }
                continue;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (n + r < p.N) {
                    float v = acc[i][t][r];
                    if (p.bias) v += p.bias[n + r];
                    if (p.R) v += p.R[coff + n + r];
                    if (p.accumulate) v += crow[n + r];
                    crow[n + r] = v;
                    if (p.C2) p.C2[coff + n + r] = silu_train(v);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// out[N][K+1] = dY^T [X | 1]  (two-stage over M)
// ------------------------------------------------------------------------------------------------------------
struct GemmTnArgs {
    const float* dY; RowMap ym; const float* X; RowMap xm; float* partial;
    long long M, rows_per_slice; int N, K, S;
    const float* x_scale; long long x_scale_rpi;     // optional per-image scale of X's columns (SE gate [img][K])
};

// 32 rows of dY [32 n] and X [64 k] per step are staged in LDS with 16-byte global loads (rows padded to 48 / 80 floats:
// the two 16-lane groups of a half-wave land on disjoint banks), then every wave feeds 8 of those rows to the matrix cores.
template <int VY, bool VX>
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTnArgs p) {
    constexpr int RT = 32, SY = 48, SX = 80;
    __shared__ float sY[RT * SY];
    __shared__ float sX[RT * SX];
    __shared__ float sm[4][32 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c16 = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 32, k0 = blockIdx.y * 64;
    const long long mb = (long long)blockIdx.z * p.rows_per_slice;
    long long me = mb + p.rows_per_slice;
    if (me > p.M) me = p.M;
    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int Kx = p.K + 1;
    const int yr = tid >> 3, yc = (tid & 7) * 4;             // dY piece of this thread: row yr, columns yc..yc+3
    const int xr = tid >> 4, xc = (tid & 15) * 4;            // X pieces: rows xr and xr + 16, columns xc..xc+3
    f32x4 vy, vx[2];
    auto fetch = [&](long long m0) {                          // 32 rows of dY / X -> registers (loads only)
        vy = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            const long long m = m0 + yr;
            if (m < me) {
                const float* row = p.dY + row_off(p.ym, m);
                const int n = n0 + yc;
                if (VY == 4 && n + 3 < p.N) vy = *reinterpret_cast<const f32x4*>(row + n);
                else if (VY == 2 && n + 1 < p.N) {              // rows are 8-byte aligned (N even): two half-width loads
                    const f32x2 lo = *reinterpret_cast<const f32x2*>(row + n);
                    const f32x2 hi = n + 3 < p.N ? *reinterpret_cast<const f32x2*>(row + n + 2) : f32x2{0.f, 0.f};
                    vy = f32x4{lo[0], lo[1], hi[0], hi[1]};
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) vy[e] = n + e < p.N ? row[n + e] : 0.f;
                }
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            vx[h] = f32x4{0.f, 0.f, 0.f, 0.f};
            const long long m = m0 + xr + 16 * h;
            if (m < me) {
                const float* row = p.X + row_off(p.xm, m);
                const int kk = k0 + xc;
                const float* gr = p.x_scale ? p.x_scale + (m / p.x_scale_rpi) * p.K : nullptr;
                if (VX && kk + 3 < p.K) {
                    vx[h] = *reinterpret_cast<const f32x4*>(row + kk);
                    if (gr) vx[h] *= *reinterpret_cast<const f32x4*>(gr + kk);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        vx[h][e] = kk + e < p.K ? (gr ? row[kk + e] * gr[kk + e] : row[kk + e]) : (kk + e == p.K ? 1.f : 0.f);
                }
            }
        }
    };
    fetch(mb);
    for (long long m0 = mb; m0 < me; m0 += RT) {
        __syncthreads();                                      // the previous step's operands have been consumed
        *reinterpret_cast<f32x4*>(sY + yr * SY + yc) = vy;
        *reinterpret_cast<f32x4*>(sX + xr * SX + xc) = vx[0];
        *reinterpret_cast<f32x4*>(sX + (xr + 16) * SX + xc) = vx[1];
        __syncthreads();
        if (m0 + RT < me) fetch(m0 + RT);                     // the next step's rows are in flight while this one is multiplied
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const int r = 8 * wave + 4 * st + g;
            float a[2], b[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = sY[r * SY + 16 * i + c16];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = sX[r * SX + 16 * j + c16];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    // D[row = n local (4g + r)][col = k local (c16)]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) sm[wave][(16 * i + 4 * g + r) * 64 + 16 * j + c16] = acc[i][j][r];
    __syncthreads();
    float* out = p.partial + (long long)blockIdx.z * p.N * Kx;
    for (int e = threadIdx.x; e < 32 * 64; e += 256) {
        const int n = n0 + e / 64, kk = k0 + e % 64;
        if (n < p.N && kk < Kx) out[(long long)n * Kx + kk] = ((sm[0][e] + sm[1][e]) + sm[2][e]) + sm[3][e];
    }
}

// The same product for 16-byte aligned operands with N % 4 == 0 and K % 4 == 0 (every conv of the network except the 810-wide
// class head): rows are fetched TWO steps ahead with unconditional loads on clamped addresses (a load under a branch, or one
// whose result feeds a select at once, makes the compiler wait for it on the spot: MI355X notes in DESIGN.md), the masks are
// applied when a step's registers go to LDS.  With one step in flight a workgroup had 12 KB on the wire for about half of the
// time: 1.8 TB/s on the 315 MB operands of the early layers.  The staging tiles and the epilogue buffer share their LDS.
// DENSE: both operands are plain row-major (row * ld): no row map in the loop (its 64-bit divisions and level tables cost more
// than the step itself)
// SCALED: X is multiplied by the image's gate row; the gate pieces travel with the stage and are applied at the LDS store (no
// branch and no use of a loaded value inside the fetch: the waits stay counted)
// VYW = 2: dY rows are only 8-byte aligned (N even: the 810-wide class head) - the dY piece travels as two 8-byte halves
template <bool DENSE, bool SCALED, int VYW>
__global__ __launch_bounds__(256) void gemm_tn_kernel_v(GemmTnArgs p) {
    constexpr int RT = 32, SY = 48, SX = 80;
    __shared__ float lds[4 * 32 * 64];
    float* sY = lds;
    float* sX = lds + RT * SY;
    float (*sm)[32 * 64] = reinterpret_cast<float (*)[32 * 64]>(lds);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c16 = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 32, k0 = blockIdx.y * 64;
    const long long mb = (long long)blockIdx.z * p.rows_per_slice;
    long long me = mb + p.rows_per_slice;
    if (me > p.M) me = p.M;
    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int Kx = p.K + 1;
    const int yr = tid >> 3, yc = (tid & 7) * 4;
    const int xr = tid >> 4, xc = (tid & 15) * 4;
    const int n = n0 + yc, kk = k0 + xc;
    const bool nok = n < p.N, nok2 = n + 2 < p.N, kok = kk < p.K, kone = kk == p.K;     // N even: a half is inside or outside as a whole
    const int nl = nok ? n : 0, nl2 = nok2 ? n + 2 : 0, kl = kok ? kk : 0;
    struct Stage { f32x4 vy, vx0, vx1, g0, g1; };
    auto fetch = [&](long long m0) {
        Stage s;
        long long my = m0 + yr, m0x = m0 + xr, m1x = m0 + xr + 16;
        my = my < me ? my : me - 1; m0x = m0x < me ? m0x : me - 1; m1x = m1x < me ? m1x : me - 1;
        const float* yrow = p.dY + (DENSE ? my * p.ym.ld : row_off(p.ym, my));
        if constexpr (VYW == 4) s.vy = *reinterpret_cast<const f32x4*>(yrow + nl);
        else {
            const f32x2 lo = *reinterpret_cast<const f32x2*>(yrow + nl), hi = *reinterpret_cast<const f32x2*>(yrow + nl2);
            s.vy = f32x4{lo[0], lo[1], hi[0], hi[1]};
        }
        s.vx0 = *reinterpret_cast<const f32x4*>(p.X + (DENSE ? m0x * p.xm.ld : row_off(p.xm, m0x)) + kl);
        s.vx1 = *reinterpret_cast<const f32x4*>(p.X + (DENSE ? m1x * p.xm.ld : row_off(p.xm, m1x)) + kl);
        if constexpr (SCALED) {                               // rows < 2^31 (checked by the launcher): 32-bit division
            const unsigned i0 = (unsigned)m0x / (unsigned)p.x_scale_rpi, i1 = (unsigned)m1x / (unsigned)p.x_scale_rpi;
            s.g0 = *reinterpret_cast<const f32x4*>(p.x_scale + (long long)i0 * p.K + kl);
            s.g1 = *reinterpret_cast<const f32x4*>(p.x_scale + (long long)i1 * p.K + kl);
        }
        return s;
    };
    const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f}, one4 = f32x4{1.f, 0.f, 0.f, 0.f};
    // a slice past the end of M (rows_per_slice is rounded up to whole steps) is empty: no step runs and its partial row is
    // written as zeros below - the second stage adds every row.  The clamps above stay valid (me - 1 = M - 1 then).
    Stage s0 = fetch(mb), s1 = fetch(mb + RT);
    // one step: the stage's registers go to LDS (masks applied here), are refilled at once with the rows two steps ahead, and the
    // tile is multiplied.  Two stages ping-pong without register copies (a copy `s0 = s1` would wait for s1's loads).
    auto step = [&](Stage& sg, long long m0) {
        __syncthreads();                                      // the previous step's operands have been consumed
        {
            const bool ry = m0 + yr < me, r0 = m0 + xr < me, r1 = m0 + xr + 16 < me;
            if constexpr (SCALED) { sg.vx0 *= sg.g0; sg.vx1 *= sg.g1; }
            if constexpr (VYW == 4) *reinterpret_cast<f32x4*>(sY + yr * SY + yc) = (ry && nok) ? sg.vy : zero4;
            else {
                const bool lo = ry && nok, hi = ry && nok2;
                *reinterpret_cast<f32x4*>(sY + yr * SY + yc) = f32x4{lo ? sg.vy[0] : 0.f, lo ? sg.vy[1] : 0.f, hi ? sg.vy[2] : 0.f, hi ? sg.vy[3] : 0.f};
            }
            *reinterpret_cast<f32x4*>(sX + xr * SX + xc) = r0 ? (kok ? sg.vx0 : (kone ? one4 : zero4)) : zero4;
            *reinterpret_cast<f32x4*>(sX + (xr + 16) * SX + xc) = r1 ? (kok ? sg.vx1 : (kone ? one4 : zero4)) : zero4;
        }
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        sg = fetch(m0 + 2 * RT);                              // clamped past the end of the slice, masked when stored
        __builtin_amdgcn_sched_barrier(0);                    // the loads go out HERE, before the tile is multiplied
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const int r = 8 * wave + 4 * st + g;
            float a[2], b[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = sY[r * SY + 16 * i + c16];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = sX[r * SX + 16 * j + c16];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };
    for (long long m0 = mb; m0 < me; m0 += 2 * RT) {
        step(s0, m0);
        step(s1, m0 + RT);                                    // past the end of the slice: an all-zero tile
    }
    __syncthreads();                                          // the staging tiles are dead: their LDS becomes the epilogue buffer
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) sm[wave][(16 * i + 4 * g + r) * 64 + 16 * j + c16] = acc[i][j][r];
    __syncthreads();
    float* out = p.partial + (long long)blockIdx.z * p.N * Kx;
    for (int e = threadIdx.x; e < 32 * 64; e += 256) {
        const int nn = n0 + e / 64, kq = k0 + e % 64;
        if (nn < p.N && kq < Kx) out[(long long)nn * Kx + kq] = ((sm[0][e] + sm[1][e]) + sm[2][e]) + sm[3][e];
    }
}

// Wide-N form of the pipelined kernel (N >= 256: the class head's 810 columns): a workgroup covers 128 n x 64 k and each wave owns
// 32 of the n columns over ALL 32 rows of a step - no cross-wave reduction, X is re-read by N / 128 workgroups instead of N / 32,
// and a row of dY is fetched in 512-byte runs (the 32-column form read 128 bytes per row and workgroup out of 3 240-byte rows).
template <bool DENSE, int VYW>
__global__ __launch_bounds__(256) void gemm_tn_kernel_w(GemmTnArgs p) {
    constexpr int RT = 32, SY = 144, SX = 80;
    __shared__ float sY[RT * SY];
    __shared__ float sX[RT * SX];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c16 = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 128, k0 = blockIdx.y * 64;
    const long long mb = (long long)blockIdx.z * p.rows_per_slice;
    long long me = mb + p.rows_per_slice;
    if (me > p.M) me = p.M;
    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int Kx = p.K + 1;
    const int yr = tid >> 3, yc = (tid & 7) * 16;             // dY: row yr, 16 columns from yc
    const int xr = tid >> 4, xc = (tid & 15) * 4;
    const int kk = k0 + xc;
    const bool kok = kk < p.K, kone = kk == p.K;
    const int kl = kok ? kk : 0;
    constexpr int NP = 16 / VYW;                              // pieces of VYW floats
    int nl[NP];
    bool nok[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) { const int n = n0 + yc + VYW * q; nok[q] = n < p.N; nl[q] = nok[q] ? n : 0; }
    struct Stage { float vy[16]; f32x4 vx0, vx1; };
    auto fetch = [&](long long m0) {
        Stage s;
        long long my = m0 + yr, m0x = m0 + xr, m1x = m0 + xr + 16;
        my = my < me ? my : me - 1; m0x = m0x < me ? m0x : me - 1; m1x = m1x < me ? m1x : me - 1;
        const float* yrow = p.dY + (DENSE ? my * p.ym.ld : row_off(p.ym, my));
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            if constexpr (VYW == 4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(yrow + nl[q]);
                s.vy[4 * q] = v[0]; s.vy[4 * q + 1] = v[1]; s.vy[4 * q + 2] = v[2]; s.vy[4 * q + 3] = v[3];
            } else {
                const f32x2 v = *reinterpret_cast<const f32x2*>(yrow + nl[q]);
                s.vy[2 * q] = v[0]; s.vy[2 * q + 1] = v[1];
            }
        }
        s.vx0 = *reinterpret_cast<const f32x4*>(p.X + (DENSE ? m0x * p.xm.ld : row_off(p.xm, m0x)) + kl);
        s.vx1 = *reinterpret_cast<const f32x4*>(p.X + (DENSE ? m1x * p.xm.ld : row_off(p.xm, m1x)) + kl);
        return s;
    };
    const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f}, one4 = f32x4{1.f, 0.f, 0.f, 0.f};
    Stage s0 = fetch(mb), s1 = fetch(mb + RT);
    auto step = [&](Stage& sg, long long m0) {
        __syncthreads();
        {
            const bool ry = m0 + yr < me, r0 = m0 + xr < me, r1 = m0 + xr + 16 < me;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (ry && nok[(4 * q4 + e) / VYW]) ? sg.vy[4 * q4 + e] : 0.f;
                *reinterpret_cast<f32x4*>(sY + yr * SY + yc + 4 * q4) = v;
            }
            *reinterpret_cast<f32x4*>(sX + xr * SX + xc) = r0 ? (kok ? sg.vx0 : (kone ? one4 : zero4)) : zero4;
            *reinterpret_cast<f32x4*>(sX + (xr + 16) * SX + xc) = r1 ? (kok ? sg.vx1 : (kone ? one4 : zero4)) : zero4;
        }
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        sg = fetch(m0 + 2 * RT);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int st = 0; st < 8; ++st) {
            const int r = 4 * st + g;
            float a[2], b[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = sY[r * SY + 32 * wave + 16 * i + c16];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = sX[r * SX + 16 * j + c16];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };
    for (long long m0 = mb; m0 < me; m0 += 2 * RT) {
        step(s0, m0);
        step(s1, m0 + RT);
    }
    // D[row = n local (4g + r)][col = k local (c16)]: every wave writes its own 32 x 64 block of the slice's partial row
    float* out = p.partial + (long long)blockIdx.z * p.N * Kx;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int nn = n0 + 32 * wave + 16 * i + 4 * g + r, kq = k0 + 16 * j + c16;
                if (nn < p.N && kq < Kx) out[(long long)nn * Kx + kq] = acc[i][j][r];
            }
}

// out[g][l] (+)= sum_s in[g][s][l] in index order
// trC > 0: the first trC * trT entries of a row are a [trT][trC] matrix that is written transposed, [trC][trT] (depthwise tap
// gradients [k*k][C] -> the parameter's [C][k*k]); entries behind it keep their place
struct ReduceMidArgs { const float* in; float* out; long long L; int G, S, accumulate; float alpha; int trC, trT; };
__global__ __launch_bounds__(256) void reduce_mid_kernel(ReduceMidArgs p) {
    // 16 consecutive l per workgroup x 16 lanes over s: lane j sums s = j, j+16, ... in order, then the 16 lane sums are
    // added in lane order - a fixed association, so the result is bitwise reproducible
    __shared__ float sm[16][17];
    const int li = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const long long l = (long long)blockIdx.x * 16 + li;
    const int gi = blockIdx.y;
    float s = 0.f;
    if (l < p.L) {
        const float* src = p.in + (long long)gi * p.S * p.L + l;
        int i = sl;
        for (; i + 48 < p.S; i += 64) {
            const float a0 = src[(long long)i * p.L], a1 = src[(long long)(i + 16) * p.L];
            const float a2 = src[(long long)(i + 32) * p.L], a3 = src[(long long)(i + 48) * p.L];
            s += a0; s += a1; s += a2; s += a3;
        }
        for (; i < p.S; i += 16) s += src[(long long)i * p.L];
    }
    sm[sl][li] = s;
    __syncthreads();
    if (sl == 0 && l < p.L) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) t += sm[j][li];
        long long o = l;
        if (p.trC > 0 && l < (long long)p.trC * p.trT) { const long long tt = l / p.trC; o = (l - tt * p.trC) * p.trT + tt; }
        float* dst = p.out + (long long)gi * p.L + o;
        t *= p.alpha;
        *dst = p.accumulate ? *dst + t : t;
    }
}

// Second stage of a BatchNorm column reduction with the layer's bookkeeping in the same launch (one launch less per BN):
//   kind 0 (forward, after the centred second pass): var[c] = alpha * sum_s partial[s][c], then effdet_train_bn_finalize's
//           arithmetic (running statistics, rstd, scale, shift) with the batch mean given;
//   kind 1 (backward, partial rows [2][C] = {sum dy, sum dy (c - mean)}): effdet_train_bn_bwd_prep's d gamma, d beta, v1, v3.
// Same fixed summation order as reduce_mid_kernel (16 lanes over s, lane sums added in lane order).
struct ReduceBnArgs {
    const float* in; int S, C, kind; float alpha;
    const float* mean; const float* gamma; const float* beta; float* running_mean; float* running_var; long long* nbt;
    float momentum, unbias, eps; float* scale; float* shift; float* rstd_out;
    const float* rstd_in; float invM; float* dgamma; float* dbeta; float* v1; float* v3;
};
__global__ __launch_bounds__(256) void reduce_bn_kernel(ReduceBnArgs p) {
    __shared__ float sm[2][16][17];
    const int li = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + li;
    const int W = p.kind == 1 ? 2 : 1;
    float s0 = 0.f, s1 = 0.f;
    if (c < p.C) {
        for (int i = sl; i < p.S; i += 16) {
            s0 += p.in[((long long)i * W) * p.C + c];
            if (W == 2) s1 += p.in[((long long)i * W + 1) * p.C + c];
        }
    }
    sm[0][sl][li] = s0;
    sm[1][sl][li] = s1;
    __syncthreads();
    if (sl != 0 || c >= p.C) return;
    float t0 = 0.f, t1 = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) { t0 += sm[0][j][li]; t1 += sm[1][j][li]; }
    if (p.kind == 0) {
        if (c == 0 && p.nbt) *p.nbt += 1;
        const float m = p.mean[c], v = t0 * p.alpha;
        p.running_mean[c] = p.running_mean[c] * (1.0f - p.momentum) + p.momentum * m;
        p.running_var[c] = p.running_var[c] * (1.0f - p.momentum) + p.momentum * (v * p.unbias);
        const float rs = 1.0f / sqrtf(v + p.eps);
        const float sc = p.gamma[c] * rs;
        p.rstd_out[c] = rs;
        p.scale[c] = sc;
        p.shift[c] = p.beta[c] - m * sc;
    } else {
        const float rs = p.rstd_in[c];
        p.dgamma[c] = t1 * rs;
        p.dbeta[c] = t0;
        p.v1[c] = t0 * p.invM;
        p.v3[c] = rs * rs * t1 * p.invM;
    }
}

// second stage of the weight-gradient GEMM: partial [S][N][K+1] -> dense dW [N][K] followed by dsum [N] (same fixed
// summation order as reduce_mid_kernel)
struct ReduceSplitArgs { const float* in; float* out; int N, K, S; };
__global__ __launch_bounds__(256) void reduce_split_kernel(ReduceSplitArgs p) {
    __shared__ float sm[16][17];
    const int li = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const long long L = (long long)p.N * (p.K + 1);
    const long long l = (long long)blockIdx.x * 16 + li;
    float s = 0.f;
    if (l < L) {
        const float* src = p.in + l;
        int i = sl;
        for (; i + 48 < p.S; i += 64) {
            const float a0 = src[(long long)i * L], a1 = src[(long long)(i + 16) * L];
            const float a2 = src[(long long)(i + 32) * L], a3 = src[(long long)(i + 48) * L];
            s += a0; s += a1; s += a2; s += a3;
        }
        for (; i < p.S; i += 16) s += src[(long long)i * L];
    }
    sm[sl][li] = s;
    __syncthreads();
    if (sl == 0 && l < L) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) t += sm[j][li];
        const long long n = l / (p.K + 1);
        const int kk = (int)(l - n * (p.K + 1));
        if (kk < p.K) p.out[n * p.K + kk] = t; else p.out[(long long)p.N * p.K + n] = t;
    }
}

}  // namespace
// out[g][l] (+)= alpha * sum_s in[g][s][l] in fixed order (declared in common.h: the second stage of every two-stage reduction)
int effdet_launch_reduce_mid_tr(hipStream_t st, const float* in, int G, int S, long long L, float* out, int accumulate, float alpha,
                                int trC, int trT) {
    ReduceMidArgs a{in, out, L, G, S, accumulate, alpha, trC, trT};
    const long long blocks = (L + 15) / 16;
    if (blocks > 0x7fffffffLL || G > 65535) return EFFDET_EINVAL;
    hipLaunchKernelGGL(reduce_mid_kernel, dim3((unsigned)blocks, (unsigned)G), dim3(256), 0, st, a);
    return effdet_check_launch();
}
int effdet_launch_reduce_mid(hipStream_t st, const float* in, int G, int S, long long L, float* out, int accumulate, float alpha) {
    return effdet_launch_reduce_mid_tr(st, in, G, S, L, out, accumulate, alpha, 0, 0);
}
namespace {
inline int launch_reduce_mid(hipStream_t st, const float* in, int G, int S, long long L, float* out, int accumulate, float alpha = 1.0f) {
    return effdet_launch_reduce_mid(st, in, G, S, L, out, accumulate, alpha);
}

DEV float silu_grad(float z) { const float s = sigmoid_train(z); return s * (1.0f + z * (1.0f - s)); }

// ------------------------------------------------------------------------------------------------------------
// depthwise backward
// ------------------------------------------------------------------------------------------------------------
struct DwBwdArgs {
    const float* dY; const float* X; const float* taps; float* dX; float* partial;
    const float* Z;                            // dx kernel, optional: dX is multiplied by silu'(Z) (Z = pre-activation of the layer below)
    int B, H, W, C, k, stride, Ho, Wo, pad_t, pad_l; long long segs_per_chunk; int seg;
};

// dX[b,iy,ix,c] = sum_taps dY[b,(iy+pad-ky)/s,(ix+pad-kx)/s,c] * w[ky,kx,c]   (4 channels per thread)
__global__ __launch_bounds__(256) void dw_bwd_dx_kernel(DwBwdArgs p) {
    const int C4 = p.C / 4;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)p.B * p.H * p.W * C4;
    if (i >= total) return;
    const int c = (int)(i % C4) * 4;
    long long px = i / C4;
    const int ix = (int)(px % p.W); px /= p.W;
    const int iy = (int)(px % p.H);
    const int b = (int)(px / p.H);
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int ky = 0; ky < p.k; ++ky) {
        const int ty = iy + p.pad_t - ky;
        if (ty < 0 || ty % p.stride) continue;
        const int oy = ty / p.stride;
        if (oy >= p.Ho) continue;
        for (int kx = 0; kx < p.k; ++kx) {
            const int tx = ix + p.pad_l - kx;
            if (tx < 0 || tx % p.stride) continue;
            const int ox = tx / p.stride;
            if (ox >= p.Wo) continue;
            const f32x4 d = *reinterpret_cast<const f32x4*>(p.dY + (((long long)b * p.Ho + oy) * p.Wo + ox) * p.C + c);
            const f32x4 w = *reinterpret_cast<const f32x4*>(p.taps + (long long)(ky * p.k + kx) * p.C + c);
            acc += d * w;
        }
    }
    const long long o = (((long long)b * p.H + iy) * p.W + ix) * p.C + c;
    if (p.Z) {
        const f32x4 z = *reinterpret_cast<const f32x4*>(p.Z + o);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] *= silu_grad(z[j]);
    }
    *reinterpret_cast<f32x4*>(p.dX + o) = acc;
}

// Stride 1: a thread owns 4 consecutive pixels of a row (and 4 channels): per tap row it loads the 4 + k - 1 values of dY those
// pixels share and the k taps once - (4 + k - 1 + k) / 4 loads per pixel and tap row instead of 2 k.  (The one-pixel form above
// moves k * k * 2 vectors per output through L1 / L2: ~6.6 TB/s of cache traffic for 2.2 TB/s of HBM traffic on the 160 x 160 maps.)
template <int KS>
__global__ __launch_bounds__(256) void dw_bwd_dx_s1_kernel(DwBwdArgs p) {
    constexpr int PX = 4, NW = PX + KS - 1;
    const int C4 = p.C / 4;
    const int strips = (p.W + PX - 1) / PX;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)p.B * p.H * strips * C4;
    if (i >= total) return;
    const int c = (int)(i % C4) * 4;
    long long q = i / C4;
    const int xs = (int)(q % strips); q /= strips;
    const int iy = (int)(q % p.H);
    const long long b = q / p.H;
    const int ix0 = xs * PX;
    f32x4 acc[PX];
#pragma unroll
    for (int u = 0; u < PX; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < KS; ++ky) {
        const int oy = iy + p.pad_t - ky;                     // stride 1: Ho == H
        if (oy < 0 || oy >= p.Ho) continue;
        const float* drow = p.dY + ((b * p.Ho + oy) * p.Wo) * p.C + c;
        f32x4 d[NW], w[KS];
        // d[j] = dY[oy][ix0 + pad_l - (KS - 1) + j]
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int ox = ix0 + p.pad_l - (KS - 1) + j;
            d[j] = (ox >= 0 && ox < p.Wo) ? *reinterpret_cast<const f32x4*>(drow + (long long)ox * p.C) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) w[kx] = *reinterpret_cast<const f32x4*>(p.taps + (long long)(ky * KS + kx) * p.C + c);
        // pixel ix0 + u, tap kx reads dY column ix0 + u + pad_l - kx = d[u + KS - 1 - kx]
#pragma unroll
        for (int kx = 0; kx < KS; ++kx)
#pragma unroll
            for (int u = 0; u < PX; ++u) acc[u] += d[u + KS - 1 - kx] * w[kx];
    }
#pragma unroll
    for (int u = 0; u < PX; ++u) {
        const int ix = ix0 + u;
        if (ix >= p.W) break;
        const long long o = ((b * p.H + iy) * p.W + ix) * p.C + c;
        f32x4 v = acc[u];
        if (p.Z) {
            const f32x4 z = *reinterpret_cast<const f32x4*>(p.Z + o);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] *= silu_grad(z[j]);
        }
        *reinterpret_cast<f32x4*>(p.dX + o) = v;
    }
}

// partial[chunk][k*k+1][C]: taps gradient + sum of dY.  block = 64 channels x 4 lanes; a lane walks row segments of
// `seg` output pixels with a k x k register window of X that slides by the stride: S*k loads per output pixel instead
// of k*k (the kernel is load-issue bound).
template <int KK, int S>
__global__ __launch_bounds__(256) void dw_bwd_dw_kernel(DwBwdArgs p) {
    constexpr int T = KK * KK;
    __shared__ float sm[4][T + 1][64];
    const int cl = threadIdx.x & 63, pl = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cl;
    const bool cv = c < p.C;
    float acc[T + 1];
#pragma unroll
    for (int t = 0; t <= T; ++t) acc[t] = 0.f;
    const int segs_x = (p.Wo + p.seg - 1) / p.seg;
    const long long nseg = (long long)p.B * p.Ho * segs_x;
    const long long sb = (long long)blockIdx.x * p.segs_per_chunk;
    long long se = sb + p.segs_per_chunk;
    if (se > nseg) se = nseg;
    if (cv) {
        for (long long sg = sb + pl; sg < se; sg += 4) {
            const int sx = (int)(sg % segs_x);
            const long long r = sg / segs_x;
            const int oy = (int)(r % p.Ho);
            const long long b = r / p.Ho;
            const int x0 = sx * p.seg;
            const int x1 = x0 + p.seg < p.Wo ? x0 + p.seg : p.Wo;
            const float* Xb = p.X + b * p.H * p.W * p.C + c;
            const float* dYr = p.dY + (b * p.Ho + oy) * p.Wo * p.C + c;
            long long rowoff[KK];
            bool rv[KK];
#pragma unroll
            for (int ky = 0; ky < KK; ++ky) {
                const int iy = oy * S + ky - p.pad_t;
                rv[ky] = iy >= 0 && iy < p.H;
                rowoff[ky] = (long long)(rv[ky] ? iy : 0) * p.W * p.C;
            }
            float xw[KK][KK];
#pragma unroll
            for (int ky = 0; ky < KK; ++ky)
#pragma unroll
                for (int kx = S; kx < KK; ++kx) {
                    const int ix = x0 * S + (kx - S) - p.pad_l;
                    xw[ky][kx] = (rv[ky] && ix >= 0 && ix < p.W) ? Xb[rowoff[ky] + (long long)ix * p.C] : 0.f;
                }
            for (int ox = x0; ox < x1; ++ox) {
#pragma unroll
                for (int ky = 0; ky < KK; ++ky) {
#pragma unroll
                    for (int kx = 0; kx < KK - S; ++kx) xw[ky][kx] = xw[ky][kx + S];
#pragma unroll
                    for (int kx = KK - S; kx < KK; ++kx) {
                        const int ix = ox * S + kx - p.pad_l;
                        xw[ky][kx] = (rv[ky] && ix >= 0 && ix < p.W) ? Xb[rowoff[ky] + (long long)ix * p.C] : 0.f;
                    }
                }
                const float d = dYr[(long long)ox * p.C];
                acc[T] += d;
#pragma unroll
                for (int ky = 0; ky < KK; ++ky)
#pragma unroll
                    for (int kx = 0; kx < KK; ++kx) acc[ky * KK + kx] += d * xw[ky][kx];
            }
        }
    }
#pragma unroll
    for (int t = 0; t <= T; ++t) sm[pl][t][cl] = acc[t];
    __syncthreads();
    if (pl == 0 && cv) {
        float* out = p.partial + (long long)blockIdx.x * (T + 1) * p.C;
#pragma unroll
        for (int t = 0; t <= T; ++t) out[(long long)t * p.C + c] = ((sm[0][t][cl] + sm[1][t][cl]) + sm[2][t][cl]) + sm[3][t][cl];
    }
}

// ------------------------------------------------------------------------------------------------------------
// depthwise forward of the training path: conv_dw + folded BN -> Z (pre-activation), A = silu(Z), SE pool partial rows of A.
// A thread owns 4 consecutive output pixels of a row and 4 channels and, per tap row, loads the input values those pixels share
// once (4 S + k - S of them) together with the k taps.  Workgroup = 64 channels (16 quads) x 16 strip lanes, 8 strips per lane;
// pool partial row = one workgroup's sum over its 512 output pixels (lanes added in lane order).
// ------------------------------------------------------------------------------------------------------------
struct DwFwdArgs {
    const float* X; float* Z; float* A; const float* taps; const float* scale; const float* shift; float* pool_partial;
    int B, H, W, C, Ho, Wo, pad_t, pad_l, strips_x, nstrips, blocks_per_image;
};
constexpr int DWF_PX = 4, DWF_SPL = 8;                       // pixels per strip, strips per lane

template <int KS, int S>
__global__ __launch_bounds__(256) void dw_fwd_train_kernel(DwFwdArgs p) {
    constexpr int PX = DWF_PX, NW = PX * S + KS - S;
    __shared__ float sm[16][64];
    const int cq = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int c = blockIdx.z * 64 + cq * 4;
    const bool cv = c < p.C;
    const long long b = blockIdx.y;
    f32x4 pool = f32x4{0.f, 0.f, 0.f, 0.f};
    if (cv) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(p.scale + c), sh = *reinterpret_cast<const f32x4*>(p.shift + c);
        const float* Xb = p.X + b * p.H * p.W * p.C + c;
        for (int it = 0; it < DWF_SPL; ++it) {
            const int strip = (blockIdx.x * DWF_SPL + it) * 16 + sl;
            if (strip >= p.nstrips) break;
            const int oy = strip / p.strips_x, ox0 = (strip - oy * p.strips_x) * PX;
            f32x4 acc[PX];
#pragma unroll
            for (int u = 0; u < PX; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ky = 0; ky < KS; ++ky) {
                const int iy = oy * S + ky - p.pad_t;
                if (iy < 0 || iy >= p.H) continue;
                const float* xrow = Xb + (long long)iy * p.W * p.C;
                f32x4 x[NW], w[KS];
#pragma unroll
                for (int j = 0; j < NW; ++j) {
                    const int ix = ox0 * S - p.pad_l + j;
                    x[j] = (ix >= 0 && ix < p.W) ? *reinterpret_cast<const f32x4*>(xrow + (long long)ix * p.C) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) w[kx] = *reinterpret_cast<const f32x4*>(p.taps + (long long)(ky * KS + kx) * p.C + c);
#pragma unroll
                for (int kx = 0; kx < KS; ++kx)
#pragma unroll
                    for (int u = 0; u < PX; ++u) acc[u] += x[u * S + kx] * w[kx];
            }
#pragma unroll
            for (int u = 0; u < PX; ++u) {
                const int ox = ox0 + u;
                if (ox >= p.Wo) break;
                const long long o = ((b * p.Ho + oy) * p.Wo + ox) * p.C + c;
                const f32x4 z = acc[u] * sc + sh;
                *reinterpret_cast<f32x4*>(p.Z + o) = z;
                if (p.A) {
                    f32x4 a;
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[j] = silu_train(z[j]);
                    *reinterpret_cast<f32x4*>(p.A + o) = a;
                    pool += a;
                }
            }
        }
    }
    if (p.pool_partial == nullptr) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) sm[sl][cq * 4 + j] = pool[j];
    __syncthreads();
    const int cl = threadIdx.x;
    if (cl < 64 && blockIdx.z * 64 + cl < p.C) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sm[k][cl];
        p.pool_partial[(b * p.blocks_per_image + blockIdx.x) * p.C + blockIdx.z * 64 + cl] = t;
    }
}

// ------------------------------------------------------------------------------------------------------------
// element-wise family (4 elements per thread; channel = index % C, image = index / (hw*C))
// ------------------------------------------------------------------------------------------------------------
struct EwArgs {
    int op; float* out; const float* a; const float* b; const float* c;
    const float* v0; const float* v1; const float* v2; const float* v3;
    float s0, s1, s2, s3; long long n; int C; long long hwC;
    const float* sdev;                         // optional device float[4] overriding s0..s3 (graph-capturable steps)
    float* out2;                               // optional second output: silu(out)
};

// second derivative of z * sigmoid(z): sigma (1 - sigma) (2 + z (1 - 2 sigma))   (double backward of the MetaHead, infer.py:658)
DEV float silu_grad2(float z) { const float s = sigmoid_train(z); return s * (1.0f - s) * (2.0f + z * (1.0f - 2.0f * s)); }

__global__ __launch_bounds__(256) void ew_kernel(EwArgs p) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= p.n) return;
    if (p.sdev) { p.s0 = p.sdev[0]; p.s1 = p.sdev[1]; p.s2 = p.sdev[2]; p.s3 = p.sdev[3]; }
    const int ch = (int)(i % p.C);
    const f32x4 a = *reinterpret_cast<const f32x4*>(p.a + i);
    f32x4 o;
    switch (p.op) {
    case 0:                                             // SiLU forward
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = silu_train(a[j]);
        break;
    case 1: {                                           // SiLU backward: a = z, b = d(out)
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.b + i);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = b[j] * silu_grad(a[j]);
        break; }
    case 2: {                                           // a + b
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.b + i);
        o = a + b;
        break; }
    case 3: {                                           // per-channel affine: a * v0[c] + v1[c]   (v1 may be NULL)
        const f32x4 s = *reinterpret_cast<const f32x4*>(p.v0 + ch);
        o = a * s;
        if (p.v1) o += *reinterpret_cast<const f32x4*>(p.v1 + ch);
        break; }
    case 4: {                                           // SE gate forward: a * v0[img, c]
        const long long img = i / p.hwC;
        o = a * *reinterpret_cast<const f32x4*>(p.v0 + img * p.C + ch);
        break; }
    case 5: {                                           // SE gate backward: a * gate[img, c] + ds[img, c] * s0
        const long long img = i / p.hwC;
        o = a * *reinterpret_cast<const f32x4*>(p.v0 + img * p.C + ch)
            + *reinterpret_cast<const f32x4*>(p.v1 + img * p.C + ch) * p.s0;
        break; }
    case 6: {                                           // BN (batch statistics) backward:
        // d conv = v0[c] * (dy - v1[c] - (conv - v2[c]) * v3[c]);  v0 = gamma*rstd, v1 = sum(dy)/M, v2 = mean,
        // v3 = rstd^2 * sum(dy*(conv-mean))/M;  a = dy, b = conv output
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.b + i);
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(p.v0 + ch), v1 = *reinterpret_cast<const f32x4*>(p.v1 + ch);
        const f32x4 v2 = *reinterpret_cast<const f32x4*>(p.v2 + ch), v3 = *reinterpret_cast<const f32x4*>(p.v3 + ch);
        o = v0 * (a - v1 - (b - v2) * v3);
        break; }
    case 7: {                                           // FpnCombine 'fastattn': sum_i (x_i * w_i) / den  (efficientdet.py:240-242)
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.b + i);
        o = (a * p.s0) / p.s3 + (b * p.s1) / p.s3;
        if (p.c) o = o + (*reinterpret_cast<const f32x4*>(p.c + i) * p.s2) / p.s3;
        break; }
    case 8:                                             // a * s0
        o = a * p.s0;
        break;
    case 10: {                                          // a * b
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.b + i);
        o = a * b;
        break; }
    case 11: {                                          // a * b * silu''(c)
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.b + i);
        const f32x4 z = *reinterpret_cast<const f32x4*>(p.c + i);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = a[j] * b[j] * silu_grad2(z[j]);
        break; }
    case 12: {                                          // SE gate backward, then SiLU backward: (a * gate + ds * s0) * silu'(c)
        const long long img = i / p.hwC;
        const f32x4 z = *reinterpret_cast<const f32x4*>(p.c + i);
        o = a * *reinterpret_cast<const f32x4*>(p.v0 + img * p.C + ch)
            + *reinterpret_cast<const f32x4*>(p.v1 + img * p.C + ch) * p.s0;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] *= silu_grad(z[j]);
        break; }
    case 13: {                                          // stochastic depth + shortcut: a * v0[img, c] + b
        const long long img = i / p.hwC;
        o = a * *reinterpret_cast<const f32x4*>(p.v0 + img * p.C + ch) + *reinterpret_cast<const f32x4*>(p.b + i);
        break; }
    default:                                            // 9: weighted sum without the division ('attn' / 'sum')
    {
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.b + i);
        o = a * p.s0 + b * p.s1;
        if (p.c) o = o + *reinterpret_cast<const f32x4*>(p.c + i) * p.s2;
        break; }
    }
    *reinterpret_cast<f32x4*>(p.out + i) = o;
    if (p.out2) {
        f32x4 q;
#pragma unroll
        for (int j = 0; j < 4; ++j) q[j] = silu_train(o[j]);
        *reinterpret_cast<f32x4*>(p.out2 + i) = q;
    }
}

// ------------------------------------------------------------------------------------------------------------
// per-channel reductions over the rows of [G][R][C] tensors (two-stage)
// ------------------------------------------------------------------------------------------------------------
struct ColArgs {
    int mode; const float* a; const float* b; const float* v; float* partial;
    long long R, rows_per_slice; int C, S;
};
// modes: 0 sum a; 1 sum a*b; 2 sum (a - v[c])^2; 3 sum a*(b - v[c]); 4: both 0 and 3 in one pass (out [G][2][C])
// V = channels per thread (4: 16-byte loads, a wave reads 4 whole rows of a 64-channel group per instruction; 1: any C).
// Workgroup = 64 channels x (256 * V / 64) row lanes; the row lanes' sums are added in lane order (fixed association).
template <int V> struct ColVec;
template <> struct ColVec<1> { typedef float type; static DEV float get(float v, int) { return v; } };
template <> struct ColVec<4> { typedef f32x4 type; static DEV float get(const f32x4& v, int j) { return v[j]; } };
template <int V>
__global__ __launch_bounds__(256) void col_reduce_kernel(ColArgs p) {
    constexpr int TPG = 64 / V, RL = 256 / TPG;
    typedef typename ColVec<V>::type vec;
    __shared__ float sm[RL][64];
    __shared__ float sm2[RL][64];
    const int ct = threadIdx.x % TPG, rl = threadIdx.x / TPG;
    const int c = blockIdx.y * 64 + ct * V;
    const int gi = blockIdx.z, s = blockIdx.x;
    const bool cv = c < p.C;                                  // V > 1 only with C % V == 0: a vector is inside or outside as a whole
    const long long rb = (long long)s * p.rows_per_slice;
    long long re = rb + p.rows_per_slice;
    if (re > p.R) re = p.R;
    float acc[V], acc2[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { acc[j] = 0.f; acc2[j] = 0.f; }
    if (cv) {
        float vc[V];
#pragma unroll
        for (int j = 0; j < V; ++j) vc[j] = (p.mode >= 2) ? p.v[c + j] : 0.f;
        const long long base = (long long)gi * p.R * p.C + c;
        for (long long r = rb + rl; r < re; r += RL) {
            const vec a = *reinterpret_cast<const vec*>(p.a + base + r * p.C);
            vec b = a;
            if (p.mode == 1 || p.mode >= 3) b = *reinterpret_cast<const vec*>(p.b + base + r * p.C);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float aj = ColVec<V>::get(a, j), bj = ColVec<V>::get(b, j);
                if (p.mode == 0) acc[j] += aj;
                else if (p.mode == 1) acc[j] += aj * bj;
                else if (p.mode == 2) { const float d = aj - vc[j]; acc[j] += d * d; }
                else if (p.mode == 3) acc[j] += aj * (bj - vc[j]);
                else { acc[j] += aj; acc2[j] += aj * (bj - vc[j]); }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < V; ++j) { sm[rl][ct * V + j] = acc[j]; sm2[rl][ct * V + j] = acc2[j]; }
    __syncthreads();
    const int cl = threadIdx.x;
    if (cl < 64 && blockIdx.y * 64 + cl < p.C) {
        const int W = p.mode == 4 ? 2 : 1;
        float* dst = p.partial + ((long long)gi * p.S + s) * W * p.C + blockIdx.y * 64 + cl;
        float t = 0.f, t2 = 0.f;
#pragma unroll
        for (int k = 0; k < RL; ++k) { t += sm[k][cl]; t2 += sm2[k][cl]; }
        dst[0] = t;
        if (W == 2) dst[p.C] = t2;
    }
}

// ------------------------------------------------------------------------------------------------------------
// spatial helpers
// ------------------------------------------------------------------------------------------------------------
struct SpArgs { int op; const float* in; const float* aux; float* out; int B, H, W, C, Ho, Wo, pad_t, pad_l; };

__global__ __launch_bounds__(256) void spatial_kernel(SpArgs p) {
    const int C4 = p.C / 4;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p.op == 0) {                                    // nearest x2 upsample: in [B,H,W,C] -> out [B,2H,2W,C]
        const long long total = (long long)p.B * 2 * p.H * 2 * p.W * C4;
        if (i >= total) return;
        const int c = (int)(i % C4) * 4;
        long long px = i / C4;
        const int x = (int)(px % (2 * p.W)); px /= 2 * p.W;
        const int y = (int)(px % (2 * p.H));
        const long long b = px / (2 * p.H);
        *reinterpret_cast<f32x4*>(p.out + ((b * 2 * p.H + y) * 2 * p.W + x) * p.C + c) =
            *reinterpret_cast<const f32x4*>(p.in + ((b * p.H + y / 2) * p.W + x / 2) * p.C + c);
        return;
    }
    const long long total = (long long)p.B * p.H * p.W * C4;
    if (i >= total) return;
    const int c = (int)(i % C4) * 4;
    long long px = i / C4;
    const int x = (int)(px % p.W); px /= p.W;
    const int y = (int)(px % p.H);
    const long long b = px / p.H;
    if (p.op == 1) {                                    // upsample backward: in = d(out) [B,2H,2W,C] -> out [B,H,W,C]
        const float* s = p.in + ((b * 2 * p.H + 2 * y) * 2 * p.W + 2 * x) * p.C + c;
        const long long rs = (long long)2 * p.W * p.C;
        const f32x4 o = (*reinterpret_cast<const f32x4*>(s) + *reinterpret_cast<const f32x4*>(s + p.C)) +
                        (*reinterpret_cast<const f32x4*>(s + rs) + *reinterpret_cast<const f32x4*>(s + rs + p.C));
        *reinterpret_cast<f32x4*>(p.out + ((b * p.H + y) * p.W + x) * p.C + c) = o;
        return;
    }
    // op 2: 3x3/s2 TF-SAME max-pool backward.  in = pool input X [B,H,W,C], aux = dY [B,Ho,Wo,C].  A pixel receives dY of
    // every window whose first maximum (row-major scan, strict >: torch.max_pool2d's choice) it is.
    const f32x4 me = *reinterpret_cast<const f32x4*>(p.in + ((b * p.H + y) * p.W + x) * p.C + c);
    f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int oy = (y + p.pad_t - 2 + 1) / 2; oy <= (y + p.pad_t) / 2; ++oy) {
        if (oy < 0 || oy >= p.Ho) continue;
        for (int ox = (x + p.pad_l - 2 + 1) / 2; ox <= (x + p.pad_l) / 2; ++ox) {
            if (ox < 0 || ox >= p.Wo) continue;
            // is (y, x) the first maximum of window (oy, ox)?  earlier positions must be strictly smaller, later ones <=
            bool first[4] = {true, true, true, true};
            for (int ky = 0; ky < 3; ++ky) {
                const int yy = oy * 2 + ky - p.pad_t;
                if (yy < 0 || yy >= p.H) continue;
                for (int kx = 0; kx < 3; ++kx) {
                    const int xx = ox * 2 + kx - p.pad_l;
                    if (xx < 0 || xx >= p.W || (yy == y && xx == x)) continue;
                    const f32x4 q = *reinterpret_cast<const f32x4*>(p.in + ((b * p.H + yy) * p.W + xx) * p.C + c);
                    const bool before = yy < y || (yy == y && xx < x);
#pragma unroll
                    for (int j = 0; j < 4; ++j) first[j] = first[j] && (before ? q[j] < me[j] : q[j] <= me[j]);
                }
            }
            const f32x4 d = *reinterpret_cast<const f32x4*>(p.aux + ((b * p.Ho + oy) * p.Wo + ox) * p.C + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] += first[j] ? d[j] : 0.f;
        }
    }
    *reinterpret_cast<f32x4*>(p.out + ((b * p.H + y) * p.W + x) * p.C + c) = o;
}

// conv_stem patches: X NCHW [B,3,H,W] -> col [B*Ho*Wo][32], k = (ky*3+kx)*3+ci, 27..31 zero (3x3/s2 TF-SAME)
struct Im2colArgs { const float* X; float* col; int B, H, W, Ho, Wo, pad_t, pad_l; };
__global__ __launch_bounds__(256) void im2col_stem_kernel(Im2colArgs p) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)p.B * p.Ho * p.Wo * 32;
    if (i >= total) return;
    const int k = (int)(i & 31);
    long long px = i >> 5;
    const int ox = (int)(px % p.Wo); px /= p.Wo;
    const int oy = (int)(px % p.Ho);
    const long long b = px / p.Ho;
    float v = 0.f;
    if (k < 27) {
        const int ci = k % 3, kx = (k / 3) % 3, ky = k / 9;
        const int iy = oy * 2 + ky - p.pad_t, ix = ox * 2 + kx - p.pad_l;
        if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) v = p.X[((b * 3 + ci) * p.H + iy) * p.W + ix];
    }
    p.col[i] = v;
}

// ------------------------------------------------------------------------------------------------------------
// SqueezeExcite backward (one 1024-thread workgroup per image: every phase is a short latency-bound loop)
// ------------------------------------------------------------------------------------------------------------
struct SeBwdArgs {
    const float* pool_sum; float inv_hw; const float* gate; const float* dgate;
    const float* W1; const float* b1; const float* W2t;
    float* ds; float* pgrad; int C, R;
};
// gate = sigmoid(u), u = W2 r + b2, r = silu(rp), rp = W1 s + b1, s = pool_sum / hw.
// pgrad[img] = { dW1 [R][C], db1 [R], dW2t [R][C], db2 [C] } for this image; ds[img][c] = d loss / d s.
__global__ __launch_bounds__(1024) void se_bwd_kernel(SeBwdArgs p) {
    extern __shared__ float sh[];
    float* s = sh;                    // [C]
    float* du = sh + p.C;             // [C]
    float* rp = du + p.C;             // [R]
    float* drp = rp + p.R;            // [R]
    const int img = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const long long bc = (long long)img * p.C;
    float* pg = p.pgrad + (long long)img * (2LL * p.R * p.C + p.R + p.C);
    float* dW1 = pg; float* db1 = pg + (long long)p.R * p.C; float* dW2t = db1 + p.R; float* db2 = dW2t + (long long)p.R * p.C;
    for (int c = tid; c < p.C; c += 1024) {
        s[c] = p.pool_sum[bc + c] * p.inv_hw;
        const float g = p.gate[bc + c];
        const float d = p.dgate[bc + c] * g * (1.0f - g);
        du[c] = d;
        db2[c] = d;
    }
    __syncthreads();
    for (int r = wave; r < p.R; r += 16) {
        float a = 0.f, d = 0.f;
        for (int c = lane; c < p.C; c += 64) {
            a += p.W1[(long long)r * p.C + c] * s[c];
            d += p.W2t[(long long)r * p.C + c] * du[c];
        }
        a = wave_reduce_sum(a);
        d = wave_reduce_sum(d);
        if (lane == 0) {
            const float z = a + p.b1[r];
            rp[r] = z;
            const float dz = d * silu_grad(z);
            drp[r] = dz;
            db1[r] = dz;
        }
    }
    __syncthreads();
    for (int c = tid; c < p.C; c += 1024) {
        float acc = 0.f;
        const float sc = s[c], dc = du[c];
        for (int r0 = 0; r0 < p.R; r0 += 8) {                 // 8 rows of W1 requested before the first is used (same r order)
            float w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w[u] = p.W1[(long long)(r0 + u < p.R ? r0 + u : p.R - 1) * p.C + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = r0 + u;
                if (r < p.R) {
                    const float dz = drp[r];
                    acc += w[u] * dz;
                    dW1[(long long)r * p.C + c] = dz * sc;
                    dW2t[(long long)r * p.C + c] = silu_train(rp[r]) * dc;
                }
            }
        }
        p.ds[bc + c] = acc;
    }
}


// ------------------------------------------------------------------------------------------------------------
// parameter-sized helpers of conv + BatchNorm(running statistics): one launch instead of a dozen tiny tensor ops
// ------------------------------------------------------------------------------------------------------------
struct FoldArgs {
    const float* W; int N, K; const float* gamma; const float* beta; const float* mean; const float* var; float eps;
    float* Wf; float* WfT; float* WT; float* scale; float* shift; float* rstd;
};
// scale = gamma * rsqrt(var + eps), shift = beta - mean * scale; Wf [N][K] = W * scale[n], WfT [K][N] its transpose,
// WT [K][N] = W transposed (each optional).  One workgroup per output channel.
__global__ __launch_bounds__(256) void fold_bn_kernel(FoldArgs p) {
    const int n = blockIdx.x;
    const float rs = 1.0f / sqrtf(p.var[n] + p.eps);
    const float sc = p.gamma[n] * rs;
    if (threadIdx.x == 0) {
        p.scale[n] = sc;
        p.shift[n] = p.beta[n] - p.mean[n] * sc;
        p.rstd[n] = rs;
    }
    for (int k = threadIdx.x; k < p.K; k += 256) {
        const float w = p.W[(long long)n * p.K + k];
        if (p.Wf) p.Wf[(long long)n * p.K + k] = w * sc;
        if (p.WfT) p.WfT[(long long)k * p.N + n] = w * sc;
        if (p.WT) p.WT[(long long)k * p.N + n] = w;
    }
}

struct FinArgs {
    const float* dWext; int N, K, transposed; const float* W; const float* scale; const float* rstd; const float* mean;
    float* dW; float* dgamma; float* dbeta;
};
// z = scale * conv(x; W) + shift, dWraw = dz^T x, dsum = sum dz  ->  dW = scale * dWraw,
// d gamma = rstd * (sum_k W * dWraw - mean * dsum), d beta = dsum.   dWext: [N][K] then [N] sums (or [(K+1)][N] when transposed).
__global__ __launch_bounds__(256) void convbn_grads_kernel(FinArgs p) {
    __shared__ float sm[4];
    const int n = blockIdx.x;
    const float sc = p.scale[n];
    float acc = 0.f;
    for (int k = threadIdx.x; k < p.K; k += 256) {
        const float v = p.transposed ? p.dWext[(long long)k * p.N + n] : p.dWext[(long long)n * p.K + k];
        p.dW[(long long)n * p.K + k] = sc * v;
        acc += p.W[(long long)n * p.K + k] * v;
    }
    acc = wave_reduce_sum(acc);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float tot = ((sm[0] + sm[1]) + sm[2]) + sm[3];
        const float dsum = p.dWext[(long long)p.K * p.N + n];      // both layouts keep the N sums behind the N*K gradients
        p.dgamma[n] = p.rstd[n] * (tot - p.mean[n] * dsum);
        p.dbeta[n] = dsum;
    }
}

struct BnFinArgs {
    const float* mean; const float* var; const float* gamma; const float* beta;
    float* running_mean; float* running_var; long long* nbt; int C, train; float momentum, unbias, eps;
    float* scale; float* shift; float* rstd;
};
// nn.BatchNorm2d bookkeeping in one launch: running statistics (training: r = (1-m) r + m * batch, unbiased variance),
// rstd = 1/sqrt(var + eps), scale = gamma * rstd, shift = beta - mean * scale
__global__ __launch_bounds__(256) void bn_finalize_kernel(BnFinArgs p) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c == 0 && p.train && p.nbt) *p.nbt += 1;
    if (c >= p.C) return;
    const float m = p.mean[c], v = p.var[c];
    if (p.train) {
        p.running_mean[c] = p.running_mean[c] * (1.0f - p.momentum) + p.momentum * m;
        p.running_var[c] = p.running_var[c] * (1.0f - p.momentum) + p.momentum * (v * p.unbias);
    }
    const float rs = 1.0f / sqrtf(v + p.eps);
    const float sc = p.gamma[c] * rs;
    p.rstd[c] = rs;
    p.scale[c] = sc;
    p.shift[c] = p.beta[c] - m * sc;
}

struct BnBwdArgs { const float* s1; const float* s2c; const float* rstd; int C; float invM; float* dgamma; float* dbeta; float* v1; float* v3; };
// d gamma = rstd * sum(dy (c - mean)), d beta = sum(dy); v1 = sum(dy)/M, v3 = rstd^2 * sum(dy (c - mean))/M for op 6 of the
// element-wise family
__global__ __launch_bounds__(256) void bn_bwd_prep_kernel(BnBwdArgs p) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= p.C) return;
    const float rs = p.rstd[c], s1 = p.s1[c], s2 = p.s2c[c];
    p.dgamma[c] = s2 * rs;
    p.dbeta[c] = s1;
    p.v1[c] = s1 * p.invM;
    p.v3[c] = rs * rs * s2 * p.invM;
}

// ------------------------------------------------------------------------------------------------------------
// Table-driven forms of the parameter-sized helpers: ONE launch for every conv of a stage instead of one per conv.
// ------------------------------------------------------------------------------------------------------------
struct PrepOp {                                 // mirrored by train_engine._PrepOp (ctypes)
    int kind, rows, cols; float eps;            // kind 0: transpose src [rows][cols] -> dst0 [cols][rows]; 1: fold_bn (rows = N, cols = K); 2: BiFPN edge weights
    const float* src; const float* gamma; const float* beta; const float* mean; const float* var;
    float* dst0; float* dst1; float* dst2; float* scale; float* shift; float* rstd;      // fold: Wf, WfT, WT (each optional)
};
__global__ __launch_bounds__(256) void prep_table_kernel(const PrepOp* ops, int n) {
    const PrepOp p = ops[blockIdx.y];
    const long long total = (long long)p.rows * p.cols;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int r = (int)(e / p.cols), c = (int)(e - (long long)r * p.cols);
        if (p.kind == 2) {                                   // BiFPN edge weights (rows = n inputs, cols = 1, eps = method): dst0 = {w0, w1, w2, den}
            if (e != 0) break;
            const int method = (int)p.eps;
            float w[3] = {0.f, 0.f, 0.f};
            float den = 1.0f;
            if (method == 0) {
                float sum = 0.f;
                for (int i = 0; i < p.rows; ++i) { w[i] = fmaxf(p.src[i], 0.f); sum += w[i]; }
                den = sum + 0.0001f;
            } else if (method == 1) {
                float m = p.src[0];
                for (int i = 1; i < p.rows; ++i) m = fmaxf(m, p.src[i]);
                float sum = 0.f;
                for (int i = 0; i < p.rows; ++i) { w[i] = expf(p.src[i] - m); sum += w[i]; }
                for (int i = 0; i < p.rows; ++i) w[i] = w[i] / sum;
            }
            p.dst0[0] = w[0]; p.dst0[1] = w[1]; p.dst0[2] = w[2]; p.dst0[3] = den;
            break;
        }
        const float w = p.src[e];
        if (p.kind == 0) { p.dst0[(long long)c * p.rows + r] = w; continue; }
        const float rs = 1.0f / sqrtf(p.var[r] + p.eps);
        const float sc = p.gamma[r] * rs;
        if (c == 0) { p.scale[r] = sc; p.shift[r] = p.beta[r] - p.mean[r] * sc; p.rstd[r] = rs; }
        if (p.dst0) p.dst0[e] = w * sc;
        if (p.dst1) p.dst1[(long long)c * p.rows + r] = w * sc;
        if (p.dst2) p.dst2[(long long)c * p.rows + r] = w;
    }
}

struct GradOp {                                 // mirrored by train_engine._GradOp: effdet_train_convbn_grads per table row
    const float* dWext; const float* W; const float* scale; const float* rstd; const float* mean;
    float* dW; float* dgamma; float* dbeta; int N, K, transposed, pad;
};
__global__ __launch_bounds__(256) void grads_table_kernel(const GradOp* ops, int n) {
    __shared__ float sm[4];
    const GradOp p = ops[blockIdx.y];
    const int nn = blockIdx.x;
    if (nn >= p.N) return;                       // uniform per workgroup
    const float sc = p.scale[nn];
    float acc = 0.f;
    for (int k = threadIdx.x; k < p.K; k += 256) {
        const float v = p.transposed ? p.dWext[(long long)k * p.N + nn] : p.dWext[(long long)nn * p.K + k];
        p.dW[(long long)nn * p.K + k] = sc * v;
        acc += p.W[(long long)nn * p.K + k] * v;
    }
    acc = wave_reduce_sum(acc);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float tot = ((sm[0] + sm[1]) + sm[2]) + sm[3];
        const float dsum = p.dWext[(long long)p.K * p.N + nn];
        p.dgamma[nn] = p.rstd[nn] * (tot - p.mean[nn] * dsum);
        p.dbeta[nn] = dsum;
    }
}

}  // namespace

// table: n PrepOp records in device memory (layout above); max_elems = the largest rows * cols in the table
extern "C" int effdet_train_prep_table(void* stream, const void* table, int n, long long max_elems) {
    EFFDET_ENTER();
    if (!table || n <= 0 || n > 65535 || max_elems <= 0) return EFFDET_EINVAL;
    long long gx = (max_elems + 255) / 256;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(prep_table_kernel, dim3((unsigned)gx, (unsigned)n), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       static_cast<const PrepOp*>(table), n);
    return effdet_check_launch();
}

// table: n GradOp records in device memory; max_n = the largest N in the table
extern "C" int effdet_train_grads_table(void* stream, const void* table, int n, int max_n) {
    EFFDET_ENTER();
    if (!table || n <= 0 || n > 65535 || max_n <= 0) return EFFDET_EINVAL;
    hipLaunchKernelGGL(grads_table_kernel, dim3((unsigned)max_n, (unsigned)n), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       static_cast<const GradOp*>(table), n);
    return effdet_check_launch();
}

// ================================================================================================================
// C ABI
// ================================================================================================================
static int launch_gemm_nt(hipStream_t st, GemmNtArgs& p) {
    const float* A = p.A; const float* W = p.W; const float* bias = p.bias; float* C = p.C; float* C2 = p.C2;
    const long long M = p.M; const int K = p.K, N = p.N;
    long long gx = (M + 127) / 128;                          // 4 waves x 32 rows
    if (gx > 0x7fffffffLL) return EFFDET_EINVAL;
    // deep K on few rows: too few workgroups with a long serial chain each -> K split over the workgroup's waves
    const bool splitk = K >= 384 && gx * ((N + 63) / 64) < 1024;
    if (splitk) gx = (M + 31) / 32;
    const bool vec = K % 4 == 0 && p.am.ld % 4 == 0 && p.am.img_stride % 4 == 0 &&
                     reinterpret_cast<uintptr_t>(A) % 16 == 0 && reinterpret_cast<uintptr_t>(W) % 16 == 0;
    p.vec_out = N % 4 == 0 && p.cm.ld % 4 == 0 && p.cm.img_stride % 4 == 0 && reinterpret_cast<uintptr_t>(C) % 16 == 0 &&
                (bias == nullptr || reinterpret_cast<uintptr_t>(bias) % 16 == 0) &&
                (C2 == nullptr || reinterpret_cast<uintptr_t>(C2) % 16 == 0) &&
                (p.R == nullptr || reinterpret_cast<uintptr_t>(p.R) % 16 == 0);
    if (!p.vec_out && N % 2 == 0 && p.cm.ld % 2 == 0 && p.cm.img_stride % 2 == 0 && reinterpret_cast<uintptr_t>(C) % 8 == 0 &&
        (bias == nullptr || reinterpret_cast<uintptr_t>(bias) % 8 == 0) && (C2 == nullptr || reinterpret_cast<uintptr_t>(C2) % 8 == 0) &&
        (p.R == nullptr || reinterpret_cast<uintptr_t>(p.R) % 8 == 0)) p.vec_out = 2;
    if (gx * ((N + 63) / 64) > 0x7fffffffLL) return EFFDET_EINVAL;
    const dim3 grid((unsigned)(gx * ((N + 63) / 64)));
    const bool vec2 = K % 2 == 0 && p.am.ld % 2 == 0 && p.am.img_stride % 2 == 0 &&
                      reinterpret_cast<uintptr_t>(A) % 8 == 0 && reinterpret_cast<uintptr_t>(W) % 8 == 0;
    const bool fast = vec && K >= 4 && p.am.nlev == 0 && p.am.img_stride == 0 && M < 0x7fffffffLL &&
                      (!p.a_scale || (reinterpret_cast<uintptr_t>(p.a_scale) % 16 == 0 && p.a_scale_rpi < 0x7fffffffLL));
    if (fast) {
        if (splitk) hipLaunchKernelGGL((gemm_nt_kernel<4, 4, true>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((gemm_nt_kernel<4, 1, true>), grid, dim3(256), 0, st, p);
        return effdet_check_launch();
    }
    if (splitk) {
        if (vec) hipLaunchKernelGGL((gemm_nt_kernel<4, 4>), grid, dim3(256), 0, st, p);
        else if (vec2) hipLaunchKernelGGL((gemm_nt_kernel<2, 4>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((gemm_nt_kernel<1, 4>), grid, dim3(256), 0, st, p);
    } else {
        if (vec) hipLaunchKernelGGL((gemm_nt_kernel<4, 1>), grid, dim3(256), 0, st, p);
        else if (vec2) hipLaunchKernelGGL((gemm_nt_kernel<2, 1>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((gemm_nt_kernel<1, 1>), grid, dim3(256), 0, st, p);
    }
    return effdet_check_launch();
}

extern "C" int effdet_train_gemm_nt(void* stream, const float* A, long long a_rpi, long long a_img_stride, long long a_ld,
                                    const float* W, const float* bias, float* C, long long c_rpi, long long c_img_stride,
                                    long long c_ld, long long M, int K, int N, int accumulate, float* C2) {
    EFFDET_ENTER();
    if (!A || !W || !C || M <= 0 || K <= 0 || N <= 0) return EFFDET_EINVAL;
    GemmNtArgs p;
    p.A = A; p.W = W; p.bias = bias; p.C = C; p.M = M; p.K = K; p.N = N; p.accumulate = accumulate; p.C2 = C2;
    p.R = nullptr; p.a_scale = nullptr; p.a_scale_rpi = 1;
    p.am = make_rowmap(a_rpi, a_img_stride, a_ld, M, K);
    p.cm = make_rowmap(c_rpi, c_img_stride, c_ld, M, N);
    return launch_gemm_nt(reinterpret_cast<hipStream_t>(stream), p);
}

// Dense rows with the fused forms of an MBConv block: C = (A * a_scale[row / a_scale_rows]) W^T + bias + R, C2 = silu(C)
// (a_scale [M / a_scale_rows][K]: the SE gate applied while A is loaded; R [M][N]: the shortcut; each optional)
extern "C" int effdet_train_gemm_nt_fused(void* stream, const float* A, const float* a_scale, long long a_scale_rows, const float* W,
                                          const float* bias, const float* R, float* C, float* C2, long long M, int K, int N) {
    EFFDET_ENTER();
    if (!A || !W || !C || M <= 0 || K <= 0 || N <= 0 || (a_scale && a_scale_rows <= 0)) return EFFDET_EINVAL;
    if (a_scale && (K % 4 || reinterpret_cast<uintptr_t>(a_scale) % 16)) return EFFDET_EINVAL;
    GemmNtArgs p;
    p.A = A; p.W = W; p.bias = bias; p.C = C; p.M = M; p.K = K; p.N = N; p.accumulate = 0; p.C2 = C2;
    p.R = R; p.a_scale = a_scale; p.a_scale_rpi = a_scale ? a_scale_rows : 1;
    p.am = make_rowmap(0, 0, 0, M, K);
    p.cm = make_rowmap(0, 0, 0, M, N);
    return launch_gemm_nt(reinterpret_cast<hipStream_t>(stream), p);
}

// The same GEMM over the rows of a level-major packed pyramid (train_levels.hip); the side flagged `*_packed` is the image-major
// head tensor [B][sum_l H_l W_l][pk_ld] (pk_img_stride floats per image), the other side is dense level-major rows.
extern "C" int effdet_train_gemm_nt_levels(void* stream, const float* A, int a_packed, const float* W, const float* bias, float* C,
                                           int c_packed, int B, int L, const int* Hs, const int* Ws, long long pk_img_stride,
                                           long long pk_ld, int K, int N, float* C2) {
    EFFDET_ENTER();
    if (!A || !W || !C || K <= 0 || N <= 0 || (a_packed && c_packed) || (C2 && c_packed)) return EFFDET_EINVAL;
    GemmNtArgs p;
    RowMap lm;
    const long long M = make_levels_rowmap(lm, B, L, Hs, Ws, pk_img_stride > 0 ? pk_img_stride : 1, pk_ld > 0 ? pk_ld : 1);
    if (M <= 0) return EFFDET_EINVAL;
    p.A = A; p.W = W; p.bias = bias; p.C = C; p.M = M; p.K = K; p.N = N; p.accumulate = 0; p.C2 = C2;
    p.R = nullptr; p.a_scale = nullptr; p.a_scale_rpi = 1;
    p.am = a_packed ? lm : make_rowmap(0, 0, 0, M, K);
    p.cm = c_packed ? lm : make_rowmap(0, 0, 0, M, N);
    if ((a_packed && pk_ld < K) || (c_packed && pk_ld < N)) return EFFDET_EINVAL;
    return launch_gemm_nt(reinterpret_cast<hipStream_t>(stream), p);
}

static int tn_slices(long long M, int N, int K) {
    const long long tiles = (long long)(N >= 256 ? (N + 127) / 128 : (N + 31) / 32) * ((K + 1 + 63) / 64);   // N >= 256: the 128-column form
    long long S = (1024 + tiles - 1) / tiles;                     // aim at >= 1024 workgroups
    const long long by_rows = (M + 255) / 256;                    // at least 256 rows per slice
    if (S > by_rows) S = by_rows;
    const long long cap = (16LL << 20) / ((long long)N * (K + 1)); // <= 16 M floats of partials
    if (S > cap) S = cap;
    if (S < 1) S = 1;
    if (S > 65535) S = 65535;
    return (int)S;
}

extern "C" long long effdet_train_gemm_tn_workspace_floats(long long M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return EFFDET_EINVAL;
    return (long long)tn_slices(M, N, K) * N * (K + 1);
}

static int launch_gemm_tn(hipStream_t st, GemmTnArgs& p, float* out, float* workspace, long long workspace_floats) {
    const long long M = p.M; const int N = p.N, K = p.K;
    const int S = tn_slices(M, N, K);
    if (workspace_floats < (long long)S * N * (K + 1)) return EFFDET_EINVAL;
    p.partial = workspace; p.S = S;
    long long rps = (M + S - 1) / S;
    rps = (rps + 31) / 32 * 32;
    p.rows_per_slice = rps;
    const dim3 grid((unsigned)((N + 31) / 32), (unsigned)((K + 1 + 63) / 64), (unsigned)S);
    const bool vy = p.ym.ld % 4 == 0 && p.ym.img_stride % 4 == 0 && reinterpret_cast<uintptr_t>(p.dY) % 16 == 0;
    const bool vx = p.xm.ld % 4 == 0 && p.xm.img_stride % 4 == 0 && reinterpret_cast<uintptr_t>(p.X) % 16 == 0;
    const bool vy2 = N % 2 == 0 && p.ym.ld % 2 == 0 && p.ym.img_stride % 2 == 0 && reinterpret_cast<uintptr_t>(p.dY) % 8 == 0;
    if (N >= 256 && vx && K % 4 == 0 && !p.x_scale && ((vy && N % 4 == 0) || vy2)) {
        const bool dense_w = p.ym.nlev == 0 && p.xm.nlev == 0 && p.ym.img_stride == 0 && p.xm.img_stride == 0;
        const dim3 gw((unsigned)((N + 127) / 128), (unsigned)((K + 1 + 63) / 64), (unsigned)S);
        if (vy && N % 4 == 0) {
            if (dense_w) hipLaunchKernelGGL((gemm_tn_kernel_w<true, 4>), gw, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((gemm_tn_kernel_w<false, 4>), gw, dim3(256), 0, st, p);
        } else {
            if (dense_w) hipLaunchKernelGGL((gemm_tn_kernel_w<true, 2>), gw, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((gemm_tn_kernel_w<false, 2>), gw, dim3(256), 0, st, p);
        }
        int rcw = effdet_check_launch();
        if (rcw) return rcw;
        ReduceSplitArgs rw{workspace, out, N, K, S};
        const long long rbw = ((long long)N * (K + 1) + 15) / 16;
        if (rbw > 0x7fffffffLL) return EFFDET_EINVAL;
        hipLaunchKernelGGL(reduce_split_kernel, dim3((unsigned)rbw), dim3(256), 0, st, rw);
        return effdet_check_launch();
    }
    const bool xs_ok = !p.x_scale || (reinterpret_cast<uintptr_t>(p.x_scale) % 16 == 0 && M < 0x7fffffffLL && p.x_scale_rpi < 0x7fffffffLL);
    const bool dense = p.ym.nlev == 0 && p.xm.nlev == 0 && p.ym.img_stride == 0 && p.xm.img_stride == 0;
    if (vy && vx && N % 4 == 0 && K % 4 == 0 && xs_ok) {
        if (dense && p.x_scale) hipLaunchKernelGGL((gemm_tn_kernel_v<true, true, 4>), grid, dim3(256), 0, st, p);
        else if (dense) hipLaunchKernelGGL((gemm_tn_kernel_v<true, false, 4>), grid, dim3(256), 0, st, p);
        else if (p.x_scale) hipLaunchKernelGGL((gemm_tn_kernel_v<false, true, 4>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((gemm_tn_kernel_v<false, false, 4>), grid, dim3(256), 0, st, p);
    }
    else if (vy2 && vx && K % 4 == 0 && !p.x_scale) {
        if (dense) hipLaunchKernelGGL((gemm_tn_kernel_v<true, false, 2>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((gemm_tn_kernel_v<false, false, 2>), grid, dim3(256), 0, st, p);
    }
    else if (vy && vx) hipLaunchKernelGGL((gemm_tn_kernel<4, true>), grid, dim3(256), 0, st, p);
    else if (vx) {
        if (vy2) hipLaunchKernelGGL((gemm_tn_kernel<2, true>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((gemm_tn_kernel<1, true>), grid, dim3(256), 0, st, p);
    }
    else if (vy) hipLaunchKernelGGL((gemm_tn_kernel<4, false>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((gemm_tn_kernel<1, false>), grid, dim3(256), 0, st, p);
    int rc = effdet_check_launch();
    if (rc) return rc;
    ReduceSplitArgs r{workspace, out, N, K, S};
    const long long rb = ((long long)N * (K + 1) + 15) / 16;
    if (rb > 0x7fffffffLL) return EFFDET_EINVAL;
    hipLaunchKernelGGL(reduce_split_kernel, dim3((unsigned)rb), dim3(256), 0, st, r);
    return effdet_check_launch();
}

extern "C" int effdet_train_gemm_tn(void* stream, const float* dY, long long y_rpi, long long y_img_stride, long long y_ld,
                                    const float* X, long long x_rpi, long long x_img_stride, long long x_ld,
                                    long long M, int N, int K, float* out, float* workspace, long long workspace_floats) {
    EFFDET_ENTER();
    if (!dY || !X || !out || !workspace || M <= 0 || N <= 0 || K <= 0) return EFFDET_EINVAL;
    GemmTnArgs p;
    p.dY = dY; p.X = X; p.M = M; p.N = N; p.K = K; p.x_scale = nullptr; p.x_scale_rpi = 1;
    p.ym = make_rowmap(y_rpi, y_img_stride, y_ld, M, N);
    p.xm = make_rowmap(x_rpi, x_img_stride, x_ld, M, K);
    return launch_gemm_tn(reinterpret_cast<hipStream_t>(stream), p, out, workspace, workspace_floats);
}

// dense rows, X scaled per image while it is loaded: out = dY^T [X * x_scale[row / x_scale_rows] | 1]
extern "C" int effdet_train_gemm_tn_scaled(void* stream, const float* dY, const float* X, const float* x_scale, long long x_scale_rows,
                                           long long M, int N, int K, float* out, float* workspace, long long workspace_floats) {
    EFFDET_ENTER();
    if (!dY || !X || !x_scale || x_scale_rows <= 0 || !out || !workspace || M <= 0 || N <= 0 || K <= 0) return EFFDET_EINVAL;
    if (K % 4 || reinterpret_cast<uintptr_t>(x_scale) % 16) return EFFDET_EINVAL;
    GemmTnArgs p;
    p.dY = dY; p.X = X; p.M = M; p.N = N; p.K = K; p.x_scale = x_scale; p.x_scale_rpi = x_scale_rows;
    p.ym = make_rowmap(0, 0, 0, M, N);
    p.xm = make_rowmap(0, 0, 0, M, K);
    return launch_gemm_tn(reinterpret_cast<hipStream_t>(stream), p, out, workspace, workspace_floats);
}

// weight gradient over the rows of a level-major packed pyramid: X dense level-major rows, dY the image-major head tensor
// (y_packed) or dense level-major rows.  Workspace: effdet_train_gemm_tn_workspace_floats(B * sum H W, N, K).
extern "C" int effdet_train_gemm_tn_levels(void* stream, const float* dY, int y_packed, const float* X, int B, int L, const int* Hs,
                                           const int* Ws, long long pk_img_stride, long long pk_ld, int N, int K, float* out,
                                           float* workspace, long long workspace_floats) {
    EFFDET_ENTER();
    if (!dY || !X || !out || !workspace || N <= 0 || K <= 0) return EFFDET_EINVAL;
    GemmTnArgs p;
    RowMap lm;
    const long long M = make_levels_rowmap(lm, B, L, Hs, Ws, pk_img_stride > 0 ? pk_img_stride : 1, pk_ld > 0 ? pk_ld : 1);
    if (M <= 0 || (y_packed && pk_ld < N)) return EFFDET_EINVAL;
    p.dY = dY; p.X = X; p.M = M; p.N = N; p.K = K; p.x_scale = nullptr; p.x_scale_rpi = 1;
    p.ym = y_packed ? lm : make_rowmap(0, 0, 0, M, N);
    p.xm = make_rowmap(0, 0, 0, M, K);
    return launch_gemm_tn(reinterpret_cast<hipStream_t>(stream), p, out, workspace, workspace_floats);
}

extern "C" int effdet_train_reduce_mid(void* stream, const float* in, int G, int S, long long L, float* out, int accumulate) {
    EFFDET_ENTER();
    if (!in || !out || G <= 0 || S <= 0 || L <= 0) return EFFDET_EINVAL;
    return launch_reduce_mid(reinterpret_cast<hipStream_t>(stream), in, G, S, L, out, accumulate);
}

// (k may carry EFFDET_PAD_SYMMETRIC: stripped here, so the callers' kernel dispatch sees the plain size)
static int dw_fill(DwBwdArgs& a, int B, int H, int W, int C, int& k, int stride) {
    const int sym = take_pad_flag(k);
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4 || (k != 3 && k != 5) || (stride != 1 && stride != 2)) return EFFDET_EINVAL;
    a.B = B; a.H = H; a.W = W; a.C = C; a.k = k; a.stride = stride;
    a.Ho = same_out(H, stride); a.Wo = same_out(W, stride);
    a.pad_t = pad_before(H, k, stride, sym); a.pad_l = pad_before(W, k, stride, sym);
    return 0;
}

static int dwf_fill(DwFwdArgs& a, int B, int H, int W, int C, int& k, int stride) {
    const int sym = take_pad_flag(k);
    if (B <= 0 || B > 65535 || H <= 0 || W <= 0 || C <= 0 || C % 4 || (k != 3 && k != 5) || (stride != 1 && stride != 2)) return EFFDET_EINVAL;
    a.B = B; a.H = H; a.W = W; a.C = C;
    a.Ho = same_out(H, stride); a.Wo = same_out(W, stride);
    a.pad_t = pad_before(H, k, stride, sym); a.pad_l = pad_before(W, k, stride, sym);
    a.strips_x = (a.Wo + DWF_PX - 1) / DWF_PX;
    a.nstrips = a.strips_x * a.Ho;
    a.blocks_per_image = (a.nstrips + 16 * DWF_SPL - 1) / (16 * DWF_SPL);
    return 0;
}

// SE pool partial rows per image that effdet_train_dwconv_fwd writes for this geometry
extern "C" int effdet_train_dwconv_fwd_parts(int H, int W, int C, int k, int stride) {
    DwFwdArgs a;
    if (dwf_fill(a, 1, H, W, C, k, stride)) return EFFDET_EINVAL;
    return a.blocks_per_image;
}

// training forward of conv_dw + folded BN + SiLU: Z = pre-activation (kept for the backward), A = silu(Z) (optional), and the
// SE pool partial rows of A ([B][effdet_train_dwconv_fwd_parts][C], optional) from the same pass
extern "C" int effdet_train_dwconv_fwd(void* stream, const float* X, float* Z, float* A, const float* Wt, const float* scale,
                                       const float* shift, float* pool_partial, int B, int H, int W, int C, int k, int stride) {
    EFFDET_ENTER();
    DwFwdArgs a;
    if (!X || !Z || !Wt || !scale || !shift || (!A && pool_partial) || dwf_fill(a, B, H, W, C, k, stride)) return EFFDET_EINVAL;
    a.X = X; a.Z = Z; a.A = A; a.taps = Wt; a.scale = scale; a.shift = shift; a.pool_partial = pool_partial;
    const dim3 grid((unsigned)a.blocks_per_image, (unsigned)B, (unsigned)((C + 63) / 64));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (k == 3 && stride == 1) hipLaunchKernelGGL((dw_fwd_train_kernel<3, 1>), grid, dim3(256), 0, st, a);
    else if (k == 3) hipLaunchKernelGGL((dw_fwd_train_kernel<3, 2>), grid, dim3(256), 0, st, a);
    else if (stride == 1) hipLaunchKernelGGL((dw_fwd_train_kernel<5, 1>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((dw_fwd_train_kernel<5, 2>), grid, dim3(256), 0, st, a);
    return effdet_check_launch();
}

static int launch_dw_bwd_dx(void* stream, const float* dY, const float* taps, const float* Z, float* dX,
                            int B, int H, int W, int C, int k, int stride);
extern "C" int effdet_train_dwconv_bwd_dx(void* stream, const float* dY, const float* taps, float* dX,
                                          int B, int H, int W, int C, int k, int stride) {
    EFFDET_ENTER();
    return launch_dw_bwd_dx(stream, dY, taps, nullptr, dX, B, H, W, C, k, stride);
}
// the same, followed by the SiLU backward of the layer below in the same pass: dX = (conv^T dY) * silu'(Z), Z [B][H][W][C]
extern "C" int effdet_train_dwconv_bwd_dx_silu(void* stream, const float* dY, const float* taps, const float* Z, float* dX,
                                               int B, int H, int W, int C, int k, int stride) {
    EFFDET_ENTER();
    if (!Z) return EFFDET_EINVAL;
    return launch_dw_bwd_dx(stream, dY, taps, Z, dX, B, H, W, C, k, stride);
}
static int launch_dw_bwd_dx(void* stream, const float* dY, const float* taps, const float* Z, float* dX,
                            int B, int H, int W, int C, int k, int stride) {
    DwBwdArgs a;
    if (!dY || !taps || !dX || dw_fill(a, B, H, W, C, k, stride)) return EFFDET_EINVAL;
    a.dY = dY; a.taps = taps; a.dX = dX; a.X = nullptr; a.partial = nullptr; a.segs_per_chunk = 0; a.seg = 0; a.Z = Z;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (stride == 1) {
        const long long total = (long long)B * H * ((W + 3) / 4) * (C / 4);
        const long long blocks = (total + 255) / 256;
        if (blocks > 0x7fffffffLL) return EFFDET_EINVAL;
        if (k == 3) hipLaunchKernelGGL(dw_bwd_dx_s1_kernel<3>, dim3((unsigned)blocks), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(dw_bwd_dx_s1_kernel<5>, dim3((unsigned)blocks), dim3(256), 0, st, a);
        return effdet_check_launch();
    }
    const long long total = (long long)B * H * W * (C / 4);
    const long long blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffLL) return EFFDET_EINVAL;
    hipLaunchKernelGGL(dw_bwd_dx_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
    return effdet_check_launch();
}

static long long dw_chunks(const DwBwdArgs& a, long long* spc, int* seg) {
    *seg = a.Wo <= 48 ? a.Wo : 32;
    const long long nseg = (long long)a.B * a.Ho * ((a.Wo + *seg - 1) / *seg);
    const int cgroups = (a.C + 63) / 64;
    long long chunks = (1024 + cgroups - 1) / cgroups;            // aim at >= 1024 workgroups
    long long per = (nseg + chunks - 1) / chunks;
    if (per < 4) per = 4;
    per = (per + 3) / 4 * 4;
    *spc = per;
    return (nseg + per - 1) / per;
}

extern "C" long long effdet_train_dwconv_bwd_dw_workspace_floats(int B, int H, int W, int C, int k, int stride) {
    DwBwdArgs a;
    if (dw_fill(a, B, H, W, C, k, stride)) return EFFDET_EINVAL;
    long long per;
    int seg;
    return dw_chunks(a, &per, &seg) * (k * k + 1) * C;
}

extern "C" int effdet_train_dwconv_bwd_dw(void* stream, const float* dY, const float* X, float* out,
                                          int B, int H, int W, int C, int k, int stride, float* workspace, long long workspace_floats,
                                          int cmajor) {
    EFFDET_ENTER();
    DwBwdArgs a;
    if (!dY || !X || !out || !workspace || dw_fill(a, B, H, W, C, k, stride)) return EFFDET_EINVAL;
    long long per;
    int seg;
    const long long chunks = dw_chunks(a, &per, &seg);
    if (workspace_floats < chunks * (k * k + 1) * C || chunks > 0x7fffffffLL) return EFFDET_EINVAL;
    a.dY = dY; a.X = X; a.taps = nullptr; a.dX = nullptr; a.partial = workspace; a.segs_per_chunk = per; a.seg = seg; a.Z = nullptr;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)chunks, (unsigned)((C + 63) / 64));
    if (k == 3 && stride == 1) hipLaunchKernelGGL((dw_bwd_dw_kernel<3, 1>), grid, dim3(256), 0, st, a);
    else if (k == 3) hipLaunchKernelGGL((dw_bwd_dw_kernel<3, 2>), grid, dim3(256), 0, st, a);
    else if (stride == 1) hipLaunchKernelGGL((dw_bwd_dw_kernel<5, 1>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((dw_bwd_dw_kernel<5, 2>), grid, dim3(256), 0, st, a);
    int rc = effdet_check_launch();
    if (rc) return rc;
    return effdet_launch_reduce_mid_tr(st, workspace, 1, (int)chunks, (long long)(k * k + 1) * C, out, 0, 1.0f, cmajor ? C : 0, k * k);
}

extern "C" int effdet_train_ew(void* stream, int op, float* out, const float* a, const float* b, const float* c,
                               const float* v0, const float* v1, const float* v2, const float* v3,
                               float s0, float s1, float s2, float s3, long long n, int C, long long hw, const float* sdev, float* out2) {
    EFFDET_ENTER();
    if (!out || !a || n <= 0 || n % 4 || C <= 0 || C % 4 || op < 0 || op > 13) return EFFDET_EINVAL;
    const bool need_b = op == 1 || op == 2 || op == 6 || op == 7 || op == 9 || op == 10 || op == 11 || op == 13;
    if (need_b && !b) return EFFDET_EINVAL;
    if ((op == 11 || op == 12) && !c) return EFFDET_EINVAL;
    if ((op == 3 || op == 4 || op == 5 || op == 6 || op == 12 || op == 13) && !v0) return EFFDET_EINVAL;
    if ((op == 5 || op == 6 || op == 12) && !v1) return EFFDET_EINVAL;
    if (op == 6 && (!v2 || !v3)) return EFFDET_EINVAL;
    if ((op == 4 || op == 5 || op == 12 || op == 13) && hw <= 0) return EFFDET_EINVAL;
    if (op == 7 && s3 == 0.f && !sdev) return EFFDET_EINVAL;
    EwArgs p{op, out, a, b, c, v0, v1, v2, v3, s0, s1, s2, s3, n, C, (hw > 0 ? hw : 1) * C, sdev, out2};
    const long long blocks = (n / 4 + 255) / 256;
    if (blocks > 0x7fffffffLL) return EFFDET_EINVAL;
    hipLaunchKernelGGL(ew_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
    return effdet_check_launch();
}

static int col_slices(int G, long long R, int C, long long* rps) {
    const long long cg = (long long)((C + 63) / 64) * G;
    long long S = (1024 + cg - 1) / cg;
    const long long by_rows = (R + 63) / 64;
    if (S > by_rows) S = by_rows;
    if (S < 1) S = 1;
    long long per = (R + S - 1) / S;
    per = (per + 3) / 4 * 4;
    *rps = per;
    return (int)((R + per - 1) / per);
}

extern "C" long long effdet_train_col_reduce_workspace_floats(int G, long long R, int C) {
    if (G <= 0 || R <= 0 || C <= 0) return EFFDET_EINVAL;
    long long rps;
    return (long long)col_slices(G, R, C, &rps) * G * C * 2;          // mode 4 keeps two sums per slice
}

extern "C" int effdet_train_col_reduce(void* stream, int mode, const float* a, const float* b, const float* v,
                                       int G, long long R, int C, float* out, float* workspace, long long workspace_floats, float alpha) {
    EFFDET_ENTER();
    if (!a || !out || !workspace || G <= 0 || G > 65535 || R <= 0 || C <= 0 || mode < 0 || mode > 4) return EFFDET_EINVAL;
    if ((mode == 1 || mode >= 3) && !b) return EFFDET_EINVAL;
    if (mode >= 2 && !v) return EFFDET_EINVAL;
    long long rps;
    const int S = col_slices(G, R, C, &rps);
    const int Wd = mode == 4 ? 2 : 1;
    if (workspace_floats < (long long)S * G * C * Wd) return EFFDET_EINVAL;
    ColArgs p{mode, a, b, v, workspace, R, rps, C, S};
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const bool vec = C % 4 == 0 && reinterpret_cast<uintptr_t>(a) % 16 == 0 && (!b || reinterpret_cast<uintptr_t>(b) % 16 == 0);
    const dim3 grid((unsigned)S, (unsigned)((C + 63) / 64), (unsigned)G);
    if (vec) hipLaunchKernelGGL(col_reduce_kernel<4>, grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL(col_reduce_kernel<1>, grid, dim3(256), 0, st, p);
    int rc = effdet_check_launch();
    if (rc) return rc;
    return launch_reduce_mid(st, workspace, G, S, (long long)C * Wd, out, 0, alpha);
}

// BatchNorm (batch statistics) forward, second half: var = mean((a - mean)^2) over the R rows and the layer's bookkeeping
// (effdet_train_col_reduce mode 2 + effdet_train_bn_finalize in two launches instead of three); mean [C] = the batch mean
extern "C" int effdet_train_bn_var_finalize(void* stream, const float* a, const float* mean, long long R, int C, const float* gamma,
                                            const float* beta, float* running_mean, float* running_var, long long* num_batches_tracked,
                                            float momentum, float unbias, float eps, float* scale, float* shift, float* rstd,
                                            float* workspace, long long workspace_floats) {
    EFFDET_ENTER();
    if (!a || !mean || !gamma || !beta || !running_mean || !running_var || !scale || !shift || !rstd || !workspace || R <= 0 || C <= 0)
        return EFFDET_EINVAL;
    long long rps;
    const int S = col_slices(1, R, C, &rps);
    if (workspace_floats < (long long)S * C) return EFFDET_EINVAL;
    ColArgs p{2, a, nullptr, mean, workspace, R, rps, C, S};
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const bool vec = C % 4 == 0 && reinterpret_cast<uintptr_t>(a) % 16 == 0;
    const dim3 grid((unsigned)S, (unsigned)((C + 63) / 64), 1);
    if (vec) hipLaunchKernelGGL(col_reduce_kernel<4>, grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL(col_reduce_kernel<1>, grid, dim3(256), 0, st, p);
    int rc = effdet_check_launch();
    if (rc) return rc;
    ReduceBnArgs q{};
    q.in = workspace; q.S = S; q.C = C; q.kind = 0; q.alpha = 1.0f / (float)R;
    q.mean = mean; q.gamma = gamma; q.beta = beta; q.running_mean = running_mean; q.running_var = running_var; q.nbt = num_batches_tracked;
    q.momentum = momentum; q.unbias = unbias; q.eps = eps; q.scale = scale; q.shift = shift; q.rstd_out = rstd;
    hipLaunchKernelGGL(reduce_bn_kernel, dim3((unsigned)((C + 15) / 16)), dim3(256), 0, st, q);
    return effdet_check_launch();
}

// BatchNorm backward, first half: sums of dy and dy (c - mean) over the R rows, then d gamma, d beta, v1, v3 (out [4][C]):
// effdet_train_col_reduce mode 4 + effdet_train_bn_bwd_prep in two launches instead of three
extern "C" int effdet_train_bn_bwd_sums(void* stream, const float* dy, const float* c, const float* mean, const float* rstd,
                                        long long R, int C, float* out, float* workspace, long long workspace_floats) {
    EFFDET_ENTER();
    if (!dy || !c || !mean || !rstd || !out || !workspace || R <= 0 || C <= 0) return EFFDET_EINVAL;
    long long rps;
    const int S = col_slices(1, R, C, &rps);
    if (workspace_floats < (long long)S * C * 2) return EFFDET_EINVAL;
    ColArgs p{4, dy, c, mean, workspace, R, rps, C, S};
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const bool vec = C % 4 == 0 && reinterpret_cast<uintptr_t>(dy) % 16 == 0 && reinterpret_cast<uintptr_t>(c) % 16 == 0;
    const dim3 grid((unsigned)S, (unsigned)((C + 63) / 64), 1);
    if (vec) hipLaunchKernelGGL(col_reduce_kernel<4>, grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL(col_reduce_kernel<1>, grid, dim3(256), 0, st, p);
    int rc = effdet_check_launch();
    if (rc) return rc;
    ReduceBnArgs q{};
    q.in = workspace; q.S = S; q.C = C; q.kind = 1; q.rstd_in = rstd; q.invM = 1.0f / (float)R;
    q.dgamma = out; q.dbeta = out + C; q.v1 = out + 2LL * C; q.v3 = out + 3LL * C;
    hipLaunchKernelGGL(reduce_bn_kernel, dim3((unsigned)((C + 15) / 16)), dim3(256), 0, st, q);
    return effdet_check_launch();
}

extern "C" int effdet_train_spatial(void* stream, int op, const float* in, const float* aux, float* out,
                                    int B, int H, int W, int C) {
    EFFDET_ENTER();
    const int sym = take_pad_flag(op);
    if (!in || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4 || op < 0 || op > 2 || (op == 2 && !aux)) return EFFDET_EINVAL;
    SpArgs p{op, in, aux, out, B, H, W, C, same_out(H, 2), same_out(W, 2), pad_before(H, 3, 2, sym), pad_before(W, 3, 2, sym)};
    const long long total = (long long)B * H * W * (C / 4) * (op == 0 ? 4 : 1);
    const long long blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffLL) return EFFDET_EINVAL;
    hipLaunchKernelGGL(spatial_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
    return effdet_check_launch();
}

extern "C" int effdet_train_im2col_stem(void* stream, const float* X, float* col, int B, int H, int W) {
    EFFDET_ENTER();
    const int sym = take_pad_flag(B);
    if (!X || !col || B <= 0 || H <= 0 || W <= 0) return EFFDET_EINVAL;
    Im2colArgs p{X, col, B, H, W, same_out(H, 2), same_out(W, 2), pad_before(H, 3, 2, sym), pad_before(W, 3, 2, sym)};
    const long long total = (long long)B * p.Ho * p.Wo * 32;
    const long long blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffLL) return EFFDET_EINVAL;
    hipLaunchKernelGGL(im2col_stem_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
    return effdet_check_launch();
}

extern "C" int effdet_train_se_bwd(void* stream, const float* pool_sum, int hw, const float* gate, const float* dgate,
                                   const float* W1, const float* b1, const float* W2t, float* ds, float* pgrad,
                                   int B, int C, int R) {
    EFFDET_ENTER();
    if (!pool_sum || !gate || !dgate || !W1 || !b1 || !W2t || !ds || !pgrad || hw <= 0 || B <= 0 || C <= 0 || R <= 0) return EFFDET_EINVAL;
    const size_t sh = (size_t)(2 * C + 2 * R) * sizeof(float);
    if (sh > 64 * 1024) return EFFDET_EINVAL;
    SeBwdArgs p{pool_sum, 1.0f / (float)hw, gate, dgate, W1, b1, W2t, ds, pgrad, C, R};
    hipLaunchKernelGGL(se_bwd_kernel, dim3(B), dim3(1024), sh, reinterpret_cast<hipStream_t>(stream), p);
    return effdet_check_launch();
}

extern "C" int effdet_train_fold_bn(void* stream, const float* W, int N, int K, const float* gamma, const float* beta,
                                    const float* mean, const float* var, float eps,
                                    float* Wf, float* WfT, float* WT, float* scale, float* shift, float* rstd) {
    EFFDET_ENTER();
    if (!W || !gamma || !beta || !mean || !var || !scale || !shift || !rstd || N <= 0 || K <= 0) return EFFDET_EINVAL;
    FoldArgs p{W, N, K, gamma, beta, mean, var, eps, Wf, WfT, WT, scale, shift, rstd};
    hipLaunchKernelGGL(fold_bn_kernel, dim3((unsigned)N), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
    return effdet_check_launch();
}

extern "C" int effdet_train_convbn_grads(void* stream, const float* dWext, int N, int K, int transposed, const float* W,
                                         const float* scale, const float* rstd, const float* mean,
                                         float* dW, float* dgamma, float* dbeta) {
    EFFDET_ENTER();
    if (!dWext || !W || !scale || !rstd || !mean || !dW || !dgamma || !dbeta || N <= 0 || K <= 0) return EFFDET_EINVAL;
    FinArgs p{dWext, N, K, transposed, W, scale, rstd, mean, dW, dgamma, dbeta};
    hipLaunchKernelGGL(convbn_grads_kernel, dim3((unsigned)N), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
    return effdet_check_launch();
}

extern "C" int effdet_train_bn_finalize(void* stream, const float* mean, const float* var, const float* gamma, const float* beta,
                                        float* running_mean, float* running_var, long long* num_batches_tracked, int C, int train,
                                        float momentum, float unbias, float eps, float* scale, float* shift, float* rstd) {
    EFFDET_ENTER();
    if (!mean || !var || !gamma || !beta || !scale || !shift || !rstd || C <= 0) return EFFDET_EINVAL;
    if (train && (!running_mean || !running_var)) return EFFDET_EINVAL;
    BnFinArgs p{mean, var, gamma, beta, running_mean, running_var, num_batches_tracked, C, train, momentum, unbias, eps, scale, shift, rstd};
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
    return effdet_check_launch();
}

extern "C" int effdet_train_bn_bwd_prep(void* stream, const float* s1, const float* s2c, const float* rstd, int C, float inv_m,
                                        float* dgamma, float* dbeta, float* v1, float* v3) {
    EFFDET_ENTER();
    if (!s1 || !s2c || !rstd || !dgamma || !dbeta || !v1 || !v3 || C <= 0) return EFFDET_EINVAL;
    BnBwdArgs p{s1, s2c, rstd, C, inv_m, dgamma, dbeta, v1, v3};
    hipLaunchKernelGGL(bn_bwd_prep_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
    return effdet_check_launch();
}
