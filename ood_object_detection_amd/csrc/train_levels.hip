// Training-path kernels that act on the WHOLE feature pyramid in one launch (the head towers of
// effdet/efficientdet.py:438-452 share their conv weights over the levels, only the BatchNorm layers are per level).
//
// Layout: the "packed pyramid" [sum_l B*H_l*W_l][C] float32, level-major - rows of level l are the NHWC tensor [B][H_l][W_l][C]
// of that level, the levels one behind the other.  A 1x1 conv over it is one GEMM over all rows (train_net.hip); the kernels
// here are the operators that need the geometry or the level of a row:
//   effdet_train_levels_dw            depthwise 3x3 / s1 TF-SAME over every level (forward, or d input with flipped taps)
//   effdet_train_levels_dw_bwd_dw     d taps summed over all levels (the taps are shared)
//   effdet_train_levels_col_reduce    per-level per-channel sums (BatchNorm batch statistics forward / backward)
//   effdet_train_levels_bn_finalize   nn.BatchNorm2d bookkeeping of the L per-level layers in one launch
//   effdet_train_levels_bn_bwd_prep   d gamma / d beta and the vectors of the BN backward, per level
//   effdet_train_levels_ew            per-level per-channel affine (+ SiLU) and the BN (batch statistics) backward
// Reductions are two-stage with a fixed order (partial rows, then effdet_launch_reduce_mid): bitwise reproducible.  Measured and
// dropped: finishing the reduction inside the first launch (the workgroup that arrives last at an agent-scope counter adds the
// partial rows) - with the 150 - 770 partial rows these shapes need, that one workgroup's serial pass cost 120 - 560 us.
#include "common.h"

namespace {

constexpr int MAXL = 8;
struct Levels {
    int n, B;
    int H[MAXL], W[MAXL];
    long long row0[MAXL + 1];               // rows of level l: [row0[l], row0[l + 1])
};

inline int fill_levels(Levels& lv, int B, int L, const int* Hs, const int* Ws) {
    if (B <= 0 || L <= 0 || L > MAXL || !Hs || !Ws) return EFFDET_EINVAL;
    lv.n = L; lv.B = B;
    lv.row0[0] = 0;
    for (int l = 0; l < MAXL; ++l) {
        if (l < L) {
            if (Hs[l] <= 0 || Ws[l] <= 0) return EFFDET_EINVAL;
            lv.H[l] = Hs[l]; lv.W[l] = Ws[l];
            lv.row0[l + 1] = lv.row0[l] + (long long)B * Hs[l] * Ws[l];
        } else { lv.H[l] = 1; lv.W[l] = 1; lv.row0[l + 1] = lv.row0[l]; }
    }
    return 0;
}

DEV float lv_silu_grad(float z) { const float s = sigmoid_train(z); return s * (1.0f + z * (1.0f - s)); }

DEV int level_of(const Levels& lv, long long row) {
    int l = 0;
#pragma unroll
    for (int i = 1; i < MAXL; ++i) l = (i < lv.n && row >= lv.row0[i]) ? i : l;
    return l;
}

// ------------------------------------------------------------------------------------------------------------
// depthwise 3x3 / stride 1 / TF-SAME (pad 1) over the packed pyramid; flip: taps mirrored = d input of the same conv
// ------------------------------------------------------------------------------------------------------------
struct LvDwArgs { const float* X; const float* taps; float* Y; Levels lv; long long strip0[MAXL + 1]; int C, flip; };

// A thread owns 4 consecutive pixels of one row of one level (and 4 channels): per tap row it loads the 6 input values those
// pixels share and the 3 taps once (the one-pixel form read 9 + 9 vectors per output: 1.8 TB/s on the 17 MB pyramid).
__global__ __launch_bounds__(256) void lv_dw_kernel(LvDwArgs p) {
    constexpr int PX = 4;
    const int C4 = p.C / 4;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= p.strip0[p.lv.n] * C4) return;
    const int c = (int)(i % C4) * 4;
    const long long strip = i / C4;
    int l = 0;
#pragma unroll
    for (int q = 1; q < MAXL; ++q) l = (q < p.lv.n && strip >= p.strip0[q]) ? q : l;
    const int H = p.lv.H[l], W = p.lv.W[l];
    const int sx = (W + PX - 1) / PX;
    const long long ls = strip - p.strip0[l];
    const int x0 = (int)(ls % sx) * PX;
    const long long t = ls / sx;
    const int y = (int)(t % H);
    const long long b = t / H;
    const long long img0 = p.lv.row0[l] + b * H * W;          // row of pixel (0, 0) of this image and level
    f32x4 acc[PX];
#pragma unroll
    for (int u = 0; u < PX; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int yy = y + ky - 1;
        if (yy < 0 || yy >= H) continue;
        const float* xrow = p.X + (img0 + (long long)yy * W) * p.C + c;
        f32x4 x[PX + 2], w[3];
#pragma unroll
        for (int j = 0; j < PX + 2; ++j) {
            const int xx = x0 - 1 + j;
            x[j] = (xx >= 0 && xx < W) ? *reinterpret_cast<const f32x4*>(xrow + (long long)xx * p.C) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int tt = p.flip ? 8 - (ky * 3 + kx) : ky * 3 + kx;
            w[kx] = *reinterpret_cast<const f32x4*>(p.taps + (long long)tt * p.C + c);
        }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int u = 0; u < PX; ++u) acc[u] += x[u + kx] * w[kx];
    }
#pragma unroll
    for (int u = 0; u < PX; ++u)
        if (x0 + u < W) *reinterpret_cast<f32x4*>(p.Y + (img0 + (long long)y * W + x0 + u) * p.C + c) = acc[u];
}

// d taps[t][c] = sum over every row of every level of dY[row][c] * X[row + tap t][c].  Workgroup = 64 channels (16 threads x 4
// channels, 16-byte loads) x 16 row lanes over a chunk of rows; partial [chunk][9][C].
struct LvDwWArgs { const float* dY; const float* X; float* partial; Levels lv; int C; long long rows_per_chunk; int chunks; };

__global__ __launch_bounds__(256) void lv_dw_bwd_dw_kernel(LvDwWArgs p) {
    __shared__ float sm[16][9][64];
    const int ct = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.y * 64 + ct * 4;
    const bool cv = c < p.C;
    f32x4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const long long rb = (long long)blockIdx.x * p.rows_per_chunk;
    long long re = rb + p.rows_per_chunk;
    const long long R = p.lv.row0[p.lv.n];
    if (re > R) re = R;
    if (cv) {
        for (long long row = rb + rl; row < re; row += 16) {
            const int l = level_of(p.lv, row);
            const int H = p.lv.H[l], W = p.lv.W[l];
            const long long local = row - p.lv.row0[l];
            const int x = (int)(local % W);
            const int y = (int)((local / W) % H);
            const f32x4 d = *reinterpret_cast<const f32x4*>(p.dY + row * p.C + c);
            const float* xb = p.X + (row - ((long long)y * W + x)) * p.C + c;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int yy = y + ky - 1;
                const bool yv = yy >= 0 && yy < H;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int xx = x + kx - 1;
                    if (yv && xx >= 0 && xx < W)
                        acc[ky * 3 + kx] += d * *reinterpret_cast<const f32x4*>(xb + ((long long)yy * W + xx) * p.C);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) sm[rl][t][ct * 4 + j] = acc[t][j];
    __syncthreads();
    for (int e = threadIdx.x; e < 9 * 64; e += 256) {
        const int t = e >> 6, cl = e & 63;
        if (blockIdx.y * 64 + cl >= p.C) continue;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += sm[k][t][cl];
        p.partial[((long long)blockIdx.x * 9 + t) * p.C + blockIdx.y * 64 + cl] = s;
    }
}

// ------------------------------------------------------------------------------------------------------------
// per-level per-channel sums.  modes: 0 sum a; 2 sum (a - v[l][c] * vscale[l])^2; 4 {sum a, sum a * (b - v[l][c])}
// pre (mode 4): a is first multiplied by silu'(pre) - the SiLU backward of the layer above, not stored
// ------------------------------------------------------------------------------------------------------------
struct LvColArgs {
    int mode; const float* a; const float* b; const float* v; const float* pre; float* partial;
    Levels lv; float vscale[MAXL]; int C, S;
};

// workgroup = 64 channels (16 threads x 4 channels, 16-byte loads) x 16 row lanes; the row lanes' sums are added in lane order
__global__ __launch_bounds__(256) void lv_col_reduce_kernel(LvColArgs p) {
    __shared__ float sm[16][64];
    __shared__ float sm2[16][64];
    const int ct = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.y * 64 + ct * 4;
    const int l = blockIdx.z, s = blockIdx.x;
    const bool cv = c < p.C;
    const long long R = p.lv.row0[l + 1] - p.lv.row0[l];
    long long per = (R + p.S - 1) / p.S;
    per = (per + 3) / 4 * 4;
    const long long rb = p.lv.row0[l] + (long long)s * per;
    long long re = rb + per;
    if (re > p.lv.row0[l + 1]) re = p.lv.row0[l + 1];
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f}, acc2 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (cv) {
        f32x4 vc = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.mode >= 2) vc = *reinterpret_cast<const f32x4*>(p.v + (long long)l * p.C + c) * p.vscale[l];
        for (long long r = rb + rl; r < re; r += 16) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(p.a + r * p.C + c);
            if (p.mode == 0) acc += a;
            else if (p.mode == 2) { const f32x4 d = a - vc; acc += d * d; }
            else {
                f32x4 ad = a;
                if (p.pre) {
                    const f32x4 z = *reinterpret_cast<const f32x4*>(p.pre + r * p.C + c);
#pragma unroll
                    for (int j = 0; j < 4; ++j) ad[j] = a[j] * lv_silu_grad(z[j]);
                }
                acc += ad;
                acc2 += ad * (*reinterpret_cast<const f32x4*>(p.b + r * p.C + c) - vc);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { sm[rl][ct * 4 + j] = acc[j]; sm2[rl][ct * 4 + j] = acc2[j]; }
    __syncthreads();
    const int Wd = p.mode == 4 ? 2 : 1;
    const int cl = threadIdx.x;
    if (cl < 64 && blockIdx.y * 64 + cl < p.C) {
        float* dst = p.partial + ((long long)l * p.S + s) * Wd * p.C + blockIdx.y * 64 + cl;
        float t = 0.f, t2 = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) { t += sm[k][cl]; t2 += sm2[k][cl]; }
        dst[0] = t;
        if (Wd == 2) dst[p.C] = t2;
    }
}

// ------------------------------------------------------------------------------------------------------------
// BatchNorm bookkeeping of the L per-level layers (separate parameter tensors) in one launch
// ------------------------------------------------------------------------------------------------------------
struct LvBnFinArgs {
    const float* sum; const float* sq;                       // [L][C] raw sums (batch statistics; unused for layers in eval mode)
    const float* gamma[MAXL]; const float* beta[MAXL]; float* rmean[MAXL]; float* rvar[MAXL]; long long* nbt[MAXL];
    int train[MAXL]; float invM[MAXL], unbias[MAXL], momentum[MAXL], eps[MAXL];
    int L, C;
    float* mean; float* scale; float* shift; float* rstd;    // [L][C]
};

__global__ __launch_bounds__(256) void lv_bn_finalize_kernel(LvBnFinArgs p) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int l = blockIdx.y;
    if (c == 0 && p.train[l] && p.nbt[l]) *p.nbt[l] += 1;
    if (c >= p.C) return;
    const long long o = (long long)l * p.C + c;
    float m, v;
    if (p.train[l]) {
        m = p.sum[o] * p.invM[l];
        v = p.sq[o] * p.invM[l];
        p.rmean[l][c] = p.rmean[l][c] * (1.0f - p.momentum[l]) + p.momentum[l] * m;
        p.rvar[l][c] = p.rvar[l][c] * (1.0f - p.momentum[l]) + p.momentum[l] * (v * p.unbias[l]);
    } else { m = p.rmean[l][c]; v = p.rvar[l][c]; }
    const float rs = 1.0f / sqrtf(v + p.eps[l]);
    const float sc = p.gamma[l][c] * rs;
    p.mean[o] = m;
    p.rstd[o] = rs;
    p.scale[o] = sc;
    p.shift[o] = p.beta[l][c] - m * sc;
}

struct LvBnBwdArgs { const float* sums; const float* rstd; float invM[MAXL]; int L, C; float* dgamma; float* dbeta; float* v1; float* v3; };
// sums [L][2][C] = {sum dy, sum dy (c - mean)} -> d gamma, d beta, v1 = sum(dy)/M, v3 = rstd^2 sum(dy (c - mean))/M   (all [L][C])
__global__ __launch_bounds__(256) void lv_bn_bwd_prep_kernel(LvBnBwdArgs p) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int l = blockIdx.y;
    if (c >= p.C) return;
    const long long o = (long long)l * p.C + c;
    const float s1 = p.sums[((long long)l * 2) * p.C + c], s2 = p.sums[((long long)l * 2 + 1) * p.C + c];
    const float rs = p.rstd[o];
    p.dgamma[o] = s2 * rs;
    p.dbeta[o] = s1;
    p.v1[o] = s1 * p.invM[l];
    p.v3[o] = rs * rs * s2 * p.invM[l];
}

// ------------------------------------------------------------------------------------------------------------
// element-wise with per-level per-channel vectors [L][C]
//   op 3: out = a * v0 + v1 (v1 optional), out2 = silu(out) optional
//   op 6: BN (batch statistics) backward  out = v0 * (a - v1 - (b - v2) * v3);  per level `train` = 0: out = a * v0
//   pre (op 6 only): a is first multiplied by silu'(z) (z = pre): the SiLU backward of the layer above in the same pass
// ------------------------------------------------------------------------------------------------------------
struct LvEwArgs {
    int op; float* out; float* out2; const float* a; const float* b; const float* pre;
    const float* v0; const float* v1; const float* v2; const float* v3; Levels lv; int train[MAXL]; int C;
};
__global__ __launch_bounds__(256) void lv_ew_kernel(LvEwArgs p) {
    const int C4 = p.C / 4;
    const long long i4 = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i4 >= p.lv.row0[p.lv.n] * C4) return;
    const long long row = i4 / C4;
    const int c = (int)(i4 - row * C4) * 4;
    const int l = level_of(p.lv, row);
    const long long i = i4 * 4, o = (long long)l * p.C + c;
    f32x4 a = *reinterpret_cast<const f32x4*>(p.a + i);
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(p.v0 + o);
    f32x4 r;
    if (p.op == 3) {
        r = a * v0;
        if (p.v1) r += *reinterpret_cast<const f32x4*>(p.v1 + o);
    } else {
        if (p.pre) {
            const f32x4 z = *reinterpret_cast<const f32x4*>(p.pre + i);
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] *= lv_silu_grad(z[j]);
        }
        if (p.train[l]) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(p.b + i);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(p.v1 + o), v2 = *reinterpret_cast<const f32x4*>(p.v2 + o);
            const f32x4 v3 = *reinterpret_cast<const f32x4*>(p.v3 + o);
            r = v0 * (a - v1 - (b - v2) * v3);
        } else r = a * v0;
    }
    *reinterpret_cast<f32x4*>(p.out + i) = r;
    if (p.out2) {
        f32x4 q;
#pragma unroll
        for (int j = 0; j < 4; ++j) q[j] = silu_train(r[j]);
        *reinterpret_cast<f32x4*>(p.out2 + i) = q;
    }
}

inline int lv_col_slices(const Levels& lv, int C) {
    const long long cg = (long long)((C + 63) / 64) * lv.n;
    long long S = (768 + cg - 1) / cg;
    long long rmax = 0;
    for (int l = 0; l < lv.n; ++l) { const long long r = lv.row0[l + 1] - lv.row0[l]; if (r > rmax) rmax = r; }
    const long long by_rows = (rmax + 63) / 64;
    if (S > by_rows) S = by_rows;
    if (S < 1) S = 1;
    return (int)S;
}
inline long long lv_dw_chunks(const Levels& lv, int C, long long* per) {
    const long long R = lv.row0[lv.n];
    const int cg = (C + 63) / 64;
    long long chunks = (768 + cg - 1) / cg;
    long long p = (R + chunks - 1) / chunks;
    if (p < 64) p = 64;
    p = (p + 3) / 4 * 4;
    *per = p;
    return (R + p - 1) / p;
}

}  // namespace

extern "C" int effdet_train_levels_dw(void* stream, const float* X, const float* taps, float* Y, int B, int L, const int* Hs,
                                      const int* Ws, int C, int flip) {
    EFFDET_ENTER();
    LvDwArgs p;
    if (!X || !taps || !Y || C <= 0 || C % 4 || fill_levels(p.lv, B, L, Hs, Ws)) return EFFDET_EINVAL;
    p.X = X; p.taps = taps; p.Y = Y; p.C = C; p.flip = flip ? 1 : 0;
    p.strip0[0] = 0;
    for (int l = 0; l < MAXL; ++l) p.strip0[l + 1] = p.strip0[l] + (l < L ? (long long)B * Hs[l] * ((Ws[l] + 3) / 4) : 0);
    const long long blocks = (p.strip0[L] * (C / 4) + 255) / 256;
    if (blocks > 0x7fffffffLL) return EFFDET_EINVAL;
    hipLaunchKernelGGL(lv_dw_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
    return effdet_check_launch();
}

extern "C" long long effdet_train_levels_workspace_floats(int B, int L, const int* Hs, const int* Ws, int C) {
    Levels lv;
    if (C <= 0 || fill_levels(lv, B, L, Hs, Ws)) return EFFDET_EINVAL;
    long long per;
    const long long a = lv_dw_chunks(lv, C, &per) * 9 * C;
    const long long b = (long long)lv_col_slices(lv, C) * L * 2 * C;
    return a > b ? a : b;
}

extern "C" int effdet_train_levels_dw_bwd_dw(void* stream, const float* dY, const float* X, float* out, int B, int L,
                                             const int* Hs, const int* Ws, int C, float* workspace, long long workspace_floats,
                                             int cmajor) {
    EFFDET_ENTER();
    LvDwWArgs p;
    if (!dY || !X || !out || !workspace || C <= 0 || C % 4 || fill_levels(p.lv, B, L, Hs, Ws)) return EFFDET_EINVAL;
    long long per;
    const long long chunks = lv_dw_chunks(p.lv, C, &per);
    if (workspace_floats < chunks * 9 * C || chunks > 65535) return EFFDET_EINVAL;
    p.dY = dY; p.X = X; p.partial = workspace; p.C = C; p.rows_per_chunk = per; p.chunks = (int)chunks;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(lv_dw_bwd_dw_kernel, dim3((unsigned)chunks, (unsigned)((C + 63) / 64)), dim3(256), 0, st, p);
    const int rc = effdet_check_launch();
    if (rc) return rc;
    return effdet_launch_reduce_mid_tr(st, workspace, 1, (int)chunks, 9LL * C, out, 0, 1.0f, cmajor ? C : 0, 9);
}

extern "C" int effdet_train_levels_col_reduce(void* stream, int mode, const float* a, const float* b, const float* v,
                                              const float* pre, const float* vscale, int B, int L, const int* Hs, const int* Ws, int C, float* out,
                                              float* workspace, long long workspace_floats) {
    EFFDET_ENTER();
    LvColArgs p;
    if (!a || !out || !workspace || C <= 0 || C % 4 || (mode != 0 && mode != 2 && mode != 4) || fill_levels(p.lv, B, L, Hs, Ws))
        return EFFDET_EINVAL;
    if ((mode == 4 && !b) || (mode >= 2 && !v)) return EFFDET_EINVAL;
    const int S = lv_col_slices(p.lv, C);
    const int cg = (C + 63) / 64;
    if (workspace_floats < (long long)S * L * 2 * C) return EFFDET_EINVAL;
    p.mode = mode; p.a = a; p.b = b; p.v = v; p.pre = mode == 4 ? pre : nullptr; p.partial = workspace; p.C = C; p.S = S;
    for (int l = 0; l < MAXL; ++l) p.vscale[l] = (vscale && l < L) ? vscale[l] : 1.0f;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(lv_col_reduce_kernel, dim3((unsigned)S, (unsigned)cg, (unsigned)L), dim3(256), 0, st, p);
    const int rc = effdet_check_launch();
    if (rc) return rc;
    return effdet_launch_reduce_mid(st, workspace, L, S, (long long)C * (mode == 4 ? 2 : 1), out, 0, 1.0f);
}

extern "C" int effdet_train_levels_bn_finalize(void* stream, const float* sum, const float* sq, int L, int C, const void* const* gamma,
                                               const void* const* beta, void* const* running_mean, void* const* running_var,
                                               void* const* num_batches_tracked, const int* train, const float* inv_m,
                                               const float* unbias, const float* momentum, const float* eps,
                                               float* mean, float* scale, float* shift, float* rstd) {
    EFFDET_ENTER();
    if (L <= 0 || L > MAXL || C <= 0 || !gamma || !beta || !running_mean || !running_var || !train || !inv_m || !unbias || !momentum ||
        !eps || !mean || !scale || !shift || !rstd) return EFFDET_EINVAL;
    LvBnFinArgs p;
    p.sum = sum; p.sq = sq; p.L = L; p.C = C; p.mean = mean; p.scale = scale; p.shift = shift; p.rstd = rstd;
    for (int l = 0; l < MAXL; ++l) {
        const int k = l < L ? l : 0;
        if (!gamma[k] || !beta[k] || !running_mean[k] || !running_var[k]) return EFFDET_EINVAL;
        if (train[k] && (!sum || !sq)) return EFFDET_EINVAL;
        p.gamma[l] = static_cast<const float*>(gamma[k]); p.beta[l] = static_cast<const float*>(beta[k]);
        p.rmean[l] = static_cast<float*>(running_mean[k]); p.rvar[l] = static_cast<float*>(running_var[k]);
        p.nbt[l] = num_batches_tracked ? static_cast<long long*>(num_batches_tracked[k]) : nullptr;
        p.train[l] = train[k]; p.invM[l] = inv_m[k]; p.unbias[l] = unbias[k]; p.momentum[l] = momentum[k]; p.eps[l] = eps[k];
    }
    hipLaunchKernelGGL(lv_bn_finalize_kernel, dim3((unsigned)((C + 255) / 256), (unsigned)L), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), p);
    return effdet_check_launch();
}

extern "C" int effdet_train_levels_bn_bwd_prep(void* stream, const float* sums, const float* rstd, const float* inv_m, int L, int C,
                                               float* dgamma, float* dbeta, float* v1, float* v3) {
    EFFDET_ENTER();
    if (!sums || !rstd || !inv_m || L <= 0 || L > MAXL || C <= 0 || !dgamma || !dbeta || !v1 || !v3) return EFFDET_EINVAL;
    LvBnBwdArgs p;
    p.sums = sums; p.rstd = rstd; p.L = L; p.C = C; p.dgamma = dgamma; p.dbeta = dbeta; p.v1 = v1; p.v3 = v3;
    for (int l = 0; l < MAXL; ++l) p.invM[l] = inv_m[l < L ? l : 0];
    hipLaunchKernelGGL(lv_bn_bwd_prep_kernel, dim3((unsigned)((C + 255) / 256), (unsigned)L), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), p);
    return effdet_check_launch();
}

extern "C" int effdet_train_levels_ew(void* stream, int op, float* out, float* out2, const float* a, const float* b, const float* pre,
                                      const float* v0, const float* v1, const float* v2, const float* v3, const int* train,
                                      int B, int L, const int* Hs, const int* Ws, int C) {
    EFFDET_ENTER();
    LvEwArgs p;
    if (!out || !a || !v0 || C <= 0 || C % 4 || (op != 3 && op != 6) || fill_levels(p.lv, B, L, Hs, Ws)) return EFFDET_EINVAL;
    bool any_train = false;
    for (int l = 0; l < MAXL; ++l) { p.train[l] = (op == 6 && train) ? train[l < L ? l : 0] : 0; any_train = any_train || (l < L && p.train[l]); }
    if (op == 6 && any_train && (!b || !v1 || !v2 || !v3)) return EFFDET_EINVAL;
    if (op == 3 && pre) return EFFDET_EINVAL;
    p.op = op; p.out = out; p.out2 = out2; p.a = a; p.b = b; p.pre = pre; p.v0 = v0; p.v1 = v1; p.v2 = v2; p.v3 = v3; p.C = C;
    const long long blocks = (p.lv.row0[L] * (C / 4) + 255) / 256;
    if (blocks > 0x7fffffffLL) return EFFDET_EINVAL;
    hipLaunchKernelGGL(lv_ew_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
    return effdet_check_launch();
}
