// Training-path kernels of one BiFPN node's FpnCombine (effdet/efficientdet.py:180-245: every input is resampled to the node's
// resolution - identity, nearest x2 upsample, or 3x3 / s2 TF-SAME max-pool - then fused with normalised edge weights, then SiLU).
// float32, NHWC.  The resampled inputs are never materialised: each kernel reads the SOURCE tensors at their own resolution.
//   effdet_train_fpn_weights   edge_weights parameter -> {w0, w1, w2, den} on the device ('fastattn': relu, den = sum + 1e-4;
//                              'attn': softmax, den = 1; 'sum': ones, den = 1)
//   effdet_train_fpn_combine   fused = sum_i (R_i(x_i) * w_i) / den,  act = silu(fused)                       (forward)
//   effdet_train_fpn_dots      S[i][c] = sum_pixels dfused * R_i(x_i),  dfused = dact * silu'(fused)            (backward 1)
//   effdet_train_fpn_wgrad     S -> d edge_weights (closed form of the normalisation)                           (backward 2)
//   effdet_train_fpn_input_bwd d x_i = (w_i / den) * R_i^T(dfused) (+ an earlier gradient of the same tensor)   (backward 3)
#include "common.h"

namespace {

struct FpnIn { const float* p; int h, w, delta, pad_t, pad_l; };     // delta: 0 same size, +1 source is coarser (x2 upsample), -1 finer (max-pool)
struct FpnArgs {
    FpnIn in[3]; int n, method;                                     // method 0: 'fastattn' arithmetic (x * w) / den; 1: x * w
    const float* wdev;                                              // {w0, w1, w2, den}
    const float* dact; const float* fused; float* out; float* out2; float* partial;
    int B, H, W, C; long long rows_per_slice; int S;
};

DEV float fpn_silu_grad(float z) { const float s = sigmoid_train(z); return s * (1.0f + z * (1.0f - s)); }

// value of resampled input `in` at pixel (b, y, x), 4 channels from c
DEV f32x4 fpn_sample(const FpnIn& in, long long b, int y, int x, int c, int C) {
    if (in.delta == 0) return *reinterpret_cast<const f32x4*>(in.p + ((b * in.h + y) * in.w + x) * C + c);
    if (in.delta > 0) return *reinterpret_cast<const f32x4*>(in.p + ((b * in.h + (y >> 1)) * in.w + (x >> 1)) * C + c);
    f32x4 m = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int yy = y * 2 + ky - in.pad_t;
        if (yy < 0 || yy >= in.h) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int xx = x * 2 + kx - in.pad_l;
            if (xx < 0 || xx >= in.w) continue;
            const f32x4 q = *reinterpret_cast<const f32x4*>(in.p + ((b * in.h + yy) * in.w + xx) * C + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) m[j] = fmaxf(m[j], q[j]);
        }
    }
    return m;
}

__global__ __launch_bounds__(256) void fpn_combine_kernel(FpnArgs p) {
    const int C4 = p.C / 4;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)p.B * p.H * p.W * C4) return;
    const int c = (int)(i % C4) * 4;
    long long px = i / C4;
    const int x = (int)(px % p.W); px /= p.W;
    const int y = (int)(px % p.H);
    const long long b = px / p.H;
    const float den = p.wdev[3];
    f32x4 o;
    const f32x4 a0 = fpn_sample(p.in[0], b, y, x, c, p.C), a1 = fpn_sample(p.in[1], b, y, x, c, p.C);
    if (p.method == 0) {
        o = (a0 * p.wdev[0]) / den + (a1 * p.wdev[1]) / den;
        if (p.n > 2) o = o + (fpn_sample(p.in[2], b, y, x, c, p.C) * p.wdev[2]) / den;
    } else {
        o = a0 * p.wdev[0] + a1 * p.wdev[1];
        if (p.n > 2) o = o + fpn_sample(p.in[2], b, y, x, c, p.C) * p.wdev[2];
    }
    *reinterpret_cast<f32x4*>(p.out + i * 4) = o;
    f32x4 q;
#pragma unroll
    for (int j = 0; j < 4; ++j) q[j] = silu_train(o[j]);
    *reinterpret_cast<f32x4*>(p.out2 + i * 4) = q;
}

// partial[slice][i][c] = sum over the slice's pixels of dfused * R_i(x_i); workgroup = 64 channels (16 threads x 4 channels,
// 16-byte loads) x 16 pixel lanes, the pixel lanes' sums added in lane order
__global__ __launch_bounds__(256) void fpn_dots_kernel(FpnArgs p) {
    __shared__ float sm[16][3][64];
    const int ct = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.y * 64 + ct * 4;
    const bool cv = c < p.C;
    const long long R = (long long)p.B * p.H * p.W;
    const long long rb = (long long)blockIdx.x * p.rows_per_slice;
    long long re = rb + p.rows_per_slice;
    if (re > R) re = R;
    f32x4 acc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (cv) {
        for (long long r = rb + rl; r < re; r += 16) {
            const int x = (int)(r % p.W);
            const long long t = r / p.W;
            const int y = (int)(t % p.H);
            const long long b = t / p.H;
            f32x4 d = *reinterpret_cast<const f32x4*>(p.dact + r * p.C + c);
            const f32x4 z = *reinterpret_cast<const f32x4*>(p.fused + r * p.C + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) d[j] *= fpn_silu_grad(z[j]);
#pragma unroll
            for (int i = 0; i < 3; ++i)
                if (i < p.n) acc[i] += d * fpn_sample(p.in[i], b, y, x, c, p.C);
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) sm[rl][i][ct * 4 + j] = acc[i][j];
    __syncthreads();
    const int cl = threadIdx.x & 63, i = threadIdx.x >> 6;
    if (i < p.n && blockIdx.y * 64 + cl < p.C) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sm[k][i][cl];
        p.partial[((long long)blockIdx.x * p.n + i) * p.C + blockIdx.y * 64 + cl] = t;
    }
}

// one workgroup: S_i = sum_c dots[i][c] (fixed order), then the closed form of the normalisation's derivative
struct FpnWgArgs { const float* dots; const float* wdev; const float* ewp; float* grad; int n, C, method; };
__global__ __launch_bounds__(64) void fpn_wgrad_kernel(FpnWgArgs p) {
    __shared__ float S[3];
    const int lane = threadIdx.x;
    for (int i = 0; i < p.n; ++i) {
        float s = 0.f;
        for (int c = lane; c < p.C; c += 64) s += p.dots[(long long)i * p.C + c];
        // lane sums added in lane order: a fixed association
        float tot = 0.f;
        for (int l = 0; l < 64; ++l) tot += __shfl(s, l, 64);
        if (lane == 0) S[i] = tot;
    }
    __syncthreads();
    if (lane >= p.n) return;
    const float den = p.wdev[3];
    float sw = 0.f;
    for (int i = 0; i < p.n; ++i) sw += S[i] * p.wdev[i];
    if (p.method == 0) {                                   // w_i = relu(e_i), out = sum x_i w_i / (sum w + eps)
        const float dw = S[lane] / den - sw / (den * den);
        p.grad[lane] = p.ewp[lane] > 0.f ? dw : 0.f;
    } else {                                               // softmax
        p.grad[lane] = p.wdev[lane] * (S[lane] - sw);
    }
}

struct FpnWArgs { const float* ewp; float* wdev; int n, method; };
__global__ __launch_bounds__(64) void fpn_weights_kernel(FpnWArgs p) {
    if (threadIdx.x != 0) return;
    float w[3] = {0.f, 0.f, 0.f};
    float den = 1.0f;
    if (p.method == 0) {
        float s = 0.f;
        for (int i = 0; i < p.n; ++i) { w[i] = fmaxf(p.ewp[i], 0.f); s += w[i]; }
        den = s + 0.0001f;
    } else if (p.method == 1) {
        float m = p.ewp[0];
        for (int i = 1; i < p.n; ++i) m = fmaxf(m, p.ewp[i]);
        float s = 0.f;
        for (int i = 0; i < p.n; ++i) { w[i] = expf(p.ewp[i] - m); s += w[i]; }
        for (int i = 0; i < p.n; ++i) w[i] = w[i] / s;
    } else {
        for (int i = 0; i < p.n; ++i) w[i] = 1.0f;
    }
    p.wdev[0] = w[0]; p.wdev[1] = w[1]; p.wdev[2] = w[2]; p.wdev[3] = den;
}

// d x = coef * R^T(dfused) (+ acc), one thread per 4 channels of a SOURCE pixel
struct FpnBwdArgs {
    FpnIn in; int idx; const float* wdev; const float* dact; const float* fused; const float* acc; float* out; int B, H, W, C;
};
DEV f32x4 fpn_dfused(const FpnBwdArgs& p, long long b, int y, int x, int c) {
    const long long o = ((b * p.H + y) * p.W + x) * p.C + c;
    const f32x4 d = *reinterpret_cast<const f32x4*>(p.dact + o), z = *reinterpret_cast<const f32x4*>(p.fused + o);
    f32x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = d[j] * fpn_silu_grad(z[j]);
    return r;
}
__global__ __launch_bounds__(256) void fpn_input_bwd_kernel(FpnBwdArgs p) {
    const int C4 = p.C / 4;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)p.B * p.in.h * p.in.w * C4) return;
    const int c = (int)(i % C4) * 4;
    long long px = i / C4;
    const int x = (int)(px % p.in.w); px /= p.in.w;
    const int y = (int)(px % p.in.h);
    const long long b = px / p.in.h;
    const float coef = p.wdev[p.idx] / p.wdev[3];
    f32x4 g;
    if (p.in.delta == 0) g = fpn_dfused(p, b, y, x, c);
    else if (p.in.delta > 0) {
        g = (fpn_dfused(p, b, 2 * y, 2 * x, c) + fpn_dfused(p, b, 2 * y, 2 * x + 1, c)) +
            (fpn_dfused(p, b, 2 * y + 1, 2 * x, c) + fpn_dfused(p, b, 2 * y + 1, 2 * x + 1, c));
    } else {
        // 3x3 / s2 max-pool backward: a pixel receives dfused of every window whose first maximum (row-major scan, strict >:
        // torch.max_pool2d's choice) it is
        const f32x4 me = *reinterpret_cast<const f32x4*>(p.in.p + i * 4);
        g = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int oy = (y + p.in.pad_t - 2 + 1) / 2; oy <= (y + p.in.pad_t) / 2; ++oy) {
            if (oy < 0 || oy >= p.H) continue;
            for (int ox = (x + p.in.pad_l - 2 + 1) / 2; ox <= (x + p.in.pad_l) / 2; ++ox) {
                if (ox < 0 || ox >= p.W) continue;
                bool first[4] = {true, true, true, true};
                for (int ky = 0; ky < 3; ++ky) {
                    const int yy = oy * 2 + ky - p.in.pad_t;
                    if (yy < 0 || yy >= p.in.h) continue;
                    for (int kx = 0; kx < 3; ++kx) {
                        const int xx = ox * 2 + kx - p.in.pad_l;
                        if (xx < 0 || xx >= p.in.w || (yy == y && xx == x)) continue;
                        const f32x4 q = *reinterpret_cast<const f32x4*>(p.in.p + ((b * p.in.h + yy) * p.in.w + xx) * p.C + c);
                        const bool before = yy < y || (yy == y && xx < x);
#pragma unroll
                        for (int j = 0; j < 4; ++j) first[j] = first[j] && (before ? q[j] < me[j] : q[j] <= me[j]);
                    }
                }
                const f32x4 d = fpn_dfused(p, b, oy, ox, c);
#pragma unroll
                for (int j = 0; j < 4; ++j) g[j] += first[j] ? d[j] : 0.f;
            }
        }
    }
    g = g * coef;
    if (p.acc) g = *reinterpret_cast<const f32x4*>(p.acc + i * 4) + g;
    *reinterpret_cast<f32x4*>(p.out + i * 4) = g;
}

inline int fpn_fill_in(FpnIn& in, const float* p, int h, int w, int H, int W, int sym) {
    in.p = p; in.h = h; in.w = w; in.pad_t = 0; in.pad_l = 0;
    if (!p || h <= 0 || w <= 0) return EFFDET_EINVAL;
    if (h == H && w == W) in.delta = 0;
    else if (2 * h == H && 2 * w == W) in.delta = 1;
    else if (same_out(h, 2) == H && same_out(w, 2) == W) {
        in.delta = -1; in.pad_t = pad_before(h, 3, 2, sym); in.pad_l = pad_before(w, 3, 2, sym);
    } else return EFFDET_EINVAL;
    return 0;
}

inline int fpn_slices(long long R, int C, long long* rps) {
    const long long cg = (C + 63) / 64;
    long long S = (512 + cg - 1) / cg;
    const long long by_rows = (R + 63) / 64;
    if (S > by_rows) S = by_rows;
    if (S < 1) S = 1;
    long long per = (R + S - 1) / S;
    per = (per + 3) / 4 * 4;
    *rps = per;
    return (int)((R + per - 1) / per);
}

}  // namespace

extern "C" int effdet_train_fpn_weights(void* stream, const float* edge_weights, int n, int method, float* wdev) {
    EFFDET_ENTER();
    if (!wdev || n < 2 || n > 3 || method < 0 || method > 2 || (method < 2 && !edge_weights)) return EFFDET_EINVAL;
    FpnWArgs p{edge_weights, wdev, n, method};
    hipLaunchKernelGGL(fpn_weights_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), p);
    return effdet_check_launch();
}

// srcs / hs / ws: the n source tensors [B][hs[i]][ws[i]][C] (same size as the node, half of it, or twice it)
extern "C" int effdet_train_fpn_combine(void* stream, int n, const void* const* srcs, const int* hs, const int* ws, int method,
                                        const float* wdev, float* fused, float* act, int B, int H, int W, int C) {
    EFFDET_ENTER();
    const int sym = take_pad_flag(method);
    if (n < 2 || n > 3 || !srcs || !hs || !ws || !wdev || !fused || !act || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4 ||
        method < 0 || method > 2) return EFFDET_EINVAL;
    FpnArgs p{};
    for (int i = 0; i < 3; ++i)
        if (fpn_fill_in(p.in[i], static_cast<const float*>(srcs[i < n ? i : 0]), hs[i < n ? i : 0], ws[i < n ? i : 0], H, W, sym)) return EFFDET_EINVAL;
    p.n = n; p.method = method == 0 ? 0 : 1; p.wdev = wdev; p.out = fused; p.out2 = act; p.B = B; p.H = H; p.W = W; p.C = C;
    const long long blocks = ((long long)B * H * W * (C / 4) + 255) / 256;
    if (blocks > 0x7fffffffLL) return EFFDET_EINVAL;
    hipLaunchKernelGGL(fpn_combine_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
    return effdet_check_launch();
}

extern "C" long long effdet_train_fpn_dots_workspace_floats(int B, int H, int W, int C) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return EFFDET_EINVAL;
    long long rps;
    return (long long)fpn_slices((long long)B * H * W, C, &rps) * 3 * C;
}

// dots [n][C]; grad [n] = d edge_weights (method 2 'sum': no parameter, grad may be NULL and only the dots are produced)
extern "C" int effdet_train_fpn_wgrad(void* stream, int n, const void* const* srcs, const int* hs, const int* ws, int method,
                                      const float* wdev, const float* edge_weights, const float* dact, const float* fused,
                                      float* dots, float* grad, int B, int H, int W, int C, float* workspace,
                                      long long workspace_floats) {
    EFFDET_ENTER();
    const int sym = take_pad_flag(method);
    if (n < 2 || n > 3 || !srcs || !hs || !ws || !wdev || !dact || !fused || !dots || !workspace || B <= 0 || H <= 0 || W <= 0 ||
        C <= 0 || C % 4 || method < 0 || method > 2 || (method < 2 && (!grad || !edge_weights))) return EFFDET_EINVAL;
    FpnArgs p{};
    for (int i = 0; i < 3; ++i)
        if (fpn_fill_in(p.in[i], static_cast<const float*>(srcs[i < n ? i : 0]), hs[i < n ? i : 0], ws[i < n ? i : 0], H, W, sym)) return EFFDET_EINVAL;
    long long rps;
    const int S = fpn_slices((long long)B * H * W, C, &rps);
    if (workspace_floats < (long long)S * n * C) return EFFDET_EINVAL;
    p.n = n; p.method = method; p.wdev = wdev; p.dact = dact; p.fused = fused; p.partial = workspace; p.B = B; p.H = H; p.W = W; p.C = C;
    p.rows_per_slice = rps; p.S = S;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(fpn_dots_kernel, dim3((unsigned)S, (unsigned)((C + 63) / 64)), dim3(256), 0, st, p);
    int rc = effdet_check_launch();
    if (rc) return rc;
    rc = effdet_launch_reduce_mid(st, workspace, 1, S, (long long)n * C, dots, 0, 1.0f);
    if (rc || method == 2) return rc;
    FpnWgArgs q{dots, wdev, edge_weights, grad, n, C, method};
    hipLaunchKernelGGL(fpn_wgrad_kernel, dim3(1), dim3(64), 0, st, q);
    return effdet_check_launch();
}

// gradient of source tensor `idx` ([B][h][w][C]); acc (optional, same shape): an earlier gradient of that tensor, added in
extern "C" int effdet_train_fpn_input_bwd(void* stream, int idx, const float* src, int h, int w, const float* wdev, const float* dact,
                                          const float* fused, const float* acc, float* out, int B, int H, int W, int C) {
    EFFDET_ENTER();
    const int sym = take_pad_flag(idx);
    FpnBwdArgs p{};
    if (idx < 0 || idx > 2 || !wdev || !dact || !fused || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4 ||
        fpn_fill_in(p.in, src, h, w, H, W, sym)) return EFFDET_EINVAL;
    p.idx = idx; p.wdev = wdev; p.dact = dact; p.fused = fused; p.acc = acc; p.out = out; p.B = B; p.H = H; p.W = W; p.C = C;
    const long long blocks = ((long long)B * h * w * (C / 4) + 255) / 256;
    if (blocks > 0x7fffffffLL) return EFFDET_EINVAL;
    hipLaunchKernelGGL(fpn_input_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
    return effdet_check_launch();
}
