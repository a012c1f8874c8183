// Training-side operators of the hot path, compiled with -ffp-contract=off.
//
//   effdet_detection_loss   loss_fn of the fork (effdet/loss.py:224-298): one-hot targets (-1 / -2 -> no hot),
//                           alpha-weighted BCE-with-logits (`new_focal_loss`, :49-95 - gamma is NOT applied, the
//                           modulating factor is commented out in the reference), label smoothing, `!= -2` mask,
//                           Huber box loss on targets != 0 (:104-118, :171-179), normaliser sum(num_positives)+1,
//                           total = cls + w_box * box; plus the gradients w.r.t. the class / box head outputs.
//   effdet_label_anchors    AnchorLabeler.batch_label_anchors (effdet/anchors.py:384-438) =
//                           IouSimilarity (region_similarity_calculator.py:24-73) -> ArgMaxMatcher with thresholds
//                           0.5/0.5 and force_match_for_each_row (argmax_matcher.py:116-146) -> class targets - 1,
//                           FasterRcnnBoxCoder.encode (box_coder.py:81-110, eps 1e-8), num_positives.
//
// Head outputs use the packed layout of the inference path: cls [B, N, C], box [B, N, 4]; targets are the
// per-level reference tensors flattened and concatenated in the same anchor order: cls_t [B, N] int64,
// box_t [B, N, 4] fp32.
#include "common.h"

namespace {

constexpr int LT = 256;

DEV float block_sum(float v, float* sm) {
    v = wave_reduce_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[wave] = v;
    __syncthreads();
    return sm[0] + sm[1] + sm[2] + sm[3];
}

struct LossArgs {
    const void* cls; const void* box; int dtype;
    const long long* cls_t; const float* box_t; const float* num_pos;
    int B; long long N; int C;
    float alpha, delta, box_w, ls;
    void* gcls; void* gbox;
    float* partial;            // [2][nblocks]
    long long ncls_blocks, nbox_blocks;
};

DEV float ld_f(const void* p, int dtype, long long i) {
    return dtype == 0 ? reinterpret_cast<const float*>(p)[i] : (float)reinterpret_cast<const bf16_t*>(p)[i];
}
DEV void st_f(void* p, int dtype, long long i, float v) {
    if (dtype == 0) reinterpret_cast<float*>(p)[i] = v; else reinterpret_cast<bf16_t*>(p)[i] = (bf16_t)v;
}

constexpr int LE = 16;                  // elements per thread: a block reduces LE * LT elements to one partial

__global__ __launch_bounds__(LT) void loss_kernel(LossArgs p) {
    __shared__ float sm[4];
    __shared__ float s_norm;
    if (threadIdx.x == 0) {
        float n = 1.0f;
        for (int b = 0; b < p.B; ++b) n += p.num_pos[b];          // sum(num_positives) + 1, same order everywhere
        s_norm = n;
    }
    __syncthreads();
    const float norm = s_norm;
    const long long blk = blockIdx.x;
    float acc = 0.f;
    if (blk < p.ncls_blocks) {
        const long long total = (long long)p.B * p.N * p.C;
        const float inv_norm = 1.0f / norm;
#pragma unroll 4
        for (int e = 0; e < LE; ++e) {
            const long long i = (blk * LE + e) * LT + threadIdx.x;
            if (i >= total) break;
            const long long bn = i / p.C;
            const int c = (int)(i - bn * p.C);
            const long long t = p.cls_t[bn];
            const float x = ld_f(p.cls, p.dtype, i);
            const float hot = (t == c) ? 1.0f : 0.0f;
            const float alpha_f = hot * p.alpha + (1.0f - hot) * (1.0f - p.alpha);
            const float ts = p.ls > 0.f ? hot * (1.0f - p.ls) + 0.5f * p.ls : hot;
            // softplus(-|x|) on the hardware exp2 / log2 (the step evaluates 7 M of these per image): e = exp(-|x|) in (0, 1], so
            // 1 + e is exact to half an ulp of 1 and log(1 + e) carries an absolute error <= 6e-8 - against terms of size
            // max(x, 0) - x t + ... that are summed into a loss of O(1e2); the gradient below does not use it
            const float en = __builtin_amdgcn_exp2f(fabsf(x) * -1.4426950408889634f);
            const float ce = fmaxf(x, 0.f) - x * ts + __builtin_amdgcn_logf(1.0f + en) * 0.6931471805599453f;
            const float mask = (t != -2) ? 1.0f : 0.0f;
            acc += inv_norm * alpha_f * ce * mask;
            if (p.gcls) {
                const float sg = x >= 0.f ? __builtin_amdgcn_rcpf(1.0f + en) : en * __builtin_amdgcn_rcpf(1.0f + en);   // sigmoid(x) from e = exp(-|x|)
                st_f(p.gcls, p.dtype, i, inv_norm * alpha_f * (sg - ts) * mask);
            }
        }
        const float s = block_sum(acc, sm);
        if (threadIdx.x == 0) p.partial[blk] = s;
    } else {
        const long long bb = blk - p.ncls_blocks;
        const long long total = (long long)p.B * p.N * 4;
        const float bnorm = norm * 4.0f;
#pragma unroll 4
        for (int e = 0; e < LE; ++e) {
            const long long i = (bb * LE + e) * LT + threadIdx.x;
            if (i >= total) break;
            const float t = p.box_t[i];
            const float x = ld_f(p.box, p.dtype, i);
            const float w = (t != 0.0f) ? 1.0f : 0.0f;
            const float err = x - t;
            const float a = fabsf(err);
            const float q = fminf(a, p.delta);
            const float lin = a - q;
            acc += (0.5f * q * q + p.delta * lin) * w;
            if (p.gbox) {
                const float g = (a <= p.delta) ? err : (err > 0.f ? p.delta : -p.delta);
                st_f(p.gbox, p.dtype, i, p.box_w * g * w / bnorm);
            }
        }
        const float s = block_sum(acc, sm);
        if (threadIdx.x == 0) p.partial[p.ncls_blocks + bb] = s;
    }
}

// fixed-order final sum of the per-block partials: 1024 threads stride over them, then a tree through LDS
__global__ __launch_bounds__(1024) void loss_finish_kernel(LossArgs p, float* out3) {
    __shared__ float smc[1024], smb[1024];
    float norm = 1.0f;
    for (int b = 0; b < p.B; ++b) norm += p.num_pos[b];
    float c = 0.f, bx = 0.f;
    for (long long i = threadIdx.x; i < p.ncls_blocks; i += 1024) c += p.partial[i];
    for (long long i = threadIdx.x; i < p.nbox_blocks; i += 1024) bx += p.partial[p.ncls_blocks + i];
    smc[threadIdx.x] = c; smb[threadIdx.x] = bx;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { smc[threadIdx.x] += smc[threadIdx.x + o]; smb[threadIdx.x] += smb[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float cs = smc[0], bs = smb[0];
        const float box_loss = bs / (norm * 4.0f);
        out3[1] = cs; out3[2] = box_loss; out3[0] = cs + p.box_w * box_loss;
    }
}

// ---------------------------------------------------------------------------------------------- labeler
constexpr int MAXGT = 512;

struct LabelArgs {
    const float* anchors; const float* gt_boxes; const long long* gt_cls;
    int B, Mmax; long long N; float thr;
    long long* cls_t; float* box_t; float* num_pos; long long* match_out;
    int* match0;                       // [B][N]
    int* force_row;                    // [B][N] (0x7F7F7F7F = not forced)
    unsigned long long* best;          // [B][Mmax][nblk] per-block best (iou, lowest anchor) of every gt row
    int nblk;
};

DEV float iou_yxyx(const float* g, float ga, float a0, float a1, float a2, float a3, float aa) {
    const float ih = fmaxf(fminf(g[2], a2) - fmaxf(g[0], a0), 0.f);
    const float iw = fmaxf(fminf(g[3], a3) - fmaxf(g[1], a1), 0.f);
    const float inter = ih * iw;
    const float uni = ga + aa - inter;
    return inter == 0.0f ? 0.0f : inter / uni;
}

// compacts the valid ground-truth rows of image b into LDS; returns their count
DEV int load_gt(const LabelArgs& p, int b, float (*gb)[4], float* garea, long long* glab) {
    __shared__ int cnt;
    if (threadIdx.x == 0) {
        int m = 0;
        for (int i = 0; i < p.Mmax && m < MAXGT; ++i) {
            const long long c = p.gt_cls[(long long)b * p.Mmax + i];
            if (c > -1) {
                const float* g = p.gt_boxes + ((long long)b * p.Mmax + i) * 4;
                gb[m][0] = g[0]; gb[m][1] = g[1]; gb[m][2] = g[2]; gb[m][3] = g[3];
                garea[m] = (g[2] - g[0]) * (g[3] - g[1]);
                glab[m] = c;
                ++m;
            }
        }
        cnt = m;
    }
    __syncthreads();
    return cnt;
}

__global__ __launch_bounds__(LT) void label_match_kernel(LabelArgs p) {
    __shared__ float gb[MAXGT][4];
    __shared__ float garea[MAXGT];
    __shared__ long long glab[MAXGT];
    __shared__ unsigned long long wbest[4];
    const int b = blockIdx.y;
    const int M = load_gt(p, b, gb, garea, glab);
    const long long n = (long long)blockIdx.x * LT + threadIdx.x;
    const bool ok = n < p.N;
    float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    if (ok) { const float* a = p.anchors + n * 4; a0 = a[0]; a1 = a[1]; a2 = a[2]; a3 = a[3]; }
    const float aa = (a2 - a0) * (a3 - a1);
    float bv = -1.0f; int bm = -1;
    for (int m = 0; m < M; ++m) {
        const float v = iou_yxyx(gb[m], garea[m], a0, a1, a2, a3, aa);
        if (v > bv) { bv = v; bm = m; }                      // first maximum, like torch.max on CPU
        // per-row (gt) best anchor of this block: max iou, ties -> lowest anchor index
        unsigned long long key = ok ? (((unsigned long long)__float_as_uint(v) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned int)n)) : 0ull;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const unsigned long long other = __shfl_xor(key, o, 64); key = other > key ? other : key; }
        __syncthreads();
        if ((threadIdx.x & 63) == 0) wbest[threadIdx.x >> 6] = key;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long k = wbest[0];
            for (int w = 1; w < 4; ++w) k = wbest[w] > k ? wbest[w] : k;
            p.best[((long long)b * p.Mmax + m) * p.nblk + blockIdx.x] = k;
        }
    }
    if (ok) p.match0[(long long)b * p.N + n] = (M == 0 || p.thr > bv) ? -1 : bm;
}

__global__ __launch_bounds__(LT) void label_force_kernel(LabelArgs p) {
    __shared__ float gb[MAXGT][4];
    __shared__ float garea[MAXGT];
    __shared__ long long glab[MAXGT];
    const int b = blockIdx.x;
    const int M = load_gt(p, b, gb, garea, glab);
    for (int m = threadIdx.x; m < M; m += LT) {
        unsigned long long k = 0ull;
        const unsigned long long* src = p.best + ((long long)b * p.Mmax + m) * p.nblk;
        for (int q = 0; q < p.nblk; ++q) k = src[q] > k ? src[q] : k;
        const unsigned int col = 0xFFFFFFFFu - (unsigned int)(k & 0xFFFFFFFFull);
        atomicMin(&p.force_row[(long long)b * p.N + col], m);          // several rows on one column: lowest row wins
    }
}

__global__ __launch_bounds__(LT) void label_targets_kernel(LabelArgs p) {
    __shared__ float gb[MAXGT][4];
    __shared__ float garea[MAXGT];
    __shared__ long long glab[MAXGT];
    __shared__ float sm[4];
    const int b = blockIdx.y;
    load_gt(p, b, gb, garea, glab);
    const long long n = (long long)blockIdx.x * LT + threadIdx.x;
    float pos = 0.f;
    if (n < p.N) {
        const long long o = (long long)b * p.N + n;
        const int f = p.force_row[o];
        const int m = (f != 0x7F7F7F7F) ? f : p.match0[o];           // 0x7F7F7F7F = memset pattern 'not forced'
        if (p.match_out) p.match_out[o] = m;
        long long cls = -1;
        float ty = 0.f, tx = 0.f, th = 0.f, tw = 0.f;
        if (m >= 0) {
            pos = 1.f;
            cls = glab[m] - 1;
            const float* a = p.anchors + n * 4;
            const float wa0 = a[3] - a[1], ha0 = a[2] - a[0];
            const float yca = a[0] + ha0 / 2.f, xca = a[1] + wa0 / 2.f;
            const float w0 = gb[m][3] - gb[m][1], h0 = gb[m][2] - gb[m][0];
            const float yc = gb[m][0] + h0 / 2.f, xc = gb[m][1] + w0 / 2.f;
            const float ha = ha0 + 1e-8f, wa = wa0 + 1e-8f, h = h0 + 1e-8f, w = w0 + 1e-8f;
            tx = (xc - xca) / wa; ty = (yc - yca) / ha;
            tw = logf(w / wa); th = logf(h / ha);
        }
        p.cls_t[o] = cls;
        p.box_t[o * 4 + 0] = ty; p.box_t[o * 4 + 1] = tx; p.box_t[o * 4 + 2] = th; p.box_t[o * 4 + 3] = tw;
    }
    const float s = block_sum(pos, sm);
    if (threadIdx.x == 0 && s != 0.f) atomicAdd(&p.num_pos[b], s);       // integer-valued floats: order independent
}

}  // namespace

extern "C" long long effdet_detection_loss_workspace_floats(int B, long long N, int C) {
    if (B <= 0 || N <= 0 || C <= 0) return EFFDET_EINVAL;
    const long long per = (long long)LT * LE;
    const long long ncls = ((long long)B * N * C + per - 1) / per, nbox = ((long long)B * N * 4 + per - 1) / per;
    return ncls + nbox;
}

extern "C" int effdet_detection_loss(void* stream, int dtype, const void* cls, const void* box,
                                     const long long* cls_t, const float* box_t, const float* num_positives,
                                     int B, long long N, int C, float alpha, float delta, float box_loss_weight,
                                     float label_smoothing, float* out3, void* grad_cls, void* grad_box,
                                     float* workspace, long long workspace_floats) {
    EFFDET_ENTER();
    if (!cls || !box || !cls_t || !box_t || !num_positives || !out3 || !workspace || (dtype & ~1)) return EFFDET_EINVAL;
    const long long need = effdet_detection_loss_workspace_floats(B, N, C);
    if (need <= 0 || workspace_floats < need) return EFFDET_EINVAL;
    LossArgs a{cls, box, dtype, cls_t, box_t, num_positives, B, N, C, alpha, delta, box_loss_weight, label_smoothing,
               grad_cls, grad_box, workspace, 0, 0};
    a.ncls_blocks = ((long long)B * N * C + (long long)LT * LE - 1) / ((long long)LT * LE);
    a.nbox_blocks = ((long long)B * N * 4 + (long long)LT * LE - 1) / ((long long)LT * LE);
    if (a.ncls_blocks + a.nbox_blocks > 0x7fffffffLL) return EFFDET_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(loss_kernel, dim3((unsigned)(a.ncls_blocks + a.nbox_blocks)), dim3(LT), 0, st, a);
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(1024), 0, st, a, out3);
    return effdet_check_launch();
}

extern "C" long long effdet_label_anchors_workspace_bytes(int B, int Mmax, long long N) {
    if (B <= 0 || Mmax < 0 || N <= 0) return EFFDET_EINVAL;
    const long long nblk = (N + LT - 1) / LT;
    return (long long)B * N * 8 + (long long)B * (Mmax > 0 ? Mmax : 1) * nblk * 8;
}

extern "C" int effdet_label_anchors(void* stream, const float* anchors, const float* gt_boxes, const long long* gt_cls,
                                    int B, int Mmax, long long N, float match_threshold,
                                    long long* cls_t, float* box_t, float* num_positives, long long* match,
                                    void* workspace, long long workspace_bytes) {
    EFFDET_ENTER();
    if (!anchors || !cls_t || !box_t || !num_positives || !workspace || B <= 0 || Mmax < 0 || Mmax > MAXGT || N <= 0) return EFFDET_EINVAL;
    if (Mmax > 0 && (!gt_boxes || !gt_cls)) return EFFDET_EINVAL;
    if (workspace_bytes < effdet_label_anchors_workspace_bytes(B, Mmax, N)) return EFFDET_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    LabelArgs a{anchors, gt_boxes, gt_cls, B, Mmax, N, match_threshold, cls_t, box_t, num_positives, match,
                nullptr, nullptr, nullptr, (int)((N + LT - 1) / LT)};
    char* ws = reinterpret_cast<char*>(workspace);
    a.match0 = reinterpret_cast<int*>(ws);
    a.force_row = reinterpret_cast<int*>(ws + (size_t)B * N * 4);
    a.best = reinterpret_cast<unsigned long long*>(ws + (size_t)B * N * 8);
    if (hipMemsetAsync(a.force_row, 0x7F, (size_t)B * N * 4, st) != hipSuccess) return EFFDET_ELAUNCH;    // 0x7F7F7F7F
    if (hipMemsetAsync(num_positives, 0, (size_t)B * 4, st) != hipSuccess) return EFFDET_ELAUNCH;
    hipLaunchKernelGGL(label_match_kernel, dim3(a.nblk, B), dim3(LT), 0, st, a);
    if (Mmax > 0) hipLaunchKernelGGL(label_force_kernel, dim3(B), dim3(LT), 0, st, a);
    hipLaunchKernelGGL(label_targets_kernel, dim3(a.nblk, B), dim3(LT), 0, st, a);
    return effdet_check_launch();
}

// ------------------------------------------------------------------------------------------------
// Optimizer half of the pretrain step (pretrain.py:272-276): torch.nn.utils.clip_grad_norm_(params, 10.)
// followed by torch.optim.Adam.step(), on flat float32 buffers (parameters, gradients and both moments of a
// parameter group concatenated).  Fixed-order two-stage reductions: bitwise reproducible.
// ------------------------------------------------------------------------------------------------
namespace {

constexpr int SQN_BLOCKS = 1024;

__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float* g, long long n, float* partial) {
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s += g[i] * g[i];
    s = wave_reduce_sum(s);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

__global__ __launch_bounds__(256) void sqnorm_final_kernel(const float* partial, int nb, float* out, int accumulate) {
    float s = 0.f;
    for (int i = threadIdx.x; i < nb; i += 256) s += partial[i];
    s = wave_reduce_sum(s);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = (part[0] + part[1]) + (part[2] + part[3]);
        out[0] = accumulate ? out[0] + t : t;
    }
}

struct AdamArgs {
    float* p; const float* g; float* m; float* v; long long n;
    float lr, beta1, beta2, eps, bc1, bc2_sqrt, max_norm;
    const float* sqnorm;                       // total squared gradient norm (all groups), or null: no clipping
    const float* bc_dev;                       // optional device {bc1, bc2_sqrt} overriding the host values
};

// clip coefficient exactly as clip_grad_norm_: clamp(max_norm / (total_norm + 1e-6), max = 1), applied to the gradient;
// then Adam in torch's order: m.lerp_(g, 1-b1); v = v*b2 + (1-b2) g^2; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ __launch_bounds__(256) void adam_clip_kernel(AdamArgs a) {
    float coef = 1.f;
    if (a.sqnorm != nullptr) {
        coef = a.max_norm / (sqrtf(a.sqnorm[0]) + 1e-6f);
        coef = coef > 1.f ? 1.f : coef;
    }
    if (a.bc_dev != nullptr) { a.bc1 = a.bc_dev[0]; a.bc2_sqrt = a.bc_dev[1]; }
    const float step_size = a.lr / a.bc1;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (long long)gridDim.x * 256) {
        const float g = a.g[i] * coef;
        float m = a.m[i], v = a.v[i];
        m = m + (1.f - a.beta1) * (g - m);
        v = v * a.beta2 + (1.f - a.beta2) * g * g;
        const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
        a.p[i] = a.p[i] - step_size * (m / denom);
        a.m[i] = m; a.v[i] = v;
    }
}

}  // namespace

extern "C" long long effdet_sqnorm_workspace_floats(void) { return SQN_BLOCKS; }

/* out[0] (+)= sum(g^2); accumulate != 0 adds to the value already there (several parameter groups). */
extern "C" int effdet_sqnorm(void* stream, const float* g, long long n, float* workspace, float* out, int accumulate) {
    EFFDET_ENTER();
    if (!g || !workspace || !out || n <= 0) return EFFDET_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    long long nb = (n + 255) / 256; if (nb > SQN_BLOCKS) nb = SQN_BLOCKS;
    hipLaunchKernelGGL(sqnorm_partial_kernel, dim3((unsigned)nb), dim3(256), 0, st, g, n, workspace);
    hipLaunchKernelGGL(sqnorm_final_kernel, dim3(1), dim3(256), 0, st, workspace, (int)nb, out, accumulate);
    return effdet_check_launch();
}

extern "C" int effdet_adam_clip_step(void* stream, float* p, const float* g, float* m, float* v, long long n,
                                     float lr, float beta1, float beta2, float eps, int step,
                                     float max_norm, const float* sqnorm) {
    EFFDET_ENTER();
    if (!p || !g || !m || !v || n <= 0 || step <= 0 || !(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f)) return EFFDET_EINVAL;
    AdamArgs a{p, g, m, v, n, lr, beta1, beta2, eps,
               (float)(1.0 - pow((double)beta1, step)), (float)sqrt(1.0 - pow((double)beta2, step)), max_norm, sqnorm, nullptr};
    long long nb = (n + 255) / 256; if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(adam_clip_kernel, dim3((unsigned)nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    return effdet_check_launch();
}

extern "C" int effdet_adam_clip_step_dev(void* stream, float* p, const float* g, float* m, float* v, long long n,
                                         float lr, float beta1, float beta2, float eps, const float* bc_dev,
                                         float max_norm, const float* sqnorm) {
    EFFDET_ENTER();
    if (!p || !g || !m || !v || !bc_dev || n <= 0 || !(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f)) return EFFDET_EINVAL;
    AdamArgs a{p, g, m, v, n, lr, beta1, beta2, eps, 1.f, 1.f, max_norm, sqnorm, bc_dev};
    long long nb = (n + 255) / 256; if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(adam_clip_kernel, dim3((unsigned)nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    return effdet_check_launch();
}
