// Detection evaluation on the device: the PASCAL-style mAP / CorLoc the fork computes inside its training loop
// (pretrain.py:246-252, ObjectDetectionEvaluator(evaluate_corlocs=True)).  Compiled with -ffp-contract=off: IoU must round
// like the reference's separate float32 numpy operations (effdet/evaluation/np_box_ops.py).
//
//   effdet_eval_match   one workgroup per image: detections in descending score order are matched greedily to the
//                       ground-truth boxes of their class (per_image_evaluation.py:377-405): the box with the largest IoU
//                       (first on ties), IoU >= threshold and not taken -> true positive.  Boxes with ymax <= ymin or
//                       xmax <= xmin are dropped (:514-538).  CorLoc (:143-176): the top-scoring detection of a class
//                       overlaps a box of that class.  Per-class counters are integer atomics (order independent).
//   effdet_eval_ap      metrics.py:4-90 per class without a sort: the rank of a detection inside its class is counted
//                       (score descending, ties by lower index), precision = cumulative TP / rank in float64, the VOC
//                       envelope is a max over the later ranks, and recall advances by 1/num_gt at every true positive.
//                       O(n^2) pair counting - the loop evaluates <= 100 detections x a handful of images per iteration.
#include "common.h"

namespace {

constexpr int ET = 256;

struct MatchArgs {
    const float* det; const int* det_count; const float* gt_boxes; const long long* gt_cls;
    int B, max_det, M, C; float thr;
    int* tp; int* gt_count; int* gt_imgs; int* correct_imgs;
};

__global__ __launch_bounds__(ET) void eval_match_kernel(MatchArgs p) {
    extern __shared__ int sh[];
    int* seen = sh;                                  // [C] class already had its top-scoring detection
    int* taken = sh + p.C;                           // [M]
    int* gcls = taken + p.M;                         // [M] 0-based class or -1
    float* red_v = reinterpret_cast<float*>(gcls + p.M);   // [4]
    int* red_i = reinterpret_cast<int*>(red_v + 4);         // [4]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int c = tid; c < p.C; c += ET) seen[c] = 0;
    for (int j = tid; j < p.M; j += ET) {
        const long long c1 = p.gt_cls[(long long)b * p.M + j];
        gcls[j] = (c1 >= 1 && c1 <= p.C) ? (int)(c1 - 1) : -1;
        taken[j] = 0;
    }
    __syncthreads();
    // ground-truth instance / image counters
    for (int j = tid; j < p.M; j += ET) {
        const int c = gcls[j];
        if (c < 0) continue;
        atomicAdd(p.gt_count + c, 1);
        bool first = true;
        for (int q = 0; q < j; ++q) first = first && gcls[q] != c;
        if (first) atomicAdd(p.gt_imgs + c, 1);
    }
    const int n = min(p.det_count[b], p.max_det);
    for (int i = 0; i < p.max_det; ++i) {
        int* out = p.tp + (long long)b * p.max_det + i;
        if (i >= n) { if (tid == 0) *out = -1; continue; }
        const float* d = p.det + ((long long)b * p.max_det + i) * 6;
        const float x1 = d[0], y1 = d[1], x2 = d[2], y2 = d[3];
        const int c = (int)d[5] - 1;
        if (!(y1 < y2 && x1 < x2) || c < 0 || c >= p.C) { if (tid == 0) *out = -1; continue; }      // uniform
        const float area_d = (y2 - y1) * (x2 - x1);
        float best = -1.f; int bj = 0x7fffffff;
        for (int j = tid; j < p.M; j += ET) {
            if (gcls[j] != c) continue;
            const float* g = p.gt_boxes + ((long long)b * p.M + j) * 4;             // yxyx
            const float ih = fmaxf(0.f, fminf(y2, g[2]) - fmaxf(y1, g[0]));
            const float iw = fmaxf(0.f, fminf(x2, g[3]) - fmaxf(x1, g[1]));
            const float inter = ih * iw;
            const float area_g = (g[2] - g[0]) * (g[3] - g[1]);
            const float iou = inter / (area_d + area_g - inter);
            if (iou > best) { best = iou; bj = j; }                                 // ascending j: first maximum kept
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o, 64);
            const int oj = __shfl_xor(bj, o, 64);
            if (ov > best || (ov == best && oj < bj)) { best = ov; bj = oj; }
        }
        __syncthreads();
        if (lane == 0) { red_v[wave] = best; red_i[wave] = bj; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < ET / 64; ++w)
                if (red_v[w] > best || (red_v[w] == best && red_i[w] < bj)) { best = red_v[w]; bj = red_i[w]; }
            int label = 0;
            const bool has_gt = bj != 0x7fffffff;
            if (has_gt && best >= p.thr && !taken[bj]) { label = 1; taken[bj] = 1; }
            *out = label;
            if (!seen[c]) {                                                          // top-scoring detection of class c
                seen[c] = 1;
                if (has_gt && best >= p.thr) atomicAdd(p.correct_imgs + c, 1);
            }
        }
        __syncthreads();
    }
}

struct ApArgs {
    const float* scores; const int* classes; const int* tp; int n, C;
    const int* gt_count; int* rank; double* prec; double* env; double* ap;
};

// pass 1: rank inside the class (1-based) and precision at that rank
__global__ __launch_bounds__(ET) void eval_rank_kernel(ApArgs p) {
    const int i = blockIdx.x * ET + threadIdx.x;
    if (i >= p.n) return;
    const int c = p.classes[i];
    if (c < 0 || p.tp[i] < 0) { p.rank[i] = 0; p.prec[i] = 0.0; return; }
    const float s = p.scores[i];
    int rank = 0, ctp = 0;
    for (int j = 0; j < p.n; ++j) {
        if (p.classes[j] != c || p.tp[j] < 0) continue;
        const float sj = p.scores[j];
        const bool before = sj > s || (sj == s && j <= i);
        rank += before ? 1 : 0;
        ctp += (before && p.tp[j] > 0) ? 1 : 0;
    }
    p.rank[i] = rank;
    p.prec[i] = (double)ctp / (double)rank;
}

// pass 2: VOC envelope at every true positive = max precision over the ranks >= its own
__global__ __launch_bounds__(ET) void eval_env_kernel(ApArgs p) {
    const int i = blockIdx.x * ET + threadIdx.x;
    if (i >= p.n) return;
    const int c = p.classes[i];
    double e = 0.0;
    if (c >= 0 && p.tp[i] > 0) {
        const int r = p.rank[i];
        for (int j = 0; j < p.n; ++j)
            if (p.classes[j] == c && p.tp[j] >= 0 && p.rank[j] >= r) e = fmax(e, p.prec[j]);
    }
    p.env[i] = e;
}

// pass 3: AP_c = sum over the class's true positives of (recall step) * envelope; one workgroup per class, fixed order
__global__ __launch_bounds__(ET) void eval_ap_kernel(ApArgs p) {
    __shared__ double sm[ET];
    const int c = blockIdx.x, tid = threadIdx.x;
    const int ng = p.gt_count[c];
    double acc = 0.0;
    for (int i = tid; i < p.n; i += ET)
        if (p.classes[i] == c && p.tp[i] > 0) acc += p.env[i];
    sm[tid] = acc;
    __syncthreads();
    for (int o = ET / 2; o > 0; o >>= 1) {
        if (tid < o) sm[tid] += sm[tid + o];
        __syncthreads();
    }
    if (tid == 0) p.ap[c] = ng > 0 ? sm[0] / (double)ng : __builtin_nan("");
}

}  // namespace

extern "C" int effdet_eval_match(void* stream, const float* det, const int* det_count, const float* gt_boxes,
                                 const long long* gt_cls, int B, int max_det, int M, int num_classes, float iou_threshold,
                                 int* tp, int* gt_count, int* gt_imgs, int* correct_imgs) {
    EFFDET_ENTER();
    if (!det || !det_count || !gt_boxes || !gt_cls || !tp || !gt_count || !gt_imgs || !correct_imgs) return EFFDET_EINVAL;
    if (B <= 0 || max_det <= 0 || M <= 0 || num_classes <= 0) return EFFDET_EINVAL;
    const size_t sh = (size_t)(num_classes + 2 * M + 8) * sizeof(int);
    if (sh > 64 * 1024) return EFFDET_EINVAL;
    MatchArgs a{det, det_count, gt_boxes, gt_cls, B, max_det, M, num_classes, iou_threshold, tp, gt_count, gt_imgs, correct_imgs};
    hipLaunchKernelGGL(eval_match_kernel, dim3(B), dim3(ET), sh, reinterpret_cast<hipStream_t>(stream), a);
    return effdet_check_launch();
}

extern "C" long long effdet_eval_ap_workspace_bytes(int n) {
    if (n <= 0) return EFFDET_EINVAL;
    return (long long)n * (sizeof(int) + 2 * sizeof(double)) + 16;
}

extern "C" int effdet_eval_ap(void* stream, const float* scores, const int* classes, const int* tp, int n, int num_classes,
                              const int* gt_count, double* ap, void* workspace, long long workspace_bytes) {
    EFFDET_ENTER();
    if (!scores || !classes || !tp || !gt_count || !ap || !workspace || n <= 0 || n > 65536 || num_classes <= 0) return EFFDET_EINVAL;
    if (workspace_bytes < effdet_eval_ap_workspace_bytes(n) || reinterpret_cast<uintptr_t>(workspace) % 8) return EFFDET_EINVAL;
    ApArgs a;
    a.scores = scores; a.classes = classes; a.tp = tp; a.n = n; a.C = num_classes; a.gt_count = gt_count; a.ap = ap;
    a.prec = reinterpret_cast<double*>(workspace);
    a.env = a.prec + n;
    a.rank = reinterpret_cast<int*>(a.env + n);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const unsigned blocks = (unsigned)((n + ET - 1) / ET);
    hipLaunchKernelGGL(eval_rank_kernel, dim3(blocks), dim3(ET), 0, st, a);
    hipLaunchKernelGGL(eval_env_kernel, dim3(blocks), dim3(ET), 0, st, a);
    hipLaunchKernelGGL(eval_ap_kernel, dim3((unsigned)num_classes), dim3(ET), 0, st, a);
    return effdet_check_launch();
}
