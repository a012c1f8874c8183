// Post-processing kernels (compiled with -ffp-contract=off: the box arithmetic must round exactly
// like the reference's separate fp32 multiply / add ops).
//
//   effdet_topk_select       _post_process (effdet/bench.py:35-56): top-k over the N*C class logits of
//                            every image, descending, ties by lower flat index; then index // C,
//                            index % C and the box / logit gathers.
//   effdet_decode_threshold  generate_detections part 1 (effdet/anchors.py:132-144): anchor gather,
//                            decode_box_outputs (:51-85), clip_boxes_xyxy (:88-92), sigmoid,
//                            `score > 0.01` order-preserving compaction, boxes.max().
//   effdet_nms_hard          torchvision batched_nms semantics (call site effdet/anchors.py:150).
//   effdet_nms_soft          batched_soft_nms / soft_nms (effdet/soft_nms.py:42-169).
//                            Both stop after max_det picks (anchors.py:153 keeps only those) and emit
//                            [B, max_det, 6] rows x1,y1,x2,y2,score,class+1 (anchors.py:154-166).
//   effdet_gather_ood        energy / max-logit of the anchors behind the kept detections.
//
// Top-k is an MSB-first radix select on 64-bit composite keys (order-preserving float key << 32 |
// ~index), 11 bits per pass.  After each pass the per-image state knows how many elements lie in and
// above the bin that holds the k-th key; as soon as that candidate set fits the 16384-entry LDS sort
// buffer the remaining passes return at once, the candidates are compacted and one workgroup per
// image bitonic-sorts them.
//
// Anchor prefilter (used when there are many more anchors than k).  Let M_a be the largest logit of anchor
// a and t the k-th largest M_a.  The k anchors with M_a >= t each own a pair (a, argmax) with logit >= t, so
// the k-th largest PAIR is >= t, hence every pair of the exact top-k belongs to an anchor with M_a >= t.
// Stage 1 radix-selects that anchor set on the [B, n_anchors] row maxima (22 key bits: a superset is fine),
// stage 2 runs the exact select above on the C logits of those ~k anchors only, with the original flat
// indices in the keys, so results (ties included) are identical to the dense select while the full logits
// tensor is never scanned.  M_a comes for free from the class head (the OOD max-logit output); without it
// one row-max pass over the logits computes it.
#include "common.h"

namespace {

constexpr int TOPK_CAP = 16384;
constexpr int HIST_BINS = 2048;
constexpr int NPASS = 6;
__constant__ int kPassShift[NPASS] = {53, 42, 32, 21, 10, 0};
__constant__ int kPassBits[NPASS] = {11, 11, 10, 11, 11, 10};

struct TopkState {            // one per image, 64 bytes
    unsigned long long prefix;   // bits fixed so far (right-aligned)
    int bits_done;
    int done;
    unsigned int c_hi;           // elements strictly above the prefix range
    unsigned int cand_total;     // candidates = c_hi + |bin|
    unsigned int cand_count;     // compaction cursor
    unsigned int pad[9];
};

DEV unsigned int float_key(float f) {
    const unsigned int u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
DEV float key_float(unsigned int k) {
    const unsigned int u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}
DEV unsigned long long comp_key(float f, unsigned int idx) {
    return ((unsigned long long)float_key(f) << 32) | (unsigned long long)(0xFFFFFFFFu - idx);
}

// Iterate the elements [seg0, seg1) of one image row, 16 bytes at a time where aligned.
template <typename T, typename Fn>
DEV void for_each_element(const T* row, long long seg0, long long seg1, int tid, int nthreads, Fn fn) {
    constexpr int EPC = VecTraits<T>::EPC;
    // first element index >= seg0 whose address is 16-byte aligned
    const uintptr_t addr0 = reinterpret_cast<uintptr_t>(row + seg0);
    long long head = ((16 - (addr0 & 15)) & 15) / (long long)sizeof(T);
    if (head > seg1 - seg0) head = seg1 - seg0;
    for (long long i = seg0 + tid; i < seg0 + head; i += nthreads) fn(to_f<T>(row[i]), (unsigned int)i);
    const long long v0 = seg0 + head;
    const long long nvec = (seg1 - v0) / EPC;
    for (long long v = tid; v < nvec; v += nthreads) {
        const long long i = v0 + v * EPC;
        if constexpr (sizeof(T) == 4) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(row + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) fn(x[e], (unsigned int)(i + e));
        } else {
            const bf16x8 x = *reinterpret_cast<const bf16x8*>(row + i);
#pragma unroll
            for (int e = 0; e < 8; ++e) fn((float)x[e], (unsigned int)(i + e));
        }
    }
    for (long long i = v0 + nvec * EPC + tid; i < seg1; i += nthreads) fn(to_f<T>(row[i]), (unsigned int)i);
}

// Element sources of the select: count(b) elements per image, each a (value, flat index) pair.
template <typename T>
struct DenseSrc {                      // every logit of the image
    const T* X; long long L;
    DEV long long count(int) const { return L; }
    template <typename Fn> DEV void for_each(int b, long long seg0, long long seg1, int tid, int nth, Fn fn) const {
        for_each_element<T>(X + (long long)b * L, seg0, seg1, tid, nth, fn);
    }
};
template <bool ROUND_BF16>
struct RowMaxSrc {                     // per-anchor maxima (fp32), rounded like the stored logits
    const float* M; long long n_anchors;
    DEV long long count(int) const { return n_anchors; }
    DEV static float val(float f) { return ROUND_BF16 ? (float)(bf16_t)f : f; }
    template <typename Fn> DEV void for_each(int b, long long seg0, long long seg1, int tid, int nth, Fn fn) const {
        const float* row = M + (long long)b * n_anchors;
        for (long long i0 = seg0; i0 < seg1; i0 += 4LL * nth) {      // 4 loads in flight per thread
            float x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const long long i = i0 + tid + (long long)u * nth; x[u] = i < seg1 ? row[i] : 0.f; }
#pragma unroll
            for (int u = 0; u < 4; ++u) { const long long i = i0 + tid + (long long)u * nth; if (i < seg1) fn(val(x[u]), (unsigned int)i); }
        }
    }
};
template <typename T>
struct PairSrc {                       // the C logits of every selected anchor
    const T* X; long long L; const int* asel; const TopkState* st1; long long n_anchors; int C;
    DEV long long count(int b) const {
        long long n = st1[b].cand_count; if (n > n_anchors) n = n_anchors;
        return n * C;
    }
    template <typename Fn> DEV void for_each(int b, long long seg0, long long seg1, int tid, int nth, Fn fn) const {
        // Row-oriented: a wave walks whole anchor rows (the row's first element decides which segment owns it),
        // 4 rows at a time so that the anchor-index loads and then all row loads are in flight together.
        const T* img = X + (long long)b * L;
        const int* al = asel + (long long)b * n_anchors;
        const long long rb = (seg0 + C - 1) / C, re = (seg1 + C - 1) / C;
        const int lane = tid & 63, wave = tid >> 6, nw = nth >> 6;
        constexpr int RB = 4, CL = 2;                         // rows per batch; 64-lane column strips (C <= 128 fast path)
        for (long long r0 = rb + wave * RB; r0 < re; r0 += (long long)nw * RB) {
            unsigned int a[RB];
#pragma unroll
            for (int u = 0; u < RB; ++u) a[u] = r0 + u < re ? (unsigned int)al[r0 + u] : 0xFFFFFFFFu;
            T x[RB][CL];
#pragma unroll
            for (int u = 0; u < RB; ++u)
#pragma unroll
                for (int q = 0; q < CL; ++q) {
                    const int c = lane + 64 * q;
                    if (a[u] != 0xFFFFFFFFu && c < C) x[u][q] = img[(long long)a[u] * C + c];
                }
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                if (a[u] == 0xFFFFFFFFu) continue;            // uniform per wave
#pragma unroll
                for (int q = 0; q < CL; ++q) {
                    const int c = lane + 64 * q;
                    if (c < C) fn(to_f<T>(x[u][q]), a[u] * (unsigned int)C + (unsigned int)c);
                }
                for (int c = lane + 64 * CL; c < C; c += 64)   // wider rows: the rest, unbatched
                    fn(to_f<T>(img[(long long)a[u] * C + c]), a[u] * (unsigned int)C + (unsigned int)c);
            }
        }
    }
};

// Append slot for the lanes of a wave whose `pred` holds: one atomic per wave instead of one per element
DEV unsigned int wave_append(unsigned int* counter, bool pred) {
    const unsigned long long m = __ballot(pred);
    if (m == 0ull) return 0u;                                 // uniform: nothing to append in this wave
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)m) - 1;
    unsigned int base = 0;
    if (pred && lane == leader) base = atomicAdd(counter, (unsigned int)__popcll(m));
    base = __shfl(base, leader < 0 ? 0 : leader, 64);
    return base + (unsigned int)__popcll(m & ((1ull << lane) - 1ull));
}

template <typename Src>
__global__ __launch_bounds__(256) void topk_hist_kernel(Src src, int pass, TopkState* state, unsigned int* hist) {
    const int b = blockIdx.y;
    const TopkState st = state[b];
    if (st.done) return;
    __shared__ unsigned int h[HIST_BINS];
    for (int i = threadIdx.x; i < HIST_BINS; i += 256) h[i] = 0;
    __syncthreads();
    const long long L = src.count(b);
    const long long per = (L + gridDim.x - 1) / gridDim.x;
    const long long seg0 = per * blockIdx.x;
    long long seg1 = seg0 + per; if (seg1 > L) seg1 = L;
    const int shift = kPassShift[pass];
    const unsigned int mask = (1u << kPassBits[pass]) - 1u;
    const int pshift = 64 - st.bits_done;
    const unsigned long long prefix = st.prefix;
    if (seg0 < seg1) {
        // logits cluster in a handful of bins: run-length aggregate per thread so that identical consecutive
        // bins cost one LDS atomic instead of one per element (same-address LDS atomics serialise)
        unsigned int last_bin = 0xFFFFFFFFu, run = 0;
        src.for_each(b, seg0, seg1, threadIdx.x, 256, [&](float f, unsigned int idx) {
            const unsigned long long key = comp_key(f, idx);
            if (pass == 0 || (key >> pshift) == prefix) {
                const unsigned int bin = (unsigned int)(key >> shift) & mask;
                if (bin == last_bin) { ++run; }
                else { if (run) atomicAdd(&h[last_bin], run); last_bin = bin; run = 1; }
            }
        });
        if (run) atomicAdd(&h[last_bin], run);
    }
    __syncthreads();
    unsigned int* gh = hist + (long long)b * HIST_BINS;
    for (int i = threadIdx.x; i < HIST_BINS; i += 256) if (h[i]) atomicAdd(&gh[i], h[i]);
}

__global__ __launch_bounds__(256) void topk_find_kernel(int pass, int last_pass, int k, unsigned int cap, TopkState* state, unsigned int* hist) {
    const int b = blockIdx.x, tid = threadIdx.x;
    TopkState st = state[b];
    if (st.done) return;
    unsigned int* gh = hist + (long long)b * HIST_BINS;
    __shared__ unsigned int part[256];
    __shared__ unsigned int sel[3];      // bin, elements above it (inside this prefix), elements in it
    // thread t owns bins [8t, 8t+8); suffix sums from the top
    unsigned int loc[8], s = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) { loc[e] = gh[tid * 8 + e]; s += loc[e]; gh[tid * 8 + e] = 0; }
    // exclusive suffix sums over the 256 per-thread totals (Hillis-Steele, 8 steps)
    unsigned int incl = s;
    part[tid] = incl;
    __syncthreads();
#pragma unroll
    for (int off = 1; off < 256; off <<= 1) {
        const unsigned int add = tid + off < 256 ? part[tid + off] : 0u;
        __syncthreads();
        incl += add;
        part[tid] = incl;
        __syncthreads();
    }
    const unsigned int need = (unsigned int)k - st.c_hi;
    unsigned int above = incl - s;       // elements in bins owned by higher threads
    if (above < need && above + s >= need) {
        for (int e = 7; e >= 0; --e) {
            if (above + loc[e] >= need) { sel[0] = tid * 8 + e; sel[1] = above; sel[2] = loc[e]; break; }
            above += loc[e];
        }
    }
    __syncthreads();
    if (tid == 0) {
        const int bits = kPassBits[pass];
        st.prefix = (st.prefix << bits) | (unsigned long long)sel[0];
        st.bits_done += bits;
        st.c_hi += sel[1];
        st.cand_total = st.c_hi + sel[2];
        if (st.cand_total <= cap || pass == last_pass) st.done = 1;
        state[b] = st;
    }
}

// Compaction: candidates are gathered in LDS and flushed with ONE global reservation per workgroup
// (thousands of same-address global atomics per image would serialise at the L2).
constexpr int COLLECT_BUF = 1024;

template <typename Src, typename Item, typename Make>
DEV void collect_common(const Src& src, const TopkState* thr, TopkState* sp, int b, Item* out, unsigned int out_cap, Make make) {
    __shared__ Item buf[COLLECT_BUF];
    __shared__ unsigned int cnt, base;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    const int pshift = 64 - thr->bits_done;
    const unsigned long long prefix = thr->prefix;
    const long long L = src.count(b);
    const long long per = (L + gridDim.x - 1) / gridDim.x;
    const long long seg0 = per * blockIdx.x;
    long long seg1 = seg0 + per; if (seg1 > L) seg1 = L;
    if (seg0 < seg1) {
        src.for_each(b, seg0, seg1, threadIdx.x, 256, [&](float f, unsigned int idx) {
            const unsigned long long key = comp_key(f, idx);
            const bool take = (key >> pshift) >= prefix;
            const unsigned int pos = wave_append(&cnt, take);
            if (take) {
                if (pos < (unsigned int)COLLECT_BUF) buf[pos] = make(key, idx);
                else {                                         // buffer full: straight to memory
                    const unsigned int g = atomicAdd(&sp->cand_count, 1u);
                    if (g < out_cap) out[g] = make(key, idx);
                }
            }
        });
    }
    __syncthreads();
    const unsigned int n = cnt < (unsigned int)COLLECT_BUF ? cnt : (unsigned int)COLLECT_BUF;
    if (threadIdx.x == 0 && n) base = atomicAdd(&sp->cand_count, n);
    __syncthreads();
    for (unsigned int i = threadIdx.x; i < n; i += 256) if (base + i < out_cap) out[base + i] = buf[i];
}

template <typename Src>
__global__ __launch_bounds__(256) void topk_collect_kernel(Src src, const TopkState* thr, TopkState* state, unsigned long long* cand) {
    const int b = blockIdx.y;
    collect_common(src, thr + b, state + b, b, cand + (long long)b * TOPK_CAP, (unsigned int)TOPK_CAP,
                   [](unsigned long long key, unsigned int) { return key; });
}

// Stage 1 in ONE launch (round 4; it was hist + find + hist + find + collect = 5 launches of 5 - 12 us each): a workgroup of 1024
// threads per image walks the image's row maxima (0.3 - 0.8 MB, L2-resident: the class head has just written them) three times -
// two 11-bit radix passes with the histogram in LDS, then the compaction of the anchors that reach the selected prefix - and
// resets the image's stage-2 state, so no memset launch is needed either.  Same arithmetic and the same selected SET as the
// multi-launch form (the order inside `asel` is arbitrary in both).
template <typename Src>
__global__ __launch_bounds__(1024) void anchor_select_kernel(Src src, int k, TopkState* state1, TopkState* state2, int* asel) {
    const int b = blockIdx.x, tid = threadIdx.x;
    __shared__ unsigned int h[HIST_BINS];
    __shared__ unsigned int part[1024];
    __shared__ unsigned int sel[3];
    __shared__ unsigned int s_cnt;
    const long long L = src.count(b);
    unsigned long long prefix = 0;
    int bits_done = 0;
    unsigned int c_hi = 0, cand_total = 0;
    for (int pass = 0; pass < 2; ++pass) {
        for (int i = tid; i < HIST_BINS; i += 1024) h[i] = 0;
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        const int shift = kPassShift[pass];
        const unsigned int mask = (1u << kPassBits[pass]) - 1u;
        const int pshift = 64 - bits_done;
        unsigned int last_bin = 0xFFFFFFFFu, run = 0;
        src.for_each(b, 0, L, tid, 1024, [&](float f, unsigned int idx) {
            const unsigned long long key = comp_key(f, idx);
            if (pass == 0 || (key >> pshift) == prefix) {
                const unsigned int bin = (unsigned int)(key >> shift) & mask;
                if (bin == last_bin) { ++run; }
                else { if (run) atomicAdd(&h[last_bin], run); last_bin = bin; run = 1; }
            }
        });
        if (run) atomicAdd(&h[last_bin], run);
        __syncthreads();
        // thread t owns bins 2t, 2t + 1; suffix sums from the top over the 1024 per-thread totals (Hillis-Steele, 10 steps)
        const unsigned int l0 = h[2 * tid], l1 = h[2 * tid + 1], sown = l0 + l1;
        unsigned int incl = sown;
        part[tid] = incl;
        __syncthreads();
#pragma unroll
        for (int off = 1; off < 1024; off <<= 1) {
            const unsigned int add = tid + off < 1024 ? part[tid + off] : 0u;
            __syncthreads();
            incl += add;
            part[tid] = incl;
            __syncthreads();
        }
        const unsigned int need = (unsigned int)k - c_hi;
        unsigned int above = incl - sown;
        if (above < need && above + sown >= need) {
            if (above + l1 >= need) { sel[0] = 2 * tid + 1; sel[1] = above; sel[2] = l1; }
            else { sel[0] = 2 * tid; sel[1] = above + l1; sel[2] = l0; }
        }
        __syncthreads();
        const int bits = kPassBits[pass];
        prefix = (prefix << bits) | (unsigned long long)sel[0];
        bits_done += bits;
        c_hi += sel[1];
        cand_total = c_hi + sel[2];
        __syncthreads();
        if (cand_total <= (unsigned int)k + 1024u) break;           // (the multi-launch form's early stop)
    }
    // compaction: anchors whose key reaches the prefix
    {
        const int pshift = 64 - bits_done;
        int* out = asel + (long long)b * L;
        src.for_each(b, 0, L, tid, 1024, [&](float f, unsigned int idx) {
            const unsigned long long key = comp_key(f, idx);
            const bool take = (key >> pshift) >= prefix;
            const unsigned int pos = wave_append(&s_cnt, take);
            if (take && pos < (unsigned int)L) out[pos] = (int)idx;
        });
    }
    __syncthreads();
    if (tid == 0) {
        TopkState st{};
        st.prefix = prefix; st.bits_done = bits_done; st.done = 1; st.c_hi = c_hi; st.cand_total = cand_total; st.cand_count = s_cnt;
        state1[b] = st;
        state2[b] = TopkState{};                                     // stage 2 starts from a clean state (no memset launch)
    }
}

// After the fast stage-2 compaction (every pair whose key reaches the stage-1 threshold): when the candidates
// fit the sort buffer they already contain the exact top k.  Otherwise (massive ties) this workgroup redoes
// the image with the full multi-pass radix select over the gathered rows - slow, but only degenerate inputs
// get here.
template <typename T>
DEV void pair_finish(const PairSrc<T>& src, int k, TopkState* state, unsigned long long* cand) {
    const int b = blockIdx.x, tid = threadIdx.x;
    TopkState* sp = state + b;
    __shared__ unsigned int h[HIST_BINS];
    __shared__ unsigned long long s_prefix;
    __shared__ unsigned int s_chi, s_total, s_cnt;
    __shared__ int s_bits, s_done;
    if (sp->cand_count <= (unsigned int)TOPK_CAP) {                 // uniform across the workgroup
        __syncthreads();                                            // (everyone has read cand_count before it may be touched)
        if (tid == 0) { sp->cand_total = sp->cand_count; sp->done = 1; }
        return;
    }
    if (tid == 0) { s_prefix = 0; s_chi = 0; s_bits = 0; s_done = 0; s_cnt = 0; }
    __syncthreads();
    const long long L = src.count(b);
    for (int pass = 0; pass < NPASS; ++pass) {
        for (int i = tid; i < HIST_BINS; i += 1024) h[i] = 0;
        __syncthreads();
        const int shift = kPassShift[pass];
        const unsigned int mask = (1u << kPassBits[pass]) - 1u;
        const int pshift = 64 - s_bits;
        const unsigned long long prefix = s_prefix;
        src.for_each(b, 0, L, tid, 1024, [&](float f, unsigned int idx) {
            const unsigned long long key = comp_key(f, idx);
            if (pass == 0 || (key >> pshift) == prefix) atomicAdd(&h[(unsigned int)(key >> shift) & mask], 1u);
        });
        __syncthreads();
        if (tid == 0) {
            const unsigned int need = (unsigned int)k - s_chi;
            unsigned int above = 0;
            int bin = (int)mask;
            for (; bin > 0; --bin) { if (above + h[bin] >= need) break; above += h[bin]; }
            s_prefix = (s_prefix << kPassBits[pass]) | (unsigned long long)bin;
            s_bits += kPassBits[pass];
            s_chi += above;
            s_total = s_chi + h[bin];
            if (s_total <= (unsigned int)TOPK_CAP || pass == NPASS - 1) s_done = 1;
        }
        __syncthreads();
        if (s_done) break;
    }
    {
        const int pshift = 64 - s_bits;
        const unsigned long long prefix = s_prefix;
        unsigned long long* out = cand + (long long)b * TOPK_CAP;
        src.for_each(b, 0, L, tid, 1024, [&](float f, unsigned int idx) {
            const unsigned long long key = comp_key(f, idx);
            if ((key >> pshift) >= prefix) {
                const unsigned int pos = atomicAdd(&s_cnt, 1u);
                if (pos < (unsigned int)TOPK_CAP) out[pos] = key;
            }
        });
    }
    __syncthreads();
    if (tid == 0) { sp->cand_total = s_total; sp->done = 1; }
}

// Row maxima of the logits, for callers that do not have them: one wave per anchor row
template <typename T>
__global__ __launch_bounds__(256) void row_max_kernel(const T* X, long long rows, int C, float* M) {
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    float m = -INFINITY;
    for (int c = lane; c < C; c += 64) m = fmaxf(m, to_f<T>(X[r * C + c]));
    m = wave_reduce_max(m);
    if (lane == 0) M[r] = m;
}

// FINISH: the launch follows the prefilter's pair compaction and first closes it (pair_finish: a no-op unless the candidates
// overflow the sort buffer) - one launch instead of two
template <typename T, bool FINISH>
__global__ __launch_bounds__(1024) void topk_sort_kernel(TopkState* state, unsigned long long* cand,
                                                         int k, int C, const T* cls_all, const T* box_all,
                                                         long long L, long long n_anchors,
                                                         T* out_cls, T* out_box, long long* out_idx, long long* out_cls_id,
                                                         PairSrc<T> fsrc) {
    // Merge sort, descending, TOPK_CAP slots (unused ones hold key 0, below every real key): every thread sorts
    // its 16 keys in registers, then ten merge rounds double the run length.  In a round a thread owns 16
    // consecutive output slots: a merge-path binary search finds where they start in the two input runs, a
    // 16-step sequential merge reads the heads from LDS, and the outputs are written back in place after
    // the barrier.  O(n log n) compare work instead of the bitonic network's O(n log^2 n).
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
    const int b = blockIdx.x, tid = threadIdx.x;
    if constexpr (FINISH) {
        pair_finish<T>(fsrc, k, state, cand);
        __threadfence_block();
        __syncthreads();
    }
    unsigned int n = FINISH ? (state[b].cand_count <= (unsigned int)TOPK_CAP ? state[b].cand_count : state[b].cand_total) : state[b].cand_total;
    if (n > (unsigned int)TOPK_CAP) n = TOPK_CAP;
    constexpr int E = TOPK_CAP / 1024;                       // 16 keys per thread
    auto slot = [](int i) { return i + (i >> 4); };          // one pad slot per 16: the per-thread 128-byte rows spread over the banks
    const unsigned long long* src = cand + (long long)b * TOPK_CAP;
    unsigned long long v[E];
#pragma unroll
    for (int m = 0; m < E; ++m) { const int i = tid * E + m; v[m] = i < (int)n ? src[i] : 0ull; }
#pragma unroll
    for (int size = 2; size <= E; size <<= 1)
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1)
#pragma unroll
            for (int m = 0; m < E; ++m)
                if ((m & stride) == 0) {
                    const bool desc = (m & size) == 0;
                    const unsigned long long x = v[m], y = v[m | stride];
                    const bool sw = (x < y) == desc;
                    v[m] = sw ? y : x; v[m | stride] = sw ? x : y;
                }
#pragma unroll
    for (int m = 0; m < E; ++m) keys[slot(tid * E + m)] = v[m];
    __syncthreads();
    for (int run = E; run < TOPK_CAP; run <<= 1) {
        const int o0 = tid * E;                               // first output slot of this thread
        const int pair0 = o0 & ~(2 * run - 1);                // start of the X run; Y follows at pair0 + run
        const int d = o0 - pair0;                             // outputs of this pair that precede ours
        const int X0 = pair0, Y0 = pair0 + run;
        int lo = d > run ? d - run : 0, hi = d < run ? d : run;
        while (lo < hi) {                                     // xi = how many of the first d outputs come from X
            const int mid = (lo + hi) >> 1;
            if (keys[slot(X0 + mid)] > keys[slot(Y0 + d - 1 - mid)]) lo = mid + 1; else hi = mid;
        }
        int xi = lo, yi = d - lo;
        unsigned long long hx = xi < run ? keys[slot(X0 + xi)] : 0ull, hy = yi < run ? keys[slot(Y0 + yi)] : 0ull;
#pragma unroll
        for (int o = 0; o < E; ++o) {
            const bool takex = yi >= run || (xi < run && hx > hy);
            v[o] = takex ? hx : hy;
            if (takex) { ++xi; hx = xi < run ? keys[slot(X0 + xi)] : 0ull; }
            else       { ++yi; hy = yi < run ? keys[slot(Y0 + yi)] : 0ull; }
        }
        __syncthreads();                                      // every thread has read its inputs
#pragma unroll
        for (int o = 0; o < E; ++o) keys[slot(o0 + o)] = v[o];
        __syncthreads();
    }
    for (int i = tid; i < k; i += 1024) {
        const unsigned long long key = keys[slot(i)];
        // fewer than k candidates can only happen with NaN logits (the per-anchor maxima skip NaNs, the key order
        // does not): stay in bounds and emit element 0 for the missing ranks
        unsigned int flat = 0xFFFFFFFFu - (unsigned int)(key & 0xFFFFFFFFull);
        if (i >= (int)n || (long long)flat >= L) flat = 0u;
        const long long o = (long long)b * k + i;
        const long long anchor = flat / (unsigned int)C;
        out_idx[o] = anchor;
        out_cls_id[o] = flat % (unsigned int)C;
        out_cls[o] = cls_all[(long long)b * L + flat];
        if (box_all != nullptr) {
            const T* bs = box_all + ((long long)b * n_anchors + anchor) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) out_box[o * 4 + e] = bs[e];
        }
    }
}

// ------------------------------------------------------------------------------------------------
struct DecodeArgs {
    const void* cls; const void* box; int dtype; int box_dtype;
    const float* anchors;
    const long long* indices; const long long* classes;
    const float* img_scale; const float* img_size;
    int k;
    float* boxes; float* scores; int* cls_out; int* src; int* count; float* maxcoord;
    long long gather_anchors;      // > 0: `box` is the full [B, gather_anchors, 4] head output, rows taken through `indices`
};

DEV float ld_as_float(const void* p, int dtype, long long i) {
    return dtype == 0 ? reinterpret_cast<const float*>(p)[i] : (float)reinterpret_cast<const bf16_t*>(p)[i];
}

DEV void decode_threshold_body(const DecodeArgs& p) {
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ int wcount[4];
    __shared__ float wmax[4];
    __shared__ int base_s;
    if (tid == 0) base_s = 0;
    float mx = -INFINITY;
    const bool clip = p.img_scale != nullptr && p.img_size != nullptr;
    float cw = 0.f, chh = 0.f;
    if (clip) { const float s = p.img_scale[b]; cw = p.img_size[b * 2] / s; chh = p.img_size[b * 2 + 1] / s; }
    __syncthreads();
    for (int i0 = 0; i0 < p.k; i0 += 256) {
        const int i = i0 + tid;
        bool keep = false;
        float x1 = 0, y1 = 0, x2 = 0, y2 = 0, sc = 0;
        if (i < p.k) {
            const long long o = (long long)b * p.k + i;
            const float* a = p.anchors + p.indices[o] * 4;
            const float ya = (a[0] + a[2]) / 2, xa = (a[1] + a[3]) / 2;
            const float ha = a[2] - a[0], wa = a[3] - a[1];
            const long long br = p.gather_anchors > 0 ? (long long)b * p.gather_anchors + p.indices[o] : o;
            const float ty = ld_as_float(p.box, p.box_dtype, br * 4 + 0), tx = ld_as_float(p.box, p.box_dtype, br * 4 + 1);
            const float th = ld_as_float(p.box, p.box_dtype, br * 4 + 2), tw = ld_as_float(p.box, p.box_dtype, br * 4 + 3);
            const float w = expf(tw) * wa, h = expf(th) * ha;
            const float yc = ty * ha + ya, xc = tx * wa + xa;
            y1 = yc - h / 2.f; x1 = xc - w / 2.f; y2 = yc + h / 2.f; x2 = xc + w / 2.f;
            if (clip) {
                x1 = fminf(fmaxf(x1, 0.f), cw); y1 = fminf(fmaxf(y1, 0.f), chh);
                x2 = fminf(fmaxf(x2, 0.f), cw); y2 = fminf(fmaxf(y2, 0.f), chh);
            }
            sc = 1.0f / (1.0f + expf(-ld_as_float(p.cls, p.dtype, o)));
            keep = sc > 0.01f;
        }
        const unsigned long long m = __ballot(keep);
        if (lane == 0) wcount[wave] = __popcll(m);
        __syncthreads();
        int off = base_s;
        for (int w = 0; w < wave; ++w) off += wcount[w];
        const int total = wcount[0] + wcount[1] + wcount[2] + wcount[3];
        if (keep) {
            const int pos = off + __popcll(m & ((1ull << lane) - 1ull));
            const long long o = (long long)b * p.k + pos;
            p.boxes[o * 4 + 0] = x1; p.boxes[o * 4 + 1] = y1; p.boxes[o * 4 + 2] = x2; p.boxes[o * 4 + 3] = y2;
            p.scores[o] = sc;
            p.cls_out[o] = (int)p.classes[(long long)b * p.k + i];
            p.src[o] = i;
            mx = fmaxf(mx, fmaxf(fmaxf(x1, y1), fmaxf(x2, y2)));
        }
        __syncthreads();
        if (tid == 0) base_s += total;
        __syncthreads();
    }
    mx = wave_reduce_max(mx);
    if (lane == 0) wmax[wave] = mx;
    __syncthreads();
    if (tid == 0) {
        p.count[b] = base_s;
        p.maxcoord[b] = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    }
}

__global__ __launch_bounds__(256) void decode_threshold_kernel(DecodeArgs p) { decode_threshold_body(p); }

// ------------------------------------------------------------------------------------------------
struct NmsArgs {
    const float* boxes; const float* scores; const int* classes; const int* src; const int* count;
    const float* maxcoord; int k;
    double iou_thr; int max_det;
    const float* img_scale;
    float* det; int* det_count; int* keep_src;
    int gaussian; float sigma; float soft_iou_thr; float score_thr;
};

DEV void write_det(const NmsArgs& p, int b, int slot, int pos, float score) {
    const long long o = (long long)b * p.k + pos;
    const float s = p.img_scale ? p.img_scale[b] : 1.0f;
    float* d = p.det + ((long long)b * p.max_det + slot) * 6;
    if (p.img_scale) { d[0] = p.boxes[o * 4] * s; d[1] = p.boxes[o * 4 + 1] * s; d[2] = p.boxes[o * 4 + 2] * s; d[3] = p.boxes[o * 4 + 3] * s; }
    else { d[0] = p.boxes[o * 4]; d[1] = p.boxes[o * 4 + 1]; d[2] = p.boxes[o * 4 + 2]; d[3] = p.boxes[o * 4 + 3]; }
    d[4] = score;
    d[5] = (float)(p.classes[o] + 1);
    p.keep_src[(long long)b * p.max_det + slot] = p.src[o];
}

// torchvision-CPU IoU test: suppress when (float IoU, widened) > double threshold
DEV bool nms_suppresses(float ix1, float iy1, float ix2, float iy2, float iarea,
                        float jx1, float jy1, float jx2, float jy2, float jarea, double thr) {
    const float xx1 = fmaxf(ix1, jx1), yy1 = fmaxf(iy1, jy1);
    const float xx2 = fminf(ix2, jx2), yy2 = fminf(iy2, jy2);
    const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
    const float inter = w * h;
    const float ovr = inter / (iarea + jarea - inter);
    return (double)ovr > thr;
}

constexpr int NMS_MAX_DET = 512;

DEV void nms_hard_body(const NmsArgs& p) {
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ float kx1[NMS_MAX_DET], ky1[NMS_MAX_DET], kx2[NMS_MAX_DET], ky2[NMS_MAX_DET], kar[NMS_MAX_DET];
    __shared__ float cx1[256], cy1[256], cx2[256], cy2[256], car[256];
    __shared__ unsigned long long alive_mask[4];
    __shared__ int nkept_s;
    const int n = p.count[b];
    const float off1 = p.maxcoord[b] + 1.0f;
    if (tid == 0) nkept_s = 0;
    // zero-fill outputs
    for (int i = tid; i < p.max_det * 6; i += 256) p.det[(long long)b * p.max_det * 6 + i] = 0.f;
    for (int i = tid; i < p.max_det; i += 256) p.keep_src[(long long)b * p.max_det + i] = -1;
    __syncthreads();
    for (int c0 = 0; c0 < n; c0 += 256) {
        const int nk = nkept_s;
        if (nk >= p.max_det) break;
        const int i = c0 + tid;
        bool alive = i < n;
        float x1 = 0, y1 = 0, x2 = 0, y2 = 0, ar = 0;
        if (alive) {
            const long long o = (long long)b * p.k + i;
            const float offs = (float)p.classes[o] * off1;
            x1 = p.boxes[o * 4] + offs; y1 = p.boxes[o * 4 + 1] + offs;
            x2 = p.boxes[o * 4 + 2] + offs; y2 = p.boxes[o * 4 + 3] + offs;
            ar = (x2 - x1) * (y2 - y1);
            for (int j = 0; j < nk; ++j)
                if (nms_suppresses(kx1[j], ky1[j], kx2[j], ky2[j], kar[j], x1, y1, x2, y2, ar, p.iou_thr)) { alive = false; break; }
        }
        cx1[tid] = x1; cy1[tid] = y1; cx2[tid] = x2; cy2[tid] = y2; car[tid] = ar;
        // sequential resolution inside the chunk
        for (;;) {
            const unsigned long long m = __ballot(alive);
            if (lane == 0) alive_mask[wave] = m;
            __syncthreads();
            int pivot = -1;
#pragma unroll
            for (int w = 3; w >= 0; --w) if (alive_mask[w]) pivot = w * 64 + __ffsll((long long)alive_mask[w]) - 1;
            const int nk2 = nkept_s;
            if (pivot < 0 || nk2 >= p.max_det) { __syncthreads(); break; }
            if (tid == pivot) {
                kx1[nk2] = x1; ky1[nk2] = y1; kx2[nk2] = x2; ky2[nk2] = y2; kar[nk2] = ar;
                write_det(p, b, nk2, c0 + pivot, p.scores[(long long)b * p.k + c0 + pivot]);
                nkept_s = nk2 + 1;
                alive = false;
            } else if (alive && tid > pivot) {
                if (nms_suppresses(cx1[pivot], cy1[pivot], cx2[pivot], cy2[pivot], car[pivot], x1, y1, x2, y2, ar, p.iou_thr))
                    alive = false;
            }
            __syncthreads();
        }
    }
    __syncthreads();
    if (tid == 0) p.det_count[b] = nkept_s;
}

__global__ __launch_bounds__(256) void nms_hard_kernel(NmsArgs p) { nms_hard_body(p); }

// generate_detections' hard-NMS path in ONE launch (round 4): the image's workgroup decodes + thresholds its k candidates into
// the compacted arrays and goes straight on to the greedy NMS over them (same 256 threads, same arithmetic, same arrays)
__global__ __launch_bounds__(256) void detections_hard_kernel(DecodeArgs d, NmsArgs n) {
    decode_threshold_body(d);
    __threadfence();                      // the compacted candidates / count / maxcoord of THIS image are visible to its own workgroup
    __syncthreads();
    nms_hard_body(n);
}

// Soft-NMS: every thread keeps its candidates (index i = tid + 1024*q) in registers.
constexpr int SOFT_Q = 8;         // supports k <= 8192

__global__ __launch_bounds__(1024) void nms_soft_kernel(NmsArgs p) {
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ unsigned long long wbest[16];
    __shared__ float top[5];
    __shared__ unsigned long long best_s;
    const int n = p.count[b];
    const float off1 = p.maxcoord[b] + 1.0f;
    for (int i = tid; i < p.max_det * 6; i += 1024) p.det[(long long)b * p.max_det * 6 + i] = 0.f;
    for (int i = tid; i < p.max_det; i += 1024) p.keep_src[(long long)b * p.max_det + i] = -1;
    float x1[SOFT_Q], y1[SOFT_Q], x2[SOFT_Q], y2[SOFT_Q], sc[SOFT_Q];
    const int nq = (n + 1023) / 1024;
#pragma unroll
    for (int q = 0; q < SOFT_Q; ++q) {
        sc[q] = -1.f; x1[q] = y1[q] = x2[q] = y2[q] = 0.f;
        const int i = tid + 1024 * q;
        if (q < nq && i < n) {
            const long long o = (long long)b * p.k + i;
            const float offs = (float)p.classes[o] * off1;
            x1[q] = p.boxes[o * 4] + offs; y1[q] = p.boxes[o * 4 + 1] + offs;
            x2[q] = p.boxes[o * 4 + 2] + offs; y2[q] = p.boxes[o * 4 + 3] + offs;
            sc[q] = p.scores[o];
        }
    }
    int count = 0;
    for (; count < p.max_det; ++count) {
        // argmax over alive scores, ties -> lowest index
        unsigned long long best = 0ull;
#pragma unroll
        for (int q = 0; q < SOFT_Q; ++q) if (q < nq && sc[q] >= 0.f) {
            const unsigned long long key = ((unsigned long long)__float_as_uint(sc[q]) << 32) |
                                           (unsigned long long)(0xFFFFFFFFu - (unsigned int)(tid + 1024 * q));
            // +1 in the top word so that a live score of 0.0 still beats "nothing"
            const unsigned long long k1 = key + (1ull << 32);
            best = k1 > best ? k1 : best;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long other = __shfl_xor(best, o, 64);
            best = other > best ? other : best;
        }
        if (lane == 0) wbest[wave] = best;
        __syncthreads();
        if (tid == 0) {
            unsigned long long m = 0ull;
            for (int w = 0; w < 16; ++w) m = wbest[w] > m ? wbest[w] : m;
            best_s = m;
        }
        __syncthreads();
        const unsigned long long bsel = best_s;
        if (bsel == 0ull) break;
        const int ti = (int)(0xFFFFFFFFu - (unsigned int)(bsel & 0xFFFFFFFFull));
        if (tid == (ti & 1023)) {
            const int q = ti >> 10;
            float tx1 = 0, ty1 = 0, tx2 = 0, ty2 = 0, ts = 0;
#pragma unroll
            for (int qq = 0; qq < SOFT_Q; ++qq) if (qq == q) { tx1 = x1[qq]; ty1 = y1[qq]; tx2 = x2[qq]; ty2 = y2[qq]; ts = sc[qq]; }
            top[0] = tx1; top[1] = ty1; top[2] = tx2; top[3] = ty2; top[4] = ts;
            write_det(p, b, count, ti, ts);
        }
        __syncthreads();
        const float tx1 = top[0], ty1 = top[1], tx2 = top[2], ty2 = top[3];
        const float tarea = (tx2 - tx1) * (ty2 - ty1);
#pragma unroll
        for (int q = 0; q < SOFT_Q; ++q) if (q < nq && sc[q] >= 0.f) {
            const float area2 = (x2[q] - x1[q]) * (y2[q] - y1[q]);
            const float w = fmaxf(fminf(tx2, x2[q]) - fmaxf(tx1, x1[q]), 0.f);
            const float h = fmaxf(fminf(ty2, y2[q]) - fmaxf(ty1, y1[q]), 0.f);
            const float inter = w * h;
            const float iou = inter > 0.f ? inter / (tarea + area2 - inter) : 0.f;
            float decay;
            if (p.gaussian) decay = expf(-(iou * iou) / p.sigma);
            else decay = iou > p.soft_iou_thr ? 1.0f - iou : 1.0f;
            const float ns = sc[q] * decay;
            const bool keep = (ns > p.score_thr) && ((tid + 1024 * q) != ti);
            sc[q] = keep ? ns : -1.f;
        }
        __syncthreads();
    }
    if (tid == 0) p.det_count[b] = count;
}

// The same algorithm for any n (the stand-alone soft_nms API, effdet/soft_nms.py:42-112, has no size limit): the working
// scores live in a caller-provided scratch row, boxes are re-read from global memory (L2) each round.
__global__ __launch_bounds__(1024) void nms_soft_global_kernel(NmsArgs p, float* sc_ws) {
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ unsigned long long wbest[16];
    __shared__ float top[5];
    __shared__ unsigned long long best_s;
    const int n = p.count[b];
    const float off1 = p.maxcoord[b] + 1.0f;
    float* sc = sc_ws + (long long)b * p.k;
    const float* bx = p.boxes + (long long)b * p.k * 4;
    const int* cls = p.classes + (long long)b * p.k;
    for (int i = tid; i < p.max_det * 6; i += 1024) p.det[(long long)b * p.max_det * 6 + i] = 0.f;
    for (int i = tid; i < p.max_det; i += 1024) p.keep_src[(long long)b * p.max_det + i] = -1;
    for (int i = tid; i < n; i += 1024) sc[i] = p.scores[(long long)b * p.k + i];
    __syncthreads();
    int count = 0;
    for (; count < p.max_det; ++count) {
        unsigned long long best = 0ull;
        for (int i = tid; i < n; i += 1024) {
            const float s = sc[i];
            if (s >= 0.f) {
                const unsigned long long k1 = (((unsigned long long)__float_as_uint(s) << 32) |
                                               (unsigned long long)(0xFFFFFFFFu - (unsigned int)i)) + (1ull << 32);
                best = k1 > best ? k1 : best;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long other = __shfl_xor(best, o, 64);
            best = other > best ? other : best;
        }
        if (lane == 0) wbest[wave] = best;
        __syncthreads();
        if (tid == 0) {
            unsigned long long m = 0ull;
            for (int w = 0; w < 16; ++w) m = wbest[w] > m ? wbest[w] : m;
            best_s = m;
            if (m != 0ull) {
                const int ti = (int)(0xFFFFFFFFu - (unsigned int)(m & 0xFFFFFFFFull));
                const float offs = (float)cls[ti] * off1;
                top[0] = bx[ti * 4] + offs; top[1] = bx[ti * 4 + 1] + offs; top[2] = bx[ti * 4 + 2] + offs; top[3] = bx[ti * 4 + 3] + offs;
                top[4] = sc[ti];
                write_det(p, b, count, ti, sc[ti]);
            }
        }
        __syncthreads();
        const unsigned long long bsel = best_s;
        if (bsel == 0ull) break;
        const int ti = (int)(0xFFFFFFFFu - (unsigned int)(bsel & 0xFFFFFFFFull));
        const float tx1 = top[0], ty1 = top[1], tx2 = top[2], ty2 = top[3];
        const float tarea = (tx2 - tx1) * (ty2 - ty1);
        for (int i = tid; i < n; i += 1024) {
            const float s = sc[i];
            if (s >= 0.f) {
                const float offs = (float)cls[i] * off1;
                const float x1 = bx[i * 4] + offs, y1 = bx[i * 4 + 1] + offs, x2 = bx[i * 4 + 2] + offs, y2 = bx[i * 4 + 3] + offs;
                const float area2 = (x2 - x1) * (y2 - y1);
                const float w = fmaxf(fminf(tx2, x2) - fmaxf(tx1, x1), 0.f);
                const float h = fmaxf(fminf(ty2, y2) - fmaxf(ty1, y1), 0.f);
                const float inter = w * h;
                const float iou = inter > 0.f ? inter / (tarea + area2 - inter) : 0.f;
                float decay;
                if (p.gaussian) decay = expf(-(iou * iou) / p.sigma);
                else decay = iou > p.soft_iou_thr ? 1.0f - iou : 1.0f;
                const float ns = s * decay;
                sc[i] = ((ns > p.score_thr) && (i != ti)) ? ns : -1.f;
            }
        }
        __syncthreads();
    }
    if (tid == 0) p.det_count[b] = count;
}

__global__ void gather_ood_kernel(const int* keep_src, const long long* indices, const float* energy,
                                  const float* maxlogit, long long n_anchors, int k, int max_det, int B,
                                  float* out_energy, float* out_maxlogit, long long* out_anchor) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * max_det) return;
    const int b = i / max_det;
    const int s = keep_src[i];
    float e = 0.f, m = 0.f;
    long long a = -1;
    if (s >= 0) {
        a = indices[(long long)b * k + s];
        e = energy[(long long)b * n_anchors + a];
        m = maxlogit[(long long)b * n_anchors + a];
    }
    out_energy[i] = e; out_maxlogit[i] = m;
    if (out_anchor) out_anchor[i] = a;
}

// Sources that walk scalar elements (row maxima, gathered anchor rows) are latency bound per thread: give
// every thread only a few elements
inline int sparse_segments(long long L) {
    long long s = (L + 4095) / 4096;                  // 16 elements per thread
    if (s < 1) s = 1;
    if (s > 65535) s = 65535;
    return (int)s;
}

inline int topk_segments(int B, long long L) {
    long long s = 2048 / (B > 0 ? B : 1);
    const long long by_work = (L + 16383) / 16384;
    if (s > by_work) s = by_work;
    if (s < 1) s = 1;
    return (int)s;
}

}  // namespace

namespace {

// Image-level OOD score from the per-anchor energies: max_a(-energy_a); one workgroup per image
__global__ __launch_bounds__(256) void ood_image_score_kernel(const float* energy, long long N, float* out) {
    const float* row = energy + (long long)blockIdx.x * N;
    float m = -INFINITY;
    for (long long i = threadIdx.x; i < N; i += 256) m = fmaxf(m, -row[i]);
    m = wave_reduce_max(m);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
}

// AUROC by exhaustive pair counting (exact, ties = 1/2): counts[0] += #{pos > neg}, counts[1] += #{pos == neg}
__global__ __launch_bounds__(256) void auroc_count_kernel(const float* pos, const float* neg, int np, int nn,
                                                          unsigned long long* counts) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    unsigned long long gt = 0, eq = 0;
    if (i < np) {
        const float p = pos[i];
        for (int j = 0; j < nn; ++j) { const float q = neg[j]; gt += p > q; eq += p == q; }
    }
    gt = wave_reduce_sum_u64(gt); eq = wave_reduce_sum_u64(eq);
    if ((threadIdx.x & 63) == 0) { atomicAdd(&counts[0], gt); atomicAdd(&counts[1], eq); }     // integer adds: order independent
}

}  // namespace

// The fork's novelty ("OOD-ish") score (infer.py:425-427, 465-471, 607-616): with proj = F.normalize(ProjectionNet
// output, p=2) and the cluster prototypes the episode code picked (`max_idxs`),
//     soft_thresh = sigmoid(dot_mult * (conf + dot_add));  sim_avg_i = mean_j <proj_i, proj_proto_j>;  sim_max_i = max_j ...
//     score_i = soft_thresh_i * sim_avg_i  (sim_target 'avg')   |   soft_thresh_i * sim_max_i  (sim_target 'max', without the
//     target_clust factor of :467, which belongs to the loss).
// One workgroup normalises the m prototypes into LDS once, then each wave scores anchors: the embedding row is normalised in
// registers (F.normalize: x / max(||x||, 1e-12)) and dotted with every prototype.
namespace {
__global__ __launch_bounds__(256) void novelty_score_kernel(const float* embds, const float* confs, const long long* proto, int n, int d, int m,
                                                            float dot_mult, float dot_add, int use_max,
                                                            float* score, float* soft_thresh, float* sim) {
    extern __shared__ float pl_[];                     // [m][d] normalised prototypes
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ int bad_;                               // a prototype index outside [0, n): never dereferenced (stale max_idxs, an
    if (tid == 0) bad_ = 0;                            // index into another concatenation); every output row is poisoned with NaN
    __syncthreads();
    for (int j = wave; j < m; j += 4) {
        const long long pj = proto[j];
        const bool ok = pj >= 0 && pj < (long long)n;                 // wave-uniform
        const float* row = embds + (ok ? pj : 0) * (long long)d;
        if (!ok && lane == 0) bad_ = 1;
        float ss = 0.f;
        for (int c = lane; c < d; c += 64) { const float v = row[c]; ss += v * v; }
        ss = wave_reduce_sum(ss);
        const float inv = ok ? 1.0f / fmaxf(sqrtf(ss), 1e-12f) : 0.f;
        for (int c = lane; c < d; c += 64) pl_[j * d + c] = row[c] * inv;
    }
    __syncthreads();
    const float poison = bad_ ? __builtin_nanf("") : 0.f;
    for (long long i = (long long)blockIdx.x * 4 + wave; i < n; i += (long long)gridDim.x * 4) {
        const float* row = embds + i * d;
        float ss = 0.f;
        for (int c = lane; c < d; c += 64) { const float v = row[c]; ss += v * v; }
        ss = wave_reduce_sum(ss);
        const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
        float acc_sum = 0.f, acc_max = -INFINITY;
        for (int j = 0; j < m; ++j) {
            float dot = 0.f;
            for (int c = lane; c < d; c += 64) dot += (row[c] * inv) * pl_[j * d + c];
            dot = wave_reduce_sum(dot);
            acc_sum += dot; acc_max = fmaxf(acc_max, dot);
        }
        if (lane == 0) {
            const float st = 1.0f / (1.0f + expf(-(dot_mult * (confs[i] + dot_add))));
            const float sv = use_max ? acc_max : acc_sum / (float)m;
            soft_thresh[i] = st; sim[i] = sv + poison; score[i] = st * sv + poison;
        }
    }
}
}  // namespace

extern "C" int effdet_novelty_score(void* stream, const float* embds, const float* confs, const long long* proto_idx, int n, int d, int m,
                                    float dot_mult, float dot_add, int use_max, float* score, float* soft_thresh, float* sim) {
    EFFDET_ENTER();
    if (!embds || !confs || !proto_idx || !score || !soft_thresh || !sim || n <= 0 || d <= 0 || m <= 0) return EFFDET_EINVAL;
    const size_t lds = (size_t)m * d * 4;
    if (lds > 64 * 1024) return EFFDET_EINVAL;                      // up to 64 prototypes of 256 dimensions
    int blocks = (n + 3) / 4; if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(novelty_score_kernel, dim3(blocks), dim3(256), lds, reinterpret_cast<hipStream_t>(stream),
                       embds, confs, proto_idx, n, d, m, dot_mult, dot_add, use_max, score, soft_thresh, sim);
    return effdet_check_launch();
}

extern "C" int effdet_ood_image_score(void* stream, const float* energy, int B, long long N, float* out) {
    EFFDET_ENTER();
    if (!energy || !out || B <= 0 || N <= 0) return EFFDET_EINVAL;
    hipLaunchKernelGGL(ood_image_score_kernel, dim3(B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), energy, N, out);
    return effdet_check_launch();
}

extern "C" int effdet_auroc_counts(void* stream, const float* pos, const float* neg, int n_pos, int n_neg,
                                   unsigned long long* counts) {
    EFFDET_ENTER();
    if (!pos || !neg || !counts || n_pos <= 0 || n_neg <= 0) return EFFDET_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (hipMemsetAsync(counts, 0, 16, st) != hipSuccess) return EFFDET_ELAUNCH;
    hipLaunchKernelGGL(auroc_count_kernel, dim3((n_pos + 255) / 256), dim3(256), 0, st, pos, neg, n_pos, n_neg, counts);
    return effdet_check_launch();
}

extern "C" long long effdet_topk_workspace_bytes(int B, long long n_anchors) {
    if (B <= 0 || n_anchors <= 0) return EFFDET_EINVAL;
    // 2 states | histogram | candidate keys | selected anchors | row maxima (when the caller has none)
    return (long long)B * (2 * sizeof(TopkState) + HIST_BINS * 4 + (long long)TOPK_CAP * 8 + n_anchors * 8);
}

namespace {

template <typename T>
int topk_run(hipStream_t st, const T* cls_all, const float* anchor_max, int B, long long n_anchors, int C,
             const T* box_all, int k, T* out_cls, T* out_box, long long* out_indices, long long* out_classes, char* ws) {
    const long long L = n_anchors * (long long)C;
    TopkState* state1 = reinterpret_cast<TopkState*>(ws);
    TopkState* state2 = state1 + B;
    char* q = ws + (size_t)B * 2 * sizeof(TopkState);
    unsigned int* hist = reinterpret_cast<unsigned int*>(q);               q += (size_t)B * HIST_BINS * 4;
    unsigned long long* cand = reinterpret_cast<unsigned long long*>(q);   q += (size_t)B * TOPK_CAP * 8;
    int* asel = reinterpret_cast<int*>(q);                                 q += (size_t)B * n_anchors * 4;
    float* rowmax = reinterpret_cast<float*>(q);
    const bool prefilter = n_anchors >= 4LL * k;
    // (the prefilter's first kernel initialises both states itself and needs no global histogram)
    if (!prefilter && hipMemsetAsync(ws, 0, (size_t)B * (2 * sizeof(TopkState) + HIST_BINS * 4), st) != hipSuccess) return EFFDET_ELAUNCH;
    // the sort buffer needs cand_total <= TOPK_CAP; stopping the radix passes at the next power of two >= k
    // keeps the bitonic network as small as the request allows
    unsigned int cap = 1024; while (cap < (unsigned int)k) cap <<= 1;
    if (prefilter) {
        if (!anchor_max) {
            hipLaunchKernelGGL(row_max_kernel<T>, dim3((unsigned int)(((long long)B * n_anchors + 3) / 4)), dim3(256), 0, st,
                               cls_all, (long long)B * n_anchors, C, rowmax);
            anchor_max = rowmax;
        }
        RowMaxSrc<sizeof(T) == 2> src1{anchor_max, n_anchors};
        hipLaunchKernelGGL(anchor_select_kernel<decltype(src1)>, dim3(B), dim3(1024), 0, st, src1, k, state1, state2, asel);
        // stage 2, fast path: every pair of the selected anchors that reaches the stage-1 threshold
        PairSrc<T> src2{cls_all, L, asel, state1, n_anchors, C};
        const int S2 = sparse_segments(2LL * k * C);
        hipLaunchKernelGGL(topk_collect_kernel<PairSrc<T>>, dim3(S2, B), dim3(256), 0, st, src2, (const TopkState*)state1, state2, cand);
    } else {
        DenseSrc<T> src{cls_all, L};
        const int S = topk_segments(B, L);
        for (int pass = 0; pass < NPASS; ++pass) {
            hipLaunchKernelGGL(topk_hist_kernel<DenseSrc<T>>, dim3(S, B), dim3(256), 0, st, src, pass, state2, hist);
            hipLaunchKernelGGL(topk_find_kernel, dim3(B), dim3(256), 0, st, pass, NPASS - 1, k, cap, state2, hist);
        }
        hipLaunchKernelGGL(topk_collect_kernel<DenseSrc<T>>, dim3(S, B), dim3(256), 0, st, src, (const TopkState*)state2, state2, cand);
    }
    const size_t sort_lds = (size_t)(TOPK_CAP + TOPK_CAP / 16) * 8;
    static bool attr_done = false;               // one per T
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(topk_sort_kernel<T, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sort_lds) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(topk_sort_kernel<T, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sort_lds) != hipSuccess)
            return EFFDET_ELAUNCH;
        attr_done = true;
    }
    PairSrc<T> fsrc{cls_all, L, asel, state1, n_anchors, C};
    if (prefilter)
        hipLaunchKernelGGL((topk_sort_kernel<T, true>), dim3(B), dim3(1024), sort_lds, st, state2, cand, k, C, cls_all, box_all, L, n_anchors,
                           out_cls, out_box, out_indices, out_classes, fsrc);
    else
        hipLaunchKernelGGL((topk_sort_kernel<T, false>), dim3(B), dim3(1024), sort_lds, st, state2, cand, k, C, cls_all, box_all, L, n_anchors,
                           out_cls, out_box, out_indices, out_classes, fsrc);
    return effdet_check_launch();
}

}  // namespace

extern "C" int effdet_topk_select(void* stream, int dtype, const void* cls_all, const float* anchor_max, int B,
                                  long long n_anchors, int C, const void* box_all, int k,
                                  void* out_cls, void* out_box, long long* out_indices, long long* out_classes,
                                  void* workspace, long long workspace_bytes) {
    EFFDET_ENTER();
    const long long L = n_anchors * (long long)C;
    if (!cls_all || !out_cls || !out_indices || !out_classes || !workspace || B <= 0 || C <= 0 || n_anchors <= 0) return EFFDET_EINVAL;
    if (k <= 0 || k > TOPK_CAP || k > L || L > 0x7fffffffLL || (dtype & ~1)) return EFFDET_EINVAL;
    if (box_all && !out_box) return EFFDET_EINVAL;
    if (workspace_bytes < effdet_topk_workspace_bytes(B, n_anchors)) return EFFDET_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    char* ws = reinterpret_cast<char*>(workspace);
    if (dtype == 0)
        return topk_run<float>(st, (const float*)cls_all, anchor_max, B, n_anchors, C, (const float*)box_all, k,
                               (float*)out_cls, (float*)out_box, out_indices, out_classes, ws);
    return topk_run<bf16_t>(st, (const bf16_t*)cls_all, anchor_max, B, n_anchors, C, (const bf16_t*)box_all, k,
                            (bf16_t*)out_cls, (bf16_t*)out_box, out_indices, out_classes, ws);
}

extern "C" int effdet_decode_threshold(void* stream, int dtype, const void* cls_topk, const void* box_topk,
                                       const float* anchors, const long long* indices, const long long* classes,
                                       const float* img_scale, const float* img_size, int B, int k,
                                       float* boxes, float* scores, int* classes_out, int* src, int* count, float* maxcoord) {
    EFFDET_ENTER();
    if (!cls_topk || !box_topk || !anchors || !indices || !classes || !boxes || !scores || !classes_out || !src || !count || !maxcoord) return EFFDET_EINVAL;
    if (B <= 0 || k <= 0 || (dtype & ~3)) return EFFDET_EINVAL;
    DecodeArgs a{cls_topk, box_topk, dtype & 1, (dtype & 2) ? 0 : (dtype & 1), anchors, indices, classes, img_scale, img_size, k, boxes, scores, classes_out, src, count, maxcoord, 0};
    hipLaunchKernelGGL(decode_threshold_kernel, dim3(B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    return effdet_check_launch();
}

// Same, with the box regressions read straight from the box head's [B, n_anchors, 4] output through `indices`: the
// top-k then does not depend on the box head at all and can run beside it on another stream.
extern "C" int effdet_decode_threshold_gather(void* stream, int dtype, const void* cls_topk, const void* box_all, long long n_anchors,
                                              const float* anchors, const long long* indices, const long long* classes,
                                              const float* img_scale, const float* img_size, int B, int k,
                                              float* boxes, float* scores, int* classes_out, int* src, int* count, float* maxcoord) {
    EFFDET_ENTER();
    if (!cls_topk || !box_all || !anchors || !indices || !classes || !boxes || !scores || !classes_out || !src || !count || !maxcoord) return EFFDET_EINVAL;
    if (B <= 0 || k <= 0 || n_anchors <= 0 || (dtype & ~3)) return EFFDET_EINVAL;
    DecodeArgs a{cls_topk, box_all, dtype & 1, (dtype & 2) ? 0 : (dtype & 1), anchors, indices, classes, img_scale, img_size, k, boxes, scores, classes_out, src, count, maxcoord, n_anchors};
    hipLaunchKernelGGL(decode_threshold_kernel, dim3(B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    return effdet_check_launch();
}

// decode + threshold (effdet/anchors.py:132-146) and hard NMS + top max_det (:147-166) of every image in one launch; arguments as
// effdet_decode_threshold[_gather] (n_anchors > 0: box rows through `indices`) followed by effdet_nms_hard's
extern "C" int effdet_detections_hard(void* stream, int dtype, const void* cls_topk, const void* box, long long n_anchors,
                                      const float* anchors, const long long* indices, const long long* classes,
                                      const float* img_scale_clip, const float* img_size, int B, int k,
                                      float* boxes, float* scores, int* classes_out, int* src, int* count, float* maxcoord,
                                      double iou_threshold, int max_det, const float* img_scale_out,
                                      float* det, int* det_count, int* keep_src) {
    EFFDET_ENTER();
    if (!cls_topk || !box || !anchors || !indices || !classes || !boxes || !scores || !classes_out || !src || !count || !maxcoord ||
        !det || !det_count || !keep_src) return EFFDET_EINVAL;
    if (B <= 0 || k <= 0 || n_anchors < 0 || (dtype & ~3) || max_det <= 0 || max_det > NMS_MAX_DET) return EFFDET_EINVAL;
    DecodeArgs d{cls_topk, box, dtype & 1, (dtype & 2) ? 0 : (dtype & 1), anchors, indices, classes, img_scale_clip, img_size, k, boxes, scores,
                 classes_out, src, count, maxcoord, n_anchors};
    NmsArgs n{boxes, scores, classes_out, src, count, maxcoord, k, iou_threshold, max_det, img_scale_out, det, det_count, keep_src, 1, 0.5f, 0.3f, 0.001f};
    hipLaunchKernelGGL(detections_hard_kernel, dim3(B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), d, n);
    return effdet_check_launch();
}

static int nms_common(NmsArgs& a, int B, bool soft, void* stream) {
    if (!a.boxes || !a.scores || !a.classes || !a.src || !a.count || !a.maxcoord || !a.det || !a.det_count || !a.keep_src) return EFFDET_EINVAL;
    if (B <= 0 || a.k <= 0 || a.max_det <= 0 || (!soft && a.max_det > NMS_MAX_DET)) return EFFDET_EINVAL;     // the hard kernel keeps its picks in LDS
    if (soft && (a.k > 1024 * SOFT_Q || a.max_det > a.k)) return EFFDET_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (soft) hipLaunchKernelGGL(nms_soft_kernel, dim3(B), dim3(1024), 0, st, a);
    else hipLaunchKernelGGL(nms_hard_kernel, dim3(B), dim3(256), 0, st, a);
    return effdet_check_launch();
}

extern "C" int effdet_nms_hard(void* stream, const float* boxes, const float* scores, const int* classes, const int* src,
                               const int* count, const float* maxcoord, int B, int k, double iou_threshold, int max_det,
                               const float* img_scale, float* det, int* det_count, int* keep_src) {
    EFFDET_ENTER();
    NmsArgs a{boxes, scores, classes, src, count, maxcoord, k, iou_threshold, max_det, img_scale, det, det_count, keep_src, 1, 0.5f, 0.3f, 0.001f};
    return nms_common(a, B, false, stream);
}

extern "C" int effdet_nms_soft(void* stream, const float* boxes, const float* scores, const int* classes, const int* src,
                               const int* count, const float* maxcoord, int B, int k,
                               int method_gaussian, float sigma, float iou_threshold, float score_threshold, int max_det,
                               const float* img_scale, float* det, int* det_count, int* keep_src) {
    EFFDET_ENTER();
    if (!(sigma > 0.f)) return EFFDET_EINVAL;
    NmsArgs a{boxes, scores, classes, src, count, maxcoord, k, (double)iou_threshold, max_det, img_scale, det, det_count, keep_src,
              method_gaussian ? 1 : 0, sigma, iou_threshold, score_threshold};
    return nms_common(a, B, true, stream);
}

extern "C" int effdet_nms_soft_large(void* stream, const float* boxes, const float* scores, const int* classes, const int* src,
                                     const int* count, const float* maxcoord, int B, int k,
                                     int method_gaussian, float sigma, float iou_threshold, float score_threshold, int max_det,
                                     const float* img_scale, float* det, int* det_count, int* keep_src, float* score_scratch) {
    EFFDET_ENTER();
    if (!(sigma > 0.f) || !score_scratch) return EFFDET_EINVAL;
    if (!boxes || !scores || !classes || !src || !count || !maxcoord || !det || !det_count || !keep_src) return EFFDET_EINVAL;
    if (B <= 0 || k <= 0 || max_det <= 0 || max_det > k) return EFFDET_EINVAL;
    NmsArgs a{boxes, scores, classes, src, count, maxcoord, k, (double)iou_threshold, max_det, img_scale, det, det_count, keep_src,
              method_gaussian ? 1 : 0, sigma, iou_threshold, score_threshold};
    hipLaunchKernelGGL(nms_soft_global_kernel, dim3(B), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), a, score_scratch);
    return effdet_check_launch();
}

extern "C" int effdet_gather_ood(void* stream, const int* keep_src, const long long* indices, const float* energy,
                                 const float* maxlogit, long long n_anchors, int B, int k, int max_det,
                                 float* out_energy, float* out_maxlogit, long long* out_anchor) {
    EFFDET_ENTER();
    if (!keep_src || !indices || !energy || !maxlogit || !out_energy || !out_maxlogit || B <= 0 || k <= 0 || max_det <= 0) return EFFDET_EINVAL;
    const int total = B * max_det;
    hipLaunchKernelGGL(gather_ood_kernel, dim3((total + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       keep_src, indices, energy, maxlogit, n_anchors, k, max_det, B, out_energy, out_maxlogit, out_anchor);
    return effdet_check_launch();
}

// ProjectionNet.weighted_median (effdet/efficientdet.py:748-760): per column of embds [n][d], sort ascending carrying the
// anchor confidences, cumulative sum, first position where it reaches half the total -> that value.  One workgroup
// per column, n <= 1024; equal values keep their original order (stable), as the reference's sort does on the CPU.
namespace {
__global__ __launch_bounds__(256) void weighted_median_kernel(const float* embds, const float* confs, int n, int d,
                                                              float* med, float* conf_sum) {
    __shared__ unsigned long long key[1024];           // ordered float key << 32 | row
    __shared__ float cs[1024];
    const int col = blockIdx.x, tid = threadIdx.x;
    int P = 1; while (P < n) P <<= 1;
    for (int i = tid; i < P; i += 256)
        key[i] = i < n ? (((unsigned long long)float_key(embds[(long long)i * d + col]) << 32) | (unsigned)i) : ~0ull;
    __syncthreads();
    for (int size = 2; size <= P; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < (P >> 1); i += 256) {
                const int pos = 2 * i - (i & (stride - 1));
                const unsigned long long a = key[pos], c = key[pos + stride];
                const bool asc = (pos & size) == 0;
                if ((a > c) == asc) { key[pos] = c; key[pos + stride] = a; }
            }
            __syncthreads();
        }
    for (int i = tid; i < n; i += 256) cs[i] = confs[(unsigned)(key[i] & 0xFFFFFFFFull)];
    __syncthreads();
    if (tid == 0) {                                     // n <= 1024: a serial scan in the reference's summation order
        float total = 0.f;
        for (int i = 0; i < n; ++i) total += confs[i];
        float run = 0.f; int idx = 0; bool found = false;
        for (int i = 0; i < n; ++i) { run += cs[i]; if (!found && run >= total / 2) { idx = i; found = true; } }
        med[col] = key_float((unsigned)(key[idx] >> 32));
        if (col == 0) conf_sum[0] = total;
    }
}
}  // namespace

extern "C" int effdet_weighted_median(void* stream, const float* embds, const float* confs, int n, int d, float* med, float* conf_sum) {
    EFFDET_ENTER();
    if (!embds || !confs || !med || !conf_sum || n <= 0 || n > 1024 || d <= 0) return EFFDET_EINVAL;
    hipLaunchKernelGGL(weighted_median_kernel, dim3(d), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), embds, confs, n, d, med, conf_sum);
    return effdet_check_launch();
}
