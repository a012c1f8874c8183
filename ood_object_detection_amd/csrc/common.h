// Shared device helpers for the gfx950 (MI355X / CDNA4) EfficientDet kernels.
// Wave = 64 lanes. Activations are NHWC; T is float (parity mode) or __bf16 (throughput mode).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/effdet_hip.h"

#define EFFDET_OK 0
#define EFFDET_EINVAL (-22)
#define EFFDET_ELAUNCH (-5)

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define DEV __device__ __forceinline__

// Other HIP users of the process (e.g. the host framework probing a pointer) can leave a stale,
// non-fatal error code behind; drop it on entry so effdet_check_launch() reports only our launch.
#define EFFDET_ENTER() (void)hipGetLastError()

extern thread_local int effdet_last_hip_error;      // defined in abi.hip
#define EFFDET_DEVICE_ERROR_CODE 0x7EFFDE7          /* effdet_last_hip_error value: the device-side failure word is set */
__attribute__((visibility("hidden"))) extern volatile int* effdet_err_host;     // abi.hip: host view of the device-side failure word
__attribute__((visibility("hidden"))) int* effdet_device_error_word();         // abi.hip: its device pointer (null: unavailable)
static inline int effdet_check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { effdet_last_hip_error = (int)e; return EFFDET_ELAUNCH; }
    // a kernel of an EARLIER call may have flagged a device-side failure (see abi.hip): reported here, sticky until cleared
    if (effdet_err_host && *effdet_err_host) { effdet_last_hip_error = EFFDET_DEVICE_ERROR_CODE; return EFFDET_ELAUNCH; }
    return EFFDET_OK;
}

// ---------------------------------------------------------------------------------------------
// scalar math (fp32 everywhere inside a kernel; T only at the HBM / LDS-image boundary)
// ---------------------------------------------------------------------------------------------
DEV float silu_f(float x) { return x / (1.0f + expf(-x)); }
DEV float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }
// Hardware-transcendental forms (v_exp_f32 / v_rcp_f32, ~1 ulp each): used in bf16 throughput mode,
// where the result is rounded to 8 bits anyway; float32 parity mode keeps the correctly rounded forms.
DEV float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
DEV float fast_silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + fast_exp(-x)); }
// float32 TRAINING kernels (train_*.hip, the training form of dwconv.hip): SiLU / sigmoid on the hardware exp2 / rcp instructions
// (~1 ulp each; relative error of the result < 1e-6, against the 1e-5 ... 2e-3 bounds of the gradient parity tests).  The
// correctly rounded expf + IEEE division of silu_f cost ~10x the instructions, and the step applies ~2 G of them.  The float32
// INFERENCE parity path keeps silu_f.  -DEFFDET_TRAIN_PRECISE_SILU restores the precise forms here too.
#ifdef EFFDET_TRAIN_PRECISE_SILU
DEV float sigmoid_train(float x) { return sigmoid_f(x); }
DEV float silu_train(float x) { return silu_f(x); }
#else
DEV float sigmoid_train(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f)); }
DEV float silu_train(float x) { return x * sigmoid_train(x); }
#endif

struct bf16p_t;                                     // the two-term bf16 dtype (below): hardware transcendentals like bf16
template <typename T> struct FastMath { static constexpr bool value = sizeof(T) == 2; };
template <> struct FastMath<bf16p_t> { static constexpr bool value = true; };
template <typename T> DEV float silu_t(float x) {
    if constexpr (FastMath<T>::value) return fast_silu(x); else return silu_f(x);
}
template <typename T> DEV float exp_t(float x) {
    if constexpr (FastMath<T>::value) return fast_exp(x); else return expf(x);
}

// Folded BN + SiLU of the four accumulator values of a lane.  In bf16 throughput mode the non-transcendental part runs on
// packed fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two values per lane and instruction); the
// arithmetic - and therefore every bit of the result - is the same as four fast_silu(acc * sc + sh) calls.
typedef float f32x2 __attribute__((ext_vector_type(2)));
// The expand SiLU of the bf16 rolling-window MBConv kernels (mbconv_roll.hip, mbconv_wide.hip) on t = -log2(e) x (the scale sits in W1 and BN1's shift; its inverse, -ln 2, in the depthwise
// taps): t * rcp(2^t + addc) = -log2(e) * silu(x) for addc = 1, and 0 for addc = +inf (pixels outside the image) - one packed
// multiply and the separate mask multiplies less than silu(x) * mask (5 instead of 8.5 vector instructions per value pair)
DEV f32x4 silu4_scaled(const f32x4 t, const float addc) {
    const f32x2 t0 = {t[0], t[1]}, t1 = {t[2], t[3]};
    // (scalar adds: a packed add wants the addend duplicated into a register pair per tile, which spilled in the 5 x 5 kernels)
    const float d00 = __builtin_amdgcn_exp2f(t0[0]) + addc, d01 = __builtin_amdgcn_exp2f(t0[1]) + addc;
    const float d10 = __builtin_amdgcn_exp2f(t1[0]) + addc, d11 = __builtin_amdgcn_exp2f(t1[1]) + addc;
    const f32x2 y0 = t0 * f32x2{__builtin_amdgcn_rcpf(d00), __builtin_amdgcn_rcpf(d01)};
    const f32x2 y1 = t1 * f32x2{__builtin_amdgcn_rcpf(d10), __builtin_amdgcn_rcpf(d11)};
    return f32x4{y0[0], y0[1], y1[0], y1[1]};
}

template <typename T> DEV f32x4 bn_silu4(const f32x4 acc, const f32x4 sc, const f32x4 sh) {
    if constexpr (FastMath<T>::value) {
        const f32x2 x0 = f32x2{acc[0], acc[1]} * f32x2{sc[0], sc[1]} + f32x2{sh[0], sh[1]};
        const f32x2 x1 = f32x2{acc[2], acc[3]} * f32x2{sc[2], sc[3]} + f32x2{sh[2], sh[3]};
        const f32x2 t0 = x0 * -1.4426950408889634f, t1 = x1 * -1.4426950408889634f;
        const f32x2 d0 = f32x2{__builtin_amdgcn_exp2f(t0[0]), __builtin_amdgcn_exp2f(t0[1])} + 1.0f;
        const f32x2 d1 = f32x2{__builtin_amdgcn_exp2f(t1[0]), __builtin_amdgcn_exp2f(t1[1])} + 1.0f;
        const f32x2 y0 = x0 * f32x2{__builtin_amdgcn_rcpf(d0[0]), __builtin_amdgcn_rcpf(d0[1])};
        const f32x2 y1 = x1 * f32x2{__builtin_amdgcn_rcpf(d1[0]), __builtin_amdgcn_rcpf(d1[1])};
        return f32x4{y0[0], y0[1], y1[0], y1[1]};
    } else {
        f32x4 y;
#pragma unroll
        for (int r = 0; r < 4; ++r) y[r] = silu_f(acc[r] * sc[r] + sh[r]);
        return y;
    }
}

template <typename T> struct VecTraits;
// 16-byte chunk = 4 floats or 8 bf16
template <> struct VecTraits<float> { static constexpr int EPC = 4; };    // elements per 16-B chunk
template <> struct VecTraits<bf16_t> { static constexpr int EPC = 8; };

// A group of 8 consecutive channels held as fp32 in registers.
struct F8 { float v[8]; };

template <typename T> DEV F8 load8(const T* p);
template <> DEV F8 load8<float>(const float* p) {
    F8 r;
    const f32x4 a = *reinterpret_cast<const f32x4*>(p);
    const f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { r.v[i] = a[i]; r.v[4 + i] = b[i]; }
    return r;
}
template <> DEV F8 load8<bf16_t>(const bf16_t* p) {
    F8 r;
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = (float)a[i];
    return r;
}
template <typename T> DEV void store8(T* p, const F8& r);
template <> DEV void store8<float>(float* p, const F8& r) {
    f32x4 a, b;
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = r.v[i]; b[i] = r.v[4 + i]; }
    *reinterpret_cast<f32x4*>(p) = a;
    *reinterpret_cast<f32x4*>(p + 4) = b;
}
template <> DEV void store8<bf16_t>(bf16_t* p, const F8& r) {
    bf16x8 a;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (bf16_t)r.v[i];
    *reinterpret_cast<bf16x8*>(p) = a;
}
// 4 consecutive elements (8 bytes of bf16 / 16 bytes of f32): the accumulator piece of one lane after an MFMA
template <typename T>
DEV void store4(T* p, float a, float b, float c, float d) {
    if constexpr (sizeof(T) == 2) {
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        bf16x4 v = {(bf16_t)a, (bf16_t)b, (bf16_t)c, (bf16_t)d};
        *reinterpret_cast<bf16x4*>(p) = v;
    } else {
        *reinterpret_cast<f32x4*>(p) = f32x4{a, b, c, d};
    }
}

DEV F8 f8_zero() { F8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = 0.f;
    return r; }
DEV F8 f8_fill(float x) { F8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = x;
    return r; }

// Division of a small non-negative int by a launch-time constant: one v_mul_hi instead of the ~25-instruction
// emulated divide.  m = 2^32 / d + 1 (host side); exact while i * d < 2^32 (the uses index LDS tiles).
struct FastDiv { unsigned m; int d; };
inline FastDiv make_fastdiv(int d) {
    FastDiv f; f.d = d; f.m = d <= 1 ? 0u : (unsigned)(0x100000000ull / (unsigned)d) + 1u;
    return f;
}
DEV int fdiv(int i, FastDiv f) { return f.d <= 1 ? i : (int)__umulhi((unsigned)i, f.m); }
DEV int fmod_(int i, int q, FastDiv f) { return i - q * f.d; }

template <typename T> DEV float to_f(T x) { return (float)x; }
template <typename T> DEV T from_f(float x) { return (T)x; }

// ---------------------------------------------------------------------------------------------
// MFMA on a "64-byte K-chunk": every lane holds one 16-byte piece of A and of B.
//   bf16: 8 elements/lane  -> one v_mfma_f32_16x16x32_bf16  (K = 32)
//   f32 : 4 elements/lane  -> four v_mfma_f32_16x16x4_f32   (K = 16), exact fp32 fma chain
// Lane l supplies row/col (l & 15) and 16-byte piece (l >> 4) of the chunk for both operands, so
// the same LDS image (rows of 64-byte K-chunks) serves both dtypes.
// C/D: col = l & 15, row = 4 * (l >> 4) + reg.
// ---------------------------------------------------------------------------------------------
template <typename T> struct Frag;
template <> struct Frag<bf16_t> { bf16x8 v; };
template <> struct Frag<float> { f32x4 v; };

template <typename T> DEV Frag<T> ld_frag(const void* p) {
    Frag<T> f;
    f.v = *reinterpret_cast<const decltype(f.v)*>(p);
    return f;
}
DEV void mma_chunk(const Frag<bf16_t>& a, const Frag<bf16_t>& b, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, acc, 0, 0, 0);
}
DEV void mma_chunk(const Frag<float>& a, const Frag<float>& b, f32x4& acc) {
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[s], b.v[s], acc, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------
// dtype 2 (EFFDET_BF16X2, "accurate" mode): every value is the unevaluated sum of TWO bfloat16 numbers,
//     x ~ hi + lo,   hi = bf16(x),   lo = bf16(x - hi)           (16 significand bits, relative error <= 2^-17)
// and a product is three matrix-core instructions at the bf16 rate (hi*hi + hi*lo + lo*hi; the lo*lo term is below the
// representation error) instead of the sixteen-times slower float32 MFMA.  Storage: 8 consecutive channels are 32 bytes,
// [8 x bf16 hi][8 x bf16 lo] - so a tensor has the byte size and the channel pitch (4 bytes) of the float32 tensor of the same
// shape, and a lane's MFMA operand piece (8 K values) is two adjacent 16-byte reads.  Channel counts are multiples of 8; only
// whole groups (or their 4-channel halves) are addressable.  The splitting is done ONCE per value by the kernel that produces it.
// ---------------------------------------------------------------------------------------------
struct bf16p_t { unsigned raw; };                 // 4 bytes per channel: pointer arithmetic in channels works on group boundaries
template <> struct VecTraits<bf16p_t> { static constexpr int EPC = 8; };
template <> struct Frag<bf16p_t> { bf16x8 h, l; };
template <> DEV Frag<bf16p_t> ld_frag<bf16p_t>(const void* p) {
    Frag<bf16p_t> f;
    f.h = *reinterpret_cast<const bf16x8*>(p);
    f.l = *(reinterpret_cast<const bf16x8*>(p) + 1);
    return f;
}
DEV void mma_chunk(const Frag<bf16p_t>& a, const Frag<bf16p_t>& b, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.l, b.h, acc, 0, 0, 0);        // small terms first
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.l, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.h, acc, 0, 0, 0);
}
// Geometry of an MFMA operand image per dtype: bytes of one lane's piece and of one K-chunk (4 pieces), elements per chunk
template <typename T> struct OpGeom { static constexpr int PIECE = 16, CHUNK = 64, KPC = 64 / (int)sizeof(T); };
template <> struct OpGeom<bf16p_t> { static constexpr int PIECE = 32, CHUNK = 128, KPC = 32; };
template <typename T> struct IsPair { static constexpr bool value = false; };
template <> struct IsPair<bf16p_t> { static constexpr bool value = true; };
// kernels choose their bf16-style code paths (hardware transcendentals, depthwise on the matrix cores) by this, not by sizeof
template <typename T> struct IsFast { static constexpr bool value = sizeof(T) == 2; };
template <> struct IsFast<bf16p_t> { static constexpr bool value = true; };

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
// four values -> 8 bytes of hi + 8 bytes of lo
DEV void pair_split4(const f32x4 x, u32x2& hi, u32x2& lo) {
    const bf16x4 h = {(bf16_t)x[0], (bf16_t)x[1], (bf16_t)x[2], (bf16_t)x[3]};
    const bf16x4 l = {(bf16_t)(x[0] - (float)h[0]), (bf16_t)(x[1] - (float)h[1]), (bf16_t)(x[2] - (float)h[2]), (bf16_t)(x[3] - (float)h[3])};
    hi = __builtin_bit_cast(u32x2, h); lo = __builtin_bit_cast(u32x2, l);
}
DEV void pair_split8(const F8& x, u32x4& hi, u32x4& lo) {
    bf16x8 h, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) { h[e] = (bf16_t)x.v[e]; l[e] = (bf16_t)(x.v[e] - (float)h[e]); }
    hi = __builtin_bit_cast(u32x4, h); lo = __builtin_bit_cast(u32x4, l);
}
DEV F8 pair_join8(const u32x4 hi, const u32x4 lo) {
    const bf16x8 h = __builtin_bit_cast(bf16x8, hi), l = __builtin_bit_cast(bf16x8, lo);
    F8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r.v[e] = (float)h[e] + (float)l[e];
    return r;
}
template <> DEV F8 load8<bf16p_t>(const bf16p_t* p) {                      // p: a group boundary (channel % 8 == 0)
    return pair_join8(*reinterpret_cast<const u32x4*>(p), *(reinterpret_cast<const u32x4*>(p) + 1));
}
template <> DEV void store8<bf16p_t>(bf16p_t* p, const F8& r) {
    u32x4 hi, lo;
    pair_split8(r, hi, lo);
    *reinterpret_cast<u32x4*>(p) = hi;
    *(reinterpret_cast<u32x4*>(p) + 1) = lo;
}
// 4 consecutive channels [ch, ch + 4) (ch % 4 == 0) of the row that starts at `row` (any 8-byte aligned address)
template <typename T> DEV void row_store4(void* row, int ch, const f32x4 v) {
    if constexpr (IsPair<T>::value) {
        u32x2 hi, lo;
        pair_split4(v, hi, lo);
        char* g = reinterpret_cast<char*>(row) + (ch >> 3) * 32 + (ch & 7) * 2;
        *reinterpret_cast<u32x2*>(g) = hi;
        *reinterpret_cast<u32x2*>(g + 16) = lo;
    } else {
        store4<T>(reinterpret_cast<T*>(row) + ch, v[0], v[1], v[2], v[3]);
    }
}

// mbconv_roll.hip (internal, hidden from the C ABI): rolling-window form of the fused MBConv front half, bf16 only.
// parts = SE pool partial rows per image when the form applies to the geometry, 0 otherwise.
// pair != 0: dtype 2 (two-term bf16) instead of bfloat16
__attribute__((visibility("hidden"))) int effdet_mbconv_roll_parts(int H, int W, int Cin, int mid, int k, int stride, int pair = 0);
__attribute__((visibility("hidden"))) int effdet_mbconv_roll_launch(hipStream_t st, const void* X, const float* in_gate, void* Y, const void* W1, const float* s1, const float* t1,
                              const float* taps, const float* s2, const float* t2, float* pool_partial,
                              int B, int H, int W, int Cin, int mid, int k, int stride, int pair = 0, int sym = 0);

// mbconv_wide.hip (internal): rolling-window form for inputs wider than 64 channels (X rows shared by a workgroup through an
// LDS ring), bf16 only; parts = SE pool partial rows per image when the form applies to the geometry, 0 otherwise
__attribute__((visibility("hidden"))) int effdet_mbconv_wide_parts(int H, int W, int Cin, int mid, int k, int stride, int pair = 0);
__attribute__((visibility("hidden"))) int effdet_mbconv_wide_launch(hipStream_t st, const void* X, void* Y, const void* W1, const float* s1, const float* t1,
                              const float* taps, const float* s2, const float* t2, float* pool_partial,
                              int B, int H, int W, int Cin, int mid, int k, int stride, int pair = 0, int sym = 0);

// stem_roll.hip (internal): rolling-window form of the fused stem + stage-0 depthwise, bf16 only; parts = SE pool partial rows
// per image when the form applies, 0 otherwise
__attribute__((visibility("hidden"))) int effdet_stem_roll_parts(int H, int W, int C, int pair = 0, int sym = 0);
__attribute__((visibility("hidden"))) int effdet_stem_roll_launch(hipStream_t st, int in_dtype, const void* X, const float* mean, const float* stdv,
                            const void* Wk, const float* s1, const float* t1, const float* taps, const float* s2, const float* t2,
                            void* Y, float* pool_partial, int B, int H, int W, int C, int pair = 0, int sym = 0);

// train_net.hip (internal): out[g][l] (+)= alpha * sum_s in[g][s][l], summed in a fixed order
__attribute__((visibility("hidden"))) int effdet_launch_reduce_mid(hipStream_t st, const float* in, int G, int S, long long L, float* out,
                                                                   int accumulate, float alpha);
// the same; the leading [trT][trC] matrix of every row is written transposed ([trC][trT]), what follows it keeps its place
__attribute__((visibility("hidden"))) int effdet_launch_reduce_mid_tr(hipStream_t st, const float* in, int G, int S, long long L, float* out,
                                                                      int accumulate, float alpha, int trC, int trT);

// TF "SAME" padding: amount in front (reference semantics live in timm)
static inline int same_pad_before(int size, int k, int s) {
    int out = (size + s - 1) / s;
    int total = (out - 1) * s + k - size;
    if (total < 0) total = 0;
    return total / 2;
}
static inline int same_out(int size, int s) { return (size + s - 1) / s; }
// timm's two padding conventions (create_conv2d / create_pool2d `padding=`; effdet config.pad_type): 'same' = TF-SAME above; '' =
// static symmetric padding ((s - 1) + (k - 1)) / 2 on every side.  For the odd kernels and strides 1 | 2 of this network both give
// ceil(size / s) outputs; they differ in where the window starts when stride 2 meets an even size (one pixel earlier for '').
// Callers select '' by OR-ing EFFDET_PAD_SYMMETRIC into the entry point's dtype (or first selector) argument.
static inline int pad_before(int size, int k, int s, int symmetric) {
    return symmetric ? ((s - 1) + (k - 1)) / 2 : same_pad_before(size, k, s);
}
static inline int take_pad_flag(int& v) { const int f = (v & EFFDET_PAD_SYMMETRIC) ? 1 : 0; v &= ~EFFDET_PAD_SYMMETRIC; return f; }

DEV float wave_reduce_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// Per-channel sum of per-thread 8-channel partials.  Thread t (< nact) holds the partial sums of channel group
// (t % cgn); all NT threads must call this.  red: [NT][8] floats, red2: [NT] floats (LDS).  Two levels, so that
// no thread walks more than nact/(cgn*P) + P values; fixed order -> bitwise reproducible.  Returns, for
// tid < cgn*8, the total of channel `tid`.
template <int NT>
DEV float pool_reduce(const F8& pool, float* red, float* red2, int tid, int cgn, int nact) {
    store8<float>(red + tid * 8, pool);
    __syncthreads();
    const int C_ = cgn * 8;
    const int P = NT / C_ > 0 ? NT / C_ : 1;
    if (tid < P * C_) {
        const int part = tid / C_, c = tid % C_;
        const int g = c >> 3, q = c & 7;
        float s = 0.f;
        for (int t = g + cgn * part; t < nact; t += cgn * P) s += red[t * 8 + q];
        red2[part * C_ + c] = s;
    }
    __syncthreads();
    float tot = 0.f;
    if (tid < C_) {
        for (int part = 0; part < P; ++part) tot += red2[part * C_ + tid];
    }
    return tot;
}

DEV unsigned long long wave_reduce_sum_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)(v & 0xFFFFFFFFull), o, 64), hi = __shfl_xor((unsigned)(v >> 32), o, 64);
        v += ((unsigned long long)hi << 32) | lo;
    }
    return v;
}
DEV float wave_reduce_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
