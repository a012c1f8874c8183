// Rolling-window form of the fused MBConv front half for WIDE inputs (more than 64 input channels; bfloat16 throughput mode):
//
//   expand 1x1 conv (MFMA) -> BN1 -> SiLU -> depthwise k x k (stride 1|2, TF-SAME) -> BN2 -> SiLU (+ SE pool partial sums)
//
// replaces timm's InvertedResidual.conv_pw/bn1/act1/conv_dw/bn2/act2 (reached from effdet/efficientdet.py:837) for the late
// backbone stages (d0: blocks 3.1 ... 6.0, 40 x 40 and 20 x 20 maps with 80 ... 192 input channels), which mbconv_roll.hip
// cannot take: there every wave pulls its own copy of the X rows through its registers, and with Cin / 32 operand fragments
// per 16-pixel tile nothing is left to prefetch with.  Here the X rows are SHARED: a workgroup is a bundle of `nw` waves that
// own consecutive 16-channel tiles of the same (image, band of rows, column strip); together they stage every X row ONCE into
// an LDS ring (each lane moves one or two 16-byte pieces per row: global -> registers two row-steps ahead, registers -> LDS one
// row-step ahead), and every wave reads its MFMA B operands from there.  Everything else is the rolling-window scheme of
// mbconv_roll.hip: the wave's W1 rows (A operand, BN1 scale folded in) stay in registers for the whole band, the last KS
// expanded rows of its 16 channels live in a wave-private LDS ring, the depthwise taps run on the matrix cores against diagonal
// weight operands, no expanded row is computed twice inside a band.
//
// There is NO workgroup barrier in the row loop.  The X ring is handed over through per-slot arrival counters in LDS:
//   commit(r): the wave's pieces of row r are written (ds_write), then one lane adds 1 to cnt[r % NSX];
//   poll(r):   before its first read of row r a wave spins (s_sleep) until cnt[r % NSX] has reached nw * (r / NSX + 1).
// Per row q every wave runs  poll(q); expand(q); commit(q + S); issue-loads(q + 2S)  (S = stride = new rows per output row).
// commit(q + S) overwrites the slot of row q + S - NSX = q - S (NSX = 2 S slots): every wave has finished reading that row,
// because this wave passed poll(q), i.e. every wave has executed its commit(q), which follows its expand(q - S) in program
// order (LDS operations of one wave execute in order).  So waves may drift apart by up to one row-step, and a slow wave only
// ever delays the others at their next poll.  All waves of a workgroup run the same number of row steps and are co-resident
// by construction (same workgroup), so every spin terminates; it is bounded anyway (a broken hand-off then sets the library's device-side
// failure word - the next effdet_* call returns -5 - instead of hanging the GPU).
//
// X loads and Y stores are buffer operations with hardware range checking (out-of-image pixels: offset beyond num_records,
// the load returns zeros and the store is dropped), as in mbconv_roll.hip, so the row loop has no exec-mask branches around
// vector-memory operations and `s_waitcnt vmcnt(N)` stays counted.
#include "common.h"
#ifdef WIDE_TUNE
#include <cstdio>
#include <cstdlib>
#endif

namespace {

struct WideArgs {
    const void* X; void* Y; const void* W1;
    const float* s1; const float* t1; const float* taps; const float* s2; const float* t2;
    float* pool_partial;
    int B, H, W, Cin, mid, Ho, Wo, pad_t, pad_l;
    int TWo, nstrips, band_rows, nbands, IWs, nw, ngroups, ring_bytes, per_image;
    int xpitch, xslot_bytes, pieces_row, ppr, x_off, ring_off, lds_bytes;
    FastDiv fd_ppr;
    int* err_word;                                 // device-side failure word (abi.hip) or null
};

typedef float f32x2w __attribute__((ext_vector_type(2)));
DEV f32x4 silu4_w(const f32x4 x) {
    const f32x2w x0 = {x[0], x[1]}, x1 = {x[2], x[3]};
    const f32x2w t0 = x0 * -1.4426950408889634f, t1 = x1 * -1.4426950408889634f;
    const f32x2w d0 = f32x2w{__builtin_amdgcn_exp2f(t0[0]), __builtin_amdgcn_exp2f(t0[1])} + 1.0f;
    const f32x2w d1 = f32x2w{__builtin_amdgcn_exp2f(t1[0]), __builtin_amdgcn_exp2f(t1[1])} + 1.0f;
    const f32x2w y0 = x0 * f32x2w{__builtin_amdgcn_rcpf(d0[0]), __builtin_amdgcn_rcpf(d0[1])};
    const f32x2w y1 = x1 * f32x2w{__builtin_amdgcn_rcpf(d1[0]), __builtin_amdgcn_rcpf(d1[1])};
    return f32x4{y0[0], y0[1], y1[0], y1[1]};
}

template <int V> struct IntW { static constexpr int value = V; };

// Phase ablation for tools/wide_ablate_gpu.sh: exists only in variant builds (`make variant TAG=.. VDEFS=-DWIDE_ABLATE=n`,
// libeffdet_hip_<TAG>.so, never loaded by the package); the product library is compiled with WIDE_ABLATE = 0.
// 1: no hand-off (no poll, no arrival count)  2: no depthwise arithmetic  4: no expand arithmetic  8: no X staging  16: SiLU -> identity
#ifndef WIDE_ABLATE
#define WIDE_ABLATE 0
#endif
// polls of an arrival counter before a wave gives up (and flags the failure); -DWIDE_SPIN_LIMIT=0 exists as a variant build for the
// test of that report only (tests/test_kernels_gpu.py::test_wide_handoff_timeout_is_reported)
#ifndef WIDE_SPIN_LIMIT
#define WIDE_SPIN_LIMIT (1 << 22)
#endif
DEV f32x4 act4_w(const f32x4 x) {
    if constexpr ((WIDE_ABLATE & 16) != 0) return x; else return silu4_w(x);
}

constexpr int WIDE_HDR = 64 + 1024;                // LDS header: arrival counters (64 B) + a 1 KiB dump row for idle staging lanes

// KS taps per side, S stride, NKC = 64-byte K chunks of Cin, MT input tiles (16 px) per strip row, NO output tiles per strip
// row, NPL = 16-byte X pieces a lane stages per row.  Register budget: 128 (up to 16 waves per CU)
// T: bf16_t, or bf16p_t (dtype 2, two-term bf16: X, W1, the X ring and Y carry hi + lo and the expand GEMM takes three MFMAs per
// chunk; NKC then counts 128-byte chunks - still 32 channels each; register budget 256: workgroups of at most 8 waves, 8 waves per
// CU.  The expanded ring holds FLOAT32 there and the depthwise taps run on the vector ALU, see mbconv_roll.hip)
template <int KS, int S, int NKC, int MT, int NO, int NPL, typename T>
__global__ __launch_bounds__((IsPair<T>::value ? 512 : 1024), (IsPair<T>::value ? 2 : 4)) void mbconv_wide_kernel(WideArgs p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr bool PAIR = IsPair<T>::value;
    constexpr int PB = OpGeom<T>::PIECE, CHB = OpGeom<T>::CHUNK;     // bytes of a lane's operand piece / of a 32-channel K-chunk
    constexpr int PXB = 16 * (int)sizeof(T);                         // bytes of a ring pixel (16 channels)
    constexpr int NSX = 2 * S;                                 // X ring slots (rows)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // provably wave-uniform: everything derived from it stays scalar
    const int frow = lane & 15, kg = lane >> 4;
    // blocks are dealt round-robin over the 8 XCDs: image = (round, xcd), so one image's workgroups share an L2
    const int xcd = blockIdx.x & 7, rr_ = blockIdx.x >> 3;
    const int b = (rr_ / p.per_image) * 8 + xcd;
    if (b >= p.B) return;
    int q_ = rr_ % p.per_image;
    const int group = q_ % p.ngroups; q_ /= p.ngroups;
    const int strip = q_ % p.nstrips, band = q_ / p.nstrips;
    const int c0 = 16 * (group * p.nw + wave);
    const int cbytes = p.Cin * (int)sizeof(T), mid = p.mid;
    int* const cnt = reinterpret_cast<int*>(lds);
    char* const xring = lds + p.x_off;
    char* const ring = lds + p.ring_off + wave * p.ring_bytes;
    constexpr int rowbytes = MT * 16 * PXB;                       // [MT * 16 px][16 ch]
    constexpr int NTAP = KS * KS, NPAIR = (NTAP + 1) / 2;

    // ---- the whole LDS allocation starts as zeros: ring padding, K tails and the pixels beyond a strip are read (against
    // zero weights / dropped lanes) and must be finite
    for (int i = threadIdx.x * 16; i < p.lds_bytes; i += blockDim.x * 16) *reinterpret_cast<u32x4*>(lds + i) = u32x4{0u, 0u, 0u, 0u};

    // ---- per-wave constants
    Frag<T> wf[NKC];
    const float rs1 = p.s1[c0 + frow], rs2 = p.s2[c0 + frow];
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) {
        const int off = kc * CHB + kg * PB;
        wf[kc] = ld_frag<T>(reinterpret_cast<const char*>(p.W1) + (long long)(c0 + frow) * cbytes + (off < cbytes ? off : 0));
    }
    const int hi = kg >> 1;
    const bool dactive = (kg & 1) == (frow >> 3);
    const int dq = (frow & 7) >> 1;
    float tapv[NPAIR];
#pragma unroll
    for (int pr = 0; pr < NPAIR; ++pr) {
        const int t = 2 * pr + hi;
        tapv[pr] = p.taps[(long long)(t < NTAP ? t : 0) * mid + c0 + frow];
    }
    // the expand GEMM produces t = -log2(e) x directly (silu4_scaled, common.h), the taps carry -ln 2
    constexpr float ESC = -1.4426950408889634f, EINV = -0.6931471805599453f;
    const f32x4 sh1 = *reinterpret_cast<const f32x4*>(p.t1 + c0 + 4 * kg) * ESC;
    const f32x4 t2v = *reinterpret_cast<const f32x4*>(p.t2 + c0 + 4 * kg);
    // two-term mode: the lane's 4 channels of every tap, BN2's scale folded in (float32 vector-ALU depthwise)
    f32x4 wv[PAIR ? NTAP : 1];
    if constexpr (PAIR) {
        const f32x4 s2q = *reinterpret_cast<const f32x4*>(p.s2 + c0 + 4 * kg);
#pragma unroll
        for (int t = 0; t < NTAP; ++t) wv[t] = *reinterpret_cast<const f32x4*>(p.taps + (long long)t * mid + c0 + 4 * kg) * (s2q * EINV);
    }
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) {
        const bool kv = kc * CHB + kg * PB < cbytes;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if constexpr (PAIR) {                                 // scale the VALUE in float32, then split again
                const float w = ((float)wf[kc].h[e] + (float)wf[kc].l[e]) * (rs1 * ESC);
                const bf16_t wh = (bf16_t)w;
                wf[kc].h[e] = kv ? wh : (bf16_t)0.f;
                wf[kc].l[e] = kv ? (bf16_t)(w - (float)wh) : (bf16_t)0.f;
            } else {
                wf[kc].v[e] = kv ? (bf16_t)((float)wf[kc].v[e] * (rs1 * ESC)) : (bf16_t)0.f;
            }
        }
    }
    unsigned abits[NPAIR], abitl[PAIR ? NPAIR : 1];
#pragma unroll
    for (int pr = 0; pr < NPAIR; ++pr) {
        const bool on = dactive && 2 * pr + hi < NTAP;
        const float tw_ = tapv[pr] * (rs2 * EINV);
        const bf16_t th_ = (bf16_t)tw_;
        abits[pr] = on ? (unsigned)__builtin_bit_cast(unsigned short, th_) << (16 * (frow & 1)) : 0u;
        if constexpr (PAIR) abitl[pr] = on ? (unsigned)__builtin_bit_cast(unsigned short, (bf16_t)(tw_ - (float)th_)) << (16 * (frow & 1)) : 0u;
    }
    // the lane's diagonal operand of tap pair pr, expanded from its one non-zero dword (two-term: one per term)
    auto diag = [&](int pr, bool opaque) {
        unsigned bits = abits[pr];
        if (opaque) asm volatile("" : "+v"(bits));              // expanded at use: 13 resident operands (52 registers) do not fit
        const u32x4 fr = {dq == 0 ? bits : 0u, dq == 1 ? bits : 0u, dq == 2 ? bits : 0u, dq == 3 ? bits : 0u};
        Frag<T> af;
        if constexpr (PAIR) {
            unsigned bl = abitl[pr];
            if (opaque) asm volatile("" : "+v"(bl));
            const u32x4 fl = {dq == 0 ? bl : 0u, dq == 1 ? bl : 0u, dq == 2 ? bl : 0u, dq == 3 ? bl : 0u};
            af.h = __builtin_bit_cast(bf16x8, fr);
            af.l = __builtin_bit_cast(bf16x8, fl);
        } else {
            af.v = __builtin_bit_cast(bf16x8, fr);
        }
        return af;
    };
    // 3 x 3: the five diagonal A operands stay expanded in registers for the whole band
    constexpr bool ARES = KS == 3;
    Frag<T> afr[ARES ? NPAIR : 1];
    if constexpr (ARES) {
#pragma unroll
        for (int pr = 0; pr < NPAIR; ++pr) afr[pr] = diag(pr, false);
    }

    const int oy_b = band * p.band_rows, oy_e = min(p.Ho, oy_b + p.band_rows);
    const int ox0 = strip * p.TWo, tw = min(p.TWo, p.Wo - ox0);
    const int ix0 = ox0 * S - p.pad_l, iy_top = oy_b * S - p.pad_t;
    constexpr int OOB = 0x7FFFFFF0;
    // expand: inside-the-image mask of the lane's pixel per tile (two-term: an AND mask; bf16: silu4_scaled's addend, 1 or +inf)
    unsigned cmask[MT];
    float caddc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int c = 16 * t + frow, ix = ix0 + c;
        const bool inside = c < p.IWs && ix >= 0 && ix < p.W;
        cmask[t] = inside ? 0xFFFFFFFFu : 0u;
        caddc[t] = inside ? 1.f : __builtin_inff();
    }
    const int xlane = frow * p.xpitch + kg * PB;                  // B operand of the expand: pixel frow of a tile, piece kg of a chunk
    const int xtile = 16 * p.xpitch;
    // The last K-chunk's lanes beyond Cin meet zero weights, but what they read must be FINITE (0 * NaN = NaN): beyond the last
    // pixel of the last X slot lies the first expanded ring, whose float32 words (two-term mode) read as bf16 pairs can be NaN
    // patterns.  Those lanes re-read the pixel's first piece instead.
    const int xlast = ((NKC - 1) * CHB + kg * PB < cbytes) ? (NKC - 1) * CHB : -kg * PB;
    char* const ring_e = ring + frow * PXB;                       // expand store: pixel frow of a tile (channels 4*kg.. by row_store4)
    // depthwise B operand of tile 0; tile u: + u * 16 * S * PXB  (two-term: the lane's 4 float32 channels of the pixel)
    const char* const dl = ring + frow * S * PXB + (PAIR ? kg * 16 : (kg & 1) * (PXB / 2));
    const char* const dlh = dl + hi * PXB;                        // ... for a pair of taps in one window row (second tap = next pixel)
    // output offsets: all lanes of the tiles before the last are inside the strip
    // (two-term: byte offset of the hi half of the lane's 4 channels inside their 8-channel group; lo: + 16)
    const int yoff0 = PAIR ? (ox0 + frow) * mid * 4 + ((c0 + 4 * kg) >> 3) * 32 + ((c0 + 4 * kg) & 7) * 2 : ((ox0 + frow) * mid + c0 + 4 * kg) * 2;
    const bool ylast_ok = 16 * (NO - 1) + frow < tw;
    const float vlast = ylast_ok ? 1.f : 0.f;
    const int yoff_last = ylast_ok ? yoff0 : OOB;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.X)) + (long long)b * p.H * p.W * cbytes, 0, p.H * p.W * cbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(p.Y) + (long long)b * p.Ho * p.Wo * mid * (int)sizeof(T), 0, p.Ho * p.Wo * mid * (int)sizeof(T), 0x00020000);

    // ---- X staging: this lane's pieces of a row (the same for every row)
    int sgoff[NPL], sloff[NPL];
    bool smine[NPL];
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        const int i = (int)threadIdx.x + j * (int)blockDim.x;
        const int px = fdiv(i, p.fd_ppr), piece = i - px * p.ppr;
        const int ix = ix0 + px;
        smine[j] = i < p.pieces_row;
        sgoff[j] = (smine[j] && ix >= 0 && ix < p.W) ? ix * cbytes + piece * 16 : OOB;
        sloff[j] = smine[j] ? p.x_off + px * p.xpitch + piece * 16 : 64 + lane * 16;      // idle lanes: the dump row
    }
    u32x4 xs[S][NPL];
    auto issue = [&](int rel, u32x4 (&dst)[NPL]) {
        if constexpr ((WIDE_ABLATE & 8) != 0) return;
        int iy = iy_top + rel;
        iy = iy < 0 ? 0 : (iy >= p.H ? p.H - 1 : iy);             // rows outside the image: any valid row (expand_row writes zeros for them)
        const int rowoff = iy * p.W * cbytes;                      // wave-uniform: the buffer op's scalar offset
#pragma unroll
        for (int j = 0; j < NPL; ++j) dst[j] = __builtin_amdgcn_raw_buffer_load_b128(xrs, sgoff[j], rowoff, 0);
    };
    auto commit = [&](int rel, const u32x4 (&src)[NPL]) {
        const int so = (rel & (NSX - 1)) * p.xslot_bytes;
        if constexpr ((WIDE_ABLATE & 8) != 0) return;
#pragma unroll
        for (int j = 0; j < NPL; ++j) *reinterpret_cast<u32x4*>(lds + sloff[j] + (smine[j] ? so : 0)) = src[j];
        if constexpr ((WIDE_ABLATE & 1) != 0) return;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // the wave's pieces are in LDS before its arrival is counted
        if (lane == 0) __hip_atomic_fetch_add(cnt + (rel & (NSX - 1)), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto poll = [&](int rel) {
        if constexpr ((WIDE_ABLATE & 1) != 0) return;
        const int target = p.nw * (rel / NSX + 1);
        int* c = cnt + (rel & (NSX - 1));
        bool arrived = false;
#pragma unroll 1
        for (int spin = 0; spin < WIDE_SPIN_LIMIT; ++spin) {
            const int v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            if (v >= target) { arrived = true; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        // Out of spins (cannot happen while the hand-off protocol holds): the wave goes on - a hung grid is worse than a wrong tile -
        // but says so in the library's device-side failure word, which the next effdet_* call reports as -5 (abi.hip)
        if (!arrived && lane == 0 && p.err_word != nullptr)
            __hip_atomic_fetch_or(p.err_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("" ::: "memory");                               // no X read moves above the poll
    };
    auto expand_row = [&](int rel, int slot_bytes) {
        const int iy = iy_top + rel;
        if ((WIDE_ABLATE & 4) != 0 || iy < 0 || iy >= p.H) {       // TF-SAME pads the EXPANDED map: rows outside the image are zeros
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                if constexpr (PAIR) *reinterpret_cast<f32x4*>(ring_e + slot_bytes + 16 * PXB * t + kg * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
                else row_store4<T>(ring_e + slot_bytes + 16 * PXB * t, 4 * kg, f32x4{0.f, 0.f, 0.f, 0.f});
            }
            return;
        }
        const char* xb = xring + (rel & (NSX - 1)) * p.xslot_bytes + xlane;
        // the B operands of tile t + 1 are requested right after the MFMAs of tile t, so their LDS latency hides behind tile t's SiLU
        Frag<T> xf[NKC];
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc) xf[kc] = ld_frag<T>(xb + (kc + 1 < NKC ? kc * CHB : xlast));
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            f32x4 acc = sh1;
#pragma unroll
            for (int kc = 0; kc < NKC; ++kc) mma_chunk(wf[kc], xf[kc], acc);
            if (t + 1 < MT) {
                xb += xtile;
#pragma unroll
                for (int kc = 0; kc < NKC; ++kc) xf[kc] = ld_frag<T>(xb + (kc + 1 < NKC ? kc * CHB : xlast));
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (PAIR || (WIDE_ABLATE & 16) != 0) {
                // two-term: pixels beyond the strip's input width read whatever follows the X slot - float32 ring words that can be NaN
                // patterns as bf16 pairs - and NaN * 0 stays NaN: those lanes are cleared bitwise (one AND mask register per tile)
                // (elements are copied to scalars first: __builtin_bit_cast applied to an ext-vector element lvalue read element 0 for all four)
                const f32x4 a_ = (WIDE_ABLATE & 16) ? acc : silu4_scaled(acc, 1.f);
                const unsigned cm = cmask[t];
                const float e0 = a_[0], e1 = a_[1], e2 = a_[2], e3 = a_[3];
                const f32x4 ev = {__builtin_bit_cast(float, __builtin_bit_cast(unsigned, e0) & cm), __builtin_bit_cast(float, __builtin_bit_cast(unsigned, e1) & cm),
                                  __builtin_bit_cast(float, __builtin_bit_cast(unsigned, e2) & cm), __builtin_bit_cast(float, __builtin_bit_cast(unsigned, e3) & cm)};
                if constexpr (PAIR) *reinterpret_cast<f32x4*>(ring_e + slot_bytes + 16 * PXB * t + kg * 16) = ev;      // float32 ring
                else row_store4<T>(ring_e + slot_bytes + 16 * PXB * t, 4 * kg, ev);
            } else {
                // bf16: the border mask rides in the SiLU's addend (+inf -> t * 0; what the masked lanes read - zeros from the staging,
                // or bf16 activations behind the X slot - is finite): no mask instructions, one register per tile
                row_store4<T>(ring_e + slot_bytes + 16 * PXB * t, 4 * kg, silu4_scaled(acc, caddc[t]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // one row event: wait for the row, expand it, hand over the row S steps ahead, fetch the row 2 S steps ahead
    auto event = [&](int rel, int slot_bytes, u32x4 (&regs)[NPL]) {
        poll(rel);
        expand_row(rel, slot_bytes);
        commit(rel + S, regs);
        issue(rel + 2 * S, regs);
    };

    __syncthreads();                                              // LDS zeroed (the only workgroup barrier of the kernel)
    // ---- pipeline fill: rows 0 .. S-1 in LDS, rows S .. 2S-1 in flight
#pragma unroll
    for (int r = 0; r < S; ++r) { issue(r, xs[r]); commit(r, xs[r]); }
#pragma unroll
    for (int r = 0; r < S; ++r) issue(S + r, xs[r]);
    int next_rel = 0;
    // prologue: the first KS - S rows of the band's window (ring slots 0 .. KS-S-1)
#pragma unroll
    for (int r = 0; r < KS - S; ++r) { event(r, r * rowbytes, xs[r % S]); }
    next_rel = KS - S;

    float pl[4] = {0.f, 0.f, 0.f, 0.f};
    int yrow = oy_b * p.Wo * mid * (int)sizeof(T);            // byte offset of the output row inside the image (scalar offset)
    const int ypitch = p.Wo * mid * (int)sizeof(T);
    const int ytile = 16 * mid * (int)sizeof(T);
    int oy = oy_b;
    // One output row.  PH = ring slot of the first row of its KS-row window.
    auto step = [&](auto PHC) {
        constexpr int PH = decltype(PHC)::value;
#pragma unroll
        for (int r = 0; r < S; ++r) event(next_rel + r, ((PH + KS - S + r) % KS) * rowbytes, xs[(KS - S + r) % S]);
        next_rel += S;
        __builtin_amdgcn_sched_barrier(0);
        // ---- depthwise: one output row out of the ring.  The taps are taken in pairs (2 pr, 2 pr + 1); lanes of the upper K half
        // (hi) read the second tap.  Where both taps lie in the same window row the second one is the next pixel (+32 bytes: folded
        // into the lane's base `dlh`) and the offset is an immediate; the few pairs that straddle two rows (2 of 13, 1 of 5) add
        // hi * (offb - offa).  The B operands are requested in batches of G pairs x NO tiles before the batch's first MFMA, and the
        // diagonal A operands of the batch are expanded meanwhile (5 x 5; the 3 x 3 kernels keep theirs resident).
        int hsel = hi;
        asm volatile("" : "+v"(hsel));                          // keeps the straddling pairs' offsets inside the row loop
        f32x4 acc[NO];
#pragma unroll
        for (int u = 0; u < NO; ++u) acc[u] = t2v;
        if constexpr (PAIR) {
            // float32 depthwise on the vector ALU (see mbconv_roll.hip): one 16-byte ring read and two packed FMAs per tap and tile
#pragma unroll
            for (int dy = 0; dy < ((WIDE_ABLATE & 2) ? 0 : KS); ++dy) {
                f32x4 e[KS][NO];
#pragma unroll
                for (int dx = 0; dx < KS; ++dx)
#pragma unroll
                    for (int u = 0; u < NO; ++u)
                        e[dx][u] = *reinterpret_cast<const f32x4*>(dl + ((PH + dy) % KS) * rowbytes + dx * PXB + u * (16 * S * PXB));
#pragma unroll
                for (int dx = 0; dx < KS; ++dx)
#pragma unroll
                    for (int u = 0; u < NO; ++u) acc[u] = e[dx][u] * wv[dy * KS + dx] + acc[u];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#ifndef WIDE_G
#define WIDE_G 12
#endif
        constexpr int G = WIDE_G / NO < NPAIR ? WIDE_G / NO : NPAIR;     // (two-term: the same count of twice as large fragments, at twice the register budget)
#pragma unroll
        for (int p0 = 0; p0 < ((WIDE_ABLATE & 2) || PAIR ? 0 : NPAIR); p0 += G) {
            Frag<T> bq[G][NO];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int pr = p0 + g;
                if (pr < NPAIR) {
                    const int ta = 2 * pr, tb = 2 * pr + 1 < NTAP ? 2 * pr + 1 : 2 * pr;        // constants after unrolling
                    const int offa = ((PH + ta / KS) % KS) * rowbytes + (ta % KS) * PXB;
                    const int offb = ((PH + tb / KS) % KS) * rowbytes + (tb % KS) * PXB;
                    const bool same_row = ta / KS == tb / KS;
                    const char* src = same_row ? dlh + offa : dl + offa + hsel * (offb - offa);
#pragma unroll
                    for (int u = 0; u < NO; ++u) bq[g][u] = ld_frag<T>(src + u * (16 * S * PXB));
                }
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int pr = p0 + g;
                if (pr < NPAIR) {
                    Frag<T> af;
                    if constexpr (ARES) af = afr[pr]; else af = diag(pr, true);
#pragma unroll
                    for (int u = 0; u < NO; ++u) mma_chunk(af, bq[g][u], acc[u]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);                  // batches stay batches: hoisting every read of the row would spill
        }
#pragma unroll
        for (int u = 0; u < NO; ++u) {
            const f32x4 ov = act4_w(acc[u]);
            const bool last = u == NO - 1;
            if (last) {
#pragma unroll
                for (int r = 0; r < 4; ++r) pl[r] += ov[r] * vlast;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) pl[r] += ov[r];
            }
            const int yo = last ? yoff_last : yoff0;
            if constexpr (PAIR) {                               // (an out-of-range offset + 16 is out of range too: the store is dropped)
                u32x2 oh, ol;
                pair_split4(ov, oh, ol);
                __builtin_amdgcn_raw_buffer_store_b64(oh, yrs, yo, yrow + u * ytile, 0);
                __builtin_amdgcn_raw_buffer_store_b64(ol, yrs, yo + 16, yrow + u * ytile, 0);
            } else {
                const bf16x4 ob = {(bf16_t)ov[0], (bf16_t)ov[1], (bf16_t)ov[2], (bf16_t)ov[3]};
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, ob), yrs, yo, yrow + u * ytile, 0);
            }
        }
        yrow += ypitch;
        ++oy;
    };
#pragma unroll 1
    while (oy < oy_e) {
        if constexpr (KS == 3) {
            step(IntW<0>{});
            if (oy < oy_e) step(IntW<(S) % 3>{});
            if (oy < oy_e) step(IntW<(2 * S) % 3>{});
        } else {
            step(IntW<0>{});
            if (oy < oy_e) step(IntW<(S) % 5>{});
            if (oy < oy_e) step(IntW<(2 * S) % 5>{});
            if (oy < oy_e) step(IntW<(3 * S) % 5>{});
            if (oy < oy_e) step(IntW<(4 * S) % 5>{});
        }
    }
    if (p.pool_partial != nullptr) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = pl[r];
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
            pl[r] = v;
        }
        if (frow == 0) {
            float* dst = p.pool_partial + ((long long)b * (p.nstrips * p.nbands) + band * p.nstrips + strip) * mid + c0 + 4 * kg;
            *reinterpret_cast<f32x4*>(dst) = f32x4{pl[0], pl[1], pl[2], pl[3]};
        }
    }
}

struct WideGeometry {
    bool use; int TWo, nstrips, IWs, IWa, band_rows, nbands, nw, ngroups, ring_bytes, nkc, xpitch, xslot_bytes, npl, wg_per_cu; size_t lds;
};

// Geometry depends on the map and channel sizes only - never on the batch - so that an image's result (including the order
// in which its SE pool partials are summed) is the same at every batch size.
WideGeometry pick_wide(int H, int W, int Cin, int mid, int k, int stride, bool pair = false) {
    WideGeometry g{};
    g.use = false;
    const int esz = pair ? 4 : 2, pxb = 16 * esz;                  // bytes per channel / per 16-channel ring pixel
    const int cbytes = Cin * esz;
    g.nkc = (Cin + 31) / 32;                                    // K-chunks of 32 channels (64 bytes; two-term bf16: 128 bytes)
#ifdef WIDE_NARROW       /* experiment (variant builds only): the shared-X form for inputs of 8 ... 64 channels too */
    if (g.nkc > 6 || mid % 16 || Cin % 8) return g;
#else
    if ((g.nkc < 3 || g.nkc > 6) || mid % 16 || Cin % 8) return g;        // narrower inputs: mbconv_roll.hip; wider: the band x slice form
#endif
    const int Ho = same_out(H, stride), Wo = same_out(W, stride);
    // one strip per row where the row fits 64 input pixels, else equal strips
    int best_ns = 0;
    for (int ns = 1; ns <= (Wo + 7) / 8; ++ns) {
        const int two = (Wo + ns - 1) / ns;
        if ((ns - 1) * two >= Wo) continue;
        const int iws = (two - 1) * stride + k;
        if (iws > (pair ? 32 : (k == 5 ? 48 : 64))) continue;       // (two-term: rings are twice as large per pixel - narrower strips)
        best_ns = ns; g.TWo = two; g.IWs = iws; g.IWa = (iws + 15) / 16 * 16;
        break;
    }
    if (!best_ns) return g;
    g.nstrips = best_ns;
    const int mt = g.IWa / 16, no = (g.TWo + 15) / 16;
    if (mt > 4 || no > 4) return g;
    // 16-pixel tiles on narrow rows waste lanes (a 20-pixel row: 24 of 32 input and 20 of 32 output pixels used), and every lane
    // pays the SiLU: measured on d0's 20 x 20 maps (blocks 5.1 ... 6.0) this form ran 0.104 / 0.079 ms against 0.079 / 0.062 ms of
    // the band x slice form, whose expand walks flat pixel tiles - those maps stay there
    {
        const double we = stride * mt * (1.0 + 0.06 * g.nkc), wd = no * (1.2 + 0.05 * ((k * k + 1) / 2));   // relative cost of the two phases
        const double use = (we * g.IWs / (16.0 * mt) + wd * g.TWo / (16.0 * no)) / (we + wd);
        if (use < 0.75 && !pair) return g;                          // (the two-term mode has no band x slice form to fall back to)
    }
    g.ring_bytes = k * g.IWa * pxb;
    // X ring: pixel pitch = Cin bytes rounded up to 16 * (2 mod 4): the 16-lane groups of ds_read_b128 then touch 16 distinct
    // 16-byte slots of the 256-byte bank row (conflict-free B operand reads)
    int P = cbytes / 16;
    while (P % 4 != 2) ++P;
    g.xpitch = P * 16;
    g.xslot_bytes = (g.IWs * g.xpitch + 15) / 16 * 16;
    const int nsx = 2 * stride;
    const size_t xbytes = (size_t)nsx * g.xslot_bytes;
    // Waves per workgroup (a divisor of the channel-tile count), workgroups per CU and bands per image are chosen together by a
    // small cost model of the launch at the reference batch of 64 images on 256 CUs (the geometry itself never depends on the
    // actual batch): the busiest CU gets L = ceil(64 * workgroups per image / 256) workgroups and runs them `per_cu` at a time;
    // a round costs (rows of the band + the k - stride re-expanded halo rows) x the waves that share the CU (the row loop is
    // issue-bound, so per-row time grows with the resident waves; below 8 waves the CU is latency-bound instead).
    const int tiles = mid / 16;
    const size_t budget = 160 * 1024;
    // lanes beyond the strip's last output pixel read (and drop) ring pixels up to (16 NO - 1) S + k - 1: the last wave's last
    // ring row needs that much readable, zeroed LDS behind it
    const int over = ((16 * no - 1) * stride + k) * pxb - g.IWa * pxb;
    const size_t tail_pad = (size_t)((over > 0 ? over : 0) + 64 + 15) / 16 * 16;
    const int npair = (k * k + 1) / 2;
    const double ce = mt * (1.0 + 0.06 * g.nkc), cd = no * (1.2 + 0.05 * npair);
    double best = -1.0;
    for (int d = 2; d <= 16; ++d) {
        if (tiles % d) continue;
        const size_t l = WIDE_HDR + xbytes + (size_t)d * g.ring_bytes + tail_pad;
        if (l > budget) continue;
        const int npl = (g.IWs * (cbytes / 16) + d * 64 - 1) / (d * 64);          // 16-byte X pieces a lane stages per row
        if (npl > (pair ? 4 : 1)) continue;                                        // bf16: one piece per lane and row
        int cap = (int)(budget / l);
        const int rcap = (pair ? 8 : 16) / d;                                      // register budget: 128 (two-term: 256) VGPRs = 16 (8) waves per CU
        if (rcap < 1) continue;
        if (cap > rcap) cap = rcap;
        for (int per_cu = 1; per_cu <= cap; ++per_cu) {
            for (int nb = 1; nb <= 4; ++nb) {
                const int rows = (Ho + nb - 1) / nb;
                if (nb > 1 && rows * stride < k) break;
                const int nbe = (Ho + rows - 1) / rows;
                const long long wgs = 64LL * (tiles / d) * g.nstrips * nbe;
                const int L = (int)((wgs + 255) / 256);
                const int conc = L < per_cu ? L : per_cu;
                const int waves = conc * d;
                const double cost = (double)((L + per_cu - 1) / per_cu) * (rows * (stride * ce + cd) + (k - stride) * ce) * (waves > 8 ? waves : 8);
                if (best < 0 || cost < best * 0.999 || (cost < best * 1.001 && d > g.nw)) {
                    best = cost; g.nw = d; g.wg_per_cu = per_cu; g.lds = l; g.nbands = nbe; g.band_rows = rows; g.npl = npl;
                }
            }
        }
    }
    if (best < 0) return g;
#ifdef WIDE_TUNE
    // geometry sweep of tools/wide_tune_gpu.sh (variant build only): EFFDET_WIDE_TUNE="nw,per_cu,nbands"
    if (const char* tv = getenv("EFFDET_WIDE_TUNE")) {
        int d = 0, pc = 0, nb = 0;
        if (sscanf(tv, "%d,%d,%d", &d, &pc, &nb) == 3 && d >= 1 && d <= 16 && tiles % d == 0 && nb >= 1) {
            const size_t l = WIDE_HDR + xbytes + (size_t)d * g.ring_bytes + tail_pad;
            const int rows = (Ho + nb - 1) / nb;
            if (l * pc <= budget && (g.IWs * (cbytes / 16) + d * 64 - 1) / (d * 64) <= 1 && d * pc <= 16) {
                g.nw = d; g.wg_per_cu = pc; g.lds = l; g.band_rows = rows; g.nbands = (Ho + rows - 1) / rows;
            }
        }
    }
#endif
    g.ngroups = tiles / g.nw;
    g.use = true;
    return g;
}

template <int KS, int S, int NKC, int MT, int NO, typename T>
void (*wide_kernel_npl(int npl))(WideArgs) {
    if constexpr (IsPair<T>::value) {
        switch (npl) {
            case 1: return mbconv_wide_kernel<KS, S, NKC, MT, NO, 1, T>;
            case 2: return mbconv_wide_kernel<KS, S, NKC, MT, NO, 2, T>;
            case 3: return mbconv_wide_kernel<KS, S, NKC, MT, NO, 3, T>;
            case 4: return mbconv_wide_kernel<KS, S, NKC, MT, NO, 4, T>;
            default: return nullptr;
        }
    } else {
        return npl == 1 ? mbconv_wide_kernel<KS, S, NKC, MT, NO, 1, T> : nullptr;
    }
}

template <int KS, int S, int NKC, typename T>
void (*wide_kernel_for(int mt, int no, int npl))(WideArgs) {
    // MT = ceil(IWs / 16) input tiles, NO = ceil(TWo / 16) output tiles
    if constexpr (S == 1) {
        if (mt == 2) return no == 1 ? wide_kernel_npl<KS, S, NKC, 2, 1, T>(npl) : no == 2 ? wide_kernel_npl<KS, S, NKC, 2, 2, T>(npl) : nullptr;
        if (mt == 3) return no == 2 ? wide_kernel_npl<KS, S, NKC, 3, 2, T>(npl) : no == 3 ? wide_kernel_npl<KS, S, NKC, 3, 3, T>(npl) : nullptr;
        if (mt == 4) return no == 3 ? wide_kernel_npl<KS, S, NKC, 4, 3, T>(npl) : no == 4 ? wide_kernel_npl<KS, S, NKC, 4, 4, T>(npl) : nullptr;
    } else {
        if (mt == 2) return no == 1 ? wide_kernel_npl<KS, S, NKC, 2, 1, T>(npl) : nullptr;
        if (mt == 3) return no == 1 ? wide_kernel_npl<KS, S, NKC, 3, 1, T>(npl) : no == 2 ? wide_kernel_npl<KS, S, NKC, 3, 2, T>(npl) : nullptr;
        if (mt == 4) return no == 2 ? wide_kernel_npl<KS, S, NKC, 4, 2, T>(npl) : nullptr;
    }
    return nullptr;
}

template <int KS, int S, typename T>
int launch_wide_ks(hipStream_t st, const WideArgs& r, const WideGeometry& g) {
    void (*kern)(WideArgs) = nullptr;
    const int mt = g.IWa / 16, no = (g.TWo + 15) / 16;
    if constexpr (IsPair<T>::value) {
        if (mt != 2) return EFFDET_EINVAL;                        // two-term strips are at most 32 input pixels wide
        switch (g.nkc) {
            case 3: kern = no == 1 ? wide_kernel_npl<KS, S, 3, 2, 1, T>(g.npl) : (no == 2 && S == 1) ? wide_kernel_npl<KS, S, 3, 2, 2, T>(g.npl) : nullptr; break;
            case 4: kern = no == 1 ? wide_kernel_npl<KS, S, 4, 2, 1, T>(g.npl) : (no == 2 && S == 1) ? wide_kernel_npl<KS, S, 4, 2, 2, T>(g.npl) : nullptr; break;
            case 5: kern = no == 1 ? wide_kernel_npl<KS, S, 5, 2, 1, T>(g.npl) : (no == 2 && S == 1) ? wide_kernel_npl<KS, S, 5, 2, 2, T>(g.npl) : nullptr; break;
            case 6: kern = no == 1 ? wide_kernel_npl<KS, S, 6, 2, 1, T>(g.npl) : (no == 2 && S == 1) ? wide_kernel_npl<KS, S, 6, 2, 2, T>(g.npl) : nullptr; break;
            default: break;
        }
    } else
    switch (g.nkc) {
#ifdef WIDE_NARROW
        case 1: kern = wide_kernel_for<KS, S, 1, T>(mt, no, g.npl); break;
        case 2: kern = wide_kernel_for<KS, S, 2, T>(mt, no, g.npl); break;
#endif
        case 3: kern = wide_kernel_for<KS, S, 3, T>(mt, no, g.npl); break;
        case 4: kern = wide_kernel_for<KS, S, 4, T>(mt, no, g.npl); break;
        case 5: kern = wide_kernel_for<KS, S, 5, T>(mt, no, g.npl); break;
        case 6: kern = wide_kernel_for<KS, S, 6, T>(mt, no, g.npl); break;
        default: break;
    }
    if (kern == nullptr) return EFFDET_EINVAL;
    if (g.lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return EFFDET_ELAUNCH;
    }
    const int rounds = (r.B + 7) / 8;
    hipLaunchKernelGGL(kern, dim3(rounds * r.per_image * 8), dim3(g.nw * 64), g.lds, st, r);
    return effdet_check_launch();
}

bool wide_supported(const WideGeometry& g, int k, int stride) {
    if (!g.use) return false;
    const int mt = g.IWa / 16, no = (g.TWo + 15) / 16;
    if (stride == 1) return (mt == 2 && (no == 1 || no == 2)) || (mt == 3 && (no == 2 || no == 3)) || (mt == 4 && (no == 3 || no == 4));
    return (mt == 2 && no == 1) || (mt == 3 && (no == 1 || no == 2)) || (mt == 4 && no == 2);
}

}  // namespace

// internal (not part of the C ABI): used by mbconv.hip's launcher
int effdet_mbconv_wide_parts(int H, int W, int Cin, int mid, int k, int stride, int pair) {
    const WideGeometry g = pick_wide(H, W, Cin, mid, k, stride, pair != 0);
    return wide_supported(g, k, stride) ? g.nstrips * g.nbands : 0;
}

int effdet_mbconv_wide_launch(hipStream_t st, const void* X, void* Y, const void* W1, const float* s1, const float* t1,
                              const float* taps, const float* s2, const float* t2, float* pool_partial,
                              int B, int H, int W, int Cin, int mid, int k, int stride, int pair, int sym) {
    const WideGeometry g = pick_wide(H, W, Cin, mid, k, stride, pair != 0);
    if (!wide_supported(g, k, stride)) return EFFDET_EINVAL;
#ifdef WIDE_TUNE
    static thread_local long long printed_ = -1;
    const long long key_ = ((((long long)H * 4096 + W) * 4096 + Cin) * 8 + k) * 4 + stride;
    if (getenv("EFFDET_GEOM_DEBUG") && printed_ != key_ && ((printed_ = key_), true))   // host-side print of the chosen geometry (tools/mbconv_layers.py)
        fprintf(stderr, "mbconv_wide H=%d W=%d Cin=%d mid=%d k=%d s=%d: nw=%d groups=%d strips=%d bands=%d x %d rows, %d WG/CU, lds=%zu, MT=%d NO=%d NKC=%d, WGs(B=%d)=%d\n",
                H, W, Cin, mid, k, stride, g.nw, g.ngroups, g.nstrips, g.nbands, g.band_rows, g.wg_per_cu, g.lds, g.IWa / 16, (g.TWo + 15) / 16,
                g.nkc, B, ((B + 7) / 8) * 8 * g.nstrips * g.nbands * g.ngroups);
#endif
    WideArgs r{};
    r.X = X; r.Y = Y; r.W1 = W1; r.s1 = s1; r.t1 = t1; r.taps = taps; r.s2 = s2; r.t2 = t2; r.pool_partial = pool_partial;
    r.B = B; r.H = H; r.W = W; r.Cin = Cin; r.mid = mid; r.Ho = same_out(H, stride); r.Wo = same_out(W, stride);
    r.pad_t = pad_before(H, k, stride, sym); r.pad_l = pad_before(W, k, stride, sym);
    r.TWo = g.TWo; r.nstrips = g.nstrips; r.band_rows = g.band_rows; r.nbands = g.nbands; r.IWs = g.IWs; r.nw = g.nw; r.ngroups = g.ngroups;
    r.ring_bytes = g.ring_bytes; r.per_image = g.nstrips * g.nbands * g.ngroups;
    r.xpitch = g.xpitch; r.xslot_bytes = g.xslot_bytes; r.ppr = Cin * (pair ? 4 : 2) / 16; r.pieces_row = g.IWs * r.ppr;
    r.x_off = WIDE_HDR; r.ring_off = WIDE_HDR + 2 * stride * g.xslot_bytes; r.lds_bytes = (int)g.lds;
    r.fd_ppr = make_fastdiv(r.ppr);
    r.err_word = effdet_device_error_word();
    if (pair) {
        if (k == 3) return stride == 1 ? launch_wide_ks<3, 1, bf16p_t>(st, r, g) : launch_wide_ks<3, 2, bf16p_t>(st, r, g);
        return stride == 1 ? launch_wide_ks<5, 1, bf16p_t>(st, r, g) : launch_wide_ks<5, 2, bf16p_t>(st, r, g);
    }
    if (k == 3) return stride == 1 ? launch_wide_ks<3, 1, bf16_t>(st, r, g) : launch_wide_ks<3, 2, bf16_t>(st, r, g);
    return stride == 1 ? launch_wide_ks<5, 1, bf16_t>(st, r, g) : launch_wide_ks<5, 2, bf16_t>(st, r, g);
}
