// Rolling-window form of the fused MBConv front half (bfloat16 throughput mode):
//
//   expand 1x1 conv (MFMA) -> BN1 -> SiLU -> depthwise k x k (stride 1|2, TF-SAME) -> BN2 -> SiLU (+ SE pool partial sums)
//
// replaces timm's InvertedResidual.conv_pw/bn1/act1/conv_dw/bn2/act2 (reached from effdet/efficientdet.py:837), like
// mbconv.hip's two forms, but every WAVE is an autonomous streaming engine and there is no workgroup barrier at all.
// A wave owns (image, band of output rows, column strip, 16 expanded channels).  It walks its band one output row at a
// time and keeps the last KS expanded input rows of its strip in a private LDS ring [KS rows][MT * 16 px][16 ch], so no
// expanded row is ever computed twice (no y halo).  Per output row:
//   expand:    S new input rows.  X pixels go straight from global memory (L2) into the MFMA B operand - the loads for
//              output row oy + 1 are issued before the depthwise phase of row oy, so their latency hides behind it; the
//              wave's W1 rows (A operand, BN1 scale and - for the gated form - the producing block's SE gate folded in) live
//              in registers for the whole band; the BN1 shift is the accumulator's initial value; SiLU in registers; one
//              8-byte LDS store per lane (4 channels of one pixel); zero outside the image (TF-SAME pads the EXPANDED map);
//   depthwise: one output row on the matrix cores (A = diag(w[t0]) | diag(w[t1]) with BN2's scale folded in, B = two
//              shifted reads of the ring), BN2 shift as the initial accumulator, SiLU, SE pool sums, 8-byte NHWC stores.
// LDS reads and writes of one wave execute in order, which is all the synchronisation the ring needs.  The row loop is
// unrolled over the KS ring phases so that every LDS offset is an immediate.  The x halo (KS - S columns per strip) is the
// only recomputation.  Workgroups are bundles of `wpg` such waves (same image / band / strip, consecutive channel tiles);
// blockIdx is remapped so that all workgroups of an image run on one XCD (its L2 then holds that image's X rows once).
//
// X loads and Y stores are BUFFER operations with hardware range checking: a lane whose pixel lies outside the strip /
// image gets an out-of-range offset (the load returns zeros, the store is dropped), so neither needs an exec-mask branch.
// The loop body is therefore straight-line code and the compiler can COUNT the memory operations in `s_waitcnt vmcnt(N)`:
// the wait for the prefetched X rows does not also wait for the younger output stores (vmcnt retires in issue order).
#include "common.h"

namespace {

struct RollArgs {
    const void* X; void* Y; const void* W1; const float* in_gate;
    const float* s1; const float* t1; const float* taps; const float* s2; const float* t2;
    float* pool_partial;
    int B, H, W, Cin, mid, Ho, Wo, pad_t, pad_l;
    int TWo, nstrips, band_rows, nbands, IWs, wpg, ngroups, ring_bytes, per_image;
};

// SiLU of four accumulator values on packed fp32 instructions (the BN shift already sits in the accumulator)
typedef float f32x2_ __attribute__((ext_vector_type(2)));
DEV f32x4 silu4_fast(const f32x4 x) {
    const f32x2_ x0 = {x[0], x[1]}, x1 = {x[2], x[3]};
    const f32x2_ t0 = x0 * -1.4426950408889634f, t1 = x1 * -1.4426950408889634f;
    const f32x2_ d0 = f32x2_{__builtin_amdgcn_exp2f(t0[0]), __builtin_amdgcn_exp2f(t0[1])} + 1.0f;
    const f32x2_ d1 = f32x2_{__builtin_amdgcn_exp2f(t1[0]), __builtin_amdgcn_exp2f(t1[1])} + 1.0f;
    const f32x2_ y0 = x0 * f32x2_{__builtin_amdgcn_rcpf(d0[0]), __builtin_amdgcn_rcpf(d0[1])};
    const f32x2_ y1 = x1 * f32x2_{__builtin_amdgcn_rcpf(d1[0]), __builtin_amdgcn_rcpf(d1[1])};
    return f32x4{y0[0], y0[1], y1[0], y1[1]};
}

template <int V> struct IntC { static constexpr int value = V; };
#ifndef ROLL_PFD2
#define ROLL_PFD2 0
#endif

// Phase ablation for timing experiments: exists only in variant builds (`make variant TAG=.. UNIT=mbconv_roll VDEFS=-DROLL_ABLATE=n`,
// libeffdet_hip_<TAG>.so, never loaded by the package); the product library is compiled with ROLL_ABLATE = 0.
// 1: no depthwise arithmetic  2: no expand arithmetic  4: no Y stores  8: no X loads  16: SiLU -> identity  32: X loads always hit (row 0)
#ifndef ROLL_ABLATE
#define ROLL_ABLATE 0
#endif
#ifndef ROLL_SYNC
#define ROLL_SYNC 0
#endif
DEV f32x4 roll_act(const f32x4 x) {
    if constexpr ((ROLL_ABLATE & 16) != 0) return x; else return silu4_fast(x);
}

// Pixels a lane beyond the strip's last output may read past the end of a ring row (its window starts at pixel (16 NO - 1) S at
// most; + 1: the hi lanes of the single last tap read the pixel after the window, against zero weights - which still must not
// meet a NaN): behind the LAST slot that is past the wave's ring, so every wave's ring carries this many zeroed pad pixels
constexpr int roll_pad_px(int ks, int s, int mt, int no) {
    const int over = (16 * no - 1) * s + ks + 1 - 16 * mt;
    return over > 0 ? over : 0;
}

// NO = output tiles (16 px) per strip row: a compile-time count, so that every output row issues the same number of loads
// and stores and the compiler can count them in its waits
// T: bf16_t, or bf16p_t (dtype 2, two-term bf16: X, W1 and Y carry hi + lo and the expand GEMM takes three MFMAs per chunk; NKC
// then counts 128-byte chunks - still 32 channels each.  The expanded ring holds FLOAT32 there (64 bytes per 16-channel pixel) and
// the depthwise taps run on the vector ALU in float32: with two-term operands the diagonal-MFMA form costs 3 x 16 cycles per tap
// pair and 16 px x 16 ch tile (240 / 624 cycles for 3 x 3 / 5 x 5) against 72 / 200 cycles of packed float32 FMAs, and the
// expanded values need no splitting at all)
// NJ: channel tiles (16 expanded channels each) per wave.  Every tile of a wave uses the SAME X fragments: with NJ = 1 each of the
// mid / 16 waves of a strip fetches the strip's X rows from L2 itself, which costs the 3 x 3 blocks of the high-resolution stages 20 - 40 %
// of their time (profiles/r04_roll_ablation.txt); NJ tiles per wave divide the readers (and the X loads / waits per unit of work) by NJ.
template <int KS, int S, int NKC, int MT, int NO, typename T, int NJ = 1>
__global__ __launch_bounds__(512, (IsPair<T>::value ? (KS == 3 && NKC == 1 && MT <= 3 && NJ == 1 ? 3 : 2) : (NKC <= 2 ? 4 : 3))) void mbconv_roll_kernel(RollArgs p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr bool PAIR = IsPair<T>::value;
    constexpr int PB = OpGeom<T>::PIECE, CHB = OpGeom<T>::CHUNK;     // bytes of a lane's operand piece / of a 32-channel K-chunk
    constexpr int PXB = 16 * (int)sizeof(T);                         // bytes of a ring pixel (16 channels)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // provably wave-uniform: everything derived from it stays scalar
    const int frow = lane & 15, kg = lane >> 4;
    // blocks are dealt round-robin over the 8 XCDs: image = (round, xcd), so one image's workgroups share an L2
    const int xcd = blockIdx.x & 7, rr_ = blockIdx.x >> 3;
    const int b = (rr_ / p.per_image) * 8 + xcd;
    if (b >= p.B) return;
    int q = rr_ % p.per_image;
    const int group = q % p.ngroups; q /= p.ngroups;
    const int strip = q % p.nstrips, band = q / p.nstrips;
    const int c0 = 16 * NJ * (group * p.wpg + wave);              // first channel of the wave's NJ tiles
    // an odd tile count leaves the strip's last wave with fewer tiles: njw of them are real (wave-uniform); the others read tile 0's
    // constants and skip all their work and stores
    const int njw = NJ == 1 ? 1 : min(NJ, p.mid / 16 - NJ * (group * p.wpg + wave));
    auto cj = [&](int j) { return c0 + (j < njw ? 16 * j : 0); };
    const int cbytes = p.Cin * (int)sizeof(T), mid = p.mid;
    char* ring = lds + wave * p.ring_bytes;
    constexpr int rowbytes = MT * 16 * PXB;                       // [MT * 16 px][16 ch]
    constexpr int NTAP = KS * KS, NPAIR = (NTAP + 1) / 2;
    constexpr int OTN = NO;
    constexpr int RB1 = KS * rowbytes + roll_pad_px(KS, S, MT, NO) * PXB;      // one channel tile's ring + its zeroed pad

    // ---- per-wave constants.  Every load of the prologue is issued before the first use.
    // W1 rows (A operand of the expand): lane (frow, kg) holds 8 consecutive K of channel c0 + frow per 64-byte chunk
    Frag<T> wf[NJ][NKC];
    const bool gated = p.in_gate != nullptr;
    float rs1[NJ], rs2[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) { rs1[j] = p.s1[cj(j) + frow]; rs2[j] = p.s2[cj(j) + frow]; }
    f32x4 g0[NKC], g1[NKC];
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) {
        const int off = kc * CHB + kg * PB;
        const bool kv = off < cbytes;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            wf[j][kc] = ld_frag<T>(reinterpret_cast<const char*>(p.W1) + (long long)(cj(j) + frow) * cbytes + (kv ? off : 0));
        if (gated) {
            const float* g = p.in_gate + (long long)b * p.Cin + (kv ? off / (int)sizeof(T) : 0);
            g0[kc] = *reinterpret_cast<const f32x4*>(g); g1[kc] = *reinterpret_cast<const f32x4*>(g + 4);
        }
    }
    const int hi = kg >> 1;
    const bool dactive = (kg & 1) == (frow >> 3);
    const int dq = (frow & 7) >> 1;
    float tapv[NJ][NPAIR];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int pr = 0; pr < NPAIR; ++pr) {
            const int t = 2 * pr + hi;
            tapv[j][pr] = p.taps[(long long)(t < NTAP ? t : 0) * mid + cj(j) + frow];
        }
    // the expand GEMM produces t = -log2(e) x directly (silu4_scaled, common.h), the taps carry -ln 2
    constexpr float ESC = -1.4426950408889634f, EINV = -0.6931471805599453f;
    f32x4 sh1[NJ], t2v[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        sh1[j] = *reinterpret_cast<const f32x4*>(p.t1 + cj(j) + 4 * kg) * ESC;
        t2v[j] = *reinterpret_cast<const f32x4*>(p.t2 + cj(j) + 4 * kg);
    }
    // two-term mode: the lane's 4 channels of every tap, BN2's scale folded in (float32 vector-ALU depthwise)
    f32x4 wv[PAIR ? NJ : 1][PAIR ? NTAP : 1];
    if constexpr (PAIR) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const f32x4 s2q = *reinterpret_cast<const f32x4*>(p.s2 + cj(j) + 4 * kg);
#pragma unroll
            for (int t = 0; t < NTAP; ++t) wv[j][t] = *reinterpret_cast<const f32x4*>(p.taps + (long long)t * mid + cj(j) + 4 * kg) * (s2q * EINV);
        }
    }
    // BN1's scale of the row's channel is folded into the bf16 weights (the shift is the accumulator's initial value), and so
    // is the SE gate of the producing block along K where that block's project conv was composed into W1
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) {
        const bool kv = kc * CHB + kg * PB < cbytes;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float ge = gated ? (e < 4 ? g0[kc][e & 3] : g1[kc][e & 3]) : 1.f;
            if constexpr (PAIR) {                                 // scale the VALUE in float32, then split again
                const float w = ((float)wf[j][kc].h[e] + (float)wf[j][kc].l[e]) * (rs1[j] * ge * ESC);
                const bf16_t wh = (bf16_t)w;
                wf[j][kc].h[e] = kv ? wh : (bf16_t)0.f;
                wf[j][kc].l[e] = kv ? (bf16_t)(w - (float)wh) : (bf16_t)0.f;
            } else {
                wf[j][kc].v[e] = kv ? (bf16_t)((float)wf[j][kc].v[e] * (rs1[j] * ge * ESC)) : (bf16_t)0.f;
            }
        }
    }
    // diagonal tap operands: the lane's single non-zero dword of diag(w[t0]) | diag(w[t1]), BN2's scale folded in
    unsigned abits[NJ][NPAIR], abitl[PAIR ? NPAIR : 1];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int pr = 0; pr < NPAIR; ++pr) {
        const bool on = dactive && 2 * pr + hi < NTAP;
        const float tw_ = tapv[j][pr] * (rs2[j] * EINV);
        const bf16_t th_ = (bf16_t)tw_;
        abits[j][pr] = on ? (unsigned)__builtin_bit_cast(unsigned short, th_) << (16 * (frow & 1)) : 0u;
        if constexpr (PAIR) abitl[pr] = on ? (unsigned)__builtin_bit_cast(unsigned short, (bf16_t)(tw_ - (float)th_)) << (16 * (frow & 1)) : 0u;
    }
    // the lane's diagonal operand of tap pair pr of channel tile j, expanded from its one non-zero dword (two-term: one per term)
    auto diag = [&](int pr, int j = 0) {
        unsigned bits = abits[j][pr];
        if constexpr (NJ > 1) asm volatile("" : "+v"(bits));      // several tiles: expanded at use (NJ x 5 resident operands would not fit)
        if constexpr (KS == 5) asm volatile("" : "+v"(bits));     // 13 resident operands (52 registers) do not fit: expanded at use
        const u32x4 fr = {dq == 0 ? bits : 0u, dq == 1 ? bits : 0u, dq == 2 ? bits : 0u, dq == 3 ? bits : 0u};
        Frag<T> af;
        if constexpr (PAIR) {
            unsigned bl = abitl[pr];
            if constexpr (KS == 5) asm volatile("" : "+v"(bl));
            const u32x4 fl = {dq == 0 ? bl : 0u, dq == 1 ? bl : 0u, dq == 2 ? bl : 0u, dq == 3 ? bl : 0u};
            af.h = __builtin_bit_cast(bf16x8, fr);
            af.l = __builtin_bit_cast(bf16x8, fl);
        } else {
            af.v = __builtin_bit_cast(bf16x8, fr);
        }
        return af;
    };

    const int oy_b = band * p.band_rows, oy_e = min(p.Ho, oy_b + p.band_rows);
    const int ox0 = strip * p.TWo, tw = min(p.TWo, p.Wo - ox0);
    const int ix0 = ox0 * S - p.pad_l, iy_top = oy_b * S - p.pad_t;
    char* const ring_e = ring + frow * PXB;                   // expand store: pixel frow of a tile (channels 4*kg.. by row_store4)
    // Lane constants per tile, the same for every row.  Expand: column validity and byte offset inside an X row.  The loads
    // are UNCONDITIONAL (no exec branches between them and the waits): a pixel outside the strip / image reads a clamped,
    // valid pixel and is zeroed by the mask after the SiLU; the K tail beyond Cin of the last chunk reads the pixel's first
    // bytes against zero weights.  Depthwise: LDS read address and byte offset inside a Y row.
    constexpr int OOB = 0x7FFFFFF0;                          // beyond every buffer's num_records: load -> 0, store -> dropped
    float cmask[MT];
    int xoff[MT], xoffl[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int c = 16 * t + frow, ix = ix0 + c;
        const bool inside = c < p.IWs && ix >= 0 && ix < p.W;
        cmask[t] = inside ? 1.f : __builtin_inff();              // silu4_scaled's addend
        xoff[t] = inside ? ix * cbytes + kg * PB : OOB;
        xoffl[t] = (inside && (NKC - 1) * CHB + kg * PB < cbytes) ? ix * cbytes + (NKC - 1) * CHB + kg * PB : OOB;
    }
    // Depthwise ring reads: ONE lane base per kind of tap pair, every tile / tap / ring-phase offset an instruction immediate (no
    // address arithmetic in the row loop; round 4 - it was 1 + NO vector adds per pair).  A K = 32 operand holds two taps x 16
    // channels; lanes of the upper half (hi) read the pair's second tap.  Both taps in one window row: second tap = next pixel
    // (`dsame`).  A pair that straddles two window rows reads (row r, last column) | (row r + 1, column 0): the byte distance is
    // rowbytes - (KS-1) PXB while the two ring slots are consecutive (`dnw`, on the hi lanes) and (KS-1)(rowbytes + PXB) the other
    // way round when the window wraps in the ring (`dwr`, on the lo lanes; the immediate is then the second tap's offset).
    // Lanes beyond the strip's last output pixel are NOT redirected any more: they read finite ring contents (at most ROLL_PAD_PX
    // pixels past the last slot: the wave's zeroed pad) and their results are dropped (store out of range, pool mask 0).
    const char* const dl0 = ring + frow * S * PXB + (PAIR ? kg * 16 : (kg & 1) * (PXB / 2));
    const char* const dsame = dl0 + hi * PXB;
    const char* const dnw = dl0 + hi * (rowbytes - (KS - 1) * PXB);
    const char* const dwr = dl0 + (1 - hi) * (KS - 1) * (rowbytes + PXB);
    constexpr int TILEB = 16 * S * PXB;                          // ring bytes between the windows of two output tiles
    {   // the pad behind the last ring slot: read (never used) by lanes beyond the strip - must be finite
        constexpr int PADB = roll_pad_px(KS, S, MT, NO) * PXB;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int o = 0; o < PADB; o += 64 * 16)
                if (o + lane * 16 < PADB) *reinterpret_cast<u32x4*>(ring + j * RB1 + KS * rowbytes + o + lane * 16) = u32x4{0u, 0u, 0u, 0u};
    }
    int yoff[OTN];
#pragma unroll
    for (int u = 0; u < OTN; ++u) {
        const int oxl = 16 * u + frow;
        const bool ok = oxl < tw;
        // two-term: byte offset of the hi half of the lane's 4 channels inside their 8-channel group (lo: + 16)
        yoff[u] = ok ? (PAIR ? (ox0 + oxl) * mid * 4 + ((c0 + 4 * kg) >> 3) * 32 + ((c0 + 4 * kg) & 7) * 2 : ((ox0 + oxl) * mid + c0 + 4 * kg) * 2) : OOB;
    }
    // buffer descriptors of this image's X and Y (wave-uniform: kernel arguments and blockIdx only)
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.X)) + (long long)b * p.H * p.W * cbytes, 0, p.H * p.W * cbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(p.Y) + (long long)b * p.Ho * p.Wo * mid * (int)sizeof(T), 0, p.Ho * p.Wo * mid * (int)sizeof(T), 0x00020000);
    // One input row = MT x NKC operand fragments, loaded PFD output rows ahead of their use.  PFD = 2 where the registers allow
    // (bf16, stride 1, at most 4 fragments per row): round 4's ablation builds (profiles/r04_roll_ablation.txt) showed the wait for
    // the X rows - L2 hits, but fetched by every channel-tile wave and queued behind the previous row's output stores in the
    // in-order vmcnt - to be the largest single item of a row step (block 1.1: 0.288 ms, 0.176 without the X loads)
    constexpr int PFD = ROLL_PFD2 && !PAIR && S == 1 && MT * NKC <= 4 ? 2 : 1;
    Frag<T> xq[PFD * S][MT][NKC];
    auto load_row = [&](int rel, Frag<T> (&dst)[MT][NKC]) {
        int iy = iy_top + rel;
        iy = iy < 0 ? 0 : (iy >= p.H ? p.H - 1 : iy);             // rows outside the image: any valid row (zeroed by the row mask)
        const int rowoff = (ROLL_ABLATE & 32) ? 0 : iy * p.W * cbytes;      // wave-uniform: the buffer op's scalar offset  (ablation 32: every row = row 0, an L1 / L2 hit)
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int kc = 0; kc < NKC; ++kc) {
                const int xo = kc + 1 < NKC ? xoff[t] + kc * CHB : xoffl[t];
                const u32x4 v = (ROLL_ABLATE & 8) ? u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u} : __builtin_amdgcn_raw_buffer_load_b128(xrs, xo, rowoff, 0);
                if constexpr (PAIR) {
                    // (an out-of-range offset + 16 stays out of range: OOB is far beyond num_records)
                    const u32x4 v2 = __builtin_amdgcn_raw_buffer_load_b128(xrs, xo + 16, rowoff, 0);
                    dst[t][kc].h = __builtin_bit_cast(bf16x8, v);
                    dst[t][kc].l = __builtin_bit_cast(bf16x8, v2);
                } else {
                    dst[t][kc].v = __builtin_bit_cast(bf16x8, v);
                }
            }
    };
    auto expand_row = [&](int rel, int slot_bytes, const Frag<T> (&src)[MT][NKC]) {
        const int iy = iy_top + rel;
        const bool rowin = iy >= 0 && iy < p.H;                           // wave-uniform
        // rows outside the image: zeros (the padding applies to the EXPANDED map); pixels outside it: +inf in the SiLU's addend
        auto put = [&](int j, int t, const f32x4 v) {
            if constexpr (PAIR) *reinterpret_cast<f32x4*>(ring_e + j * RB1 + slot_bytes + 16 * PXB * t + kg * 16) = v;      // float32 ring
            else row_store4<T>(ring_e + j * RB1 + slot_bytes + 16 * PXB * t, 4 * kg, v);
        };
        if (rowin) {
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {                         // every channel tile multiplies the same X fragments
                    if (j >= njw) continue;
                    f32x4 acc = sh1[j];
#pragma unroll
                    for (int kc = 0; kc < ((ROLL_ABLATE & 2) ? 0 : NKC); ++kc) mma_chunk(wf[j][kc], src[t][kc], acc);
                    put(j, t, (ROLL_ABLATE & 16) ? (cmask[t] == 1.f ? acc : f32x4{0.f, 0.f, 0.f, 0.f}) : silu4_scaled(acc, cmask[t]));
                }
        } else {
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int j = 0; j < NJ; ++j) put(j, t, f32x4{0.f, 0.f, 0.f, 0.f});
        }
    };

    // 4 channels of one output pixel -> Y (range-checked: an out-of-range offset drops the store; OOB + 16 is out of range too)
    auto store_out = [&](const f32x4 ov, int yo, int yrow_) {
        if constexpr ((ROLL_ABLATE & 4) != 0) { if (ov[0] == 12345.678f) __builtin_amdgcn_raw_buffer_store_b32(1u, yrs, yo, yrow_, 0); return; }
        if constexpr (PAIR) {
            u32x2 oh, ol;
            pair_split4(ov, oh, ol);
            __builtin_amdgcn_raw_buffer_store_b64(oh, yrs, yo, yrow_, 0);
            __builtin_amdgcn_raw_buffer_store_b64(ol, yrs, yo + 16, yrow_, 0);
        } else {
            const bf16x4 ob = {(bf16_t)ov[0], (bf16_t)ov[1], (bf16_t)ov[2], (bf16_t)ov[3]};
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, ob), yrs, yo, yrow_, 0);
        }
    };
    float pl[NJ][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) pl[j][r] = 0.f;
    int next_rel = 0;
    // prologue: the first KS - S rows of the band's window (ring slots 0 .. KS-S-1), loaded and expanded on the spot
#pragma unroll 1
    for (; next_rel < KS - S; ++next_rel) {
        load_row(next_rel, xq[0]);
        expand_row(next_rel, next_rel * rowbytes, xq[0]);
    }
#pragma unroll
    for (int d = 0; d < PFD; ++d)
#pragma unroll
        for (int r = 0; r < S; ++r) load_row(next_rel + d * S + r, xq[d * S + r]);
    int yrow = oy_b * p.Wo * mid * (int)sizeof(T);            // byte offset of the output row inside the image (scalar offset)
    const int ypitch = p.Wo * mid * (int)sizeof(T);
    int oy = oy_b;
    // One output row.  PH = ring slot of the first row of its KS-row window.
    auto step = [&](auto PHC, auto SLC) {
        constexpr int PH = decltype(PHC)::value;
        constexpr int SL = decltype(SLC)::value % PFD;          // register slot of this step's rows (fetched PFD iterations ago)
        // ---- expand the S new input rows of this output row, then fetch the rows of the step PFD ahead into the slot just used
#pragma unroll
        for (int r = 0; r < S; ++r) expand_row(next_rel + r, ((PH + KS - S + r) % KS) * rowbytes, xq[SL * S + r]);
        next_rel += S;
#if ROLL_SYNC
        __builtin_amdgcn_s_barrier();                                // the workgroup's channel-tile waves ask for the same X row together
#endif
#pragma unroll
        for (int r = 0; r < S; ++r) load_row(next_rel + (PFD - 1) * S + r, xq[SL * S + r]);      // past the band's end: clamped rows, never used
        __builtin_amdgcn_sched_barrier(0);
        // ---- depthwise: one output row out of the ring
        constexpr int OT = OTN < 2 ? OTN : 2;                   // output tiles in flight together
        // lane base + immediate of a tap pair's ring read (ta, tb, offa, offb: constants after unrolling; a single last tap has
        // tb = ta: its hi lanes carry zero weights and read the next pixel)
        auto pair_addr = [&](int ta, int tb, int offa, int offb) -> const char* {
            if (ta / KS == tb / KS) return dsame + offa;
            return offb > offa ? dnw + offa : dwr + offb;
        };
        if constexpr (PAIR) {
            // float32 depthwise on the vector ALU: per tap one 16-byte ring read (4 channels of the lane's pixel) and two packed FMAs
            // per output tile; ring offsets are immediates (PH is a template constant), one window row of taps per batch
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
            if (j >= njw) continue;
            f32x4 acc[OTN];
#pragma unroll
            for (int u = 0; u < OTN; ++u) acc[u] = t2v[j];
#pragma unroll
            for (int dy = 0; dy < KS; ++dy) {
                f32x4 e[KS][OTN];
#pragma unroll
                for (int dx = 0; dx < KS; ++dx)
#pragma unroll
                    for (int u = 0; u < OTN; ++u)
                        e[dx][u] = *reinterpret_cast<const f32x4*>(dl0 + j * RB1 + u * TILEB + ((PH + dy) % KS) * rowbytes + dx * PXB);
#pragma unroll
                for (int dx = 0; dx < KS; ++dx)
#pragma unroll
                    for (int u = 0; u < OTN; ++u) acc[u] = e[dx][u] * wv[j][dy * KS + dx] + acc[u];
                __builtin_amdgcn_sched_barrier(0);              // rows stay rows: hoisting every read of the window would spill
            }
#pragma unroll
            for (int u = 0; u < OTN; ++u) {
                const f32x4 ov = roll_act(acc[u]);
                const int yo = yoff[u];
                const float vm = yo == OOB ? 0.f : 1.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) pl[j][r] += ov[r] * vm;
                store_out(ov, yo + 64 * j, yrow);               // channel tile j: two 8-channel groups of 32 bytes further
            }
            }
        } else if constexpr (KS == 5) {
            // 5 x 5 (round 3, as in mbconv_wide.hip): every tap pair is expanded ONCE per row and applied to all NO tiles (the row used
            // to be walked in groups of two tiles, each re-expanding the 13 diagonal operands); the B operands of a batch of G pairs x
            // NO tiles are requested before the batch's first MFMA and the operands are expanded while they travel; pairs whose two
            // taps lie in one window row read at an immediate offset from `dl + hi * 32` (the second tap is the next pixel), only the
            // two pairs that straddle rows select between two offsets
            f32x4 acc[OTN];
#pragma unroll
            for (int u = 0; u < OTN; ++u) acc[u] = t2v[0];
            constexpr int G = 8 / OTN < NPAIR ? 8 / OTN : NPAIR;      // (two-term: the same count of twice as large fragments, with twice the registers)      // 8 B-operand fragments in flight: the X rows prefetched for the next step keep 24 - 32 registers
#pragma unroll
            for (int p0 = 0; p0 < NPAIR; p0 += G) {
                Frag<T> bq[G][OTN];
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const int pr = p0 + g;
                    if (pr < NPAIR) {
                        const int ta = 2 * pr, tb = 2 * pr + 1 < NTAP ? 2 * pr + 1 : 2 * pr;        // constants after unrolling
                        const int offa = ((PH + ta / KS) % KS) * rowbytes + (ta % KS) * PXB;
                        const int offb = ((PH + tb / KS) % KS) * rowbytes + (tb % KS) * PXB;
#pragma unroll
                        for (int u = 0; u < OTN; ++u) bq[g][u] = ld_frag<T>(pair_addr(ta, tb, offa, offb) + u * TILEB);
                    }
                }
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const int pr = p0 + g;
                    if (pr < NPAIR) {
                        const Frag<T> af = diag(pr);
#pragma unroll
                        for (int u = 0; u < ((ROLL_ABLATE & 1) ? 0 : OTN); ++u) mma_chunk(af, bq[g][u], acc[u]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);              // batches stay batches: hoisting every read of the row would spill
            }
#pragma unroll
            for (int u = 0; u < OTN; ++u) {
                const f32x4 ov = roll_act(acc[u]);
                const int yo = yoff[u];
                const float vm = yo == OOB ? 0.f : 1.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) pl[0][r] += ov[r] * vm;
                store_out(ov, yo, yrow);
            }
        } else {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int u0 = 0; u0 < OTN; u0 += OT) {
            if (j < njw) {
                f32x4 acc[OT];
#pragma unroll
                for (int u = 0; u < OT; ++u) acc[u] = t2v[j];
#pragma unroll
                for (int pr = 0; pr < NPAIR; ++pr) {
                    const Frag<T> af = diag(pr, j);
                    const int ta = 2 * pr, tb = 2 * pr + 1 < NTAP ? 2 * pr + 1 : 2 * pr;        // constants after unrolling
                    const int offa = ((PH + ta / KS) % KS) * rowbytes + (ta % KS) * PXB, offb = ((PH + tb / KS) % KS) * rowbytes + (tb % KS) * PXB;
                    const char* const src = pair_addr(ta, tb, offa, offb) + j * RB1;
#pragma unroll
                    for (int u = 0; u < OT; ++u) {
                        if (u0 + u < OTN && !(ROLL_ABLATE & 1)) mma_chunk(af, ld_frag<T>(src + (u0 + u) * TILEB), acc[u]);
                    }
                    // keep the scheduler from hoisting all 2 x 13 ring reads (4 registers each) to the top of the row
                    if constexpr (KS == 5) { if (pr % 4 == 3) __builtin_amdgcn_sched_barrier(0); }
                }
#pragma unroll
                for (int u = 0; u < OT; ++u) {
                    if (u0 + u < OTN) {
                        const f32x4 ov = roll_act(acc[u]);
                        const int yo = yoff[u0 + u < OTN ? u0 + u : 0];
                        const float vm = yo == OOB ? 0.f : 1.f;
#pragma unroll
                        for (int r = 0; r < 4; ++r) pl[j][r] += ov[r] * vm;
                        store_out(ov, yo + 32 * j, yrow);            // channel tile j: + 16 channels (an out-of-range offset stays out of range)
                    }
                }
            }
        }
        }
        yrow += ypitch;
        ++oy;
    };
#pragma unroll 1
    while (oy < oy_e) {
        // KS ring phases x PFD register slots, unrolled so that both are compile-time constants in every step
        if constexpr (KS == 3) {
            step(IntC<0>{}, IntC<0>{});
            if (oy < oy_e) step(IntC<(S) % 3>{}, IntC<1>{});
            if (oy < oy_e) step(IntC<(2 * S) % 3>{}, IntC<2>{});
            if constexpr (PFD == 2) {
                if (oy < oy_e) step(IntC<0>{}, IntC<3>{});
                if (oy < oy_e) step(IntC<(S) % 3>{}, IntC<4>{});
                if (oy < oy_e) step(IntC<(2 * S) % 3>{}, IntC<5>{});
            }
        } else {
            step(IntC<0>{}, IntC<0>{});
            if (oy < oy_e) step(IntC<(S) % 5>{}, IntC<1>{});
            if (oy < oy_e) step(IntC<(2 * S) % 5>{}, IntC<2>{});
            if (oy < oy_e) step(IntC<(3 * S) % 5>{}, IntC<3>{});
            if (oy < oy_e) step(IntC<(4 * S) % 5>{}, IntC<4>{});
            if constexpr (PFD == 2) {
                if (oy < oy_e) step(IntC<0>{}, IntC<5>{});
                if (oy < oy_e) step(IntC<(S) % 5>{}, IntC<6>{});
                if (oy < oy_e) step(IntC<(2 * S) % 5>{}, IntC<7>{});
                if (oy < oy_e) step(IntC<(3 * S) % 5>{}, IntC<8>{});
                if (oy < oy_e) step(IntC<(4 * S) % 5>{}, IntC<9>{});
            }
        }
    }
    if (p.pool_partial != nullptr) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = pl[j][r];
                v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
                pl[j][r] = v;
            }
            if (frow == 0 && j < njw) {
                float* dst = p.pool_partial + ((long long)b * (p.nstrips * p.nbands) + band * p.nstrips + strip) * mid + c0 + 16 * j + 4 * kg;
                *reinterpret_cast<f32x4*>(dst) = f32x4{pl[j][0], pl[j][1], pl[j][2], pl[j][3]};
            }
        }
    }
}

struct RollGeometry { bool use; int TWo, nstrips, IWs, IWa, band_rows, nbands, wpg, ngroups, ring_bytes, nkc, nj; size_t lds; };
// Channel tiles per wave (kernel parameter NJ).  Measured at d0 / 640 / batch 64 (profiles/r04_roll_nj.txt): two tiles per wave take
// block 1.0 (320 x 320, 32 -> 96 channels, 3 x 3 / s2: 64 bytes of X per pixel) from 0.360 to 0.328 ms; the 16- and 24-channel inputs of
// the other 3 x 3 blocks gain nothing, three tiles per wave (128 registers, spills in the 48-pixel strips) lose 6 - 15 %.  So: two tiles
// where a stride-2 3 x 3 block reads at least 64 bytes per input pixel and the tile count is even; ROLL_NJ_MAX widens that for A/B builds.
#ifndef ROLL_NJ_MAX
#define ROLL_NJ_MAX 0
#endif
#ifndef ROLL_NJ_PAIR
#define ROLL_NJ_PAIR 1
#endif

// Geometry depends on the map and channel sizes only - never on the batch - so that an image's result (including the order
// in which its SE pool partials are summed) is the same at every batch size.
RollGeometry pick_roll(int H, int W, int Cin, int mid, int k, int stride, bool pair = false) {
    RollGeometry g{};
    g.use = false;
    g.nkc = (Cin + 31) / 32;                                    // K-chunks of 32 channels (64 bytes; two-term bf16: 128 bytes)
    // Inputs wider than 64 channels stay with mbconv.hip's band x channel-slice form.  Both ways of feeding such a layer to
    // 16-channel waves were built and measured on d0 (40 x 40 and 20 x 20 maps, round 2): X through each wave's own registers
    // (Cin / 32 fragments per pixel tile: no registers left to prefetch, 3 waves per SIMD) ran 1.3 - 2x slower, and X staged
    // once per workgroup in LDS with a barrier per row ran within +-10 % of that form - those small maps simply do not have
    // enough rows x channel tiles to fill the chip with row-streaming waves - so neither is kept.
    if (g.nkc > 2 || mid % 16 || Cin % 8) return g;
    const int Ho = same_out(H, stride), Wo = same_out(W, stride);
    const int npair = (k * k + 1) / 2;
    // waves x KS rows x IWa x 32 B must fit the CU's LDS.  (two-term: 64 B per ring pixel and twice the registers per prefetched X
    // fragment - strips of at most 48 input pixels keep the 3 x 3 kernels at three waves per SIMD)
    const int iwa_max = (k == 5 || pair) ? 48 : 64;
    long long best = -1;
    for (int ns = 1; ns <= (Wo + 7) / 8; ++ns) {
        const int two = (Wo + ns - 1) / ns;
        if ((ns - 1) * two >= Wo) continue;
        const int iws = (two - 1) * stride + k, iwa = iws <= 32 ? 32 : (iws + 15) / 16 * 16;
        if (iwa > iwa_max) continue;
        // the prefetched rows (stride x IWa/16 tiles x K-chunks fragments) must fit the register budget without spills
        const int mt = iwa / 16;
        if (pair && k == 5 && g.nkc > 1 && mt >= 3) continue;     // (two-term, 25 resident tap vectors + two-term X fragments: spills at 256 registers)
        if (stride == 1) {
            if (mt == 4 && g.nkc > 1 && k == 5) continue;
        } else {
            if (mt >= 3 && (g.nkc > 1 || (mt == 4 && k == 5))) continue;
        }
        // rough issue cycles per output row of the strip set: expand tiles (MFMAs + epilogue) + depthwise tiles
        // (round 4: constants re-derived from the row loop's listing after its diet - an expand tile is 1 MFMA per K chunk + 8 exp / rcp +
        // packed multiplies / converts ~ 110 cycles, a 3 x 3 depthwise tile 5 MFMAs + the same SiLU + pool / store ~ 155; the old 64 : 136
        // put block 1.1 on five 48-pixel strips, 0.288 ms, where three 64-pixel strips run 0.262: profiles/r04_roll_nj.txt)
        long long cost = (long long)ns * ((iwa / 16) * stride * (g.nkc * 16 + 94) + ((two + 15) / 16) * (npair * 19 + 60));
#ifdef ROLL_PAIR_WIDE    /* experiment (variant builds only): two-term mode takes the widest strips its registers allow - measured
                            round 4: block 1.0 0.794 ms either way (28 or 44 strips x bands per image), 2.0 0.38 vs 0.30 ms */
        if (pair) cost = ns;
#endif
        if (best < 0 || cost < best) { best = cost; g.TWo = two; g.nstrips = ns; g.IWs = iws; g.IWa = iwa; }
    }
    if (best < 0) return g;
    g.nj = 1;
    if (k == 3 && g.nkc == 1) {
        // two-term mode (twice the X bytes per pixel): two tiles per wave wherever there are at least two (3 x 3, one K chunk), an odd
        // count leaves the last wave of a strip with one; bf16: stride-2 blocks reading >= 64 bytes per input pixel
        if (ROLL_NJ_MAX == 0) { if (pair ? (ROLL_NJ_PAIR && mid / 16 >= 2) : (stride == 2 && Cin >= 32 && (mid / 16) % 2 == 0)) g.nj = 2; }
        else for (int n = 2; n <= ROLL_NJ_MAX; ++n) if ((mid / 16) % n == 0) g.nj = n;
    }
    // per wave: nj x (KS slots + the zeroed pad)
    g.ring_bytes = g.nj * (k * g.IWa + roll_pad_px(k, stride, g.IWa / 16, (g.TWo + 15) / 16)) * (pair ? 64 : 32);
    // waves per workgroup: a divisor of the wave count per strip that packs the CU's 16 wave slots
    const int tiles = (mid / 16 + g.nj - 1) / g.nj;              // waves per strip
    int bestfill = -1;
    for (int d = 1; d <= 8; ++d) {
        if (tiles % d) continue;
        const int slots = (pair && g.nj > 1) ? 8 : 16;            // (two tiles per wave in two-term mode: 170 - 250 registers, two waves per SIMD)
        const int fill = (slots / d) * d;
        if (fill > bestfill || (fill == bestfill && d > g.wpg)) { bestfill = fill; g.wpg = d; }
    }
    g.ngroups = tiles / g.wpg;
    g.nbands = Ho / 40 > 0 ? Ho / 40 : 1;
    g.band_rows = (Ho + g.nbands - 1) / g.nbands;
    g.nbands = (Ho + g.band_rows - 1) / g.band_rows;
    g.lds = (size_t)g.wpg * g.ring_bytes;
    if (g.lds > 160 * 1024) return g;                           // (cannot happen for the strip widths above; a launch would fail)
    g.use = true;
    return g;
}

template <int KS, int S, int NKC, typename T, int NJ = 1>
void (*roll_kernel_for(int mt, int no))(RollArgs) {
    // MT = ceil(IWs / 16) input tiles, NO = ceil(TWo / 16) output tiles: stride 1 -> NO in {MT - 1, MT}; stride 2 -> MT in {2 NO - 1 .. 2 NO + 1}
    if constexpr (S == 1) {
        if (mt == 2) return no == 1 ? mbconv_roll_kernel<KS, S, NKC, 2, 1, T, NJ> : no == 2 ? mbconv_roll_kernel<KS, S, NKC, 2, 2, T, NJ> : nullptr;
        if constexpr (NKC <= 4) {
            if (mt == 3) return no == 2 ? mbconv_roll_kernel<KS, S, NKC, 3, 2, T, NJ> : no == 3 ? mbconv_roll_kernel<KS, S, NKC, 3, 3, T, NJ> : nullptr;
        }
        if constexpr (NKC <= 3 && KS == 3) {
            if (mt == 4) return no == 3 ? mbconv_roll_kernel<KS, S, NKC, 4, 3, T, NJ> : no == 4 ? mbconv_roll_kernel<KS, S, NKC, 4, 4, T, NJ> : nullptr;
        }
    } else {
        if constexpr (NKC <= 4) {
            if (mt == 2) return no == 1 ? mbconv_roll_kernel<KS, S, NKC, 2, 1, T, NJ> : nullptr;
        }
        if constexpr (NKC == 1) {
            if (mt == 3) return no == 1 ? mbconv_roll_kernel<KS, S, NKC, 3, 1, T, NJ> : no == 2 ? mbconv_roll_kernel<KS, S, NKC, 3, 2, T, NJ> : nullptr;
            if constexpr (KS == 3) {
                if (mt == 4) return no == 2 ? mbconv_roll_kernel<KS, S, NKC, 4, 2, T, NJ> : nullptr;
            }
        }
    }
    return nullptr;
}

template <int KS, int S, typename T>
int launch_roll_ks(hipStream_t st, const RollArgs& r, const RollGeometry& g) {
    void (*kern)(RollArgs) = nullptr;
    const int mt = g.IWa / 16, no = (g.TWo + 15) / 16;
    switch (g.nkc) {
        case 1:
            if constexpr (KS == 3) {
                if (g.nj == 2) { kern = roll_kernel_for<KS, S, 1, T, 2>(mt, no); break; }
#if ROLL_NJ_MAX >= 3
                if (g.nj == 3) { kern = roll_kernel_for<KS, S, 1, T, 3>(mt, no); break; }
#endif
            }
            if (g.nj != 1) return EFFDET_EINVAL;
            kern = roll_kernel_for<KS, S, 1, T>(mt, no); break;
        case 2: kern = roll_kernel_for<KS, S, 2, T>(mt, no); break;
        default: break;
    }
    if (kern == nullptr) return EFFDET_EINVAL;
    if (g.lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return EFFDET_ELAUNCH;
    }
    const int rounds = (r.B + 7) / 8;
    hipLaunchKernelGGL(kern, dim3(rounds * r.per_image * 8), dim3(g.wpg * 64), g.lds, st, r);
    return effdet_check_launch();
}

}  // namespace

// internal (not part of the C ABI): used by mbconv.hip's launcher
int effdet_mbconv_roll_parts(int H, int W, int Cin, int mid, int k, int stride, int pair) {
    const RollGeometry g = pick_roll(H, W, Cin, mid, k, stride, pair != 0);
    return g.use ? g.nstrips * g.nbands : 0;
}

int effdet_mbconv_roll_launch(hipStream_t st, const void* X, const float* in_gate, void* Y, const void* W1, const float* s1, const float* t1,
                              const float* taps, const float* s2, const float* t2, float* pool_partial,
                              int B, int H, int W, int Cin, int mid, int k, int stride, int pair, int sym) {
    const RollGeometry g = pick_roll(H, W, Cin, mid, k, stride, pair != 0);
    if (!g.use) return EFFDET_EINVAL;
    RollArgs r{X, Y, W1, in_gate, s1, t1, taps, s2, t2, pool_partial, B, H, W, Cin, mid, same_out(H, stride), same_out(W, stride),
               pad_before(H, k, stride, sym), pad_before(W, k, stride, sym), g.TWo, g.nstrips, g.band_rows, g.nbands, g.IWs,
               g.wpg, g.ngroups, g.ring_bytes, g.nstrips * g.nbands * g.ngroups};
    if (pair) {
        if (k == 3) return stride == 1 ? launch_roll_ks<3, 1, bf16p_t>(st, r, g) : launch_roll_ks<3, 2, bf16p_t>(st, r, g);
        return stride == 1 ? launch_roll_ks<5, 1, bf16p_t>(st, r, g) : launch_roll_ks<5, 2, bf16p_t>(st, r, g);
    }
    if (k == 3) return stride == 1 ? launch_roll_ks<3, 1, bf16_t>(st, r, g) : launch_roll_ks<3, 2, bf16_t>(st, r, g);
    return stride == 1 ? launch_roll_ks<5, 1, bf16_t>(st, r, g) : launch_roll_ks<5, 2, bf16_t>(st, r, g);
}
