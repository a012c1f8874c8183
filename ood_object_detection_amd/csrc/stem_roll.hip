// Rolling-window form of the fused network entry (bfloat16 throughput mode):
//
//   conv_stem 3x3 / s2 (TF-SAME) + bn1 + SiLU  ->  blocks.0.0.conv_dw 3x3 / s1 + bn1 + SiLU  (+ SE pool partial sums)
//
// replaces timm's conv_stem / bn1 / act1 and blocks.0.0.conv_dw / bn1 / act1 (reached from effdet/efficientdet.py:837), like
// stem_dw.hip, with the structure of mbconv_roll.hip: every WAVE is an autonomous streaming engine without workgroup
// barriers.  A wave owns (image, band of output rows, column strip) and ALL NJ = C / 16 channel tiles (the im2col operand is
// the expensive part: it is built once and used for every tile).  Per output row it
//   input:    fetches the 2 new rows of the 3 input planes (NCHW float32 / bfloat16 / raw uint8 with the loader's normalisation)
//             one output row ahead, and interleaves them as [px][c0 c1 c2 0] bf16 (8 bytes per pixel) into a private 4-row LDS ring;
//   stem:     one new stem row on the matrix cores: per kernel row ky the 3 taps x 4 (padded) channels of a pixel are 24
//             contiguous bytes of that image, so with K ordered (ky, kx, c4) a lane's MFMA operand is ONE 16-byte LDS read;
//             two MFMAs (ky 0 | 1, ky 2 | -) per 16 pixels x 16 channels, BN scale folded into the weights, shift as the
//             initial accumulator, SiLU, zero outside the stem map, 8-byte store into the tile's 3-row ring;
//   depthwise: the output row from the rings (diag(w[t0]) | diag(w[t1]) operands, mbconv_roll.hip), BN shift as the initial
//             accumulator, SiLU, pool sums, range-checked 8-byte buffer stores.
// Every input row is read from HBM once per strip (the tile form re-read a 37 x 37 patch per 16 x 16 outputs: 4.4x).
// Needs an even pad_l (even W) and C = 32 (the b0 .. b2 stems); anything else takes stem_dw.hip's tile form.
#include "common.h"
#include <type_traits>

namespace {

struct SrArgs {
    const void* X; const void* Wk;                // X NCHW; Wk [C][32] bf16 with k = (ky*3 + kx)*3 + ci (the engine's layout)
    float nmean[3], nstd[3];
    const float* s1; const float* t1; const float* taps; const float* s2; const float* t2;
    void* Y; float* pool_partial;
    int B, H, W, C, Ho, Wo, pad_t, pad_l;
    int TWo, nstrips, band_rows, nbands, wpg, per_image;
    int jsplit;                                   // channel-tile groups per (band, strip): 1 = a wave owns all C / 16 tiles; C / 16 = one tile per wave
};

typedef float f32x2r __attribute__((ext_vector_type(2)));
DEV f32x4 silu4r(const f32x4 x) {
    const f32x2r x0 = {x[0], x[1]}, x1 = {x[2], x[3]};
    const f32x2r t0 = x0 * -1.4426950408889634f, t1 = x1 * -1.4426950408889634f;
    const f32x2r d0 = f32x2r{__builtin_amdgcn_exp2f(t0[0]), __builtin_amdgcn_exp2f(t0[1])} + 1.0f;
    const f32x2r d1 = f32x2r{__builtin_amdgcn_exp2f(t1[0]), __builtin_amdgcn_exp2f(t1[1])} + 1.0f;
    const f32x2r y0 = x0 * f32x2r{__builtin_amdgcn_rcpf(d0[0]), __builtin_amdgcn_rcpf(d0[1])};
    const f32x2r y1 = x1 * f32x2r{__builtin_amdgcn_rcpf(d1[0]), __builtin_amdgcn_rcpf(d1[1])};
    return f32x4{y0[0], y0[1], y1[0], y1[1]};
}

template <int V> struct IntS { static constexpr int value = V; };

constexpr int SR_TW = 30;                   // output columns per strip: 32 stem columns with the halo = 2 pixel tiles
constexpr int SR_IPX = 72;                  // input pixels staged per row (2 * 32 + 1 = 65 used), 8 bytes each
constexpr int SR_IROW = SR_IPX * 8;         // 576 bytes
constexpr int SR_PAD_PX = 4;                // ring pixels a lane beyond the strip reads past its row (2 taps + the single last tap's partner)
// IN: 0 float32, 1 bfloat16, 2 uint8 (normalised on the fly).  NJ = C / 16 channel tiles.
// T: bf16_t, or bf16p_t (dtype 2, two-term bf16: the staged image, the stem weights, the stem ring and Y carry hi + lo and every
// MFMA becomes three; Wk is then plain float32 [C][32]; the image's hi and lo planes are staged as two sets of input rows)
template <int IN, int NJ, typename T>
__global__ __launch_bounds__(512, (IsPair<T>::value ? 2 : 4)) void stem_roll_kernel(SrArgs p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr bool PAIR = IsPair<T>::value;
    constexpr int PXB = 16 * (int)sizeof(T);        // bytes of a ring pixel (16 channels)
    constexpr int SR_ROWB = 32 * PXB;               // one stem row of a channel tile in its ring
    constexpr int SR_RING = 3 * SR_ROWB;            // per channel tile: 3 stem rows x 32 px x 16 ch
    constexpr int SR_PADB = SR_PAD_PX * PXB;        // zeroed pixels behind the wave's last ring (read, never used, by lanes beyond the strip)
    // the stem GEMM produces t = -log2(e) x directly (silu4_scaled, common.h: scale in the weights / BN shift, -ln 2 in the taps)
    constexpr float ESC = -1.4426950408889634f, EINV = -0.6931471805599453f;
    constexpr int IPL = PAIR ? 2 : 1;               // staged image planes (hi | hi, lo)
    // readfirstlane: the wave index is wave-uniform, but the compiler cannot know that of threadIdx.x >> 6 - and this kernel's band
    // and strip (hence the scalar offsets of every buffer load and store) derive from it.  Left as a vector value each buffer
    // operation was wrapped in a waterfall loop (readfirstlane + exec mask + branch) and waited for on the spot: the input rows
    // "fetched three steps ahead" were in fact loaded one dword at a time with `s_waitcnt vmcnt(0)` after each (rounds 1 - 2).
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int frow = lane & 15, kg = lane >> 4;
    const int xcd = blockIdx.x & 7, rr_ = blockIdx.x >> 3;
    const int b = (rr_ / p.per_image) * 8 + xcd;
    if (b >= p.B) return;
    // (band, strip[, channel-tile group]) of this wave
    int q = (rr_ % p.per_image) * p.wpg + wave;
    const int jb = q % p.jsplit;                                    // first channel tile of this wave
    q /= p.jsplit;
    const int strip = q % p.nstrips, band = q / p.nstrips;
    if (band >= p.nbands) return;                                   // waves are autonomous: no barrier follows
    const int C = p.C;
    char* const wl = lds + wave * (IPL * 4 * SR_IROW + NJ * SR_RING + SR_PADB);
    char* const irows = wl;                                         // [IPL][4][SR_IPX][4] bf16 interleaved input rows
    char* const rings = wl + IPL * 4 * SR_IROW;                     // [NJ][3][32 px][16 ch]
    constexpr int ILO = 4 * SR_IROW;                                // two-term: the lo plane's rows follow the hi plane's

    // ---- constants.  Stem weights as MFMA A operands: chunk 0 = (ky 0 | ky 1), chunk 1 = (ky 2 | zero); inside a ky the 16
    // K slots are (kx, c4) = 12 real values + 4 zeros; the lane (row m = frow, pieces of 8 K) holds K = 8*kg .. 8*kg + 7.
    Frag<T> wf[NJ][2];
    f32x4 sh1[NJ], t2v[NJ];
    unsigned abits[NJ][5];
    f32x4 wv[PAIR ? NJ : 1][PAIR ? 9 : 1];          // two-term mode: float32 taps (BN2 scale folded in) of the lane's 4 channels, see mbconv_roll.hip
    const int hi = kg >> 1;
    const bool dactive = (kg & 1) == (frow >> 3);
    const int dq = (frow & 7) >> 1;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int ch = 16 * (j + jb) + frow;
        const float rs1 = p.s1[ch], rs2 = p.s2[ch];
        typedef typename std::conditional<PAIR, float, bf16_t>::type WT;
        const WT* wrow = reinterpret_cast<const WT*>(p.Wk) + (long long)ch * 32;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int ky = 2 * c + (kg >> 1);                       // kernel row of this lane's half of the chunk
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int slot = 8 * (kg & 1) + e;                  // 0..15 inside the ky: kx = slot / 4, ci = slot % 4
                const int kx = slot >> 2, ci = slot & 3;
                const bool real = ky < 3 && kx < 3 && ci < 3;
                const float w = real ? (float)wrow[(ky * 3 + kx) * 3 + ci] * (rs1 * ESC) : 0.f;
                if constexpr (PAIR) {
                    const bf16_t wh = (bf16_t)w;
                    wf[j][c].h[e] = wh;
                    wf[j][c].l[e] = (bf16_t)(w - (float)wh);
                } else {
                    wf[j][c].v[e] = (bf16_t)w;
                }
            }
        }
        if constexpr (PAIR) {
            const f32x4 s2q = *reinterpret_cast<const f32x4*>(p.s2 + 16 * (j + jb) + 4 * kg);
#pragma unroll
            for (int t = 0; t < 9; ++t) wv[j][t] = *reinterpret_cast<const f32x4*>(p.taps + (long long)t * C + 16 * (j + jb) + 4 * kg) * (s2q * EINV);
        }
        sh1[j] = *reinterpret_cast<const f32x4*>(p.t1 + 16 * (j + jb) + 4 * kg) * ESC;
        t2v[j] = *reinterpret_cast<const f32x4*>(p.t2 + 16 * (j + jb) + 4 * kg);
#pragma unroll
        for (int pr = 0; pr < 5; ++pr) {
            const int t = 2 * pr + hi;
            const bool on = dactive && t < 9;
            const float wv = on ? p.taps[(long long)(t < 9 ? t : 0) * C + ch] * (rs2 * EINV) : 0.f;
            abits[j][pr] = (unsigned)__builtin_bit_cast(unsigned short, (bf16_t)wv) << (16 * (frow & 1));
        }
    }
    const int oy_b = band * p.band_rows, oy_e = min(p.Ho, oy_b + p.band_rows);
    const int ox0 = strip * p.TWo, tw = min(p.TWo, p.Wo - ox0);
    const int sx0 = ox0 - 1;                                        // stem column of ring column 0
    const int ix0 = 2 * sx0 - p.pad_l;                              // input column of staged pixel 0 (even)
    // lane constants.  Staging: lane l handles input pixels l and l + 64 (the 65th) of a row.
    constexpr int OOB = 0x7FFFFFF0;
    const int esz = IN == 0 ? 4 : (IN == 1 ? 2 : 1);
    float cmask[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int sx = sx0 + 16 * t + frow;
        cmask[t] = (sx >= 0 && sx < p.Wo) ? 1.f : __builtin_inff();      // silu4_scaled's addend: +inf -> 0 outside the stem map
    }
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.X)) + (long long)b * 3 * p.H * p.W * esz, 0, 3 * p.H * p.W * esz, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(p.Y) + (long long)b * p.Ho * p.Wo * C * (int)sizeof(T), 0, p.Ho * p.Wo * C * (int)sizeof(T), 0x00020000);
    const int plane = p.H * p.W * esz;
    // depthwise ring reads: three lane bases, every tile / tap / phase / channel-tile offset an immediate (mbconv_roll.hip); lanes
    // beyond the strip read finite ring contents (behind the wave's last ring: the zeroed pad) and are dropped at the store
    const char* const dl0 = rings + frow * PXB + (PAIR ? kg * 16 : (kg & 1) * (PXB / 2));      // (two-term: a float32 ring, 4 channels per lane)
    const char* const dsame = dl0 + hi * PXB;
    const char* const dnw = dl0 + hi * (SR_ROWB - 2 * PXB);
    const char* const dwr = dl0 + (1 - hi) * 2 * (SR_ROWB + PXB);
    if (lane * 16 < SR_PADB) *reinterpret_cast<u32x4*>(rings + NJ * SR_RING + lane * 16) = u32x4{0u, 0u, 0u, 0u};
    static_assert(SR_PADB <= 64 * 16, "one 16-byte store per lane zeroes the pad");
    int yoff[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int oxl = 16 * u + frow;
        const bool ok = oxl < tw;
        // (two-term: byte offset of the hi half of channels 4 kg .. inside their 8-channel group; channel tile j adds 64 bytes)
        yoff[u] = ok ? (PAIR ? (ox0 + oxl) * C * 4 + ((4 * kg) >> 3) * 32 + ((4 * kg) & 7) * 2 : ((ox0 + oxl) * C + 4 * kg) * 2) : OOB;
    }
    char* const ring_e = rings + frow * PXB;
    // im2col operand address inside a staged input row: pixel 2 * (16 t + frow) + 2 * (kg & 1), i.e. 16-byte aligned
    const int xf_lane = (2 * frow + 2 * (kg & 1)) * 8;

    // One input row = 3 planes, fetched THREE output rows ahead (an HBM round trip is several output rows of work long): lanes 0 .. 32
    // load TWO adjacent pixels per plane and row - a dword of bfloat16, a dwordx2 of float32 or a 16-bit pair of uint8 (ix0, W and
    // pad_l are even, so a pair never straddles the image border) - 6 loads per step, and commit them with one 16-byte LDS store
    // per lane (two-term mode: one per term).  Raw values wait in registers; uint8 is normalised at the commit.
    // (round 4: float32 / uint8 inputs used to take 12 scalar loads per step, one step ahead: 0.42 / 0.40 ms against 0.26 ms of
    // the bfloat16 input at d0 / 640 / batch 64)
    constexpr int PFD = 3;
    u32x2 rawq[PFD][2][3];                                            // .x (and .y for float32): the pair's raw bits
    const bool dvok = lane < 33 && ix0 + 2 * lane >= 0 && ix0 + 2 * lane < p.W;
    const int dvoff = dvok ? (ix0 + 2 * lane) * esz : OOB;
    auto fetch_rows = [&](int iy0, auto SLC) {                        // input rows iy0, iy0 + 1 -> register slot SL
        constexpr int SL = decltype(SLC)::value;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            int iy = iy0 + r;
            iy = iy < 0 ? 0 : (iy >= p.H ? p.H - 1 : iy);             // rows outside the image: any valid row, zeroed at the commit
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) {
                const int so = ci * plane + iy * p.W * esz;
                if constexpr (IN == 0) rawq[SL][r][ci] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(xrs, dvoff, so, 0));
                else if constexpr (IN == 1) rawq[SL][r][ci] = u32x2{__builtin_amdgcn_raw_buffer_load_b32(xrs, dvoff, so, 0), 0u};
                else rawq[SL][r][ci] = u32x2{(unsigned)__builtin_amdgcn_raw_buffer_load_b16(xrs, dvoff, so, 0), 0u};
            }
        }
    };
    auto commit_rows = [&](int iy0, auto SLC) {                       // raw -> interleaved bf16 pixels in ring slots (iy & 3)
        constexpr int SL = decltype(SLC)::value;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            char* dst = irows + ((iy0 + r) & 3) * SR_IROW;
            const bool rin = iy0 + r >= 0 && iy0 + r < p.H;           // rows / columns outside the image are TF-SAME zero padding of the INPUT
            if constexpr (IN == 1) {
                // {c0 c1} | {c2 0} of pixel 2*lane, then of pixel 2*lane + 1
                const unsigned m = rin ? 0xFFFFFFFFu : 0u;
                const unsigned d0 = rawq[SL][r][0][0] & m, d1 = rawq[SL][r][1][0] & m, d2 = rawq[SL][r][2][0] & m;
                const u32x4 v = {(d0 & 0xFFFFu) | (d1 << 16), d2 & 0xFFFFu, (d0 >> 16) | (d1 & 0xFFFF0000u), d2 >> 16};
                if (lane < 36) *reinterpret_cast<u32x4*>(dst + lane * 16) = v;
            } else {
                float f[2][3];                                        // [pixel of the pair][plane]; uint8: zero AFTER normalisation too
                const bool valid = rin && dvok;
#pragma unroll
                for (int ci = 0; ci < 3; ++ci) {
                    float v0, v1;
                    if constexpr (IN == 0) {
                        // (elements copied to scalars first: __builtin_bit_cast applied to an ext-vector element lvalue reads element 0)
                        const u32x2 q2 = rawq[SL][r][ci];
                        const unsigned a0 = q2[0], a1 = q2[1];
                        v0 = __builtin_bit_cast(float, a0);
                        v1 = __builtin_bit_cast(float, a1);
                    } else {
                        const float mean = ci == 0 ? p.nmean[0] : ci == 1 ? p.nmean[1] : p.nmean[2];
                        const float sd = ci == 0 ? p.nstd[0] : ci == 1 ? p.nstd[1] : p.nstd[2];
                        v0 = ((float)(rawq[SL][r][ci][0] & 0xFFu) - mean) / sd;
                        v1 = ((float)((rawq[SL][r][ci][0] >> 8) & 0xFFu) - mean) / sd;
                    }
                    f[0][ci] = valid ? v0 : 0.f;
                    f[1][ci] = valid ? v1 : 0.f;
                }
                const bf16x8 vh = {(bf16_t)f[0][0], (bf16_t)f[0][1], (bf16_t)f[0][2], (bf16_t)0.f, (bf16_t)f[1][0], (bf16_t)f[1][1], (bf16_t)f[1][2], (bf16_t)0.f};
                if (lane < 36) *reinterpret_cast<bf16x8*>(dst + lane * 16) = vh;
                if constexpr (PAIR) {
                    const bf16x8 vl = {(bf16_t)(f[0][0] - (float)vh[0]), (bf16_t)(f[0][1] - (float)vh[1]), (bf16_t)(f[0][2] - (float)vh[2]), (bf16_t)0.f,
                                       (bf16_t)(f[1][0] - (float)vh[4]), (bf16_t)(f[1][1] - (float)vh[5]), (bf16_t)(f[1][2] - (float)vh[6]), (bf16_t)0.f};
                    if (lane < 36) *reinterpret_cast<bf16x8*>(dst + ILO + lane * 16) = vl;
                }
            }
        }
    };
    // stem row sy -> ring slot `slot` of every channel tile (zeros when sy is outside the stem map: the depthwise conv pads IT)
    auto stem_row = [&](int sy, int slot_bytes) {
        const int iy0 = 2 * sy - p.pad_t;                             // first of its three input rows
        const bool rowin = sy >= 0 && sy < p.Ho;                        // wave-uniform
        auto put = [&](int j, int t, const f32x4 v) {
            if constexpr (PAIR) *reinterpret_cast<f32x4*>(ring_e + j * SR_RING + slot_bytes + 16 * PXB * t + kg * 16) = v;
            else row_store4<T>(ring_e + j * SR_RING + slot_bytes + 16 * PXB * t, 4 * kg, v);
        };
        if (!rowin) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int j = 0; j < NJ; ++j) put(j, t, f32x4{0.f, 0.f, 0.f, 0.f});
            return;
        }
        const char* r0 = irows + ((iy0 + (kg >> 1)) & 3) * SR_IROW + xf_lane;          // chunk 0: ky = kg >> 1 (0 | 1)
        const char* r1 = irows + ((iy0 + 2) & 3) * SR_IROW + xf_lane;                  // chunk 1: ky = 2 (upper half: zero weights)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            Frag<T> x0, x1;
            if constexpr (PAIR) {                                     // hi and lo planes of the staged image
                x0.h = *reinterpret_cast<const bf16x8*>(r0 + t * 256); x0.l = *reinterpret_cast<const bf16x8*>(r0 + ILO + t * 256);
                x1.h = *reinterpret_cast<const bf16x8*>(r1 + t * 256); x1.l = *reinterpret_cast<const bf16x8*>(r1 + ILO + t * 256);
            } else {
                x0 = ld_frag<T>(r0 + t * 256); x1 = ld_frag<T>(r1 + t * 256);
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                f32x4 acc = sh1[j];
                mma_chunk(wf[j][0], x0, acc);
                mma_chunk(wf[j][1], x1, acc);
                put(j, t, silu4_scaled(acc, cmask[t]));
            }
        }
    };

    float pl[NJ][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) pl[j][r] = 0.f;
    // ---- prologue: stem rows oy_b - 1 and oy_b (ring slots 0, 1).  Input rows of stem row s: 2s - pad_t .. + 2.
    const int s_first = oy_b - 1;
    {
        const int iyA = 2 * s_first - p.pad_t;
        fetch_rows(iyA, IntS<0>{}); commit_rows(iyA, IntS<0>{});
        fetch_rows(iyA + 2, IntS<0>{}); commit_rows(iyA + 2, IntS<0>{});
        stem_row(s_first, 0);
        fetch_rows(iyA + 3, IntS<0>{}); commit_rows(iyA + 3, IntS<0>{});   // (row iyA + 3 again with iyA + 4: rows are committed in pairs)
        stem_row(s_first + 1, SR_ROWB);
    }
    int snext = s_first + 2;                                          // next stem row to compute
    // the two new input rows of the next three stem rows (the first row of each was committed with its predecessor)
    fetch_rows(2 * snext - p.pad_t + 1, IntS<0>{});
    if constexpr (PFD == 3) {
        fetch_rows(2 * (snext + 1) - p.pad_t + 1, IntS<1>{});
        fetch_rows(2 * (snext + 2) - p.pad_t + 1, IntS<2>{});
    }
    int yrow = oy_b * p.Wo * C * (int)sizeof(T);
    const int ypitch = p.Wo * C * (int)sizeof(T);
    int oy = oy_b;
    auto step = [&](auto PHC) {
        constexpr int PH = decltype(PHC)::value;                      // ring slot of stem row oy - 1
        const int iyn = 2 * snext - p.pad_t + 1;
        constexpr int SLOT = PFD == 3 ? PH : 0;
        commit_rows(iyn, IntS<SLOT>{});                              // fetched PFD steps ago into register slot SLOT
        stem_row(snext, ((PH + 2) % 3) * SR_ROWB);
        ++snext;
        fetch_rows(2 * (snext + PFD - 1) - p.pad_t + 1, IntS<SLOT>{});   // rows of the step PFD ahead, into the slot just emptied
        __builtin_amdgcn_sched_barrier(0);
        auto pair_addr = [&](int ta, int tb, int offa, int offb) -> const char* {      // (mbconv_roll.hip)
            if (ta / 3 == tb / 3) return dsame + offa;
            return offb > offa ? dnw + offa : dwr + offb;
        };
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            f32x4 acc[2] = {t2v[j], t2v[j]};
            if constexpr (PAIR) {
                // float32 depthwise on the vector ALU (mbconv_roll.hip): one 16-byte ring read + two packed FMAs per tap and tile
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    f32x4 e[3][2];
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                        for (int u = 0; u < 2; ++u)
                            e[dx][u] = *reinterpret_cast<const f32x4*>(dl0 + u * 16 * PXB + j * SR_RING + ((PH + dy) % 3) * SR_ROWB + dx * PXB);
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                        for (int u = 0; u < 2; ++u) acc[u] = e[dx][u] * wv[j][dy * 3 + dx] + acc[u];
                }
            } else {
#pragma unroll
            for (int pr = 0; pr < 5; ++pr) {
                unsigned bits = abits[j][pr];
                asm volatile("" : "+v"(bits));        // expand the diagonal operand at use: NJ x 5 hoisted fragments would spill
                const u32x4 fr = {dq == 0 ? bits : 0u, dq == 1 ? bits : 0u, dq == 2 ? bits : 0u, dq == 3 ? bits : 0u};
                Frag<bf16_t> af;
                af.v = __builtin_bit_cast(bf16x8, fr);
                const int ta = 2 * pr, tb = 2 * pr + 1 < 9 ? 2 * pr + 1 : 2 * pr;
                const int offa = ((PH + ta / 3) % 3) * SR_ROWB + (ta % 3) * PXB, offb = ((PH + tb / 3) % 3) * SR_ROWB + (tb % 3) * PXB;
                const char* const src = pair_addr(ta, tb, offa, offb) + j * SR_RING;
#pragma unroll
                for (int u = 0; u < 2; ++u) mma_chunk(af, ld_frag<bf16_t>(src + u * 16 * PXB), acc[u]);
            }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const f32x4 ov = silu4r(acc[u]);
                const float vm = yoff[u] == OOB ? 0.f : 1.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) pl[j][r] += ov[r] * vm;
                if constexpr (PAIR) {                   // (an out-of-range offset + 16 is out of range too: the store is dropped)
                    u32x2 oh, ol;
                    pair_split4(ov, oh, ol);
                    const int yo = yoff[u] == OOB ? OOB : yoff[u] + 64 * (j + jb);
                    __builtin_amdgcn_raw_buffer_store_b64(oh, yrs, yo, yrow, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(ol, yrs, yo + 16, yrow, 0);
                } else {
                    const bf16x4 ob = {(bf16_t)ov[0], (bf16_t)ov[1], (bf16_t)ov[2], (bf16_t)ov[3]};
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, ob), yrs, yoff[u] == OOB ? OOB : yoff[u] + 32 * (j + jb), yrow, 0);
                }
            }
        }
        yrow += ypitch;
        ++oy;
    };
#pragma unroll 1
    while (oy < oy_e) {
        step(IntS<0>{});
        if (oy < oy_e) step(IntS<1>{});
        if (oy < oy_e) step(IntS<2>{});
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
            if (frow == 0) {
                float* dst = p.pool_partial + ((long long)b * (p.nstrips * p.nbands) + band * p.nstrips + strip) * C + 16 * (j + jb) + 4 * kg;
                *reinterpret_cast<f32x4*>(dst) = f32x4{pl[j][0], pl[j][1], pl[j][2], pl[j][3]};
            }
        }
    }
}

struct SrGeometry { bool use; int TWo, nstrips, band_rows, nbands, wpg, per_image; size_t lds; };

// map sizes only (never the batch): an image's result is the same at every batch size
SrGeometry pick_stem_roll(int H, int W, int C, bool pair = false, int sym = 0) {
    SrGeometry g{};
    g.use = false;
    if (C != 32) return g;                                            // 2 channel tiles (b0 .. b2 stems); wider stems spill at 128 registers
    if (pad_before(W, 3, 2, sym) % 2) return g;                       // the 16-byte operand reads need an even left pad (never with pad_type='')
    const int Ho = same_out(H, 2), Wo = same_out(W, 2);
    if (Wo < 16 || Ho < 8) return g;
    g.nstrips = (Wo + SR_TW - 1) / SR_TW;
    g.TWo = (Wo + g.nstrips - 1) / g.nstrips;
    g.nbands = Ho / 20 > 0 ? Ho / 20 : 1;                           // ~20-row bands: enough waves to fill the chip several times over
    g.band_rows = (Ho + g.nbands - 1) / g.nbands;
    g.nbands = (Ho + g.band_rows - 1) / g.band_rows;
    g.wpg = 4;
    // (one channel tile per wave - jsplit = C / 16 - was measured for the two-term mode, whose rings allow only two waves per SIMD
    // with all tiles in one wave: three waves per SIMD but every tile wave staging the input rows again ran 0.82 ms against 0.72)
    const int jsplit = 1;
    g.per_image = (g.nstrips * g.nbands * jsplit + g.wpg - 1) / g.wpg;
    g.lds = (size_t)g.wpg * ((pair ? 2 : 1) * 4 * SR_IROW + (C / 16 / jsplit) * 3 * 32 * (pair ? 64 : 32) + SR_PAD_PX * (pair ? 64 : 32));
    g.use = true;
    return g;
}

}  // namespace

int effdet_stem_roll_parts(int H, int W, int C, int pair, int sym) {
    const SrGeometry g = pick_stem_roll(H, W, C, pair != 0, sym);
    return g.use ? g.nstrips * g.nbands : 0;
}

int effdet_stem_roll_launch(hipStream_t st, int in_dtype, const void* X, const float* mean, const float* stdv, const void* Wk,
                            const float* s1, const float* t1, const float* taps, const float* s2, const float* t2,
                            void* Y, float* pool_partial, int B, int H, int W, int C, int pair, int sym) {
    const SrGeometry g = pick_stem_roll(H, W, C, pair != 0, sym);
    if (!g.use) return EFFDET_EINVAL;
    if (pair && in_dtype == 1) return EFFDET_EINVAL;                 // two-term mode takes float32 or raw uint8 images
    SrArgs a;
    a.X = X; a.Wk = Wk;
    for (int i = 0; i < 3; ++i) { a.nmean[i] = in_dtype == 2 ? mean[i] : 0.f; a.nstd[i] = in_dtype == 2 ? stdv[i] : 1.f; }
    a.s1 = s1; a.t1 = t1; a.taps = taps; a.s2 = s2; a.t2 = t2; a.Y = Y; a.pool_partial = pool_partial;
    a.B = B; a.H = H; a.W = W; a.C = C; a.Ho = same_out(H, 2); a.Wo = same_out(W, 2);
    a.pad_t = pad_before(H, 3, 2, sym); a.pad_l = pad_before(W, 3, 2, sym);
    a.TWo = g.TWo; a.nstrips = g.nstrips; a.band_rows = g.band_rows; a.nbands = g.nbands; a.wpg = g.wpg; a.per_image = g.per_image;
    a.jsplit = 1;
    void (*kern)(SrArgs) = nullptr;
    const int nj = C / 16;
#define SR_PICK(IN_, T_) (nj == 2 ? stem_roll_kernel<IN_, 2, T_> : nullptr)
    if (pair) kern = in_dtype == 0 ? SR_PICK(0, bf16p_t) : SR_PICK(2, bf16p_t);
    else kern = in_dtype == 0 ? SR_PICK(0, bf16_t) : in_dtype == 1 ? SR_PICK(1, bf16_t) : SR_PICK(2, bf16_t);
#undef SR_PICK
    if (g.lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return EFFDET_ELAUNCH;
    }
    if (kern == nullptr) return EFFDET_EINVAL;
    const int rounds = (B + 7) / 8;
    hipLaunchKernelGGL(kern, dim3(rounds * g.per_image * 8), dim3(g.wpg * 64), g.lds, st, a);
    return effdet_check_launch();
}
