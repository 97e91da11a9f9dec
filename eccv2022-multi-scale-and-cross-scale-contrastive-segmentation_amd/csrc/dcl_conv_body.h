// dcl_conv_body.h -- the tile body of the direct f16x3 convolution (conv_body) and what it needs, shared by the translation units
// that instantiate it: dcl_conv3x3.hip (the kernels of dcl_conv3x3_f16x3 / dcl_conv1x1_f16x3) and dcl_conv3x3_pre.hip (the same tiles
// with the producer norm's affine map + ReLU applied while the input patch is staged).  Design notes: head of dcl_conv3x3.hip.
#pragma once
#include <cstring>
#include <type_traits>
#include <utility>

#include "dcl_common.h"

// launch arguments (one definition for every translation unit that instantiates conv_body)
struct ConvArgs {
    const float *x;
    const uint4 *wp;
    float *y;
    const float *addend;            // optional tensor of y's shape added in the epilogue (residual gradient)
    const float *bias;              // optional [Cout] added in the epilogue
    const float *xamax, *wamax;     // max|x| as xcount partial maxima (e.g. per channel), max|w| (1 value)
    int xcount;
    int N, Cin, Cout;
    int H, W;                       // (virtual) input height / width the 3x3 window slides over
    int Hs, Ws, up;                 // stored input size; up = 2: the stored tensor is the virtual one sampled at even
                                    // coordinates, zeros in between (data gradient of a stride-2 convolution)
    int Ho, Wo;                     // output size (= H, W for stride 1)
    int tiles_x, tiles_y, nchunk, groups;
    int phases;                     // 1: data gradient of a stride-2 convolution, one output parity class per workgroup
    int onetap;                     // 1: 1x1 convolution (weights packed with one tap)
    const float *pre_sc, *pre_sh;   // PRE tiles only: per-input-channel affine map of the producer norm; the operand is
                                    // relu(x * pre_sc[c] + pre_sh[c]), formed while the patch is staged (see conv_body, PRE)
};

// dcl_conv3x3_pre.hip: the interleaved stride-1 tiles with the producer norm's map + ReLU in the staging (conv_body, PRE);
// ws2 = the (1, 4) wave-split tile that serves the automatic (2, 2) choice.  False: no such tile (nothing launched).
bool dcl_conv_pre_launch(const ConvArgs &a, int R, int P, int stride, bool ws2, dim3 grid, hipStream_t stream);
bool dcl_conv_pre_has_tile(int R, int P, int stride, bool ws2);

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// Bound probes (tools/probes/conv_bounds.sh builds the library with -DDCL_CONV_PROBE=<bits>; 0 in the product): 1 = no
// MFMAs, 2 = no patch loads, 4 = no output stores, 8 = no weight-fragment loads.  Results are wrong, times tell which part
// of a launch bounds it.
#ifndef DCL_CONV_PROBE
#define DCL_CONV_PROBE 0
#endif
// (bit 16, tools/probes/pk_mix.py: no matrix instruction, but operands A and B stay in use -- the LDS reads, weight loads and splits
// that feed the MFMAs are all still there)
__device__ __forceinline__ f32x16 dcl_fake_mfma(half8 a, half8 b, f32x16 c)
{
    c[0] += (float)a[0] * (float)b[0];
    c[1] += (float)a[7] * (float)b[7];
    return c;
}
#define DCL_MFMA(A, B, C)                                                                              \
    ((DCL_CONV_PROBE & 1) ? (C) : (DCL_CONV_PROBE & 16) ? dcl_fake_mfma((A), (B), (C))                 \
                                                         : __builtin_amdgcn_mfma_f32_32x32x16_f16((A), (B), (C), 0, 0, 0))

constexpr int TW = 32;          // tile width in pixels (one MFMA pixel tile = 1 row x 32 columns)
constexpr int PIXB = 80;        // bytes per LDS pixel record

__device__ __forceinline__ int jrow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// Packed f16 pairs (low half = first value) of hi = f16(v * s) and lo = f16(v * s - hi) for two values.  s is a
// power of two (or 0), so v * s is exact and the fused form is the same number; v_fma_mix{lo,hi}_f16 takes f32 /
// f16 inputs, computes in f32 and writes one f16 half of the destination: 2 VALU instructions per value, no packing.
__device__ __forceinline__ void split2(float v0, float v1, float s, unsigned &hi, unsigned &lo)
{
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hi) : "v"(v0), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hi) : "v"(v1), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=&v"(lo) : "v"(v0), "v"(s), "v"(hi));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(v1), "v"(s), "v"(hi));
}


constexpr float F16_TARGET = 16384.0f;      // operands are scaled so that their absmax lands in (2^13, 2^14]

// power-of-two operand scale from the tensor's absmax (same expression in the packer and in the convolution, so
// both see the same weight scale); an all-zero tensor takes 1, inf / nan propagate into the products
__device__ __forceinline__ float pow2_scale(float amax)
{
    return amax == 0.f ? 1.f : exp2f(fminf(fmaxf(floorf(log2f(F16_TARGET / amax)), -100.f), 100.f));
}

// S = stride (1 | 2).  A stride-2 tile reads a (2 * 4P + 1) x 65 patch, so S = 2 is instantiated with P = 1 only.
//
// PH ("phases"): data gradient of a stride-2 convolution, dx = conv3x3(dy zero-inserted at the odd coordinates, w').  An
// output pixel (2 i + py, 2 j + px) only sees the taps whose input coordinate is even: ky = 1 for py = 0 (stored row
// i), ky = 0 and 2 for py = 1 (stored rows i and i + 1), the same along x -- 1, 2, 2 or 4 taps instead of 9.  Each
// workgroup takes ONE parity class: it runs the stride-1 tile over the STORED dy with the taps remapped (kernel tap
// 1 -> ky 1 | 0, kernel tap 2 -> ky 2, kernel tap 0 unused), skips the others, and scatters its tile to the class's
// pixels.  A quarter of the matrix work of the zero-inserted formulation (which spends 3/4 of it on zeros).
//
// MODE 2 ("one tap"): a 1x1 convolution -- the same tile, patch staging and epilogue with ONE MFMA group per chunk
// (the centre pixel of the patch); the weights are packed with one tap per (tile, chunk) (pack_item, taps = 1) and
// streamed one chunk ahead.
// ---- compile-time decode of the k-th MFMA of a chunk (order: kx, tile row rr, pass, valid tap row ky, channel tile r)
struct MfmaAt {
    int kx, rr, pass, ky, r, first_of_group;
};
template <int R, int P>
constexpr MfmaAt mfma_at(int k)
{
    const int NM = 9 * R * P;
    MfmaAt m{};
    m.kx = k / NM;
    int rem = k - m.kx * NM;
    for (int rr = 0; rr < P + 2; ++rr) {
        const int klo = rr - (P - 1) > 0 ? rr - (P - 1) : 0, khi = rr < 2 ? rr : 2;     // valid ky: rr - ky in [0, P)
        const int npk = khi - klo + 1;
        if (rem < 3 * R * npk) {
            m.rr = rr;
            m.pass = rem / (R * npk);
            const int q = rem - m.pass * R * npk;
            m.ky = klo + q / R;
            m.r = q % R;
            m.first_of_group = rem == 0;
            return m;
        }
        rem -= 3 * R * npk;
    }
    return m;
}
template <int K, int N, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (K < N) {
        f(std::integral_constant<int, K>{});
        static_for<K + 1, N>(f);
    }
}

//
// IL ("interleaved", stride-1 3x3 tiles with two chunks of look-ahead, Cin % 16 == 0): the staging of the NEXT chunks --
// the global loads of chunk c + 2, the split + LDS write of chunk c + 1, the weight fragments of the next kx step -- is
// spread over chunk c's MFMAs by scheduling groups (after every MFMA: <= V loads, <= A VALU, <= 1 LDS write) instead of
// sitting in blocks between fences.  Probe builds of this file (DCL_CONV_PROBE) showed the blocks do NOT overlap the
// matrix work: on every shape time(no MFMAs) + time(no loads, no stores) = time(product), e.g. head 2.2 + 6.5 = 8.8 ms,
// 96 channels 21 + 45 = 63 us -- at one wave per SIMD the 40-94 vector-memory instructions and ~160 VALU of a chunk's
// staging are issued while the matrix pipe idles.
//
// WS ("wave split", 1 | 2): 1 = the four waves of a workgroup stack along the rows, every wave owns all R channel tiles;
// 2 = two waves along the rows x two along the channel tiles (the workgroup covers 2 P rows x 2 R channel tiles, a wave
// P rows x R tiles).  For the 48 / 64-channel layers -- three or four K chunks, tile (R, P) = (2, 2) -- every wave
// streaming BOTH channel tiles' weight fragments is 1 KiB from L1 per ~3 MFMAs and wave, 85 B / clk at two workgroups per
// CU against the L1's 64: the (1, 4) wave of WS = 2 does the same 108 MFMAs per chunk on half the fragments.
//
// PRE ("pre-activation", interleaved tiles only): the convolution's input is relu(bn(z)) of the tensor z it is handed -- the
// BasicBlock's conv1 -> bn1 -> relu -> conv2 (reference models/HRNet.py:77-93) without the normalised tensor ever being written:
// every staged value goes through v = max(fma(z, sc[c], sh[c]), 0) -- the norm kernels' own bn_eval + ReLU, bit for bit -- in
// front of its f16 split.  Halo pixels outside the image stay zero AFTER the map because they are masked through the operand
// scale (gsc = 0), which multiplies the mapped value.  The staging items are dealt out so that an item's channel octet is
// wave-uniform (waves 0-1: channels 0-7 of the chunk, waves 2-3: channels 8-15) and the eight (sc, sh) pairs of a chunk are
// scalar registers: no vector register is spent on them (the (3, 4) tile has none to spare).
template <int R, int P, int S, int MODE = 0, bool IL = false, int WS = 1, bool PRE = false>
__device__ __forceinline__ void conv_body(const ConvArgs &a, const int bid)
{
    constexpr bool PH = MODE == 1, T1 = MODE == 2;
    static_assert(!PRE || (IL && MODE == 0), "pre-activation: interleaved 3x3 tiles only");
    constexpr int WR = 4 / WS;                    // waves along the rows
    static_assert(WS == 1 || (WS == 2 && IL), "wave split: interleaved stride-1 tiles only");
    static_assert(!IL || (MODE == 0 && (S == 1 || P == 1)), "interleaved staging: stride-1 3x3 tiles, stride 2 with one row per wave");
    static_assert(MODE == 0 || S == 1, "phases / one tap: stride-1 tile");
    constexpr int LW = S * (TW - 1) + 3;          // patch width incl. halo: 34 | 65
    constexpr int ROWS = S * (WR * P - 1) + 3;    // 4P + 2 | 8P + 1 (WS = 1)
    constexpr int TP = ROWS * LW;
    constexpr int NITEM = PRE ? (TP + 127) / 128 : (2 * TP + 255) / 256;
    constexpr int BUFB = TP * PIXB;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUFB];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, h = lane >> 5, li = lane & 31;
    // channel octet (0 | 1) of staging item m of this thread: by position in the item sequence, or (PRE) by wave pair
    const int oct_u = PRE ? __builtin_amdgcn_readfirstlane(tid >> 7) : 0;
    auto item_oct = [&](int m) { return PRE ? oct_u : (tid + 256 * m >= TP ? 1 : 0); };
    // XCD-aware decode of the 1-D grid (consecutive workgroup ids go round-robin over the 8 XCDs): the `groups`
    // channel groups of one pixel tile get ids xcd + 8 * (g + groups * hi), i.e. the SAME XCD and adjacent dispatch
    // slots, so that the input patch they all read is fetched from HBM once and then served by that XCD's L2
    // (the head convolution has 8 groups: 12.0 GiB -> 1.8 GiB of FETCH_SIZE per launch).
    int bx, cg;
    {
        // (phases: the four parity classes of a tile are four more "groups" -- same input patch, same XCD)
        const int ngrp = PH ? 4 * a.groups : a.groups;
        const int ntile = a.tiles_x * a.tiles_y * a.N, n8 = ntile & ~7, main_blocks = n8 * ngrp;
        if (bid < main_blocks) {
            const int xcd = bid & 7, rest = bid >> 3;
            cg = rest % ngrp;
            // each XCD takes a CONTIGUOUS eighth of the tile sequence, so that vertically / horizontally adjacent
            // tiles (which share halo rows and columns) also share an L2
            bx = xcd * (n8 >> 3) + rest / ngrp;
        } else {
            const int rest = bid - main_blocks;
            cg = rest % ngrp;
            bx = n8 + rest / ngrp;
        }
    }
    // parity class (py, px) and its taps: kernel tap k reads original tap ko[k]; bit k of the mask = tap in use
    const int phase = PH ? cg / a.groups : 0;
    if (PH)
        cg -= phase * a.groups;
    const int py = phase >> 1, px = phase & 1;
    const int kym = PH ? (py ? 6 : 2) : 7, kxm = PH ? (px ? 6 : 2) : 7;
    const int kyo1 = PH ? (py ? 0 : 1) : 1, kxo1 = PH ? (px ? 0 : 1) : 1;       // kernel tap 1 (tap 2 -> 2, tap 0 -> 0)
    const int tx = bx % a.tiles_x;
    bx /= a.tiles_x;
    const int ty = bx % a.tiles_y;
    const int n = bx / a.tiles_y;
    const int x0 = tx * TW, y0 = ty * WR * P;
    const int wrow = wave / WS;                   // this wave's row group (= wave for WS = 1)
    const int T0 = (cg * WS + wave % WS) * R;
    const size_t plane = (size_t)a.Hs * a.Ws;                // stored input plane
    const size_t oplane = (size_t)a.Ho * a.Wo;
    const float *xb = a.x + (size_t)n * a.Cin * plane;
    // staging work items: (octet of channels, patch pixel); consecutive lanes -> consecutive pixels.
    // Every load is unconditional from a clamped, always valid address (a branch around a load makes the compiler
    // wait for it at the join, i.e. in front of the MFMAs it is meant to overlap): halo pixels outside the image
    // read pixel 0 and are zeroed through their operand scale (0 instead of s); channels past Cin read channel
    // Cin - 1 and meet the zero weights the packer wrote for k >= Cin.
    int goff[NITEM], loff[NITEM];
    float gsc[NITEM];               // operand scale of the item (0 for halo pixels outside the image), set below
    bool gok[NITEM];
#pragma unroll
    for (int m = 0; m < NITEM; ++m) {
        const int it = min(tid + 256 * m, 2 * TP - 1);          // surplus items repeat the last one
        const int oct = PRE ? oct_u : (it >= TP ? 1 : 0);
        const int pix = PRE ? min((tid & 127) + 128 * m, TP - 1) : it - oct * TP;
        const int r = pix / LW, c = pix - r * LW;
        const int gy = S * y0 + r - 1, gx = S * x0 + c - 1;
        bool ok = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        int sy = gy, sx = gx;
        if (a.up == 2) {                                        // zero-inserted input: only even coordinates exist
            ok = ok && !((gy | gx) & 1);
            sy = gy >> 1;
            sx = gx >> 1;
            ok = ok && sy < a.Hs && sx < a.Ws;
        }
        // stride 2: the columns of a patch row are stored de-interleaved, [33 even | 32 odd] -- tap kx of output column li
        // reads patch column 2 li + kx, i.e. consecutive lanes read consecutive records of ONE parity plane (80-byte stride:
        // conflict-free, like stride 1) instead of every second record (160-byte stride: 2-way bank conflicts)
        const int cs = S == 2 ? ((c & 1) ? (LW + 1) / 2 + (c >> 1) : (c >> 1)) : c;
        loff[m] = (r * LW + cs) * PIXB + oct * 16;
        goff[m] = ok ? sy * a.Ws + sx : 0;
        gok[m] = ok;
    }
    const bool ragged = (a.Cin & 15) != 0;                       // last chunk has channels past Cin
    // two register sets: gA holds the patch of the NEXT chunk (split and written to LDS while the current chunk's
    // MFMAs run), gB receives the loads of the chunk after that
    float gA[NITEM][8], gB[NITEM][8];
    auto load_items = [&](int c, float (&g)[NITEM][8]) {
        if (DCL_CONV_PROBE & 2) {
#pragma unroll
            for (int m = 0; m < NITEM; ++m)
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    g[m][e] = 1.0f;
        } else if (!ragged || c + 1 < a.nchunk) {
#pragma unroll
            for (int m = 0; m < NITEM; ++m) {
                const float *xc = xb + (size_t)(16 * c + 8 * item_oct(m)) * plane + goff[m];
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    g[m][e] = xc[e * plane];
            }
        } else {
#pragma unroll
            for (int m = 0; m < NITEM; ++m) {
                const int ch0 = 16 * c + 8 * item_oct(m);
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    g[m][e] = xb[(size_t)min(ch0 + e, a.Cin - 1) * plane + goff[m]];
            }
        }
    };
    auto write_items = [&](unsigned char *buf, const float (&g)[NITEM][8]) {
#pragma unroll
        for (int m = 0; m < NITEM; ++m) {
            unsigned hh[4], ll[4];
            split2(g[m][0], g[m][1], gsc[m], hh[0], ll[0]);
            split2(g[m][2], g[m][3], gsc[m], hh[1], ll[1]);
            split2(g[m][4], g[m][5], gsc[m], hh[2], ll[2]);
            split2(g[m][6], g[m][7], gsc[m], hh[3], ll[3]);
            *(uint4 *)(buf + loff[m]) = make_uint4(hh[0], hh[1], hh[2], hh[3]);
            *(uint4 *)(buf + loff[m] + 32) = make_uint4(ll[0], ll[1], ll[2], ll[3]);
        }
    };

    // PRE: the (sc, sh) pairs of the eight channels this wave stages of chunk c, in SCALAR registers.  Explicit s_load: left to
    // the compiler the uniform loads became vector loads into 16 vector registers that stayed live through the chunk (and an "s"
    // operand was handed a vector register all the same).  The compiler does not know these instructions load, so their
    // completion is awaited by hand (pre_wait: lgkmcnt(0) right in front of the chunk's first use -- stricter than any wait the
    // compiler counts for its own LDS reads, which therefore stay correct with one more operation in the counter).
    typedef float f32x8s __attribute__((ext_vector_type(8)));
    f32x8s psc, psh;
    auto load_pre = [&](int c) {
        if constexpr (PRE) {
            const unsigned long long ps = (unsigned long long)(uintptr_t)(a.pre_sc + 16 * c + 8 * oct_u),
                                     ph = (unsigned long long)(uintptr_t)(a.pre_sh + 16 * c + 8 * oct_u);
            // (readfirstlane returns a signed int: through unsigned words, or a set bit 31 of the low word sign-extends into the high one)
            const unsigned slo = __builtin_amdgcn_readfirstlane((unsigned)ps), shi = __builtin_amdgcn_readfirstlane((unsigned)(ps >> 32)),
                           hlo = __builtin_amdgcn_readfirstlane((unsigned)ph), hhi = __builtin_amdgcn_readfirstlane((unsigned)(ph >> 32));
            const unsigned long long us = ((unsigned long long)shi << 32) | slo, uh = ((unsigned long long)hhi << 32) | hlo;
            asm volatile("s_load_dwordx8 %0, %2, 0x0\n\ts_load_dwordx8 %1, %3, 0x0" : "=&s"(psc), "=&s"(psh) : "s"(us), "s"(uh));
        }
    };
    auto pre_wait = [&]() {
        if constexpr (PRE)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(psc), "+s"(psh));
    };
    // v = max(fma(z, sc, sh), 0) with both constants in scalar registers: one VOP3 instruction takes ONE scalar operand on this
    // target, so the shift is moved into the destination first; volatile so that the three instructions stay in their slot among
    // the MFMAs
    auto pre_act = [](float &v, float sc, float sh) {
        if constexpr (PRE)
            asm volatile("v_mov_b32 %0, %2\n\tv_fma_f32 %0, %1, %3, %0\n\tv_max_f32 %0, 0, %0"
                         : "=&v"(v) : "v"(v), "s"(sh), "s"(sc));
    };

    // A fragments: wp[(((T * nchunk + c) * 9 + ky * 3 + kx) * 2 + part) * 64 + lane]; channel tiles past the last
    // one (a workgroup tile wider than Cout) re-read the last tile, their results are never stored
    const int mtiles = (a.Cout + 31) / 32;
    const uint4 *wa[R];
#pragma unroll
    for (int r = 0; r < R; ++r)
        wa[r] = a.wp + (size_t)min(T0 + r, mtiles - 1) * a.nchunk * (T1 ? 1 : 9) * 2 * 64 + lane;
    auto load_A = [&](half8 (&A)[3][R][2], int c, int kx) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int part = 0; part < 2; ++part) {
                    const int tap = PH ? (ky == 1 ? kyo1 : ky) * 3 + (kx == 1 ? kxo1 : kx) : ky * 3 + kx;
                    const uint4 v = (DCL_CONV_PROBE & 8) ? make_uint4(tap, c, r, part)
                                                         : wa[r][(((size_t)c * 9 + tap) * 2 + part) * 64];
                    A[ky][r][part] = __builtin_bit_cast(half8, v);
                }
    };

    f32x16 acc[R][P];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int q = 0; q < 16; ++q)
                acc[r][p][q] = 0.f;

    // Weight fragments are streamed AD kx-steps ahead of their use into a ring of three register sets indexed by
    // kx (static indices, no copies): one step ahead when a step holds >= 54 MFMAs (R = 3), two steps ahead for
    // the smaller tiles, whose steps are shorter than an L2 round trip.
    constexpr int AD = (R >= 3 || (R == 2 && P == 2)) ? 1 : 2;   // (2, 2) must stay under 256 registers
    constexpr bool O2 = S == 1 && ((R == 2 && P == 2) || WS == 2) && !T1;              // the tile that runs at two workgroups per CU
                                                                 // ((3, 1) was tried: its spills cost more than it gains)
    constexpr bool LA2 = !O2;                                    // two chunks of patch look-ahead (one for those)
    constexpr bool BPIPE = LA2 && !PH;                           // B fragments one group ahead (not for (2, 2): registers;
                                                                 // not with skipped taps: plain reads there)
    half8 Ab[3][3][R][2];
    const int nsteps = 3 * a.nchunk;
    half8 A1[T1 ? 2 : 1][R][2];                                     // one tap: weight fragments of two chunks
    auto load_A1 = [&](half8 (&A)[R][2], int c) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int part = 0; part < 2; ++part)
                A[r][part] = __builtin_bit_cast(half8, wa[r][((size_t)c * 2 + part) * 64]);
    };
    load_items(0, gA);
    if constexpr (T1) {
        load_A1(A1[0], 0);
    } else {
        load_A(Ab[0], 0, 0);
        if (AD == 2)
            load_A(Ab[1], 0, 1);
    }
    // operand scale of x (max over the producer's partial maxima, exchanged between the waves through LDS) -- after
    // the first patch and weight fragments are in flight, so that its memory round trip hides behind theirs
    float xs;
    {
        float m = 0.f;
        for (int i = tid; i < a.xcount; i += 256)
            m = fmaxf(m, a.xamax[i]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
            m = fmaxf(m, __shfl_xor(m, o, 64));
        float *wm = (float *)lds;
        if (lane == 0)
            wm[wave] = m;
        __syncthreads();
        m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
        __syncthreads();
        xs = pow2_scale(m);
    }

#pragma unroll
    for (int m = 0; m < NITEM; ++m)
        gsc[m] = gok[m] ? xs : 0.f;
    // Epilogue addend (residual gradient) / bias: the accumulators START at (addend + bias) * (x scale * w scale) instead of
    // at zero, so the epilogue is scaling + stores only.  Loading them there -- 16 loads per accumulator tile, then its 16
    // stores -- put s_waitcnt vmcnt(0) in front of every tile's stores, i.e. behind the previous tile's (stores count in
    // vmcnt on this target): a dozen store round trips per wave in a row.
    if (a.addend || a.bias) {
        const float sc2 = xs * pow2_scale(a.wamax[0]);
        const int col0 = PH ? 2 * (x0 + li) + px : x0 + li;
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const int row = PH ? 2 * (y0 + P * wrow + p) + py : y0 + P * wrow + p;
                const int cob = (T0 + r) * 32 + 4 * h;
                const bool ok = row < a.Ho && col0 < a.Wo && cob < a.Cout;
                const size_t o0 = (((size_t)n * a.Cout + min(cob, a.Cout - 1)) * a.Ho + min(row, a.Ho - 1)) * a.Wo +
                                  min(col0, a.Wo - 1);
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int k = ok ? min((q & 3) + 8 * (q >> 2), a.Cout - 1 - cob) : 0;
                    const float ad = (a.addend ? a.addend[o0 + (size_t)k * oplane] : 0.f) +
                                     (a.bias ? a.bias[min(cob, a.Cout - 1) + k] : 0.f);
                    acc[r][p][q] = ad * sc2;
                }
            }
    }
    if constexpr (PRE) {
        load_pre(0);
        pre_wait();
#pragma unroll
        for (int m = 0; m < NITEM; ++m)
#pragma unroll
            for (int e = 0; e < 8; ++e)
                pre_act(gA[m][e], psc[e], psh[e]);
    }
    write_items(lds, gA);
    // (IL with P = 4: ONE register set -- a chunk's items are loaded during the first half of the previous chunk's slices
    // and split + written during its second half; the two-set scheme spills at (3, 4))
    constexpr bool ONESET = IL && (P == 4 || O2);
    if (LA2 && !ONESET)
        load_items(min(1, a.nchunk - 1), gA);
    __syncthreads();

    const int brow = (S * P * wrow) * LW + S * li;      // patch pixel of this lane's output pixel, tap (0, 0)
    if constexpr (T1) {
        // one tap: chunk c = P B fragments (patch pixel (p + 1, li + 1)) x R channel tiles x 3 passes; the weight
        // fragments of chunk c + 1 are fetched while chunk c runs (two register sets, loop unrolled by two)
        auto chunk = [&](auto PHASE, int c) {
            constexpr int ph = decltype(PHASE)::value;
            const unsigned char *cur = lds + (c & 1) * BUFB;
            const bool more = c + 1 < a.nchunk;
            __builtin_amdgcn_sched_barrier(0);
            load_items(min(c + 2, a.nchunk - 1), gB);
            load_A1(A1[ph ^ 1], min(c + 1, a.nchunk - 1));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const unsigned char *bp = cur + (brow + (p + 1) * LW + 1) * PIXB + h * 16;
                const half8 bh = *(const half8 *)bp, bl = *(const half8 *)(bp + 32);
#pragma unroll
                for (int pass = 0; pass < 3; ++pass)
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        acc[r][p] = DCL_MFMA(A1[ph][r][pass == 2 ? 1 : 0],
                                                                           pass == 1 ? bl : bh, acc[r][p]);
                if (p == 0 && more)
                    write_items(lds + ((c + 1) & 1) * BUFB, gA);
            }
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
#pragma unroll
            for (int m = 0; m < NITEM; ++m)
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    gA[m][e] = gB[m][e];
        };
        int c = 0;
        for (; c + 2 <= a.nchunk; c += 2) {
            chunk(std::integral_constant<int, 0>{}, c);
            chunk(std::integral_constant<int, 1>{}, c + 1);
        }
        if (c < a.nchunk)
            chunk(std::integral_constant<int, 0>{}, c);
    } else if constexpr (IL) {
    // Every MFMA of a chunk is followed by AT MOST a few instructions of staging work and a scheduling fence, so the
    // staging is issued in the shadow of the matrix pipe (an MFMA occupies it for 32 cycles and the wave issues in order:
    // a block of 8 loads behind 6 MFMAs still leaves the pipe idle for most of the block).  Micro-operations, by MFMA index
    // k of the chunk (NT = 27 R P MFMAs): the 6 R weight-fragment loads of the next kx step at the first MFMAs of a step;
    // item m's eight global loads one per MFMA from LB(m); its split in four pieces of 4 VALU and its two LDS writes from
    // WB(m).  Two register sets (loads of chunk c + 2, writes of chunk c + 1), or ONE for the tiles that cannot afford two
    // (P = 4, and the (2, 2) tile at two workgroups per CU): loads of chunk c + 1 in the first half of the chunk, writes in
    // the second.  (Scheduling groups -- "1 MFMA, <= V loads, <= A VALU, <= 1 LDS write" x 324 -- were tried first and
    // ignored by the scheduler; fences per (tile row, pass) slice left blocks of 8 loads: +13-23 % on the 96 ... 512-channel
    // shapes, little on the head.)
    constexpr int NT = 27 * R * P, NM = 9 * R * P;
    constexpr int HALF = NT / 2;
    unsigned sh_[NITEM][4], sl_[NITEM][4];
    for (int c = 0; c < a.nchunk; ++c) {
        const unsigned char *cur = lds + (c & 1) * BUFB;
        unsigned char *nxt = lds + ((c + 1) & 1) * BUFB;            // (past the last chunk: written, never read)
        const int c2 = min(c + (ONESET ? 1 : 2), a.nchunk - 1);
        half8 bq[2][2];
        auto read_b = [&](int g, half8 (&dst)[2]) {
            // stride 2 (P = 1: fragment group g = (kx, ky)): patch row 2 wrow + ky, de-interleaved column slot of 2 li + kx
            const int rr = g % (P + 2), kx = g / (P + 2);
            const unsigned char *bp = S == 2
                ? cur + ((2 * P * wrow + rr) * LW + (kx == 1 ? (LW + 1) / 2 + li : li + (kx >> 1))) * PIXB + h * 16
                : cur + (brow + rr * LW + kx) * PIXB + h * 16;
            dst[0] = *(const half8 *)bp;
            dst[1] = *(const half8 *)(bp + 32);
        };
        load_pre(min(c + 1, a.nchunk - 1));        // (PRE) the map of the chunk that is split + written under this chunk's MFMAs
        __builtin_amdgcn_sched_barrier(0);
        read_b(0, bq[0]);
        static_for<0, NT>([&](auto KC) {
            constexpr int k = decltype(KC)::value;
            constexpr MfmaAt at = mfma_at<R, P>(k);
            constexpr int g = at.kx * (P + 2) + at.rr;
            if constexpr (at.first_of_group && g + 1 < 3 * (P + 2))
                read_b(g + 1, bq[(g + 1) & 1]);
            acc[at.r][at.rr - at.ky] = DCL_MFMA(Ab[at.kx][at.ky][at.r][at.pass == 2 ? 1 : 0],
                                                at.pass == 1 ? bq[g & 1][1] : bq[g & 1][0], acc[at.r][at.rr - at.ky]);
            // ---- micro-operations of this MFMA
            {   // weight fragments of step kx + AD: fragment f = (tap row, tile, part) at MFMA kx * NM + f
                constexpr int f = k - at.kx * NM;
                if constexpr (f < 6 * R) {
                    const int s2 = min(3 * c + at.kx + AD, nsteps - 1), c3 = s2 / 3, kx3 = s2 % 3;
                    constexpr int ky = f / (2 * R), r = (f / 2) % R, part = f & 1;
                    const uint4 v = (DCL_CONV_PROBE & 8) ? make_uint4(ky, c3, r, part)
                                                         : wa[r][(((size_t)c3 * 9 + ky * 3 + kx3) * 2 + part) * 64];
                    Ab[(at.kx + AD) % 3][ky][r][part] = __builtin_bit_cast(half8, v);
                }
            }
            static_for<0, NITEM>([&](auto MC) {
                constexpr int m = decltype(MC)::value;
                // first MFMA of the item's eight loads / of its four split pieces + two writes: spread over the chunk (one
                // set: loads in the first half, writes in the second), every micro-operation inside [0, NT)
                // (PRE: four more slots in front of the split, one per pair of values, for the norm's map + ReLU)
                constexpr int WL = PRE ? 10 : 6, PS = PRE ? 4 : 0;
                constexpr int lb = ONESET ? m * (HALF - 8) / NITEM : m * (NT - 8) / NITEM + (NT - 8) / (2 * NITEM);
                constexpr int wb = (ONESET ? HALF + m * (NT - HALF - WL) / NITEM : m * (NT - WL) / NITEM) + PS;
                static_assert(lb >= 0 && lb + 8 <= NT && wb - PS >= 0 && wb + 6 <= NT && (!ONESET || lb + 8 <= wb - PS),
                              "staging micro-operations must fall inside the chunk");
                if constexpr (k >= lb && k < lb + 8) {                  // one of the item's eight loads
                    constexpr int e = k - lb;
                    const float *xc = xb + (size_t)(16 * c2 + 8 * item_oct(m)) * plane + goff[m];
                    const float v = (DCL_CONV_PROBE & 2) ? 1.0f : xc[e * plane];
                    if constexpr (ONESET)
                        gA[m][e] = v;
                    else
                        gB[m][e] = v;
                }
                if constexpr (PRE && k >= wb - PS && k < wb) {          // norm + ReLU of a quarter of the item's values
                    constexpr int q = k - (wb - PS);
                    if constexpr (m == 0 && q == 0)
                        pre_wait();                                     // the chunk's (sc, sh) pairs have arrived
                    pre_act(gA[m][2 * q], psc[2 * q], psh[2 * q]);
                    pre_act(gA[m][2 * q + 1], psc[2 * q + 1], psh[2 * q + 1]);
                }
                if constexpr (k >= wb && k < wb + 4) {                  // a quarter of the split
                    constexpr int q = k - wb;
                    split2(gA[m][2 * q], gA[m][2 * q + 1], gsc[m], sh_[m][q], sl_[m][q]);
                }
                if constexpr (k == wb + 4)
                    *(uint4 *)(nxt + loff[m]) = make_uint4(sh_[m][0], sh_[m][1], sh_[m][2], sh_[m][3]);
                if constexpr (k == wb + 5)
                    *(uint4 *)(nxt + loff[m] + 32) = make_uint4(sl_[m][0], sl_[m][1], sl_[m][2], sl_[m][3]);
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        __syncthreads();
        if (!ONESET) {
#pragma unroll
            for (int m = 0; m < NITEM; ++m)
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    gA[m][e] = gB[m][e];
        }
    }
    } else {
    for (int c = 0; c < a.nchunk; ++c) {
        const unsigned char *cur = lds + (c & 1) * BUFB;
        const bool more = c + 1 < a.nchunk;
        // Two chunks of look-ahead: the loads of chunk c + 2 are issued here (pinned above the MFMAs; past the end
        // they re-read the last chunk and are dropped); the patch of chunk c + 1, loaded one iteration ago, is split
        // and written to the other LDS buffer in the shadow of this chunk's MFMAs (after the first kx step).
        __builtin_amdgcn_sched_barrier(0);
        if (LA2)
            load_items(min(c + 2, a.nchunk - 1), gB);
        else
            load_items(min(c + 1, a.nchunk - 1), gA);
        __builtin_amdgcn_sched_barrier(0);
        // B fragments (stride 1) are read ONE group ahead of the MFMAs that consume them -- group g = (kx, tile row
        // rr) -- into two alternating register pairs, so that the LDS latency hides behind the previous group's MFMAs
        // (the scheduling groups below pin "2 DS reads, then the group's MFMAs"; left alone the reads sit right in
        // front of their first use and every group starts with an exposed ~100-cycle wait)
        half8 bq[2][2];
        auto read_b = [&](int g, half8 (&dst)[2]) {
            const unsigned char *bp = cur + (brow + (g % (P + 2)) * LW + g / (P + 2)) * PIXB + h * 16;
            dst[0] = *(const half8 *)bp;
            dst[1] = *(const half8 *)(bp + 32);
        };
        if (S == 1 && BPIPE)
            read_b(0, bq[0]);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            {
                const int s2 = min(3 * c + kx + AD, nsteps - 1);      // step whose fragments are fetched now
                load_A(Ab[(kx + AD) % 3], s2 / 3, s2 % 3);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (S == 1 && (!PH || ((kxm >> kx) & 1))) {
                // a fragment depends on (tile row rr = p + ky, kx) only: read once, used by every (p, ky) pair
#pragma unroll
                for (int rr = 0; rr < P + 2; ++rr) {
                    const int g = kx * (P + 2) + rr;
                    if (PH && rr == 0)
                        continue;                                       // kernel tap row 0 is never in use
                    if (!BPIPE)
                        read_b(g, bq[g & 1]);
                    else if (g + 1 < 3 * (P + 2))
                        read_b(g + 1, bq[(g + 1) & 1]);
                    const half8 bh = bq[g & 1][0], bl = bq[g & 1][1];
#pragma unroll
                    for (int pass = 0; pass < 3; ++pass)
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky) {
                            const int p = rr - ky;
                            if (p >= 0 && p < P && (!PH || ((kym >> ky) & 1))) {
#pragma unroll
                                for (int r = 0; r < R; ++r)
                                    acc[r][p] = DCL_MFMA(Ab[kx][ky][r][pass == 2 ? 1 : 0], pass == 1 ? bl : bh, acc[r][p]);
                            }
                        }
                    // number of (p, ky) pairs of this tile row
                    const int npk = (rr < 3 ? rr + 1 : 3) - (rr > P - 1 ? rr - (P - 1) : 0);
                    if (!BPIPE)
                        continue;
                    if (g + 1 < 3 * (P + 2))
                        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    if (npk == 1)
                        __builtin_amdgcn_sched_group_barrier(0x008, 3 * R, 0);
                    else if (npk == 2)
                        __builtin_amdgcn_sched_group_barrier(0x008, 6 * R, 0);
                    else
                        __builtin_amdgcn_sched_group_barrier(0x008, 9 * R, 0);
                }
            } else if (S == 2) {
                // stride 2: output pixel (p, li) reads patch pixel (2 p' + ky, 2 li + kx)
#pragma unroll
                for (int p = 0; p < P; ++p)
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        const int cslot = kx == 1 ? (LW + 1) / 2 + li : li + (kx >> 1);      // de-interleaved columns, see loff
                        const unsigned char *bp = cur + ((S * P * wrow + S * p + ky) * LW + cslot) * PIXB + h * 16;
                        const half8 bh = *(const half8 *)bp;
                        const half8 bl = *(const half8 *)(bp + 32);
#pragma unroll
                        for (int pass = 0; pass < 3; ++pass)
#pragma unroll
                            for (int r = 0; r < R; ++r)
                                acc[r][p] = DCL_MFMA(Ab[kx][ky][r][pass == 2 ? 1 : 0], pass == 1 ? bl : bh, acc[r][p]);
                    }
            }
            if (LA2 && kx == 0 && more)
                write_items(lds + ((c + 1) & 1) * BUFB, gA);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!LA2 && more)
            write_items(lds + ((c + 1) & 1) * BUFB, gA);
        __syncthreads();
        if (LA2) {
#pragma unroll
            for (int m = 0; m < NITEM; ++m)
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    gA[m][e] = gB[m][e];
        }
    }

    }
    const float inv = 1.0f / (xs * pow2_scale(a.wamax[0]));
    const int col = PH ? 2 * (x0 + li) + px : x0 + li;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int row = PH ? 2 * (y0 + P * wrow + p) + py : y0 + P * wrow + p;
            const int cob = (T0 + r) * 32 + 4 * h;
            if (row < a.Ho && col < a.Wo && cob < a.Cout && !((DCL_CONV_PROBE & 4) && acc[r][p][0] != 12345.f)) {
                const size_t o0 = (((size_t)n * a.Cout + cob) * a.Ho + row) * a.Wo + col;
                float *yp = a.y + o0;
                if (cob + 28 <= a.Cout) {            // whole channel tile inside Cout
#pragma unroll
                    for (int q = 0; q < 16; ++q)
                        yp[(size_t)((q & 3) + 8 * (q >> 2)) * oplane] = acc[r][p][q] * inv;
                } else {
#pragma unroll
                    for (int q = 0; q < 16; ++q)
                        if (cob + (q & 3) + 8 * (q >> 2) < a.Cout)
                            yp[(size_t)((q & 3) + 8 * (q >> 2)) * oplane] = acc[r][p][q] * inv;
                }
            }
        }
}

}  // namespace
