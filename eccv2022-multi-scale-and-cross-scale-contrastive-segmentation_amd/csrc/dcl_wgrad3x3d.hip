// dcl_wgrad3x3d.hip -- weight gradient of the 3x3 / stride 1 / pad 1 convolution, operands staged by LDS-DMA.
//
// Arithmetic, tiling, work split, slabs and reductions are those of dcl_wgrad3x3.hip (k_wgrad3x3): one wave = NCO x NCI
// tiles of 16 x 16 (co x ci) x 9 taps, walking down a 32-pixel strip, dY rows r - 1 .. r + 1 as f16 fragments in a
// register ring, the three kx windows of an X row cut out of 10 split values.  What differs is how the rows get there.
//
// k_wgrad3x3 loads them in MFMA order: lane (q, j) reads pixels 8 q .. 8 q + 7 of channel row j -- 16 different rows in
// any 16 neighbouring lanes.  The vector-memory front end processes such an instruction one lane at a time
// (TCP_TOTAL_CACHE_ACCESSES / SQ_INSTS_VMEM_RD = 64, counters in profiles/): 640 cache-access cycles per wave and row
// step for the (3, 1) tile against 1296 cycles of MFMA work, times four waves per CU -- the kernel is bound by the L1
// tag rate at ~45 % matrix-pipe utilisation.  Here every row is fetched with global_load_lds_dwordx4, lanes running
// ALONG the row: 10 (9) consecutive lanes take the 160 (128) contiguous bytes a tile row needs -- strip, 4-pixel halo
// pieces left and right -- so an instruction touches ~14 cache lines instead of 64; the data lands in a private LDS
// ring of the wave (no barriers), [tile][row][piece] with one pad piece per row (strides of 11 / 9 pieces: the 16 rows a
// ds_read_b128 pass reads fall into 16 different bank groups), and is read back in MFMA order two row steps later.
// Converted fragments never go through LDS; the registers that used to hold raw rows in flight are gone.
#include <type_traits>

#include "dcl_common.h"
#include "dcl_wgrad.h"

// Bound probes (tools/probes/conv_bounds.sh; 0 in the product): 1 = no MFMAs, 2 = no LDS-DMA.  Results are wrong.
#ifndef DCL_WG_PROBE
#define DCL_WG_PROBE 0
#endif
#define DCL_WMFMA(A, B, C) ((DCL_WG_PROBE & 1) ? (C) : __builtin_amdgcn_mfma_f32_16x16x32_f16((A), (B), (C), 0, 0, 0))

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr float F16_TARGET = 16384.0f;

__device__ __forceinline__ float pow2_scale(float amax)
{
    return amax == 0.f ? 1.f : exp2f(fminf(fmaxf(floorf(log2f(F16_TARGET / amax)), -100.f), 100.f));
}

__device__ __forceinline__ void split2(float v0, float v1, float s, unsigned &hi, unsigned &lo)
{
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hi) : "v"(v0), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hi) : "v"(v1), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=&v"(lo) : "v"(v0), "v"(s), "v"(hi));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(v1), "v"(s), "v"(hi));
}

__device__ __forceinline__ void split1(float v0, float s, unsigned &hi, unsigned &lo)
{
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hi) : "v"(v0), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=&v"(lo) : "v"(v0), "v"(s), "v"(hi));
}

__device__ __forceinline__ half8 as_half8(u32x4 v) { return __builtin_bit_cast(half8, v); }

// one LDS-DMA wave instruction: lane i copies the 16 bytes at gbase + voff(i) to LDS byte address lds_dst + 16 i.
// (inline asm: the compiler puts s_waitcnt vmcnt(0) in front of every LDS read that follows the builtin.)
__device__ __forceinline__ const float *uniform_ptr(const float *p)
{
    const unsigned long long v = (unsigned long long)(uintptr_t)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const float *)(uintptr_t)(((unsigned long long)hi << 32) | lo);
}

// gbase and lds_dst must be wave-uniform (scalar registers)
__device__ __forceinline__ void dma16(const void *gbase, unsigned voff, unsigned lds_dst)
{
    if (DCL_WG_PROBE & 2)
        return;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(gbase), "s"(lds_dst)
                 : "memory");
}

template <int N>
__device__ __forceinline__ void dma_wait()
{
    static_assert(N >= 0 && N < 64, "vmcnt immediate");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

constexpr int APIECES = 9, BPIECES = 11;          // 16-byte pieces per staged row: 8 (+ 2 halo) + 1 pad
#ifndef DCL_WG_SPREAD
#define DCL_WG_SPREAD 1
#endif
constexpr bool g_wgrad_spread = DCL_WG_SPREAD != 0;      // LDS-DMA instructions dealt out among the MFMAs (see step())

// WAVE = true (a.wave_mode): between 129 and 256 tile pairs a workgroup per pair leaves CUs empty (the head's 144 -> 720
// launch: 135 pairs on 256 CUs) and a second workgroup per pair would run as a second round.  Here the S = 128 / ceil(pairs / 8)
// pixel splits of a pair go to single WAVES: XCD x owns a contiguous run of pairs (~pairs / 8 of them: neighbours share their dY
// channel rows in that XCD's L2), its 128 waves (32 workgroups of the 256-workgroup grid) take (pair, split) jobs in order --
// wave_mode 2: pair fastest, so the four waves of a workgroup walk the same pixels for four neighbouring pairs (shared dY rows
// hit in the CU's L1: 1.89 ms against 1.99 with split fastest, 2.83 with a workgroup per pair on 12 x (144 -> 720) x 128 x 256) --
// and a wave writes its own slab (no LDS reduction: the four waves of a workgroup belong to different pairs).
//
// PRE: x is the RAW output z of the convolution in front of a training-mode norm and the operand is relu(z sc[ci] + sh[ci]) --
// the weight gradient of the BasicBlock's conv2 (reference models/HRNet.py:77-93) without the normalised tensor in memory
// (dcl_conv3x3_pre.hip has the forward).  A lane converts values of ONE input channel per ci tile (row j of the tile), so the
// map costs two registers per tile and one fma + one max per value in front of the split; the halo values and the pixels past
// the row end are masked through the operand scale, i.e. after the map.
template <int NCO, int NCI, bool WAVE = false, bool PRE = false>
__global__ __launch_bounds__(256, 1) void k_wgrad3x3d(WgradArgs a)
{
    constexpr int NIA = (NCO * 16 * APIECES + 63) / 64, NIB = (NCI * 16 * BPIECES + 63) / 64, NI = NIA + NIB;
    constexpr int NS = 3;                                   // ring slots: rows are fetched two steps ahead
    constexpr int SLOTB = NI * 1024, STAGEB = 4 * NS * SLOTB;
    constexpr bool LDSRED = !WAVE && NCO * NCI <= 4;
    constexpr int NREG = NCO * NCI * 36;
    constexpr int REDB = LDSRED ? 2 * NREG * 64 * 4 : 0;
    constexpr int SMEMB = STAGEB > REDB ? STAGEB : REDB;
    static_assert(SMEMB + 64 <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEMB];
    __shared__ float wm[8];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q4 = lane >> 4, j = lane & 15;

    // operand scales from the producers' partial maxima -- computed AFTER the first rows are in flight (a wave with an
    // empty run does it right away: every wave passes the barrier inside exactly once)
    float sx = 0.f, sg = 0.f;
    bool have_scales = false;
    auto scales = [&]() {
        float mx = 0.f, mg = 0.f;
        for (int i = tid; i < a.xcount; i += 256)
            mx = fmaxf(mx, a.xamax[i]);
        for (int i = tid; i < a.gcount; i += 256)
            mg = fmaxf(mg, a.gamax[i]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mx = fmaxf(mx, __shfl_xor(mx, o, 64));
            mg = fmaxf(mg, __shfl_xor(mg, o, 64));
        }
        if (lane == 0) {
            wm[wave] = mx;
            wm[4 + wave] = mg;
        }
        __syncthreads();
        sx = pow2_scale(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3])));
        sg = pow2_scale(fmaxf(fmaxf(wm[4], wm[5]), fmaxf(wm[6], wm[7])));
        have_scales = true;
    };
    // workgroup -> (tile pair, pixel split): as k_wgrad3x3
    int pair, xsplit, wsplit = 0;
    if (WAVE) {
        const int xcd = blockIdx.x & 7, base = a.npairs >> 3, rem = a.npairs & 7;
        const int mine = base + (xcd < rem ? 1 : 0), first = xcd * base + min(xcd, rem);
        const int job = (int)(blockIdx.x >> 3) * 4 + wave;
        const int pl = a.wave_mode == 2 ? job % mine : job / a.S, sp = a.wave_mode == 2 ? job / mine : job - pl * a.S;
        xsplit = 0;
        if (a.wave_mode == 2 ? sp < a.S : pl < mine) {
            pair = first + pl;
            wsplit = sp;
        } else {                                // no job for this wave: an empty run of pair 0 (it still meets the barrier)
            pair = 0;
            wsplit = a.S;
        }
    } else if (a.rect_mode) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int unit = (slot >> 5) * 8 + xcd, idx = slot & 31;
        const int ncog = a.npairs / a.ncig, rects_i = (a.ncig + a.rect_i - 1) / a.rect_i;
        const int nrect = ((ncog + a.rect_c - 1) / a.rect_c) * rects_i;
        const int rect = unit % nrect;
        xsplit = unit / nrect;
        const int cg = (rect / rects_i) * a.rect_c + idx / a.rect_i, ci = (rect % rects_i) * a.rect_i + idx % a.rect_i;
        if (xsplit >= a.nx || cg >= ncog || ci >= a.ncig)
            return;
        pair = cg * a.ncig + ci;
    } else {
        const int nx8 = a.nx & ~7, main_blocks = nx8 * a.npairs;
        if ((int)blockIdx.x < main_blocks) {
            const int xcd = blockIdx.x & 7, rest = blockIdx.x >> 3;
            pair = rest % a.npairs;
            xsplit = (rest / a.npairs) * 8 + xcd;
        } else {
            const int rest = blockIdx.x - main_blocks;
            pair = rest % a.npairs;
            xsplit = nx8 + rest / a.npairs;
        }
    }
    const int split = WAVE ? wsplit : xsplit * 4 + wave;
    const int cog = pair / a.ncig, cig = pair - cog * a.ncig;
    const int co0 = cog * NCO * 16, ci0 = cig * NCI * 16;
    const size_t plane = (size_t)a.H * a.W;
    bool ci_ok[NCI];
#pragma unroll
    for (int u = 0; u < NCI; ++u)
        ci_ok[u] = ci0 + 16 * u < a.Cin;
    float psc[NCI], psh[NCI];               // PRE: the norm's map of this lane's channel of ci tile u
#pragma unroll
    for (int u = 0; u < NCI; ++u) {
        const int ch = ci_ok[u] ? ci0 + 16 * u + j : ci0 + j;
        psc[u] = PRE ? a.pre_sc[ch] : 1.f;
        psh[u] = PRE ? a.pre_sh[ch] : 0.f;
    }
    auto pre = [&](float v, int u) { return PRE ? fmaxf(__builtin_fmaf(v, psc[u], psh[u]), 0.f) : v; };

    // staging geometry of this lane, per DMA instruction m: which (tile, row, piece) of the slot image it carries
    // (piece = 64 m + lane in [tile][row][piece] order; the tail of the last instruction repeats piece 0)
    unsigned chanA[NIA], chanB[NIB];        // byte offset of the lane's channel row from the image's channel 0
    int pieceA[NIA], pieceB[NIB];           // first pixel of its piece relative to the strip (A: 0 .. 28, B: -4 .. 32)
#pragma unroll
    for (int m = 0; m < NIA; ++m) {
        int P = 64 * m + lane;
        if (P >= NCO * 16 * APIECES)
            P = 0;
        const int row = P / APIECES, pc = min(P - row * APIECES, APIECES - 2);     // (tile, row) flat; pad -> last piece
        chanA[m] = (unsigned)((size_t)(co0 + row) * plane * 4);
        pieceA[m] = 4 * pc;
    }
#pragma unroll
    for (int m = 0; m < NIB; ++m) {
        int P = 64 * m + lane;
        if (P >= NCI * 16 * BPIECES)
            P = 0;
        const int row = P / BPIECES, pc = min(P - row * BPIECES, BPIECES - 2);
        const int u = row >> 4;
        const int ch = (ci0 + 16 * u < a.Cin) ? ci0 + row : ci0 + (row & 15);       // ragged last ci group: tile 0 again
        chanB[m] = (unsigned)((size_t)ch * plane * 4);
        pieceB[m] = 4 * pc - 4;
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char *)smem +
                          (unsigned)wave * (NS * SLOTB);
    const unsigned char *my = smem + wave * (NS * SLOTB);
    // MFMA-order read offsets inside a slot: lane (q4, j) reads pixels 8 q4 .. 8 q4 + 7 of row j
    const unsigned ardo = (unsigned)(j * APIECES + 2 * q4) * 16;
    const unsigned brdo = (unsigned)(NIA * 1024 + (j * BPIECES + 1 + 2 * q4) * 16);

    f32x4 acc[NCO][NCI][9];
#pragma unroll
    for (int t = 0; t < NCO; ++t)
#pragma unroll
        for (int u = 0; u < NCI; ++u)
#pragma unroll
            for (int k = 0; k < 9; ++k)
                acc[t][u][k] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Which rows: split `split` of S over the columns (image, strip) x H rows -- or, a.grp = G > 1 (workgroup form only), the G
    // waves w % G of a workgroup take G ADJACENT strips over the SAME rows (sub-range w / G of the workgroup's share of the
    // "super-columns" (image, G strips)).  An x row piece with its halo touches three 128-byte lines, two of them the neighbouring
    // strips' own: with the neighbours in the same CU at the same time they are L1 / L2 hits instead of a second and third fetch
    // through the fabric (FETCH_SIZE of the 64 -> 64 launch was 2.05x the tensors: 1 + 3 lines per dY + x row against 1 + 1).
    const int G = WAVE ? 1 : a.grp;
    const long long T = (long long)(a.units / G) * a.H;
    const int q = G > 1 ? xsplit * (4 / G) + wave / G : split, Sq = G > 1 ? a.nx * (4 / G) : a.S;
    long long t = min(T, T * q / Sq);
    const long long t1 = min(T, T * (q + 1) / Sq);
    if (t >= t1)
        scales();
    // a.band < H (wave form): a column is `band` rows of a strip and the strips of an image follow each other band by band, so
    // that the halo lines of an x row -- the neighbouring strips' own lines -- are touched again `band` row steps later, while
    // the XCD's L2 still has them, not H row steps later
    const int RB = a.band, nb = a.H / RB;
    while (t < t1) {
        const int colb = (int)(t / RB);                         // (image, band, strip) or, one band, (image, strip)
        const int rr0 = (int)(t - (long long)colb * RB);
        const int len = (int)min((long long)(RB - rr0), t1 - t);
        t += len;
        const int sgs = a.strips / G;
        const int strip = (colb % sgs) * G + (G > 1 ? wave % G : 0);
        const int band = (colb / sgs) % nb;
        const int n = colb / (sgs * nb);
        const int r0 = band * RB + rr0, r1 = r0 + len;
        const int px0 = strip * 32, px = px0 + 8 * q4;
        const bool oct_ok = px < a.W;
        float sx_c, sx_l, sx_r;                                 // set once the scales are known (below)
        // per-lane source offsets of this column (pieces clamped into the row: what falls outside is masked by the scales)
        unsigned offA[NIA], offB[NIB];
#pragma unroll
        for (int m = 0; m < NIA; ++m)
            offA[m] = chanA[m] + 4u * (unsigned)min(max(px0 + pieceA[m], 0), a.W - 4);
#pragma unroll
        for (int m = 0; m < NIB; ++m)
            offB[m] = chanB[m] + 4u * (unsigned)min(max(px0 + pieceB[m], 0), a.W - 4);
        const float *dyn = a.dy + (size_t)n * a.Cout * plane, *xn = a.x + (size_t)n * a.Cin * plane;

        // group g = (dY row g + 1, X row g) -> ring slot s
        auto dma_group = [&](int g, int s) {
            const float *ab = uniform_ptr(dyn + (size_t)min(max(g + 1, 0), a.H - 1) * a.W);
            const float *bb = uniform_ptr(xn + (size_t)min(max(g, 0), a.H - 1) * a.W);
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)s * SLOTB);
#pragma unroll
            for (int m = 0; m < NIA; ++m)
                dma16(ab, offA[m], dst + m * 1024);
#pragma unroll
            for (int m = 0; m < NIB; ++m)
                dma16(bb, offB[m], dst + (NIA + m) * 1024);
        };
        auto read_A = [&](int s, f32x4 (&dst)[NCO][2]) {
            const unsigned char *p = my + s * SLOTB + ardo;
#pragma unroll
            for (int t2 = 0; t2 < NCO; ++t2) {
                dst[t2][0] = *(const f32x4 *)(p + t2 * (16 * APIECES * 16));
                dst[t2][1] = *(const f32x4 *)(p + t2 * (16 * APIECES * 16) + 16);
            }
        };
        auto read_B = [&](int s, f32x4 (&dst)[NCI][2], float (&l)[NCI], float (&rr)[NCI]) {
            const unsigned char *p = my + s * SLOTB + brdo;
#pragma unroll
            for (int u = 0; u < NCI; ++u) {
                const unsigned char *pu = p + u * (16 * BPIECES * 16);
                dst[u][0] = *(const f32x4 *)pu;
                dst[u][1] = *(const f32x4 *)(pu + 16);
                l[u] = *(const float *)(pu - 4);
                rr[u] = *(const float *)(pu + 32);
            }
        };
        auto cvt_A = [&](int y, const f32x4 (&src)[NCO][2], half8 (&dst)[NCO][2]) {
            const float scale = (oct_ok && y >= 0 && y < a.H) ? sg : 0.f;
#pragma unroll
            for (int t2 = 0; t2 < NCO; ++t2) {
                unsigned h[4], l[4];
                split2(src[t2][0].x, src[t2][0].y, scale, h[0], l[0]);
                split2(src[t2][0].z, src[t2][0].w, scale, h[1], l[1]);
                split2(src[t2][1].x, src[t2][1].y, scale, h[2], l[2]);
                split2(src[t2][1].z, src[t2][1].w, scale, h[3], l[3]);
                dst[t2][0] = as_half8(u32x4{h[0], h[1], h[2], h[3]});
                dst[t2][1] = as_half8(u32x4{l[0], l[1], l[2], l[3]});
            }
        };
        auto cvt_B = [&](const f32x4 (&src)[NCI][2], const float (&l)[NCI], const float (&rr)[NCI],
                         half8 (&dst)[3][NCI][2]) {
#pragma unroll
            for (int u = 0; u < NCI; ++u) {
                const float sc = ci_ok[u] ? sx_c : 0.f, sl = ci_ok[u] ? sx_l : 0.f, sr = ci_ok[u] ? sx_r : 0.f;
                unsigned h[5], q[5];
                unsigned hl, ql, hr, qr, hm, qm;
                split1(pre(l[u], u), sl, hl, ql);
                split1(pre(src[u][0].x, u), sc, hm, qm);
                h[0] = __builtin_amdgcn_perm(hm, hl, 0x05040100u);
                q[0] = __builtin_amdgcn_perm(qm, ql, 0x05040100u);
                split2(pre(src[u][0].y, u), pre(src[u][0].z, u), sc, h[1], q[1]);
                split2(pre(src[u][0].w, u), pre(src[u][1].x, u), sc, h[2], q[2]);
                split2(pre(src[u][1].y, u), pre(src[u][1].z, u), sc, h[3], q[3]);
                split1(pre(src[u][1].w, u), sc, hm, qm);
                split1(pre(rr[u], u), sr, hr, qr);
                h[4] = __builtin_amdgcn_perm(hr, hm, 0x05040100u);
                q[4] = __builtin_amdgcn_perm(qr, qm, 0x05040100u);
                unsigned a1h[4], a1l[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a1h[e] = __builtin_amdgcn_alignbit(h[e + 1], h[e], 16);
                    a1l[e] = __builtin_amdgcn_alignbit(q[e + 1], q[e], 16);
                }
                dst[0][u][0] = as_half8(u32x4{h[0], h[1], h[2], h[3]});
                dst[0][u][1] = as_half8(u32x4{q[0], q[1], q[2], q[3]});
                dst[1][u][0] = as_half8(u32x4{a1h[0], a1h[1], a1h[2], a1h[3]});
                dst[1][u][1] = as_half8(u32x4{a1l[0], a1l[1], a1l[2], a1l[3]});
                dst[2][u][0] = as_half8(u32x4{h[1], h[2], h[3], h[4]});
                dst[2][u][1] = as_half8(u32x4{q[1], q[2], q[3], q[4]});
            }
        };

        // Column start: groups r0 - 2 (dY row r0 - 1 only), r0 - 1, r0 fill the ring and are converted; then groups
        // r0 + 1, r0 + 2 are put in flight.  Steady state, step i (X row r = r0 + i, ph = i % 6):
        //   wait until group r + 1 (issued two steps ago) has landed; read it out of slot (i + 1) % 3,
        //   MFMAs of tap row ky = 2 (dY row r - 1); underneath, dY row r + 2 / X row r + 1 are split into fragments,
        //   group r + 3 is issued into slot i % 3 (group r's, read one step ago),
        //   MFMAs of tap rows ky = 1, 0.
        half8 A[3][NCO][2], B[2][3][NCI][2];
        dma_wait<0>();                          // nothing of the previous column is still landing in the ring
        dma_group(r0 - 2, 0);
        dma_group(r0 - 1, 1);
        dma_group(r0, 2);
        if (!have_scales)
            scales();
        sx_c = oct_ok ? sx : 0.f;
        sx_l = (oct_ok && px > 0) ? sx : 0.f;
        sx_r = (px + 8 < a.W) ? sx : 0.f;
        dma_wait<0>();
        {
            f32x4 ra[NCO][2], rb[NCI][2];
            float rl[NCI], rr[NCI];
            read_A(0, ra);
            cvt_A(r0 - 1, ra, A[0]);
            read_A(1, ra);
            cvt_A(r0, ra, A[1]);
            read_A(2, ra);
            read_B(2, rb, rl, rr);
            cvt_A(r0 + 1, ra, A[2]);
            cvt_B(rb, rl, rr, B[0]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the ring has been read: its slots may be refilled
        dma_group(r0 + 1, 1);
        dma_group(r0 + 2, 2);

        auto mfma_row = [&](auto SLOT, auto BSET, auto KY) {
            constexpr int slot = decltype(SLOT)::value, bs = decltype(BSET)::value, ky = decltype(KY)::value;
#pragma unroll
            for (int pass = 0; pass < 3; ++pass)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int t2 = 0; t2 < NCO; ++t2)
#pragma unroll
                        for (int u = 0; u < NCI; ++u)
                            acc[t2][u][ky * 3 + kx] = DCL_WMFMA(A[slot][t2][pass == 2 ? 1 : 0], B[bs][kx][u][pass == 1 ? 1 : 0],
                                acc[t2][u][ky * 3 + kx]);
        };
        auto step = [&](auto PH, int r) {
            constexpr int ph = decltype(PH)::value;
            using I = std::integral_constant<int, ph % 3>;             // register slot of dY row r - 1
            using I1 = std::integral_constant<int, (ph + 1) % 3>;
            using I2 = std::integral_constant<int, (ph + 2) % 3>;
            using BS = std::integral_constant<int, ph % 2>;
            f32x4 ra[NCO][2], rb[NCI][2];
            float rl[NCI], rr[NCI];
            dma_wait<NI>();                                            // group r + 1 is in LDS (group r + 2 may be landing)
            read_A((ph + 1) % 3, ra);
            read_B((ph + 1) % 3, rb, rl, rr);
            mfma_row(I{}, BS{}, std::integral_constant<int, 2>{});
            cvt_A(r + 2, ra, A[ph % 3]);
            cvt_B(rb, rl, rr, B[(ph + 1) % 2]);
            // Group r + 3 goes into slot ph % 3 (group r's, read one step ago).  Its NI LDS-DMA instructions are dealt out
            // one per (pass, kx) sub-block of the remaining two tap rows (18 sub-blocks of NCO x NCI MFMAs), each fenced
            // so that it stays there: as ONE block between the tap rows (~5 NI scalar + vector-memory instructions) they were
            // issued with the matrix pipe idle -- probe builds: time(no DMA) = 0.80 x time(product) on the BasicBlock shapes.
            __builtin_amdgcn_sched_barrier(0);
            if (!g_wgrad_spread) {
                dma_group(r + 3, ph % 3);
                __builtin_amdgcn_sched_barrier(0);
                mfma_row(I1{}, BS{}, std::integral_constant<int, 1>{});
                mfma_row(I2{}, BS{}, std::integral_constant<int, 0>{});
            } else {
                const int g3 = r + 3;
                const float *ab = uniform_ptr(dyn + (size_t)min(max(g3 + 1, 0), a.H - 1) * a.W);
                const float *bb = uniform_ptr(xn + (size_t)min(max(g3, 0), a.H - 1) * a.W);
                const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(ph % 3) * SLOTB);
                static_assert(NI <= 18, "one DMA instruction per sub-block");
#pragma unroll
                for (int sb = 0; sb < 18; ++sb) {
                    const int kyi = sb / 9, pass = (sb % 9) / 3, kx = sb % 3;      // tap row ky = 1, then ky = 0
#pragma unroll
                    for (int t2 = 0; t2 < NCO; ++t2)
#pragma unroll
                        for (int u = 0; u < NCI; ++u) {
                            if (kyi == 0)
                                acc[t2][u][3 + kx] = DCL_WMFMA(A[(ph + 1) % 3][t2][pass == 2 ? 1 : 0],
                                                               B[ph % 2][kx][u][pass == 1 ? 1 : 0], acc[t2][u][3 + kx]);
                            else
                                acc[t2][u][kx] = DCL_WMFMA(A[(ph + 2) % 3][t2][pass == 2 ? 1 : 0],
                                                           B[ph % 2][kx][u][pass == 1 ? 1 : 0], acc[t2][u][kx]);
                        }
                    if (sb < NIA)
                        dma16(ab, offA[sb], dst + sb * 1024);
                    else if (sb < NI)
                        dma16(bb, offB[sb - NIA], dst + sb * 1024);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        int r = r0;
        for (; r + 6 <= r1; r += 6) {
            step(std::integral_constant<int, 0>{}, r);
            step(std::integral_constant<int, 1>{}, r + 1);
            step(std::integral_constant<int, 2>{}, r + 2);
            step(std::integral_constant<int, 3>{}, r + 3);
            step(std::integral_constant<int, 4>{}, r + 4);
            step(std::integral_constant<int, 5>{}, r + 5);
        }
        if (r < r1)
            step(std::integral_constant<int, 0>{}, r);
        if (r + 1 < r1)
            step(std::integral_constant<int, 1>{}, r + 1);
        if (r + 2 < r1)
            step(std::integral_constant<int, 2>{}, r + 2);
        if (r + 3 < r1)
            step(std::integral_constant<int, 3>{}, r + 3);
        if (r + 4 < r1)
            step(std::integral_constant<int, 4>{}, r + 4);
    }
    dma_wait<0>();

    if (LDSRED) {
        __syncthreads();                        // every wave is done with its staging ring: reuse it for the reduction
        float(*red)[NREG][64] = (float(*)[NREG][64])smem;
        auto put = [&](int b) {
#pragma unroll
            for (int t2 = 0; t2 < NCO; ++t2)
#pragma unroll
                for (int u = 0; u < NCI; ++u)
#pragma unroll
                    for (int k = 0; k < 9; ++k)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            red[b][((t2 * NCI + u) * 9 + k) * 4 + q][lane] = acc[t2][u][k][q];
        };
        auto add = [&](int b) {
#pragma unroll
            for (int t2 = 0; t2 < NCO; ++t2)
#pragma unroll
                for (int u = 0; u < NCI; ++u)
#pragma unroll
                    for (int k = 0; k < 9; ++k)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            acc[t2][u][k][q] += red[b][((t2 * NCI + u) * 9 + k) * 4 + q][lane];
        };
        if (wave & 1)
            put(wave >> 1);
        __syncthreads();
        if (!(wave & 1))
            add(wave >> 1);
        __syncthreads();
        if (wave == 2)
            put(0);
        __syncthreads();
        if (wave != 0)
            return;
        add(0);
    } else if (split >= a.S)
        return;
    const float inv = 1.0f / (sx * sg);
    float *out = a.part + (size_t)(LDSRED ? xsplit : split) * 9 * a.Cout * a.Cin;
#pragma unroll
    for (int t2 = 0; t2 < NCO; ++t2)
#pragma unroll
        for (int u = 0; u < NCI; ++u)
#pragma unroll
            for (int k = 0; k < 9; ++k)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int co = co0 + 16 * t2 + 4 * q4 + q, ci = ci0 + 16 * u + j;
                    if (ci_ok[u])
                        out[((size_t)k * a.Cout + co) * a.Cin + ci] = acc[t2][u][k][q] * inv;
                }
}


// ---- 1x1 convolution: dw[co][ci] = sum_{n, y, x} dy[n, co, y, x] * x[n, ci, y, x] -----------------------------------
// Same staging and work split with one tap: no halo pieces, no dY ring; a wave owns NCO x NCI tiles (up to 4 x 4) and per
// row step issues 3 NCO NCI MFMAs on the fragments converted one step earlier while the next row is being split.
struct Wgrad1Args {
    const float *x, *dy;
    float *part;                 // [nx][Cout][Cin]
    const float *xamax, *gamax;
    int xcount, gcount;
    int N, Cin, Cout, H, W;
    int strips, units, S, ncig, npairs, nx;
};

template <int NCO, int NCI>
__global__ __launch_bounds__(256, 1) void k_wgrad1x1d(Wgrad1Args a)
{
    constexpr int NIA = (NCO * 16 * APIECES + 63) / 64, NIB = (NCI * 16 * APIECES + 63) / 64, NI = NIA + NIB;
    // ring slots: rows are fetched NS - 1 steps ahead.  The steps are short (one tap) and the kernel is bound by HBM, not
    // by the matrix pipe: what matters is the volume in flight, 4 waves x (NS - 1) x NI KiB per CU
    constexpr int NS = NI > 13 ? 2 : 3, D = NS - 1;
    constexpr int SLOTB = NI * 1024, STAGEB = 4 * NS * SLOTB;
    constexpr int NREG = NCO * NCI * 4;
    constexpr int REDB = 2 * NREG * 64 * 4;
    constexpr int SMEMB = STAGEB > REDB ? STAGEB : REDB;
    static_assert(SMEMB + 64 <= 160 * 1024, "LDS");
    static_assert(D * NI < 64, "vmcnt");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEMB];
    __shared__ float wm[8];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q4 = lane >> 4, j = lane & 15;

    // operand scales from the producers' partial maxima -- computed AFTER the first rows are in flight (a wave with an
    // empty run does it right away: every wave passes the barrier inside exactly once)
    float sx = 0.f, sg = 0.f;
    bool have_scales = false;
    auto scales = [&]() {
        float mx = 0.f, mg = 0.f;
        for (int i = tid; i < a.xcount; i += 256)
            mx = fmaxf(mx, a.xamax[i]);
        for (int i = tid; i < a.gcount; i += 256)
            mg = fmaxf(mg, a.gamax[i]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mx = fmaxf(mx, __shfl_xor(mx, o, 64));
            mg = fmaxf(mg, __shfl_xor(mg, o, 64));
        }
        if (lane == 0) {
            wm[wave] = mx;
            wm[4 + wave] = mg;
        }
        __syncthreads();
        sx = pow2_scale(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3])));
        sg = pow2_scale(fmaxf(fmaxf(wm[4], wm[5]), fmaxf(wm[6], wm[7])));
        have_scales = true;
    };
    int pair, xsplit;
    {
        const int nx8 = a.nx & ~7, main_blocks = nx8 * a.npairs;
        if ((int)blockIdx.x < main_blocks) {
            const int xcd = blockIdx.x & 7, rest = blockIdx.x >> 3;
            pair = rest % a.npairs;
            xsplit = (rest / a.npairs) * 8 + xcd;
        } else {
            const int rest = blockIdx.x - main_blocks;
            pair = rest % a.npairs;
            xsplit = nx8 + rest / a.npairs;
        }
    }
    const int split = xsplit * 4 + wave;
    const int cog = pair / a.ncig, cig = pair - cog * a.ncig;
    const int co0 = cog * NCO * 16, ci0 = cig * NCI * 16;
    const size_t plane = (size_t)a.H * a.W;

    unsigned chanA[NIA], chanB[NIB];
    int pieceA[NIA], pieceB[NIB];
#pragma unroll
    for (int m = 0; m < NIA; ++m) {
        int P = 64 * m + lane;
        if (P >= NCO * 16 * APIECES)
            P = 0;
        const int row = P / APIECES, pc = min(P - row * APIECES, APIECES - 2);
        chanA[m] = (unsigned)((size_t)(co0 + row) * plane * 4);
        pieceA[m] = 4 * pc;
    }
#pragma unroll
    for (int m = 0; m < NIB; ++m) {
        int P = 64 * m + lane;
        if (P >= NCI * 16 * APIECES)
            P = 0;
        const int row = P / APIECES, pc = min(P - row * APIECES, APIECES - 2);
        chanB[m] = (unsigned)((size_t)(ci0 + row) * plane * 4);
        pieceB[m] = 4 * pc;
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char *)smem +
                          (unsigned)wave * (NS * SLOTB);
    const unsigned char *my = smem + wave * (NS * SLOTB);
    const unsigned ardo = (unsigned)(j * APIECES + 2 * q4) * 16;
    const unsigned brdo = (unsigned)(NIA * 1024) + ardo;

    f32x4 acc[NCO][NCI];
#pragma unroll
    for (int t = 0; t < NCO; ++t)
#pragma unroll
        for (int u = 0; u < NCI; ++u)
            acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};

    const long long T = (long long)a.units * a.H;
    long long t = min(T, T * split / a.S);
    const long long t1 = min(T, T * (split + 1) / a.S);
    if (t >= t1)
        scales();
    while (t < t1) {
        const int col = (int)(t / a.H);
        const int r0 = (int)(t - (long long)col * a.H);
        const int r1 = (int)min((long long)a.H, r0 + (t1 - t));
        t += r1 - r0;
        const int strip = col % a.strips;
        const int n = col / a.strips;
        const int px0 = strip * 32, px = px0 + 8 * q4;
        const bool oct_ok = px < a.W;
        float sx_c, sg_c;
        unsigned offA[NIA], offB[NIB];
#pragma unroll
        for (int m = 0; m < NIA; ++m)
            offA[m] = chanA[m] + 4u * (unsigned)min(px0 + pieceA[m], a.W - 4);
#pragma unroll
        for (int m = 0; m < NIB; ++m)
            offB[m] = chanB[m] + 4u * (unsigned)min(px0 + pieceB[m], a.W - 4);
        const float *dyn = a.dy + (size_t)n * a.Cout * plane, *xn = a.x + (size_t)n * a.Cin * plane;

        auto dma_group = [&](int g, int s) {           // rows g of dY and X -> ring slot s
            const size_t ro = (size_t)min(g, a.H - 1) * a.W;
            const float *ab = uniform_ptr(dyn + ro), *bb = uniform_ptr(xn + ro);
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)s * SLOTB);
#pragma unroll
            for (int m = 0; m < NIA; ++m)
                dma16(ab, offA[m], dst + m * 1024);
#pragma unroll
            for (int m = 0; m < NIB; ++m)
                dma16(bb, offB[m], dst + (NIA + m) * 1024);
        };
        auto read_rows = [&](int s, f32x4 (&da)[NCO][2], f32x4 (&db)[NCI][2]) {
            const unsigned char *p = my + s * SLOTB;
#pragma unroll
            for (int t2 = 0; t2 < NCO; ++t2) {
                da[t2][0] = *(const f32x4 *)(p + ardo + t2 * (16 * APIECES * 16));
                da[t2][1] = *(const f32x4 *)(p + ardo + t2 * (16 * APIECES * 16) + 16);
            }
#pragma unroll
            for (int u = 0; u < NCI; ++u) {
                db[u][0] = *(const f32x4 *)(p + brdo + u * (16 * APIECES * 16));
                db[u][1] = *(const f32x4 *)(p + brdo + u * (16 * APIECES * 16) + 16);
            }
        };
        auto cvt8 = [&](const f32x4 (&src)[2], float scale, half8 (&dst)[2]) {
            unsigned h[4], l[4];
            split2(src[0].x, src[0].y, scale, h[0], l[0]);
            split2(src[0].z, src[0].w, scale, h[1], l[1]);
            split2(src[1].x, src[1].y, scale, h[2], l[2]);
            split2(src[1].z, src[1].w, scale, h[3], l[3]);
            dst[0] = as_half8(u32x4{h[0], h[1], h[2], h[3]});
            dst[1] = as_half8(u32x4{l[0], l[1], l[2], l[3]});
        };

        half8 FA[2][NCO][2], FB[2][NCI][2];
        dma_wait<0>();
#pragma unroll
        for (int g = 0; g <= D; ++g)
            dma_group(r0 + g, g);
        if (!have_scales)
            scales();
        sx_c = oct_ok ? sx : 0.f;
        sg_c = oct_ok ? sg : 0.f;
        dma_wait<D * NI>();
        {
            f32x4 ra[NCO][2], rb[NCI][2];
            read_rows(0, ra, rb);
#pragma unroll
            for (int t2 = 0; t2 < NCO; ++t2)
                cvt8(ra[t2], sg_c, FA[0][t2]);
#pragma unroll
            for (int u = 0; u < NCI; ++u)
                cvt8(rb[u], sx_c, FB[0][u]);
        }
        // step i (row r = r0 + i, ph = i % 6): fragments of row r are in set ph % 2; group r + 1 sits in slot
        // (i + 1) % NS, group r + NS goes to slot i % NS (row r's, read one step ago)
        auto step = [&](auto PH, int r) {
            constexpr int ph = decltype(PH)::value, fs = ph % 2;
            f32x4 ra[NCO][2], rb[NCI][2];
            dma_wait<(D - 1) * NI>();
            read_rows((ph + 1) % NS, ra, rb);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // slot ph % NS was read a step ago; this one now
            dma_group(r + NS, ph % NS);
#pragma unroll
            for (int pass = 0; pass < 3; ++pass)
#pragma unroll
                for (int t2 = 0; t2 < NCO; ++t2)
#pragma unroll
                    for (int u = 0; u < NCI; ++u)
                        acc[t2][u] = DCL_WMFMA(FA[fs][t2][pass == 2 ? 1 : 0],
                                                                            FB[fs][u][pass == 1 ? 1 : 0], acc[t2][u]);
#pragma unroll
            for (int t2 = 0; t2 < NCO; ++t2)
                cvt8(ra[t2], sg_c, FA[fs ^ 1][t2]);
#pragma unroll
            for (int u = 0; u < NCI; ++u)
                cvt8(rb[u], sx_c, FB[fs ^ 1][u]);
        };
        int r = r0;
        for (; r + 6 <= r1; r += 6) {
            step(std::integral_constant<int, 0>{}, r);
            step(std::integral_constant<int, 1>{}, r + 1);
            step(std::integral_constant<int, 2>{}, r + 2);
            step(std::integral_constant<int, 3>{}, r + 3);
            step(std::integral_constant<int, 4>{}, r + 4);
            step(std::integral_constant<int, 5>{}, r + 5);
        }
        if (r < r1)
            step(std::integral_constant<int, 0>{}, r);
        if (r + 1 < r1)
            step(std::integral_constant<int, 1>{}, r + 1);
        if (r + 2 < r1)
            step(std::integral_constant<int, 2>{}, r + 2);
        if (r + 3 < r1)
            step(std::integral_constant<int, 3>{}, r + 3);
        if (r + 4 < r1)
            step(std::integral_constant<int, 4>{}, r + 4);
    }
    dma_wait<0>();
    __syncthreads();
    {
        float(*red)[NREG][64] = (float(*)[NREG][64])smem;
        auto put = [&](int b) {
#pragma unroll
            for (int t2 = 0; t2 < NCO; ++t2)
#pragma unroll
                for (int u = 0; u < NCI; ++u)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        red[b][(t2 * NCI + u) * 4 + q][lane] = acc[t2][u][q];
        };
        auto add = [&](int b) {
#pragma unroll
            for (int t2 = 0; t2 < NCO; ++t2)
#pragma unroll
                for (int u = 0; u < NCI; ++u)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        acc[t2][u][q] += red[b][(t2 * NCI + u) * 4 + q][lane];
        };
        if (wave & 1)
            put(wave >> 1);
        __syncthreads();
        if (!(wave & 1))
            add(wave >> 1);
        __syncthreads();
        if (wave == 2)
            put(0);
        __syncthreads();
        if (wave != 0)
            return;
        add(0);
    }
    const float inv = 1.0f / (sx * sg);
    float *out = a.part + (size_t)xsplit * a.Cout * a.Cin;
#pragma unroll
    for (int t2 = 0; t2 < NCO; ++t2)
#pragma unroll
        for (int u = 0; u < NCI; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                out[(size_t)(co0 + 16 * t2 + 4 * q4 + q) * a.Cin + ci0 + 16 * u + j] = acc[t2][u][q] * inv;
}

// dw[i] = sum_s part[s][i], slabs added in fixed order (8 partial sums combined in order through LDS)
__global__ __launch_bounds__(256) void k_slab_sum(const float *__restrict__ part, int S, int total, float *__restrict__ dw)
{
    __shared__ float sh[8][32];
    const int lane = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int idx = blockIdx.x * 32 + lane;
    float s0 = 0.f, s1 = 0.f;
    if (idx < total) {
        int k = g;
        for (; k + 8 < S; k += 16) {
            s0 += part[(size_t)k * total + idx];
            s1 += part[(size_t)(k + 8) * total + idx];
        }
        if (k < S)
            s0 += part[(size_t)k * total + idx];
    }
    sh[g][lane] = s0 + s1;
    __syncthreads();
    if (g == 0 && idx < total) {
        float s = sh[0][lane];
#pragma unroll
        for (int q = 1; q < 8; ++q)
            s += sh[q][lane];
        dw[idx] = s;
    }
}

struct W1Plan {
    int nco, nci, ncig, npairs, units, S, nx;
};

W1Plan w1_plan(int N, int Cin, int Cout, int H, int W)
{
    const int cot = Cout / 16, cit = Cin / 16;
    W1Plan p;
    p.nco = cot % 4 == 0 ? 4 : (cot % 3 == 0 ? 3 : (cot % 2 == 0 ? 2 : 1));
    p.nci = cit % 4 == 0 ? 4 : (cit % 3 == 0 ? 3 : (cit % 2 == 0 ? 2 : 1));
    if (p.nco == 4 && p.nci == 4)                   // the staging ring of a (4, 4) wave does not fit: 4 waves x 2 x 20 KiB
        (cot >= cit ? p.nci : p.nco) = 2;
    p.ncig = cit / p.nci;
    p.npairs = (cot / p.nco) * p.ncig;
    p.units = N * ((W + 31) / 32);
    p.nx = 256 / p.npairs;
    if (p.nx < 1)
        p.nx = 1;
    p.S = 4 * p.nx;
    if ((long long)p.S > (long long)p.units * H)
        p.S = p.units * H;
    p.nx = (p.S + 3) / 4;
    return p;
}

}  // namespace

bool dcl_wgrad_dma_supported(int nco, int nci) { return nco >= 1 && nco <= 3 && nci >= 1 && nci <= 2; }
bool dcl_wgrad_dma_wave_mode_supported(int nco, int nci) { return nco == 3 && nci == 1; }

void dcl_wgrad_dma_launch(const WgradArgs &a, int nco, int nci, dim3 grid, hipStream_t s)
{
    if (a.pre_sc) {             // the input operand through the producer norm's map + ReLU (k_wgrad3x3d, PRE)
        if (a.wave_mode) {
            hipLaunchKernelGGL((k_wgrad3x3d<3, 1, true, true>), dim3(256), dim3(256), 0, s, a);
            dcl_note_kernel("k_wgrad3x3d_pre<3,1,true>");
            return;
        }
#define DCL_WGD_CASE(o, i)       \
    if (nco == o && nci == i)    \
        hipLaunchKernelGGL((k_wgrad3x3d<o, i, false, true>), grid, dim3(256), 0, s, a);
        DCL_WGD_CASE(2, 2)
        DCL_WGD_CASE(1, 2)
        DCL_WGD_CASE(3, 1)
        DCL_WGD_CASE(2, 1)
        DCL_WGD_CASE(1, 1)
#undef DCL_WGD_CASE
        dcl_note_kernel("k_wgrad3x3d_pre<%d,%d,false>", nco, nci);
        return;
    }
    if (a.wave_mode) {
        hipLaunchKernelGGL((k_wgrad3x3d<3, 1, true>), dim3(256), dim3(256), 0, s, a);
        dcl_note_kernel("k_wgrad3x3d<3,1,true>");
        return;
    }
#define DCL_WGD_CASE(o, i)       \
    if (nco == o && nci == i)    \
        hipLaunchKernelGGL((k_wgrad3x3d<o, i>), grid, dim3(256), 0, s, a);
    DCL_WGD_CASE(2, 2)
    DCL_WGD_CASE(1, 2)
    DCL_WGD_CASE(3, 1)
    DCL_WGD_CASE(2, 1)
    DCL_WGD_CASE(1, 1)
#undef DCL_WGD_CASE
    dcl_note_kernel("k_wgrad3x3d<%d,%d,false>", nco, nci);
}

extern "C" int dcl_wgrad1x1_splits(int N, int Cin, int Cout, int H, int W)
{
    if (N <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || (Cin & 15) || (Cout & 15) || (W & 7))
        return 0;
    return w1_plan(N, Cin, Cout, H, W).nx;
}

extern "C" int dcl_wgrad1x1_f16x3(const float *x, const float *dy, int N, int Cin, int Cout, int H, int W,
                                  const float *xamax, int xcount, const float *gamax, int gcount, float *part, float *dw,
                                  void *stream)
{
    DCL_CHECK_ARG(x && dy && xamax && gamax && part && dw, "null pointer");
    DCL_CHECK_ARG(N > 0 && H > 0 && W > 0 && xcount > 0 && gcount > 0, "bad shape");
    DCL_CHECK_ARG(Cin > 0 && Cout > 0 && (Cin & 15) == 0 && (Cout & 15) == 0, "channel counts must be multiples of 16");
    DCL_CHECK_ARG((W & 7) == 0, "W must be a multiple of 8");
    DCL_CHECK_ARG((size_t)(Cin > Cout ? Cin : Cout) * H * W * 4 < ((size_t)1 << 32), "image too large");
    DCL_CHECK_ARG(((((uintptr_t)x) | ((uintptr_t)dy)) & 15) == 0, "tensors must be 16-byte aligned");
    const W1Plan p = w1_plan(N, Cin, Cout, H, W);
    Wgrad1Args a;
    a.x = x; a.dy = dy; a.part = part; a.xamax = xamax; a.gamax = gamax; a.xcount = xcount; a.gcount = gcount;
    a.N = N; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
    a.strips = (W + 31) / 32;
    a.units = p.units; a.S = p.S; a.ncig = p.ncig; a.npairs = p.npairs; a.nx = p.nx;
    const dim3 grid((unsigned)(p.npairs * p.nx));
    hipStream_t s = (hipStream_t)stream;
#define DCL_W1_CASE(o, i)            \
    if (p.nco == o && p.nci == i)    \
        hipLaunchKernelGGL((k_wgrad1x1d<o, i>), grid, dim3(256), 0, s, a);
#define DCL_W1_ROW(o) DCL_W1_CASE(o, 1) DCL_W1_CASE(o, 2) DCL_W1_CASE(o, 3)
    DCL_W1_ROW(1)
    DCL_W1_ROW(2)
    DCL_W1_ROW(3)
    DCL_W1_ROW(4)
    DCL_W1_CASE(1, 4)
    DCL_W1_CASE(2, 4)
    DCL_W1_CASE(3, 4)
#undef DCL_W1_ROW
#undef DCL_W1_CASE
    dcl_note_kernel("k_wgrad1x1d<%d,%d>", p.nco, p.nci);
    DCL_LAUNCH_CHECK();
    const int total = Cout * Cin;
    hipLaunchKernelGGL(k_slab_sum, dim3((total + 31) / 32), dim3(256), 0, s, part, p.nx, total, dw);
    DCL_LAUNCH_CHECK();
    return 0;
}
