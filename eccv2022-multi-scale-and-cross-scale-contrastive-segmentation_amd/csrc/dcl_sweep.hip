// dcl_sweep.hip -- K4 / K5: fused similarity + masked InfoNCE forward and backward on the f32
// matrix cores of gfx950 (v_mfma_f32_32x32x2_f32: exact fp32, 157 TFLOP/s dense peak).
//
// One kernel template, three modes, one structure:
//   * a workgroup = 4 waves owns DCL_ROW_TILE = 128 anchor rows; each wave keeps ITS 32 rows of A
//     (32 x 256 f32) in 128 VGPRs for the whole sweep -- A is read from HBM/L2 exactly once;
//   * the contrast bank B streams through LDS in chunks of 32 rows (32 x 1 KiB, +16 B row pad so the
//     ds_read_b128 operand fetch is bank-conflict free), double buffered and filled by LDS-DMA
//     (global_load_lds_dwordx4: one 1-KiB wave-instruction = one bank row), one barrier per chunk;
//   * the similarity tile is computed TRANSPOSED, X[j][i] = <B_j, A_i>, so the MFMA result has the
//     anchor i on the lane and the 32 columns j in the 16 accumulator registers x 2 lane halves:
//     row reductions over j are in-lane adds (no shuffles), and the accumulator registers are,
//     as they stand, the A-operand of the second product dA[i][:] += H[j][i] * B[j][:] (backward);
//   * the N1 x N2 similarity matrix never exists in memory.
//
// Replaces (reference, /root/reference): losses/DenseContrastiveLossV2.py:150 (matmul/div),
// :154-171 (get_masks2), :173-192 (get_loss); losses/DenseContrastiveLossV2_ms.py:114, :118-130,
// :132-161; and the autograd backward of all of them.
#include <limits.h>

#include <mutex>
#include <type_traits>
#include <unordered_map>

#include "dcl_common.h"

namespace {

// Bound probes (tools/probes/sweep_ab.sh builds variants of the library with -DDCL_SWEEP_PROBE=<bits>, sweep_ab.py times them next to
// the shipped one in one process; 0 in the product; results wrong, times meaningful).  Forward sweeps: 1 = no chunk DMA after the
// first, 2 = no exp epilogue, 4 = no LDS operand reads, 8 = one A row for every lane (no strided panel loads).  Pipelined backward:
// 16 = every chunk DMA re-reads chunk 0, 32 = no exp / split epilogue, 64 = no transposing reads, 128 = no S operand reads,
// 256 = no workgroup barrier per chunk.  Round 5, N = 9 804 (profiles/r05_sweep_probes.txt): forward 148 us -> 130 / 146 / 146 /
// 119, all four 98; backward 310 us -> 303 / 282 / 297 / 270 / 311, all 241: no single resource gates either sweep, the bare MFMA
// stream is at 0.49-0.60 of the nominal f16x3 roofline (the clock drops to ~1.75 GHz in these loops).
#ifndef DCL_SWEEP_PROBE
#define DCL_SWEEP_PROBE 0
#endif
constexpr int CP = DCL_CP;              // 256 channels (padded)
constexpr int ROWF = CP + 4;            // LDS row stride in floats (1040 B)
constexpr int CJ = 32;                  // bank rows (similarity columns) per chunk
constexpr int BM = DCL_ROW_TILE;        // anchor rows per workgroup
constexpr int BUF_FLOATS = CJ * ROWF;   // 8320 floats = 33,280 B per buffer

enum { MODE_Z = 0, MODE_POS = 1, MODE_BWD = 2 };

// f16x3 mode: the similarity product runs on the f16 matrix cores at fp32-equivalent accuracy.  Every
// normalised embedding x (|x| <= 1) is stored as two halves, hi = f16(x * 2^10) and lo = f16(x * 2^10 - hi)
// (22 mantissa bits together), and <a, b> = (hi_a.hi_b + hi_a.lo_b + lo_a.hi_b) / 2^20 with f32 accumulation:
// three v_mfma_f32_32x32x16_f16 (32 cycles, K = 16 each) replace eight v_mfma_f32_32x32x2_f32 (64 cycles,
// K = 2 each) -- 5.3x fewer matrix-pipe cycles; measured |dS| 8.3e-6 vs 7.2e-6 for plain fp32.
// The dropped lo.lo term is 2^-22 relative.  A bank row in this format is [256 hi | 256 lo] halves = 1 KiB,
// the same size as the f32 row, so staging and LDS geometry are unchanged.
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __attribute__((address_space(3))) fp16x4 LDS_FP16X4;
constexpr int ROWH = 2 * ROWF;          // LDS row stride in halves (1040 B)
constexpr float F16_SCALE_SQ_INV = 1.0f / 1048576.0f;   // 2^-20

struct SweepArgs {
    const float *A;        // [N1pad, CP]
    const float *B;        // [N2pad, CP]
    const _Float16 *Ah;    // f16x3 mode: [N1pad, 2*CP] halves (hi | lo), else unused
    const _Float16 *Bh;
    int N1, N2, V1;
    const int32_t *rng_lo; // per A slot: positive column range [lo, hi) in B
    const int32_t *rng_hi;
    float inv_tau, c1;     // c1 = inv_tau * log2(e)
    int intra;
    int nsplit;
    float *zpart;          // MODE_Z out / MODE_POS in: [nsplit, N1pad]
    int zsplits;           // MODE_POS: number of zpart slabs to sum
    int accumulate;        // MODE_POS: add to rowloss / W instead of overwriting (multi-segment banks)
    float *Z, *rowloss, *W;        // MODE_POS out: [N1pad]
    const float *rstat, *cstat;    // MODE_BWD in: [N1pad,4], [N2pad,4] = {Z, cW, cZ, 0}
    int use_row, use_col;
    float *dpart;          // MODE_BWD out: [nsplit, N1pad, CP]  (stream-K: [N1pad, CP], the finished gradient)
    // stream-K backward (SK): gridDim.x persistent workgroups share the (row block, chunk) sequence
    int N1pad;             // rows of the padded anchor bank
    float *sk_ws;          // [gridDim.x][BM][CP] partial tiles of the workgroups whose range ends inside a row block
    int *sk_err;           // the error word (hand-overs that timed out): word 0 of the caller's flags buffer, whatever the grid size
    int *sk_flags;         // [gridDim.x] (the caller's buffer from word 1): flags[g] == sk_seq: workgroup g's partial tile of THIS launch
                           // is in sk_ws (any other value = not yet: stale values of earlier or aborted launches never match)
                           // word (number of hand-overs that timed out, ever; never reset by the kernel)
    int sk_nslice;         // 1 | 4 | 8 column slices (see k_sweep, SK): slice s writes its own finished slab dpart[s][N1pad][CP]
    int sk_seq;            // launch number of this (flags, workspace) pair, never 0 (host: dcl_infonce_bwd_streamk)
    long long sk_timeout;  // s_memrealtime ticks (100 MHz) an owner waits for one partial tile before it gives up
    int sk_probe;          // timing probe (dcl_infonce_set_streamk(2)): no flag traffic, no waiting -- WRONG results
    // MODE_Z, optional: keep the raw similarities of the POSITIVE columns, spos[i * spos_ld + (j - lo_i)] (row i's positive range
    // is contiguous: class-major bank).  k_pos_finish then evaluates the positives' terms from them once Z_i is complete --
    // the MODE_POS sweep (a second pass over ~1/K of the similarity matrix: A panels, chunks, MFMAs) disappears.
    float *spos;
    int spos_ld;
};

// One LDS-DMA wave-instruction: 64 lanes x 16 B from per-lane global addresses to the wave-uniform LDS address
// `lds_dst` + lane * 16.  Issued through inline asm ON PURPOSE: hipcc tracks a builtin LDS-DMA as a pending LDS write
// and puts `s_waitcnt vmcnt(0)` in front of the next ds_read that may alias it -- i.e. directly behind the prefetch,
// which exposes the whole DMA latency in every chunk (measured: 41 % of the backward sweep's wave cycles parked in
// s_waitcnt).  An asm DMA is invisible to that bookkeeping; its completion is counted by hand (`dma_wait<N>()` +
// `s_barrier`, then the reads), cdna_hip_programming.md "Pipelining across barriers".
__device__ __forceinline__ void dma_row(const void *gsrc, const float *lds_dst)
{
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane(
        (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float *)lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(dst)
                 : "memory");
}

// wait until at most N of this wave's vector-memory operations (the asm DMAs: the loops below issue no other ones
// that matter) are outstanding and every LDS read has returned, then the workgroup barrier
template <int N>
__device__ __forceinline__ void dma_wait_barrier()
{
    static_assert(N >= 0 && N < 16, "vmcnt immediate");
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

__device__ __forceinline__ void stage_chunk(float *buf, const float *B, int j0, int wave, int lane)
{
    // 32 bank rows per chunk, 8 per wave; one LDS-DMA wave-instruction moves one 1-KiB row:
    // LDS destination = wave-uniform row base + lane * 16 B, global source is per lane.
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int row = wave * 8 + r;
        dma_row(B + (size_t)(j0 + row) * CP + lane * 4, buf + row * ROWF);
    }
}

__device__ __forceinline__ void stage_chunk_h(float *buf, const _Float16 *Bh, int j0, int wave, int lane)
{
    // same geometry as stage_chunk: one 1-KiB wave-instruction per bank row ([256 hi | 256 lo] halves)
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int row = wave * 8 + r;
        dma_row(Bh + (size_t)(j0 + row) * (2 * CP) + lane * 8, buf + row * ROWF);
    }
}

__device__ __forceinline__ int jrow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// Packed f16 pair (low half = first value) of hi = f16(v * s) and lo = f16(v * s - hi) for two values; s is a power of
// two, so v * s is exact and the fused form is the same number.  v_fma_mix{lo,hi}_f16: f32 / f16 inputs, f32
// arithmetic, one f16 half of the destination written -- 2 VALU per value, no separate packing (the compiler's own
// lowering of the C expression takes 3+ and a v_pack).
__device__ __forceinline__ void split2(float v0, float v1, float s, unsigned &hi, unsigned &lo)
{
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hi) : "v"(v0), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hi) : "v"(v1), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=&v"(lo) : "v"(v0), "v"(s), "v"(hi));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(v1), "v"(s), "v"(hi));
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// LDS operand fetch, software-pipelined by hand: the reads for step q+1 are issued right after the
// FIRST MFMA of step q, so they land under the remaining MFMAs of that step (sched_group_barrier pins
// "1 MFMA, n DS reads, rest of the MFMAs").  Left to itself hipcc emits read -> s_waitcnt lgkmcnt(0)
// -> MFMAs, which at one wave per SIMD (backward) exposes the LDS latency in every group (measured:
// 67-71 % of the f32 MFMA peak before this change).
#define DCL_SCHED_MFMA_DS_MFMA(n_ds, n_mfma)                     \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);           \
    __builtin_amdgcn_sched_group_barrier(0x100, (n_ds), 0);      \
    __builtin_amdgcn_sched_group_barrier(0x008, (n_mfma) - 1, 0)

// SK (backward, f16x3): "stream-K" partition.  The (row block, chunk) units of the launch form ONE sequence, row block
// major; gridDim.x persistent workgroups (one per CU) take equal contiguous ranges of it.  A range covers at most the tail
// of one row block, whole row blocks, and the head of another: per piece ("segment") the workgroup loads that row block's A
// panel and runs the chunk pipeline.  The workgroup that finishes a row block (its segment contains the last chunk)
// owns the result: it waits for the partial tiles of the workgroups that covered the earlier chunks of the row block --
// always LOWER workgroup ids, which the dispatcher starts first (the wait is bounded all the same, see below) -- adds them in ascending id
// order (fixed order: bitwise reproducible) and writes the finished 128 x 256 tile.  Every other segment is the last one of
// its workgroup and leaves one partial tile in sk_ws[g].  Against one slab per (row block, column split) that is
// <= 255 partial tiles of 128 KiB instead of 13 x 77 (131 MB written, then read again by K6), one or two A-panel
// prologues per CU instead of four, and no tail round.
// PF (pipelined f16x3 backward): how many steps ahead of the MFMAs that consume them the LDS operand reads are issued.
// At one wave per SIMD nothing hides an LDS read that is not back when its MFMAs are due, and a step is only 96
// matrix-pipe cycles (3 MFMAs): with PF = 1 the reads of step q + 1 have those 96 cycles, less than the LDS round trip
// while four waves keep the LDS pipe ~70 % busy.
template <int MODE, bool USE_COL, bool F16, bool SK = false, int PF = 1>
__global__ __launch_bounds__(256, MODE == MODE_BWD ? 1 : 2) void k_sweep(SweepArgs p)
{
    static_assert(!SK || (MODE == MODE_BWD && F16), "stream-K: f16x3 backward only");
    // per buffer: one 32-row chunk of the contrast bank, f32 rows or (hi | lo) half rows (same 1040-B stride);
    // in f16x3 mode both products of the backward read the half rows
    constexpr int BUF = BUF_FLOATS;
    // f16x3 backward: software-pipelined over chunks (S recompute of chunk c + 1 under the epilogue of chunk c), four
    // buffers: chunk c (second product), chunk c + 1 (S recompute), chunks c + 2 and c + 3 (LDS-DMA in flight)
    constexpr bool PIPE = (MODE == MODE_BWD) && F16;
    constexpr int NBUF = PIPE ? 4 : 2;
    constexpr int CSTF = 256;                      // floats per column-statistics piece (one 1-KiB DMA: 64 rows x 16 B)
    // PIPE image: rows of exactly 1 KiB (no pad), 16-byte slots XOR-swizzled by the row (below); else padded rows
    constexpr int PBUF = CJ * CP;                  // floats per PIPE buffer (32 KiB)
    __shared__ __attribute__((aligned(1024))) float lds[PIPE ? NBUF * PBUF + (USE_COL ? NBUF * CSTF : 0) : NBUF * BUF];

    const int tid = threadIdx.x, lane0 = tid & 63;
    // (SK: the wave index as a scalar, so that what the compiler derives from it and hoists out of the segment loop sits
    // in SGPRs; the other instantiations keep the code they were tuned with)
    const int wave = SK ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6);
    // stream-K: this workgroup's range [su, su1) of the unit sequence (unit = (row block, chunk), row block major)
    // SK, column slices (p.sk_nslice = 4 | 8): the chunk axis is cut into slices and every slice is swept by the workgroups of
    // ONE XCD (8 slices) or one pair of XCDs (4) -- workgroup ids go round-robin over the 8 XCDs, so "XCD of workgroup g" is
    // g & 7 -- as a stream-K problem of its own (local workgroup index sk_j of sk_G, slice-local chunk count sk_nchunk, global
    // chunks [sk_cs0, sk_cs0 + sk_nchunk)).  A slice of the (hi | lo) bank is N2 / nslice KiB: 2.5 MB at N2 = 9 804 and 4 slices,
    // resident in the 4-MiB L2 of the XCDs that sweep it, so the bank is fetched through the fabric once per XCD instead of once
    // per workgroup (FETCH_SIZE 283 MB per launch for a 10-MB bank with one slice, profiles/r03_loss_pmc_fetch.csv).  Each slice
    // leaves its own finished [N1pad, CP] slab; K6 adds the slabs in slice order (fixed order: bitwise reproducible).
    int sk_slice = 0, sk_j = blockIdx.x, sk_G = gridDim.x, sk_cs0 = 0;
    int sk_nchunk = (p.N2 + CJ - 1) / CJ;
    if (SK && p.sk_nslice > 1) {
        const int g = blockIdx.x, nct = sk_nchunk;
        if (p.sk_nslice == 8) {
            sk_slice = g & 7;
            sk_j = g >> 3;
        } else {
            sk_slice = (g & 7) >> 1;
            sk_j = (g >> 3) * 2 + (g & 1);
        }
        sk_G = gridDim.x / p.sk_nslice;
        sk_cs0 = nct * sk_slice / p.sk_nslice;
        sk_nchunk = nct * (sk_slice + 1) / p.sk_nslice - sk_cs0;
    }
    long long su = 0, su1 = 0;
    if (SK) {
        const long long U = (long long)(p.N1pad / BM) * sk_nchunk;
        su = U * sk_j / sk_G;
        su1 = U * (sk_j + 1) / sk_G;
        if (su >= su1)
            return;
    }
    for (bool sk_first = true;; sk_first = false) {      // one trip unless SK: one trip per segment of the range
    // SK: the lane id is laundered through an empty asm in every trip -- as loop invariants, the lane-dependent addresses
    // of the prologue, the fragment reads and the 128 tile stores would be hoisted out of the segment loop, kept live
    // across it and spilled (this kernel has no register to spare)
    int lane = lane0;
    if (SK)
        asm volatile("" : "+v"(lane));
    const int h = lane >> 5, li = lane & 31;
    // SK: the segments of the range are taken LAST FIRST.  The last segment is the one that may end inside a row block,
    // i.e. the partial tile other workgroups wait for: it is published before anything else, and the one segment that may
    // have to wait for other workgroups' tiles (the first: the tail of a row block) comes at the very end.  In range
    // order every owner would sit in its wait before it even started the tile the NEXT owner waits for -- a dependency
    // chain through all row blocks (measured: 9.6 ms instead of 0.3).
    const int rb = SK ? (int)((su1 - 1) / sk_nchunk) : blockIdx.x, split = SK ? 0 : blockIdx.y;
    const int N1pad = SK ? p.N1pad : gridDim.x * BM;
    const int i = rb * BM + wave * 32 + li;     // this lane's anchor row (both lane halves)
    if (SK && !sk_first)
        dma_wait_barrier<0>();                  // every wave has left the previous segment's chunk buffers

    // ---- A panel -> registers.  Lane half h holds k in {8q + 4h .. 8q + 4h + 3}: the MFMA sums over
    // all k, so any k order works as long as both operands use the same one.
    float a[F16 ? 1 : 128];
    half8 ahi[F16 ? 16 : 1], alo[F16 ? 16 : 1];
    if (PIPE) {
        // lane half h holds k in {16 kb + 8h .. 16 kb + 8h + 7} of k-block kb (MFMA 32x32x16 operand map).  Straight
        // from memory that is 32 load instructions per wave that each touch 32 different 128-byte lines (row stride
        // 1 KiB between lanes); instead the wave's 32 rows travel as 32 coalesced 1-KiB LDS-DMA rows into chunk buffer
        // `wave` (free until the sweep starts; same slot swizzle as the chunks, so the fragment reads are
        // conflict-free) and are read back as fragments.
        const unsigned g_li = (((unsigned)li & 3u) << 2) | (((unsigned)li >> 2) & 3u);
#pragma unroll 8
        for (int r = 0; r < 32; ++r) {
            const unsigned g_r = (((unsigned)r & 3u) << 2) | (((unsigned)r >> 2) & 3u);
            dma_row(p.Ah + (size_t)(rb * BM + wave * 32 + r) * (2 * CP) + ((unsigned)lane ^ g_r) * 8,
                    lds + wave * PBUF + r * CP);
        }
        dma_wait_barrier<0>();
        const unsigned a_off = (unsigned)wave * (PBUF * 4) + (unsigned)li * 1024u + ((((unsigned)h) ^ g_li) << 4);
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) {
            const char *a16 = (const char *)lds + (a_off ^ ((unsigned)kb << 5));
            ahi[kb] = *(const half8 *)a16;
            alo[kb] = *(const half8 *)(a16 + 512);
        }
        dma_wait_barrier<0>();          // every wave holds its fragments: the buffers may take the first chunks
    } else if (F16) {
        const _Float16 *arow = p.Ah + (size_t)((DCL_SWEEP_PROBE & 8) ? 0 : i) * (2 * CP) + 8 * h;
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) {
            ahi[kb] = *(const half8 *)(arow + 16 * kb);
            alo[kb] = *(const half8 *)(arow + CP + 16 * kb);
        }
    } else {
        const float *arow = p.A + (size_t)i * CP + 4 * h;
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const f32x4 v = *(const f32x4 *)(arow + 8 * q);
            a[4 * q + 0] = v.x;
            a[4 * q + 1] = v.y;
            a[4 * q + 2] = v.z;
            a[4 * q + 3] = v.w;
        }
    }

    // ---- positive column range of this anchor, and of the wave / workgroup
    int lo = 0, hi = 0;
    if (i < p.N1) {
        const int u = i / p.V1;
        lo = p.rng_lo[u];
        hi = p.rng_hi[u];
    }
    const unsigned span = (unsigned)(hi - lo);
    const int wlo = __builtin_amdgcn_readfirstlane(wave_min_i(hi > lo ? lo : INT_MAX));
    const int whi = __builtin_amdgcn_readfirstlane(wave_max_i(hi > lo ? hi : 0));

    // ---- chunk range
    int c0, c1;
    if (MODE == MODE_POS) {
        int *red = (int *)lds;
        if (lane == 0) {
            red[wave] = wlo;
            red[4 + wave] = whi;
        }
        __syncthreads();
        const int blo = min(min(red[0], red[1]), min(red[2], red[3]));
        const int bhi = max(max(red[4], red[5]), max(red[6], red[7]));
        __syncthreads();
        if (bhi > blo) {
            c0 = blo / CJ;
            c1 = (bhi + CJ - 1) / CJ;
        } else {
            c0 = c1 = 0;
        }
    } else if (SK) {
        c0 = sk_cs0 + (int)max(0LL, su - (long long)rb * sk_nchunk);          // (global chunk indices)
        c1 = sk_cs0 + (int)(su1 - (long long)rb * sk_nchunk);
    } else {
        const int nchunk = (p.N2 + CJ - 1) / CJ;
        c0 = (int)((long long)split * nchunk / p.nsplit);
        c1 = (int)((long long)(split + 1) * nchunk / p.nsplit);
    }

    // ---- per-mode row state
    float zi = 0.f;            // MODE_Z accumulator / MODE_POS: Z_i
    float rl = 0.f, wsum = 0.f;
    float rZ = 0.f, rcW = 0.f, rcZ = 0.f;
    f32x16 dacc[MODE == MODE_BWD ? 8 : 1];
    if (MODE == MODE_POS) {
        for (int s = 0; s < p.zsplits; ++s)
            zi += p.zpart[(size_t)s * N1pad + i];
    }
    if (MODE == MODE_BWD) {
#pragma unroll
        for (int ct = 0; ct < 8; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                dacc[ct][r] = 0.f;
        if (p.use_row) {
            const f32x4 st = *(const f32x4 *)(p.rstat + (size_t)i * 4);
            rZ = st.x;
            rcW = st.y;
            rcZ = st.z;
        }
    }
    // f16x3 second product: H is scaled by one power of two G for the whole launch so that |H * G| <= 2^14.
    // The bound max|H| <= 2 * max_rows max(e^{1/tau} |cW|, |coef|) comes from dcl_infonce_prep_stats (extra row
    // of the stat arrays); within it every |H| down to 2^-28 of the bound keeps a normal f16 hi part, which
    // covers the whole range of e^{s}, s in [-1/tau, 1/tau], at tau = 0.1.
    float hG = 1.f, hInv = 1.f;
    if (F16 && MODE == MODE_BWD) {
        float mmax = 0.f;
        if (p.use_row)
            mmax = p.rstat[(size_t)N1pad * 4];
        if (USE_COL)
            mmax = fmaxf(mmax, p.cstat[(size_t)((p.N2 + BM - 1) / BM * BM) * 4]);
        hG = (mmax == 0.f) ? 1.f : exp2f(fminf(floorf(log2f(8192.0f / mmax)), 100.f));
        hInv = 1.0f / (hG * 1024.0f);
    }

    // ---- sweep
    auto stage = [&](float *dst, int j0s) {
        if (F16) {
            stage_chunk_h(dst, p.Bh, j0s, wave, lane);
        } else {
            stage_chunk(dst, p.B, j0s, wave, lane);
        }
    };
    if constexpr (PIPE) {
        // ---- f16x3 backward, two-stage software pipeline over chunks.  At one wave per SIMD nothing else can fill the
        // matrix pipe while this wave runs its exp / scale / split epilogue, so the epilogue of chunk c is interleaved,
        // instruction by instruction, with the 48 MFMAs of the S recompute of chunk c + 1 (independent work: a second
        // accumulator tile); the second product of chunk c follows.  Chunks c + 2 and c + 3 are in flight as asm
        // LDS-DMA (counted vmcnt + raw barrier), together with their 32 rows of column statistics {Z, cW, cZ, 0}: the
        // loop contains no compiler-visible global load at all.
        constexpr int NV = USE_COL ? 9 : 8;        // DMA wave-instructions per chunk and wave
        // LDS image of a chunk: 32 rows x 1 KiB, row r = [32 hi slots | 32 lo slots] of 16 B; the slot that holds
        // piece q (8 halves) of row r is q ^ g(r), g(r) = ((r & 3) << 2) | ((r >> 2) & 3) (4 bits, so hi / lo stay
        // apart).  With it BOTH operand fetches are bank-conflict free: the S recompute reads one piece of 16 different
        // rows per LDS cycle (ds_read_b128 lane groups), the second product reads 4 pieces x 4 rows (ds_read_b64_tr_b16,
        // 32 lanes per cycle) -- on padded rows the latter was 4-way conflicted and the LDS array, not the matrix
        // pipe, paced the loop (SQ_LDS_BANK_CONFLICT 56 % of SQ_LDS_IDX_ACTIVE).  The DMA fills rows lane-linearly, so
        // the swizzle is applied on the SOURCE address: lane l of row r fetches piece l ^ g(r).
        constexpr unsigned PBUFB = PBUF * 4;       // bytes per buffer
        const char *lb = (const char *)lds;
        float *cst = lds + NBUF * PBUF;
        const unsigned trq = (lane & 15) >> 2, trp = lane & 3, gq = (lane >> 4) & 1;
        const int nrows_c = (p.N2 + BM - 1) / BM * BM;             // rows of the column-statistics array
        auto gsw = [](unsigned r) { return ((r & 3u) << 2) | ((r >> 2) & 3u); };
        auto dma_piece = [&](int slot, int j0s, int row) {
            dma_row(p.Bh + (size_t)(((DCL_SWEEP_PROBE & 16) ? 0 : j0s) + row) * (2 * CP) + ((unsigned)lane ^ gsw(row)) * 8,
                    lds + slot * PBUF + row * CP);
        };
        auto dma_stats = [&](int slot, int j0s) {       // every wave copies the same piece: uniform DMA count per wave
            dma_row(p.cstat + (size_t)min(j0s + lane, nrows_c - 1) * 4, cst + slot * CSTF);
        };
        auto stage_p = [&](int slot, int j0s) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
                dma_piece(slot, j0s, wave * 8 + r);
            if (USE_COL)
                dma_stats(slot, j0s);
        };
        // S recompute operand: lane (h, li) reads piece 2 kb + h of row li -> slot (2 kb + h) ^ g(li); the buffers are
        // 1-KiB aligned, so the whole address is (row base + ((h ^ g) << 4)) ^ (kb << 5): one v_xor per k-block
        const unsigned s_off = (unsigned)li * 1024u + ((((unsigned)h) ^ gsw(li)) << 4);
        // second product operand (transposing read): lane (gq, trq, trp) supplies 8 bytes of row R + trq, piece
        // 4 ct + 2 gq + (trp >> 1) (R = 16 kb + 4 h, or + 8) -> slot ((4 ct) ^ (trq << 2)) | ((2 gq + (trp >> 1)) ^ hh),
        // hh = h (rows R ..) or h + 2 (rows R + 8 ..): address = lane constant ^ (ct << 6) + kb * 16 KiB
        const unsigned lowp = 2 * gq + (trp >> 1);
        const unsigned pa_off = (4 * h + trq) * 1024u + (trq << 6) + ((lowp ^ (unsigned)h) << 4) + 8 * (trp & 1);
        const unsigned pb_off = (4 * h + trq + 8) * 1024u + (trq << 6) + ((lowp ^ (unsigned)(h + 2)) << 4) + 8 * (trp & 1);
        // S tile of the chunk in buffer `sslot`: X[j][i] = sum_k B[j][k] A[i][k] (3 MFMAs per k-block), optionally
        // carrying the epilogue of the PREVIOUS tile (`eacc`, column statistics in `cs`) two elements per odd k-block
        // ... and, `WITH_DMA`, the nine LDS-DMA pieces of the chunk that is staged during this iteration, one per even
        // k-block (issued among the MFMAs)
        f32x16 accN;
        u32x4 hhv[2], hlv[2];
        auto s_tile = [&](int sslot, auto WITH_E, const f32x16 &eacc, const float *cs, auto WITH_DMA, int dslot,
                          int dj0) {
            constexpr bool with_e = decltype(WITH_E)::value;
            constexpr bool with_dma = decltype(WITH_DMA)::value;
            const char *sb = lb + (unsigned)sslot * PBUFB;
            auto rd = [&](int kb, half8 &hi8, half8 &lo8) {
                if ((DCL_SWEEP_PROBE & 128) && kb > 1)
                    return;
                const char *a = sb + (s_off ^ ((unsigned)kb << 5));
                hi8 = *(const half8 *)a;
                lo8 = *(const half8 *)(a + 512);
            };
            half8 qh[PF + 1], ql[PF + 1];          // ring of operand fragments, static indices after unrolling
#pragma unroll
            for (int q = 0; q < PF; ++q)
                rd(q, qh[q], ql[q]);
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * PF, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r)
                accN[r] = 0.f;
#pragma unroll
            for (int kb = 0; kb < 16; ++kb) {
                rd(kb + PF < 16 ? kb + PF : 15, qh[(kb + PF) % (PF + 1)], ql[(kb + PF) % (PF + 1)]);
                const half8 bh = qh[kb % (PF + 1)], bl = ql[kb % (PF + 1)];
                accN = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ahi[kb], accN, 0, 0, 0);
                accN = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, alo[kb], accN, 0, 0, 0);
                accN = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, ahi[kb], accN, 0, 0, 0);
                if (with_e && (kb & 1)) {
                    // elements r0, r0 + 1 of the previous tile: H = e^{s} (cW_i + cW_j) on tiles without positives,
                    // scaled by G and split into f16 (hi, lo)
                    const int r0 = kb - 1;
                    const float w0 = USE_COL ? cs[jrow(r0, h) * 4 + 1] : 0.f;
                    const float w1 = USE_COL ? cs[jrow(r0 + 1, h) * 4 + 1] : 0.f;
                    const float v0 = (DCL_SWEEP_PROBE & 32) ? eacc[r0] : __builtin_amdgcn_exp2f(eacc[r0] * p.c1) * (rcW + w0);
                    const float v1 = (DCL_SWEEP_PROBE & 32) ? eacc[r0 + 1] : __builtin_amdgcn_exp2f(eacc[r0 + 1] * p.c1) * (rcW + w1);
                    unsigned hi, lo2;
                    if (DCL_SWEEP_PROBE & 32) {
                        hi = __float_as_uint(v0);
                        lo2 = __float_as_uint(v1);
                    } else {
                        split2(v0, v1, hG, hi, lo2);
                    }
                    hhv[r0 >> 3][(r0 & 7) >> 1] = hi;
                    hlv[r0 >> 3][(r0 & 7) >> 1] = lo2;
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, USE_COL ? 4 : 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);       // first half of the epilogue VALU
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);       // the rest
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                } else {
                    if (with_dma) {
                        dma_piece(dslot, dj0, wave * 8 + (kb >> 1));
                        if (USE_COL && kb == 14)
                            dma_stats(dslot, dj0);
                    }
                    DCL_SCHED_MFMA_DS_MFMA(2, 3);
                }
            }
        };
        const int nck = c1 - c0;
        f32x16 zero16;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            zero16[r] = 0.f;
        if (nck > 0) {
#pragma unroll
            for (int q = 0; q < 3; ++q)             // past the last chunk the last one is staged again (uniform DMA count)
                stage_p(q, min(c0 + q, c1 - 1) * CJ);
            dma_wait_barrier<0>();                 // once per workgroup: everything staged so far has landed
            s_tile(0, std::false_type{}, zero16, cst, std::false_type{}, 0, 0);
        }
        for (int k = 0; k < nck; ++k) {
            const int c = c0 + k;
            // chunk c + 1 (issued two iterations ago) has landed once at most the NV DMAs of chunk c + 2 are still
            // outstanding; behind the barrier every wave has also left the second product of chunk c - 1, whose buffer
            // takes chunk c + 3 during this iteration's S recompute
            if (DCL_SWEEP_PROBE & 256)
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NV) : "memory");
            else
                dma_wait_barrier<NV>();
            const int dslot = (k + 3) & 3, dj0 = min(c + 3, c1 - 1) * CJ;
            const int nxt = (k + 1) & 3;                       // past the last chunk: stale data, result unused
            const float *cs = cst + (k & 3) * CSTF;
            const int j0 = c * CJ;
            const f32x16 acc = accN;
            const bool plain = !((j0 < whi) && (j0 + CJ > wlo)) && (j0 + CJ <= p.N2);  // wave-uniform
            if (plain) {
                s_tile(nxt, std::true_type{}, acc, cs, std::true_type{}, dslot, dj0);
            } else {
                // tile with positives / ragged edge: masked epilogue first (VALU only), then the plain S recompute
                f32x16 hv;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int jj = j0 + jrow(r, h);
                    const float e = __builtin_amdgcn_exp2f(acc[r] * p.c1);
                    const bool valid = jj < p.N2;
                    const bool inr = (unsigned)(jj - lo) < span;
                    bool pos = inr && valid;
                    if (p.intra)
                        pos = pos && (jj != i);
                    f32x4 cst4 = {0.f, 0.f, 0.f, 0.f};                      // {Z_j, cW_j, cZ_j, -}
                    if (USE_COL)
                        cst4 = *(const f32x4 *)(cs + jrow(r, h) * 4);
                    const float hn = e * (rcW + cst4.y);
                    const float hp = -(rcZ * __builtin_amdgcn_rcpf(e + rZ) + cst4.z * __builtin_amdgcn_rcpf(e + cst4.x));
                    hv[r] = pos ? hp : ((valid && !inr) ? hn : 0.f);
                }
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    unsigned hi, lo2;
                    split2(hv[r], hv[r + 1], hG, hi, lo2);
                    hhv[r >> 3][(r & 7) >> 1] = hi;
                    hlv[r >> 3][(r & 7) >> 1] = lo2;
                }
                s_tile(nxt, std::false_type{}, zero16, cs, std::true_type{}, dslot, dj0);
            }
            // second product: dA[i][c] += sum_j H[j][i] * B[j0 + j][c] (see the generic loop below for the operand
            // maps); the four transposing reads of step (kb, ct) + 1 are issued behind the first MFMA of step (kb, ct)
            const char *hb = lb + (unsigned)(k & 3) * PBUFB;
            union TR { half8 v; fp16x4 q[2]; };
            auto read_b = [&](int st, TR &bh, TR &bl) {
                if ((DCL_SWEEP_PROBE & 64) && st > 1)
                    return;
                const int kb = st >> 3, ct = st & 7;
                const char *pa = hb + kb * 16384 + (pa_off ^ ((unsigned)ct << 6));
                const char *pb = hb + kb * 16384 + (pb_off ^ ((unsigned)ct << 6));
                bh.q[0] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((LDS_FP16X4 *)pa);
                bh.q[1] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((LDS_FP16X4 *)pb);
                bl.q[0] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((LDS_FP16X4 *)(pa + 512));
                bl.q[1] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((LDS_FP16X4 *)(pb + 512));
            };
            TR th[PF + 1], tl[PF + 1];
#pragma unroll
            for (int q = 0; q < PF; ++q)
                read_b(q, th[q], tl[q]);
            __builtin_amdgcn_sched_group_barrier(0x100, 4 * PF, 0);
#pragma unroll
            for (int st = 0; st < 16; ++st) {
                const int kb = st >> 3, ct = st & 7;
                const half8 hh = __builtin_bit_cast(half8, hhv[kb]), hl = __builtin_bit_cast(half8, hlv[kb]);
                read_b(st + PF < 16 ? st + PF : 15, th[(st + PF) % (PF + 1)], tl[(st + PF) % (PF + 1)]);
                const half8 bhv = th[st % (PF + 1)].v, blv = tl[st % (PF + 1)].v;
                dacc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hh, bhv, dacc[ct], 0, 0, 0);
                dacc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hh, blv, dacc[ct], 0, 0, 0);
                dacc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hl, bhv, dacc[ct], 0, 0, 0);
                DCL_SCHED_MFMA_DS_MFMA(4, 3);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // no LDS-DMA may outlive the loop
    } else {
    if (c0 < c1)
        stage(lds, c0 * CJ);
    for (int c = c0; c < c1; ++c) {
        dma_wait_barrier<0>();  // chunk c has landed (asm LDS-DMA: waited for by hand); everyone left chunk c-1
        const float *cur = lds + ((c - c0) & 1) * BUF;
        const float *buf = cur;
        if (c + 1 < c1 && !(DCL_SWEEP_PROBE & 1))
            stage(lds + ((c + 1 - c0) & 1) * BUF, (c + 1) * CJ);
        const int j0 = c * CJ;

        // column statistics for the backward epilogue: issue early, consume after the MFMA chain
        float ccw[16];
        if (MODE == MODE_BWD) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                ccw[r] = USE_COL ? p.cstat[(size_t)(j0 + jrow(r, h)) * 4 + 1] : 0.f;
        }

        // X[j][i] = sum_k B[j0 + j][k] * A[i][k]
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            acc[r] = 0.f;
        if (F16) {
            const _Float16 *bt = (const _Float16 *)cur + li * ROWH + 8 * h;
            half8 bh = *(const half8 *)bt;
            half8 bl = *(const half8 *)(bt + CP);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      // prologue reads of k-block 0
#pragma unroll
            for (int kb = 0; kb < 16; ++kb) {
                const int kn = (DCL_SWEEP_PROBE & 4) ? 0 : (kb < 15 ? kb + 1 : 15);
                const half8 nh = (DCL_SWEEP_PROBE & 4) ? bh : *(const half8 *)(bt + 16 * kn);
                const half8 nl = (DCL_SWEEP_PROBE & 4) ? bl : *(const half8 *)(bt + CP + 16 * kn);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ahi[kb], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, alo[kb], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, ahi[kb], acc, 0, 0, 0);
                DCL_SCHED_MFMA_DS_MFMA(2, 3);
                bh = nh;
                bl = nl;
            }
        } else {
            const float *bt = buf + li * ROWF + 4 * h;
            f32x4 b = *(const f32x4 *)bt;
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // prologue read of step 0
#pragma unroll
            for (int q = 0; q < 32; ++q) {
                const f32x4 bn = *(const f32x4 *)(bt + 8 * (q < 31 ? q + 1 : 31));
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, a[4 * q + 0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, a[4 * q + 1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, a[4 * q + 2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, a[4 * q + 3], acc, 0, 0, 0);
                DCL_SCHED_MFMA_DS_MFMA(1, 4);
                b = bn;
            }
        }

        const bool plain = !((j0 < whi) && (j0 + CJ > wlo)) && (j0 + CJ <= p.N2);  // wave-uniform
        if (MODE == MODE_Z) {
            if (plain) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    zi += (DCL_SWEEP_PROBE & 2) ? acc[r] : __builtin_amdgcn_exp2f(acc[r] * p.c1);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int jj = j0 + jrow(r, h);
                    const float e = __builtin_amdgcn_exp2f(acc[r] * p.c1);
                    const bool inr = (unsigned)(jj - lo) < span;
                    zi += (jj < p.N2 && !inr) ? e : 0.f;
                    if (p.spos && inr && jj < p.N2)         // (rows >= N1 have an empty range)
                        p.spos[(size_t)i * p.spos_ld + (jj - lo)] = acc[r];
                }
            }
        } else if (MODE == MODE_POS) {
            if (!plain) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int jj = j0 + jrow(r, h);
                    const float e = __builtin_amdgcn_exp2f(acc[r] * p.c1);
                    bool pos = ((unsigned)(jj - lo) < span) && (jj < p.N2);
                    if (p.intra)
                        pos = pos && (jj != i);
                    const float t = e + zi;
                    rl += pos ? (acc[r] * p.inv_tau - logf(t)) : 0.f;
                    wsum += pos ? __builtin_amdgcn_rcpf(t) : 0.f;
                }
            }
        } else {
            // Tiles without positives (the common case): H = e * (cW_i + cW_j), no masks.
            // (Folding this epilogue into the second product's loop was tried: the VALU ops then sit right in
            // front of the MFMA that consumes them and cost 3-5 % -- kept as one block per tile.)
            if (plain) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    acc[r] = __builtin_amdgcn_exp2f(acc[r] * p.c1) * (rcW + ccw[r]);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int jj = j0 + jrow(r, h);
                    const float e = __builtin_amdgcn_exp2f(acc[r] * p.c1);
                    const bool valid = jj < p.N2;
                    const bool inr = (unsigned)(jj - lo) < span;
                    bool pos = inr && valid;
                    if (p.intra)
                        pos = pos && (jj != i);
                    float cZ = 0.f, ccZ = 0.f;
                    if (USE_COL) {
                        cZ = p.cstat[(size_t)jj * 4 + 0];
                        ccZ = p.cstat[(size_t)jj * 4 + 2];
                    }
                    const float hn = e * (rcW + ccw[r]);
                    const float hp = -(rcZ * __builtin_amdgcn_rcpf(e + rZ) +
                                       ccZ * __builtin_amdgcn_rcpf(e + cZ));
                    acc[r] = pos ? hp : ((valid && !inr) ? hn : 0.f);
                }
            }
            if (F16) {
                // dA[i][c] += sum_j H[j][i] * B[j0 + j][c] as f16x3: the accumulator registers 8kb .. 8kb+7 of
                // this lane ARE, in order, the 8 k-slots of the A operand of k-block kb (k-slot (h, e) <-> chunk
                // row 16 kb + 4 h + (e & 3) + 8 (e >> 2)); the matching B operand -- 4 consecutive chunk rows at
                // one channel per lane -- comes straight out of the row-major half rows with the transposing LDS
                // read ds_read_b64_tr_b16 (lane 4q+p of a 16-lane group supplies &tile[row q][col 4p] and
                // receives column (lane % 16), rows 0..3; probed in tools/probes/tr_probe.hip).
                const _Float16 *hb = (const _Float16 *)cur;
                const int trq = (lane & 15) >> 2, trp = lane & 3;
                const int colb = 16 * ((lane >> 4) & 1) + 4 * trp;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    half8 hh, hl;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float v = acc[8 * kb + e] * hG;
                        const _Float16 t = (_Float16)v;
                        hh[e] = t;
                        hl[e] = (_Float16)(v - (float)t);
                    }
                    const _Float16 *pa = hb + (16 * kb + 4 * h + trq) * ROWH + colb;
                    const _Float16 *pb = pa + 8 * ROWH;
#pragma unroll
                    for (int ct = 0; ct < 8; ++ct) {
                        union { half8 v; fp16x4 q[2]; } bh, bl;
                        bh.q[0] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((LDS_FP16X4 *)(pa + 32 * ct));
                        bh.q[1] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((LDS_FP16X4 *)(pb + 32 * ct));
                        bl.q[0] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((LDS_FP16X4 *)(pa + CP + 32 * ct));
                        bl.q[1] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((LDS_FP16X4 *)(pb + CP + 32 * ct));
                        dacc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hh, bh.v, dacc[ct], 0, 0, 0);
                        dacc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hh, bl.v, dacc[ct], 0, 0, 0);
                        dacc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hl, bh.v, dacc[ct], 0, 0, 0);
                    }
                }
            } else {
            // dA[i][c] += sum_j H[j][i] * B[j0 + j][c]: acc register r IS the A-operand of k-step r.
            // Column permutation: MFMA (g, e) writes column li <-> channel c = 128 g + 4 li + e, so one
            // ds_read_b128 per (r, g) feeds four MFMAs and the final store is 16 B per lane.
            {
                const float *b0 = buf + 4 * li;
                f32x4 u0 = *(const f32x4 *)(b0 + jrow(0, h) * ROWF);
                f32x4 u1 = *(const f32x4 *)(b0 + jrow(0, h) * ROWF + 128);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // prologue reads of step 0
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rn = r < 15 ? r + 1 : 15;
                    const f32x4 n0 = *(const f32x4 *)(b0 + jrow(rn, h) * ROWF);
                    const f32x4 n1 = *(const f32x4 *)(b0 + jrow(rn, h) * ROWF + 128);
                    dacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(acc[r], u0.x, dacc[0], 0, 0, 0);
                    dacc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(acc[r], u0.y, dacc[1], 0, 0, 0);
                    dacc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(acc[r], u0.z, dacc[2], 0, 0, 0);
                    dacc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(acc[r], u0.w, dacc[3], 0, 0, 0);
                    dacc[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(acc[r], u1.x, dacc[4], 0, 0, 0);
                    dacc[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(acc[r], u1.y, dacc[5], 0, 0, 0);
                    dacc[6] = __builtin_amdgcn_mfma_f32_32x32x2f32(acc[r], u1.z, dacc[6], 0, 0, 0);
                    dacc[7] = __builtin_amdgcn_mfma_f32_32x32x2f32(acc[r], u1.w, dacc[7], 0, 0, 0);
                    DCL_SCHED_MFMA_DS_MFMA(2, 8);
                    u0 = n0;
                    u1 = n1;
                }
            }
            }
        }
    }

    }   // !PIPE

    // ---- outputs
    if (MODE == MODE_Z) {
        zi += __shfl_xor(zi, 32, 64);
        if (h == 0)
            p.zpart[(size_t)split * N1pad + i] = zi;
    } else if (MODE == MODE_POS) {
        rl += __shfl_xor(rl, 32, 64);
        wsum += __shfl_xor(wsum, 32, 64);
        if (h == 0) {
            p.Z[i] = zi;
            if (p.accumulate) {
                rl += p.rowloss[i];
                wsum += p.W[i];
            }
            p.rowloss[i] = rl;
            p.W[i] = wsum;
        }
    } else {
        if (SK) {
            // natural channel order (column li of MFMA ct <-> channel 32 ct + li), unscaled by 1 / (G * 2^10)
            const int g = blockIdx.x;
            // global workgroup id of local index jj of this slice (inverse of the decode at the top)
            auto sk_gid = [&](int jj) {
                return p.sk_nslice == 8 ? jj * 8 + sk_slice : (p.sk_nslice == 4 ? (jj >> 1) * 8 + sk_slice * 2 + (jj & 1) : jj);
            };
            // element (r, ct) of this lane sits at toff + KOFF(r, ct) of a 128 x 256 tile, KOFF a compile-time constant.
            // toff is laundered through an empty asm: as a loop invariant of the segment loop the compiler would hoist all
            // 128 (64-bit) store addresses out of it and spill them (1 KiB of scratch per lane)
            unsigned toff = (unsigned)((wave * 32 + 4 * h) * CP + li);
            asm volatile("" : "+v"(toff));
#define DCL_KOFF(r, ct) ((unsigned)(jrow((r), 0) * CP + 32 * (ct)))
            if (c1 == sk_cs0 + sk_nchunk) {
                // owner of row block rb (of this slice): the earlier chunks (if any) were covered by the slice's workgroups
                // gf .. sk_j - 1 (local indices), each of which leaves exactly one partial tile (its last segment) in sk_ws
                float *out = p.dpart + ((size_t)sk_slice * p.N1pad + (size_t)rb * BM) * CP;
                if (c0 > sk_cs0) {
                    const long long U = (long long)(p.N1pad / BM) * sk_nchunk, ub = (long long)rb * sk_nchunk;
                    int gf = sk_j - 1;                           // first contributor: the workgroup holding unit (rb, 0)
                    while (gf > 0 && U * gf / sk_G > ub)
                        --gf;
#pragma unroll
                    for (int ct = 0; ct < 8; ++ct)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            dacc[ct][r] *= hInv;
                    // Everything that crosses workgroups -- the flags and the partial tiles -- moves as relaxed AGENT-scope
                    // atomic accesses (sc1: served by memory, not by this XCD's L2) and is ordered by hand (the writer
                    // waits for its stores, then a barrier, then the flag).  Acquire / release fences at agent scope
                    // would do it too, but on this multi-XCD part they invalidate / write back the WHOLE L2 of the XCD
                    // each time: with the owners spinning on acquire loads the launch took 10 ms instead of 0.3.
                    // The wait is BOUNDED: deadlock-freedom rests on every contributor (a lower workgroup id) being or becoming
                    // resident, which other streams' persistent kernels, CU masks or several ranks on one device can break.
                    // After sk_timeout ticks without the flag the owner counts an error in *sk_err (the caller reads that word:
                    // the launch's gradient is then invalid and it falls back to the column-split form) and goes on.
                    if (tid == 0 && p.sk_probe == 0) {
                        const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
                        bool late = false;
                        for (int gp = gf; gp < sk_j && !late; ++gp)
                            while (__hip_atomic_load(p.sk_flags + sk_gid(gp), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != p.sk_seq) {
                                __builtin_amdgcn_s_sleep(8);
                                if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > p.sk_timeout) {
                                    late = true;
                                    break;
                                }
                            }
                        if (late)
                            __hip_atomic_fetch_add(p.sk_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    __syncthreads();
                    for (int gp = gf; gp < sk_j; ++gp) {
                        const float *pt = p.sk_ws + (size_t)sk_gid(gp) * BM * CP;
#pragma unroll
                        for (int ct = 0; ct < 8; ++ct) {
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                dacc[ct][r] += __hip_atomic_load(pt + toff + DCL_KOFF(r, ct), __ATOMIC_RELAXED,
                                                                 __HIP_MEMORY_SCOPE_AGENT);
                            __builtin_amdgcn_sched_barrier(0);       // 16 loads in flight, not 128 (registers)
                        }
                    }
                    // (no flag reset: the next launch on this pair waits for ITS sequence number)
#pragma unroll
                    for (int ct = 0; ct < 8; ++ct)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            out[toff + DCL_KOFF(r, ct)] = dacc[ct][r];
                } else {
#pragma unroll
                    for (int ct = 0; ct < 8; ++ct)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            out[toff + DCL_KOFF(r, ct)] = dacc[ct][r] * hInv;
                }
            } else {
                float *out = p.sk_ws + (size_t)g * BM * CP;
#pragma unroll
                for (int ct = 0; ct < 8; ++ct)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        __hip_atomic_store(out + toff + DCL_KOFF(r, ct), dacc[ct][r] * hInv, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's tile rows have reached memory
                __syncthreads();
                if (tid == 0 && p.sk_probe == 0)
                    __hip_atomic_store(p.sk_flags + g, p.sk_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#undef DCL_KOFF
        } else if (F16) {
            // natural channel order (column li of MFMA ct <-> channel 32 ct + li), unscaled by 1 / (G * 2^10)
            float *out = p.dpart + ((size_t)split * N1pad + (size_t)rb * BM + wave * 32) * CP + li;
#pragma unroll
            for (int ct = 0; ct < 8; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    out[(size_t)jrow(r, h) * CP + 32 * ct] = dacc[ct][r] * hInv;
        } else {
            float *out = p.dpart + ((size_t)split * N1pad + (size_t)rb * BM + wave * 32) * CP + 4 * li;
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    f32x4 v;
                    v.x = dacc[4 * g + 0][r];
                    v.y = dacc[4 * g + 1][r];
                    v.z = dacc[4 * g + 2][r];
                    v.w = dacc[4 * g + 3][r];
                    *(f32x4 *)(out + (size_t)jrow(r, h) * CP + 128 * g) = v;
                }
        }
    }
    if (!SK)
        break;
    su1 -= c1 - c0;
    if (su >= su1)
        break;
    }   // segments
}

// loss = -(1/N1) sum_i rowloss_i / P_i     (DenseContrastiveLossV2.py:188-189; ms:148-156)
__device__ __forceinline__ float positives_of(const int32_t *pcount, const int32_t *rng_lo,
                                              const int32_t *rng_hi, int u, int intra)
{
    // explicit per-slot count (multi-segment banks) or derived from the single positive range
    const int P = pcount ? pcount[u] : (rng_hi[u] - rng_lo[u] - (intra ? 1 : 0));
    return intra ? (float)P : (float)max(P, 1);      // cross-scale: where(P > 0, P, 1)  (ms:148-152)
}

__global__ __launch_bounds__(1024) void k_loss_reduce(const float *__restrict__ rowloss,
                                                     const int32_t *__restrict__ rng_lo,
                                                     const int32_t *__restrict__ rng_hi,
                                                     const int32_t *__restrict__ pcount, int N1,
                                                     int V1, int intra, float *__restrict__ loss)
{
    __shared__ float part[16];
    float acc = 0.f;
    for (int i = threadIdx.x; i < N1; i += 1024) {
        const float Pn = positives_of(pcount, rng_lo, rng_hi, i / V1, intra);
        acc += rowloss[i] / Pn;       // intra with P == 0: 0/0 = NaN, as in the reference
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0)
        part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int w = 0; w < 16; ++w)
            s += part[w];
        loss[0] = -s / (float)N1;
    }
}

// The positives' share of the forward, from the similarities MODE_Z kept (SweepArgs::spos): row i's
//   rowloss_i = sum_p (s_ip / tau - log(e^{s_ip / tau} + Z_i)),   W_i = sum_p 1 / (e^{s_ip / tau} + Z_i),   Z_i = sum of the zpart slabs
// -- exactly what MODE_POS computes per element (same exp2 / log / rcp), one wave per row, lanes along the positive range.
__global__ __launch_bounds__(256) void k_pos_finish(const float *__restrict__ spos, int ld, int N1, int N1pad, int V1,
                                                   const int32_t *__restrict__ rng_lo, const int32_t *__restrict__ rng_hi,
                                                   float inv_tau, float c1, int intra, const float *__restrict__ zpart,
                                                   int zsplits, float *__restrict__ Z, float *__restrict__ rowloss,
                                                   float *__restrict__ W)
{
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= N1pad)
        return;
    float zi = 0.f;
    for (int s = 0; s < zsplits; ++s)
        zi += zpart[(size_t)s * N1pad + i];
    int lo = 0, hi = 0;
    if (i < N1) {
        lo = rng_lo[i / V1];
        hi = rng_hi[i / V1];
    }
    float rl = 0.f, ws = 0.f;
    const float *row = spos + (size_t)i * ld;
    for (int q = lane; q < hi - lo; q += 64) {
        const float sv = row[q];
        const float t = __builtin_amdgcn_exp2f(sv * c1) + zi;
        const bool pos = !(intra && lo + q == i);
        rl += pos ? (sv * inv_tau - logf(t)) : 0.f;
        ws += pos ? __builtin_amdgcn_rcpf(t) : 0.f;
    }
    rl = wave_sum(rl);
    ws = wave_sum(ws);
    if (lane == 0) {
        Z[i] = zi;
        rowloss[i] = rl;
        W[i] = ws;
    }
}

__global__ __launch_bounds__(256) void k_prep_stats(const float *__restrict__ Z,
                                                   const float *__restrict__ W,
                                                   const int32_t *__restrict__ rng_lo,
                                                   const int32_t *__restrict__ rng_hi,
                                                   const int32_t *__restrict__ pcount, int N1,
                                                   int N1pad, int V1, int intra, float wscale,
                                                   float inv_tau, const float *__restrict__ grad_out,
                                                   float *__restrict__ stat)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N1pad)
        return;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    if (i < N1) {
        const float gs = wscale * inv_tau * (grad_out ? grad_out[0] : 1.0f);
        const float Pn = positives_of(pcount, rng_lo, rng_hi, i / V1, intra);
        const float coef = gs / ((float)N1 * Pn);
        o.x = Z[i];
        o.y = coef * W[i];
        o.z = coef * Z[i];
        // bound of |dL/ds| terms this row can produce: negatives e^{s} |cW| <= e^{1/tau} |cW|, positives <= |coef|
        const float m = fmaxf(expf(inv_tau) * fabsf(o.y), fabsf(coef));
        atomicMax((unsigned int *)(stat + (size_t)N1pad * 4), __float_as_uint(m));    // m >= 0: uint order
    }
    *(f32x4 *)(stat + (size_t)i * 4) = o;
}

int check_common(const float *A, int N1, int V1, const float *B, int N2, const int32_t *lo,
                 const int32_t *hi, int nsplit)
{
    if (!A || !B || !lo || !hi) {
        dcl_set_error("null pointer");
        return DCL_EINVAL;
    }
    if (N1 <= 0 || N2 <= 0 || V1 <= 0 || nsplit <= 0 || nsplit > 64) {
        dcl_set_error("bad sizes (N1=%d N2=%d V1=%d nsplit=%d)", N1, N2, V1, nsplit);
        return DCL_EINVAL;
    }
    return 0;
}

}  // namespace

static SweepArgs fwd_args(const float *A, int N1, int V1, const float *B, int N2,
                          const int32_t *rng_lo, const int32_t *rng_hi, float inv_tau, int intra,
                          const void *Ah, const void *Bh)
{
    SweepArgs p = {};
    p.A = A; p.B = B; p.N1 = N1; p.N2 = N2; p.V1 = V1;
    p.Ah = (const _Float16 *)Ah; p.Bh = (const _Float16 *)Bh;
    p.rng_lo = rng_lo; p.rng_hi = rng_hi;
    // f16x3: the accumulators hold 2^20 * <a, b>; fold the factor into both logit scales
    const float k = (Ah && Bh) ? F16_SCALE_SQ_INV : 1.0f;
    p.inv_tau = inv_tau * k; p.c1 = inv_tau * 1.4426950408889634f * k;
    p.intra = intra;
    return p;
}

extern "C" int dcl_infonce_zsweep(const float *A, int N1, int V1, const float *B, int N2,
                                  const int32_t *rng_lo, const int32_t *rng_hi, float inv_tau,
                                  int nsplit, float *zpart, const void *Ah, const void *Bh, void *stream)
{
    int rc = check_common(A, N1, V1, B, N2, rng_lo, rng_hi, nsplit);
    if (rc)
        return rc;
    DCL_CHECK_ARG(zpart, "null output pointer");
    DCL_CHECK_ARG((Ah == nullptr) == (Bh == nullptr), "Ah and Bh must be given together");
    SweepArgs p = fwd_args(A, N1, V1, B, N2, rng_lo, rng_hi, inv_tau, 0, Ah, Bh);
    p.nsplit = nsplit; p.zpart = zpart;
    const int RB = dcl_round_up(N1, BM) / BM;
    if (Ah)
        hipLaunchKernelGGL((k_sweep<MODE_Z, false, true>), dim3(RB, nsplit), dim3(256), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL((k_sweep<MODE_Z, false, false>), dim3(RB, nsplit), dim3(256), 0, (hipStream_t)stream, p);
    DCL_LAUNCH_CHECK();
    return 0;
}

extern "C" int dcl_infonce_zsweep_keep(const float *A, int N1, int V1, const float *B, int N2,
                                       const int32_t *rng_lo, const int32_t *rng_hi, float inv_tau,
                                       int nsplit, float *zpart, const void *Ah, const void *Bh, float *spos, int spos_ld,
                                       void *stream)
{
    int rc = check_common(A, N1, V1, B, N2, rng_lo, rng_hi, nsplit);
    if (rc)
        return rc;
    DCL_CHECK_ARG(zpart && spos && spos_ld > 0, "null output pointer");
    DCL_CHECK_ARG((Ah == nullptr) == (Bh == nullptr), "Ah and Bh must be given together");
    SweepArgs p = fwd_args(A, N1, V1, B, N2, rng_lo, rng_hi, inv_tau, 0, Ah, Bh);
    p.nsplit = nsplit; p.zpart = zpart;
    p.spos = spos; p.spos_ld = spos_ld;
    const int RB = dcl_round_up(N1, BM) / BM;
    if (Ah)
        hipLaunchKernelGGL((k_sweep<MODE_Z, false, true>), dim3(RB, nsplit), dim3(256), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL((k_sweep<MODE_Z, false, false>), dim3(RB, nsplit), dim3(256), 0, (hipStream_t)stream, p);
    DCL_LAUNCH_CHECK();
    return 0;
}

extern "C" int dcl_infonce_pos_finish(const float *spos, int spos_ld, int N1, int V1, const int32_t *rng_lo,
                                      const int32_t *rng_hi, float inv_tau, int intra, int f16x3, const float *zpart,
                                      int zsplits, float *Z, float *rowloss, float *W, void *stream)
{
    DCL_CHECK_ARG(spos && rng_lo && rng_hi && zpart && Z && rowloss && W, "null pointer");
    DCL_CHECK_ARG(N1 > 0 && V1 > 0 && spos_ld > 0 && zsplits > 0, "bad sizes");
    const float k = f16x3 ? F16_SCALE_SQ_INV : 1.0f;            // (as fwd_args: the kept values are 2^20 <a, b> in f16x3 mode)
    const int N1pad = dcl_round_up(N1, BM);
    hipLaunchKernelGGL(k_pos_finish, dim3((N1pad + 3) / 4), dim3(256), 0, (hipStream_t)stream, spos, spos_ld, N1, N1pad, V1,
                       rng_lo, rng_hi, inv_tau * k, inv_tau * 1.4426950408889634f * k, intra, zpart, zsplits, Z, rowloss, W);
    DCL_LAUNCH_CHECK();
    return 0;
}

extern "C" int dcl_infonce_possweep(const float *A, int N1, int V1, const float *B, int N2,
                                    const int32_t *rng_lo, const int32_t *rng_hi, float inv_tau,
                                    int intra, const float *zpart, int zsplits, int accumulate,
                                    float *Z, float *rowloss, float *W, const void *Ah, const void *Bh,
                                    void *stream)
{
    int rc = check_common(A, N1, V1, B, N2, rng_lo, rng_hi, 1);
    if (rc)
        return rc;
    DCL_CHECK_ARG(zpart && Z && rowloss && W && zsplits > 0, "bad arguments");
    DCL_CHECK_ARG((Ah == nullptr) == (Bh == nullptr), "Ah and Bh must be given together");
    SweepArgs p = fwd_args(A, N1, V1, B, N2, rng_lo, rng_hi, inv_tau, intra, Ah, Bh);
    p.nsplit = 1; p.zpart = const_cast<float *>(zpart); p.zsplits = zsplits; p.accumulate = accumulate;
    p.Z = Z; p.rowloss = rowloss; p.W = W;
    const int RB = dcl_round_up(N1, BM) / BM;
    if (Ah)
        hipLaunchKernelGGL((k_sweep<MODE_POS, false, true>), dim3(RB, 1), dim3(256), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL((k_sweep<MODE_POS, false, false>), dim3(RB, 1), dim3(256), 0, (hipStream_t)stream, p);
    DCL_LAUNCH_CHECK();
    return 0;
}

extern "C" int dcl_infonce_loss(const float *rowloss, const int32_t *rng_lo, const int32_t *rng_hi,
                                const int32_t *pcount, int N1, int V1, int intra, float *loss,
                                void *stream)
{
    DCL_CHECK_ARG(rowloss && loss && (pcount || (rng_lo && rng_hi)) && N1 > 0 && V1 > 0, "bad arguments");
    hipLaunchKernelGGL(k_loss_reduce, dim3(1), dim3(1024), 0, (hipStream_t)stream, rowloss, rng_lo,
                       rng_hi, pcount, N1, V1, intra, loss);
    DCL_LAUNCH_CHECK();
    return 0;
}

extern "C" int dcl_infonce_fwd(const float *A, int N1, int V1, const float *B, int N2,
                               const int32_t *rng_lo, const int32_t *rng_hi, float inv_tau,
                               int intra, int nsplit, float *zpart, float *Z, float *rowloss,
                               float *W, float *loss, void *stream)
{
    int rc = dcl_infonce_zsweep(A, N1, V1, B, N2, rng_lo, rng_hi, inv_tau, nsplit, zpart, nullptr,
                                nullptr, stream);
    if (rc)
        return rc;
    rc = dcl_infonce_possweep(A, N1, V1, B, N2, rng_lo, rng_hi, inv_tau, intra, zpart, nsplit, 0, Z,
                              rowloss, W, nullptr, nullptr, stream);
    if (rc)
        return rc;
    DCL_CHECK_ARG(loss, "null output pointer");
    return dcl_infonce_loss(rowloss, rng_lo, rng_hi, nullptr, N1, V1, intra, loss, stream);
}

extern "C" int dcl_infonce_prep_stats(const float *Z, const float *W, const int32_t *rng_lo,
                                      const int32_t *rng_hi, const int32_t *pcount, int N1, int V1,
                                      int intra, float wscale, float inv_tau,
                                      const float *grad_out, float *stat, void *stream)
{
    DCL_CHECK_ARG(Z && W && stat && (pcount || (rng_lo && rng_hi)), "null pointer");
    DCL_CHECK_ARG(N1 > 0 && V1 > 0, "bad sizes");
    const int N1pad = dcl_round_up(N1, BM);
    hipError_t e = hipMemsetAsync(stat + (size_t)N1pad * 4, 0, 4 * sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) {
        dcl_set_error("dcl_infonce_prep_stats: memset failed: %s", hipGetErrorString(e));
        return (int)e;
    }
    hipLaunchKernelGGL(k_prep_stats, dim3(N1pad / 256 + 1), dim3(256), 0, (hipStream_t)stream, Z, W,
                       rng_lo, rng_hi, pcount, N1, N1pad, V1, intra, wscale, inv_tau, grad_out, stat);
    DCL_LAUNCH_CHECK();
    return 0;
}

extern "C" int dcl_infonce_bwd(const float *A, int N1, int V1, const float *B, int N2,
                               const int32_t *rng_lo, const int32_t *rng_hi, float inv_tau,
                               int intra, int use_row, int use_col, const float *rstat,
                               const float *cstat, int nsplit, float *dpart, const void *Ah,
                               const void *Bh, void *stream)
{
    int rc = check_common(A, N1, V1, B, N2, rng_lo, rng_hi, nsplit);
    if (rc)
        return rc;
    DCL_CHECK_ARG(dpart, "null output pointer");
    DCL_CHECK_ARG(!use_row || rstat, "use_row needs rstat");
    DCL_CHECK_ARG(!use_col || cstat, "use_col needs cstat");
    DCL_CHECK_ARG((Ah == nullptr) == (Bh == nullptr), "Ah and Bh must be given together");
    SweepArgs p = fwd_args(A, N1, V1, B, N2, rng_lo, rng_hi, inv_tau, intra, Ah, Bh);
    p.nsplit = nsplit;
    p.rstat = rstat; p.cstat = cstat; p.use_row = use_row; p.use_col = use_col;
    p.dpart = dpart;
    const int RB = dcl_round_up(N1, BM) / BM;
    const dim3 grid(RB, nsplit);
    hipStream_t st = (hipStream_t)stream;
    if (Ah) {
        if (use_col)
            hipLaunchKernelGGL((k_sweep<MODE_BWD, true, true>), grid, dim3(256), 0, st, p);
        else
            hipLaunchKernelGGL((k_sweep<MODE_BWD, false, true>), grid, dim3(256), 0, st, p);
    } else {
        if (use_col)
            hipLaunchKernelGGL((k_sweep<MODE_BWD, true, false>), grid, dim3(256), 0, st, p);
        else
            hipLaunchKernelGGL((k_sweep<MODE_BWD, false, false>), grid, dim3(256), 0, st, p);
    }
    DCL_LAUNCH_CHECK();
    return 0;
}

// ---- stream-K backward (f16x3 operands) ----------------------------------------------------------------------------
static int g_streamk = 1;
static int g_sk_prefetch = 2;      // operand read-ahead of the pipelined backward (tuning hook: 1, 2 or 3 steps)

extern "C" int dcl_infonce_set_streamk(int on)
{
    // 0 off, 1 on, 2 = timing probe without the inter-workgroup hand-over (results invalid); + 16 * read-ahead steps
    // (1..3, 0 = keep): tuning hook for the operand prefetch depth of the pipelined kernel
    g_streamk = on & 15;
    if ((on >> 4) >= 1 && (on >> 4) <= 3)
        g_sk_prefetch = on >> 4;
    return 0;
}

// number of CUs of the current device (queried once per device): the persistent grid is one workgroup per CU
static int sk_cu_count()
{
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64)
        return 256;
    if (cus[dev] == 0) {
        hipDeviceProp_t prop;
        cus[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                       ? prop.multiProcessorCount : 256;
    }
    return cus[dev];
}

static int g_sk_slices = 4;        // column slices of the stream-K backward (1 | 4 | 8), see k_sweep; used when the grid is a
                                   // whole number of XCD rounds and every slice keeps >= 16 chunks

extern "C" int dcl_infonce_set_streamk_slices(int n)
{
    if (n != 1 && n != 4 && n != 8)
        return DCL_EINVAL;
    g_sk_slices = n;
    return 0;
}

static int sk_slices_for(int G, int N2)
{
    const int nchunk = (N2 + CJ - 1) / CJ;
    return (g_sk_slices > 1 && G % (8 * g_sk_slices) == 0 && nchunk >= 16 * g_sk_slices) ? g_sk_slices : 1;
}

// slabs dcl_infonce_bwd_streamk writes for these sizes (dout f32 [slabs][N1pad][DCL_CP], to be summed in order): 0 = stream-K off
extern "C" int dcl_infonce_bwd_streamk_slabs(int N1, int N2)
{
    const int G = dcl_infonce_bwd_streamk_workgroups(N1, N2);
    return G > 0 ? sk_slices_for(G, N2) : 0;
}

static long long g_sk_timeout_ticks = 200000000LL;      // 2 s of the 100-MHz s_memrealtime clock per hand-over

extern "C" int dcl_infonce_set_streamk_timeout_ms(int ms)
{
    g_sk_timeout_ticks = (long long)(ms > 0 ? ms : 2000) * 100000LL;
    return 0;
}

// 0 = the column-split form should be used (switch off, or fewer units than workgroups would make empty ranges pointless)
extern "C" int dcl_infonce_bwd_streamk_workgroups(int N1, int N2)
{
    if (!g_streamk || N1 <= 0 || N2 <= 0)
        return 0;
    const long long units = (long long)(dcl_round_up(N1, BM) / BM) * ((N2 + CJ - 1) / CJ);
    const int cus = sk_cu_count();                    // one persistent workgroup per CU of THIS device
    return units >= cus ? cus : (int)units;
}

// launch numbers per (flags, workspace) pair: a flag is valid for the launch whose number it holds, so nothing has to be
// zero on entry and a launch that was aborted (or timed out) leaves nothing behind that a later launch could mistake
static int sk_next_seq(const void *flags)
{
    static std::mutex mu;
    static std::unordered_map<const void *, int> seq;
    std::lock_guard<std::mutex> lock(mu);
    int &s = seq[flags];
    s = s >= 0x7ffffff0 ? 1 : s + 1;
    return s;
}

extern "C" int dcl_infonce_bwd_streamk(const float *A, int N1, int V1, const float *B, int N2,
                                       const int32_t *rng_lo, const int32_t *rng_hi, float inv_tau, int intra,
                                       int use_row, int use_col, const float *rstat, const float *cstat, float *dout,
                                       float *ws, int32_t *flags, const void *Ah, const void *Bh, void *stream)
{
    int rc = check_common(A, N1, V1, B, N2, rng_lo, rng_hi, 1);
    if (rc)
        return rc;
    DCL_CHECK_ARG(dout && ws && flags, "null output / workspace pointer");
    DCL_CHECK_ARG(Ah && Bh, "the stream-K backward takes the (hi | lo) half rows of both banks");
    DCL_CHECK_ARG(!use_row || rstat, "use_row needs rstat");
    DCL_CHECK_ARG(!use_col || cstat, "use_col needs cstat");
    const int G = dcl_infonce_bwd_streamk_workgroups(N1, N2);
    DCL_CHECK_ARG(G > 0, "stream-K is switched off (dcl_infonce_set_streamk)");
    SweepArgs p = fwd_args(A, N1, V1, B, N2, rng_lo, rng_hi, inv_tau, intra, Ah, Bh);
    p.nsplit = 1;
    p.rstat = rstat; p.cstat = cstat; p.use_row = use_row; p.use_col = use_col;
    p.dpart = dout;
    p.N1pad = dcl_round_up(N1, BM);
    p.sk_ws = ws;
    p.sk_err = flags;              // word 0: a FIXED place, so that launches of different grid sizes that share the buffer (a short
    p.sk_flags = flags + 1;        // cross-scale term has fewer units than CUs) all count into the word the caller reads
    p.sk_nslice = sk_slices_for(G, N2);
    p.sk_seq = sk_next_seq(flags);
    p.sk_timeout = g_sk_timeout_ticks;
    p.sk_probe = g_streamk == 2 ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
#define DCL_SK_LAUNCH(PFV)                                                                                    \
    do {                                                                                                      \
        if (use_col)                                                                                          \
            hipLaunchKernelGGL((k_sweep<MODE_BWD, true, true, true, PFV>), dim3(G), dim3(256), 0, st, p);     \
        else                                                                                                  \
            hipLaunchKernelGGL((k_sweep<MODE_BWD, false, true, true, PFV>), dim3(G), dim3(256), 0, st, p);    \
    } while (0)
    if (g_sk_prefetch == 1)
        DCL_SK_LAUNCH(1);
    else if (g_sk_prefetch == 3)
        DCL_SK_LAUNCH(3);
    else
        DCL_SK_LAUNCH(2);
#undef DCL_SK_LAUNCH
    DCL_LAUNCH_CHECK();
    return 0;
}
