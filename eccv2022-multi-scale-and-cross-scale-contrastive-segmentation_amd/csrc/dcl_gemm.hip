// dcl_gemm.hip -- C[M, N] = A[M, K] . B[N, K]^T (+ bias) on the f16 matrix cores at fp32-equivalent accuracy
// (split-f16: hi.hi + hi.lo + lo.hi with f32 accumulation), f32 operands straight from HBM, gfx950 only.
//
// Replaces the library's fp32 GEMMs behind the token-major Linears of the Swin backbone (reference models/Swin.py:62-76
// Mlp, :198-230 qkv / proj, :357-362 PatchMerging reduction: forward y = x W^T + b, data gradient dx = dy W, weight
// gradient dW = dy^T x), the decoder's and HRNet's large 1x1 convolutions (models/HRNet.py:63-100, :236-262; batched
// over the images) and the head's tap products ("the head without its up-sampled input").
//
// Both operands are addressed as (row, k): "k-major" = the contraction index is the contiguous one (x[m][k], W[n][k]),
// otherwise the ROW index is contiguous (element (r, k) at p[k * ld + r]: dy^T, x^T, W read as its transpose).  All
// four combinations run through one LDS image, so the matrix loop is the same code:
//
//   staging   global f32 -> registers (one k-step of 32 ahead, 8 x 16-byte loads per thread) -> split into hi / lo f16
//             ONCE per element -> LDS.  k-major rows: a lane loads 4 consecutive k (8 lanes = one 128-byte line of a
//             row).  Row-major-contiguous operands: a lane loads 4 consecutive rows at 8 consecutive k and transposes
//             the 8 x 4 block in its registers, so the LDS writes are whole 16-byte fragments either way.
//   LDS       [k-group of 8][row] 16-byte units (hi) + the same (lo): an MFMA operand fragment (32 rows x 8 k per
//             half-wave) is one ds_read_b128 per lane from 256 consecutive bytes per 16 lanes -- conflict-free; rows are
//             rotated within 16-row groups and the k-groups padded by 64 bytes so that both kinds of WRITES are
//             conflict-free too.  Two stages (double buffer), one workgroup barrier per k-step.
//   waves     WM x WN waves, each TM x TN accumulator tiles of 32 x 32 (v_mfma_f32_32x32x16_f16); the first half of
//             the waves splits + stores the next stage BEFORE its matrix work, the second half AFTER it: the two waves
//             that share a SIMD are in opposite phases, one's vector work runs under the other's MFMAs.
//   grid      one workgroup per (tile, batch, k-split), XCD-contiguous tile ranges; k-splits write slabs that a second
//             kernel sums in fixed order (deterministic), used where M x N alone cannot fill 256 CUs (weight gradients).
//
// Roofline: MFMA; algorithmic FLOP = 2 M N K, issued as 3 f16 passes -> peak 2500 / 3 = 833 TFLOP/s.
#include "dcl_common.h"
#include <type_traits>

// A/B and bound probes (tools/probes/gemm_ab.sh); the product build leaves them at their defaults
#ifndef DCL_GEMM_PROBE
#define DCL_GEMM_PROBE 0        // bits: 1 no MFMAs, 2 no split + LDS stores, 4 no global loads, 8 no split (raw bits
                                // stored), 16 split but no LDS stores, 32 no result stores (results wrong)
#endif
#ifndef DCL_GEMM_PINGPONG
#define DCL_GEMM_PINGPONG 1     // 0: every wave stages before its matrix work
#endif
#ifndef DCL_GEMM_PF
#define DCL_GEMM_PF 0           // 1: fragment reads issued one MFMA group ahead (fenced; measured: no gain); 0: compiler order
#endif
#define GEMM_MFMA(A, B, C) ((DCL_GEMM_PROBE & 1) ? (C) : __builtin_amdgcn_mfma_f32_32x32x16_f16((A), (B), (C), 0, 0, 0))

namespace {

constexpr float F16_TARGET_G = 16384.0f;

__device__ __forceinline__ float pow2_scale_g(float amax)
{
    return amax == 0.f ? 1.f : exp2f(fminf(fmaxf(floorf(log2f(F16_TARGET_G / amax)), -100.f), 100.f));
}

// packed f16 pairs of hi = f16(v * s), lo = f16(v * s - hi) for two values (see dcl_conv3x3.hip)
__device__ __forceinline__ void split2g(float v0, float v1, float s, unsigned &hi, unsigned &lo)
{
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hi) : "v"(v0), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hi) : "v"(v1), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=&v"(lo) : "v"(v0), "v"(s), "v"(hi));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(v1), "v"(s), "v"(hi));
}

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

struct GemmArgs {
    const float *A, *B;
    float *C;                       // result, or the slab workspace when splitk > 1
    const float *bias;              // [N] or null (ignored when splitk > 1: the slab sum adds it)
    const float *a_amax, *b_amax;   // a_count / b_count partial maxima each: upper bounds of max|A|, max|B|
    int a_count, b_count;
    float *c_amax;                  // optional: atomic max of |C| (as int bits; caller zero-initialises)
    long lda, ldb, ldc;
    long sA, sB, sC;                // batch strides in elements
    int M, N, K;
    int batch, splitk;
    int tiles_m, tiles_n;
    int accumulate;                 // C += (splitk == 1 only)
    float *rowsum;                  // optional (row-contiguous A only): [z][M] sums over k of A(m, k) -- the bias gradient of a
                                    // Linear rides on its weight-gradient GEMM (A = dy^T); written by the tile column 0 workgroups
    // fused epilogues (template parameter EP; splitk == 1 only).  aux / C2 are addressed like C (ldc, batch stride sC).
    //   EP_GELU_FWD  C = v (the pre-activation, kept for the backward), C2 = gelu(v)          (Mlp.fc1 + act, Swin.py:62-76)
    //   EP_GELU_BWD  C = v * gelu'(aux)                                                        (data gradient of fc2 -> fc1's dy)
    //   EP_RESIDUAL  C = aux + rowscale[row / rows_per_scale] * v  (rowscale NULL: 1)          (shortcut + drop_path(branch), :318-321)
    float *C2;
    const float *aux;
    const float *rowscale;
    int rows_per_scale;
    // A-operand scale (template parameter ASC): element (row, k) of A is multiplied by a_scale[t / a_scale_group], t = the TOKEN
    // index of the element = its row for a k-major A (a Linear's data gradient: A = dy) and its k for a row-contiguous A (the
    // weight gradient: A = dy^T).  The per-sample factor of DropPath on a branch's gradient (Swin.py:318-321: dy_branch = dy *
    // mask / keep) rides in the power-of-two scale of the f16 split: no pass over dy.  Row-contiguous A: a_scale_group % 32 == 0.
    const float *a_scale;
    int a_scale_group;
};

enum { EP_NONE = 0, EP_GELU_FWD = 1, EP_GELU_BWD = 2, EP_RESIDUAL = 3 };

// nn.GELU() (approximate = 'none') and its derivative -- ATen's formulas (aten/src/ATen/native/cuda/ActivationGeluKernel.cu:
// 0.5 x (1 + erf(x / sqrt 2)); cdf + x pdf) -- for TWO values at a time on the packed f32 instructions (v_pk_fma_f32: one issue
// slot per pair).  The epilogue of a tile is exposed time (one workgroup per CU: nothing else runs under it), and with the math
// library's erff it cost 46 vector instructions per value.  erf: N. Juffa's single-precision form, two polynomials joined at
// |a| = 0.927734375, < 1 ulp (checked against float64 over [-6, 6]: 0.99 ulp); both are evaluated for the pair, branch-free.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 sp2(float v) { return f32x2{v, v}; }
__device__ __forceinline__ f32x2 erf2(f32x2 a)
{
    const f32x2 t = {fabsf(a.x), fabsf(a.y)};
    const f32x2 s = a * a;
    f32x2 r = pk_fma(sp2(-1.72853470e-5f), t, sp2(3.83197126e-4f));
    const f32x2 u = pk_fma(sp2(-3.88396438e-3f), t, sp2(2.42546219e-2f));
    r = pk_fma(r, s, u);
    r = pk_fma(r, t, sp2(-1.06777877e-1f));
    r = pk_fma(r, t, sp2(-6.34846687e-1f));
    r = pk_fma(r, t, sp2(-1.28717512e-1f));
    r = pk_fma(r, t, -t);
    f32x2 q = pk_fma(sp2(-5.96761703e-4f), s, sp2(4.99119423e-3f));
    q = pk_fma(q, s, sp2(-2.67681349e-2f));
    q = pk_fma(q, s, sp2(1.12819925e-1f));
    q = pk_fma(q, s, sp2(-3.76125336e-1f));
    q = pk_fma(q, s, sp2(1.28379166e-1f));
    q = pk_fma(q, a, a);
    const float bx = copysignf(1.0f - __expf(r.x), a.x), by = copysignf(1.0f - __expf(r.y), a.y);
    return f32x2{t.x > 0.927734375f ? bx : q.x, t.y > 0.927734375f ? by : q.y};
}
__device__ __forceinline__ f32x2 gelu2(f32x2 x)
{
    const f32x2 e = erf2(x * sp2(0.70710678118654752440f));
    return pk_fma(sp2(0.5f) * x, e, sp2(0.5f) * x);
}
__device__ __forceinline__ f32x2 gelu_grad2(f32x2 x)
{
    const f32x2 e = erf2(x * sp2(0.70710678118654752440f));
    const f32x2 cdf = pk_fma(sp2(0.5f), e, sp2(0.5f));
    const f32x2 h = sp2(-0.5f) * x * x;
    const f32x2 pdf = f32x2{__expf(h.x), __expf(h.y)} * sp2(0.39894228040143267794f);
    return pk_fma(x, pdf, cdf);
}

// LDS image of one operand tile and stage: [hi | lo][k-group of 8 k: 4][slot: BX + 2] 16-byte units.  The two pad units
// make the k-group stride 8 banks (mod 32, the store banking) so that a 16-lane group of ds_write_b64 -- two rows x four
// k-groups -- lands on 32 distinct banks.  Slot of row r: k-major operands: r (ds_read_b128 serves the lane groups
// {0-3, 12-15, 20-27} and {4-11, 16-19, 28-31}: 16 distinct 16-byte slots of the 256-byte bank row each).  Row-contiguous
// operands, whose stores are 8-lane groups writing rows 4 l + i: within a 32-row block, row 4 m + i sits at slot
// 8 i + (m ^ (i >> 1)) -- the 8 lanes of a store hit 8 distinct slots (mod 8), and a read group's 16 rows stay on 16
// distinct slots (mod 16) because {0, 3, 5, 6} ^ 1 = {1, 2, 4, 7}.
template <bool KM>
__device__ __forceinline__ int slot_of(int r)
{
    if constexpr (KM)
        return r;
    const int m = (r >> 2) & 7, i = r & 3;
    return (r & ~31) | (8 * i + (m ^ (i >> 1)));
}

// One thread's share of an operand tile (BX rows x 32 k = BX units of 32 elements; unit u): where it loads from and
// where it stores to.  All per-k-step address arithmetic is scalar (a uniform base pointer advances; the lane part is a
// 32-bit byte offset computed once).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct StageState {
    unsigned voff[8];               // lane byte offsets into the operand (buffer addressing: 32-bit, bounds-checked)
    unsigned wa, wb;                // LDS byte offsets (from the operand's hi image of stage 0)
    unsigned step, ldb;             // bytes per k-step (scalar offset advance); row-contiguous: bytes per k row
};

template <int BX, bool KM>
struct Stager {
    static constexpr int KG = BX + 2;

    static __device__ __forceinline__ void init(StageState &q, int u, long ld, int row0, int rows)
    {
        if constexpr (KM) {
            const int k4 = u & 7;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = min(row0 + (u >> 3) + j * (BX / 8), rows - 1);
                q.voff[j] = (unsigned)(((long)r * ld + 4 * k4) * 4);
            }
            q.wa = (unsigned)((((k4 >> 1) * KG + (u >> 3)) * 2 + (k4 & 1)) * 8);
            q.wb = 0;
            q.step = 32 * 4;
            q.ldb = 0;
        } else {
            const int m4 = u % (BX / 4), kg = u / (BX / 4);
            const int r = min(row0 + 4 * m4, rows - 4);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                q.voff[j] = (unsigned)(((long)(8 * kg) * ld + r) * 4);
            const int blk = m4 >> 3, m = m4 & 7;
            q.wa = (unsigned)((kg * KG + 32 * blk + m) * 16);
            q.wb = (unsigned)((kg * KG + 32 * blk + (m ^ 1)) * 16);
            q.step = (unsigned)(32 * ld * 4);
            q.ldb = (unsigned)(ld * 4);
        }
    }
    static __device__ __forceinline__ void load(const StageState &q, __amdgpu_buffer_rsrc_t rsrc, unsigned soff,
                                                float4 (&v)[8])
    {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            u32x4 w;
            if constexpr (KM)
                w = __builtin_amdgcn_raw_buffer_load_b128(rsrc, q.voff[j], soff, 0);
            else
                w = __builtin_amdgcn_raw_buffer_load_b128(rsrc, q.voff[0], soff + j * q.ldb, 0);
            v[j] = __builtin_bit_cast(float4, w);
        }
    }
    // rs: optional per-row scales of this thread's 8 rows (k-major operands with an A-operand scale), else s for all
    static __device__ __forceinline__ void store(const StageState &q, char *__restrict__ hi, float s,
                                                 const float4 (&v)[8], const float *rs = nullptr)
    {
        if constexpr (KM) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                uint2 h, l;
                const float sj = rs ? rs[j] : s;
                if (DCL_GEMM_PROBE & 8) {
                    h = uint2{__float_as_uint(v[j].x), __float_as_uint(v[j].y)};
                    l = uint2{__float_as_uint(v[j].z), __float_as_uint(v[j].w)};
                } else {
                    split2g(v[j].x, v[j].y, sj, h.x, l.x);
                    split2g(v[j].z, v[j].w, sj, h.y, l.y);
                }
                if (DCL_GEMM_PROBE & 16) {
                    asm volatile("" ::"v"(h.x), "v"(h.y), "v"(l.x), "v"(l.y));
                    continue;
                }
                *reinterpret_cast<uint2 *>(hi + q.wa + j * (BX / 8) * 16) = h;
                *reinterpret_cast<uint2 *>(hi + q.wa + j * (BX / 8) * 16 + 4 * KG * 16) = l;
            }
        } else {
            const float *f = reinterpret_cast<const float *>(&v[0]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint4 h, l;
                split2g(f[0 * 4 + i], f[1 * 4 + i], s, h.x, l.x);
                split2g(f[2 * 4 + i], f[3 * 4 + i], s, h.y, l.y);
                split2g(f[4 * 4 + i], f[5 * 4 + i], s, h.z, l.z);
                split2g(f[6 * 4 + i], f[7 * 4 + i], s, h.w, l.w);
                char *p = hi + ((i >> 1) ? q.wb : q.wa) + 8 * i * 16;
                *reinterpret_cast<uint4 *>(p) = h;
                *reinterpret_cast<uint4 *>(p + 4 * KG * 16) = l;
            }
        }
    }
};

__device__ __forceinline__ void lds_barrier()
{
    // LDS traffic only: __syncthreads() would also drain vmcnt, i.e. the global loads of the NEXT k-step
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int TM, int TN, int WM, int WN, bool AKM, bool BKM, int EP = EP_NONE, bool ASC = false>
__global__ __launch_bounds__(64 * WM * WN) void k_gemm(GemmArgs a)
{
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, NT = 64 * WM * WN;
    constexpr int KGA = BM + 2, KGB = BN + 2;
    constexpr int A_BYTES = 8 * KGA * 16, B_BYTES = 8 * KGB * 16;
    constexpr int STAGE = A_BYTES + B_BYTES;            // bytes per stage: A hi | A lo | B hi | B lo
    static_assert(BM + BN <= NT, "one staging unit per thread");
    static_assert(BM % 64 == 0 && BN % 64 == 0, "staging roles are wave-uniform");
    extern __shared__ uint4 lds_u4[];
    char *lds = reinterpret_cast<char *>(lds_u4);

    // ---- which tile: XCD x gets a contiguous range of the (z, tile_m, tile_n) order
    const int G = gridDim.x;
    int g = blockIdx.x;
    {
        const int x = g & 7, i = g >> 3;
        g = x * (G >> 3) + min(x, G & 7) + i;
    }
    const int tiles = a.tiles_m * a.tiles_n;
    const int z = g / tiles, t = g - z * tiles;
    const int tm = t / a.tiles_n, tn = t - tm * a.tiles_n;
    const int b = z / a.splitk, ks = z - b * a.splitk;
    const int nk = (a.K + 31) / 32;     // ragged only with two row-contiguous operands: k rows past K read as 0 (buffer bounds)
    const int kbeg = (int)((long)nk * ks / a.splitk), kend = (int)((long)nk * (ks + 1) / a.splitk);
    const int row0 = tm * BM, col0 = tn * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const bool stA = wave < BM / 64;                    // wave-uniform staging roles
    const bool stB = !stA && wave < (BM + BN) / 64;
    const bool early = !DCL_GEMM_PINGPONG || wave < (WM * WN) / 2;            // split + store before (true) or after (false) the MFMAs:
                                                        // waves w and w + 4 share a SIMD
    float ma = 0.f, mb = 0.f;
    for (int i = lane; i < a.a_count; i += 64)
        ma = fmaxf(ma, a.a_amax[i]);
    for (int i = lane; i < a.b_count; i += 64)
        mb = fmaxf(mb, a.b_amax[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ma = fmaxf(ma, __shfl_xor(ma, o, 64));
        mb = fmaxf(mb, __shfl_xor(mb, o, 64));
    }
    const float sa = pow2_scale_g(ma), sb = pow2_scale_g(mb);

    // the operand this wave stages, as a bounds-checked buffer (reads past the last element return 0), and the scalar
    // byte offset of the k-step it loads next
    const float *gbase = stA ? a.A + (long)b * a.sA : a.B + (long)b * a.sB;
    const long gelems = stA ? (AKM ? (long)(a.M - 1) * a.lda + a.K : (long)(a.K - 1) * a.lda + a.M)
                            : (BKM ? (long)(a.N - 1) * a.ldb + a.K : (long)(a.K - 1) * a.ldb + a.N);
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(gbase), 0, (int)(unsigned)(gelems * 4), 0x00020000);
    StageState q;
    if (stA)
        Stager<BM, AKM>::init(q, tid, a.lda, row0, a.M);
    else
        Stager<BN, BKM>::init(q, stB ? tid - BM : 0, a.ldb, col0, a.N);
    const unsigned gstep = q.step;
    unsigned goff = (unsigned)kbeg * gstep;
    // A-operand scale: k-major A -> one factor per staged row (this thread's 8 rows, fixed for the launch); row-contiguous A ->
    // one factor per k-step (32 consecutive tokens of one sample), fetched one k-step before the split that uses it
    // (row-contiguous A: the factors -- at most 64, checked by the entry point -- sit in one VGPR, lane l = factor l, and are picked
    // with v_readlane by a scalar index that is advanced with the k-steps: no memory operation inside the k loop.  A scalar load per
    // k-step made the compiler wait on lgkmcnt, the counter of the LDS fragment reads as well.)
    float arow[8];
    float atab = 0.f, ak_cur = 1.f;
    int aidx = 0, arem = 0;                 // sample of the NEXT k-step to be stored, and that k-step's first token within it
    if constexpr (ASC) {
        if constexpr (AKM) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = min(row0 + (tid >> 3) + j * (BM / 8), a.M - 1);
                arow[j] = stA ? sa * a.a_scale[r / a.a_scale_group] : sa;
            }
        } else {
            const int nfac = (a.K + a.a_scale_group - 1) / a.a_scale_group;
            atab = lane < nfac ? a.a_scale[lane] : 0.f;
            aidx = __builtin_amdgcn_readfirstlane((kbeg * 32) / a.a_scale_group);
            arem = __builtin_amdgcn_readfirstlane(kbeg * 32 - aidx * a.a_scale_group);
            ak_cur = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, atab), min(aidx, 63)));
            arem += 32;
            if (arem >= a.a_scale_group) {
                arem -= a.a_scale_group;
                ++aidx;
            }
        }
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[i][j][r] = 0.f;

    // fragment addresses: lane (row l & 31, k-group l >> 5 of the substep); tiles, substeps and hi / lo are immediates
    const unsigned fa = (unsigned)(((lane >> 5) * KGA + slot_of<AKM>(32 * wm * TM + (lane & 31))) * 16);
    const unsigned fb = (unsigned)(A_BYTES + ((lane >> 5) * KGB + slot_of<BKM>(32 * wn * TN + (lane & 31))) * 16);

    float4 v0[8];               // the k-step in flight: loaded one k-step before its split + LDS store
    auto load = [&](float4 (&v)[8]) {
        if (DCL_GEMM_PROBE & 4)
            return;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            v[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, q.voff[j], goff + j * q.ldb, 0));
        goff += gstep;
    };
    float rs[4] = {0.f, 0.f, 0.f, 0.f};     // row sums of this thread's four A rows over the k rows it staged
    // ak: (ASC, row-contiguous A) the factor of the k-step being stored
    auto store = [&](int stage, const float4 (&v)[8], float ak = 1.f) {
        if (DCL_GEMM_PROBE & 2)
            return;
        char *base = lds + stage * STAGE;
        if (!AKM && stA && a.rowsum) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if constexpr (ASC) {
                    rs[0] = fmaf(ak, v[j].x, rs[0]);
                    rs[1] = fmaf(ak, v[j].y, rs[1]);
                    rs[2] = fmaf(ak, v[j].z, rs[2]);
                    rs[3] = fmaf(ak, v[j].w, rs[3]);
                } else {
                    rs[0] += v[j].x;
                    rs[1] += v[j].y;
                    rs[2] += v[j].z;
                    rs[3] += v[j].w;
                }
            }
        }
        if (stA) {
            if constexpr (ASC && AKM)
                Stager<BM, AKM>::store(q, base, sa, v, arow);
            else
                Stager<BM, AKM>::store(q, base, ASC ? sa * ak : sa, v);
        } else if (stB)
            Stager<BN, BKM>::store(q, base + A_BYTES, sb, v);
    };

    // Loads are issued unconditionally (k-steps past the end read zeros through the buffer bounds, or the next k-split's
    // rows, and are never stored).  A second register set (loads two k-steps ahead) was built and bought nothing: the
    // compiler's vmcnt bookkeeping across the rotated loop waits for the NEWEST loads at every store (vmcnt(7..0) where
    // vmcnt(15..8) would do), and asm loads the compiler cannot see are not safe at 250+ live registers.
    load(v0);
    store(0, v0, ak_cur);
    load(v0);
    lds_barrier();

    // one k-step: `va` holds k-step k + 1 (stored into the other stage, then refilled with k-step k + 2)
    auto kstep = [&](int k, float4 (&va)[8]) {
        const int cur = (k - kbeg) & 1;
        const bool more = k + 1 < kend;
        float ak = 1.f;
        if constexpr (ASC && !AKM) {            // factor of k-step k + 1, the one stored in this trip
            ak = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, atab), min(aidx, 63)));
            arem += 32;
            if (arem >= a.a_scale_group) {
                arem -= a.a_scale_group;
                ++aidx;
            }
        }
        if (early) {
            if (more)
                store(cur ^ 1, va, ak);
            load(va);
        }
        const char *st = lds + cur * STAGE;
#if DCL_GEMM_PF
        // groups: X(s) = {A hi, B lo}, Y(s) = {A lo, B hi} of substep s; each is requested one MFMA group before its
        // first use (the fences keep the compiler from sinking the reads to their uses)
        h8 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
        auto readX = [&](int s) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                ah[s][i] = *reinterpret_cast<const h8 *>(st + fa + (32 * i + 2 * s * KGA) * 16);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bl[s][j] = *reinterpret_cast<const h8 *>(st + fb + (32 * j + 2 * s * KGB + 4 * KGB) * 16);
        };
        auto readY = [&](int s) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                al[s][i] = *reinterpret_cast<const h8 *>(st + fa + (32 * i + 2 * s * KGA + 4 * KGA) * 16);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bh[s][j] = *reinterpret_cast<const h8 *>(st + fb + (32 * j + 2 * s * KGB) * 16);
        };
        auto mm = [&](h8 (&x)[TM], h8 (&y)[TN]) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = GEMM_MFMA(x[i], y[j], acc[i][j]);
        };
        readX(0);
        readY(0);
        __builtin_amdgcn_sched_barrier(0);
        mm(ah[0], bl[0]);
        __builtin_amdgcn_sched_barrier(0);
        readX(1);
        __builtin_amdgcn_sched_barrier(0);
        mm(al[0], bh[0]);
        mm(ah[0], bh[0]);
        __builtin_amdgcn_sched_barrier(0);
        readY(1);
        __builtin_amdgcn_sched_barrier(0);
        mm(ah[1], bl[1]);
        mm(al[1], bh[1]);
        mm(ah[1], bh[1]);
        __builtin_amdgcn_sched_barrier(0);
#else
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            h8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ah[i] = *reinterpret_cast<const h8 *>(st + fa + (32 * i + 2 * s * KGA) * 16);
                al[i] = *reinterpret_cast<const h8 *>(st + fa + (32 * i + 2 * s * KGA + 4 * KGA) * 16);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bh[j] = *reinterpret_cast<const h8 *>(st + fb + (32 * j + 2 * s * KGB) * 16);
                bl[j] = *reinterpret_cast<const h8 *>(st + fb + (32 * j + 2 * s * KGB + 4 * KGB) * 16);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = GEMM_MFMA(ah[i], bl[j], acc[i][j]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = GEMM_MFMA(al[i], bh[j], acc[i][j]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = GEMM_MFMA(ah[i], bh[j], acc[i][j]);
        }
#endif
        if (!early) {
            if (more)
                store(cur ^ 1, va, ak);
            load(va);
        }
        lds_barrier();
    };
    for (int k = kbeg; k < kend; ++k)
        kstep(k, v0);

    if (!AKM && a.rowsum && tn == 0) {       // (uniform) the four k-groups of a row meet in LDS; the stages are free now
        float *red = reinterpret_cast<float *>(lds);
        if (stA) {
            const int m4 = tid % (BM / 4), kg = tid / (BM / 4);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                red[kg * BM + 4 * m4 + i] = rs[i];
        }
        __syncthreads();
        if (tid < BM && row0 + tid < a.M)
            a.rowsum[(long)z * a.M + row0 + tid] = (red[tid] + red[BM + tid]) + (red[2 * BM + tid] + red[3 * BM + tid]);
    }
    // ---- epilogue: acc register r of tile (i, j) is row 8 (r / 4) + 4 (lane / 32) + r % 4, column lane % 32
    const float inv = 1.0f / (sa * sb);
    float *C = a.C + (long)z * a.sC;                    // sC: batch stride, or the slab stride when splitk > 1
    float mx = 0.f;
    const bool full = row0 + BM <= a.M && col0 + BN <= a.N;     // uniform: interior tiles store without per-element tests
    auto epilogue = [&](auto full_c) {
        constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = col0 + 32 * (wn * TN + j) + (lane & 31);
            const bool cok = FULL || col < a.N;
            const float bv = (a.bias && cok) ? a.bias[col] : 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int rbase = row0 + 32 * (wm * TM + i) + 4 * (lane >> 5);
                float *p0 = C + (long)rbase * a.ldc + col;
                if constexpr (EP == EP_NONE) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int dr = 8 * (r >> 2) + (r & 3);
                        if (FULL || (cok && rbase + dr < a.M)) {
                            float *p = p0 + (long)dr * a.ldc;
                            float val = acc[i][j][r] * inv + bv;
                            if (a.accumulate)
                                val += *p;
                            if (!(DCL_GEMM_PROBE & 32))
                                *p = val;
                            mx = fmaxf(mx, fabsf(val));
                        }
                    }
                } else {
                    // 32-bit element offsets from the uniform bases of C / C2 / aux (the entry point checks M * ldc < 2^30): one
                    // address register per element instead of a 64-bit pointer per array -- with those the 256 x 256 tile spilled
                    const unsigned ldc32 = (unsigned)a.ldc;
                    const unsigned o0 = (unsigned)rbase * ldc32 + (unsigned)col;
                    const float *__restrict__ auxz = a.aux + (long)z * a.sC;
                    float *__restrict__ c2z = a.C2 + (long)z * a.sC;
                    // the second operand of the epilogue (h / the shortcut) is requested for the whole 32 x 32 tile before the
                    // first value is used: 16 independent loads in flight per lane
                    float xin[16];
                    if constexpr (EP == EP_GELU_BWD || EP == EP_RESIDUAL) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int dr = 8 * (r >> 2) + (r & 3);
                            xin[r] = (FULL || (cok && rbase + dr < a.M)) ? auxz[o0 + (unsigned)dr * ldc32] : 0.f;
                        }
                    }
                    // per-sample factor of the rows of this tile: ONE division per tile -- rows_per_scale >= 32 (checked by the entry
                    // point) and dr <= 27, so row (rbase + dr) lies in sample q0 or q0 + 1
                    int q0 = 0, rem0 = 0;
                    if constexpr (EP == EP_RESIDUAL) {
                        if (a.rowscale) {
                            q0 = rbase / a.rows_per_scale;
                            rem0 = rbase - q0 * a.rows_per_scale;
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const int dr = 8 * (r >> 2) + (r & 3);                  // rows dr, dr + 1
                        const bool ok0 = FULL || (cok && rbase + dr < a.M), ok1 = FULL || (cok && rbase + dr + 1 < a.M);
                        const unsigned o = o0 + (unsigned)dr * ldc32;
                        f32x2 val = pk_fma(f32x2{acc[i][j][r], acc[i][j][r + 1]}, sp2(inv), sp2(bv));
                        if constexpr (EP == EP_GELU_FWD) {
                            const f32x2 gv = gelu2(val);
                            if (ok0) {
                                C[o] = val.x;
                                c2z[o] = gv.x;
                            }
                            if (ok1) {
                                C[o + ldc32] = val.y;
                                c2z[o + ldc32] = gv.y;
                            }
                        } else {
                            if constexpr (EP == EP_GELU_BWD) {
                                val = val * gelu_grad2(f32x2{xin[r], xin[r + 1]});
                            } else {
                                f32x2 sc = sp2(1.f);
                                if (a.rowscale) {
                                    // (rows past M in a partial tile: clamped to the last sample's factor, never read past the
                                    // [M / rows_per_scale] array; their results are not stored)
                                    const int qmax = (a.M - 1) / a.rows_per_scale;
                                    sc.x = a.rowscale[min(q0 + (rem0 + dr >= a.rows_per_scale ? 1 : 0), qmax)];
                                    sc.y = a.rowscale[min(q0 + (rem0 + dr + 1 >= a.rows_per_scale ? 1 : 0), qmax)];
                                }
                                val = pk_fma(sc, val, f32x2{xin[r], xin[r + 1]});
                            }
                            if (ok0)
                                C[o] = val.x;
                            if (ok1)
                                C[o + ldc32] = val.y;
                        }
                        if (ok0)
                            mx = fmaxf(mx, fabsf(val.x));
                        if (ok1)
                            mx = fmaxf(mx, fabsf(val.y));
                    }
                }
            }
        }
    };
    if (full)
        epilogue(std::true_type{});
    else
        epilogue(std::false_type{});
    if (a.c_amax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
            mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        if (lane == 0 && mx > __builtin_nontemporal_load(a.c_amax))
            atomicMax(reinterpret_cast<int *>(a.c_amax), __float_as_int(mx));
    }
}

// C[i] = (accumulate ? C[i] : 0) + bias[col] + sum over the slabs in ascending order; 4 x 4 elements per thread
__global__ __launch_bounds__(256) void k_gemm_slab_sum(const float *__restrict__ ws, int slabs, long slab_stride,
                                                       const float *__restrict__ bias, float *__restrict__ C, long ldc,
                                                       int M, int N, int accumulate, float *c_amax)
{
    const long total = (long)M * N / 4;
    float mx = 0.f;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const long q = ((long)blockIdx.x * 4 + it) * 256 + threadIdx.x;     // quad index over M * N / 4
        if (q >= total)
            break;
        const long e = q * 4;
        const int row = (int)(e / N), col = (int)(e - (long)row * N);
        float4 s = *reinterpret_cast<const float4 *>(ws + e);
        // eight slabs' loads in flight, added in ascending order (one load + one dependent add per trip left the kernel waiting
        // a memory latency per slab: 18 us on average for 117 launches of a Swin-T step)
        int k = 1;
        for (; k + 8 <= slabs; k += 8) {
            float4 p[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                p[j] = *reinterpret_cast<const float4 *>(ws + (long)(k + j) * slab_stride + e);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                s.x += p[j].x, s.y += p[j].y, s.z += p[j].z, s.w += p[j].w;
        }
        for (; k < slabs; ++k) {
            const float4 p = *reinterpret_cast<const float4 *>(ws + (long)k * slab_stride + e);
            s.x += p.x, s.y += p.y, s.z += p.z, s.w += p.w;
        }
        if (bias)
            s.x += bias[col], s.y += bias[col + 1], s.z += bias[col + 2], s.w += bias[col + 3];
        float *p = C + (long)row * ldc + col;
        if (accumulate)
            s.x += p[0], s.y += p[1], s.z += p[2], s.w += p[3];
        *reinterpret_cast<float4 *>(p) = s;
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(s.x), fabsf(s.y)), fmaxf(fabsf(s.z), fabsf(s.w))));
    }
    if (c_amax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
            mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        // one atomic per wave at most, and only while it would still raise the maximum (a plain read first: tens of
        // thousands of same-address atomics serialise in L2 -- 0.4 ms on a 6400 x 1536 result)
        if ((threadIdx.x & 63) == 0 && mx > __builtin_nontemporal_load(c_amax))
            atomicMax(reinterpret_cast<int *>(c_amax), __float_as_int(mx));
    }
}

// out[m] = sum over the k-split slabs, ascending
__global__ __launch_bounds__(256) void k_gemm_rowsum_sum(const float *__restrict__ ws, int slabs, int M, float *__restrict__ out)
{
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M)
        return;
    float s = ws[m];
    for (int k = 1; k < slabs; ++k)
        s += ws[(long)k * M + m];
    out[m] = s;
}

int g_gemm_tile = 0;        // 0 = automatic; 1 = 256 x 256, 2 = 256 x 128, 3 = 128 x 256, 4 = 128 x 128, 5 = 256 x 192

struct TileCfg { int bm, bn, nt; };
constexpr int NTILES = 5;
constexpr TileCfg TILES[NTILES] = {{256, 256, 512}, {256, 128, 512}, {128, 256, 512}, {128, 128, 256}, {256, 192, 512}};

// Cost of running (tile c, k-split s) in units of one 256 x 256 x 32 tile step / 65536: rounds of 256 workgroups x the work
// of one workgroup (tile area x its k-steps + ~1.5 steps of prologue / epilogue; smaller tiles pay 4-25 % more per FLOP),
// plus the slab traffic of a split (each slab written, read by the sum, the result written: ~0.04 units per float at 5 TB/s).
// Tile and split are chosen TOGETHER: a weight gradient with 36 tiles of 256 x 256 is better served by 7 splits of those
// (252 workgroups, operand traffic ~ 1 / BM + 1 / BN) than by 4 splits of 72 smaller tiles (288 = a second, almost empty round).
double plan_cost(int c, int s, int M, int N, int K, int batch)
{
    const double eff[NTILES] = {1.0, 0.90, 0.90, 0.75, 0.96};
    const long nk = (K + 31) / 32;
    const long wg = (long)((M + TILES[c].bm - 1) / TILES[c].bm) * ((N + TILES[c].bn - 1) / TILES[c].bn) * batch * s;
    const long rounds = (wg + 255) / 256;
    const double steps = (double)((nk + s - 1) / s) + 1.5;
    double cost = (double)rounds * TILES[c].bm * TILES[c].bn / eff[c] * steps;
    if (s > 1)
        cost += (2.0 * s + 1.0) * (double)M * N * batch * 0.04;
    return cost;
}

// best tile for a given split (s >= 1), or best (tile, split) when *split == 0 (auto: splits up to K / 256, at most 64)
int plan_gemm(int M, int N, int K, int batch, int *split)
{
    const int forced = (g_gemm_tile >= 1 && g_gemm_tile <= NTILES) ? g_gemm_tile - 1 : -1;
    const long nk = (K + 31) / 32;
    const int smax = *split > 0 ? *split : (int)(nk / 8 < 1 ? 1 : (nk / 8 > 64 ? 64 : nk / 8));
    const int smin = *split > 0 ? *split : 1;
    int best = 0, bs = smin;
    double bc = 1e300;
    for (int c = 0; c < NTILES; ++c) {
        if (forced >= 0 && c != forced)
            continue;
        for (int s = smin; s <= smax; ++s) {
            const double cost = plan_cost(c, s, M, N, K, batch);
            if (cost < bc)
                bc = cost, best = c, bs = s;
        }
    }
    *split = bs;
    return best;
}

template <int TM, int TN, int WM, int WN, int EP = EP_NONE, bool ASC = false>
int launch_gemm(const GemmArgs &a, int akm, int bkm, hipStream_t stream)
{
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr size_t lds_bytes = 2 * (8 * (BM + 2) + 8 * (BN + 2)) * sizeof(uint4);
    const dim3 grid((unsigned)(a.tiles_m * a.tiles_n * a.batch * a.splitk)), block(64 * WM * WN);
#define DCL_GEMM_LAUNCH(AK, BK)                                                                                       \
    do {                                                                                                              \
        static bool attr_done = false;                                                                                \
        if (!attr_done) {                                                                                             \
            (void)hipFuncSetAttribute((const void *)k_gemm<TM, TN, WM, WN, AK, BK, EP, ASC>,                                   \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);                    \
            attr_done = true;                                                                                         \
        }                                                                                                             \
        hipLaunchKernelGGL((k_gemm<TM, TN, WM, WN, AK, BK, EP, ASC>), grid, block, lds_bytes, stream, a);                      \
    } while (0)
    // fused epilogues exist for the operand layouts of the products that use them: a Linear's forward (both k-major) and its
    // data gradient (dy k-major, W read as its transpose)
    if constexpr (EP == EP_GELU_FWD || EP == EP_RESIDUAL) {
        DCL_CHECK_ARG(akm && bkm, "this epilogue needs two k-major operands (a Linear's forward)");
        DCL_GEMM_LAUNCH(true, true);
    } else if constexpr (EP == EP_GELU_BWD) {
        DCL_CHECK_ARG(akm && !bkm, "the GELU-backward epilogue needs a k-major A and a row-contiguous B (a Linear's data gradient)");
        DCL_GEMM_LAUNCH(true, false);
    } else if constexpr (ASC) {
        // the A-operand scale exists for the two products of a Linear's backward: dy k-major / dy^T row-contiguous, W row-contiguous
        DCL_CHECK_ARG(!bkm, "the A-operand scale needs a row-contiguous B (a Linear's data / weight gradient)");
        if (akm)
            DCL_GEMM_LAUNCH(true, false);
        else
            DCL_GEMM_LAUNCH(false, false);
    } else {
    if (akm && bkm)
        DCL_GEMM_LAUNCH(true, true);
    else if (akm)
        DCL_GEMM_LAUNCH(true, false);
    else if (bkm)
        DCL_GEMM_LAUNCH(false, true);
    else
        DCL_GEMM_LAUNCH(false, false);
    }
#undef DCL_GEMM_LAUNCH
    DCL_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" int dcl_gemm_set_tile(int tile)
{
    DCL_CHECK_ARG(tile >= 0 && tile <= NTILES, "tile must be 0 (automatic) .. 5");
    g_gemm_tile = tile;
    return 0;
}

extern "C" int dcl_gemm_supported(int M, int N, int K, int64_t lda, int a_kmajor, int64_t ldb, int b_kmajor)
{
    if (M < 1 || N < 1 || K < 32)
        return 0;
    if (K % 32 && (a_kmajor || b_kmajor))       // a ragged contraction needs k rows that run out of the buffer
        return 0;
    if (lda % 4 || ldb % 4)
        return 0;
    if (!a_kmajor && (M % 4 || M < 4))
        return 0;
    if (!b_kmajor && (N % 4 || N < 4))
        return 0;
    // lane offsets are 32-bit byte offsets from a uniform base
    const int64_t ea = a_kmajor ? (int64_t)M * lda : (int64_t)(K + 31) * lda + M;
    const int64_t eb = b_kmajor ? (int64_t)N * ldb : (int64_t)(K + 31) * ldb + N;
    if (ea * 4 >= ((int64_t)1 << 32) || eb * 4 >= ((int64_t)1 << 32))
        return 0;
    return 1;
}

extern "C" int64_t dcl_gemm_workspace_floats(int M, int N, int batch, int splitk)
{
    return splitk > 1 ? ((int64_t)M * N + M) * batch * splitk : 0;       // slabs + the slabs of the optional row sums
}

extern "C" int dcl_gemm_suggest_splitk(int M, int N, int K, int batch)
{
    int s = 0;
    (void)plan_gemm(M, N, K, batch, &s);
    return s;
}

static int gemm_impl(const float *A, int64_t lda, int a_kmajor, int64_t strideA, const float *B, int64_t ldb,
                     int b_kmajor, int64_t strideB, int M, int N, int K, int batch, const float *a_amax,
                     int a_count, const float *b_amax, int b_count, const float *bias, float *C, int64_t ldc, int64_t strideC,
                     int accumulate, float *c_amax, int splitk, float *ws, float *a_rowsum, void *stream,
                     int ep, float *C2, const float *aux, const float *rowscale, int rows_per_scale,
                     const float *a_scale = nullptr, int a_scale_group = 1)
{
    DCL_CHECK_ARG(!a_scale || (a_scale_group >= 1 && batch == 1 && (a_kmajor || a_scale_group % 32 == 0)),
                  "A-operand scale: batch 1, group >= 1, and a multiple of 32 for a row-contiguous A");
    DCL_CHECK_ARG(!a_scale || ep == EP_NONE || ep == EP_GELU_BWD, "A-operand scale: plain or GELU-backward epilogue only");
    DCL_CHECK_ARG(!a_scale || a_kmajor || (K + a_scale_group - 1) / a_scale_group <= 64,
                  "A-operand scale on a row-contiguous A: at most 64 factors");
    DCL_CHECK_ARG(!a_rowsum || (!a_kmajor && batch == 1), "a_rowsum needs a row-contiguous A and batch 1");
    DCL_CHECK_ARG(A && B && C && a_amax && b_amax, "null pointer");
    DCL_CHECK_ARG(batch >= 1 && splitk >= 1 && a_count >= 1 && b_count >= 1, "batch, splitk and the absmax counts must be >= 1");
    DCL_CHECK_ARG(dcl_gemm_supported(M, N, K, lda, a_kmajor, ldb, b_kmajor),
                  "unsupported shape (K % 32 unless both operands are row-contiguous, leading dimensions % 4, row-contiguous operands need rows % 4)");
    DCL_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0 && strideA % 4 == 0 && strideB % 4 == 0,
                  "operands must be 16-byte aligned");
    DCL_CHECK_ARG(splitk == 1 || (ws && N % 4 == 0 && ldc % 4 == 0 && ((uintptr_t)C % 16) == 0),
                  "split-k needs a workspace and N, ldc multiples of 4");
    DCL_CHECK_ARG(splitk <= (K + 31) / 32, "more k-splits than k-steps");
    GemmArgs a;
    a.A = A, a.B = B, a.lda = lda, a.ldb = ldb, a.sA = strideA, a.sB = strideB;
    a.a_amax = a_amax, a.b_amax = b_amax, a.a_count = a_count, a.b_count = b_count;
    a.M = M, a.N = N, a.K = K, a.batch = batch, a.splitk = splitk;
    a.rowsum = a_rowsum ? (splitk > 1 ? ws + (long)M * N * batch * splitk : a_rowsum) : nullptr;
    a.C2 = C2, a.aux = aux, a.rowscale = rowscale, a.rows_per_scale = rows_per_scale > 0 ? rows_per_scale : 1;
    a.a_scale = a_scale, a.a_scale_group = a_scale_group;
    if (splitk > 1) {
        a.C = ws, a.ldc = N, a.sC = (long)M * N, a.bias = nullptr, a.c_amax = nullptr, a.accumulate = 0;
    } else {
        a.C = C, a.ldc = ldc, a.sC = strideC, a.bias = bias, a.c_amax = c_amax, a.accumulate = accumulate;
    }
    int sfix = splitk;
    int c = plan_gemm(M, N, K, batch, &sfix);
    if (ep != EP_NONE && c == 0)
        c = 4;          // the fused epilogues on 8 accumulator tiles per wave spill (144-172 B): 256 x 192 instead of 256 x 256
    a.tiles_m = (M + TILES[c].bm - 1) / TILES[c].bm;
    a.tiles_n = (N + TILES[c].bn - 1) / TILES[c].bn;
    int rc;
#define DCL_GEMM_TILES(EPV, ASCV)                                                                                \
    switch (c) {                                                                                                 \
    case 0: rc = launch_gemm<2, (EPV == EP_NONE ? 4 : 3), 4, 2, EPV, ASCV>(a, a_kmajor, b_kmajor, (hipStream_t)stream); break; \
    case 1: rc = launch_gemm<2, 2, 4, 2, EPV, ASCV>(a, a_kmajor, b_kmajor, (hipStream_t)stream); break;          \
    case 2: rc = launch_gemm<2, 2, 2, 4, EPV, ASCV>(a, a_kmajor, b_kmajor, (hipStream_t)stream); break;          \
    case 4: rc = launch_gemm<2, 3, 4, 2, EPV, ASCV>(a, a_kmajor, b_kmajor, (hipStream_t)stream); break;          \
    default: rc = launch_gemm<2, 2, 2, 2, EPV, ASCV>(a, a_kmajor, b_kmajor, (hipStream_t)stream); break;         \
    }
    if (a_scale) {
        if (ep == EP_GELU_BWD) {
            DCL_GEMM_TILES(EP_GELU_BWD, true);
        } else {
            DCL_GEMM_TILES(EP_NONE, true);
        }
    } else {
        switch (ep) {
        case EP_GELU_FWD: DCL_GEMM_TILES(EP_GELU_FWD, false); break;
        case EP_GELU_BWD: DCL_GEMM_TILES(EP_GELU_BWD, false); break;
        case EP_RESIDUAL: DCL_GEMM_TILES(EP_RESIDUAL, false); break;
        default: DCL_GEMM_TILES(EP_NONE, false); break;
        }
    }
#undef DCL_GEMM_TILES
    if (rc != 0)
        return rc;
    if (splitk > 1) {
        for (int b = 0; b < batch; ++b) {
            const long quads = (long)M * N / 4;
            hipLaunchKernelGGL(k_gemm_slab_sum, dim3((unsigned)((quads + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream,
                               ws + (long)b * splitk * M * N, splitk, (long)M * N, bias, C + (long)b * strideC, (long)ldc, M, N,
                               accumulate, c_amax);
        }
        if (a_rowsum)
            hipLaunchKernelGGL(k_gemm_rowsum_sum, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                               ws + (long)M * N * batch * splitk, splitk, M, a_rowsum);
        DCL_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int dcl_gemm_f16x3(const float *A, int64_t lda, int a_kmajor, int64_t strideA, const float *B, int64_t ldb,
                              int b_kmajor, int64_t strideB, int M, int N, int K, int batch, const float *a_amax,
                              int a_count, const float *b_amax, int b_count, const float *bias, float *C, int64_t ldc, int64_t strideC,
                              int accumulate, float *c_amax, int splitk, float *ws, float *a_rowsum, void *stream)
{
    return gemm_impl(A, lda, a_kmajor, strideA, B, ldb, b_kmajor, strideB, M, N, K, batch, a_amax, a_count, b_amax, b_count, bias,
                     C, ldc, strideC, accumulate, c_amax, splitk, ws, a_rowsum, stream, EP_NONE, nullptr, nullptr, nullptr, 1);
}

extern "C" int dcl_gemm_f16x3_ascaled(const float *A, int64_t lda, int a_kmajor, const float *B, int64_t ldb, int b_kmajor, int M,
                                      int N, int K, const float *a_amax, int a_count, const float *b_amax, int b_count,
                                      float *C, int64_t ldc, float *c_amax, int splitk, float *ws, float *a_rowsum,
                                      const float *a_scale, int a_scale_group, int ep, const float *aux, void *stream)
{
    DCL_CHECK_ARG(a_scale, "null a_scale");
    DCL_CHECK_ARG(ep == EP_NONE || (ep == EP_GELU_BWD && aux && splitk == 1), "ep must be 0, or 2 (GELU backward) with aux and no k-split");
    DCL_CHECK_ARG(ep == EP_NONE || (ldc >= N && (int64_t)M * ldc < ((int64_t)1 << 30)), "the fused epilogue addresses C with 32-bit element offsets");
    return gemm_impl(A, lda, a_kmajor, 0, B, ldb, b_kmajor, 0, M, N, K, 1, a_amax, a_count, b_amax, b_count, nullptr, C, ldc, 0, 0,
                     c_amax, splitk, ws, a_rowsum, stream, ep, nullptr, aux, nullptr, 1, a_scale, a_scale_group);
}

extern "C" int dcl_gemm_f16x3_ep(const float *A, int64_t lda, int a_kmajor, const float *B, int64_t ldb, int b_kmajor, int M, int N,
                                 int K, const float *a_amax, int a_count, const float *b_amax, int b_count, const float *bias,
                                 float *C, int64_t ldc, float *c_amax, int ep, float *C2, const float *aux,
                                 const float *rowscale, int rows_per_scale, void *stream)
{
    DCL_CHECK_ARG(ep >= EP_GELU_FWD && ep <= EP_RESIDUAL, "ep must be 1 (GELU forward), 2 (GELU backward) or 3 (residual)");
    DCL_CHECK_ARG(ep != EP_GELU_FWD || C2, "the GELU-forward epilogue needs the second output C2");
    DCL_CHECK_ARG(ep == EP_GELU_FWD || aux, "this epilogue needs its second input aux");
    DCL_CHECK_ARG(!rowscale || rows_per_scale >= 32, "rows_per_scale must be >= 32 (a tile row group spans at most two samples)");
    DCL_CHECK_ARG(ldc >= N && (int64_t)M * ldc < ((int64_t)1 << 30), "the fused epilogues address C with 32-bit element offsets: M * ldc < 2^30");
    return gemm_impl(A, lda, a_kmajor, 0, B, ldb, b_kmajor, 0, M, N, K, 1, a_amax, a_count, b_amax, b_count, bias, C, ldc, 0, 0,
                     c_amax, 1, nullptr, nullptr, stream, ep, C2, aux, rowscale, rows_per_scale);
}
