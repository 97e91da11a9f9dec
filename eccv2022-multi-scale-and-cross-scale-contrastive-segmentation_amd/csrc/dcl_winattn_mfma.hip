// dcl_winattn_mfma.hip -- Swin window attention forward on the f16 matrix cores at fp32-equivalent accuracy (split-f16:
// hi.hi + hi.lo + lo.hi, f32 accumulation).  Same semantics, arguments and token mapping as k_winattn_fwd
// (dcl_winattn.hip: reference models/Swin.py:198-230 inside :286-318); that kernel computes a window's 49 x 49 scores and
// 49 x 32 outputs with in-lane fp32 FMAs (~3 500 vector instructions per lane and window-head), this one with 48 MFMAs.
//
// One wave per (image, window, head), waves persistent over the windows of ONE head (its bias table stays in LDS).
//   S^T = K Q^T     keys along the accumulator registers, queries along the lanes (v_mfma_f32_32x32x16_f16 C layout:
//                   register r of lane l = row 8 (r / 4) + 4 (l / 32) + r % 4, column l % 32): the softmax over the keys
//                   of a query is an in-lane reduction over 32 registers plus ONE cross-lane step (l ^ 32) -- with the
//                   queries along the registers it would be five shuffle steps per register.
//                   Operand fragments are 8 consecutive floats of a token's q / k row: loaded straight from the qkv
//                   tensor, split in registers; operand scales are powers of two from the wave's own absmax.
//   O^T = V^T P^T   P^T never leaves the registers: the accumulator registers 8 s .. 8 s + 7 of a key tile ARE the B
//                   fragment of k-step s once the contraction index of that step is taken in the order
//                   kappa(s, half, t) = 16 s + 8 (t / 4) + 4 half + t % 4; the A fragment (V^T) is gathered in the same
//                   order, 8 dword loads per k-step with the lanes along the 32 channels of a v row (128-byte segments).
// The zero-padded border tokens carry qkv = the projection's bias, shifted windows mask by region id, as in dcl_winattn.hip.
#include "dcl_common.h"

namespace {

constexpr int WS = 7, NT = 49, HD = 32;
constexpr int BST = 52;                 // row stride of the bias table in LDS (16-byte groups, conflict-free b128 reads)

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

struct WmArgs {
    const float *qkv;       // [B, H*W, 3*C]
    const float *qkv_bias;  // [3*C]: qkv of the padded tokens
    const float *bias;      // [heads, 49, 49]
    float *out;             // [B, H*W, C]
    float *lse;             // [B, nW, heads, 49] or null
    int B, H, W, Hp, Wp, shift, heads, C;
    int nWx, nW;
    float scale;
    int nwaves;
};

__device__ __forceinline__ void token_of_m(const WmArgs &a, int wy, int wx, int t, int &row, bool &real, int &rid)
{
    const int r = t / WS, c = t - r * WS;
    const int y = wy * WS + r, x = wx * WS + c;
    int ys = y + a.shift, xs = x + a.shift;
    ys -= ys >= a.Hp ? a.Hp : 0;
    xs -= xs >= a.Wp ? a.Wp : 0;
    real = ys < a.H && xs < a.W;
    row = real ? ys * a.W + xs : -1;
    const int ry = y < a.Hp - WS ? 0 : (y < a.Hp - a.shift ? 1 : 2);
    const int rx = x < a.Wp - WS ? 0 : (x < a.Wp - a.shift ? 1 : 2);
    rid = ry * 3 + rx;
}

__device__ __forceinline__ float pow2_scale_w(float amax)
{
    return amax == 0.f ? 1.f : exp2f(fminf(fmaxf(floorf(log2f(16384.0f / amax)), -100.f), 100.f));
}

__device__ __forceinline__ void split2w(float v0, float v1, float s, unsigned &hi, unsigned &lo)
{
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hi) : "v"(v0), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hi) : "v"(v1), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=&v"(lo) : "v"(v0), "v"(s), "v"(hi));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(v1), "v"(s), "v"(hi));
}

// 8 floats -> MFMA operand fragments (hi, lo)
__device__ __forceinline__ void split8(const float (&v)[8], float s, h8 &hi, h8 &lo)
{
    unsigned h0, h1, h2, h3, l0, l1, l2, l3;
    split2w(v[0], v[1], s, h0, l0);
    split2w(v[2], v[3], s, h1, l1);
    split2w(v[4], v[5], s, h2, l2);
    split2w(v[6], v[7], s, h3, l3);
    hi = __builtin_bit_cast(h8, u32x4v{h0, h1, h2, h3});
    lo = __builtin_bit_cast(h8, u32x4v{l0, l1, l2, l3});
}

__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_s_waitcnt(0xc07f);                   // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
}

#define WMFMA(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_f16((A), (B), (C), 0, 0, 0)

// The splits are inline asm: the compiler's hazard recognizer does not see a VALU write behind them, so an MFMA that reads
// a fragment right after its last v_fma_mixhi gets no wait states and can read the register before the high half has
// landed (observed: 1e-4 errors in one of ~10^5 scores).  A fenced s_nop between the splits and the MFMAs that consume them.
__device__ __forceinline__ void split_to_mfma_fence()
{
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 4");
    __builtin_amdgcn_sched_barrier(0);
}

__global__ __launch_bounds__(64) void k_winattn_fwd_mfma(WmArgs a)
{
    __shared__ __attribute__((aligned(16))) float Bs[NT * BST];       // bias of this wave's head, [query][key]
    __shared__ __attribute__((aligned(16))) int Ts[64], Rs[64];       // token row (-1 padded, -2 none) / region id
    const int lane = threadIdx.x;
    const int gw = blockIdx.x;
    const int hd = gw % a.heads;
    for (int idx = lane; idx < NT * NT; idx += 64) {
        const int q = idx / NT, k = idx - q * NT;
        Bs[q * BST + k] = a.bias[(size_t)hd * NT * NT + idx];
    }
    const int h = lane >> 5, l32 = lane & 31;
    const long long nbw = (long long)a.B * a.nW;
    const int stride = a.nwaves / a.heads;
    const size_t C3 = (size_t)3 * a.C;
    const float *bq = a.qkv_bias + hd * HD, *bk = bq + a.C, *bv = bk + a.C;

    for (long long bw = gw / a.heads; bw < nbw; bw += stride) {
        const int win = (int)(bw % a.nW), b = (int)(bw / a.nW);
        const int wy = win / a.nWx, wx = win - wy * a.nWx;
        wave_lds_sync();                                  // the previous window's table reads are done
        {
            int row, rid;
            bool real;
            token_of_m(a, wy, wx, lane < NT ? lane : NT - 1, row, real, rid);
            Ts[lane] = lane < NT ? (real ? b * a.H * a.W + row : -1) : -2;
            Rs[lane] = rid;
        }
        wave_lds_sync();
        // this lane's two tokens as key rows / query columns: t = 32 i + l32
        int tok[2] = {Ts[l32], Ts[32 + l32]};
        // ---- Q and K fragments: 8 consecutive channels (8 h + 16 ks ..) of the token's q / k row
        float qv[2][2][8], kv[2][2][8];
        float mq = 0.f, mk = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float *pq = tok[i] >= 0 ? a.qkv + (size_t)tok[i] * C3 + hd * HD : bq;
            const float *pk = tok[i] >= 0 ? pq + a.C : bk;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const f32x4 q0 = *(const f32x4 *)(pq + 8 * h + 16 * ks), q1 = *(const f32x4 *)(pq + 8 * h + 16 * ks + 4);
                const f32x4 k0 = *(const f32x4 *)(pk + 8 * h + 16 * ks), k1 = *(const f32x4 *)(pk + 8 * h + 16 * ks + 4);
                const float z = tok[i] == -2 ? 0.f : 1.f;
                const float qa[8] = {q0.x * z, q0.y * z, q0.z * z, q0.w * z, q1.x * z, q1.y * z, q1.z * z, q1.w * z};
                const float ka[8] = {k0.x * z, k0.y * z, k0.z * z, k0.w * z, k1.x * z, k1.y * z, k1.z * z, k1.w * z};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    qv[i][ks][e] = qa[e];
                    kv[i][ks][e] = ka[e];
                    mq = fmaxf(mq, fabsf(qa[e]));
                    mk = fmaxf(mk, fabsf(ka[e]));
                }
            }
        }
        // (the V^T gather of the second product is issued here: its latency runs under the first product and the softmax)
        float vv[4][8];
        float mv = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int k0 = 32 * i + 16 * s + 4 * h;                          // keys k0 .. k0 + 3 and k0 + 8 .. k0 + 11
                const int4 ta = *(const int4 *)(Ts + min(k0, 60)), tb = *(const int4 *)(Ts + min(k0 + 8, 60));
                const int tk[8] = {ta.x, ta.y, ta.z, ta.w, tb.x, tb.y, tb.z, tb.w};
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int key = k0 + 8 * (t >> 2) + (t & 3);
                    const bool valid = key < NT;
                    const float *pv = (valid && tk[t] >= 0) ? a.qkv + (size_t)tk[t] * C3 + 2 * a.C + hd * HD : bv;
                    const float x = valid ? pv[l32] : 0.f;
                    vv[2 * i + s][t] = x;
                    mv = fmaxf(mv, fabsf(x));
                }
            }
        const float sq = pow2_scale_w(wave_max(mq)), sk = pow2_scale_w(wave_max(mk));
        h8 qh[2][2], ql[2][2], kh[2][2], kl[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                split8(qv[i][ks], sq, qh[i][ks], ql[i][ks]);
                split8(kv[i][ks], sk, kh[i][ks], kl[i][ks]);
            }
        split_to_mfma_fence();
        // ---- S^T[key tile i][query tile j]
        f32x16 st[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    st[i][j][r] = 0.f;
            }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    st[i][j] = WMFMA(kh[i][ks], ql[j][ks], st[i][j]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    st[i][j] = WMFMA(kl[i][ks], qh[j][ks], st[i][j]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    st[i][j] = WMFMA(kh[i][ks], qh[j][ks], st[i][j]);
        }
        // ---- scale, bias, shift mask; keys >= 49 drop out; softmax over the keys of each query column
        const float c = a.scale / (sq * sk);
        float mx[2], sum[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = min(32 * j + l32, NT - 1);
            const int ridq = Rs[q];
            float m = -INFINITY;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int k0 = 32 * i + 8 * g + 4 * h;                       // keys k0 .. k0 + 3 <-> r = 4 g .. 4 g + 3
                    if (i == 1 && g == 3)
                        continue;                                               // keys >= 56: none valid
                    const f32x4 bb = *(const f32x4 *)(Bs + q * BST + min(k0, BST - 4));
                    int4 rr = *(const int4 *)(Rs + min(k0, 60));
                    const float bvs[4] = {bb.x, bb.y, bb.z, bb.w};
                    const int rv[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float s = st[i][j][4 * g + e] * c + bvs[e];
                        if (a.shift > 0)
                            s += rv[e] != ridq ? -100.f : 0.f;
                        s = (k0 + e < NT) ? s : -INFINITY;
                        st[i][j][4 * g + e] = s;
                        m = fmaxf(m, s);
                    }
                }
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            float l = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (i == 1 && r >= 12) {
                        st[i][j][r] = 0.f;
                        continue;
                    }
                    const float p = __expf(st[i][j][r] - m);
                    st[i][j][r] = p;
                    l += p;
                }
            l += __shfl_xor(l, 32, 64);
            mx[j] = m;
            sum[j] = l;
        }
        // ---- O^T = V^T P^T: four k-steps of 16 keys (tile i, step s); the last one holds key 48 only
        f32x16 ot[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                ot[j][r] = 0.f;
        const float sv = pow2_scale_w(wave_max(mv));
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                h8 vh, vl;
                split8(vv[2 * i + s], sv, vh, vl);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float pp[8];
#pragma unroll
                    for (int t = 0; t < 8; ++t)
                        pp[t] = st[i][j][8 * s + t];
                    h8 ph, pl;
                    split8(pp, 1.0f, ph, pl);
                    split_to_mfma_fence();
                    ot[j] = WMFMA(vh, pl, ot[j]);
                    ot[j] = WMFMA(vl, ph, ot[j]);
                    ot[j] = WMFMA(vh, ph, ot[j]);
                }
            }
        // ---- outputs: column = query 32 j + l32, rows = channels 8 g + 4 h + 0..3
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = 32 * j + l32;
            if (q < NT) {
                if (a.lse && h == 0)
                    a.lse[(((size_t)b * a.nW + win) * a.heads + hd) * NT + q] = mx[j] + logf(sum[j]);
                if (tok[j] >= 0) {
                    const float inv = 1.0f / (sum[j] * sv);
                    float *op = a.out + (size_t)tok[j] * a.C + hd * HD + 4 * h;
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *(f32x4 *)(op + 8 * g) = f32x4{ot[j][4 * g] * inv, ot[j][4 * g + 1] * inv, ot[j][4 * g + 2] * inv,
                                                       ot[j][4 * g + 3] * inv};
                }
            }
        }
    }
}

// ---- backward ------------------------------------------------------------------------------------------------------
// Per window-head: recompute the scores in BOTH orientations (S^T = K Q^T: keys along the registers, queries along the lanes;
// S = Q K^T: the other way round) together with dP^T = V dO^T and dP = dO V^T -- 96 MFMAs instead of a transposition of two
// 49 x 49 matrices through LDS -- because an accumulator tile can only serve as the B operand of a product that contracts
// over its REGISTER index:  dQ^T = K^T dS^T  (contracts over keys: orientation T),
//                           dK^T = Q^T dS and dV^T = dO^T P  (contract over queries: orientation N).
// delta[q] = sum_k P dP is an in-lane reduction in orientation T and reaches orientation N through a 64-float LDS table, as
// do the log-sum-exp values.  The transposed A operands (K^T, dO^T, Q^T: lanes along the 32 channels, contraction index in
// the kappa order of the accumulator registers) are read from a [token][36] LDS tile the fragment loads fill.  The bias
// gradient accumulates in LDS across the windows of the wave (its head is fixed) and is written once.
struct WbArgs {
    const float *qkv, *qkv_bias, *bias, *lse, *dout;
    float *dqkv, *dpad, *dbias_part, *dqkv_amax;
    int B, H, W, Hp, Wp, shift, heads, C;
    int nWx, nW, npad;
    float scale;
    int nwaves;
};

constexpr int XST = 36;                 // row stride of the [token][channel] tile (conflict-free b128 stores)
constexpr int NONE = -0x40000000;       // "no token" in the token table

__global__ __launch_bounds__(64) void k_winattn_bwd_mfma(WbArgs a)
{
    __shared__ __attribute__((aligned(16))) float Bs[NT * BST];       // bias [q][k]
    __shared__ __attribute__((aligned(16))) float Bt[NT * BST];       // bias [k][q]
    __shared__ __attribute__((aligned(16))) float Db[NT * BST];       // bias-gradient accumulator [q][k]
    __shared__ __attribute__((aligned(16))) float Xs[64 * XST];       // transposition tile [token][channel]
    __shared__ __attribute__((aligned(16))) int Ts[64], Rs[64];       // token row (>= 0 real, < 0: -1 - pad index, NONE)
    __shared__ __attribute__((aligned(16))) float Ls[64], Ds[64];     // lse[q], delta[q]
    const int lane = threadIdx.x;
    const int gw = blockIdx.x;
    const int hd = gw % a.heads;
    for (int idx = lane; idx < NT * NT; idx += 64) {
        const int q = idx / NT, k = idx - q * NT;
        const float v = a.bias[(size_t)hd * NT * NT + idx];
        Bs[q * BST + k] = v;
        Bt[k * BST + q] = v;
    }
    for (int idx = lane; idx < NT * BST; idx += 64)
        Db[idx] = 0.f;
    const int h = lane >> 5, l32 = lane & 31;
    const long long nbw = (long long)a.B * a.nW;
    const int stride = a.nwaves / a.heads;
    const size_t C3 = (size_t)3 * a.C;
    const int L = a.H * a.W;
    float gmax = 0.f;

    for (long long bw = gw / a.heads; bw < nbw; bw += stride) {
        const int win = (int)(bw % a.nW), b = (int)(bw / a.nW);
        const int wy = win / a.nWx, wx = win - wy * a.nWx;
        wave_lds_sync();
        {
            // token lane of the window: real row, or the index of the padded token (order of dcl_winattn.hip's token_of)
            const int t = lane < NT ? lane : NT - 1;
            const int r = t / WS, c = t - r * WS;
            const int y = wy * WS + r, x = wx * WS + c;
            int ys = y + a.shift, xs = x + a.shift;
            ys -= ys >= a.Hp ? a.Hp : 0;
            xs -= xs >= a.Wp ? a.Wp : 0;
            const bool real = ys < a.H && xs < a.W;
            int row;
            if (real)
                row = b * L + ys * a.W + xs;
            else if (ys < a.H)
                row = -1 - (ys * (a.Wp - a.W) + (xs - a.W));
            else
                row = -1 - (a.H * (a.Wp - a.W) + (ys - a.H) * a.Wp + xs);
            const int ry = y < a.Hp - WS ? 0 : (y < a.Hp - a.shift ? 1 : 2);
            const int rx = x < a.Wp - WS ? 0 : (x < a.Wp - a.shift ? 1 : 2);
            Ts[lane] = lane < NT ? row : NONE;
            Rs[lane] = ry * 3 + rx;
            Ls[lane] = lane < NT ? a.lse[(((size_t)b * a.nW + win) * a.heads + hd) * NT + lane] : 0.f;
        }
        wave_lds_sync();
        const int tok[2] = {Ts[l32], Ts[32 + l32]};
        // ---- fragments of Q, K, V (qkv rows, or the projection's bias for padded tokens) and dO (zero unless real)
        h8 fh[4][2][2], fl[4][2][2];            // [q k v do][tile][k-step]
        float fs[4];
        {
            float raw[4][2][2][8];
            float mxv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bool real = tok[i] >= 0, none = tok[i] == NONE;
                const float *pq = real ? a.qkv + (size_t)tok[i] * C3 + hd * HD : a.qkv_bias + hd * HD;
                const float *pd = a.dout + (size_t)(real ? tok[i] : 0) * a.C + hd * HD;
                const float zq = none ? 0.f : 1.f, zd = real ? 1.f : 0.f;
#pragma unroll
                for (int X = 0; X < 4; ++X) {
                    const float *p = X < 3 ? pq + X * a.C : pd;
                    const float z = X < 3 ? zq : zd;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const f32x4 v0 = *(const f32x4 *)(p + 8 * h + 16 * ks), v1 = *(const f32x4 *)(p + 8 * h + 16 * ks + 4);
                        const float e[8] = {v0.x * z, v0.y * z, v0.z * z, v0.w * z, v1.x * z, v1.y * z, v1.z * z, v1.w * z};
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            raw[X][i][ks][u] = e[u];
                            mxv[X] = fmaxf(mxv[X], fabsf(e[u]));
                        }
                    }
                }
            }
#pragma unroll
            for (int X = 0; X < 4; ++X) {
                fs[X] = pow2_scale_w(wave_max(mxv[X]));
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
                        split8(raw[X][i][ks], fs[X], fh[X][i][ks], fl[X][i][ks]);
            }
            // K rows into the transposition tile (for K^T in phase A)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    float *d = Xs + (32 * i + l32) * XST + 8 * h + 16 * ks;
                    *(f32x4 *)d = f32x4{raw[1][i][ks][0], raw[1][i][ks][1], raw[1][i][ks][2], raw[1][i][ks][3]};
                    *(f32x4 *)(d + 4) = f32x4{raw[1][i][ks][4], raw[1][i][ks][5], raw[1][i][ks][6], raw[1][i][ks][7]};
                }
        }
        split_to_mfma_fence();
        // product of two fragment sets: out[i][j] = sum_d A[i] (rows) . B[j] (columns)
        auto prod = [&](int A, int Bx, f32x16 (&o)[2][2]) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        o[i][j][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        o[i][j] = WMFMA(fh[A][i][ks], fl[Bx][j][ks], o[i][j]);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        o[i][j] = WMFMA(fl[A][i][ks], fh[Bx][j][ks], o[i][j]);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        o[i][j] = WMFMA(fh[A][i][ks], fh[Bx][j][ks], o[i][j]);
            }
        };
        // out^T[channel][column tile j] += X^T (from the LDS tile, kappa order) . M (accumulator tiles [i][j] as B operands)
        auto apply = [&](const f32x16 (&m)[2][2], float ms, f32x16 (&o)[2]) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    o[j][r] = 0.f;
            float xv[4][8];
            float mx = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const int row = 32 * i + 16 * s + 8 * (t >> 2) + 4 * h + (t & 3);
                        const float v = Xs[row * XST + l32];
                        xv[2 * i + s][t] = v;
                        mx = fmaxf(mx, fabsf(v));
                    }
            const float xs = pow2_scale_w(wave_max(mx));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    h8 xh, xl;
                    split8(xv[2 * i + s], xs, xh, xl);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        float pp[8];
#pragma unroll
                        for (int t = 0; t < 8; ++t)
                            pp[t] = m[i][j][8 * s + t];
                        h8 ph, pl;
                        split8(pp, ms, ph, pl);
                        split_to_mfma_fence();
                        o[j] = WMFMA(xh, pl, o[j]);
                        o[j] = WMFMA(xl, ph, o[j]);
                        o[j] = WMFMA(xh, ph, o[j]);
                    }
                }
            return xs;
        };
        // rows of a [token][channel] tensor (this lane's fragments) into the transposition tile
        auto to_tile = [&](int part) {
            wave_lds_sync();                              // earlier reads of the tile are done
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bool real = tok[i] >= 0, none = tok[i] == NONE;
                const float *p = part == 3 ? a.dout + (size_t)(real ? tok[i] : 0) * a.C + hd * HD
                                           : (real ? a.qkv + (size_t)tok[i] * C3 + part * a.C + hd * HD
                                                   : a.qkv_bias + part * a.C + hd * HD);
                const float z = part == 3 ? (real ? 1.f : 0.f) : (none ? 0.f : 1.f);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    f32x4 v0 = *(const f32x4 *)(p + 8 * h + 16 * ks), v1 = *(const f32x4 *)(p + 8 * h + 16 * ks + 4);
                    v0.x *= z; v0.y *= z; v0.z *= z; v0.w *= z; v1.x *= z; v1.y *= z; v1.z *= z; v1.w *= z;
                    float *d = Xs + (32 * i + l32) * XST + 8 * h + 16 * ks;
                    *(f32x4 *)d = v0;
                    *(f32x4 *)(d + 4) = v1;
                }
            }
            wave_lds_sync();
        };
        // column tile j of out^T[channel][token] -> part `part` of the tokens' dqkv rows (or dpad rows)
        auto store_out = [&](const f32x16 (&o)[2], float mul, int part) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (tok[j] == NONE)
                    continue;
                float *op = (tok[j] >= 0 ? a.dqkv + (size_t)tok[j] * C3
                                         : a.dpad + ((size_t)b * a.npad + (-1 - tok[j])) * C3) + part * a.C + hd * HD + 4 * h;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = f32x4{o[j][4 * g] * mul, o[j][4 * g + 1] * mul, o[j][4 * g + 2] * mul, o[j][4 * g + 3] * mul};
                    *(f32x4 *)(op + 8 * g) = v;
                    gmax = fmaxf(fmaxf(gmax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
                }
            }
        };

        // ================= phase A: keys along the registers, queries along the lanes =================
        f32x16 m1[2][2], m2[2][2];
        prod(1, 0, m1);                                   // S^T (unscaled) = K Q^T
        prod(2, 3, m2);                                   // dP^T = V dO^T
        const float cS = a.scale / (fs[0] * fs[1]), cP = 1.0f / (fs[2] * fs[3]);
        float dsmax = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = min(32 * j + l32, NT - 1);
            const bool qok = 32 * j + l32 < NT;
            const int ridq = Rs[q];
            const float lq = Ls[q];
            float delta = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int k0 = 32 * i + 8 * g + 4 * h;
                    const f32x4 bb = *(const f32x4 *)(Bs + q * BST + min(k0, BST - 4));
                    const int4 rr = *(const int4 *)(Rs + min(k0, 60));
                    const float bvs[4] = {bb.x, bb.y, bb.z, bb.w};
                    const int rv[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float s = m1[i][j][4 * g + e] * cS + bvs[e];
                        if (a.shift > 0)
                            s += rv[e] != ridq ? -100.f : 0.f;
                        const float p = (k0 + e < NT && qok) ? __expf(s - lq) : 0.f;
                        const float dp = m2[i][j][4 * g + e] * cP;
                        m1[i][j][4 * g + e] = p;
                        m2[i][j][4 * g + e] = dp;
                        delta += p * dp;
                    }
                }
            delta += __shfl_xor(delta, 32, 64);
            if (h == 0)
                Ds[32 * j + l32] = delta;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int k0 = 32 * i + 8 * g + 4 * h;
                    float ds[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        ds[e] = m1[i][j][4 * g + e] * (m2[i][j][4 * g + e] - delta);
                        m1[i][j][4 * g + e] = ds[e];
                        dsmax = fmaxf(dsmax, fabsf(ds[e]));
                    }
                    if (qok && k0 < NT) {                 // bias gradient: [q][k0 .. k0 + 3] (padding columns stay 0)
                        f32x4 *dbp = (f32x4 *)(Db + q * BST + k0);
                        f32x4 v = *dbp;
                        v.x += ds[0]; v.y += ds[1]; v.z += ds[2]; v.w += ds[3];
                        *dbp = v;
                    }
                }
        }
        const float sds = pow2_scale_w(wave_max(dsmax));
        wave_lds_sync();                                  // K rows are in the tile, delta table written
        {
            f32x16 o[2];
            const float xs = apply(m1, sds, o);           // dQ^T = K^T dS^T
            store_out(o, a.scale / (xs * sds), 0);
        }
        // ================= phase B: queries along the registers, keys along the lanes =================
        prod(0, 1, m1);                                   // S = Q K^T
        prod(3, 2, m2);                                   // dP = dO V^T
        dsmax = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {                     // column tile j: key 32 j + l32
            const int k = min(32 * j + l32, NT - 1);
            const bool kok = 32 * j + l32 < NT;
            const int ridk = Rs[k];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int q0 = 32 * i + 8 * g + 4 * h;
                    const f32x4 bb = *(const f32x4 *)(Bt + k * BST + min(q0, BST - 4));
                    const int4 rr = *(const int4 *)(Rs + min(q0, 60));
                    const f32x4 ll = *(const f32x4 *)(Ls + min(q0, 60)), dd = *(const f32x4 *)(Ds + min(q0, 60));
                    const float bvs[4] = {bb.x, bb.y, bb.z, bb.w}, lv[4] = {ll.x, ll.y, ll.z, ll.w}, dv[4] = {dd.x, dd.y, dd.z, dd.w};
                    const int rv[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float s = m1[i][j][4 * g + e] * cS + bvs[e];
                        if (a.shift > 0)
                            s += rv[e] != ridk ? -100.f : 0.f;
                        const float p = (q0 + e < NT && kok) ? __expf(s - lv[e]) : 0.f;
                        const float ds = p * (m2[i][j][4 * g + e] * cP - dv[e]);
                        m1[i][j][4 * g + e] = p;
                        m2[i][j][4 * g + e] = ds;
                        dsmax = fmaxf(dsmax, fabsf(ds));
                    }
                }
        }
        const float sds2 = pow2_scale_w(wave_max(dsmax));
        {
            f32x16 o[2];
            to_tile(3);                                   // dO rows
            const float xs = apply(m1, 1.0f, o);          // dV^T = dO^T P
            store_out(o, 1.0f / xs, 2);
            to_tile(0);                                   // Q rows
            const float xq = apply(m2, sds2, o);          // dK^T = Q^T dS
            store_out(o, a.scale / (xq * sds2), 1);
        }
    }
    // ---- bias gradient of this wave, absmax of everything it wrote
    wave_lds_sync();
    {
        float *dst = a.dbias_part + (size_t)gw * NT * NT;
        for (int idx = lane; idx < NT * NT; idx += 64) {
            const int q = idx / NT, k = idx - q * NT;
            dst[idx] = Db[q * BST + k];
        }
    }
    if (a.dqkv_amax) {
        gmax = wave_max(gmax);
        if (lane == 0)
            atomicMax((int *)a.dqkv_amax + (blockIdx.x & (DCL_AMAX_SLOTS - 1)), __float_as_int(gmax));
    }
}

}  // namespace

// host entry: same arguments as dcl_winattn_fwd (declared in dcl_winattn.hip's dispatcher)
int dcl_winattn_fwd_mfma_launch(const float *qkv, const float *qkv_bias, const float *bias, int B, int H, int W, int C,
                                int heads, int shift, float scale, float *out, float *lse, int nwaves, hipStream_t stream)
{
    WmArgs a = {};
    a.qkv = qkv; a.qkv_bias = qkv_bias; a.bias = bias; a.out = out; a.lse = lse;
    a.B = B; a.H = H; a.W = W; a.C = C; a.heads = heads; a.shift = shift; a.scale = scale;
    a.Hp = (H + WS - 1) / WS * WS;
    a.Wp = (W + WS - 1) / WS * WS;
    a.nWx = a.Wp / WS;
    a.nW = (a.Hp / WS) * a.nWx;
    a.nwaves = nwaves;
    hipLaunchKernelGGL(k_winattn_fwd_mfma, dim3((unsigned)nwaves), dim3(64), 0, stream, a);
    return 0;
}

int dcl_winattn_bwd_mfma_launch(const float *qkv, const float *qkv_bias, const float *bias, const float *lse,
                                const float *dout, int B, int H, int W, int C, int heads, int shift, float scale, float *dqkv,
                                float *dpad, float *dbias_part, float *dqkv_amax, int nwaves, hipStream_t stream)
{
    WbArgs a = {};
    a.qkv = qkv; a.qkv_bias = qkv_bias; a.bias = bias; a.lse = lse; a.dout = dout;
    a.dqkv = dqkv; a.dpad = dpad; a.dbias_part = dbias_part; a.dqkv_amax = dqkv_amax;
    a.B = B; a.H = H; a.W = W; a.C = C; a.heads = heads; a.shift = shift; a.scale = scale;
    a.Hp = (H + WS - 1) / WS * WS;
    a.Wp = (W + WS - 1) / WS * WS;
    a.nWx = a.Wp / WS;
    a.nW = (a.Hp / WS) * a.nWx;
    a.npad = a.Hp * a.Wp - H * W;
    a.nwaves = nwaves;
    hipLaunchKernelGGL(k_winattn_bwd_mfma, dim3((unsigned)nwaves), dim3(64), 0, stream, a);
    return 0;
}
