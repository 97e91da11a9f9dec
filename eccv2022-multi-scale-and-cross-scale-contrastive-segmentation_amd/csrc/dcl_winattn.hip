// dcl_winattn.hip -- Swin window attention, forward and backward, fp32.
//
// Replaces WindowAttention.forward (reference models/Swin.py:198-230: q * scale, q k^T, + relative-position bias,
// + shifted-window mask, softmax, attn v) TOGETHER WITH the data movement SwinTransformerBlock.forward wraps around it
// (:286-318: zero-pad to a multiple of the window, cyclic roll by -shift, window_partition, and after the attention
// window_reverse, roll back, crop): the qkv projection is token-wise, so it commutes with all of those permutations
// and is applied to the tokens in their natural [B, H*W, 3C] order; this kernel then gathers the 49 tokens of a window
// straight from that tensor (source pixel = ((7 wy + r + shift) mod Hp, (7 wx + c + shift) mod Wp)) and writes the
// result back to the same natural positions.  Four full copies of the activation per block disappear, and so does the
// [B_, heads, 49, 49] bias + mask tensor the SDPA call needed.
//   * tokens in the zero-padded border carry qkv = the projection's BIAS in the reference (Linear(0) = b): the kernel
//     reads the bias vector for them, and their gradient goes to `dqkv` rows past H*W (the host adds them to the bias
//     gradient);
//   * the shifted-window mask (-100 between different wrap-around regions, Swin.py:448-466) is computed from the token
//     coordinates, never materialised.
// Work decomposition: one wave per (image, window, head); lane i owns query row i (49 of 64 lanes active): the score
// row, the softmax and the output row are in-lane (no cross-lane reductions); K and V rows are broadcast reads from the
// wave's private LDS region.  head_dim = 32 and window 7 x 7 (every Swin variant the reference ships).  The problem is
// memory- and latency-bound (25 KB in / out per 0.3 MFLOP); plain fp32 FMAs -- the reference's own arithmetic.
// Backward: recompute S and P from the saved log-sum-exp, dP = dO V^T, dS = P (dP - sum_j P dP), dQ = dS K (rows,
// in-lane); dK = dS^T Q and dV = P^T dO need the transposes: dS and P go through LDS so that lane j owns key row j.
// The bias gradient (sum of dS over windows) accumulates in registers across the windows a wave processes -- waves are
// assigned a fixed head -- and is written as one partial per wave (summed by the host in fixed order: deterministic).
#include "dcl_common.h"

namespace {

constexpr int WS = 7, NT = 49, HD = 32;
constexpr int TS = 49;                 // row stride of the transposed 49 x 49 matrices in LDS (odd: lane-strided reads
                                       // of pass 3 hit distinct banks)

struct WaArgs {
    const float *qkv;       // [B, H*W, 3*C]
    const float *qkv_bias;  // [3*C] or null (zeros): qkv of the padded tokens
    const float *bias;      // [heads, 49, 49] dense relative-position bias
    float *out;             // fwd: [B, H*W, C]
    float *lse;             // [B, nW, heads, 49]
    // backward
    const float *dout;      // [B, H*W, C]
    float *dqkv;            // [B, H*W, 3*C]
    float *dqkv_amax;       // optional: DCL_AMAX_SLOTS partial maxima of |dqkv| (integer atomic max), or null
    float *dpad;            // [B, npad, 3*C]  gradient of the padded tokens' qkv (= of the projection's bias)
    float *dbias_part;      // [nwaves, 49, 49]  partial bias gradients, wave w holds head w % heads
    int B, H, W, Hp, Wp, shift, heads, C;
    int nWx, nW, npad;
    float scale;
    int nwaves;
};

// token t of window (wy, wx) -> row of the [L + npad] token axis (natural order; >= L for padded tokens) and its
// shifted-window region id
__device__ __forceinline__ void token_of(const WaArgs &a, int wy, int wx, int t, int &row, bool &real, int &rid)
{
    const int r = t / WS, c = t - r * WS;
    const int y = wy * WS + r, x = wx * WS + c;                    // coordinates in the rolled, padded frame
    int ys = y + a.shift, xs = x + a.shift;
    ys -= ys >= a.Hp ? a.Hp : 0;
    xs -= xs >= a.Wp ? a.Wp : 0;
    real = ys < a.H && xs < a.W;
    if (real)
        row = ys * a.W + xs;
    else if (ys < a.H)
        row = a.H * a.W + ys * (a.Wp - a.W) + (xs - a.W);
    else
        row = a.H * a.W + a.H * (a.Wp - a.W) + (ys - a.H) * a.Wp + xs;
    const int ry = y < a.Hp - WS ? 0 : (y < a.Hp - a.shift ? 1 : 2);
    const int rx = x < a.Wp - WS ? 0 : (x < a.Wp - a.shift ? 1 : 2);
    rid = ry * 3 + rx;
}

// pointer to the `part` (0 = q, 1 = k, 2 = v) vector of head `hd` of a token: its row of qkv, or -- zero-padded border
// -- the projection's bias vector (always global memory: no generic / flat loads)
__device__ __forceinline__ const float *qkv_ptr(const WaArgs &a, int b, int row, bool real, int part, int hd)
{
    const int off = part * a.C + hd * HD;
    if (real)
        return a.qkv + ((size_t)b * a.H * a.W + row) * (3 * a.C) + off;
    return a.qkv_bias + off;
}

// rows of K / V (or Q / dO) of the window into LDS [49][32]: 392 float4 pieces over 64 lanes; ptr_of(t, keep) returns a
// valid global pointer and whether the row is kept (else zeros are stored)
template <typename PtrOf>
__device__ __forceinline__ void load_rows(float *dst, int lane, PtrOf ptr_of)
{
    for (int idx = lane; idx < NT * 8; idx += 64) {
        const int t = idx >> 3, seg = idx & 7;
        bool keep;
        const float *p = ptr_of(t, keep);
        f32x4 v = *(const f32x4 *)(p + seg * 4);
        if (!keep)
            v = f32x4{0.f, 0.f, 0.f, 0.f};
        *(f32x4 *)(dst + t * HD + seg * 4) = v;
    }
}

__global__ __launch_bounds__(256) void k_winattn_fwd(WaArgs a)
{
    __shared__ __attribute__((aligned(16))) float lds[4][2][NT * HD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long total = (long long)a.B * a.nW * a.heads;
    const long long prob = (long long)blockIdx.x * 4 + wave;
    if (prob >= total)
        return;
    const int hd = (int)(prob % a.heads);
    const int win = (int)((prob / a.heads) % a.nW);
    const int b = (int)(prob / ((long long)a.heads * a.nW));
    const int wy = win / a.nWx, wx = win - wy * a.nWx;
    float *Ks = lds[wave][0], *Vs = lds[wave][1];

    const int i = lane < NT ? lane : NT - 1;              // surplus lanes shadow the last row, nothing is stored
    int row_i, rid_i;
    bool real_i;
    token_of(a, wy, wx, i, row_i, real_i, rid_i);
    float q[HD];
    {
        const float *qp = qkv_ptr(a, b, row_i, real_i, 0, hd);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const f32x4 v = *(const f32x4 *)(qp + 4 * s);
            q[4 * s + 0] = v.x * a.scale;
            q[4 * s + 1] = v.y * a.scale;
            q[4 * s + 2] = v.z * a.scale;
            q[4 * s + 3] = v.w * a.scale;
        }
    }
    auto kv_of = [&](int part) {
        return [&, part](int t, bool &keep) {
            int row, rid;
            bool real;
            token_of(a, wy, wx, t, row, real, rid);
            keep = true;
            return qkv_ptr(a, b, row, real, part, hd);
        };
    };
    load_rows(Ks, lane, kv_of(1));
    load_rows(Vs, lane, kv_of(2));
    // Scores in blocks of 7 keys with an online softmax (running max m and sum l, accumulator rescaled when m grows):
    // a fully unrolled 49-key loop makes the compiler hoist all 392 broadcast reads into registers and spill.
    const float *brow = a.bias + ((size_t)hd * NT + i) * NT;
    __builtin_amdgcn_s_waitcnt(0xc07f);                   // lgkmcnt(0): this wave's LDS writes have landed
    __builtin_amdgcn_wave_barrier();
    float m = -INFINITY, l = 0.f;
    float o[HD];
#pragma unroll
    for (int k = 0; k < HD; ++k)
        o[k] = 0.f;
#pragma unroll 1
    for (int jb = 0; jb < WS; ++jb) {
        float s7[WS];
#pragma unroll
        for (int jj = 0; jj < WS; ++jj) {
            const int j = jb * WS + jj;
            float acc = 0.f;
#pragma unroll
            for (int k4 = 0; k4 < 8; ++k4) {
                const f32x4 kv = *(const f32x4 *)(Ks + j * HD + 4 * k4);          // broadcast read
                acc += q[4 * k4 + 0] * kv.x;
                acc += q[4 * k4 + 1] * kv.y;
                acc += q[4 * k4 + 2] * kv.z;
                acc += q[4 * k4 + 3] * kv.w;
            }
            acc += brow[j];
            if (a.shift > 0) {
                int row, rid;
                bool real;
                token_of(a, wy, wx, j, row, real, rid);
                acc += rid != rid_i ? -100.f : 0.f;
            }
            s7[jj] = acc;
        }
        float mb = s7[0];
#pragma unroll
        for (int jj = 1; jj < WS; ++jj)
            mb = fmaxf(mb, s7[jj]);
        const float mn = fmaxf(m, mb);
        const float corr = expf(m - mn);                  // 0 on the first block (m = -inf)
        l *= corr;
#pragma unroll
        for (int k = 0; k < HD; ++k)
            o[k] *= corr;
#pragma unroll
        for (int jj = 0; jj < WS; ++jj) {
            const int j = jb * WS + jj;
            const float pj = expf(s7[jj] - mn);
            l += pj;
#pragma unroll
            for (int k4 = 0; k4 < 8; ++k4) {
                const f32x4 vv = *(const f32x4 *)(Vs + j * HD + 4 * k4);
                o[4 * k4 + 0] += pj * vv.x;
                o[4 * k4 + 1] += pj * vv.y;
                o[4 * k4 + 2] += pj * vv.z;
                o[4 * k4 + 3] += pj * vv.w;
            }
        }
        m = mn;
    }
    const float inv = 1.0f / l;
    if (lane < NT) {
        if (a.lse)
            a.lse[(((size_t)b * a.nW + win) * a.heads + hd) * NT + lane] = m + logf(l);
        if (real_i) {
            float *op = a.out + ((size_t)b * a.H * a.W + row_i) * a.C + hd * HD;
#pragma unroll
            for (int s4 = 0; s4 < 8; ++s4)
                *(f32x4 *)(op + 4 * s4) = f32x4{o[4 * s4] * inv, o[4 * s4 + 1] * inv, o[4 * s4 + 2] * inv, o[4 * s4 + 3] * inv};
        }
    }
}

// one wave per problem as in the forward, one wave per workgroup; LDS: K, Q (scaled) and V as [49][32] -- dO takes V's
// place once pass 1 is through with it -- and two transposed 49 x 49 matrices [j][i]: P^T and dP^T -> dS^T. 38 KB, so
// four workgroups share a CU (one wave per SIMD: the kernel is VALU-bound and holds ~400 VGPRs). The running
// bias-gradient row of query i lives in lane i's registers.
constexpr int BWD_LDS = 3 * NT * HD + 2 * NT * TS;

__global__ __launch_bounds__(64) void k_winattn_bwd(WaArgs a)
{
    __shared__ __attribute__((aligned(16))) float lds[BWD_LDS];
    const int lane = threadIdx.x;
    const int gw = blockIdx.x;                                // global wave id; its head is fixed: gw % heads
    const int hd = gw % a.heads;
    float *Ks = lds, *Vs = Ks + NT * HD, *Qs = Vs + NT * HD, *Gs = Vs;
    float *PT = Qs + NT * HD, *DT = PT + NT * TS;
    const int i = lane < NT ? lane : NT - 1;
    const bool act = lane < NT;
    float db[WS][WS];
    float gmax = 0.f;                                         // max |dq|, |dk|, |dv| this lane wrote
#pragma unroll
    for (int j = 0; j < NT; ++j)
        db[j / WS][j % WS] = 0.f;
    const long long nbw = (long long)a.B * a.nW;              // (image, window) pairs, strided over the waves of a head
    const int stride = a.nwaves / a.heads;
    for (long long bw = gw / a.heads; bw < nbw; bw += stride) {
        const int win = (int)(bw % a.nW), b = (int)(bw / a.nW);
        const int wy = win / a.nWx, wx = win - wy * a.nWx;
        int row_i, rid_i;
        bool real_i;
        token_of(a, wy, wx, i, row_i, real_i, rid_i);
        auto tok_ptr = [&](int part) {
            return [&, part](int t, bool &keep) {
                int row, rid;
                bool real;
                token_of(a, wy, wx, t, row, real, rid);
                keep = true;
                return qkv_ptr(a, b, row, real, part, hd);
            };
        };
        // dO of a padded token is zero (its output is cropped away): clamped pointer + keep flag
        auto dout_ptr = [&](int t, bool &keep) {
            int row, rid;
            bool real;
            token_of(a, wy, wx, t, row, real, rid);
            keep = real;
            return a.dout + ((size_t)b * a.H * a.W + (real ? row : 0)) * a.C + hd * HD;
        };
        __builtin_amdgcn_wave_barrier();                      // previous iteration's LDS reads are done
        load_rows(Ks, lane, tok_ptr(1));
        load_rows(Vs, lane, tok_ptr(2));
        float q[HD], g[HD];
        {
            const float *qp = qkv_ptr(a, b, row_i, real_i, 0, hd);
            bool keep_g;
            const float *gp = dout_ptr(i, keep_g);
            const float gm = keep_g ? 1.f : 0.f;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const f32x4 v = *(const f32x4 *)(qp + 4 * s);
                f32x4 w = *(const f32x4 *)(gp + 4 * s);
                w.x *= gm; w.y *= gm; w.z *= gm; w.w *= gm;
                q[4 * s + 0] = v.x * a.scale; q[4 * s + 1] = v.y * a.scale;
                q[4 * s + 2] = v.z * a.scale; q[4 * s + 3] = v.w * a.scale;
                g[4 * s + 0] = w.x; g[4 * s + 1] = w.y; g[4 * s + 2] = w.z; g[4 * s + 3] = w.w;
            }
        }
        if (act) {                                           // scaled Q rows for dK
#pragma unroll
            for (int s4 = 0; s4 < 8; ++s4)
                *(f32x4 *)(Qs + lane * HD + 4 * s4) = f32x4{q[4 * s4], q[4 * s4 + 1], q[4 * s4 + 2], q[4 * s4 + 3]};
        }
        const float lse = a.lse[(((size_t)b * a.nW + win) * a.heads + hd) * NT + i];
        const float *brow = a.bias + ((size_t)hd * NT + i) * NT;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        // pass 1: P = exp(S - lse), dP = dO V^T, D = sum_j P dP; P and dP parked transposed in LDS
        float D = 0.f;
#pragma unroll 1
        for (int jb = 0; jb < WS; ++jb) {
#pragma unroll
            for (int jj = 0; jj < WS; ++jj) {
                const int j = jb * WS + jj;
                float acc = 0.f, acc2 = 0.f;
#pragma unroll
                for (int k4 = 0; k4 < 8; ++k4) {
                    const f32x4 kv = *(const f32x4 *)(Ks + j * HD + 4 * k4);
                    const f32x4 vv = *(const f32x4 *)(Vs + j * HD + 4 * k4);
                    acc += q[4 * k4 + 0] * kv.x; acc += q[4 * k4 + 1] * kv.y;
                    acc += q[4 * k4 + 2] * kv.z; acc += q[4 * k4 + 3] * kv.w;
                    acc2 += g[4 * k4 + 0] * vv.x; acc2 += g[4 * k4 + 1] * vv.y;
                    acc2 += g[4 * k4 + 2] * vv.z; acc2 += g[4 * k4 + 3] * vv.w;
                }
                acc += brow[j];
                if (a.shift > 0) {
                    int row, rid;
                    bool real;
                    token_of(a, wy, wx, j, row, real, rid);
                    acc += rid != rid_i ? -100.f : 0.f;
                }
                const float pj = expf(acc - lse);
                D += pj * acc2;
                if (act) {
                    PT[j * TS + lane] = pj;
                    DT[j * TS + lane] = acc2;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();                      // V is done with: dO rows take its place
        load_rows(Gs, lane, dout_ptr);
        // pass 2: dS = P (dP - D); dQ = scale * dS K; dS^T replaces dP^T. The bias gradient accumulates in registers:
        // the loop over key rows stays rolled, so the 7 x 7 accumulator rotates by one row per trip (49 moves against
        // ~1100 FMAs) and every index is a constant; seven trips bring it back into place
        float dq[HD];
#pragma unroll
        for (int k = 0; k < HD; ++k)
            dq[k] = 0.f;
#pragma unroll 1
        for (int jb = 0; jb < WS; ++jb) {
#pragma unroll
            for (int jj = 0; jj < WS; ++jj) {
                const int j = jb * WS + jj;
                const float ds = PT[j * TS + i] * (DT[j * TS + i] - D);
                if (act)
                    DT[j * TS + lane] = ds;
                db[0][jj] += ds;
#pragma unroll
                for (int k4 = 0; k4 < 8; ++k4) {
                    const f32x4 kv = *(const f32x4 *)(Ks + j * HD + 4 * k4);
                    dq[4 * k4 + 0] += ds * kv.x; dq[4 * k4 + 1] += ds * kv.y;
                    dq[4 * k4 + 2] += ds * kv.z; dq[4 * k4 + 3] += ds * kv.w;
                }
            }
#pragma unroll
            for (int jj = 0; jj < WS; ++jj) {
                const float t = db[0][jj];
#pragma unroll
                for (int r = 0; r + 1 < WS; ++r)
                    db[r][jj] = db[r + 1][jj];
                db[WS - 1][jj] = t;
            }
        }
        if (act) {
            float *dqp = (real_i ? a.dqkv + ((size_t)b * a.H * a.W + row_i) * (3 * a.C)
                                 : a.dpad + ((size_t)b * a.npad + (row_i - a.H * a.W)) * (3 * a.C)) + hd * HD;
#pragma unroll
            for (int s4 = 0; s4 < 8; ++s4) {
                const f32x4 o = f32x4{dq[4 * s4] * a.scale, dq[4 * s4 + 1] * a.scale, dq[4 * s4 + 2] * a.scale,
                                      dq[4 * s4 + 3] * a.scale};
                *(f32x4 *)(dqp + 4 * s4) = o;
                gmax = fmaxf(fmaxf(gmax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        // pass 3, lane j owns key row j: dK_j = sum_i dS^T[j][i] Qs[i] (Q carries the scale), dV_j = sum_i P^T[j][i] dO_i
        float dk[HD], dv[HD];
#pragma unroll
        for (int k = 0; k < HD; ++k)
            dk[k] = dv[k] = 0.f;
#pragma unroll 1
        for (int ib = 0; ib < WS; ++ib) {
#pragma unroll
            for (int i2 = 0; i2 < WS; ++i2) {
                const int ii = ib * WS + i2;
                const float dst = DT[i * TS + ii], pt = PT[i * TS + ii];
#pragma unroll
                for (int k4 = 0; k4 < 8; ++k4) {
                    const f32x4 qv = *(const f32x4 *)(Qs + ii * HD + 4 * k4);
                    const f32x4 gv = *(const f32x4 *)(Gs + ii * HD + 4 * k4);
                    dk[4 * k4 + 0] += dst * qv.x; dk[4 * k4 + 1] += dst * qv.y;
                    dk[4 * k4 + 2] += dst * qv.z; dk[4 * k4 + 3] += dst * qv.w;
                    dv[4 * k4 + 0] += pt * gv.x; dv[4 * k4 + 1] += pt * gv.y;
                    dv[4 * k4 + 2] += pt * gv.z; dv[4 * k4 + 3] += pt * gv.w;
                }
            }
        }
        if (act) {
            float *dkp = (real_i ? a.dqkv + ((size_t)b * a.H * a.W + row_i) * (3 * a.C)
                                 : a.dpad + ((size_t)b * a.npad + (row_i - a.H * a.W)) * (3 * a.C)) + a.C + hd * HD;
            float *dvp = dkp + a.C;
#pragma unroll
            for (int s4 = 0; s4 < 8; ++s4) {
                *(f32x4 *)(dkp + 4 * s4) = f32x4{dk[4 * s4], dk[4 * s4 + 1], dk[4 * s4 + 2], dk[4 * s4 + 3]};
                *(f32x4 *)(dvp + 4 * s4) = f32x4{dv[4 * s4], dv[4 * s4 + 1], dv[4 * s4 + 2], dv[4 * s4 + 3]};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    gmax = fmaxf(gmax, fmaxf(fabsf(dk[4 * s4 + e]), fabsf(dv[4 * s4 + e])));
            }
        }
    }
    if (a.dqkv_amax) {                                       // absmax side channel for the qkv Linear's backward GEMMs
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
            gmax = fmaxf(gmax, __shfl_xor(gmax, o, 64));
        if (lane == 0)
            atomicMax((int *)a.dqkv_amax + (blockIdx.x & (DCL_AMAX_SLOTS - 1)), __float_as_int(gmax));
    }
    if (act) {                                               // dbias_part[gw][i][:] = lane i's accumulator
        float *dst = a.dbias_part + ((size_t)gw * NT + lane) * NT;
#pragma unroll
        for (int j = 0; j < NT; ++j)
            dst[j] = db[j / WS][j % WS];
    }
}

int fill_args(WaArgs &a, int B, int H, int W, int C, int heads, int shift, float scale)
{
    if (B <= 0 || H <= 0 || W <= 0 || heads <= 0 || C != heads * HD || shift < 0 || shift >= WS)
        return DCL_EINVAL;
    a.B = B; a.H = H; a.W = W; a.C = C; a.heads = heads; a.shift = shift; a.scale = scale;
    a.Hp = (H + WS - 1) / WS * WS;
    a.Wp = (W + WS - 1) / WS * WS;
    a.nWx = a.Wp / WS;
    a.nW = (a.Hp / WS) * a.nWx;
    a.npad = a.Hp * a.Wp - H * W;
    return 0;
}

}  // namespace

extern "C" int dcl_winattn_npad(int H, int W)
{
    const int Hp = (H + WS - 1) / WS * WS, Wp = (W + WS - 1) / WS * WS;
    return Hp * Wp - H * W;
}

int dcl_winattn_fwd_mfma_launch(const float *qkv, const float *qkv_bias, const float *bias, int B, int H, int W, int C,
                                int heads, int shift, float scale, float *out, float *lse, int nwaves, hipStream_t stream);
extern "C" int dcl_winattn_bwd_waves(int B, int H, int W, int heads);
int dcl_winattn_bwd_mfma_launch(const float *qkv, const float *qkv_bias, const float *bias, const float *lse,
                                const float *dout, int B, int H, int W, int C, int heads, int shift, float scale, float *dqkv,
                                float *dpad, float *dbias_part, float *dqkv_amax, int nwaves, hipStream_t stream);
static int g_winattn_mfma = 3;      // bit 0: forward on the matrix cores, bit 1: backward (A/B: dcl_winattn_set_mfma)

extern "C" int dcl_winattn_set_mfma(int mask)
{
    DCL_CHECK_ARG(mask >= 0 && mask <= 3, "mask: bit 0 forward, bit 1 backward");
    g_winattn_mfma = mask;
    return 0;
}

extern "C" int dcl_winattn_fwd(const float *qkv, const float *qkv_bias, const float *bias, int B, int H, int W, int C,
                               int heads, int shift, float scale, float *out, float *lse, void *stream)
{
    DCL_CHECK_ARG(qkv && qkv_bias && bias && out, "null pointer (pass a zero vector for a projection without bias)");
    WaArgs a = {};
    DCL_CHECK_ARG(fill_args(a, B, H, W, C, heads, shift, scale) == 0,
                  "bad shape (window 7, head_dim 32: C must be 32 * heads; 0 <= shift < 7)");
    DCL_CHECK_ARG(((((uintptr_t)qkv) | ((uintptr_t)out) | ((uintptr_t)qkv_bias)) & 15) == 0, "16-byte alignment");
    a.qkv = qkv; a.qkv_bias = qkv_bias; a.bias = bias; a.out = out; a.lse = lse;
    if (g_winattn_mfma & 1) {       // matrix-core kernel (dcl_winattn_mfma.hip); token rows are 32-bit there
        DCL_CHECK_ARG((long long)B * H * W < (1LL << 31), "too many tokens");
        dcl_winattn_fwd_mfma_launch(qkv, qkv_bias, bias, B, H, W, C, heads, shift, scale, out, lse,
                                    dcl_winattn_bwd_waves(B, H, W, heads), (hipStream_t)stream);
        DCL_LAUNCH_CHECK();
        return 0;
    }
    const long long total = (long long)B * a.nW * heads;
    hipLaunchKernelGGL(k_winattn_fwd, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
    DCL_LAUNCH_CHECK();
    return 0;
}

extern "C" int dcl_winattn_bwd_waves(int B, int H, int W, int heads)
{
    if (B <= 0 || H <= 0 || W <= 0 || heads <= 0)
        return 0;
    const long long nbw = (long long)B * ((H + WS - 1) / WS) * ((W + WS - 1) / WS);
    long long per_head = 2048 / heads;                       // ~8 waves per CU
    if (per_head < 1)
        per_head = 1;
    if (per_head > nbw)
        per_head = nbw;
    return (int)(per_head * heads);
}

extern "C" int dcl_winattn_bwd(const float *qkv, const float *qkv_bias, const float *bias, const float *lse,
                               const float *dout, int B, int H, int W, int C, int heads, int shift, float scale,
                               float *dqkv, float *dpad, float *dbias_part, float *dqkv_amax, void *stream)
{
    DCL_CHECK_ARG(qkv && qkv_bias && bias && lse && dout && dqkv && dbias_part, "null pointer");
    DCL_CHECK_ARG(dpad || dcl_winattn_npad(H, W) == 0, "dpad is required when H or W is not a multiple of 7");
    WaArgs a = {};
    DCL_CHECK_ARG(fill_args(a, B, H, W, C, heads, shift, scale) == 0,
                  "bad shape (window 7, head_dim 32: C must be 32 * heads; 0 <= shift < 7)");
    DCL_CHECK_ARG(((((uintptr_t)qkv) | ((uintptr_t)dout) | ((uintptr_t)dqkv) | ((uintptr_t)qkv_bias)) & 15) == 0,
                  "16-byte alignment");
    a.qkv = qkv; a.qkv_bias = qkv_bias; a.bias = bias; a.lse = const_cast<float *>(lse); a.dout = dout;
    a.dqkv = dqkv; a.dpad = dpad; a.dbias_part = dbias_part; a.dqkv_amax = dqkv_amax;
    a.nwaves = dcl_winattn_bwd_waves(B, H, W, heads);
    if ((g_winattn_mfma & 2) && (long long)B * H * W < (1LL << 30)) {     // matrix-core kernel (dcl_winattn_mfma.hip)
        dcl_winattn_bwd_mfma_launch(qkv, qkv_bias, bias, lse, dout, B, H, W, C, heads, shift, scale, dqkv, dpad, dbias_part,
                                    dqkv_amax, a.nwaves, (hipStream_t)stream);
        DCL_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(k_winattn_bwd, dim3((unsigned)a.nwaves), dim3(64), 0, (hipStream_t)stream, a);
    DCL_LAUNCH_CHECK();
    return 0;
}
