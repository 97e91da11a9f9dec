// pk_mfma_hazard.hip -- stand-alone reproducer of the fault described in DESIGN.md section 7 ("Packed FP32 beside MFMA"), MI355X:
//
//   VICTIM     a packed-FP32 instruction (v_pk_fma_f32, v_pk_mul_f32, v_pk_add_f32) whose op_sel selects the HIGH half of src1 for
//              the LOW result (op_sel:[0,1,0] / [0,1]) -- what the compiler emits to broadcast the second of two packed scalars;
//   AGGRESSOR  another wave on the same SIMD that issues MFMAs while operand fragments ARRIVE in its registers (ds_read_b128 or
//              global loads in flight under the matrix instructions: the software pipeline of every real tile loop);
//   RESULT     the victim's LOW result is wrong in lanes 48-63 (the last quarter-wave), ~1e-5 of the executions; the high result,
//              lanes 0-47, other op_sel bits (src0, src2, op_sel_hi), the plain forms and the scalar instructions are never wrong;
//              no mismatch beside MFMA loops with resident operands, beside the same arrivals with vector FMAs, or alone.
//
//     hipcc -O3 --offload-arch=gfx950 tools/probes/pk_mfma_hazard.hip -o tools/probes/variants/pk_mfma_hazard
//     gpurun -- tools/probes/variants/pk_mfma_hazard 20          (output of a run: profiles/r05_pk_mfma_hazard.txt)
//
// Every packed result is compared with the same arithmetic on scalar v_*_f32 instructions (inline asm) in the same lane; mismatches
// are counted per lane and half.  The file also builds as a shared library (-shared -fPIC) whose pkh_launch_* entry points let
// tools/probes/pk_mix.py put these kernels beside the library's own (how the pair was narrowed: the library's norm backward beside
// synthetic loops -> nothing; synthetic victims beside the library's matrix kernels -> the fifth instruction of the norm sequence,
// the one with op_sel:[0,1,0]; then the aggressor: a convolution without its MFMA instructions -> nothing, MFMA loops on resident
// operands -> nothing, MFMA loops with arriving operands -> this table).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                                      \
    do {                                                                                              \
        hipError_t e_ = (x);                                                                          \
        if (e_ != hipSuccess) {                                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));                 \
            exit(1);                                                                                  \
        }                                                                                             \
    } while (0)

// OP: 0 = v_pk_fma_f32, 1 = v_pk_mul_f32, 2 = v_pk_add_f32
template <int OP>
__global__ __launch_bounds__(256) void k_pk(const float *__restrict__ in, unsigned *__restrict__ bad, int iters)
{
    const int lane = threadIdx.x & 63;
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    f32x2 a = {in[4 * g], in[4 * g + 1]}, b = {in[4 * g + 2], in[4 * g + 3]};
    f32x2 c = {0.25f, -0.5f};
    float s0 = c.x, s1 = c.y;
    unsigned nbad_lo = 0, nbad_hi = 0;
    for (int i = 0; i < iters; ++i) {
        f32x2 p;
        float q0, q1;
        if (OP == 0) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(p) : "v"(a), "v"(b), "v"(c));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(q0) : "v"(a.x), "v"(b.x), "v"(s0));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(q1) : "v"(a.y), "v"(b.y), "v"(s1));
        } else if (OP == 1) {
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p) : "v"(a), "v"(c));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(q0) : "v"(a.x), "v"(s0));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(q1) : "v"(a.y), "v"(s1));
        } else {
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p) : "v"(b), "v"(c));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(q0) : "v"(b.x), "v"(s0));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(q1) : "v"(b.y), "v"(s1));
        }
        nbad_lo += __float_as_uint(p.x) != __float_as_uint(q0);
        nbad_hi += __float_as_uint(p.y) != __float_as_uint(q1);
        // keep the chain bounded: fold back into (-1, 1), the scalar chain continues from the SCALAR results
        c.x = q0 - truncf(q0);
        c.y = q1 - truncf(q1);
        s0 = c.x;
        s1 = c.y;
    }
    if (nbad_lo)
        atomicAdd(bad + lane, nbad_lo);
    if (nbad_hi)
        atomicAdd(bad + 64 + lane, nbad_hi);
}

// The instruction sequence of the norm backward (k_bn_bwd_apply: o = k (g - mg - (x - m) is mgx) on two elements), operands m, is
// in SCALAR registers, modifiers as the compiler emitted them, fed from memory like there: a burst of packed operations behind loads.
// MODE 0: as described; 1: (mg, mgx) from the kernel arguments, no LDS, no barriers; 2: ONLY the LDS broadcast is checked (bit
// pattern of what every lane reads against the arguments); 3: as 0 without global loads in the loop (operands from registers)
template <int MODE>
__global__ __launch_bounds__(256) void k_pk_bn(const float *__restrict__ in, unsigned *__restrict__ bad, int iters, float m, float is,
                                               float mg, float mgx, float k)
{
    const int lane = threadIdx.x & 63;
    const size_t g0 = (size_t)blockIdx.x * 256 + threadIdx.x;
    unsigned nbad_lo = 0, nbad_hi = 0;
    f32x2 ms = {m, m}, iss = {is, is};
    f32x2 kk = {k, k};
    __shared__ float bc[2];
    const size_t total = (size_t)gridDim.x * 256;
    // four vectors per trip, all eight loads issued before the first is used (the later ones are still in flight during the first
    // packed operations, as in the norm kernel); (mg, mgx) come from an LDS broadcast behind a barrier
    f32x4 v0 = *(const f32x4 *)(in + 4 * (g0 % total));
    for (int i = 0; i < iters; i += 4) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE == 3) {
                v[u] = v0;
                v[u].x += 0.001f * (float)(i + u);
            } else {
                v[u] = *(const f32x4 *)(in + 4 * ((g0 + (size_t)(i + u) * 256 * 7) % total));
            }
        }
        f32x2 mm = {mg, mgx};
        if (MODE != 1) {
            if (threadIdx.x == 0) {
                bc[0] = mg;
                bc[1] = mgx;
            }
            __syncthreads();
            mm = *(const f32x2 *)bc;
        }
        if (MODE == 2) {
            nbad_lo += __float_as_uint(mm.x) != __float_as_uint(mg);
            nbad_hi += __float_as_uint(mm.y) != __float_as_uint(mgx);
            asm volatile("" : "+v"(v[0]));          // (the loads stay)
            __syncthreads();
            if (threadIdx.x == 0) {                 // next trip writes again: make the slot differ in between
                bc[0] = 0.f;
                bc[1] = 0.f;
            }
            __syncthreads();
            continue;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const f32x2 x = {v[u].x, v[u].y}, gr = {v[u].z, v[u].w};
            f32x2 t, w, p;
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(x), "s"(ms));
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(w) : "v"(gr), "v"(mm));
            asm volatile("v_pk_mul_f32 %0, %1, %0 op_sel_hi:[0,1]" : "+v"(t) : "s"(iss));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "+v"(t) : "v"(mm), "v"(w));
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(p) : "v"(kk), "v"(t));
            float q[2];
            for (int e = 0; e < 2; ++e) {
                float a, b, c;
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a) : "v"(x[e]), "v"(m));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(b) : "v"(gr[e]), "v"(mg));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a) : "v"(is), "v"(a));
                asm volatile("v_fma_f32 %0, -%1, %2, %3" : "=v"(c) : "v"(a), "v"(mgx), "v"(b));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(q[e]) : "v"(k), "v"(c));
            }
            nbad_lo += __float_as_uint(p.x) != __float_as_uint(q[0]);
            nbad_hi += __float_as_uint(p.y) != __float_as_uint(q[1]);
        }
        if (MODE != 1)
            __syncthreads();
    }
    if (nbad_lo)
        atomicAdd(bad + lane, nbad_lo);
    if (nbad_hi)
        atomicAdd(bad + 64 + lane, nbad_hi);
}


// Which instruction of the sequence is it?  The five packed instructions one after the other as in k_pk_bn<1> (operands from the
// arguments, no LDS, no loads in the loop); after EVERY one its result is compared with the scalar chain's value at that stage:
// bad[stage * 2 + half] counts the first stage at which a lane's value differs (later stages inherit the difference).
__global__ __launch_bounds__(256) void k_pk_stage(const float *__restrict__ in, unsigned *__restrict__ bad, int iters, float m, float is,
                                                  float mg, float mgx, float k)
{
    const size_t g0 = (size_t)blockIdx.x * 256 + threadIdx.x;
    const f32x2 ms = {m, m}, iss = {is, is}, mm = {mg, mgx}, kk = {k, k};
    const f32x4 v0 = *(const f32x4 *)(in + 4 * g0);
    unsigned cnt[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
        f32x4 v = v0;
        v.x += 0.001f * (float)i;
        v.w -= 0.002f * (float)i;
        const f32x2 x = {v.x, v.y}, gr = {v.z, v.w};
        f32x2 t, w, p;
        float a[2], b[2], c[2], q[2];
        for (int e = 0; e < 2; ++e) {
            asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[e]) : "v"(x[e]), "v"(m));
            asm volatile("v_sub_f32 %0, %1, %2" : "=v"(b[e]) : "v"(gr[e]), "v"(mg));
        }
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(x), "s"(ms));
        bool done[2] = {false, false};
        for (int e = 0; e < 2; ++e)
            if (__float_as_uint(t[e]) != __float_as_uint(a[e])) { cnt[0 + e]++; done[e] = true; }
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(w) : "v"(gr), "v"(mm));
        for (int e = 0; e < 2; ++e)
            if (!done[e] && __float_as_uint(w[e]) != __float_as_uint(b[e])) { cnt[2 + e]++; done[e] = true; }
        for (int e = 0; e < 2; ++e)
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[e]) : "v"(is), "v"(a[e]));
        asm volatile("v_pk_mul_f32 %0, %1, %0 op_sel_hi:[0,1]" : "+v"(t) : "s"(iss));
        for (int e = 0; e < 2; ++e)
            if (!done[e] && __float_as_uint(t[e]) != __float_as_uint(a[e])) { cnt[4 + e]++; done[e] = true; }
        for (int e = 0; e < 2; ++e)
            asm volatile("v_fma_f32 %0, -%1, %2, %3" : "=v"(c[e]) : "v"(a[e]), "v"(mgx), "v"(b[e]));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "+v"(t) : "v"(mm), "v"(w));
        for (int e = 0; e < 2; ++e)
            if (!done[e] && __float_as_uint(t[e]) != __float_as_uint(c[e])) { cnt[6 + e]++; done[e] = true; }
        for (int e = 0; e < 2; ++e)
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(q[e]) : "v"(k), "v"(c[e]));
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(p) : "v"(kk), "v"(t));
        for (int e = 0; e < 2; ++e)
            if (!done[e] && __float_as_uint(p[e]) != __float_as_uint(q[e])) { cnt[8 + e]++; done[e] = true; }
    }
    for (int j = 0; j < 10; ++j)
        if (cnt[j])
            atomicAdd(bad + j, cnt[j]);
}


// One v_pk_fma_f32 per trip with the modifiers of form F against the same arithmetic on v_fma_f32:
//   0 plain | 1 op_sel:[0,1,0] (low result takes src1's HIGH half) | 2 neg_lo:[1,0,0] neg_hi:[1,0,0] | 3 both (the norm kernel's form)
//   4 op_sel_hi:[1,0,1] (high result takes src1's LOW half) | 5 op_sel:[1,0,0] (low result takes src0's high half)
//   6 v_pk_mul_f32 op_sel:[0,1] | 7 v_pk_add_f32 op_sel:[0,1] | 8 v_pk_fma_f32 op_sel:[0,0,1] (low result takes src2's high half)
template <int F>
__global__ __launch_bounds__(256) void k_pk_form(const float *__restrict__ in, unsigned *__restrict__ bad, int iters)
{
    const int lane = threadIdx.x & 63;
    const size_t g0 = (size_t)blockIdx.x * 256 + threadIdx.x;
    const f32x4 v0 = *(const f32x4 *)(in + 4 * g0);
    unsigned nlo = 0, nhi = 0;
    for (int i = 0; i < iters; ++i) {
        f32x2 t = {v0.x + 0.001f * (float)i, v0.y}, mm = {v0.z, v0.w - 0.002f * (float)i}, w = {v0.y, v0.x}, p;
        float q0, q1;
        if (F == 0) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(p) : "v"(t), "v"(mm), "v"(w));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(q0) : "v"(t.x), "v"(mm.x), "v"(w.x));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(q1) : "v"(t.y), "v"(mm.y), "v"(w.y));
        } else if (F == 1) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(p) : "v"(t), "v"(mm), "v"(w));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(q0) : "v"(t.x), "v"(mm.y), "v"(w.x));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(q1) : "v"(t.y), "v"(mm.y), "v"(w.y));
        } else if (F == 2) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(p) : "v"(t), "v"(mm), "v"(w));
            asm volatile("v_fma_f32 %0, -%1, %2, %3" : "=v"(q0) : "v"(t.x), "v"(mm.x), "v"(w.x));
            asm volatile("v_fma_f32 %0, -%1, %2, %3" : "=v"(q1) : "v"(t.y), "v"(mm.y), "v"(w.y));
        } else if (F == 3) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(p) : "v"(t), "v"(mm), "v"(w));
            asm volatile("v_fma_f32 %0, -%1, %2, %3" : "=v"(q0) : "v"(t.x), "v"(mm.y), "v"(w.x));
            asm volatile("v_fma_f32 %0, -%1, %2, %3" : "=v"(q1) : "v"(t.y), "v"(mm.y), "v"(w.y));
        } else if (F == 4) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(p) : "v"(t), "v"(mm), "v"(w));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(q0) : "v"(t.x), "v"(mm.x), "v"(w.x));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(q1) : "v"(t.y), "v"(mm.x), "v"(w.y));
        } else if (F == 5) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=v"(p) : "v"(t), "v"(mm), "v"(w));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(q0) : "v"(t.y), "v"(mm.x), "v"(w.x));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(q1) : "v"(t.y), "v"(mm.y), "v"(w.y));
        } else if (F == 6) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(p) : "v"(t), "v"(mm));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(q0) : "v"(t.x), "v"(mm.y));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(q1) : "v"(t.y), "v"(mm.y));
        } else if (F == 7) {
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(p) : "v"(t), "v"(mm));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(q0) : "v"(t.x), "v"(mm.y));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(q1) : "v"(t.y), "v"(mm.y));
        } else if (F == 8) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]" : "=v"(p) : "v"(t), "v"(mm), "v"(w));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(q0) : "v"(t.x), "v"(mm.x), "v"(w.y));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(q1) : "v"(t.y), "v"(mm.y), "v"(w.y));
        } else if (F == 9) {            // the op_sel_hi forms the packed builds of dcl_gemm.hip / dcl_sweep.hip contain
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(p) : "v"(t), "v"(mm), "v"(w));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(q0) : "v"(t.x), "v"(mm.x), "v"(w.x));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(q1) : "v"(t.x), "v"(mm.y), "v"(w.y));
        } else if (F == 10) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,1,0]" : "=v"(p) : "v"(t), "v"(mm), "v"(w));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(q0) : "v"(t.x), "v"(mm.x), "v"(w.x));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(q1) : "v"(t.y), "v"(mm.y), "v"(w.x));
        } else if (F == 11) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(p) : "v"(t), "v"(mm), "v"(w));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(q0) : "v"(t.x), "v"(mm.x), "v"(w.x));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(q1) : "v"(t.y), "v"(mm.x), "v"(w.x));
        } else if (F == 12) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(p) : "v"(t), "v"(mm));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(q0) : "v"(t.x), "v"(mm.x));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(q1) : "v"(t.x), "v"(mm.y));
        } else {
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(p) : "v"(t), "v"(mm));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(q0) : "v"(t.x), "v"(mm.x));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(q1) : "v"(t.y), "v"(mm.x));
        }
        nlo += __float_as_uint(p.x) != __float_as_uint(q0);
        nhi += __float_as_uint(p.y) != __float_as_uint(q1);
    }
    if (nlo)
        atomicAdd(bad + lane, nlo);
    if (nhi)
        atomicAdd(bad + 64 + lane, nhi);
}

// KIND: 0 = VALU FMAs, 1 = v_mfma_f32_16x16x32_f16, 2 = v_mfma_f32_32x32x16_f16, 3 = v_mfma_f32_16x16x16_f16, 4 = v_mfma_f32_32x32x8_f16
template <int KIND>
__global__ __launch_bounds__(256) void k_mfma(float *__restrict__ out, int iters)
{
    const int lane = threadIdx.x & 63;
    half8 a8, b8;
    half4 a4, b4;
    for (int i = 0; i < 8; ++i) {
        a8[i] = (_Float16)(0.001f * (lane + i));
        b8[i] = (_Float16)(0.002f * (lane - i));
    }
    for (int i = 0; i < 4; ++i) {
        a4[i] = a8[i];
        b4[i] = b8[i];
    }
    // four INDEPENDENT accumulators per shape: the matrix pipe is issued to back to back, as in the product kernels
    f32x4 c4[4];
    f32x16 c16[4];
    for (int u = 0; u < 4; ++u) {
        c4[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < 16; ++i)
            c16[u][i] = 0.f;
    }
    float f = 0.5f;
    __shared__ half8 ops[2 * 256];
    ops[threadIdx.x] = a8;
    ops[256 + threadIdx.x] = b8;
    __syncthreads();
    for (int i = 0; i < iters; ++i) {
        a8 = ops[(threadIdx.x + i) & 255];              // LDS operand reads between the matrix instructions, as in a real tile loop
        b8 = ops[256 + ((threadIdx.x + 2 * i) & 255)];
        for (int e = 0; e < 4; ++e) {
            a4[e] = a8[e];
            b4[e] = b8[e];
        }
        if (KIND == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f) : "v"(0.999f), "v"(0.001f));
        } else if (KIND == 1) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                c4[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c4[u], 0, 0, 0);
        } else if (KIND == 2) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                c16[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, c16[u], 0, 0, 0);
        } else if (KIND == 3) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                c4[u] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c4[u], 0, 0, 0);
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                c16[u] = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, c16[u], 0, 0, 0);
        }
    }
    float r = f;
    for (int u = 0; u < 4; ++u) {
        r += c4[u][0] + c4[u][1] + c4[u][2] + c4[u][3];
        for (int i = 0; i < 16; ++i)
            r += c16[u][i];
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = r;
}

template <int KIND>
static void launch_mfma(float *out, int blocks, int iters, hipStream_t st)
{
    hipLaunchKernelGGL(k_mfma<KIND>, dim3(blocks), dim3(256), 0, st, out, iters);
}


// ---- entry points for tools/probes/pk_coresident.py (the library's kernels on one stream, these on the other) -----------------
extern "C" int pkh_launch_mfma(int kind, int blocks, int iters, float *out, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (kind == 0) launch_mfma<0>(out, blocks, iters, st);
    else if (kind == 1) launch_mfma<1>(out, blocks, iters, st);
    else if (kind == 2) launch_mfma<2>(out, blocks, iters, st);
    else if (kind == 3) launch_mfma<3>(out, blocks, iters, st);
    else launch_mfma<4>(out, blocks, iters, st);
    return (int)hipGetLastError();
}
extern "C" int pkh_launch_victim(int op, int blocks, int iters, const float *in, unsigned *bad, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (op == 0) hipLaunchKernelGGL(k_pk<0>, dim3(blocks), dim3(256), 0, st, in, bad, iters);
    else if (op == 1) hipLaunchKernelGGL(k_pk<1>, dim3(blocks), dim3(256), 0, st, in, bad, iters);
    else if (op == 2) hipLaunchKernelGGL(k_pk<2>, dim3(blocks), dim3(256), 0, st, in, bad, iters);
    else if (op == 3) hipLaunchKernelGGL(k_pk_bn<0>, dim3(blocks), dim3(256), 0, st, in, bad, iters, 0.31f, 1.7f, 0.013f, -0.021f, 2.9f);
    else if (op == 4) hipLaunchKernelGGL(k_pk_bn<1>, dim3(blocks), dim3(256), 0, st, in, bad, iters, 0.31f, 1.7f, 0.013f, -0.021f, 2.9f);
    else if (op == 5) hipLaunchKernelGGL(k_pk_bn<2>, dim3(blocks), dim3(256), 0, st, in, bad, iters, 0.31f, 1.7f, 0.013f, -0.021f, 2.9f);
    else if (op == 6) hipLaunchKernelGGL(k_pk_bn<3>, dim3(blocks), dim3(256), 0, st, in, bad, iters, 0.31f, 1.7f, 0.013f, -0.021f, 2.9f);
    else if (op == 7) hipLaunchKernelGGL(k_pk_stage, dim3(blocks), dim3(256), 0, st, in, bad, iters, 0.31f, 1.7f, 0.013f, -0.021f, 2.9f);
    else if (op == 10) hipLaunchKernelGGL(k_pk_form<0>, dim3(blocks), dim3(256), 0, st, in, bad, iters);
    else if (op == 11) hipLaunchKernelGGL(k_pk_form<1>, dim3(blocks), dim3(256), 0, st, in, bad, iters);
    else if (op == 12) hipLaunchKernelGGL(k_pk_form<2>, dim3(blocks), dim3(256), 0, st, in, bad, iters);
    else if (op == 13) hipLaunchKernelGGL(k_pk_form<3>, dim3(blocks), dim3(256), 0, st, in, bad, iters);
    else if (op == 14) hipLaunchKernelGGL(k_pk_form<4>, dim3(blocks), dim3(256), 0, st, in, bad, iters);
    else if (op == 15) hipLaunchKernelGGL(k_pk_form<5>, dim3(blocks), dim3(256), 0, st, in, bad, iters);
    else if (op == 16) hipLaunchKernelGGL(k_pk_form<6>, dim3(blocks), dim3(256), 0, st, in, bad, iters);
    else if (op == 17) hipLaunchKernelGGL(k_pk_form<7>, dim3(blocks), dim3(256), 0, st, in, bad, iters);
    else if (op == 18) hipLaunchKernelGGL(k_pk_form<8>, dim3(blocks), dim3(256), 0, st, in, bad, iters);
    else if (op == 19) hipLaunchKernelGGL(k_pk_form<9>, dim3(blocks), dim3(256), 0, st, in, bad, iters);
    else if (op == 20) hipLaunchKernelGGL(k_pk_form<10>, dim3(blocks), dim3(256), 0, st, in, bad, iters);
    else if (op == 21) hipLaunchKernelGGL(k_pk_form<11>, dim3(blocks), dim3(256), 0, st, in, bad, iters);
    else if (op == 22) hipLaunchKernelGGL(k_pk_form<12>, dim3(blocks), dim3(256), 0, st, in, bad, iters);
    else hipLaunchKernelGGL(k_pk_form<13>, dim3(blocks), dim3(256), 0, st, in, bad, iters);
    return (int)hipGetLastError();
}

// Aggressor candidates WITHOUT matrix instructions: loops of the mixed-precision VOP3P instructions the library's matrix kernels split
// their operands with (v_fma_mixlo_f16 / v_fma_mixhi_f16, with and without op_sel), and of v_pk_fma_f32 itself.
// KIND: 0 v_fma_mixlo/hi_f16 plain | 1 the split2 sequence of the library (op_sel / op_sel_hi modifiers) | 2 v_pk_fma_f32 with op_sel:[0,1,0]
template <int KIND>
__global__ __launch_bounds__(256) void k_mix(float *__restrict__ out, int iters)
{
    float v0 = 0.001f * (float)threadIdx.x, v1 = 0.5f - v0, sc = 1024.f;
    unsigned hi = 0, lo = 0;
    f32x2 t = {v0, v1}, mm = {v1, v0}, w = {0.25f, 0.75f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0) {
                asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hi) : "v"(v0), "v"(sc));
                asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hi) : "v"(v1), "v"(sc));
            } else if (KIND == 1) {
                asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hi) : "v"(v0), "v"(sc));
                asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hi) : "v"(v1), "v"(sc));
                asm volatile("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=&v"(lo) : "v"(v0), "v"(sc), "v"(hi));
                asm volatile("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(v1), "v"(sc), "v"(hi));
            } else {
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,1,0]" : "+v"(t) : "v"(mm), "v"(w));
            }
        }
        v0 += 1e-6f;
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = v0 + __uint_as_float(hi) + __uint_as_float(lo) + t.x + t.y;
}
extern "C" int pkh_launch_mix(int kind, int blocks, int iters, float *out, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (kind == 0) hipLaunchKernelGGL(k_mix<0>, dim3(blocks), dim3(256), 0, st, out, iters);
    else if (kind == 1) hipLaunchKernelGGL(k_mix<1>, dim3(blocks), dim3(256), 0, st, out, iters);
    else hipLaunchKernelGGL(k_mix<2>, dim3(blocks), dim3(256), 0, st, out, iters);
    return (int)hipGetLastError();
}

// A matrix-instruction loop that OCCUPIES most of a SIMD's register file like the library's tiles do (NACC accumulator tiles of 16
// registers, all live, cycled through v_mfma_f32_32x32x16_f16): a co-resident wave then gets its registers behind them.
template <int NACC>
__global__ __launch_bounds__(256) void k_mfma_fat(float *__restrict__ out, int iters, const float *__restrict__ rnd)
{
    const int lane = threadIdx.x & 63;
    half8 a8[2], b8[2];
    for (int u = 0; u < 2; ++u)
        for (int i = 0; i < 8; ++i) {
            // rnd != NULL: operands with random bit patterns (as a real tile's), else smooth small numbers
            a8[u][i] = rnd ? (_Float16)(100.f * rnd[(threadIdx.x * 16 + i + 8 * u) & 0xffff]) : (_Float16)(0.001f * (lane + i + u));
            b8[u][i] = rnd ? (_Float16)(100.f * rnd[(threadIdx.x * 16 + 4096 + i + 8 * u) & 0xffff]) : (_Float16)(0.002f * (lane - i - u));
        }
    f32x16 acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t)
        for (int i = 0; i < 16; ++i)
            acc[t][i] = 0.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int t = 0; t < NACC; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8[t & 1], b8[(t >> 1) & 1], acc[t], 0, 0, 0);
    }
    float r = 0.f;
#pragma unroll
    for (int t = 0; t < NACC; ++t)
        for (int i = 0; i < 16; ++i)
            r += acc[t][i];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = r;
}
extern "C" int pkh_launch_fat(int nacc, int blocks, int iters, float *out, const float *rnd, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (nacc <= 4) hipLaunchKernelGGL(k_mfma_fat<4>, dim3(blocks), dim3(256), 0, st, out, iters, rnd);
    else if (nacc <= 8) hipLaunchKernelGGL(k_mfma_fat<8>, dim3(blocks), dim3(256), 0, st, out, iters, rnd);
    else if (nacc <= 12) hipLaunchKernelGGL(k_mfma_fat<12>, dim3(blocks), dim3(256), 0, st, out, iters, rnd);
    else hipLaunchKernelGGL(k_mfma_fat<14>, dim3(blocks), dim3(256), 0, st, out, iters, rnd);
    return (int)hipGetLastError();
}

// A matrix loop whose B fragments ARRIVE FROM LDS WHILE MATRIX INSTRUCTIONS RUN (software pipeline as in the library's tiles: the
// ds_read_b128 of the next group is issued in front of the current group's MFMAs), MODE 1: the fragments arrive from global memory
// instead (loads in flight under the MFMAs), MODE 2: both.
template <int MODE>
__global__ __launch_bounds__(256) void k_mfma_pipe(float *__restrict__ out, int iters, const float *__restrict__ rnd)
{
    __shared__ half8 frag[1024];
    for (int i = threadIdx.x; i < 1024; i += 256)
        for (int e = 0; e < 8; ++e)
            frag[i][e] = (_Float16)(10.f * rnd[(i * 8 + e) & 0xffff]);
    __syncthreads();
    const half8 *gf = (const half8 *)rnd;
    half8 a8 = frag[threadIdx.x], bq[2][2];
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t)
        for (int i = 0; i < 16; ++i)
            acc[t][i] = 0.f;
    bq[0][0] = frag[threadIdx.x];
    bq[0][1] = frag[256 + threadIdx.x];
    for (int i = 0; i < iters; ++i) {
        const int nb = (i + 1) & 1, cb = i & 1;
        if (MODE == 0 || MODE == 2 || MODE >= 3) {
            bq[nb][0] = frag[(threadIdx.x + 7 * i) & 1023];
            bq[nb][1] = frag[(threadIdx.x + 13 * i + 512) & 1023];
        }
        if (MODE == 1) {
            bq[nb][0] = gf[(threadIdx.x + 7 * i) & 4095];
            bq[nb][1] = gf[(threadIdx.x + 13 * i + 512) & 4095];
        }
        if (MODE == 2)
            a8 = gf[(threadIdx.x + 5 * i) & 4095];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (MODE == 3) {                        // control: the same arrivals, vector FMAs instead of matrix instructions
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    acc[t][e] = __builtin_fmaf((float)a8[e], (float)bq[cb][0][e], acc[t][e]) + (float)bq[cb][1][e];
            } else if (MODE == 4) {                 // LDS arrivals, v_mfma_f32_16x16x32_f16 (the weight gradients' shape)
                f32x4 c = {acc[t][0], acc[t][1], acc[t][2], acc[t][3]};
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, bq[cb][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, bq[cb][1], c, 0, 0, 0);
                acc[t][0] = c[0]; acc[t][1] = c[1]; acc[t][2] = c[2]; acc[t][3] = c[3];
            } else if (MODE == 5) {                 // LDS arrivals, v_mfma_f32_32x32x2f32 (fp32 matrix instructions)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32((float)a8[0], (float)bq[cb][0][0], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32((float)a8[1], (float)bq[cb][1][0], acc[t], 0, 0, 0);
            } else {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, bq[cb][0], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, bq[cb][1], acc[t], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    float r = 0.f;
    for (int t = 0; t < 4; ++t)
        for (int i = 0; i < 16; ++i)
            r += acc[t][i];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = r;
}
extern "C" int pkh_launch_pipe(int mode, int blocks, int iters, float *out, const float *rnd, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (mode == 0) hipLaunchKernelGGL(k_mfma_pipe<0>, dim3(blocks), dim3(256), 0, st, out, iters, rnd);
    else if (mode == 1) hipLaunchKernelGGL(k_mfma_pipe<1>, dim3(blocks), dim3(256), 0, st, out, iters, rnd);
    else if (mode == 2) hipLaunchKernelGGL(k_mfma_pipe<2>, dim3(blocks), dim3(256), 0, st, out, iters, rnd);
    else if (mode == 3) hipLaunchKernelGGL(k_mfma_pipe<3>, dim3(blocks), dim3(256), 0, st, out, iters, rnd);
    else if (mode == 4) hipLaunchKernelGGL(k_mfma_pipe<4>, dim3(blocks), dim3(256), 0, st, out, iters, rnd);
    else hipLaunchKernelGGL(k_mfma_pipe<5>, dim3(blocks), dim3(256), 0, st, out, iters, rnd);
    return (int)hipGetLastError();
}

int main(int argc, char **argv)
{
    // The stand-alone pair: victims k_pk_form<F> on stream A, aggressors on stream B; every packed result is checked against the
    // scalar instruction's in the same lane.  Expected on MI355X: mismatches ONLY for the forms with op_sel[1] = 1 (the low result
    // takes src1's high half), ONLY in the low half of lanes 48-63, ONLY beside the loops whose operands arrive under the MFMAs.
    const int rounds = argc > 1 ? atoi(argv[1]) : 20;
    const int blocks = 512;
    float *in, *out, *rnd;
    unsigned *bad;
    std::vector<float> h((size_t)blocks * 256 * 4), hr(65536);
    unsigned seed = 12345u;
    for (auto &v : h) {
        seed = seed * 1664525u + 1013904223u;
        v = ((seed >> 8) & 0xffff) / 65536.0f * 2.0f - 1.0f;
    }
    for (auto &v : hr) {
        seed = seed * 1664525u + 1013904223u;
        v = ((seed >> 8) & 0xffff) / 65536.0f * 2.0f - 1.0f;
    }
    CHECK(hipMalloc(&in, h.size() * 4));
    CHECK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&rnd, hr.size() * 4));
    CHECK(hipMemcpy(rnd, hr.data(), hr.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    CHECK(hipMalloc(&bad, 128 * 4));
    hipStream_t sa, sb;
    CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    const char *aggr[] = {"nothing", "MFMA loop, operands resident in registers (14 tiles)", "MFMA loop, B fragments arriving from LDS",
                          "MFMA loop, B fragments arriving from global memory", "the LDS arrivals with vector FMAs instead of MFMAs",
                          "v_mfma_f32_16x16x32_f16 loop, LDS arrivals", "v_mfma_f32_32x32x2f32 loop, LDS arrivals"};
    const int forms[] = {0, 1, 6, 7, 5, 8, 4, 9, 10, 11, 12, 13};
    const char *fname[] = {"v_pk_fma_f32 (plain)", "v_pk_fma_f32 op_sel:[0,1,0]", "v_pk_mul_f32 op_sel:[0,1]", "v_pk_add_f32 op_sel:[0,1]",
                           "v_pk_fma_f32 op_sel:[1,0,0]", "v_pk_fma_f32 op_sel:[0,0,1]", "v_pk_fma_f32 op_sel_hi:[1,0,1]",
                           "v_pk_fma_f32 op_sel_hi:[0,1,1]", "v_pk_fma_f32 op_sel_hi:[1,1,0]", "v_pk_fma_f32 op_sel_hi:[1,0,0]",
                           "v_pk_mul_f32 op_sel_hi:[0,1]", "v_pk_add_f32 op_sel_hi:[1,0]"};
    const int nag = argc > 2 ? atoi(argv[2]) : 5;              // (2nd argument: only the first n neighbours)
    for (int ag = (argc > 3 ? atoi(argv[3]) : 0); ag < nag; ++ag)
        for (int f = 0; f < 12; ++f) {
            CHECK(hipMemset(bad, 0, 128 * 4));
            CHECK(hipDeviceSynchronize());
            for (int r = 0; r < rounds; ++r) {
                for (int k = 0; k < 4; ++k) {
                    if (ag == 1) pkh_launch_fat(14, blocks, 4000, out, rnd, sb);
                    if (ag == 2) pkh_launch_pipe(0, blocks, 6000, out, rnd, sb);
                    if (ag == 3) pkh_launch_pipe(1, blocks, 6000, out, rnd, sb);
                    if (ag == 4) pkh_launch_pipe(3, blocks, 6000, out, rnd, sb);
                    if (ag == 5) pkh_launch_pipe(4, blocks, 6000, out, rnd, sb);
                    if (ag == 6) pkh_launch_pipe(5, blocks, 6000, out, rnd, sb);
                }
                for (int k = 0; k < 6; ++k)
                    pkh_launch_victim(10 + forms[f], blocks, 2000, in, bad, sa);
                CHECK(hipDeviceSynchronize());
            }
            unsigned hb[128];
            CHECK(hipMemcpy(hb, bad, sizeof(hb), hipMemcpyDeviceToHost));
            unsigned long long lo = 0, hi = 0, q[4] = {0, 0, 0, 0};
            for (int l = 0; l < 64; ++l) {
                lo += hb[l];
                hi += hb[64 + l];
                q[l / 16] += hb[l] + hb[64 + l];
            }
            printf("%-30s beside %-55s: low half %10llu, high half %llu; lanes 0-15 %llu, 16-31 %llu, 32-47 %llu, 48-63 %llu\n", fname[f],
                   aggr[ag], lo, hi, q[0], q[1], q[2], q[3]);
            fflush(stdout);
        }
    return 0;
}
