// dcl_host_rng.cpp -- host half of the sampling step: the permutation draws.
//
// The reference draws one torch.randperm(count) per (image, class) pair from PyTorch's global CPU
// generator (losses/DenseContrastiveLossV2.py:121) and keeps the first V entries.  To stay
// bit-identical to it under a fixed seed while avoiding ~700 Python-level randperm calls per step,
// this file advances THE SAME generator state natively: at::mt19937 (MT19937, 32-bit output) and
// randperm_cpu's forward Fisher-Yates (z = draw % (n - i), n - 1 draws per call; PyTorch 2.10,
// n < 2^32 / 20).  The state is exchanged with torch.get_rng_state() / set_rng_state() (legacy
// CPUGeneratorImplState byte layout, offsets below); tests pin both against torch.randperm itself.
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/dcl_hip.h"

void dcl_set_error(const char *fmt, ...);

namespace {

// byte layout of the tensor returned by torch.get_rng_state() (5056 bytes)
constexpr size_t OFF_LEFT = 8;      // int32
constexpr size_t OFF_SEEDED = 12;   // int32
constexpr size_t OFF_NEXT = 16;     // uint64
constexpr size_t OFF_STATE = 24;    // uint64[624], each holding a 32-bit word
constexpr size_t STATE_BYTES = 5056;
constexpr int MT_N = 624, MT_M = 397;

struct Mt {
    uint32_t s[MT_N];
    int left;
    uint32_t next;

    // the three-segment form of the MT19937 twist (no modulo in the index: the first two loops auto-vectorise; same words
    // as the one-loop form with wrapped indices: s[k + M] is an OLD word for k < N - M and the NEW word k + M - N beyond)
    void twist()
    {
        auto mix = [](uint32_t a, uint32_t b) {
            const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
            return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        };
        for (int k = 0; k < MT_N - MT_M; ++k)
            s[k] = s[k + MT_M] ^ mix(s[k], s[k + 1]);
        for (int k = MT_N - MT_M; k < MT_N - 1; ++k)
            s[k] = s[k + MT_M - MT_N] ^ mix(s[k], s[k + 1]);
        s[MT_N - 1] = s[MT_M - 1] ^ mix(s[MT_N - 1], s[0]);
        left = MT_N;
        next = 0;
    }
    // k draws whose values are not needed: the state advances exactly as k calls of draw() would move it, without the
    // per-word tempering (randperm consumes n - 1 draws per call, the sampling keeps the first V positions: at the benchmark
    // shape 97 % of the ~516 k draws of a step are skipped)
    void skip(int64_t k)
    {
        while (k > 0) {
            if (left > 1) {
                const int64_t m = k < (int64_t)(left - 1) ? k : (int64_t)(left - 1);
                left -= (int)m;
                next += (uint32_t)m;
                k -= m;
            } else {            // left == 1: this draw twists (and emits word 0)
                twist();
                next = 1;
                k -= 1;
            }
        }
    }
    inline uint32_t draw()
    {
        if (--left == 0)
            twist();
        uint32_t y = s[next++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
};

}  // namespace

// at::mt19937::next_state() twists when --left_ hits 0 and twists the array as a whole; the
// sequential form above produces the same stream as long as (left, next) follow ATen's convention:
// left counts down from 624 after a twist, next indexes the word to emit.
extern "C" int dcl_host_randperm_select(uint8_t *rng_state_host, int64_t state_bytes,
                                        const int32_t *counts_host, int T, int V,
                                        int32_t *sel_host)
{
    if (!rng_state_host || !counts_host || !sel_host || T < 0 || V < 0) {
        dcl_set_error("dcl_host_randperm_select: bad arguments");
        return DCL_EINVAL;
    }
    if (state_bytes != (int64_t)STATE_BYTES) {
        dcl_set_error("dcl_host_randperm_select: unexpected torch RNG state size %lld (expected %zu)",
                      (long long)state_bytes, STATE_BYTES);
        return DCL_EUNSUPPORTED;
    }
    Mt g;
    int32_t left32, seeded;
    uint64_t next64;
    memcpy(&left32, rng_state_host + OFF_LEFT, 4);
    memcpy(&seeded, rng_state_host + OFF_SEEDED, 4);
    memcpy(&next64, rng_state_host + OFF_NEXT, 8);
    for (int k = 0; k < MT_N; ++k) {
        uint64_t w;
        memcpy(&w, rng_state_host + OFF_STATE + 8 * (size_t)k, 8);
        g.s[k] = (uint32_t)w;
    }
    g.left = left32;
    g.next = (uint32_t)next64;
    if (g.left < 1 || g.left > MT_N || g.next > (uint32_t)MT_N) {
        dcl_set_error("dcl_host_randperm_select: implausible generator state (left=%d next=%u)",
                      g.left, g.next);
        return DCL_EUNSUPPORTED;
    }
    // forward Fisher-Yates on a SPARSE identity permutation: only positions < V are kept (they are final after step V - 1),
    // so at most 2 V entries ever differ from the identity -- a small open-addressed table instead of an n-entry array per pair
    std::vector<int64_t> keys;
    std::vector<int32_t> vals;
    size_t cap = 64;
    while (cap < (size_t)(4 * V + 8))
        cap <<= 1;
    keys.assign(cap, -1);
    vals.assign(cap, 0);
    auto slot = [&](int64_t key) {
        size_t h = ((uint64_t)key * 0x9E3779B97F4A7C15ull) >> 32 & (cap - 1);
        while (keys[h] != -1 && keys[h] != key)
            h = (h + 1) & (cap - 1);
        return h;
    };
    auto get = [&](int64_t key) {
        const size_t h = slot(key);
        return keys[h] == key ? vals[h] : (int32_t)key;
    };
    std::vector<size_t> touched;
    std::vector<int32_t> dense;
    auto put = [&](int64_t key, int32_t v) {
        const size_t h = slot(key);
        if (keys[h] == -1)
            touched.push_back(h);
        keys[h] = key;
        vals[h] = v;
    };
    for (int t = 0; t < T; ++t) {
        const int64_t n = counts_host[t];
        if (n < V || n >= (int64_t)(0xffffffffu / 20)) {
            dcl_set_error("dcl_host_randperm_select: pair %d has %lld pixels (V=%d)", t,
                          (long long)n, V);
            return DCL_EINVAL;
        }
        int32_t *out = sel_host + (size_t)t * V;
        int64_t i = 0;
        if (n <= 1024) {
            // short lists: a dense array is cheaper than hashing
            dense.resize((size_t)n);
            for (int64_t k = 0; k < n; ++k)
                dense[(size_t)k] = (int32_t)k;
            for (; i < n - 1 && i < V; ++i) {
                const int64_t z = (int64_t)(g.draw() % (uint32_t)(n - i));
                std::swap(dense[(size_t)i], dense[(size_t)(z + i)]);
            }
            g.skip(n - 1 - i);
            memcpy(out, dense.data(), sizeof(int32_t) * (size_t)V);
            continue;
        }
        for (size_t h : touched)
            keys[h] = -1;
        touched.clear();
        // positions < V are final after step i = V - 1; later steps only consume draws
        for (; i < n - 1 && i < V; ++i) {
            const int64_t z = (int64_t)(g.draw() % (uint32_t)(n - i));
            const int32_t a = get(i), b = get(z + i);
            put(i, b);
            put(z + i, a);
        }
        g.skip(n - 1 - i);
        for (int v = 0; v < V; ++v)
            out[v] = get(v);
    }
    left32 = g.left;
    next64 = g.next;
    memcpy(rng_state_host + OFF_LEFT, &left32, 4);
    memcpy(rng_state_host + OFF_NEXT, &next64, 8);
    for (int k = 0; k < MT_N; ++k) {
        const uint64_t w = g.s[k];
        memcpy(rng_state_host + OFF_STATE + 8 * (size_t)k, &w, 8);
    }
    return 0;
}
