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

    void twist()
    {
        for (int k = 0; k < MT_N; ++k) {
            const uint32_t y = (s[k] & 0x80000000u) | (s[(k + 1) % MT_N] & 0x7fffffffu);
            s[k] = s[(k + MT_M) % MT_N] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        left = MT_N;
        next = 0;
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
    std::vector<int32_t> perm;
    for (int t = 0; t < T; ++t) {
        const int64_t n = counts_host[t];
        if (n < V || n >= (int64_t)(0xffffffffu / 20)) {
            dcl_set_error("dcl_host_randperm_select: pair %d has %lld pixels (V=%d)", t,
                          (long long)n, V);
            return DCL_EINVAL;
        }
        perm.resize((size_t)n);
        for (int64_t i = 0; i < n; ++i)
            perm[(size_t)i] = (int32_t)i;
        int64_t i = 0;
        // positions < V are final after step i = V - 1; later steps only consume draws
        for (; i < n - 1 && i < V; ++i) {
            const int64_t z = (int64_t)(g.draw() % (uint32_t)(n - i));
            const int32_t tmp = perm[(size_t)i];
            perm[(size_t)i] = perm[(size_t)(z + i)];
            perm[(size_t)(z + i)] = tmp;
        }
        for (; i < n - 1; ++i)
            (void)g.draw();
        memcpy(sel_host + (size_t)t * V, perm.data(), sizeof(int32_t) * (size_t)V);
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
