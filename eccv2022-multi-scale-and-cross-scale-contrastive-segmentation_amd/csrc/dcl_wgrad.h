// dcl_wgrad.h -- launch arguments shared by the per-wave weight-gradient kernels (dcl_wgrad3x3.hip, dcl_wgrad3x3d.hip)
#pragma once
#include "dcl_common.h"

struct WgradArgs {
    const float *x, *dy;
    float *part;                 // [S][9][Cout][Cin]
    const float *xamax, *gamax;
    int xcount, gcount;
    int N, Cin, Cout, H, W;
    int Hd, Wd;                  // stored size of dy (= H, W; or the even samples of a zero-inserted dy: stride 2)
    int strips, nseg, units, S, ncig, npairs, nx;
    int rect_c, rect_i, rect_mode;  // many pairs: pair-grid rectangle that shares an XCD (rect_c * rect_i = 32)
    int wave_mode;               // 1: S pixel splits per pair dealt out to WAVES (k_wgrad3x3d<.., true>), one slab per split
    int band;                    // rows per column of the traversal (H, or a divisor of H: wave form, see k_wgrad3x3d)
    int grp;                     // 1 | 2 | 4: the waves of a workgroup walk `grp` ADJACENT strips over the same rows (k_wgrad3x3d)
    const float *pre_sc, *pre_sh;   // k_wgrad3x3d PRE forms: the operand is relu(x * pre_sc[ci] + pre_sh[ci]); NULL otherwise
};

// dcl_wgrad3x3d.hip: the stride-1 kernel with LDS-DMA operand staging; same grid, slabs and arguments as k_wgrad3x3
bool dcl_wgrad_dma_supported(int nco, int nci);
void dcl_wgrad_dma_launch(const WgradArgs &a, int nco, int nci, dim3 grid, hipStream_t s);
bool dcl_wgrad_dma_wave_mode_supported(int nco, int nci);
