/* The same convolution (stride 1, automatic tile) with the batch-norm statistics of its OUTPUT reduced in the epilogue: per
 * pixel tile t and output channel c, part[(c * ntile + t) * 2 + {0, 1}] = sum over the tile's pixels of (y - pivot[c]),
 * (y - pivot[c])^2 -- the partial sums dcl_bn_apply_parts(ns = ntile) combines, so the norm that follows the convolution
 * (reference models/HRNet.py:77-93) needs no statistics pass over y.  ntile = dcl_conv3x3_bnstats_tiles(N, Cin, Cout, H, W);
 * 0 = no such kernel for the shape (Cin % 16 != 0, or a tile other than the BasicBlock tiles): use dcl_conv3x3_f16x3 +
 * dcl_bn_stats_part.  pivot f32 [Cout]: the norm's running mean; pivot_out f32 [Cout] receives a copy (the apply kernel
 * updates the running mean).  Fixed summation order: bitwise reproducible. */

int dcl_conv3x3_bnstats_tiles(int N, int Cin, int Cout, int H, int W);
int dcl_conv3x3_bnstats_f16x3(const float *x, int N, int Cin, int H, int W, const void *wp, int Cout,
                              const float *xamax, int xcount, const float *wamax, const float *addend,
                              const float *bias, float *y, const float *pivot, float *part, float *pivot_out,
                              void *stream);
