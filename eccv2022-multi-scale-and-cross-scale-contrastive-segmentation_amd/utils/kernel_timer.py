"""In-step timing of the library's launches with HIP events (bench.py's `roofline` / `roofline_hbm` legs).

``with KernelTimer() as kt: step()`` wraps the C-ABI entry points of ``libdcl_hip.so`` that launch the step's kernels:
every call is bracketed by two HIP events recorded ON THE STREAM THE CALL LAUNCHES ON (the caller's current stream -- the
branches of an exchange module run on their own streams, ``torch.cuda.Event`` on another stream would see nothing), and its
ALGORITHMIC work -- FLOPs for the matrix kernels, bytes for the streaming ones, computed from the call's own arguments with
the formulas of SURVEY.md section 8(d) / DESIGN.md section 3 -- is noted next to it.  After a synchronise, ``rows()`` groups
the calls by kernel symbol: the convolution / weight-gradient entry points report the symbol they chose through
``dcl_last_kernel`` (tile and variant are picked inside the library), the others have one kernel per entry point.

The bracket of a call holds the kernel(s) it launched plus the dispatch gap behind the first event (a few microseconds): on
the 60-90-us convolution kernels that is within 5 % of rocprofv3's per-kernel durations of the same step
(profiles/r04_step_kernels.csv), on the 5-30-us batch-norm kernels it is an upper bound of the kernel time.

Measurement only: nothing here runs inside the timed region of bench.py, and the wrappers are removed on exit."""
import torch

from .. import _lib

MFMA_F16X3_PEAK_TFLOPS = 2500.0 / 3.0      # every algorithmic FLOP = three f16 MFMA passes (hi.hi + hi.lo + lo.hi)
HBM_PEAK_GBS = 8000.0


def _conv3x3(a):
    # (x, N, Cin, H, W, wp, Cout, xamax, xcount, wamax, addend, bias, y, stride, in_up, Hout, Wout, tile_r, tile_p, stream)
    n, ci, h, w, co, stride, in_up = a[1], a[2], a[3], a[4], a[6], a[13], a[14]
    ho, wo = a[15], a[16]
    if in_up == 2:          # data gradient of a stride-2 convolution: every stored gradient pixel meets all 9 taps once
        px = h * w
    elif stride == 2:
        px = ((h - 1) // 2 + 1) * ((w - 1) // 2 + 1)
    else:
        px = h * w
        ho, wo = h, w
    flops = 2.0 * n * co * ci * 9 * px
    byts = 4.0 * n * (ci * h * w + co * ho * wo) + (4.0 * n * co * ho * wo if a[10] else 0.0)
    return flops, byts, f"{n}x({ci}->{co})x{h}x{w}" + (" s2" if stride == 2 else "") + (" s2-dgrad" if in_up == 2 else "")


def _conv_smallcin(a):      # (x, N, Cin, H, W, w, Cout, bias, y): the stem's 3-channel stride-2 convolution, HBM-bound
    n, ci, h, w, co = a[1], a[2], a[3], a[4], a[6]
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    return 2.0 * n * co * ci * 9 * ho * wo, 4.0 * n * (ci * h * w + co * ho * wo), f"{n}x({ci}->{co})x{h}x{w} s2"


def _wgrad_smallcin(a):     # (x, N, Cin, H, W, gy, Cout, part, dw): the stem's weight gradient, HBM-bound
    n, ci, h, w, co = a[1], a[2], a[3], a[4], a[6]
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    return 2.0 * n * co * ci * 9 * ho * wo, 4.0 * n * (ci * h * w + co * ho * wo), f"{n}x({ci}->{co})x{h}x{w} s2 wgrad"


def _conv1x1(a):
    n, ci, h, w, co = a[1], a[2], a[3], a[4], a[6]
    return 2.0 * n * co * ci * h * w, 4.0 * n * (ci + co) * h * w + (4.0 * n * co * h * w if a[10] else 0.0), \
        f"{n}x({ci}->{co})x{h}x{w} 1x1"


def _wgrad3x3(a):
    # (x, dy, N, Cin, Cout, H, W, xamax, xcount, gamax, gcount, stride, part, dw, stream)
    n, ci, co, h, w, stride = a[2], a[3], a[4], a[5], a[6], a[11]
    hd, wd = ((h - 1) // 2 + 1, w // 2) if stride == 2 else (h, w)
    return 2.0 * n * co * ci * 9 * hd * wd, 4.0 * n * (ci * h * w + co * hd * wd), \
        f"{n}x({ci}->{co})x{h}x{w}" + (" s2" if stride == 2 else "")


def _wgrad1x1(a):
    n, ci, co, h, w = a[2], a[3], a[4], a[5], a[6]
    return 2.0 * n * co * ci * h * w, 4.0 * n * (ci + co) * h * w, f"{n}x({ci}->{co})x{h}x{w} 1x1"


def _gemm(a):
    m, n, k, batch = a[8], a[9], a[10], a[11]
    return 2.0 * m * n * k * batch, 4.0 * batch * (m * k + n * k + m * n), f"{batch}x[{m}x{k}].[{n}x{k}]^T"


def _gemm_ep(a):            # dcl_gemm_f16x3_ep(A, lda, akm, B, ldb, bkm, M, N, K, ...): + one more [M, N] tensor read or written
    m, n, k = a[6], a[7], a[8]
    return 2.0 * m * n * k, 4.0 * (m * k + n * k + 2.0 * m * n), f"[{m}x{k}].[{n}x{k}]^T"


def _sweep(factor):
    def f(a):
        n1, n2 = a[1], a[4]
        return factor * n1 * n2 * 256.0, 4.0 * 256 * (n1 + n2) + (4.0 * 256 * n1 if factor == 4.0 else 0.0), f"{n1}x{n2}"
    return f


def _bn_stats(a):           # (x, N, C, HW, ...): one read of x
    n, c, hw = a[1], a[2], a[3]
    return 0.0, 4.0 * n * c * hw, f"{n}x{c}x{hw}"


def _bn_apply(a):           # (x, res, part, count, eps, momentum, gamma, beta, N, C, HW, relu, y, ...): read x (+ res), write y
    n, c, hw = a[8], a[9], a[10]
    return 0.0, 4.0 * n * c * hw * (3 if a[1] else 2), f"{n}x{c}x{hw}"


def _bn_apply_parts(a):     # (x, res, part, ns, count, eps, momentum, gamma, beta, N, C, HW, relu, y, ...)
    n, c, hw = a[9], a[10], a[11]
    return 0.0, 4.0 * n * c * hw * (3 if a[1] else 2), f"{n}x{c}x{hw}"


def _bn_bwd_reduce(a):      # (dy, x, y, mean, invstd, gamma, beta, N, C, HW, relu, part): read dy, x (+ 1/32 packed mask)
    n, c, hw, relu = a[7], a[8], a[9], a[10]
    return 0.0, 4.0 * n * c * hw * (2 + (1.0 / 32 if relu == 2 else (1 if (relu == 1 and a[2]) else 0))), f"{n}x{c}x{hw}"


def _bn_bwd_apply(a):       # (dy, x, y, ..., N, C, HW, relu, dx, dres, ...): read dy, x (+ mask), write dx (+ dres)
    n, c, hw, relu = a[10], a[11], a[12], a[13]
    extra = 1.0 / 32 if relu == 2 else (1 if (relu == 1 and a[2]) else 0)
    return 0.0, 4.0 * n * c * hw * (3 + extra + (1 if a[15] else 0)), f"{n}x{c}x{hw}"


def _gather(a):             # (feat, sn, sc, sp, C, pix, pair_b, slot_pair, T, V, bank, nrm, bank_h): N rows read, f32 (+ half) rows written
    c, t, v = a[4], a[8], a[9]
    return 0.0, 4.0 * t * v * c * (3 if a[12] else 2), f"N={t * v} C={c}"


def _scatter(a):            # (slabs, nslab, bank, nrm, pix, pair_b, slot_pair, T, V, C, dfeat, ...): slabs + bank rows read, N rows scattered
    ns, t, v, c = a[1], a[7], a[8], a[9]
    return 0.0, 4.0 * t * v * (256 * (ns + 1) + c), f"N={t * v} C={c} slabs={ns}"


def _label_hist(a):         # (label, n, H, W, scale, K, ...): strided int64 read (8 B per kept pixel) + uint8 write
    n, h, w, s = a[1], a[2], a[3], a[4]
    px = n * (h // s) * (w // s)
    return 0.0, 9.0 * px, f"{n}x{h // s}x{w // s}"


def _rank_select(a):        # (lbl_s, seg_hist, n, hw, K, pair_b, pair_k, T, V, sel, pix): T*V queries, each writes 4 B and scans <= 256 B
    t, v = a[7], a[8]
    return 0.0, 8.0 * t * v, f"T={t} V={v}"


def _upsample_fwd(a):       # (x, addend, planes, h, w, H, W, ...)
    pl, h, w, H, W = a[2], a[3], a[4], a[5], a[6]
    return 0.0, 4.0 * pl * (h * w + H * W * (2 if a[1] else 1)), f"{pl}x{h}x{w}->{H}x{W}"


def _upsample_bwd(a):       # (dy, planes, h, w, H, W, ...)
    pl, h, w, H, W = a[1], a[2], a[3], a[4], a[5]
    return 0.0, 4.0 * pl * (h * w + H * W), f"{pl}x{h}x{w}<-{H}x{W}"


def _head_norm_dz(a):       # (dl, z, wt, c0, c1, N, K, C, HW, dz, amax, stream): read z and the K-channel dl, write dz
    n, k, c, hw = a[5], a[6], a[7], a[8]
    return 2.0 * n * k * c * hw, 4.0 * n * hw * (2 * c + k), f"{n}x({k}->{c})x{hw}"


def _conv3x3_pre(a):
    # (x, N, Cin, H, W, wp, Cout, xamax, xcount, wamax, pre_sc, pre_sh, bias, y, stride, tile_r, tile_p, stream): the convolution of
    # relu(bn(x)) with the map applied in the operand staging -- same work as dcl_conv3x3_f16x3 on the written tensor
    n, ci, h, w, co, stride = a[1], a[2], a[3], a[4], a[6], a[14]
    ho, wo = ((h - 1) // 2 + 1, (w - 1) // 2 + 1) if stride == 2 else (h, w)
    return 2.0 * n * co * ci * 9 * ho * wo, 4.0 * n * (ci * h * w + co * ho * wo), \
        f"{n}x({ci}->{co})x{h}x{w}" + (" s2" if stride == 2 else "") + " pre"


def _wgrad3x3_pre(a):
    # (x, dy, N, Cin, Cout, H, W, xamax, xcount, gamax, gcount, pre_sc, pre_sh, stride, part, dw, stream)
    n, ci, co, h, w, stride = a[2], a[3], a[4], a[5], a[6], a[13]
    hd, wd = ((h - 1) // 2 + 1, w // 2) if stride == 2 else (h, w)
    return 2.0 * n * co * ci * 9 * hd * wd, 4.0 * n * (ci * h * w + co * hd * wd), \
        f"{n}x({ci}->{co})x{h}x{w}" + (" s2" if stride == 2 else "") + " pre"


# entry point -> (work model, bound, fixed kernel symbol or None = ask dcl_last_kernel)
MODELS = {
    "dcl_conv3x3_f16x3": (_conv3x3, "mfma", None),
    "dcl_conv3x3_s2_smallcin": (_conv_smallcin, "hbm", "k_conv3x3_s2_smallcin"),
    "dcl_wgrad3x3_s2_smallcin": (_wgrad_smallcin, "hbm", "k_wgrad_stem"),
    "dcl_conv1x1_f16x3": (_conv1x1, "mfma", None),
    "dcl_wgrad3x3_f16x3": (_wgrad3x3, "mfma", None),
    "dcl_wgrad1x1_f16x3": (_wgrad1x1, "mfma", None),
    "dcl_gemm_f16x3": (_gemm, "mfma", "k_gemm"),
    "dcl_gemm_f16x3_ep": (_gemm_ep, "mfma", "k_gemm"),
    "dcl_gemm_f16x3_ascaled": (_gemm_ep, "mfma", "k_gemm"),
    "dcl_infonce_zsweep": (_sweep(2.0), "mfma", "k_sweep<MODE_Z>"),
    "dcl_infonce_zsweep_keep": (_sweep(2.0), "mfma", "k_sweep<MODE_Z>"),
    "dcl_infonce_bwd_streamk": (_sweep(4.0), "mfma", "k_sweep<MODE_BWD,stream-K>"),
    "dcl_infonce_bwd": (_sweep(4.0), "mfma", "k_sweep<MODE_BWD>"),
    "dcl_conv3x3_pre_f16x3": (_conv3x3_pre, "mfma", None),
    "dcl_wgrad3x3_pre_f16x3": (_wgrad3x3_pre, "mfma", None),
    "dcl_head_norm_dz": (_head_norm_dz, "hbm", "k_head_norm_dz"),
    "dcl_bn_stats_part": (_bn_stats, "hbm", "k_bn_stats"),
    "dcl_bn_stats_pre": (_bn_stats, "hbm", "k_bn_stats_pre"),
    "dcl_bn_stats_minmax_part": (_bn_stats, "hbm", "k_bn_stats_mm"),
    "dcl_bn_apply_fused": (_bn_apply, "hbm", "k_bn_apply"),
    "dcl_bn_apply_parts": (_bn_apply_parts, "hbm", "k_bn_apply"),
    "dcl_bn_bwd_reduce_part": (_bn_bwd_reduce, "hbm", "k_bn_bwd_reduce"),
    "dcl_bn_bwd_apply_fused": (_bn_bwd_apply, "hbm", "k_bn_bwd_apply"),
    "dcl_gather_normalize": (_gather, "hbm", "k_gather_normalize"),
    "dcl_normalize_bwd_scatter": (_scatter, "hbm", "k_normalize_bwd_scatter"),
    "dcl_label_hist": (_label_hist, "hbm", "k_label_hist"),
    "dcl_rank_select": (_rank_select, "hbm", "k_rank_select"),
    "dcl_upsample_bilinear_fwd": (_upsample_fwd, "hbm", "k_upsample_fwd"),
    "dcl_upsample_bilinear_bwd": (_upsample_bwd, "hbm", "k_upsample_bwd_rows"),
}


class KernelTimer:
    def __init__(self, names=None):
        self.names = list(names or MODELS)
        self.calls = []                 # (entry point, kernel symbol, flops, bytes, shape, e0, e1)
        self._orig = {}

    def __enter__(self):
        L = _lib.lib()
        L.dcl_trace_kernels(1)
        for name in self.names:
            fn = getattr(L, name)
            self._orig[name] = fn
            setattr(L, name, self._wrap(L, name, fn))
        return self

    def _wrap(self, L, name, fn):
        model, _, fixed = MODELS[name]
        calls = self.calls

        def timed(*a):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*a)
            e1.record()
            sym = fixed or (L.dcl_last_kernel() or b"?").decode()
            flops, byts, shape = model(a)
            if name == "dcl_bn_apply_fused":
                sym = f"k_bn_apply<{'true' if a[11] else 'false'},{'true' if a[1] else 'false'}>"
            elif name == "dcl_bn_apply_parts":
                sym = f"k_bn_apply<{'true' if a[12] else 'false'},{'true' if a[1] else 'false'}>"
            elif name in ("dcl_bn_bwd_reduce_part", "dcl_bn_bwd_apply_fused"):
                sym = f"{fixed}<{'true' if a[10 if name == 'dcl_bn_bwd_reduce_part' else 13] else 'false'}>"
            calls.append((name, sym, flops, byts, shape, e0, e1))
            return rc
        return timed

    def __exit__(self, *exc):
        L = _lib.lib()
        for name, fn in self._orig.items():
            setattr(L, name, fn)
        L.dcl_trace_kernels(0)
        return False

    def rows(self):
        """One row per kernel symbol, largest total time first (call after torch.cuda.synchronize())."""
        agg = {}
        for name, sym, flops, byts, shape, e0, e1 in self.calls:
            ms = e0.elapsed_time(e1)
            r = agg.setdefault(sym, {"kernel": sym, "entry": name, "bound": MODELS[name][1], "calls": 0, "total_ms": 0.0,
                                     "flops": 0.0, "bytes": 0.0, "shapes": {}})
            r["calls"] += 1
            r["total_ms"] += ms
            r["flops"] += flops
            r["bytes"] += byts
            sh = r["shapes"].setdefault(shape, [0, 0.0, flops, byts])
            sh[0] += 1
            sh[1] += ms
        out = sorted(agg.values(), key=lambda r: -r["total_ms"])
        for r in out:
            sec = r["total_ms"] * 1e-3
            if r["bound"] == "mfma":
                r["achieved"], r["peak"], r["unit"] = r["flops"] / sec / 1e12, MFMA_F16X3_PEAK_TFLOPS, "TFLOP/s"
            else:
                r["achieved"], r["peak"], r["unit"] = r["bytes"] / sec / 1e9, HBM_PEAK_GBS, "GB/s"
            r["frac"] = r["achieved"] / r["peak"]
        return out
