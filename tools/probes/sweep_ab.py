"""Forward (Z) and backward InfoNCE sweeps of the benchmark term (N = 9 804, C = 256, f16x3) timed with HIP events, for the
shipped library and for variant builds of csrc/dcl_sweep.hip loaded into the SAME process (alternating runs: clock drift hits
all alike).

    bash tools/probes/sweep_ab.sh build NAME=-DFLAG ...     # here: tools/probes/variants/libdcl_<NAME>.so
    gpurun -- python tools/probes/sweep_ab.py [NAME ...]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import mscs_amd  # noqa: F401,E402
from mscs_amd import _lib  # noqa: E402
from mscs_amd.losses import DenseContrastiveLossV2_ms  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    args = type("A", (), dict(batch=12, height=512, width=1024, scales=3, classes=20, labels="iid", feature_layout="nchw"))()
    mod = DenseContrastiveLossV2_ms(bench.loss_config(3, True))
    label, feats = bench.synth_loss_inputs(args, dev, 0)
    torch.manual_seed(0)
    mod(label, feats).backward()
    torch.cuda.synchronize()
    st = mod.last_state
    t = st.terms[0]
    A = st.scales[t.a]
    N, V = A.plan.N, A.plan.V
    Npad = A.bank.shape[0]
    libs = {"shipped": _lib.lib()}
    for name in sys.argv[1:]:
        path = os.path.join(ROOT, "tools", "probes", "variants", f"libdcl_{name}.so")
        L = ctypes.CDLL(path)
        for fn, argtypes in _lib.SIGNATURES.items():
            f = getattr(L, fn)
            f.argtypes = argtypes
            f.restype = ctypes.c_int
        libs[name] = L
    p = _lib.ptr
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ns = int(libs["shipped"].dcl_suggest_nsplit(N, N))
    ld = (int(t.max_span) + 3) // 4 * 4 if getattr(t, "max_span", 0) else 2048
    zpart = torch.empty((ns, Npad), device=dev)
    spos = torch.empty((Npad, ld), device=dev)
    stat = torch.empty((Npad + 1, 4), device=dev)
    ref = {}

    def fwd(L, keep):
        if keep:
            return L.dcl_infonce_zsweep_keep(p(A.bank), N, V, p(A.bank), N, p(t.rng_lo), p(t.rng_hi), 1.0 / t.tau, ns, p(zpart),
                                             p(A.bank_h), p(A.bank_h), p(spos), ld, stream)
        return L.dcl_infonce_zsweep(p(A.bank), N, V, p(A.bank), N, p(t.rng_lo), p(t.rng_hi), 1.0 / t.tau, ns, p(zpart),
                                    p(A.bank_h), p(A.bank_h), stream)

    def timeit(fn, iters=30):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3

    Lb = libs["shipped"]
    _lib.check(Lb.dcl_infonce_prep_stats(p(t.Z), p(t.W), p(t.rng_lo), p(t.rng_hi), None, N, V, 1, 1.0, 1.0 / t.tau, None, p(stat), stream), "prep")
    G = int(Lb.dcl_infonce_bwd_streamk_workgroups(N, N))
    nsl = int(Lb.dcl_infonce_bwd_streamk_slabs(N, N))
    dout = torch.empty((nsl, Npad, 256), device=dev)
    ws = torch.empty((G, 128, 256), device=dev)
    flags = {name: torch.zeros(G + 1, dtype=torch.int32, device=dev) for name in libs}

    def bwd(name, L):
        return L.dcl_infonce_bwd_streamk(p(A.bank), N, V, p(A.bank), N, p(t.rng_lo), p(t.rng_hi), 1.0 / t.tau, 1, 1, 1, p(stat), p(stat),
                                         p(dout), p(ws), p(flags[name]), p(A.bank_h), p(A.bank_h), stream)

    print(f"N = {N}, V = {V}, nsplit = {ns}, kept-positives row = {ld} floats, stream-K workgroups {G}, slabs {nsl}")
    if os.environ.get("SWEEP_AB_SCAN"):
        # launch time against the number of contrast columns (fixed cost per launch vs cost per chunk), shipped library
        for n2 in (N, 3 * N // 4, N // 2, N // 4, N // 8):
            n2 = n2 // 32 * 32
            fl = torch.zeros(G + 1, dtype=torch.int32, device=dev)
            us = timeit(lambda: Lb.dcl_infonce_bwd_streamk(p(A.bank), N, V, p(A.bank), n2, p(t.rng_lo), p(t.rng_hi), 1.0 / t.tau, 1, 1, 1,
                                                          p(stat), p(stat), p(dout), p(ws), p(fl), p(A.bank_h), p(A.bank_h), stream), 20)
            uz = timeit(lambda: Lb.dcl_infonce_zsweep(p(A.bank), N, V, p(A.bank), n2, p(t.rng_lo), p(t.rng_hi), 1.0 / t.tau, ns, p(zpart),
                                                     p(A.bank_h), p(A.bank_h), stream), 20)
            print(f"  N2 = {n2:5d}: bwd {us:7.1f} us   zsweep {uz:7.1f} us")
    if os.environ.get("SWEEP_AB_XMAP"):
        for rep in range(3):
            for on in (0, 1):
                Lb.dcl_infonce_set_zsweep_xmap(on)
                for keep in (False, True):
                    us = timeit(lambda: fwd(Lb, keep))
                    zs = zpart.sum(0)
                    ref.setdefault((keep, "zx"), zs.clone())
                    same = torch.equal(zs, ref[(keep, "zx")])
                    print(f"  rep {rep} xmap {on} zsweep{'_keep' if keep else '     '}: {us:7.1f} us = {2.0 * N * N * 256 / us / 1e6 / 833.3:.3f} of the roofline; Z bitwise equal: {same}")
        Lb.dcl_infonce_set_zsweep_xmap(1)
        return
    for rep in range(3):
        for name, L in libs.items():
            us = timeit(lambda: bwd(name, L), 20)
            flops = 4.0 * N * N * 256
            print(f"  rep {rep} {name:12s} bwd stream-K : {us:7.1f} us = {flops / us / 1e6:6.1f} TFLOP/s ({flops / us / 1e6 / 833.3:.3f} of the f16x3 roofline)")
        if "bwd" in os.environ.get("SWEEP_AB_ONLY", ""):
            continue
        for name, L in libs.items():
            for keep in (False, True):
                us = timeit(lambda: fwd(L, keep))
                zs = zpart.sum(0)
                if (keep, "z") not in ref:
                    ref[(keep, "z")] = zs.clone()
                err = ((zs - ref[(keep, "z")]).abs().max() / ref[(keep, "z")].abs().max()).item()
                flops = 2.0 * N * N * 256
                print(f"  rep {rep} {name:12s} zsweep{'_keep' if keep else '     '}: {us:7.1f} us = {flops / us / 1e6:6.1f} TFLOP/s "
                      f"({flops / us / 1e6 / 833.3:.3f} of the f16x3 roofline)   max rel dZ vs first = {err:.1e}")


if __name__ == "__main__":
    main()
