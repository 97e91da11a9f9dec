#!/usr/bin/env python3
"""Alternating A/B timing of dcl_gemm_f16x3 builds (tools/probes/gemm_ab.sh): every tools/probes/_build/gemm_*.so is
loaded into this process; per shape and GEMM kind the variants run round-robin, REPS rounds, median per variant."""
import ctypes
import glob
import os
import statistics
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
SIG = [vp, i64, i32, i64, vp, i64, i32, i64, i32, i32, i32, i32, vp, i32, vp, i32, vp, vp, i64, i64, i32, vp, i32, vp, vp, vp]


def main():
    shapes = [tuple(int(v) for v in s.split(",")) for s in sys.argv[1:]] or \
        [(25600, 768, 3072), (25600, 3072, 768), (6400, 1536, 4608), (102400, 384, 1152), (409600, 192, 576)]
    libs = {}
    for f in sorted(glob.glob(os.path.join(HERE, "_build", "gemm_*.so"))):
        l = ctypes.CDLL(f)
        l.dcl_gemm_f16x3.argtypes = SIG
        l.dcl_gemm_suggest_splitk.argtypes = [i32] * 4
        libs[os.path.basename(f)[5:-3]] = l
    names = list(libs)
    dev = "cuda"
    st = torch.cuda.current_stream().cuda_stream
    print("shape,kind," + ",".join(f"{n}_ms" for n in names) + "," + ",".join(f"{n}_tflops" for n in names))
    for (m, k, n) in shapes:
        x = torch.randn(m, k, device=dev)
        w = torch.randn(n, k, device=dev) * 0.05
        gy = torch.randn(m, n, device=dev) * 1e-3
        am = {id(t): t.abs().max().reshape(1) for t in (x, w, gy)}
        y, gx, gw = torch.empty(m, n, device=dev), torch.empty(m, k, device=dev), torch.empty(n, k, device=dev)
        ws = torch.empty(64 << 20, device=dev)
        kinds = {
            # A, lda, akm, B, ldb, bkm, M, N, K, C, ldc
            "fwd": (x, k, 1, w, k, 1, m, n, k, y, n),
            "dgrad": (gy, n, 1, w, k, 0, m, k, n, gx, k),
            "wgrad": (gy, n, 0, x, k, 0, n, k, m, gw, k),
        }
        for kind, (A, lda, akm, B, ldb, bkm, M, N, K, C, ldc) in kinds.items():
            def run(l):
                sk = l.dcl_gemm_suggest_splitk(M, N, K, 1)
                rc = l.dcl_gemm_f16x3(A.data_ptr(), lda, akm, 0, B.data_ptr(), ldb, bkm, 0, M, N, K, 1, am[id(A)].data_ptr(), 1,
                                      am[id(B)].data_ptr(), 1, None, C.data_ptr(), ldc, 0, 0, None, sk, ws.data_ptr(), None, st)
                assert rc == 0, rc
            times = {nm: [] for nm in names}
            for nm in names:
                run(libs[nm])
            torch.cuda.synchronize()
            for rep in range(7):
                for nm in names:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(5):
                        run(libs[nm])
                    e1.record()
                    torch.cuda.synchronize()
                    times[nm].append(e0.elapsed_time(e1) / 5)
            med = [statistics.median(times[nm]) for nm in names]
            fl = 2.0 * M * N * K
            print(f"{m}x{k}x{n},{kind}," + ",".join(f"{t:.4f}" for t in med) + "," + ",".join(f"{fl / t / 1e9:.0f}" for t in med),
                  flush=True)


if __name__ == "__main__":
    main()
