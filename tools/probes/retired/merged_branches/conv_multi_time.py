"""One launch for the 3x3 convolutions of several HRNet branches (dcl_conv3x3_f16x3_multi) against single launches.

    python tools/probes/conv_multi_time.py [--batch 12]

Times, with HIP events: (a) the branch-1..3 convolutions of a stage-4 depth as three launches on one stream, (b) as one
merged launch (tile_p variants), (c) the same two next to branch 0's convolution on a second stream -- and checks that the
merged results are bitwise those of the single launches with the same tile."""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mscs_amd  # noqa: F401,E402
from mscs_amd import _lib  # noqa: E402
from mscs_amd.models import ops  # noqa: E402
from mscs_amd.models.amax import amax_of  # noqa: E402


class Job(ctypes.Structure):
    _fields_ = [("x", ctypes.c_void_p), ("wp", ctypes.c_void_p), ("xamax", ctypes.c_void_p), ("wamax", ctypes.c_void_p),
                ("addend", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("y", ctypes.c_void_p),
                ("N", ctypes.c_int), ("Cin", ctypes.c_int), ("Cout", ctypes.c_int), ("H", ctypes.c_int), ("W", ctypes.c_int),
                ("xcount", ctypes.c_int), ("tile_p", ctypes.c_int), ("reserved", ctypes.c_int)]


def timeit(fn, iters=50):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=12)
    a = ap.parse_args()
    dev = torch.device("cuda")
    L = _lib.lib()
    n = a.batch
    shapes = [(48, 128, 256), (96, 64, 128), (192, 32, 64), (384, 16, 32)]
    gen = torch.Generator(device=dev).manual_seed(0)
    T = []
    for c, h, w in shapes:
        x = torch.randn(n, c, h, w, device=dev, generator=gen).relu_()
        wt = torch.randn(c, c, 3, 3, device=dev, generator=gen) * (2.0 / (9 * c)) ** 0.5
        xa, wa = amax_of(x), amax_of(wt)
        wp = ops.conv3x3_pack(wt, wa)
        T.append(dict(x=x, wp=wp, xa=xa, wa=wa, c=c, h=h, w=w, y=torch.empty(n, c, h, w, device=dev), y2=torch.empty(n, c, h, w, device=dev)))

    def single(t, p=0, out="y"):
        ops.conv3x3_launch(t["x"], t["wp"], t["c"], t["xa"], t["wa"], t[out], tile_r=3 if p else 0, tile_p=p)

    def jobs_of(ts, ps, out="y2"):
        arr = (Job * len(ts))()
        for k, (t, p) in enumerate(zip(ts, ps)):
            arr[k] = Job(t["x"].data_ptr(), t["wp"].data_ptr(), t["xa"].data_ptr(), t["wa"].data_ptr(), None, None, t[out].data_ptr(),
                         n, t["c"], t["c"], t["h"], t["w"], t["xa"].numel(), p, 0)
        return arr

    def multi(arr, stream=None):
        st = _lib.stream_ptr(dev)
        _lib.check(L.dcl_conv3x3_f16x3_multi(ctypes.cast(arr, ctypes.c_void_p), len(arr), st), "multi")

    coarse = T[1:]
    # bitwise check
    for ps in ((4, 4, 4), (4, 4, 2), (2, 2, 2)):
        for t, p in zip(coarse, ps):
            single(t, p)
        arr = jobs_of(coarse, ps)
        multi(arr)
        torch.cuda.synchronize()
        ok = all(torch.equal(t["y"], t["y2"]) for t in coarse)
        print(f"tile_p {ps}: merged launch bitwise equal to single launches: {ok}")
    print("single launches (automatic tiles), us:", [round(timeit(lambda t=t: single(t)), 1) for t in T])
    print("three coarse launches in a row, us:", round(timeit(lambda: [single(t) for t in coarse]), 1))
    for ps in ((4, 4, 4), (4, 4, 2), (4, 2, 2), (2, 2, 2), (0, 0, 0)):
        arr = jobs_of(coarse, ps)
        print(f"merged b1+b2+b3 tile_p {ps}: {timeit(lambda: multi(arr)):.1f} us")
    arr2 = jobs_of(coarse[:2], (4, 4))
    print(f"two launches b1, b2: {timeit(lambda: [single(t) for t in coarse[:2]]):.1f} us; merged b1+b2 (4,4): {timeit(lambda: multi(arr2)):.1f} us")
    arr2 = jobs_of(coarse[:2], (4, 2))
    print(f"merged b1+b2 (4,2): {timeit(lambda: multi(arr2)):.1f} us")
    # next to branch 0 on a second stream
    s1 = torch.cuda.Stream()
    s = [torch.cuda.Stream() for _ in range(3)]

    def four_streams():
        cur = torch.cuda.current_stream()
        for k in range(3):
            s[k].wait_stream(cur)
            with torch.cuda.stream(s[k]):
                single(coarse[k])
        single(T[0])
        for k in range(3):
            cur.wait_stream(s[k])

    def two_streams(arr):
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur)
        with torch.cuda.stream(s1):
            multi(arr)
        single(T[0])
        cur.wait_stream(s1)

    print(f"all four branches, one launch each in a row: {timeit(lambda: [single(t) for t in T]):.1f} us")
    print(f"all four branches on four streams: {timeit(four_streams):.1f} us")
    for ps in ((4, 4, 4), (4, 4, 2), (2, 2, 2)):
        arr = jobs_of(coarse, ps)
        print(f"branch 0 + merged coarse {ps} on two streams: {timeit(lambda: two_streams(arr)):.1f} us")


if __name__ == "__main__":
    main()
