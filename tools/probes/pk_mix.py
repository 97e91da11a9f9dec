"""Which side of the packed-FP32 / MFMA pair is special?  Mixes the library's kernels with the synthetic ones of
tools/probes/pk_mfma_hazard.hip (built as tools/probes/variants/libpkh.so; the LIBRARY must be the packed build:
DCL_LIB_PATH=tools/probes/variants/libdcl_packed.so):

    real victim (norm backward)      beside synthetic MFMA loops of every shape
    synthetic victims (k_pk, k_pk_bn) beside the library's matrix kernels (48-channel convolution, GEMM, weight gradient)"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import mscs_amd  # noqa: F401,E402
from mscs_amd import _lib  # noqa: E402
from mscs_amd.models import fused_bn, ops  # noqa: E402
from mscs_amd.models.amax import amax_of  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
P = ctypes.CDLL(os.path.join(ROOT, "tools/probes/variants/libpkh.so"))
P.pkh_launch_mfma.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
P.pkh_launch_victim.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
print("library:", os.environ.get("DCL_LIB_PATH", "product (no packed FP32: expect zeros)"))

# real victim
bn = fused_bn.FusedBatchNorm2d(48).to(dev).train()
x = torch.randn(2, 48, 64, 96, device=dev, requires_grad=True)
res = torch.randn(2, 48, 64, 96, device=dev)
dy = torch.randn(2, 48, 64, 96, device=dev) * 1e-4
y = bn(x, residual=res, relu=True)
ref = torch.autograd.grad(y, x, dy, retain_graph=True)[0].clone()
# real aggressors
c48 = torch.randn(12, 48, 128, 256, device=dev).relu_()
w48 = torch.randn(48, 48, 3, 3, device=dev) * 0.05
sx, sw = amax_of(c48), amax_of(w48)
wp = ops.conv3x3_pack(w48, sw)
o48 = torch.empty_like(c48)
g48 = torch.randn(12, 48, 128, 256, device=dev)
lx = torch.randn(16384, 384, device=dev)
lw = torch.randn(1536, 384, device=dev) * 0.05
# synthetic buffers
mf_out = torch.empty(512 * 256, device=dev)
pk_in = torch.rand(512 * 256 * 4, device=dev) * 2 - 1
bad = torch.zeros(128, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
kinds = ["VALU v_fma_f32", "v_mfma_f32_16x16x32_f16", "v_mfma_f32_32x32x16_f16", "v_mfma_f32_16x16x16_f16", "v_mfma_f32_32x32x8_f16"]

print("-- real victim (norm backward, this library) beside synthetic loops")
for kind in range(5):
    nbad = total = 0
    for it in range(40):
        with torch.cuda.stream(sb):
            for _ in range(4):
                P.pkh_launch_mfma(kind, 512, 3000 if kind in (2, 4) else 6000, mf_out.data_ptr(), sb.cuda_stream)
        outs = []
        with torch.cuda.stream(sa):
            for _ in range(12):
                outs.append(torch.autograd.grad(y, x, dy, retain_graph=True)[0])
        torch.cuda.synchronize()
        for o in outs:
            total += 1
            nbad += 0 if torch.equal(o, ref) else 1
    print(f"   beside {kinds[kind]:26s}: {nbad} of {total} norm backwards differ", flush=True)

if "stage" in sys.argv:
    names = ["1 v_pk_add x - s[m]", "2 v_pk_add g - mm", "3 v_pk_mul s[is] * t", "4 v_pk_fma -t mm.y + w", "5 v_pk_mul k * t"]
    fn = lambda: ops.conv3x3_launch(c48, wp, 48, sx, sw, o48)
    for rep in range(2):
        bad.zero_()
        torch.cuda.synchronize()
        for it in range(30):
            with torch.cuda.stream(sb):
                for _ in range(8):
                    fn()
            with torch.cuda.stream(sa):
                for _ in range(6):
                    P.pkh_launch_victim(7, 512, 2000, pk_in.data_ptr(), bad.data_ptr(), sa.cuda_stream)
            torch.cuda.synchronize()
        b = bad.cpu().tolist()
        print("first differing stage beside the 48-channel convolution:", {names[j]: (b[2 * j], b[2 * j + 1]) for j in range(5)}, "(low half, high half)")
    sys.exit(0)
if "mixaggr" in sys.argv:
    P.pkh_launch_mix.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    for kind, kname in enumerate(["v_fma_mixlo/hi_f16 (plain)", "the library's split2 sequence (v_fma_mix with op_sel)", "v_pk_fma_f32 op_sel:[0,1,0] (another wave's)"]):
        for mf in (None, 1, 2):
            bad.zero_()
            torch.cuda.synchronize()
            for it in range(30):
                with torch.cuda.stream(sb):
                    for _ in range(4):
                        P.pkh_launch_mix(kind, 512, 4000, mf_out.data_ptr(), sb.cuda_stream)
                        if mf is not None:
                            P.pkh_launch_mfma(mf, 512, 3000, mf_out.data_ptr(), sb.cuda_stream)
                with torch.cuda.stream(sa):
                    for _ in range(6):
                        P.pkh_launch_victim(11, 512, 2000, pk_in.data_ptr(), bad.data_ptr(), sa.cuda_stream)
                torch.cuda.synchronize()
            b = bad.cpu()
            print(f"   victim v_pk_fma_f32 op_sel:[0,1,0] beside {kname} {'+ ' + kinds[mf] + ' kernels' if mf is not None else ''}: low half {int(b[:64].sum())}, high half {int(b[64:].sum())}, lanes 48-63 {int(b[48:64].sum() + b[112:128].sum())}", flush=True)
    sys.exit(0)
if "pipe" in sys.argv:
    P.pkh_launch_pipe.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    rnd = torch.randn(65536, device=dev)
    for mode, mname in enumerate(["B fragments arriving from LDS under the MFMAs", "B fragments arriving from global memory under the MFMAs", "both (+ A from global memory)"]):
        for blocks in (256, 512):
            bad.zero_()
            torch.cuda.synchronize()
            for it in range(30):
                with torch.cuda.stream(sb):
                    for _ in range(4):
                        P.pkh_launch_pipe(mode, blocks, 6000, mf_out.data_ptr(), rnd.data_ptr(), sb.cuda_stream)
                with torch.cuda.stream(sa):
                    for _ in range(6):
                        P.pkh_launch_victim(11, 512, 2000, pk_in.data_ptr(), bad.data_ptr(), sa.cuda_stream)
                torch.cuda.synchronize()
            b = bad.cpu()
            print(f"   victim v_pk_fma_f32 op_sel:[0,1,0] beside an MFMA loop with {mname}, {blocks} workgroups: low half {int(b[:64].sum())}, high half {int(b[64:].sum())}, lanes 48-63 {int(b[48:64].sum() + b[112:128].sum())}", flush=True)
    sys.exit(0)
if "fat" in sys.argv:
    P.pkh_launch_fat.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    rnd = torch.randn(65536, device=dev)
    for nacc in (4, 14):
        for blocks in (256, 512):
            bad.zero_()
            torch.cuda.synchronize()
            for it in range(30):
                with torch.cuda.stream(sb):
                    for _ in range(4):
                        P.pkh_launch_fat(nacc, blocks, 4000, mf_out.data_ptr(), rnd.data_ptr(), sb.cuda_stream)
                with torch.cuda.stream(sa):
                    for _ in range(6):
                        P.pkh_launch_victim(11, 512, 2000, pk_in.data_ptr(), bad.data_ptr(), sa.cuda_stream)
                torch.cuda.synchronize()
            b = bad.cpu()
            print(f"   victim v_pk_fma_f32 op_sel:[0,1,0] beside a {nacc}-tile MFMA loop on RANDOM operands ({16 * nacc}+ registers per lane), {blocks} workgroups: low half {int(b[:64].sum())}, high half {int(b[64:].sum())}, lanes 48-63 {int(b[48:64].sum() + b[112:128].sum())}", flush=True)
    sys.exit(0)
if "one" in sys.argv:          # victim: v_pk_fma_f32 op_sel:[0,1,0]; aggressor: the 48-channel convolution of the loaded library (probe builds)
    fn = lambda: ops.conv3x3_launch(c48, wp, 48, sx, sw, o48)
    bad.zero_()
    torch.cuda.synchronize()
    for it in range(30):
        with torch.cuda.stream(sb):
            for _ in range(8):
                fn()
        with torch.cuda.stream(sa):
            for _ in range(6):
                P.pkh_launch_victim(11, 512, 2000, pk_in.data_ptr(), bad.data_ptr(), sa.cuda_stream)
        torch.cuda.synchronize()
    b = bad.cpu()
    print(f"   victim v_pk_fma_f32 op_sel:[0,1,0] beside the 48-channel convolution of {os.path.basename(os.environ.get('DCL_LIB_PATH', 'product'))}: low half {int(b[:64].sum())}, high half {int(b[64:].sum())}", flush=True)
    sys.exit(0)
if "forms" in sys.argv:
    forms = ["plain", "op_sel:[0,1,0]", "neg_lo/hi:[1,0,0]", "op_sel:[0,1,0] + neg (the norm kernel's)", "op_sel_hi:[1,0,1]", "op_sel:[1,0,0]",
             "(v_pk_mul_f32) op_sel:[0,1]", "(v_pk_add_f32) op_sel:[0,1]", "op_sel:[0,0,1]"]
    fn = lambda: ops.conv3x3_launch(c48, wp, 48, sx, sw, o48)
    for f, fname in enumerate(forms):
        bad.zero_()
        torch.cuda.synchronize()
        for it in range(30):
            with torch.cuda.stream(sb):
                for _ in range(8):
                    fn()
            with torch.cuda.stream(sa):
                for _ in range(6):
                    P.pkh_launch_victim(10 + f, 512, 2000, pk_in.data_ptr(), bad.data_ptr(), sa.cuda_stream)
            torch.cuda.synchronize()
        b = bad.cpu()
        q = [int(b[16 * i:16 * i + 16].sum() + b[64 + 16 * i:64 + 16 * i + 16].sum()) for i in range(4)]
        print(f"   v_pk_fma_f32 {fname:42s} beside the 48-channel convolution: low half {int(b[:64].sum()):9d}, high half {int(b[64:].sum()):9d}; lanes 0-15 {q[0]}, 16-31 {q[1]}, 32-47 {q[2]}, 48-63 {q[3]}", flush=True)
    sys.exit(0)
print("-- synthetic victims beside the library's matrix kernels")
aggr = {"convolution 48 ch (32x32x16)": lambda: ops.conv3x3_launch(c48, wp, 48, sx, sw, o48),
        "GEMM 16384 x 384 -> 1536": lambda: ops.linear_f16x3(lx, lw),
        "weight gradient 48 ch (16x16x32)": lambda: ops.conv3x3_wgrad(c48, g48)}
for name, fn in aggr.items():
    for op, oname in enumerate(["v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "norm sequence", "norm seq., no LDS", "LDS broadcast only",
                                "norm seq., no loads"]):
        bad.zero_()
        torch.cuda.synchronize()
        for it in range(30):
            with torch.cuda.stream(sb):
                for _ in range(8):
                    fn()
            with torch.cuda.stream(sa):
                for _ in range(6):
                    P.pkh_launch_victim(op, 512, 2000, pk_in.data_ptr(), bad.data_ptr(), sa.cuda_stream)
            torch.cuda.synchronize()
        b = bad.cpu()
        q = [int(b[16 * i:16 * i + 16].sum() + b[64 + 16 * i:64 + 16 * i + 16].sum()) for i in range(4)]
        print(f"   {oname:20s} beside {name:34s}: {int(b.sum()):9d} mismatching results (low half {int(b[:64].sum())}, high half {int(b[64:].sum())}; "
              f"lanes 0-15 {q[0]}, 16-31 {q[1]}, 32-47 {q[2]}, 48-63 {q[3]})", flush=True)
