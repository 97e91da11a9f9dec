"""Element-wise kernels must return the same bits whatever runs beside them.

On MI355X a packed-FP32 instruction whose op_sel takes the high half of src1 for the low result (v_pk_fma_f32 ... op_sel:[0,1,0]:
the compiler's broadcast of the second of two packed scalars) returns a wrong low result in lanes 48-63 while another wave on the
same SIMD issues MFMAs with operands still arriving in its registers (stand-alone reproducer: tools/probes/pk_mfma_hazard.hip).  With
the library built WITH packed FP32, 15-25 % of the norm backwards below differed from the one computed alone (up to 1e-2 of max) as
soon as a matrix kernel of another HIP stream shared the compute units -- the situation of every HRNet exchange module (one stream
per branch, reference models/HRNet.py:263-267).  The library is therefore built without packed FP32 (csrc/Makefile NOPK); this file
holds the library-level reproducer as a regression test and checks the build."""
import os
import re
import shutil
import subprocess

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "eccv2022-multi-scale-and-cross-scale-contrastive-segmentation_amd", "csrc")


FAULTY_FORM = re.compile(r"\bv_pk_[a-z0-9_]+ .*op_sel:\[[01],1")      # a packed instruction whose LOW result takes src1's HIGH half


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs the ROCm toolchain (hipcc)")
def test_library_is_built_without_the_faulty_packed_form(tmp_path):
    """Device assembly with the Makefile's own flags: the norm kernels (the ones the finding was made on) and the resize kernels
    (explicit two-float vector arithmetic in the source) hold no packed-FP32 instruction at all; the two translation units that
    keep packed FP32 (csrc/Makefile PACKED: the sweeps, the GEMM) hold none with op_sel[1] set -- the build itself fails otherwise."""
    mk = open(os.path.join(CSRC, "Makefile")).read()
    assert "-packed-fp32-ops" in mk and "$(NOPK)" in mk
    assert 'v_pk_(fma|mul|add)_f32' in mk, "the generic rule checks every NOPK unit's device assembly at build time"
    packed = re.search(r"^PACKED = (.*)$", mk, re.M).group(1).split()
    assert set(packed) <= {"dcl_sweep", "dcl_gemm"} and "op_sel:" in mk, "new PACKED members need the evidence of DESIGN.md section 7"
    for src in ("dcl_bn", "dcl_resize"):
        assert src not in packed
        # (the Makefile's own flags; the assembly goes to the test's directory, nothing is written into the source tree)
        flags = subprocess.run(["make", "-C", CSRC, "-s", "--no-print-directory", "print-cxxflags"], capture_output=True, text=True)
        assert flags.returncode == 0 and "-packed-fp32-ops" in flags.stdout, flags.stderr[-2000:]
        dst = os.path.join(str(tmp_path), f"{src}.s")
        out = subprocess.run(["hipcc", *flags.stdout.split(), "-S", "--cuda-device-only", os.path.join(CSRC, f"{src}.hip"), "-o", dst],
                             capture_output=True, text=True)
        assert out.returncode == 0, out.stderr[-2000:]
        asm = open(dst).read()
        assert "s_endpgm" in asm
        assert not re.search(r"\bv_pk_(fma|mul|add)_f32\b", asm), f"{src}: packed FP32 instructions in the device code"
    # the guard's pattern finds the form in a packed build of the norm kernels (8 instructions: the broadcasts of mean_gx)
    out = subprocess.run(["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-S", "--cuda-device-only",
                          os.path.join(CSRC, "dcl_bn.hip"), "-o", "-"], capture_output=True, text=True)
    assert out.returncode == 0 and FAULTY_FORM.search(out.stdout), "the packed build of dcl_bn.hip no longer shows the faulty form?"
    # ... and not in what the build wrote for the PACKED units (present after `make`; the build stops if it ever does)
    for src in packed:
        path = os.path.join(CSRC, "build", f"{src}.pk.s")
        if os.path.exists(path):
            asm = open(path).read()
            assert "v_pk_" in asm and not FAULTY_FORM.search(asm), src


@pytest.mark.gpu
def test_norm_backward_is_bitwise_stable_beside_matrix_kernels():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import mscs_amd  # noqa: F401
    from mscs_amd.models import fused_bn, ops
    from mscs_amd.models.amax import amax_of
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    bn = fused_bn.FusedBatchNorm2d(48).to(dev).train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
    x = torch.randn(2, 48, 64, 96, device=dev, requires_grad=True)
    res = torch.randn(2, 48, 64, 96, device=dev)
    dy = torch.randn(2, 48, 64, 96, device=dev) * 1e-4
    y = bn(x, residual=res, relu=True)
    ref = torch.autograd.grad(y, x, dy, retain_graph=True)[0].clone()
    # the neighbours: a convolution at two workgroups per CU, a weight gradient, a GEMM -- all leave room for other waves on their SIMDs
    c48 = torch.randn(12, 48, 128, 256, device=dev).relu_()
    w48 = torch.randn(48, 48, 3, 3, device=dev) * 0.05
    sx, sw = amax_of(c48), amax_of(w48)
    wp = ops.conv3x3_pack(w48, sw)
    o48 = torch.empty_like(c48)
    g48 = torch.randn(12, 48, 128, 256, device=dev)
    lx = torch.randn(16384, 384, device=dev)
    lw = torch.randn(1536, 384, device=dev) * 0.05
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    for beside in (lambda: ops.conv3x3_launch(c48, wp, 48, sx, sw, o48), lambda: ops.conv3x3_wgrad(c48, g48),
                   lambda: ops.linear_f16x3(lx, lw)):
        for _ in range(20):
            with torch.cuda.stream(sb):
                for _ in range(8):
                    beside()
            outs = []
            with torch.cuda.stream(sa):
                for _ in range(12):
                    outs.append(torch.autograd.grad(y, x, dy, retain_graph=True)[0])
            torch.cuda.synchronize()
            for o in outs:
                assert torch.equal(o, ref)
