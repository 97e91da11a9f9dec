"""Register / scratch / LDS figures of every kernel in the built library, read from the code objects inside
libdcl_hip.so (no recompilation):

    python tools/scratch_survey.py            # kernels with scratch > 0, then a one-line total
    python tools/scratch_survey.py --all      # every kernel

The library embeds one clang offload bundle per translation unit (section .hip_fatbin); each holds a gfx950 ELF whose
NT_AMDGPU_METADATA note lists, per kernel, .private_segment_fixed_size (scratch bytes per lane), .vgpr_count,
.agpr_count, .sgpr_count and .group_segment_fixed_size.  `llvm-readelf --notes` prints that note as YAML."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
CXXFILT = "c++filt"


def code_objects(path):
    blob = open(path, "rb").read()
    # device ELFs: ELFCLASS64, little endian, e_machine = EM_AMDGPU (224) at offset 18
    offs = [m.start() for m in re.finditer(b"\x7fELF\x02\x01\x01", blob)]
    out = []
    for i, o in enumerate(offs):
        if int.from_bytes(blob[o + 18:o + 20], "little") != 224:
            continue
        end = offs[i + 1] if i + 1 < len(offs) else len(blob)
        out.append(blob[o:end])
    return out


def kernels(elf_bytes):
    with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
        f.write(elf_bytes)
        name = f.name
    try:
        txt = subprocess.run([READELF, "--notes", name], capture_output=True, text=True).stdout
    finally:
        os.unlink(name)
    res = []
    for blk in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
        blk = ".agpr_count:" + blk
        def g(k):
            m = re.search(r"\." + k + r":\s+(\S+)", blk)
            return m.group(1) if m else "?"
        res.append(dict(name=g("name"), vgpr=g("vgpr_count"), agpr=g("agpr_count"), sgpr=g("sgpr_count"),
                        scratch=g("private_segment_fixed_size"), lds=g("group_segment_fixed_size")))
    return res


def demangle(names):
    p = subprocess.run([CXXFILT], input="\n".join(names), capture_output=True, text=True)
    return [re.sub(r"\(anonymous namespace\)::", "", s) for s in p.stdout.splitlines()]


def survey(lib_path=None):
    if lib_path is None:
        from mscs_amd import _lib
        lib_path = _lib.LIB_PATH
    ks = []
    for co in code_objects(lib_path):
        ks += kernels(co)
    for k, d in zip(ks, demangle([k["name"] for k in ks])):
        k["demangled"] = re.sub(r"\(.*", "", d)
    return ks


if __name__ == "__main__":
    ks = survey()
    show_all = "--all" in sys.argv
    spill = [k for k in ks if k["scratch"] not in ("0", "?")]
    for k in (ks if show_all else spill):
        print(f"{k['demangled'][:70]:72s} vgpr={k['vgpr']:>4s} agpr={k['agpr']:>4s} sgpr={k['sgpr']:>4s} "
              f"scratch={k['scratch']:>5s} lds={k['lds']:>6s}")
    print(f"{len(ks)} kernels, {len(spill)} with scratch")
