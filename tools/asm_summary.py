"""Per-kernel register / scratch / LDS summary of a device assembly file (make -C csrc <file>.s):
    python tools/asm_summary.py eccv2022-..._amd/csrc/dcl_sweep.s"""
import re
import sys

txt = open(sys.argv[1]).read()
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    name, body = m.group(1), m.group(2)
    def g(k):
        r = re.search(r"\.amdhsa_" + k + r"\s+(\S+)", body)
        return r.group(1) if r else "?"
    short = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", name)[:60]
    print(f"{short:62s} vgpr_next={g('next_free_vgpr'):>4s} accum_off={g('accum_offset'):>4s} "
          f"scratch={g('private_segment_fixed_size'):>5s} lds={g('group_segment_fixed_size'):>6s}")
