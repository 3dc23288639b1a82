"""Per-kernel resource summary of a gfx950 assembly listing (hipcc -S --cuda-device-only): registers, spills, LDS, and instruction counts of
interest.  Usage: python tools/isa_report.py file.s [name-regex]"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
meta = {}
for blk in re.split(r"\n  - \.agpr_count", s[s.find("amdhsa.kernels"):])[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    meta[g("name")] = dict(agpr=blk.split("\n")[0].strip(": "), vgpr=g("vgpr_count"), sgpr=g("sgpr_count"), spill=g("vgpr_spill_count"), lds=g("group_segment_fixed_size"), scratch=g("private_segment_fixed_size"))
for name, m in meta.items():
    if pat and not pat.search(name):
        continue
    i = s.find("\n" + name + ":")
    j = s.find(".Lfunc_end", i)
    c = Counter(re.findall(r"^\s+([a-z_0-9]+)", s[i:j], re.M))
    keys = [k for k in c if "mfma" in k or k.startswith(("scratch_", "ds_read", "ds_write", "global_load_lds", "v_permlane", "v_accvgpr", "v_exp", "s_barrier"))]
    print(name[:110])
    print("   ", m, {k: c[k] for k in sorted(keys)})
