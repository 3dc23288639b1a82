"""A/B kernel builds: recompile ONE source of the library with extra -D flags and link it with the cached objects of the others.
    python tools/build_variant.py <name> <source.hip>[,<source2.hip>...] [-DFLAG ...]   ->  pixart_sigma_amd/variants/lib_<name>.so
Run a benchmark against it with PXA_LIB_PATH=pixart_sigma_amd/variants/lib_<name>.so (pixart_sigma_amd/lib.py)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pixart_sigma_amd import build as B

name, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
OPERAND = os.environ.get("VARIANT_OPERAND", "bf16")          # f16: the variant of the fp16-operand library (run it with PXA_OPERAND_DTYPE=f16)
flags = [*B.VARIANTS[OPERAND][1], *flags]
B.build(variants=(OPERAND,))
out_dir = os.path.join(B.HERE, "variants")
os.makedirs(out_dir, exist_ok=True)
objs = []
for s in sorted(f for f in os.listdir(B.CSRC) if f.endswith(".hip")):
    path = os.path.join(B.CSRC, s)
    if s in [os.path.basename(x) for x in src.split(",")]:
        obj = os.path.join(out_dir, f"{s[:-4]}.{name}.o")
        subprocess.run([B._hipcc(), *B.FLAGS, *B.PER_FILE_FLAGS.get(s, []), *flags, "-I", B.INCLUDE, "-c", path, "-o", obj], check=True)
    else:
        obj = os.path.join(B.OBJ, f"{s[:-4]}.{OPERAND}.{B._digest(path, B.VARIANTS[OPERAND][1])}.o")
    objs.append(obj)
lib = os.path.join(out_dir, f"lib_{name}.so")
subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib], check=True)
print(lib)
