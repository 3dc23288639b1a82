"""Static check of hand-counted LDS waits (csrc/attn.hip, attn_bwd_dkv2_kernel<1>): the loop's LDS reads are inline asm, so the compiler inserts no
s_waitcnt for them and a miscounted lgkmcnt would read a fragment register before its data has landed - a timing-dependent wrong result that a
parity test can miss.  This replays the EMITTED ISA of the kernel's innermost loop: lgkmcnt retires LDS reads in issue order, so
    ds_read*            -> push its destination registers on a FIFO
    s_waitcnt lgkmcnt(N)-> pop from the front until N entries are left
    anything else       -> must not name (read OR write) a register that is still in the FIFO
and the FIFO must be empty at the loop's back edge (nothing in flight across basic blocks, where the compiler may insert copies).
Usage: python tools/check_lds_waits.py [kernel-name-substring]   (compiles csrc/attn.hip to assembly with the library's flags; CPU only)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def regs(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def check(asm_text, kernel_sub):
    lines = asm_text.splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + kernel_sub + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end]
    heads = [i for i, l in enumerate(body) if "Loop Header" in l]
    assert heads, "no loop found"
    h = heads[-1]
    label = body[h].split(":")[0].strip()
    back = max(i for i in range(h, len(body)) if re.search(r"s_cbranch\w+\s+" + re.escape(label) + r"\b", body[i]) or re.search(r"s_branch\s+" + re.escape(label) + r"\b", body[i]))
    fifo, n_reads, n_waits, n_mfma, errors = [], 0, 0, 0, []
    for rnd in range(2):                      # twice around: the state at the back edge feeds the next iteration
        for i in range(h, back + 1):
            l = body[i].split(";")[0].strip()
            if not l or l.endswith(":") or l.startswith("."):
                continue
            op = l.split()[0]
            if op.startswith("ds_read"):
                dst = l.split()[1].rstrip(",")
                for pend in fifo:
                    if pend & regs(l):
                        errors.append(f"line {i}: `{l}` touches registers of a read still in flight")
                fifo.append(regs(dst))
                n_reads += rnd == 0
                continue
            if op == "s_waitcnt":
                m = re.search(r"lgkmcnt\((\d+)\)", l)
                if m:
                    n = int(m.group(1))
                    while len(fifo) > n:
                        fifo.pop(0)
                    n_waits += rnd == 0
                continue
            if op in ("s_load_dword", "s_load_dwordx2", "s_load_dwordx4", "s_buffer_load_dword"):
                errors.append(f"line {i}: scalar memory read inside the loop (returns out of order in lgkmcnt)")
            n_mfma += (rnd == 0) and op.startswith("v_mfma")
            touched = regs(l)
            for pend in fifo:
                if pend & touched:
                    errors.append(f"line {i}: `{l}` uses v{sorted(pend & touched)} before the LDS read that fills it was waited for")
        if fifo:
            errors.append(f"{len(fifo)} LDS reads still in flight at the loop's back edge")
    return dict(reads=n_reads, waits=n_waits, mfma=n_mfma, errors=errors, lines=back - h + 1)


def compile_asm():
    from pixart_sigma_amd import build as B
    src = os.path.join(B.CSRC, "attn.hip")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "attn.s")
        subprocess.run([B._hipcc(), *B.FLAGS, *B.PER_FILE_FLAGS.get("attn.hip", []), "-I", B.INCLUDE, "-S", "--cuda-device-only", src, "-o", out],
                       check=True, capture_output=True)
        return open(out).read()


if __name__ == "__main__":
    sub = sys.argv[1] if len(sys.argv) > 1 else "attn_bwd_dkv2_kernelILi1E"
    r = check(compile_asm(), sub)
    print({k: v for k, v in r.items() if k != "errors"})
    for e in r["errors"][:20]:
        print("ERROR", e)
    sys.exit(1 if r["errors"] else 0)
