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
    """registers named in `tok`: arch VGPRs as n, accumulator VGPRs (a0 ...) as 1000 + n"""
    out = set()
    for m in re.finditer(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b", tok):
        if m.group(1):
            base = 1000 if m.group(1) == "a" else 0
            out.update(range(base + int(m.group(2)), base + int(m.group(3)) + 1))
        else:
            out.add((1000 if m.group(4) == "a" else 0) + int(m.group(5)))
    return out


def check(asm_text, kernel_sub, inflight_at_back_edge=False):
    """inflight_at_back_edge: the loop body opens with its own lgkmcnt(0) (attn_fwd4_kernel: the K row reads of the next tile are waited for in
    front of the barrier), so reads may cross the back edge; the second replay round then checks that wait."""
    lines = asm_text.splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + kernel_sub + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end]
    heads = [i for i, l in enumerate(body) if "Loop Header" in l]
    assert heads, "no loop found"
    h = heads[-1]
    label = body[h].split(":")[0].strip()
    # the loop's text: the header block and every labelled block the compiler marks "in Loop: Header=<label>" - a rotated loop keeps its latch block
    # in FRONT of the header (attn_fwd4_kernel), so replay order = header ... last block behind it, then the blocks in front
    tag = "Header=" + label.lstrip(".L")
    lab = [i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)] + [len(body)]
    seq_after, seq_before = [], []
    for a, b in zip(lab, lab[1:]):
        if a == h or tag in body[a]:
            (seq_after if a >= h else seq_before).append(range(a, b))
    order = [i for r in seq_after + seq_before for i in r]
    fifo, n_reads, n_waits, n_mfma, errors = [], 0, 0, 0, []
    for rnd in range(2):                      # twice around: the state at the back edge feeds the next iteration
        for i in order:
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
        if fifo and not inflight_at_back_edge:
            errors.append(f"{len(fifo)} LDS reads still in flight at the loop's back edge")
    return dict(reads=n_reads, waits=n_waits, mfma=n_mfma, errors=errors, lines=len(order))


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
    r = check(compile_asm(), sub, inflight_at_back_edge=any(k in sub for k in ("fwd4", "dkv4", "dkv5", "dq4")))
    print({k: v for k, v in r.items() if k != "errors"})
    for e in r["errors"][:20]:
        print("ERROR", e)
    sys.exit(1 if r["errors"] else 0)
