"""Static checks on the EMITTED gfx950 ISA of the hand-scheduled kernels (hipcc cross-compiles without a GPU; ~10 s):
* hand-counted LDS waits (tools/check_lds_waits.py): every register an inline-asm ds_read fills is waited for before its first use;
* the one-wave-per-SIMD forward kernel keeps its promises: no scratch, no v_accvgpr traffic on the tile loop's hot path (the compiler's own split of a
  512-register kernel cost 177 spills and ~440 moves per tile before the accumulator half was pinned through asm constraints), one barrier per tile;
* M0 (the LDS-DMA destination, written inside common.h lds_dma16 / attn.hip dma_pair without a clobber the compiler would honour) has no other user
  in any kernel of attn.hip / gemm.hip (a compiler-generated M0 use between the statements would be corrupted silently)."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    from pixart_sigma_amd import build as B
    out = {}
    td = tmp_path_factory.mktemp("isa")
    for src in ("attn.hip", "gemm.hip", "gemm_nt4.hip"):
        for tag, extra in (("bf16", []), ("f16", ["-DPXA_OPERAND_F16"])):
            if src == "gemm.hip" and tag == "f16":
                continue
            o = str(td / f"{src}.{tag}.s")
            subprocess.run([B._hipcc(), *B.FLAGS, *B.PER_FILE_FLAGS.get(src, []), *extra, "-I", B.INCLUDE, "-S", "--cuda-device-only", os.path.join(B.CSRC, src), "-o", o],
                           check=True, capture_output=True)
            out[(src, tag)] = open(o).read()
    return out


@pytest.mark.parametrize("kernel,tag", [("attn_bwd_dkv2_kernelILi1E", "bf16"), ("attn_bwd_dq2_kernel", "bf16"), ("attn_fwd4_kernel", "bf16"), ("attn_fwd4_kernel", "f16"),
                                        ("attn_bwd_dkv4_kernel", "bf16"), ("attn_bwd_dkv4_kernel", "f16"), ("attn_bwd_dq4_kernel", "bf16"), ("attn_bwd_dq4_kernel", "f16"),
                                        ("attn_bwd_dkv5_kernel", "bf16"), ("attn_bwd_dkv5_kernel", "f16")])
def test_hand_counted_lds_waits(asm, kernel, tag):
    import check_lds_waits as C
    r = C.check(asm[("attn.hip", tag)], kernel, inflight_at_back_edge=kernel.endswith(("4_kernel", "5_kernel")))
    assert not r["errors"], r["errors"][:5]
    assert r["reads"] > 0 and r["waits"] > 0 and r["mfma"] > 0


def test_lds_wait_checker_catches_a_wrong_count(asm):
    """mutation: relax one counted wait of the forward kernel's second-product loop by one - the replay must object"""
    import check_lds_waits as C
    text = asm[("attn.hip", "f16")]
    i = text.index("attn_fwd4_kernel")
    m = re.search(r"s_waitcnt lgkmcnt\(([3-9])\)", text[text.index("Loop Header", i):])
    j = text.index("Loop Header", i) + m.start()
    bad = text[:j] + f"s_waitcnt lgkmcnt({int(m.group(1)) + 3})" + text[j + len(m.group(0)):]
    assert C.check(bad, "attn_fwd4_kernel", inflight_at_back_edge=True)["errors"]


@pytest.mark.parametrize("tag", ["bf16", "f16"])
def test_fwd4_register_ownership(asm, tag):
    text = asm[("attn.hip", tag)]
    name = re.search(r"^(_Z\w*attn_fwd4_kernel\w*):", text, re.M).group(1)
    meta = text[text.index(".amdhsa_kernel " + name):]
    meta = meta[:meta.index(".end_amdhsa_kernel")]
    assert re.search(r"\.amdhsa_private_segment_fixed_size\s+0\b", meta), "scratch in attn_fwd4_kernel"
    body = text[text.index("\n" + name + ":"):]
    body = body[:body.index(".Lfunc_end")]
    assert "scratch_" not in body
    lines = body.split("\n")
    h = max(i for i, l in enumerate(lines) if "Loop Header" in l)
    # hot path of one tile body = from a body's barrier up to the branch that skips the slow path
    bars = [i for i in range(h, len(lines)) if re.match(r"\s+s_barrier", lines[i])]
    assert len(bars) == 4, "one barrier per tile body, four bodies (4-slot rings)"
    for b0 in bars:
        end = next(i for i in range(b0, len(lines)) if re.match(r"\s+s_cbranch_vccz", lines[i]))
        hot = [l for l in lines[b0:end] if re.match(r"\s+[a-z]", l)]
        ops = [l.split()[0] for l in hot]
        assert not any("accvgpr" in o for o in ops), "register-file traffic on the hot path"
        assert sum(o.startswith("v_mfma_f32_32x32x16") for o in ops) == 20 and sum(o.startswith("v_mfma_f32_16x16x32") for o in ops) == 40
        assert sum(o.startswith("global_load_lds") for o in ops) == 6 and sum(o == "s_barrier" for o in ops) == 1


@pytest.mark.parametrize("kernel,n32,n16", [("attn_bwd_dkv4_kernel", 88, 0), ("attn_bwd_dkv5_kernel", 40, 80), ("attn_bwd_dq4_kernel", 40, 40)])
@pytest.mark.parametrize("tag", ["bf16", "f16"])
def test_one_wave_backward_kernels_keep_the_tile_loop_clean(asm, kernel, n32, n16, tag):
    """The round-4 backward kernels: no scratch, and per 64-row tile of the loop exactly the contracted MFMAs, ONE barrier, no register-file traffic."""
    text = asm[("attn.hip", tag)]
    name = re.search(r"^(_Z\w*" + kernel + r"\w*):", text, re.M).group(1)
    body = text[text.index("\n" + name + ":"):]
    body = body[:body.index(".Lfunc_end")]
    assert "scratch_" not in body
    lines = body.split("\n")
    h = max(i for i, l in enumerate(lines) if "Loop Header" in l)
    end = next(i for i in range(h + 1, len(lines)) if re.match(r"^\.LBB", lines[i]))
    ops = [l.split()[0] for l in lines[h:end] if re.match(r"\s+[a-z]", l)]
    assert not any("accvgpr" in o for o in ops)
    assert sum(o.startswith("v_mfma_f32_32x32x16") for o in ops) == n32 and sum(o.startswith("v_mfma_f32_16x16x32") for o in ops) == n16
    assert sum(o == "s_barrier" for o in ops) == 1


@pytest.mark.parametrize("tnb,bias", [(8, 1), (8, 0), (4, 1), (4, 0)])
@pytest.mark.parametrize("tag", ["bf16", "f16"])
def test_nt4_gemm_register_ownership_and_waits(asm, tnb, bias, tag):
    """gemm_nt4_kernel (one wave per SIMD NT GEMM): all 8 x TNB accumulator tiles in the accumulator half and NO scratch anywhere (an asynchronous load whose
    destination the allocator spills is a wrong result); the steady-state loop of four k-units is exactly 4 x 8 x TNB MFMAs, 4 x (8 + TNB) fragment reads,
    2 line pairs of LDS-DMA pieces and two barriers, with no register-file traffic; and the replay of its LDS waits finds every fragment waited for."""
    import check_lds_waits as C
    text = asm[("gemm_nt4.hip", tag)]
    name = re.search(r"^(_Z\w*gemm_nt4_kernelILi%dELb%dE\w*):" % (tnb, bias), text, re.M).group(1)
    body = text[text.index("\n" + name + ":"):]
    body = body[:body.index(".Lfunc_end")]
    assert "scratch_" not in body
    meta = text[text.index(".amdhsa_kernel " + name):]
    meta = meta[:meta.index(".end_amdhsa_kernel")]
    assert int(re.search(r"\.amdhsa_accum_offset (\d+)", meta).group(1)) <= 256
    lines = body.split("\n")
    h = max(i for i, l in enumerate(lines) if "Inner Loop Header" in l)
    end = next(i for i in range(h + 1, len(lines)) if re.search(r"s_cbranch_scc", lines[i]))
    ops = [l.split()[0] for l in lines[h:end] if re.match(r"\s+[a-z]", l)]
    assert not any("accvgpr" in o or o.startswith("v_mov") for o in ops)
    assert sum(o.startswith("v_mfma_f32_16x16x32") for o in ops) == 4 * 8 * tnb
    assert sum(o == "ds_read_b128" for o in ops) == 4 * (8 + tnb)
    assert sum(o.startswith("global_load_lds") for o in ops) == 4 * (4 + tnb // 2)
    assert sum(o == "s_barrier" for o in ops) == 4
    r = C.check(text, "gemm_nt4_kernelILi%dELb%dE" % (tnb, bias), inflight_at_back_edge=True)
    assert not r["errors"], r["errors"][:5]


def test_m0_has_no_other_user(asm):
    for key, text in asm.items():
        for l in text.split("\n"):
            c = l.split(";")[0]
            if re.search(r"\bm0\b", c):
                assert re.match(r"\s+s_(mov_b32|add_u32) m0,", c), (key, l.strip())
            assert not re.match(r"\s+(s_movrel|v_movrel|s_sendmsg|ds_gws|v_interp)", c), (key, l.strip())


def test_row_kernels_keep_their_register_budget(tmp_path):
    """Round 5: a third register accumulator set took ln_mod_bwd_kernel<9> from 284 to 334 registers and every call of it from 204 to 345 us (+8 ms per training step),
    unnoticed by an A/B whose two sides ran the same kernel.  The instance without the fused bias gradient must stay the round-4 kernel (284 registers incl. the
    accumulator-half spill space), the one with it may add its LDS addressing only; gate_bwd_kernel<9> leaves room for two waves per SIMD; nothing spills to scratch."""
    from pixart_sigma_amd import build as B
    o = str(tmp_path / "norm.s")
    subprocess.run([B._hipcc(), *B.FLAGS, "-DPXA_OPERAND_F16", "-I", B.INCLUDE, "-S", "--cuda-device-only", os.path.join(B.CSRC, "norm.hip"), "-o", o], check=True, capture_output=True)
    text = open(o).read()
    meta = {}
    for blk in re.split(r"\n  - \.agpr_count", text[text.find("amdhsa.kernels"):])[1:]:
        g = lambda k: re.search(r"\." + k + r":\s+(\S+)", blk).group(1)
        meta[g("name")] = (int(g("vgpr_count")), int(g("private_segment_fixed_size")))
    def of(sub):
        hits = [v for k, v in meta.items() if sub in k]
        assert len(hits) == 1, (sub, [k for k in meta if sub in k])
        return hits[0]
    plain, fused, gate = of("ln_mod_bwd_kernelILi9ELb0E"), of("ln_mod_bwd_kernelILi9ELb1E"), of("gate_bwd_kernelILi9E")
    assert plain[0] <= 288 and fused[0] <= plain[0] + 24, (plain, fused)
    assert gate[0] <= 256, gate
    assert plain[1] == 0 and fused[1] == 0 and gate[1] == 0, (plain, fused, gate)


def test_persistent_gemm_instances_use_no_scratch(asm):
    """Round 6: the PAIR && SEG instances of gemm_pers_kernel (the VAE's 128-channel convolutions as 512 x 128 paired items) first compiled with one lambda NOT
    inlined - the closure (kernel arguments, item state, the DMA pointer arrays) went to 816 bytes of scratch per lane and the kernel ran 16x slower with correct
    results (profiles/r6_03_*).  No instance of the persistent kernel may make a call or hold more than a few spilled registers in its private segment."""
    text = asm[("gemm.hip", "bf16")]
    names = re.findall(r"^\s+\.amdhsa_kernel (_Z\w*gemm_pers_kernel\w*)", text, re.M)
    assert len(names) >= 20
    for name in names:
        meta = text[text.index(".amdhsa_kernel " + name):]
        meta = meta[:meta.index(".end_amdhsa_kernel")]
        m = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", meta)
        assert m and int(m.group(1)) <= 64, (name, m and m.group(1))       # (the run-time-epilogue NN flavour <1, 3, 0> spills 16 bytes: a few registers, not a closure)
        body = text[text.index("\n" + name + ":"):]
        body = body[:body.index(".end_amdhsa_kernel")]
        assert "s_swappc_b64" not in body, name
