"""Per-layer table of one AutoencoderKL.decode from a launch-ordered kernel trace (tools/export_trace.py): time, algorithmic TFLOP/s and GB/s, which
roofline bounds the layer and the fraction of it reached.
The schedule is re-derived from the public SD / SDXL VAE config (block_out (128, 256, 512, 512), 2 + 1 layers per decoder block) in the order
pixart_sigma_amd/vae/autoencoder_kl.py:decode issues its GEMMs; every GEMM launch of the trace consumes one schedule entry, and the row kernels in front of it
(GroupNorm finalize / apply + SiLU + upsample, im2col) are charged to the same layer, the ones behind it (residual add) as well.
Usage: python tools/vae_layer_table.py TRACE.csv BATCH [PX]"""
import csv
import sys

MFMA_PEAK, HBM_PEAK = 2.5e15, 8.0e12


def decode_schedule(B, px=512, chans=(128, 256, 512, 512), layers=2, latent=4, out_ch=3, phases=False, conv_out_gemm=True):
    """[(label, group, flops, bytes)] per GEMM launch.  bytes = one read of the layer's input + one write of its output in 16-bit (what a fully fused
    layer would move); flops = 2 m n k at the unpadded sizes."""
    s, h = [], px // 8
    rev = list(reversed(chans))

    def conv(label, group, H, ci, co, taps=9, out_bytes=2, up=1):
        Ho = H * up
        s.append((label, group, 2.0 * B * Ho * Ho * ci * co * taps, B * (H * H * ci * 2 + Ho * Ho * co * out_bytes), f"{Ho}x{Ho} {ci}->{co}" + (" 1x1" if taps == 1 else "")))

    def resnet(name, group, H, ci, co):
        conv(name + ".conv1", group, H, ci, co)
        if ci != co:
            conv(name + ".shortcut", group, H, ci, co, taps=1)
        conv(name + ".conv2", group, H, co, co)

    conv("post_quant_conv", "stem", h, latent, latent, taps=1)
    conv("conv_in", "stem", h, latent, rev[0])
    c = rev[0]
    resnet("mid.res0", f"mid {h}x{h} C{c}", h, c, c)
    conv("mid.attn.qkv", "mid attention", h, c, 3 * c, taps=1)
    for i in range(B):
        n = h * h
        s.append(("mid.attn.scores", "mid attention", 2.0 * n * n * c, n * c * 4 + n * n * 4, f"{n}x{n}x{c}"))
        s.append(("mid.attn.pv", "mid attention", 2.0 * n * n * c, n * n * 4 + n * c * 4, f"{n}x{n}x{c}"))
    conv("mid.attn.out", "mid attention", h, c, c, taps=1)
    resnet("mid.res1", f"mid {h}x{h} C{c}", h, c, c)
    H = h
    for bi, co in enumerate(rev):
        for r in range(layers + 1):
            resnet(f"up{bi}.res{r}", f"up{bi} {H}x{H} C{co}", H, c if r == 0 else co, co)
        c = co
        if bi < len(rev) - 1:
            # since round 6 the upsampling convolution runs as four low-res 2 x 2 phase convolutions (16 of the 36 products): four launches that share one
            # line of the table; the FLOPs charged stay the ALGORITHMIC ones of the reference's 3 x 3 convolution over the upsampled grid
            n0 = len(s)
            conv(f"up{bi}.upsample", f"up{bi} upsample conv -> {2 * H}x{2 * H} C{co}", H, co, co, up=2)
            if phases:
                lab, grp, fl, by, shp = s.pop(n0)
                s.extend([(lab, grp, fl / 4, by / 4, shp + " (4 phases)")] * 4)
            H *= 2
    if conv_out_gemm:                                    # (round 6: a direct kernel, no GEMM launch - its time lands behind the last GEMM)
        conv("conv_out", "conv_out", H, c, out_ch, out_bytes=4)
    return s


def main():
    path, B = sys.argv[1], int(sys.argv[2])
    px = int(sys.argv[3]) if len(sys.argv) > 3 else 512
    rows = [r for r in csv.DictReader(l for l in open(path) if not l.startswith("#"))]
    starts = [i for i, r in enumerate(rows) if "nchw_to_grid" in r["kernel"]]      # decode() begins with the latent's layout change: keep the LAST decode only
    if starts:
        rows = rows[starts[-1]:]
    n_gemm = sum(1 for r in rows if "gemm" in r["kernel"] and "splitk" not in r["kernel"])
    direct_out = any("conv3x3_small_out" in r["kernel"] for r in rows)
    sched = decode_schedule(B, px, conv_out_gemm=not direct_out)
    if len(sched) != n_gemm:                            # the phase-decomposed upsampling convolutions: 3 more launches each
        sched = decode_schedule(B, px, phases=True, conv_out_gemm=not direct_out)
    segs, cur = [], []
    for r in rows:
        cur.append(r)
        k = r["kernel"]
        if "gemm" in k and "splitk" not in k:
            segs.append(cur)
            cur = []
    tail = cur
    # trailing row kernels of a layer (vae_add, split-K reduction) belong to the layer whose GEMM they follow
    for i in range(1, len(segs)):
        while segs[i] and any(t in segs[i][0]["kernel"] for t in ("vae_add", "splitk")):
            segs[i - 1].append(segs[i].pop(0))
    if len(segs) != len(sched):
        print(f"WARNING: {len(segs)} GEMM launches in the trace, {len(sched)} in the schedule - table by launch order only")
    lines = {}
    order = []
    for i, seg in enumerate(segs):
        label, group, fl, by, shape = sched[i] if i < len(sched) else (f"gemm{i}", "unmatched", 0.0, 0.0, "?")
        key = label if not label.startswith("mid.attn.") else "mid.attn (qkv, 64 x (scores, softmax, pv), out)"
        if key not in lines:
            lines[key] = dict(group=group, shape=shape, total=0.0, gemm=0.0, rows_=0.0, flops=0.0, bytes=0.0, n=0)
            order.append(key)
        L = lines[key]
        for r in seg:
            d = float(r["duration_us"])
            L["total"] += d
            if "gemm" in r["kernel"] and "splitk" not in r["kernel"]:
                L["gemm"] += d
            else:
                L["rows_"] += d
        L["flops"] += fl
        L["bytes"] += by
        L["n"] += 1
    t_tail = sum(float(r["duration_us"]) for r in tail)
    tot = sum(L["total"] for L in lines.values()) + t_tail
    span = float(rows[-1]["start_us"]) + float(rows[-1]["duration_us"]) - float(rows[0]["start_us"])
    print(f"AutoencoderKL.decode, batch {B} x {px}px: {tot / 1e3:.2f} ms of kernel time ({span / 1e3:.2f} ms first start to last end), {sum(x[2] for x in sched) / 1e12:.1f} TFLOP algorithmic")
    print(f"{'layer':34s} {'shape':18s} {'ms':>7s} {'GEMM ms':>8s} {'row-kernel ms':>13s} {'TFLOP/s':>8s} {'alg GB/s':>9s} {'bound':>5s} {'frac':>5s}")
    for key in order:
        L = lines[key]
        t = L["total"] * 1e-6
        tf, gb = L["flops"] / t / 1e12, L["bytes"] / t / 1e9
        t_m, t_h = L["flops"] / MFMA_PEAK, L["bytes"] / HBM_PEAK
        bound, frac = ("mfma", t_m / t) if t_m >= t_h else ("hbm", t_h / t)
        print(f"{key[:34]:34s} {L['shape']:18s} {L['total'] / 1e3:7.2f} {L['gemm'] / 1e3:8.2f} {L['rows_'] / 1e3:13.2f} {tf:8.0f} {gb:9.0f} {bound:>5s} {frac:5.2f}")
    print(f"{'(behind the last GEMM: conv_out as a direct kernel with its GroupNorm finalize / cast)' if direct_out else '(behind the last GEMM: crop / permute / cast)':53s} {t_tail / 1e3:7.2f}")
    by_group = {}
    for key in order:
        L = lines[key]
        g = by_group.setdefault(L["group"], [0.0, 0.0])
        g[0] += L["total"]
        g[1] += L["flops"]
    print("\nby resolution level:")
    for g, (t, fl) in by_group.items():
        print(f"  {g:44s} {t / 1e3:7.2f} ms  {100 * t / tot:5.1f} %  {fl / (t * 1e-6) / 1e12:6.0f} TFLOP/s")


if __name__ == "__main__":
    main()
