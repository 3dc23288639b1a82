"""Per-GEMM table of the in-step counter dumps (tools/pmc_step.sh -> gpurun_out/<tag>_pmc_step_{1,2,3}.csv): the launches of one kernel instance are told
apart by their position in the training step's fixed kernel sequence (engine.block_fwd / block_bwd).  Durations are those under the counter pass.
Usage: python tools/pmc_step_table.py gpurun_out/r5_01 [steps_in_dump=3]"""
import csv, collections, sys
tag = sys.argv[1]
NSTEP = int(sys.argv[2]) if len(sys.argv) > 2 else 3
D, DFF, R, LT = 1152, 4608, 65536, 4800
# instance -> cyclic labels of its launches inside a block (forward order, then backward order); (label, m, n, k, compulsory operand + output bytes)
CYC = {
    "gemm_pers_kernel<0, 0, 2, false>": ["qkv", "attn.proj", "q_linear", "cross.proj", "fc2"],
    "gemm_pers_kernel<0, 1, 0, false>": ["fc1+GELU"],
    "gemm_pers_kernel<0, 0, 0, false>": ["kv_linear"],
    "gemm_pers_kernel<1, 2, 0, false>": ["fc2 dX x GELU'"],
    "gemm_pers_kernel<1, 0, 2, false>": ["fc1 dX", "cross.proj dX", "q_linear dX", "attn.proj dX", "qkv dX"],
    "gemm_pers_kernel<2, 0, 1, false>": ["fc1 dW", "cross.proj dW", "q_linear dW", "attn.proj dW", "qkv dW"],
    "gemm_pers_kernel<2, 0, 0, false>": ["fc2 dW"],
}
SHAPE = {"qkv": (R, 3 * D, D), "attn.proj": (R, D, D), "q_linear": (R, D, D), "cross.proj": (R, D, D), "fc2": (R, D, DFF), "fc1+GELU": (R, DFF, D),
         "kv_linear": (LT, 2 * D, D), "fc2 dX x GELU'": (R, DFF, D), "fc1 dX": (R, D, DFF), "cross.proj dX": (R, D, D), "q_linear dX": (R, D, D),
         "attn.proj dX": (R, D, D), "qkv dX": (R, D, 3 * D), "fc1 dW": (DFF, D, R), "cross.proj dW": (D, D, R), "q_linear dW": (D, D, R),
         "attn.proj dW": (D, D, R), "qkv dW": (3 * D, D, R), "fc2 dW": (D, DFF, R)}
data = collections.defaultdict(lambda: collections.defaultdict(dict))     # kernel -> order -> counter -> value (+ 'dur')
for i in (1, 2, 3):
    try:
        rows = list(csv.DictReader(open(f"{tag}_pmc_step_{i}.csv")))
    except FileNotFoundError:
        continue
    per = collections.defaultdict(lambda: collections.defaultdict(dict))
    for r in rows:
        k = r["kernel"].replace("void ", "").split("(")[0]
        per[k][int(r["order"])][r["counter"]] = float(r["value"])
        per[k][int(r["order"])]["dur_us_pass%d" % i] = float(r["dur_ns"]) / 1e3
    for k, d in per.items():
        orders = sorted(d)
        n = len(orders) // NSTEP
        for j, o in enumerate(orders[n:]):                    # drop the warm-up step; index inside the step
            data[k][(j % n)].setdefault("_n", 0)
            for c, v in d[o].items():
                data[k][j % n][c] = data[k][j % n].get(c, 0.0) + v / (NSTEP - 1)
print(f"{'GEMM (in the step)':18s} {'us':>7s} {'TFLOP/s':>8s} {'fetch MB':>9s} {'x A+B':>6s} {'L2 hit':>7s} {'MFMA busy':>9s} {'eff. GHz':>8s} {'wait':>6s}")
for k, labels in CYC.items():
    if k not in data:
        continue
    pos = sorted(data[k])
    # the caption MLP's launches use the same instances at the very start (forward) / end (backward) of a step: drop positions beyond 28 x len(labels) from the
    # side they sit on (forward instances: the first ones; backward instances: the last ones)
    extra = len(pos) - 28 * len(labels)
    fwd = k.startswith("gemm_pers_kernel<0")
    pos = pos[extra:] if fwd else pos[:len(pos) - extra] if extra else pos
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for j, ps in enumerate(pos):
        lab = labels[j % len(labels)]
        for c, v in data[k][ps].items():
            agg[lab][c] += v / 28
    for lab in labels:
        a = agg[lab]
        m, n, kk = SHAPE[lab]
        us = a.get("dur_us_pass1", 0.0)
        fl = 2.0 * m * n * kk
        comp = (m * kk + n * kk) * 2 / 1e6
        fetch = a.get("FETCH_SIZE", 0.0) * 1e3 * 2 / 1e6          # KB -> MB, x 2: the gfx950 correction (MI355X_MICROARCH.md, HBM section)
        hit = a.get("TCC_HIT_sum", 0.0) / max(1.0, a.get("TCC_HIT_sum", 0.0) + a.get("TCC_MISS_sum", 0.0))
        busy = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(1.0, 32 * a.get("SQ_BUSY_CYCLES", 0.0)) if "SQ_BUSY_CYCLES" in a else 0.0
        us3 = a.get("dur_us_pass3", 0.0)
        ghz = a.get("GRBM_GUI_ACTIVE", 0.0) / 8 / max(1e-9, us3 * 1e3) if us3 else 0.0
        wait = a.get("SQ_WAIT_ANY", 0.0) / max(1.0, a.get("SQ_WAVE_CYCLES", 0.0))
        print(f"{lab:18s} {us:7.1f} {fl / us / 1e6 if us else 0:8.0f} {fetch:9.0f} {fetch / comp if comp else 0:6.2f} {hit:7.3f} {busy:9.3f} {ghz:8.2f} {wait:6.2f}   {k}")
