"""Training-runtime protocol on the GPU: the device-side loss scaler of the fused optimizers (GradScaler semantics without a host sync),
the RCCL path of the benchmark at world size 1, and the training entry point end to end."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tiny_model():
    from pixart_sigma_amd import PixArtMS
    torch.manual_seed(0)
    m = PixArtMS(depth=1, input_size=8, model_max_length=8, class_dropout_prob=0.0).cuda()
    m.prepare("cuda")
    return m


def test_loss_scaler_matches_unscaled_step_skips_on_overflow_and_grows():
    """(1) gradients x 4096 with scaler(4096) give the update of the unscaled gradients; (2) an inf in the gradient buffer skips the step
    (weights, moments and the applied-step count untouched), halves the scale and counts a skipped step; (3) after `growth_interval` clean
    steps the scale doubles - torch.cuda.amp.GradScaler.update() semantics, all on the device."""
    from pixart_sigma_amd.dp import FusedAdamW, LossScaler
    ma, mb = _tiny_model(), _tiny_model()
    g = torch.Generator(device="cuda").manual_seed(1)
    grads = [torch.randn(ma._store.total, device="cuda", generator=g) * 1e-3 for _ in range(3)]
    oa = FusedAdamW(ma, lr=1e-3, weight_decay=3e-2, eps=1e-10, max_grad_norm=0.01)
    sc = LossScaler("cuda", init_scale=4096.0, growth_interval=2)
    ob = FusedAdamW(mb, lr=1e-3, weight_decay=3e-2, eps=1e-10, max_grad_norm=0.01, scaler=sc)
    for gr in grads[:2]:
        ma._store.grad.copy_(gr)
        mb._store.grad.copy_(gr * sc.value)
        oa.step()
        ob.step()
    # The two optimizers see the same gradients up to the fp32 atomic-add order of the global-norm reduction (4096 block partials:
    # ~1e-6 relative on the norm, run to run) -> the clip coefficient -> at most lr * 1e-5 per step on a weight: atol = 3 steps of that.
    ATOL = 3 * 1e-3 * 1e-5
    assert torch.allclose(mb._store.master, ma._store.master, rtol=1e-6, atol=ATOL)
    assert torch.allclose(ob.last_norm, oa.last_norm, rtol=1e-5)                  # the reported norm is the unscaled one
    assert sc.value == 8192.0 and sc.steps_applied == 2 and sc.steps_skipped == 0     # two clean steps -> grown once
    # overflow
    before = (mb._store.master.clone(), ob.m.clone(), ob.v.clone(), mb._store.shadow.clone())
    mb._store.grad.copy_(grads[2] * sc.value)
    mb._store.grad[12345] = float("inf")
    ob.step()
    assert sc.found_inf and sc.value == 4096.0 and sc.steps_skipped == 1 and sc.steps_applied == 2
    assert all(torch.equal(a, b) for a, b in zip(before, (mb._store.master, ob.m, ob.v, mb._store.shadow)))
    # the next clean step uses bias correction step 3 = applied count, like the unscaled optimizer's third step
    ma._store.grad.copy_(grads[2])
    mb._store.grad.copy_(grads[2] * sc.value)
    oa.step()
    ob.step()
    assert not sc.found_inf and sc.steps_applied == 3
    d = (mb._store.master - ma._store.master).abs()
    tol = ATOL + 1e-6 * ma._store.master.abs()
    assert bool((d <= tol).all()), f"max |diff| {float(d.max()):.3e}, worst excess over tolerance {float((d - tol).max()):.3e}, elements over {int((d > tol).sum())} of {d.numel()}"


def test_came_skips_on_overflow():
    from pixart_sigma_amd.dp import FusedCAME, LossScaler
    m = _tiny_model()
    sc = LossScaler("cuda", init_scale=1024.0)
    opt = FusedCAME(m, lr=1e-3, scaler=sc)
    m._store.grad.normal_(std=1e-3)
    m._store.grad.mul_(sc.value)
    opt.step()
    w1 = m._store.master.clone()
    m._store.grad.fill_(float("nan"))
    opt.step()
    assert sc.found_inf and sc.value == 512.0 and torch.equal(m._store.master, w1)


def _bench(extra_env, launcher):
    env = dict(os.environ, **extra_env)
    cmd = launcher + [os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "0", "--image-size", "256", "--batch", "4",
                      "--no-cpu-baseline", "--no-kernel-roofline", "--no-torch-baseline", "--no-other-dtype", "--no-configs"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_bench_under_torchrun_with_a_real_rccl_group_matches_plain_run():
    """`python -m torch.distributed.run --nproc-per-node 1 bench.py` builds a nccl (= RCCL) process group of one rank and, with
    PXA_DP_FORCE_COLLECTIVES=1, really all-reduces every gradient bucket from the engine's hooks while backward runs - the N > 1 code path,
    as far as one GPU can execute it.  The loss after two optimizer steps must equal the run without a process group up to the run-to-run
    noise of the fp32 atomics in the reductions (measured 1e-5 relative between two identical runs)."""
    plain = _bench({}, [sys.executable])
    port = str(29500 + os.getpid() % 2000)
    pg = _bench({"PXA_DP_FORCE_COLLECTIVES": "1", "PXA_DP_TRACE": "1", "PXA_DP_TRACE_PRINT": "0"}, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                                                    "--master-addr", "127.0.0.1", "--master-port", port])
    print("\nfinal_loss plain", plain["final_loss"], "rccl world-1", pg["final_loss"])
    assert pg["process_group"] == "nccl" and plain["process_group"] is None
    assert abs(pg["final_loss"] - plain["final_loss"]) < 1e-4 * abs(plain["final_loss"])
    assert pg["n_gpus"] == 1 and pg["config"]["parallelism"] == "dp1"
    # round 6: the per-bucket trace of the last step rides in the bench line (PXA_DP_TRACE=1): final, blocks 27 .. 0, then 'cond' - every one of them launched
    # from a hook (the last from autograd's end-of-backward callback), none from finish()
    tr = pg["dp_trace"]
    names = [b["bucket"] for b in tr["buckets"]]
    assert names == ["final"] + [f"blocks.{i}" for i in reversed(range(28))] + ["cond"], names
    assert all(b["launched_from"] == "hook" for b in tr["buckets"]) and tr["exposed_ms"] >= 0.0 and tr["world"] == 1
    assert all(b["passed_ms"] >= b["ready_ms"] for b in tr["buckets"]) and "dp_trace" not in plain


def test_train_entry_point_fp16_accumulation_schedule(tmp_path):
    """train_scripts/train.py on synthetic data: fp16 operands + loss scaling (the reference's mixed_precision='fp16'), gradient accumulation 2,
    auto-lr (sqrt rule) and a 4-step warm-up; the logged learning rates follow the schedule and the loss is finite."""
    cfg = tmp_path / "cfg.py"
    cfg.write_text("image_size = 256\ntrain_batch_size = 2\nmodel_max_length = 16\nmixed_precision = 'fp16'\ngradient_accumulation_steps = 2\n"
                   "auto_lr = dict(rule='sqrt')\nlr_schedule = 'constant'\nlr_schedule_args = dict(num_warmup_steps=4)\nlog_interval = 1\n"
                   "optimizer = dict(type='AdamW', lr=2e-5, weight_decay=3e-2, eps=1e-10)\nsave_model_steps = 3\n")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train_scripts", "train.py"), str(cfg), "--synthetic", "--max-steps", "3", "--work-dir", str(tmp_path / "w")],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=dict(os.environ))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("step ")]
    print("\n" + r.stdout[-1200:])
    assert len(lines) == 3 and "operands fp16" in r.stdout and "loss_scale" in lines[0]
    base = 2e-5 * (2 * 1 * 2 / 256) ** 0.5
    lrs = [float(l.split(" lr ")[1].split()[0]) for l in lines]
    assert all(abs(lr - base * k / 4) < 1e-9 for k, lr in enumerate(lrs))           # warm-up factor k / 4 at optimizer step k
    assert all(float(l.split("loss ")[1].split()[0]) == float(l.split("loss ")[1].split()[0]) for l in lines)   # not nan
    ck = torch.load(tmp_path / "w" / "checkpoints" / "epoch_1_step_3.pth", map_location="cpu", weights_only=False)
    assert ck["lr_scheduler"]["last_step"] == 3 and "loss_scaler" in ck and ck["optimizer"]["layout"][0][0] == "x_embedder.proj.weight"
    # unknown config keys are refused, not ignored
    bad = tmp_path / "bad.py"
    bad.write_text("image_size = 256\nuse_fancy_thing = True\n")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train_scripts", "train.py"), str(bad), "--synthetic", "--max-steps", "1"], capture_output=True, text=True,
                       timeout=300, cwd=ROOT)
    assert r.returncode != 0 and "use_fancy_thing" in (r.stdout + r.stderr)


@pytest.mark.parametrize("algo,steps", [("dpm-solver", 3), ("iddpm", 4), ("sa-solver", 5)])
def test_inference_entry_point_samplers(tmp_path, algo, steps):
    """scripts/inference.py with the reference's CLI (reference scripts/inference.py:24-44, 86-133) on synthetic caption features: all three samplers run
    the HIP denoiser end to end (256px, two prompts, batch 2 -> model batch 4 with CFG) and write finite latents of the right shape."""
    txt = tmp_path / "prompts.txt"
    txt.write_text("a red cube\na blue sphere\n")
    env = {k: v for k, v in os.environ.items() if k not in ("PXA_OPERAND_DTYPE", "PXA_LIB_PATH")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "inference.py"), "--image_size", "256", "--txt_file", str(txt), "--bs", "2", "--synthetic",
                        "--sampling_algo", algo, "--step", str(steps), "--save_name", f"pytest_{algo}", "--pipeline_load_from", str(tmp_path / "none")],
                       capture_output=True, text=True, timeout=900, cwd=str(tmp_path), env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert f"in {steps} steps" in r.stdout
    lat = torch.load(tmp_path / "output" / f"pytest_{algo}" / "latents_0.pt", map_location="cpu")
    assert tuple(lat.shape) == (2, 4, 32, 32) and torch.isfinite(lat).all() and lat.std() > 0.1
