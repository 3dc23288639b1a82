# GPU-box session r04a (round 3): patch_embed_bwd with 8 gradient rows in flight per thread: parity + time at the headline shape
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "patch_embed" 2>&1 | tail -2 > gpurun_out/r04a_patch_embed.txt
python - >> gpurun_out/r04a_patch_embed.txt 2>&1 <<'PY'
import torch, sys
sys.path.insert(0, '.')
from pixart_sigma_amd import ops
from tools.kbench import timed
B, Hl = 16, 128
x = torch.randn(B, 4, Hl, Hl, device="cuda"); dtok = torch.randn(B * 4096, 1152, device="cuda")
dw = torch.zeros(1152, 16, device="cuda"); db = torch.zeros(1152, device="cuda")
t = timed(lambda: ops.patch_embed_bwd(x, dtok, dw, db), iters=20, warm=2)
print(f"patch_embed_bwd B16 128x128 latent: {t*1e6:.1f} us  {dtok.numel()*4/t/1e12:.2f} TB/s")
PY
cat gpurun_out/r04a_patch_embed.txt
