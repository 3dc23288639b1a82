"""rel-L2 of the HIP VAE (decode image, encode moments) against oracle/vae_ref.py (fp32, CPU) with shared random weights, full
SD / SDXL config, 64x64 and 128x128 images.  Run once per operand build:  python tools/vae_parity.py ; PXA_OPERAND_DTYPE=f16 python tools/vae_parity.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from oracle.vae_ref import AutoencoderKLRef, randomize_  # noqa: E402
from pixart_sigma_amd import lib  # noqa: E402
from pixart_sigma_amd.vae import AutoencoderKL  # noqa: E402


def rel(a, b):
    a, b = a.double().cpu(), b.double()
    return float((a - b).norm() / b.norm())


ref = randomize_(AutoencoderKLRef(), seed=5)
vae = AutoencoderKL()
vae.load_state_dict(ref.state_dict())
vae = vae.cuda()
out = {"operand": lib.OPERAND}
g = torch.Generator().manual_seed(6)
for px in (64, 128):
    z, x = torch.randn(1, 4, px // 8, px // 8, generator=g), torch.randn(1, 3, px, px, generator=g)
    with torch.no_grad():
        want, (mean, logvar) = ref.decode(z), ref.encode_moments(x)
    d = vae.encode(x.cuda()).latent_dist
    out[f"{px}px"] = {"decode": rel(vae.decode(z.cuda()).sample, want), "encode_mean": rel(d.mean, mean), "encode_logvar": rel(d.logvar, logvar)}
# BASELINE config 5's VAE shape: full-architecture 512px decode, batch 2, against the fp32 restatement evaluated on the same GPU
z = torch.randn(2, 4, 64, 64, generator=g)
ref = ref.cuda()
with torch.no_grad():
    want = ref.decode(z.cuda()).cpu()
out["512px_b2"] = {"decode": rel(vae.decode(z.cuda()).sample, want)}
print(json.dumps(out))
