"""Host-side orchestration of the PixArt-Sigma denoiser on the HIP kernels: which kernel runs when, which
activations are kept for backward, where weight gradients land.  Pure sequencing — all arithmetic is in
libpixart_hip.so (ops.py -> C ABI).  Mirrors the dataflow of PixArtMS.forward / PixArtMSBlock.forward
(reference diffusion/model/nets/PixArtMS.py:71-79,165-211) with these fusions:

  block l:  x_in  = x2[l-1] + gate_mlp[l-1]*u3[l-1] ; xn1 = LN(x_in)(1+scale_msa)+shift_msa      (ln_mod_fwd, 1 pass)
            qkv   = xn1 Wqkv^T + b                                                                (gemm NT)
            a     = softmax(q k^T/sqrt(72)) v          [k,v optionally KV-compressed]             (attn_fwd)
            u1    = a Wproj^T + b                                                                 (gemm NT)
            x1    = x_in + gate_msa*u1 ; x1b = bf16(x1)                                           (ln_mod_fwd, no LN)
            qc    = x1b Wq^T + b ; kvc = y Wkv^T + b ; c = varlen-attn(qc, kvc) ; u2 = c Wcp^T+b  (gemm, attn_fwd, gemm)
            x2    = x1 + u2 ; xn2 = LN(x2)(1+scale_mlp)+shift_mlp                                 (ln_mod_fwd)
            h     = gelu(xn2 W1^T + b)  (pre-activation kept) ; u3 = h W2^T + b                   (gemm+GELU, gemm)
  final:    x3 = x2 + gate_mlp*u3 ; LN+modulate ; Linear(D,32) ; unpatchify

The residual stream, LayerNorm statistics, softmax and all accumulators are fp32; GEMM / attention operands and the
stored branch activations are bf16 (DESIGN.md "Numerics").  Backward is hand-sequenced (no autograd inside): weight
gradients are accumulated straight into the flat fp32 gradient buffer (ParamStore.grad).
"""
import math
import os
import weakref

import numpy as np
import torch

from . import ops
from .ops import BF16, F32, NN, NT, TN


def sincos_pos_embed(embed_dim, h, w, pe_interpolation, base_size):
    """Host float64 table, same arithmetic as the reference's get_2d_sincos_pos_embed (PixArt.py:258-307):
    float32 grid coordinates, float64 omega/sin/cos, w (column) coordinate in the first half.  Cached per geometry
    on the device instead of being rebuilt in numpy on every forward (PixArtMS.py:177-182)."""
    grid_h = np.arange(h, dtype=np.float32) / (h / base_size) / pe_interpolation
    grid_w = np.arange(w, dtype=np.float32) / (w / base_size) / pe_interpolation
    gw, gh = np.meshgrid(grid_w, grid_h)
    quarter = embed_dim // 4
    omega = np.arange(quarter, dtype=np.float64)
    omega /= embed_dim / 4.0
    omega = 1.0 / 10000 ** omega

    def enc(pos):
        out = np.einsum("m,d->md", pos.reshape(-1), omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)
    return np.concatenate([enc(gw), enc(gh)], axis=1)


def _capturing():
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


class ParamStore:
    """All model parameters in ONE flat fp32 buffer (master), ONE flat fp32 gradient buffer and ONE flat bf16 shadow
    (the GEMM operands), sharing offsets.  nn.Parameters are re-pointed to views of the master so optimizers, state
    dicts and the data-parallel all-reduce see ordinary tensors, while the fused AdamW / grad-norm / all-reduce kernels
    work on contiguous ranges.  Order = forward order (embedders, block 0..L-1, final layer) so a block's gradients are a
    contiguous bucket that completes at a known point of the hand-sequenced backward."""
    ALIGN = 64

    def __init__(self, named_params, device, group_of=None):
        """group_of: optional name -> group label; consecutive parameters with one label form a bucket (store.groups)."""
        self.names, self.offset, self.shape, self.numel = [], {}, {}, {}
        self.groups = {}
        off = 0
        for name, p in named_params:
            self.names.append(name)
            self.offset[name], self.shape[name], self.numel[name] = off, tuple(p.shape), p.numel()
            end = off + (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
            if group_of is not None:
                gname = group_of(name)
                if gname in self.groups:
                    assert self.groups[gname][1] == off, f"group {gname} is not contiguous at {name}"
                    self.groups[gname] = (self.groups[gname][0], end)
                else:
                    self.groups[gname] = (off, end)
            off = end
        self.total = off
        self.device = device
        self.master = torch.zeros(off, dtype=F32, device=device)
        self.grad = torch.zeros(off, dtype=F32, device=device)
        self.shadow = torch.zeros(off, dtype=BF16, device=device)
        self.params = dict(named_params)
        self._versions = None
        self.generation = 0        # bumped whenever the shadow weights change (re-cast here, fused optimizer steps): keys everything cached from them
        self._on_change = []       # callables run at every bump: buffers DERIVED from the weights are rewritten in place there (Engine._refresh_qs)
        for name, p in named_params:
            v = self.view(self.master, name)
            v.copy_(p.data)
            p.data = v
            p.grad = self.view(self.grad, name)

    def view(self, flat, name):
        o = self.offset[name]
        return flat[o:o + self.numel[name]].view(self.shape[name])

    def w(self, name):   # bf16 shadow weight (2-D)
        return self.view(self.shadow, name)

    def f(self, name):   # fp32 master (bias / table)
        return self.view(self.master, name)

    def g(self, name):   # fp32 gradient accumulator
        return self.view(self.grad, name)

    def range_of(self, prefix):
        """[start, end) element range of all parameters whose name starts with prefix (contiguous by construction)."""
        idx = [i for i, n in enumerate(self.names) if n.startswith(prefix)]
        assert idx and idx == list(range(idx[0], idx[-1] + 1)), prefix
        last = self.names[idx[-1]]
        end = self.offset[last] + (self.numel[last] + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        return self.offset[self.names[idx[0]]], end

    def bias_range(self, prefix):
        """[start, end) of the contiguous run of '*.bias' parameters under `prefix` (the model orders a block's weights first,
        then its biases, so that fused bias-gradient partials can be folded into the gradient buffer with one launch)."""
        idx = [i for i, n in enumerate(self.names) if n.startswith(prefix) and n.endswith(".bias")]
        assert idx and idx == list(range(idx[0], idx[-1] + 1)), f"biases under {prefix} are not contiguous in the flat store"
        last = self.names[idx[-1]]
        return self.offset[self.names[idx[0]]], self.offset[last] + (self.numel[last] + self.ALIGN - 1) // self.ALIGN * self.ALIGN

    def owns(self, p, name):
        return p.data_ptr() == self.master.data_ptr() + 4 * self.offset[name] and p.device == self.master.device

    def owns_all(self, named_params):
        """True iff EVERY parameter still lives at its offset of this store's master buffer (a stand-alone block engine, .to(),
        or an assignment to p.data re-points parameters one by one; checking only the first one would keep reading stale copies)."""
        return all(n in self.offset and self.owns(p, n) for n, p in named_params)

    def bump(self):
        """The shadow weights changed (a re-cast, a fused optimizer step): new generation, and every buffer derived from the weights is rewritten IN PLACE now -
        so whatever holds its address (a captured HIP graph) keeps reading current values, exactly as it does for the shadow itself."""
        self.generation += 1
        for ref in list(self._on_change):            # weak references: a store must not keep a discarded engine (and its buffers) alive
            fn = ref()
            if fn is None:
                self._on_change.remove(ref)
            else:
                fn()

    def refresh_shadow(self, force=False):
        """Re-cast master -> bf16 shadow if any parameter was modified by torch ops since the last cast (the fused AdamW
        kernel refreshes the shadow itself and does not bump versions)."""
        try:
            vers = tuple(p._version for p in self.params.values())
        except RuntimeError:      # parameters created under torch.inference_mode() carry no version counter: re-cast every call
            vers, force = None, True
        if force or vers != self._versions:
            ops.cast_bf16(self.master, self.shadow)
            self._versions = vers
            self.bump()

    def attach_grads(self):
        """Make every p.grad the view of the flat buffer; returns True if the buffer had to be (re)zeroed."""
        missing = [n for n, p in self.params.items() if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * self.offset[n]]
        if not missing:
            return False
        if len(missing) == len(self.params):
            self.grad.zero_()
        for n in missing:
            p = self.params[n]
            v = self.view(self.grad, n)
            if len(missing) != len(self.params):
                v.zero_()
                if p.grad is not None:
                    v.copy_(p.grad)
            p.grad = v
        return True


def _splitk(M, N, K):
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    return max(1, min((640 + tiles - 1) // tiles, max(1, K // 512)))


class Engine:
    """Forward/backward sequencing for one PixArtMS instance."""

    def __init__(self, store, cfg):
        self.S, self.cfg = store, cfg
        self._pos_cache = {}
        self._len_cache = {}
        # Inference-only cache of everything that depends on the text alone: the caption MLP output and every block's cross-attention K / V
        # (kv_linear of the packed caption rows).  A sampler calls the model 20+ times with the SAME caption tensor (scripts/inference.py,
        # DPM_Solver.sample): those 1 + depth GEMMs per call (28 of them on M <= 4,800 rows: the 512px inference tail) run once.
        # Key = identity + version of y / the drop mask / y_null and the lengths; any training forward (save != None) clears it - weights may change.
        self._text_cache = None
        self.grad_ready_hook = None   # callable(prefix) fired when a parameter group's gradients are complete
        # Round 5: the softmax scale rides in the qkv projection.  The forward multiplies by a COPY of every block's attn.qkv weight / bias whose q rows
        # carry scale * log2 e (cast from the fp32 master in one rounding; ops.scale_copy, one launch per run of equally spaced blocks after each optimizer
        # step), so q leaves the GEMM as the exponent's argument: the attention kernels take q k^T straight into exp2 (pxa_attn_args.q_prescaled) - one
        # multiply per score less in the dK/dV kernel, and the fp16 forward's folded first product loses its second rounding of q.  The backward is
        # untouched: dq comes out with respect to the unscaled queries and the dX / dW GEMMs read the true weights.  Off under qk_norm (the LayerNorm behind
        # the projection would normalise the factor away) and with PXA_Q_PRESCALE=0 (A/B).
        self.prescale = (not cfg.get("qk_norm")) and os.environ.get("PXA_Q_PRESCALE", "1") != "0" and cfg["hidden_size"] // cfg["num_heads"] == 72
        # The copy lives in ONE pair of buffers per engine, allocated here and rewritten in place whenever the shadow weights change (ParamStore.bump):
        # a HIP graph captured over the forward (DPM_Solver.sample_graphed) bakes these addresses exactly as it bakes the shadow's, and replays after an
        # optimizer step / weight load read current values (the per-generation allocation it replaces left such a graph reading a freed block,
        # and cost the training step a 223 MB allocate / free).
        self._qs = None
        if self.prescale:
            D, depth = cfg["hidden_size"], cfg["depth"]
            self._qs = (torch.empty((depth, 3 * D, D), dtype=BF16, device=store.device), torch.empty((depth, 3 * D), dtype=F32, device=store.device))
            store._on_change.append(weakref.WeakMethod(self._refresh_qs))

    # ------------------------------------------------------------------ helpers
    def pos_table(self, h, w):
        key = (h, w)
        if key not in self._pos_cache:
            c = self.cfg
            tab = sincos_pos_embed(c["hidden_size"], h, w, c["pe_interpolation"], c["base_size"])
            self._pos_cache[key] = torch.from_numpy(tab).to(F32).to(self.S.device).contiguous()
        return self._pos_cache[key]

    # Item order of the token GEMMs (round 5, pxa_gemm_args.items_descending).  Every row kernel and attention kernel sweeps the token rows upwards, so the rows it
    # wrote LAST are the ones still in the 256 MB Infinity Cache when the next GEMM starts: that GEMM walks its tiles downwards, and its own output then ends with the
    # first rows - fresh for the ascending consumer behind it.  The second GEMM of a GEMM -> GEMM pair (fc2 behind fc1; fc1's dX behind fc2's dX) follows its
    # producer by walking upwards.  Bit-identical results; -1.7 ... -2.1 ms per training step (profiles/r5_16 / r5_17_step_ab_reverse.txt).  PXA_GEMM_ASCENDING=1: off (A/B).
    @staticmethod
    def _desc(name):
        return not name.endswith(("mlp.fc2", "kv_linear", "y_proj.fc1", "y_proj.fc2"))

    def _lin(self, x, name, **kw):
        return ops.gemm(x, self.S.w(name + ".weight"), NT, bias=self.S.f(name + ".bias"), descending=self._desc(name), **kw)

    def _lin_bwd(self, dy, x, name, need_dx=True, dx_kw=None, bias_done=False):
        """dW += dy^T x ; db += colsum(dy) (unless the kernel that produced dy already accumulated it) ; returns dx = dy W (bf16)."""
        S = self.S
        M, N = S.shape[name + ".weight"]
        # (order measured: the weight gradient FIRST - its pass over dy leaves dy in the Infinity Cache for the dX GEMM; dX first is 1.6 ms per step slower,
        # profiles/r5_20_step_ab_dx_first.txt)
        ops.gemm(dy, x, TN, out_f32=S.g(name + ".weight"), accumulate=True, split_k=0)
        if not bias_done:
            ops.colsum(dy, S.g(name + ".bias"))
        if need_dx:                 # dX of fc1 follows fc2's dX GEMM (upwards); the text-row and caption-MLP GEMMs are too small to care
            desc = not name.endswith(("mlp.fc1", "kv_linear", "y_proj.fc1", "y_proj.fc2"))
            return ops.gemm(dy, S.w(name + ".weight"), NN, descending=desc, **(dx_kw or {}))
        return None

    def _refresh_qs(self):
        """Rewrite the prescaled qkv weight / bias copies from the fp32 master (one rounding), in place: one launch per run of equally spaced blocks."""
        S, D, depth = self.S, self.cfg["hidden_size"], self.cfg["depth"]
        w, b = self._qs
        offw = [S.offset[f"blocks.{i}.attn.qkv.weight"] for i in range(depth)]
        offb = [S.offset[f"blocks.{i}.attn.qkv.bias"] for i in range(depth)]
        i = 0
        while i < depth:                                  # runs of equally spaced blocks (KV-compressed blocks carry extra parameters)
            n, st = 1, 0
            if i + 1 < depth:
                st = offw[i + 1] - offw[i]
                while i + n < depth and offw[i + n] - offw[i + n - 1] == st and offb[i + n] - offb[i + n - 1] == st:
                    n += 1
            ops.scale_copy(S.master[offw[i]:], st if n > 1 else 0, n, D * D, 3 * D * D, ops.Q_PRESCALE, out_bf16=w[i:i + n])
            ops.scale_copy(S.master[offb[i]:], st if n > 1 else 0, n, D, 3 * D, ops.Q_PRESCALE, out_f32=b[i:i + n])
            i += n

    def _qkv_prescaled(self, l):
        """(weight (3D, D) in the operand type, bias (3D,) fp32) of block l with the q rows times scale * log2 e: views of the engine's persistent copy."""
        return self._qs[0][l], self._qs[1][l]

    # ------------------------------------------------------------------ caption branch
    def caption_fwd(self, y, row_idx, L, drop, y_null):
        """y (B*L, 4096) fp32 -> packed y_emb (Ltot, D) bf16  (CaptionEmbedder, PixArt_blocks.py:400-407 after masked_select).
        y_null: the y_embedding buffer substituted for dropped samples (token_drop)."""
        S = self.S
        yb = ops.gather_rows_bf16(y, row_idx, L, alt=y_null if drop is not None else None, drop=drop)
        hpre = torch.empty((yb.shape[0], S.shape["y_embedder.y_proj.fc1.weight"][0]), dtype=BF16, device=y.device)
        h = self._lin(yb, "y_embedder.y_proj.fc1", act=ops.ACT_GELU_SAVE_GRAD, out2=hpre)   # hpre holds GELU'(pre-activation)
        ye = self._lin(h, "y_embedder.y_proj.fc2")
        return ye, (yb, hpre, h)

    def caption_bwd(self, dye_f32, saved):
        yb, hpre, h = saved
        dye = torch.empty(dye_f32.shape, dtype=BF16, device=dye_f32.device)
        ops.gate_bwd(dye_f32, du=dye, rows_per_batch=dye_f32.shape[0])
        dh = self._lin_bwd(dye, h, "y_embedder.y_proj.fc2", dx_kw=dict(act=ops.ACT_MUL_AUX, aux=hpre))
        self._lin_bwd(dh, yb, "y_embedder.y_proj.fc1", need_dx=False)

    # ------------------------------------------------------------------ block
    def block_fwd(self, l, x_prev, u_prev, gate_prev, ctx):
        """Returns (x2, u3, saved).  x_prev/u_prev/gate_prev: residual, pending gated branch of the previous block."""
        S, c = self.S, self.cfg
        B, N, D, H = ctx["B"], ctx["N"], c["hidden_size"], c["num_heads"]
        p = f"blocks.{l}."
        mod = ctx["mod"][l]                                   # (B, 6, D) fp32 view
        sm, scm, gm, sl, scl, gl = (mod[:, i] for i in range(6))
        st = 6 * D
        r = ops.ln_mod_fwd(x_prev, sm, scm, st, u=u_prev, gate=gate_prev, gate_stride=st, rows_per_batch=N, want_stats=True)
        x_in, xn1, mean1, rstd1 = r["x"], r["xn"], r["mean"], r["rstd"]
        pre = dict(q_prescaled=True) if self.prescale else {}
        if self.prescale:
            wq, bq = self._qkv_prescaled(l)
            qkv = ops.gemm(xn1, wq, NT, bias=bq, descending=True)
        else:
            qkv = self._lin(xn1, p + "attn.qkv")
        qkn = None
        if c.get("qk_norm"):        # q_norm / k_norm on the full-resolution q, k column blocks, in place (PixArt_blocks.py:133-134)
            qkn = (ops.ln_affine_fwd(qkv[:, :D], S.f(p + "attn.q_norm.weight"), S.f(p + "attn.q_norm.bias")),
                   ops.ln_affine_fwd(qkv[:, D:2 * D], S.f(p + "attn.k_norm.weight"), S.f(p + "attn.k_norm.bias")))
        sr = c["kv_scale_factor"] if l in c["kv_layers"] else 1
        a = torch.empty((B * N, D), dtype=BF16, device=qkv.device)
        lse = torch.empty((B, H, N), dtype=F32, device=qkv.device)
        s3 = (N * 3 * D, 3 * D, 72)
        kc = vc = None
        if sr > 1:
            hh, ww = ctx["hw"]
            if c["kv_sampling"] == "conv":
                cw, cb = S.f(p + "attn.sr.weight"), S.f(p + "attn.sr.bias")
                lw, lb = S.f(p + "attn.norm.weight"), S.f(p + "attn.norm.bias")
                kc = ops.kv_compress_fwd(qkv[:, D:2 * D], N * 3 * D, 3 * D, cw, cb, lw, lb, B, hh, ww, D, sr)
                vc = ops.kv_compress_fwd(qkv[:, 2 * D:], N * 3 * D, 3 * D, cw, cb, lw, lb, B, hh, ww, D, sr)
                Nk = kc.shape[1]
                sk = (Nk * D, D, 72)
                ops.attention_fwd(qkv[:, :D], kc, vc, a, lse, B, H, N, Nk, (s3, sk, sk, (N * D, D, 72)), **pre)
            elif c["kv_sampling"] == "uniform_every":
                Nk = (N + sr - 1) // sr
                sk = (N * 3 * D, 3 * D * sr, 72)
                ops.attention_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], a, lse, B, H, N, Nk, (s3, sk, sk, (N * D, D, 72)), **pre)
            elif c["kv_sampling"] in ("uniform", "ave"):      # 'ave' = nearest interpolation = the same strided pick (PixArt_blocks.py:110-115)
                Nk = (hh // sr) * (ww // sr)
                kc = torch.empty((B, Nk, D), dtype=BF16, device=qkv.device)
                vc = torch.empty((B, Nk, D), dtype=BF16, device=qkv.device)
                ops.kv_pick(qkv[:, D:2 * D], kc, N * 3 * D, 3 * D, B, hh, ww, D, sr)
                ops.kv_pick(qkv[:, 2 * D:], vc, N * 3 * D, 3 * D, B, hh, ww, D, sr)
                sk = (Nk * D, D, 72)
                ops.attention_fwd(qkv[:, :D], kc, vc, a, lse, B, H, N, Nk, (s3, sk, sk, (N * D, D, 72)), **pre)
            else:
                raise ValueError(f"unknown kv sampling mode {c['kv_sampling']!r}")
        else:
            ops.attention_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], a, lse, B, H, N, N, (s3, s3, s3, (N * D, D, 72)), **pre)
        u1 = self._lin(a, p + "attn.proj")
        r = ops.ln_mod_fwd(x_in, u=u1, gate=gm, gate_stride=st, want_xn=False, want_xb=True, rows_per_batch=N)
        x1, x1b = r["x"], r["xb"]
        qc = self._lin(x1b, p + "cross_attn.q_linear")
        ye = ctx["ye"]
        kvc = ctx["kvc"][l] if ctx.get("kvc") is not None else None      # inference: constant over the sampler's steps (Engine._text_cache)
        if kvc is None:
            kvc = self._lin(ye, p + "cross_attn.kv_linear")
            if ctx.get("kvc") is not None:
                ctx["kvc"][l] = kvc
        cr = torch.empty((B * N, D), dtype=BF16, device=qkv.device)
        lse_c = torch.empty((B, H, N), dtype=F32, device=qkv.device)
        sc_str = ((N * D, D, 72), (0, 2 * D, 72), (0, 2 * D, 72), (N * D, D, 72))
        ops.attention_fwd(qc, kvc[:, :D], kvc[:, D:], cr, lse_c, B, H, N, ctx["max_len"], sc_str,
                          kv_start=ctx["kv_start"], kv_len=ctx["kv_len"], max_kv_len=ctx["max_len"], kv_len_host=ctx.get("lens_host"))
        u2 = self._lin(cr, p + "cross_attn.proj")
        r = ops.ln_mod_fwd(x1, sl, scl, st, u=u2, x_out=x1, rows_per_batch=N, want_stats=True)
        x2, xn2, mean2, rstd2 = r["x"], r["xn"], r["mean"], r["rstd"]
        if ctx.get("need_grad_aux", True):
            hpre = torch.empty((B * N, S.shape[p + "mlp.fc1.weight"][0]), dtype=BF16, device=qkv.device)
            h = self._lin(xn2, p + "mlp.fc1", act=ops.ACT_GELU_SAVE_GRAD, out2=hpre)   # hpre holds GELU'(pre-activation), bf16
        else:                                      # inference / the discarded forward of a checkpointed step: no backward reads GELU' - one output, half the
            hpre = None                            # epilogue's transcendentals and stores
            h = self._lin(xn2, p + "mlp.fc1", act=ops.ACT_GELU)
        u3 = self._lin(h, p + "mlp.fc2")
        saved = dict(x_in=x_in, mean1=mean1, rstd1=rstd1, xn1=xn1, qkv=qkv, a=a, lse=lse, u1=u1, x1b=x1b, qc=qc, kvc=kvc,
                     cr=cr, lse_c=lse_c, x2=x2, mean2=mean2, rstd2=rstd2, xn2=xn2, hpre=hpre, h=h, u3=u3, kc=kc, vc=vc, sr=sr, qkn=qkn)
        return x2, u3, gl, saved

    def block_bwd(self, l, G, sv, ctx):
        """G: fp32 (R, D) gradient w.r.t. this block's output residual x3 = x2 + gate_mlp*u3; overwritten with the
        gradient w.r.t. x_in.  Accumulates parameter gradients, d(mod) and d(y_emb)."""
        S, c = self.S, self.cfg
        B, N, D, H = ctx["B"], ctx["N"], c["hidden_size"], c["num_heads"]
        R = B * N
        p = f"blocks.{l}."
        mod, dmod = ctx["mod"][l], ctx["dmod"][l]
        st = 6 * D
        dev = G.device
        # bias gradients ride along with the kernels that produce the output gradients (no separate column-sum passes); they add
        # into PXA_COLSUM_SLOTS partial rows laid out like this block's bias range of the flat gradient buffer
        bs, be = S.bias_range(p)
        part = torch.zeros((ops.COLSUM_SLOTS, be - bs), dtype=F32, device=dev)

        def pb(name, lo=0, hi=None):
            o = S.offset[p + name] - bs
            return part[:, o + lo:o + (S.numel[p + name] if hi is None else hi)]
        # ---- MLP branch: x3 = x2 + gate_mlp * u3
        du = torch.empty((R, D), dtype=BF16, device=dev)
        ops.gate_bwd(G, u=sv["u3"], gate=mod[:, 5], mod_stride=st, du=du, dgate=dmod[:, 5], dmod_stride=st, rows_per_batch=N,
                     dbias=pb("mlp.fc2.bias"))
        dh = self._lin_bwd(du, sv["h"], p + "mlp.fc2", dx_kw=dict(act=ops.ACT_MUL_AUX, aux=sv["hpre"], colsum=pb("mlp.fc1.bias")), bias_done=True)
        dxn = self._lin_bwd(dh, sv["xn2"], p + "mlp.fc1", bias_done=True)
        del dh
        # ---- cross attention: x2 = x1 + u2 (no gate, no norm): du2 = bf16(G2) comes out of the LN backward pass itself
        # (round 5: cross_attn.proj's bias gradient = column sums of G2 ride in this pass; PXA_FUSED_CPROJ_BIAS=0: the separate colsum pass, A/B)
        fused_cb = os.environ.get("PXA_FUSED_CPROJ_BIAS", "1") != "0"
        ops.ln_mod_bwd(dxn, sv["x2"], sv["mean2"], sv["rstd2"], mod[:, 4], st, G, G, dmod[:, 3], dmod[:, 4], st, N, dx_bf16=du,
                       dbias=pb("cross_attn.proj.bias") if fused_cb else None)
        dc = self._lin_bwd(du, sv["cr"], p + "cross_attn.proj", bias_done=fused_cb)
        dqc = torch.empty((R, D), dtype=BF16, device=dev)
        dkvc = torch.empty_like(sv["kvc"])
        delta = torch.empty((B, H, N), dtype=F32, device=dev)
        sc_str = ((N * D, D, 72), (0, 2 * D, 72), (0, 2 * D, 72), (N * D, D, 72))
        ops.attention_bwd(sv["qc"], sv["kvc"][:, :D], sv["kvc"][:, D:], sv["cr"], dc, sv["lse_c"], delta, dqc, dkvc[:, :D], dkvc[:, D:],
                          B, H, N, ctx["max_len"], sc_str, ((N * D, D, 72), (0, 2 * D, 72), (0, 2 * D, 72)),
                          kv_start=ctx["kv_start"], kv_len=ctx["kv_len"], max_kv_len=ctx["max_len"], kv_len_host=ctx.get("lens_host"))
        gq = self._lin_bwd(dqc, sv["x1b"], p + "cross_attn.q_linear")
        # 4,800 text rows fill 95 of 256 CUs with 256 x 256 tiles: split_k = 0 lets the library's (tile, split) model choose (128 x 128 here: 72 -> 47 us)
        self._lin_bwd(dkvc, ctx["ye"], p + "cross_attn.kv_linear", dx_kw=dict(out_f32=ctx["dye"], accumulate=True, split_k=0))
        # ---- self attention: x1 = x_in + gate_msa * u1 ; G1 = G + gq
        ops.gate_bwd(G, add=gq, u=sv["u1"], gate=mod[:, 2], mod_stride=st, dx_out=G, du=du, dgate=dmod[:, 2], dmod_stride=st, rows_per_batch=N,
                     dbias=pb("attn.proj.bias"))
        da = self._lin_bwd(du, sv["a"], p + "attn.proj", bias_done=True)
        qkv = sv["qkv"]
        s3 = (N * 3 * D, 3 * D, 72)
        so = (N * D, D, 72)
        sr = sv["sr"]
        pre = dict(q_prescaled=True) if self.prescale else {}
        if sr == 1:
            dqkv = torch.empty_like(qkv)
            ops.attention_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], sv["a"], da, sv["lse"], delta, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:],
                              B, H, N, N, (s3, s3, s3, so), (s3, s3, s3), **pre)
        elif c["kv_sampling"] == "uniform_every":          # strided keys: gradients land on the picked tokens, the rest stay zero
            dqkv = torch.zeros_like(qkv)
            Nk = (N + sr - 1) // sr
            sk = (N * 3 * D, 3 * D * sr, 72)
            ops.attention_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], sv["a"], da, sv["lse"], delta, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:],
                              B, H, N, Nk, (s3, sk, sk, so), (s3, sk, sk), **pre)
        else:                                               # compressed K/V buffers: attention backward, then the compression's backward
            hh, ww = ctx["hw"]
            kc, vc = sv["kc"], sv["vc"]
            Nk = kc.shape[1]
            sk = (Nk * D, D, 72)
            covered = (hh % sr == 0) and (ww % sr == 0) and c["kv_sampling"] == "conv"
            dqkv = torch.empty_like(qkv) if covered else torch.zeros_like(qkv)
            dkc, dvc = torch.empty_like(kc), torch.empty_like(vc)
            ops.attention_bwd(qkv[:, :D], kc, vc, sv["a"], da, sv["lse"], delta, dqkv[:, :D], dkc, dvc, B, H, N, Nk, (s3, sk, sk, so), (s3, sk, sk), **pre)
            if c["kv_sampling"] == "conv":
                cw, cb, lw = S.f(p + "attn.sr.weight"), S.f(p + "attn.sr.bias"), S.f(p + "attn.norm.weight")
                gcw, gcb = S.g(p + "attn.sr.weight"), S.g(p + "attn.sr.bias")
                glw, glb = S.g(p + "attn.norm.weight"), S.g(p + "attn.norm.bias")
                for dyc, lo in ((dkc, D), (dvc, 2 * D)):   # the same sr / norm parameters process K and V (PixArt_blocks.py:138-139)
                    ops.kv_compress_bwd(dyc, qkv[:, lo:lo + D], N * 3 * D, 3 * D, cw, cb, lw, dqkv[:, lo:lo + D], N * 3 * D, 3 * D,
                                        gcw, gcb, glw, glb, B, hh, ww, D, sr)
            else:
                ops.kv_pick(dkc, dqkv[:, D:2 * D], N * 3 * D, 3 * D, B, hh, ww, D, sr, backward=True)
                ops.kv_pick(dvc, dqkv[:, 2 * D:], N * 3 * D, 3 * D, B, hh, ww, D, sr, backward=True)
        if sv.get("qkn") is not None:                        # back through q_norm / k_norm: dqkv's q, k blocks become d(raw q), d(raw k)
            (qs, qm, qr), (ks_, km, kr) = sv["qkn"]
            ops.ln_affine_bwd(dqkv[:, :D], qs, qm, qr, S.f(p + "attn.q_norm.weight"), S.g(p + "attn.q_norm.weight"), S.g(p + "attn.q_norm.bias"))
            ops.ln_affine_bwd(dqkv[:, D:2 * D], ks_, km, kr, S.f(p + "attn.k_norm.weight"), S.g(p + "attn.k_norm.weight"), S.g(p + "attn.k_norm.bias"))
        # the attention kernels can also emit the q/k/v bias-gradient sums (pxa_attn_args.d*_colsum), but the cross-lane row
        # reductions cost them more (~0.5 ms/block) than one streaming column-sum pass over dqkv (~0.2 ms/block): measured, not used
        dxn = self._lin_bwd(dqkv, sv["xn1"], p + "attn.qkv")
        ops.ln_mod_bwd(dxn, sv["x_in"], sv["mean1"], sv["rstd1"], mod[:, 1], st, G, G, dmod[:, 0], dmod[:, 1], st, N)
        ops.colsum_reduce(part, S.grad[bs:be])
        if self.grad_ready_hook:
            self.grad_ready_hook(f"blocks.{l}")
        return G

    # ------------------------------------------------------------------ whole core
    def forward(self, x, y, mod, fin_mod, row_idx, lens, drop, save, y_null=None):
        """x (B,4,Hl,Wl) fp32, y (B*L,4096) fp32, mod (L,B,6,D) fp32, fin_mod (B,2,D) fp32 -> (out (B,2C,Hl,Wl), saved)."""
        S, c = self.S, self.cfg
        B, _, Hl, Wl = x.shape
        h, w = Hl // 2, Wl // 2
        N, D, depth = h * w, c["hidden_size"], c["depth"]
        dev = x.device
        lk = tuple(int(v) for v in lens)
        if lk not in self._len_cache:                        # per-sample text lengths -> device index tensors, uploaded once per distinct set
            if len(self._len_cache) > 64:
                self._len_cache.clear()
            starts = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int32)
            self._len_cache[lk] = (torch.tensor(lens, dtype=torch.int32, device=dev), torch.from_numpy(starts).to(dev))
        kv_len, kv_start = self._len_cache[lk]
        ctx = dict(B=B, N=N, hw=(h, w), mod=mod, kv_len=kv_len, kv_start=kv_start, max_len=int(max(lens)), lens_host=lk, need_grad_aux=(save == "all"))
        L = y.shape[0] // B
        tkey = None
        if save or torch.is_grad_enabled():
            self._text_cache = None
        elif _capturing() or os.environ.get("PXA_TEXT_CACHE", "1") == "0":
            pass                                  # a captured graph must own every buffer it reads: no cache inside sample_graphed's capture
        else:
            tkey = (S.generation, y.data_ptr(), y._version, tuple(y.shape), lk, None if drop is None else (drop.data_ptr(), drop._version),
                    None if y_null is None else (y_null.data_ptr(), y_null._version), row_idx.data_ptr())
        if tkey is not None and self._text_cache is not None and self._text_cache["key"] == tkey:
            ye, cap_saved = self._text_cache["ye"], None
            ctx["kvc"] = self._text_cache["kvc"]
        else:
            ye, cap_saved = self.caption_fwd(y, row_idx, L, drop, y_null)
            if tkey is not None:
                ctx["kvc"] = [None] * depth
                self._text_cache = dict(key=tkey, ye=ye, kvc=ctx["kvc"], y=y)      # holds y so that its address cannot be reused by another tensor
        ctx["ye"] = ye
        xt = ops.patch_embed_fwd(x, S.f("x_embedder.proj.weight"), S.f("x_embedder.proj.bias"), self.pos_table(h, w))
        x_prev, u_prev, gate_prev = xt, None, None
        blocks = []
        for l in range(depth):
            x_prev, u_prev, gate_prev, sv = self.block_fwd(l, x_prev, u_prev, gate_prev, ctx)
            blocks.append(sv if save == "all" else (dict(x_in=sv["x_in"]) if save == "ckpt" else None))
        r = ops.ln_mod_fwd(x_prev, fin_mod[:, 0], fin_mod[:, 1], 2 * D, u=u_prev, gate=gate_prev, gate_stride=6 * D, rows_per_batch=N, want_stats=True)
        lin = ops.gemm(r["xn"], S.w("final_layer.linear.weight"), NT, bias=S.f("final_layer.linear.bias"), out_dtype=F32)
        out = ops.unpatchify_fwd(lin, B, h, w, c["out_channels"])
        saved = None
        if save:
            saved = dict(ctx=ctx, blocks=blocks, cap=cap_saved, x=x, x3=r["x"], meanf=r["mean"], rstdf=r["rstd"], xnf=r["xn"], fin_mod=fin_mod, mode=save)
        return out, saved

    def backward(self, dout, saved):
        """dout (B,2C,Hl,Wl) fp32 -> (dmod (L,B,6,D), dfin (B,2,D)); parameter gradients go to the flat buffer."""
        S, c = self.S, self.cfg
        ctx = saved["ctx"]
        B, N, D, depth = ctx["B"], ctx["N"], c["hidden_size"], c["depth"]
        h, w = ctx["hw"]
        dev = dout.device
        ctx["dmod"] = torch.zeros_like(ctx["mod"])
        dfin = torch.zeros_like(saved["fin_mod"])
        ctx["dye"] = torch.zeros(ctx["ye"].shape, dtype=F32, device=dev)
        dlin = ops.patchify_bwd(dout.contiguous(), h, w)
        dxn = self._lin_bwd(dlin, saved["xnf"], "final_layer.linear")
        G = torch.empty((B * N, D), dtype=F32, device=dev)
        ops.ln_mod_bwd(dxn, saved["x3"], saved["meanf"], saved["rstdf"], saved["fin_mod"][:, 1], 2 * D, None, G, dfin[:, 0], dfin[:, 1], 2 * D, N)
        if self.grad_ready_hook:
            self.grad_ready_hook("final")
        ctx["need_grad_aux"] = True                      # the recomputed forwards of a checkpointed step feed block_bwd
        for l in reversed(range(depth)):
            sv = saved["blocks"][l]
            if saved["mode"] == "ckpt":   # recompute this block's activations from its saved input (auto_grad_checkpoint semantics)
                sv = self._recompute(l, sv, ctx)
            G = self.block_bwd(l, G, sv, ctx)
            saved["blocks"][l] = None
        ops.patch_embed_bwd(saved["x"], G, S.g("x_embedder.proj.weight"), S.g("x_embedder.proj.bias"))
        self.caption_bwd(ctx["dye"], saved["cap"])
        return ctx["dmod"], dfin

    def _recompute(self, l, sv, ctx):
        """Re-run block l forward from its saved input x_in (x_in already includes the previous block's gated MLP branch)."""
        return self.block_fwd(l, sv["x_in"], None, None, ctx)[3]
