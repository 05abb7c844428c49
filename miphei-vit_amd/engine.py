"""HIP execution engine of the MIPHEI-ViT generator (forward, backward, fused training step).

Sequences the C-ABI kernels of ``libmiphei_hip.so`` for the module graph the reference builds in
``/root/reference/src/generators/mipheivit.py`` (``ViTMatte.forward`` :106-110, ``Encoder.forward`` :153-163,
``Detail_Capture.forward`` :207-220), the timm ViT it wraps (``foundation_models.py:53-57``), LoRA
(``lora.py:8-33``) and the hot part of ``ModelModule.training_step`` (``/root/reference/src/models.py:87-143``).

Data layout in HBM
  * tokens: rows of a [B*N, D] matrix; residual stream f32, every GEMM operand bf16 (f32 accumulate on MFMA)
  * frozen weights: bf16 [out, in] (K-contiguous "B^T" operands) plus their transposes for the dgrad GEMMs,
    SwiGLU fc1 rows interleaved in groups of 32 (a|b) so the gate is applied in the fc1 epilogue
  * decoder activations: NHWC bf16; every Fusion_Block input is one "concat" buffer [B,H,W,Cskip+Cup] written in
    place by its producers (no torch.cat), BatchNorm+ReLU of a producer is applied by the consumer-side gather
  * trainable parameters (LoRA + decoder) live in ONE flat f32 buffer (module parameters are views into it) with a
    matching flat gradient buffer: one clip+Adam launch and at most two all-reduce buckets per step.
There is no fallback path: without the HIP library / a ROCm device this raises.
"""
from __future__ import annotations

import math
from types import SimpleNamespace as NS

import torch

from . import _lib as L
from . import ops
from .ops import (A_CONV3, A_CONV3_T, A_PATCH, ACCUM_BF16, ATOMIC, EPI_DGELU, EPI_DSWIGLU, EPI_GELU, EPI_PATCH, EPI_RESID,
                  EPI_STATS, EPI_STORE, EPI_SWIGLU, OUT_F32, RELU)
from .resample import taps

NSLOTS = ops.STAT_SLOTS          # statistic slots per BatchNorm layer (32; 256 = one per writer block with MIPHEI_DETERMINISTIC=1)
BN_EPS, BN_MOM = 1e-5, 0.1
CONV_CH = (3, 48, 96, 192)
FUS_OUT = (256, 128, 64, 32)
HEAD_C, HEAD_HID = 32, 16


def _pad8(n):
    return (n + 7) // 8 * 8


def swiglu_pack_index(hidden):
    """Row order of the packed fc1 weight: per 32 gate columns, 32 'a' rows then their 32 'b' rows."""
    H = hidden // 2
    g = torch.arange(H)
    idx = torch.empty(hidden, dtype=torch.long)
    idx[(g // 32) * 64 + g % 32] = g
    idx[(g // 32) * 64 + 32 + g % 32] = H + g
    return idx


class HipEngine:
    def __init__(self, model):
        self.model = model
        self._flat = None
        self._opt_stash = None       # Adam state carried across re-flattening (see invalidate)
        self._nonfinite = None       # device int32: sticky NaN-guard flag written by the Adam kernel
        self.lora_group = None       # ViT blocks whose LoRA weight-gradient products share one launch (_encoder_bwd).  None = automatic:
        #                              10 when a gradient exchange consumes the blocks group by group (on_lora_block_done), otherwise all
        #                              blocks in one pair of launches (385 us against 436 us for four groups of 10 at the benchmark shape,
        #                              +0.3 % on the step)
        self.use_chunked_conv = True  # fusion blocks 0-2 on the chunked direct convolution (False: implicit GEMM, for A/B runs)
        self.attn_residual = True     # keep the bf16 rounding residual of the attention output for the backward's D term (A/B switch)
        self.bn_fold = True           # eval mode: BatchNorm of the ConvStream folded into its convolutions (_bn_fold; A/B switch)
        self._stats_gen = 0           # train-mode forwards so far: the kernels update the running statistics through raw pointers,
        #                               which tensor._version does not see (key of the eval-mode fold)
        self.invalidate()

    # ------------------------------------------------------------------ state management
    def invalidate(self):
        """Drop everything derived from the module's tensors (``.to()`` / ``load_state_dict`` / ``set_input_size`` replace
        or change them).  The optimiser state is NOT derived from them: Adam's moments and step count are stashed, keyed by
        the flat layout, and put back when the flat buffer is rebuilt (torch.optim state survives those calls too)."""
        if self._flat is not None and self._flat.m is not None:
            self._opt_stash = self.optimizer_state_dict()
        self._frozen = None
        self._flat = None
        self._ws = {}
        self._pack_key = None
        self._fold_key = None
        self._saved = None

    def operand_mode(self):
        """"f16" when the module's parameters are fp16 (`generator.eval().cuda().half()`, the reference's evaluation convention,
        /root/reference/evaluation/eval_orion.py:191, 214-215): the forward then runs on libmiphei_hip_f16.so -- IEEE fp16 operands on
        v_mfma_f32_*_f16, the weights exactly as the module holds them -- instead of re-rounding them to bf16.  Everything else is
        the bf16 library.  A change of mode drops every derived cache."""
        try:
            p = next(self.model.encoder.vit.parameters())
        except StopIteration:
            return "bf16"
        mode = "f16" if p.dtype == torch.float16 else "bf16"
        if mode != getattr(self, "_mode_cached", mode):
            self.invalidate()
        self._mode_cached = mode
        return mode

    def params_changed(self):
        """The trainable parameters were rewritten in place through the flat buffer (Adam step, a rank-0 broadcast): such writes do not
        bump ``tensor._version``, so both derived caches -- the bf16 packs and the eval-mode BatchNorm fold, whose key would otherwise
        come out identical -- are dropped explicitly."""
        self._pack_key = None
        self._fold_key = None

    @property
    def device(self):
        return next(self.model.parameters()).device

    def _require_gpu(self):
        dev = self.device
        if dev.type != "cuda":
            raise RuntimeError("the MIPHEI-ViT HIP engine needs the model on a ROCm device (model.cuda()); "
                               "there is no CPU fallback")
        return dev

    def _config(self):
        vit = self.model.encoder.vit
        g = vit.patch_embed.grid_size
        S = vit.patch_embed.img_size
        if g[0] != g[1] or S[0] != S[1]:
            raise NotImplementedError("square tiles only")
        qkv0 = vit.blocks[0].attn.qkv
        lora = hasattr(qkv0, "lora_q")
        c = NS(D=vit.embed_dim, L=len(vit.blocks), H=vit.num_heads, patch=vit.patch_embed.patch_size[0], grid=g[0], S=S[0],
               nreg=vit.reg_tokens, prefix=vit.num_prefix_tokens, swiglu=vit.mlp_type == "swiglu", hidden=vit.hidden,
               eps=vit.ln_eps, lora=lora, dec=getattr(self.model, "decoder", None) is not None)
        c.NH = self.model.decoder.num_heads if c.dec else 0
        c.Dh = c.D // c.H
        c.ntok = c.grid * c.grid + c.prefix
        c.Hg = c.hidden // 2 if c.swiglu else c.hidden
        if lora:
            c.rank = qkv0.lora_q.rank
            c.alpha = float(qkv0.lora_q.alpha)
            if 2 * c.rank > 16 or (2 * c.rank) % 8:
                raise NotImplementedError("LoRA rank must be 4 or 8")
        if c.D % 8 or c.Dh % 8 or c.Dh > 64:
            raise NotImplementedError("embed_dim % 8 == 0 and head_dim in {8..64} required")
        if c.swiglu and c.hidden % 128:
            raise NotImplementedError("SwiGLU hidden size must be a multiple of 128")
        if c.NH > 16:
            raise NotImplementedError("at most 16 output heads")
        return c

    # ------------------------------------------------------------------ flat trainable parameters
    def _heads(self):
        dec = self.model.decoder
        return [getattr(dec, f"segmentation_head_{i}") for i in range(dec.num_heads)]

    def _ensure_flat(self):
        if self._flat is not None:
            return self._flat
        dev = self._require_gpu()
        c = self._config()
        vit, dec = self.model.encoder.vit, getattr(self.model, "decoder", None)
        groups = []  # (name, [params]) in flat order
        if c.lora:
            for blk in vit.blocks:
                q = blk.attn.qkv
                groups += [q.lora_q.A, q.lora_q.B, q.lora_v.A, q.lora_v.B]
        n_lora = sum(p.numel() for p in groups)
        head_off, heads, convs = {}, [], []
        if dec is not None:
            convs = [cv for cv in dec.convstream.convs] + [fb.conv for fb in dec.fusion_blks]
            for cv in convs:
                groups += [cv.conv.weight, cv.bn.weight, cv.bn.bias]
            heads = self._heads()
            stacked = [("W1", lambda h: h[0].psi[0].weight), ("b1", lambda h: h[0].psi[0].bias),
                       ("bnw", lambda h: h[0].psi[1].weight), ("bnb", lambda h: h[0].psi[1].bias),
                       ("W2", lambda h: h[0].psi[3].weight), ("b2", lambda h: h[0].psi[3].bias),
                       ("W3", lambda h: h[1].weight), ("b3", lambda h: h[1].bias)]
            off = sum(p.numel() for p in groups)
            for name, get in stacked:
                head_off[name] = off
                for h in heads:
                    groups.append(get(h))
                    off += get(h).numel()
        for p in groups:
            if p.dtype != torch.float32:
                raise RuntimeError("training needs fp32 master parameters (do not call .half()/.bfloat16() on a model "
                                   "you train)")
        n = sum(p.numel() for p in groups)
        flat = torch.empty(n, device=dev, dtype=torch.float32)
        gflat = torch.zeros(n, device=dev, dtype=torch.float32)
        o = 0
        gview = {}
        for p in groups:
            k = p.numel()
            flat[o:o + k].copy_(p.data.reshape(-1))
            p.data = flat[o:o + k].view(p.shape)
            gview[id(p)] = gflat[o:o + k].view(p.shape)
            p.grad = gview[id(p)] if p.requires_grad else None
            o += k
        f = NS(flat=flat, gflat=gflat, n=n, n_lora=n_lora, params=groups, m=None, v=None, step=0, gview=gview)
        names = {id(p): k for k, p in self.model.named_parameters()}
        f.layout = [(names.get(id(p), f"param{i}"), p.numel()) for i, p in enumerate(groups)]
        if self._opt_stash is not None:
            self._restore_optimizer(f, self._opt_stash)
            self._opt_stash = None
        NH = c.NH

        def hv(buf, name, shape):
            k = int(torch.tensor(shape).prod())
            return buf[head_off[name]:head_off[name] + k].view(shape)

        for buf, pre in (((flat, ""), (gflat, "d")) if dec is not None else ()):
            setattr(f, pre + "W1", hv(buf, "W1", (NH, HEAD_HID, HEAD_C)))
            setattr(f, pre + "b1", hv(buf, "b1", (NH * HEAD_HID,)))
            setattr(f, pre + "bnw", hv(buf, "bnw", (NH * HEAD_HID,)))
            setattr(f, pre + "bnb", hv(buf, "bnb", (NH * HEAD_HID,)))
            setattr(f, pre + "W2", hv(buf, "W2", (NH, HEAD_HID)))
            setattr(f, pre + "b2", hv(buf, "b2", (NH,)))
            setattr(f, pre + "W3", hv(buf, "W3", (NH, HEAD_C, 9)))
            setattr(f, pre + "b3", hv(buf, "b3", (NH,)))
        if c.lora:
            per = 4 * c.rank * c.D
            for buf, pre in ((flat, ""), (gflat, "d")):
                reg = buf[:n_lora].view(c.L, 4, c.rank * c.D)
                setattr(f, pre + "Aq", reg[:, 0].view(c.L, c.D, c.rank))
                setattr(f, pre + "Bq", reg[:, 1].view(c.L, c.rank, c.D))
                setattr(f, pre + "Av", reg[:, 2].view(c.L, c.D, c.rank))
                setattr(f, pre + "Bv", reg[:, 3].view(c.L, c.rank, c.D))
            assert per * c.L == n_lora
        if dec is None:   # encoder-only engine (UNETR baseline: its decoder parameters live in UnetrEngine)
            f.convs = []
            self._flat = f
            return f
        # running statistics: heads stacked, all num_batches_tracked share one tensor
        bns = [cv.bn for cv in convs] + [h[0].psi[1] for h in heads]
        nbt = torch.stack([b.num_batches_tracked.to(dev) for b in bns]).contiguous()
        for i, b in enumerate(bns):
            b.num_batches_tracked = nbt[i]
        f.nbt = nbt
        rm = torch.cat([h[0].psi[1].running_mean.detach().float() for h in heads]).to(dev).contiguous()
        rv = torch.cat([h[0].psi[1].running_var.detach().float() for h in heads]).to(dev).contiguous()
        for i, h in enumerate(heads):
            h[0].psi[1].running_mean = rm[i * HEAD_HID:(i + 1) * HEAD_HID]
            h[0].psi[1].running_var = rv[i * HEAD_HID:(i + 1) * HEAD_HID]
        f.head_rm, f.head_rv = rm, rv
        f.convs = convs
        self._flat = f
        return f

    # ------------------------------------------------------------------ frozen encoder weights -> bf16 operands
    def _ensure_frozen(self):
        if self._frozen is not None:
            return self._frozen
        dev = self._require_gpu()
        c = self._config()
        vit = self.model.encoder.vit
        bf = L.operand_torch_dtype()

        def w16(t):
            return t.detach().to(device=dev, dtype=bf).contiguous()

        def f32(t):
            return t.detach().to(device=dev, dtype=torch.float32).contiguous()

        fz = NS(blocks=[])
        # patch-embed weight in the K order of the in-kernel window gather (MVIT_A_PATCH): k = (dy * patch + dx) * 8 + c over the
        # 8-channel NHWC image (3 colour channels, 5 zero)
        wp = vit.patch_embed.proj.weight.detach().to(dev)                      # [D, 3, patch, patch]
        wpp = torch.zeros(c.D, c.patch, c.patch, 8, device=dev, dtype=bf)
        wpp[..., :3] = wp.permute(0, 2, 3, 1)
        fz.wpatch, fz.bpatch = wpp.reshape(c.D, c.patch * c.patch * 8).contiguous(), f32(vit.patch_embed.proj.bias)
        fz.pos = f32(vit.pos_embed.reshape(-1, c.D))
        fz.cls = f32(vit.cls_token.reshape(-1))
        fz.reg = f32(vit.reg_token.reshape(-1, c.D)) if c.nreg else fz.cls
        fz.nw, fz.nb = f32(vit.norm.weight), f32(vit.norm.bias)
        idx = swiglu_pack_index(c.hidden).to(dev) if c.swiglu else None
        for blk in vit.blocks:
            lin = blk.attn.qkv.qkv if c.lora else blk.attn.qkv
            b = NS()
            b.n1w, b.n1b, b.n2w, b.n2b = f32(blk.norm1.weight), f32(blk.norm1.bias), f32(blk.norm2.weight), f32(blk.norm2.bias)
            b.wqkv, b.bqkv = w16(lin.weight), f32(lin.bias)
            b.wproj, b.bproj = w16(blk.attn.proj.weight), f32(blk.attn.proj.bias)
            w1, b1 = blk.mlp.fc1.weight.detach().to(dev), blk.mlp.fc1.bias.detach().to(dev)
            if c.swiglu:
                w1, b1 = w1[idx], b1[idx]
            b.wfc1, b.bfc1 = w16(w1), f32(b1)
            b.wfc2, b.bfc2 = w16(blk.mlp.fc2.weight), f32(blk.mlp.fc2.bias)
            b.ls1, b.ls2 = f32(blk.ls1.gamma), f32(blk.ls2.gamma)
            b.t = None
            fz.blocks.append(b)
        self._frozen = fz
        return fz

    def _ensure_frozen_bwd(self):
        fz = self._ensure_frozen()
        for b in fz.blocks:
            if b.t is None:
                b.t = NS(wqkv=b.wqkv.t().contiguous(), wproj=b.wproj.t().contiguous(), wfc1=b.wfc1.t().contiguous(),
                         wfc2=b.wfc2.t().contiguous())
        return fz

    # ------------------------------------------------------------------ per-step packs of the trainable weights
    def _fus3_perm(self, cin, dev):
        """[up | img] channel order of the last fusion block, built once per device: a host tensor moved with .to(dev) inside the
        step is a pageable copy that waits for the whole stream (the step's prologue would run launch-bound every step)"""
        key = (cin, str(dev))
        cache = self.__dict__.setdefault("_perm_cache", {})
        if key not in cache:
            cache[key] = torch.cat([torch.arange(3, cin, device=dev), torch.arange(0, 3, device=dev)])
        return cache[key]

    def _pack_trainable(self, need_bwd):
        c = self._config()
        dev = self._require_gpu()
        dec = getattr(self.model, "decoder", None)
        vit = self.model.encoder.vit
        params = (list(dec.parameters()) if dec is not None else [vit.pos_embed, vit.cls_token]) + ([p for blk in vit.blocks for p in blk.attn.qkv.lora_q.parameters()] +
                                           [p for blk in vit.blocks for p in blk.attn.qkv.lora_v.parameters()] if c.lora else [])
        key = (tuple(p._version for p in params), tuple(p.data_ptr() for p in params[:4]), need_bwd)
        if self._pack_key == key:
            return self._pack
        bf = L.operand_torch_dtype()
        pk = NS()
        f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32)
        fl = self._flat
        if c.lora:
            r, D = c.rank, c.D
            if fl is not None and need_bwd:
                # training: every bf16 operand of the adapters from the flat parameter region in ONE launch
                pk.AcatT = torch.empty(c.L, 2 * r, D, device=dev, dtype=bf)            # [L, 2r, D]: B operand of t = h @ A
                pk.Acat16 = torch.empty(c.L, D, 2 * r, device=dev, dtype=bf)           # B2 operand of the dh1 GEMM
                pk.B2 = torch.empty(c.L, 3 * D, 2 * r, device=dev, dtype=bf)           # [L, 3D, 2r]: qkv K-extension
                bqv = torch.empty(c.L, 2, r, D, device=dev, dtype=bf)                  # alpha * Bq, alpha * Bv
                ops.lora_pack(fl.flat[:fl.n_lora], pk.AcatT, pk.Acat16, pk.B2, bqv, c.L, D, r, c.alpha)
                pk.Bq16, pk.Bv16 = bqv[:, 0], bqv[:, 1]                                 # [L, r, D] views (row stride D)
            else:
                if fl is not None:
                    Aq, Av, Bq, Bv = fl.Aq, fl.Av, fl.Bq, fl.Bv
                else:
                    Aq = torch.stack([f32(b.attn.qkv.lora_q.A) for b in vit.blocks])
                    Av = torch.stack([f32(b.attn.qkv.lora_v.A) for b in vit.blocks])
                    Bq = torch.stack([f32(b.attn.qkv.lora_q.B) for b in vit.blocks])
                    Bv = torch.stack([f32(b.attn.qkv.lora_v.B) for b in vit.blocks])
                Acat = torch.cat([Aq, Av], dim=2).contiguous()                         # [L, D, 2r] f32
                B2 = torch.zeros(c.L, 3 * D, 2 * r, device=dev, dtype=bf)
                B2[:, :D, :r] = (c.alpha * Bq).transpose(1, 2)
                B2[:, 2 * D:, r:] = (c.alpha * Bv).transpose(1, 2)
                pk.B2 = B2
                pk.AcatT = Acat.transpose(1, 2).to(bf).contiguous()
                if not need_bwd:
                    # inference: fold the adapters into the packed projection, W' = W + alpha * (A B)^T on the q and v rows
                    # (SURVEY.md section 8f row 1 "LoRA merge"); the rank-2r K extension and the x @ A product disappear
                    pk.wqkv_merged = []
                    for l, blk in enumerate(vit.blocks):
                        wm = f32(blk.attn.qkv.qkv.weight).clone()
                        wm[:D] += c.alpha * (Aq[l] @ Bq[l]).t()
                        wm[2 * D:] += c.alpha * (Av[l] @ Bv[l]).t()
                        pk.wqkv_merged.append(wm.to(bf).contiguous())
                else:
                    pk.Acat16 = Acat.to(bf)
                    pk.Bq16, pk.Bv16 = (c.alpha * Bq).to(bf).contiguous(), (c.alpha * Bv).to(bf).contiguous()
        if dec is None:   # encoder-only engine (bare registry model: embedding extraction)
            self._pack_key, self._pack = key, pk
            return pk
        convs = [cv for cv in dec.convstream.convs] + [fb.conv for fb in dec.fusion_blks]
        pk.wk, pk.wd, pk.cin, pk.cin_pad, pk.perm = [], [], [], [], []
        pk.wdir_f = pk.wdir_b = None
        pk.wch_f, pk.wch_b = {}, {}          # fusion blocks 0-2: operands of the chunked direct convolution (forward / input gradient)
        packs, cpacks = [], []
        for i, cv in enumerate(convs):
            w = f32(cv.conv.weight).contiguous()                                    # [Cout, Cin, 3, 3]
            cout, cin = w.shape[0], w.shape[1]
            last = i == len(convs) - 1                                              # fus3: internal order [up(64) | img(3)]
            perm = self._fus3_perm(cin, dev) if last else None
            cp = _pad8(cin)
            chunked = self.use_chunked_conv and 3 <= i < len(convs) - 1 and cin % 8 == 0 and cout % 8 == 0
            if chunked:
                # wide fusion blocks (1728 -> 256, 352 -> 128, 176 -> 64): direct convolution on LDS-staged tiles, input channels in
                # chunks of 32, output channels in slices of 64 (csrc/conv_chunked.hip); no implicit-GEMM weight layouts needed
                # (packed by ONE launch below, into buffers that persist across steps)
                bufs = self.__dict__.setdefault("_wch_bufs", {})
                for dg in ((False, True) if need_bwd else (False,)):
                    key_ = (i, dg, tuple(w.shape), str(dev), bf)
                    if key_ not in bufs:
                        bufs[key_] = torch.empty(ops.conv3x3_chunked_pack_elems(w, dg), device=dev, dtype=bf)
                    (pk.wch_b if dg else pk.wch_f)[i] = bufs[key_]
                    cpacks.append((w, bufs[key_], dg))
                wk = wd = None
            else:
                wk = torch.empty(cout, 9 * cp, device=dev, dtype=bf)
                wd = torch.empty(cp, 9 * cout, device=dev, dtype=bf) if need_bwd else None
                packs.append((w, wk, wd, 3 if last else 0))
            pk.wk.append(wk)
            if need_bwd:
                pk.wd.append(wd)
            if last and ops.conv3x3_direct_supported(cp, cout):
                # last fusion block (67 -> 32 at full resolution): direct convolution with LDS-staged halo tiles, forward and
                # (64 up-sampled channels only) input gradient
                pk.wdir_f = ops.pack_conv3x3_direct(w, cout, cp, rot=3)
                if need_bwd and ops.conv3x3_direct_supported(_pad8(cout), FUS_OUT[2]):
                    pk.wdir_b = ops.pack_conv3x3_direct(w, FUS_OUT[2], _pad8(cout), rot=3, dgrad=True)
            pk.cin.append(cin)
            pk.cin_pad.append(cp)
            pk.perm.append(perm)
        ops.pack_conv3x3_weights_multi(packs)      # all implicit-GEMM layouts in one launch
        if cpacks:
            ops.pack_conv3x3_chunked_multi(cpacks)  # all chunked-kernel operands in one launch (persistent buffers)
        heads = self._heads()
        st = lambda get, shape: torch.stack([f32(get(h)).reshape(-1) for h in heads]).reshape(shape).contiguous()
        NH = c.NH
        if fl is not None:
            pk.W1, pk.b1, pk.bnw, pk.bnb, pk.W2, pk.b2, pk.b3 = fl.W1, fl.b1, fl.bnw, fl.bnb, fl.W2, fl.b2, fl.b3
            pk.W3k = fl.W3.transpose(1, 2).contiguous()
        else:
            pk.W1 = st(lambda h: h[0].psi[0].weight, (NH, HEAD_HID, HEAD_C))
            pk.b1 = st(lambda h: h[0].psi[0].bias, (NH * HEAD_HID,))
            pk.bnw = st(lambda h: h[0].psi[1].weight, (NH * HEAD_HID,))
            pk.bnb = st(lambda h: h[0].psi[1].bias, (NH * HEAD_HID,))
            pk.W2 = st(lambda h: h[0].psi[3].weight, (NH, HEAD_HID))
            pk.b2 = st(lambda h: h[0].psi[3].bias, (NH,))
            pk.W3k = st(lambda h: h[1].weight, (NH, HEAD_C, 9)).transpose(1, 2).contiguous()   # [NH, 9, 32]
            pk.b3 = st(lambda h: h[1].bias, (NH,))
        pk.bn = [NS(w=f32(cv.bn.weight).contiguous(), b=f32(cv.bn.bias).contiguous()) for cv in convs]
        self._pack_key, self._pack = key, pk
        return pk

    # ------------------------------------------------------------------ workspaces
    def _workspace(self, B, train):
        c = self._config()
        key = (B, c.S, train)
        if key in self._ws:
            return self._ws[key]
        dev = self._require_gpu()
        bf = L.operand_torch_dtype()
        e = lambda *s, dt=bf: torch.empty(*s, device=dev, dtype=dt)
        z = lambda *s, dt=bf: torch.zeros(*s, device=dev, dtype=dt)
        M, D, S = B * c.ntok, c.D, c.S
        w = NS(B=B, M=M)
        w.img8 = e(B, S, S, 8)            # bf16 NHWC image (3 colour channels + 5 zero): patch-embed gather and decoder operand
        nl = c.L if train else 1
        w.x_in = [e(M, D, dt=torch.float32) for _ in range(nl + 1)] if train else [e(M, D, dt=torch.float32)]
        w.x_mid = [e(M, D, dt=torch.float32) for _ in range(nl)]
        # per-block saves of one kind sit in ONE allocation at a uniform stride: the batched LoRA weight-gradient products
        # (_encoder_bwd) address block l's operand as base + l * stride
        w.h1_all = e(nl, M, D)
        w.h1 = list(w.h1_all.unbind(0))
        w.h2 = e(M, D)
        w.t_all = e(nl, M, 2 * c.rank) if c.lora else None
        w.t = list(w.t_all.unbind(0)) if c.lora else None
        w.qkv = [e(M, 3 * D) for _ in range(nl)]
        w.o = [e(M, D) for _ in range(nl)]
        # bf16 rounding residual of o (D term of the attention backward): allocated by the first training forward that runs with
        # attn_residual on (650 MB at B = 16 / 40 blocks), see _attn_res()
        w.ores = None
        w.ores_on = False     # latched by every forward: the backward reads the residual iff the forward of THIS step wrote it
        w.lse = [e(B, c.H, c.ntok, dt=torch.float32) for _ in range(nl)]
        w.u = [e(M, c.hidden) for _ in range(nl)] if train else None
        w.g = e(M, c.Hg)
        w.tok = e(M, D)
        if not c.dec:
            if train:
                self._alloc_encoder_bwd(w, c, e, z)
            self._ws[key] = w
            return w
        G = S // 16
        s1, s2, s3 = S // 2, S // 4, S // 8
        w.res = (S, s1, s2, s3, G)
        w.feat = e(B, G, G, D)
        w.cat = [e(B, s3, s3, CONV_CH[3] + D), e(B, s2, s2, CONV_CH[2] + FUS_OUT[0]), e(B, s1, s1, CONV_CH[1] + FUS_OUT[1]),
                 z(B, S, S, _pad8(CONV_CH[0] + FUS_OUT[2]))]
        if max(t.numel() for t in w.cat) * 2 >= 2 ** 31:
            # the decoder kernels address their NHWC buffers with 32-bit byte offsets (raw buffer descriptors); the implicit-GEMM
            # fallback has the same limit and its weight layouts are not even packed when the chunked kernels are on
            raise ValueError(f"decoder buffers of batch {B} at {S}x{S} exceed the 2 GiB range of 32-bit buffer offsets: split the "
                             "batch (at most 372 tiles at 256 px, 93 at 512 px)")
        w.pre_c = [e(B * s1 * s1, 48), e(B * s2 * s2, 96), e(B * s3 * s3, 192)]
        w.pre_f = [e(B * s3 * s3, 256), e(B * s2 * s2, 128), e(B * s1 * s1, 64), e(B * S * S, 32)]
        w.F3 = e(B * S * S, HEAD_C)
        w.G = e(B * S * S, 16)
        w.out = e(B, c.NH, S, S, dt=torch.float32)
        chans = [48, 96, 192, 256, 128, 64, 32]
        nch = c.NH * HEAD_HID
        w.bnp = [NS(scale=e(ch, dt=torch.float32), shift=e(ch, dt=torch.float32), mean=e(ch, dt=torch.float32),
                    rstd=e(ch, dt=torch.float32)) for ch in chans]
        w.hbn = NS(scale=e(nch, dt=torch.float32), shift=e(nch, dt=torch.float32), mean=e(nch, dt=torch.float32),
                   rstd=e(nch, dt=torch.float32))
        # f64 arena: forward stats | heads moments | mom_sum | (backward) stats | loss | sqnorm
        sizes = [NSLOTS * 2 * ch for ch in chans] + [NSLOTS * (32 + 1024), 32 + 1024]
        w.arena_f = z(sum(sizes), dt=torch.float64)
        offs = [0]
        for s_ in sizes:
            offs.append(offs[-1] + s_)
        w.stats_f = [w.arena_f[offs[i]:offs[i + 1]] for i in range(7)]
        w.mom = w.arena_f[offs[7]:offs[8]]
        w.mom_sum = w.arena_f[offs[8]:offs[9]]
        if train:
            # everything the backward pass accumulates into lives in ONE zero-filled byte buffer (one fill per step):
            # f64 BN-backward statistics | f32 weight-gradient scratch of the TN GEMMs | head-bias slots
            pixc = [8, 48, 96, _pad8(192 + D), _pad8(96 + 256), _pad8(48 + 128), _pad8(3 + 64)]
            wsz = [9 * cp * ch for cp, ch in zip(pixc, chans)] + [c.NH * 9 * HEAD_C]
            sizes_b = [NSLOTS * 2 * ch for ch in chans]
            nb64, nw32, ns32 = sum(sizes_b), sum(wsz), 64 * 32
            w.zbuf = z(nb64 * 8 + (nw32 + ns32) * 4, dt=torch.uint8)
            w.arena_b = w.zbuf[:nb64 * 8].view(torch.float64)
            w.wscr = w.zbuf[nb64 * 8:nb64 * 8 + nw32 * 4].view(torch.float32)
            w.db3_slots = w.zbuf[nb64 * 8 + nw32 * 4:].view(torch.float32).view(64, 32)
            ob = [0]
            for s_ in sizes_b:
                ob.append(ob[-1] + s_)
            w.stats_b = [w.arena_b[ob[i]:ob[i + 1]] for i in range(7)]
            w.scal = z(2, dt=torch.float64)          # loss accumulator | gradient square norm (one fill, in loss_and_grad)
            w.loss_acc, w.sqn = w.scal[0:1], w.scal[1:2]
            # gradients / scratch
            w.dY = e(B, c.NH, S, S, dt=torch.float32)
            w.cscr = e(ops.heads_conv_bwd_scratch_bytes(B * S * S) // 4, dt=torch.float32)
            w.dG = e(B * S * S, 16, dt=torch.float32)
            w.dXc = e(B * S * S, HEAD_C, dt=torch.float32)
            w.dF3 = e(B * S * S, HEAD_C)
            w.hscr = e(ops.heads_gate_bwd_scratch_bytes() // 4, dt=torch.float32)
            w.dpre_c = [e(B * s1 * s1, 48), e(B * s2 * s2, 96), e(B * s3 * s3, 192)]
            w.dpre_f = [e(B * s3 * s3, 256), e(B * s2 * s2, 128), e(B * s1 * s1, 64), e(B * S * S, 32)]
            w.dcat = [e(B * s3 * s3, CONV_CH[3] + D), e(B * s2 * s2, CONV_CH[2] + FUS_OUT[0]),
                      e(B * s1 * s1, CONV_CH[1] + FUS_OUT[1]), e(B * S * S, FUS_OUT[2])]
            w.dFpost = [e(B * s3 * s3, 256), e(B * s2 * s2, 128), e(B * s1 * s1, 64)]
            w.dfeat = e(B, G, G, D)
            ow = [0]
            for s_ in wsz:
                ow.append(ow[-1] + s_)
            w.dWt = [w.wscr[ow[i]:ow[i + 1]].view(9 * pixc[i], chans[i]) for i in range(7)]
            w.dW3 = w.wscr[ow[7]:ow[8]].view(c.NH * 9, HEAD_C)
            self._alloc_encoder_bwd(w, c, e, z)
        self._ws[key] = w
        return w

    @staticmethod
    def _alloc_encoder_bwd(w, c, e, z):
        M, D, B = w.M, c.D, w.B
        w.dtok = z(M, D)
        w.dx = e(M, D, dt=torch.float32)
        w.dy = e(M, D)
        w.du = e(M, c.hidden)
        w.dh = e(M, D)
        w.do = e(M, D)
        # d(qkv) and d(t) of every block are kept until the block's group has run its batched weight-gradient products
        # (L x 48.5 MB at B = 16: 1.9 GB of the 288)
        w.dqkv_all = e(c.L if c.lora else 1, M, 3 * D)
        w.dsum = e(B, c.H, c.ntok, dt=torch.float32)
        w.dt_all = e(c.L, M, 2 * c.rank) if c.lora else None

    # ------------------------------------------------------------------ forward
    def _drop_path_factors(self, w, c):
        """Per-row DropPath factors of this step, [L, 2, M] f32 (branch 0 = attention, 1 = MLP), or None.

        timm builds the blocks with ``drop_path = linspace(0, drop_path_rate, depth)[l]`` and applies, in train mode,
        ``x = x + drop_path(ls(branch(norm(x))))`` with one Bernoulli(keep) draw per SAMPLE, kept samples scaled by 1/keep
        (stochastic depth; reference: ``drop_path_rate=drop_rate`` at src/generators/unet.py:37 -> foundation_models.py:53-57).
        The draws are torch RNG on the device (plumbing); the factors are consumed inside the residual epilogue of the proj / fc2
        GEMMs and, for the gradient, inside the LayerNorm-backward kernel that emits the branch gradient."""
        vit = self.model.encoder.vit
        rate = float(getattr(vit, "drop_path_rate", 0.0) or 0.0)
        if rate <= 0.0 or not vit.training:
            return None
        keep = 1.0 - torch.linspace(0.0, rate, c.L, device=w.tok.device).view(c.L, 1, 1)
        draw = (torch.rand(c.L, 2, w.B, device=w.tok.device) < keep).float() / keep
        return draw.repeat_interleave(c.ntok, dim=2).contiguous()

    def _attn_res(self, w, i, train):
        """residual buffer of block i for this forward (None = off).  The A/B switch ``attn_residual`` is latched per forward in
        ``w.ores_on`` so that a toggle between forward and backward cannot make the backward read an unwritten residual."""
        on = bool(train and self.attn_residual and self._config().lora)   # (only the encoder backward reads it)
        if i == 0:
            w.ores_on = on
        if not w.ores_on:
            return None
        if w.ores is None:
            w.ores = [torch.empty(w.M, self._config().D, device=w.tok.device, dtype=L.operand_torch_dtype()) for _ in range(len(w.o))]
        return w.ores[i]

    def _encoder_fwd(self, w, x, train, pk, taps=None, img8=None):
        """taps: {block index: bf16 [M, D] buffer} receives the residual stream after that block (forward_intermediates);
        img8: bf16 NHWC [B,S,S,8] image already written by the input stage (else converted from x here)"""
        c, fz = self._config(), self._ensure_frozen()
        B, M, D = w.B, w.M, c.D
        w.dpath = self._drop_path_factors(w, c) if train else None
        dp = w.dpath
        P = c.grid * c.grid
        if img8 is not None:
            if img8.dtype != L.operand_torch_dtype() or tuple(img8.shape) != tuple(w.img8.shape) or not img8.is_contiguous():
                raise ValueError(f"img8: expected a contiguous bf16 tensor {tuple(w.img8.shape)}")
            im8 = img8            # read in place (the decoder and its backward use the same buffer)
        else:
            im8 = w.img8
            ops.image_to_nhwc(x, im8, 8, nzero=5)
        w.img8_cur = im8
        X = w.x_in[0]
        ops.prefix_tokens(X, fz.cls, fz.reg, B, c.ntok, D, c.nreg)
        # patch embedding: 14x14x3 windows gathered from the NHWC image by the GEMM's A loader (no im2col buffer)
        ops.gemm(im8, fz.wpatch, X, M=B * P, amode=A_PATCH, conv=(c.S, c.S, 8, 8, c.grid, c.grid, c.patch), bias=fz.bpatch,
                 pos=fz.pos, epi=EPI_PATCH, patch=(P, c.ntok, c.prefix), flags=OUT_F32)
        scale = c.Dh ** -0.5
        for l, b in enumerate(fz.blocks):
            i = l if train else 0
            xin = w.x_in[l] if train else w.x_in[0]
            xmid = w.x_mid[i]
            xout = w.x_in[l + 1] if train else w.x_in[0]
            if c.lora and not train:
                ops.layernorm_fwd(xin, b.n1w, b.n1b, w.h1[i], c.eps)
                ops.gemm(w.h1[i], pk.wqkv_merged[l], w.qkv[i], bias=b.bqkv)
            elif c.lora:
                # LN1 and the adapters' down-projection t = LN1(x) @ [A_q | A_v] in one pass over the row
                ops.layernorm_lora_fwd(xin, b.n1w, b.n1b, w.h1[i], pk.AcatT[l], w.t[i], c.eps)
                ops.gemm(w.h1[i], b.wqkv, w.qkv[i], bias=b.bqkv, a2=w.t[i], b2=pk.B2[l], K2=2 * c.rank)
            else:
                ops.layernorm_fwd(xin, b.n1w, b.n1b, w.h1[i], c.eps)
                ops.gemm(w.h1[i], b.wqkv, w.qkv[i], bias=b.bqkv)
            ops.attention_fwd(w.qkv[i], w.o[i], w.lse[i], B, c.ntok, c.H, c.Dh, scale, out_res=self._attn_res(w, i, train))
            ops.gemm(w.o[i], b.wproj, xmid, bias=b.bproj, gamma=b.ls1, aux=xin, epi=EPI_RESID, flags=OUT_F32,
                     rowscale=None if dp is None else dp[l, 0])
            ops.layernorm_fwd(xmid, b.n2w, b.n2b, w.h2, c.eps)
            ops.gemm(w.h2, b.wfc1, w.g, bias=b.bfc1, aux=(w.u[l] if train and c.lora else None),   # (saved for the encoder backward only)
                     epi=EPI_SWIGLU if c.swiglu else EPI_GELU)
            ops.gemm(w.g, b.wfc2, xout, bias=b.bfc2, gamma=b.ls2, aux=xmid, epi=EPI_RESID, flags=OUT_F32,
                     rowscale=None if dp is None else dp[l, 1])
            if taps is not None and l in taps:
                ops.cast_bf16(xout, taps[l])
        xf = w.x_in[c.L] if train else w.x_in[0]
        ops.layernorm_fwd(xf, fz.nw, fz.nb, w.tok, c.eps)
        return w.tok

    def _bn(self, w, i, pk, conv_mod, count, bn_train):
        bp, bn = w.bnp[i], conv_mod.bn
        rm, rv = bn.running_mean, bn.running_var
        if rm.dtype != torch.float32:  # .half() / .bfloat16() inference models: the kernels read f32 statistics
            if bn_train:
                raise RuntimeError("train-mode BatchNorm needs fp32 running statistics (model.float())")
            rm, rv = rm.float(), rv.float()
        ops.bn_finalize(w.stats_f[i], pk.bn[i].w, pk.bn[i].b, rm, rv, bp.scale, bp.shift,
                        bp.mean, bp.rstd, pk.bn[i].w.numel(), NSLOTS, count, BN_EPS, BN_MOM, bn_train)

    def _bn_fold(self, pk, convs):
        """Eval-mode BatchNorm as constants (SURVEY.md section 8f row 1 "BN fold", section 8d config 5): with running statistics
        y = scale * conv(x) + shift, scale = weight / sqrt(running_var + eps), shift = bias - running_mean * scale
        (nn.BatchNorm2d.eval() inside Basic_Conv3x3 / Fusion_Block, /root/reference/src/generators/mipheivit.py:20-41, 76-93).
        ConvStream: scale goes into the packed convolution weights and shift + ReLU into the GEMM epilogue (bias, MVIT_RELU), which
        writes the concat slice directly -- no pre-activation buffer, no bn_finalize / bn_relu_apply launches.  Fusion blocks:
        scale / shift are computed here once instead of by one bn_finalize launch per layer and forward (their consumers -- the
        up-sampling gather, the last block's apply pass -- take them as before).  Cached on the parameter and buffer versions."""
        bns = [cv.bn for cv in convs]
        key = (self._pack_key, self._stats_gen, tuple((b.running_mean._version, b.running_var._version, b.running_mean.data_ptr()) for b in bns))
        if self._fold_key == key:
            return self._fold
        dev = self._require_gpu()
        f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32)
        fold = NS(wk=[], scale=[], shift=[])
        packs = []
        for i, cv in enumerate(convs):
            bn = cv.bn
            scale = f32(bn.weight) * torch.rsqrt(f32(bn.running_var) + BN_EPS)
            shift = f32(bn.bias) - f32(bn.running_mean) * scale
            fold.scale.append(scale.contiguous())
            fold.shift.append(shift.contiguous())
            if i < 3:
                wf = (f32(cv.conv.weight) * scale[:, None, None, None]).contiguous()
                wk = torch.empty_like(pk.wk[i])
                packs.append((wf, wk, None, 0))
                fold.wk.append(wk)
        ops.pack_conv3x3_weights_multi(packs)
        self._fold_key, self._fold = key, fold
        return fold

    def _decoder_fwd(self, w, x, bn_train, pk, convs):
        c = self._config()
        B, D = w.B, c.D
        S, s1, s2, s3, G = w.res
        dev = x.device
        mode = "bicubic" if c.patch != 16 else "identity"
        ty = taps(mode, c.grid, G, dev)
        ops.resample2d(w.tok[c.prefix:], w.feat, ty, ty, B=B, h=c.grid, w=c.grid, H=G, W=G, C=D, ld_src=D, ld_dst=D,
                       src_bstride=c.ntok * D, dst_bstride=G * G * D)
        fold = self._bn_fold(pk, convs) if (not bn_train and self.bn_fold) else None
        im8 = w.img8_cur          # written (or adopted from the input stage) by _encoder_fwd
        # (the image slice of the last concat buffer is copied from img8 by the up-sampling kernel that fills the rest of it)
        # ConvStream: conv3x3 s2 -> BN -> ReLU, written into the skip slice of the matching concat buffer
        src = [(im8, S, 8, 8), (w.cat[2], s1, 48, w.cat[2].shape[-1]), (w.cat[1], s2, 96, w.cat[1].shape[-1])]
        dst = [(w.cat[2], s1), (w.cat[1], s2), (w.cat[0], s3)]
        for i in range(3):
            a, r_in, cin, ld = src[i]
            d, r_out = dst[i]
            Mo = B * r_out * r_out
            cout = CONV_CH[i + 1]
            if fold is not None:
                ops.gemm(a, fold.wk[i], d, M=Mo, N=cout, ldc=d.shape[-1], amode=A_CONV3, conv=(r_in, r_in, cin, ld, r_out, r_out, 2),
                         bias=fold.shift[i], flags=RELU)
                continue
            if bn_train:
                ops.gemm(a, pk.wk[i], w.pre_c[i], M=Mo, amode=A_CONV3, conv=(r_in, r_in, cin, ld, r_out, r_out, 2),
                         epi=EPI_STATS, stats=w.stats_f[i], nslots=NSLOTS)
            else:
                ops.gemm(a, pk.wk[i], w.pre_c[i], M=Mo, amode=A_CONV3, conv=(r_in, r_in, cin, ld, r_out, r_out, 2))
            self._bn(w, i, pk, convs[i], Mo, bn_train)
            ops.bn_relu_apply(w.pre_c[i], w.bnp[i].scale, w.bnp[i].shift, d, Mo, cout, cout, d.shape[-1])
        # Fusion blocks: bilinear x2 of the previous stage (BN+ReLU fused into the gather) -> concat slice -> conv3x3
        assert s3 == 2 * G
        ops.upsample2x_bilinear(w.feat, w.cat[0].view(-1)[CONV_CH[3]:], B=B, h=G, w=G, C=D, ld_src=D,
                                ld_dst=w.cat[0].shape[-1], src_bstride=G * G * D, dst_bstride=s3 * s3 * w.cat[0].shape[-1])
        res = [s3, s2, s1, S]
        for j in range(4):
            r = res[j]
            Mo = B * r * r
            cat = w.cat[j]
            cp = cat.shape[-1]
            i = 3 + j
            if j == 3 and pk.wdir_f is not None and cat.numel() * 2 < 2 ** 31:    # (32-bit byte offsets of its raw buffer)
                ops.conv3x3_direct(cat, pk.wdir_f, w.pre_f[j], B=B, H=r, W=r, cin_pad=cp, ldx=cp, cout=FUS_OUT[j], ldy=FUS_OUT[j],
                                   stats=w.stats_f[i] if bn_train else None, nslots=NSLOTS)
            elif i in pk.wch_f and cat.numel() * 2 < 2 ** 31:
                ops.conv3x3_chunked(cat, pk.wch_f[i], w.pre_f[j], B=B, H=r, W=r, cin=cp, ldx=cp, cout=FUS_OUT[j], ldy=FUS_OUT[j],
                                    stats=w.stats_f[i] if bn_train else None, nslots=NSLOTS)
            elif bn_train:
                ops.gemm(cat, pk.wk[i], w.pre_f[j], M=Mo, amode=A_CONV3, conv=(r, r, cp, cp, r, r, 1), epi=EPI_STATS,
                         stats=w.stats_f[i], nslots=NSLOTS)
            else:
                ops.gemm(cat, pk.wk[i], w.pre_f[j], M=Mo, amode=A_CONV3, conv=(r, r, cp, cp, r, r, 1))
            if fold is None:
                self._bn(w, i, pk, convs[i], Mo, bn_train)
            bscale, bshift = (w.bnp[i].scale, w.bnp[i].shift) if fold is None else (fold.scale[i], fold.shift[i])
            if j < 3:
                nxt = w.cat[j + 1]
                off = 0 if j == 2 else CONV_CH[2 - j]
                ops.upsample2x_bilinear(w.pre_f[j], nxt.view(-1)[off:], B=B, h=r, w=r, C=FUS_OUT[j], ld_src=FUS_OUT[j],
                                        ld_dst=nxt.shape[-1], src_bstride=r * r * FUS_OUT[j],
                                        dst_bstride=4 * r * r * nxt.shape[-1], scale=bscale, shift=bshift,
                                        extra8=im8 if j == 2 else None)
            else:
                ops.bn_relu_apply(w.pre_f[3], bscale, bshift, w.F3, Mo, 32, 32, 32)
        # heads
        Mp = B * S * S
        fl = self._flat
        rm, rv = (fl.head_rm, fl.head_rv) if fl is not None else self._head_running()
        if bn_train:
            ops.heads_moments(w.F3, w.mom, Mp, NSLOTS)
        ops.heads_bn_from_moments(w.mom, pk.W1, pk.b1, pk.bnw, pk.bnb, rm, rv, w.hbn.scale, w.hbn.shift, w.hbn.mean,
                                  w.hbn.rstd, w.mom_sum, c.NH, NSLOTS, Mp, BN_EPS, BN_MOM, bn_train)
        ops.heads_gate_fwd(w.F3, pk.W1, pk.b1, w.hbn.scale, w.hbn.shift, pk.W2, pk.b2, w.G, Mp, c.NH)
        ops.heads_conv_fwd(w.F3, w.G, pk.W3k, pk.b3, w.out, B, S, S, c.NH)
        return w.out

    def _head_running(self):
        """stacked running statistics of the head BatchNorms for a model whose parameters were not flattened"""
        dev = self.device
        heads = self._heads()
        rm = torch.cat([h[0].psi[1].running_mean.detach().float() for h in heads]).to(dev).contiguous()
        rv = torch.cat([h[0].psi[1].running_var.detach().float() for h in heads]).to(dev).contiguous()
        return rm, rv

    def forward(self, x, train=False, bn_train=None, img8=None):
        """Generator forward.  train=True keeps the activations the backward pass needs (one graph in flight).
        img8: optional bf16 NHWC [B,S,S,8] copy of x (channels 3..7 zero) as the input stage writes it; replaces the engine's own
        NCHW -> NHWC conversion of the decoder's image operand."""
        mode = self.operand_mode()
        if mode == "f16" and train:
            raise RuntimeError("training needs fp32 master parameters (do not call .half() on a model that trains): the fp16 operand "
                               "library is the evaluation convention generator.eval().cuda().half()")
        with L.operands(mode):
            return self._forward(x, train, bn_train, img8)

    def _forward(self, x, train, bn_train, img8):
        dev = self._require_gpu()
        c = self._config()
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] != c.S or x.shape[3] != c.S:
            raise ValueError(f"expected input [B,3,{c.S},{c.S}], got {tuple(x.shape)}")
        in_dtype = x.dtype
        x = x.detach().to(device=dev, dtype=torch.float32).contiguous()
        if bn_train is None:
            bn_train = self.model.decoder.training
        if train or bn_train or all(p.dtype == torch.float32 for p in self.model.decoder.parameters()):
            self._ensure_flat()
        pk = self._pack_trainable(need_bwd=train)
        w = self._workspace(x.shape[0], train)
        if bn_train:
            w.arena_f.zero_()
        dec = self.model.decoder
        convs = [cv for cv in dec.convstream.convs] + [fb.conv for fb in dec.fusion_blks]
        self._encoder_fwd(w, x, train, pk, img8=img8)
        out = self._decoder_fwd(w, x, bn_train, pk, convs)
        if bn_train:
            self._flat.nbt.add_(1)
            self._stats_gen += 1
        if train:
            self._saved = NS(w=w, pk=pk, x=x, convs=convs, bn_train=bn_train)
        return out if in_dtype == torch.float32 else out.to(in_dtype)

    # ------------------------------------------------------------------ hipGraph-captured inference
    def capture_inference(self, batch):
        """Capture the eval-mode forward for a fixed batch size into a hipGraph (BASELINE config 5).

        Returns (run, x_static, out_static): copy a batch into x_static, call run(), read out_static.  Every launch of
        the forward goes through the C-ABI on the capture stream; nothing allocates or synchronises once the workspace,
        weight packs and tap tables exist, so one warm-up call precedes the capture."""
        dev = self._require_gpu()
        c = self._config()
        x_static = torch.zeros(batch, 3, c.S, c.S, device=dev, dtype=torch.float32)
        self.forward(x_static, train=False, bn_train=False)      # warm-up: allocations, packs, LDS attributes
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out_static = self.forward(x_static, train=False, bn_train=False)
        return graph.replay, x_static, out_static

    # ------------------------------------------------------------------ backward
    def _wgrad(self, w, i, src, r_in, cin_pad, ld, r_out, stride, dpre, cout):
        """dW^T[(ky,kx,c), co] += sum_pixels im2col(X)[m, (ky,kx,c)] * dY[m, co]: TN MFMA GEMM, window gathered on the
        fly from the NHWC input (no im2col / transposed copies), split over the pixel range."""
        B = w.B
        Mo = B * r_out * r_out
        K9 = 9 * cin_pad
        it, jt = (128, 32) if cout <= 32 else (64, 128)
        tiles = ((K9 + it - 1) // it) * ((cout + jt - 1) // jt)
        ms = max(1, min(768 // tiles, (Mo + 255) // 256))     # one round of blocks (3 per CU): 105 us for the three ConvStream layers, 124 at 1024
        ops.gemm_tn(src, dpre, w.dWt[i], M=Mo, I=K9, J=cout, ldb=cout, ldci=cout, msplit=ms,
                    conv=(r_in, r_in, cin_pad, ld, r_out, r_out, stride))

    def backward(self, dY, on_decoder_done=None, on_lora_block_done=None):
        """Gradients of every trainable parameter into the flat gradient buffer (views = param.grad).

        on_decoder_done() is called when the decoder gradients are complete, on_lora_block_done(l) when block l's LoRA
        gradients are (l descending): hooks of the data-parallel exchange, which all-reduces finished slices meanwhile."""
        sv = self._saved
        if sv is None:
            raise RuntimeError("backward() needs a preceding forward(train=True)")
        if not sv.bn_train:
            raise NotImplementedError("backward through eval-mode BatchNorm is not part of the training path")
        # (the transposed frozen weights -- 2.2 GB -- are only built when the encoder backward will run: decoder-only training with a
        #  frozen, adapter-free encoder never reads them)
        c, fl = self._config(), self._ensure_flat()
        fz = self._ensure_frozen_bwd() if c.lora else self._ensure_frozen()
        if not c.lora and any(p.requires_grad for p in self.model.encoder.vit.parameters()):
            # get_vitmatte(use_lora=False) leaves every encoder weight trainable (reference mipheivit.py:224-231: full fine-tuning with
            # layer-wise lr decay, models.py:347-357); that needs the weight gradients of 1.1 G frozen-layout parameters and is outside the
            # LoRA hot path.  With the encoder frozen (requires_grad_(False) on generator.encoder) the decoder trains on fixed features.
            raise NotImplementedError("training a fully unfrozen encoder is outside the MIPHEI-ViT (LoRA) hot path: freeze it "
                                      "(model.encoder.requires_grad_(False): decoder-only training) or build with use_lora=True")
        w, pk, convs = sv.w, sv.pk, sv.convs
        B, D, M = w.B, c.D, w.M
        S, s1, s2, s3, G = w.res
        dev = dY.device
        Mp = B * S * S
        w.zbuf.zero_()
        fl.gflat.zero_()
        w.wgrad_n_major_set = set()      # conv layers whose weight-gradient scratch is output-channel major this step
        dY = dY.to(torch.float32).contiguous()
        # ---- heads
        ops.heads_conv_bwd(dY, w.out, w.F3, w.G, pk.W3k, w.cscr, w.dG, w.dXc, w.dW3, w.db3_slots, B, S, S, c.NH)
        fl.db3.add_(w.db3_slots.sum(0)[:c.NH])
        fl.dW3.add_(w.dW3.view(c.NH, 9, HEAD_C).transpose(1, 2))
        ops.heads_gate_bwd(w.F3, w.G, w.dG, w.dXc, pk.W1, pk.b1, w.hbn.scale, w.hbn.shift, w.hbn.mean, w.hbn.rstd, pk.bnw,
                           pk.W2, w.mom_sum, w.hscr, fl.dW1, fl.dbnw, fl.dbnb, fl.dW2, fl.db2, w.dF3, Mp, c.NH)
        # ---- fusion blocks (reverse)
        res = [s3, s2, s1, S]
        dy_post, ld_post = w.dF3, HEAD_C
        for j in (3, 2, 1, 0):
            i, r = 3 + j, res[j]
            Mo = B * r * r
            cat = w.cat[j]
            cp = cat.shape[-1]
            cout = FUS_OUT[j]
            bp = w.bnp[i]
            bn = convs[i].bn
            ops.bn_relu_bwd(dy_post, ld_post, w.pre_f[j], bp.scale, bp.shift, bp.mean, bp.rstd, pk.bn[i].w, w.stats_b[i],
                            fl.gview[id(bn.weight)], fl.gview[id(bn.bias)], w.dpre_f[j], Mo, cout, NSLOTS)
            if (i in pk.wch_f or (j == 3 and self.use_chunked_conv)) and not ops.DETERMINISTIC and cat.numel() * 2 < 2 ** 31 \
                    and w.dpre_f[j].numel() * 2 < 2 ** 31:
                # weight gradient on the chunked LDS-staged tiles (output-channel-major scratch, see the unpack below); the last
                # fusion block (72 -> 32) too: 127 us against 155 us for conv_direct.hip's whole-weight variant
                ops.conv3x3_chunked_wgrad(cat, w.dpre_f[j], w.dWt[i], B=B, H=r, W=r, cin=cp, cin_pad=cp, ldx=cp, cout=cout, ldy=cout)
                w.wgrad_n_major_set.add(i)
            elif j == 3 and pk.wdir_f is not None and cat.numel() * 2 < 2 ** 31 and (cp, cout) == (72, 32) and not ops.DETERMINISTIC:
                ops.conv3x3_direct_wgrad(cat, w.dpre_f[j], w.dWt[i], B=B, H=r, W=r, cin_pad=cp, ldx=cp, cout=cout, ldy=cout)
                w.wgrad_n_major_set.add(i)
            else:
                self._wgrad(w, i, cat, r, cp, cp, r, 1, w.dpre_f[j], cout)
            # dgrad into the concat-gradient buffer (fus3: only the 64 upsampled channels carry gradient)
            ncols = FUS_OUT[2] if j == 3 else cp
            dcat = w.dcat[j]
            if j == 3 and pk.wdir_b is not None and w.dpre_f[j].numel() * 2 < 2 ** 31:
                ops.conv3x3_direct(w.dpre_f[j], pk.wdir_b, dcat, B=B, H=r, W=r, cin_pad=cout, ldx=cout, cout=ncols,
                                   ldy=dcat.shape[-1])
            elif i in pk.wch_b and w.dpre_f[j].numel() * 2 < 2 ** 31:
                ops.conv3x3_chunked(w.dpre_f[j], pk.wch_b[i], dcat, B=B, H=r, W=r, cin=cout, ldx=cout, cout=ncols, ldy=dcat.shape[-1])
            else:
                ops.gemm(w.dpre_f[j], pk.wd[i], dcat, M=Mo, N=ncols, amode=A_CONV3_T, conv=(r, r, cout, cout, r, r, 1),
                         ldc=dcat.shape[-1])
            if j > 0:
                # adjoint of the bilinear x2 upsample -> gradient w.r.t. relu(BN(pre_f[j-1]))
                off = 0 if j == 3 else CONV_CH[3 - j]
                rp = res[j - 1]
                cprev = FUS_OUT[j - 1]
                ops.upsample2x_bilinear_bwd(dcat.view(-1)[off:], w.dFpost[j - 1], B=B, h=rp, w=rp, C=cprev,
                                            ld_dout=dcat.shape[-1], ld_din=cprev, dout_bstride=r * r * dcat.shape[-1],
                                            din_bstride=rp * rp * cprev)
                dy_post, ld_post = w.dFpost[j - 1], cprev
        # ---- encoder feature gradient: adjoint bilinear (s3 -> G), adjoint regrid (G -> token grid)
        dcat0 = w.dcat[0]
        ops.upsample2x_bilinear_bwd(dcat0.view(-1)[CONV_CH[3]:], w.dfeat, B=B, h=G, w=G, C=D, ld_dout=dcat0.shape[-1], ld_din=D,
                                    dout_bstride=s3 * s3 * dcat0.shape[-1], din_bstride=G * G * D)
        mode = "bicubic" if c.patch != 16 else "identity"
        tr = taps(mode, c.grid, G, dev, adjoint=True)
        ops.resample2d(w.dfeat, w.dtok[c.prefix:], tr, tr, B=B, h=G, w=G, H=c.grid, W=c.grid, C=D, ld_src=D, ld_dst=D,
                       src_bstride=G * G * D, dst_bstride=c.ntok * D)
        # ---- ConvStream (reverse): dD_k = skip slice of the concat gradient (+ dgrad of the next conv, accumulated)
        skip = [(w.dcat[2], s1), (w.dcat[1], s2), (w.dcat[0], s3)]
        srcs = [(w.img8_cur, S, 8, 8), (w.cat[2], s1, 48, w.cat[2].shape[-1]), (w.cat[1], s2, 96, w.cat[1].shape[-1])]
        for i in (2, 1, 0):
            dsk, r_out = skip[i]
            Mo = B * r_out * r_out
            cout = CONV_CH[i + 1]
            bp, bn = w.bnp[i], convs[i].bn
            ops.bn_relu_bwd(dsk, dsk.shape[-1], w.pre_c[i], bp.scale, bp.shift, bp.mean, bp.rstd, pk.bn[i].w, w.stats_b[i],
                            fl.gview[id(bn.weight)], fl.gview[id(bn.bias)], w.dpre_c[i], Mo, cout, NSLOTS)
            a, r_in, cin, ld = srcs[i]
            self._wgrad(w, i, a, r_in, cin, ld, r_out, 2, w.dpre_c[i], cout)
            if i > 0:
                tgt, _ = skip[i - 1]
                ops.gemm(w.dpre_c[i], pk.wd[i], tgt, M=B * r_in * r_in, N=cin, amode=A_CONV3_T,
                         conv=(r_out, r_out, cout, cout, r_in, r_in, 2), ldc=tgt.shape[-1], flags=ACCUM_BF16)
        # conv weight gradients: dWt [(ky,kx,c_pad), cout] -> parameter layout [cout, cin, ky, kx]
        ops.unpack_conv3x3_wgrad_multi([(w.dWt[i], fl.gview[id(cv.conv.weight)], pk.cin_pad[i], 3 if pk.perm[i] is not None else 0,
                                         i in w.wgrad_n_major_set)
                                        for i, cv in enumerate(convs)])
        if on_decoder_done is not None:
            on_decoder_done()
        # ---- encoder (LoRA gradients; frozen weights need dgrad only).  No adapters = a frozen encoder (checked above): nothing upstream
        # of the decoder has a gradient
        if c.lora:
            self._encoder_bwd(w, pk, fl, fz, on_block_done=on_lora_block_done)
        return fl.gflat

    def lora_blocks(self):
        c = self._config()
        return c.L if c.lora else 0

    def _encoder_bwd(self, w, pk, fl, fz, from_tokens=True, inject=None, on_block_done=None):
        """Backward of the ViT blocks.  from_tokens: start from the gradient of the final-norm tokens in w.dtok (MIPHEI-ViT);
        otherwise the caller has put the gradient of the last block's output into w.dx (f32).  inject(l) is called when w.dx
        holds the gradient of block l's output coming from later blocks and may add to it (forward_intermediates taps)."""
        c = self._config()
        B, D, M = w.B, c.D, w.M
        r_ = c.rank
        scale = c.Dh ** -0.5
        lsplit = max(1, min(512 // ((D + 127) // 128), (M + 255) // 256))
        if self.lora_group is None:
            G = c.L if on_block_done is None else 10
        else:
            G = max(1, int(self.lora_group))
        last = fz.blocks[c.L - 1]
        dp = getattr(w, "dpath", None)          # DropPath factors of the forward pass this backward belongs to
        rs = (lambda l, br: None) if dp is None else (lambda l, br: dp[l, br])
        if from_tokens:
            ops.layernorm_bwd(w.dtok, w.x_in[c.L], fz.nw, w.dx, last.ls2, w.dy, c.eps, accumulate=False,
                              rowscale_next=rs(c.L - 1, 1))
        else:
            ops.scale_cols_cast(w.dx, last.ls2, w.dy, rowscale=rs(c.L - 1, 1))
        for l in range(c.L - 1, -1, -1):
            b = fz.blocks[l]
            # MLP branch: dy = ls2 * dx
            ops.gemm(w.dy, b.t.wfc2, w.du, aux=w.u[l], epi=EPI_DSWIGLU if c.swiglu else EPI_DGELU)
            ops.gemm(w.du, b.t.wfc1, w.dh)
            ops.layernorm_bwd(w.dh, w.x_mid[l], b.n2w, w.dx, b.ls1, w.dy, c.eps, accumulate=True, rowscale_next=rs(l, 0))
            # attention branch: dy = ls1 * dx
            ops.gemm(w.dy, b.t.wproj, w.do)
            dqkv, dt = w.dqkv_all[l], w.dt_all[l]
            ops.attention_bwd(w.qkv[l], w.o[l], w.do, w.lse[l], w.dsum, dqkv, B, c.ntok, c.H, c.Dh, scale, out_res=w.ores[l] if w.ores_on else None)
            dq, dv = dqkv, dqkv.view(-1)[2 * D:]
            # dt_q = dq @ (a B_q)^T, dt_v = dv @ (a B_v)^T: one launch
            ops.skinny_xw2(dq, pk.Bq16[l], dt, dv, pk.Bv16[l], dt.view(-1)[r_:], ldx=3 * D, ldw=D, ldo=2 * r_, M=M, K=D, R=r_)
            if l % G == 0:
                # LoRA weight gradients of blocks l .. l+G-1 on the TN MFMA GEMM, both adapters per pass and the whole group
                # per launch: dB = t^T [dq | . | dv] (rows of B_q from the q columns, rows of B_v from the v columns),
                # dA = ([dt_q | dt_v]^T h)^T (columns of A_q, A_v).  One block's products are 4 steps of m per workgroup -
                # latency-bound at 13 us each; a group of 10 streams its operands at the fabric rate.  dA's slices: one round of
                # blocks (3 per CU x 256 CUs; 12 column tiles x 10 blocks x 6 slices = 720: 41.7 us against 49.9 with 10 slices); groups of
                # more than 16 blocks: 4 slices (40 blocks: 138.8 us; 152.1 / 141.7 / 153.9 with 2 / 6 / 8).
                n = min(G, c.L - l)
                gsplit = max(1, min(lsplit, -(-1024 // (n * 2 * ((D + 127) // 128)))))
                ops.gemm_tn(w.t[l], dqkv, fl.dBq[l], M=M, I=2 * r_, J=3 * D, lda=2 * r_, ldb=3 * D, ldci=D, ldcj=1,
                            msplit=gsplit, c2=fl.dBv[l], isplit=r_, j1=D, jlo2=2 * D, batch=n, stride_a=M * 2 * r_,
                            stride_b=M * 3 * D, stride_c=4 * r_ * D)
                ops.gemm_tn(dt, w.h1[l], fl.dAq[l], M=M, I=2 * r_, J=D, lda=2 * r_, ldb=D, ldci=1, ldcj=r_,
                            msplit=max(1, min(lsplit, 768 // (n * ((D + 127) // 128)) if n <= 16 else 4)), c2=fl.dAv[l], isplit=r_, batch=n,
                            stride_a=M * 2 * r_,
                            stride_b=M * D, stride_c=4 * r_ * D)
                for j in range(l + n - 1, l - 1, -1):
                    if c.alpha != 1.0:   # B2 carries alpha*B: d(B) = alpha * d(alpha*B); scaled before the slice may be exchanged
                        fl.dBq[j].mul_(c.alpha)
                        fl.dBv[j].mul_(c.alpha)
                    if on_block_done is not None:
                        on_block_done(j)
            if l > 0:
                ops.gemm(dqkv, b.t.wqkv, w.dh, a2=dt, b2=pk.Acat16[l], K2=2 * r_)
                if inject is not None:
                    inject(l - 1)
                ops.layernorm_bwd(w.dh, w.x_in[l], b.n1w, w.dx, fz.blocks[l - 1].ls2, w.dy, c.eps, accumulate=True,
                                  rowscale_next=rs(l - 1, 1))

    # ------------------------------------------------------------------ fused training step
    def loss_and_grad(self, out, target, marker_weights, lambda_factor, grad_scale=1.0):
        """WeightedMSELoss value (device scalar, f64) and dL/d(out) in the workspace."""
        w = self._saved.w
        w.scal.zero_()
        w.sqn_fresh = True
        # grad_scale multiplies dL/d(out) only (the loss value keeps lambda_factor): 1/world in data-parallel runs, so that the SUM
        # all-reduce of the gradient buckets already is the average -- every backward kernel is linear in dY, and for world = 2^k the
        # scaling commutes with every rounding (bf16 and f32 share the exponent range), i.e. the bits equal a post-exchange division
        ops.wmse_fwd_bwd(out, target.to(torch.float32).contiguous(), marker_weights, w.loss_acc, w.dY,
                         float(lambda_factor) * float(grad_scale))
        B, C, H, W = out.shape
        return w.loss_acc * (float(lambda_factor) / (C * B * H * W)), w.dY

    def adam_step(self, lr, betas=(0.5, 0.999), eps=1e-7, max_norm=1.0):
        fl, w = self._flat, self._saved.w
        if fl.m is None:
            fl.m, fl.v = torch.zeros_like(fl.flat), torch.zeros_like(fl.flat)
        fl.step += 1
        # the square-norm accumulator is zeroed by loss_and_grad (one fill for loss + norm); a caller that reaches adam_step
        # without it (external dY, custom loss, a second adam_step) gets its own fill, so the norm never accumulates across steps
        if not getattr(w, "sqn_fresh", False):
            w.sqn.zero_()
        w.sqn_fresh = False
        ops.sqnorm(fl.gflat, w.sqn)
        ops.adam_clip_step(fl.flat, fl.gflat, fl.m, fl.v, w.sqn, float(lr), betas[0], betas[1], eps,
                           1.0 - betas[0] ** fl.step, 1.0 - betas[1] ** fl.step, float(max_norm),
                           nonfinite=self.nonfinite_flag())
        self.params_changed()  # parameters changed in place through the flat buffer
        return w.sqn

    def nonfinite_flag(self):
        """Device int32 scalar the Adam kernel sets (and then honours: no further updates) when a gradient norm is NaN/Inf."""
        dev = self._require_gpu()
        if self._nonfinite is None or self._nonfinite.device != dev:
            self._nonfinite = torch.zeros(1, device=dev, dtype=torch.int32)
        return self._nonfinite

    # ------------------------------------------------------------------ optimiser state (checkpoint / resume)
    def optimizer_state_dict(self):
        """Adam state of the fused step: {"step", "exp_avg", "exp_avg_sq", "layout"}; the moments are flat f32 tensors in the
        order ``layout`` = [(parameter name, numel)] gives (reference: the optimizer state Lightning checkpoints carry)."""
        fl = self._flat
        if fl is None or fl.m is None:
            return self._opt_stash if (fl is None and self._opt_stash is not None) else {"step": 0, "exp_avg": None,
                                                                                         "exp_avg_sq": None, "layout": None}
        return {"step": int(fl.step), "exp_avg": fl.m.detach().clone(), "exp_avg_sq": fl.v.detach().clone(),
                "layout": [tuple(x) for x in fl.layout]}

    def load_optimizer_state_dict(self, sd):
        if sd is None or sd.get("exp_avg") is None:
            return
        if self._flat is None:
            self._opt_stash = sd          # applied when the flat buffer is built
        else:
            self._restore_optimizer(self._flat, sd)

    @staticmethod
    def _restore_optimizer(f, sd):
        if sd.get("exp_avg") is None:
            return
        if [tuple(x) for x in sd["layout"]] != [tuple(x) for x in f.layout]:
            raise RuntimeError("optimizer state does not match the trainable parameters of this generator "
                               f"({len(sd['layout'])} vs {len(f.layout)} tensors in the flat layout)")
        f.m = sd["exp_avg"].to(device=f.flat.device, dtype=torch.float32).clone()
        f.v = sd["exp_avg_sq"].to(device=f.flat.device, dtype=torch.float32).clone()
        f.step = int(sd["step"])

    def grad_buckets(self):
        """(decoder gradients, LoRA gradients): contiguous slices of the flat gradient buffer for all-reduce."""
        fl = self._ensure_flat()
        return fl.gflat[fl.n_lora:], fl.gflat[:fl.n_lora]

    # ------------------------------------------------------------------ autograd bridge (generic callers)
    def forward_autograd(self, x):
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.model.parameters())
        if not needs_grad:
            return self.forward(x, train=False).clone()
        fl = self._ensure_flat()
        return _GeneratorFn.apply(self, x, *[p for p in fl.params if p.requires_grad])


class _GeneratorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, engine, x, *params):
        ctx.engine = engine
        ctx.n = len(params)
        return engine.forward(x, train=True).clone()

    @staticmethod
    def backward(ctx, dY):
        eng = ctx.engine
        fl = eng._flat
        keep = fl.gflat.clone()           # gradients accumulated by earlier backward calls (p.grad are views)
        eng.backward(dY)
        new = fl.gflat.clone()
        fl.gflat.copy_(keep)
        grads, o = [], 0
        for p in fl.params:
            k = p.numel()
            if p.requires_grad:
                grads.append(new[o:o + k].view(p.shape))
            o += k
        return (None, None, *grads)


class _BareEncoder(torch.nn.Module):
    """Engine owner of a registry model used outside ViTMatte (embedding extraction: `FOUNDATION_MODEL_REGISTRY[name](...)`
    called directly, reference preprocessings/artifacts_detection/extract_embeddings.py:41-42): `.encoder.vit`, no decoder."""

    def __init__(self, vit):
        super().__init__()
        object.__setattr__(self, "encoder", NS(vit=vit))

    def parameters(self, recurse=True):
        return self.encoder.vit.parameters(recurse)


def encoder_tokens(vit, x):
    """Encoder-only forward for a bare VisionTransformer: final-norm tokens [B, N, D] in fp32."""
    eng = vit.__dict__.get("_engine_owner")
    if eng is None:
        eng = HipEngine(_BareEncoder(vit))
        object.__setattr__(vit, "_engine_owner", eng)
    dev = eng._require_gpu()
    x = x.detach().to(device=dev, dtype=torch.float32).contiguous()
    c = eng._config()
    if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] != c.S or x.shape[3] != c.S:
        raise ValueError(f"expected [B,3,{c.S},{c.S}] input, got {tuple(x.shape)}")
    with L.operands(eng.operand_mode()):       # fp16 parameters (`.half()`): the fp16-operand library
        pk = eng._pack_trainable(need_bwd=False)
        w = eng._workspace(x.shape[0], False)
        tok = eng._encoder_fwd(w, x, False, pk)
        return tok.view(x.shape[0], c.ntok, c.D).float()
