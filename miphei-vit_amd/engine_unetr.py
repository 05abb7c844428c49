"""Kernel sequencing of the UNETR baseline generator (`generators/unet.py`, reference src/generators/unet.py).

Every layer of the reference graph maps onto kernels the MIPHEI-ViT path already has:
  * ViT encoder with `forward_intermediates` taps: `HipEngine._encoder_fwd(..., taps=...)`; backward through
    `HipEngine._encoder_bwd` with the tap gradients injected into the residual-gradient stream
  * nearest 18->16 re-grid (nn.Upsample(scale_factor), unet.py:190-209): tap-table resample and its adjoint
  * Conv2DBlock: implicit-GEMM conv3x3 with bias and BatchNorm statistics in the epilogue -> bn_finalize -> bn_relu_apply;
    backward = fused BN+ReLU backward, TN weight-gradient GEMM on the virtual im2col, adjoint implicit GEMM
  * ConvTranspose2d(k2, s2): dense GEMM against the [4*Cout, Cin] repacked weight + `mvit_pixel_shuffle2x` into a channel slice of
    the concat buffer of the consuming stage (torch.cat never materialises); backward = inverse shuffle + dense / TN GEMMs
  * final conv1x1 and the fused per-marker heads.
  * Dropout(drop_rate) behind every block's ReLU and DropPath(drop_rate) in the ViT blocks (the reference trains this baseline with
    `model.dropout: 0.1`, configs/model/unet.yaml:2 -> generators/__init__.py:36): counter-based masks recomputed in the backward
    kernels, per-sample DropPath factors inside the residual GEMM epilogues (engine._drop_path_factors).
Activations are NHWC bf16, one buffer per graph node (kept for the backward pass).  Training goes through the autograd bridge
(`generator(x)`, `loss.backward()`, any torch optimiser); the fused single-sequence step of MIPHEI-ViT is not built for this baseline.
"""
from __future__ import annotations

from types import SimpleNamespace as NS

import torch

from . import ops
from .engine import BN_EPS, BN_MOM, HEAD_C, HEAD_HID, NSLOTS, HipEngine, _BareEncoder, _pad8
from .ops import A_CONV3, A_CONV3_T, EPI_STATS
from .resample import taps


class UnetrEngine:
    def __init__(self, model):
        self.model = model
        self._enc = None
        self._ws = {}
        self._saved = None
        self._drop_step = 0          # dropout masks are functions of (seed, step, layer, element): nothing is stored
        self._flat = None            # fused step: flat f32 buffer of the trainable parameters outside the ViT (+ gradient twin)
        self._opt_stash = None
        self._nonfinite = None

    def invalidate(self):
        if self._flat is not None and self._flat.m is not None:
            self._opt_stash = self.optimizer_state_dict()
        self._flat = None
        self._ws = {}
        self._saved = None
        if self._enc is not None:
            self._enc.invalidate()

    # ------------------------------------------------------------------ fused training step (flat buffers, one clip + Adam)
    def _ensure_flat(self):
        """Trainable parameters outside the ViT (pyramids, decoder, heads) in one flat f32 buffer with a gradient twin, as
        HipEngine does for MIPHEI-ViT; the LoRA adapters live in the encoder engine's own flat buffer."""
        if self._flat is not None:
            return self._flat
        enc = self._encoder_engine()
        dev = enc._require_gpu()
        vit_ids = {id(p) for p in self.model.encoder.model.parameters()}
        named = [(k, p) for k, p in self.model.named_parameters() if p.requires_grad and id(p) not in vit_ids]
        n = sum(p.numel() for _, p in named)
        flat, gflat = torch.empty(n, device=dev), torch.zeros(n, device=dev)
        gview, o = {}, 0
        for _, p in named:
            if p.dtype != torch.float32:
                raise RuntimeError("training needs fp32 master parameters")
            k = p.numel()
            flat[o:o + k].copy_(p.data.reshape(-1))
            p.data = flat[o:o + k].view(p.shape)
            gview[id(p)] = gflat[o:o + k].view(p.shape)
            p.grad = gview[id(p)]
            o += k
        self._flat = NS(flat=flat, gflat=gflat, n=n, gview=gview, m=None, v=None, step=0, layout=[(k, p.numel()) for k, p in named])
        if self._opt_stash is not None:
            self.load_optimizer_state_dict(self._opt_stash)
            self._opt_stash = None
        return self._flat

    def lora_blocks(self):
        return self._encoder_engine().lora_blocks()

    def grad_buckets(self):
        """(decoder-side gradients, LoRA gradients): the two contiguous regions the data-parallel exchange all-reduces"""
        return self._ensure_flat().gflat, self._encoder_engine().grad_buckets()[1]

    def param_buffers(self):
        return [self._ensure_flat().flat, self._encoder_engine()._ensure_flat().flat]

    def grad_buffers(self):
        return [self._ensure_flat().gflat, self._encoder_engine()._ensure_flat().gflat]

    @property
    def _pack_key(self):
        return self._encoder_engine()._pack_key

    @_pack_key.setter
    def _pack_key(self, v):
        self._encoder_engine()._pack_key = v

    def params_changed(self):
        self._encoder_engine().params_changed()

    def nonfinite_flag(self):
        dev = self._encoder_engine()._require_gpu()
        if self._nonfinite is None or self._nonfinite.device != dev:
            self._nonfinite = torch.zeros(1, device=dev, dtype=torch.int32)
        return self._nonfinite

    def loss_and_grad(self, out, target, marker_weights, lambda_factor, grad_scale=1.0):
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

    def backward_fused(self, dY, on_decoder_done=None, on_lora_block_done=None):
        """Gradients straight into the flat gradient buffers (decoder side here, LoRA in the encoder engine's)."""
        fl, efl = self._ensure_flat(), self._encoder_engine()._ensure_flat()
        fl.gflat.zero_()
        efl.gflat.zero_()
        grads = self.backward(dY, fused=True, on_decoder_done=on_decoder_done, on_lora_block_done=on_lora_block_done)
        return grads

    def adam_step(self, lr, betas=(0.5, 0.999), eps=1e-7, max_norm=1.0):
        fl, efl, w = self._flat, self._encoder_engine()._flat, self._saved.w
        for f in (fl, efl):
            if f.m is None:
                f.m, f.v = torch.zeros_like(f.flat), torch.zeros_like(f.flat)
            f.step += 1
        if not getattr(w, "sqn_fresh", False):   # zeroed with the loss accumulator by loss_and_grad; any other caller: own fill
            w.sqn.zero_()
        w.sqn_fresh = False
        ops.sqnorm(fl.gflat, w.sqn)           # one global norm over both buffers
        ops.sqnorm(efl.gflat, w.sqn)
        for f in (fl, efl):
            ops.adam_clip_step(f.flat, f.gflat, f.m, f.v, w.sqn, float(lr), betas[0], betas[1], eps, 1.0 - betas[0] ** f.step,
                               1.0 - betas[1] ** f.step, float(max_norm), nonfinite=self.nonfinite_flag())
        self._encoder_engine().params_changed()
        return w.sqn

    def optimizer_state_dict(self):
        fl, enc = self._flat, self._encoder_engine()
        if fl is None or fl.m is None:
            return self._opt_stash if self._opt_stash is not None else {"step": 0, "exp_avg": None, "exp_avg_sq": None, "layout": None}
        e = enc.optimizer_state_dict()
        return {"step": int(fl.step), "exp_avg": torch.cat([fl.m, e["exp_avg"]]), "exp_avg_sq": torch.cat([fl.v, e["exp_avg_sq"]]),
                "layout": [tuple(x) for x in fl.layout] + [tuple(x) for x in e["layout"]]}

    def load_optimizer_state_dict(self, sd):
        if sd is None or sd.get("exp_avg") is None:
            return
        if self._flat is None:
            self._opt_stash = sd
            return
        fl, enc = self._flat, self._encoder_engine()
        efl = enc._ensure_flat()
        want = [tuple(x) for x in fl.layout] + [tuple(x) for x in efl.layout]
        if [tuple(x) for x in sd["layout"]] != want:
            raise RuntimeError("optimizer state does not match the trainable parameters of this generator")
        dev = fl.flat.device
        fl.m, fl.v = sd["exp_avg"][:fl.n].to(dev).clone(), sd["exp_avg_sq"][:fl.n].to(dev).clone()
        efl.m, efl.v = sd["exp_avg"][fl.n:].to(dev).clone(), sd["exp_avg_sq"][fl.n:].to(dev).clone()
        fl.step = efl.step = int(sd["step"])

    def capture_inference(self, batch):
        """hipGraph capture of the eval-mode forward for a fixed batch (as HipEngine.capture_inference)."""
        enc = self._encoder_engine()
        dev = enc._require_gpu()
        c = enc._config()
        x_static = torch.zeros(batch, 3, c.S, c.S, device=dev, dtype=torch.float32)
        self._forward(x_static, train=False)          # warm-up: allocations, packs, LDS attributes
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out_static = self._forward(x_static, train=False)
        return graph.replay, x_static, out_static

    def _encoder_engine(self):
        if self._enc is None:
            vit = self.model.encoder.model
            self._enc = HipEngine(_BareEncoder(vit))
            object.__setattr__(vit, "_engine_owner", self._enc)
        return self._enc

    # ------------------------------------------------------------------ graph
    def _graph(self, G, S):
        """[(kind, name, src, H, modules, dst, dst_off)]; buffers {name: (resolution, channels)}"""
        m = self.model
        up, dec = m.encoder.feature_upsampler, m.decoder
        D, bott, s11, s12 = up.embed_dim, up.bottleneck_dim, up.skip_dim_11, up.skip_dim_12
        buf = {"img8": (S, 8), "a0": (S, 32), "cat0": (S, 128), "cat1": (8 * G, 256), "cat2": (4 * G, 512), "cat3": (2 * G, 2 * bott),
               "feat0": (G, D), "feat1": (G, D), "feat2": (G, D), "feat3": (G, D), "t01": (2 * G, s11), "t02": (4 * G, s12),
               "t11": (2 * G, s11), "c30": (2 * G, bott), "c31": (2 * G, bott), "c32": (2 * G, bott), "c20": (4 * G, 256),
               "c21": (4 * G, 256), "c10": (8 * G, 128), "c11": (8 * G, 128), "c00": (S, 64), "c01": (S, 64), "F3": (S, 32)}
        L = []

        def conv(name, src, blk, dst, off=0, cin=None):
            L.append(("conv", name, src, buf[src][0], (blk.block[0], blk.block[1]), dst, off, cin or buf[src][1]))

        def convT(name, src, ct, dst, off=0):
            L.append(("convT", name, src, buf[src][0], ct, dst, off, None))

        def deconv(name, src, blk, dst, off=0):
            mid = "m_" + name
            buf[mid] = (2 * buf[src][0], blk.block[0].out_channels)
            convT(name + ".t", src, blk.block[0], mid)
            L.append(("conv", name + ".c", mid, buf[mid][0], (blk.block[1], blk.block[2]), dst, off, buf[mid][1]))

        conv("s0", "img8", up.convsteam[0], "a0", cin=3)
        conv("s1", "a0", up.convsteam[1], "cat0")
        deconv("u0.1", "feat0", up.upsampler0[1], "t01")
        deconv("u0.2", "t01", up.upsampler0[2], "t02")
        deconv("u0.3", "t02", up.upsampler0[3], "cat1")
        deconv("u1.1", "feat1", up.upsampler1[1], "t11")
        deconv("u1.2", "t11", up.upsampler1[2], "cat2")
        deconv("u2.1", "feat2", up.upsampler2[1], "cat3")
        convT("bott", "feat3", dec.bottleneck_upsampler, "cat3", bott)
        conv("d3.0", "cat3", dec.decoder3_upsampler[0], "c30")
        conv("d3.1", "c30", dec.decoder3_upsampler[1], "c31")
        conv("d3.2", "c31", dec.decoder3_upsampler[2], "c32")
        convT("d3.3", "c32", dec.decoder3_upsampler[3], "cat2", 256)
        conv("d2.0", "cat2", dec.decoder2_upsampler[0], "c20")
        conv("d2.1", "c20", dec.decoder2_upsampler[1], "c21")
        convT("d2.2", "c21", dec.decoder2_upsampler[2], "cat1", 128)
        conv("d1.0", "cat1", dec.decoder1_upsampler[0], "c10")
        conv("d1.1", "c10", dec.decoder1_upsampler[1], "c11")
        convT("d1.2", "c11", dec.decoder1_upsampler[2], "cat0", 64)
        conv("d0.0", "cat0", dec.decoder0_header[0], "c00")
        conv("d0.1", "c00", dec.decoder0_header[1], "c01")
        return L, buf

    # ------------------------------------------------------------------ workspace
    def _workspace(self, B, S, dev, train):
        key = (B, S, train)
        if key in self._ws:
            return self._ws[key]
        m = self.model
        G = S // 16
        bf = torch.bfloat16
        e = lambda *s, dt=bf: torch.empty(*s, device=dev, dtype=dt)
        z = lambda *s, dt=bf: torch.zeros(*s, device=dev, dtype=dt)
        w = NS(B=B, S=S, G=G, train=train)
        w.layers, shapes = self._graph(G, S)
        w.buf = {k: e(B * r * r, ch) for k, (r, ch) in shapes.items() if k != "img8"}
        w.res = {k: r for k, (r, ch) in shapes.items()}
        maxel = max(B * r * r * ch for r, ch in shapes.values())
        w.pre = {}          # pre-BatchNorm conv outputs (kept per layer in train mode, one shared buffer otherwise)
        w.pre_shared = e(maxel)
        w.tmpT = e(4 * maxel if False else max(B * w.res[l[2]] ** 2 * 4 * l[4].out_channels for l in w.layers if l[0] == "convT"))
        w.bnp = {}
        w.bn_shared = NS(scale=e(512, dt=torch.float32), shift=e(512, dt=torch.float32), mean=e(512, dt=torch.float32),
                         rstd=e(512, dt=torch.float32))
        w.stats = z(NSLOTS * 2 * 512, dt=torch.float64)
        NH = m.num_heads
        nch = NH * HEAD_HID
        Mp = B * S * S
        w.G_ = e(Mp, 16)
        w.out = e(B, NH, S, S, dt=torch.float32)
        w.hbn = NS(scale=e(nch, dt=torch.float32), shift=e(nch, dt=torch.float32), mean=e(nch, dt=torch.float32),
                   rstd=e(nch, dt=torch.float32))
        w.mom = z(NSLOTS * (32 + 1024), dt=torch.float64)
        w.mom_sum = z(32 + 1024, dt=torch.float64)
        if train:
            for l in w.layers:
                if l[0] == "conv":
                    cout = l[4][0].out_channels
                    w.pre[l[1]] = e(B * l[3] * l[3], cout)
                    w.bnp[l[1]] = NS(scale=e(cout, dt=torch.float32), shift=e(cout, dt=torch.float32), mean=e(cout, dt=torch.float32),
                                     rstd=e(cout, dt=torch.float32))
            w.dbuf = {k: e(B * r * r, ch) for k, (r, ch) in shapes.items() if k != "img8"}
            w.dpre = e(maxel)
            w.dtmpT = e(w.tmpT.numel())
            w.stats_b = z(NSLOTS * 2 * 512, dt=torch.float64)
            wmax = max(9 * _pad8(l[7]) * l[4][0].out_channels for l in w.layers if l[0] == "conv")
            wmax = max(wmax, max(4 * l[4].out_channels * l[4].in_channels for l in w.layers if l[0] == "convT"))
            w.wscr = z(wmax, dt=torch.float32)
            w.bscr = z(4 * 512 * 8, dt=torch.float32)
            w.ones = torch.ones(Mp, 8, device=dev, dtype=bf)
            w.cscr = e(ops.heads_conv_bwd_scratch_bytes(Mp) // 4 + 1, dt=torch.float32)
            w.hscr = e(ops.heads_gate_bwd_scratch_bytes() // 4, dt=torch.float32)
            w.dG = e(Mp, 16, dt=torch.float32)
            w.dXc = e(Mp, HEAD_C, dt=torch.float32)
            w.dW3 = e(NH * 9, HEAD_C, dt=torch.float32)
            w.db3_slots = z(64, 32, dt=torch.float32)
            w.scal = z(2, dt=torch.float64)          # loss accumulator | gradient square norm (fused step)
            w.loss_acc, w.sqn = w.scal[0:1], w.scal[1:2]
            w.dY = e(B, NH, S, S, dt=torch.float32)
        self._ws[key] = w
        return w

    @staticmethod
    def _f32(t):
        return t.detach().to(torch.float32).contiguous()

    def _view(self, w, name, off=0):
        t = w.buf[name]
        return (t.view(-1)[off:] if off else t), t.shape[-1]

    # ------------------------------------------------------------------ layer forward
    def _conv_fwd(self, w, rec, bn_train, st):
        _, name, src, H, (conv, bn), dst, off, cin = rec
        B = w.B
        M = B * H * H
        a, ld_src = self._view(w, src)
        d, ld_dst = self._view(w, dst, off)
        cout = conv.out_channels
        cp = _pad8(cin)
        assert cp == ld_src, (name, cp, ld_src)
        wk = torch.empty(cout, 9 * cp, device=a.device, dtype=torch.bfloat16)
        wd = torch.empty(cp, 9 * cout, device=a.device, dtype=torch.bfloat16) if w.train else None
        ops.pack_conv3x3_weights(self._f32(conv.weight), wk, wd)
        pre = w.pre[name] if w.train else w.pre_shared[:M * cout].view(M, cout)
        bnp = w.bnp[name] if w.train else w.bn_shared
        bias = self._f32(conv.bias) if conv.bias is not None else None
        stt = w.stats[:NSLOTS * 2 * cout]
        if bn_train:
            stt.zero_()
            ops.gemm(a, wk, pre, M=M, amode=A_CONV3, conv=(H, H, cp, ld_src, H, H, 1), bias=bias, epi=EPI_STATS, stats=stt,
                     nslots=NSLOTS)
        else:
            ops.gemm(a, wk, pre, M=M, amode=A_CONV3, conv=(H, H, cp, ld_src, H, H, 1), bias=bias)
        rm, rv = bn.running_mean, bn.running_var
        if rm.dtype != torch.float32:
            if bn_train:
                raise RuntimeError("train-mode BatchNorm needs fp32 running statistics (model.float())")
            rm, rv = rm.float(), rv.float()
        gamma = self._f32(bn.weight)
        ops.bn_finalize(stt, gamma, self._f32(bn.bias), rm, rv, bnp.scale, bnp.shift, bnp.mean, bnp.rstd, cout, NSLOTS, M, BN_EPS,
                        BN_MOM, bn_train)
        if bn_train:
            bn.num_batches_tracked += 1
        # nn.Dropout(drop_rate) behind the ReLU of every Conv2DBlock / Deconv2DBlock (reference unet.py:441-519), train mode only
        drop_p = float(self.model.decoder.drop_rate or 0.0) if bn_train else 0.0
        drop_seed = (st["_seed"] + 0x632BE59BD9B4E019 * (1 + len(st))) & 0xFFFFFFFFFFFFFFFF if drop_p > 0 else 0
        ops.bn_relu_apply(pre, bnp.scale, bnp.shift, d, M, cout, cout, ld_dst, drop_p=drop_p, drop_seed=drop_seed)
        st[name] = NS(wd=wd, gamma=gamma, drop_p=drop_p, drop_seed=drop_seed)

    def _convT_fwd(self, w, rec, st):
        _, name, src, H, ct, dst, off, _ = rec
        B = w.B
        M = B * H * H
        a, ld_src = self._view(w, src)
        d, ld_dst = self._view(w, dst, off)
        cin, cout = ct.in_channels, ct.out_channels
        wt = self._f32(ct.weight).permute(2, 3, 1, 0).reshape(4 * cout, cin).to(torch.bfloat16).contiguous()
        b4 = self._f32(ct.bias).repeat(4).contiguous()
        tmp = w.tmpT[:M * 4 * cout].view(M, 4 * cout)
        ops.gemm(a, wt, tmp, M=M, K=cin, lda=ld_src, bias=b4)
        ops.pixel_shuffle2x(tmp, d, B, H, H, cout, ld_dst)
        st[name] = NS(wt=wt)

    # ------------------------------------------------------------------ forward
    def forward(self, x, train=None):
        m = self.model
        if train:        # fused step (ModelModule.training_step): no autograd graph, gradients go to the flat buffers
            self._ensure_flat()
            return self._forward(x, train=True)
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in m.parameters())
        if not needs_grad or not m.training:
            # eval mode never records a graph (backward through eval-mode BatchNorm is not part of the training path)
            return self._forward(x, train=False).clone()
        params = [p for p in m.parameters() if p.requires_grad]
        return _UnetrFn.apply(self, x, *params)

    def _forward(self, x, train):
        m = self.model
        enc = self._encoder_engine()
        dev = enc._require_gpu()
        in_dtype = x.dtype
        x = x.detach().to(device=dev, dtype=torch.float32).contiguous()
        c = enc._config()
        if train and not c.lora:
            raise NotImplementedError("training the UNETR baseline with a fully unfrozen encoder is outside this path (use *_lora)")
        B, S = x.shape[0], x.shape[-1]
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] != c.S or S != c.S:
            raise ValueError(f"expected [B,3,{c.S},{c.S}] input, got {tuple(x.shape)}")
        bn_train = m.training
        w = self._workspace(B, S, dev, train)
        D, G = c.D, w.G
        # ---- ViT with the four intermediate taps (block outputs, no final norm)
        pk = enc._pack_trainable(need_bwd=train)
        if train:
            enc._ensure_frozen_bwd()
            enc._ensure_flat()
            pk = enc._pack_trainable(need_bwd=True)
        we = enc._workspace(B, train)
        if not hasattr(we, "tap16"):
            we.tap16 = [torch.empty(we.M, D, device=dev, dtype=torch.bfloat16) for _ in range(4)]
        layers = m.encoder.extract_layers
        enc._encoder_fwd(we, x, train, pk, taps={l: we.tap16[i] for i, l in enumerate(layers)})
        ty = taps("nearest" if c.patch != 16 else "identity", c.grid, G, dev)
        for i in range(4):
            ops.resample2d(we.tap16[i][c.prefix:], w.buf[f"feat{i}"], ty, ty, B=B, h=c.grid, w=c.grid, H=G, W=G, C=D, ld_src=D,
                           ld_dst=D, src_bstride=c.ntok * D, dst_bstride=G * G * D)
        w.buf["img8"] = we.img8_cur          # bf16 NHWC image: written once by the encoder engine (its patch-embed gather reads it too)
        self._drop_step += 1
        st = {"_seed": (torch.initial_seed() * 0x9E3779B97F4A7C15 + self._drop_step * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF}
        for rec in w.layers:
            if rec[0] == "conv":
                self._conv_fwd(w, rec, bn_train, st)
            else:
                self._convT_fwd(w, rec, st)
        c1 = m.decoder.decoder0_header[2]
        st["c1x1"] = NS(w=self._f32(c1.weight).view(32, 64).to(torch.bfloat16).contiguous())
        ops.gemm(w.buf["c01"], st["c1x1"].w, w.buf["F3"], M=B * S * S, K=64, lda=64, bias=self._f32(c1.bias))
        out = self._heads_fwd(w, bn_train, st)
        self._saved = NS(w=w, we=we, pk=pk, st=st, bn_train=bn_train, c=c) if train else None
        return out.to(in_dtype) if in_dtype.is_floating_point else out

    def _heads(self):
        return [getattr(self.model, f"segmentation_head_{i}") for i in range(self.model.num_heads)]

    def _heads_fwd(self, w, bn_train, st):
        m = self.model
        NH, B, S = m.num_heads, w.B, w.S
        Mp = B * S * S
        heads = self._heads()
        F3 = w.buf["F3"]
        dev = F3.device
        stk = lambda get, shape: torch.stack([self._f32(get(h)).reshape(-1) for h in heads]).reshape(shape).contiguous()
        hp = NS(W1=stk(lambda h: h[0].psi[0].weight, (NH * HEAD_HID, HEAD_C)), b1=stk(lambda h: h[0].psi[0].bias, (NH * HEAD_HID,)),
                bnw=stk(lambda h: h[0].psi[1].weight, (NH * HEAD_HID,)), bnb=stk(lambda h: h[0].psi[1].bias, (NH * HEAD_HID,)),
                W2=stk(lambda h: h[0].psi[3].weight, (NH * HEAD_HID,)), b2=stk(lambda h: h[0].psi[3].bias, (NH,)),
                W3k=stk(lambda h: h[1].weight, (NH, HEAD_C, 9)).transpose(1, 2).contiguous(), b3=stk(lambda h: h[1].bias, (NH,)))
        rm = torch.cat([h[0].psi[1].running_mean.detach().float() for h in heads]).to(dev).contiguous()
        rv = torch.cat([h[0].psi[1].running_var.detach().float() for h in heads]).to(dev).contiguous()
        if bn_train:
            w.mom.zero_()
            ops.heads_moments(F3, w.mom, Mp, NSLOTS)
        ops.heads_bn_from_moments(w.mom, hp.W1, hp.b1, hp.bnw, hp.bnb, rm, rv, w.hbn.scale, w.hbn.shift, w.hbn.mean, w.hbn.rstd,
                                  w.mom_sum, NH, NSLOTS, Mp, BN_EPS, BN_MOM, bn_train)
        if bn_train:
            with torch.no_grad():
                for i, h in enumerate(heads):
                    h[0].psi[1].running_mean.copy_(rm[HEAD_HID * i:HEAD_HID * (i + 1)])
                    h[0].psi[1].running_var.copy_(rv[HEAD_HID * i:HEAD_HID * (i + 1)])
                    h[0].psi[1].num_batches_tracked += 1
        ops.heads_gate_fwd(F3, hp.W1, hp.b1, w.hbn.scale, w.hbn.shift, hp.W2, hp.b2, w.G_, Mp, NH)
        ops.heads_conv_fwd(F3, w.G_, hp.W3k, hp.b3, w.out, B, S, S, NH)
        st["heads"] = hp
        return w.out

    # ------------------------------------------------------------------ backward
    def backward(self, dY, fused=False, on_decoder_done=None, on_lora_block_done=None):
        """-> {id(param): gradient tensor} for every trainable parameter of the generator.  fused: the decoder-side gradients
        are also written into the flat gradient buffer and the LoRA gradients stay in the encoder engine's (no copies handed
        out); the hooks are those of HipEngine.backward (data-parallel exchange)."""
        sv = self._saved
        if sv is None:
            raise RuntimeError("backward() needs a preceding training-mode forward")
        if not sv.bn_train:
            raise NotImplementedError("backward through eval-mode BatchNorm is not part of the training path")
        m, w, st, c = self.model, sv.w, sv.st, sv.c
        B, S, G = w.B, w.S, w.G
        Mp = B * S * S
        dev = dY.device
        grads = {}
        dY = dY.to(torch.float32).contiguous()
        # ---- heads
        NH = m.num_heads
        hp = st["heads"]
        heads = self._heads()
        nch = NH * HEAD_HID
        z = lambda *s: torch.zeros(*s, device=dev)
        dW1, dbnw, dbnb, dW2, db2 = z(nch, HEAD_C), z(nch), z(nch), z(nch), z(NH)
        w.db3_slots.zero_()
        F3 = w.buf["F3"]
        ops.heads_conv_bwd(dY, w.out, F3, w.G_, hp.W3k, w.cscr, w.dG, w.dXc, w.dW3, w.db3_slots, B, S, S, NH)
        ops.heads_gate_bwd(F3, w.G_, w.dG, w.dXc, hp.W1, hp.b1, w.hbn.scale, w.hbn.shift, w.hbn.mean, w.hbn.rstd, hp.bnw, hp.W2,
                           w.mom_sum, w.hscr, dW1, dbnw, dbnb, dW2, db2, w.dbuf["F3"], Mp, NH)
        db3 = w.db3_slots.sum(0)[:NH]
        dW3 = w.dW3.view(NH, 9, HEAD_C).transpose(1, 2)
        for i, h in enumerate(heads):
            sl = slice(HEAD_HID * i, HEAD_HID * (i + 1))
            grads[id(h[0].psi[0].weight)] = dW1[sl].reshape(h[0].psi[0].weight.shape)
            grads[id(h[0].psi[0].bias)] = torch.zeros_like(h[0].psi[0].bias)        # bias in front of a train-mode BatchNorm
            grads[id(h[0].psi[1].weight)] = dbnw[sl]
            grads[id(h[0].psi[1].bias)] = dbnb[sl]
            grads[id(h[0].psi[3].weight)] = dW2[sl].reshape(h[0].psi[3].weight.shape)
            grads[id(h[0].psi[3].bias)] = db2[i:i + 1]
            grads[id(h[1].weight)] = dW3[i].reshape(h[1].weight.shape).contiguous()
            grads[id(h[1].bias)] = db3[i:i + 1]
        # ---- conv1x1
        c1 = m.decoder.decoder0_header[2]
        dF3 = w.dbuf["F3"]
        ops.gemm(dF3, st["c1x1"].w.t().contiguous(), w.dbuf["c01"], M=Mp, K=32, lda=32)
        gw = w.wscr[:32 * 64].view(32, 64)
        gw.zero_()
        ops.gemm_tn(dF3, w.buf["c01"], gw, M=Mp, I=32, J=64, lda=32, ldb=64, ldci=64, msplit=max(1, min(64, Mp // 4096)))
        grads[id(c1.weight)] = gw.clone().view(c1.weight.shape)
        grads[id(c1.bias)] = self._colsum(w, dF3, Mp, 32)
        # ---- decoder / pyramids in reverse
        for rec in reversed(w.layers):
            if rec[0] == "conv":
                self._conv_bwd(w, rec, st, grads)
            else:
                self._convT_bwd(w, rec, st, grads)
        # ---- encoder: adjoint re-grid of the four feature gradients, injected into the residual-gradient stream
        enc, we, pk = self._enc, sv.we, sv.pk
        fz, fl = enc._ensure_frozen_bwd(), enc._ensure_flat()
        D = c.D
        if not hasattr(we, "dtap"):
            we.dtap = [torch.zeros(we.M, D, device=dev, dtype=torch.bfloat16) for _ in range(4)]
        ta = taps("nearest" if c.patch != 16 else "identity", c.grid, G, dev, adjoint=True)
        layers = m.encoder.extract_layers
        for i in range(4):
            ops.resample2d(w.dbuf[f"feat{i}"], we.dtap[i][c.prefix:], ta, ta, B=B, h=G, w=G, H=c.grid, W=c.grid, C=D, ld_src=D,
                           ld_dst=D, src_bstride=G * G * D, dst_bstride=c.ntok * D)
        where = {l: i for i, l in enumerate(layers)}
        assert layers[-1] == c.L - 1
        if fused:
            gv = self._flat.gview
            for pid, g in grads.items():          # decoder-side gradients into their slices of the flat gradient buffer
                gv[pid].copy_(g.reshape(gv[pid].shape))
            if on_decoder_done is not None:
                on_decoder_done()
        else:
            keep = fl.gflat.clone() # LoRA .grad tensors are views of the flat gradient buffer: hand out copies, restore
            fl.gflat.zero_()
        we.dx.copy_(we.dtap[3])

        def inject(l):
            if l in where:
                we.dx.add_(we.dtap[where[l]])

        enc._encoder_bwd(we, pk, fl, fz, from_tokens=False, inject=inject, on_block_done=on_lora_block_done if fused else None)
        if fused:
            return grads
        new = fl.gflat.clone()
        fl.gflat.copy_(keep)
        o = 0
        for p in fl.params:
            k = p.numel()
            grads[id(p)] = new[o:o + k].view(p.shape)
            o += k
        return grads

    def _colsum(self, w, a, M, C):
        """sum over the rows of a bf16 [M, C] matrix (C % 8 == 0) on the TN GEMM against a ones column"""
        out = w.bscr[:C * 8].view(C, 8)
        out.zero_()
        ops.gemm_tn(a, w.ones, out, M=M, I=C, J=8, lda=a.shape[-1] if a.dim() == 2 else C, ldb=8, ldci=8,
                    msplit=max(1, min(64, M // 4096)))
        return out[:, 0].clone()

    def _conv_bwd(self, w, rec, st, grads):
        _, name, src, H, (conv, bn), dst, off, cin = rec
        B = w.B
        M = B * H * H
        cout = conv.out_channels
        cp = _pad8(cin)
        a, ld_src = self._view(w, src)
        dd = w.dbuf[dst]
        dy = dd.view(-1)[off:] if off else dd
        bnp, pre = w.bnp[name], w.pre[name]
        dpre = w.dpre[:M * cout].view(M, cout)
        sb = w.stats_b[:NSLOTS * 2 * cout]
        sb.zero_()
        dgam, dbet = torch.zeros(cout, device=a.device), torch.zeros(cout, device=a.device)
        ops.bn_relu_bwd(dy, dd.shape[-1], pre, bnp.scale, bnp.shift, bnp.mean, bnp.rstd, st[name].gamma, sb, dgam, dbet, dpre, M, cout,
                        NSLOTS, drop_p=st[name].drop_p, drop_seed=st[name].drop_seed)
        grads[id(bn.weight)], grads[id(bn.bias)] = dgam, dbet
        if conv.bias is not None:
            grads[id(conv.bias)] = torch.zeros_like(conv.bias)      # bias in front of a train-mode BatchNorm: zero gradient
        K9 = 9 * cp
        dWt = w.wscr[:K9 * cout].view(K9, cout)
        dWt.zero_()
        it, jt = (128, 32) if cout <= 32 else ((128, 64) if cout <= 64 else (128, 128))
        tiles = ((K9 + it - 1) // it) * ((cout + jt - 1) // jt)
        ms = max(1, min(1024 // tiles, (M + 255) // 256))
        ops.gemm_tn(a, dpre, dWt, M=M, I=K9, J=cout, ldb=cout, ldci=cout, msplit=ms, conv=(H, H, cp, ld_src, H, H, 1))
        grads[id(conv.weight)] = dWt.view(3, 3, cp, cout)[:, :, :cin].permute(3, 2, 0, 1).contiguous()
        if src != "img8":
            ds = w.dbuf[src]
            ops.gemm(dpre, st[name].wd, ds, M=M, N=cp, amode=A_CONV3_T, conv=(H, H, cout, cout, H, H, 1), ldc=ds.shape[-1])

    def _convT_bwd(self, w, rec, st, grads):
        _, name, src, H, ct, dst, off, _ = rec
        B = w.B
        M = B * H * H
        cin, cout = ct.in_channels, ct.out_channels
        a, ld_src = self._view(w, src)
        dd = w.dbuf[dst]
        dy = dd.view(-1)[off:] if off else dd
        dtmp = w.dtmpT[:M * 4 * cout].view(M, 4 * cout)
        ops.pixel_shuffle2x(dtmp, dy, B, H, H, cout, dd.shape[-1], inverse=True)
        gW = w.wscr[:4 * cout * cin].view(4 * cout, cin)
        gW.zero_()
        ops.gemm_tn(dtmp, a, gW, M=M, I=4 * cout, J=cin, lda=4 * cout, ldb=ld_src, ldci=cin, msplit=max(1, min(32, M // 1024)))
        grads[id(ct.weight)] = gW.view(2, 2, cout, cin).permute(3, 2, 0, 1).contiguous()
        grads[id(ct.bias)] = self._colsum(w, dtmp, M, 4 * cout).view(4, cout).sum(0)
        ops.gemm(dtmp, st[name].wt.t().contiguous(), w.dbuf[src], M=M, K=4 * cout, lda=4 * cout)


class _UnetrFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, engine, x, *params):
        ctx.engine = engine
        ctx.pids = [id(p) for p in params]
        return engine._forward(x, train=True).clone()

    @staticmethod
    def backward(ctx, dY):
        g = ctx.engine.backward(dY)
        return (None, None, *[g.get(pid) for pid in ctx.pids])
