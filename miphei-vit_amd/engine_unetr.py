"""Kernel sequencing of the UNETR baseline generator (`generators/unet.py`, reference src/generators/unet.py).

Forward only in this round (train-mode or eval-mode BatchNorm; inference with the LoRA adapters merged): every layer of the
reference graph maps onto kernels the MIPHEI-ViT path already has --
  * ViT encoder with `forward_intermediates` taps: `HipEngine._encoder_fwd(..., taps=...)`
  * nearest 18->16 re-grid (nn.Upsample(scale_factor), unet.py:190-209): tap-table resample
  * Conv2DBlock: implicit-GEMM conv3x3 with bias and BatchNorm statistics in the epilogue -> bn_finalize -> bn_relu_apply
  * ConvTranspose2d(k2, s2): dense GEMM against the [4*Cout, Cin] repacked weight + `mvit_pixel_shuffle2x` into a channel slice of
    the concat buffer of the consuming stage (torch.cat never materialises)
  * final conv1x1 and the fused per-marker heads.
Activations are NHWC bf16.  The backward pass of this baseline is not built yet (training_step raises).
"""
from __future__ import annotations

from types import SimpleNamespace as NS

import torch

from . import ops
from .engine import BN_EPS, BN_MOM, HEAD_C, HEAD_HID, NSLOTS, HipEngine, _BareEncoder, _pad8
from .ops import A_CONV3, EPI_STATS
from .resample import taps


class UnetrEngine:
    def __init__(self, model):
        self.model = model
        self._enc = None
        self._ws = {}

    def invalidate(self):
        self._ws = {}
        if self._enc is not None:
            self._enc.invalidate()

    def _encoder_engine(self):
        if self._enc is None:
            vit = self.model.encoder.model
            self._enc = HipEngine(_BareEncoder(vit))
            object.__setattr__(vit, "_engine_owner", self._enc)
        return self._enc

    # ------------------------------------------------------------------ workspace
    def _workspace(self, B, S, dev):
        key = (B, S)
        if key in self._ws:
            return self._ws[key]
        m = self.model
        up = m.encoder.feature_upsampler
        D, bott, s11, s12 = up.embed_dim, up.bottleneck_dim, up.skip_dim_11, up.skip_dim_12
        G = S // 16
        bf = torch.bfloat16
        e = lambda *s, dt=bf: torch.empty(*s, device=dev, dtype=dt)
        w = NS(B=B, S=S, G=G)
        w.img8 = e(B, S, S, 8)
        w.feat = [e(B, G, G, D) for _ in range(4)]
        # concat buffers of the decoder stages: [skip | up-convolved]
        w.cat3 = e(B, 2 * G, 2 * G, 2 * bott)
        w.cat2 = e(B, 4 * G, 4 * G, 512)
        w.cat1 = e(B, 8 * G, 8 * G, 256)
        w.cat0 = e(B, S, S, 128)
        maxpix = B * S * S
        w.pre = e(maxpix * 64)            # pre-BatchNorm conv output of the current layer (largest: 64 ch at full res)
        w.act = [e(maxpix * 64), e(maxpix * 64)]   # ping-pong activations inside a chain
        w.tmpT = e(maxpix * 64)           # ConvTranspose GEMM output before the pixel shuffle (largest: 4*64 ch at S/2)
        w.bn = NS(scale=e(512, dt=torch.float32), shift=e(512, dt=torch.float32), mean=e(512, dt=torch.float32),
                  rstd=e(512, dt=torch.float32))
        w.stats = torch.zeros(NSLOTS * 2 * 512, device=dev, dtype=torch.float64)
        NH = m.num_heads
        nch = NH * HEAD_HID
        w.F3 = e(maxpix, HEAD_C)
        w.G_ = e(maxpix, 16)
        w.out = e(B, NH, S, S, dt=torch.float32)
        w.hbn = NS(scale=e(nch, dt=torch.float32), shift=e(nch, dt=torch.float32), mean=e(nch, dt=torch.float32),
                   rstd=e(nch, dt=torch.float32))
        w.mom = torch.zeros(NSLOTS * (32 + 1024), device=dev, dtype=torch.float64)
        w.mom_sum = torch.zeros(32 + 1024, device=dev, dtype=torch.float64)
        self._ws[key] = w
        return w

    # ------------------------------------------------------------------ layer helpers
    @staticmethod
    def _f32(t):
        return t.detach().to(torch.float32).contiguous()

    def _conv3(self, w, src, H, cin, ld_src, conv, bn, dst, ld_dst, bn_train):
        """Conv2d 3x3 (bias) -> BatchNorm -> ReLU on an NHWC [B,H,H,*] source; result into dst (row stride ld_dst)"""
        B = w.B
        M = B * H * H
        cout = conv.out_channels
        cp = _pad8(cin)
        wk = torch.empty(cout, 9 * cp, device=src.device, dtype=torch.bfloat16)
        ops.pack_conv3x3_weights(self._f32(conv.weight), wk, None)
        pre = w.pre[:M * cout].view(M, cout)
        bias = self._f32(conv.bias) if conv.bias is not None else None
        if bn_train:
            st = w.stats[:NSLOTS * 2 * cout]
            st.zero_()
            ops.gemm(src, wk, pre, M=M, amode=A_CONV3, conv=(H, H, cp, ld_src, H, H, 1), bias=bias, epi=EPI_STATS, stats=st,
                     nslots=NSLOTS)
        else:
            st = w.stats[:NSLOTS * 2 * cout]
            ops.gemm(src, wk, pre, M=M, amode=A_CONV3, conv=(H, H, cp, ld_src, H, H, 1), bias=bias)
        rm, rv = bn.running_mean, bn.running_var
        if rm.dtype != torch.float32:
            if bn_train:
                raise RuntimeError("train-mode BatchNorm needs fp32 running statistics (model.float())")
            rm, rv = rm.float(), rv.float()
        ops.bn_finalize(st, self._f32(bn.weight), self._f32(bn.bias), rm, rv, w.bn.scale, w.bn.shift, w.bn.mean, w.bn.rstd, cout,
                        NSLOTS, M, BN_EPS, BN_MOM, bn_train)
        if bn_train:
            bn.num_batches_tracked += 1
        ops.bn_relu_apply(pre, w.bn.scale, w.bn.shift, dst, M, cout, cout, ld_dst)

    def _convT(self, w, src, H, ct, dst, ld_dst):
        """ConvTranspose2d(k2, s2) (bias) of an NHWC [B,H,H,Cin] source into dst = NHWC [B,2H,2H,*] slice (row stride ld_dst)"""
        B = w.B
        M = B * H * H
        cin, cout = ct.in_channels, ct.out_channels
        wt = self._f32(ct.weight).permute(2, 3, 1, 0).reshape(4 * cout, cin).to(torch.bfloat16).contiguous()
        b4 = self._f32(ct.bias).repeat(4).contiguous()
        tmp = w.tmpT[:M * 4 * cout].view(M, 4 * cout)
        ops.gemm(src, wt, tmp, M=M, K=cin, lda=cin, bias=b4)
        ops.pixel_shuffle2x(tmp, dst, B, H, H, cout, ld_dst)

    def _deconv_block(self, w, src, H, blk, dst, ld_dst, bn_train):
        """Deconv2DBlock: ConvTranspose -> conv3x3 -> BN -> ReLU; src NHWC [B,H,H,Cin] -> dst NHWC [B,2H,2H,*]"""
        ct, conv, bn = blk.block[0], blk.block[1], blk.block[2]
        cout = ct.out_channels
        B = w.B
        mid = w.act[1][:B * 4 * H * H * cout].view(B * 4 * H * H, cout)
        self._convT(w, src, H, ct, mid, cout)
        self._conv3(w, mid, 2 * H, cout, cout, conv, bn, dst, ld_dst, bn_train)

    # ------------------------------------------------------------------ forward
    def forward(self, x):
        m = self.model
        if torch.is_grad_enabled() and any(p.requires_grad for p in m.parameters()) and m.training:
            # the reference trains this baseline too; its backward is not built on this path yet
            pass
        enc = self._encoder_engine()
        dev = enc._require_gpu()
        in_dtype = x.dtype
        x = x.detach().to(device=dev, dtype=torch.float32).contiguous()
        c = enc._config()
        B, S = x.shape[0], x.shape[-1]
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] != c.S or S != c.S:
            raise ValueError(f"expected [B,3,{c.S},{c.S}] input, got {tuple(x.shape)}")
        bn_train = m.training
        w = self._workspace(B, S, dev)
        up, dec = m.encoder.feature_upsampler, m.decoder
        D, G = c.D, w.G
        # ---- ViT with the four intermediate taps (block outputs, no final norm)
        pk = enc._pack_trainable(need_bwd=False)
        we = enc._workspace(B, False)
        if not hasattr(we, "tap16"):
            we.tap16 = [torch.empty(we.M, D, device=dev, dtype=torch.bfloat16) for _ in range(4)]
        layers = m.encoder.extract_layers
        enc._encoder_fwd(we, x, False, pk, taps={l: we.tap16[i] for i, l in enumerate(layers)})
        ty = taps("nearest" if c.patch != 16 else "identity", c.grid, G, dev)
        for i in range(4):
            ops.resample2d(we.tap16[i][c.prefix:], w.feat[i], ty, ty, B=B, h=c.grid, w=c.grid, H=G, W=G, C=D, ld_src=D,
                           ld_dst=D, src_bstride=c.ntok * D, dst_bstride=G * G * D)
        # ---- conv stem on the image -> skip z0 = cat0[..., :64]
        ops.image_to_nhwc(x, w.img8, 8, nzero=5)
        a0 = w.act[0][:B * S * S * 32].view(B * S * S, 32)
        self._conv3(w, w.img8, S, 3, 8, up.convsteam[0].block[0], up.convsteam[0].block[1], a0, 32, bn_train)
        self._conv3(w, a0, S, 32, 32, up.convsteam[1].block[0], up.convsteam[1].block[1], w.cat0, 128, bn_train)
        # ---- feature pyramids (Deconv2DBlock chains) -> skips z1, z2, z3 in the concat buffers
        s11, s12, bott = up.skip_dim_11, up.skip_dim_12, up.bottleneck_dim
        t0 = w.act[0][:B * 4 * G * G * s11].view(-1, s11)
        self._deconv_block(w, w.feat[0].view(-1, D), G, up.upsampler0[1], t0, s11, bn_train)
        t1 = w.act[0][B * 4 * G * G * s11:B * 4 * G * G * s11 + B * 16 * G * G * s12].view(-1, s12)
        self._deconv_block(w, t0, 2 * G, up.upsampler0[2], t1, s12, bn_train)
        self._deconv_block(w, t1, 4 * G, up.upsampler0[3], w.cat1, 256, bn_train)                  # z1: 128 ch at 8G
        self._deconv_block(w, w.feat[1].view(-1, D), G, up.upsampler1[1], t0, s11, bn_train)
        self._deconv_block(w, t0, 2 * G, up.upsampler1[2], w.cat2, 512, bn_train)                  # z2: 256 ch at 4G
        self._deconv_block(w, w.feat[2].view(-1, D), G, up.upsampler2[1], w.cat3, 2 * bott, bn_train)  # z3: bott ch at 2G
        # ---- decoder
        self._convT(w, w.feat[3].view(-1, D), G, dec.bottleneck_upsampler, w.cat3.view(-1)[bott:], 2 * bott)
        h = self._chain(w, w.cat3.view(-1, 2 * bott), 2 * G, dec.decoder3_upsampler, 3, bn_train)
        self._convT(w, h, 2 * G, dec.decoder3_upsampler[3], w.cat2.view(-1)[256:], 512)
        h = self._chain(w, w.cat2.view(-1, 512), 4 * G, dec.decoder2_upsampler, 2, bn_train)
        self._convT(w, h, 4 * G, dec.decoder2_upsampler[2], w.cat1.view(-1)[128:], 256)
        h = self._chain(w, w.cat1.view(-1, 256), 8 * G, dec.decoder1_upsampler, 2, bn_train)
        self._convT(w, h, 8 * G, dec.decoder1_upsampler[2], w.cat0.view(-1)[64:], 128)
        h = self._chain(w, w.cat0.view(-1, 128), S, dec.decoder0_header, 2, bn_train)
        c1 = dec.decoder0_header[2]
        ops.gemm(h, self._f32(c1.weight).view(32, 64).to(torch.bfloat16).contiguous(), w.F3, M=B * S * S, K=64, lda=64,
                 bias=self._f32(c1.bias))
        out = self._heads_fwd(w, bn_train)
        return out.to(in_dtype) if in_dtype.is_floating_point else out

    def _chain(self, w, src, H, seq, n, bn_train):
        """n Conv2DBlocks of a decoder stage; returns the last activation [B*H*H, C]"""
        cur, ld = src, src.shape[-1]
        cin = ld
        for k in range(n):
            conv, bn = seq[k].block[0], seq[k].block[1]
            cout = conv.out_channels
            dst = w.act[k & 1][:w.B * H * H * cout].view(-1, cout)
            self._conv3(w, cur, H, cin, ld, conv, bn, dst, cout, bn_train)
            cur, ld, cin = dst, cout, cout
        return cur

    def _heads_fwd(self, w, bn_train):
        m = self.model
        NH, B, S = m.num_heads, w.B, w.S
        Mp = B * S * S
        heads = [getattr(m, f"segmentation_head_{i}") for i in range(NH)]
        dev = w.F3.device
        st = lambda get, shape: torch.stack([self._f32(get(h)).reshape(-1) for h in heads]).reshape(shape).contiguous()
        W1 = st(lambda h: h[0].psi[0].weight, (NH * HEAD_HID, HEAD_C))
        b1 = st(lambda h: h[0].psi[0].bias, (NH * HEAD_HID,))
        bnw = st(lambda h: h[0].psi[1].weight, (NH * HEAD_HID,))
        bnb = st(lambda h: h[0].psi[1].bias, (NH * HEAD_HID,))
        W2 = st(lambda h: h[0].psi[3].weight, (NH * HEAD_HID,))
        b2 = st(lambda h: h[0].psi[3].bias, (NH,))
        W3k = st(lambda h: h[1].weight, (NH, HEAD_C, 9)).transpose(1, 2).contiguous()
        b3 = st(lambda h: h[1].bias, (NH,))
        rm = torch.cat([h[0].psi[1].running_mean.detach().float() for h in heads]).to(dev).contiguous()
        rv = torch.cat([h[0].psi[1].running_var.detach().float() for h in heads]).to(dev).contiguous()
        if bn_train:
            w.mom.zero_()
            ops.heads_moments(w.F3, w.mom, Mp, NSLOTS)
        ops.heads_bn_from_moments(w.mom, W1, b1, bnw, bnb, rm, rv, w.hbn.scale, w.hbn.shift, w.hbn.mean, w.hbn.rstd, w.mom_sum, NH,
                                  NSLOTS, Mp, BN_EPS, BN_MOM, bn_train)
        if bn_train:
            with torch.no_grad():
                for i, h in enumerate(heads):
                    h[0].psi[1].running_mean.copy_(rm[HEAD_HID * i:HEAD_HID * (i + 1)])
                    h[0].psi[1].running_var.copy_(rv[HEAD_HID * i:HEAD_HID * (i + 1)])
                    h[0].psi[1].num_batches_tracked += 1
        ops.heads_gate_fwd(w.F3, W1, b1, w.hbn.scale, w.hbn.shift, W2, b2, w.G_, Mp, NH)
        ops.heads_conv_fwd(w.F3, w.G_, W3k, b3, w.out, B, S, S, NH)
        return w.out
