"""VAE decoder on the HIP kernels: the `self.vae.decode(z).sample` call of `AnimationPipeline.decode_latents`
(reference motionclone/pipelines/pipeline_animation.py:249-263), SURVEY.md 8(f) rank 1.

The VAE is `diffusers==0.16.0` `AutoencoderKL` (not reference source; restated in oracle/vae_ref.py with that version's
state-dict keys).  Same design as the UNet engine: channels-last fp16 token matrices `[(frame y x), C]`, every conv an
implicit GEMM, GroupNorm+SiLU fused, the nearest-2x upsample folded into the following conv's gather.  The reference
decodes frame by frame; here a chunk of frames is one batch (same per-frame arithmetic).  The single-head 512-channel
AttentionBlock of the mid block runs as GEMMs per frame: S = q k^T / sqrt(C) -> fp32 row softmax -> P v, with v^T formed
directly by a GEMM (W_v . n^T) and the value bias added after P (rows of P sum to one).
"""
import math

import torch

from . import ops
from .engine import Weights
from .ops import CONV_S1, CONV_UP

SD15_VAE_CONFIG = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
                       layers_per_block=2, norm_num_groups=32, scaling_factor=0.18215)
EPS = 1e-6           # resnet_eps handed to every block and to conv_norm_out by diffusers' Decoder
ACT_BUDGET = 1.2e9   # bytes of the largest activation of one chunk (a GEMM operand must stay below 2 GiB)


class VaeDecoderEngine:
    def __init__(self, state_dict, cfg=None, device="cuda"):
        self.cfg = dict(cfg or SD15_VAE_CONFIG)
        self.dev = torch.device(device)
        assert self.cfg["norm_num_groups"] == 32, "kernels are specialised for GroupNorm(32)"
        assert all(c % 64 == 0 for c in self.cfg["block_out_channels"])
        self.w = Weights(state_dict, self.cfg, self.dev)
        sd, w, lat = state_dict, self.w, self.cfg["latent_channels"]
        # post_quant_conv as a [lat, 64] GEMM on the 64-padded latent tokens; conv_out rows padded 3 -> 4
        pq = torch.zeros(lat, 64)
        pq[:, :lat] = sd["post_quant_conv.weight"].reshape(lat, lat).float()
        self.pq_w = w._h(pq)
        self.pq_b = w._f(sd["post_quant_conv.bias"]).unsqueeze(0)
        co = w.conv("decoder.conv_out.weight")
        self.out_c = co.shape[0]
        n4 = (self.out_c + 3) // 4 * 4
        self.co_w = torch.zeros((n4, co.shape[1]), dtype=torch.float16, device=self.dev)
        self.co_w[:self.out_c] = co
        cb = torch.zeros(n4)
        cb[:self.out_c] = sd["decoder.conv_out.bias"].float()
        self.co_b = w._f(cb).unsqueeze(0)

    # ---- blocks --------------------------------------------------------------------------------------------
    def _resnet(self, p, x, n, H, W):
        """ResnetBlock2D without time embedding (diffusers 0.16.0 models/resnet.py)"""
        w, hw, T = self.w, H * W, n * H * W
        st1 = ops.gn_stats(x, None, n, hw, EPS)
        h = ops.gn_apply(x, None, st1, w.vec(p + "norm1.weight"), w.vec(p + "norm1.bias"), True, n, hw)
        h = ops.gemm(h, w.conv(p + "conv1.weight"), bias=w.vec(p + "conv1.bias").unsqueeze(0), mode=CONV_S1,
                     geom=(H, W, H, W), m_out=T)
        st2 = ops.gn_stats(h, None, n, hw, EPS)
        h2 = ops.gn_apply(h, None, st2, w.vec(p + "norm2.weight"), w.vec(p + "norm2.bias"), True, n, hw)
        del h
        if (p + "conv_shortcut.weight") in w.sd:
            x = ops.gemm(x, w.lin(p + "conv_shortcut.weight"), bias=w.vec(p + "conv_shortcut.bias").unsqueeze(0))
        return ops.gemm(h2, w.conv(p + "conv2.weight"), bias=w.vec(p + "conv2.bias").unsqueeze(0), residual=x,
                        mode=CONV_S1, geom=(H, W, H, W), m_out=T)

    def _attention(self, p, x, n, H, W):
        """AttentionBlock, one head (diffusers 0.16.0 models/attention.py): softmax(q k^T / sqrt(C)) v, fp32 softmax"""
        w, hw = self.w, H * W
        C = x.shape[1]
        if hw % 64:
            raise NotImplementedError("VAE attention needs a multiple of 64 latent pixels per frame (got %d)" % hw)
        st = ops.gn_stats(x, None, n, hw, EPS)
        nx = ops.gn_apply(x, None, st, w.vec(p + "group_norm.weight"), w.vec(p + "group_norm.bias"), False, n, hw)
        q = ops.gemm(nx, w.lin(p + "query.weight"), bias=w.vec(p + "query.bias").unsqueeze(0))
        k = ops.gemm(nx, w.lin(p + "key.weight"), bias=w.vec(p + "key.bias").unsqueeze(0))
        o = ops.empty((n * hw, C), x)
        wv, bv = w.lin(p + "value.weight"), w.vec(p + "value.bias").unsqueeze(0)
        scale = 1.0 / math.sqrt(C)
        for f in range(n):
            rows = slice(f * hw, (f + 1) * hw)
            s = ops.gemm(q[rows], k[rows], alpha=scale)            # [hw, hw] scores
            ops.softmax_rows_(s)
            vt = ops.gemm(wv, nx[rows])                            # v^T = W_v . n^T   [C, hw] (bias added after P)
            ops.gemm(s, vt, bias=bv, out=o[rows])                  # P v + b_v
        return ops.gemm(o, w.lin(p + "proj_attn.weight"), bias=w.vec(p + "proj_attn.bias").unsqueeze(0), residual=x)

    # ---- decode ----------------------------------------------------------------------------------------------
    def chunk_frames(self, h, w):
        ch = self.cfg["block_out_channels"]
        up = 2 ** (len(ch) - 1)
        per_frame = 2.0 * (h * up) * (w * up) * max(ch[0], ch[1] if len(ch) > 1 else ch[0])
        return max(1, int(ACT_BUDGET // per_frame))

    def decode_tokens(self, z, scale=1.0):
        """z [n, latent, h, w] fp16 -> (tokens [(n H W), 4] fp16 with the first 3 columns = RGB, H, W)"""
        cfg, w = self.cfg, self.w
        n, lat, h, wd = z.shape
        L, ch = cfg["layers_per_block"], tuple(cfg["block_out_channels"])
        T = n * h * wd
        zin = ops.latent_to_cl(z.to(torch.float16).permute(1, 0, 2, 3).unsqueeze(0).contiguous(), 64)   # [T, 64]
        pq = torch.zeros((T, 64), dtype=torch.float16, device=self.dev)
        ops.gemm(zin, self.pq_w, bias=self.pq_b, alpha=float(scale), out=pq[:, :lat])
        x = ops.gemm(pq, w.conv("decoder.conv_in.weight", pad_cin=64), bias=w.vec("decoder.conv_in.bias").unsqueeze(0),
                     mode=CONV_S1, geom=(h, wd, h, wd), m_out=T)
        del zin, pq
        x = self._resnet("decoder.mid_block.resnets.0.", x, n, h, wd)
        x = self._attention("decoder.mid_block.attentions.0.", x, n, h, wd)
        x = self._resnet("decoder.mid_block.resnets.1.", x, n, h, wd)
        H, W = h, wd
        for i in range(len(ch)):
            for j in range(L + 1):
                x = self._resnet("decoder.up_blocks.%d.resnets.%d." % (i, j), x, n, H, W)
            if i != len(ch) - 1:   # Upsample2D: nearest 2x folded into the conv's gather
                u = "decoder.up_blocks.%d.upsamplers.0.conv." % i
                x = ops.gemm(x, w.conv(u + "weight"), bias=w.vec(u + "bias").unsqueeze(0), mode=CONV_UP,
                             geom=(H, W, 2 * H, 2 * W), m_out=n * 4 * H * W)
                H, W = 2 * H, 2 * W
        st = ops.gn_stats(x, None, n, H * W, EPS)
        y = ops.gn_apply(x, None, st, w.vec("decoder.conv_norm_out.weight"), w.vec("decoder.conv_norm_out.bias"), True,
                         n, H * W)
        del x
        out = ops.gemm(y, self.co_w, bias=self.co_b, mode=CONV_S1, geom=(H, W, H, W), m_out=n * H * W)
        return out, H, W

    def decode(self, z):
        """AutoencoderKL.decode(z).sample: [n, latent, h, w] -> [n, 3, 8h, 8w] fp16 (frames batched in chunks)"""
        n = z.shape[0]
        step = self.chunk_frames(z.shape[2], z.shape[3])
        outs = []
        for i in range(0, n, step):
            tok, H, W = self.decode_tokens(z[i:i + step])
            m = min(step, n - i)
            outs.append(ops.cl_to_latent(tok, 1, self.out_c, m, H, W)[0].permute(1, 0, 2, 3))
        return torch.cat(outs)

    def decode_video(self, latents):
        """decode_latents (pipeline_animation.py:249-263) up to the host copy: [1, 4, F, h, w] -> float32 [1, 3, F, H, W]"""
        b, c, F, h, w = latents.shape
        assert b == 1
        z = latents[0].permute(1, 0, 2, 3)
        step = self.chunk_frames(h, w)
        outs = []
        for i in range(0, F, step):
            tok, H, W = self.decode_tokens(z[i:i + step], scale=1.0 / self.cfg["scaling_factor"])
            outs.append(ops.video_post(tok, self.out_c, min(step, F - i), H, W))
        return torch.cat(outs, dim=2)
