"""VAE on the HIP kernels, SURVEY.md 8(f) rank 1: the `self.vae.decode(z).sample` call of
`AnimationPipeline.decode_latents` (reference motionclone/pipelines/pipeline_animation.py:249-263) and the
`self.vae.encode(x).latent_dist.sample()` calls of `obtain_motion_representation` / `sample_video`
(motionclone/utils/motionclone_functions.py:31,64-65,125).

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
from .ops import CONV_S1, CONV_S2, CONV_UP

SD15_VAE_CONFIG = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
                       layers_per_block=2, norm_num_groups=32, scaling_factor=0.18215)
EPS = 1e-6           # resnet_eps handed to every block and to conv_norm_out by diffusers' Decoder
ACT_BUDGET = 1.2e9   # bytes of the largest activation of one chunk (a GEMM operand must stay below 2 GiB)


class VaeBlocks:
    """ResnetBlock2D / AttentionBlock shared by the encoder and the decoder"""

    def __init__(self, state_dict, cfg=None, device="cuda"):
        self.cfg = dict(cfg or SD15_VAE_CONFIG)
        self.dev = torch.device(device)
        assert self.cfg["norm_num_groups"] == 32, "kernels are specialised for GroupNorm(32)"
        assert all(c % 64 == 0 for c in self.cfg["block_out_channels"])
        self.w = Weights(state_dict, self.cfg, self.dev)

    def _padded_1x1(self, name, kpad=64):
        """1x1 conv on a few channels as a GEMM over 64-padded token rows: weight [N, kpad] (zeros beyond Cin)"""
        wt = self.w.sd[name + "weight"]
        n, k = wt.shape[0], wt.shape[1]
        m = torch.zeros(n, kpad)
        m[:, :k] = wt.reshape(n, k).float()
        return self.w._h(m), self.w._f(self.w.sd[name + "bias"]).unsqueeze(0)

    def _resnet(self, p, x, n, H, W):
        """ResnetBlock2D without time embedding (diffusers 0.16.0 models/resnet.py)"""
        w, hw, T = self.w, H * W, n * H * W
        h, st1 = ops.gn_fwd(x, None, w.vec(p + "norm1.weight"), w.vec(p + "norm1.bias"), True, n, hw, EPS)
        h = ops.gemm(h, w.conv(p + "conv1.weight"), bias=w.vec(p + "conv1.bias").unsqueeze(0), mode=CONV_S1,
                     geom=(H, W, H, W), m_out=T)
        h2, st2 = ops.gn_fwd(h, None, w.vec(p + "norm2.weight"), w.vec(p + "norm2.bias"), True, n, hw, EPS)
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
        nx, st = ops.gn_fwd(x, None, w.vec(p + "group_norm.weight"), w.vec(p + "group_norm.bias"), False, n, hw, EPS)
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



class VaeDecoderEngine(VaeBlocks):
    def __init__(self, state_dict, cfg=None, device="cuda"):
        super().__init__(state_dict, cfg, device)
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

    # ---- decode ----------------------------------------------------------------------------------------------
    def chunk_frames(self, h, w):
        ch = self.cfg["block_out_channels"]
        up = 2 ** (len(ch) - 1)
        per_frame = 2.0 * (h * up) * (w * up) * max(ch[0], ch[1] if len(ch) > 1 else ch[0])
        return max(1, int(ACT_BUDGET // per_frame))

    @ops.scoped
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


class DiagonalGaussianDistribution:
    """diffusers 0.16.0 models/vae.py: posterior of AutoencoderKL.encode.  `sample()` draws the normal noise with torch's
    generator (global RNG when None, as the reference relies on - SURVEY.md quirk 10); the arithmetic
    mean + exp(0.5 * clamp(logvar, -30, 20)) * noise runs in one HIP kernel."""

    def __init__(self, moments_tokens, n, lat, h, w):
        self._tok, self._n, self._lat, self._h, self._w = moments_tokens, n, lat, h, w

    def _shape(self):
        return (self._n, self._lat, self._h, self._w)

    def sample(self, generator=None):
        if generator is None:   # the reference draws from the global generator; a launcher lane has its own copy of that stream
            from . import lanes
            generator = lanes.serial_generator()
        noise = torch.randn(self._shape(), generator=generator, device=self._tok.device, dtype=torch.float16)
        return ops.vae_sample(self._tok, noise, self._lat)

    def mode(self):
        return ops.vae_mode(self._tok, self._n, self._lat, self._h, self._w)

    @property
    def mean(self):
        return self.mode()


class VaeEncoderEngine(VaeBlocks):
    """AutoencoderKL.encode: Encoder.forward + quant_conv -> posterior parameters (diffusers 0.16.0 models/vae.py)"""

    def __init__(self, state_dict, cfg=None, device="cuda"):
        super().__init__(state_dict, cfg, device)
        self.q_w, self.q_b = self._padded_1x1("quant_conv.")

    def chunk_frames(self, H, W):
        per_frame = 2.0 * H * W * self.cfg["block_out_channels"][0]
        return max(1, int(ACT_BUDGET // per_frame))

    @ops.scoped
    def encode_tokens(self, x):
        """x [n, 3, H, W] fp16 in [-1, 1] -> (moment tokens [(n h w), 2 * latent], h, w)"""
        cfg, w = self.cfg, self.w
        n, cin, H, W = x.shape
        L, ch, lat = cfg["layers_per_block"], tuple(cfg["block_out_channels"]), cfg["latent_channels"]
        if (H % (2 ** (len(ch) - 1))) or (W % (2 ** (len(ch) - 1))):
            raise ValueError("image size must be a multiple of %d" % 2 ** (len(ch) - 1))
        xin = ops.latent_to_cl(x.to(torch.float16).permute(1, 0, 2, 3).unsqueeze(0).contiguous(), 64)
        h = ops.gemm(xin, w.conv("encoder.conv_in.weight", pad_cin=64), bias=w.vec("encoder.conv_in.bias").unsqueeze(0),
                     mode=CONV_S1, geom=(H, W, H, W), m_out=n * H * W)
        del xin
        for i in range(len(ch)):
            for j in range(L):
                h = self._resnet("encoder.down_blocks.%d.resnets.%d." % (i, j), h, n, H, W)
            if i != len(ch) - 1:   # Downsample2D(padding=0): zero row / column only at the bottom / right
                d = "encoder.down_blocks.%d.downsamplers.0.conv." % i
                h = ops.gemm(h, w.conv(d + "weight"), bias=w.vec(d + "bias").unsqueeze(0), mode=CONV_S2,
                             geom=(H, W, H // 2, W // 2), m_out=n * (H // 2) * (W // 2), pad_front=False)
                H, W = H // 2, W // 2
        h = self._resnet("encoder.mid_block.resnets.0.", h, n, H, W)
        h = self._attention("encoder.mid_block.attentions.0.", h, n, H, W)
        h = self._resnet("encoder.mid_block.resnets.1.", h, n, H, W)
        st = ops.gn_stats(h, None, n, H * W, EPS)
        y = ops.gn_apply(h, None, st, w.vec("encoder.conv_norm_out.weight"), w.vec("encoder.conv_norm_out.bias"), True,
                         n, H * W)
        T = n * H * W
        m64 = torch.zeros((T, 64), dtype=torch.float16, device=self.dev)
        ops.gemm(y, w.conv("encoder.conv_out.weight"), bias=w.vec("encoder.conv_out.bias").unsqueeze(0), mode=CONV_S1,
                 geom=(H, W, H, W), m_out=T, out=m64[:, :2 * lat])
        moments = ops.gemm(m64, self.q_w, bias=self.q_b)     # quant_conv, [T, 2 * latent]
        return moments, H, W

    def encode(self, x):
        """-> DiagonalGaussianDistribution over [n, latent, H/8, W/8] (frames batched in chunks)"""
        n = x.shape[0]
        step = self.chunk_frames(x.shape[2], x.shape[3])
        toks = []
        for i in range(0, n, step):
            tok, h, w = self.encode_tokens(x[i:i + step])
            toks.append(tok)
        return DiagonalGaussianDistribution(torch.cat(toks) if len(toks) > 1 else toks[0], n,
                                            self.cfg["latent_channels"], h, w)
