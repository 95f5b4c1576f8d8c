"""Execution engine of the guided-DDIM hot path on MI355X.

One `UNet3DEngine` holds the packed fp16 weights of a reference `UNet3DConditionModel`
(reference/motionclone/models/unet.py:38-249) and runs

  * `forward`      - unet_customized_forward (motionclone_functions.py:478-662) as a fixed sequence of
                     HIP kernel launches over channels-last token matrices [(b f y x), C];
  * `backward`     - the data-gradient of the guidance loss w.r.t. the latent, i.e. what
                     torch.autograd.grad does at motionclone_functions.py:236, as a hand-written
                     reverse schedule (a tape of closures recorded by the in-graph half of the
                     forward: conv_in .. up_blocks[guidance_block]); weight gradients are never formed;
  * `guided_step` / `plain_step` / `extract_representation` - single_step_video (:173-257) and the
                     model part of obtain_motion_representation (:74-79).

Layout decisions: activations never leave [tokens, C]; the reference's einops
round-trips, head reshapes, skip `torch.cat`s and the nearest-2x upsample are index math inside the
kernels; q|k|v projections are fused into one GEMM; the 22 time_emb_proj Linears are one GEMM per
forward and enter conv1 as a per-batch bias; gradients are carried at `grad_scale` x their value so
fp16 gradient activations stay in range (the scale is divided out in the last kernel).
"""
import math

import torch

from . import ops

DENSE, CONV_S1, CONV_S2, CONV_UP, TCONV_S2 = ops.DENSE, ops.CONV_S1, ops.CONV_S2, ops.CONV_UP, ops.TCONV_S2
CIN_PAD = 64


def default_config():
    """SD-1.5 UNet + configs/model_config/model_config.yaml (the only architecture the reference ships)."""
    return dict(in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
                cross_attention_dim=768, attention_heads=8, norm_num_groups=32, norm_eps=1e-5,
                down_has_attn=(True, True, True, False), up_has_attn=(False, True, True, True),
                motion_heads=8, motion_pe_max_len=32, motion_mid_block=False)


class Geo:
    """token geometry of one resolution level"""

    def __init__(self, B, F, H, W):
        self.B, self.F, self.H, self.W = B, F, H, W
        self.frames = B * F
        self.hw = H * W
        self.T = self.frames * self.hw

    def down(self):
        return Geo(self.B, self.F, self.H // 2, self.W // 2)

    def up(self):
        return Geo(self.B, self.F, self.H * 2, self.W * 2)


def _pe_table(max_len, dim):
    pos = torch.arange(max_len, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, dim, 2, dtype=torch.float32) * (-math.log(10000.0) / dim))
    pe = torch.zeros(max_len, dim)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


class Weights:
    """Packs a reference state-dict into kernel layouts on `device` (load-time only)."""

    def __init__(self, sd, cfg, device):
        self.sd = sd
        self.cfg = cfg
        self.dev = device
        self._c = {}

    def _get(self, key, fn):
        if key not in self._c:
            t = fn()
            # packed on whichever stream touches the weight first; lanes on OTHER streams look it up right after this returns,
            # so the (cold-path) packing is completed before it is published.  Never reached under graph capture: the eager
            # pass that precedes every capture has filled the cache.
            if isinstance(t, torch.Tensor) and t.is_cuda and not torch.cuda.is_current_stream_capturing():
                torch.cuda.current_stream(t.device).synchronize()
            self._c[key] = t
        return self._c[key]

    def _h(self, t):
        return t.detach().to(self.dev, torch.float16).contiguous()

    def _f(self, t):
        return t.detach().to(self.dev, torch.float32).contiguous()

    def vec(self, name):
        return self._get(("vec", name), lambda: self._f(self.sd[name]))

    def lin(self, name):
        """nn.Linear / 1x1 conv weight as [N, K]"""
        return self._get(("lin", name), lambda: self._h(self.sd[name].reshape(self.sd[name].shape[0], -1)))

    def lin_t(self, name):
        """transposed copy [K, N] for the data-gradient GEMM"""
        return self._get(("lin_t", name),
                         lambda: self._h(self.sd[name].reshape(self.sd[name].shape[0], -1).t()))

    def cat_lin(self, names):
        return self._get(("cat", tuple(names)), lambda: self._h(torch.cat([self.sd[n] for n in names], 0)))

    def cat_lin_t(self, names):
        return self._get(("cat_t", tuple(names)), lambda: self._h(torch.cat([self.sd[n] for n in names], 0).t()))

    def conv(self, name, pad_cin=0):
        def mk():
            w = self.sd[name].to(torch.float32).permute(0, 2, 3, 1)  # [Cout, 3, 3, Cin]
            if pad_cin and w.shape[3] < pad_cin:
                w = torch.nn.functional.pad(w, (0, pad_cin - w.shape[3]))
            return self._h(ops.pack_conv_k(w.reshape(w.shape[0], 9, w.shape[3])))
        return self._get(("conv", name, pad_cin), mk)

    def conv_dgrad(self, name, stride=1, pad_cin=0):
        """[Cin, 9*Cout]: flipped taps for stride 1 (a conv again), plain taps for the stride-2 transpose"""
        def mk():
            w = self.sd[name].to(torch.float32)
            if stride == 1:
                w = w.flip(2, 3)
            w = w.permute(1, 2, 3, 0)  # [Cin, 3, 3, Cout]
            w = ops.pack_conv_k(w.reshape(w.shape[0], 9, w.shape[3]))
            if pad_cin and w.shape[0] < pad_cin:
                w = torch.nn.functional.pad(w, (0, 0, 0, pad_cin - w.shape[0]))
            return self._h(w)
        return self._get(("conv_d", name, stride, pad_cin), mk)

    def geglu_lin(self, name):
        """FeedForward's first Linear with rows interleaved (h_j, gate_j) for the fused GEGLU epilogue"""
        return self._get(("geglu_w", name), lambda: self._h(ops.interleave_geglu(self.sd[name])))

    def geglu_vec(self, name):
        return self._get(("geglu_b", name), lambda: self._f(ops.interleave_geglu(self.sd[name])).unsqueeze(0))

    def pe(self, dim):
        return self._get(("pe", dim), lambda: self._f(_pe_table(self.cfg["motion_pe_max_len"], dim)))


class Tape:
    """reverse schedule: closures appended in forward order, run backwards; grads keyed by tensor identity"""

    def __init__(self, grad_batch=None):
        self.fns = []
        self.grads = {}                # id(activation) -> gradient
        self.alive = {}                # id(activation) -> activation, while its gradient is pending
        # differentiate only these batch elements of a B > 1 forward: an index b, or a range (b0, b1) = elements b0 .. b1 - 1
        # (the conditional halves of several videos batched as [u_1 .. u_V | c_1 .. c_V]); None: all
        self.grad_batch = grad_batch
        self.grad_scale = 1.0          # the gradients on this tape are carried at this multiple of their value

    def add(self, fn):
        self.fns.append(fn)

    def give(self, x, dx):
        """accumulate dx into the gradient of activation x"""
        key = id(x)
        if key in self.grads:
            g = self.grads[key]
            ops.add(g, dx, out=g)
        else:
            self.grads[key] = dx
            self.alive[key] = x            # held until its gradient is taken: id(x) stays unique meanwhile

    def take(self, x):
        self.alive.pop(id(x), None)
        return self.grads.pop(id(x), None)

    def run(self):
        """closures run in reverse order and are DROPPED as they finish: the activations a block saved for its backward are
        released (to the graph's memory pool, for the blocks still to come) as soon as that block is done, instead of
        staying alive until the whole backward has run (round 5)"""
        fns, self.fns = self.fns, []
        while fns:
            fn = fns.pop()
            fn()
            del fn

    def whole_batch(self):
        """the same tape for modules whose forward ran on the differentiated batch elements ONLY (the shared prefix of a
        duplicated batch, UNet3DEngine.forward(dup=True)): nothing is sliced there"""
        return _WholeBatchTape(self)


class _WholeBatchTape:
    """Tape proxy with grad_batch = None; every other attribute (closures, gradients, grad_scale, latent_grad) is the tape's"""

    def __init__(self, tape):
        object.__setattr__(self, "_t", tape)

    grad_batch = None

    def __getattr__(self, k):
        return getattr(self._t, k)

    def __setattr__(self, k, v):
        setattr(self._t, k, v)


class UNet3DEngine:
    def __init__(self, state_dict, cfg=None, device="cuda", guidance_block=1, grad_scale=1024.0, guidance_blocks=None):
        """guidance_blocks: input_config.motion_guidance_blocks (e.g. ['up_blocks.1']); a temporal attention is hooked when
        its name contains one of the entries (util.py:434-440), and the up blocks up to the index of the LAST entry
        stay in the differentiated half (motionclone_functions.py:602).  guidance_block = that index when no list is given."""
        self.cfg = dict(cfg or default_config())
        self.dev = torch.device(device)
        self.w = Weights(state_dict, self.cfg, self.dev)
        if guidance_blocks is None:
            guidance_blocks = ["up_blocks.%d" % guidance_block]
        self.guidance_blocks = tuple(guidance_blocks)
        self.guidance_block = int(self.guidance_blocks[-1].split(".")[-1])
        for blk in self.guidance_blocks:
            kind, _, idx = blk.partition(".")
            if kind not in ("down_blocks", "up_blocks", "mid_block") or (kind == "up_blocks" and idx.split(".")[0].isdigit()
                                                                         and int(idx.split(".")[0]) > self.guidance_block):
                # an up block behind the last entry runs under no_grad and after the extraction's early return in the
                # reference (:627-652): its recorded q / k would be stale there
                raise NotImplementedError("motion_guidance_blocks entry %r cannot be honoured (last entry: %r)"
                                          % (blk, self.guidance_blocks[-1]))
        self.grad_scale = float(grad_scale)
        if self.dev.type == "cuda":
            ops.prepare_tile_counters(self.dev)   # the tile loop's zeroed counter slab must exist before any graph capture
        # guided / plain steps feed the UNet a batch whose halves differ in the text only: run what precedes the first
        # cross-attention once (forward: dup).  False = the duplicated batch of rounds 1-4 (A/B: bench.py --no-shared-prefix)
        self.share_prefix = True
        self.G = self.cfg["norm_num_groups"]
        assert self.G == 32, "kernels are specialised for GroupNorm(32)"
        ch = self.cfg["block_out_channels"]
        assert all(c % 64 == 0 for c in ch) and self.cfg["cross_attention_dim"] % 64 == 0
        self._build_temb_plan()

    # ---- time embedding: all time_emb_proj Linears as one GEMM ----------------------------------
    def _resnet_names(self):
        L = self.cfg["layers_per_block"]
        names = []
        for i in range(4):
            names += ["down_blocks.%d.resnets.%d." % (i, j) for j in range(L)]
        names += ["mid_block.resnets.0.", "mid_block.resnets.1."]
        for i in range(4):
            names += ["up_blocks.%d.resnets.%d." % (i, j) for j in range(L + 1)]
        return names

    def _build_temb_plan(self):
        sd = self.w.sd
        names = self._resnet_names()
        self.temb_off = {}
        off = 0
        for n in names:
            c = sd[n + "time_emb_proj.weight"].shape[0]
            self.temb_off[n] = (off, c)
            off += c
        self.temb_w = self.w.cat_lin([n + "time_emb_proj.weight" for n in names])
        self.temb_b = torch.cat([sd[n + "time_emb_proj.bias"].float() + sd[n + "conv1.bias"].float()
                                 for n in names]).to(self.dev).contiguous()

    def _time_bias(self, t, B, like):
        """-> fp32 [1, sum Cout]: conv1.bias + time_emb_proj(silu(time_embedding(t)))  (resnet.py:191-195).  The timestep
        is one scalar for the whole batch (unet.py:386-391 broadcasts it), so the bias row is shared by every batch element:
        one row is computed, and each resnet's slice of it is a contiguous view (no per-resnet copies)."""
        w = self.w
        tt = torch.full((1,), float(t), dtype=torch.float32, device=self.dev)
        e = ops.timestep_embed(tt, self.cfg["block_out_channels"][0], like)
        e = ops.gemm(e, w.lin("time_embedding.linear_1.weight"), bias=w.vec("time_embedding.linear_1.bias").unsqueeze(0))
        e = ops.gemm(ops.silu(e), w.lin("time_embedding.linear_2.weight"),
                     bias=w.vec("time_embedding.linear_2.bias").unsqueeze(0))
        allp = ops.gemm(ops.silu(e), self.temb_w)
        return (allp.float() + self.temb_b.unsqueeze(0)).contiguous()

    # ---- batch slicing for the backward ------------------------------------------------------------------
    @staticmethod
    def _brange(geo, b):
        """grad_batch -> (b0, b1): None = every batch element, an int = that one, a pair = the range"""
        if b is None:
            return 0, geo.B
        if isinstance(b, (tuple, list)):
            return int(b[0]), int(b[1])
        return int(b), int(b) + 1

    @staticmethod
    def _bslice(geo, b):
        """The forward may run B = 2 (uncond | cond) - or B = 2 V for V videos batched as [u_1 .. u_V | c_1 .. c_V] - while only
        the batch elements `b` (an index or a range (b0, b1)) are differentiated (the reference runs them as separate B = 1
        calls, motionclone_functions.py:216-223 - per-sample results are identical).  Returns the geometry of the differentiated
        part and a function that cuts it out of any saved per-token / per-frame / per-batch tensor; token order is batch-major,
        so every slice is a contiguous row range (a view)."""
        if b is None:
            return geo, (lambda t: t)
        b0, b1 = UNet3DEngine._brange(geo, b)
        T1 = geo.T // geo.B

        def cut(t):
            if t is None:
                return None
            n = t.shape[0]
            if n == geo.T:
                return t[b0 * T1:b1 * T1]
            if n == geo.frames:
                return t[b0 * geo.F:b1 * geo.F]
            per = n // geo.B          # per-batch rows (text keys/values, time bias)
            return t[b0 * per:b1 * per]
        return Geo(b1 - b0, geo.F, geo.H, geo.W), cut

    def _gn_gemm(self, x, gname, bname, wname, wbias, fr, hw):
        """GroupNorm(32, eps 1e-6, no activation) + the 1x1 proj_in (attention.py:105-117, motion_module.py:145-151) ->
        (proj_in output, GroupNorm statistics for the backward).  At the K = 320 level one launch for norm-apply + GEMM
        (mc_norm_gemm_f16: the normalised tensor never reaches HBM), elsewhere GroupNorm then GEMM."""
        w = self.w
        gN, bN, W, bias = w.vec(gname), w.vec(bname), w.lin(wname), w.vec(wbias).unsqueeze(0)
        gp = ops.gnp_of(x, hw)       # statistics left by the epilogue of the GEMM that produced x (round 6), or None
        r = ops.norm_gemm(x, W, 2, gN, bN, bias=bias, hw=hw, eps=1e-6, gnp=gp)
        if r is not None:
            return r
        hn, st = ops.gn_fwd(x, None, gN, bN, False, fr, hw, 1e-6, gnp=gp)
        return ops.gemm(hn, W, bias=bias), st

    def _ln_gemm(self, h, gname, bname, W, tape, bias=None, pe=None, hw=0):
        """LayerNorm (+ temporal position table) + Linear -> (output, (mean, rstd) per row or None without a tape)"""
        w = self.w
        g, b = w.vec(gname), w.vec(bname)
        r = ops.norm_gemm(h, W, 1, g, b, bias=bias, pe=pe, hw=hw, save_stats=tape is not None)
        if r is not None:
            return r
        n, ls = ops.layernorm_fwd(h, g, b, pe=pe, hw=hw, save_stats=tape is not None)
        return ops.gemm(n, W, bias=bias), ls

    def _ff_geglu(self, h, gname, bname, prefix, geo, tape):
        """LayerNorm + FeedForward's first Linear + GEGLU on the rows `h` -> (h * gelu(gate), taped pre-activation or None,
        LayerNorm statistics or None).  No backward: the product is formed in the GEMM epilogue and the [T, 8C] pre-activation
        never exists.  With a tape only the differentiated batch element needs it: its rows take the unfused Linear + geglu
        kernel, the rows of the other batch element (the unconditional half of a guided step) stay on the fused epilogue -
        half the geglu launches' bytes and a third of that GEMM's output traffic gone.  At the K = 320 level the LayerNorm
        runs inside the GEMMs (mc_norm_gemm_f16)."""
        w = self.w
        wname, biasname = prefix + "ff.net.0.proj.weight", prefix + "ff.net.0.proj.bias"
        g, b = w.vec(gname), w.vec(bname)
        T, gb = geo.T, (tape.grad_batch if tape is not None else None)
        if tape is None:
            r = ops.norm_gemm(h, w.geglu_lin(wname), 1, g, b, bias=w.geglu_vec(biasname), save_stats=False, geglu=True)
            if r is not None:
                return r[0], None, None
            n, _ = ops.layernorm_fwd(h, g, b, save_stats=False)
            return ops.gemm(n, w.geglu_lin(wname), bias=w.geglu_vec(biasname), geglu=True), None, None
        T1 = T // geo.B
        b0, b1 = self._brange(geo, gb)
        lo, hi = b0 * T1, b1 * T1
        W1, bias1 = w.lin(wname), w.vec(biasname).unsqueeze(0)
        gg = ops.empty((T, W1.shape[0] // 2), h)
        ls = ops.empty((T, 2), h, torch.float32)       # only rows lo .. hi (the differentiated batch elements) are read later
        r = ops.norm_gemm(h[lo:hi], W1, 1, g, b, bias=bias1, stats=ls[lo:hi])
        n = None
        if r is not None:
            ff1 = r[0]
        else:                                          # outside the streaming kernel's shapes: LayerNorm, then the GEMMs
            n, ls = ops.layernorm_fwd(h, g, b, save_stats=True)
            ff1 = ops.gemm(n[lo:hi], W1, bias=bias1)
        ops.geglu_fwd(ff1, out=gg[lo:hi])
        for a, e in ((0, lo), (hi, T)):
            if e <= a:
                continue
            if n is None and ops.norm_gemm(h[a:e], w.geglu_lin(wname), 1, g, b, bias=w.geglu_vec(biasname), save_stats=False,
                                           geglu=True, out=gg[a:e]) is not None:
                continue
            if n is None:
                n, _ = ops.layernorm_fwd(h, g, b, save_stats=False)
            ops.gemm(n[a:e], w.geglu_lin(wname), bias=w.geglu_vec(biasname), geglu=True, out=gg[a:e])
        return gg, ff1, ls

    # ---- modules -----------------------------------------------------------------------------------
    def _resnet(self, p, x, x2, tb_all, geo, tape):
        """ResnetBlock3D.forward (resnet.py:183-213); x2 = skip tensor of the up-block concat or None"""
        w, cfg = self.w, self.cfg
        eps = cfg["norm_eps"]
        fr, hw, H, W = geo.frames, geo.hw, geo.H, geo.W
        off, cout = self.temb_off[p]
        tb = tb_all[:, off:off + cout]        # one shared row [1, cout] (a contiguous view), or one row per batch element
        rpb = 0
        if tb_all.shape[0] > 1:
            tb, rpb = tb.contiguous(), geo.F * hw
        g1w, b1 = w.vec(p + "norm1.weight"), w.vec(p + "norm1.bias")
        g2, b2 = w.vec(p + "norm2.weight"), w.vec(p + "norm2.bias")
        h1, st1 = ops.gn_fwd(x, x2, g1w, b1, True, fr, hw, eps, gnp=ops.gnp_of(x, hw) if x2 is None else None)
        # (round 6) norm2's statistics come from conv1's epilogue where the library runs a one-pass ring kernel (gp; else None)
        h2, gp = ops.gemm(h1, w.conv(p + "conv1.weight"), bias=tb, rows_per_batch=rpb, mode=CONV_S1,
                          geom=(H, W, H, W), m_out=geo.T, gn_hw=hw)
        del h1
        h3, st2 = ops.gn_fwd(h2, None, g2, b2, True, fr, hw, eps, gnp=gp)
        has_sc = (p + "conv_shortcut.weight") in w.sd
        if has_sc:
            sc = ops.gemm(x, w.lin(p + "conv_shortcut.weight"), a2=x2,
                          bias=w.vec(p + "conv_shortcut.bias").unsqueeze(0))
        else:
            assert x2 is None
            sc = x
        # (the block's output is normalised next by Transformer3DModel.norm / the motion module's norm / the next block's norm1:
        # its statistics ride on the tensor, ops.gnp_of)
        out, gpo = ops.gemm(h3, w.conv(p + "conv2.weight"), bias=w.vec(p + "conv2.bias").unsqueeze(0), residual=sc,
                            mode=CONV_S1, geom=(H, W, H, W), m_out=geo.T, gn_hw=hw)
        ops.tag_gnp(out, gpo, hw)
        if tape is not None:
            g1, cut = self._bslice(geo, tape.grad_batch)
            bx, bx2, bst1, bh2, bst2 = cut(x), cut(x2), cut(st1), cut(h2), cut(st2)
            H1, W1, fr1, hw1 = g1.H, g1.W, g1.frames, g1.hw

            def bwd():
                dout = tape.take(out)
                if dout is None:
                    return
                dh3 = ops.gemm(dout, w.conv_dgrad(p + "conv2.weight"), mode=CONV_S1, geom=(H1, W1, H1, W1), m_out=g1.T)
                dh2 = ops.gn_bwd(bh2, None, dh3, bst2, g2, b2, True, fr1, hw1)
                dh1 = ops.gemm(dh2, w.conv_dgrad(p + "conv1.weight"), mode=CONV_S1, geom=(H1, W1, H1, W1), m_out=g1.T)
                if has_sc:
                    dx = ops.gn_bwd(bx, bx2, dh1, bst1, g1w, b1, True, fr1, hw1)
                    ops.gemm(dout, w.lin_t(p + "conv_shortcut.weight"), residual=dx, out=dx)
                else:
                    dx = ops.gn_bwd(bx, None, dh1, bst1, g1w, b1, True, fr1, hw1, out=dout, accumulate=True)
                c1 = x.shape[1]
                if x2 is None:
                    tape.give(x, dx)
                else:
                    tape.give(x, dx[:, :c1])
                    tape.give(x2, dx[:, c1:])
            tape.add(bwd)
        return out

    def _spatial(self, p, x, text2d, n_text, geo, tape, shared_geo=None):
        """Transformer3DModel + BasicTransformerBlock (attention.py:95-142,256-300).

        shared_geo (forward(dup=True)): the batch of `geo` is two copies of the same rows that differ in their text only
        ([u_1 .. u_V | c_1 .. c_V] of the same latents, motionclone_functions.py:216-223,248-253) and `x` holds ONE copy
        (geometry shared_geo, V batch elements).  Everything in front of the cross-attention - GroupNorm, proj_in, the
        self-attention with its output projection, LayerNorm + attn2.to_q - does not see the text: it runs once on the V
        elements and both halves use the result (the reference computes it twice with identical values).  The backward
        differentiates the conditional half only, which IS the shared rows: nothing is sliced in front of the junction."""
        w, cfg = self.w, self.cfg
        heads = cfg["attention_heads"]
        C = x.shape[1]
        d = C // heads
        fr, hw, T = geo.frames, geo.hw, geo.T
        gA = shared_geo if shared_geo is not None else geo
        b = p + "transformer_blocks.0."
        gN, bN = w.vec(p + "norm.weight"), w.vec(p + "norm.bias")
        h0, st = self._gn_gemm(x, p + "norm.weight", p + "norm.bias", p + "proj_in.weight", p + "proj_in.bias", gA.frames, hw)
        # self-attention
        qkv, ls1 = self._ln_gemm(h0, b + "norm1.weight", b + "norm1.bias",
                                 w.cat_lin([b + "attn1.to_q.weight", b + "attn1.to_k.weight", b + "attn1.to_v.weight"]), tape)
        a1, lse1 = ops.attn_fwd(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], hw, hw, heads, d, gA.frames,
                                need_lse=tape is not None)
        h1 = ops.gemm(a1, w.lin(b + "attn1.to_out.0.weight"), bias=w.vec(b + "attn1.to_out.0.bias").unsqueeze(0),
                      residual=h0)
        # cross-attention to the text (K/V are frame-invariant: computed once per batch element)
        q2, ls2 = self._ln_gemm(h1, b + "norm2.weight", b + "norm2.bias", w.lin(b + "attn2.to_q.weight"), tape)
        if shared_geo is not None:     # the junction: both halves continue from the same rows
            h1B, q2B, xB = ops.dup_rows(h1), ops.dup_rows(q2), ops.dup_rows(x)
        else:
            h1B, q2B, xB = h1, q2, x
        kv = ops.gemm(text2d, w.cat_lin([b + "attn2.to_k.weight", b + "attn2.to_v.weight"]))
        a2, lse2 = ops.attn_fwd(q2B, kv[:, :C], kv[:, C:], hw, n_text, heads, d, fr, kv_bdiv=geo.F,
                                need_lse=tape is not None)
        h2 = ops.gemm(a2, w.lin(b + "attn2.to_out.0.weight"), bias=w.vec(b + "attn2.to_out.0.bias").unsqueeze(0),
                      residual=h1B)
        del h1B, q2B
        # GEGLU feed-forward
        gg, bff1, ls3 = self._ff_geglu(h2, b + "norm3.weight", b + "norm3.bias", b, geo, tape)
        h3 = ops.gemm(gg, w.lin(b + "ff.net.2.weight"), bias=w.vec(b + "ff.net.2.bias").unsqueeze(0), residual=h2)
        del gg
        out = ops.gemm(h3, w.lin(p + "proj_out.weight"), bias=w.vec(p + "proj_out.bias").unsqueeze(0), residual=xB)
        del xB
        if tape is None:
            return out

        g1, cut = self._bslice(geo, tape.grad_batch)
        cutA = (lambda t: t) if shared_geo is not None else cut      # tensors in front of the junction hold the shared rows
        bx, bst, bh0, bh1, bh2 = cutA(x), cutA(st), cutA(h0), cutA(h1), cut(h2)
        bls1, bls2, bls3, bqkv, ba1, blse1 = cutA(ls1), cutA(ls2), cut(ls3), cutA(qkv), cutA(a1), cutA(lse1)
        bq2, bkv, ba2, blse2 = cutA(q2), cut(kv), cut(a2), cut(lse2)   # (bff1: already the differentiated rows)
        fr1, hw1, T1 = g1.frames, g1.hw, g1.T
        assert bx.shape[0] == T1 and bq2.shape[0] == T1 and ba2.shape[0] == T1

        def bwd():
            dout = tape.take(out)
            if dout is None:
                return
            dh3 = ops.gemm(dout, w.lin_t(p + "proj_out.weight"))
            dg = ops.gemm(dh3, w.lin_t(b + "ff.net.2.weight"))
            dff1 = ops.geglu_bwd(dg, bff1)
            dn3 = ops.gemm(dff1, w.lin_t(b + "ff.net.0.proj.weight"))
            dh2 = ops.layernorm_bwd(dn3, bh2, bls3, w.vec(b + "norm3.weight"), add=dh3)
            da2 = ops.gemm(dh2, w.lin_t(b + "attn2.to_out.0.weight"))
            dq2, _, _ = ops.attn_bwd(bq2, bkv[:, :C], bkv[:, C:], ba2, da2, blse2, hw1, n_text, heads, d, fr1,
                                     kv_bdiv=g1.F, need_dkv=False)
            dn2 = ops.gemm(dq2, w.lin_t(b + "attn2.to_q.weight"))
            dh1 = ops.layernorm_bwd(dn2, bh1, bls2, w.vec(b + "norm2.weight"), add=dh2)
            da1 = ops.gemm(dh1, w.lin_t(b + "attn1.to_out.0.weight"))
            dqkv = ops.empty((T1, 3 * C), x)
            ops.attn_bwd(bqkv[:, :C], bqkv[:, C:2 * C], bqkv[:, 2 * C:], ba1, da1, blse1, hw1, hw1, heads, d, fr1,
                         dq=dqkv[:, :C], dk=dqkv[:, C:2 * C], dv=dqkv[:, 2 * C:])
            dn1 = ops.gemm(dqkv, w.cat_lin_t([b + "attn1.to_q.weight", b + "attn1.to_k.weight", b + "attn1.to_v.weight"]))
            dh0 = ops.layernorm_bwd(dn1, bh0, bls1, w.vec(b + "norm1.weight"), add=dh1)
            dhn = ops.gemm(dh0, w.lin_t(p + "proj_in.weight"))
            dx = ops.gn_bwd(bx, None, dhn, bst, gN, bN, False, fr1, hw1, out=dout, accumulate=True)
            tape.give(x, dx)
        tape.add(bwd)
        return out

    def _motion(self, name, x, geo, tape, record, seeds):
        """VanillaTemporalModule (motion_module.py:80-85,137-161,213-225,274-345).
        record: dict collecting the fused q|k|v buffer of hooked attentions (MySelfAttnProcessor.record_qkv);
        seeds: {attention name: (ref_idx u8, ref_val f32, coef)} guidance seeds for the backward."""
        w, cfg = self.w, self.cfg
        heads = cfg["motion_heads"]
        C = x.shape[1]
        d = C // heads
        fr, hw, T = geo.frames, geo.hw, geo.T
        p = name + ".temporal_transformer."
        b = p + "transformer_blocks.0."
        gN, bN = w.vec(p + "norm.weight"), w.vec(p + "norm.bias")
        pe = w.pe(C)[:geo.F].contiguous()
        h, st = self._gn_gemm(x, p + "norm.weight", p + "norm.bias", p + "proj_in.weight", p + "proj_in.bias", fr, hw)
        saved = []
        n_attn = 2 if (b + "attention_blocks.1.to_q.weight") in w.sd else 1   # SparseCtrl modules have one
        for a in range(n_attn):
            ap = b + "attention_blocks.%d." % a
            aname = ap[:-1]
            qkv, ls = self._ln_gemm(h, b + "norms.%d.weight" % a, b + "norms.%d.bias" % a,
                                    w.cat_lin([ap + "to_q.weight", ap + "to_k.weight", ap + "to_v.weight"]), tape, pe=pe, hw=hw)
            if record is not None and self._hooked(aname):   # the guidance read-out sees the differentiated batch element only
                rg, rcut = self._bslice(geo, tape.grad_batch if tape is not None else None)
                record[aname] = dict(qkv=rcut(qkv), C=C, heads=heads, d=d, geo=rg)
            o = ops.tattn_fwd(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], geo.B, geo.F, hw, heads, d)
            hnext = ops.gemm(o, w.lin(ap + "to_out.0.weight"), bias=w.vec(ap + "to_out.0.bias").unsqueeze(0), residual=h)
            saved.append((h, ls, qkv, aname, ap))
            h = hnext
        h2 = h
        gg, bff1, lsf = self._ff_geglu(h2, b + "ff_norm.weight", b + "ff_norm.bias", b, geo, tape)
        h3 = ops.gemm(gg, w.lin(b + "ff.net.2.weight"), bias=w.vec(b + "ff.net.2.bias").unsqueeze(0), residual=h2)
        del gg
        out = ops.gemm(h3, w.lin(p + "proj_out.weight"), bias=w.vec(p + "proj_out.bias").unsqueeze(0), residual=x)
        if tape is None:
            return out

        g1, cut = self._bslice(geo, tape.grad_batch)
        bx, bst, bh2, blsf = cut(x), cut(st), cut(h2), cut(lsf)
        bsaved = [(cut(hin), cut(ls), cut(qkv), aname, ap) for (hin, ls, qkv, aname, ap) in saved]
        fr1, hw1, T1 = g1.frames, g1.hw, g1.T

        def bwd():
            dout = tape.take(out)
            has_seed = seeds is not None and any(s[3] in seeds for s in bsaved)
            if dout is None and not has_seed:
                return
            dh = None
            if dout is not None:
                dh3 = ops.gemm(dout, w.lin_t(p + "proj_out.weight"))
                dg = ops.gemm(dh3, w.lin_t(b + "ff.net.2.weight"))
                dff1 = ops.geglu_bwd(dg, bff1)
                dn = ops.gemm(dff1, w.lin_t(b + "ff.net.0.proj.weight"))
                dh = ops.layernorm_bwd(dn, bh2, blsf, w.vec(b + "ff_norm.weight"), add=dh3)
            for a in reversed(range(n_attn)):
                hin, ls, qkv, aname, ap = bsaved[a]
                seed = seeds.get(aname) if seeds is not None and self._hooked(aname) else None
                if dh is None and seed is None:
                    continue
                da = ops.gemm(dh, w.lin_t(ap + "to_out.0.weight")) if dh is not None else None
                dqkv = ops.empty((T1, 3 * C), x)
                ops.tattn_bwd(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], da, dqkv[:, :C], dqkv[:, C:2 * C],
                              dqkv[:, 2 * C:], g1.B, g1.F, hw1, heads, d,
                              ref_idx=seed[0] if seed else None, ref_val=seed[1] if seed else None,
                              seed_coef=seed[2] if seed else 0.0)
                dn = ops.gemm(dqkv, w.cat_lin_t([ap + "to_q.weight", ap + "to_k.weight", ap + "to_v.weight"]))
                dh = ops.layernorm_bwd(dn, hin, ls, w.vec(b + "norms.%d.weight" % a), add=dh)
            dhn = ops.gemm(dh, w.lin_t(p + "proj_in.weight"))
            if dout is not None:
                dx = ops.gn_bwd(bx, None, dhn, bst, gN, bN, False, fr1, hw1, out=dout, accumulate=True)
            else:
                dx = ops.gn_bwd(bx, None, dhn, bst, gN, bN, False, fr1, hw1)
            tape.give(x, dx)
        tape.add(bwd)
        return out

    def _downsample(self, p, x, geo, tape):
        w = self.w
        g2 = geo.down()
        out = ops.gemm(x, w.conv(p + "weight"), bias=w.vec(p + "bias").unsqueeze(0), mode=CONV_S2,
                       geom=(geo.H, geo.W, g2.H, g2.W), m_out=g2.T)
        if tape is not None:
            gb, _ = self._bslice(geo, tape.grad_batch)

            def bwd():
                dout = tape.take(out)
                if dout is None:
                    return
                dx = ops.gemm(dout, w.conv_dgrad(p + "weight", stride=2), mode=TCONV_S2,
                              geom=(g2.H, g2.W, geo.H, geo.W), m_out=gb.T)
                tape.give(x, dx)
            tape.add(bwd)
        return out, g2

    def _upsample(self, p, x, geo, tape):
        w = self.w
        g2 = geo.up()
        out = ops.gemm(x, w.conv(p + "weight"), bias=w.vec(p + "bias").unsqueeze(0), mode=CONV_UP,
                       geom=(geo.H, geo.W, g2.H, g2.W), m_out=g2.T)
        if tape is not None:
            gb, _ = self._bslice(geo, tape.grad_batch)
            gb2 = gb.up()

            def bwd():
                dout = tape.take(out)
                if dout is None:
                    return
                du = ops.gemm(dout, w.conv_dgrad(p + "weight"), mode=CONV_S1, geom=(g2.H, g2.W, g2.H, g2.W), m_out=gb2.T)
                tape.give(x, ops.sumpool2(du, gb.frames, gb.H, gb.W))
            tape.add(bwd)
        return out, g2

    # ---- whole network --------------------------------------------------------------------------------
    @ops.scoped
    def forward(self, latents, t, text, tape=None, record=None, seeds=None, only_motion_feature=False,
                down_residuals=None, mid_residual=None, dup=False):
        """latents [B, 4, F, H, W] fp16, text [B, n_text, xdim] fp16 -> eps as a token matrix [(b f y x), 4].
        With `tape`, the blocks up to up_blocks[guidance_block] record their backward (motionclone_functions.py
        :601-625); later blocks never do (the reference runs them under no_grad, :629-652).

        dup=True (round 5): latents [V, 4, F, H, W] stand for the batch [x_1 .. x_V | x_1 .. x_V] with text [2 V, ...] =
        [u_1 .. u_V | c_1 .. c_V] - what single_step_video feeds the UNet (`noisy.expand(2, ...)` at :248-253; the two B = 1
        calls on equal latents at :216-223).  The two halves differ in their text only, and the text enters at the first
        cross-attention: conv_in, the first ResnetBlock3D and the first transformer's self-attention part give the same
        values for both halves, so they run ONCE on the V elements (_spatial: shared_geo); from the junction on the batch is
        2 V.  Same values as the duplicated batch up to the fp32 summation order of a GEMM whose tile choice follows the row
        count (tests/test_engine_parity.py)."""
        cfg, w = self.cfg, self.w
        assert latents.dtype == torch.float16 and text.dtype == torch.float16
        B, CL, F, H, W = latents.shape
        L = cfg["layers_per_block"]
        geoA = None                  # geometry of the shared prefix (dup): V batch elements
        if dup:
            if text.shape[0] != 2 * B:
                raise ValueError("forward(dup=True): %d latents need text [%d, ...], got %s" % (B, 2 * B, tuple(text.shape)))
            if cfg["down_has_attn"][0]:
                geoA = Geo(B, F, H, W)
            else:                    # no cross-attention in the first block: nothing to share the way _spatial does
                latents = torch.cat([latents, latents], 0)
            B = 2 * B
        geo = Geo(B, F, H, W)
        n_text = text.shape[1]
        text2d = text.reshape(B * n_text, text.shape[2]).contiguous()
        tb_all = self._time_bias(t, B, latents)
        gb = self.guidance_block
        tapeA = tape.whole_batch() if (geoA is not None and tape is not None) else tape

        def hook(nm):   # classify_blocks (util.py:434-440) matches the FULL attention-module names by substring: a motion
            # module is hooked when any of its attentions is (entries longer than the module prefix included)
            return any(self._hooked(nm + ".temporal_transformer.transformer_blocks.0.attention_blocks.%d" % a) for a in (0, 1))

        geo_in = geoA if geoA is not None else geo
        x_in = ops.latent_to_cl(latents, CIN_PAD)
        x = ops.gemm(x_in, w.conv("conv_in.weight", CIN_PAD), bias=w.vec("conv_in.bias").unsqueeze(0), mode=CONV_S1,
                     geom=(H, W, H, W), m_out=geo_in.T)
        if tape is not None:
            x0 = x

            def bwd_in():
                dout = tape.take(x0)
                if dout is None:
                    return
                b0, b1 = self._brange(geo_in, tapeA.grad_batch)
                nb = b1 - b0
                dxin = ops.gemm(dout, w.conv_dgrad("conv_in.weight", pad_cin=CIN_PAD), mode=CONV_S1,
                                geom=(H, W, H, W), m_out=nb * F * H * W)
                tape.latent_grad = ops.cl_to_latent(dxin, nb, CL, F, H, W, scale=1.0 / tape.grad_scale, f32=True)
            tape.add(bwd_in)
        if geoA is not None:
            # the skip copy of conv_in's output serves both halves of the batch; gradients (all of the differentiated,
            # i.e. shared, rows) pass straight through to the one copy
            xs = ops.dup_rows(x)
            if tape is not None:
                tape.add(lambda xs=xs, x0=x: (lambda g: tape.give(x0, g) if g is not None else None)(tape.take(xs)))
            skips = [(xs, geo)]
        else:
            skips = [(x, geo)]
        for i in range(4):
            for j in range(L):
                if geoA is not None and i == 0 and j == 0:      # the shared prefix (see the docstring)
                    x = self._resnet("down_blocks.0.resnets.0.", x, None, tb_all, geoA, tapeA)
                    x = self._spatial("down_blocks.0.attentions.0.", x, text2d, n_text, geo, tape, shared_geo=geoA)
                else:
                    x = self._resnet("down_blocks.%d.resnets.%d." % (i, j), x, None, tb_all, geo, tape)
                    if cfg["down_has_attn"][i]:
                        x = self._spatial("down_blocks.%d.attentions.%d." % (i, j), x, text2d, n_text, geo, tape)
                nm = "down_blocks.%d.motion_modules.%d" % (i, j)
                x = self._motion(nm, x, geo, tape, record if hook(nm) else None, seeds if hook(nm) else None)
                skips.append((x, geo))
            if i < 3:
                x, geo = self._downsample("down_blocks.%d.downsamplers.0.conv." % i, x, geo, tape)
                skips.append((x, geo))
        if down_residuals is not None:
            # SparseCtrl: res_samples[i] += residual[i] on the skip copies only (motionclone_functions.py:581-587);
            # the residuals are constants, so the gradient of the sum goes to the skip tensor unchanged
            assert len(down_residuals) == len(skips)
            for i, r in enumerate(down_residuals):
                sk, sg = skips[i]
                summed = ops.add(sk, r)
                skips[i] = (summed, sg)
                if tape is not None:
                    tape.add(lambda sk=sk, summed=summed: (lambda g: tape.give(sk, g) if g is not None else None)(tape.take(summed)))
        x = self._resnet("mid_block.resnets.0.", x, None, tb_all, geo, tape)
        x = self._spatial("mid_block.attentions.0.", x, text2d, n_text, geo, tape)
        x = self._resnet("mid_block.resnets.1.", x, None, tb_all, geo, tape)
        if mid_residual is not None:   # :595-598
            xm = x
            x = ops.add(xm, mid_residual)
            if tape is not None:
                tape.add(lambda xm=xm, xs=x: (lambda g: tape.give(xm, g) if g is not None else None)(tape.take(xs)))
        for i in range(4):
            in_graph = i <= gb
            if not in_graph and only_motion_feature:
                return None
            tp = tape if in_graph else None
            for j in range(L + 1):
                skip, sgeo = skips.pop()
                assert sgeo.T == geo.T
                x = self._resnet("up_blocks.%d.resnets.%d." % (i, j), x, skip, tb_all, geo, tp)
                if cfg["up_has_attn"][i]:
                    x = self._spatial("up_blocks.%d.attentions.%d." % (i, j), x, text2d, n_text, geo, tp)
                nm = "up_blocks.%d.motion_modules.%d" % (i, j)
                is_hooked = in_graph and hook(nm)
                x = self._motion(nm, x, geo, tp, record if is_hooked else None, seeds if is_hooked else None)
            if i < 3:
                # the upsampler of block i feeds block i+1: in the graph only while i+1 <= guidance block
                x, geo = self._upsample("up_blocks.%d.upsamplers.0.conv." % i, x, geo, tape if i < gb else None)
        hn, st = ops.gn_fwd(x, None, w.vec("conv_norm_out.weight"), w.vec("conv_norm_out.bias"), True, geo.frames, geo.hw, cfg["norm_eps"])
        eps = ops.gemm(hn, w.conv("conv_out.weight"), bias=w.vec("conv_out.bias").unsqueeze(0), mode=CONV_S1,
                       geom=(geo.H, geo.W, geo.H, geo.W), m_out=geo.T)
        return eps

    # ---- guidance layer -----------------------------------------------------------------------------------
    def _hooked(self, attention_name):
        return any(blk in attention_name for blk in self.guidance_blocks)

    def hooked_names(self):
        """names of the hooked temporal attentions in module order (= the keys of the reference's .pt dict)"""
        L = self.cfg["layers_per_block"]
        mods = ["down_blocks.%d.motion_modules.%d" % (i, j) for i in range(4) for j in range(L)]
        mods += ["up_blocks.%d.motion_modules.%d" % (i, j) for i in range(self.guidance_block + 1) for j in range(L + 1)]
        names = [m + ".temporal_transformer.transformer_blocks.0.attention_blocks.%d" % a for m in mods for a in range(2)]
        return [n for n in names if any(blk in n for blk in self.guidance_blocks)]

    @ops.scoped
    def extract_representation(self, noisy_latents, t, uncond_text, down_residuals=None, mid_residual=None):
        """model part of obtain_motion_representation (motionclone_functions.py:74-79): partial forward to the
        guidance block, P = softmax(scale q k^T) of the hooked temporal attentions, top-1 value/index."""
        record = {}
        self.forward(noisy_latents, t, uncond_text, record=record, only_motion_feature=True,
                     down_residuals=down_residuals, mid_residual=mid_residual)
        rep = {}
        for name in self.hooked_names():
            r = record[name]
            C, g = r["C"], r["geo"]
            val, idx = ops.tattn_top1(r["qkv"][:, :C], r["qkv"][:, C:2 * C], g.B, g.F, g.hw, r["heads"], r["d"])
            rep[name] = [val, idx]
        return rep

    def prepare_representation(self, rep):
        """reference .pt dict {name: [values [BN, heads, F, 1], indices uint8]} -> device tensors for the kernels.
        A LIST of such dicts (V videos batched in one forward, guided_eps_and_grad) is concatenated along the (b, pixel) axis
        in video order - the order of the conditional halves [c_1 .. c_V] in the batch."""
        if isinstance(rep, (list, tuple)):
            parts = [self.prepare_representation(r) for r in rep]
            if len(parts) == 1:
                return parts[0]
            return {name: (torch.cat([p[name][0] for p in parts], 0).contiguous(), torch.cat([p[name][1] for p in parts], 0).contiguous())
                    for name in parts[0]}
        out = {}
        for name, (val, idx) in rep.items():
            out[name] = (idx.to(self.dev, torch.uint8).contiguous(), val.to(self.dev, torch.float32).contiguous())
        return out

    @ops.scoped
    def guided_eps_and_grad(self, latents, t, text_cond, rep_dev, weight, want_loss=False, down_residuals=None,
                            mid_residual=None, text_uncond=None):
        """eps_c forward with the in-graph half taped + backward of  weight * sum_m mse_m  w.r.t. the latent
        (motionclone_functions.py:221-236).  With `text_uncond` the un-guided eps_u of :216-219 is produced by the same
        launch sequence: one B = 2 forward over [uncond | cond] whose tape differentiates batch element 1 only
        (residuals, if any, are then the B = 2 SparseCtrl outputs).

        V > 1 videos at once (latents [V, 4, F, H, W], text_* [V, n, dim], `rep_dev` = prepare_representation of the list of
        their representations; needs text_uncond): ONE B = 2 V forward over [u_1 .. u_V | c_1 .. c_V], the tape differentiates
        the conditional halves; every video's loss is its own mean (the seed coefficient uses the per-video element count), so
        each video gets exactly the gradient of its separate call.
        Returns (eps_c tokens, grad fp32 [V,4,F,H,W], loss (summed over the videos) or None[, eps_u tokens])."""
        batched = text_uncond is not None
        V = latents.shape[0]
        if V > 1 and not batched:
            raise ValueError("several videos per call need text_uncond (one [u_1 .. u_V | c_1 .. c_V] forward)")
        tape = Tape(grad_batch=(V, 2 * V) if batched else None)
        # The loss is a MEAN over the attention maps (F.mse_loss, :229), so the gradient per element shrinks with the size of
        # the problem: 4e-5 at the latent for config 2, 6e-6 for config 5 (32 f x 96^2), where `grad_scale` = 1024 left the deep
        # layers' fp16 gradient activations in the subnormal range (gradient 1.8e-2 from the fp32 oracle against 7e-3 at every
        # smaller size, tests/test_fullsize_parity.py).  The scale therefore follows the map size in powers of two from the
        # validated point (config 2: 32768 elements per hooked attention): exact to undo, same dynamic range at every size.
        numel_max = max((idx.numel() // V for idx, _ in rep_dev.values()), default=1)
        tape.grad_scale = self.grad_scale * float(2 ** max(0, round(math.log2(max(1.0, numel_max / 32768.0)))))
        seeds = {}
        for name, (idx, val) in rep_dev.items():
            numel = idx.numel() // V              # per video: F.mse_loss averages over ONE video's map
            seeds[name] = (idx, val, tape.grad_scale * float(weight) * 2.0 / numel)
        record = {}
        if batched:
            text2 = torch.cat([text_uncond, text_cond], 0)
            if self.share_prefix:    # [u_1 .. u_V | c_1 .. c_V] on the same latents: what does not see the text runs once
                eps2 = self.forward(latents, t, text2, tape=tape, record=record, seeds=seeds, down_residuals=down_residuals,
                                    mid_residual=mid_residual, dup=True)
            else:
                lat2 = latents.expand(2, -1, -1, -1, -1) if V == 1 else torch.cat([latents, latents], 0)
                eps2 = self.forward(lat2, t, text2, tape=tape, record=record, seeds=seeds, down_residuals=down_residuals,
                                    mid_residual=mid_residual)
            T1 = eps2.shape[0] // 2
            eps_u, eps_c = eps2[:T1], eps2[T1:]
        else:
            eps_c = self.forward(latents, t, text_cond, tape=tape, record=record, seeds=seeds,
                                 down_residuals=down_residuals, mid_residual=mid_residual)
        loss = None
        if want_loss:
            total = None
            for name, (idx, val) in rep_dev.items():
                r = record[name]
                C, g = r["C"], r["geo"]
                lm = ops.tattn_loss(r["qkv"][:, :C], r["qkv"][:, C:2 * C], idx, val, g.B, g.F, g.hw, r["heads"], r["d"])
                total = lm if total is None else total + lm
            loss = total * (float(weight) * V)   # tattn_loss is the mean over all V maps; the sum of the V means is V x that
        tape.latent_grad = None
        tape.run()
        grad = tape.latent_grad
        assert grad is not None, "guidance gradient did not reach the latent"
        if batched:
            return eps_c, grad, loss, eps_u
        return eps_c, grad, loss


class ControlNetEngine(UNet3DEngine):
    """SparseControlNetModel.forward (reference motionclone/models/sparse_controlnet.py:450-587) for the
    configuration of configs/sparsectrl/latent_condition.yaml (BASELINE config 4: i2v_rgb): the noisy input is
    replaced by zeros (= a broadcast of conv_in.bias, :516-518), the condition is VAE latent + mask through one
    3x3 conv (:176-184, 522-527), followed by the down blocks (motion modules with a single temporal attention),
    the mid block and 12 + 1 output 1x1 convs scaled by `conditioning_scale` (:557-574).  Always inference-only
    (motionclone_functions.py:177)."""

    def _resnet_names(self):
        L = self.cfg["layers_per_block"]
        names = []
        for i in range(4):
            names += ["down_blocks.%d.resnets.%d." % (i, j) for j in range(L)]
        return names + ["mid_block.resnets.0.", "mid_block.resnets.1."]

    def _cond_embedding(self, cond, mask, F, H, W):
        """controlnet_cond_embedding(cat(cond, mask)) + conv_in.bias (:516-527) as a token matrix [(f y x), C0].
        latent_condition.yaml: one 3x3 conv on VAE latent + mask at latent resolution.  image_condition.yaml (scribble /
        sketch): SparseControlNetConditioningEmbedding (:49-82) on pixels + mask at 8x the latent resolution - conv_in,
        then (same-width, stride-2 widening) conv pairs down to the latent grid, SiLU after each, conv_out; every conv an
        implicit GEMM on 64-padded channels-last rows (padding channels stay zero through SiLU).  Recomputed at every call,
        as in the reference."""
        w = self.w
        cm = torch.cat([cond, mask], dim=1).to(torch.float16)
        p = "controlnet_cond_embedding."
        cin_b = w.vec("conv_in.bias")
        if (p + "weight") in w.sd:
            if tuple(cm.shape[2:]) != (F, H, W):
                raise ValueError("latent condition %s does not match the sample grid %s" % (tuple(cm.shape), (F, H, W)))
            bias = (w.vec(p + "bias") + cin_b).unsqueeze(0).contiguous()
            e = ops.gemm(ops.latent_to_cl(cm, CIN_PAD), w.conv(p + "weight", CIN_PAD), bias=bias, mode=CONV_S1,
                         geom=(H, W, H, W), m_out=F * H * W)
        else:
            Hs, Ws = cm.shape[3], cm.shape[4]
            nblk = 0
            while (p + "blocks.%d.weight" % nblk) in w.sd:
                nblk += 1
            if cm.shape[2] != F or (Hs >> (nblk // 2), Ws >> (nblk // 2)) != (H, W) or Hs % (1 << (nblk // 2)):
                raise ValueError("pixel condition %s does not reduce to the sample grid %s" % (tuple(cm.shape), (F, H, W)))
            x = ops.latent_to_cl(cm, 64)
            names = [("conv_in.", 1)] + [("blocks.%d." % i, 2 if i % 2 else 1) for i in range(nblk)] + [("conv_out.", 1)]
            for name, stride in names:
                last = name == "conv_out."
                cout = w.sd[p + name + "weight"].shape[0]
                Ho, Wo = (Hs, Ws) if stride == 1 else (Hs // 2, Ws // 2)
                rows, cpad = F * Ho * Wo, (cout + 63) // 64 * 64
                buf = (torch.zeros if cpad != cout else torch.empty)((rows, cpad), dtype=torch.float16, device=self.dev)
                b = w.vec(p + name + "bias")
                ops.gemm(x, w.conv(p + name + "weight", pad_cin=x.shape[1]),
                         bias=((b + cin_b) if last else b).unsqueeze(0).contiguous(),
                         mode=CONV_S1 if stride == 1 else CONV_S2, geom=(Hs, Ws, Ho, Wo), m_out=rows, out=buf[:, :cout])
                x = buf if last else ops.silu(buf)
                Hs, Ws = Ho, Wo
            e = x
        return e

    @ops.scoped
    def forward(self, sample_shape, t, text, cond, mask, conditioning_scale=1.0):
        """sample_shape = (B, 4, F, H, W); cond [1, Cc, F, H, W] latent condition (or [1, 3, F, 8H, 8W] pixels for the
        scribble embedding; zeros on unconditioned frames), mask [1, 1, F, ...] at the condition's resolution;
        text [B, n, dim] -> (list of 12 residual token matrices [(b f y x), C], mid residual)"""
        cfg, w = self.cfg, self.w
        B, _, F, H, W = sample_shape
        L = cfg["layers_per_block"]
        geo = Geo(B, F, H, W)
        n_text = text.shape[1]
        text2d = text.reshape(B * n_text, text.shape[2]).contiguous()
        tb_all = self._time_bias(t, B, text)
        e = self._cond_embedding(cond, mask, F, H, W)
        # the same condition for every batch element: the batch elements differ in their text only, so (as in
        # UNet3DEngine.forward(dup=True)) what precedes the first cross-attention runs once for a [u | c] batch
        share = self.share_prefix and B == 2 and cfg["down_has_attn"][0]
        geoA = Geo(1, F, H, W) if share else None
        feats = [torch.cat([e] * B, dim=0) if B > 1 else e]
        x = e if share else feats[0]
        for i in range(4):
            for j in range(L):
                if share and i == 0 and j == 0:
                    x = self._resnet("down_blocks.0.resnets.0.", x, None, tb_all, geoA, None)
                    x = self._spatial("down_blocks.0.attentions.0.", x, text2d, n_text, geo, None, shared_geo=geoA)
                else:
                    x = self._resnet("down_blocks.%d.resnets.%d." % (i, j), x, None, tb_all, geo, None)
                    if cfg["down_has_attn"][i]:
                        x = self._spatial("down_blocks.%d.attentions.%d." % (i, j), x, text2d, n_text, geo, None)
                x = self._motion("down_blocks.%d.motion_modules.%d" % (i, j), x, geo, None, None, None)
                feats.append(x)
            if i < 3:
                x, geo = self._downsample("down_blocks.%d.downsamplers.0.conv." % i, x, geo, None)
                feats.append(x)
        x = self._resnet("mid_block.resnets.0.", x, None, tb_all, geo, None)
        x = self._spatial("mid_block.attentions.0.", x, text2d, n_text, geo, None)
        x = self._resnet("mid_block.resnets.1.", x, None, tb_all, geo, None)
        sc = float(conditioning_scale)
        down = [ops.gemm(f, w.lin("controlnet_down_blocks.%d.weight" % i),
                         bias=(w.vec("controlnet_down_blocks.%d.bias" % i) * sc).unsqueeze(0).contiguous(), alpha=sc)
                for i, f in enumerate(feats)]
        mid = ops.gemm(x, w.lin("controlnet_mid_block.weight"),
                       bias=(w.vec("controlnet_mid_block.bias") * sc).unsqueeze(0).contiguous(), alpha=sc)
        return down, mid


def split_residuals(down, mid, b, B):
    """rows of batch element b out of B from every residual (the reference's tensor[[b], ...], :205-208)"""
    def rows(t):
        n = t.shape[0] // B
        return t[b * n:(b + 1) * n]
    return [rows(d) for d in down], rows(mid)
