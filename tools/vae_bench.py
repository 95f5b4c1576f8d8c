"""VAE decode timing on cuda:0: decode_latents of one config-2 video (16 latent frames 64x64 -> 16 x 512x512), SD-1.5
AutoencoderKL architecture with synthetic weights.  Prints one JSON line (frames/s, TFLOP/s from the conv/GEMM MACs)."""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from motionclone_amd import ops, spec  # noqa: E402
from motionclone_amd.models.vae import vae_param_shapes  # noqa: E402
from motionclone_amd.vae_engine import SD15_VAE_CONFIG, VaeDecoderEngine  # noqa: E402


def decoder_flops(cfg, h, w):
    """2 * MACs of one frame: convs, 1x1 shortcuts, the mid-block attention"""
    shapes = vae_param_shapes(cfg)
    ch = cfg["block_out_channels"]
    fl = 0.0
    res = {}
    H, W = h, w
    res["decoder.conv_in"] = res["decoder.mid_block"] = (H, W)
    for i in range(len(ch)):
        res["decoder.up_blocks.%d.resnets" % i] = (H, W)
        if i != len(ch) - 1:
            H, W = 2 * H, 2 * W
            res["decoder.up_blocks.%d.upsamplers" % i] = (H, W)
    res["decoder.conv_out"] = (H, W)
    for name, s in shapes.items():
        if not name.startswith("decoder.") or not name.endswith("weight") or len(s) < 2:
            continue
        key = max((k for k in res if name.startswith(k)), key=len)
        hh, ww = res[key]
        macs = hh * ww
        for d in s:
            macs *= d
        fl += 2.0 * macs
    C = ch[-1]
    fl += 2.0 * 2 * (h * w) ** 2 * C   # q k^T and P v
    return fl


def main():
    dev = torch.device("cuda:0")
    F, h, w = 16, 64, 64
    cfg = dict(SD15_VAE_CONFIG)
    g = torch.Generator().manual_seed(4242)
    sd = {}
    for name, shape in vae_param_shapes(cfg).items():
        if name.startswith("encoder") or name.startswith("quant"):
            continue
        fan = 1
        for d in shape[1:]:
            fan *= d
        sd[name] = (1.0 + 0.1 * torch.randn(shape, generator=g)) if ("norm" in name and name.endswith("weight")) else \
            (0.02 * torch.randn(shape, generator=g) if name.endswith("bias") else (torch.rand(shape, generator=g) * 2 - 1) / fan ** 0.5)
    eng = VaeDecoderEngine(sd, cfg, dev)
    lat = (0.18215 * torch.randn(1, 4, F, h, w, generator=torch.Generator().manual_seed(11))).half().to(dev)
    for _ in range(2):
        eng.decode_video(lat)
    torch.cuda.synchronize()
    t0 = time.time()
    K = 5
    for _ in range(K):
        v = eng.decode_video(lat)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / K
    fl = decoder_flops(cfg, h, w) * F
    print(json.dumps(dict(metric="VAE decode_latents, 16 frames 64x64 -> 512x512 (SD-1.5 AutoencoderKL arch)",
                          sec_per_video=dt, frames_per_s=F / dt, tflop_per_video=fl / 1e12, tflops=fl / dt / 1e12,
                          chunk_frames=eng.chunk_frames(h, w), out_shape=list(v.shape))))


if __name__ == "__main__":
    main()
