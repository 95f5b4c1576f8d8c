"""CLIP text encoder on the HIP kernels: `self.text_encoder(input_ids)[0]` of
`AnimationPipeline._encode_prompt` (reference motionclone/pipelines/pipeline_animation.py:160-247).

The model is `transformers` `CLIPTextModel` (the reference loads it with `from_pretrained(..., subfolder="text_encoder")`,
t2v_video_sample.py:24; SD-1.5: 12 layers, width 768, 12 heads, 77 positions, quick_gelu, causal mask, final LayerNorm),
driven from a flat state-dict with the HF key names.  One prompt pair is a [154, 768] token matrix: twelve layers of
LayerNorm -> fused q|k|v GEMM -> causal flash attention -> out-proj + residual -> LayerNorm -> fc1 -> quick_gelu -> fc2 +
residual, all through the same kernels as the UNet (the query scaling 1/sqrt(d) of HF's attention is the softmax scale)."""
import torch

from . import ops
from .engine import Weights

SD15_CLIP_CONFIG = dict(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
                        num_attention_heads=12, max_position_embeddings=77, layer_norm_eps=1e-5, hidden_act="quick_gelu")


class ClipTextEngine:
    def __init__(self, state_dict, cfg=None, device="cuda"):
        self.cfg = dict(cfg or SD15_CLIP_CONFIG)
        if self.cfg.get("hidden_act", "quick_gelu") != "quick_gelu":
            raise NotImplementedError("only the quick_gelu CLIP text tower of SD-1.x is built")
        self.dev = torch.device(device)
        sd = {k[len("text_model."):] if k.startswith("text_model.") else k: v for k, v in state_dict.items()}
        self.w = Weights(sd, self.cfg, self.dev)
        C, H = self.cfg["hidden_size"], self.cfg["num_attention_heads"]
        assert C % H == 0 and C % 64 == 0 and (C // H) % 8 == 0 and self.cfg["intermediate_size"] % 64 == 0
        self.tok = self.w._h(sd["embeddings.token_embedding.weight"])
        self.pos = self.w._h(sd["embeddings.position_embedding.weight"])

    @ops.scoped
    def forward(self, input_ids):
        """input_ids int64 [B, S] -> last_hidden_state fp16 [B, S, C]"""
        cfg, w = self.cfg, self.w
        B, S = input_ids.shape
        C, H = cfg["hidden_size"], cfg["num_attention_heads"]
        d = C // H
        if S > cfg["max_position_embeddings"]:
            raise ValueError("sequence of %d tokens exceeds max_position_embeddings" % S)
        eps = cfg["layer_norm_eps"]
        x = ops.clip_embed(input_ids.to(self.dev).contiguous(), self.tok, self.pos)
        for i in range(cfg["num_hidden_layers"]):
            p = "encoder.layers.%d." % i
            n1, _ = ops.layernorm_fwd(x, w.vec(p + "layer_norm1.weight"), w.vec(p + "layer_norm1.bias"), eps=eps, save_stats=False)
            qkv = ops.gemm(n1, w.cat_lin([p + "self_attn.q_proj.weight", p + "self_attn.k_proj.weight", p + "self_attn.v_proj.weight"]),
                           bias=self._qkv_bias(p))
            a = ops.attn_fwd_causal(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], S, H, d, B)
            x = ops.gemm(a, w.lin(p + "self_attn.out_proj.weight"), bias=w.vec(p + "self_attn.out_proj.bias").unsqueeze(0),
                         residual=x)
            n2, _ = ops.layernorm_fwd(x, w.vec(p + "layer_norm2.weight"), w.vec(p + "layer_norm2.bias"), eps=eps, save_stats=False)
            h = ops.quick_gelu(ops.gemm(n2, w.lin(p + "mlp.fc1.weight"), bias=w.vec(p + "mlp.fc1.bias").unsqueeze(0)))
            x = ops.gemm(h, w.lin(p + "mlp.fc2.weight"), bias=w.vec(p + "mlp.fc2.bias").unsqueeze(0), residual=x)
        out, _ = ops.layernorm_fwd(x, w.vec("final_layer_norm.weight"), w.vec("final_layer_norm.bias"), eps=eps, save_stats=False)
        return out.reshape(B, S, C)

    def _qkv_bias(self, p):
        w = self.w
        return w._get(("qkv_b", p), lambda: torch.cat([w.vec(p + "self_attn.q_proj.bias"), w.vec(p + "self_attn.k_proj.bias"),
                                                       w.vec(p + "self_attn.v_proj.bias")]).unsqueeze(0).contiguous())
