"""Drop-in text encoder for `AnimationPipeline._encode_prompt` (reference pipeline_animation.py:160-247): the surface of
`transformers.CLIPTextModel` the reference touches - `from_pretrained(path, subfolder="text_encoder")`, `.config`
(`max_position_embeddings`, and `use_attention_mask` which SD-1.5's config does not set), `.dtype`, `.device`, `.to()`,
HF state-dict keys, and `model(input_ids, attention_mask=None)[0]` = last_hidden_state [B, 77, 768] - on the HIP engine."""
import json
import os

import torch
from torch import nn

from ..clip_engine import SD15_CLIP_CONFIG, ClipTextEngine
from .unet import FrozenConfig, ParamNode


def clip_param_shapes(cfg):
    C, I, L = cfg["hidden_size"], cfg["intermediate_size"], cfg["num_hidden_layers"]
    s = {"text_model.embeddings.token_embedding.weight": (cfg["vocab_size"], C),
         "text_model.embeddings.position_embedding.weight": (cfg["max_position_embeddings"], C)}
    for i in range(L):
        p = "text_model.encoder.layers.%d." % i
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s[p + "self_attn.%s.weight" % n] = (C, C)
            s[p + "self_attn.%s.bias" % n] = (C,)
        for n in ("layer_norm1", "layer_norm2"):
            s[p + n + ".weight"] = (C,)
            s[p + n + ".bias"] = (C,)
        s[p + "mlp.fc1.weight"] = (I, C)
        s[p + "mlp.fc1.bias"] = (I,)
        s[p + "mlp.fc2.weight"] = (C, I)
        s[p + "mlp.fc2.bias"] = (C,)
    s["text_model.final_layer_norm.weight"] = (C,)
    s["text_model.final_layer_norm.bias"] = (C,)
    return s


class CLIPTextModel(nn.Module):
    def __init__(self, **config):
        super().__init__()
        cfg = dict(SD15_CLIP_CONFIG)
        cfg.update({k: v for k, v in config.items() if k in cfg})
        self.engine_config = cfg
        self.config = FrozenConfig(**cfg)
        for name, shape in clip_param_shapes(cfg).items():
            node = self
            parts = name.split(".")
            for part in parts[:-1]:
                if part not in node._modules:
                    node.add_module(part, ParamNode())
                node = node._modules[part]
            node.register_parameter(parts[-1], nn.Parameter(torch.empty(shape, dtype=torch.float16), requires_grad=False))
        self._engine = None
        self._engine_key = None

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    @classmethod
    def from_pretrained(cls, pretrained_model_path, subfolder=None, **kw):
        if subfolder is not None:
            pretrained_model_path = os.path.join(pretrained_model_path, subfolder)
        with open(os.path.join(pretrained_model_path, "config.json"), "r") as f:
            config = json.load(f)
        config = config.get("text_config", config)
        model = cls(**config)
        st = os.path.join(pretrained_model_path, "model.safetensors")
        from .. import checkpoints
        sd = checkpoints.read(st if os.path.isfile(st) else os.path.join(pretrained_model_path, "pytorch_model.bin"))
        model.load_state_dict(sd)
        return model

    def load_state_dict(self, state_dict, strict=True, **kw):
        # buffers of older transformers checkpoints that are not parameters
        state_dict = {k: v for k, v in state_dict.items() if not k.endswith("position_ids")}
        if state_dict and not any(k.startswith("text_model.") for k in state_dict):
            # transformers >= 5 dropped the `text_model.` level that 4.28.1 (the reference's pin) and SD checkpoints carry
            state_dict = {"text_model." + k: v for k, v in state_dict.items()}
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self._engine = None
        return out

    def engine(self):
        p0 = next(self.parameters())
        key = (p0.device, p0.data_ptr())
        if self._engine is None or self._engine_key != key:
            self._engine = ClipTextEngine(dict(self.state_dict()), self.engine_config, p0.device)
            self._engine_key = key
        return self._engine

    def forward(self, input_ids, attention_mask=None, **unused):
        if attention_mask is not None:
            raise NotImplementedError("SD-1.5's text encoder config has no use_attention_mask; padding is attended as in the reference")
        hidden = self.engine().forward(input_ids)
        return (hidden,)
