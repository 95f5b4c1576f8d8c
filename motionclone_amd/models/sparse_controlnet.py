"""Drop-in `SparseControlNetModel` (reference motionclone/models/sparse_controlnet.py:85-587) backed by
`ControlNetEngine`: same constructor / `from_unet` / state-dict keys / `forward` arguments for the configurations of
configs/sparsectrl/latent_condition.yaml (i2v_rgb: VAE-latent condition, one-conv embedding) and
configs/sparsectrl/image_condition.yaml (i2v_sketch: pixel-space condition through SparseControlNetConditioningEmbedding).  `forward` returns the 12 down residuals and the mid residual as
channels-last token matrices, which `UNet3DConditionModel.forward` and the sampler consume directly."""
import torch
from torch import nn

from .. import spec
from ..engine import ControlNetEngine, default_config
from .unet import FrozenConfig, ParamNode, _build_tree


class SparseControlNetModel(nn.Module):
    def __init__(self, in_channels=4, conditioning_channels=3, block_out_channels=(320, 640, 1280, 1280),
                 layers_per_block=2, cross_attention_dim=1280, attention_head_dim=8, num_attention_heads=None,
                 norm_num_groups=32, norm_eps=1e-5, use_motion_module=True, motion_module_resolutions=(1, 2, 4, 8),
                 motion_module_mid_block=False, motion_module_type="Vanilla", motion_module_kwargs=None,
                 concate_conditioning_mask=True, use_simplified_condition_embedding=False,
                 set_noisy_sample_input_to_zero=False, conditioning_embedding_out_channels=(16, 32, 96, 256), **unused):
        super().__init__()
        mk = dict(motion_module_kwargs or {})
        if not (set_noisy_sample_input_to_zero and concate_conditioning_mask):
            raise NotImplementedError("both SparseCtrl configurations of the reference (configs/sparsectrl/*.yaml) zero the "
                                      "noisy input and concatenate the conditioning mask; other combinations are not built")
        if mk.get("attention_block_types", ["Temporal_Self"]) != ["Temporal_Self"]:
            raise NotImplementedError("SparseCtrl motion modules use a single Temporal_Self attention")
        heads = num_attention_heads or attention_head_dim
        heads = heads if isinstance(heads, int) else heads[0]
        self.engine_config = dict(default_config(), in_channels=in_channels, block_out_channels=tuple(block_out_channels),
                                  layers_per_block=layers_per_block, cross_attention_dim=cross_attention_dim,
                                  attention_heads=heads, norm_num_groups=norm_num_groups, norm_eps=norm_eps,
                                  motion_heads=mk.get("num_attention_heads", 8),
                                  motion_pe_max_len=mk.get("temporal_position_encoding_max_len", 32))
        self.config = FrozenConfig(in_channels=in_channels, conditioning_channels=conditioning_channels,
                                   block_out_channels=tuple(block_out_channels), global_pool_conditions=False)
        self.use_simplified_condition_embedding = bool(use_simplified_condition_embedding)
        self.set_noisy_sample_input_to_zero = True
        _build_tree(self, spec.controlnet_param_shapes(self.engine_config, conditioning_channels,
                                                       simplified=self.use_simplified_condition_embedding,
                                                       embedding_channels=tuple(conditioning_embedding_out_channels)),
                    self.engine_config, torch.float16)
        self._engine = None
        self._engine_key = None

    @classmethod
    def from_unet(cls, unet, controlnet_conditioning_channel_order="rgb", conditioning_embedding_out_channels=None,
                  load_weights_from_unet=True, controlnet_additional_kwargs=None):
        """reference :316-360: same widths as the UNet; conv_in / time embedding / down / mid weights copied over"""
        c = unet.config
        model = cls(in_channels=c.in_channels, block_out_channels=c.block_out_channels,
                    layers_per_block=c.layers_per_block, cross_attention_dim=c.cross_attention_dim,
                    attention_head_dim=c.attention_head_dim, norm_num_groups=c.norm_num_groups, norm_eps=c.norm_eps,
                    **dict(controlnet_additional_kwargs or {}))
        if load_weights_from_unet:
            own = model.state_dict()
            src = {k: v for k, v in unet.state_dict().items()
                   if k in own and own[k].shape == v.shape and "motion_modules." not in k}
            model.load_state_dict(src, strict=False)
        return model

    def load_state_dict(self, state_dict, strict=True, **kw):
        state_dict = {k: v for k, v in state_dict.items() if "pos_encoder.pe" not in k}
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self._engine = None
        return out

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    def engine(self):
        p0 = next(self.parameters())
        key = (p0.device, p0.data_ptr())
        if self._engine is None or self._engine_key != key:
            self._engine = ControlNetEngine(dict(self.state_dict()), self.engine_config, p0.device)
            self._engine_key = key
        return self._engine

    def forward(self, sample, timestep, encoder_hidden_states, controlnet_cond, conditioning_mask=None,
                conditioning_scale=1.0, class_labels=None, attention_mask=None, cross_attention_kwargs=None,
                guess_mode=False, return_dict=True):
        if guess_mode or class_labels is not None or attention_mask is not None:
            raise NotImplementedError("guess_mode / class_labels / attention_mask are never used on the MotionClone path")
        t = int(timestep) if not torch.is_tensor(timestep) else int(timestep.reshape(-1)[0].item())
        B = sample.shape[0]
        text = encoder_hidden_states.to(torch.float16)
        if text.shape[0] != B:
            text = text.repeat(B // text.shape[0], 1, 1)      # reference :489
        down, mid = self.engine().forward(tuple(sample.shape), t, text, controlnet_cond, conditioning_mask,
                                          conditioning_scale)
        return (down, mid)


ParamNode  # re-exported for isinstance checks
