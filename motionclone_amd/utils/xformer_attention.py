"""Q/K capture hook and its installer (reference motionclone/utils/xformer_attention.py:17-52)."""
from .util import classify_blocks


class MySelfAttnProcessor:
    """Holds references to the hooked attention's query / key.  With the HIP engine both point at the fused
    q|k|v token-matrix record of that module (dict: qkv, C, heads, d, geo) rather than to reshaped copies."""

    def __init__(self, attention_op=None):
        self.attention_op = attention_op
        self.query = None
        self.key = None

    def __call__(self, attn, hidden_states, query, key, value, attention_mask):
        self.key = key
        self.query = query

    def record_qkv(self, attn, hidden_states, query, key, value, attention_mask):
        self.key = key
        self.query = query

    def record_attn_mask(self, attn, hidden_states, query, key, value, attention_mask):
        self.attn = attn
        self.attention_mask = attention_mask


def prep_unet_attention(unet, motion_gudiance_blocks):
    """install a processor on every temporal attention whose name contains a guidance block (reference :45-52)"""
    for name, module in unet.named_modules():
        if "VersatileAttention" in type(module).__name__ and classify_blocks(motion_gudiance_blocks, name):
            module.set_processor(MySelfAttnProcessor())
    return unet
