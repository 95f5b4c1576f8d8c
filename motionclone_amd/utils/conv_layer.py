"""reference motionclone/utils/conv_layer.py: prep_unet_conv swaps ResnetBlock3D.forward of the up blocks for a
numerically identical closure that records a hidden state nobody on this path reads.  The HIP
engine computes the same ResnetBlock3D; the installer is kept so entry scripts run unchanged."""


def prep_unet_conv(unet):
    for i in range(len(unet.up_blocks)):
        for j in range(len(unet.up_blocks[i].resnets)):
            unet.up_blocks[i].resnets[j].record_hidden_state = None
    return unet
