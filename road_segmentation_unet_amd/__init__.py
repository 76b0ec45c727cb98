"""MI355X-native U-Net forward/backward path of aschneuw/road-segmentation-unet (hand-written HIP kernels behind the
C ABI of include/rsu.h). Host mirrors of the reference interface: unet.input_size_needed / unet.forward,
model.ConvolutionalModel, images (tiler)."""
from . import unet  # noqa: F401
from .unet import UNet, forward, input_size_needed  # noqa: F401
