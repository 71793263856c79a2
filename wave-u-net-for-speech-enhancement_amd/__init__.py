"""MI355X-native Wave-U-Net hot path (forward / backward / loss / data-parallel gradient exchange)
behind the plugin API of haoxiangsnr/Wave-U-Net-for-Speech-Enhancement.

The package directory name contains hyphens, so it is addressed the way the reference addresses
every plugin - by string through importlib (util/utils.py:55-72):

    importlib.import_module("wave-u-net-for-speech-enhancement_amd.model").Model()
"""
from .model import Model  # noqa: F401
from .loss import mse_loss, l1_loss, smooth_l1_loss  # noqa: F401
