from .modules import *  # noqa: F401,F403
from .modules import ActFun, NormLayer, VGG_downscaler, VGG_upscaler, SimpleParamNet, ConvLSTM, ConvLSTMLayer  # noqa: F401
from .utils import set_gpu, split_feature, get_layer_size, Flatten, UnFlatten, free_bits_kl, batch_reduce  # noqa: F401
