# same export list as reference core/models/__init__.py:1-14 (ResNet/VGG backbones are out of scope)
from .attention import PositionalEncoding, MultiheadedAttention, UniModalAttention, PrototypeAttention
from .bn_inception import bninception, BNInception
from .dataparallel import DataParallel
from .model_builder import build_model
from .model import TBNModel, Fusion, Classifier
from .contrast_loss import ContrastLoss
