# flake8: noqa
"""Same public names as /root/reference/chainer_mask_rcnn/models/__init__.py:1-9."""
from . import utils

from .mask_rcnn import MaskRCNN
from .mask_rcnn_resnet import MaskRCNNResNet
from .mask_rcnn_train_chain import MaskRCNNTrainChain
from .region_proposal_network import RegionProposalNetwork
from .resnet_extractor import ResNet101Extractor
from .resnet_extractor import ResNet50Extractor
