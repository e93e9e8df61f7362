"""Model classes of the hot path; the public names are those of the reference's
``chainer_mask_rcnn.models`` package."""
from . import utils  # noqa: F401
from . import mask_rcnn as _m
from . import mask_rcnn_resnet as _mr
from . import mask_rcnn_train_chain as _tc
from . import region_proposal_network as _rpn
from . import resnet_extractor as _re

MaskRCNN = _m.MaskRCNN
MaskRCNNResNet = _mr.MaskRCNNResNet
MaskRCNNTrainChain = _tc.MaskRCNNTrainChain
RegionProposalNetwork = _rpn.RegionProposalNetwork
ResNet50Extractor = _re.ResNet50Extractor
ResNet101Extractor = _re.ResNet101Extractor

__all__ = ['MaskRCNN', 'MaskRCNNResNet', 'MaskRCNNTrainChain', 'RegionProposalNetwork',
           'ResNet50Extractor', 'ResNet101Extractor', 'utils']
