import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(scope='session')
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no ROCm device')
    # The test models are small (64 sampled RoIs, 224x320 images): lower the work threshold of
    # the Winograd route (a performance heuristic, functions/conv.py:uses_winograd) so that the
    # model-level parity tests run the SAME route as the full-size configurations — the RoI
    # head's 3x3 layers and the RPN conv1 on Winograd backward / inference forward.
    # Restored at session end so that CPU tests of the routing policy sharing this process see
    # the shipped default whatever the test order.
    from chainer_mask_rcnn_amd.functions import conv
    saved = conv.WINOGRAD_MIN_WORK
    conv.WINOGRAD_MIN_WORK = 1 << 24
    try:
        yield torch.device('cuda:0')
    finally:
        conv.WINOGRAD_MIN_WORK = saved
