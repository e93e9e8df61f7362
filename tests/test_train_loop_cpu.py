"""Host logic of tools/train_loop.py (the counterpart of examples/train_common.py:200-231): the
SerialIterator restatement consumes the GLOBAL np.random stream as chainer's does — one permutation
at construction, one in-place shuffle whenever a batch reaches the end of the data, batches wrapping
into the new order — and TransformDataset applies its transform at access time."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
import train_loop as TL


def test_serial_iterator_order_and_rng_consumption():
    data = list(range(7))
    np.random.seed(3)
    it = TL.SerialIterator(data, 3)
    got = [it.next(), it.next()]
    assert it.epoch == 0 and not it.is_new_epoch and it.crosses_epoch()
    b3 = it.next()
    assert it.epoch == 1 and it.is_new_epoch and it.current_position == 2
    after = np.random.randint(0, 1 << 30)
    # replay of what chainer's iterator draws: a permutation at construction, then ONE in-place
    # shuffle of the same array when the third batch reaches the end of the data
    np.random.seed(3)
    order = np.random.permutation(7)
    assert got[0] == [int(j) for j in order[:3]] and got[1] == [int(j) for j in order[3:6]]
    last = int(order[6])
    np.random.shuffle(order)
    assert b3 == [last] + [int(j) for j in order[:2]]
    assert np.random.randint(0, 1 << 30) == after        # the global stream is at the same place
    # exact multiple of the batch size: shuffle, no wrap
    np.random.seed(1)
    it = TL.SerialIterator(list(range(4)), 2)
    it.next()
    assert it.crosses_epoch()
    it.next()
    assert it.epoch == 1 and it.current_position == 0
    # unshuffled
    it = TL.SerialIterator(list(range(5)), 2, shuffle=False)
    assert [it.next(), it.next(), it.next()] == [[0, 1], [2, 3], [4, 0]]


def test_transform_dataset_applies_at_access_time():
    calls = []
    ds = TL.TransformDataset([10, 20, 30], lambda v: (calls.append(v), v + 1)[1])
    assert len(ds) == 3 and not calls
    assert ds[1] == 21 and calls == [20]


def test_synthetic_instances_contract():
    d = TL.SyntheticInstances(2, seed=1, height=60, width=80, n_gt=3, virtual_len=10)
    assert len(d) == 10
    img, bbox, label, mask = d[7]
    assert img.dtype == np.uint8 and img.shape == (60, 80, 3)
    assert bbox.dtype == np.float32 and bbox.shape == (3, 4)
    assert label.dtype == np.int32 and mask.dtype == np.int32 and mask.shape == (3, 60, 80)
    for g in range(3):                                   # tight boxes of the masks (mask_to_bbox)
        ys, xs = np.nonzero(mask[g])
        assert tuple(bbox[g]) == (ys.min(), xs.min(), ys.max() + 1, xs.max() + 1)
