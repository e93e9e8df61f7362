"""COCO instance-segmentation dataset — same class name, constructor, example layout and
annotation handling as the reference's
/root/reference/chainer_mask_rcnn/datasets/coco.py:18-183, without its
third-party readers: the annotation index that ``pycocotools.coco.COCO`` builds is built here
from the JSON directly, COCO run-length masks (``pycocotools.mask.frPyObjects`` / ``decode``)
are decoded by ``rle_decode`` below, images are read with Pillow (the reference uses
``skimage.io.imread``, which returns the same RGB uint8 array for JPEG files).  Polygons are
rasterised with ``PIL.ImageDraw`` exactly as the reference does (:136-143).

    dataset[i] -> img (H,W,3) uint8 RGB, bboxes (G,4) f32 (y1,x1,y2,x2), labels (G,) i32,
                  masks (G,H,W) i32 {0,1} [, crowds (G,) i32] [, areas (G,) f32]

which is what ``datasets.MaskRCNNTransform`` consumes (datasets/transforms.py:10-51).
There is no network here: the data must already be under ``root_dir``.
"""
import json
import os.path as osp

import numpy as np


def rle_counts_from_string(s):
    """COCO's compressed RLE string -> run lengths (pycocotools maskApi.c rleFrString): 5 data
    bits per character (offset 48), bit 5 = continuation, sign-extended, and from the third
    run on each value is a difference to the run two places back."""
    if isinstance(s, str):
        s = s.encode('ascii')
    counts, p = [], 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = s[p] - 48
            x |= (c & 0x1f) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(counts) > 2:
            x += counts[-2]
        counts.append(x)
    return counts


def rle_decode(segmentation, height, width):
    """{'counts': list | str | bytes, 'size': [h, w]} -> (h, w) uint8 mask
    (pycocotools.mask.frPyObjects + decode: runs alternate 0 / 1 in COLUMN-major order)."""
    h, w = segmentation['size']
    counts = segmentation['counts']
    if not isinstance(counts, (list, tuple)):
        counts = rle_counts_from_string(counts)
    flat = np.zeros(h * w, dtype=np.uint8)
    pos, v = 0, 0
    for c in counts:
        if v:
            flat[pos:pos + c] = 1
        pos += c
        v ^= 1
    return flat.reshape((w, h)).T


def mask_to_bbox(mask):
    """utils/geometry.py:150-166: (y1, x1, y2, x2) of the non-zero pixels (raises on an empty
    mask, as the reference does)."""
    where = np.argwhere(mask)
    (y1, x1), (y2, x2) = where.min(0), where.max(0) + 1
    return y1, x1, y2, x2


class COCOInstanceSegmentationDataset(object):

    class_names = None  # initialized by __init__
    root_dir = osp.expanduser('~/data/datasets/COCO')

    def __init__(self, split, use_crowd=False, return_crowd=False, return_area=False,
                 root_dir=None):
        if root_dir is not None:
            self.root_dir = root_dir
        if split == 'train':
            split = split + '2014'
            data_type = 'train2014'
        elif split in ['val', 'minival', 'valminusminival']:
            split = split + '2014'
            data_type = 'val2014'
        else:
            raise ValueError
        ann_file = osp.join(self.root_dir, 'annotations/instances_%s.json' % split)
        if not osp.exists(ann_file):
            raise IOError('%s not found; the reference downloads it (coco.py:24-50), this '
                          'build has no network access: place the data there' % ann_file)
        self._use_crowd = use_crowd
        self._return_crowd = return_crowd
        self._return_area = return_area

        with open(ann_file) as f:
            data = json.load(f)
        # the index pycocotools.coco.COCO.createIndex builds
        self._anns_of_img = {}
        for ann in data.get('annotations', []):
            self._anns_of_img.setdefault(ann['image_id'], []).append(ann)
        self.img_fname = osp.join(self.root_dir, data_type, 'COCO_%s_{:012}.jpg' % data_type)

        # set class_names (:83-94)
        cat_id_to_class_id = {}
        class_names = []
        for cat in sorted(data.get('categories', []), key=lambda x: x['id']):
            cat_id_to_class_id[cat['id']] = len(class_names)
            class_names.append(cat['name'])
        class_names = np.asarray(class_names)
        class_names.setflags(write=0)
        self.cat_id_to_class_id = cat_id_to_class_id
        self.class_names = class_names

        # filter images without any annotations (:96-102)
        self.img_ids = [img['id'] for img in data.get('images', [])
                        if len(self._anns_of_img.get(img['id'], [])) >= 1]

    def __len__(self):
        return len(self.img_ids)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self.get_example(j) for j in range(*i.indices(len(self)))]
        return self.get_example(i)

    def get_example(self, i):
        import PIL.Image
        img_id = self.img_ids[i]
        # Every file ends up as HxWx3 uint8 RGB whatever its JPEG colour model: grayscale is
        # replicated (the reference's cv2.COLOR_GRAY2RGB), CMYK / palette / alpha files go through
        # PIL's colour conversion (skimage.io.imread, which the reference calls, does the same);
        # slicing the first three planes of a CMYK array would be wrong pixel data.
        with PIL.Image.open(self.img_fname.format(img_id)) as f:
            img = np.asarray(f if f.mode in ('L', 'RGB') else f.convert('RGB'))
        if img.ndim == 2:
            img = np.repeat(img[:, :, None], 3, axis=2)
        example = self._annotations_to_example(self._anns_of_img[img_id], img.shape[0], img.shape[1])
        return tuple([img] + example)

    # -- annotations -> (bboxes, labels, masks[, crowds][, areas]) -------------------------------
    # Contract of the reference's datasets/coco.py:123-176 (pinned by tests/golden/coco_example.npz):
    # an instance is kept when it has a segmentation, is not a crowd region (unless use_crowd) and
    # — for run-length masks — decodes to the image size; its box is the tight box of its mask.
    @staticmethod
    def _rasterise(segmentation, height, width):
        """Polygon list or COCO RLE dict -> (height, width) bool mask, or None when a run-length
        mask does not match the image (malformed minival annotations are dropped)."""
        if isinstance(segmentation, list):
            import PIL.Image
            import PIL.ImageDraw
            canvas = PIL.Image.new('L', (width, height), 0)
            pen = PIL.ImageDraw.Draw(canvas)
            for ring in segmentation:
                pts = np.asarray(ring).reshape(-1, 2)
                pen.polygon(xy=list(map(tuple, pts)), outline=1, fill=1)
            return np.asarray(canvas) == 1
        decoded = rle_decode(segmentation, height, width)
        return decoded == 1 if decoded.shape == (height, width) else None

    def _kept_instances(self, anns, height, width):
        for ann in anns:
            if 'segmentation' not in ann or (ann['iscrowd'] == 1 and not self._use_crowd):
                continue
            inst = self._rasterise(ann['segmentation'], height, width)
            if inst is not None:
                yield ann, inst

    def _annotations_to_example(self, anns, height, width):
        kept = list(self._kept_instances(anns, height, width))
        col = lambda fn, dt: np.asarray([fn(a, m) for a, m in kept], dtype=dt)
        example = [
            col(lambda a, m: mask_to_bbox(m), np.float32).reshape((-1, 4)),       # y1, x1, y2, x2
            col(lambda a, m: self.cat_id_to_class_id[a['category_id']], np.int32),
            col(lambda a, m: m, np.int32).reshape((-1, height, width)),
        ]
        if self._return_crowd:
            example.append(col(lambda a, m: a['iscrowd'], np.int32))
        if self._return_area:
            example.append(col(lambda a, m: a['area'], np.float32))
        return example
