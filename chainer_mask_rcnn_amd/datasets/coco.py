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
        anns = self._anns_of_img[img_id]
        img = np.asarray(PIL.Image.open(self.img_fname.format(img_id)))
        if img.ndim == 2:
            img = np.repeat(img[:, :, None], 3, axis=2)       # cv2.COLOR_GRAY2RGB
        elif img.shape[2] == 4:
            img = img[:, :, :3]
        example = self._annotations_to_example(anns, img.shape[0], img.shape[1])
        return tuple([img] + example)

    def _annotations_to_example(self, anns, height, width):
        import PIL.Image
        import PIL.ImageDraw
        bboxes, labels, masks, crowds, areas = [], [], [], [], []
        for ann in anns:
            if 'segmentation' not in ann:
                continue
            if not self._use_crowd and ann['iscrowd'] == 1:
                continue
            class_id = self.cat_id_to_class_id[ann['category_id']]
            if isinstance(ann['segmentation'], list):
                # polygon
                mask = PIL.Image.fromarray(np.zeros((height, width), dtype=np.uint8))
                for seg in ann['segmentation']:
                    xy = np.array(seg).reshape((-1, 2))
                    xy = [tuple(xy_i) for xy_i in xy]
                    PIL.ImageDraw.Draw(mask).polygon(xy=xy, outline=1, fill=1)
                mask = np.asarray(mask)
            else:
                # run-length mask
                mask = rle_decode(ann['segmentation'], height, width)
                # FIXME (reference): some of minival annotations are malformed.
                if mask.shape != (height, width):
                    continue
            mask = mask == 1
            bboxes.append(mask_to_bbox(mask))  # y1, x1, y2, x2
            masks.append(mask)
            labels.append(class_id)
            crowds.append(ann['iscrowd'])
            areas.append(ann['area'])
        bboxes = np.asarray(bboxes, dtype=np.float32).reshape((-1, 4))
        labels = np.asarray(labels, dtype=np.int32)
        masks = np.asarray(masks, dtype=np.int32).reshape((-1, height, width))
        example = [bboxes, labels, masks]
        if self._return_crowd:
            example.append(np.asarray(crowds, dtype=np.int32))
        if self._return_area:
            example.append(np.asarray(areas, dtype=np.float32))
        return example
