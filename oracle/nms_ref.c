/*
 * ORACLE (test infrastructure, not product code).
 *
 * Greedy non-maximum suppression as performed by chainercv's
 * `non_maximum_suppression` CPU path (chainercv is an un-vendored dependency
 * of the reference: requirements.txt:2 `chainercv>=0.9.0`; call sites
 * /root/reference/chainer_mask_rcnn/models/mask_rcnn.py:193-194 and, through
 * ProposalCreator, models/region_proposal_network.py:136-138).  Restated from
 * its published algorithm — "parity unpinned": the
 * reference holds no golden vectors for it; pinned here by a brute-force
 * O(n^2) definition in tests/test_oracle_boxes.py.
 *
 * Boxes are (y_min, x_min, y_max, x_max) fp32, already sorted by descending
 * score.  Box i is selected unless IoU(i, s) >= thresh for some previously
 * selected s.  All arithmetic fp32, no FMA (-ffp-contract=off):
 *   tl = max(b_i[:2], b_s[:2]); br = min(b_i[2:], b_s[2:])
 *   inter = (br_y - tl_y) * (br_x - tl_x) * (tl_y < br_y && tl_x < br_x)
 *   iou = inter / (area_i + area_s - inter)
 */
#include <stdint.h>
#include <stdlib.h>

int oracle_nms_sorted(const float *bbox, int n, float thresh, int limit,
                      int32_t *keep)
{
    float *area = (float *)malloc(sizeof(float) * (n > 0 ? n : 1));
    for (int i = 0; i < n; ++i)
        area[i] = (bbox[4 * i + 2] - bbox[4 * i + 0]) *
                  (bbox[4 * i + 3] - bbox[4 * i + 1]);
    int cnt = 0;
    for (int i = 0; i < n; ++i) {
        const float *b = bbox + 4 * i;
        int suppressed = 0;
        for (int k = 0; k < cnt; ++k) {
            const float *s = bbox + 4 * keep[k];
            float tly = b[0] > s[0] ? b[0] : s[0];
            float tlx = b[1] > s[1] ? b[1] : s[1];
            float bry = b[2] < s[2] ? b[2] : s[2];
            float brx = b[3] < s[3] ? b[3] : s[3];
            float inter = (bry - tly) * (brx - tlx);
            if (!(tly < bry && tlx < brx)) inter = inter * 0.f;
            float iou = inter / (area[i] + area[keep[k]] - inter);
            if (iou >= thresh) { suppressed = 1; break; }
        }
        if (!suppressed) {
            keep[cnt++] = i;
            if (limit > 0 && cnt >= limit) break;
        }
    }
    free(area);
    return cnt;
}
