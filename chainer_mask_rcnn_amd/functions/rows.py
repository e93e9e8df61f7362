"""Row fan-out of the RoI head: the box branch consumes every res5 row, the mask branch only
the foreground rows (MaskRCNNTrainChain.mask_branch_fg_only).  Plain ``index_select`` makes
autograd materialise a zero-filled full-size gradient for the row subset and add it to the
box branch's gradient (three extra passes over the (R,2048,7,7) tensor); this node adds the
subset's gradient rows into the full gradient in place instead."""
import torch

from ._layout import nhwc


class _FanoutRowsFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, rows):
        x = nhwc(x)
        phys = x.permute(0, 2, 3, 1)                       # dense (N,H,W,C)
        sub = phys.index_select(0, rows).permute(0, 3, 1, 2)
        ctx.save_for_backward(rows)
        ctx.shape = tuple(x.shape)
        return x.view_as(x), sub

    @staticmethod
    def backward(ctx, g_full, g_rows):
        rows, = ctx.saved_tensors
        if g_full is None:
            if g_rows is None:
                return None, None
            n, c, h, w = ctx.shape
            g_full = torch.zeros((n, h, w, c), dtype=g_rows.dtype,
                                 device=g_rows.device).permute(0, 3, 1, 2)
        else:
            g_full = nhwc(g_full)
        if g_rows is not None:
            # g_full is the tensor the box branch's backward just produced (or the zeros
            # above): accumulate into it rather than allocating a third one
            g_full.permute(0, 2, 3, 1).index_add_(0, rows, nhwc(g_rows).permute(0, 2, 3, 1))
        return g_full, None


def fanout_rows(x, rows):
    """Returns ``(x, x[rows])``; gradients of the two outputs are combined in place."""
    return _FanoutRowsFn.apply(x, rows)
