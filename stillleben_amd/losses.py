"""Negative-IoU loss (interface of the reference's stillleben/losses.py:5-21; pure torch tensor ops)."""


def neg_iou_loss(predict, target):
    """predict, target: BxCxHxW.  Returns (loss, loss image) like the reference: the scalar 1 - mean IoU over the batch and the
    per-pixel 1 - intersection / (union + 1e-6), detached."""
    dims = tuple(range(predict.ndimension())[1:])
    inter_img = predict * target
    union_img = predict + target - predict * target
    intersect = inter_img.sum(dims)
    union = union_img.sum(dims) + 1e-6
    loss_img = (1.0 - inter_img / (union_img + 1e-6)).detach().clone()
    return 1.0 - (intersect / union).sum() / intersect.nelement(), loss_img
