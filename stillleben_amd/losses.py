"""Negative-IoU loss (interface of the reference's stillleben/losses.py; pure torch tensor ops)."""


def neg_iou_loss(predict, target):
    dims = tuple(range(predict.ndimension())[1:])
    intersect = (predict * target).sum(dims)
    union = (predict + target - predict * target).sum(dims) + 1e-6
    return 1.0 - (intersect / union).sum() / intersect.nelement()
