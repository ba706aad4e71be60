"""Loss helpers referenced by the DARTS model wrapper (mirror of utils/util_loss.py:8-64).  local_global_loss with the mean-squared
criterion runs on the device in one call (risp_local_global_l2: no boolean indexing, no host read of the flags); any other criterion, and
tensors the kernel does not take, follow the reference's torch formulation."""
import torch
import torch.nn.functional as TF


def latency_loss(img_in, img_gt, latency, target_latency, w, fidelity_loss):
    """fidelity * (latency / target_latency) ** w  ->  (loss, latency term)"""
    term = (latency / target_latency) ** w
    return fidelity_loss(img_in, img_gt) * term, term


def local_global_loss(img_in, img_gt, glb_flag, loss_func):
    """Images flagged local (<1): loss after a detached per-channel mean-matching gain clamped to
    [0.5, 2]; images flagged global (>=1): loss on 1/4-scale bilinear down-samples."""
    if (getattr(loss_func, 'kind', None) == 'l2' and img_in.is_cuda and img_in.dim() == 4 and img_in.shape == img_gt.shape
            and img_in.dtype == img_gt.dtype == torch.float32 and img_in.shape[2] % 4 == 0 and img_in.shape[3] % 4 == 0
            and img_in.is_contiguous() and img_gt.is_contiguous() and not img_gt.requires_grad
            and (img_in.data_ptr() | img_gt.data_ptr()) % 16 == 0 and glb_flag.numel() == img_in.shape[0]):
        from ... import functional as F
        return F.local_global_l2(img_in, img_gt, glb_flag)
    total = 0.
    local = glb_flag < 1
    if local.any():
        a, b = img_in[local], img_gt[local]
        mean_a = a.mean((2, 3), keepdim=True).clamp(0, None) + 1e-6
        gain = (b.mean((2, 3), keepdim=True) / mean_a).clamp(0.5, 2.).detach()
        total = total + loss_func(a * gain, b)
    if (~local).any():
        small = lambda t: TF.interpolate(t, scale_factor=0.25, mode='bilinear', align_corners=False)
        total = total + loss_func(small(img_in[~local]), small(img_gt[~local]))
    return total
