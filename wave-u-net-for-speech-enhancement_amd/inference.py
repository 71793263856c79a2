"""chunked inference.

The reference enhances a long recording by zero-padding it to a multiple of `sample_length`, splitting it into
chunks and running the model on every chunk SEQUENTIALLY at batch 1 with a device->host copy per chunk
(/root/reference/enhancement.py:57-69, trainer/trainer.py:66-78).  In eval mode BatchNorm uses running statistics,
so chunks are independent: they are stacked into one [n_chunks, 1, sample_length] batch, run through ONE forward
(optionally in slabs of `max_batch` chunks) and trimmed back - same numbers, one launch sequence, one copy.
"""
import torch


@torch.no_grad()
def enhance(model, mixture, sample_length=16384, max_batch=256, to_host=False):
    """mixture: float32 [1, 1, T] (any T, the reference's batch-1 contract, enhancement.py:50) on the model's device.
    Returns the enhanced waveform [1, 1, T] - on the model's device, or with to_host=True as a host tensor (what the reference
    hands to the file writer, enhancement.py:66-71): every slab goes into one pinned buffer with an asynchronous copy,
    one synchronisation at the end instead of one per chunk."""
    if mixture.dim() != 3 or mixture.shape[0] != 1 or mixture.shape[1] != 1:
        raise ValueError("Only support batch size is 1 in enhancement stage.")       # reference wording, enhancement.py:50
    if model.training:
        raise RuntimeError("enhance() needs model.eval(): chunks are only independent with running BatchNorm statistics")
    T = mixture.shape[-1]
    pad = (-T) % sample_length                                   # enhancement.py:57-59
    if pad:
        mixture = torch.cat([mixture, torch.zeros(1, 1, pad, device=mixture.device, dtype=mixture.dtype)], dim=-1)
    chunks = mixture.reshape(-1, 1, sample_length)               # == torch.split(..., sample_length, dim=-1) stacked
    n = chunks.shape[0]
    # equal slabs (586 chunks -> 196 + 196 + 194, not 256 + 256 + 74): no small ragged forward at the end, one context size
    nslab = (n + max_batch - 1) // max_batch
    max_batch = (n + nslab - 1) // nslab
    if to_host and mixture.is_cuda:
        host = torch.empty(n, 1, sample_length, dtype=mixture.dtype, pin_memory=True)
        for i in range(0, n, max_batch):
            host[i:i + max_batch].copy_(model(chunks[i:i + max_batch].contiguous()), non_blocking=True)
        torch.cuda.current_stream(mixture.device).synchronize()
        return host.reshape(1, 1, -1)[:, :, :T]
    outs = [model(chunks[i:i + max_batch].contiguous()) for i in range(0, n, max_batch)]
    enhanced = torch.cat(outs, dim=0).reshape(1, 1, -1)          # enhancement.py:68
    enhanced = enhanced[:, :, :T]                                # enhancement.py:69 trims the padding
    return enhanced.cpu() if to_host else enhanced
