"""segment_csr of torch-scatter as the reference uses it (custom_functions.py:132-133: per-ray sums of per-sample gradients)."""
import torch

__all__ = ['segment_csr']


def segment_csr(src: torch.Tensor, indptr: torch.Tensor, out: torch.Tensor | None = None, reduce: str = 'sum') -> torch.Tensor:
    if reduce not in ('sum', 'add', 'mean', 'min', 'max'):
        raise ValueError(f'segment_csr: unsupported reduce {reduce!r}')
    res = torch.segment_reduce(src, 'sum' if reduce == 'add' else reduce, offsets=indptr.to(torch.int64), axis=0, initial=0 if reduce in ('sum', 'add') else None)
    if out is not None:
        out.copy_(res)
        return out
    return res
