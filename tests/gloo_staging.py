"""Test scaffolding for the two-ranks-on-ONE-GPU runs (RCCL refuses two ranks on one device, so those tests use gloo): gloo is never handed a
device tensor.  `install()` wraps torch.distributed.all_gather_into_tensor and all_reduce so that, on a gloo group, device tensors are staged explicitly -
input .cpu() (which drains the current stream), the collective on host tensors, the result copied back on the current stream.  The product
module (ikflow_amd/dist.py) therefore carries no gloo-specific code; under RCCL (`nccl`) the wrapper passes straight through.
No oracle import here: bench.py's test-only backend switch (IKF_BENCH_TEST_BACKEND=gloo) uses this module too."""
import torch
import torch.distributed as dist

_orig = None
_orig_reduce = None


def install() -> None:
    global _orig, _orig_reduce
    if _orig is not None:
        return
    _orig = dist.all_gather_into_tensor
    _orig_reduce = dist.all_reduce

    def staged_reduce(tensor, op=dist.ReduceOp.SUM, group=None, async_op=False):
        if not (tensor.is_cuda and dist.get_backend(group) == "gloo"):
            return _orig_reduce(tensor, op=op, group=group, async_op=async_op)
        assert not async_op, "the staged gloo form is synchronous"
        host = tensor.detach().cpu()
        _orig_reduce(host, op=op, group=group)
        tensor.copy_(host)
        return None

    def staged(output_tensor, input_tensor, group=None, async_op=False):
        if not (input_tensor.is_cuda and dist.get_backend(group) == "gloo"):
            return _orig(output_tensor, input_tensor, group=group, async_op=async_op)
        assert not async_op, "the staged gloo form is synchronous"
        host_in = input_tensor.detach().cpu()
        host_out = torch.empty(output_tensor.shape, dtype=output_tensor.dtype)
        _orig(host_out, host_in, group=group)
        output_tensor.copy_(host_out)
        return None

    dist.all_gather_into_tensor = staged
    dist.all_reduce = staged_reduce
