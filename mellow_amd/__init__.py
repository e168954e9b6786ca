"""mellow_amd — MI355X-native inference engine for the Mellow audio-language model.

`from mellow_amd import MellowWrapper` mirrors `from mellow import MellowWrapper` of the reference."""
import os as _os

# A CONVENIENCE, not a requirement: HIP maps streams onto a few hardware queues (4 by default).  The engine itself measures
# whether the side stream of its split LM prefill runs beside its main stream and replaces it / falls back to one chain when
# it does not (mellow_prefill_parts, include/mellow_hip.h), so a raw C-ABI consumer needs nothing from this file.  A serving
# pool (serve.py) wants one queue per context on top of that; with 4 queues two contexts can land on ONE queue and serialise
# (right answers, no overlap: measured as `pipelined` 543 -> 443 responses/s).  The variable is read when the HIP runtime
# initialises, i.e. at the process's first GPU call -- a host that touches the GPU before importing this package should export it itself.
if "GPU_MAX_HW_QUEUES" not in _os.environ:
    _os.environ["GPU_MAX_HW_QUEUES"] = "8"
    import sys as _sys
    _torch = _sys.modules.get("torch")      # only a hint for EnginePool's warning: was the GPU runtime already up at this point?
    if _torch is not None and _torch.cuda.is_initialized():
        _os.environ["MELLOW_HWQ_SET_BY_IMPORT"] = "1"

__all__ = ["MellowWrapper"]


def __getattr__(name):
    if name == "MellowWrapper":
        from .wrapper import MellowWrapper
        return MellowWrapper
    raise AttributeError(name)
