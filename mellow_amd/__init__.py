"""mellow_amd — MI355X-native inference engine for the Mellow audio-language model.

`from mellow_amd import MellowWrapper` mirrors `from mellow import MellowWrapper` of the reference."""
import os as _os

# HIP maps streams onto a few hardware queues (4 by default).  An engine uses up to two streams for its split LM prefill and a
# serving pool one main stream per context; with 4 queues two of them can land on ONE queue and serialise (right answers, no
# overlap: measured as `pipelined` 543 -> 443 responses/s).  The variable is read when the HIP runtime initialises, i.e. at the
# process's first GPU call -- a host that touches the GPU before importing this package should export it itself.
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
