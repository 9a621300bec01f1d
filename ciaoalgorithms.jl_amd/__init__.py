"""ciaoalgorithms.jl_amd -- MI355X-native finite-sum inner loops (SVRG / SAGA+SAG / Finito / LFinito) behind the
CIAOAlgorithms.jl solver API.

The directory name contains a dot, so it is loaded under the importable alias `ciaoalgorithms_jl_amd`
(see /ciao_loader.py).  Layout:
    csrc/            HIP kernels (gfx950) + the extern "C" ABI  -> libciao_hip.so (declared in /include/ciao_hip.h)
    _lib.py          ctypes binding of that ABI (what a Julia `ccall` would bind)
    device.py        Context / PackedF / ProxG: torch device tensors -> ABI calls
    operators.py     ProximalOperators-style descriptions of f_i and g, and their packing
    sampling.py      injected index streams (the reference's RNG draws are an explicit input here)
    solvers.py       SVRG / SAGA / SAG / Finito constructors, functors, iterator(), solution()
    parallel.py      row sharding + torch.distributed (RCCL) all-reduce hook
    julia/           the Julia wrapper module (written against the same ABI; cannot run in this image)
Importing this package does not need a GPU; creating a Context does.  There is no CPU fallback.
"""
from . import _lib
from .sampling import FixedStream, IndexStream

__all__ = ["_lib", "IndexStream", "FixedStream"]


def __getattr__(name):
    # torch-dependent modules are imported lazily so that `import ciaoalgorithms_jl_amd` stays cheap
    import importlib
    for mod in ("solvers", "operators", "device", "parallel"):
        m = importlib.import_module(f"{__name__}.{mod}")
        if hasattr(m, name):
            return getattr(m, name)
        if name == mod:
            return m
    raise AttributeError(name)
