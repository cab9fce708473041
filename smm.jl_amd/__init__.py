"""smm.jl_amd — MI355X (gfx950) backend for the BGP parallel-tempering hot path of
floswald/SMM.jl, plus the host-side mirror of the reference's API surface for that path
(MProb / Eval / MAlgoBGP / BGPChain; src/SMM.jl:31-57).

Import as `import smm_jl_amd` (alias module at the repo root).
The compute path is libsmmhip.so (hand-written HIP, csrc/); there is no CPU fallback.
"""
from . import _abi
from ._abi import SMMHipError
from .backend import BGPContext, BGPOpts, Problem, Tables, hip_context

__all__ = ["_abi", "SMMHipError", "BGPContext", "BGPOpts", "Problem", "Tables", "hip_context"]
