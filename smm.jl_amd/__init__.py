"""smm.jl_amd — MI355X (gfx950) backend for the BGP parallel-tempering hot path of
floswald/SMM.jl, plus the host-side mirror of the reference's API surface for that path
(MProb / Eval / MAlgoBGP / BGPChain; src/SMM.jl:31-57).

Import as `import smm_jl_amd` (alias module at the repo root).
The compute path is libsmmhip.so (hand-written HIP, csrc/); there is no CPU fallback.
"""
from . import _abi
from ._abi import SMMHipError
from .backend import BGPContext, BGPOpts, Problem, Tables, hip_context, register_user_objective
from .host import (CI, BGPChain, Eval, MAlgoBGP, MProb, addEvalFunc, addMoment, addParam, addSampledParam, allAccepted,
                   banana, best, computeNextIteration, dataMoment, dataMomentd, dataMomentW, dataMomentWd,
                   evaluateObjective, fill, dense_sim, dense_sim2, history, mean, median, ms_names, objfunc_norm, param, paramd, params,
                   ps2s_names, ps_names, readMalgo, restart, run, save, serialNormal, setMoments, setValue, snorm_impl,
                   summary, user_objective)
from .callers import (FD_gradient, Slice, doSlices, evaluateObjectives, getSigma, get_stdErrors, optSlices, range_length)

__all__ = [n for n in dir() if not n.startswith("_")] + ["_abi"]
