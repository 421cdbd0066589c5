"""pympc_amd -- MI355X-native implementation of pyMPC's QP hot path.

``MPCController`` is a drop-in for ``pyMPC.mpc.MPCController`` (reference: pyMPC/mpc.py); the QP
build and the OSQP-style ADMM solve run in hand-written HIP (csrc/mpcqp*.h) behind the C ABI of
include/mpcqp.h.  ``BatchMPCController`` is the additive batched surface (many independent
controllers of equal dimensions on one GPU, sharded over GPUs with torch.distributed).
"""
from .controller import MPCController
from .batch import BatchMPCController
from .unconstrained import unconstrained_gains

__all__ = ['MPCController', 'BatchMPCController', 'unconstrained_gains']
