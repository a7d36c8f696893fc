"""
pytorch-graphsage_amd -- MI355X (gfx950) native GraphSAGE training hot path behind the plugin
surface of bkj/pytorch-graphsage.  Directory name has a hyphen: import it with
`importlib.import_module("pytorch-graphsage_amd")`, or run `pytorch-graphsage_amd/train.py`.

    csrc/ + libgsage_hip.so   hand-written HIP kernels behind the C ABI of include/gsage.h
    _native.py                ctypes binding (fails loudly when the library is missing)
    ops.py / store.py         tensor-level operators, HBM data layouts
    nn_modules.py, models.py, problem.py, helpers.py, lr.py, train.py
                              same names and interfaces as the reference's files
    dist.py                   RCCL data-parallel gradient sync
    engine/                   the whole train_step as recorded launches (captured autograd path; fused mean /
                              pool / attention engines on a shared base)
"""
from . import _native, dist, engine, helpers, nn_modules, ops, optim, problem, store               # noqa: F401
from .helpers import set_seeds, to_numpy                            # noqa: F401
from .lr import LRSchedule                                          # noqa: F401
from .models import GSSupervised                                    # noqa: F401
from .nn_modules import aggregator_lookup, prep_lookup, sampler_lookup   # noqa: F401
from .problem import DeviceMetrics, NodeProblem, ProblemLosses, ProblemMetrics, batch_metric   # noqa: F401
from .store import DenseAdj, DeviceCSR, FeatureStore, RowRef                  # noqa: F401

__all__ = ["GSSupervised", "NodeProblem", "aggregator_lookup", "prep_lookup", "sampler_lookup",
           "set_seeds", "to_numpy", "LRSchedule", "FeatureStore", "DeviceCSR", "RowRef", "ops"]
