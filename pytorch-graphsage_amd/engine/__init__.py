"""
engine -- the whole train_step (reference models.py:97-104) as recorded launches.

  captured.CapturedTrainStep   GSSupervised.train_step under autograd, captured once as hipGraphs
  common.FusedTrainStep        what the fused engines share (buckets, sampler descriptor, head, finalise + Adam,
                               command lists / graphs, batch queue + software pipeline, data-parallel order)
  mean.FusedMeanTrainStep      mean aggregators (BASELINE configs[1], [4])
  pool.FusedPoolTrainStep      max-pool / mean-pool aggregators (configs[2])
  attn.FusedAttnTrainStep      attention aggregators, optionally over trainable node embeddings (configs[3])

`fused_engine_for(model, feats)` picks the engine that covers a model, or explains why none does.
"""
import sys

from .captured import CapturedTrainStep                      # noqa: F401
from .common import FusedTrainStep, _PrepDesc, _ReduceDesc   # noqa: F401
from .mean import FusedMeanTrainStep
from .pool import FusedPoolTrainStep
from .attn import FusedAttnTrainStep

ENGINES = (FusedMeanTrainStep, FusedPoolTrainStep, FusedAttnTrainStep)


def why_no_fused_engine(model, feats, ddp=None):
    """{engine name: what it does not cover} for every fused engine (empty when one covers the model)."""
    out = {}
    for cls in ENGINES:
        why = cls.why_not(model, feats, ddp)
        if why is None:
            return {}
        out[cls.__name__] = why
    return out


def fused_engine_for(model, feats, explain=False, ddp=None):
    """The fused train-step engine that covers (model, feats), or None (callers then fall back to
    GSSupervised.train_step, optionally captured by CapturedTrainStep).  explain=True: when none does, say on
    stderr what each engine misses, so nobody lands on the slow path without knowing.  ddp: the data-parallel handle
    the engine will be built with (dist.DataParallel), so that what an engine cannot do under it is said HERE."""
    for cls in ENGINES:
        if cls.supports(model, feats, ddp):
            return cls
    if explain:
        why = why_no_fused_engine(model, feats, ddp)
        print("gsage: no fused train-step engine covers this model -- " +
              "; ".join("%s: %s" % kv for kv in why.items()), file=sys.stderr)
    return None
