"""
helpers.py -- counterparts of the reference's helpers.py:14-25.

`set_seeds` seeds numpy's global legacy stream and torch exactly like the reference, because
in rng="compat" mode the epoch shuffle (problem.py:146) and the sampler's `sel`
(nn_modules.py:88) are drawn from that one stream.  `to_numpy` is the working version of the
reference helper (which recurses forever on torch >= 0.4).
"""
import numpy as np
import torch


class LegacyStreamOnDevice(object):
    """numpy's global legacy MT19937 stream, lent to the GPU.

    In rng="compat" mode the sampler's `sel` comes from the very stream the reference draws from
    (nn_modules.py:88).  Drawing it on the host costs ~1 ms of numpy plus a 560 KB H2D copy per
    Reddit-sized step; `gsage_mt_choice_device` consumes the same words on the device.  The stream is one
    sequence, so at any time it lives in exactly one place:
      acquire(device)  numpy -> device (2.5 KB, once; no-op while the device already holds it)
      release()        device -> numpy, before ANYTHING draws from np.random on the host (the epoch
                       shuffle of NodeProblem.iterate, evaluation in host mode, user code) -- the callers
                       in this package do it; a host draw that slips in between is detected (the numpy
                       state is compared with the snapshot taken at acquire) and raises
      drop()           forget the device copy (np.random.seed re-defines the stream)
    Enabled per process with `enabled = True` (train.py does for CUDA runs with --rng compat)."""

    enabled = False

    def __init__(self):
        self.state, self.snapshot = None, None

    @property
    def on_device(self):
        return self.snapshot is not None

    def acquire(self, device):
        if self.on_device and self.state.device == device:
            return self.state
        self.release()
        name, key, pos, has_gauss, cached = np.random.get_state()
        assert name == "MT19937"
        host = np.concatenate([np.asarray(key, dtype=np.uint32), np.array([pos], dtype=np.uint32)])
        self.state = torch.from_numpy(host.view(np.int32).copy()).to(device)
        self.snapshot = (np.array(key, dtype=np.uint32, copy=True), int(pos), int(has_gauss), float(cached))
        return self.state

    def release(self):
        if not self.on_device:
            return
        name, key, pos, has_gauss, cached = np.random.get_state()
        k0, p0, g0, c0 = self.snapshot
        if int(pos) != p0 or not np.array_equal(np.asarray(key, dtype=np.uint32), k0):
            self.snapshot = None
            raise RuntimeError("np.random was used on the host while the GPU held the legacy stream "
                               "(helpers.legacy_stream.release() must precede host draws)")
        arr = self.state.cpu().numpy().view(np.uint32)
        np.random.set_state(("MT19937", arr[:624].copy(), int(arr[624]), g0, c0))
        self.snapshot = None

    def drop(self):
        self.snapshot = None


legacy_stream = LegacyStreamOnDevice()


def set_seeds(seed=0):
    legacy_stream.drop()                 # a re-seeded stream supersedes whatever the device held
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)


def to_numpy(x):
    if hasattr(x, "materialize"):
        x = x.materialize()
    return x.detach().cpu().float().numpy() if x.is_floating_point() else x.detach().cpu().numpy()
