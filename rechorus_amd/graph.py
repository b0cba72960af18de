"""hipGraph replay of the dense training step (model(batch) -> loss -> backward -> optimizer.step).

At the reference's default batch size (256) a step is ~25 kernels of a few microseconds each, and the Python
between them (autograd Functions, module calls, ctypes) costs several times more than the GPU work.  The
whole step is therefore recorded ONCE per batch shape with HIP stream capture (torch.cuda.CUDAGraph: the
engine launches on torch's current stream, takes its scratch from caches that are warm by then, never
synchronises) and replayed with the next batch copied into static input buffers.  Adam's step count lives
in device memory (HipOptimizer(capturable=True), rc_dense_update_multi_dev), so bias correction advances
across replays.  Eligibility is decided by the runner (helpers/BaseRunner.py): host-free forward
(no host-side candidate shuffle; torch's dropout replays with a fresh Philox offset), HipOptimizer, CUDA tensors.

ROCm caveat (measured on ROCm 7.0 / MI355X, repro: tools/repro_hipgraph_fault.py): with the runtime's
default "AQL packet capture" fast path for graphs, ONE device-to-host copy on the default stream between
two launches of an instantiated graph (a `.item()` on any tensor, e.g. the epoch loss) makes the next
launch fault ("Memory access fault by GPU").  DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 selects the runtime's
regular graph launch path, which does not have the problem.  The variable is read when HIP initialises,
so `enable()` sets it at package import if the process has not touched the GPU yet, and `usable()` tells
the runner whether graphs may be used in this process; otherwise training stays eager.
"""
import os
import warnings

import torch

_ENV = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"
_OK = False
# eager steps on the default stream after a capture (the last, smaller batch of an epoch) meet AccumulateGrad
# nodes that autograd created on the capture stream; the streams are joined correctly, the notice is noise
warnings.filterwarnings("ignore", message="The AccumulateGrad node's stream does not match")


def _hsa_initialised():
    """the ROCm runtime opens /dev/kfd when it initialises; after that its environment flags are frozen"""
    try:
        for fd in os.listdir("/proc/self/fd"):
            try:
                if os.readlink(os.path.join("/proc/self/fd", fd)) == "/dev/kfd":
                    return True
            except OSError:
                pass
    except OSError:
        return True  # cannot tell: assume the worst
    return False


def enable():
    """select the safe graph launch path for this process (called at package import).  If the variable is
    already 0 (launcher, tests/conftest.py, bench.py) nothing is to do; if the runtime is already up without
    it, graphs stay disabled for this process."""
    global _OK
    if os.environ.get(_ENV) == "0":
        _OK = True
    elif _hsa_initialised():
        _OK = False
    else:
        os.environ[_ENV] = "0"
        _OK = True


def usable():
    return _OK


class GraphedStep:
    """captured training step for one feed-dict shape; `run(batch)` trains on `batch` and returns the loss"""

    WARMUP = 2  # eager steps before capture: optimizer state, workspaces and autotuned paths exist by then

    def __init__(self, model, loss_of=None):
        self.model = model
        # loss_of(model, batch) -> scalar loss; default: the reference's `model.loss(model(batch))`
        self.loss_of = loss_of or (lambda m, b: m.loss(m(b)))
        self.seen = 0
        self.graph = None
        self.static = None
        self.groups = None
        self.loss = None

    @staticmethod
    def signature(batch):
        return tuple((k, tuple(v.shape), v.dtype) for k, v in sorted(batch.items()) if isinstance(v, torch.Tensor))

    def _announce(self, on):
        """while a whole step (forward, backward, optimizer.step()) runs under this object's control, the model may see its
        optimizer (`_step_optimizer`): the FM family's field gather then hands small batches to HipOptimizer's rows mode, which
        is only correct when backward and step() are certain to follow the forward (the tables get no .grad)"""
        opt = getattr(self.model, 'optimizer', None)
        if on and getattr(opt, 'rows_ok', None) is not None and opt.rows_ok():
            object.__setattr__(self.model, '_step_optimizer', opt)
        elif '_step_optimizer' in self.model.__dict__:
            object.__delattr__(self.model, '_step_optimizer')
        if not on and getattr(opt, 'rows_abort', None) is not None:
            opt.rows_abort()    # a step that raised between the forward and step() must not poison the next one
        from . import engine, nn as hnn
        engine.clear_bumped_early()
        # the head's forward launch may leave the backward fan-out for exactly this seed buffer (nn._CtrHeadFn)
        hnn.UNIT_LOSS_GRAD = getattr(self, '_one', None) if on else None

    def _eager(self, batch):
        model = self.model
        model.optimizer.zero_grad()
        self._announce(True)
        try:
            loss = self.loss_of(model, batch)
            one = getattr(self, '_one', None)
            if one is not None and loss.dim() == 0 and loss.dtype == torch.float32 and loss.device == one.device:
                loss.backward(gradient=one)
            else:
                loss.backward()
            model.optimizer.step()
        finally:
            self._announce(False)
        return loss.detach().reshape(1)

    def run(self, batch):
        """warm-up steps and replays run on the caller's current stream; only the capture itself happens on
        the side stream torch.cuda.graph provides"""
        if self.graph is None:
            self.seen += 1
            if self.seen <= self.WARMUP:
                return self._eager(batch)
            self._capture(batch)
        else:
            # one launch per dtype: the batch's tensors are concatenated straight into the flat buffer the static inputs view
            # (ten 4.7 us copies per DeepFM step otherwise -- a tenth of the replayed step at B = 1,024)
            for flat, keys in self.groups:
                if len(keys) == 1:
                    self.static[keys[0]].copy_(batch[keys[0]])
                else:
                    torch.cat([batch[k].reshape(-1) for k in keys], out=flat)
        self.graph.replay()
        return self.loss.detach().reshape(1).clone()

    def _capture(self, batch):
        model = self.model
        # static inputs: one flat buffer per dtype, every tensor of the feed dict a view into it
        self.static, self.groups = dict(batch), []
        by_dtype = {}
        for k, v in sorted(batch.items()):
            if isinstance(v, torch.Tensor):
                by_dtype.setdefault((v.dtype, v.device), []).append(k)
        for (dtype, dev), keys in by_dtype.items():
            flat = torch.empty(sum(batch[k].numel() for k in keys), dtype=dtype, device=dev)
            o = 0
            for k in keys:
                n = batch[k].numel()
                self.static[k] = flat[o:o + n].view(batch[k].shape)
                self.static[k].copy_(batch[k])
                o += n
            self.groups.append((flat, keys))
        model.optimizer.zero_grad()  # grads must be None: the captured backward allocates them in the graph's pool
        dev = next((p.device for p in model.parameters()), None)      # where the loss will live (a batch may hold no tensor)
        if dev is None:
            dev = next((v.device for v in batch.values() if isinstance(v, torch.Tensor)), None)
        self._one = torch.ones((), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with warnings.catch_warnings():
            # autograd notes that the AccumulateGrad nodes were created on the default stream (warm-up); the
            # capture stream joins it before and after, which is what we want
            warnings.simplefilter('ignore', UserWarning)
            one = None
            self._announce(True)
            try:
                with torch.cuda.graph(self.graph):
                    self.loss = self.loss_of(model, self.static)
                    # the seed gradient of the scalar loss from a buffer filled once, not a ones_like fill in every replay (a launch
                    # of its own: ~4.6 us of a 0.26 ms DeepFM step)
                    one = self._one if (self.loss.dim() == 0 and self.loss.dtype == torch.float32 and self.loss.device == self._one.device) else None
                    self.loss.backward(gradient=one) if one is not None else self.loss.backward()
                    model.optimizer.step()
            finally:
                self._announce(False)
