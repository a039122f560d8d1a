"""Data-parallel gradient exchange of the path: frames shard over ranks, one process per GPU, the only
exchange besides naiveSyncBN's statistics is the gradient all-reduce (RCCL over xGMI; ``backend='nccl'`` IS RCCL on ROCm).

The reference trains under MMDistributedDataParallel (mmdet3d/apis/seq_training_apis.py:146-150, tools/train.py): bucketed
all-reduces launched from autograd hooks while the backward pass is still running.  Here the same idea is sized for this
network and for point-to-point xGMI links (7 x ~153 GB/s per GPU: a ring is per-link bound, so few LARGE messages):

  * ONE persistent flat fp32 buffer holds every gradient (8.4 MB for SST-base); it is allocated once, every ``p.grad``
    is a view into it after the exchange - no per-step ``torch.cat``, no per-step allocation;
  * the buffer is cut into a few buckets in REVERSE registration order (the order the backward pass produces gradients:
    last encoder layer first, voxel encoder last); a bucket is sent as soon as its last gradient exists
    (``register_post_accumulate_grad_hook``), asynchronously, so the encoder stack's buckets travel while the backward of
    the voxel encoder / index stages still runs; only the last bucket is exposed;
  * buckets are sent STRICTLY in index order (bucket b only after b - 1), on a process group of their own: RCCL matches
    collectives by issue order per communicator, so (a) every rank must issue the buckets in the same order whatever subset
    of its parameters received a gradient (a class head without points on one rank), and (b) they must not interleave with
    naiveSyncBN's blocking all-reduces of the backward pass, which run on the default group;
  * ``finish()`` sends whatever did not complete (a bucket that holds a parameter without a gradient this step is only
    complete here; such parameters travel as zeros, as in DDP with ``find_unused_parameters``), waits, and divides by the
    world size inside the same pass (``ReduceOp.AVG`` where the backend has it);
  * one backward pass per ``finish()``: a hook that fires for a bucket already on the wire raises instead of losing the
    gradient (gradient accumulation over several backward passes needs ``overlap=False``);
  * the bucket communicator is created ONCE per process and shared by every reducer (``dist.new_group`` is collective over
    all ranks and communicators are never freed by ``remove()``: one per reducer leaked them - ADVICE round 4);
    ``ReduceOp.AVG`` support is decided at construction by a blocking one-element probe on that communicator (an
    asynchronous refusal mid-step could not be caught), not on the first bucket of a step;
  * ``static_graph`` (default True) is the caller's statement that EVERY rank produces the same set of gradients every step
    (true for SST / FSD training steps on any frame).  Only then may a bucket leave mid-backward: with per-rank sets of unused
    parameters rank A would send bucket b before, rank B after one of naiveSyncBN's blocking all-reduces of the same backward
    pass (default communicator), and two collectives issued in different relative order on different ranks may deadlock when
    their kernels cannot co-run.  ``static_graph=False``: every bucket leaves in ``finish()`` - after the backward pass and all
    its norm-layer collectives - in index order on every rank;
  * the copy into the flat buffer (one ``_foreach_copy_`` per bucket, 8.4 MB for SST-base: ~10 us per step) is skipped for a
    gradient that already IS its slot (``p.grad`` left pointing at the view and accumulated into in place).  Pre-pointing
    every gradient is not the default: autograd then accumulates in place, one small ``add_`` launch per parameter
    (~100 launches, ~0.4 ms) instead of one multi-tensor copy.
"""
import torch
import torch.distributed as dist


# keyed by id(group) WITH the group object kept in the entry: a bare id() can be reused after destroy_process_group() +
# init_process_group() in one process (test harnesses, elastic restarts) and would hand back a destroyed communicator or a stale
# AVG decision; an entry that holds its group keeps that id taken (ProcessGroup objects take no weak references)
_BUCKET_GROUP = {}      # id(default group) -> (default group, the process-wide bucket communicator)
_AVG_SUPPORT = {}       # id(group) -> (group, {backend: bool})


def bucket_group():
    """the communicator the gradient buckets travel on: created on first use (collective: every rank builds its first reducer
    at the same point of the program), then shared by every reducer of the process"""
    world = dist.group.WORLD
    entry = _BUCKET_GROUP.get(id(world))
    if entry is None or entry[0] is not world:
        entry = _BUCKET_GROUP[id(world)] = (world, dist.new_group())
    return entry[1]


def avg_supported(group, device):
    """does the backend of ``group`` reduce with ReduceOp.AVG?  One blocking one-element all-reduce, once per communicator
    (collective).  RCCL builds that follow NCCL >= 2.10 do; gloo refuses at the call (-> sum and divide)."""
    entry = _AVG_SUPPORT.get(id(group))
    if entry is None or entry[0] is not group:
        entry = _AVG_SUPPORT[id(group)] = (group, {})
    per_group = entry[1]
    key = dist.get_backend(group)
    if key not in per_group:
        try:
            probe = torch.ones(1, dtype=torch.float32, device=device)
            dist.all_reduce(probe, op=dist.ReduceOp.AVG, group=group)
            if probe.is_cuda:
                torch.cuda.current_stream(device).synchronize()
            per_group[key] = bool(abs(float(probe.item()) - 1.0) < 1e-6)
        except (RuntimeError, ValueError, TypeError, NotImplementedError):
            per_group[key] = False
    return per_group[key]


class GradBucketReducer(object):
    """reducer = GradBucketReducer(params, n_buckets=2); per step: ``p.grad = None`` for all -> backward (hooks fire) ->
    ``reducer.finish()`` -> every ``p.grad`` is a view of the averaged flat buffer."""

    def __init__(self, params, n_buckets=2, group=None, overlap=True, static_graph=True):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError('GradBucketReducer: no parameters')
        dev, dtype = self.params[0].device, self.params[0].dtype
        if any(p.device != dev or p.dtype != dtype for p in self.params):
            raise ValueError('GradBucketReducer: parameters must share one device and dtype')
        self.overlap = bool(overlap) and bool(static_graph)   # per-rank gradient sets: nothing leaves before finish()
        self.static_graph = bool(static_graph)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        if group is None and self.world > 1:
            group = bucket_group()
        self.group = group
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, dtype=dtype, device=dev)
        # layout: reverse registration order, so that the gradients produced first sit in the first bucket
        order = list(reversed(range(len(self.params))))
        self.views, off = [None] * len(self.params), 0
        target = -(-total // max(1, int(n_buckets)))
        self.buckets, cur, cur_n = [], [], 0      # bucket = [start, end, parameter indices]
        start = 0
        for i in order:
            p = self.params[i]
            self.views[i] = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()
            cur.append(i)
            cur_n += p.numel()
            if cur_n >= target and len(self.buckets) < n_buckets - 1:
                self.buckets.append((start, off, cur))
                start, cur, cur_n = off, [], 0
        if cur:
            self.buckets.append((start, off, cur))
        self.bucket_of = {i: b for b, (_, _, idx) in enumerate(self.buckets) for i in idx}
        self._pending = [len(idx) for _, _, idx in self.buckets]
        self._sent = [False] * len(self.buckets)
        self._next = 0                        # buckets [0, _next) are on the wire
        self._work = []
        self._avg = self.world > 1 and avg_supported(self.group, dev)     # decided once, by a blocking probe
        self._hooks = [p.register_post_accumulate_grad_hook(self._make_hook(i)) for i, p in enumerate(self.params)]

    def _make_hook(self, i):
        def hook(param):
            if not self.overlap:               # everything leaves in finish(): any number of backward passes before it
                return
            b = self.bucket_of[i]
            if self._sent[b] or self._pending[b] <= 0:
                raise RuntimeError('GradBucketReducer: a second gradient arrived for a bucket that was already sent - one backward '
                                   'pass per finish(); use overlap=False to accumulate over several backward passes')
            self._pending[b] -= 1
            if self.overlap:
                while self._next < len(self.buckets) and self._pending[self._next] == 0:
                    self._send(self._next)
        return hook

    def _send(self, b):
        start, end, idx = self.buckets[b]
        have = [i for i in idx if self.params[i].grad is not None]
        missing = [i for i in idx if self.params[i].grad is None]
        # gradients -> their slots of the persistent buffer: one multi-tensor copy
        have = [i for i in have if self.params[i].grad.data_ptr() != self.views[i].data_ptr()]    # already in its slot
        if have:
            torch._foreach_copy_([self.views[i] for i in have], [self.params[i].grad for i in have])
        for i in missing:
            self.views[i].zero_()
        assert b == self._next, 'buckets leave in index order'
        self._sent[b] = True
        self._next = b + 1
        if self.world > 1:
            chunk = self.flat[start:end]
            op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
            self._work.append(dist.all_reduce(chunk, op=op, group=self.group, async_op=True))

    def finish(self):
        """send what is left, wait for every bucket, average, point every ``p.grad`` at its slot.  A parameter that produced no
        gradient this step travels as zeros and ends with a ZERO gradient (not None): an optimizer with weight decay / momentum
        then updates it, unlike torch DDP with find_unused_parameters (which leaves None) - the models of this package use every
        parameter in every step; a caller with conditional branches should drop those parameters from the reducer."""
        for b in range(self._next, len(self.buckets)):
            self._send(b)
        for w in self._work:
            w.wait()
        if self.world > 1 and not self._avg:
            self.flat.div_(self.world)
        for i, p in enumerate(self.params):
            p.grad = self.views[i]
        self._work = []
        self._sent = [False] * len(self.buckets)
        self._next = 0
        self._pending = [len(idx) for _, _, idx in self.buckets]
        return self.flat

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
