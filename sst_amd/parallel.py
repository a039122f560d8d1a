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
  * ``finish()`` sends whatever did not complete (parameters without a gradient this step count as zeros, as in DDP with
    ``find_unused_parameters``), waits, and divides by the world size inside the same pass (``ReduceOp.AVG`` where the
    backend has it).
"""
import torch
import torch.distributed as dist


class GradBucketReducer(object):
    """reducer = GradBucketReducer(params, n_buckets=2); per step: ``p.grad = None`` for all -> backward (hooks fire) ->
    ``reducer.finish()`` -> every ``p.grad`` is a view of the averaged flat buffer."""

    def __init__(self, params, n_buckets=2, group=None, overlap=True):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError('GradBucketReducer: no parameters')
        dev, dtype = self.params[0].device, self.params[0].dtype
        if any(p.device != dev or p.dtype != dtype for p in self.params):
            raise ValueError('GradBucketReducer: parameters must share one device and dtype')
        self.group, self.overlap = group, overlap
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
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
        self._work = []
        self._avg = dist.is_initialized() and dist.get_backend(group) == 'nccl'
        self._hooks = [p.register_post_accumulate_grad_hook(self._make_hook(i)) for i, p in enumerate(self.params)]

    def _make_hook(self, i):
        def hook(param):
            b = self.bucket_of[i]
            self._pending[b] -= 1
            if self._pending[b] == 0 and self.overlap and not self._sent[b]:
                self._send(b)
        return hook

    def _send(self, b):
        start, end, idx = self.buckets[b]
        have = [i for i in idx if self.params[i].grad is not None]
        missing = [i for i in idx if self.params[i].grad is None]
        # gradients -> their slots of the persistent buffer: one multi-tensor copy
        if have:
            torch._foreach_copy_([self.views[i] for i in have], [self.params[i].grad for i in have])
        for i in missing:
            self.views[i].zero_()
        self._sent[b] = True
        if self.world > 1:
            op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
            self._work.append(dist.all_reduce(self.flat[start:end], op=op, group=self.group, async_op=True))

    def finish(self):
        """send what is left, wait for every bucket, average, point every ``p.grad`` at its slot"""
        for b in range(len(self.buckets)):
            if not self._sent[b]:
                self._send(b)
        for w in self._work:
            w.wait()
        if self.world > 1 and not self._avg:
            self.flat.div_(self.world)
        for i, p in enumerate(self.params):
            p.grad = self.views[i]
        self._work = []
        self._sent = [False] * len(self.buckets)
        self._pending = [len(idx) for _, _, idx in self.buckets]
        return self.flat

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
