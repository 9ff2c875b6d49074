"""W ranks of the PRODUCT iteration in ONE process on ONE device, talking through device-local, ASYNCHRONOUS stand-in
collectives.  Test infrastructure.

Why: the GPU box has a single MI355X and RCCL refuses two ranks on one device, so the multi-rank tests that exist stage
`all_to_all_single` / `all_gather_into_tensor` through the host -- every such round trip synchronises the device and would
hide a missing stream dependency (a forgotten `wait_event` / `record_stream` between the exchange's side stream and the
renderer's stream).  Here nothing ever waits for the device on behalf of a collective:

* every rank is a thread with its OWN HIP stream as torch's current stream (current streams are thread-local) and its own
  view of the process-global state the mirror reads (`utils.GLOBAL_RANK`, `DEFAULT_GROUP`, the deferred-backward sink,
  the side streams ...), swapped in and out when the baton changes hands;
* exactly one rank runs at a time (a baton): a rank runs until it enters a collective, posts its buffers together with an
  event recorded on the stream the collective is issued on, and hands the baton on.  When all W ranks have posted, each
  one in turn makes ITS stream wait for the peers' events (device-side `hipStreamWaitEvent`, the host does not wait),
  copies its segments out of the peers' buffers with plain device copies, records a "read" event, and a second round
  makes every source's stream wait for its readers -- the completion semantics of an RCCL collective on a stream;
* `async_op=True` collectives run on a per-rank communication stream that first waits for the issuing stream, like
  ProcessGroupNCCL's; `work.wait()` is a stream-level join;
* the k-th collective of every rank must be the same operation on the same group (checked): a rank-dependent ORDER of
  collectives -- what the token chain of the exchange's autograd nodes exists to prevent -- fails as a mismatch / deadlock
  instead of hanging a node.

Autograd: a backward pass over device tensors normally runs on the engine's one worker thread per device, which a
blocked rank would starve; the ranks therefore run under `torch.autograd.set_multithreading_enabled(False)` (backward
nodes execute on the calling thread, on the streams their forward ran on -- the engine's stream semantics are unchanged).

Use:
    world = FakeWorld(W, device)
    results = world.run(fn)          # fn(rank) -> result, executed by W cooperating threads
"""
import threading

import torch
import torch.distributed as dist


class FakeGroup:
    """duck type of a process group: size() / rank().  `rank=None`: the rank of whoever holds the baton (for groups
    created inside the world by dist.new_group, which all ranks share as ONE object)"""

    def __init__(self, world, rank=None):
        self._world, self._rank = world, rank

    def size(self):
        return self._world.W

    def rank(self):
        return self._world.current if self._rank is None else self._rank


class _Work:
    def __init__(self, stream=None, issuing=None, keep=()):
        self._stream, self._issuing, self._keep = stream, issuing, keep

    def wait(self, *a, **k):
        if self._stream is not None:
            torch.cuda.current_stream().wait_stream(self._stream)
        self._keep = ()
        return True

    def is_completed(self):
        return self._stream is None or self._stream.query()


class CollectiveMismatch(RuntimeError):
    pass


class FakeWorld:
    def __init__(self, W, device, swap=None):
        self.W, self.device = W, torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.current = 0
        self._cv = threading.Condition()
        self._seq = [0] * W            # per rank: index of its next collective
        self._slots = {}               # collective index -> {"tag": ..., "posts": {rank: payload}}
        self._waiting = [None] * W     # per rank: the slot index it waits for, or None
        self._finished = [False] * W
        self._error = None
        self._saved = [None] * W       # per rank: snapshot of the swapped process-global state
        self._swap = swap or default_swap_list()
        self.groups = [FakeGroup(self, r) for r in range(W)]
        self.streams = [torch.cuda.Stream(self.device) for _ in range(W)] if self.cuda else [None] * W
        self.comm_streams = [torch.cuda.Stream(self.device) for _ in range(W)] if self.cuda else [None] * W
        self.log = []                  # (collective index, tag) in rendezvous order: what the ranks agreed on

    # ------------------------------------------------------------------ process-global state, one view per rank
    def _snapshot(self):
        return [get() for get, _ in self._swap]

    def _restore(self, values):
        for (_, put), v in zip(self._swap, values):
            put(v)

    # ------------------------------------------------------------------ the baton
    def _runnable(self, r):
        if self._finished[r]:
            return False
        k = self._waiting[r]
        return k is None or len(self._slots[k]["posts"]) == self.W

    def _hand_over(self, me):
        """called with the lock held by the rank that holds the baton: pick the next runnable rank (round-robin)"""
        for d in range(1, self.W + 1):
            r = (me + d) % self.W
            if self._runnable(r):
                self.current = r
                self._cv.notify_all()
                return
        if all(self._finished):
            self.current = -1
            self._cv.notify_all()
            return
        stuck = {r: (self._waiting[r], self._slots[self._waiting[r]]["tag"]) for r in range(self.W)
                 if not self._finished[r] and self._waiting[r] is not None}
        self._error = CollectiveMismatch(f"deadlock: no rank can proceed; waiting on (collective index, tag): {stuck}; "
                                         f"finished: {self._finished}")
        self.current = -2
        self._cv.notify_all()

    def _yield(self, me):
        """give the baton away and return when it comes back (state of the process globals swapped around the wait)"""
        with self._cv:
            self._saved[me] = self._snapshot()
            self._hand_over(me)
            while self.current != me and self._error is None:
                self._cv.wait()
            if self._error is not None:
                raise CollectiveMismatch(f"rank {me}: aborted -- {self._error}")
            self._restore(self._saved[me])

    def rendezvous(self, tag, payload):
        """every rank's k-th call meets here: -> [payload of rank 0, ..., payload of rank W-1]"""
        me = self.current
        k = self._seq[me]
        self._seq[me] += 1
        with self._cv:
            slot = self._slots.setdefault(k, {"tag": tag, "posts": {}, "left": self.W})
            if slot["tag"] != tag:
                self._error = CollectiveMismatch(f"collective #{k}: rank {me} calls {tag!r} where rank(s) "
                                                 f"{sorted(slot['posts'])} called {slot['tag']!r}")
                self.current = -2
                self._cv.notify_all()
                raise self._error
            slot["posts"][me] = payload
            if len(slot["posts"]) == self.W:
                self.log.append((k, tag))
            self._waiting[me] = k
        # always pass the baton at a collective (keeps the ranks in lock-step: a deterministic interleaving)
        self._yield(me)
        with self._cv:
            self._waiting[me] = None
            posts = [slot["posts"][r] for r in range(self.W)]
            slot["left"] -= 1
            if slot["left"] == 0:
                del self._slots[k]
        return posts

    # ------------------------------------------------------------------ running the ranks
    def run(self, fn, timeout=600.0):
        results, errors = [None] * self.W, [None] * self.W
        base = self._snapshot()
        for r in range(self.W):
            self._saved[r] = list(base)

        def body(r):
            try:
                with self._cv:
                    while self.current != r and self._error is None:
                        self._cv.wait()
                    if self._error is not None:
                        return
                    self._restore(self._saved[r])
                if self.cuda:
                    torch.cuda.set_device(self.device)
                    torch.cuda.set_stream(self.streams[r])
                with torch.autograd.set_multithreading_enabled(False):
                    results[r] = fn(r)
            except BaseException as e:  # noqa: BLE001
                errors[r] = e
                with self._cv:
                    if self._error is None:
                        self._error = e
            finally:
                with self._cv:
                    self._finished[r] = True
                    self._waiting[r] = None
                    self._saved[r] = self._snapshot()
                    if self._error is not None:
                        self.current = -2
                        self._cv.notify_all()
                    elif self.current == r:
                        self._hand_over(r)

        self.current = -3  # nobody yet
        threads = [threading.Thread(target=body, args=(r,), name=f"fake-rank-{r}", daemon=True) for r in range(self.W)]
        patches = _install(self)
        try:
            for t in threads:
                t.start()
            with self._cv:
                self.current = 0
                self._cv.notify_all()
            for t in threads:
                t.join(timeout)
                if t.is_alive():
                    with self._cv:
                        self._error = self._error or TimeoutError(f"{t.name} still running after {timeout} s")
                        self.current = -2
                        self._cv.notify_all()
        finally:
            _uninstall(patches)
            self._restore(base)
        first = next((e for e in errors if e is not None and not isinstance(e, CollectiveMismatch)), None) or \
            next((e for e in errors if e is not None), None) or self._error
        if first is not None:
            raise first
        if self.cuda:
            torch.cuda.synchronize(self.device)
        return results


# ---------------------------------------------------------------------- which process globals every rank owns
def _attr(obj, name):
    return (lambda: getattr(obj, name, None)), (lambda v: setattr(obj, name, v))


def _item(container, key):
    return (lambda: container[key]), (lambda v: container.__setitem__(key, v))


def _dict_contents(d):
    def put(v):
        d.clear()
        d.update(v)

    return (lambda: dict(d)), put


def default_swap_list():
    """(getter, setter) pairs of the process-global state of the product that belongs to ONE rank"""
    import diff_gaussian_rasterization as dgr
    import gaussian_renderer as gr
    import gaussian_renderer.workload_division as wd
    import utils.general_utils as utils

    pairs = [_attr(utils, n) for n in ("ARGS", "LOG_FILE", "CUR_ITER", "GLOBAL_RANK", "LOCAL_RANK", "WORLD_SIZE",
                                        "DP_GROUP", "MP_GROUP", "DEFAULT_GROUP", "IN_NODE_GROUP", "TIMERS",
                                        "DENSIFY_ITER", "IMG_H", "IMG_W", "TILE_Y", "TILE_X")]
    pairs.append(_item(dgr._DEFERRED_SINK, 0))
    pairs.append(_dict_contents(gr._SIDE_STREAMS))
    pairs.append(_dict_contents(gr.exchange_stats))
    pairs.append(_dict_contents(wd._BALANCE))
    pairs.append(_attr(dgr._RenderGaussians, "_pending"))
    return pairs


# ---------------------------------------------------------------------- the stand-in collectives
def _install(world):
    names = ["all_to_all_single", "all_gather_into_tensor", "all_gather", "all_reduce", "barrier", "broadcast",
             "all_gather_object", "new_group", "is_initialized", "get_world_size", "get_rank", "get_backend"]
    saved = {n: getattr(dist, n) for n in names}
    cuda = world.cuda

    def issue_stream(async_op):
        """-> (stream the copies run on, issuing stream): the current stream, or the rank's communication stream made
        to wait for it (async_op)"""
        if not cuda:
            return None, None
        cur = torch.cuda.current_stream()
        if not async_op:
            return cur, cur
        comm = world.comm_streams[world.current]
        comm.wait_stream(cur)
        return comm, cur

    def ready_event(stream):
        if stream is None:
            return None
        ev = torch.cuda.Event()
        ev.record(stream)
        return ev

    def complete(tag, stream, async_op, keep):
        """second round: nobody's buffers are released / overwritten before every peer has read them"""
        dones = world.rendezvous(tag + "/read", ready_event(stream))
        if stream is not None:
            for ev in dones:
                stream.wait_event(ev)
        if async_op:
            return _Work(stream if cuda else None, keep=keep)
        return None

    def on(stream):
        return torch.cuda.stream(stream) if stream is not None else _Null()

    def all_to_all_single(output, input, output_split_sizes=None, input_split_sizes=None, group=None, async_op=False):
        W, me = world.W, world.current
        if input_split_sizes is None:
            input_split_sizes = [input.shape[0] // W] * W
        if output_split_sizes is None:
            output_split_sizes = [output.shape[0] // W] * W
        if sum(input_split_sizes) != input.shape[0] or sum(output_split_sizes) != output.shape[0]:
            raise ValueError(f"all_to_all_single: split sizes {input_split_sizes} / {output_split_sizes} do not add up to "
                             f"the buffers' rows {input.shape[0]} / {output.shape[0]}")
        stream, _ = issue_stream(async_op)
        posts = world.rendezvous("all_to_all_single", (input, list(input_split_sizes), ready_event(stream)))
        with on(stream):
            o = 0
            for i in range(W):
                src, splits, ev = posts[i]
                if stream is not None:
                    stream.wait_event(ev)
                lo, n = sum(splits[:me]), splits[me]
                if n != output_split_sizes[i]:
                    raise CollectiveMismatch(f"all_to_all_single: rank {me} expects {output_split_sizes[i]} rows from "
                                             f"rank {i}, which sends {n}")
                if n:
                    output[o:o + n].copy_(src[lo:lo + n], non_blocking=True)
                o += n
        return complete("all_to_all_single", stream, async_op, (posts, output))

    def all_gather_into_tensor(output, input, group=None, async_op=False):
        W = world.W
        stream, _ = issue_stream(async_op)
        posts = world.rendezvous("all_gather_into_tensor", (input, ready_event(stream)))
        with on(stream):
            out = output.view(W, -1)
            for i in range(W):
                src, ev = posts[i]
                if stream is not None:
                    stream.wait_event(ev)
                out[i].copy_(src.reshape(-1), non_blocking=True)
        return complete("all_gather_into_tensor", stream, async_op, (posts, output))

    def all_gather(tensor_list, tensor, group=None, async_op=False):
        W = world.W
        stream, _ = issue_stream(async_op) if tensor.is_cuda else (None, None)
        posts = world.rendezvous("all_gather", (tensor, ready_event(stream)))
        with on(stream):
            for i in range(W):
                src, ev = posts[i]
                if stream is not None:
                    stream.wait_event(ev)
                tensor_list[i].copy_(src, non_blocking=True)
        return complete("all_gather", stream, async_op, (posts, tensor_list))

    def all_reduce(tensor, op=dist.ReduceOp.SUM, group=None, async_op=False):
        W = world.W
        stream, _ = issue_stream(async_op) if tensor.is_cuda else (None, None)
        posts = world.rendezvous("all_reduce", (tensor, ready_event(stream)))
        with on(stream):
            acc = None
            for i in range(W):  # rank order: every rank forms the same sum
                src, ev = posts[i]
                if stream is not None:
                    stream.wait_event(ev)
                if acc is None:
                    acc = src.clone()
                elif op == dist.ReduceOp.SUM:
                    acc += src
                elif op == dist.ReduceOp.MAX:
                    acc = torch.maximum(acc, src)
                elif op == dist.ReduceOp.MIN:
                    acc = torch.minimum(acc, src)
                else:
                    raise NotImplementedError(f"fake all_reduce: {op}")
        # nobody overwrites its tensor before every peer has read it
        dones = world.rendezvous("all_reduce/read", ready_event(stream))
        with on(stream):
            if stream is not None:
                for ev in dones:
                    stream.wait_event(ev)
            tensor.copy_(acc)
        return _Work(stream, keep=(posts, acc)) if async_op else None

    def broadcast(tensor, src=0, group=None, async_op=False):
        stream, _ = issue_stream(async_op) if tensor.is_cuda else (None, None)
        posts = world.rendezvous("broadcast", (tensor, src, ready_event(stream)))
        if any(p[1] != src for p in posts):
            raise CollectiveMismatch(f"broadcast: roots differ {[p[1] for p in posts]}")
        with on(stream):
            if world.current != src:
                t, _, ev = posts[src]
                if stream is not None:
                    stream.wait_event(ev)
                tensor.copy_(t, non_blocking=True)
        return complete("broadcast", stream, async_op, (posts, tensor))

    def barrier(group=None, async_op=False, **kw):
        world.rendezvous("barrier", None)
        return _Work() if async_op else None

    def all_gather_object(object_list, obj, group=None):
        posts = world.rendezvous("all_gather_object", obj)
        for i in range(world.W):
            object_list[i] = posts[i]

    def new_group(*a, **k):
        # a collective in the real library: every rank makes the call; all share the one dynamic-rank group object
        posts = world.rendezvous("new_group", None)
        del posts
        if not hasattr(world, "_extra_group"):
            world._extra_group = FakeGroup(world, None)
        return world._extra_group

    patched = dict(all_to_all_single=all_to_all_single, all_gather_into_tensor=all_gather_into_tensor,
                   all_gather=all_gather, all_reduce=all_reduce, barrier=barrier, broadcast=broadcast,
                   all_gather_object=all_gather_object, new_group=new_group, is_initialized=lambda: True,
                   get_world_size=lambda group=None: world.W, get_rank=lambda group=None: world.current,
                   get_backend=lambda group=None: "fake-device-local")
    for n, f in patched.items():
        setattr(dist, n, f)
    return saved


def _uninstall(saved):
    for n, f in saved.items():
        setattr(dist, n, f)


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False
