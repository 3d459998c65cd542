"""Thread-per-stage frame pipeline: the ``processors=[...]`` contract of the reference's forked
DataLoader (RTL/dataloader.py:112-131, :734-751, :1026-1053) for torch 2.x.

The reference file cannot even be imported on torch >= 1.9 (``torch._six``, RTL/dataloader.py:15),
so this module re-creates the part of it the reconstruction demo relies on (RTL/main.py:326-464):

* ``processors`` is a list of callables; stage k runs on its own daemon thread, fed by a FIFO
  queue from stage k-1 (dataloader.py:734-751); different frames occupy different stages at the
  same time and come out in submission order;
* every stage thread starts with ``torch.set_num_threads(1)`` and ``torch.cuda.set_device``
  (dataloader.py:1029-1031);
* an exception inside a stage is captured, travels down the remaining queues untouched and is
  re-raised in the consumer (``ExceptionWrapper``, dataloader.py:1042-1047, :909-914);
* at most ``max_in_flight`` frames are admitted at once (the reference primes ``2 * num_workers``,
  dataloader.py:776-777, :891).

Additions:

* the stage threads run with the autograd grad mode of the thread that iterates the pipeline (grad
  mode is thread-local in torch: without this a ``with torch.no_grad():`` around the consuming loop
  would not reach the stages);

* each stage thread owns a HIP stream; a frame is handed to the next stage together with an event
  recorded on the producer's stream, and the consumer's stream waits on it.  Stages therefore
  overlap on the GPU (not only on the host) without any ``synchronize`` call;
* a processor may be a ``Coalesced(fn, fn_many)``: when frames have queued up in front of its stage
  (the stage is the bottleneck), the stage takes up to ``max_batch`` of them at once and serves them
  with ONE call of ``fn_many(list) -> list`` -- one batched encoder pass, one ``mp_recon_batch``, one
  host sync for all their vertex counts -- instead of ``len(list)`` calls of ``fn``.  Per-frame
  semantics and FIFO order are unchanged (results leave in submission order; with an empty queue a
  frame is served alone, by ``fn``); an exception inside ``fn_many`` is re-tried frame by frame so
  that it lands on the frame that caused it -- which RE-RUNS ``fn`` on frames ``fn_many`` may already
  have worked on: the processors of a Coalesced stage must be idempotent per frame (no counters
  advanced per call, no state that a second evaluation of the same frame would corrupt; the
  reference's stateful camera step, RTL/main.py:330-336, is NOT a candidate).  A ``fn_many`` that finds
  it cannot batch raises ``Coalesced.CannotBatch`` BEFORE touching any state: the frames are then
  served by ``fn`` one by one without an error being recorded.  The reference's stages are one Python thread per
  frame step on a host that spends most of its time in the GIL; this is what the drop-in surface
  needs to reach the batched kernels.
"""
import queue
import sys
import threading

import torch

import os

_END = object()
_tls = threading.local()
SWITCH_INTERVAL = float(os.environ.get("MONOPORT_STAGE_SWITCH_INTERVAL", "0"))
# HIP stream priority per stage index ("4:-1,5:0": lower = served first by the GPU's dispatcher; default 0 for all)
STAGE_PRIORITY = {int(k): int(v) for k, v in (kv.split(":") for kv in os.environ.get("MONOPORT_STAGE_PRIORITY", "").split(",") if kv)}


def stage_kind():
    """What the calling host thread is: None (not a stage thread), "stage" (a per-frame stage of a StagePipeline)
    or "coalesced" (a stage whose processor is a ``Coalesced``).  The encoders use it to decide whether a call is
    worth replaying as a recorded plan (modeling/backbones.py: ENCODER_PLAN)."""
    return getattr(_tls, "kind", None)
# Stage streams are kept for the life of the process, one per (device, stage index): torch's caching
# allocator pools memory PER STREAM, so a pipeline that made fresh streams every time it is iterated
# would find none of the blocks its predecessor cached and go back to hipMalloc for every activation
# (measured: 1.5-1.9 s for the first batched encoder call of a pass).
_STREAMS = {}
_STREAMS_LOCK = threading.Lock()


def stage_stream(device, idx):
    """The HIP stream stage ``idx`` of every StagePipeline on ``device`` runs on (public: a caller can warm a
    stage up on its own stream, e.g. run a batched encoder once per batch size a Coalesced stage can meet, so
    that the allocator pool of that stream holds the blocks before the first frame arrives)."""
    key = (str(device), idx)
    with _STREAMS_LOCK:
        st = _STREAMS.get(key)
        if st is None:
            st = _STREAMS[key] = torch.cuda.Stream(device=device, priority=STAGE_PRIORITY.get(idx, 0))
    return st


def _record_streams(obj, stream, depth=0):
    """Tell the caching allocator that ``stream`` uses every CUDA tensor reachable from ``obj``
    (dict / list / tuple nesting as in the data_dict of RTL/main.py:326-452): a tensor allocated
    on the producer stage's stream must not be recycled while a later stage still reads it."""
    if torch.is_tensor(obj):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif depth < 4:
        if isinstance(obj, dict):
            for v in obj.values():
                _record_streams(v, stream, depth + 1)
        elif isinstance(obj, (list, tuple)):
            for v in obj:
                _record_streams(v, stream, depth + 1)


class StageError:
    """An exception raised inside a stage, re-raised by the consumer with its stage index."""

    def __init__(self, stage, exc_info):
        self.stage = stage
        self.exc_type, self.exc, self.tb = exc_info

    def reraise(self):
        raise RuntimeError("stage %d failed: %s: %s"
                           % (self.stage, self.exc_type.__name__, self.exc)) from self.exc


class Coalesced:
    """A stage processor that can serve several queued frames in one call (see the module
    docstring): ``fn(item) -> item`` and ``fn_many([item, ...]) -> [item, ...]`` (same order)."""

    class CannotBatch(Exception):
        """Raised by ``fn_many`` before any side effect: serve these frames one by one through ``fn``."""

    def __init__(self, fn, fn_many, max_batch=8, max_pending=1):
        """``max_pending``: batches of this stage that may be in flight on the GPU at once.  Before it takes
        frames off its queue the stage waits until at most ``max_pending - 1`` of its earlier batches are
        still running -- so batches are formed at the pace of the GPU, not of the host: a stage whose host
        side is fast (a recorded encoder plan returns in under a millisecond) would otherwise pick every
        frame up the moment it arrives and never see two of them together.  0 = no throttle.  Measured on the
        reference's processors list (bench.py --mode dropin, 16 frames per call, 48 in flight): 0 / 1 / 2 / 3 ->
        154 / 159 / 153 / 158 recon/s, +- 4 between runs; 1 ships."""
        self.fn, self.fn_many, self.max_batch = fn, fn_many, max(1, int(max_batch))
        self.max_pending = max(0, int(max_pending))

    def __call__(self, item):
        return self.fn(item)


class StagePipeline:
    def __init__(self, source, processors, device=None, max_in_flight=2, stage_streams=True):
        """``source``: iterable of input items (the reference's data stream);
        ``processors``: list of callables ``item -> item`` (e.g. the lambdas of RTL/main.py:326-452);
        ``device``: torch device of the stage threads (None = CPU only, no streams)."""
        self.source = source
        self.processors = list(processors)
        self.device = torch.device(device) if device is not None else None
        self.max_in_flight = max(1, int(max_in_flight))
        self.use_streams = bool(stage_streams and self.device is not None
                                and self.device.type == "cuda")
        self._threads = []
        self._grad_enabled = True
        self._stop = threading.Event()

    # ---- worker bodies ------------------------------------------------------------------------
    def _stage_loop(self, idx, fn, q_in, q_out):
        torch.set_num_threads(1)
        # autograd's grad mode is thread-local: a `with torch.no_grad():` around the loop that consumes the
        # pipeline would not reach the stage threads, and netG.filter -- which RTL/main.py:367-370 calls
        # undecorated -- would stay on the differentiable (MIOpen) path instead of the inference kernels
        torch.set_grad_enabled(self._grad_enabled)
        _tls.kind = "coalesced" if isinstance(fn, Coalesced) else "stage"
        stream = None
        if self.device is not None and self.device.type == "cuda":
            torch.cuda.set_device(self.device)
            if self.use_streams:
                stream = stage_stream(self.device, idx)
        held = None  # an item taken off the queue while coalescing that has to wait for its turn
        pending = []  # completion events of this stage's batches that may still be running (Coalesced stages)
        while True:
            item = held if held is not None else q_in.get()
            held = None
            if item is _END:
                q_out.put(_END)
                return
            payload, event = item
            if isinstance(payload, StageError):
                q_out.put((payload, None))
                continue
            batch = [item]
            if isinstance(fn, Coalesced):  # whatever else is ALREADY waiting, up to max_batch frames
                if fn.max_pending and stream is not None:  # ... once the GPU has room for another batch of this stage
                    pending = [ev for ev in pending if not ev.query()]
                    while len(pending) >= fn.max_pending:
                        pending.pop(0).synchronize()
                while len(batch) < fn.max_batch:
                    try:
                        nxt = q_in.get_nowait()
                    except queue.Empty:
                        break
                    if nxt is _END or isinstance(nxt[0], StageError):
                        held = nxt
                        break
                    batch.append(nxt)

            def run(items):
                payloads = [p for p, _ in items]
                if stream is None:
                    results = fn.fn_many(payloads) if len(items) > 1 else [fn(payloads[0])]
                    return [(r, None) for r in results]
                with torch.cuda.stream(stream):
                    for p, ev in items:
                        if ev is not None:
                            stream.wait_event(ev)
                        _record_streams(p, stream)
                    results = fn.fn_many(payloads) if len(items) > 1 else [fn(payloads[0])]
                    done = torch.cuda.Event()
                    done.record(stream)
                return [(r, done) for r in results]

            try:
                outs = run(batch)
                if len(outs) != len(batch):
                    raise RuntimeError("fn_many returned %d results for %d frames" % (len(outs), len(batch)))
            except Exception:  # noqa: BLE001 -- forwarded to the consumer like ExceptionWrapper
                first = sys.exc_info()
                outs = []
                if len(batch) == 1:
                    outs.append((StageError(idx, first), None))
                for one in (batch if len(batch) > 1 else []):  # frame by frame: the error lands on its frame
                    try:
                        outs.extend(run([one]))
                    except Exception:  # noqa: BLE001
                        outs.append((StageError(idx, sys.exc_info()), None))
            if isinstance(fn, Coalesced) and outs and outs[-1][1] is not None:
                pending.append(outs[-1][1])
            for o in outs:
                q_out.put(o)

    def _feeder(self, q0, slots):
        try:
            for item in self.source:
                slots.acquire()
                if self._stop.is_set():  # the consumer went away (an error was re-raised, or it stopped iterating)
                    break
                q0.put((item, None))
        except Exception:  # noqa: BLE001
            q0.put((StageError(-1, sys.exc_info()), None))
        q0.put(_END)

    # ---- consumer ---------------------------------------------------------------------------------
    def __iter__(self):
        self._grad_enabled = torch.is_grad_enabled()  # of the consuming thread, handed to the stage threads
        queues = [queue.Queue() for _ in range(len(self.processors) + 1)]
        slots = threading.Semaphore(self.max_in_flight)
        self._threads = [threading.Thread(target=self._feeder, args=(queues[0], slots), daemon=True)]
        for i, fn in enumerate(self.processors):
            self._threads.append(threading.Thread(target=self._stage_loop,
                                                  args=(i, fn, queues[i], queues[i + 1]),
                                                  daemon=True))
        self._stop.clear()
        # CPython hands the interpreter lock to a waiting thread only when the holder blocks or after the switch
        # interval (5 ms by default): with eight stage threads a frame's hand-over to the next stage can sit behind a
        # neighbour stage's pure-Python stretch for that long.  A shorter interval while the pipeline runs bounds it
        # (MONOPORT_STAGE_SWITCH_INTERVAL seconds, 0 = leave the interpreter's setting alone).
        switch_old = sys.getswitchinterval()
        if SWITCH_INTERVAL > 0:
            sys.setswitchinterval(SWITCH_INTERVAL)
        for t in self._threads:
            t.start()
        finished = False
        try:
            while True:
                item = queues[-1].get()
                if item is _END:
                    finished = True
                    break
                payload, event = item
                slots.release()
                if isinstance(payload, StageError):
                    payload.reraise()
                if event is not None:
                    # the consumer reads the result on its own current stream
                    cur = torch.cuda.current_stream(self.device)
                    cur.wait_event(event)
                    _record_streams(payload, cur)
                yield payload
        finally:
            if not finished:
                # an error was re-raised above, or the consumer stopped iterating: shut the pipeline down instead of
                # leaving its threads parked on their queues (the reference's loader dies with the process,
                # RTL/dataloader.py:909-914).  The feeder stops admitting, the frames already inside run through and
                # are dropped here, the end marker follows them down the stages.
                self._stop.set()
                for _ in range(self.max_in_flight + 1):
                    slots.release()
                try:
                    while queues[-1].get(timeout=30) is not _END:
                        slots.release()
                except queue.Empty:
                    pass
            for t in self._threads:
                t.join(timeout=5)
            if SWITCH_INTERVAL > 0:
                sys.setswitchinterval(switch_old)
