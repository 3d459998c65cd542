"""processors=[...] semantics of the stage pipeline (RTL/dataloader.py:734-751, :1026-1053): FIFO
order, one thread per stage, exception forwarding, bounded in-flight frames.  CPU only."""
import threading
import time

import pytest

from monoport_amd.stage_pipeline import StagePipeline


def test_fifo_order_and_dict_passing():
    procs = [lambda d: {"x": d},
             lambda d: {**d, "y": d["x"] * 2},
             lambda d: {**d, "z": d["y"] + 1}]
    out = list(StagePipeline(range(20), procs))
    assert [o["z"] for o in out] == [2 * i + 1 for i in range(20)]


def test_each_stage_has_its_own_thread_and_frames_overlap():
    seen = [set(), set()]
    active = [0]
    peak = [0]
    lock = threading.Lock()

    def stage(k):
        def fn(x):
            seen[k].add(threading.get_ident())
            with lock:
                active[0] += 1
                peak[0] = max(peak[0], active[0])
            time.sleep(0.01)
            with lock:
                active[0] -= 1
            return x
        return fn

    list(StagePipeline(range(10), [stage(0), stage(1)], max_in_flight=2))
    assert len(seen[0]) == 1 and len(seen[1]) == 1 and seen[0] != seen[1]
    assert peak[0] == 2  # two frames in two different stages at once


def test_in_flight_is_bounded():
    entered = []
    gate = threading.Event()

    def slow(x):
        entered.append(x)
        gate.wait(2)
        return x

    admitted = []

    def source():
        for i in range(10):
            admitted.append(i)
            yield i

    results = []
    consumer = threading.Thread(
        target=lambda: results.extend(StagePipeline(source(), [slow, lambda x: x], max_in_flight=3)))
    consumer.start()
    time.sleep(0.3)
    assert entered == [0]        # stage 0 holds frame 0 ...
    assert len(admitted) <= 4    # ... frames 1, 2 are queued, the feeder blocks on the 4th
    gate.set()
    consumer.join(10)
    assert results == list(range(10))


def test_exception_is_forwarded_to_consumer():
    def boom(x):
        if x == 3:
            raise ValueError("frame 3 is bad")
        return x

    got = []
    with pytest.raises(RuntimeError, match="stage 1 failed: ValueError: frame 3 is bad"):
        for v in StagePipeline(range(6), [lambda x: x, boom, lambda x: x]):
            got.append(v)
    assert got == [0, 1, 2]


def test_none_results_pass_through():
    """Empty reconstructions propagate None through later stages (RTL/recon.py:32-33)."""
    out = list(StagePipeline(range(4), [lambda x: None if x % 2 else x,
                                        lambda v: None if v is None else v + 10]))
    assert out == [10, None, 12, None]


def test_coalesced_stage_serves_queued_frames_together_in_order():
    """A Coalesced processor takes whatever is already queued in front of its stage (up to max_batch)
    in ONE fn_many call; results leave in submission order; a frame that arrives alone is served by
    fn; an exception inside fn_many is re-tried frame by frame and lands on the frame that caused it."""
    import threading
    import time
    from monoport_amd.stage_pipeline import Coalesced, StagePipeline
    gate = threading.Event()
    sizes, singles = [], []

    def slow_first(x):  # holds the first frame until the others have queued up behind it
        if x == 0:
            gate.wait(5)
        return x

    def one(x):
        singles.append(x)
        if x == 5:
            raise ValueError("frame five")
        return x * 10

    def many(xs):
        sizes.append(len(xs))
        if 5 in xs:
            raise ValueError("somewhere in the batch")
        return [x * 10 for x in xs]

    def source():
        for i in range(8):
            yield i
        time.sleep(0.2)
        gate.set()

    # stage 0 parks frame 0; frames 1.. cannot overtake it (FIFO), so they queue up in front of stage 0
    # ... and, once released, arrive at stage 1 in a burst
    pipe = StagePipeline(source(), [Coalesced(slow_first, lambda xs: [slow_first(x) for x in xs], 8),
                                    Coalesced(one, many, max_batch=4)], device=None, max_in_flight=8)
    got, err = [], None
    try:
        for v in pipe:
            got.append(v)
    except RuntimeError as e:
        err = e
    assert got == [0, 10, 20, 30, 40]  # FIFO, up to the failing frame
    assert err is not None and "frame five" in str(err) and isinstance(err.__cause__, ValueError)
    assert max(sizes) > 1 and max(sizes) <= 4  # frames were served together


def test_stage_threads_inherit_the_consumers_grad_mode():
    """torch's grad mode is thread-local; the stage threads take the mode of the thread that iterates the
    pipeline, so `with torch.no_grad():` around the consuming loop reaches netG.filter in its stage."""
    import torch
    from monoport_amd.stage_pipeline import StagePipeline
    seen = []
    pipe = lambda: StagePipeline(iter(range(3)), [lambda x: (seen.append(torch.is_grad_enabled()), x)[1]], device=None)
    with torch.no_grad():
        assert list(pipe()) == [0, 1, 2]
    assert seen == [False] * 3
    seen.clear()
    assert list(pipe()) == [0, 1, 2] and seen == [True] * 3


def test_pipeline_shuts_down_after_an_error_or_an_early_exit():
    """An exception re-raised in the consumer (RTL/dataloader.py:909-914) or a consumer that stops iterating must not
    leave the feeder and the stage threads parked on their queues: the feeder stops admitting from an ENDLESS
    source, the frames inside run through and the end marker follows them (round 6: the soak leg's requirement)."""
    import itertools
    from monoport_amd.stage_pipeline import StagePipeline

    def boom(x):
        if x == 7:
            raise ValueError("frame seven")
        return x

    pipe = StagePipeline(itertools.count(), [lambda x: x, boom, lambda x: x], device=None, max_in_flight=4)
    got = []
    with pytest.raises(RuntimeError, match="stage 1 failed: ValueError: frame seven"):
        for v in pipe:
            got.append(v)
    assert got == list(range(7))
    assert pipe._threads and not any(t.is_alive() for t in pipe._threads)

    pipe = StagePipeline(itertools.count(), [lambda x: x + 1], device=None, max_in_flight=3)
    for v in pipe:
        if v == 5:
            break
    time.sleep(0.05)
    for t in pipe._threads:
        t.join(timeout=5)
    assert not any(t.is_alive() for t in pipe._threads)
