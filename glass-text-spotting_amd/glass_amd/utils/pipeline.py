"""Host-side software pipelining of inference steps.

A step has three places where the host must read a few integers back from the GPU before it can size the next
stage (detection counts -> recognizer batch; surviving counts -> Instances views; word counts / characters).
Run one step at a time and the GPU idles ~2 ms of 42 around those reads while Python refills the launch queue,
and the latency-bound tail kernels (NMS, finalize, word post-processing) run alone on the chip.

The step functions are therefore written as generators that `yield ReadBack(tensor, ...)` where they used to call
`.cpu()`: `drive()` runs one synchronously (the plain API), `run_pipelined()` keeps `depth` steps in flight, each on
its own HIP stream, advancing them round-robin one segment at a time - while the host waits for step k's
read-back, step k+1's kernels are already queued, and k's tail overlaps k+1's backbone.
"""
from __future__ import annotations

import contextlib
from collections import deque
from typing import Callable, Generator, Iterable, List, Optional

import torch


_STREAMS: dict = {}


class ReadBack:
    """Request to copy small device tensors to the host.  `start()` enqueues the copies (pinned memory,
    non-blocking) and an event on the current stream; `wait()` blocks on that event and returns the host tensors."""

    def __init__(self, *tensors: torch.Tensor):
        self.tensors = tensors
        self.host: Optional[List[torch.Tensor]] = None
        self.event: Optional[torch.cuda.Event] = None

    def start(self) -> "ReadBack":
        self.host = []
        for t in self.tensors:
            if t.is_cuda:
                h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                h.copy_(t, non_blocking=True)
            else:
                h = t
            self.host.append(h)
        if any(t.is_cuda for t in self.tensors):
            self.event = torch.cuda.Event(blocking=True)      # let the host thread sleep, not spin, while it waits
            self.event.record()
        return self

    def wait(self) -> List[torch.Tensor]:
        if self.event is not None:
            self.event.synchronize()
        return self.host


class StepOutput(list):
    """The list a step generator returns (`list[{'instances': Instances}]` / `list[Instances]`, the reference's return
    types) carrying that step's padded device-resident tensors as attributes (`.batch`: BatchedDetections of the
    meta-arch, `.words`: the word post-processor's padded outputs).  Per-step state travels WITH the result: with
    several steps in flight a `model.last_*` attribute is overwritten by whichever step ran last."""
    batch = None
    words = None


def segment_scoped(gen: Generator, enter: Callable, leave: Callable):
    """Run `gen` with `tok = enter()` ... `leave(tok)` around EVERY segment (the code between two yields) instead of
    around the whole generator: process-global settings (conv precision) must not stay switched while the other
    in-flight steps of `run_pipelined` execute their segments."""
    val, exc, first = None, None, True
    while True:
        tok = enter()
        try:
            if exc is not None:
                req = gen.throw(exc)                 # an exception thrown in at our yield belongs to the step body
            else:
                req = next(gen) if first else gen.send(val)
        except StopIteration as e:
            return e.value
        finally:
            leave(tok)
        first, exc = False, None
        try:
            val = yield req
        except GeneratorExit:
            # closed at a yield (a failing step of run_pipelined, a caller that gives up): run the body's `finally`
            # blocks now, inside the scope, not whenever the garbage collector finds the generator
            tok = enter()
            try:
                gen.close()
            finally:
                leave(tok)
            raise
        except BaseException as e:                   # noqa: BLE001 - forwarded, not handled
            exc = e


def _record_stream(obj, stream, depth: int = 0) -> None:
    """caching-allocator hygiene for results that leave their producing stream: the blocks were allocated from the
    side stream's pool and are consumed on `stream`; without record_stream they could be recycled by the next step on
    the side stream while kernels queued on `stream` still read them."""
    if isinstance(obj, torch.Tensor):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, dict):
        for v in obj.values():
            _record_stream(v, stream, depth + 1)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _record_stream(v, stream, depth + 1)
        for name in ("batch", "words"):
            if getattr(obj, name, None) is not None:
                _record_stream(getattr(obj, name), stream, depth + 1)
    elif depth < 6 and hasattr(obj, "__dict__"):
        for v in vars(obj).values():
            _record_stream(v, stream, depth + 1)


class _HostStream:
    """stand-in for a HIP stream when the schedule runs on host tensors only"""
    def wait_stream(self, other) -> None:
        pass


def _on(stream):
    return contextlib.nullcontext() if isinstance(stream, _HostStream) else torch.cuda.stream(stream)


def drive(gen: Generator):
    """Run a step generator to completion, serving each read-back immediately (the synchronous API)."""
    # grad mode is thread-global state: it is set around every segment here (a `with torch.no_grad()` that spans a
    # `yield` inside the generator would leak into / be clobbered by the other in-flight steps)
    try:
        with torch.no_grad():
            req = next(gen)
        while True:
            host = req.start().wait()
            with torch.no_grad():
                req = gen.send(host)
    except StopIteration as e:
        return e.value


def run_pipelined(make_steps: Iterable[Callable[[], Generator]], depth: int = 2, device=None) -> list:
    """Run the step generators produced by `make_steps` with up to `depth` in flight; returns their results in
    order.  Every step runs under its own stream (round-robin over `depth` streams); the host work of the
    in-flight steps is interleaved segment by segment in a fixed order, so ranks of a multi-GPU job that run the
    same schedule also issue their collectives in the same order."""
    if depth <= 1:
        return [drive(mk()) for mk in make_steps]
    # HIP multiplexes the streams of one priority class onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default).
    # With the null stream and the local extractor's two side streams already in the normal class, two more
    # normal-priority streams end up sharing a hardware queue with one of them and the in-flight steps serialise
    # where they should overlap (measured: 200 images/s with both normal vs 210 with the classes alternated or with
    # GPU_MAX_HW_QUEUES=8; 204 unpipelined).  The alternation is about queue placement, not about urgency.
    # The streams are created once per (device, depth): the caching allocator keeps one pool per stream, so fresh
    # streams on every call would start with cold pools (hipMalloc stalls in the first steps, memory growing per call).
    on_gpu = torch.cuda.is_available() and (device is None or torch.device(device).type == "cuda")
    key = (str(device), depth)
    if not on_gpu:
        # host-only use (the gloo tests of the step schedule): same interleaving, no streams
        streams, main = [_HostStream() for _ in range(depth)], _HostStream()
    else:
        if key not in _STREAMS:
            _STREAMS[key] = [torch.cuda.Stream(device=device, priority=(-1 if i % 2 == 0 else 0)) for i in range(depth)]
        streams = _STREAMS[key]
        main = torch.cuda.current_stream(device)
    todo = deque(enumerate(make_steps))
    active = deque()                 # [index, generator, stream, pending ReadBack | None]
    results = {}

    def advance(slot) -> bool:
        idx, gen, st, req = slot
        with _on(st), torch.no_grad():
            try:
                nxt = next(gen) if req is None else gen.send(req.wait())
                slot[3] = nxt.start()
                return True
            except StopIteration as e:
                results[idx] = e.value
                if not isinstance(main, _HostStream):
                    _record_stream(e.value, main)
                return False

    while todo or active:
        while todo and len(active) < depth:
            idx, mk = todo.popleft()
            st = streams[idx % depth]
            st.wait_stream(main)                      # inputs prepared on the caller's stream
            slot = [idx, None, st, None]
            with _on(st):
                slot[1] = mk()
            if advance(slot):
                active.append(slot)
            else:
                main.wait_stream(st)
        if active:
            slot = active.popleft()
            if advance(slot):
                active.append(slot)
            else:
                main.wait_stream(slot[2])
    return [results[i] for i in sorted(results)]
