"""Launch plan of a forward: the ordered C-ABI calls of one `Generator.forward`, recorded once and replayed.

`Generator._forward_hip` decides a forward's schedule - which kernel serves which layer (asked of the library's shape queries), which
buffers it reads and writes, which launches go to the side stream and where the two streams meet.  None of that depends on the values of
the inputs: for a given (shape, mode, precision, algorithm, fusion switches, parameter storage) the same calls go out with the same
argument structs every time; only four pointers change - `x`, `spk_emb`, `noise` and the freshly allocated output.  So the first forward of
a configuration runs through the planner with a `Recorder` in front of the library: every launching entry point it calls (the ones whose last
argument is the stream; the host-only shape queries pass through) is appended to a `Tape` with its argument objects, together with the
stream operations between them.  Every later forward of that configuration is `Tape.replay`: a loop of `fn(*args, stream)` over prebuilt
ctypes structs - no shape queries, no struct construction, no Python per layer (vec2wav/train.py:246-291 calls the generator once per
validation utterance; at B = 1 the forward is launch-bound).

A tape owns nothing but references: the structs it replays point into buffers of the module that recorded it (`Generator._ws`, `_wts`,
fold plans, the split-over-C_in slab), and the module drops its tapes whenever it replaces one of those buffers."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import torch

from . import _hip

K_CALL, K_FORK, K_JOIN, K_MARK, K_NEED, K_PY = range(6)
MAIN, SIDE, SIDE2 = 0, 1, 2        # SIDE2: the conditioning chain's own queue in the bf16-storage train-mode plan (forward_plan.weights)


class Step:
    __slots__ = ('kind', 'fn', 'args', 'sid', 'tag', 'name')

    def __init__(self, kind, fn=None, args=None, sid=MAIN, tag=None, name=None):
        self.kind, self.fn, self.args, self.sid, self.tag, self.name = kind, fn, args, sid, tag, name


def _struct_slots(obj, out, path=()):
    """Every pointer-valued slot of a ctypes struct / array (nested ones too) as (container, key) pairs."""
    if isinstance(obj, C.Array):
        if issubclass(obj._type_, (C.Structure, C.Array)):
            for i in range(len(obj)):
                _struct_slots(obj[i], out)
        elif obj._type_ is C.c_void_p:
            for i in range(len(obj)):
                out.append((obj, i))
        return
    if isinstance(obj, C.Structure):
        for name, typ in obj._fields_:
            if typ is C.c_void_p:
                out.append((obj, name))
            elif isinstance(typ, type) and issubclass(typ, (C.Structure, C.Array)):
                _struct_slots(getattr(obj, name), out)


class TapeNotOwned(RuntimeError):
    pass


def _in_ranges(ranges, v):
    import bisect
    i = bisect.bisect_right(ranges, (v, float('inf'))) - 1
    return i >= 0 and ranges[i][0] <= v < ranges[i][1]


def owned_ranges(tensors):
    """Sorted, merged [lo, hi) device address ranges of `tensors` (storage extents)."""
    spans = []
    for t in tensors:
        if t is None or not t.is_cuda:
            continue
        st = t.untyped_storage()
        if st.nbytes():
            spans.append((st.data_ptr(), st.data_ptr() + st.nbytes()))
    spans.sort()
    out = []
    for lo, hi in spans:
        if out and lo <= out[-1][1]:
            out[-1] = (out[-1][0], max(out[-1][1], hi))
        else:
            out.append((lo, hi))
    return out


def tensors_in(obj, depth=4, seen=None):
    """Every torch.Tensor reachable from `obj` through dicts, sequences and object attributes (the fold caches hold plans whose descriptor
    tables are tensors)."""
    seen = set() if seen is None else seen
    if id(obj) in seen or depth < 0:
        return
    seen.add(id(obj))
    if isinstance(obj, torch.Tensor):
        yield obj
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from tensors_in(v, depth - 1, seen)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            yield from tensors_in(v, depth - 1, seen)
    elif hasattr(obj, '__dict__') and not isinstance(obj, type):
        yield from tensors_in(vars(obj), depth - 1, seen)


class Tape:
    """The recorded schedule.  `binds`: name -> device address at record time of the tensors that differ from call to call."""

    def __init__(self):
        self.steps: List[Step] = []
        self.keep: list = []                 # objects the recorded structs point into and nobody else may hold (plans, temporaries)
        self.patches: Dict[str, list] = {}   # bind name -> [(container, key)] slots holding that address
        self.events: Dict[str, torch.cuda.Event] = {}
        self.out = None                      # (shape, dtype) of the output tensor
        self.launches = 0

    def finalize(self, binds: Dict[str, int], owned=None):
        """`owned` (optional): sorted, non-overlapping [lo, hi) address ranges of everything the recording module keeps alive (parameters, buffers,
        workspaces, folded weights, slabs, plan tables).  Every pointer-valued struct slot of a recorded call must then lie in one of them or be a
        bind: a wrapper that allocated a scratch tensor inside the planned forward would otherwise leave a dangling address in the tape
        (ADVICE r05).  Raises TapeNotOwned - the caller keeps planning that configuration instead of replaying it."""
        by_addr = {}
        for name, addr in binds.items():
            if addr in by_addr:
                raise RuntimeError(f'schedule: {name} and {by_addr[addr]} share an address')
            by_addr[addr] = name
        self.patches = {name: [] for name in binds}
        for st in self.steps:
            if st.kind != K_CALL:
                continue
            self.launches += 1
            for i, a in enumerate(st.args):
                if isinstance(a, int):
                    if a in by_addr:
                        self.patches[by_addr[a]].append((st.args, i))
                    continue
                obj = getattr(a, '_obj', a)          # C.byref(struct) keeps its struct in ._obj
                slots = []
                _struct_slots(obj, slots)
                for cont, key in slots:
                    v = cont[key] if isinstance(key, int) else getattr(cont, key)
                    if v in by_addr:
                        self.patches[by_addr[v]].append((cont, key))
                    elif v and owned is not None and not _in_ranges(owned, v):
                        raise TapeNotOwned(f'schedule: {st.name} was recorded with a pointer ({key} = {v:#x}) into memory the module does not own')
        for name in binds:
            if name != 'y' and not self.patches[name]:
                raise RuntimeError(f'schedule: no recorded call reads {name}')

    def kernel_names(self) -> Dict[str, list]:
        """tag -> names of the kernels the tagged main-stream calls of this plan launch, in order (asked of the library call by call through
        a v2w_name_sink: host-only, nothing runs).  bench.py labels its per-launch timings with these."""
        out: Dict[str, list] = {}
        for st in self.steps:
            if st.kind == K_CALL and st.tag is not None:
                _rc, names = _hip.kernel_names(st.fn, *st.args)
                out.setdefault(st.tag, []).extend(names)
        return out

    def replay(self, main, side, binds: Dict[str, int], profile: Optional[list] = None, side2=None):
        for name, addr in binds.items():
            for cont, key in self.patches[name]:
                if isinstance(key, int):
                    cont[key] = addr
                else:
                    setattr(cont, key, addr)
        streams = (main, side, side if side2 is None else side2)
        handles = tuple(q.cuda_stream for q in streams)
        open_tag, e0 = None, None
        for st in self.steps:
            k = st.kind
            if profile is not None and st.tag != open_tag:
                if open_tag is not None:
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record(main)
                    profile.append((open_tag, e0, e1))
                open_tag = st.tag
                if open_tag is not None:
                    e0 = torch.cuda.Event(enable_timing=True)
                    e0.record(main)
            if k == K_CALL:
                rc = st.fn(*st.args, handles[st.sid])
                if rc != 0:
                    _hip.check(rc, st.name)
            elif k == K_NEED:
                main.wait_event(self.events[st.name])
            elif k == K_MARK:
                self.events[st.name].record(streams[st.sid])
            elif k == K_FORK:
                for q in streams[1:1 + st.args]:
                    q.wait_stream(main)
            elif k == K_JOIN:
                for q in streams[1:1 + st.args]:
                    main.wait_stream(q)
            else:
                st.fn()
        if profile is not None and open_tag is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record(main)
            profile.append((open_tag, e0, e1))


class NameProbe:
    """Stands in front of the library while ONE tagged step of a profiled, planned forward runs (forward_plan.ForwardPlanner.timed): every
    launching call is made, and asked for its kernel names through a v2w_name_sink."""

    def __init__(self, lib, names: list):
        self._lib, self._names = lib, names

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if name not in _hip.LAUNCHERS:
            return fn

        def call(*args):
            rc = fn(*args)
            if rc == 0:
                self._names.extend(_hip.kernel_names(fn, *args[:-1])[1])
            return rc
        return call


class Recorder:
    """Stands in front of the library while a planner runs: launching calls that succeed are taped (and executed), queries pass through."""

    def __init__(self, lib, main, side, side2=None):
        self._lib = lib
        self.tape = Tape()
        self.main, self.side, self.side2 = main, side, side if side2 is None else side2
        self._sid = {self.side2.cuda_stream: SIDE2, side.cuda_stream: SIDE, main.cuda_stream: MAIN}
        self.nfork = 1
        self.tag = None

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if name not in _hip.LAUNCHERS:
            return fn

        def call(*args):
            rc = fn(*args)
            if rc == 0:
                h = args[-1]
                h = h.value if isinstance(h, C.c_void_p) else h
                sid = self._sid.get(h)
                if sid is None:
                    raise RuntimeError(f'schedule: {name} launched on a stream the plan does not know')
                self.tape.steps.append(Step(K_CALL, fn, list(args[:-1]), sid, self.tag if sid == MAIN else None, name))
            return rc
        return call

    # ---- stream operations of the plan (executed now, taped for the replays)
    def stream(self, sid):
        return (self.main, self.side, self.side2)[sid]

    def fork(self, sides=1):
        """The first `sides` side streams wait for the main stream (and `join` brings the same ones back)."""
        self.nfork = sides
        for q in (self.side, self.side2)[:sides]:
            q.wait_stream(self.main)
        self.tape.steps.append(Step(K_FORK, args=sides))

    def join(self):
        for q in (self.side, self.side2)[:self.nfork]:
            self.main.wait_stream(q)
        self.tape.steps.append(Step(K_JOIN, args=self.nfork))

    def mark(self, name, sid=SIDE):
        ev = self.tape.events.get(name)
        if ev is None:
            ev = self.tape.events[name] = torch.cuda.Event()
        ev.record(self.stream(sid))
        self.tape.steps.append(Step(K_MARK, name=name, sid=sid))

    def need(self, name):
        self.main.wait_event(self.tape.events[name])
        self.tape.steps.append(Step(K_NEED, name=name))

    def py(self, fn, tag=None):
        """A step that is not a C-ABI launch (the statistics all-reduce of a data-parallel run): called now and at every replay."""
        self.tape.steps.append(Step(K_PY, fn, tag=tag))
        return fn()

    def keep(self, obj):
        self.tape.keep.append(obj)
