"""Prepared-database cache for the drop-in CLI (round 5).

The reference's `inference.py:56-65` shells out to `GestureKNN.py` once per clip, and every invocation of
`main_codebook` (GestureKNN.py:816-845) re-reads the whole speaker database from its .npz files, re-resamples the
WavLM track and rebuilds every derived table - 1.2 s around a match that takes 0.3 ms here.  What a GestureDB
holds after its constructor is a pure function of the five database-side files and the build options, so it is
written ONCE next to nothing else: one flat file per (files' sizes + mtimes, options, library version) key,

    [8-byte magic][8-byte header length][JSON header][padding to 4 KiB][tensor bytes, each 4 KiB aligned]

and a later invocation restores the object from it without launching a single build kernel: the JSON header carries
every scalar / list attribute and a table (name, dtype, shape, offset, where) of the tensors; device tensors are
streamed file -> pinned staging ring -> HBM by reader threads (`os.preadv` releases the GIL) and asynchronous
copies on one HIP stream, so the page-cache read of chunk i + 1 overlaps the H2D copy of chunk i.

Nothing here touches the matching path: a restored GestureDB is the same object the constructor would have built
(tests/test_gpu_db_cache.py compares every tensor bit for bit and the CLI's output bytes).
"""
import hashlib
import json
import os
import threading

import numpy as np
import torch

MAGIC = b"QPGDB\x00\x05\x00"
ALIGN = 4096
CHUNK = 32 << 20                  # bytes per staging slot
SLOTS = 6
VERSION = 5                       # bump when GestureDB's attributes change meaning

_DT = {"float32": torch.float32, "float64": torch.float64, "float16": torch.float16, "int16": torch.int16,
       "int32": torch.int32, "int64": torch.int64, "uint8": torch.uint8, "bool": torch.bool}


def file_key(paths, options):
    """Cache key: every file's absolute path, size and mtime (ns) + the build options + the cache and library versions."""
    from . import _lib
    h = hashlib.sha256()
    for p in paths:
        st = os.stat(p)
        h.update(("%s|%d|%d;" % (os.path.abspath(p), st.st_size, st.st_mtime_ns)).encode())
    h.update(json.dumps(options, sort_keys=True).encode())
    h.update(("v%d|lib%d" % (VERSION, int(_lib.load().qpg_version()))).encode())
    return h.hexdigest()[:24]


def default_dir():
    return os.environ.get("QPG_DB_CACHE_DIR") or os.path.join(os.environ.get("XDG_CACHE_HOME") or
                                                              os.path.expanduser("~/.cache"), "qpgesture_amd")


def cache_path(key, directory=None):
    return os.path.join(directory or default_dir(), "db_%s.qpgdb" % key)


# ---- object <-> (json, tensors) ------------------------------------------------------------------------------------
def _jsonable(v):
    try:
        json.dumps(v)
        return True
    except (TypeError, ValueError):
        return False


def _flatten(obj, prefix, tensors, seen):
    """JSON-able description of `obj`'s attributes; tensors / arrays are moved to `tensors` under prefixed names."""
    out = {}
    for k, v in obj.__dict__.items():
        name = prefix + k
        if isinstance(v, torch.Tensor):
            key = (v.data_ptr(), tuple(v.shape), str(v.dtype), str(v.device))
            if key in seen:                               # (aliases: txt_cidx is txt_r)
                out[k] = {"__alias__": seen[key]}
                continue
            seen[key] = name
            tensors[name] = v
            out[k] = {"__tensor__": name}
        elif isinstance(v, np.ndarray):
            tensors[name] = v
            out[k] = {"__array__": name}
        elif isinstance(v, torch.device):
            out[k] = {"__device__": True}
        elif isinstance(v, (bool, int, float, str, type(None))):
            out[k] = v
        elif isinstance(v, (np.integer, np.floating)):
            out[k] = v.item()
        elif isinstance(v, (list, tuple)) and all(isinstance(x, (bool, int, float, str, np.integer, np.floating)) for x in v):
            out[k] = {"__list__": [x.item() if isinstance(x, (np.integer, np.floating)) else x for x in v],
                      "tuple": isinstance(v, tuple)}
        elif isinstance(v, dict) and _jsonable(v):
            out[k] = {"__dict__": v}
        elif hasattr(v, "__dict__") and type(v).__module__.startswith("qpgesture_amd"):
            out[k] = {"__object__": type(v).__module__ + ":" + type(v).__name__,
                      "attrs": _flatten(v, name + ".", tensors, seen)}
        else:
            raise TypeError("db_cache: attribute %s of type %s is not serialisable" % (name, type(v)))
    return out


def _restore(cls, desc, tensors, device):
    import importlib
    obj = object.__new__(cls)
    for k, v in desc.items():
        if isinstance(v, dict) and "__tensor__" in v:
            val = tensors[v["__tensor__"]]
        elif isinstance(v, dict) and "__alias__" in v:
            val = tensors[v["__alias__"]]
        elif isinstance(v, dict) and "__array__" in v:
            val = tensors[v["__array__"]]
        elif isinstance(v, dict) and "__device__" in v:
            val = device
        elif isinstance(v, dict) and "__list__" in v:
            val = tuple(v["__list__"]) if v["tuple"] else list(v["__list__"])
        elif isinstance(v, dict) and "__dict__" in v:
            val = dict(v["__dict__"])
        elif isinstance(v, dict) and "__object__" in v:
            mod, name = v["__object__"].split(":")
            val = _restore(getattr(importlib.import_module(mod), name), v["attrs"], tensors, device)
        else:
            val = v
        obj.__dict__[k] = val
    return obj


# ---- file I/O --------------------------------------------------------------------------------------------------------
def save(db, path, key=""):
    """Write `db` (a GestureDB) to `path` atomically (tmp + rename).  Device tensors are copied out in chunks."""
    tensors, seen = {}, {}
    desc = _flatten(db, "", tensors, seen)
    table, off = [], 0
    for name, t in tensors.items():
        if isinstance(t, torch.Tensor):
            assert t.is_contiguous(), name
            nbytes, dt, where = t.numel() * t.element_size(), str(t.dtype).split(".")[1], ("device" if t.is_cuda else "host_t")
        else:
            t = tensors[name] = np.ascontiguousarray(t)
            nbytes, dt, where = t.nbytes, str(t.dtype), "host"
        table.append({"name": name, "dtype": dt, "shape": list(t.shape), "offset": off, "nbytes": nbytes, "where": where})
        off += (nbytes + ALIGN - 1) // ALIGN * ALIGN
    head = json.dumps({"key": key, "version": VERSION, "class": "qpgesture_amd.code_knn:GestureDB", "attrs": desc,
                       "tensors": table, "data_bytes": off}).encode()
    data0 = (16 + len(head) + ALIGN - 1) // ALIGN * ALIGN
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    tmp = "%s.tmp.%d" % (path, os.getpid())
    with open(tmp, "wb") as f:
        f.write(MAGIC)
        f.write(len(head).to_bytes(8, "little"))
        f.write(head)
        for ent in table:
            f.seek(data0 + ent["offset"])
            t = tensors[ent["name"]]
            if isinstance(t, np.ndarray):
                f.write(memoryview(t).cast("B"))
                continue
            flat = t.reshape(-1).view(torch.uint8) if t.dtype != torch.bool else t.reshape(-1).to(torch.uint8)
            for o in range(0, flat.numel(), 256 << 20):                  # (bounded host copies of a 1.5 GB array)
                f.write(memoryview(flat[o:o + (256 << 20)].cpu().numpy()))
        f.truncate(data0 + off)
    os.replace(tmp, path)
    return path


class _Stager:
    """Pinned staging ring shared by the loads of one process."""
    _inst = None

    def __init__(self):
        self.buf = torch.empty((SLOTS, CHUNK), dtype=torch.uint8).pin_memory()
        self.np = self.buf.numpy()
        self.free = [torch.cuda.Event() for _ in range(SLOTS)]

    @classmethod
    def get(cls):
        if cls._inst is None:
            cls._inst = cls()
        return cls._inst


def load(path, device, expect_key=None, n_readers=4):
    """Restore the GestureDB written by save(); None if the file is missing, foreign or keyed differently."""
    dev = torch.device(device)
    try:
        fd = os.open(path, os.O_RDONLY)
    except OSError:
        return None
    try:
        pre = os.pread(fd, 16, 0)
        if len(pre) != 16 or pre[:8] != MAGIC:
            return None
        hl = int.from_bytes(pre[8:], "little")
        head = json.loads(os.pread(fd, hl, 16).decode())
        if head.get("version") != VERSION or (expect_key is not None and head.get("key") != expect_key):
            return None
        data0 = (16 + hl + ALIGN - 1) // ALIGN * ALIGN
        if os.fstat(fd).st_size < data0 + head["data_bytes"]:
            return None
        tensors, jobs = {}, []
        for ent in head["tensors"]:
            shape, nb = tuple(ent["shape"]), ent["nbytes"]
            if ent["where"] == "host":
                a = np.empty(shape, np.dtype(ent["dtype"]))
                if nb:
                    os.preadv(fd, [memoryview(a).cast("B")], data0 + ent["offset"])
                tensors[ent["name"]] = a
            elif ent["where"] == "host_t":
                t = torch.empty(shape, dtype=_DT[ent["dtype"]])
                if nb:
                    os.preadv(fd, [memoryview(t.numpy()).cast("B")], data0 + ent["offset"])
                tensors[ent["name"]] = t
            else:
                t = torch.empty(shape, dtype=_DT[ent["dtype"]], device=dev)
                tensors[ent["name"]] = t
                flat = t.reshape(-1).view(torch.uint8) if t.dtype != torch.bool else None
                if flat is None:
                    raise TypeError("bool device tensors are not cached")
                for o in range(0, nb, CHUNK):
                    jobs.append((flat, o, min(CHUNK, nb - o), data0 + ent["offset"] + o))
        _stream_in(fd, jobs, dev, n_readers)
    finally:
        os.close(fd)
    import importlib
    mod, name = head["class"].split(":")
    return _restore(getattr(importlib.import_module(mod), name), head["attrs"], tensors, dev)


def _stream_in(fd, jobs, dev, n_readers):
    """jobs: (flat device byte tensor, offset in it, bytes, file offset).  Reader threads fill pinned slots (preadv: no GIL),
    the calling thread issues the H2D copies on a private stream and recycles a slot when its copy's event has passed."""
    if not jobs:
        return
    st = _Stager.get()
    copy_stream = torch.cuda.Stream(dev)
    lock = threading.Condition()
    state = {"next": 0, "filled": {}, "slot_free": [True] * SLOTS, "err": None}

    def reader():
        while True:
            with lock:
                while True:
                    if state["err"] is not None or state["next"] >= len(jobs):
                        return
                    slot = next((i for i in range(SLOTS) if state["slot_free"][i]), None)
                    if slot is not None:
                        break
                    lock.wait()
                j = state["next"]
                state["next"] += 1
                state["slot_free"][slot] = False
            try:
                _, _, nb, foff = jobs[j]
                got = os.preadv(fd, [memoryview(st.np[slot])[:nb]], foff)
                if got != nb:
                    raise IOError("short read")
            except Exception as e:                      # noqa: BLE001
                with lock:
                    state["err"] = e
                    lock.notify_all()
                return
            with lock:
                state["filled"][j] = slot
                lock.notify_all()

    threads = [threading.Thread(target=reader, daemon=True) for _ in range(max(1, n_readers))]
    for t in threads:
        t.start()
    pending = []                                         # (slot, event) of copies in flight
    with torch.cuda.stream(copy_stream):
        for j in range(len(jobs)):
            with lock:
                while j not in state["filled"] and state["err"] is None:
                    # recycle finished slots while waiting
                    lock.wait(timeout=0.0005)
                    for s_, ev in list(pending):
                        if ev.query():
                            pending.remove((s_, ev))
                            state["slot_free"][s_] = True
                            lock.notify_all()
                if state["err"] is not None:
                    break
                slot = state["filled"].pop(j)
            flat, o, nb, _ = jobs[j]
            flat[o:o + nb].copy_(st.buf[slot, :nb], non_blocking=True)
            ev = st.free[slot]
            ev.record(copy_stream)
            pending.append((slot, ev))
            with lock:
                for s_, e_ in list(pending):
                    if e_.query():
                        pending.remove((s_, e_))
                        state["slot_free"][s_] = True
                lock.notify_all()
    copy_stream.synchronize()
    with lock:
        state["slot_free"] = [True] * SLOTS
        lock.notify_all()
    for t in threads:
        t.join()
    if state["err"] is not None:
        raise state["err"]
    torch.cuda.current_stream(dev).wait_stream(copy_stream)
