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


def sources_id(paths, options):
    """What a cache file was built FOR, without the files' sizes / mtimes: a later save() for the same sources (the files
    were regenerated) replaces the older cache files instead of leaving them behind (prune)."""
    h = hashlib.sha256()
    for p in paths:
        h.update((os.path.abspath(p) + ";").encode())
    h.update(json.dumps(options, sort_keys=True).encode())
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


def _class_of(spec):
    """'module:Name' of a header -> the class, for THIS package's modules only: the header is data read from a file, and
    importlib.import_module on an arbitrary name runs that module's top level."""
    import importlib
    mod, name = str(spec).split(":")
    if not (mod == "qpgesture_amd" or mod.startswith("qpgesture_amd.")) or not name.isidentifier():
        raise ValueError("cache header names a foreign class: %r" % (spec,))
    cls = getattr(importlib.import_module(mod), name)
    if not isinstance(cls, type):
        raise ValueError("cache header names a non-class: %r" % (spec,))
    return cls


def _restore(cls, desc, tensors, device):
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
            val = _restore(_class_of(v["__object__"]), v["attrs"], tensors, device)
        else:
            val = v
        obj.__dict__[k] = val
    return obj


# ---- file I/O --------------------------------------------------------------------------------------------------------
def save(db, path, key="", sources=None, keep=None):
    """Write `db` (a GestureDB) to `path` atomically (tmp + rename; the tmp file is removed on ANY failure).  Device tensors
    are copied out in chunks.  sources (sources_id): older cache files of the same sources are deleted afterwards, and at
    most `keep` files (default $QPG_DB_CACHE_MAX_FILES or 8, newest first) stay in the directory."""
    tensors, seen = {}, {}
    desc = _flatten(db, "", tensors, seen)
    table, off = [], 0
    for name, t in tensors.items():
        if isinstance(t, torch.Tensor):
            if not t.is_contiguous():
                raise ValueError("db_cache.save: tensor %s is not contiguous" % name)
            nbytes, dt, where = t.numel() * t.element_size(), str(t.dtype).split(".")[1], ("device" if t.is_cuda else "host_t")
        else:
            t = tensors[name] = np.ascontiguousarray(t)
            nbytes, dt, where = t.nbytes, str(t.dtype), "host"
        table.append({"name": name, "dtype": dt, "shape": list(t.shape), "offset": off, "nbytes": nbytes, "where": where})
        off += (nbytes + ALIGN - 1) // ALIGN * ALIGN
    head = json.dumps({"key": key, "version": VERSION, "class": "qpgesture_amd.code_knn:GestureDB", "attrs": desc,
                       "tensors": table, "data_bytes": off, "sources": sources}).encode()
    data0 = (16 + len(head) + ALIGN - 1) // ALIGN * ALIGN
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    tmp = "%s.tmp.%d" % (path, os.getpid())
    try:
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
    finally:
        try:
            os.unlink(tmp)                       # (gone after a successful rename; left behind by anything else)
        except OSError:
            pass
    prune(os.path.dirname(os.path.abspath(path)), path, sources, keep)
    return path


def _read_head(path):
    """The JSON header of a cache file, None for anything that is not one (missing, foreign, truncated, corrupt)."""
    try:
        with open(path, "rb") as f:
            pre = f.read(16)
            if len(pre) != 16 or pre[:8] != MAGIC:
                return None
            hl = int.from_bytes(pre[8:], "little")
            if hl <= 0 or hl > (64 << 20):
                return None
            head = json.loads(f.read(hl).decode())
        return head if isinstance(head, dict) else None
    except (OSError, ValueError, UnicodeDecodeError):
        return None


def prune(directory, keep_path, sources=None, keep=None):
    """Eviction (one file per distinct key was written and never removed; the bench database is ~1.5 GB):
    delete the other db_*.qpgdb files of the same `sources`, then the oldest beyond `keep` files, and stale tmp files."""
    if keep is None:
        keep = int(os.environ.get("QPG_DB_CACHE_MAX_FILES", "8"))
    keep_path = os.path.abspath(keep_path)
    try:
        names = os.listdir(directory)
    except OSError:
        return
    live = []
    for n in names:
        p = os.path.join(directory, n)
        if n.startswith("db_") and ".qpgdb.tmp." in n:
            pid = n.rsplit(".", 1)[-1]
            if not (pid.isdigit() and os.path.exists("/proc/%s" % pid)):
                _unlink(p)
            continue
        if not (n.startswith("db_") and n.endswith(".qpgdb")) or os.path.abspath(p) == keep_path:
            continue
        head = _read_head(p)
        if sources is not None and head is not None and head.get("sources") == sources:
            _unlink(p)
            continue
        try:
            live.append((os.path.getmtime(p), p))
        except OSError:
            pass
    live.sort(reverse=True)
    for _, p in live[max(0, keep - 1):]:
        _unlink(p)


def _unlink(p):
    try:
        os.unlink(p)
    except OSError:
        pass


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
    """Restore the GestureDB written by save(); None if the file is missing, foreign, keyed differently, truncated or its
    header does not parse (the caller then rebuilds the database and overwrites the file) - only I/O and device errors while
    the DATA is streamed raise."""
    dev = torch.device(device)
    head = _read_head(path)
    if head is None:
        return None
    try:
        if head.get("version") != VERSION or (expect_key is not None and head.get("key") != expect_key):
            return None
        cls = _class_of(head["class"])
        plan = []
        for ent in head["tensors"]:
            where, dt = ent["where"], ent["dtype"]
            if where not in ("host", "host_t", "device"):
                raise ValueError(where)
            plan.append((str(ent["name"]), where, np.dtype(dt) if where == "host" else _DT[dt],
                         tuple(int(x) for x in ent["shape"]), int(ent["offset"]), int(ent["nbytes"])))
        data_bytes, attrs = int(head["data_bytes"]), head["attrs"]
        if not isinstance(attrs, dict) or any(o < 0 or nb < 0 or o + nb > data_bytes for _, _, _, _, o, nb in plan):
            raise ValueError("tensor table outside the data section")
    except (ValueError, KeyError, TypeError, AttributeError, ImportError):
        return None                              # a corrupt / foreign header: rebuild (this used to raise on every run)
    try:
        fd = os.open(path, os.O_RDONLY)
    except OSError:
        return None
    try:
        hl = int.from_bytes(os.pread(fd, 16, 0)[8:], "little")
        data0 = (16 + hl + ALIGN - 1) // ALIGN * ALIGN
        if os.fstat(fd).st_size < data0 + data_bytes:
            return None
        tensors, jobs = {}, []
        for name, where, dt, shape, off, nb in plan:
            if where == "host":
                a = np.empty(shape, dt)
                if a.nbytes != nb:
                    return None
                if nb:
                    os.preadv(fd, [memoryview(a).cast("B")], data0 + off)
                tensors[name] = a
            elif where == "host_t":
                t = torch.empty(shape, dtype=dt)
                if t.numel() * t.element_size() != nb:
                    return None
                if nb:
                    os.preadv(fd, [memoryview(t.numpy()).cast("B")], data0 + off)
                tensors[name] = t
            else:
                if dt == torch.bool:
                    return None                  # (save() never writes one to the device section)
                t = torch.empty(shape, dtype=dt, device=dev)
                if t.numel() * t.element_size() != nb:
                    return None
                tensors[name] = t
                flat = t.reshape(-1).view(torch.uint8)
                for o in range(0, nb, CHUNK):
                    jobs.append((flat, o, min(CHUNK, nb - o), data0 + off + o))
        _stream_in(fd, jobs, dev, n_readers)
    finally:
        os.close(fd)
    try:
        return _restore(cls, attrs, tensors, dev)
    except (ValueError, KeyError, TypeError, AttributeError, ImportError):
        return None


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
