"""Prepared-database cache (qpgesture_amd/db_cache.py): a GestureDB restored from its cache file is the object its
constructor built - every tensor bit for bit - and the drop-in CLI writes the same bytes from it."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _same(a, b, path=""):
    import torch
    assert type(a) is type(b), path
    if isinstance(a, torch.Tensor):
        assert a.dtype == b.dtype and a.shape == b.shape and a.device == b.device and torch.equal(a, b), path
    elif isinstance(a, np.ndarray):
        assert a.dtype == b.dtype and np.array_equal(a, b), path
    elif hasattr(a, "__dict__") and type(a).__module__.startswith("qpgesture_amd"):
        assert set(a.__dict__) == set(b.__dict__), path
        for k in a.__dict__:
            _same(a.__dict__[k], b.__dict__[k], path + "." + k)
    elif isinstance(a, dict):
        assert set(a) == set(b), path
        for k in a:
            _same(a[k], b[k], path + "[%s]" % k)
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b), path
        for i, (x, y) in enumerate(zip(a, b)):
            _same(x, y, path + "[%d]" % i)
    elif isinstance(a, float):
        assert a == b or (a != a and b != b), path
    else:
        assert a == b, path


@pytest.mark.parametrize("feature_dtype,wavvq", [("f32", False), ("f16", False), ("f32", True)])
def test_restored_db_is_the_built_one(tmp_path, feature_dtype, wavvq):
    import torch
    from qpgesture_amd import synth
    from qpgesture_amd.code_knn import CodeKNN, GestureDB
    from qpgesture_amd.data_processing import interp_wavlm
    N, M = 96, 2
    tr, te = synth.make_db(N, 40), synth.make_db(M, 41)
    code, sig = synth.make_codes(N, 42), synth.make_signature(43)
    db = GestureDB(code, interp_wavlm(tr["wavlm"]), tr["context"].squeeze(2), tr["phase_dense"], sig, device="cuda:0",
                   feature_dtype=feature_dtype, wavvq=tr["wavvq"] if wavvq else None)
    p = str(tmp_path / "db.qpgdb")
    db.save(p, "k1")
    assert GestureDB.load(p, "cuda:0", "other-key") is None                  # keyed differently: not this database
    assert GestureDB.load(str(tmp_path / "missing.qpgdb"), "cuda:0") is None
    db2 = GestureDB.load(p, "cuda:0", "k1")
    assert db2 is not None
    _same(db, db2, "db")
    assert db2.txt_cidx is db2.txt_r                                          # aliases stay aliases
    ti = torch.from_numpy(interp_wavlm(te["wavlm"])).cuda()
    tc = torch.from_numpy(np.ascontiguousarray(te["context"].squeeze(2))).cuda()
    a = CodeKNN(db, rng=np.random.RandomState(5)).match_clip(ti, tc, M)
    b = CodeKNN(db2, rng=np.random.RandomState(5)).match_clip(ti, tc, M)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    # a truncated file is refused, not half-loaded
    with open(p, "r+b") as f:
        f.truncate(os.path.getsize(p) - 4096)
    assert GestureDB.load(p, "cuda:0", "k1") is None


def test_cli_second_invocation_takes_the_cache_and_writes_the_same_bytes(tmp_path, capsys):
    """GestureKNN.py's command line twice on the same files: the second run restores the prepared database (no .npz of the
    database side is opened), `knn_pred` is byte-identical; touching a database file invalidates the key; --db_cache off
    never writes; a text track with exact ties still goes through the reference's NumPy ranks."""
    from qpgesture_amd import GestureKNN as cli
    from qpgesture_amd import synth
    d = str(tmp_path / "npz")
    paths = synth.write_npz_set(d, 48, 2, variant="texttie")
    cdir = str(tmp_path / "cache")

    def run(extra=()):
        outp = str(tmp_path / "out.npz")
        argv = []
        for k, v in paths.items():
            argv += ["--" + k, v]
        cli.main(argv + ["--out_knn_filename", outp, "--db_cache_dir", cdir] + list(extra))
        return np.load(outp)["knn_pred"], capsys.readouterr().out
    off, _ = run(["--db_cache", "off"])
    assert not os.path.exists(cdir) or not os.listdir(cdir)
    first, o1 = run()
    assert "prepared-database cache)" not in o1 and len(os.listdir(cdir)) == 1
    second, o2 = run()
    assert "prepared-database cache)" in o2
    assert first.dtype == np.int64 and np.array_equal(first, second) and np.array_equal(first, off)
    stable, _ = run(["--tie_rule", "stable"])                                # another key (options are part of it)
    assert len(os.listdir(cdir)) == 2
    before = set(os.listdir(cdir))
    os.utime(paths["train_codebook"], ns=(1, 1))                             # the database changed on disk
    third, o3 = run()
    # a new key - and the cache file of the SAME sources under the old key is evicted (round 6: db_cache.prune), the other
    # options' file stays
    after = set(os.listdir(cdir))
    assert "prepared-database cache)" not in o3 and np.array_equal(third, first)
    assert len(after) == 2 and len(after & before) == 1 and not [n for n in after if ".tmp." in n]
