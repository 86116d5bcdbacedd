"""bench.py's bookkeeping (no GPU): the roofline prices the SURVEY 8(d) record sizes whatever the build fetches, and a PMC summary
is quoted only for the library sources it was measured on."""
import json
import os

import bench


def counters(nodes=True):
    d = dict(n_samples=1000, n_closest=5000, n_shadow=4000, n_shaded=4500, n_node_visits=100000 if nodes else 0, n_tri_tests=30000 if nodes else 36 * 9000,
             node_bytes=80, tri_bytes=64)
    return d


def test_byte_model_uses_the_surveys_record_sizes():
    d = counters()
    base = 56 * 5000 + 292 * 4500 + 64 * 4000 + 156 * 1000
    assert bench.algorithmic_bytes(d) == base + 64 * 100000 + 48 * 30000          # NODE = 64, TRI = 48
    assert bench.algorithmic_bytes(d, fetched=True) == base + 80 * 100000 + 64 * 30000  # what this build's steps fetch: reported, never the score
    assert bench.algorithmic_bytes(counters(nodes=False)) == base                  # cbox: no BVH terms at all


def test_pmc_summary_is_quoted_only_for_the_sources_it_was_measured_on(tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.makedirs(tmp_path / "profiles")
    os.makedirs(tmp_path / "akari_render_amd" / "csrc" / "device")
    src = tmp_path / "akari_render_amd" / "csrc" / "device" / "k.h"
    src.write_text("// kernel v1\n")
    assert bench.measured_counters("c2") == (None, "no PMC summary profiles/r6_pmc_c2.json (tools/pmc_bench.sh c2)")
    h1 = bench.csrc_hash()
    json.dump({"csrc_hash": h1, "hbm_bytes_per_sample": 10.0, "valu_busy": 0.5}, open(tmp_path / "profiles" / "r4_pmc_c2.json", "w"))
    m, note = bench.measured_counters("c2")
    assert note is None and m["valu_busy"] == 0.5
    src.write_text("// kernel v2\n")                      # the kernel is edited: the summary describes another kernel now
    assert bench.csrc_hash() != h1
    m, note = bench.measured_counters("c2")
    assert m is None and "not quoted" in note and h1 in note
    d = counters(nodes=False)
    d.update(kernel_ms=100.0, n_launches=1)
    r = bench.roofline_block("c2", d)
    assert r["frac_measured"] is None and r["traffic"] is None and "not quoted" in r["measured_counters"]
    assert abs(r["frac"] - bench.algorithmic_bytes(d) / 0.1 / 1e9 / 8000.0) < 1e-12


def test_committed_pmc_summaries_carry_a_hash(root):
    for key in ("c2", "c3", "c4"):
        path = os.path.join(root, "profiles", f"r4_pmc_{key}.json")
        if os.path.exists(path):
            m = json.load(open(path))
            assert len(m.get("csrc_hash", "")) == 16 and "k_pt_pass" in m["kernel"]


def test_deadline_helper_reports_a_call_that_does_not_come_back():
    """bench.py puts akr_comm_create + the first akr_film_reduce (never run with more than one rank before a multi-GPU node exists)
    under a wall-clock deadline: a call that hangs is abandoned and reported, one that raises is reported, one that returns is used."""
    import time

    done, res = bench._with_deadline(lambda: 41 + 1, 5.0)
    assert done and res == 42
    done, res = bench._with_deadline(lambda: (_ for _ in ()).throw(RuntimeError("no communicator")), 5.0)
    assert done and isinstance(res, RuntimeError)
    t0 = time.time()
    done, res = bench._with_deadline(lambda: time.sleep(30), 0.3)
    assert not done and isinstance(res, TimeoutError) and time.time() - t0 < 5.0


def test_sample_split_needs_an_index_based_sampler():
    import subprocess
    import sys

    r = subprocess.run([sys.executable, os.path.join(bench.ROOT, "bench.py"), "--split", "samples"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode != 0 and "index-based sampler" in r.stderr


def test_native_communicator_fallbacks(monkeypatch):
    """make_native_comm with the library and torch.distributed faked: a communicator that works is used on the context it was made
    on; one whose creation raises, whose first reduce never returns, or whose unique id cannot be made gives (None, a FRESH context,
    the reason) on every rank -- the run goes on with torch.distributed.reduce instead of waiting on a stuck stream."""
    import time
    import types

    import torch as real_torch

    from akari_render_amd import capi

    made = []

    class FakeCtx:
        def __init__(self, dev):
            made.append(dev)

    class FakeFilm:
        def __init__(self, *a, **k):
            pass

    behaviour = {"mode": "ok"}

    class FakeComm:
        def __init__(self, ctx, uid, rank, world):
            if behaviour["mode"] == "raise":
                raise capi.AkariError(-6, "ncclCommInitRank: unhandled system error")
            self.closed = False

        def reduce_film(self, film, root=0, blocking=True, planes=7):
            if behaviour["mode"] == "hang":
                time.sleep(60)

        def close(self):
            self.closed = True

    def uid():
        if behaviour["mode"] == "no_rccl":
            raise capi.AkariError(-6, "librccl.so not found")
        return b"\\0" * 128

    monkeypatch.setattr(capi, "Context", FakeCtx)
    monkeypatch.setattr(capi, "Film", FakeFilm)
    monkeypatch.setattr(capi, "Comm", FakeComm)
    monkeypatch.setattr(capi, "comm_unique_id", uid)
    fake_torch = types.SimpleNamespace(zeros=lambda *a, **k: types.SimpleNamespace(data_ptr=lambda: 0), float32=real_torch.float32, int32=real_torch.int32,
                                       tensor=lambda v, dtype=None, device=None: real_torch.tensor(v, dtype=dtype), cuda=types.SimpleNamespace(synchronize=lambda d: None))
    fake_dist = types.SimpleNamespace(broadcast_object_list=lambda box, src=0: None, all_reduce=lambda t, op=None: None, ReduceOp=types.SimpleNamespace(MIN=0))
    ctx0 = object()
    comm, ctx, note = bench.make_native_comm(ctx0, 0, 2, fake_torch, fake_dist, "cpu", 3, 5.0)
    assert isinstance(comm, FakeComm) and ctx is ctx0 and note is None and made == []
    for mode, expect in (("raise", "ncclCommInitRank"), ("hang", "no answer after"), ("no_rccl", "librccl.so not found")):
        behaviour["mode"] = mode
        before = len(made)
        comm, ctx, note = bench.make_native_comm(ctx0, 0, 2, fake_torch, fake_dist, "cpu", 3, 0.3)
        assert comm is None and expect in note, (mode, note)
        if mode != "no_rccl":  # a possibly stuck stream is abandoned with its context
            assert isinstance(ctx, FakeCtx) and made[before:] == [3]
        else:
            assert ctx is ctx0
