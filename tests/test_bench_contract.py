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
    assert bench.measured_counters("c2") == (None, "no PMC summary profiles/r4_pmc_c2.json")
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
