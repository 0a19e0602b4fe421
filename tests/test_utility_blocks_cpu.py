"""The host-only drop-in blocks (wrap, unwrap, inspect, throttle: SURVEY.md 8f ranks 1 and 4)
against the bytes and messages the reference's own binaries produced for the same inputs
(tests/golden/utility.npz / utility.json, made by tests/golden/make_golden.py).  No GPU work
is involved, so these run in the CPU suite."""
import json
import os
import subprocess
import time

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "blocks", "bin")

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(BIN, "wrap")),
                                reason="blocks not built (run __graft_entry__.build())")


@pytest.fixture(scope="module")
def gold(golden_dir):
    with open(os.path.join(golden_dir, "utility.json")) as f:
        txt = json.load(f)
    return np.load(os.path.join(golden_dir, "utility.npz")), txt


def run(name, args, data=b"", seekable=False, tmp=None):
    exe = os.path.join(BIN, name)
    if seekable:
        path = os.path.join(tmp, "stdin.bin")
        with open(path, "wb") as f:
            f.write(data)
        with open(path, "rb") as f:
            p = subprocess.run([exe] + args, stdin=f, capture_output=True, timeout=60)
    else:
        p = subprocess.run([exe] + args, input=data, capture_output=True, timeout=60)
    return p.stdout, p.stderr.decode(), p.returncode


@pytest.mark.parametrize("case", ["wrap_file_count", "wrap_file_exact", "wrap_file_optimal", "wrap_file_duration",
                                  "wrap_file_norate"])
def test_wrap_seekable_input_byte_exact(gold, tmp_path, case):
    g, txt = gold
    t = txt[case]
    out, err, rc = run("wrap", t["argv"], bytes(g["raw"][: t["input_bytes"]]), True, str(tmp_path))
    assert rc == t["rc"] == 0 and err == t["stderr"]
    assert out == bytes(g[case])


@pytest.mark.parametrize("case", ["wrap_pipe_count", "wrap_pipe_exact"])
def test_wrap_pipe_input(gold, case):
    """On a pipe the reference cannot see the end coming: its last header announces a full
    packet in front of the bytes that were left (none, when the input ended on a packet
    boundary).  The drop-in's last header carries the true size; everything before is
    byte-exact."""
    g, txt = gold
    t = txt[case]
    data = bytes(g["raw"][: t["input_bytes"]])
    out, err, rc = run("wrap", t["argv"], data)
    assert rc == 0 and err == t["stderr"]
    ref = O.parse_packets(bytes(g[case]))
    got = O.parse_packets(out)
    full = [p for p in ref if len(p[4]) == p[2]]
    assert got[: len(full)] == full
    tail_ref = ref[len(full):]                       # the reference's over-announced last packet
    assert len(tail_ref) == 1 and len(tail_ref[0][4]) < tail_ref[0][2]
    if tail_ref[0][4]:
        assert len(got) == len(full) + 1
        last = got[-1]
        assert last[:2] == tail_ref[0][:2] and last[3] == tail_ref[0][3]
        assert last[2] == len(last[4]) == len(tail_ref[0][4]) and last[4] == tail_ref[0][4]
    else:
        assert len(got) == len(full)                 # no bare header after the end of the input
    assert b"".join(p[4] for p in got) == data


@pytest.mark.parametrize("case", ["unwrap_all", "unwrap_1", "unwrap_7"])
def test_unwrap(gold, case):
    g, txt = gold
    out, err, rc = run("unwrap", txt[case]["argv"][1:], bytes(g["pstream"]))
    assert (rc, err) == (0, txt[case]["stderr"]) and out == bytes(g[case])


@pytest.mark.parametrize("case", ["inspect_all", "inspect_1", "inspect_1_pass", "inspect_1_pass_all", "inspect_all_pass_all"])
def test_inspect(gold, case):
    g, txt = gold
    out, err, rc = run("inspect", txt[case]["argv"][1:], bytes(g["pstream"]))
    assert rc == 0 and err == txt[case]["stderr"]
    assert out == bytes(g[case])


@pytest.mark.parametrize("case", ["throttle_all", "throttle_1"])
def test_throttle_forwards_everything(gold, case):
    g, txt = gold
    out, err, rc = run("throttle", txt[case]["argv"][1:], bytes(g["pstream"]))
    assert (rc, err) == (0, "") and out == bytes(g[case]) == bytes(g["pstream"])


def test_throttle_paces_by_duration():
    """blocks/throttle.cpp:42-49: packet k+1 leaves no earlier than the sum of the durations
    of the paced packets before it; a foreign stream and zero durations are not paced."""
    pk = b"".join(O.pack_packet(pid, 6, b"\0" * 8, d) for pid, d in
                  [(1, 60_000_000), (2, 500_000_000), (1, 60_000_000), (1, 0), (1, 60_000_000)])
    t0 = time.perf_counter()
    out, err, rc = run("throttle", ["1"], pk)
    dt = time.perf_counter() - t0
    assert rc == 0 and out == pk
    assert 0.18 <= dt < 0.45          # 3 x 60 ms; the 500 ms packet of stream 2 does not count


@pytest.mark.parametrize("case", ["wrap_help", "unwrap_help", "inspect_help", "throttle_help", "wrap_missing",
                                  "wrap_zero_size", "wrap_zero_count", "wrap_short_duration", "wrap_mismatch",
                                  "wrap_badid", "unwrap_badid"])
def test_usage_and_error_text(gold, case):
    _, txt = gold
    t = txt[case]
    name = t["argv"][0]
    p = subprocess.run(["./" + name] + t["argv"][1:], cwd=BIN, input=b"", capture_output=True, timeout=60)
    assert p.stderr.decode() == t["stderr"]
    assert p.returncode == t["rc"] and len(p.stdout) == t["stdout_bytes"]
