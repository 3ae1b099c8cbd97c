"""bench.py at N > 1 prints ONE JSON line whatever the second (library-choice) run does: headline_then_guarded is exercised here
with fake runs, in a child process each (its failure paths leave through os._exit).  No GPU, no torch."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, time
sys.path.insert(0, {root!r})
import bench
mode, rank = sys.argv[1], int(sys.argv[2])
head = {{"value": 1000.0, "roofline": {{"frac": 0.1}}, "config": {{"parallelism": "tp2"}}}} if rank == 0 else None
lib = {{"value": 2000.0, "ms_per_step": 0.5, "transport": "p2p", "rccl_nranks": 0, "parity": {{"tokens_equal": True}},
       "config": {{"parallelism": "cls2"}}, "roofline": {{"frac": 0.2, "redundant_bytes_per_step": 1, "bytes_streamed_per_gpu_per_step": 2, "step_split": None}}}}
def second():
    if mode == "raise": raise RuntimeError("peer-to-peer handshake: no flag from a peer")
    if mode == "hang": time.sleep(60)
    if mode == "malformed": return {{"value": 5.0}} if rank == 0 else None      # a run that returns without `config` / `roofline`
    return lib if rank == 0 else None
def third():
    if mode == "raise3": raise RuntimeError("split-out run failed")
    return dict(lib, value=3000.0, config={{"parallelism": "tp2 split-out"}}) if rank == 0 else None
runs = second if mode in ("ok", "raise", "hang", "malformed") else [("library_choice", second), ("tp_split_out", third)]
out = bench.headline_then_guarded(lambda: head, runs, lambda: None, rank, 1.0)
print("RETURNED", flush=True)
"""


def run(mode, rank):
    p = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT), mode, str(rank)], capture_output=True, text=True, timeout=60)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return p, lines


def test_second_run_succeeds_one_line_with_both():
    p, lines = run("ok", 0)
    assert p.returncode == 0 and len(lines) == 1 and "RETURNED" in p.stdout, p.stderr
    d = json.loads(lines[0])
    assert d["value"] == 1000.0 and d["library_choice"]["value"] == 2000.0 and d["library_choice"]["parallelism"] == "cls2"
    assert d["library_choice"]["roofline"]["frac"] == 0.2


def test_second_run_raises_headline_still_printed():
    p, lines = run("raise", 0)
    assert p.returncode == 0 and len(lines) == 1 and "RETURNED" not in p.stdout, (p.stdout, p.stderr)
    d = json.loads(lines[0])
    assert d["value"] == 1000.0 and d["library_choice"]["value"] is None and "RuntimeError" in d["library_choice"]["skipped"]
    assert "library-choice run failed" in p.stderr


def test_second_run_hangs_watchdog_prints_headline():
    p, lines = run("hang", 0)
    assert p.returncode == 0 and len(lines) == 1 and "RETURNED" not in p.stdout, (p.stdout, p.stderr)
    assert "did not finish" in json.loads(lines[0])["library_choice"]["skipped"]


def test_other_ranks_print_nothing_and_leave():
    for mode in ("ok", "raise", "hang"):
        p, lines = run(mode, 1)
        assert p.returncode == 0 and not lines, (mode, p.stdout, p.stderr)


def test_two_further_runs_both_ride_along():
    p, lines = run("ok3", 0)
    assert p.returncode == 0 and len(lines) == 1 and "RETURNED" in p.stdout, p.stderr
    d = json.loads(lines[0])
    assert d["library_choice"]["value"] == 2000.0 and d["tp_split_out"]["value"] == 3000.0 and d["tp_split_out"]["parallelism"] == "tp2 split-out"


def test_third_run_raises_the_first_two_are_printed():
    p, lines = run("raise3", 0)
    assert p.returncode == 0 and len(lines) == 1 and "RETURNED" not in p.stdout, (p.stdout, p.stderr)
    d = json.loads(lines[0])
    assert d["value"] == 1000.0 and d["library_choice"]["value"] == 2000.0 and d["tp_split_out"]["value"] is None and "RuntimeError" in d["tp_split_out"]["skipped"]


def test_second_run_returns_without_its_figures_headline_still_printed():
    """A run that comes back without `config` / `roofline` raises inside the guard, not after it: the headline line still gets out, once."""
    p, lines = run("malformed", 0)
    assert p.returncode == 0 and len(lines) == 1 and "RETURNED" not in p.stdout, (p.stdout, p.stderr)
    d = json.loads(lines[0])
    assert d["value"] == 1000.0 and d["library_choice"]["value"] is None and "KeyError" in d["library_choice"]["skipped"]
