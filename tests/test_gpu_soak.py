"""`-m gpu`: a few seconds of each differential soak (tools/soak_fuzz.py, tools/soak_batcher.py), run in-process, so that the
soaks stay runnable and a regression in the paths they cover shows up in the ordinary test run.  The long runs are in
profiles/r01_soak.txt."""
import os
import runpy
import sys

import pytest

pytestmark = pytest.mark.gpu
TOOLS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")


def run_tool(name, seconds, seed, capsys):
    argv = sys.argv
    sys.argv = [name, str(seconds), str(seed)]
    try:
        runpy.run_path(os.path.join(TOOLS, name), run_name="__main__")  # a mismatch ends in sys.exit(1)
    finally:
        sys.argv = argv
    assert " ok: " in capsys.readouterr().out


def test_a_few_seconds_of_the_batch_soak(capsys):
    # seeds from 1: blocking calls, a handle with repeated lengths (graph replay), pinned-arena and device-resident rounds
    run_tool("soak_fuzz.py", 10, 1, capsys)


def test_a_few_seconds_of_the_batcher_soak(capsys):
    run_tool("soak_batcher.py", 6, 1, capsys)


def test_a_few_seconds_of_the_private_handles_soak(capsys):
    # 1 .. 10 client threads with a private handle each, long buffers: hand-offs admitted and refused by the device's ledger in turn
    run_tool("soak_handles.py", 8, 1, capsys)
