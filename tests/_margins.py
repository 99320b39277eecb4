"""Margin recorder of the GPU parity tests (round-4 review item 1b).

`within(name, value, bar)` asserts value <= bar and remembers (test id, name, value, bar); the session summary (tests/conftest.py)
lists every check that used more than half of its bar and, when GC_TEST_MARGINS=<path> is set, writes all of them as JSON lines so the
same suite run on several boxes can be compared (scripts/gpu_suite_repeat.sh -> profiles/r04_gpu_suite_margins.txt)."""
import os

RECORDS = []


def _test_id():
    return os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]


def within(name, value, bar, strict=False):
    value = float(value); bar = float(bar)
    RECORDS.append((_test_id(), name, value, bar))
    ok = value < bar if strict else value <= bar
    assert ok, f"{name}: {value:.6g} exceeds bar {bar:.6g} (x{value / bar if bar else float('inf'):.3f})"
    return value
