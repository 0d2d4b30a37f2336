"""Import environment for the Python reference (build container only).

TEST INFRASTRUCTURE.  Nothing under oracle/ is imported by the product
package; this module is used by the golden-vector generators and the
in-container differential fuzzer.  /root/reference does not exist on the
GPU box, so nothing here is reachable from `-m gpu` tests, smoke() or
bench.py.

What it does:
  * puts /root/reference first on sys.path (its test.py shadows stdlib test),
  * puts the local `colorama` stand-in on sys.path,
  * chdirs to /root/reference for Stormbound.__init__'s relative opens
    (games/stormbound.py:306-310),
  * requires PYTHONHASHSEED=0 (cards/s203.py:27 iterates a set of Points
    hashed by string, point.py:9-10).
"""
import os
import sys

REF = os.environ.get("MONSOON_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def setup():
    if not os.path.isdir(REF):
        raise RuntimeError(f"reference tree not found at {REF}")
    if os.environ.get("PYTHONHASHSEED") != "0":
        raise RuntimeError("run with PYTHONHASHSEED=0 (s203 set order)")
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)
    if HERE not in sys.path:
        sys.path.insert(1, HERE)
    os.chdir(REF)
