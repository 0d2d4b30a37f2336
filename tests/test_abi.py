"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/monsoon.h declares, the ctypes binding covers them, and the product fails loudly (no
CPU fallback) when there is no GPU.  No compute calls."""
import os
import re

import pytest

from conftest import REPO


def header_functions():
    text = open(os.path.join(REPO, "include", "monsoon.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(monsoon_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from monsoon_amd import _lib
    lib = _lib.load()
    names = header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/monsoon.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)


def test_card_table_matches_json():
    from monsoon_amd import _lib
    from monsoon_amd.cards import CARD_IDS, UNSUPPORTED
    lib = _lib.load()
    assert len(CARD_IDS) == 112 and CARD_IDS == sorted(CARD_IDS)
    for i, cid in enumerate(CARD_IDS):
        assert lib.monsoon_card_index(cid.encode()) == i
        assert bool(lib.monsoon_card_supported(i)) == (cid not in UNSUPPORTED)
    assert lib.monsoon_card_index(b"zzzz") == -1


def test_no_cpu_fallback_without_gpu():
    """On a box without a GPU the product must refuse to run, not compute on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from monsoon_amd import MonsoonError
    from monsoon_amd.engine import BatchEngine
    with pytest.raises(MonsoonError, match="no usable HIP device"):
        BatchEngine(4)


def test_product_does_not_import_oracle():
    """Nothing under monsoon_amd/ (Python or C++) may reference oracle/."""
    bad = []
    for root, _, files in os.walk(os.path.join(REPO, "monsoon_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".inc")):
                src = open(os.path.join(root, f), errors="ignore").read()
                for line in src.splitlines():
                    code = line.split("#")[0] if f.endswith(".py") else line.split("//")[0]
                    if re.search(r"(import\s+oracle|from\s+oracle|oracle_lib|liboracle|#include\s+\".*oracle)", code):
                        bad.append((f, line.strip()))
    assert not bad, bad
