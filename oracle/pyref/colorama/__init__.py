"""Minimal stand-in for the `colorama` package (absent from this image).

The reference engine imports Back/Fore/Style only to colour its __repr__
strings (board.py:5, unit.py:5, ...).  Every attribute resolves to "".
Test infrastructure only: used by oracle/pyref/harness.py in the build
container when importing /root/reference.
"""


class _Blank:
    def __getattr__(self, name):
        return ""


Back = _Blank()
Fore = _Blank()
Style = _Blank()
