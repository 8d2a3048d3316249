"""TEST INFRASTRUCTURE: stand-in for python/pylibcugraph/pylibcugraph/exceptions.py (one class, raised by pagerank.pyx:46,229)."""


class FailedToConvergeError(Exception):
    """An algorithm did not converge within its iteration limit."""
