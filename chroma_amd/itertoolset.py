"""The two itertools helpers the propagate path uses (reference: chroma/itertoolset.py)."""
from itertools import chain, islice, repeat  # noqa: F401  (re-exported)


def peek(iterable):
    """Return (first element, iterator that still yields every element)."""
    it = iter(iterable)
    first = next(it)
    return first, chain([first], it)


def repeating_iterator(obj, nreps):
    for _ in range(nreps):
        yield obj
