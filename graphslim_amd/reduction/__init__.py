"""Reducer factory (graphslim/reduction/registry.py:136-142) restricted to what the GCond path instantiates."""


def create_reducer(name, setting, data, args, **kwargs):
    key = str(name).lower()
    if key == "random":
        from ..sparsification.random import Random
        return Random(setting, data, args, **kwargs)
    if key == "gcond":
        from ..condensation.gcond import GCond
        return GCond(setting, data, args, **kwargs)
    raise NotImplementedError(f"graphslim_amd builds the GCond path only; reducer '{name}' is out of scope (SURVEY.md §8)")
