"""Reducer factory (graphslim/reduction/registry.py:136-142) restricted to what the GCond path instantiates."""


def create_reducer(name, setting, data, args, **kwargs):
    key = str(name).lower()
    if key == "random":
        from ..sparsification.random import Random
        return Random(setting, data, args, **kwargs)
    if key in ("kcenter", "herding"):
        from ..sparsification.model_based import Herding, KCenter
        return (KCenter if key == "kcenter" else Herding)(setting, data, args, **kwargs)
    if key == "averaging":
        from ..coarsening.averaging import Average
        return Average(setting, data, args, **kwargs)
    if key == "clustering":
        from ..coarsening.clustering import Cluster
        return Cluster(setting, data, args, **kwargs)
    if key in ("gcond", "doscond", "gcondx", "doscondx", "msgc", "sgdd"):
        from .. import condensation as cd
        cls = {"gcond": cd.GCond, "doscond": cd.DosCond, "gcondx": cd.GCondX, "doscondx": cd.DosCondX, "msgc": cd.MSGC, "sgdd": cd.SGDD}[key]
        return cls(setting, data, args, **kwargs)
    raise NotImplementedError(f"graphslim_amd builds the GCond path only; reducer '{name}' is out of scope (SURVEY.md §8)")
