"""Multi-GPU (one process per GPU, RCCL) — filled in below once the single-GPU profile is known."""


def make_engine(p, dev, rank, world):
    raise NotImplementedError("multi-GPU engine not built yet")
