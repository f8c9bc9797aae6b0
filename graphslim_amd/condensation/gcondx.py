"""GCondX / DosCondX — the structure-free variants (graphslim/condensation/gcondx.py:18-79, doscondx.py:18-64): the
synthetic adjacency is the identity, only X~ is learned."""
from .gcond import _GradMatch


class GCondX(_GradMatch):
    identity_structure = True

    def rule(self, it, ol):
        # gcondx.py:53-56: `ol % 5 < 1` steps optimizer_pge, whose parameters receive no gradient without a PGE in the
        # graph -> nothing is updated on those iterations; the others step the features
        return False, not (ol % 5 < 1)


class DosCondX(_GradMatch):
    identity_structure = True
    inner_steps = False

    def rule(self, it, ol):
        return False, True               # doscondx.py:46-48
