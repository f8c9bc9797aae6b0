"""DosCond — "Condensing Graphs via One-Step Gradient Matching" (graphslim/condensation/doscond.py:18-67): the GCond
matching step without inner model training; both optimisers step every outer iteration."""
from .gcond import _GradMatch


class DosCond(_GradMatch):
    inner_steps = False
    inference_each_step = False          # doscond.py:53: PGE.inference only at checkpoints

    def rule(self, it, ol):
        return True, True                # doscond.py:48-49
