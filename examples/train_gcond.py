"""Condense a graph with GCond and evaluate the result with a GCN — same five calls as GraphSlim's
examples/train_gcond.py (which also runs unchanged against this package through the `graphslim` alias):

    python examples/train_gcond.py -D cora -M gcond -R 0.5 -G 0
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from graphslim.config import cli
from graphslim.condensation import GCond
from graphslim.dataset import get_dataset
from graphslim.evaluation.eval_agent import Evaluator

if __name__ == "__main__":
    args = cli(standalone_mode=False)
    graph = get_dataset(args.dataset, args)
    agent = GCond(setting=args.setting, data=graph, args=args)
    reduced_graph = agent.reduce(graph, verbose=args.verbose)
    Evaluator(args).evaluate(reduced_graph, model_type="GCN")
