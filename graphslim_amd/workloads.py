"""Benchmark / test workloads of the GCond path: dataset-shaped synthetic problems at the configurations BASELINE.json names,
engine construction for them and the job's own step cadence.  Used by bench.py, the multi-GPU driver (parallel.make_engine),
tools/ and the full-size parity tests — host-side plumbing only, every number is computed by libgslim_hip.so."""
import numpy as np
import torch

WORKLOADS = {
    # name: (dataset shape, rate, ntrans, hidden, pge_hidden, outer_loop, inner_loop, lr_feat, lr_adj, fan-outs)
    "ogbn-arxiv": ("ogbn-arxiv", 0.005, 2, 256, 256, 20, 3, 0.01, 0.01, (10, 5)),
    "cora": ("cora", 0.5, 1, 256, 128, 20, 15, 1e-4, 1e-4, (10, 5)),
    # the ogbn-arxiv configuration at r = 1 % (n' = 909: 4x the PGE's pair rows): where, if anywhere, sharding pays
    "ogbn-arxiv-r0.01": ("ogbn-arxiv", 0.01, 2, 256, 256, 20, 3, 0.01, 0.01, (10, 5)),
    # BASELINE.json configs[4]: inductive, train sub-graph, wide rows (d = 602) and fan-outs [15, 8]: the gather stress case
    "reddit": ("reddit", 0.001, 1, 256, 256, 10, 1, 0.1, 0.1, (15, 8)),
    # the other BASELINE.json configs (parity-test cases at FULL size, tests/test_gpu_fullsize.py; not bench lines)
    "cora-r0.25": ("cora", 0.25, 1, 256, 128, 20, 15, 1e-4, 1e-4, (10, 5)),
    # n' = 140 with the Planetoid PGE width 128: the generated / fused PGE forms at N = 128 (4 MFMA column tiles per wave)
    "cora-r1.0": ("cora", 1.0, 1, 256, 128, 20, 15, 1e-4, 1e-4, (10, 5)),
    "citeseer-r0.005": ("citeseer", 0.005, 1, 256, 128, 20, 15, 1e-4, 1e-4, (10, 5)),
    "pubmed-r0.005": ("pubmed", 0.005, 1, 256, 128, 20, 15, 1e-4, 1e-4, (10, 5)),
    # graphslim/configs/gcond/flickr.json at r = 0.5 %: inductive + two transforms + fan-out [15, 8] (loader.py:197-210), d = 500 (not a
    # multiple of 32: H1 = relu(X W1 + b1) runs on the general tile), C = 7, n' = 223 — full-size parity case
    "flickr": ("flickr", 0.005, 2, 256, 256, 10, 1, 0.005, 0.005, (15, 8)),
    # job-level accuracy gate (tests/golden/make_job_accuracy.py): the ogbn-arxiv hyper-parameters at graph scale 0.05 with a rate
    # that keeps n' >= 128 (n' = 136): a whole job the CPU oracle finishes in minutes, through the headline shape's kernels
    "arxiv-gate": ("ogbn-arxiv", 0.03, 2, 256, 256, 10, 3, 0.01, 0.01, (10, 5)),
    # the same hyper-parameters on the balanced 8-class variant of the shape (dataset/synthetic.py "arxiv-c8"): the seed-quiet whole job
    # through the headline shape's kernels (n' = 136 >= 128, d = 128, h = P_h = 256, two transforms) — VERDICT r05 "next" 4a
    "arxiv-gate-c8": ("arxiv-c8", 0.03, 2, 256, 256, 10, 3, 0.01, 0.01, (10, 5)),
    # `--nlayers 3` (dataset/loader.py:205-206: sizes [15, 10, 5]; models/sgc.py:45-51: three propagations on both sides) with the
    # ogbn-arxiv hyper-parameters: a parity-test case (tests/test_gpu_fullsize.py), not a bench line
    "arxiv-k3": ("ogbn-arxiv", 0.005, 2, 256, 256, 20, 3, 0.01, 0.01, (15, 10, 5)),
    "cora-k3": ("cora", 0.5, 1, 256, 128, 20, 15, 1e-4, 1e-4, (15, 10, 5)),
    # `--ntrans 3` / 4 (models/sgc.py:32-35: hidden-to-hidden transforms between the first and the last) with the ogbn-arxiv
    # hyper-parameters: parity-test cases (tests/test_gpu_fullsize.py), not bench lines
    "arxiv-t3": ("ogbn-arxiv", 0.005, 3, 256, 256, 20, 3, 0.01, 0.01, (10, 5)),
    "arxiv-t4": ("ogbn-arxiv", 0.005, 4, 256, 256, 20, 3, 0.01, 0.01, (10, 5)),
    # `--condense_model GCN --nlayers 3` (models/gcn.py:8-23: one GraphConvolution per sampled hop) with the ogbn-arxiv hyper-parameters:
    # a parity-test case (the third field is the model's depth)
    "arxiv-gcn3": ("ogbn-arxiv", 0.005, 3, 256, 256, 20, 3, 0.01, 0.01, (15, 10, 5)),
    "arxiv-gcn2": ("ogbn-arxiv", 0.005, 2, 256, 256, 20, 3, 0.01, 0.01, (10, 5)),      # (two layers: the dropout parity case)
    "cora-r0.25-k3": ("cora", 0.25, 1, 256, 128, 20, 15, 1e-4, 1e-4, (15, 10, 5)),      # the quiet job of the job-level gate with three hops
}


def build_problem(name, seed=1, scale=1.0, sep=1.0):
    from graphslim_amd.dataset.loader import TransAndInd, splits
    from graphslim_amd.dataset.synthetic import make_raw
    from graphslim_amd.host import generate_labels_syn, normalize_adj_sparse
    shape, rate, T, hidden, ph, outer, inner, lrf, lra, fan = WORKLOADS[name]
    raw = splits(make_raw(shape, seed=seed, scale=scale, sep=sep))
    data = TransAndInd(raw, "ogbn-arxiv" if shape == "arxiv-c8" else shape)      # (the preprocessing of loader.py:110-121 goes by the dataset name)
    labels_train = data.labels_train.numpy()
    labels_syn, ncd, ranges = generate_labels_syn(labels_train, rate)
    setting = "ind" if shape in ("reddit", "flickr") else "trans"
    if setting == "ind":      # gcond.py:25-27: the inductive setting condenses the TRAIN sub-graph
        csr = normalize_adj_sparse(data.adj_train)
        data.bench_feat, data.bench_labels, data.bench_n = data.feat_train, data.labels_train, data.feat_train.shape[0]
    else:
        csr = normalize_adj_sparse(data.adj_full)
        data.bench_feat, data.bench_labels, data.bench_n = data.feat_full, data.labels_full, data.num_nodes
    pools = data.class_pools(setting)
    return dict(data=data, labels_syn=labels_syn, ncd=ncd, ranges=[tuple(ranges[c]) for c in range(data.nclass)], csr=csr,
                pools=pools, T=T, hidden=hidden, ph=ph, outer=outer, inner=inner, lrf=lrf, lra=lra, fan=fan, rate=rate, name=name,
                model="GCN" if "-gcn" in name else "SGC")


def make_engine(p, device, bcap=256):
    from graphslim_amd.engine import EngineShapes, GCondEngine
    d = p["data"].feat_full.shape[1]
    gcn = p.get("model", "SGC") == "GCN"
    sh = EngineShapes(p["data"].bench_n, d, p["data"].nclass, len(p["labels_syn"]), 2 if gcn else p["T"], p["hidden"], p["ph"], bcap,
                      p["fan"][0], p["fan"][-1], fmid=tuple(p["fan"][1:-1]), dropout=float(p.get("dropout", 0.0)),
                      model=p.get("model", "SGC"))
    cls_ptr = np.concatenate([[0], np.cumsum([len(q) for q in p["pools"]])])
    eng = GCondEngine(sh, p["data"].bench_feat, p["csr"], cls_ptr, np.concatenate(p["pools"]), p["labels_syn"], p["ranges"],
                      lr_feat=p["lrf"], lr_adj=p["lra"], lr_model=0.01, inner_loop=p["inner"], seed=1, device=device)
    return eng


def init_state(p, eng, seed=1):
    """X~ = features of randomly picked train nodes per class (sparsification/random.py:9-17); PGE as nn.Linear/BN default."""
    from graphslim_amd.host import init_pge_params, random_init_indices
    np.random.seed(seed)
    torch.manual_seed(seed)
    ids = p["data"].idx_train.numpy() if p["data"].bench_n == p["data"].num_nodes else np.arange(p["data"].bench_n)
    sel = random_init_indices(ids, p["data"].labels_train.numpy(), p["ncd"])
    eng.set_feat(p["data"].bench_feat[sel])
    eng.set_pge(init_pge_params(p["data"].feat_full.shape[1], p["ph"]))


def new_model(p, eng):
    from graphslim_amd.host import init_sgc_params
    d = p["data"].feat_full.shape[1]
    theta = init_sgc_params(d, p["hidden"], p["data"].nclass, p["T"])
    eng.set_model(theta)
    return theta


def run_steps(p, eng, k, cycle_len):
    """k outer iterations with the job's own mix and cadence: the first 20 % are PGE-update iterations (epoch % 50 < 10,
    gcond.py:58-61), the rest feature-update iterations, and theta is re-drawn and uploaded every `outer_loop` iterations —
    an epoch of the real job is `outer_loop` consecutive iterations on one theta (gcond.py:42-47)."""
    for i in range(k):
        if i % p["outer"] == 0:
            new_model(p, eng)
        eng.outer_step(0 if i * 5 < cycle_len else 10)


def blocks_to_batches(blk):
    """The engine's sampled blocks (BlockSet) as reference-style batches (batch_size, n_id, [outermost block, .., innermost block]) with
    CSR blocks (rowptr, col, val, shape) — what oracle.GCondState.outer_iteration takes.  The outermost hop's sources are global ids in
    the engine (not de-duplicated): they are relabelled here the way PyG does it, targets first, then new nodes in first-seen order."""
    f = {k: getattr(blk, k).cpu().numpy() for k in blk.FIELDS}
    nm = len(blk.fmid)
    out = []
    for c in range(blk.C):
        B, n1 = int(f["n_seed"][c]), int(f["n1"][c])
        v1, c1 = f["val1"][c, :B].reshape(-1), f["col1"][c, :B].reshape(-1)
        k1 = v1 != 0
        rows1 = np.repeat(np.arange(B), blk.f1)[k1]
        rp1 = np.concatenate([[0], np.cumsum(np.bincount(rows1, minlength=B))])
        blocks = [(rp1, c1[k1].astype(np.int64), v1[k1].astype(np.float32), (B, n1))]
        rows, nid_last = n1, f["nid1"][c, :n1]
        for m in range(nm):      # middle hop m: frontier m - 1 (rows) <- frontier m
            ncur = int(f[f"nm{m}"][c])
            vm, cm = f[f"valm{m}"][c, :rows].reshape(-1), f[f"colm{m}"][c, :rows].reshape(-1)
            km = vm != 0
            rr = np.repeat(np.arange(rows), blk.fmid[m])[km]
            rpm = np.concatenate([[0], np.cumsum(np.bincount(rr, minlength=rows))])
            blocks.insert(0, (rpm, cm[km].astype(np.int64), vm[km].astype(np.float32), (rows, ncur)))
            rows, nid_last = ncur, f[f"nidm{m}"][c, :ncur]
        v0, c0g = f["val0"][c, :rows].reshape(-1), f["col0"][c, :rows].reshape(-1)
        keep = v0 != 0
        rows0 = np.repeat(np.arange(rows), blk.f2)[keep]
        uniq = np.unique(np.concatenate([nid_last, c0g[keep]]))
        # targets first, then the new sources in first-seen order
        pos = -np.ones(int(uniq.max()) + 1, dtype=np.int64)
        pos[nid_last] = np.arange(rows)
        rest = c0g[keep][pos[c0g[keep]] < 0]
        _, first = np.unique(rest, return_index=True)
        new = rest[np.sort(first)]
        pos[new] = rows + np.arange(len(new))
        n_id = np.concatenate([nid_last, new])
        rp0 = np.concatenate([[0], np.cumsum(np.bincount(rows0, minlength=rows))])
        blocks.insert(0, (rp0, pos[c0g[keep]], v0[keep].astype(np.float32), (rows, len(n_id))))
        out.append((B, n_id, blocks))
    return out
